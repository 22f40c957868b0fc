// tests/emu/trace_hooks.cpp — TEST / ANALYSIS INFRASTRUCTURE ONLY (never in libmgx.so).
// Memory-access tracer for the host model of the wave programs: the traced build of the model
// (libmgxemu_trace_w8.so, `make trace`) is compiled with -fsanitize=kernel-address and outlined
// instrumentation, which turns EVERY load and store of the wave program into a call of
// __asan_loadN / __asan_storeN.  This file implements those hooks: accesses that fall into a
// registered region (the arena arrays of carve(), the graph tables, the batch streams) are tallied
// as 64-byte lines per region, per read — "distinct lines touched while one read is processed", the
// traffic a cache that holds one read's working set would see.  A CPU-side model of the fabric
// traffic the rocprofv3 PMC passes measure on the GPU (tools/traffic_model.py).
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
struct Region { uint64_t lo, hi; std::string name; uint64_t rd_lines = 0, wr_lines = 0, rd_acc = 0, wr_acc = 0; };
std::vector<Region> g_regions;               // sorted by lo once tracing starts
std::unordered_map<uint64_t, uint8_t> g_lines;   // line -> bit 0 read, bit 1 written (this read)
bool g_on = false;
uint64_t g_reads = 0;

inline int find_region(uint64_t a) {
    int lo = 0, hi = (int)g_regions.size() - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        if (a < g_regions[mid].lo) hi = mid - 1;
        else if (a >= g_regions[mid].hi) lo = mid + 1;
        else return mid;
    }
    return -1;
}
inline void touch(uint64_t a, uint64_t n, bool wr) {
    if (!g_on) return;
    const int r = find_region(a);
    if (r < 0) return;
    if (wr) ++g_regions[r].wr_acc; else ++g_regions[r].rd_acc;
    for (uint64_t l = a >> 6; l <= (a + n - 1) >> 6; ++l) g_lines[l] |= wr ? 2 : 1;
}
} // namespace

extern "C" {
void mgx_trace_region(const void *p, uint64_t bytes, const char *name) {
    if (!p || !bytes) return;
    Region r; r.lo = (uint64_t)p; r.hi = r.lo + bytes; r.name = name;
    g_regions.push_back(r);
    std::sort(g_regions.begin(), g_regions.end(), [](const Region &a, const Region &b) { return a.lo < b.lo; });
}
void mgx_trace_reset() { g_regions.clear(); g_lines.clear(); g_on = false; g_reads = 0; }
void mgx_trace_on(int on) { g_on = on != 0; }
// end of one read: fold the touched lines into the per-region totals
void mgx_trace_end_read() {
    for (auto &kv : g_lines) {
        const int r = find_region(kv.first << 6);
        const int r2 = r >= 0 ? r : find_region((kv.first << 6) + 63);
        if (r2 < 0) continue;
        if (kv.second & 1) ++g_regions[r2].rd_lines;
        if (kv.second & 2) ++g_regions[r2].wr_lines;
    }
    g_lines.clear();
    ++g_reads;
}
// report: name \t read lines \t written lines \t read accesses \t write accesses (totals over all reads), regions of one name merged
uint64_t mgx_trace_report(char *buf, uint64_t cap) {
    std::vector<Region> m;
    for (const Region &r : g_regions) {
        auto it = std::find_if(m.begin(), m.end(), [&](const Region &x) { return x.name == r.name; });
        if (it == m.end()) { m.push_back(r); continue; }
        it->rd_lines += r.rd_lines; it->wr_lines += r.wr_lines; it->rd_acc += r.rd_acc; it->wr_acc += r.wr_acc;
    }
    std::string s = "reads\t" + std::to_string(g_reads) + "\n";
    for (const Region &r : m)
        s += r.name + "\t" + std::to_string(r.rd_lines) + "\t" + std::to_string(r.wr_lines) + "\t" + std::to_string(r.rd_acc) + "\t" + std::to_string(r.wr_acc) + "\n";
    if (buf && cap) { strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0; }
    return s.size();
}

#define MGX_HOOK(n)                                                                      \
    void __asan_load##n(uintptr_t a) { touch(a, n, false); }                              \
    void __asan_store##n(uintptr_t a) { touch(a, n, true); }                              \
    void __asan_load##n##_noabort(uintptr_t a) { touch(a, n, false); }                    \
    void __asan_store##n##_noabort(uintptr_t a) { touch(a, n, true); }
MGX_HOOK(1) MGX_HOOK(2) MGX_HOOK(4) MGX_HOOK(8) MGX_HOOK(16)
void __asan_loadN(uintptr_t a, size_t n) { touch(a, n, false); }
void __asan_storeN(uintptr_t a, size_t n) { touch(a, n, true); }
void __asan_loadN_noabort(uintptr_t a, size_t n) { touch(a, n, false); }
void __asan_storeN_noabort(uintptr_t a, size_t n) { touch(a, n, true); }
void __asan_handle_no_return() {}
void __asan_init() {}
void __asan_version_mismatch_check_v8() {}
void __asan_register_globals(void *, uintptr_t) {}
void __asan_unregister_globals(void *, uintptr_t) {}
void __asan_alloca_poison(uintptr_t, size_t) {}
void __asan_allocas_unpoison(uintptr_t, uintptr_t) {}
void __asan_before_dynamic_init(const char *) {}
void __asan_after_dynamic_init() {}
}
