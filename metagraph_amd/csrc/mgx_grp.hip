// mgx_grp.hip — the sub-wave-group instantiation of the aligner's wave program (see wave_group.hpp):
// every hardware wavefront aligns 64 / MGX_GROUP reads at once, one per group of MGX_GROUP lanes.
// Same sources as the wave-per-read kernel (dev_graph.hpp, align_core.hpp); symbols live in their own
// namespace so that all instantiations coexist in libmgx.so.
#ifndef MGX_GROUP
#define MGX_GROUP 16
#endif
#define MGX_CAT2(a, b) a##b
#define MGX_CAT(a, b) MGX_CAT2(a, b)
// Two builds of this file go into libmgx.so: the product (MGX_MAX_ALT = 1: one alignment per query) and one with room for
// MGX_MAX_ALTERNATIVE_PATHS alignments per query (-DMGX_MAX_ALT=4 -DMGX_ALT_BUILD), launched when
// DBGAlignerConfig::num_alternative_paths > 1; its bigger control block stays out of the product kernel's LDS.
// A third build (-DMGX_PRIM_BUILD -DMGX_WITH_PRIMARY=1) is the product kernel with the CanonicalDBG branches compiled in: PRIMARY
// graphs with one alignment per query run the rounds at 3 waves per SIMD like every other graph, instead of sharing the
// 2-wave build of the alternative paths.
// A fourth build (-DMGX_LAB_BUILD -DMGX_ALT_BUILD -DMGX_WITH_LABELS=1 -DMGX_MAX_ALT=4) carries the label-aware extender
// (label_sets.hpp / label_driver.hpp): the batches of a mgx_labeled_aligner_create aligner, through the per-read program.
#if defined(MGX_LAB_BUILD)
#define MGX_SUFFIX(x) MGX_CAT(x, _lab)
#define mgx MGX_CAT(MGX_CAT(mgx_grp, MGX_GROUP), l)
#elif defined(MGX_ALT_BUILD)
#define MGX_SUFFIX(x) MGX_CAT(x, _alt)
#define mgx MGX_CAT(MGX_CAT(mgx_grp, MGX_GROUP), a)
#elif defined(MGX_PRIM_BUILD)
#define MGX_SUFFIX(x) MGX_CAT(x, _prim)
#define mgx MGX_CAT(MGX_CAT(mgx_grp, MGX_GROUP), p)
#else
#define MGX_SUFFIX(x) x
#define mgx MGX_CAT(mgx_grp, MGX_GROUP)
#endif
#define MGX_PARAMS_IN_LDS 1     // the kernel keeps one copy of AlignParams in LDS; the per-read program reads it with ds_ loads
#include "wave_group.hpp"
#include "align_core.hpp"

using namespace mgx;

// Waves per SIMD.  The kernel is bound by dependent round trips inside each group (halving the resident groups halves the
// throughput: 315 -> 581 -> 1080 ms per 2 M reads at 100 / 50 / 25 % of the groups, profiles/r03_ab1.txt), so occupancy is
// the lever.  Round 2 ran 2 waves: under its DRAM-saturating per-read traffic a third wave was slower (396 vs 318 ms).
// With the round-3 traffic (one 64-byte slot per column, aliased convergence entries) 3 waves — the 168-VGPR budget, 16
// scratch accesses in extend() — run 15 % faster (265 vs 313 ms).  4 waves do not fit: the control blocks of 8 groups
// alone are 10 KB of the 10 KB a wavefront would get.  The build that carries alternative paths / PRIMARY graphs has a
// bigger control block and stays at 2.
#ifndef MGX_GRP_WAVES_PER_SIMD
#ifdef MGX_ALT_BUILD
#define MGX_GRP_WAVES_PER_SIMD 2
#else
#define MGX_GRP_WAVES_PER_SIMD 3
#endif
#endif

// each group owns one read at a time, one arena slice and one slice of the dynamic LDS
template <int PHASE>
__global__ void __launch_bounds__(64, MGX_GRP_WAVES_PER_SIMD) MGX_SUFFIX(MGX_CAT(k_align_grp, MGX_GROUP))(AlignParams P, uint32_t lds_bytes, uint32_t n_groups) {
    const int g = group_id();
    // (a small batch is spread over the wavefronts: only the first `groups_per_wave` groups of each take reads, see mgx.hip)
    const uint32_t gpw = P.groups_per_wave ? P.groups_per_wave : (uint32_t)GROUPS_PER_WAVEFRONT;
    const uint32_t slot = blockIdx.x * gpw + (uint32_t)g;
    __shared__ Wave ws[GROUPS_PER_WAVEFRONT];
    int8_t *sm_rows = g_sm_rows;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&P);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&g_params);      // (a kernel argument whose address is taken would live in scratch)
        for (uint32_t x = threadIdx.x; x < sizeof(AlignParams) / 4; x += 64) dst[x] = src[x];
    }
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    for (int x = threadIdx.x; x < 6 * 128; x += 64) {
        uint32_t code = (uint32_t)(x >> 7);
        uint8_t row = code != 5 ? decode_code(code) : 0;
        sm_rows[x] = P.score_matrix[(uint32_t)(row & 127) * 128 + (x & 127)];
    }
    __syncthreads();
    if (slot >= n_groups) return;                      // the last wavefront may hold fewer groups than arena slices exist
    if ((uint32_t)g >= gpw) return;
    Wave &w = ws[g];
    uint8_t *lds = dyn_lds + (uint32_t)g * lds_bytes;
    KernelStats acc;
    memset(&acc, 0, sizeof(acc));
    const uint64_t n_items = P.n_items ? P.n_items : (P.n_items_ptr ? *P.n_items_ptr : P.n_reads);
    // next work item of this group: position `item` of the (sorted) order, or a retry position of the pass before
    auto fetch = [&](uint64_t &read, const uint8_t *&rec) -> bool {
        LV<uint64_t> rv;
        rv.v = 0;
        if (lane_id() == 0) rv.v = atomicAdd(g_params.read_cursor, 1ull);
        const uint64_t item = wave_bcast(rv, 0);
        if (item >= n_items) return false;
        read = (PHASE == PH_EXTEND && g_params.order) ? g_params.order[item] : item;
        rec = nullptr;
        if (PHASE == PH_EXTEND && g_params.resume_in) {         // a later pass: `read` is a retry position of the pass before
            rec = g_params.resume_in + read * g_params.resume_rec_bytes;
            read = g_params.resume_reads[read];
        }
        return true;
    };
    // (the product build has no per-read program at all: one alignment per seed always takes the flat loop; the build that
    // carries alternative paths keeps both, and -DMGX_KEEP_LEGACY=1 gives an A/B build that obeys MGX_NO_FLAT)
#if defined(MGX_ALT_BUILD) || defined(MGX_KEEP_LEGACY)
    const bool flat = PHASE == PH_EXTEND && n_alt_of(w) == 1 && !g_params.no_flat && !(kWithLabels && g_params.labeled);
#else
    constexpr bool flat = PHASE == PH_EXTEND;
#endif
    if (flat) {
        // rounds of service -> extension loop -> service -> trace loop (align_core.hpp, flat_service): the phases are
        // lock-step, the reads are not
        w.fs.act = ACT_FETCH;
        for (;;) {
            flat_service(w, slot, &acc, lds, lds_bytes, n_items);
            if (w.fs.act == ACT_EXIT) break;
            if (w.fs.act == ACT_EXTEND) flat_extend_all(w);
            else flat_bt_all(w);
        }
    }
#if defined(MGX_ALT_BUILD) || defined(MGX_KEEP_LEGACY) || defined(MGX_GRP_SEED_PROBE)
    else {
        for (;;) {
            uint64_t read = 0;
            const uint8_t *rec = nullptr;
            if (!fetch(read, rec)) break;
            align_read<PHASE>(w, g_params, read, slot, &acc, nullptr, sm_rows, lds, lds_bytes, rec);
        }
    }
#endif
    if (lane_id() == 0) {
        atomicAdd(&P.stats->rank_lines, acc.rank_lines);
        atomicAdd(&P.stats->select_lines, acc.select_lines);
        atomicAdd(&P.stats->bit_lines, acc.bit_lines);
        atomicAdd(&P.stats->columns, acc.columns);
        atomicAdd(&P.stats->fast_columns, acc.fast_columns);
        atomicAdd(&P.stats->extensions, acc.extensions);
        atomicAdd(&P.stats->seeds, acc.seeds);
        atomicAdd(&P.stats->capacity_errors, acc.capacity_errors);
        if (acc.seed_lines) atomicAdd(&P.stats->seed_lines, acc.seed_lines);
        for (int x = 0; x < 8; ++x) {
            atomicAdd(&P.stats->cyc[x], acc.cyc[x]);
            atomicAdd(&P.stats->xcyc[x], acc.xcyc[x]);
        }
    }
}

// n_groups = arena slices; lds_bytes = dynamic LDS per group
// phase = PH_BOTH (fused) or PH_EXTEND (after the seeding kernel)
extern "C" int MGX_SUFFIX(MGX_CAT(mgx_launch_align_grp, MGX_GROUP))(const void *params, uint32_t n_groups, uint32_t lds_bytes, int phase, void *stream) {
    const AlignParams &P = *static_cast<const AlignParams *>(params);
    const uint32_t gpw = P.groups_per_wave ? P.groups_per_wave : (uint32_t)GROUPS_PER_WAVEFRONT;
    uint32_t blocks = (n_groups + gpw - 1) / gpw;
#if defined(MGX_GRP_SEED_PROBE) && !defined(MGX_ALT_BUILD)
    if (phase == PH_SEED) {          // A/B probe: the seeding half with 8 lanes per read
        MGX_SUFFIX(MGX_CAT(k_align_grp, MGX_GROUP))<PH_SEED><<<blocks, 64, lds_bytes * GROUPS_PER_WAVEFRONT, (hipStream_t)stream>>>(P, lds_bytes, n_groups);
        return (int)hipGetLastError();
    }
#endif
    if (phase != PH_EXTEND) return (int)hipErrorInvalidValue;      // only the extension half is instantiated for sub-wave groups
    MGX_SUFFIX(MGX_CAT(k_align_grp, MGX_GROUP))<PH_EXTEND><<<blocks, 64, lds_bytes * GROUPS_PER_WAVEFRONT, (hipStream_t)stream>>>(P, lds_bytes, n_groups);
    return (int)hipGetLastError();
}
extern "C" int MGX_SUFFIX(MGX_CAT(mgx_grp_waves_per_simd, MGX_GROUP))(void) { return MGX_GRP_WAVES_PER_SIMD; }
#if defined(MGX_LAB_BUILD)
extern "C" int MGX_SUFFIX(MGX_CAT(mgx_grp_max_alt, MGX_GROUP))(void) { return MGX_MAX_ALT; }
#endif
extern "C" unsigned MGX_SUFFIX(MGX_CAT(mgx_grp_static_lds, MGX_GROUP))(void) { return (unsigned)(sizeof(Wave) * GROUPS_PER_WAVEFRONT + sizeof(AlignParams) + sizeof(g_sm_rows)); }
