// Test harness (tests/test_boss_files.py): csrc/boss_files.hpp under -fsanitize=address,undefined over mutated files.
//   files_fuzz KIND FILE ROUNDS SEED      KIND = dbg | columns
// Every mutation must end in a parsed table or in a ParseError / Unsupported — the sanitizers abort on anything else.
#include <cstdlib>
#include <random>

#include "../metagraph_amd/csrc/boss_files.hpp"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const bool dbg = std::string(argv[1]) == "dbg";
    const std::vector<uint8_t> good = mgx::files::read_whole_file(argv[2]);
    const long rounds = atol(argv[3]);
    std::mt19937_64 rng((uint64_t)atoll(argv[4]));
    long ok = 0, bad = 0;
    auto run = [&](const std::vector<uint8_t> &d) {
        try {
            if (dbg) { auto g = mgx::files::parse_dbg(d.data(), d.size()); if (g.W.size() != g.last.size()) abort(); }
            else { auto f = mgx::files::parse_columns(d.data(), d.size()); if (f.col_begin.back() != f.rows.size()) abort(); }
            ++ok;
        } catch (const mgx::files::ParseError &) { ++bad; } catch (const mgx::files::Unsupported &) { ++bad; }
    };
    run(good);
    if (ok != 1) { fprintf(stderr, "the unmodified file does not parse\n"); return 1; }
    for (long r = 0; r < rounds; ++r) {
        std::vector<uint8_t> d = good;
        switch (rng() % 5) {
            case 0: d.resize(rng() % (d.size() + 1)); break;                                    // truncation
            case 1: d[rng() % d.size()] ^= (uint8_t)(1u << (rng() % 8)); break;                 // bit flip
            case 2: { const size_t at = rng() % d.size(); for (size_t i = at; i < at + 8 && i < d.size(); ++i) d[i] = (uint8_t)rng(); break; }   // a field overwritten
            case 3: { const size_t at = rng() % d.size(); for (size_t i = at; i < at + 8 && i < d.size(); ++i) d[i] = 0xFF; break; }              // ... by all ones
            case 4: { const size_t at = rng() % d.size(), n = rng() % 64; d.insert(d.begin() + (std::ptrdiff_t)at, n, (uint8_t)rng()); break; }   // bytes inserted
        }
        run(d);
    }
    printf("%ld parsed, %ld rejected\n", ok, bad);
    return 0;
}
