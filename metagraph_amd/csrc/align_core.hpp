// align_core.hpp — the per-read wave program: seeding bookkeeping, seed extension (x-drop affine
// DP over graph neighbours), backtracking, strand handling and best-alignment selection.
//
// One wavefront owns one read.  Control flow is wave-uniform; DP columns, scans, table searches
// and array copies are lane-parallel (FOR_LANES).  All per-read state lives in a per-wave slice
// of an HBM arena (AlignParams::arena) that stays L2-hot; the only graph traffic is the 64-byte
// blocks of dev_graph.hpp.
//
// Reference being restated (A/ = M/src/graph/alignment/):
//   seeding      A/aligner_seeder_methods.cpp:49-93,153-424   (SuffixSeeder<UniMEMSeeder>)
//   extension    A/aligner_extender_methods.cpp:66-1034       (DefaultColumnExtender)
//   driver       A/dbg_aligner.cpp:193-384,531-758            (DBGAligner<>, BASIC graphs)
//   aggregator   A/aligner_aggregator.hpp:68-202              (num_alternative_paths == 1)
// Integer results are bit-exact; the four double comparisons and the one fma are kept as in the
// reference (see DESIGN.md "floating point").
#pragma once
#include "align_types.hpp"

// per-group inlining control (register pressure vs. call overhead)
#ifndef MGX_NI_MASK
#define MGX_NI_MASK 15   // depth-3 call chains fault on gfx950 (ROCm 7.2); keep at most two noinline levels
#endif
#if MGX_NI_MASK & 1
#define MGX_NI_G1 MGX_DEV_NOINLINE
#else
#define MGX_NI_G1 MGX_DEV
#endif
#if MGX_NI_MASK & 2
#define MGX_NI_G2 MGX_DEV_NOINLINE
#else
#define MGX_NI_G2 MGX_DEV
#endif
#if MGX_NI_MASK & 4
#define MGX_NI_G3 MGX_DEV_NOINLINE
#else
#define MGX_NI_G3 MGX_DEV
#endif
#if MGX_NI_MASK & 16
#define MGX_NI_G5 MGX_DEV_NOINLINE
#else
#define MGX_NI_G5 MGX_DEV
#endif
#if MGX_NI_MASK & 8
#define MGX_NI_G4 MGX_DEV_NOINLINE
#else
#define MGX_NI_G4 MGX_DEV
#endif

// kernels that keep AlignParams in LDS (mgx_grp.hip) say so; elsewhere it is a kernel argument
#if defined(MGX_PARAMS_IN_LDS) && MGX_PARAMS_IN_LDS
// The sub-wave-group kernel keeps ONE copy of AlignParams per workgroup in LDS, as a namespace-scope __shared__ object:
// every function, inlined or not, then reads it with ds_ instructions at a fixed address.  (Through a pointer kept in
// the control block the compiler cannot tell the address space, and a FLAT load makes the wave wait for every global
// load and store in flight — including the ones issued early on purpose.)
#define MGX_PARAMS_OF(w) (::mgx::g_params)
#else
#define MGX_PARAMS_OF(w) (*(w).P)
#endif

// PRIMARY graphs (DevConfig::canonical == 2, the CanonicalDBG wrapper of canon_graph.hpp) are compiled into separate
// instantiations (-DMGX_WITH_PRIMARY=1: mgx_primary.hip for seeding, the mgx_grp.hip build that also carries alternative
// paths for extension), so that the kernels every other graph runs on are the code they were without it.
#ifndef MGX_WITH_PRIMARY
#define MGX_WITH_PRIMARY 0
#endif
// Label-aware alignment (LabeledAligner / LabeledExtender, A/aligner_labeled.{hpp,cpp}) is compiled into its own instantiation
// of the extension kernel (-DMGX_WITH_LABELS=1: mgx_lab64.hip) and into the host model; AlignParams::labeled switches it on
// at run time there.  Every other kernel is the code it was without it.
#ifndef MGX_WITH_LABELS
#define MGX_WITH_LABELS 0
#endif

// timing ablations (results become WRONG) are compiled into -DMGX_PROBES builds only
#ifdef MGX_PROBES
#define MGX_ABLATED(w, bit) ((MGX_PARAMS_OF(w).ablate & (bit)) != 0)
#else
#define MGX_ABLATED(w, bit) false
#endif

namespace mgx {

constexpr bool kWithPrimary = MGX_WITH_PRIMARY != 0;
constexpr bool kWithLabels = MGX_WITH_LABELS != 0;

#if defined(MGX_PARAMS_IN_LDS) && MGX_PARAMS_IN_LDS
__shared__ AlignParams g_params;
__shared__ int8_t g_sm_rows[6 * 128];      // score-matrix rows of the 6 possible path characters ($ACGT\0) x 128
#define MGX_SM_ROWS(w) (::mgx::g_sm_rows)
#else
#define MGX_SM_ROWS(w) ((w).sm_rows)
#endif

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
MGX_DEV uint32_t encode_char(uint8_t ch) {           // kmer/alphabets.hpp:67-76 (+ kmer_extractor.cpp:33-36)
    if (ch & 0x80) return 5;
    switch (ch) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        case 'T': case 't': case 'U': case 'u': return 4;
        default: return 5;
    }
}

MGX_DEV uint8_t decode_code(uint32_t c) {            // "$ACGT"
    return c == 0 ? '$' : c == 1 ? 'A' : c == 2 ? 'C' : c == 3 ? 'G' : 'T';
}

MGX_DEV uint8_t complement_char(uint8_t c) {         // COMPL_TAB, common/seq_tools/reverse_complement.hpp:31-48
    const char *up = "TVGHEFCDIJMLKNOPQYSAABWXRZ";
    if (c >= 'A' && c <= 'Z') return (uint8_t)up[c - 'A'];
    if (c >= 'a' && c <= 'z') return (uint8_t)(up[c - 'a'] + ('a' - 'A'));
    if (c == 96) return 64;
    return c;
}

MGX_DEV uint8_t to_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

} // namespace mgx
#include "map_chain.hpp"
#include "canon_graph.hpp"
namespace mgx {

MGX_DEV uint8_t char_to_op(uint8_t a, uint8_t b) {   // initialize_opt_table (A/aligner_cigar.cpp:10-51)
    uint8_t ua = to_upper(a), ub = to_upper(b);
    bool valid = (ua == 'A' || ua == 'C' || ua == 'G' || ua == 'T');
    return (valid && ua == ub) ? OP_MATCH : OP_MISMATCH;
}

template <class T>
MGX_DEV T imin(T a, T b) { return a < b ? a : b; }
template <class T>
MGX_DEV T imax(T a, T b) { return a > b ? a : b; }
MGX_DEV int32_t iabs(int32_t a) { return a < 0 ? -a : a; }

// ------------------------------------------------------------------------------------------------
// per-wave arena
// ------------------------------------------------------------------------------------------------
// Lanes per read of the extension kernel, which fixes the width of the chain path's register window and with it the
// layout of the column slots in the arena.  The HIP build runs the extension with 8-lane groups only (mgx_grp.hip); the
// host translation unit (mgx.hip, 64-lane seeding kernel) must compute the same arena layout, hence a constant that does
// not depend on the translation unit's own WAVE.  The host model runs every lane count.
#if MGX_WAVE_EMU
constexpr int32_t EXT_LANES = WAVE;
#else
constexpr int32_t EXT_LANES = 8;
#endif
constexpr int32_t FWS = 4 * EXT_LANES;          // cells of a chain-format column slot

struct ColMeta {                 // DPTColumn minus the vectors (aligner_extender_methods.hpp:129-147), in registers
    uint32_t node;
    int32_t parent;
    int32_t offset, max_pos, trim, size;
    int32_t score;               // edge score
    uint32_t cells;              // word offset of the column's S / F record in the cell arena (see rec_words); NO_CELLS: none
    uint32_t cw;                 // path character (bits 0-7) | cells per array of the record (bits 8-30) | bit 31: chain format
    int32_t org;                 // window position of record / slot cell 0 (org <= trim)
    int32_t base;                // chain formats: S = base + 16-bit / 8-bit offset
    int32_t self;                // the column's own table index (not stored)
};
constexpr uint32_t NO_CELLS = 0xFFFFFFFFu;
constexpr uint32_t CW_CHAIN = 0x80000000u;       // a chain-format column (S relative to `base`, flags in the slot) ...
constexpr uint32_t CW_COMPACT = 0x40000000u;     // ... in the compact one-line form (see ColSlot)
MGX_DEV uint8_t col_char(const ColMeta &c) { return (uint8_t)(c.cw & 0xFF); }
MGX_DEV int32_t col_wc(const ColMeta &c) { return (int32_t)((c.cw & ~(CW_CHAIN | CW_COMPACT)) >> 8); }
MGX_DEV bool col_chain(const ColMeta &c) { return (c.cw & CW_CHAIN) != 0; }
MGX_DEV bool col_compact(const ColMeta &c) { return (c.cw & CW_COMPACT) != 0; }

// What a column leaves in HBM.  Every column owns one slot of the table (`Wave::cols`): 32 bytes of metadata, one FLAG
// byte per cell of the chain window and the window's S values as 16-bit offsets from `base` — one or two 64-byte lines
// that the chain path writes once and backtracking reads back (a 64-byte line costs the same DRAM access whether 4 or 64
// of its bytes are used, and the kernel is bound by the number of such accesses).  The flags are every fact about a cell
// that backtrack (:800-1034) tests, computed while the column and its parent are in registers / staging:
//   bit 0  S != ninf                     bit 1  S == E                    bit 2  E[j] == E[j - 1] + gap_ext  (E[-1] = ninf)
//   bit 3  S == S_parent[pos - 1] + edge score + profile   (and pos - 1 inside the parent column: the match test :963-975)
//   bit 4  S == F                        bit 5  F == F_parent[pos] + edge score + gap_ext  (deletion run :985-1003)
//   bit 6  S_parent[pos - 1] != ninf     (start-cell filter :846-853)
// Columns of the general path (any width) keep S and F as int32 arrays plus their flag bytes in a record of the cell
// arena (`cells`); a chain-format column gets such a record (S, F of its window) only when it stays behind in the
// frontier, i.e. when something may have to reload it as a parent.
// Three forms share the slot (round 3; the 128-byte slot of round 2 wrote two lines per column):
//   general   m[8] as packed by col_pack; S / F / flags in the column's record of the cell arena.
//   chain     m[8] + one flag byte per window cell here, the window's S as 16-bit offsets from `base` in the column's row
//             of a SECOND array (`Wave::cols_s16`) that only reloads-as-data touch (convergence merges, the rare S reads
//             of backtrack).
//   compact   ONE 64-byte line for everything: 16 bytes of metadata and, for the first CCELLS = 24 window cells, the flag
//             byte and S as an 8-bit offset from `base` (-128 = ninf).  Taken by a chain column whose window cells from
//             CCELLS on hold neither S nor F (nothing can ever trace into them), whose S fit 8 bits (x-drop < ~100),
//             and whose metadata fits the narrow fields; ~95 % of the columns of a short-read batch.
//             words 0-3: node | parent:24 char code:3 score code:2 -:1 tag 01 | base:22 size:5 max_pos-trim:5 | offset:16 trim:15
//             words 4 + 2 l, 5 + 2 l (l < 6): flag bytes / 8-bit S of window cells 4 l .. 4 l + 3
//             (tag: bits 31-30 of word 1 are 00 or 11 in the other forms, whose word 1 is the parent index or -1)
struct alignas(16) ColSlot {
    uint32_t m[8];
    uint8_t flags[FWS];
};
constexpr int32_t CCELLS = 24;
constexpr int8_t S8_NINF = INT8_MIN;
enum { CF_REAL = 1, CF_S_IS_E = 2, CF_E_EXT = 4, CF_MATCH = 8, CF_S_IS_F = 16, CF_F_EXT = 32, CF_SP_REAL = 64 };
constexpr int16_t S16_NINF = INT16_MIN;

// S / F record of a column in the cell arena: S[wc], F[wc] (int32) and wc flag bytes; record cell x holds window position
// org + x; positions outside [trim, trim + size + 5) or outside the record read as ninf / 0, exactly what the reference's
// vectors hold there (never-written padding).  wc is a multiple of 4; records are 16-byte aligned.
MGX_HD uint32_t rec_words(uint32_t wc) { return (2 * wc + wc / 4 + 3) & ~3u; }

// all lanes hold the same metadata; moving it to scalar registers makes every dependent branch and
// address computation scalar
MGX_DEV ColMeta uni_col(const ColMeta &c) {
    ColMeta r;
    r.node = uni(c.node); r.parent = uni(c.parent); r.offset = uni(c.offset); r.max_pos = uni(c.max_pos);
    r.trim = uni(c.trim); r.size = uni(c.size); r.score = uni(c.score); r.cells = uni(c.cells);
    r.cw = uni(c.cw); r.org = uni(c.org); r.base = uni(c.base); r.self = uni(c.self);
    return r;
}

struct BtIndex { int32_t score, neg_off_diag, neg_i, pos; };

// Alignments per query this build can keep (DBGAlignerConfig::num_alternative_paths).  The product kernel is built for 1;
// a second instantiation of the extension kernel (and the host model) is built for MGX_MAX_ALT_BUILD.
#ifndef MGX_MAX_ALT
#define MGX_MAX_ALT 1
#endif
constexpr int MAX_ALT = MGX_MAX_ALT;
// label-aware alignment: a backtracking reports up to LAB_EXT alignments (one per label subset of the seed), the per-label
// aggregator holds up to LAB_POOL_PER_ALT x num_alternative_paths alignments (DevLimits::lab_ext / lab_pool, host_common.hpp)
// — the sizes of a first run; mgx_align_batch re-runs a read that outgrows them with larger ones
constexpr int LAB_EXT = 8, LAB_POOL_PER_ALT = 32;
constexpr int N_ALN = kWithLabels ? 3 * LAB_EXT + LAB_POOL_PER_ALT * MAX_ALT : 4 * MAX_ALT;   // alignment buffers: extension results, their reversals, backward results, the best

// the convergence table of one extender (layout and entry kinds: see the "convergence checker" section)
struct ConvRec { uint32_t off; int32_t start, len; uint32_t cap; };      // a pool entry: words [off, off + cap) hold [start, start + len)
struct ConvChecker {             // SeedFilteringExtender::conv_checker_ (extender hpp:75-76)
    uint64_t *tab;               // hash slots, all levels (conv_tab_slots()); the pool entries' ConvRecs follow them
    int32_t *pool;               // DevLimits::conv_pool_words words
    uint32_t n_entries, n_recs;
    uint32_t gen;                // 1 .. 255
    uint32_t pool_top;           // words handed out since conv_clear
    uint32_t cap, base;          // current level: slots and its first slot in `tab`
    uint32_t start;              // window origin (query position) of the extension the entries belong to
    uint32_t dirty;              // highest level written since the generations last wrapped
};

constexpr uint32_t CONV_CAP0 = 512;
MGX_HD uint32_t conv_cap0(uint32_t hash_size) { return hash_size < CONV_CAP0 ? hash_size : CONV_CAP0; }
MGX_HD uint64_t conv_tab_slots(uint32_t hash_size) { return 2ull * hash_size; }          // all levels together (< 2 x the top one)

struct SdustScratch {            // working set of is_low_complexity(); lives in LDS on the device
    int16_t Ps[64], Pf[64], Pr[64], Pl[64];   // perfect intervals (at most one per window position)
    // the triplet counters and the window deque (ring): memory only where a read has fewer than 64 lanes; the 64-lane
    // instantiation keeps them in registers (RegTab64) and its kernel allocates just the first SDUST_LDS_BYTES of this struct
    int16_t cv[64], cw[64], c2[64];
    int16_t wqw[64];
};
constexpr uint32_t SDUST_LDS_BYTES_REGTAB = 4 * 64 * 2;

struct DevAln {                  // Alignment (alignment.hpp:132-331)
    uint32_t *nodes;
    uint32_t *cigar;             // len << 3 | op
    uint8_t *seq;
    int32_t n_nodes, n_cigar, seq_len;
    int32_t score, offset;
    int32_t qbegin, qlen;        // query_view within the strand's query
    int32_t orientation;         // strand of the query this alignment is on
    int32_t extra_score;
#if MGX_WITH_LABELS
    uint32_t lab;                // label_columns (alignment.hpp:285): a set of the read's label arena (label_sets.hpp), 0 = none
#endif
};

MGX_DEV int32_t aln_clipping(const DevAln &a) {
    return a.n_cigar && (a.cigar[0] & 7) == OP_CLIPPED ? (int32_t)(a.cigar[0] >> 3) : 0;
}
MGX_DEV int32_t aln_end_clipping(const DevAln &a) {
    return a.n_cigar && (a.cigar[a.n_cigar - 1] & 7) == OP_CLIPPED ? (int32_t)(a.cigar[a.n_cigar - 1] >> 3) : 0;
}

// what the extender needs to know about its seed (a Seed-derived or a reversed Alignment)
struct SeedRef {
    const uint32_t *nodes;
    const uint8_t *seq;
    int32_t n_nodes, seq_len;
    int32_t clipping, end_clipping, qlen, offset, score;
    int32_t orientation;
#if MGX_WITH_LABELS
    uint32_t lab;                // the seed's label_columns
#endif
};

struct ExtenderState {           // one per strand (Extender object in dbg_aligner.cpp:287,292)
    const uint8_t *q;            // normalized query of this strand
    const int32_t *psum;         // partial_sums_ (aligner_extender_methods.cpp:26-36)
    int32_t psum_lin;            // > 0: the strand is pure ACGT with one self-score m, so psum[x] == (L - x) * m exactly; else 0
    ConvChecker conv;
    uint32_t table_cap;          // capacity of the reference's std::vector<DPTColumn>
    int32_t rc_view;             // 1 while this extender runs on the RCDBG view
};

struct ExtendResult { int32_t n_tips; int32_t min_cell_score; int32_t table_size; };

// One DP column staged on chip (S, E, F incl. the 5-cell padding): the column being computed and its
// parent live in LDS so that the hot path of an extension never waits on the HBM arena.
// A staged array is two-tiered: the first `st_cap` cells in LDS (lo), the rest in the arena (hi, indexed by the
// same j): x-drop bands are narrow, so at 8 reads per wavefront the common columns live entirely in LDS while
// any width stays correct.  The parent column needs S and F only; E exists once, for the column being computed.
struct Tier { int32_t *lo, *hi; };
// One FLAT access through the selected pointer.  (Measured alternative: separate ds_/global_ paths behind a branch
// on j < cap, with an LDS address-space hint for `lo` — 2 % slower, the extra branches cost more than decoupling
// the LDS tier from vmcnt gains.)
MGX_DEV int32_t tget(const Tier &t, int32_t cap, int32_t j) { return *(j < cap ? t.lo + j : t.hi + j); }
MGX_DEV void tset(const Tier &t, int32_t cap, int32_t j, int32_t v) { *(j < cap ? t.lo + j : t.hi + j) = v; }
struct Staging { Tier S, F; int32_t col; };

// Loop-carried state of one extension (DefaultColumnExtender::extend): lives in LDS next to the control block so that
// the step functions of the flat extension loop are separate small functions.  fS / fF: S and F of the chain path's
// current column, four consecutive window positions per lane starting at f_org.
#ifndef MGX_NO_EXTEND
constexpr int32_t FW = 4 * WAVE;                  // cells of the register window (4 per lane)
// Cells of it the chain path may use: what a column slot holds (FWS).  The 64-lane instantiation (mgx_ext64.hip: FW = 256)
// shares the arena layout of the 8-lane groups, so its chain columns are confined to the first FWS cells of the window — lanes
// from FWS / 4 on stay ninf — and a wider band takes the general path, as it does in the 8-lane kernel.
constexpr int32_t CHW = FW < FWS ? FW : FWS;
static_assert(CHW <= FWS && CHW <= FW && (CHW & 3) == 0, "a chain column must fit its slot");
struct XState {
    int32_t xdrop_cutoff, best_score, tsize, min_cell_score, qn, nn, n_tips;
    int32_t q_top;                        // score of the frontier's top entry (INT32_MIN when empty)
    uint32_t cell_top;
    uint64_t table_size_bytes;
    int32_t start, window_size, qlen, partial_sum_offset, seed_off, seed_seq_len, psum_lin, force_fixed;
    int32_t f_idx, f_offset, f_trim, f_size, f_max_pos, f_max_val, f_org, f_n_out;
    uint32_t f_node, pad_;
    // copies of what the chain step reads every column (the seed lives in its caller's private frame and the
    // parameter block behind a generic pointer: both would be FLAT loads there)
    const uint32_t *seed_nodes;
    const uint8_t *seed_seq;
    double rel_cutoff, max_nodes_per_char, max_ram;
    int32_t go, ge, xdrop, k, Lq, max_columns, seq_lds, rc;
    int32_t n_valid, n_for, n_count, pad3_;   // children of column n_for enumerated ahead of time into Wave::out_* (chain path)
    uint32_t alias_ok, cell_words;        // alias_ok: a node's first chain column enters the convergence table as an alias
    int32_t seed_n_nodes, sn_base, sc_base, pad4_;   // seed replay: first entry of the register-cached node / character run
};
// what the backtracking keeps between its steps (bt_begin / bt_step); overlays XState, which is dead by then
struct BtState {
    int32_t es, n_max, min_path_score, produced, best_score, remaining, first_bi, stage;
    // the trace walk in progress
    int32_t j, score, pos, end_pos, n_ops, n_path, n_seq, n_trace, dummy_counter, align_offset, extra_score;
    uint32_t cur_run, last_path_node;
};
#else
struct XState { int32_t unused_; };       // seeding-only translation units (k_seed) carry no extension state in LDS
struct BtState { int32_t unused_; };
#endif

#ifdef MGX_SEED_PROBE
constexpr int XCYC_N = 8;
#else
#if defined(MGX_CHAIN_PROBE) || defined(MGX_BT_PROBE)
constexpr int XCYC_N = 8;                     // probe build: xcyc[] = the sections of chain_step (tools/probe_imbalance.py)
#else
constexpr int XCYC_N = 4;
#endif
#endif
// the flat group loop's per-read state (see flat_drive): what the group is doing, where the read's driver stands
struct FlatState {
    uint8_t act, ds, ph, s;      // activity (ACT_*), driver state (DS_*), strand phase (0 / 1), strand of the current aln_both
    uint8_t filt, pad_[3];       // a reversed backward alignment awaits its lazy filter_nodes (aln[2])
    int32_t i;                   // seed index
    uint64_t read;               // the group's current read
};

struct Wave {
#if !(defined(MGX_PARAMS_IN_LDS) && MGX_PARAMS_IN_LDS)
    const AlignParams *P;        // (the sub-wave-group kernel keeps the parameter block and the score rows as workgroup-wide
                                 // __shared__ objects: 16 bytes less per control block, which is what keeps both query
                                 // strands of a 150-bp read in LDS at 3 waves per SIMD)
#endif
    int32_t L;                   // query length
    int32_t q_lds;               // the strands q[0], q[1] live in LDS (carve)
    uint8_t *q[2];
    int32_t *psum[2];
    const uint32_t *nodes[2];
    DevSeed *seeds[2];
    uint8_t *alive[2];
    int32_t n_seeds[2];
    uint32_t num_matching[2];
    int32_t psum_lin[2];         // see ExtenderState::psum_lin
    // The seeding half's scratch pointers and the extension's loop state are never live at the same time (build_seeders
    // finishes before the first extend()), so they share their LDS bytes.
    union {
        struct {
            const uint8_t *mlen[2];      // k_map's index() match lengths per position (may be null)
            const uint2 *rng[2];         // and the (rl, ru) ranges of matches >= min_seed_length (may be null)
            uint16_t *msl, *pos_cnt, *ml;    // sub-k scratch
            uint8_t *pos_full;
            uint32_t *pos_start, *rfirst, *rlast, *alt;
            SdustScratch *sd;            // sdust scratch (LDS)
            SdustScratch *sd_own;        // carve()'s own scratch in the seeding overlay (used when the kernel passes none)
            uint32_t *pk[2];             // 2-bit packed strands (16 codes per word, first char least significant)
            uint64_t *bm[4];             // position bitmasks of the seeder: matched k-mers, MEM stops, lookup hits, seed slots
            uint8_t *dust_t;             // triplet code per position (maybe_low_complexity)
            uint64_t *dust_eq;           // per position: which of the next 61 positions hold the same triplet
            int32_t lc_any[2];           // whole-strand sdust verdict: 0 = no maskable interval anywhere, 1 = some, -1 = unknown
            int32_t lc_maybe;            // maybe_low_complexity of the read (either strand: see window_low_complexity), -1 = unknown
            int32_t inv_any[2];          // strand holds a character outside ACGT
            int32_t n_kmers;
        };
        XState x;
        BtState bt;
    };
    // extension scratch
    int32_t *cells;
    ColSlot *cols;                        // the column table: one slot per column (see ColSlot)
    int16_t *cols_s16;                    // row i: 16-bit S of chain-format column i (FWS cells)
    uint64_t *queue, *next_nodes;         // frontier / current batch (arena; the chain path rarely touches them)
    Staging st[2];
    Tier stE;                             // E of the column being computed
    int32_t st_cap;                       // cells of every staged array that live in LDS
    uint32_t blk_cache_w[16];             // target block of the last graph expansion (children live in it), as plain words
    uint32_t blk_cache_idx;
    uint32_t *tips, *prev_starts;
    BtIndex *indices;
    uint32_t *rev_ops, *rev_nodes;
    uint8_t *rev_seq;
#if !(defined(MGX_PARAMS_IN_LDS) && MGX_PARAMS_IN_LDS)
    const int8_t *sm_rows;       // score-matrix rows of the 6 possible path characters ($ACGT\\0) x 128, in LDS
#endif
    uint32_t *gen_store;         // conv-checker generation counters, persistent per arena slice
    ExtenderState ext[2];
#if MGX_WITH_LABELS
    DevAln *aln;                 // (builds with the label-aware extender hold many more alignments: the records live in the arena)
#else
    DevAln aln[N_ALN];           // [0, A): extension results, [A, 2A): reversed seeds of the backward pass, [2A, 3A): backward
                                 // extension results, [3A, 4A): the aggregator's queue (A = num_alternative_paths <= MAX_ALT)
#endif
    FlatState fs;
    int32_t have_best;           // alignments in the aggregator's queue
    int32_t seeds_done;          // seeds whose extension ran for this read in this pass (multi-pass extension)
    int32_t resume_phase, resume_i, no_limit;   // where this pass (re)starts: strand call 0 / 1, seed index; limit lifted
    LineCtr ctr;                 // BOSS block loads (lane-parallel regions add their wave sums)
    ExtendResult er;             // result of the last extend(); noinline callees must not write through
                                 // pointers into the caller's private frame, so outputs live here
    int32_t tmp_pushes;
    uint32_t out_nodes[5];       // children of the column being expanded (call_outgoing)
    int32_t out_scores[5];
    uint8_t out_chars[8];
    uint32_t cyc[8];             // phase timers (shader cycles of one read: 32 bits; LDS bytes decide the kernel's occupancy)
    uint32_t xcyc[XCYC_N];       // extend() breakdown: pop, general steps, chain steps (probe builds: per-stage timers)
    uint32_t n_columns, n_extensions, n_fast_columns;
    int32_t status;
#if MGX_WITH_LABELS
    // label-aware alignment (label_sets.hpp / label_driver.hpp)
    uint32_t *lab;               // the read's label-set arena: word 0 = 0 (the empty set), per-extension sets grow up from
    uint32_t lab_lo, lab_hi;     // lab_lo, per-read sets (seeds, alignments) grow down from lab_hi
    uint32_t *col_lab;           // [max_columns] LabeledExtender::node_labels_ of the DP table
    uint32_t *seed_lab[2];       // [max_seeds] label_columns of the seeds
    int32_t last_flushed;        // LabeledExtender::last_flushed_table_i_
    uint32_t remaining_lab;      // remaining_labels_i_
    uint32_t bt_isect, bt_diff;  // label_intersection_ / label_diff_ of the start cell being walked
    uint32_t out_lab[5];         // label sets of the children in out_nodes
    uint32_t *agg;               // the per-label aggregator's queues (label_driver.hpp)
#endif
};

MGX_DEV Block wave_blk_cache(const Wave &w) {
    Block b;
    const uint32_t *v = w.blk_cache_w;
    b.cum[0] = v[0]; b.cum[1] = v[1]; b.cum[2] = v[2]; b.cum[3] = v[3]; b.last_cum = v[4]; b.cum0 = v[5];
    b.last_bits = ((uint64_t)v[7] << 32) | v[6]; b.p0 = ((uint64_t)v[9] << 32) | v[8]; b.p1 = ((uint64_t)v[11] << 32) | v[10];
    b.p2 = ((uint64_t)v[13] << 32) | v[12]; b.pf = ((uint64_t)v[15] << 32) | v[14];
    return b;
}
MGX_DEV void wave_set_blk_cache(Wave &w, const Block &b, uint32_t idx) {
    uint32_t *v = w.blk_cache_w;
    v[0] = b.cum[0]; v[1] = b.cum[1]; v[2] = b.cum[2]; v[3] = b.cum[3]; v[4] = b.last_cum; v[5] = b.cum0;
    v[6] = (uint32_t)b.last_bits; v[7] = (uint32_t)(b.last_bits >> 32); v[8] = (uint32_t)b.p0; v[9] = (uint32_t)(b.p0 >> 32);
    v[10] = (uint32_t)b.p1; v[11] = (uint32_t)(b.p1 >> 32); v[12] = (uint32_t)b.p2; v[13] = (uint32_t)(b.p2 >> 32);
    v[14] = (uint32_t)b.pf; v[15] = (uint32_t)(b.pf >> 32);
    w.blk_cache_idx = idx;
}

// buffer roles for A = num_alternative_paths (runtime, <= MAX_ALT): [0, A) extension results, [A, 2A) their reversals (the
// seeds of the backward pass), [2A, 3A) backward extension results, [3A, 4A) the aggregator's queue
MGX_DEV int n_alt_of(const Wave &w) { return imax(1, imin((int)MGX_PARAMS_OF(w).cfg.num_alt, MAX_ALT)); }
#define Q0 (3 * n_alt_of(w))

// LocalAlignmentLess (alignment.hpp:337-348): a < b
MGX_DEV bool aln_less(const DevAln &a, const DevAln &b) {
    int32_t ca = aln_clipping(a), cb = aln_clipping(b);
    if (b.score != a.score) return b.score > a.score;
    if (a.qlen != b.qlen) return a.qlen > b.qlen;
    if (a.orientation != b.orientation) return a.orientation > b.orientation;
    return ca > cb;
}

MGX_HD uint64_t align8(uint64_t x) { return (x + 7) & ~7ull; }

// words of the per-label aggregator's state (label_driver.hpp): header, one record per label queue, reference counts
// (DevLimits::lab_queues label queues of 8 words each, lab_pool reference counts, the output order's scratch)
MGX_HD uint64_t lab_agg_words(const DevLimits &lim) { return 16 + (uint64_t)lim.lab_queues * 8 + lim.lab_pool + 8 + ((uint64_t)lim.lab_queues * 4 + 16) + lim.lab_pool + 8; }

// Test builds only (tools/fuzz_asan.sh: the host model under -fsanitize=address with -DMGX_ARENA_REDZONE=64): a poisoned gap behind
// each of the arena's larger arrays, so that an index past the end of one is reported instead of landing in its neighbour — the
// sanitizer by itself only sees the arena as one allocation.  0 in every product build: no gap, same layout.
#ifndef MGX_ARENA_REDZONE
#define MGX_ARENA_REDZONE 0
#endif
constexpr uint64_t ARENA_REDZONES = 21;                 // take_rz() calls of carve() (16 + 5 of the label part)
#if MGX_ARENA_REDZONE
#include <sanitizer/asan_interface.h>
#endif

// byte size of one wave's arena slice
MGX_HD uint64_t arena_bytes(const DevLimits &lim) {
    uint64_t L = lim.Lmax, Lp = align8(L + 8);
    uint64_t ent = (uint64_t)lim.max_columns + lim.max_path;
    uint64_t b = 0;
    b += 2 * Lp;                                        // q
    b += 2 * align8((L + 1) * 4);                       // psum
    b += 2 * align8((uint64_t)lim.max_seeds * sizeof(DevSeed));
    b += 2 * align8(lim.max_seeds);                     // alive
    b += 3 * align8((L + 1) * 2);                       // msl, pos_cnt, ml
    b += align8(L + 1);                                 // pos_full
    b += 3 * align8((L + 1) * 4);                       // pos_start, rfirst, rlast
    b += align8(sizeof(SdustScratch));
    b += 2 * align8(((L + 15) / 16 + 2) * 4) + 4 * align8(((L + 63) / 64 + 1) * 8);   // pk, bm
    b += align8((L + 8) * 8) + align8(L + 8);           // dust_eq, dust_t
    b += align8((uint64_t)lim.max_alt * 4);             // alt
    b += 16 + align8((uint64_t)lim.cell_words * 4);     // cells
    b += 64 + align8((uint64_t)lim.max_columns * sizeof(ColSlot));
    b += 64 + align8((uint64_t)lim.max_columns * FWS * 2);   // cols_s16
    b += 2 * align8((uint64_t)lim.max_columns * 8);     // queue, next_nodes
    b += align8((uint64_t)lim.max_columns * 4);         // tips
    b += align8(((uint64_t)lim.max_columns + 31) / 32 * 4);   // prev_starts
    b += align8((uint64_t)lim.max_columns * 2 * sizeof(BtIndex));
    b += 2 * align8((uint64_t)lim.max_path * 4) + align8(lim.max_path);   // rev_*
    b += 16;                                            // gen_store (generation + dirty level per extender)
    b += 6 * align8((L + 16) * 4);                      // staging
    b += 64 + 2 * (align8(2ull * lim.hash_size * 8) + align8(((uint64_t)lim.max_columns + lim.max_path) * 16) + 64 + align8((uint64_t)lim.conv_pool_words * 4));
    b += (uint64_t)lim.n_aln * (2 * align8((uint64_t)lim.max_path * 4) + align8(lim.max_path));
    if (lim.lab_words || kWithLabels)                   // builds with the label-aware extender: the alignment records themselves
        b += align8((uint64_t)lim.n_aln * sizeof(DevAln));
    if (lim.lab_words)                                  // label-aware alignment: set arena, column / seed handles, aggregator queues
        b += align8((uint64_t)lim.lab_words * 4) + align8((uint64_t)lim.max_columns * 4) + 2 * align8((uint64_t)lim.max_seeds * 4)
             + align8(lab_agg_words(lim) * 4);
    b += ARENA_REDZONES * MGX_ARENA_REDZONE;
    return (b + 63) & ~63ull;          // slices keep the 32-byte alignment of the hash slots and the 16-byte one of the cell records
}

// bytes of one resume record of the multi-pass extension (AlignParams::resume_*; layout: resume_save)
MGX_HD uint32_t resume_aln_bytes(const DevLimits &lim) { return 48u + 2u * (uint32_t)align8((uint64_t)lim.max_path * 4) + (uint32_t)align8(lim.max_path); }
MGX_HD uint32_t resume_rec_bytes(const DevLimits &lim, uint32_t n_alt) {
    return 64u + (uint32_t)align8(2ull * lim.max_seeds) + n_alt * resume_aln_bytes(lim);
}

// Carve the wave's workspace.  Small, latency-critical scalar arrays go to LDS (`lds`, `lds_bytes`)
// when they fit; everything else lives in the wave's HBM arena slice.
MGX_DEV void carve(Wave &w, const AlignParams &P, uint8_t *base, uint8_t *lds, uint32_t lds_bytes) {
    const DevLimits &lim = P.lim;
    uint64_t L = lim.Lmax, Lp = align8(L + 8);
    uint64_t ent = (uint64_t)lim.max_columns + lim.max_path;
    uint8_t *p = base;
    uint8_t *lp = lds;
    uint32_t lleft = lds_bytes;
    auto take = [&](uint64_t bytes) { uint8_t *r = p; p += align8(bytes); return r; };
#if MGX_ARENA_REDZONE
    auto take_rz = [&](uint64_t bytes) { uint8_t *r = p; p += align8(bytes); ASAN_POISON_MEMORY_REGION(p, MGX_ARENA_REDZONE); p += MGX_ARENA_REDZONE; return r; };
#else
    auto take_rz = take;
#endif
    auto take_fast = [&](uint64_t bytes) {
        uint64_t b8 = align8(bytes);
        uint8_t *r = p;
        p += b8;                                   // the arena slot is reserved either way (layout is static)
        if (b8 <= lleft) { r = lp; lp += b8; lleft -= (uint32_t)b8; }
        return r;
    };
    // persistent fast arrays
    // both strands or neither: the query is read with LDS or global instructions according to q_lds, never generic ones
    w.q_lds = 2 * Lp <= lleft ? 1 : 0;
    for (int s = 0; s < 2; ++s) w.q[s] = w.q_lds ? take_fast(Lp) : take(Lp);
    // overlay: the seeding tables and the extension's column staging are never live at the same time
    uint8_t *lp_mark = lp;
    uint32_t lleft_mark = lleft;
    w.msl = (uint16_t *)take_fast((L + 1) * 2);
    w.pos_cnt = (uint16_t *)take_fast((L + 1) * 2);
    w.ml = (uint16_t *)take_fast((L + 1) * 2);
    w.pos_full = take_fast(L + 1);
    w.pos_start = (uint32_t *)take_fast((L + 1) * 4);
    w.rfirst = (uint32_t *)take_fast((L + 1) * 4);
    w.rlast = (uint32_t *)take_fast((L + 1) * 4);
    w.sd_own = (SdustScratch *)take_fast(sizeof(SdustScratch));
    {
        // the four k-mer bitmaps are read together: all of them in LDS or none (three in LDS and one in the arena measured
        // 10 % slower for the whole seeding kernel than none)
        const uint64_t one = align8(((L + 63) / 64 + 1) * 8);
        const bool fits = 4 * one <= lleft;
        for (int b = 0; b < 4; ++b) w.bm[b] = (uint64_t *)(fits ? take_fast(one) : take(one));
    }
    w.dust_eq = (uint64_t *)take_fast((L + 8) * 8);
    w.dust_t = take_fast(L + 8);
    uint8_t *lp_seed_end = lp;
    uint32_t lleft_seed_end = lleft;
    lp = lp_mark;
    lleft = lleft_mark;
    {
        // staging: 5 arrays (S, F of two slots + E); the arena holds them at full size, LDS the first st_cap cells
        const uint64_t full = L + 16;
        uint64_t cap = (lleft / 20) & ~3ull;
        if (cap > full) cap = full;
        if (cap < 16) cap = 0;
        w.st_cap = (int32_t)cap;
        int32_t *hi[6];
        for (int a = 0; a < 6; ++a) hi[a] = (int32_t *)take(full * 4);          // (6th slot kept for layout stability)
        int32_t *lo[5];
        for (int a = 0; a < 5; ++a) { lo[a] = (int32_t *)lp; lp += cap * 4; lleft -= (uint32_t)(cap * 4); }
        w.st[0].S = { lo[0], hi[0] }; w.st[0].F = { lo[1], hi[1] };
        w.st[1].S = { lo[2], hi[2] }; w.st[1].F = { lo[3], hi[3] };
        w.stE = { lo[4], hi[4] };
        w.st[0].col = -1; w.st[1].col = -1;
    }
    if (lleft_seed_end < lleft) { lp = lp_seed_end; lleft = lleft_seed_end; }     // past the larger side of the overlay
    for (int s = 0; s < 2; ++s) w.pk[s] = (uint32_t *)take_fast(((L + 15) / 16 + 2) * 4);
    for (int s = 0; s < 2; ++s) w.psum[s] = (int32_t *)take_fast((L + 1) * 4);
    for (int s = 0; s < 2; ++s) w.seeds[s] = (DevSeed *)take_rz((uint64_t)lim.max_seeds * sizeof(DevSeed));
    for (int s = 0; s < 2; ++s) w.alive[s] = take_rz(lim.max_seeds);
    w.alt = (uint32_t *)take_rz((uint64_t)lim.max_alt * 4);
    p = (uint8_t *)(((uint64_t)p + 15) & ~15ull);                // cell records are written with 16-byte stores
    w.cells = (int32_t *)take_rz((uint64_t)lim.cell_words * 4);
    p = (uint8_t *)(((uint64_t)p + 63) & ~63ull);                // a chain-format slot is exactly two 64-byte lines
    w.cols = (ColSlot *)take_rz((uint64_t)lim.max_columns * sizeof(ColSlot));
    p = (uint8_t *)(((uint64_t)p + 63) & ~63ull);
    w.cols_s16 = (int16_t *)take_rz((uint64_t)lim.max_columns * FWS * 2);
    w.queue = (uint64_t *)take_rz((uint64_t)lim.max_columns * 8);
    w.next_nodes = (uint64_t *)take_rz((uint64_t)lim.max_columns * 8);
    w.tips = (uint32_t *)take_rz((uint64_t)lim.max_columns * 4);
    w.prev_starts = (uint32_t *)take_rz(((uint64_t)lim.max_columns + 31) / 32 * 4);
    w.indices = (BtIndex *)take_rz((uint64_t)lim.max_columns * 2 * sizeof(BtIndex));
    w.rev_ops = (uint32_t *)take_rz((uint64_t)lim.max_path * 4);
    w.rev_nodes = (uint32_t *)take_rz((uint64_t)lim.max_path * 4);
    w.rev_seq = take_rz(lim.max_path);
    w.gen_store = (uint32_t *)take(16);
    for (int s = 0; s < 2; ++s) {
        p = (uint8_t *)(((uint64_t)p + 63) & ~63ull);              // level 0 of the hash table starts a line
        w.ext[s].conv.tab = (uint64_t *)take(2ull * lim.hash_size * 8);          // conv_tab_slots()
        take(((uint64_t)lim.max_columns + lim.max_path) * 16);                   // ConvRecs (conv_recs())
        w.ext[s].conv.pool = (int32_t *)take((uint64_t)lim.conv_pool_words * 4);
    }
    // (the label part of the arena follows the alignment buffers of ALL n_aln alignments, also in builds that use fewer)
    uint8_t *aln_end = p + (uint64_t)lim.n_aln * (2 * align8((uint64_t)lim.max_path * 4) + align8(lim.max_path));
#if MGX_WITH_LABELS
    w.aln = (DevAln *)aln_end;
    aln_end += align8((uint64_t)lim.n_aln * sizeof(DevAln));
#endif
    for (int a = 0; (kWithLabels || a < N_ALN) && a < (int)lim.n_aln; ++a) {        // (label builds: the records live in the arena, n_aln of them)
        w.aln[a].nodes = (uint32_t *)take((uint64_t)lim.max_path * 4);
        w.aln[a].cigar = (uint32_t *)take((uint64_t)lim.max_path * 4);
        w.aln[a].seq = take(lim.max_path);
    }
#if MGX_WITH_LABELS
    p = aln_end;
    if (lim.lab_words) {
        w.lab = (uint32_t *)take_rz((uint64_t)lim.lab_words * 4);
        w.col_lab = (uint32_t *)take_rz((uint64_t)lim.max_columns * 4);
        for (int s = 0; s < 2; ++s) w.seed_lab[s] = (uint32_t *)take_rz((uint64_t)lim.max_seeds * 4);
        w.agg = (uint32_t *)take_rz(lab_agg_words(lim) * 4);
    } else {
        w.lab = nullptr; w.col_lab = nullptr; w.seed_lab[0] = w.seed_lab[1] = nullptr; w.agg = nullptr;
    }
#endif
}

// LDS bytes that hold every "fast" array of carve() for a given Lmax
MGX_HD uint32_t fast_lds_bytes(uint32_t Lmax) {
    uint64_t L = Lmax, Lp = align8(L + 8);
    uint64_t persistent = 2 * Lp + 2 * align8((L + 1) * 4) + 2 * align8(((L + 15) / 16 + 2) * 4);
    uint64_t seeding = 3 * align8((L + 1) * 2) + align8(L + 1) + 3 * align8((L + 1) * 4) + align8(sizeof(SdustScratch))
                       + 4 * align8(((L + 63) / 64 + 1) * 8) + align8((L + 8) * 8) + align8(L + 8);
    uint64_t staging = 6 * align8((L + 16) * 4);
    return (uint32_t)(persistent + (seeding > staging ? seeding : staging));
}

MGX_DEV int32_t score_of(const AlignParams &P, uint8_t graph_char, uint8_t query_char) {
    return P.score_matrix[(uint32_t)(graph_char & 127) * 128 + (query_char & 127)];
}

// fill the 6 x 128 score rows ($, A, C, G, T, '\\0'); dst may be LDS
MGX_DEV void load_score_rows(const AlignParams &P, int8_t *dst) {
    for (int32_t base = 0; base < 6 * 128; base += WAVE) {
        FOR_LANES(l) {
            int32_t x = base + l;
            if (x < 6 * 128) {
                uint32_t code = (uint32_t)(x >> 7);
                uint8_t row = code != 5 ? decode_code(code) : 0;
                dst[x] = P.score_matrix[(uint32_t)(row & 127) * 128 + (x & 127)];
            }
        }
    }
    wave_sync();
}

// profile_score_[encode(c)][start + trim + j] (aligner_extender_methods.cpp:38-59): column char vs
// the query character one before absolute window position; 0 in the first cell and the padding
MGX_DEV int32_t profile_at(const Wave &w, const uint8_t *q, int32_t L, uint8_t c, int32_t abs_pos) {
    if (abs_pos < 1 || abs_pos > L) return 0;
    return MGX_SM_ROWS(w)[encode_char(c) * 128 + (q[abs_pos - 1] & 127)];      // row 5 = score_matrix['\0']
}

MGX_DEV uint8_t profile_op_at(const uint8_t *q, int32_t L, uint8_t c, int32_t abs_pos) {
    if (abs_pos < 1 || abs_pos > L) return OP_CLIPPED;
    uint32_t code = encode_char(c);
    uint8_t row = code != 5 ? decode_code(code) : 0;
    return char_to_op(row, q[abs_pos - 1]);
}

// ------------------------------------------------------------------------------------------------
// sdust — lh3's symmetric DUST restated (github.com/lh3/sdust, sdust.c; called at
// A/aligner_seeder_methods.cpp:22-29 with T = 20, W = 64).  Wave-uniform scalar code; `sd` is a
// 2048-word scratch area.  Returns whether any interval is masked.
// ------------------------------------------------------------------------------------------------
// 64-entry tables of sdust in memory (any lane count) ...
struct MemTab64 {
    int16_t *p;
    MGX_DEV int32_t get(int i) const { return p[i]; }
    MGX_DEV void add(int i, int32_t d) { p[i] = (int16_t)(p[i] + d); }
    MGX_DEV void set(int i, int32_t x) { p[i] = (int16_t)x; }
    MGX_DEV void fill(int32_t x) { for (int i = 0; i < 64; ++i) p[i] = (int16_t)x; }
};
// ... or, for a 64-lane wave, one entry per lane in a register (RegTab64 of wave.hpp)
MGX_DEV void tab_copy(MemTab64 &dst, const MemTab64 &src) { for (int i = 0; i < 64; ++i) dst.p[i] = src.p[i]; }
#if MGX_HAS_REGTAB
MGX_DEV void tab_copy(RegTab64 &dst, const RegTab64 &src) { dst = src; }
#endif
template <class Tab>
MGX_DEV bool sdust_core(const uint8_t *s, int32_t l_seq, SdustScratch *sd, Tab cv, Tab cw, Tab c2, Tab wq) {
    constexpr int T = 20, W = 64, WLEN = 3, WMSK = 63;
    int16_t *Ps = sd->Ps, *Pf = sd->Pf, *Pr = sd->Pr, *Pl = sd->Pl;
    cv.fill(0); cw.fill(0);
    int wfront = 0, wcount = 0;
    int Pn = 0;
    bool have_res = false;
    int res_f = 0;
    int rv = 0, rw = 0, Lw = 0;
    int l = 0;
    uint32_t t = 0;
    auto wat = [&](int idx) { return wq.get((wfront + idx) & 63); };
    auto save_masked = [&](int start) {
        if (Pn == 0 || Ps[Pn - 1] >= start) return;
        int ps = Ps[Pn - 1], pf = Pf[Pn - 1];
        bool saved = false;
        if (have_res) {
            if (ps <= res_f) { saved = true; res_f = res_f > pf ? res_f : pf; }
        }
        if (!saved) { have_res = true; res_f = pf; }
        int i;
        for (i = Pn - 1; i >= 0 && Ps[i] < start; --i) {}
        Pn = i + 1;
    };
    for (int i = 0; i <= l_seq; ++i) {
        int b = 4;
        if (i < l_seq) {
            uint32_t code = encode_char(s[i]);
            b = code >= 1 && code <= 4 ? (int)code - 1 : 4;
        }
        if (b < 4) {
            ++l;
            t = (t << 2 | (uint32_t)b) & WMSK;
            if (l >= WLEN) {
                int start = (l - W > 0 ? l - W : 0) + (i + 1 - l);
                save_masked(start);
                // shift_window
                if (wcount >= W - WLEN + 1) {
                    int sv = wq.get(wfront & 63);
                    wfront = (wfront + 1) & 63;
                    --wcount;
                    cw.add(sv, -1); rw -= cw.get(sv);
                    if (Lw > wcount) { --Lw; cv.add(sv, -1); rv -= cv.get(sv); }
                }
                wq.set((wfront + wcount) & 63, (int32_t)t);
                ++wcount;
                ++Lw;
                rw += cw.get((int)t); cw.add((int)t, 1);
                rv += cv.get((int)t); cv.add((int)t, 1);
                if (cv.get((int)t) * 10 > T << 1) {
                    int sv;
                    do {
                        sv = wat(wcount - Lw);
                        cv.add(sv, -1); rv -= cv.get(sv);
                        --Lw;
                    } while (sv != (int)t);
                }
                if (rw * 10 > Lw * T) {
                    // find_perfect
                    tab_copy(c2, cv);
                    int r = rv, max_r = 0, max_l = 0;
                    for (int ii = wcount - Lw - 1; ii >= 0; --ii) {
                        int tt = wat(ii);
                        r += c2.get(tt); c2.add(tt, 1);
                        int new_r = r, new_l = wcount - ii - 1;
                        if (new_r * 10 > T * new_l) {
                            int j;
                            for (j = 0; j < Pn && Ps[j] >= ii + start; ++j) {
                                if (max_r == 0 || Pr[j] * max_l > max_r * Pl[j]) { max_r = Pr[j]; max_l = Pl[j]; }
                            }
                            if (max_r == 0 || new_r * max_l >= max_r * new_l) {
                                max_r = new_r; max_l = new_l;
                                if (Pn < 63) {
                                    for (int m = Pn; m > j; --m) { Ps[m] = Ps[m - 1]; Pf[m] = Pf[m - 1]; Pr[m] = Pr[m - 1]; Pl[m] = Pl[m - 1]; }
                                    ++Pn;
                                    Ps[j] = (int16_t)(ii + start); Pf[j] = (int16_t)(wcount + (WLEN - 1) + start);
                                    Pr[j] = (int16_t)new_r; Pl[j] = (int16_t)new_l;
                                }
                            }
                        }
                    }
                }
            }
        } else {
            int start = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);
            while (Pn) save_masked(start++);
            l = 0; t = 0;
        }
    }
    return have_res;
}

MGX_NI_G1 bool is_low_complexity(const uint8_t *s, int32_t l_seq, SdustScratch *sd) {
#if MGX_HAS_REGTAB
    RegTab64 cv, cw, c2, wq;
    wq.fill(0); c2.fill(0);
    return sdust_core(s, l_seq, sd, cv, cw, c2, wq);
#else
    MemTab64 cv = { sd->cv }, cw = { sd->cw }, c2 = { sd->c2 }, wq = { sd->wqw };
    return sdust_core(s, l_seq, sd, cv, cw, c2, wq);
#endif
}

// Lane-parallel, conservative companion of is_low_complexity(): returns false only if NO interval of at most 62
// consecutive triplets anywhere in the strand reaches a DUST score above T (sum over triplets of c(c-1)/2 pairs,
// times 10, greater than T times (number of triplets - 1): the test find_perfect applies, sdust.c).  Every interval
// sdust can mask — on the whole strand or on any window of it — is such an interval, so `false` proves that no
// window is low-complexity; `true` only means "run the exact algorithm".  One lane per end position instead of
// sdust's serial state machine: per position a 61-bit mask of the following positions with the same triplet, then
// r(start, end) accumulates popcounts while the start walks back.  Strands with a non-ACGT character are not judged
// here (`true`).
MGX_DEV bool maybe_low_complexity(Wave &w, int s, uint8_t *tc_at = nullptr, uint64_t *eq_at = nullptr) {
    constexpr int32_t T = 20, SPAN = 61;
    const int32_t L = w.L;
    const uint8_t *q = w.q[s];
    uint8_t *tc = tc_at ? tc_at : w.dust_t;
    uint64_t *eq = eq_at ? eq_at : w.dust_eq;
    uint64_t invalid = 0;
    for (int32_t base = 0; base < L; base += WAVE) {
        LV<bool> bad;
        FOR_LANES(l) {
            int32_t i = base + l;
            bad[l] = false;
            if (i < L) {
                uint8_t t = 255;
                if (i >= 2) {
                    uint32_t a = encode_char(q[i - 2]), b = encode_char(q[i - 1]), c = encode_char(q[i]);
                    if (a >= 1 && a <= 4 && b >= 1 && b <= 4 && c >= 1 && c <= 4) t = (uint8_t)(((a - 1) << 4) | ((b - 1) << 2) | (c - 1));
                    else bad[l] = true;
                }
                tc[i] = t;
            }
        }
        invalid |= wave_ballot(bad);
    }
    wave_sync();
    // sdust keeps its triplet window across an invalid character (only the run length restarts), so intervals can
    // span it with fewer triplets than positions: leave such strands to the exact algorithm
    if (invalid) return true;
    for (int32_t base = 0; base < L; base += WAVE) {
        FOR_LANES(l) {
            int32_t p = base + l;
            if (p < L) {
                uint64_t m = 0;
                const uint8_t t = tc[p];
                if (t != 255)
                    for (int32_t d = 1; d <= SPAN && p + d < L; ++d)
                        if (tc[p + d] == t) m |= 1ull << (d - 1);
                eq[p] = m;
            }
        }
    }
    wave_sync();
    uint64_t any = 0;
    for (int32_t base = 0; base < L; base += WAVE) {
        LV<bool> hit;
        FOR_LANES(l) {
            int32_t e = base + l;
            bool h = false;
            if (e < L && tc[e] != 255) {
                // eight masks per round, fetched together (the test of each d depends on the running sum, and with a break
                // after every load the 61 loads of a lane were issued one after the other)
                int32_t r = 0;
                for (int32_t d0 = 1; d0 <= SPAN && e - d0 >= 0 && !h; d0 += 8) {
                    uint64_t v[8];
                    for (int t = 0; t < 8; ++t) { const int32_t d = d0 + t; v[t] = (d <= SPAN && e - d >= 0) ? eq[e - d] : 0ull; }
                    for (int t = 0; t < 8; ++t) {
                        const int32_t d = d0 + t;
                        if (d <= SPAN && e - d >= 0) {
                            r += popc64(v[t] & ((1ull << d) - 1));    // pairs (e - d, y) with y in (e - d, e]
                            h |= r * 10 > T * d;
                        }
                    }
                }
            }
            hit[l] = h;
        }
        any |= wave_ballot(hit);
    }
    wave_sync();
    return any != 0;
}

// Window test with an exact shortcut.  A window is flagged iff it contains an interval (<= 64 words) whose
// DUST score exceeds T; any such interval also exists in the whole strand, so if sdust on the whole strand
// masks nothing, no window can be flagged.  Only strands with a masked region pay the per-window runs.
MGX_DEV bool window_low_complexity(Wave &w, int s, int32_t begin, int32_t len) {
    if (w.lc_any[s] < 0) {
        // The conservative test is mirror-symmetric — an interval of the reverse complement holds the reverse complements
        // of the same triplets, so the same pairs of equal triplets at the same distances — hence one evaluation serves both
        // strands of a read (the exact algorithm below still runs per strand).
        if (w.lc_maybe < 0) w.lc_maybe = maybe_low_complexity(w, s) ? 1 : 0;
        w.lc_any[s] = (w.lc_maybe && is_low_complexity(w.q[s], w.L, w.sd)) ? 1 : 0;
    }
    if (!w.lc_any[s]) return false;
    return is_low_complexity(w.q[s] + begin, len, w.sd);
}

// ------------------------------------------------------------------------------------------------
// query preparation: AlignmentResults ctor (A/alignment.cpp:1348-1372) + partial sums
// ------------------------------------------------------------------------------------------------
// Linear partial sums: if every character of a strand is in ACGT and score(A,A) == score(C,C) == score(G,G) ==
// score(T,T) == m > 0, then partial_sums_[x] == (L - x) * m exactly and the extender's per-cell test needs no table.
MGX_DEV void detect_linear_psum(Wave &w) {
    const AlignParams &P = MGX_PARAMS_OF(w);
    const int32_t L = w.L;
    const int32_t mA = score_of(P, 'A', 'A');
    const bool same = mA > 0 && score_of(P, 'C', 'C') == mA && score_of(P, 'G', 'G') == mA && score_of(P, 'T', 'T') == mA;
    for (int s = 0; s < 2; ++s) {
        uint64_t other = 0;
        for (int32_t base = 0; same && base < L; base += WAVE) {
            LV<bool> bad;
            FOR_LANES(l) {
                int32_t j = base + l;
                bad[l] = false;
                if (j < L) { uint8_t c = w.q[s][j]; bad[l] = !(c == 'A' || c == 'C' || c == 'G' || c == 'T'); }
            }
            other |= wave_ballot(bad);
        }
        w.psum_lin[s] = (same && !other) ? mA : 0;
    }
}

// word j of read r's 2-bit packed strands (k_pack_reads, graph_build.hpp): 32 codes per word, every read starts a word
MGX_HD uint64_t packed_word_begin(uint64_t byte_offset, uint64_t read) { return (byte_offset >> 5) + read; }

MGX_NI_G1 void prepare_query(Wave &w, const char *raw, bool for_seeding, bool for_extension, uint64_t read = 0) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    const int32_t L = w.L;
    for (int32_t base = 0; base < L; base += WAVE) {
        FOR_LANES(l) {
            int32_t j = base + l;
            if (j < L) {
                uint8_t c = (uint8_t)raw[j];
                uint8_t f = (c & 0x80) ? 127 : to_upper(c);
                w.q[0][j] = f;
                w.q[1][L - 1 - j] = complement_char(f);
            }
        }
    }
    wave_sync();
    if (for_extension) detect_linear_psum(w); else { w.psum_lin[0] = w.psum_lin[1] = 0; }
    // partial_sums_[i] = sum_{j >= i} score(q[j], q[j]); partial_sums_[L] = 0 (seed scores and the extender only)
    for (int s = 0; for_extension && s < 2; ++s) {
        if (w.psum_lin[s]) continue;                       // (L - x) * m: no table needed
        int32_t carry = 0;
        int32_t nchunks = (L + WAVE - 1) / WAVE;
        for (int32_t ch = nchunks - 1; ch >= 0; --ch) {
            // process positions [ch*64, ch*64+64) from high to low: lane l handles position top - l
            int32_t top = imin(L, (ch + 1) * WAVE) - 1;
            LV<int32_t> x;
            FOR_LANES(l) {
                int32_t j = top - l;
                x[l] = (j >= ch * WAVE) ? score_of(P, w.q[s][j], w.q[s][j]) : 0;
            }
            LV<int32_t> ex = wave_prefix_sum_excl(x);
            FOR_LANES(l) {
                int32_t j = top - l;
                if (j >= ch * WAVE) w.psum[s][j] = carry + ex[l] + x[l];
            }
            carry += wave_sum(x);
        }
        FOR_LANES(l) { if (l == 0) w.psum[s][L] = 0; }
    }
    // 2-bit packed strands for the suffix-range table keys (the seeder's lookups only): as k_pack_reads left them where the
    // batch was packed for the mapping kernel (same codes, same order: a 64-bit word of 32 codes is two of these words; an
    // invalid or missing position holds 0 there as here), else encoded here
    for (int s = 0; for_seeding && s < 2; ++s) {
        const int32_t nw = (L + 15) / 16 + 1;
        if (P.pkw[s] && P.ivw[s]) {
            const uint64_t wb = packed_word_begin((uint64_t)(raw - P.seqs), read);
            const int32_t n64 = (L + 31) >> 5;
            uint64_t bad_any = 0;
            for (int32_t base = 0; base < nw; base += WAVE) {
                LV<bool> bad;
                FOR_LANES(l) {
                    const int32_t wi = base + l;
                    bad[l] = false;
                    if (wi < nw) {
                        const int32_t j = wi >> 1;
                        uint32_t v = 0;
                        if (j < n64) {
                            const uint64_t q = gld(P.pkw[s] + wb + (uint64_t)j);
                            v = (wi & 1) ? (uint32_t)(q >> 32) : (uint32_t)q;
                            if (!(wi & 1)) bad[l] = gld(P.ivw[s] + wb + (uint64_t)j) != 0;
                        }
                        w.pk[s][wi] = v;
                    }
                }
                bad_any |= wave_ballot(bad);
            }
            w.inv_any[s] = bad_any ? 1 : 0;
            continue;
        }
        uint64_t bad_any = 0;
        for (int32_t base = 0; base < nw; base += WAVE) {
            LV<bool> bad;
            FOR_LANES(l) {
                int32_t wi = base + l;
                bad[l] = false;
                if (wi < nw) {
                    uint32_t v = 0;
                    for (int32_t j = 0; j < 16; ++j) {
                        int32_t p = wi * 16 + j;
                        if (p < L) {
                            uint32_t code = encode_char(w.q[s][p]);
                            if (code < 1 || code > 4) bad[l] = true;
                            else v |= (code - 1) << (2 * j);
                        }
                    }
                    w.pk[s][wi] = v;
                }
            }
            bad_any |= wave_ballot(bad);
        }
        w.inv_any[s] = bad_any ? 1 : 0;
    }
    wave_sync();
}

// first position >= from whose bit equals `val` in a bitmask over [0, n); n if none
MGX_DEV int32_t bits_next(const uint64_t *wds, int32_t n, int32_t from, bool val) {
    while (from < n) {
        uint64_t x = wds[from >> 6];
        if (!val) x = ~x;
        x >>= (from & 63);
        if (x) { int32_t p = from + ctz64(x); return p < n ? p : n; }
        from = (from | 63) + 1;
    }
    return n;
}
MGX_DEV bool bits_test(const uint64_t *wds, int32_t i) { return (wds[i >> 6] >> (i & 63)) & 1; }

// ------------------------------------------------------------------------------------------------
// seeding
// ------------------------------------------------------------------------------------------------
// bitmasks of the strand's k-mer positions: bm[0] = matched (node != 0), bm[1] = MEM stop (terminus of a
// matched k-mer, or unmatched); one coalesced pass over nodes[] instead of per-position scalar loads
MGX_DEV void kmer_masks(Wave &w, int s) {
    const AlignParams &P = MGX_PARAMS_OF(w);
    const int32_t n = w.n_kmers;
    const uint32_t *nodes = w.nodes[s];
    const int32_t nwords = (imax(n, 0) + 63) / 64 + 1;
    FOR_LANES(l) { if (l == 0) for (int32_t x = 0; x < nwords; ++x) { w.bm[0][x] = 0; w.bm[1][x] = 0; } }
    wave_sync();
    for (int32_t base = 0; base < n; base += WAVE) {
        LV<bool> mt, st;
        FOR_LANES(l) {
            int32_t i = base + l;
            mt[l] = false; st[l] = false;
            if (i < n) {
                uint32_t v = nodes[i];
                if (v) {
                    bool term = (i + 1 == n) || nodes[i + 1] == 0;
                    if (!term) {
                        // (PRIMARY graphs: ids above n are reverse complements, canon_graph.hpp)
                        const bool rc_id = kWithPrimary && P.cfg.canonical >= 2 && v > P.g.n;
                        const uint64_t u = rc_id ? v - P.g.n : v;
                        term = (P.g.terminus[(rc_id ? P.g.n_blocks : 0u) + (u >> 6)] >> (u & 63)) & 1;
                    }
                    mt[l] = true; st[l] = term;
                } else {
                    st[l] = true;
                }
            }
        }
        const uint64_t bm = wave_ballot(mt), bs = wave_ballot(st);
        FOR_LANES(l) { if (l == 0) { w.bm[0][base >> 6] |= bm << (base & 63); w.bm[1][base >> 6] |= bs << (base & 63); } }
        wave_sync();
    }
    w.ctr.bit_lines += (uint32_t)imax(n, 0);
}

MGX_DEV uint32_t num_exact_matching(const uint64_t *matched, int32_t n, int32_t k) {
    // A/aligner_seeder_methods.cpp:49-65, run by run over the matched-k-mer mask
    uint32_t num_matching = 0, last_match_count = 0;
    int32_t i = 0;
    while (i < n) {
        if (bits_test(matched, i)) {
            int32_t j = bits_next(matched, n, i + 1, false);
            num_matching += (uint32_t)k + (uint32_t)(j - i) - 1 - last_match_count;
            last_match_count = (uint32_t)k;
            i = j;
        } else {
            int32_t j = bits_next(matched, n, i + 1, true);
            uint32_t zeros = (uint32_t)(j - i);
            last_match_count = last_match_count > zeros ? last_match_count - zeros : 0;
            i = j;
        }
    }
    return num_matching;
}

MGX_DEV bool push_seed(Wave &w, int s, int32_t clip, int32_t len, int32_t offset, int32_t n_nodes, uint32_t node) {
    if (w.n_seeds[s] >= (int32_t)MGX_PARAMS_OF(w).lim.max_seeds) { w.status = ST_CAPACITY; return false; }
    DevSeed sd;
    sd.clipping = (uint16_t)clip; sd.length = (uint16_t)len; sd.offset = (uint16_t)offset;
    sd.n_nodes = (uint16_t)n_nodes; sd.node = node;
    w.seeds[s][w.n_seeds[s]] = sd;
    w.alive[s][w.n_seeds[s]] = 1;
    ++w.n_seeds[s];
    return true;
}

// MEMSeeder::get_seeds / ExactSeeder::get_seeds into w.seeds[s] (A/aligner_seeder_methods.cpp:67-93,360-424)
// MANY: one seed per matched k-mer (max_seed_length <= k; see make_seeder)
template <bool MANY>
MGX_NI_G2 void base_seeds(Wave &w, int s) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    const DevConfig &cfg = P.cfg;
    const int32_t k = (int32_t)P.g.k, L = w.L, n = w.n_kmers;
    const uint32_t *nodes = w.nodes[s];
    w.n_seeds[s] = 0;
    if ((double)w.num_matching[s] < cfg.min_exact_match * (double)L) return;
    if (MANY) {
        // ExactSeeder::get_seeds
        if (cfg.max_seed_length < (uint32_t)k) return;
        // One seed per matched k-mer (label-aware alignment clamps max_seed_length to k: ~120 seeds per strand of a 150-bp
        // read).  When no window of the strand can be low-complexity — the whole-strand verdict window_low_complexity()
        // caches — the seeds are written one k-mer per lane, in position order; else position by position.
        const int32_t i0 = bits_next(w.bm[0], n, 0, true);       // the first matched k-mer (kmer_masks)
        if (i0 >= n) return;
        bool plain = !cfg.seed_complexity_filter;
        if (!plain) { (void)window_low_complexity(w, s, i0, k); plain = w.lc_any[s] == 0; }
        if (plain) {
            const int32_t max_seeds = (int32_t)MGX_PARAMS_OF(w).lim.max_seeds;
            int32_t ns = 0;
            for (int32_t base = i0; base < n; base += WAVE) {
                LV<bool> has;
                LV<uint32_t> nd;
                FOR_LANES(l) { const int32_t i = base + l; nd[l] = i < n ? gld(nodes + i) : 0u; has[l] = nd[l] != 0; }
                const uint64_t mk = wave_ballot(has);
                const int32_t cnt = popc64(mk);
                if (ns + cnt > max_seeds) { w.status = ST_CAPACITY; w.n_seeds[s] = ns; return; }
                FOR_LANES(l) {
                    if (has[l]) {
                        const int32_t pos = ns + popc64(mk & ((1ull << l) - 1));
                        DevSeed sd;
                        sd.clipping = (uint16_t)(base + l); sd.length = (uint16_t)k; sd.offset = 0; sd.n_nodes = 1; sd.node = nd[l];
                        w.seeds[s][pos] = sd;
                        w.alive[s][pos] = 1;
                    }
                }
                ns += cnt;
            }
            w.n_seeds[s] = ns;
            wave_sync();
            return;
        }
        for (int32_t i = 0; i < n; ++i) {
            if (nodes[i]) {
                if (!cfg.seed_complexity_filter || !window_low_complexity(w, s, i, k))
                    if (!push_seed(w, s, i, k, 0, 1, nodes[i])) return;
            }
        }
        return;
    }
    // UniMEMSeeder (seeder hpp:116-135) over the masks of kmer_masks(): a MEM runs from a matched k-mer to the
    // first stop position (terminus of a matched k-mer, inclusive, or an unmatched k-mer, exclusive)
    const uint64_t *matched = w.bm[0], *stop = w.bm[1];
    int32_t it = 0;
    while (it < n) {
        it = bits_next(matched, n, it, true);
        if (it >= n) break;
        int32_t next = bits_next(stop, n, it, true);
        if (next < n && bits_test(matched, next)) ++next;
        int32_t mem_length = (next - it) + k - 1;
        if ((uint32_t)mem_length >= cfg.min_seed_length)
            if (!push_seed(w, s, it, mem_length, 0, next - it, 0)) return;
        it = next;
    }
}

constexpr uint32_t DEFERRED_RANGE = 0xFFFFFFFFu;     // rfirst[] marker: match length known, range not fetched yet
// rfirst[] marker: the range holds ONE node and rlast[] is its last edge.  index_range returns (succ_last(rl), ru), both last
// edges of nodes (boss.hpp:756-763), so first == last means one node, and select_last(rank_last(first)) == first: neither
// the two ranks nor the select of dbg_succinct.cpp:349-375 need a memory access then (the common case).
constexpr uint32_t SINGLE_NODE = 0xFFFFFFFEu;

// BOSS::index_range (boss.hpp:720-764) for one lane: codes q[i .. i + len); returns matched length,
// *first = succ_last(rl), *last = ru
MGX_DEV int32_t index_range_lane(const Wave &w, int s, int32_t i, int32_t len, int32_t min_len,
                                 uint64_t *first, uint64_t *last, LineCtr &ctr) {
    const DevGraph &g = MGX_PARAMS_OF(w).g;
    const uint8_t *q = w.q[s] + i;
    *first = 0; *last = 0;
    if (len == 0) { *first = 1; *last = 1; return 0; }
    const bool clean = !w.inv_any[s];                            // whole strand is ACGT: no per-window check
    if (!clean)
        for (int32_t j = 0; j < len; ++j) if (encode_char(q[j]) == 5) return 0;
    uint64_t rl = 1, ru = 0;
    int32_t it = 1;
    bool have = false;
    if (g.prefix_len && (int32_t)g.prefix_len <= len) {
        // get_initial_range via the suffix-range table (boss.hpp:645-663)
        uint32_t key = 0;
        if (clean) {
            const uint32_t *pk = w.pk[s];
            const uint32_t wi = (uint32_t)i >> 4, sh = 2 * ((uint32_t)i & 15);
            const uint64_t two = ((uint64_t)pk[wi + 1] << 32) | pk[wi];
            key = (uint32_t)(two >> sh) & ((1u << (2 * g.prefix_len)) - 1);
        } else {
            for (uint32_t j = 0; j < g.prefix_len; ++j) key |= (encode_char(q[j]) - 1) << (2 * j);
        }
        prefix_range(g, key, &rl, &ru, ctr);
        if (rl <= ru) { have = true; it = (int32_t)g.prefix_len; }
        else if (min_len > (int32_t)g.prefix_len) return 0;      // the match is shorter than prefix_len < min_len:
                                                                 // the caller discards it (dbg_succinct.cpp:346-347)
    }
    if (!have) {                                                 // "start search from scratch" (boss.hpp:739-754)
        initial_range(g, encode_char(q[0]), &rl, &ru);
        if (rl > ru) return 0;
        it = 1;
    }
    for (; it < len; ++it)
        if (!tighten_range(g, &rl, &ru, encode_char(q[it]), ctr)) break;
    *first = succ_last(g, rl, ctr);
    *last = ru;
    return it;
}

// per-column timers of extend() (mgx_stats.extend_cycles): a profiling aid that costs an s_memtime + lgkmcnt
// drain per reading, measured to be free on gfx950 (769 vs 773 ms per 2 M reads); -DMGX_XTIMERS=0 removes them
#ifndef MGX_XTIMERS
#define MGX_XTIMERS 1
#endif
MGX_DEV uint64_t xclock() {
#if MGX_XTIMERS
    return cycle_clock();
#else
    return 0;
#endif
}

#ifdef MGX_SEED_PROBE
#define SEED_T(slot, t0) { uint64_t t_ = cycle_clock(); w.xcyc[slot] += t_ - t0; t0 = t_; }
#else
#define SEED_T(slot, t0)
#endif
// SuffixSeeder::generate_seeds on a CanonicalDBG (A/aligner_seeder_methods.cpp:251-314): sub-k matches of the query's
// reverse complement.  index_range matches query prefix -> node suffix; the nodes wanted are those whose PREFIX spells the
// match (suffix_to_prefix :95-139: a depth-first walk that tightens the range by every possible next character until the
// label is complete), reported as their reverse complements at query position j = L - i - match length.  Runs after the
// forward-strand positions with the same per-position bookkeeping (append_suffix_seed :195-213): msl / pos_start / pos_cnt
// over w.alt, which keeps the nodes of one position contiguous — entries added to a position that already holds nodes from
// the first phase are joined by moving the earlier ones to the end of w.alt.
// The reference's walk uses a LIFO stack of ranges: a node's children are tightened for c = 1..4, complete ones are
// reported at once and the others pushed, so incomplete children are visited in the order 4..1.  Tightening has no side
// effects, so here each level keeps only its parent range and the next character to try (w.indices as scratch).
MGX_DEV void primary_rc_suffix_seeds(Wave &w, int s, uint32_t alt_n) {
    const AlignParams &P = MGX_PARAMS_OF(w);
    const DevConfig &cfg = P.cfg;
    const DevGraph &g = P.g;
    const int32_t k = (int32_t)g.k, L = w.L, boss_k = k - 1;
    const int32_t msl0 = (int32_t)cfg.min_seed_length;
    const int32_t nslots = L - msl0 + 1;
    const int so = 1 - s;                                       // the other strand's text is this one's reverse complement
    uint64_t *lvl = (uint64_t *)w.indices;                      // per level: rl, ru, next character
    auto append = [&](int32_t jj, uint32_t node, int32_t sl) -> bool {
        const uint32_t cap = MGX_PARAMS_OF(w).lim.max_alt;
        if (sl > (int32_t)w.msl[jj]) { w.pos_cnt[jj] = 0; w.pos_full[jj] = 0; }
        w.msl[jj] = (uint16_t)sl;
        if (w.pos_cnt[jj] == 0) {
            w.pos_start[jj] = alt_n; w.pos_full[jj] = 0;
        } else if (w.pos_start[jj] + (uint32_t)w.pos_cnt[jj] != alt_n) {
            const uint32_t c0 = (uint32_t)w.pos_cnt[jj], from = w.pos_start[jj];
            if (alt_n + c0 >= cap) { w.status = ST_CAPACITY; return false; }
            for (uint32_t x = 0; x < c0; ++x) w.alt[alt_n + x] = w.alt[from + x];
            w.pos_start[jj] = alt_n;
            alt_n += c0;
        }
        if (alt_n >= cap) { w.status = ST_CAPACITY; return false; }
        w.alt[alt_n++] = node;
        ++w.pos_cnt[jj];
        w.bm[3][jj >> 6] |= 1ull << (jj & 63);
        for (++jj; jj < nslots && sl > (int32_t)w.msl[jj]; ++jj) {
            w.msl[jj] = (uint16_t)sl--;
            w.pos_cnt[jj] = 0;
            w.pos_full[jj] = 0;
        }
        return true;
    };
    // The look-up length of position i depends on min_seed_length[] as updated so far (:262-268), which only ever grows during
    // this phase.  So all look-ups run first, one position per lane, at the lengths the bounds allow now; the sequential pass
    // below takes a result as it is unless its match is longer than the (possibly smaller) length allowed by then — only that
    // case changes the range — and redoes just those.  (w.ml / w.rfirst / w.rlast are free after the forward phase.)
    auto allowed_length = [&](int32_t i) -> int32_t {            // 0: the position is skipped
        int32_t max_len = (int32_t)imin<uint32_t>(imin<uint32_t>(cfg.max_seed_length, (uint32_t)boss_k), (uint32_t)(L - i));
        int32_t j_min = L - i - max_len;
        const int32_t j_max = L - i - msl0;
        while (j_min <= j_max && (int32_t)w.msl[j_min] > max_len) { ++j_min; --max_len; }
        return j_min > j_max ? 0 : max_len;
    };
    for (int32_t base = 0; base + msl0 <= L; base += WAVE) {
        LV<int32_t> nr, ns;
        FOR_LANES(l) {
            LineCtr lc = { 0, 0, 0 };
            const int32_t i = base + l;
            if (i + msl0 <= L) {
                const int32_t max_len = allowed_length(i);
                uint16_t m = 0;
                uint32_t f = 0, la = 0;
                if (max_len) {
                    uint64_t first, last;
                    const int32_t sl = index_range_lane(w, so, i, max_len, msl0, &first, &last, lc);
                    if (sl >= msl0) { m = (uint16_t)sl; f = (uint32_t)first; la = (uint32_t)last; }
                }
                w.ml[i] = m; w.rfirst[i] = f; w.rlast[i] = la;
            }
            nr[l] = (int32_t)(lc.rank_lines + lc.bit_lines); ns[l] = (int32_t)lc.select_lines;
        }
        w.ctr.rank_lines += (uint32_t)wave_sum(nr);
        w.ctr.select_lines += (uint32_t)wave_sum(ns);
    }
    wave_sync();
    for (int32_t i = 0; i + msl0 <= L; ++i) {
        if (!w.ml[i]) continue;                                  // skipped or shorter than min_seed_length then: still so now
        const int32_t max_len = allowed_length(i);
        if (!max_len) continue;
        uint64_t first = w.rfirst[i], last = w.rlast[i];
        int32_t sl = w.ml[i];
        if (sl > max_len) sl = index_range_lane(w, so, i, max_len, msl0, &first, &last, w.ctr);
        if (sl < msl0) continue;
        const int32_t j = L - i - sl;
        if (sl < (int32_t)w.msl[j]) continue;
        if (cfg.seed_complexity_filter && window_low_complexity(w, s, j, sl)) continue;
        auto report = [&](uint64_t a, uint64_t b) -> bool {
            for (uint64_t e = a; e <= b; ++e) {
                if (!in_graph(g, e)) continue;
                // CanonicalDBG::reverse_complement(node) (:515-549)
                uint32_t id = (uint32_t)(e + g.n);
                if (!(k & 1)) {
                    if (cfg.canonical == 3) {
                        ++w.ctr.bit_lines;
                        if ((gld(primary_tables(g).pal + (e >> 6)) >> (e & 63)) & 1) id = (uint32_t)e;
                    } else {
                        const Spell sp = base_spelling(g, e, w.ctr);
                        if (kmer_is_palindrome(sp, k)) id = (uint32_t)e;
                    }
                }
                if (!append(j, id, sl)) return false;
            }
            return true;
        };
        const uint64_t rl0 = pred_last(g, first - 1, w.ctr) + 1;
        if (sl == boss_k) { if (!report(rl0, last)) return; continue; }
        int32_t d = 0;
        lvl[0] = rl0; lvl[1] = last; lvl[2] = (sl + 1 == boss_k) ? 1 : 4;
        while (d >= 0) {
            const bool leaves = sl + d + 1 == boss_k;
            const uint32_t c = (uint32_t)lvl[3 * d + 2];
            if (leaves ? c > 4 : c < 1) { --d; continue; }
            lvl[3 * d + 2] = leaves ? c + 1 : c - 1;
            uint64_t a = lvl[3 * d], b = lvl[3 * d + 1];
            if (!tighten_range(g, &a, &b, c, w.ctr)) continue;
            if (leaves) { if (!report(a, b)) return; }
            else { ++d; lvl[3 * d] = a; lvl[3 * d + 1] = b; lvl[3 * d + 2] = (sl + d + 1 == boss_k) ? 1 : 4; }
        }
    }
}

// A strand the seeder has nothing to say about — the usual fate of one of a read's two strands: no k-mer of it is in the graph
// (no MEM, num_matching 0) and no position can report a sub-k seed: k_map's index() matched fewer than min_seed_length characters
// at every k-mer position (its match lengths), and the walks of the read-tail positions, made here side by side, end below it
// too.  make_seeder() would set up its tables, look the same things up and report no seed; this answers in one pass.  `false`
// whenever anything is not known for certain (a match length k_map did not record, a position that does reach min_seed_length,
// PRIMARY graphs, whose wrapper also seeds from the other strand): the full seeder runs.
MGX_NI_G2 bool strand_without_seeds(Wave &w, int s) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    const DevConfig &cfg = P.cfg;
    const DevGraph &g = P.g;
    const int32_t k = (int32_t)g.k, L = w.L;
    if (cfg.canonical != 0 || !w.mlen[s] || w.n_kmers <= 0) return false;
    {
        // no matched k-mer (kmer_masks' first mask, without its terminus look-ups)
        const uint32_t *nodes = w.nodes[s];
        bool any = false;
        for (int32_t base = 0; base < w.n_kmers; base += WAVE) {
            LV<bool> mt;
            FOR_LANES(l) { const int32_t i = base + l; mt[l] = i < w.n_kmers && nodes[i] != 0; }
            any |= wave_ballot(mt) != 0;
        }
        if (any) return false;
    }
    const int32_t msl0 = (int32_t)cfg.min_seed_length;
    if ((uint32_t)L < cfg.min_seed_length || msl0 >= k) return true;            // (no sub-k seeding: base seeds only, and there are none)
    const int32_t nslots = L - msl0 + 1;
    bool maybe = false;
    for (int32_t base = 0; base < nslots; base += WAVE) {
        LV<bool> nd;
        LV<int32_t> nr, ns;
        FOR_LANES(l) {
            LineCtr lc = { 0, 0, 0 };
            const int32_t i = base + l;
            nd[l] = false;
            if (i < nslots) {
                const int32_t max_len = (int32_t)imin<uint32_t>(imin<uint32_t>(cfg.max_seed_length, (uint32_t)(k - 1)), (uint32_t)(L - i));
                if (max_len >= msl0) {
                    if (i < w.n_kmers && max_len == k - 1) {
                        const uint32_t ml = w.mlen[s][i];
                        if (ml == MLEN_LT_PREFIX) nd[l] = msl0 <= (int32_t)g.prefix_len;
                        else if (ml < MLEN_TAIL) nd[l] = (int32_t)ml >= msl0;
                        else nd[l] = true;                                      // not recorded
                    } else {
                        uint64_t first, last;
                        const int32_t m = index_range_lane(w, s, i, max_len, msl0, &first, &last, lc);
                        nd[l] = m >= msl0 && first && first <= g.n;
                    }
                }
            }
            nr[l] = (int32_t)(lc.rank_lines + lc.bit_lines); ns[l] = (int32_t)lc.select_lines;
        }
        w.ctr.rank_lines += (uint32_t)wave_sum(nr);
        w.ctr.select_lines += (uint32_t)wave_sum(ns);
        maybe |= wave_ballot(nd) != 0;
    }
    return !maybe;
}

// SuffixSeeder<UniMEMSeeder> ctor + generate_seeds (A/aligner_seeder_methods.cpp:153-358; the CanonicalDBG part above)
// Two instantiations: MANY = one seed per matched k-mer (max_seed_length <= k: label-aware alignment, ~120 base seeds per strand
// of a short read), whose bookkeeping runs one seed / one slot per lane; else a handful of MEMs, seed by seed.  (As run-time
// branches of one function the lane-parallel blocks cost the few-seed kernel 3 %: profiles/r04_ab4_seed_agg.txt.)
template <bool MANY>
MGX_NI_G2 void make_seeder(Wave &w, int s) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    const DevConfig &cfg = P.cfg;
    const DevGraph &g = P.g;
    const int32_t k = (int32_t)g.k, L = w.L;
#ifdef MGX_SEED_PROBE
    uint64_t tp = cycle_clock();
#endif
    kmer_masks(w, s);
    w.num_matching[s] = num_exact_matching(w.bm[0], w.n_kmers, k);
    SEED_T(0, tp)
    w.n_seeds[s] = 0;
    if ((uint32_t)L < cfg.min_seed_length) return;
    if (cfg.seed_complexity_filter && w.lc_maybe < 0 && cfg.min_seed_length < (uint32_t)k) {
        // The DUST pre-filter of the read (window_low_complexity), evaluated HERE instead of at the first position that asks:
        // nearly every read has such a position (the read tail), and here the per-position tables below are not in use yet, so
        // its two arrays — 61 reads of each entry per lane — can overlay them where they live, which is LDS; at their own
        // place behind those tables they were in the arena, and the filter a fifth of this kernel's time.
        const uint64_t Lm = MGX_PARAMS_OF(w).lim.Lmax;           // (carve() sized the tables for the batch's longest read)
        uint8_t *lo = (uint8_t *)w.msl;
        const uint64_t need = align8((uint64_t)(L + 8) * 8) + align8((uint64_t)L + 8);
        // the run of tables that lie back to back from msl on (carve() places them in this order; the first ones are in LDS)
        const uint8_t *at[7] = { (const uint8_t *)w.msl, (const uint8_t *)w.pos_cnt, (const uint8_t *)w.ml, w.pos_full,
                                 (const uint8_t *)w.pos_start, (const uint8_t *)w.rfirst, (const uint8_t *)w.rlast };
        const uint64_t sz[7] = { align8((Lm + 1) * 2), align8((Lm + 1) * 2), align8((Lm + 1) * 2), align8(Lm + 1),
                                 align8((Lm + 1) * 4), align8((Lm + 1) * 4), align8((Lm + 1) * 4) };
        uint64_t run = 0;
        for (int t = 0; t < 7 && at[t] == lo + run; ++t) run += sz[t];
        const bool overlay = run >= need && ((uint64_t)lo & 7) == 0;
        w.lc_maybe = (overlay ? maybe_low_complexity(w, s, lo + align8((uint64_t)(L + 8) * 8), (uint64_t *)lo)
                              : maybe_low_complexity(w, s)) ? 1 : 0;
    }
    if (cfg.min_seed_length >= (uint32_t)k) { base_seeds<MANY>(w, s); return; }

    const int32_t msl0 = (int32_t)cfg.min_seed_length;
    const int32_t nslots = L - msl0 + 1;
    base_seeds<MANY>(w, s);
    SEED_T(1, tp)
    if (w.status != ST_OK) return;
    const int32_t n_base = w.n_seeds[s];
    // min_seed_length[] and per-position seed lists
    for (int32_t base = 0; base < nslots; base += WAVE) {
        FOR_LANES(l) {
            int32_t i = base + l;
            if (i < nslots) { w.msl[i] = (uint16_t)msl0; w.pos_cnt[i] = 0; w.pos_full[i] = 0; w.pos_start[i] = 0; }
        }
    }
    {
        const int32_t nwords = (nslots + 63) / 64 + 1;
        FOR_LANES(l) { if (l == 0) for (int32_t x = 0; x < nwords; ++x) { w.bm[2][x] = 0; w.bm[3][x] = 0; } }
    }
    wave_sync();
    if (MANY) {
        // (one base seed per lane: seeds cover disjoint positions, and where their msl ranges touch they write the same value)
        for (int32_t base = 0; base < n_base; base += WAVE) {
            FOR_LANES(l) {
                const int32_t b = base + l;
                if (b < n_base) {
                    const DevSeed sd = w.seeds[s][b];
                    const int32_t i = sd.clipping;
                    for (int32_t j = 0; j < sd.n_nodes; ++j) w.msl[i + j] = (uint16_t)k;
                    if (i + sd.n_nodes < nslots) w.msl[i + sd.n_nodes] = (uint16_t)k;
                    w.pos_full[i] = 1;            // suffix_seeds[i] holds exactly this full seed
                    w.pos_cnt[i] = 1;
                    w.pos_start[i] = (uint32_t)b; // index of the full seed among the base seeds
                }
            }
        }
        wave_sync();
        for (int32_t base = 0; base < nslots; base += WAVE) {          // slots that hold a base seed
            LV<bool> fl;
            FOR_LANES(l) { const int32_t i = base + l; fl[l] = i < nslots && w.pos_full[i] != 0; }
            const uint64_t fb = wave_ballot(fl);
            FOR_LANES(l) { if (l == 0 && fb) w.bm[3][base >> 6] |= fb << (base & 63); }
        }
        wave_sync();
    } else {
        for (int32_t b = 0; b < n_base; ++b) {
            const DevSeed sd = w.seeds[s][b];
            const int32_t i = sd.clipping, nn = sd.n_nodes;
            w.bm[3][i >> 6] |= 1ull << (i & 63);
            // (a MEM of a short read covers ~100 positions: one per lane, not one after the other)
            const int32_t cover = nn + (i + nn < nslots ? 1 : 0);
            for (int32_t jb = 0; jb < cover; jb += WAVE) {
                FOR_LANES(l) { const int32_t j = jb + l; if (j < cover) w.msl[i + j] = (uint16_t)k; }
            }
            w.pos_full[i] = 1;            // suffix_seeds[i] holds exactly this full seed
            w.pos_cnt[i] = 1;
            w.pos_start[i] = (uint32_t)b; // index of the full seed among the base seeds
        }
        wave_sync();
    }
    SEED_T(2, tp)
    // Read tail (positions with fewer than k characters left): if the last k-mer is a node, its target node's
    // label ends with q[i..L) for every such i, so index_range matches all L - i characters.  The length is all
    // the replacement rules need for dominated positions; the range is fetched only if a position reports.
    // (PRIMARY graphs: only if the last k-mer was found in the base graph itself, not as a reverse complement)
    uint32_t *pre = (uint32_t *)w.cells;             // per position: parents of a single matched node (the extension's cell arena is idle)
    const bool pre_fits = (uint64_t)5 * (uint64_t)nslots <= MGX_PARAMS_OF(w).lim.cell_words;
    const bool tail_known = w.n_kmers > 0 && w.nodes[s][w.n_kmers - 1] != 0 && !w.inv_any[s]
                            && !(kWithPrimary && cfg.canonical >= 2 && w.nodes[s][w.n_kmers - 1] > g.n);
    // lane-parallel longest-prefix lookups for every position that can report a seed: the lookup of position i
    auto lookup_position = [&](const int32_t i, LineCtr &lc) -> bool {
        int32_t max_len = (int32_t)imin<uint32_t>(imin<uint32_t>(cfg.max_seed_length, (uint32_t)(k - 1)), (uint32_t)(L - i));
        uint16_t mlen = 0;
        uint32_t rf = 0, rl_ = 0;
        bool need = max_len >= (int32_t)w.msl[i];
        int32_t known = -1;                              // match length with a stored range, if any
        int32_t known_at = i;                            // the slot of rng[] that holds it
        if (need && w.mlen[s] && i < w.n_kmers && max_len == k - 1) {
            // k_map's index() walked this very chain (BOSS::index_range == index() up to the failing
            // character): skip lookups that cannot reach min_seed_length, reuse the range of those that do
            const uint32_t ml = w.mlen[s][i];
            if (ml == MLEN_LT_PREFIX) need = msl0 <= (int32_t)g.prefix_len;
            else if (ml < MLEN_TAIL) {
                need = (int32_t)ml >= msl0;
                if (need && w.rng[s]) known = (int32_t)ml;
            }
        }
        // (MANY: not deferred.  With a full seed at every matched k-mer the last-full-seed rule below (:240-244) drops the
        // tail positions' seeds one after the other without raising msl[] behind them, so ALL of them report and each
        // deferred range was a walk of ~25 dependent steps by the whole wavefront — 40 % of the label-aware seeding kernel,
        // profiles/r04_ab7_seed_sections.txt; here the tail positions walk side by side, one per lane.)
        // (Few seeds: only the first tail position reports, and raises msl[] for those behind it; looking it up here instead
        // moves 12 ms per 2 M reads from the bookkeeping to this loop and saves 1 — same file.)
        // the tail position two behind the last k-mer — the one that reports when that k-mer is covered by a MEM — has its
        // range from k_map (MLEN_TAIL, map_pipe.hpp): there the walk is one of 64 per wavefront
        if (need && tail_known && i == w.n_kmers + 1 && max_len == L - i && max_len == k - 2 && max_len >= msl0 && w.mlen[s]
                && w.rng[s] && w.mlen[s][w.n_kmers - 1] == MLEN_TAIL) {
            known = max_len; known_at = w.n_kmers - 1;
        }
        if (!MANY && known < 0 && need && tail_known && i >= w.n_kmers && max_len == L - i && max_len >= msl0) {
            mlen = (uint16_t)max_len;
            rf = DEFERRED_RANGE;
            need = false;
        }
        if (need) {
            uint64_t first, last;
            int32_t m;
            if (known >= 0) {
                const uint2 r = w.rng[s][known_at];
                ++lc.bit_lines;
                first = succ_last(g, r.x, lc);           // index_range's return (boss.hpp:756-763)
                last = r.y;
                m = known;
            } else {
                m = index_range_lane(w, s, i, max_len, msl0, &first, &last, lc);
            }
            if (m >= msl0 && first && first <= g.n) {
                mlen = (uint16_t)m;
                if (first == last) {
                    rf = SINGLE_NODE; rl_ = (uint32_t)first;
                    // the parents of that node (what the position contributes if it reports, below), fetched here,
                    // one position per lane, instead of one reporting position after the other: most of these never
                    // report — a longer match to their left covers them — but a dependent chain of three or four lines
                    // per reporting position was 30 % of this kernel
                    if (pre_fits) {
                        const int np = incoming_nodes32(g, first, pre + 5 * i + 1, 4, lc);
                        gst(pre + 5 * i, (uint32_t)np);
                    }
                }
                else { rf = rank_last(g, first, lc); rl_ = rank_last(g, last, lc); }
            }
        }
        w.ml[i] = mlen; w.rfirst[i] = rf; w.rlast[i] = rl_;
        return mlen != 0;
    };
    if (WAVE == 64 && nslots <= 3 * WAVE) {
        // Short reads (round 5): only the positions whose lookup can matter — msl[i] within reach, i.e. not covered by a MEM:
        // a dozen read-tail positions, plus ~30 per error — are looked up, COMPACTED into as few 64-lane passes as they need
        // (one, typically): a pass is a chain of six or seven dependent loads (match length, range, succ_last, bwd's rank and
        // select, the parents' scan) whatever the number of lanes that take part, and three passes per strand over mostly idle
        // lanes were a seventh of this kernel's time.  Which lane takes which position: the R-th set bit of the ballots.
        uint64_t nm[3] = { 0, 0, 0 };
        for (int c = 0; c < 3; ++c) {
            LV<bool> nd;
            FOR_LANES(l) {
                const int32_t i = c * WAVE + l;
                nd[l] = false;
                if (i < nslots) {
                    const int32_t max_len = (int32_t)imin<uint32_t>(imin<uint32_t>(cfg.max_seed_length, (uint32_t)(k - 1)), (uint32_t)(L - i));
                    nd[l] = max_len >= (int32_t)w.msl[i];
                    w.ml[i] = 0; w.rfirst[i] = 0; w.rlast[i] = 0;
                }
            }
            nm[c] = wave_ballot(nd);
        }
        wave_sync();
        const int32_t c0 = popc64(nm[0]), c1 = popc64(nm[1]), total = c0 + c1 + popc64(nm[2]);
        for (int32_t base = 0; base < total; base += WAVE) {
            LV<int32_t> nr, ns;
            FOR_LANES(l) {
                LineCtr lc = { 0, 0, 0 };
                const int32_t R = base + l;
                if (R < total) {
                    const int32_t i = R < c0 ? select64(nm[0], R + 1) : R < c0 + c1 ? WAVE + select64(nm[1], R - c0 + 1)
                                                                                     : 2 * WAVE + select64(nm[2], R - c0 - c1 + 1);
                    (void)lookup_position(i, lc);
                }
                nr[l] = (int32_t)(lc.rank_lines + lc.bit_lines); ns[l] = (int32_t)lc.select_lines;
            }
            w.ctr.rank_lines += (uint32_t)wave_sum(nr);
            w.ctr.select_lines += (uint32_t)wave_sum(ns);
        }
        wave_sync();
        for (int c = 0; c * WAVE < nslots; ++c) {
            LV<bool> hit;
            FOR_LANES(l) { const int32_t i = c * WAVE + l; hit[l] = i < nslots && w.ml[i] != 0; }
            const uint64_t hb = wave_ballot(hit);
            FOR_LANES(l) { if (l == 0) { w.bm[2][c] |= hb; w.bm[3][c] |= hb; } }
        }
    } else
    for (int32_t base = 0; base < nslots; base += WAVE) {
        LV<int32_t> nr, ns;
        LV<bool> hit;
        FOR_LANES(l) {
            LineCtr lc = { 0, 0, 0 };
            int32_t i = base + l;
            hit[l] = i < nslots && lookup_position(i, lc);
            nr[l] = (int32_t)(lc.rank_lines + lc.bit_lines); ns[l] = (int32_t)lc.select_lines;
        }
        w.ctr.rank_lines += (uint32_t)wave_sum(nr);
        w.ctr.select_lines += (uint32_t)wave_sum(ns);
        const uint64_t hb = wave_ballot(hit);
        FOR_LANES(l) { if (l == 0) { w.bm[2][base >> 6] |= hb << (base & 63); w.bm[3][base >> 6] |= hb << (base & 63); } }
    }
    wave_sync();
    SEED_T(3, tp)
    // sequential bookkeeping (:195-249)
    uint32_t alt_n = 0;
    const int32_t last_full_id = L >= k ? L - k + 1 : nslots;
    // only positions whose lookup matched >= min_seed_length characters can report (every other position takes
    // one of the two `continue`s below: msl[i] never drops under min_seed_length)
    for (int32_t i = bits_next(w.bm[2], nslots, 0, true); i < nslots; i = bits_next(w.bm[2], nslots, i + 1, true)) {
        int32_t max_len = (int32_t)imin<uint32_t>(imin<uint32_t>(cfg.max_seed_length, (uint32_t)(k - 1)), (uint32_t)(L - i));
        int32_t cur_msl = w.msl[i];
        if (max_len < cur_msl) continue;                       // lookup returns immediately (dbg_succinct.cpp:314)
        int32_t seed_length = w.ml[i];
        if (seed_length < cur_msl) continue;                   // match_size < min_match_length: no callback
        // the complexity filter is evaluated first in the reference (:226-229); it has no side effects, so
        // testing it only for positions that would report a seed is equivalent
#ifdef MGX_SEED_PROBE
        uint64_t tq = cycle_clock();
        bool lowc = cfg.seed_complexity_filter && window_low_complexity(w, s, i, cur_msl);
        SEED_T(6, tq)
        if (lowc) continue;
#else
        if (cfg.seed_complexity_filter && window_low_complexity(w, s, i, cur_msl)) continue;
#endif
        bool fresh_range = false;
        if (w.rfirst[i] == DEFERRED_RANGE) {
            fresh_range = true;
            // deferred tail lookup: this position does report, so its node range is needed after all
            uint64_t first, last;
            LineCtr lc = { 0, 0, 0 };
            const int32_t m = index_range_lane(w, s, i, max_len, msl0, &first, &last, lc);
            const bool ok = m >= msl0 && first && first <= g.n;
            uint32_t rf = 0, rl_ = 0;
            if (ok) {
                if (first == last) { rf = SINGLE_NODE; rl_ = (uint32_t)first; }
                else { rf = rank_last(g, first, lc); rl_ = rank_last(g, last, lc); }
            }
            w.ctr.rank_lines += lc.rank_lines; w.ctr.select_lines += lc.select_lines; w.ctr.bit_lines += lc.bit_lines;
            wave_sync();
            FOR_LANES(l) { if (l == 0) { w.rfirst[i] = rf; w.rlast[i] = rl_; } }
            wave_sync();
            if (!ok) continue;                                 // the eager path would not have listed it as a hit
        }
        // enumerate nodes whose suffix matches (dbg_succinct.cpp:349-392)
        uint32_t first_alt = alt_n;
        uint32_t cnt = 0;
        const bool one_node = w.rfirst[i] == SINGLE_NODE;
        const uint32_t r_begin = one_node ? 0u : w.rfirst[i], r_end = one_node ? 0u : w.rlast[i];
        const uint32_t np = (one_node && !fresh_range && pre_fits) ? pre[5 * i] : 0xFFFFFFFFu;
        if (np <= 4u) {
            // (fetched with the lookups above)
            if (alt_n + np > MGX_PARAMS_OF(w).lim.max_alt) { w.status = ST_CAPACITY; return; }
            for (uint32_t t = 0; t < np; ++t) w.alt[alt_n++] = pre[5 * i + 1 + t];
            cnt = np;
        } else
        for (uint32_t r = r_begin; r <= r_end; ++r) {
            uint64_t e = one_node ? (uint64_t)w.rlast[i] : select_last<true>(g, r, w.ctr);
            uint64_t inc[5];
            uint32_t fc[5];
            // call_incoming_to_target(bwd(e), node_last_value(e)) == parents of the node whose last edge is e
            int ni = incoming<true, false>(g, e, inc, fc, w.ctr);
            for (int t = 0; t < ni; ++t) {
                if (alt_n >= MGX_PARAMS_OF(w).lim.max_alt) { w.status = ST_CAPACITY; return; }
                w.alt[alt_n++] = (uint32_t)inc[t];
                ++cnt;
            }
        }
#ifdef MGX_SEED_PROBE
        SEED_T(7, tq)
#endif
        if (i >= last_full_id && cnt == 1 && w.msl[last_full_id - 1] == k && w.pos_full[last_full_id - 1]
                && w.pos_cnt[last_full_id - 1] == 1) {
            DevSeed fs = w.seeds[s][w.pos_start[last_full_id - 1]];
            uint32_t first_node = w.nodes[s][fs.clipping];
            if (w.alt[first_alt] == first_node) { alt_n = first_alt; continue; }
        }
        // append_suffix_seed for every alt node (:195-213)
        for (uint32_t a = 0; a < cnt; ++a) {
            int32_t ii = i;
            int32_t sl = seed_length;
            if (sl > (int32_t)w.msl[ii]) { w.pos_cnt[ii] = 0; w.pos_full[ii] = 0; }
            w.msl[ii] = (uint16_t)sl;
            if (w.pos_cnt[ii] == 0) { w.pos_start[ii] = first_alt + a; w.pos_full[ii] = 0; }
            ++w.pos_cnt[ii];
            // the positions behind i take the remainder of the match while it is longer than what they hold: position
            // i + 1 + t gets sl - t.  Every test reads the value from before this loop (a position is written after its own
            // test only), so the tests of a stretch run side by side and the first failure ends the run.
            ++ii;
            for (bool go_on = true; go_on && ii < nslots && sl > 0; ) {
                LV<bool> fail;
                FOR_LANES(l) { const int32_t p = ii + l; fail[l] = !(p < nslots && sl - l > (int32_t)w.msl[p]); }
                const uint64_t fb = wave_ballot(fail);
                const int32_t run = fb ? ctz64(fb) : WAVE;
                wave_sync();
                FOR_LANES(l) {
                    if (l < run) { const int32_t p = ii + l; w.msl[p] = (uint16_t)(sl - l); w.pos_cnt[p] = 0; w.pos_full[p] = 0; }
                }
                wave_sync();
                ii += run; sl -= run;
                go_on = run == WAVE;
            }
        }
    }
    if (kWithPrimary && cfg.canonical >= 2) { primary_rc_suffix_seeds(w, s, alt_n); if (w.status != ST_OK) return; }
    SEED_T(4, tp)
    // aggregate (:316-357): rebuild the seed list in position order
    // full seeds are already stored at [0, n_base); copy them out of the way first
    DevSeed *tmp = (DevSeed *)w.indices;         // scratch big enough for n_base <= L seeds
    uint32_t num_matching = 0;
    if (MANY) {
        for (int32_t base = 0; base < n_base; base += WAVE) {
            FOR_LANES(l) { const int32_t b = base + l; if (b < n_base) tmp[b] = w.seeds[s][b]; }
        }
        wave_sync();
        // One slot per lane: what it emits (its full seed, or its sub-k seeds if they are at most max_num_seeds_per_locus), where
        // (prefix sums over the slots in position order), and — for num_matching, whose update looks at the END of the slot
        // emitted before (:343-350) — the ends of the emitted slots in order (in rfirst[], dead by now).
        int32_t n_out = 0, n_emit = 0;
        const int32_t max_seeds = (int32_t)MGX_PARAMS_OF(w).lim.max_seeds;
        uint32_t *ends = w.rfirst;
        for (int32_t base = 0; base < nslots; base += WAVE) {
            const uint64_t word = w.bm[3][base >> 6] >> (base & 63);     // (WAVE divides 64: a chunk never straddles two words)
            if (!(WAVE == 64 ? word : (word & ((1ull << (WAVE & 63)) - 1)))) continue;
            LV<int32_t> ns, em, bg, en;
            LV<bool> fullv;
            FOR_LANES(l) {
                const int32_t i = base + l;
                int32_t cnt = (i < nslots && ((word >> l) & 1)) ? (int32_t)w.pos_cnt[i] : 0;
                const bool full = cnt && w.pos_full[i];
                int32_t end = 0, nseed = 0;
                if (full) { const DevSeed fs = tmp[w.pos_start[i]]; end = i + (int32_t)fs.length; nseed = 1; }
                else if (cnt && (uint32_t)cnt <= cfg.max_num_seeds_per_locus) { end = i + (int32_t)w.msl[i]; nseed = cnt; }
                ns[l] = nseed; em[l] = nseed ? 1 : 0; bg[l] = i; en[l] = end; fullv[l] = full;
            }
            const LV<int32_t> so = wave_prefix_sum_excl(ns), eo = wave_prefix_sum_excl(em);
            const int32_t tot = wave_sum(ns), tot_e = wave_sum(em);
            if (n_out + tot > max_seeds) { w.status = ST_CAPACITY; return; }
            FOR_LANES(l) {
                if (ns[l]) {
                    const int32_t i = bg[l];
                    ends[n_emit + eo[l]] = (uint32_t)en[l];
                    DevSeed *dst = w.seeds[s] + n_out + so[l];
                    if (fullv[l]) {
                        dst[0] = tmp[w.pos_start[i]];
                        w.alive[s][n_out + so[l]] = 1;
                    } else {
                        const int32_t sl = (int32_t)w.msl[i];
                        for (int32_t a2 = 0; a2 < ns[l]; ++a2) {
                            DevSeed sd;
                            sd.clipping = (uint16_t)i; sd.length = (uint16_t)sl; sd.offset = (uint16_t)(k - sl); sd.n_nodes = 1;
                            sd.node = w.alt[w.pos_start[i] + a2];
                            dst[a2] = sd;
                            w.alive[s][n_out + so[l] + a2] = 1;
                        }
                    }
                }
            }
            wave_sync();
            LV<int32_t> contrib;
            FOR_LANES(l) {
                int32_t c = 0;
                if (ns[l]) {
                    const int32_t e_idx = n_emit + eo[l];
                    const int32_t last_end = e_idx ? (int32_t)ends[e_idx - 1] : 0;
                    c = bg[l] < last_end ? en[l] - last_end : en[l] - bg[l];
                }
                contrib[l] = c;
            }
            num_matching += (uint32_t)wave_sum(contrib);
            n_out += tot; n_emit += tot_e;
        }
        w.n_seeds[s] = n_out;
    } else {
        for (int32_t b = 0; b < n_base; ++b) tmp[b] = w.seeds[s][b];
        w.n_seeds[s] = 0;
        int32_t last_end = 0;
        // slots that can hold seeds: positions of base seeds and of lookup hits (pos_cnt is only ever raised there)
        for (int32_t i = bits_next(w.bm[3], nslots, 0, true); i < nslots; i = bits_next(w.bm[3], nslots, i + 1, true)) {
            int32_t cnt = w.pos_cnt[i];
            if (!cnt) continue;
            bool full = w.pos_full[i];
            bool emitted = false;
            int32_t begin = i, end = 0;
            if (full) {
                DevSeed fs = tmp[w.pos_start[i]];
                if (!push_seed(w, s, fs.clipping, fs.length, 0, fs.n_nodes, fs.node)) return;
                end = begin + fs.length;
                emitted = true;
            } else if ((uint32_t)cnt <= cfg.max_num_seeds_per_locus) {
                int32_t sl = w.msl[i];
                for (int32_t a = 0; a < cnt; ++a)
                    if (!push_seed(w, s, i, sl, k - sl, 1, w.alt[w.pos_start[i] + a])) return;
                end = begin + sl;
                emitted = true;
            }
            if (emitted) {
                if (begin < last_end) num_matching += end - begin - (last_end - begin);
                else num_matching += end - begin;
                last_end = end;
            }
        }
    }
    w.num_matching[s] = num_matching;
    SEED_T(5, tp)
}

#ifndef MGX_NO_EXTEND    // translation units that only seed (mgx.hip: k_map, k_seed) skip the extension half
// ------------------------------------------------------------------------------------------------
// extension (DefaultColumnExtender::extend, A/aligner_extender_methods.cpp:412-772)
// ------------------------------------------------------------------------------------------------
// metadata of column i <-> its slot (positions and sizes fit 16 bits: Lmax <= MGX_MAX_QUERY_LENGTH; gap scores are int8)
MGX_DEV ColMeta col_unpack(const uint32_t *m, int32_t i, int32_t go, int32_t ge) {
    ColMeta c;
    if ((m[1] >> 30) == 1u) {                                    // compact form
        c.node = m[0]; c.parent = (int32_t)(m[1] & 0xFFFFFF);
        const uint32_t code = (m[1] >> 24) & 7, sc = (m[1] >> 27) & 3;
        c.base = (int32_t)(m[2] << 10) >> 10;
        c.size = (int32_t)((m[2] >> 22) & 31);
        c.offset = (int32_t)(m[3] & 0xFFFF); c.trim = (int32_t)((m[3] >> 16) & 0x7FFF);
        c.max_pos = c.trim + (int32_t)(m[2] >> 27);
        c.score = sc == 0 ? 0 : sc == 1 ? go : ge;
        c.cells = NO_CELLS; c.org = c.trim & ~3;
        c.cw = (uint32_t)decode_code(code) | ((uint32_t)CCELLS << 8) | CW_CHAIN | CW_COMPACT;
        c.self = i;
        return c;
    }
    c.node = m[0]; c.parent = (int32_t)m[1]; c.offset = (int32_t)m[2]; c.base = (int32_t)m[3]; c.cells = m[4];
    c.max_pos = (int32_t)(m[5] & 0xFFFF); c.trim = (int32_t)(m[5] >> 16);
    c.size = (int32_t)(m[6] & 0xFFFF); c.org = (int32_t)(m[6] >> 16);
    c.score = (int32_t)(int8_t)((m[7] >> 8) & 0xFF);
    c.cw = (m[7] & 0xFF) | (((m[7] >> 16) & 0x7FFF) << 8) | (m[7] & CW_CHAIN);
    c.self = i;
    return c;
}
MGX_DEV ColMeta col_load(const Wave &w, int32_t i) {
    const ColSlot *sl = w.cols + i;
    uint32_t m[8];
#if MGX_WAVE_EMU
    for (int t = 0; t < 8; ++t) m[t] = sl->m[t];
#else
    mgx_mem::load_bytes<32>(sl->m, m);
#endif
    const DevConfig &cfg = MGX_PARAMS_OF(w).cfg;
    return col_unpack(m, i, cfg.gap_open, cfg.gap_ext);
}
MGX_DEV void col_pack(const ColMeta &c, uint32_t *m) {
    m[0] = c.node; m[1] = (uint32_t)c.parent; m[2] = (uint32_t)c.offset; m[3] = (uint32_t)c.base; m[4] = c.cells;
    m[5] = ((uint32_t)c.max_pos & 0xFFFF) | ((uint32_t)c.trim << 16);
    m[6] = ((uint32_t)c.size & 0xFFFF) | ((uint32_t)c.org << 16);
    m[7] = (c.cw & 0xFF) | (((uint32_t)c.score & 0xFF) << 8) | ((uint32_t)col_wc(c) << 16) | (c.cw & CW_CHAIN);
}
// store by one lane of the wave program
MGX_DEV void col_store(Wave &w, int32_t i, const ColMeta &c) {
    uint32_t m[8];
    col_pack(c, m);
    ColSlot *sl = w.cols + i;
    FOR_LANES(l) {
        if (l == 0) {
#if MGX_WAVE_EMU
            for (int t = 0; t < 8; ++t) sl->m[t] = m[t];
#else
            mgx_mem::store_bytes<32>(sl->m, m);
#endif
        }
    }
}

// cell of column c at window position pos (absolute, like DPTColumn's trim + index); ninf outside what the reference's
// vectors hold (index < 0 or >= size + 5: undefined there, defined as ninf here and in the oracle)
MGX_DEV bool cell_idx(const ColMeta &c, int32_t pos, int32_t &x) {
    const int32_t j = pos - c.trim;
    x = pos - c.org;
    return j >= 0 && j < c.size + 5 && x < col_wc(c);
}
MGX_DEV int32_t cell_S(const Wave &w, const ColMeta &c, int32_t pos) {
    int32_t x;
    if (!cell_idx(c, pos, x)) return NINF;
    if (col_compact(c)) {
        const int32_t v = (int32_t)gld((const int8_t *)(w.cols + c.self) + 20 + 8 * (x >> 2) + (x & 3));
        return v == (int32_t)S8_NINF ? NINF : c.base + v;
    }
    if (col_chain(c)) {
        const int32_t v = (int32_t)gld(w.cols_s16 + (int64_t)c.self * FWS + x);
        return v == (int32_t)S16_NINF ? NINF : c.base + v;
    }
    return gld(w.cells + c.cells + x);
}
MGX_DEV uint32_t cell_flags(const Wave &w, const ColMeta &c, int32_t pos) {
    int32_t x;
    if (!cell_idx(c, pos, x)) return 0;
    if (col_compact(c)) return gld((const uint8_t *)(w.cols + c.self) + 16 + 8 * (x >> 2) + (x & 3));
    if (col_chain(c)) return gld(w.cols[c.self].flags + x);
    return gld((const uint8_t *)(w.cells + c.cells + 2 * col_wc(c)) + x);
}

// ------------------------------------------------------------------------------------------------
// convergence checker (SeedFilteringExtender, A/aligner_extender_methods.cpp:66-207)
// ------------------------------------------------------------------------------------------------
// conv_checker_ maps a node to the best score reached per query position in this extension.  Round 3 layout (round 2 kept a
// 32-byte slot per node in a table sized for the worst case plus an int32 copy of every column's window in a pool: one random
// line read, one written and ~100 bytes appended per column, half of the kernel's traffic):
//  * hash slots of 8 bytes: key (node id) | generation:8 kind:1 index:23.  The table grows by LEVELS of 512, 1024, ... slots
//    (level t occupies [512 (2^t - 1), 512 (2^(t+1) - 1)) of `tab`; the top level has DevLimits::hash_size slots): a short
//    extension only ever touches the 4 KB of level 0, which stay cache resident; a level is left for the next one when it is
//    half full (the live slots are re-inserted).  Generation tags make clearing O(1); after 255 generations the levels used
//    since the last wrap are zeroed.
//  * kind 0, an ALIAS: the entry's vector IS the S window of chain-format column `index` (ColSlot: the compact or the two-line
//    form) — the query range follows from the column's trim / size, the values from its 8- / 16-bit offsets and `base`.  Nothing
//    but the 8-byte slot is written for a node's first column, which is what 99 % of the columns are.  Aliases need the column
//    table to outlive the table's readers, so they are used only with one alignment per seed (see aln_both).
//  * kind 1, a POOL entry: ConvRec `index` {pool offset, query range, capacity} and int32 words in the pool, as in round 2
//    (position p of the range is pool[off + p - start]; an entry at the pool's top grows in place, else it moves there).
//    Columns of the general path, filter_nodes marks and any entry that is written a second time (an alias is first copied out).
// Keys are plain node ids: the reference adds max_index for RCDBG views (:74-75,107-108), a constant for all keys of one
// extender's table here (an extender keeps its view), and filter_nodes (:158) uses the raw id on the forward extender, whose
// view is the graph itself.

MGX_DEV uint64_t cs_make(uint32_t key, uint32_t gen, uint32_t kind, uint32_t idx) {
    return (uint64_t)key | ((uint64_t)((gen << 24) | (kind << 23) | idx) << 32);
}
MGX_DEV uint32_t cs_key(uint64_t e) { return (uint32_t)e; }
MGX_DEV uint32_t cs_gen(uint64_t e) { return (uint32_t)(e >> 56); }
MGX_DEV uint32_t cs_kind(uint64_t e) { return (uint32_t)(e >> 55) & 1u; }
MGX_DEV uint32_t cs_idx(uint64_t e) { return (uint32_t)(e >> 32) & 0x7FFFFFu; }

// set_seed (:90-98): a new generation, at level 0
MGX_DEV void conv_clear(Wave &w, ConvChecker &c) {
    const uint32_t hs = uni(MGX_PARAMS_OF(w).lim.hash_size);
    uint32_t gen = uni(c.gen) + 1;
    if (gen > 255) {
        // every slot a generation of this round can have written lies in the levels up to `dirty`
        const uint32_t cap0 = conv_cap0(hs);
        const uint32_t n = cap0 * ((2u << uni(c.dirty)) - 1);
        for (uint32_t base = 0; base < n; base += WAVE) {
            FOR_LANES(l) { const uint32_t j = base + l; if (j < n) gst(c.tab + j, (uint64_t)0); }
        }
        wave_sync();
        gen = 1;
        c.dirty = 0;
    }
    c.gen = gen; c.n_entries = 0; c.n_recs = 0; c.pool_top = 0;
    c.cap = conv_cap0(hs); c.base = 0;
}

MGX_DEV uint32_t conv_hash(uint32_t key, uint32_t mask) {
    // any mixing works (results do not depend on it)
    uint32_t h = (key ^ (key >> 15)) * 0x85EBCA6Bu;
    h ^= h >> 13;
    return h & mask;
}

// Linear probing from slot `h` of the current level, whose content `e` the caller has loaded: returns the slot of `key`
// (found, e = its content) or the free slot where it would go.
MGX_DEV uint32_t conv_probe_from(const ConvChecker &c, uint32_t key, uint32_t h, uint64_t &e, bool &found) {
    const uint64_t *lv = c.tab + c.base;
    const uint32_t gen = c.gen, mask = c.cap - 1;
    for (;;) {
        if (cs_gen(e) != gen) { found = false; return h; }
        if (cs_key(e) == key) { found = true; return h; }
        h = (h + 1) & mask;
        e = gld(lv + h);
    }
}
MGX_DEV uint32_t conv_probe(const ConvChecker &c, uint32_t key, uint64_t &e, bool &found) {
    const uint32_t h = conv_hash(key, c.cap - 1);
    e = gld(c.tab + c.base + h);
    return conv_probe_from(c, key, h, e, found);
}

// a level that is half full hands over to the next one
MGX_DEV void conv_grow(Wave &w, ConvChecker &c) {
    const uint32_t hs = uni(MGX_PARAMS_OF(w).lim.hash_size);
    while (uni(c.n_entries) * 2 > uni(c.cap) && uni(c.cap) < hs) {
        const uint32_t ocap = uni(c.cap), obase = uni(c.base), gen = uni(c.gen);
        const uint32_t ncap = 2 * ocap, nbase = obase + ocap;
        const uint64_t *ol = c.tab + obase;
        uint64_t *nl = c.tab + nbase;
        // one live slot at a time (the probe sequences of a lane-parallel insert would collide); this runs once per 256,
        // 512, ... distinct nodes of an extension
        for (uint32_t j = 0; j < ocap; ++j) {
            const uint64_t e = uni(gld(ol + j));
            if (cs_gen(e) != gen) continue;
            uint32_t h = conv_hash(cs_key(e), ncap - 1);
            while (cs_gen(uni(gld(nl + h))) == gen) h = (h + 1) & (ncap - 1);
            FOR_LANES(l) { if (l == 0) gst(nl + h, e); }
            wave_sync();
        }
        c.cap = ncap; c.base = nbase;
        uint32_t lvl = 0;
        while ((conv_cap0(hs) << lvl) < ncap) ++lvl;
        if (lvl > uni(c.dirty)) c.dirty = lvl;
    }
}

// claim free slot `slot` of the current level for a new key; false = capacity (status set)
MGX_DEV bool conv_claim(Wave &w, ConvChecker &c, uint32_t slot, uint32_t key, uint32_t kind, uint32_t idx) {
    const DevLimits &lim = MGX_PARAMS_OF(w).lim;
    const uint32_t ne = uni(c.n_entries);
    if (ne >= uni(lim.max_columns + lim.max_path) || ne * 2 >= uni(lim.hash_size)) { w.status = ST_CAPACITY; return false; }
    c.n_entries = ne + 1;
    FOR_LANES(l) { if (l == 0) gst(c.tab + c.base + slot, cs_make(key, c.gen, kind, idx)); }
    wave_sync();
    conv_grow(w, c);
    return true;
}

// the pool entries' records live right behind the hash levels (one pointer less in the LDS control block)
MGX_DEV ConvRec *conv_recs(const Wave &w, const ConvChecker &c) {
    return reinterpret_cast<ConvRec *>(c.tab + conv_tab_slots(uni(MGX_PARAMS_OF(w).lim.hash_size)));
}
MGX_DEV ConvRec conv_rec_load(const Wave &w, const ConvChecker &c, uint32_t r) {
    const uint4 v = gld(reinterpret_cast<const uint4 *>(conv_recs(w, c) + r));
    ConvRec rec; rec.off = v.x; rec.start = (int32_t)v.y; rec.len = (int32_t)v.z; rec.cap = v.w;
    return rec;
}
MGX_DEV void conv_rec_store(const Wave &w, ConvChecker &c, uint32_t r, const ConvRec &rec) {
    uint4 v; v.x = rec.off; v.y = (uint32_t)rec.start; v.z = (uint32_t)rec.len; v.w = rec.cap;
    FOR_LANES(l) { if (l == 0) gst(reinterpret_cast<uint4 *>(conv_recs(w, c) + r), v); }
}
// vec[p] for query position p of pool entry rec
MGX_DEV int32_t *conv_vec(const ConvChecker &c, const ConvRec &rec) { return c.pool + (int64_t)rec.off - (int64_t)rec.start; }

// a new pool entry for [start, start + len) (words not initialised); returns the record number or -1 (capacity)
MGX_DEV int32_t conv_new_rec(Wave &w, ConvChecker &c, int32_t start, int32_t len, ConvRec &rec) {
    const DevLimits &lim = MGX_PARAMS_OF(w).lim;
    const uint32_t nr = uni(c.n_recs), top = uni(c.pool_top);
    if (nr >= uni(lim.max_columns + lim.max_path) || (uint64_t)top + (uint32_t)len > uni(lim.conv_pool_words)) { w.status = ST_CAPACITY; return -1; }
    c.n_recs = nr + 1;
    c.pool_top = top + (uint32_t)len;
    rec.off = top; rec.start = start; rec.len = len; rec.cap = (uint32_t)len;
    conv_rec_store(w, c, nr, rec);
    return (int32_t)nr;
}

// Make pool entry r cover [ns, ns + nl), a superset of its range: in place if its allocation reaches (an entry at the pool's
// top grows there), else it moves to the pool's top (old values copied; newly covered positions are left for the caller to
// write, as with the reference's vector::insert + fill).  Updates the record and `rec`; false = out of pool.
MGX_DEV bool conv_cover(Wave &w, ConvChecker &c, uint32_t r, ConvRec &rec, int32_t ns, int32_t nl) {
    const uint32_t words = uni(MGX_PARAMS_OF(w).lim.conv_pool_words);
    const uint32_t top = uni(c.pool_top);
    if (ns == rec.start && (uint32_t)nl <= rec.cap) {
        rec.len = nl;
    } else if (ns == rec.start && rec.off + rec.cap == top) {
        if ((uint64_t)rec.off + (uint32_t)nl > words) { w.status = ST_CAPACITY; return false; }
        c.pool_top = rec.off + (uint32_t)nl;
        rec.cap = (uint32_t)nl; rec.len = nl;
    } else {
        if ((uint64_t)top + (uint32_t)nl > words) { w.status = ST_CAPACITY; return false; }
        const int32_t *src = c.pool + rec.off;
        int32_t *dst = c.pool + top + (rec.start - ns);
        for (int32_t base = 0; base < rec.len; base += WAVE) {
            LV<int32_t> v;
            FOR_LANES(l) { const int32_t j = base + l; v[l] = j < rec.len ? gld(src + j) : 0; }
            FOR_LANES(l) { const int32_t j = base + l; if (j < rec.len) gst(dst + j, v[l]); }
        }
        c.pool_top = top + (uint32_t)nl;
        rec.off = top; rec.cap = (uint32_t)nl; rec.start = ns; rec.len = nl;
    }
    rec.start = ns;
    conv_rec_store(w, c, r, rec);
    wave_sync();
    return true;
}

// fill vec positions [a, b) with `val`
MGX_DEV void fill_range(int32_t *vec, int32_t a, int32_t b, int32_t val) {
    for (int32_t base = a; base < b; base += WAVE) {
        FOR_LANES(l) { int32_t j = base + l; if (j < b) gst(vec + j, val); }
    }
}

// The query range an alias of column c stands for (update_seed_filter's arguments when the column was entered: cells
// j in [skip, size) of the column at query positions start + trim + j - 1, skip = 1 for a window that starts at 0).
MGX_DEV void conv_alias_range(const ConvChecker &c, const ColMeta &col, int32_t &qs, int32_t &len) {
    const int32_t skip = col.trim ? 0 : 1;
    qs = (int32_t)c.start + col.trim - (col.trim ? 1 : 0);
    len = col.size - skip;
}
// value of an alias at query position p (inside its range)
MGX_DEV int32_t conv_alias_at(const Wave &w, const ConvChecker &c, const ColMeta &col, int32_t p) {
    return cell_S(w, col, p - (int32_t)c.start + 1);
}

// An alias about to be written a second time becomes a pool entry with the same content; `slot` is its hash slot (current
// level).  Returns the record number (rec filled) or -1.
MGX_DEV int32_t conv_materialize(Wave &w, ConvChecker &c, uint32_t slot, uint64_t e, ConvRec &rec) {
    const ColMeta col = uni_col(col_load(w, (int32_t)cs_idx(e)));
    int32_t qs, len;
    conv_alias_range(c, col, qs, len);
    const int32_t r = conv_new_rec(w, c, qs, len, rec);
    if (r < 0) return -1;
    int32_t *vec = conv_vec(c, rec);
    for (int32_t base = 0; base < len; base += WAVE) {
        LV<int32_t> v;
        FOR_LANES(l) { const int32_t j = base + l; v[l] = j < len ? conv_alias_at(w, c, col, qs + j) : 0; }
        FOR_LANES(l) { const int32_t j = base + l; if (j < len) gst(vec + qs + j, v[l]); }
    }
    FOR_LANES(l) { if (l == 0) gst(c.tab + c.base + slot, cs_make(cs_key(e), c.gen, 1u, (uint32_t)r)); }
    wave_sync();
    return r;
}

// update_seed_filter (:100-156) for a column of the general path: S in the staged (two-tier) array s_tier, cells
// s_skip .. s_skip + size at query positions from query_start.  Returns converged score (NINF = nothing improved).
MGX_NI_G5 int32_t update_seed_filter(Wave &w, ExtenderState &E, uint32_t node, int32_t query_start,
                                   const Tier s_tier, int32_t s_skip, int32_t size) {
    MGX_ASSUME_LDS(&w);
    MGX_ASSUME_LDS(&E);
    const AlignParams &P = MGX_PARAMS_OF(w);
    const int32_t s_cap = uni(w.st_cap);
    s_skip = uni(s_skip);
#define s_cells(j) tget(s_tier, s_cap, s_skip + (j))
    auto column_max = [&]() {
        int32_t m = INT32_MIN;
        for (int32_t base = 0; base < size; base += WAVE) {
            LV<int32_t> x;
            FOR_LANES(l) { int32_t j = base + l; x[l] = j < size ? s_cells(j) : INT32_MIN; }
            m = imax(m, wave_max(x));
        }
        return m;
    };
    auto store_column = [&](int32_t *vec) {
        for (int32_t base = 0; base < size; base += WAVE) {
            FOR_LANES(l) { int32_t j = base + l; if (j < size) gst(vec + query_start + j, s_cells(j)); }
        }
        wave_sync();
    };
    if (node == 0) return column_max();
    node = uni(node); query_start = uni(query_start); size = uni(size);
    ConvChecker &C = E.conv;
    uint64_t e;
    bool found;
    const uint32_t slot = conv_probe(C, node, e, found);
    ConvRec rec;
    int32_t r;
    if (!found) {
        r = conv_new_rec(w, C, query_start, size, rec);
        if (r < 0) return NINF;
        if (!conv_claim(w, C, slot, node, 1u, (uint32_t)r)) return NINF;
        store_column(conv_vec(C, rec));
        return column_max();
    }
    if (cs_kind(e) == 0) {
        r = conv_materialize(w, C, slot, e, rec);
        if (r < 0) return NINF;
    } else {
        r = (int32_t)cs_idx(e);
        rec = conv_rec_load(w, C, (uint32_t)r);
    }
    int32_t start = uni(rec.start), len = uni(rec.len);
    if (query_start + size <= start) {
        if (!conv_cover(w, C, (uint32_t)r, rec, query_start, start + len - query_start)) return NINF;
        int32_t *vec = conv_vec(C, rec);
        fill_range(vec, query_start + size, start, NINF);
        store_column(vec);
        return column_max();
    }
    if (query_start >= start + len) {
        if (!conv_cover(w, C, (uint32_t)r, rec, start, query_start + size - start)) return NINF;
        int32_t *vec = conv_vec(C, rec);
        fill_range(vec, start + len, query_start, NINF);
        store_column(vec);
        return column_max();
    }
    {
        const int32_t ns = imin(start, query_start), ne = imax(start + len, query_start + size);
        if (ns != start || ne != start + len) { if (!conv_cover(w, C, (uint32_t)r, rec, ns, ne - ns)) return NINF; }
    }
    int32_t *vec = conv_vec(C, rec);
    if (query_start < start) fill_range(vec, query_start, start, NINF);
    if (query_start + size > start + len) fill_range(vec, start + len, query_start + size, NINF);
    wave_sync();
    int32_t max_changed = NINF;
    const double rel = P.cfg.rel_score_cutoff;
    for (int32_t base = 0; base < size; base += WAVE) {
        LV<int32_t> x;
        FOR_LANES(l) {
            int32_t j = base + l;
            x[l] = NINF;
            if (j < size) {
                int32_t sv = s_cells(j);
                int32_t vv = gld(vec + query_start + j);
                if ((double)sv > (double)vv * rel) {
                    vv = imax(vv, sv);
                    gst(vec + query_start + j, vv);
                    x[l] = vv;
                }
            }
        }
        max_changed = imax(max_changed, wave_max(x));
    }
    wave_sync();
    return max_changed;
}
#undef s_cells

// check_seed (:66-88): true when the seed is still worth extending
MGX_DEV bool check_seed(Wave &w, const ExtenderState &E, uint32_t last_node, int32_t qlen, int32_t clipping, int32_t score) {
    const ConvChecker &C = E.conv;
    uint64_t e;
    bool found;
    conv_probe(C, last_node, e, found);
    if (!found) return true;
    const int32_t pos = qlen + clipping - 1;
    if (cs_kind(e) == 0) {
        const ColMeta col = col_load(w, (int32_t)cs_idx(e));
        int32_t qs, len;
        conv_alias_range(C, col, qs, len);
        if (pos < qs || pos - qs >= len) return true;
        return conv_alias_at(w, C, col, pos) < score;
    }
    const ConvRec rec = conv_rec_load(w, C, cs_idx(e));
    if (pos < rec.start || pos - rec.start >= rec.len) return true;
    return gld(conv_vec(C, rec) + pos) < score;
}

// filter_nodes (:158-207); written out only with several alignments per seed (aln_both applies it lazily otherwise)
MGX_NI_G5 void filter_nodes(Wave &w, ExtenderState &E, uint32_t node, int32_t query_start, int32_t query_end) {
    MGX_ASSUME_LDS(&w);
    MGX_ASSUME_LDS(&E);
    const int32_t mscore = -NINF;
    int32_t size = query_end - query_start;
    ConvChecker &C = E.conv;
    uint64_t e;
    bool found;
    const uint32_t slot = conv_probe(C, node, e, found);
    ConvRec rec;
    int32_t r;
    if (!found) {
        r = conv_new_rec(w, C, query_start, size, rec);
        if (r < 0) return;
        if (!conv_claim(w, C, slot, node, 1u, (uint32_t)r)) return;
        fill_range(conv_vec(C, rec), query_start, query_end, mscore);
        wave_sync();
        return;
    }
    if (cs_kind(e) == 0) {
        r = conv_materialize(w, C, slot, e, rec);
        if (r < 0) return;
    } else {
        r = (int32_t)cs_idx(e);
        rec = conv_rec_load(w, C, (uint32_t)r);
    }
    const int32_t start = rec.start, len = rec.len;
    const int32_t ns = imin(start, query_start), ne = imax(start + len, query_end);
    if (ns != start || ne != start + len) { if (!conv_cover(w, C, (uint32_t)r, rec, ns, ne - ns)) return; }
    int32_t *vec = conv_vec(C, rec);
    if (query_end <= start) fill_range(vec, query_end, start, NINF);                   // gap below the old range
    else if (query_start >= start + len) fill_range(vec, start + len, query_start, NINF);   // gap above it
    fill_range(vec, query_start, query_end, mscore);      // mscore is the maximum, so max(v, mscore) == mscore
    wave_sync();
}

// capacity of a reference vector created with `size0` elements (+5 reserved) after `pushes` push_backs
// followed by reserve(size + 5) (DPTColumn::create :389-410, extend_ins_end :293-328; libstdc++ growth)
MGX_DEV uint32_t ref_capacity(uint32_t size0, uint32_t pushes) {
    uint32_t cap = size0 + 5;
    if (!pushes) return cap;
    uint32_t target = size0 + pushes;
    while (cap < target) cap = imax<uint32_t>(1u, 2 * cap);
    return imax(cap, target + 5);
}

MGX_DEV uint64_t queue_key(int32_t score, int32_t neg_off_diag, uint32_t idx) {
    // orders like std::tuple<score, -|off diag|, table idx, ...> (:477-480); idx is unique.
    // 24 bits of score (|score| < 2^23), 16 bits of off-diagonal distance, 24 bits of table index.
    return ((uint64_t)(uint32_t)(score + (1 << 23)) << 40) | ((uint64_t)(uint32_t)(neg_off_diag + 32768) << 24) | idx;
}
MGX_DEV int32_t key_score(uint64_t key) { return (int32_t)(uint32_t)(key >> 40) - (1 << 23); }
MGX_DEV uint32_t key_idx(uint64_t key) { return (uint32_t)(key & 0xFFFFFF); }

MGX_DEV uint64_t qget(const uint64_t *arena, int32_t i) { return gld(arena + i); }
MGX_DEV void qset(uint64_t *arena, int32_t i, uint64_t v) { gst(arena + i, v); }

// The frontier (std::priority_queue<TableIt>, :477-487) is kept as an ascending sorted array (keys are
// unique), so the maximum is at the back.  Insert = lane-parallel rank + shift.
MGX_DEV void frontier_insert(Wave &w, int32_t &qn, uint64_t key) {
    key = uni(key);
    qn = uni(qn);
    int32_t pos = 0;
    for (int32_t base = 0; base < qn; base += WAVE) {
        LV<bool> lt;
        FOR_LANES(l) { int32_t j = base + l; lt[l] = j < qn && qget(w.queue, j) < key; }
        pos += popc64(wave_ballot(lt));
    }
    // shift [pos, qn) up by one, highest chunk first
    for (int32_t top = qn; top > pos; top -= WAVE) {
        int32_t lo = imax(pos, top - WAVE);
        LV<uint64_t> v;
        FOR_LANES(l) { int32_t j = lo + l; v[l] = j < top ? qget(w.queue, j) : 0; }
        wave_sync();
        FOR_LANES(l) { int32_t j = lo + l; if (j < top) qset(w.queue, j + 1, v[l]); }
        wave_sync();
    }
    qset(w.queue, pos, key);
    ++qn;
    wave_sync();
}

#ifndef MGX_NO_EXTEND
// frontier_insert on the extension's loop state (keeps the cached top score current)
MGX_DEV void frontier_push(Wave &w, uint64_t key) {
    XState &x = w.x;
    int32_t qn = x.qn;
    x.q_top = qn ? imax(x.q_top, key_score(key)) : key_score(key);
    frontier_insert(w, qn, key);
    x.qn = qn;
}
#endif

MGX_DEV int32_t st_S(const Staging &s, int32_t cap, int32_t size, int32_t j) { return (j >= 0 && j < size + 5) ? tget(s.S, cap, j) : NINF; }
MGX_DEV int32_t st_F(const Staging &s, int32_t cap, int32_t size, int32_t j) { return (j >= 0 && j < size + 5) ? tget(s.F, cap, j) : NINF; }

// make column `idx` resident in a staging buffer; returns the buffer index (the column must have an S / F record:
// every column that can be popped from the frontier has one)
MGX_DEV int stage_column(Wave &w, int32_t idx, const ColMeta &c) {
    if (w.st[0].col == idx) return 0;
    if (w.st[1].col == idx) return 1;
    const int b = 0;
    Staging &s = w.st[b];
    const int32_t n = c.size + 5;
    const int32_t cap = w.st_cap;
    const int32_t wc = col_wc(c), shift = c.trim - c.org;
    const int32_t *recS = w.cells + c.cells, *recF = recS + wc;
    if (c.cells == NO_CELLS) { w.status = ST_CAPACITY; return b; }          // cannot happen (see above); fail loudly if it does
    for (int32_t base = 0; base < n; base += WAVE) {
        FOR_LANES(l) {
            int32_t j = base + l;
            if (j < n) {                                  // a parent's E is never read
                const int32_t x = j + shift;
                tset(s.S, cap, j, x < wc ? gld(recS + x) : NINF);
                tset(s.F, cap, j, x < wc ? gld(recF + x) : NINF);
            }
        }
    }
    s.col = idx;
    wave_sync();
    return b;
}

// write a staged column (size + 5 cells: S, F and the flag byte per cell) to the arena as a record with org == trim;
// nothing waits on these stores.  The flags relate the column to its parent (staged in `par`, nullptr for the root):
// parent cell of window position trim + j is par cell j + dp.  Returns the record's cells per array.
MGX_DEV int32_t flush_column(Wave &w, const Staging &s, uint32_t cells_off, int32_t size, int32_t ge, const Staging *par,
                             int32_t par_size, int32_t dp, int32_t edge_score, uint8_t c, int32_t abs0, const uint8_t *q) {
    int32_t *rec = (int32_t *)uni((uint64_t)(w.cells + cells_off));
    const int32_t cap = uni(w.st_cap);
    const Tier tS = s.S, tF = s.F, tE = w.stE;
    const int32_t n = uni(size) + 5;
    const int32_t wc = (n + 3) & ~3;
    uint8_t *fb = (uint8_t *)(rec + 2 * wc);
    for (int32_t base = 0; base < wc; base += WAVE) {
        FOR_LANES(l) {
            int32_t j = base + l;
            if (j < wc) {
                const int32_t sv = j < n ? tget(tS, cap, j) : NINF, fv = j < n ? tget(tF, cap, j) : NINF;
                gst(rec + j, sv);
                gst(rec + wc + j, fv);
                uint32_t fl = 0;
                if (j < n) {
                    const int32_t e = tget(tE, cap, j), ep = j ? tget(tE, cap, j - 1) : NINF;
                    if (sv != NINF) fl |= CF_REAL;
                    if (sv == e) fl |= CF_S_IS_E;
                    if (e == ep + ge) fl |= CF_E_EXT;
                    if (sv == fv) fl |= CF_S_IS_F;
                    if (par) {
                        const int32_t jp = j + dp;                       // the parent's cell at the same window position
                        const int32_t sp1 = jp - 1 >= 0 ? st_S(*par, cap, par_size, jp - 1) : NINF;      // pos - 1 >= parent's trim
                        const int32_t fp = st_F(*par, cap, par_size, jp);
                        if (jp - 1 >= 0 && sv == sp1 + edge_score + profile_at(w, q, w.L, c, abs0 + j)) fl |= CF_MATCH;
                        if (jp - 1 >= 0 && sp1 != NINF) fl |= CF_SP_REAL;
                        if (fv == fp + edge_score + ge) fl |= CF_F_EXT;
                    }
                }
                gst(fb + j, (uint8_t)fl);
            }
        }
    }
    return wc;
}

// Compute one DP column into staging buffer `cb` from its parent in buffer `pb`
// (update_column :209-290 + extend_ins_end :293-328).  Returns the final size; pushes in w.tmp_pushes.
MGX_NI_G5 int32_t compute_column(Wave &w, const ExtenderState &E, int32_t prev_size, int32_t prev_trim, int pb, int cb,
                               int32_t prev_end, int32_t begin, int32_t size, uint8_t c, int32_t init_score,
                               int32_t offset, int32_t start, int32_t window_size, int32_t xdrop_cutoff) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    const int32_t go = uni(P.cfg.gap_open), ge = uni(P.cfg.gap_ext);
    const int32_t L = uni(w.L);
    prev_size = uni(prev_size); prev_trim = uni(prev_trim); prev_end = uni(prev_end); begin = uni(begin);
    size = uni(size); init_score = uni(init_score); offset = uni(offset); start = uni(start);
    window_size = uni(window_size); xdrop_cutoff = uni(xdrop_cutoff);
    const Staging par = w.st[pb];
    const Tier cS = w.st[cb].S, cF = w.st[cb].F, cE = w.stE;
    const int32_t cap = uni(w.st_cap);
#define CS(j) tget(cS, cap, (j))
#define CS_SET(j, v) tset(cS, cap, (j), (v))
#define CE(j) tget(cE, cap, (j))
#define CE_SET(j, v) tset(cE, cap, (j), (v))
#define CF(j) tget(cF, cap, (j))
#define CF_SET(j, v) tset(cF, cap, (j), (v))
    const int32_t trim = begin;
    const int32_t max_size = window_size + 1 - trim;
    MGX_ASSUME_LDS(MGX_SM_ROWS(w));                          // the score rows are a __shared__ array of the kernel
    const int8_t *row = MGX_SM_ROWS(w) + encode_char(c) * 128;    // profile_score_[encode(c)] (:38-59)
    const uint8_t *qq = (const uint8_t *)uni((uint64_t)E.q);
    // DPTColumn::create: size + 5 cells of ninf (we initialise everything update_column may touch)
    const int32_t init_n = imin(max_size, size) + 8;
    for (int32_t base = 0; base < init_n; base += WAVE) {
        FOR_LANES(l) { int32_t j = base + l; if (j < init_n) { CS_SET(j, NINF); CE_SET(j, NINF); CF_SET(j, NINF); } }
    }
    w.st[cb].col = -1;
    wave_sync();
    const int32_t n_prev = prev_end - trim;                 // update_column's prev_end
    const int32_t n_loop = (n_prev + 3) & ~3;               // lanes computed in blocks of 4
    const int32_t dp = trim - prev_trim;                    // S_prev_v = S_prev.data() + trim - trim_prev
    int32_t e_carry = NINF;                                 // E_v[base], E_v[0] = ninf
    int32_t tmax_carry = INT32_MIN;
    for (int32_t base = 0; base < n_loop; base += WAVE) {
        LV<int32_t> m, tval;
        FOR_LANES(l) {
            int32_t j = base + l;
            int32_t mm = NINF;
            if (j < n_loop) {
                int32_t match = NINF;
                if (j) {
                    int32_t ap = start + trim + j;
                    int32_t prof = (ap >= 1 && ap <= L) ? (int32_t)row[qq[ap - 1] & 127] : 0;
                    match = st_S(par, cap, prev_size, dp + j - 1) + prof + init_score;
                }
                int32_t del = NINF;
                if (offset > 1) del = imax(st_S(par, cap, prev_size, dp + j) + go, st_F(par, cap, prev_size, dp + j) + ge) + init_score;
                CF_SET(j, del);                             // F_v[j]
                mm = imax(match, del);
            }
            m[l] = mm;
            // E[j + 1] = max(E[j] + ge, m[j] + go)  ==  max_i<=j (m[i] + go + (j - i) ge)  or the E[0] chain
            tval[l] = j < n_loop ? mm + go - j * ge : INT32_MIN;
        }
        LV<int32_t> pm = wave_prefix_max(tval);
        LV<int32_t> enext;                                  // E[j + 1]
        FOR_LANES(l) {
            int32_t j = base + l;
            int32_t t = imax(pm[l], tmax_carry);
            int32_t from_open = t + j * ge;
            int64_t fe0 = (int64_t)NINF + (int64_t)(j + 1) * ge;     // E[0] = ninf extended j + 1 times
            int32_t from_e0 = fe0 < (int64_t)INT32_MIN ? INT32_MIN : (int32_t)fe0;
            enext[l] = j < n_loop ? imax(from_open, from_e0) : NINF;
        }
        LV<int32_t> ecur = wave_shift_up1(enext, e_carry);  // E[j]
        FOR_LANES(l) {
            int32_t j = base + l;
            if (j < n_loop) {
                CE_SET(j + 1, enext[l]);                    // E_v[j + 1]
                int32_t sv = imax(m[l], ecur[l]);
                CS_SET(j, sv > xdrop_cutoff - 1 ? sv : NINF);
            }
        }
        int32_t last_lane = imin(WAVE, n_loop - base) - 1;
        e_carry = wave_bcast(enext, last_lane);
        tmax_carry = imax(tmax_carry, wave_max(tval));
    }
    wave_sync();
    if (size > imax(1, n_prev)) {                            // scalar tail (:284-289)
        int32_t j = size - 1;
        int32_t ap = start + trim + j;
        int32_t prof = (ap >= 1 && ap <= L) ? (int32_t)row[qq[ap - 1] & 127] : 0;
        int32_t match = uni(imax(st_S(par, cap, prev_size, dp + j - 1) + init_score + prof, CE(j)));
        if (match >= xdrop_cutoff) CS_SET(j, match);
    }
    wave_sync();
    // extend_ins_end
    w.tmp_pushes = 0;
    if (size < max_size) {
        const int32_t ins_score = uni(imax(CS(size - 1) + go, CE(size - 1) + ge));
        if (ins_score >= xdrop_cutoff) {
            int32_t n_push = 1;
            int32_t room = max_size - (size + 1);
            if (ge == 0) {
                n_push += room;
            } else {
                int32_t v = ins_score;
                while (n_push - 1 < room && v + ge >= xdrop_cutoff) { v += ge; ++n_push; }
            }
            for (int32_t base = 0; base < n_push + 5; base += WAVE) {
                FOR_LANES(l) {
                    int32_t t = base + l;
                    if (t < n_push) {
                        int32_t v = ins_score + t * ge;
                        CS_SET(size + t, v); CE_SET(size + t, v); CF_SET(size + t, NINF);
                    } else if (t < n_push + 5) {             // padding after the new end is ninf
                        CS_SET(size + t, NINF); CE_SET(size + t, NINF); CF_SET(size + t, NINF);
                    }
                }
            }
            w.tmp_pushes = n_push;
            size += n_push;
        }
    }
    wave_sync();
    return size;
}

// children of a column (DefaultColumnExtender::call_outgoing :330-387, non-canonical graphs)
#undef CS
#undef CE
#undef CF
#undef CS_SET
#undef CE_SET
#undef CF_SET

// children of `node` in the graph the extender runs on (the graph itself, or its RCDBG view)
MGX_DEV int graph_children(Wave &w, const ExtenderState &E, const uint32_t node, uint32_t *nodes, uint8_t *chars, int32_t *scores) {
    const AlignParams &P = MGX_PARAMS_OF(w);
    uint64_t nn[5];
    uint32_t cc[5];
    int n;
    if (kWithPrimary && P.cfg.canonical >= 2) {
        // CanonicalDBG::call_outgoing_kmers over a PRIMARY graph (canon_graph.hpp)
        const uint32_t v = uni(node);
        uint8_t codes[4];
        bool sentinel;
        if (P.cfg.canonical == 3) {                            // from the precomputed reverse-complement tables
            n = canon_children_tables(P.g, v, nodes, codes, &sentinel, w.ctr);
            for (int t = 0; t < n; ++t) { chars[t] = decode_code(codes[t]); scores[t] = 0; }
            return n;
        }
        const bool is_rc = v > P.g.n;
        Spell sp = base_spelling(P.g, is_rc ? v - P.g.n : v, w.ctr);
        if (is_rc) sp = spell_reverse_complement(sp, (int32_t)P.g.k);
        n = canon_children(P.g, v, sp, nodes, codes, &sentinel, w.ctr);
        for (int t = 0; t < n; ++t) { chars[t] = decode_code(codes[t]); scores[t] = 0; }
        return n;
    }
    if (!E.rc_view) {
        // DBGSuccinct::call_outgoing_kmers (dbg_succinct.cpp:110-139); the node's own block is usually the
        // target block of the expansion that created it
        const DevGraph &g = P.g;
        const uint64_t v = node;
        Block cur;
        if ((uint32_t)(v >> 6) == uni(w.blk_cache_idx)) cur = uni_block(wave_blk_cache(w));
        else { ++w.ctr.rank_lines; cur = load_block_uniform(g, uni((uint32_t)(v >> 6))); }
        uint32_t wv = block_W(cur, (int)(v & 63));
        if (v > 1 && wv == 0) return 0;
        Block tgt;
        const uint64_t lst = uni(fwd_from<true>(g, v, cur, wv % SIGMA, tgt, w.ctr));
        wave_set_blk_cache(w, tgt, (uint32_t)(lst >> 6));
        uint64_t first = pred_last_from<true>(g, lst - 1, ((lst - 1) >> 6) == (lst >> 6) ? tgt : load_block_uniform(g, uni((uint32_t)((lst - 1) >> 6))), w.ctr) + 1;
        if (first < 2) first = 2;
        n = 0;
        Block b = tgt;
        uint32_t bi = (uint32_t)(lst >> 6);
        for (uint64_t i = first; i <= lst; ++i) {
            if ((uint32_t)(i >> 6) != bi) { bi = (uint32_t)(i >> 6); ++w.ctr.rank_lines; b = load_block_uniform(g, uni(bi)); }
            uint32_t c = block_W(b, (int)(i & 63)) % SIGMA;
            if (c != 0 && in_graph(g, i)) { if (n < 4) { nodes[n] = (uint32_t)i; chars[n] = decode_code(c); scores[n] = 0; } ++n; }
        }
        return n < 4 ? n : 4;
    }
    // RCDBG::call_outgoing_kmers (rc_dbg.hpp:88-99): parents with the complemented first character
    n = incoming<true>(P.g, node, nn, cc, w.ctr);
    int m = 0;
    for (int t = 0; t < n; ++t) {
        if (cc[t] == 0) continue;                             // complement('$') == '$' is dropped (:381-384)
        nodes[m] = (uint32_t)nn[t]; chars[m] = complement_char(decode_code(cc[t])); scores[m] = 0;
        ++m;
    }
    return m;
}


// (the seed's node list, spelling and offset come from the extension's loop state, where extend_begin() put them: the
// extension can then be advanced step by step without its caller's SeedRef)
MGX_DEV int call_outgoing(Wave &w, const ExtenderState &E, const ColMeta &col, bool force_fixed_seed,
                          uint32_t *nodes, uint8_t *chars, int32_t *scores) {
    const AlignParams &P = MGX_PARAMS_OF(w);
    const XState &xs = w.x;
    const uint32_t *seed_nodes = xs.seed_nodes;
    const uint8_t *seed_seq = xs.seed_seq;
    const int32_t k = (int32_t)uni(P.g.k);
    const int32_t next_offset = col.offset + 1;
    const int32_t seed_pos = next_offset - uni(xs.seed_off);
    const bool in_seed = seed_pos >= 0 && seed_pos < uni(xs.seed_seq_len);
    if (in_seed && next_offset < k) {
        nodes[0] = seed_nodes[0]; chars[0] = seed_seq[seed_pos]; scores[0] = 0;
        return 1;
    }
    if (in_seed && force_fixed_seed) {
        int32_t node_i = next_offset - k + 1;
        uint32_t next_node = seed_nodes[node_i];
        nodes[0] = next_node; chars[0] = seed_seq[seed_pos];
        scores[0] = next_node ? 0 : (!col.node ? P.cfg.gap_ext : P.cfg.gap_open);
        return 1;
    }
    return graph_children(w, E, col.node, nodes, chars, scores);
}

// ------------------------------------------------------------------------------------------------
// Register-resident column chains.
//
// Almost every column of an extension is the only child of the column computed just before it (the seed replay —
// call_outgoing :344-348 — and every non-branching stretch of the graph), and the frontier hands it straight back
// (:491-504: it is the unique top of the queue).  For such chains the column never touches the staging buffers or the
// frontier arrays: a window of FW = 4 x WAVE cells, four consecutive window positions per lane starting at `org` (a
// multiple of 4, org <= trim), holds S and F of the parent; chain_step() computes the child from it in registers
// (update_column :209-290 lane-exactly incl. the 4-wide overshoot, the scalar tail and extend_ins_end), scans it,
// commits it (one record store per array) and enters it into the convergence table.  Everything that does not fit
// the pattern — several children, an equal-score batch, a band wider than the window — goes through general_step()
// with the parent spilled to a staging buffer; results are identical by construction (the same arithmetic on the same
// values in the same order) and the CPU/GPU parity tests run with the chain path on and off (AlignParams::no_fast).
//
// extend() is a FLAT loop over single steps (pop / chain step / general step): the sub-wave groups of a wavefront
// work on different reads, and a loop nest would make a group that leaves a chain wait for every other group's
// chain to end.  All loop-carried state lives in XState (LDS), so each step is a small noinline function with its own
// register allocation instead of one huge function spilling to scratch.
// ------------------------------------------------------------------------------------------------
// select one of four per-lane values by a (group-uniform) slot number
MGX_DEV int32_t pick4(int32_t a, int32_t b, int32_t c, int32_t d, int32_t s) { return s == 0 ? a : s == 1 ? b : s == 2 ? c : d; }

// value of window position `pos` of a register column (must lie inside the window)
MGX_DEV int32_t reg_at(const LV<int32_t> &A0, const LV<int32_t> &A1, const LV<int32_t> &A2, const LV<int32_t> &A3, int32_t org, int32_t pos) {
    const int32_t x = pos - org;
    LV<int32_t> t;
    FOR_LANES(l) { t[l] = pick4(A0[l], A1[l], A2[l], A3[l], x & 3); }
    return wave_bcast(t, x >> 2);
}

MGX_DEV bool fast_fits(const ColMeta &c) { return (c.trim & 3) + c.size + 3 <= CHW; }

// load column `idx` (metadata c) into the chain window from its staging buffer or its arena record
MGX_DEV void fast_load(Wave &w, const ColMeta &c, int32_t idx, LV<int32_t> *S, LV<int32_t> *F) {
    XState &x = w.x;
    x.f_idx = idx; x.f_node = c.node; x.f_offset = c.offset; x.f_trim = c.trim; x.f_size = c.size; x.f_max_pos = c.max_pos;
    const int32_t org = c.trim & ~3;
    x.f_org = org;
    const int32_t n = c.size + 5;
    const int sb = (w.st[0].col == idx) ? 0 : (w.st[1].col == idx) ? 1 : -1;
    if (sb >= 0) {
        const Staging st = w.st[sb];
        const int32_t cap = w.st_cap;
        FOR_LANES(l) {
            for (int s = 0; s < 4; ++s) {
                const int32_t j = org + 4 * l + s - c.trim;
                const bool in = j >= 0 && j < n;
                S[s][l] = in ? tget(st.S, cap, j) : NINF;
                F[s][l] = in ? tget(st.F, cap, j) : NINF;
            }
        }
    } else {
        const int32_t wc = col_wc(c);
        const int32_t *recS = w.cells + c.cells, *recF = recS + wc;
        if (c.cells == NO_CELLS) w.status = ST_CAPACITY;          // cannot happen: a poppable column has its S / F record
        FOR_LANES(l) {
            for (int s = 0; s < 4; ++s) {
                const int32_t a = org + 4 * l + s, j = a - c.trim, rx = a - c.org;
                const bool in = j >= 0 && j < n && rx < wc;
                S[s][l] = in ? gld(recS + rx) : NINF;
                F[s][l] = in ? gld(recF + rx) : NINF;
            }
        }
    }
    x.f_max_val = reg_at(S[0], S[1], S[2], S[3], org, c.max_pos);
    wave_sync();
}

// the chain's parent goes back to staging buffer 0 in the general layout (cell j = window position trim + j)
MGX_DEV void fast_spill(Wave &w, const LV<int32_t> *S, const LV<int32_t> *F) {
    XState &x = w.x;
    Staging &st = w.st[0];
    const int32_t cap = w.st_cap;
    const int32_t n = x.f_size + 5;
    FOR_LANES(l) {
        for (int s = 0; s < 4; ++s) {
            const int32_t j = x.f_org + 4 * l + s - x.f_trim;
            if (j >= 0 && j < n) { tset(st.S, cap, j, S[s][l]); tset(st.F, cap, j, F[s][l]); }
        }
    }
    // cells the window does not hold: past its end, or below an origin that moved up with the band (all under the
    // cut-off, or they would have kept the origin down): ninf
    for (int32_t base = FW - (x.f_trim - x.f_org); base < n; base += WAVE) {
        FOR_LANES(l) { int32_t j = base + l; if (j < n) { tset(st.S, cap, j, NINF); tset(st.F, cap, j, NINF); } }
    }
    for (int32_t base = 0; base < x.f_org - x.f_trim; base += WAVE) {
        FOR_LANES(l) { int32_t j = base + l; if (j < x.f_org - x.f_trim && j < n) { tset(st.S, cap, j, NINF); tset(st.F, cap, j, NINF); } }
    }
    st.col = x.f_idx;
    if (w.st[1].col == x.f_idx) w.st[1].col = -1;
    wave_sync();
}

enum { XM_POP = 0, XM_FAST = 1 };
enum { FR_CONT = 0, FR_END = 1, FR_FALLBACK = 3, FR_STOP = 4, FR_ERROR = 5 };

} // namespace mgx
#include "lane_column.hpp"
namespace mgx {
#if MGX_WITH_LABELS
#include "label_sets.hpp"
#endif

// ---- general path: one popped column `i` with all its children (staging buffers, frontier arrays) ----
// returns 0, or 1 = the extension is over (capacity error; w.status says which)
MGX_DEV int general_step(Wave &w, ExtenderState &E, const int32_t i, const bool children_ready) {
    MGX_ASSUME_LDS(&w);
    MGX_ASSUME_LDS(&E);
    XState &x = w.x;
    const AlignParams &P = MGX_PARAMS_OF(w);
    const DevConfig &cfg = P.cfg;
    const DevLimits &lim = P.lim;
    const ColMeta col = uni_col(col_load(w, i));
    const int32_t max_columns = (int32_t)uni(lim.max_columns);
    const uint32_t cell_words = uni(lim.cell_words);
    const double rel_cutoff = cfg.rel_score_cutoff, max_nodes_per_char = cfg.max_nodes_per_seq_char, max_ram = cfg.max_ram_per_alignment;
    const int32_t xdrop = uni(cfg.xdrop);
    const int32_t start = x.start, window_size = x.window_size, qlen = x.qlen, psum_lin = x.psum_lin;
    const int32_t *psum = E.psum;
    const int32_t seed_off = x.seed_off, seed_seq_len = x.seed_seq_len, seed_offset = x.seed_off - 1;
    const bool force_fixed_seed = x.force_fixed != 0;
    const int pb = uni(stage_column(w, i, col));
    const Staging par = w.st[pb];
    const int32_t cap = uni(w.st_cap);
    const int32_t next_offset = col.offset + 1;
    const int32_t prev_xdrop_cutoff = x.xdrop_cutoff;       // global_xdrop: one shared cutoff
    const bool in_seed = (next_offset - seed_off) >= 0 && (next_offset - seed_off) < seed_seq_len;
    // early cut-offs when off the optimal path (:521-547)
    if (!children_ready && uni(st_S(par, cap, col.size, col.max_pos - col.trim)) < x.best_score) {
        double node_counter = (double)x.tsize;
        if (node_counter / (double)window_size >= max_nodes_per_char) { x.qn = 0; x.nn = 0; return 0; }
        if ((double)x.table_size_bytes / 1000000.0 > max_ram) { x.qn = 0; x.nn = 0; return 0; }
    }
    // band within the x-drop cutoff (:549-560)
    int32_t begin, prev_end;
    {
        int32_t b = col.size, e = 0;
        for (int32_t base = 0; base < col.size; base += WAVE) {
            LV<bool> inr;
            FOR_LANES(l) { int32_t j = base + l; inr[l] = j < col.size && tget(par.S, cap, j) >= prev_xdrop_cutoff; }
            uint64_t mk = wave_ballot(inr);
            if (mk) {
                if (b == col.size) b = base + ctz64(mk);
                e = base + 64 - clz64(mk);
            }
        }
        begin = b + col.trim; prev_end = e + col.trim;
    }
    if (prev_end <= begin) return 0;
    // the children list lives in the LDS control block: a private array indexed at run time would sit in scratch
    uint32_t *out_nodes = w.out_nodes;
    uint8_t *out_chars = w.out_chars;
    int32_t *out_scores = w.out_scores;
    if (!children_ready) x.n_valid = 0;                       // the children list is about to be overwritten
#if MGX_WITH_LABELS
    int n_out = children_ready ? x.f_n_out : uni(call_outgoing(w, E, col, force_fixed_seed, out_nodes, out_chars, out_scores));
    wave_sync();
    if (P.labeled) {
        n_out = lab_filter_children(w, i, n_out);           // LabeledExtender::call_outgoing
        if (w.status != ST_OK) return 1;
    }
#else
    const int n_out = children_ready ? x.f_n_out : uni(call_outgoing(w, E, col, force_fixed_seed, out_nodes, out_chars, out_scores));
    wave_sync();
#endif
    if (n_out == 0) {
        if (x.n_tips < max_columns) gst(w.tips + x.n_tips++, (uint32_t)i);
        return 0;
    }
    const int32_t end = imin(prev_end, window_size) + 1;
    const int cb = 1 - pb;
    for (int oi = 0; oi < n_out; ++oi) {
        const uint32_t next = uni(out_nodes[oi]);
        const uint8_t c = (uint8_t)uni((uint32_t)to_upper(out_chars[oi]));
        const int32_t score = uni(out_scores[oi]);
        if (x.tsize >= max_columns - 1) { w.status = ST_CAPACITY; return 1; }
        int32_t size0 = end - begin;
        uint32_t need = rec_words((uint32_t)(window_size + 1 - begin + 8));     // the column may grow to the window end
        if ((uint64_t)x.cell_top + need > cell_words) { w.status = ST_CAPACITY; return 1; }
        uint32_t table_cap_before = E.table_cap;
        if ((uint32_t)x.tsize == E.table_cap) E.table_cap = imax<uint32_t>(1u, 2 * E.table_cap);
        ++w.n_columns;
        const int32_t size = uni(compute_column(w, E, col.size, col.trim, pb, cb, prev_end, begin, size0, c, score, next_offset,
                                                start, window_size, x.xdrop_cutoff));
        const int32_t pushes = uni(w.tmp_pushes);
        ColMeta cur;
        cur.node = next; cur.parent = i; cur.cw = c; cur.org = begin; cur.offset = next_offset; cur.max_pos = begin; cur.trim = begin;
        cur.score = score; cur.cells = x.cell_top; cur.size = size;
        const uint32_t cur_cap3 = 3 * ref_capacity((uint32_t)size0, (uint32_t)pushes);
        const Tier cS = w.st[cb].S;
        // scan (:643-669): min_cell_score_, max_pos (closest to the diagonal), has_extension
        const int32_t diag_i = next_offset - seed_offset;
        bool has_extension = in_seed;
        const int32_t extension_cutoff =
            uni((int32_t)fma_f64((double)x.best_score, rel_cutoff, (double)x.partial_sum_offset));
        int32_t best_s = INT32_MIN, best_d = INT32_MAX, best_j = 0;
        int32_t min_cell_score = x.min_cell_score;
        for (int32_t base = 0; base < size; base += WAVE) {
            LV<int32_t> sv, mn, dd;
            LV<bool> ext;
            FOR_LANES(l) {
                int32_t j = base + l;
                int32_t v = j < size ? tget(cS, cap, j) : INT32_MIN;
                sv[l] = v;
                mn[l] = (j < size && v != NINF) ? v : INT32_MAX;
                ext[l] = j < size && v + (psum_lin ? (qlen - (start + begin + j)) * psum_lin : psum[start + begin + j]) >= extension_cutoff;
            }
            min_cell_score = imin(min_cell_score, wave_min(mn));
            if (wave_ballot(ext)) has_extension = true;
            // arg max in the order (S desc, |pos - diag| asc, j asc) (:647-650)
            const int32_t cm = wave_max(sv);
            FOR_LANES(l) { int32_t j = base + l; dd[l] = (j < size && sv[l] == cm) ? iabs(j + begin - diag_i) : INT32_MAX; }
            const int32_t cd = wave_min(dd);
            LV<bool> hit;
            FOR_LANES(l) { hit[l] = dd[l] == cd; }
            const int32_t cj = base + ctz64(wave_ballot(hit));
            if (cm > best_s || (cm == best_s && cd < best_d)) { best_s = cm; best_d = cd; best_j = cj; }
        }
        x.min_cell_score = min_cell_score;
        cur.max_pos = best_j + begin;
        const int32_t max_val = best_s;
        if ((!in_seed && max_val < x.xdrop_cutoff) || (!in_seed && !has_extension)) {
            // pop(table.size() - 1): the vector keeps its (possibly grown) capacity
            continue;
        }
        uint32_t table_sizediff = E.table_cap - table_cap_before;
        x.table_size_bytes += (uint64_t)136 * table_sizediff + (uint64_t)cur_cap3 * 4;
        if ((int32_t)((uint32_t)max_val - (uint32_t)x.xdrop_cutoff) > xdrop) x.xdrop_cutoff = max_val - xdrop;
        x.best_score = imax(x.best_score, max_val);
        // commit the column: metadata + cells go to the arena (nothing waits on them)
        const int32_t cur_wc = flush_column(w, w.st[cb], x.cell_top, size, uni(cfg.gap_ext), &w.st[pb], col.size, begin - col.trim,
                                            score, c, start + begin, E.q);
        cur.cw |= (uint32_t)cur_wc << 8;
        cur.base = 0;
        const int32_t my_idx = x.tsize;
        cur.self = my_idx;
        col_store(w, my_idx, cur);
#if MGX_WITH_LABELS
        if (P.labeled) w.col_lab[my_idx] = w.out_lab[oi];
#endif
        w.st[cb].col = my_idx;
        x.cell_top += rec_words((uint32_t)cur_wc);
        x.tsize = my_idx + 1;
        const int32_t vec_offset = start + begin - (begin ? 1 : 0);
        const int32_t skip = begin ? 0 : 1;
        int32_t converged = update_seed_filter(w, E, next, vec_offset, cS, skip, size - skip);
        if (w.status != ST_OK) return 1;
        if (converged != NINF) {
            uint64_t key = queue_key(converged, -iabs(cur.max_pos - diag_i), (uint32_t)my_idx);
            // next_nodes[0] is the first element popped into this batch (still there unless the batch
            // has been fully consumed, in which case next_nodes.size() == 0)
            if (x.nn && converged == key_score(qget(w.next_nodes, 0))) {
                qset(w.next_nodes, x.nn++, key);
                wave_sync();
            } else {
                frontier_push(w, key);
            }
        }
    }
    return 0;
}

// ---- chain path: the only child of the window column, computed, judged and committed in registers ----
// pS / pF: the window column (S and F of the chain's current parent), loop-carried registers of extend()
#ifdef MGX_CHAIN_PROBE
#define CH_T(slot) { const uint64_t t_ = xclock(); w.xcyc[slot] += t_ - tch; tch = t_; }
#else
#define CH_T(slot)
#endif
// The seed replay reads the seed's node list and spelling WAVE entries at a time (one per lane) and hands them out by
// lane broadcast: a column then starts without a dependent global load, whose wait would also drain the stores of the
// column before (one vmcnt for loads and stores on gfx9).
struct SeedRun { LV<uint32_t> nodes, chars; };

MGX_DEV int chain_step(Wave &w, ExtenderState &E, LV<int32_t> *pS, LV<int32_t> *pF, SeedRun &run) {
    MGX_ASSUME_LDS(&w);
    MGX_ASSUME_LDS(&E);
    XState &x = w.x;
    const int32_t xdrop_cutoff = x.xdrop_cutoff;
    const int32_t start = x.start, window_size = x.window_size, qlen = x.qlen;
    const int32_t go = x.go, ge = x.ge;
    const uint64_t tx1 = xclock();
#ifdef MGX_CHAIN_PROBE
    uint64_t tch = tx1;
#endif
    // early cut-offs when off the optimal path (:521-547)
    if (x.f_max_val < x.best_score) {
        double node_counter = (double)x.tsize;
        if (node_counter / (double)window_size >= x.max_nodes_per_char) return FR_STOP;
        if ((double)x.table_size_bytes / 1000000.0 > x.max_ram) return FR_STOP;
    }
    int32_t p_org = x.f_org;
#if defined(MGX_LANE_CHECK) && MGX_WAVE_EMU
    // host-model check of lane_column() against this function, cell by cell (FW == LFW builds only: 8 lanes per read)
    static_assert(FW == LFW, "the lane check runs in the 8-lane host model");
    int32_t lc_ps[LFW], lc_pf[LFW], lc_s[LFW], lc_f[LFW];
    FOR_LANES(l) { for (int s = 0; s < 4; ++s) { lc_ps[4 * l + s] = pS[s][l]; lc_pf[4 * l + s] = pF[s][l]; } }
    const int32_t lc_p_org = x.f_org, lc_min_cell = x.min_cell_score, lc_best = x.best_score;
    LaneColumnOut lc_out;
    auto lane_run = [&](int32_t next_offset_, int32_t score_, bool in_seed_, uint8_t c_) {
        LaneColumnIn li;
        for (int cx = 0; cx < LFW; ++cx) { lc_s[cx] = lc_ps[cx]; lc_f[cx] = lc_pf[cx]; }
        li.p_org = lc_p_org; li.p_trim = x.f_trim; li.p_size = x.f_size;
        li.xdrop_cutoff = xdrop_cutoff; li.start = start; li.window_size = window_size; li.qlen = qlen; li.go = go; li.ge = ge;
        li.next_offset = next_offset_; li.score = score_; li.in_seed = in_seed_;
        li.best_score = lc_best; li.min_cell_score = lc_min_cell; li.rel_cutoff = x.rel_cutoff;
        li.partial_sum_offset = x.partial_sum_offset; li.psum_lin = x.psum_lin; li.psum = E.psum; li.seed_off = x.seed_off;
        li.q = E.q; li.row = MGX_SM_ROWS(w) + encode_char(c_) * 128; li.band_given = 0; li.band_begin = li.band_prev_end = 0;
        return lane_column(li, lc_s, lc_f, lc_out);
    };
    auto lane_fail = [&](const char *what) { fprintf(stderr, "lane_column != chain_step: %s\n", what); abort(); };
#define MGX_LC(expr) expr
#else
#define MGX_LC(expr)
#endif
    // band within the x-drop cutoff (:549-560)
    int32_t begin, prev_end;
    {
        LV<int32_t> lo, hi;
        FOR_LANES(l) {
            int32_t a0 = INT32_MAX, a1 = INT32_MIN;
            for (int s = 0; s < 4; ++s) {
                const int32_t a = p_org + 4 * l + s, j = a - x.f_trim;
                if (j >= 0 && j < x.f_size && pS[s][l] >= xdrop_cutoff) { a0 = imin(a0, a); a1 = imax(a1, a + 1); }
            }
            lo[l] = a0; hi[l] = a1;
        }
        begin = wave_min(lo); prev_end = wave_max(hi);
    }
    if (prev_end <= begin) { MGX_LC(if (lane_run(x.f_offset + 1, 0, false, 'A') != LC_EMPTY_BAND) lane_fail("empty band");) return FR_END; }
    // the child (call_outgoing :330-387)
    const int32_t next_offset = x.f_offset + 1;
    const int32_t seed_pos = next_offset - x.seed_off;
    const bool in_seed = seed_pos >= 0 && seed_pos < x.seed_seq_len;
    const int32_t k = x.k;
    uint32_t next;
    uint8_t c;
    int32_t score;
    if (in_seed && (next_offset < k || x.force_fixed)) {
        // the seed replay (:344-372): the node sequence and spelling of the seed, no graph access
        const int32_t ni = next_offset < k ? 0 : next_offset - k + 1;
        if ((uint32_t)(ni - x.sn_base) >= (uint32_t)WAVE) {
            const int32_t last = x.seed_n_nodes - 1;
            FOR_LANES(l) { run.nodes[l] = gld(x.seed_nodes + imin(ni + l, last)); }
            x.sn_base = ni;
        }
        next = wave_bcast(run.nodes, ni - x.sn_base);
        if (x.seq_lds) {
            c = lds_u8(x.seed_seq + seed_pos);
        } else {
            if ((uint32_t)(seed_pos - x.sc_base) >= (uint32_t)WAVE) {
                const int32_t last = x.seed_seq_len - 1;
                FOR_LANES(l) { run.chars[l] = gld(x.seed_seq + imin(seed_pos + l, last)); }
                x.sc_base = seed_pos;
            }
            c = (uint8_t)wave_bcast(run.chars, seed_pos - x.sc_base);
        }
        score = (next_offset < k || next) ? 0 : (!x.f_node ? ge : go);
    } else {
        int n_out;
        if (x.n_valid && x.n_for == x.f_idx) {
            n_out = x.n_count;                                    // enumerated during the previous step
        } else {
            n_out = graph_children(w, E, x.f_node, w.out_nodes, w.out_chars, w.out_scores);
            wave_sync();
        }
        x.n_valid = 0;
        if (n_out == 0) {
            if (x.n_tips < x.max_columns) gst(w.tips + x.n_tips++, (uint32_t)x.f_idx);
            return FR_END;
        }
        x.f_n_out = n_out;
        if (n_out != 1) return FR_FALLBACK;
        next = w.out_nodes[0]; c = w.out_chars[0]; score = w.out_scores[0];
    }
    c = to_upper(c);
    // the convergence table's slot for the child's node: issued now, consumed after the column is computed
    const uint32_t ckey = next;
    const uint32_t chash = conv_hash(ckey, E.conv.cap - 1);
    uint64_t cse = 0;
    if (next) cse = gld(E.conv.tab + E.conv.base + chash);
    CH_T(0)
    // Expansion of the child's own node one column ahead (forward graph only): if this child continues the chain and
    // its successor comes from the graph, the two dependent loads of fwd() (select hint, target block) travel while
    // this column is computed.  Same primitives, same results; a speculation that does not pan out is simply dropped.
    bool pf = false;
    uint32_t pf_r = 0;
    uint32_t pf_hint = 0;
    int pf_zero = 0;
    {
        const int32_t no2 = next_offset + 1, sp2 = no2 - x.seed_off;
        const bool replay2 = sp2 >= 0 && sp2 < x.seed_seq_len && (no2 < k || x.force_fixed);
        if (!x.rc && next > 1 && !replay2 && (next >> 6) == w.blk_cache_idx) {
                    const DevGraph &g = MGX_PARAMS_OF(w).g;
            const Block cur = wave_blk_cache(w);
            const uint32_t wv = block_W(cur, (int)(next & 63));
            if (wv == 0) {
                pf = true; pf_zero = 1;                          // sink dummy: no children (dbg_succinct.cpp:113)
            } else {
                pf_r = nf_of(g, wv % SIGMA) + block_rank_W(cur, (int)(next & 63), wv % SIGMA, (next >> 6) == 0);
                if (pf_r) { pf = true; pf_hint = gld(g.last_hint + ((pf_r - 1) >> 6)); }
            }
        }
    }
    const int32_t end = imin(prev_end, window_size) + 1;
    const int32_t size0 = end - begin;
    const int32_t max_size = window_size + 1 - begin;
    const int32_t n_prev = prev_end - begin, n_loop = (n_prev + 3) & ~3;
    const int32_t org = begin & ~3;
    // from here on a fallback hands the one child over through the children list (general_step reads it from there)
    x.f_n_out = 1;
    w.out_nodes[0] = next; w.out_chars[0] = c; w.out_scores[0] = score;
    if ((begin - org) + imax(n_loop, size0) > CHW) {
        MGX_LC(if (lane_run(next_offset, score, in_seed, c) != LC_FALLBACK) lane_fail("window fallback");)
        wave_sync();
        return FR_FALLBACK;
    }
    if (x.tsize >= x.max_columns - 1) { w.status = ST_CAPACITY; return FR_ERROR; }
    // (room for the record a column that stays behind in the frontier writes below: always a whole chain window of CHW cells, which
    // near the read's end is MORE than the window-sized record the general path would need — with only the latter tested, such a
    // record ran past the cell arena into the first column slots when the arena was nearly full; found by the fuzz campaign's
    // address-sanitized run; tests/test_fuzz_smoke.py keeps the world)
    if ((uint64_t)x.cell_top + imax<uint32_t>(rec_words((uint32_t)(window_size + 1 - begin + 8)), rec_words((uint32_t)CHW)) > x.cell_words) { w.status = ST_CAPACITY; return FR_ERROR; }
    // move the parent window to the child's origin (whole lanes)
    // (the parent's cell just below the new origin — under the cut-off, but the match flag of the origin cell compares
    // against its true value)
    int32_t p_below = NINF;
    if (org != p_org) {
        if (org - 1 >= p_org) p_below = reg_at(pS[0], pS[1], pS[2], pS[3], p_org, org - 1);
        const int32_t sh = (org - p_org) >> 2;
        for (int s = 0; s < 4; ++s) { pS[s] = wave_shift_down(pS[s], sh, NINF); pF[s] = wave_shift_down(pF[s], sh, NINF); }
        p_org = org;
    }
    CH_T(1)
    // update_column (:209-290): cell j = a - begin; j in [0, n_loop) is computed in blocks of four lanes
    const int8_t *row = MGX_SM_ROWS(w) + encode_char(c) * 128; // a __shared__ array of the kernel
    const uint8_t *qq = E.q;
    const bool q_lds = w.q_lds != 0;
    const LV<int32_t> Sm1_0 = wave_shift_up1(pS[3], p_below);     // parent at a - 1 for slot 0
    LV<int32_t> cS[4], cF[4], cE[4], mraw[4], tv[4], mm[4];
    // profile_score_[c][start + a] (:38-59): the query character before window position a against the column's
    // character; 0 outside the query.  All loads are issued back to back at clamped addresses (no per-cell branches:
    // a branch around a load costs a wait each) and masked afterwards.
    LV<int32_t> prof[4];
    {
        LV<uint32_t> qc[4];
        const int32_t hi = qlen - 1;
        if (q_lds) {
            FOR_LANES(l) { for (int s = 0; s < 4; ++s) qc[s][l] = lds_u8(qq + imax(0, imin(hi, start + org + 4 * l + s - 1))); }
        } else {
            FOR_LANES(l) { for (int s = 0; s < 4; ++s) qc[s][l] = gld(qq + imax(0, imin(hi, start + org + 4 * l + s - 1))); }
        }
        FOR_LANES(l) { for (int s = 0; s < 4; ++s) prof[s][l] = (int32_t)lds_i8(row + (qc[s][l] & 127)); }
        FOR_LANES(l) {
            for (int s = 0; s < 4; ++s) {
                const int32_t ap = start + org + 4 * l + s;
                prof[s][l] = (ap >= 1 && ap <= qlen) ? prof[s][l] : 0;
            }
        }
    }
    FOR_LANES(l) {
        for (int s = 0; s < 4; ++s) {
            const int32_t a = org + 4 * l + s, j = a - begin;
            const int32_t sm1 = s == 0 ? Sm1_0[l] : pS[s - 1][l];
            mraw[s][l] = sm1 + prof[s][l] + score;                      // S_prev[j - 1] + profile + init_score
            const bool in = j >= 0 && j < n_loop;
            int32_t del = NINF;
            if (next_offset > 1) del = imax(pS[s][l] + go, pF[s][l] + ge) + score;
            const int32_t match = j >= 1 ? mraw[s][l] : NINF;
            const int32_t m = imax(match, del);
            cF[s][l] = in ? del : NINF;
            mm[s][l] = m;
            tv[s][l] = in ? m + go - j * ge : INT32_MIN;
        }
    }
    // E[j + 1] = max(E[j] + ge, m[j] + go) == max_{i <= j}(m[i] + go + (j - i) ge) or the E[0] = ninf chain
    LV<int32_t> tot;
    FOR_LANES(l) {
        tv[1][l] = imax(tv[1][l], tv[0][l]); tv[2][l] = imax(tv[2][l], tv[1][l]); tv[3][l] = imax(tv[3][l], tv[2][l]);
        tot[l] = tv[3][l];
    }
    const LV<int32_t> pm = wave_prefix_max(tot);
    const LV<int32_t> ex = wave_shift_up1(pm, INT32_MIN);             // everything in earlier lanes
    LV<int32_t> en[4];                                                // E[j + 1] per cell j
    FOR_LANES(l) {
        for (int s = 0; s < 4; ++s) {
            const int32_t a = org + 4 * l + s, j = a - begin;
            const int32_t t = imax(tv[s][l], ex[l]);
            const int32_t from_open = t + j * ge;
            // E[0] = ninf extended j + 1 times, saturating at INT32_MIN (NINF = INT32_MIN + 100; |(j + 1) ge| < 2^16 here)
            const int32_t dec = (j + 1) * ge;
            const int32_t from_e0 = dec < -100 ? INT32_MIN : NINF + dec;
            en[s][l] = (j >= 0 && j < n_loop) ? imax(from_open, from_e0) : NINF;
        }
    }
    const LV<int32_t> en_up = wave_shift_up1(en[3], NINF);
    FOR_LANES(l) {
        for (int s = 0; s < 4; ++s) {
            const int32_t a = org + 4 * l + s, j = a - begin;
            const int32_t ecur = s == 0 ? en_up[l] : en[s - 1][l];    // E[j] (ninf at j == 0 and wherever nothing was computed)
            cE[s][l] = (j >= 0 && j <= n_loop) ? ecur : NINF;
            int32_t sv = NINF;
            if (j >= 0 && j < n_loop) { sv = imax(mm[s][l], ecur); if (!(sv > xdrop_cutoff - 1)) sv = NINF; }
            cS[s][l] = sv;
        }
    }
    // scalar tail (:284-289)
    if (size0 > imax(1, n_prev)) {
        FOR_LANES(l) {
            for (int s = 0; s < 4; ++s) {
                const int32_t j = org + 4 * l + s - begin;
                if (j == size0 - 1) {
                    const int32_t match = imax(mraw[s][l], cE[s][l]);
                    if (match >= xdrop_cutoff) cS[s][l] = match;
                }
            }
        }
    }
    CH_T(2)
    // extend_ins_end (:293-328)
    int32_t size = size0, pushes = 0;
    if (size0 < max_size) {
        const int32_t lastS = reg_at(cS[0], cS[1], cS[2], cS[3], org, begin + size0 - 1);
        const int32_t lastE = reg_at(cE[0], cE[1], cE[2], cE[3], org, begin + size0 - 1);
        const int32_t ins_score = imax(lastS + go, lastE + ge);
        if (ins_score >= xdrop_cutoff) {
            // while (v + ge >= cutoff) { v += ge; ++n_push; } bounded by the window end, in closed form (ge < 0)
            const int32_t room = max_size - (size0 + 1);
            int32_t n_push = 1 + room;
            // (the chain path runs with |cutoff| <= 30000 and scores below 2^24: the difference fits 32 bits)
            if (ge != 0) n_push = 1 + imin(room, (int32_t)((uint32_t)(ins_score - xdrop_cutoff) / (uint32_t)(-ge)));
            if ((begin - org) + size0 + n_push > CHW) {
                // the parent window has moved: keep it consistent for the spill
                x.f_org = p_org;
                MGX_LC(if (lane_run(next_offset, score, in_seed, c) != LC_FALLBACK) lane_fail("ins_end fallback");)
                return FR_FALLBACK;
            }
            FOR_LANES(l) {
                for (int s = 0; s < 4; ++s) {
                    const int32_t t = org + 4 * l + s - begin - size0;
                    if (t >= 0 && t < n_push) { const int32_t v = ins_score + t * ge; cS[s][l] = v; cE[s][l] = v; cF[s][l] = NINF; }
                    else if (t >= n_push) { cS[s][l] = NINF; cE[s][l] = NINF; cF[s][l] = NINF; }      // padding after the new end
                }
            }
            pushes = n_push;
            size += n_push;
        }
    }
    uint32_t table_cap_before = E.table_cap;
    if ((uint32_t)x.tsize == E.table_cap) E.table_cap = imax<uint32_t>(1u, 2 * E.table_cap);
    ++w.n_columns;
    ++w.n_fast_columns;
    Block pf_blk;
    pf_blk.cum[0] = pf_blk.cum[1] = pf_blk.cum[2] = pf_blk.cum[3] = 0; pf_blk.last_cum = 0; pf_blk.cum0 = 0;
    pf_blk.last_bits = 0; pf_blk.p0 = pf_blk.p1 = pf_blk.p2 = pf_blk.pf = 0;
    if (pf && !pf_zero) pf_blk = load_block_uniform(MGX_PARAMS_OF(w).g, pf_hint);
    // scan (:643-669): min_cell_score_, max_pos (closest to the diagonal), has_extension
    const int32_t psum_lin = x.psum_lin;
    const int32_t *psum = E.psum;
    const int32_t diag_i = next_offset - (x.seed_off - 1);
    const int32_t extension_cutoff = (int32_t)fma_f64((double)x.best_score, x.rel_cutoff, (double)x.partial_sum_offset);
    LV<int32_t> lmax, lmin;
    LV<bool> lext;
    FOR_LANES(l) {
        int32_t mx = INT32_MIN, mn = INT32_MAX;
        bool ext = false;
        for (int s = 0; s < 4; ++s) {
            const int32_t a = org + 4 * l + s, j = a - begin;
            if (j >= 0 && j < size) {
                const int32_t v = cS[s][l];
                mx = imax(mx, v);
                if (v != NINF) mn = imin(mn, v);
                ext |= v + (psum_lin ? (qlen - (start + a)) * psum_lin : psum[start + a]) >= extension_cutoff;
            }
        }
        lmax[l] = mx; lmin[l] = mn; lext[l] = ext;
    }
    x.min_cell_score = imin(x.min_cell_score, wave_min(lmin));
    const bool has_extension = in_seed || wave_ballot(lext) != 0;
    const int32_t max_val = wave_max(lmax);
    // arg max in the order (S desc, |pos - diag| asc, j asc) (:647-650): one reduction over (distance, position)
    LV<int32_t> lkey;
    FOR_LANES(l) {
        int32_t kk = INT32_MAX;
        for (int s = 0; s < 4; ++s) {
            const int32_t a = org + 4 * l + s, j = a - begin;
            if (j >= 0 && j < size && cS[s][l] == max_val) kk = imin(kk, (iabs(a - diag_i) << 12) | j);
        }
        lkey[l] = kk;
    }
    const int32_t max_pos = begin + (wave_min(lkey) & 4095);          // j < FW <= 256, distance < 2^19 (Lmax <= 32704)
    if ((!in_seed && max_val < xdrop_cutoff) || (!in_seed && !has_extension)) {      // pop(table.size() - 1)
        MGX_LC(if (lane_run(next_offset, score, in_seed, c) != LC_POP || lc_out.min_cell_score != x.min_cell_score) lane_fail("pop");)
        return FR_END;
    }
    const uint32_t cur_cap3 = 3 * ref_capacity((uint32_t)size0, (uint32_t)pushes);
    x.table_size_bytes += (uint64_t)136 * (E.table_cap - table_cap_before) + (uint64_t)cur_cap3 * 4;
    if ((int32_t)((uint32_t)max_val - (uint32_t)xdrop_cutoff) > x.xdrop) x.xdrop_cutoff = max_val - x.xdrop;
    x.best_score = imax(x.best_score, max_val);
    CH_T(3)
    // --- everything that LOADS comes first (a wait on a load also waits for every store issued before it) ---
    // update_seed_filter (:100-156), resolution: cell at window position a (j in [skip, size)) is query position
    // start + a - 1 of the node's vector.  A position outside the vector's old range holds ninf by definition, so
    // nothing is read back that this step writes.
    const int32_t my_idx = x.tsize;
    const int32_t skip = begin ? 0 : 1;
    const int32_t cn = size - skip;
    const int32_t query_start = start + begin - (begin ? 1 : 0);
    enum { CV_NONE = 0, CV_INSERT = 1, CV_BELOW = 2, CV_ABOVE = 3, CV_MERGE = 4 };
    int cv_mode = CV_NONE;
    uint32_t cv_slot = 0;
    int32_t cv_rec = -1;                       // pool record of a node seen before
    ConvRec crec;
    crec.off = 0; crec.start = 0; crec.len = 0; crec.cap = 0;
    int32_t cv_vstart = 0, cv_vlen = 0;
    int32_t *cv_vec = nullptr;
    LV<int32_t> mv[4];                         // CV_MERGE: value to store per cell ...
    LV<uint32_t> mdo;                          // ... and which cells are stored at all
    int32_t converged;
    {
        LV<int32_t> cm;
        FOR_LANES(l) {
            int32_t m = INT32_MIN;
            for (int s = 0; s < 4; ++s) {
                const int32_t j = org + 4 * l + s - begin;
                if (j >= skip && j < size) m = imax(m, cS[s][l]);
                mv[s][l] = cS[s][l];
            }
            cm[l] = m;
            mdo[l] = 0;
        }
        converged = wave_max(cm);
        if (next != 0 && !MGX_ABLATED(w, 1u)) {
            bool found;
            cv_slot = conv_probe_from(E.conv, ckey, chash, cse, found);
            if (!found) {
                cv_mode = CV_INSERT;
            } else {
                // a node seen before: its entry is written a second time, so an alias is copied out to the pool first
                if (cs_kind(cse) == 0) cv_rec = conv_materialize(w, E.conv, cv_slot, cse, crec);
                else { cv_rec = (int32_t)cs_idx(cse); crec = conv_rec_load(w, E.conv, (uint32_t)cv_rec); }
                if (cv_rec < 0) return FR_ERROR;
                cv_vec = conv_vec(E.conv, crec);
                cv_vstart = crec.start; cv_vlen = crec.len;
                if (query_start + cn <= cv_vstart) cv_mode = CV_BELOW;
                else if (query_start >= cv_vstart + cv_vlen) cv_mode = CV_ABOVE;
                else {
                    cv_mode = CV_MERGE;
                    const double rel = x.rel_cutoff;
                    LV<int32_t> xm, vold[4];
                    const int32_t vlast = cv_vstart + cv_vlen - 1;
                    FOR_LANES(l) { for (int s = 0; s < 4; ++s) vold[s][l] = gld(cv_vec + imax(cv_vstart, imin(vlast, start + org + 4 * l + s - 1))); }
                    FOR_LANES(l) {
                        int32_t m = NINF;
                        uint32_t dm = 0;
                        for (int s = 0; s < 4; ++s) {
                            const int32_t a = org + 4 * l + s, j = a - begin;
                            const int32_t pos = start + a - 1;
                            const bool cell = j >= skip && j < size;
                            const bool old = pos >= cv_vstart && pos <= vlast;
                            const int32_t sv = cS[s][l];
                            const int32_t vv = old ? vold[s][l] : NINF;
                            const bool up = (double)sv > (double)vv * rel;
                            const int32_t nv = up ? imax(vv, sv) : NINF;
                            mv[s][l] = nv;
                            if (cell && (up || !old)) dm |= 1u << s;
                            if (cell && up) m = imax(m, nv);
                        }
                        xm[l] = m;
                        mdo[l] = dm;
                    }
                    converged = wave_max(xm);
                }
            }
        }
    }
    CH_T(4)
    // the children of this column's node, enumerated ahead (see above): consume the target block now
    int pf_n = -1;
    if (pf) {
        if (pf_zero) {
            pf_n = 0;
        } else if (pf_blk.last_cum < pf_r && pf_blk.last_cum + (uint32_t)popc64(pf_blk.last_bits) >= pf_r) {
            const DevGraph &g = MGX_PARAMS_OF(w).g;
            const int lj = select64(pf_blk.last_bits, (int)(pf_r - pf_blk.last_cum));     // fwd(): last edge of the target node
            const uint64_t m = lj > 0 ? (pf_blk.last_bits & mask_upto(lj - 1)) : 0;       // pred_last(lst - 1) inside this block
            if (m) {
                const uint64_t base = (uint64_t)pf_hint << 6;
                int fj = 63 - clz64(m) + 1;
                if (base + fj < 2) fj = (int)(2 - base);
                pf_n = 0;
                for (int j = fj; j <= lj; ++j) {
                    const uint32_t cc = block_W(pf_blk, j) % SIGMA;
                    if (cc != 0 && in_graph(g, base + j)) {
                        if (pf_n < 4) { w.out_nodes[pf_n] = (uint32_t)(base + j); w.out_chars[pf_n] = decode_code(cc); w.out_scores[pf_n] = 0; }
                        ++pf_n;
                    }
                }
                if (pf_n > 4) pf_n = 4;
                wave_set_blk_cache(w, pf_blk, pf_hint);
                ++w.ctr.select_lines;
            }
        }
    }
    CH_T(5)
    // --- stores only from here on ---
    // Will the frontier hand this column straight back (:491-504: it is the unique maximum)?  Then nothing ever reloads
    // it and its slot is all it leaves behind; otherwise it also gets an S / F record.
    const bool chain_on = converged != NINF && x.nn == 0 && (x.qn == 0 || converged > x.q_top)
                          && (begin & 3) + size + 3 <= CHW;
    const bool deferred = converged != NINF && !chain_on;
    // commit: the column's slot (metadata, flag byte per cell, 16-bit S) ...
    ColMeta cur;
    cur.node = next; cur.parent = x.f_idx; cur.cw = (uint32_t)c | ((uint32_t)CHW << 8) | CW_CHAIN; cur.org = org; cur.offset = next_offset;
    cur.max_pos = max_pos; cur.trim = begin; cur.score = score; cur.size = size;
    cur.base = max_val == NINF ? 0 : max_val;
    cur.cells = deferred ? x.cell_top : NO_CELLS;
    cur.self = my_idx;
    if (!MGX_ABLATED(w, 2u)) {
        ColSlot *slot = w.cols + my_idx;
        const LV<int32_t> e_up = wave_shift_up1(cE[3], NINF);
        const int32_t ptrim = x.f_trim;
        LV<uint32_t> fwv;
        LV<int32_t> hs[4];
        LV<bool> blocks;                           // what keeps this lane's cells out of the compact form
        FOR_LANES(l) {
            uint32_t fw = 0;
            bool blk = false;
            for (int s = 0; s < 4; ++s) {
                const int32_t a = org + 4 * l + s, j = a - begin;
                const int32_t ep = j <= 0 ? NINF : (s == 0 ? e_up[l] : cE[s - 1][l]);
                const int32_t sv = cS[s][l], fv = cF[s][l];
                const int32_t sp1 = s == 0 ? Sm1_0[l] : pS[s - 1][l];       // parent at a - 1 (ninf outside the parent column)
                const bool pin = a - 1 >= ptrim;                           // pos - 1 >= the parent's trim (:963)
                uint32_t fl = 0;
                if (sv != NINF) fl |= CF_REAL;
                if (sv == cE[s][l]) fl |= CF_S_IS_E;
                if (cE[s][l] == ep + ge) fl |= CF_E_EXT;
                if (pin && sv == mraw[s][l]) fl |= CF_MATCH;
                if (sv == fv) fl |= CF_S_IS_F;
                if (fv == pF[s][l] + score + ge) fl |= CF_F_EXT;
                if (pin && sp1 != NINF) fl |= CF_SP_REAL;
                fw |= fl << (8 * s);
                hs[s][l] = sv == NINF ? (int32_t)S16_NINF : sv - cur.base;
                // compact form: cells from CCELLS on must hold neither S nor F (then no trace can reach them and their flag
                // bytes are never consulted), and S must fit 8 bits
                if (4 * l + s >= CCELLS) blk |= sv != NINF || fv != NINF;
                else blk |= sv != NINF && (sv - cur.base < -127 || sv - cur.base > 127);
            }
            fwv[l] = fw;
            blocks[l] = blk;
        }
#if defined(MGX_LANE_CHECK) && MGX_WAVE_EMU
        {
            if (lane_run(next_offset, score, in_seed, c) != LC_OK) lane_fail("return code of a committed column");
            if (lc_out.begin != begin || lc_out.size != size || lc_out.size0 != size0 || lc_out.pushes != pushes || lc_out.org != org) lane_fail("geometry");
            if (lc_out.max_val != max_val || lc_out.max_pos != max_pos || lc_out.has_extension != has_extension
                    || lc_out.min_cell_score != x.min_cell_score) lane_fail("scan");
            if (cv_mode != CV_MERGE && lc_out.converged != converged) lane_fail("converged");
            FOR_LANES(l) {
                for (int s = 0; s < 4; ++s) {
                    const int xx = 4 * l + s;
                    if (lc_s[xx] != cS[s][l] || lc_f[xx] != cF[s][l]) lane_fail("cells");
                    // (a cell that holds no S is never consulted beyond its CF_REAL bit: lane_column leaves the bytes of the
                    // blocks behind the band at 0)
                    const uint8_t fl_lane = (uint8_t)(lc_out.fw[xx >> 2] >> (8 * (xx & 3))), fl_grp = (uint8_t)(fwv[l] >> (8 * s));
                    if ((fl_grp & CF_REAL) ? fl_lane != fl_grp : (fl_lane & CF_REAL) != 0) lane_fail("flags");
                }
            }
        }
#endif
        const uint32_t ccode = encode_char(c);
        const bool compact = !deferred && !MGX_PARAMS_OF(w).no_compact && wave_ballot(blocks) == 0 && next_offset <= 0xFFFF && begin <= 0x7FFF
                             && cur.base > -(1 << 21) && cur.base < (1 << 21) && ccode <= 4 && decode_code(ccode) == c
                             && (score == 0 || score == go || score == ge) && size <= 31 && x.f_idx < (1 << 24);
        if (compact) {
            const uint32_t sc = score == 0 ? 0u : (score == go ? 1u : 2u);
            const uint32_t m0 = next, m1 = (uint32_t)x.f_idx | (ccode << 24) | (sc << 27) | (1u << 30);
            const uint32_t m2 = ((uint32_t)cur.base & 0x3FFFFF) | ((uint32_t)size << 22) | ((uint32_t)(max_pos - begin) << 27);
            const uint32_t m3 = (uint32_t)next_offset | ((uint32_t)begin << 16);
            FOR_LANES(l) {
                if (l < 8) {
                    uint2 v;
                    if (l < 6) {
                        v.x = fwv[l];
                        v.y = ((uint32_t)hs[0][l] & 0xFF) | (((uint32_t)hs[1][l] & 0xFF) << 8) | (((uint32_t)hs[2][l] & 0xFF) << 16) | ((uint32_t)hs[3][l] << 24);
                        // (a 16-bit ninf code truncates to 0x00: restore the 8-bit one)
                        if (hs[0][l] == (int32_t)S16_NINF) v.y = (v.y & ~0xFFu) | 0x80u;
                        if (hs[1][l] == (int32_t)S16_NINF) v.y = (v.y & ~0xFF00u) | 0x8000u;
                        if (hs[2][l] == (int32_t)S16_NINF) v.y = (v.y & ~0xFF0000u) | 0x800000u;
                        if (hs[3][l] == (int32_t)S16_NINF) v.y = (v.y & 0x00FFFFFFu) | 0x80000000u;
                    } else {
                        v.x = l == 6 ? m0 : m2;
                        v.y = l == 6 ? m1 : m3;
                    }
                    gst((uint2 *)slot + (l < 6 ? 2 + l : l - 6), v);
                }
            }
            cur.cw = (uint32_t)c | ((uint32_t)CCELLS << 8) | CW_CHAIN | CW_COMPACT;
        } else {
            int16_t *srow = w.cols_s16 + (int64_t)my_idx * FWS;
            FOR_LANES(l) {
                if (4 * l < CHW) {
                    gst((uint32_t *)slot->flags + l, fwv[l]);
                    uint2 hv;
                    hv.x = ((uint32_t)hs[0][l] & 0xFFFF) | ((uint32_t)hs[1][l] << 16);
                    hv.y = ((uint32_t)hs[2][l] & 0xFFFF) | ((uint32_t)hs[3][l] << 16);
                    gst((uint2 *)srow + l, hv);
                }
            }
            col_store(w, my_idx, cur);
        }
        // ... and, for a column that stays behind in the frontier, its window as an S / F record
        if (deferred) {
            int32_t *rec = w.cells + x.cell_top;
#if MGX_WAVE_EMU
            if ((uint64_t)x.cell_top + rec_words((uint32_t)CHW) > x.cell_words) { fprintf(stderr, "chain_step: S / F record past the cell arena\n"); abort(); }
#endif
            FOR_LANES(l) {
                if (4 * l < CHW) {
                    gst4(rec + 4 * l, cS[0][l], cS[1][l], cS[2][l], cS[3][l]);
                    gst4(rec + CHW + 4 * l, cF[0][l], cF[1][l], cF[2][l], cF[3][l]);
                }
            }
            x.cell_top += rec_words((uint32_t)CHW);
        }
    }
    CH_T(6)
#if MGX_WITH_LABELS
    // LabeledExtender::call_outgoing, the only-child case (aligner_labeled.cpp:224-231): the parent's labels, unseen
    if (MGX_PARAMS_OF(w).labeled) w.col_lab[my_idx] = w.col_lab[x.f_idx];
#endif
    x.tsize = my_idx + 1;
    // update_seed_filter, the stores
    if (cv_mode == CV_INSERT) {
        if (x.alias_ok) {
            // the node's first column: the table entry points at the column's own S window (its slot, just written)
            if (!conv_claim(w, E.conv, cv_slot, ckey, 0u, (uint32_t)my_idx)) return FR_ERROR;
            cv_mode = CV_NONE;
        } else {
            cv_rec = conv_new_rec(w, E.conv, query_start, cn, crec);
            if (cv_rec < 0) return FR_ERROR;
            if (!conv_claim(w, E.conv, cv_slot, ckey, 1u, (uint32_t)cv_rec)) return FR_ERROR;
            cv_vec = conv_vec(E.conv, crec);
        }
    } else if (cv_mode != CV_NONE) {
        // a node seen before: its vector grows to the union of the ranges (moving to the pool's top if it has to), the gap
        // between disjoint ranges reads ninf
        const int32_t nstart = imin(cv_vstart, query_start);
        const int32_t nend = imax(cv_vstart + cv_vlen, query_start + cn);
        if (nstart != cv_vstart || nend != cv_vstart + cv_vlen) {
            if (!conv_cover(w, E.conv, (uint32_t)cv_rec, crec, nstart, nend - nstart)) return FR_ERROR;
            cv_vec = conv_vec(E.conv, crec);
        }
        if (cv_mode == CV_BELOW) fill_range(cv_vec, query_start + cn, cv_vstart, NINF);
        else if (cv_mode == CV_ABOVE) fill_range(cv_vec, cv_vstart + cv_vlen, query_start, NINF);
    }
    if (cv_mode != CV_NONE) {
        const bool all = cv_mode != CV_MERGE;              // a new or disjoint range takes the column as it is
        FOR_LANES(l) {
            for (int s = 0; s < 4; ++s) {
                const int32_t a = org + 4 * l + s, j = a - begin;
                if (j >= skip && j < size && (all || ((mdo[l] >> s) & 1u))) gst(cv_vec + start + a - 1, mv[s][l]);
            }
        }
    }
    CH_T(7)
    if (w.status != ST_OK) return FR_ERROR;
    if (converged == NINF) return FR_END;
    if (chain_on) {
        for (int s = 0; s < 4; ++s) { pS[s] = cS[s]; pF[s] = cF[s]; }
        x.f_idx = my_idx; x.f_node = next; x.f_offset = next_offset; x.f_trim = begin; x.f_size = size; x.f_max_pos = max_pos;
        x.f_max_val = max_val; x.f_org = org;
        if (pf_n >= 0) { x.n_valid = 1; x.n_for = my_idx; x.n_count = pf_n; }
        wave_sync();
#ifndef MGX_CHAIN_PROBE
        w.xcyc[2] += xclock() - tx1;
#endif
        return FR_CONT;
    }
    {
        uint64_t key = queue_key(converged, -iabs(max_pos - diag_i), (uint32_t)my_idx);
        if (x.nn && converged == key_score(qget(w.next_nodes, 0))) {
            qset(w.next_nodes, x.nn++, key);
            wave_sync();
        } else {
            frontier_push(w, key);
        }
    }
#ifndef MGX_CHAIN_PROBE
    w.xcyc[2] += xclock() - tx1;
#endif
    return FR_END;
}

// The extension is cut into three pieces so that its steps can be driven from outside (the flat group loop of round 3, in
// which every 8-lane group advances its own read) as well as by the loop of extend():
//   extend_begin   set_seed + the root column; false = capacity (status set)
//   extend_step    one pop / chain step / general step; XS_MORE, XS_DONE (results in Wave::er) or XS_ERROR
// The chain window and the seed-replay run are registers of whoever drives the steps (ChainRegs).
struct ChainRegs {
    LV<int32_t> pS[4], pF[4];            // the chain window: S and F of the chain's current column, 4 cells per lane
    SeedRun run;
    int mode;
};
enum { XS_MORE = 0, XS_DONE = 1, XS_ERROR = 2 };

MGX_DEV void chain_regs_reset(ChainRegs &R) {
    FOR_LANES(l) { R.run.nodes[l] = 0; R.run.chars[l] = 0; }
    FOR_LANES(l) { for (int s = 0; s < 4; ++s) { R.pS[s][l] = NINF; R.pF[s][l] = NINF; } }
    R.mode = XM_POP;
}

MGX_DEV bool extend_begin(Wave &w, const int es, const SeedRef &seed, bool force_fixed_seed) {
    MGX_ASSUME_LDS(&w);
    ExtenderState &E = w.ext[es];
    ExtendResult *res = &w.er;
    const AlignParams &P = MGX_PARAMS_OF(w);
    const DevConfig &cfg = P.cfg;
    const DevLimits &lim = P.lim;
    XState &x = w.x;
    ++w.n_extensions;
    // table.clear(); prev_starts.clear()
    for (uint32_t base = 0; base < (lim.max_columns + 31) / 32; base += WAVE) {
        FOR_LANES(l) { uint32_t j = base + l; if (j < (lim.max_columns + 31) / 32) w.prev_starts[j] = 0; }
    }
    const int32_t xdrop = uni(cfg.xdrop);                     // added_xdrop == 0
    x.xdrop_cutoff = imax(-xdrop, NINF + 1);
    x.start = uni(seed.clipping);
    x.window_size = uni(w.L) - x.start;                       // trim_query_suffix == 0
    x.qlen = uni(w.L);
    x.psum_lin = uni(E.psum_lin);
    x.partial_sum_offset = x.psum_lin ? (x.qlen - (x.start + x.window_size)) * x.psum_lin : uni(E.psum[x.start + x.window_size]);
    x.seed_off = uni(seed.offset);
    x.seed_seq_len = uni(seed.seq_len);
    x.force_fixed = force_fixed_seed ? 1 : 0;
    x.seed_nodes = seed.nodes;
    x.seed_seq = seed.seq;
    x.seq_lds = (w.q_lds && seed.seq >= w.q[seed.orientation] && seed.seq < w.q[seed.orientation] + w.L) ? 1 : 0;
    x.rc = E.rc_view ? 1 : 0;
    x.n_valid = 0; x.n_for = -1; x.n_count = 0;
    x.seed_n_nodes = uni(seed.n_nodes); x.sn_base = -0x40000000; x.sc_base = -0x40000000;
    x.rel_cutoff = cfg.rel_score_cutoff; x.max_nodes_per_char = cfg.max_nodes_per_seq_char; x.max_ram = cfg.max_ram_per_alignment;
    x.go = cfg.gap_open; x.ge = cfg.gap_ext; x.xdrop = cfg.xdrop; x.k = (int32_t)P.g.k; x.Lq = (int32_t)lim.Lmax;
    x.max_columns = (int32_t)lim.max_columns; x.cell_words = lim.cell_words;
    // aliases point into the one column table: usable when nothing runs another extension between this one and the last
    // reader of its convergence table, i.e. with one alignment per seed (aln_both checks the later seeds before the backward pass)
    x.alias_ok = n_alt_of(w) == 1 && !P.no_alias ? 1u : 0u;
#if MGX_WITH_LABELS
    if (P.labeled) {
        // LabeledExtender::set_seed (aligner_labeled.cpp:139-174, no coordinates): the seed's first node counts as flushed,
        // the seed's labels are what backtracking has to account for; the per-extension sets of the last extension are dead
        x.alias_ok = 0;
        w.lab_lo = 1;
        w.last_flushed = 1;
        w.remaining_lab = seed.lab;
        w.col_lab[0] = seed.lab;
        w.bt_isect = w.bt_diff = 0;
    }
#endif
    E.conv.start = (uint32_t)x.start;
    x.cell_top = 0;
    x.tsize = 0;
    x.table_size_bytes = 0;
    x.f_n_out = 0;
    w.st[0].col = -1; w.st[1].col = -1;
    w.blk_cache_idx = 0xFFFFFFFFu;
    // root column (:455-470)
    {
        ColMeta r;
        r.node = seed.nodes[0]; r.parent = -1; r.cw = 0; r.org = 0; r.offset = x.seed_off - 1; r.max_pos = 0; r.trim = 0;
        r.score = 0; r.cells = 0; r.size = 1;
        Staging &s0 = w.st[0];
        const int32_t cap = w.st_cap;
        for (int32_t base = 0; base < 8; base += WAVE) {
            FOR_LANES(l) { int32_t j = base + l; if (j < 8) { tset(s0.S, cap, j, NINF); tset(w.stE, cap, j, NINF); tset(s0.F, cap, j, NINF); } }
        }
        wave_sync();
        int32_t sroot = (cfg.left_end_bonus && !seed.clipping) ? cfg.left_end_bonus : 0;
        FOR_LANES(l) { if (l == 0) tset(s0.S, cap, 0, sroot); }
        wave_sync();
        int32_t max_size = x.window_size + 1;
        int32_t pushes = 0;
        if (1 < max_size) {
            int32_t ins_score = imax(sroot + cfg.gap_open, NINF + cfg.gap_ext);
            if (ins_score >= x.xdrop_cutoff) {
                int32_t n_push = 1, room = max_size - 2;
                if (cfg.gap_ext == 0) n_push += room;
                else { int32_t v = ins_score; while (n_push - 1 < room && v + cfg.gap_ext >= x.xdrop_cutoff) { v += cfg.gap_ext; ++n_push; } }
                for (int32_t base = 0; base < n_push + 5; base += WAVE) {
                    FOR_LANES(l) {
                        int32_t t = base + l;
                        if (t < n_push) {
                            int32_t v = ins_score + t * cfg.gap_ext;
                            tset(s0.S, cap, 1 + t, v); tset(w.stE, cap, 1 + t, v); tset(s0.F, cap, 1 + t, NINF);
                        } else if (t < n_push + 5) {
                            tset(s0.S, cap, 1 + t, NINF); tset(w.stE, cap, 1 + t, NINF); tset(s0.F, cap, 1 + t, NINF);
                        }
                    }
                }
                pushes = n_push;
            }
        }
        r.size = 1 + pushes;
        if ((uint64_t)rec_words((uint32_t)r.size + 8) > lim.cell_words) { w.status = ST_CAPACITY; res->table_size = 0; return false; }
        const uint32_t root_cap3 = 3 * ref_capacity(1, (uint32_t)pushes);
        wave_sync();
        const int32_t root_wc = flush_column(w, s0, 0, r.size, cfg.gap_ext, nullptr, 0, 0, 0, 0, 0, E.q);
        r.cw = (uint32_t)root_wc << 8;
        r.base = 0; r.self = 0;
        s0.col = 0;
        x.cell_top = rec_words((uint32_t)root_wc);
        if (E.table_cap < 1) E.table_cap = 1;                 // emplace_back on an empty vector
        col_store(w, 0, r);
        x.tsize = 1;
        x.table_size_bytes = (uint64_t)136 * E.table_cap + (uint64_t)root_cap3 * 4;
    }
    x.min_cell_score = 0;
    x.best_score = 0;
    x.qn = 0; x.nn = 0; x.n_tips = 0;
    frontier_push(w, queue_key(0, 0, 0));
    return true;
}

MGX_DEV int extend_step(Wave &w, const int es, ChainRegs &R) {
    MGX_ASSUME_LDS(&w);
    ExtenderState &E = w.ext[es];
    ExtendResult *res = &w.er;
    const AlignParams &P = MGX_PARAMS_OF(w);
    XState &x = w.x;
    // the chain format keeps S as 16-bit offsets from the column maximum: cells live within x-drop (+ one match score) of
    // it, so any x-drop up to 30000 fits; wider (the unit tests' "no x-drop") takes the general path
    const bool use_fast = !P.no_fast && P.cfg.xdrop <= 30000;
    int &mode = R.mode;
    LV<int32_t> *pS = R.pS, *pF = R.pF;
    SeedRun &run = R.run;
    {
        int32_t gi = -1;                 // column for the general step of this iteration
        bool children_ready = false;
        if (mode == XM_POP) {
            if (x.nn == 0) {
                if (x.qn == 0) {
                    wave_sync();
                    res->n_tips = x.n_tips;
                    res->min_cell_score = x.min_cell_score;
                    res->table_size = x.tsize;
                    return XS_DONE;
                }
                uint64_t tx0 = xclock();
                // pop every entry that shares the top score, in descending tuple order (:491-500)
                int32_t qn = x.qn, nn = 0;
                const int32_t top_score = key_score(qget(w.queue, qn - 1));
                while (qn && key_score(qget(w.queue, qn - 1)) == top_score) {
                    qset(w.next_nodes, nn++, qget(w.queue, qn - 1));
                    --qn;
                }
                x.qn = qn; x.nn = nn;
                x.q_top = qn ? key_score(qget(w.queue, qn - 1)) : INT32_MIN;
                wave_sync();
#ifndef MGX_CHAIN_PROBE
                w.xcyc[0] += xclock() - tx0;
#endif
            }
            const int32_t i = (int32_t)uni(key_idx(qget(w.next_nodes, x.nn - 1)));
            --x.nn;
            const ColMeta col = uni_col(col_load(w, i));
            if (use_fast && x.nn == 0 && fast_fits(col)) {
                fast_load(w, col, i, pS, pF);
                mode = XM_FAST;
            } else {
                gi = i;
            }
        }
        if (mode == XM_FAST) {
            const int r = chain_step(w, E, pS, pF, run);
            if (r == FR_CONT) return XS_MORE;
            mode = XM_POP;
            if (r == FR_END) return XS_MORE;
            if (r == FR_STOP) { x.qn = 0; x.nn = 0; return XS_MORE; }
            if (r == FR_ERROR) { res->table_size = 0; return XS_ERROR; }
            // FR_FALLBACK: the parent goes through the general code (children already enumerated)
            fast_spill(w, pS, pF);
            gi = x.f_idx;
            children_ready = true;
        }
        if (gi >= 0) {
            const uint64_t tg = xclock();
            const int bad = general_step(w, E, gi, children_ready);
#ifndef MGX_CHAIN_PROBE
            w.xcyc[1] += xclock() - tg;
#endif
            if (bad) { res->table_size = 0; return XS_ERROR; }
        }
    }
    return XS_MORE;
}

// One function for the whole loop: a call boundary makes the callee wait for every store it issued (s_waitcnt before
// s_setpc), which would drain each column's record stores at the end of each step.
// (es: which of the wave's two extenders — taken as an index so that the extender state is addressed off the control
// block, i.e. provably in LDS, rather than through a second generic pointer)
MGX_NI_G3 void extend(Wave &w, const int es, const SeedRef &seed, bool force_fixed_seed) {
    if (!extend_begin(w, es, seed, force_fixed_seed)) return;
    ChainRegs R;
    chain_regs_reset(R);
    while (extend_step(w, es, R) == XS_MORE) {}
}

// ------------------------------------------------------------------------------------------------
// backtracking (DefaultColumnExtender::backtrack + construct_alignment, :774-1034), target_node == npos
// Writes at most one alignment into `out`; returns whether one was produced.
// ------------------------------------------------------------------------------------------------
MGX_DEV bool prev_start_test_and_set(Wave &w, int32_t j) {
    uint32_t word = gld(w.prev_starts + (j >> 5));
    bool was = (word >> (j & 31)) & 1;
    if (!was) { gst(w.prev_starts + (j >> 5), word | (1u << (j & 31))); }
    return !was;       // true when newly inserted (emplace(...).second)
}

MGX_DEV void cigar_append(uint32_t *ops, int32_t *n, uint32_t op, uint32_t num, int32_t cap, int32_t *status) {
    if (!num) return;
    if (*n == 0 || (ops[*n - 1] & 7) != op) {
        if (*n >= cap) { *status = ST_CAPACITY; return; }
        ops[(*n)++] = (num << 3) | op;
    } else {
        ops[*n - 1] += num << 3;
    }
}

MGX_DEV void seed_as_alignment(Wave &w, const SeedRef &seed, DevAln &out);

MGX_NI_G4 void copy_aln(DevAln &dst, const DevAln &src);

// Backtracking in resumable pieces (round 3): bt_begin() collects the start cells, bt_step() advances the pop / walk /
// construct cycle by a bounded number of walk steps and says when it is over.  backtrack() drives them to the end (the
// per-read program of the legacy path); the flat group loop calls bt_step() once per iteration, so that a group whose
// extension ended walks its trace while its wave-mates are still extending.  State between steps: BtState (LDS, overlaid
// with the extension's loop state, which is dead by then; the extension's results are in Wave::er).
enum { BT_POP = 0, BT_WALK = 1, BT_FINISH = 2, BT_OVER = 3 };

// seed_aln: the Alignment the seed was made from (backward pass) or nullptr for Seed-derived seeds
MGX_DEV void bt_begin(Wave &w, const int es, const SeedRef &seed, int32_t min_path_score, int n_max) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    const ExtendResult er = w.er;
    BtState &b = w.bt;
    const uint8_t *bq = w.ext[es].q;
    const bool bq_lds = w.q_lds != 0;
    auto op_at = [&](uint8_t c, int32_t abs_pos) -> uint8_t {      // profile_op_at
        if (abs_pos < 1 || abs_pos > w.L) return OP_CLIPPED;
        const uint32_t code = encode_char(c);
        const uint8_t row = code != 5 ? decode_code(code) : 0;
        return char_to_op(row, bq_lds ? lds_u8(bq + abs_pos - 1) : gld(bq + abs_pos - 1));
    };
    const DevConfig &cfg = P.cfg;
    const int32_t k = (int32_t)P.g.k;
    const int32_t seed_clipping = seed.clipping;
    const int32_t seed_offset = seed.offset - 1;
    const int32_t window_size = w.L - seed.clipping;
    const int32_t last_pos = window_size;
    const int32_t seed_dist = imax(k, seed.seq_len) - 1;
    const int32_t min_start_score = min_path_score;
    const int32_t right_end_bonus = cfg.right_end_bonus;
    const int32_t tsize = er.table_size;
#if MGX_WITH_LABELS
    if (P.labeled) { lab_flush(w, tsize); if (w.status != ST_OK) { w.bt.stage = BT_OVER; w.bt.produced = 0; return; } }     // LabeledExtender::backtrack (aligner_labeled.hpp:32-43)
#endif
#ifdef MGX_BT_PROBE
    uint64_t tbt = xclock();
#define BT_T(slot) { const uint64_t t_ = xclock(); w.xcyc[slot] += t_ - tbt; tbt = t_; }
#else
#define BT_T(slot)
#endif
    // candidate start cells (:815-867), one table column per lane; the order of `indices` is irrelevant
    // because the heap pops by the full (unique) tuple
    int32_t n_idx = 0;
    // every lane also keeps the lexicographic maximum (score, -off_diag, -i, pos) of the candidates it wrote and where it
    // wrote it: the first pop of the heap — usually the only one — then needs no pass over the list in memory
    LV<BtIndex> lbest;
    LV<int32_t> lbest_at;
    FOR_LANES(l) { lbest[l] = BtIndex{ INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN }; lbest_at[l] = -1; }
    auto bt_greater = [](const BtIndex &a, const BtIndex &c) {
        if (a.score != c.score) return a.score > c.score;
        if (a.neg_off_diag != c.neg_off_diag) return a.neg_off_diag > c.neg_off_diag;
        if (a.neg_i != c.neg_i) return a.neg_i > c.neg_i;
        return a.pos > c.pos;
    };
    for (int32_t base = 1; base < tsize; base += WAVE) {
        LV<int32_t> cnt;
        LV<BtIndex> c0, c1;
        FOR_LANES(l) {
            int32_t i = base + l;
            int32_t n = 0;
            BtIndex b0 = { 0, 0, 0, 0 }, b1 = { 0, 0, 0, 0 };
            if (i < tsize) {
                // one line per column: its slot's metadata and the flags of the start cell (S of a chain column is its
                // maximum, i.e. `base`, at max_pos); the parent is never touched
                const ColMeta col = col_load(w, i);
                if (col.offset >= seed_dist) {
                    bool is_tip = false;
                    for (int32_t t = 0; t < er.n_tips; ++t) is_tip |= w.tips[t] == (uint32_t)i;
                    for (int pass = 0; pass < 2; ++pass) {
                        int32_t start_pos;
                        if (pass == 0) start_pos = col.max_pos;
                        else {
                            if (!(col.size + col.trim == window_size + 1 && col.max_pos != last_pos)) break;
                            start_pos = last_pos;
                        }
                        // start_pos < par.trim + 1, or S_parent[start_pos - 1] == ninf: not a start cell (:846-849)
                        const uint32_t fl = cell_flags(w, col, start_pos);
                        if (!(fl & CF_REAL) || !(fl & CF_SP_REAL)) continue;
                        const int32_t sv = (pass == 0 && col_chain(col)) ? col.base : cell_S(w, col, start_pos);
                        int32_t end_bonus = start_pos == last_pos ? right_end_bonus : 0;
                        if (sv + end_bonus >= min_start_score) {
                            bool is_match = (fl & CF_MATCH) && op_at(col_char(col), seed_clipping + start_pos) == OP_MATCH;
                            if (is_match || start_pos == last_pos || is_tip) {
                                BtIndex bx;
                                bx.score = sv + end_bonus; bx.neg_off_diag = -iabs(start_pos - col.offset + seed_offset);
                                bx.neg_i = -i; bx.pos = start_pos;
                                if (n == 0) b0 = bx; else b1 = bx;
                                ++n;
                            }
                        }
                    }
                }
            }
            cnt[l] = n; c0[l] = b0; c1[l] = b1;
        }
        LV<int32_t> off = wave_prefix_sum_excl(cnt);
        FOR_LANES(l) {
            if (cnt[l] > 0) {
                w.indices[n_idx + off[l]] = c0[l];
                if (lbest_at[l] < 0 || bt_greater(c0[l], lbest[l])) { lbest[l] = c0[l]; lbest_at[l] = n_idx + off[l]; }
            }
            if (cnt[l] > 1) {
                w.indices[n_idx + off[l] + 1] = c1[l];
                if (bt_greater(c1[l], lbest[l])) { lbest[l] = c1[l]; lbest_at[l] = n_idx + off[l] + 1; }
            }
        }
        n_idx += wave_sum(cnt);
    }
    // wave-level maximum of the lanes' maxima (registers only)
    int32_t first_bi = -1;
    if (n_idx > 0) {
        LV<int32_t> v;
        FOR_LANES(l) { v[l] = lbest_at[l] >= 0 ? lbest[l].score : INT32_MIN; }
        const int32_t m_score = wave_max(v);
        FOR_LANES(l) { v[l] = (lbest_at[l] >= 0 && lbest[l].score == m_score) ? lbest[l].neg_off_diag : INT32_MIN; }
        const int32_t m_off = wave_max(v);
        FOR_LANES(l) { v[l] = (lbest_at[l] >= 0 && lbest[l].score == m_score && lbest[l].neg_off_diag == m_off) ? lbest[l].neg_i : INT32_MIN; }
        const int32_t m_i = wave_max(v);
        FOR_LANES(l) {
            v[l] = (lbest_at[l] >= 0 && lbest[l].score == m_score && lbest[l].neg_off_diag == m_off && lbest[l].neg_i == m_i) ? lbest[l].pos : INT32_MIN;
        }
        const int32_t m_pos = wave_max(v);
        LV<bool> hit;
        FOR_LANES(l) {
            hit[l] = lbest_at[l] >= 0 && lbest[l].score == m_score && lbest[l].neg_off_diag == m_off && lbest[l].neg_i == m_i && lbest[l].pos == m_pos;
        }
        first_bi = wave_bcast(lbest_at, ctz64(wave_ballot(hit)));
    }
    wave_sync();
    BT_T(3)
    b.es = es; b.n_max = n_max; b.min_path_score = min_path_score;
    b.produced = 0; b.best_score = INT32_MIN; b.remaining = n_idx; b.first_bi = first_bi;
    b.stage = BT_POP;
}

// Advances the backtracking by at most `max_walk` steps of a trace walk; true when it is over (b.produced alignments in
// outs[0 ..); up to n_max = num_alternative_paths, :1005 terminate_backtrack_start).
MGX_DEV bool bt_step(Wave &w, const SeedRef &seed, const DevAln *seed_aln, DevAln *outs, int32_t max_walk) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    const ExtendResult er = w.er;
    BtState &b = w.bt;
    const int es = b.es;
    // the extender's query: read with LDS or global instructions, never generic ones (a FLAT load waits for every store
    // in flight, and the walk below stores three values per step)
    const uint8_t *bq = w.ext[es].q;
    const bool bq_lds = w.q_lds != 0;
    auto op_at = [&](uint8_t c, int32_t abs_pos) -> uint8_t {      // profile_op_at
        if (abs_pos < 1 || abs_pos > w.L) return OP_CLIPPED;
        const uint32_t code = encode_char(c);
        const uint8_t row = code != 5 ? decode_code(code) : 0;
        return char_to_op(row, bq_lds ? lds_u8(bq + abs_pos - 1) : gld(bq + abs_pos - 1));
    };
    const DevConfig &cfg = P.cfg;
    const int32_t k = (int32_t)P.g.k;
    const int32_t seed_clipping = seed.clipping;
    const int32_t k_minus_1 = k - 1;
    const int32_t min_start_score = b.min_path_score;
    const int32_t min_trace_length = k - seed.offset;
    const int32_t cap = (int32_t)P.lim.max_path;
#ifdef MGX_BT_PROBE
    uint64_t tbt = xclock();
#endif
    if (b.stage == BT_POP) {
#if MGX_WITH_LABELS
        // LabeledExtender::terminate_backtrack_start (aligner_labeled.hpp:49-52): until every label of the seed is accounted for
        const bool lab_on = P.labeled != 0;
        if (lab_on ? !(b.remaining > 0 && w.remaining_lab) : !(b.remaining > 0 && b.produced < b.n_max)) {
#else
        if (!(b.remaining > 0 && b.produced < b.n_max)) {        // terminate_backtrack_start: extensions.size() >= num_alternative_paths
#endif
            b.stage = BT_FINISH;
        } else {
            // pop the lexicographic maximum (score, -off_diag, -i, pos) (:873-879): four lane-parallel passes
            const int32_t remaining = b.remaining;
            int32_t bi = 0;
            if (b.first_bi >= 0) {
                bi = b.first_bi;
                b.first_bi = -1;
            } else {
                int32_t m_score = INT32_MIN, m_off = INT32_MIN, m_i = INT32_MIN, m_pos = INT32_MIN;
                for (int pass = 0; pass < 4; ++pass) {
                    int32_t best = INT32_MIN;
                    for (int32_t base = 0; base < remaining; base += WAVE) {
                        LV<int32_t> v;
                        FOR_LANES(l) {
                            int32_t x = base + l;
                            int32_t val = INT32_MIN;
                            if (x < remaining) {
                                BtIndex a = w.indices[x];
                                bool ok = (pass < 1 || a.score == m_score) && (pass < 2 || a.neg_off_diag == m_off) && (pass < 3 || a.neg_i == m_i);
                                if (ok) val = pass == 0 ? a.score : pass == 1 ? a.neg_off_diag : pass == 2 ? a.neg_i : a.pos;
                            }
                            v[l] = val;
                        }
                        best = imax(best, wave_max(v));
                    }
                    if (pass == 0) m_score = best; else if (pass == 1) m_off = best; else if (pass == 2) m_i = best; else m_pos = best;
                }
                for (int32_t base = 0; base < remaining; base += WAVE) {
                    LV<bool> hit;
                    FOR_LANES(l) {
                        int32_t x = base + l;
                        bool h = false;
                        if (x < remaining) {
                            BtIndex a = w.indices[x];
                            h = a.score == m_score && a.neg_off_diag == m_off && a.neg_i == m_i && a.pos == m_pos;
                        }
                        hit[l] = h;
                    }
                    uint64_t mk = wave_ballot(hit);
                    if (mk) { bi = base + ctz64(mk); break; }
                }
            }
            BtIndex cur = w.indices[bi];
            w.indices[bi] = w.indices[remaining - 1];
            b.remaining = remaining - 1;
            const int32_t j = -cur.neg_i;
            if (!prev_start_test_and_set(w, j)) return false;       // skip_backtrack_start (next pop on the next step)
#if MGX_WITH_LABELS
            if (lab_on) {
                // LabeledExtender::skip_backtrack_start (aligner_labeled.cpp:304-326): the labels this start cell can still
                // account for; none: skip it
                lab_isect_diff(w, w.remaining_lab, w.col_lab[j], &w.bt_isect, &w.bt_diff);
                if (w.status != ST_OK) return true;
                if (!w.bt_isect) return false;
            }
#endif
            if (cur.score - er.min_cell_score < b.best_score) { b.stage = BT_FINISH; }
            else {
                b.j = j; b.score = cur.score; b.pos = cur.pos; b.end_pos = cur.pos;
                b.n_ops = 0; b.n_path = 0; b.n_seq = 0; b.n_trace = 0; b.cur_run = 0; b.dummy_counter = 0;
                b.align_offset = seed.offset; b.extra_score = 0; b.last_path_node = 0;
                b.stage = BT_WALK;
            }
        }
    }
    if (b.stage == BT_WALK) {
        int32_t score = b.score;
        int32_t n_ops = b.n_ops, n_path = b.n_path, n_seq = b.n_seq, n_trace = b.n_trace;
        uint32_t cur_run = b.cur_run;                  // rev_ops[n_ops - 1], kept in a register: appending never reads memory
        auto push_op = [&](uint32_t op, uint32_t num) {
            if (!num) return;
            if (n_ops == 0 || (cur_run & 7) != op) {
                if (n_ops >= cap) { w.status = ST_CAPACITY; return; }
                cur_run = (num << 3) | op;
                gst(w.rev_ops + n_ops++, cur_run);
            } else {
                cur_run += num << 3;
                gst(w.rev_ops + n_ops - 1, cur_run);
            }
        };
        int32_t dummy_counter = b.dummy_counter;
        int32_t pos = b.pos;
        const int32_t end_pos = b.end_pos;
        int32_t align_offset = b.align_offset;
        int32_t extra_score = b.extra_score;
        uint32_t last_path_node = b.last_path_node;
        int32_t j = b.j;
        auto append_node = [&](uint32_t node, uint8_t c, int32_t offset, uint32_t op) {
            if (n_seq >= cap) { w.status = ST_CAPACITY; return; }
            gst(w.rev_seq + n_seq++, c);
            push_op(op, 1);
            if (offset >= k_minus_1) {
                if (n_path >= cap) { w.status = ST_CAPACITY; return; }
                gst(w.rev_nodes + n_path++, node);
                last_path_node = node;
                if (!node) {
                    ++dummy_counter;
                } else if (dummy_counter) {
                    push_op(OP_NODE_INSERTION, (uint32_t)dummy_counter);
                    extra_score -= cfg.gap_open + (dummy_counter - 1) * cfg.gap_ext;
                    dummy_counter = 0;
                }
            }
        };
        // The column chain is walked parent by parent; each step reads ONE thing: the flag byte of the current cell (all
        // of backtrack's comparisons were evaluated when the column was computed).  The parent's metadata is fetched one
        // step ahead.
        // Walking parent pointers is pointer chasing: one DRAM round trip per column.  Columns of an extension are mostly
        // numbered along the path (a chain column's parent is the column before it), so the slots below the one the walk
        // is about to read are fetched WAVE at a time, one per lane; the walk's own loads then hit the L2.  A guess only:
        // a parent elsewhere is fetched on demand as before.  (Serving the walk from such a batch held in registers was
        // measured too: no faster — the walk is bound by its chain of LDS round trips, not by these loads.)
        int32_t pf_lo = INT32_MAX;
        LV<uint32_t> pf_val;
        FOR_LANES(l) { pf_val[l] = 0; }
        // prev_starts bits set along the walk are collected per 32-column word and written when the walk leaves the word
        // (nothing reads the set before the next pop)
        int32_t ps_idx = -1;
        uint32_t ps_bits = 0;
        auto ps_flush = [&]() {
            if (ps_idx >= 0 && ps_bits) gst(w.prev_starts + ps_idx, gld(w.prev_starts + ps_idx) | ps_bits);
            ps_bits = 0;
        };
        // (col, par) = (column j, its parent) at the top of every step
        ColMeta col = col_load(w, j), par = col;
        if (j) par = col_load(w, col.parent);
        bool walk_over = false;
        int32_t steps = 0;
        for (;;) {
            if (!j) { walk_over = true; break; }
            if (steps++ >= max_walk) break;
            // A run of diagonal steps at once.  The typical trace is matches / mismatches along consecutive chain columns
            // (column j - 1 is the parent of column j, one query position back per step), and walking it one dependent
            // round trip per step is what the trace loop spends its time on.  Lane l looks at the step l ahead — column
            // j - l at position pos - l — on its own: slot metadata and the cell's flag byte, two loads in flight per lane
            // instead of one per step; the leading lanes whose cell takes the match branch below and whose parent is the
            // next lane's column are applied together (same appends, same marks, in the same order).  Anything else — an
            // insertion or deletion, a column in another format, a dummy node, the end of the trace — stops the run and is
            // left to the step-by-step code.
            if ((n_ops == 0 || (cur_run & 7) != OP_DELETION) && dummy_counter == 0 && !MGX_PARAMS_OF(w).no_bt_runs) {
                LV<bool> okv, linkv, mpv, mmv;
                LV<int32_t> parv, offv, scv;
                LV<uint32_t> nodev, chv;
                FOR_LANES(l) {
                    const int32_t cl = j - l, pl = pos - l;
                    bool ok = cl >= 1 && pl >= 1;
                    ColMeta m = col;
                    if (ok && l) m = col_load(w, cl);
                    ok = ok && col_compact(m) && m.node != 0;
                    uint32_t fl = 0;
                    if (ok) fl = cell_flags(w, m, pl);
                    ok = ok && (fl & CF_REAL) && (fl & CF_MATCH) && !(fl & CF_S_IS_E);
                    okv[l] = ok;
                    linkv[l] = ok && m.parent == cl - 1;
                    parv[l] = m.parent; offv[l] = m.offset; scv[l] = ok ? m.score : 0;
                    nodev[l] = m.node; chv[l] = col_char(m);
                    mpv[l] = ok && pl == m.max_pos;
                    mmv[l] = ok && op_at(col_char(m), seed_clipping + pl) != OP_MATCH;
                }
                const uint64_t okm = wave_ballot(okv), lkm = wave_ballot(linkv);
                int32_t r = 0;
                while (r < WAVE && ((okm >> r) & 1) && (r == 0 || ((lkm >> (r - 1)) & 1))) ++r;
                if (r >= 2 && n_seq + r <= cap && n_path + r <= cap && n_ops + r <= cap) {
                    const uint64_t rmask = r >= 64 ? ~0ull : ((1ull << r) - 1);
                    // prev_starts marks, step by step as below (nothing reads them before the next pop)
                    const uint64_t mpm = wave_ballot(mpv) & rmask;
                    for (int32_t l = 0; l < r; ++l) {
                        if (!((mpm >> l) & 1)) continue;
                        const int32_t cj = j - l;
                        if ((cj >> 5) != ps_idx) { ps_flush(); ps_idx = cj >> 5; }
                        ps_bits |= 1u << (cj & 31);
                    }
                    // append_node x r: characters, CIGAR operators (one run if they are all matches), nodes from offset k - 1 on
                    FOR_LANES(l) { if (l < r) gst(w.rev_seq + n_seq + l, (uint8_t)chv[l]); }
                    n_seq += r;
                    const uint64_t mmm = wave_ballot(mmv) & rmask;
                    if (!mmm) {
                        push_op(OP_MATCH, (uint32_t)r);
                    } else {
                        for (int32_t l = 0; l < r; ++l) push_op(((mmm >> l) & 1) ? OP_MISMATCH : OP_MATCH, 1);
                    }
                    LV<bool> pnv;
                    FOR_LANES(l) { pnv[l] = l < r && offv[l] >= k_minus_1; }
                    const int32_t np = popc64(wave_ballot(pnv));          // a prefix of the run: offsets fall by one per step
                    FOR_LANES(l) { if (l < np) gst(w.rev_nodes + n_path + l, nodev[l]); }
                    if (np) last_path_node = wave_bcast(nodev, np - 1);
                    n_path += np;
                    n_trace += r;
                    LV<int32_t> scr;
                    FOR_LANES(l) { scr[l] = l < r ? scv[l] : 0; }
                    extra_score += wave_sum(scr);
                    align_offset = imin(wave_bcast(offv, r - 1), k_minus_1);
                    pos -= r;
                    j = wave_bcast(parv, r - 1);
                    steps += r - 1;
                    if (j) { col = col_load(w, j); par = col_load(w, col.parent); }
                    if (w.status != ST_OK) return true;
                    continue;
                }
            }
            ColMeta gp = par;
            if (col.parent > 0) {                                      // par is not the root
                const int32_t t = par.parent;
                if (t < pf_lo) {
#if !MGX_WAVE_EMU
                    asm volatile("" : : "v"(pf_val.v));                // the previous batch has long arrived
#endif
                    FOR_LANES(l) { pf_val[l] = gld((const uint32_t *)(w.cols + imax(0, t - l))); }
                    pf_lo = t - WAVE + 1;
                }
                gp = col_load(w, t);
            }
            align_offset = imin(col.offset, k_minus_1);
            if (pos == col.max_pos) {
                if ((j >> 5) != ps_idx) { ps_flush(); ps_idx = j >> 5; }
                ps_bits |= 1u << (j & 31);
            }
            const uint32_t fl = cell_flags(w, col, pos);
            const uint32_t last_op = n_ops ? (cur_run & 7) : 99u;
            if (!(fl & CF_REAL)) {
                j = 0;
            } else if (pos && (fl & CF_S_IS_E) && (n_ops == 0 || last_op != OP_DELETION)) {
                uint32_t lop = OP_INSERTION;
                while (lop == OP_INSERTION) {
                    push_op(lop, 1);
                    lop = (cell_flags(w, col, pos) & CF_E_EXT) ? OP_INSERTION : OP_MATCH;
                    --pos;
                    if (w.status != ST_OK) return true;
                }
            } else if (pos && (fl & CF_MATCH)) {
                ++n_trace;
                extra_score += col.score;
                append_node(col.node, col_char(col), col.offset, op_at(col_char(col), seed_clipping + pos));
                --pos;
                j = col.parent;
                col = par; par = gp;
            } else if ((fl & CF_S_IS_F) && (n_ops == 0 || last_op != OP_INSERTION)) {
                uint32_t lop = OP_DELETION;
                while (lop == OP_DELETION && j) {
                    const ColMeta c2 = col_load(w, j);
                    align_offset = imin(c2.offset, k_minus_1);
                    lop = (cell_flags(w, c2, pos) & CF_F_EXT) ? OP_DELETION : OP_MATCH;
                    ++n_trace;
                    extra_score += c2.score;
                    append_node(c2.node, col_char(c2), c2.offset, OP_DELETION);
                    j = c2.parent;
                    if (w.status != ST_OK) return true;
                }
                if (j) { col = col_load(w, j); par = col_load(w, col.parent); }
            } else {
                walk_over = true;
                break;
            }
            if (w.status != ST_OK) return true;
        }
        ps_flush();
#if !MGX_WAVE_EMU
        asm volatile("" : : "v"(pf_val.v));
#endif
        if (!walk_over) {
            // out of steps for this call: park the walk
            b.j = j; b.pos = pos; b.n_ops = n_ops; b.n_path = n_path; b.n_seq = n_seq; b.n_trace = n_trace; b.cur_run = cur_run;
            b.dummy_counter = dummy_counter; b.align_offset = align_offset; b.extra_score = extra_score; b.last_path_node = last_path_node;
            wave_sync();
            return false;
        }
#ifdef MGX_BT_PROBE
        BT_T(4)
#endif
        b.stage = BT_POP;
        if (n_trace >= min_trace_length && n_path && last_path_node) {
            const ColMeta cj = col_load(w, j);
            int32_t cur_cell_score = cell_S(w, cj, pos);
            b.best_score = imax(b.best_score, score - cur_cell_score);
            if (score - er.min_cell_score < b.best_score) {
                b.stage = BT_FINISH;
            } else if (score >= min_start_score && (!pos || cur_cell_score == 0)
                    && (pos || cur_cell_score == gld(w.cells + col_load(w, 0).cells))
                    && (cfg.allow_left_trim || !j)) {
                // construct_alignment (:774-798): clipping = pos, window = [pos, end_pos)
#if MGX_WITH_LABELS
                if (b.produced >= b.n_max) { w.status = ST_CAPACITY; return true; }      // (labels: more alignments than buffers)
#endif
                DevAln &out = outs[b.produced];
                int32_t nc = 0;
                uint32_t clip_total = (uint32_t)(seed_clipping + pos);      // cigar clip + extend_query_begin
                if (clip_total) out.cigar[nc++] = (clip_total << 3) | OP_CLIPPED;
                // reversed ops; the clipping appended last becomes the first run
                for (int32_t x = n_ops - 1; x >= 0; --x) {
                    uint32_t op = w.rev_ops[x];
                    if (nc && (out.cigar[nc - 1] & 7) == (op & 7)) out.cigar[nc - 1] += (op >> 3) << 3;
                    else { if (nc >= cap) { w.status = ST_CAPACITY; return true; } out.cigar[nc++] = op; }
                }
                uint32_t end_clip = (uint32_t)(w.L - (seed_clipping + end_pos));   // extend_query_end
                if (end_clip) {
                    if (nc && (out.cigar[nc - 1] & 7) == OP_CLIPPED) out.cigar[nc - 1] += end_clip << 3;
                    else { if (nc >= cap) { w.status = ST_CAPACITY; return true; } out.cigar[nc++] = (end_clip << 3) | OP_CLIPPED; }
                }
                wave_sync();
                for (int32_t base = 0; base < imax(n_path, n_seq); base += WAVE) {
                    FOR_LANES(l) {
                        int32_t x = base + l;
                        if (x < n_path) out.nodes[x] = w.rev_nodes[n_path - 1 - x];
                        if (x < n_seq) out.seq[x] = w.rev_seq[n_seq - 1 - x];
                    }
                }
                out.n_cigar = nc; out.n_nodes = n_path; out.seq_len = n_seq;
                out.score = score; out.offset = align_offset;
                out.qbegin = seed_clipping + pos; out.qlen = end_pos - pos;
                out.orientation = seed.orientation; out.extra_score = extra_score;
#if MGX_WITH_LABELS
                if (P.labeled) {
                    // LabeledExtender::call_alignments (aligner_labeled.cpp:328-448, no coordinates)
                    out.lab = lab_persist(w, w.bt_isect);
                    w.remaining_lab = w.bt_diff;
                    w.bt_isect = 0;
                    if (w.status != ST_OK) return true;
                } else out.lab = 0;
#endif
                wave_sync();
                ++b.produced;
            }
        }
        if (b.stage == BT_POP) return false;           // next pop on the next step
    }
    if (b.stage == BT_FINISH) {
#ifdef MGX_BT_PROBE
        BT_T(5)
#endif
        int produced = b.produced;
        if (!produced && seed.score >= b.min_path_score) {       // extensions.emplace_back(*seed_) (:1030-1031)
            if (seed_aln) copy_aln(outs[0], *seed_aln);
            else seed_as_alignment(w, seed, outs[0]);
            produced = 1;
        }
        for (int e = 0; e < produced; ++e) {
            // extension.trim_offset() (alignment.cpp:177-190)
            DevAln &out = outs[e];
            if (out.offset && out.n_nodes > 1) {
                int32_t first_dummy = out.n_nodes;      // no npos nodes can occur on this path
                for (int32_t x = 0; x < out.n_nodes; ++x) if (!out.nodes[x]) { first_dummy = x; break; }
                int32_t trim = imin(imin(out.offset, out.n_nodes - 1), first_dummy - 1);
                if (trim > 0) {
                    // erase the first `trim` nodes (wave-uniform in-place shift)
                    for (int32_t x = 0; x + trim < out.n_nodes; ++x) out.nodes[x] = out.nodes[x + trim];
                    out.n_nodes -= trim;
                    out.offset -= trim;
                }
            }
            wave_sync();
        }
        b.produced = produced;
        b.stage = BT_OVER;
    }
    return b.stage == BT_OVER;
}

// Writes up to n_max alignments into outs[0 .. ); returns how many (0 with w.status set on a capacity error).
MGX_NI_G4 int backtrack(Wave &w, const int es, const SeedRef &seed, const DevAln *seed_aln,
                      const ExtendResult &er, int32_t min_path_score, DevAln *outs, int n_max) {
    (void)er;                                           // == w.er, where extend() left it
    bt_begin(w, es, seed, min_path_score, n_max);
    while (!bt_step(w, seed, seed_aln, outs, INT32_MAX)) {}
    return w.status == ST_OK ? w.bt.produced : 0;
}

// Alignment(const Seed&, config) or a copy of an Alignment used as seed
MGX_DEV void seed_as_alignment(Wave &w, const SeedRef &seed, DevAln &out) {
    int32_t nc = 0;
    if (seed.clipping) out.cigar[nc++] = ((uint32_t)seed.clipping << 3) | OP_CLIPPED;
    // Seed-derived seeds are exact matches: cigar "cS m= eS" (alignment.hpp:162-164)
    out.cigar[nc++] = ((uint32_t)seed.qlen << 3) | OP_MATCH;
    if (seed.end_clipping) out.cigar[nc++] = ((uint32_t)seed.end_clipping << 3) | OP_CLIPPED;
    for (int32_t base = 0; base < imax(seed.n_nodes, seed.seq_len); base += WAVE) {
        FOR_LANES(l) {
            int32_t x = base + l;
            if (x < seed.n_nodes) out.nodes[x] = seed.nodes[x];
            if (x < seed.seq_len) out.seq[x] = seed.seq[x];
        }
    }
    out.n_cigar = nc; out.n_nodes = seed.n_nodes; out.seq_len = seed.seq_len;
    out.score = seed.score; out.offset = seed.offset;
    out.qbegin = seed.clipping; out.qlen = seed.qlen; out.orientation = seed.orientation; out.extra_score = 0;
#if MGX_WITH_LABELS
    out.lab = seed.lab;
#endif
    wave_sync();
}

MGX_NI_G4 void copy_aln(DevAln &dst, const DevAln &src) {
    int32_t n = imax(imax(src.n_nodes, src.n_cigar), src.seq_len);
    for (int32_t base = 0; base < n; base += WAVE) {
        FOR_LANES(l) {
            int32_t x = base + l;
            if (x < src.n_nodes) dst.nodes[x] = src.nodes[x];
            if (x < src.n_cigar) dst.cigar[x] = src.cigar[x];
            if (x < src.seq_len) dst.seq[x] = src.seq[x];
        }
    }
    dst.n_nodes = src.n_nodes; dst.n_cigar = src.n_cigar; dst.seq_len = src.seq_len;
    dst.score = src.score; dst.offset = src.offset; dst.qbegin = src.qbegin; dst.qlen = src.qlen;
    dst.orientation = src.orientation; dst.extra_score = src.extra_score;
#if MGX_WITH_LABELS
    dst.lab = src.lab;
#endif
    wave_sync();
}

// Alignment::reverse_complement for RCDBG views (alignment.cpp:547-561); false = alignment became empty
MGX_NI_G4 bool reverse_complement_aln(Wave &w, DevAln &a) {
    MGX_ASSUME_LDS(&w);
    if (a.offset) { a.n_nodes = 0; return false; }        // trim_offset() left a non-zero offset
    int32_t n = imax(imax(a.n_nodes, a.n_cigar), a.seq_len);
    for (int32_t base = 0; base < (n + 1) / 2; base += WAVE) {
        FOR_LANES(l) {
            int32_t x = base + l;
            if (x < a.n_nodes / 2) { uint32_t t = a.nodes[x]; a.nodes[x] = a.nodes[a.n_nodes - 1 - x]; a.nodes[a.n_nodes - 1 - x] = t; }
            if (x < a.n_cigar / 2) { uint32_t t = a.cigar[x]; a.cigar[x] = a.cigar[a.n_cigar - 1 - x]; a.cigar[a.n_cigar - 1 - x] = t; }
            if (x < (a.seq_len + 1) / 2) {
                uint8_t t0 = complement_char(a.seq[x]), t1 = complement_char(a.seq[a.seq_len - 1 - x]);
                a.seq[x] = t1; a.seq[a.seq_len - 1 - x] = t0;
            }
        }
    }
    wave_sync();
    a.orientation = !a.orientation;
    int32_t clip = aln_clipping(a), eclip = aln_end_clipping(a);
    a.qbegin = clip;
    a.qlen = w.L - clip - eclip;
    return true;
}

// Alignment::reverse_complement on a graph that holds the reverse complement itself (CANONICAL-mode DBGSuccinct), for
// alignments without an offset (alignment.cpp:563-565: every caller on this path has checked it): the path of the reverse
// complement is found by mapping the reversed-complemented spelling (reverse_complement_seq_path, sequence_graph.cpp:563-573
// -> BOSS::map_to_edges), here by one lane walking the byte chain of map_chain.hpp with strand = 1.
MGX_NI_G4 bool reverse_complement_aln_canonical(Wave &w, DevAln &a) {
    MGX_ASSUME_LDS(&w);
    if (a.offset) { a.n_nodes = 0; return false; }
    const DevGraph &g = MGX_PARAMS_OF(w).g;
    const int32_t n_kmers = a.seq_len - (int32_t)g.k + 1;
    LV<int32_t> nl;
    FOR_LANES(l) {
        LineCtr lc = { 0, 0, 0 };
        if (l == 0 && n_kmers > 0) {
            MapLane m;
            m.state = 0;
            bool given = false;
            auto fetch = [&](MapLane &ml) -> bool {
                if (given) return false;
                given = true;
                ml.strand = 1; ml.L = a.seq_len; ml.seq = (const char *)a.seq; ml.out = a.nodes;
                ml.out_len = nullptr; ml.out_rng = nullptr; ml.min_rng_len = 0; ml.n_kmers = n_kmers;
                return true;
            };
            while (m.state != 3) map_lane_step(g, m, lc, fetch);
        }
        nl[l] = (int32_t)(lc.rank_lines + lc.select_lines + lc.bit_lines);
    }
    w.ctr.rank_lines += (uint32_t)wave_sum(nl);
    wave_sync();
    a.n_nodes = imax(n_kmers, 0);
    // spelling and CIGAR reversed (the nodes above are already those of the reversed spelling)
    int32_t n = imax(a.n_cigar, a.seq_len);
    for (int32_t base = 0; base < (n + 1) / 2; base += WAVE) {
        FOR_LANES(l) {
            int32_t x = base + l;
            if (x < a.n_cigar / 2) { uint32_t t = a.cigar[x]; a.cigar[x] = a.cigar[a.n_cigar - 1 - x]; a.cigar[a.n_cigar - 1 - x] = t; }
            if (x < (a.seq_len + 1) / 2) {
                uint8_t t0 = complement_char(a.seq[x]), t1 = complement_char(a.seq[a.seq_len - 1 - x]);
                a.seq[x] = t1; a.seq[a.seq_len - 1 - x] = t0;
            }
        }
    }
    wave_sync();
    a.orientation = !a.orientation;
    int32_t clip = aln_clipping(a), eclip = aln_end_clipping(a);
    a.qbegin = clip;
    a.qlen = w.L - clip - eclip;
    return a.n_nodes > 0;
}

// Alignment::reverse_complement on a PRIMARY graph behind CanonicalDBG, for alignments without an offset:
// CanonicalDBG::reverse_complement(seq, path) (canonical_dbg.cpp:551-560) mirrors the path node by node — id v <-> v + n, a
// palindromic k-mer (even k) keeps its id; node x spells seq[x .. x + k) — and the rest is the plain reversal.
// (inlined into its callers: a third noinline level under aln_both is not safe on gfx950, see MGX_NI_MASK)
MGX_DEV bool reverse_complement_aln_primary(Wave &w, DevAln &a) {
    if (a.offset) { a.n_nodes = 0; return false; }
    const DevGraph &g = MGX_PARAMS_OF(w).g;
    const int32_t k = (int32_t)g.k;
    const uint32_t n = (uint32_t)g.n;
    for (int32_t base = 0; base < a.n_nodes; base += WAVE) {
        FOR_LANES(l) {
            const int32_t x = base + l;
            if (x < a.n_nodes) {
                uint32_t v = a.nodes[x];
                if (v > n) v -= n;
                else if (v) {
                    bool pal = !(k & 1) && x + k <= a.seq_len;
                    for (int32_t j = 0; pal && j < k / 2; ++j)
                        pal = encode_char(a.seq[x + j]) + encode_char(a.seq[x + k - 1 - j]) == 5;
                    if (!pal) v += n;
                }
                a.nodes[x] = v;
            }
        }
    }
    wave_sync();
    return reverse_complement_aln(w, a);
}

MGX_DEV bool reverse_complement_aln_stored(Wave &w, DevAln &a) {
    if (kWithPrimary && MGX_PARAMS_OF(w).cfg.canonical >= 2) return reverse_complement_aln_primary(w, a);
    return reverse_complement_aln_canonical(w, a);
}

MGX_DEV SeedRef seedref_from_aln(const DevAln &a) {
    SeedRef s;
    s.nodes = a.nodes; s.seq = a.seq; s.n_nodes = a.n_nodes; s.seq_len = a.seq_len;
    s.clipping = aln_clipping(a); s.end_clipping = aln_end_clipping(a); s.qlen = a.qlen;
    s.offset = a.offset; s.score = a.score; s.orientation = a.orientation;
#if MGX_WITH_LABELS
    s.lab = a.lab;
#endif
    return s;
}

// ------------------------------------------------------------------------------------------------
// aggregator (AlignmentAggregator, num_alternative_paths == 1, unlabeled; A/aligner_aggregator.hpp)
// ------------------------------------------------------------------------------------------------
MGX_DEV int32_t global_cutoff(const Wave &w) {              // :141-149
    if (!w.have_best) return NINF;
    int mx = 0;                                             // std::max_element: the first of equal maxima
    for (int t = 1; t < w.have_best; ++t) if (aln_less(w.aln[Q0 + mx], w.aln[Q0 + t])) mx = t;
    int32_t cur_max = w.aln[Q0 + mx].score;
    return cur_max > 0 ? (int32_t)((double)cur_max * MGX_PARAMS_OF(w).cfg.rel_score_cutoff) : cur_max;
}

MGX_DEV bool aln_equal(const Wave &w, const DevAln &a, const DevAln &b) {     // Alignment::operator== (alignment.hpp:261-269)
    if (a.orientation != b.orientation || a.offset != b.offset || a.score != b.score || a.qlen != b.qlen
            || a.seq_len != b.seq_len || a.n_cigar != b.n_cigar || a.n_nodes != b.n_nodes) return false;
    const uint8_t *qa = w.q[a.orientation] + a.qbegin, *qb = w.q[b.orientation] + b.qbegin;
    for (int32_t x = 0; x < a.qlen; ++x) if (qa[x] != qb[x]) return false;
    for (int32_t x = 0; x < a.seq_len; ++x) if (a.seq[x] != b.seq[x]) return false;
    for (int32_t x = 0; x < a.n_cigar; ++x) if (a.cigar[x] != b.cigar[x]) return false;
    for (int32_t x = 0; x < a.n_nodes; ++x) if (a.nodes[x] != b.nodes[x]) return false;
    return true;
}

MGX_NI_G4 void add_alignment(Wave &w, const DevAln &a) {
    MGX_ASSUME_LDS(&w);       // :68-138 (unlabeled): a queue of at most num_alternative_paths alignments
    const DevConfig &dc = MGX_PARAMS_OF(w).cfg;
    const int n_alt = dc.post_chain ? imax(1, imin((int)dc.agg_cap, N_ALN - 3 * n_alt_of(w))) : n_alt_of(w);
    if (!w.have_best) { copy_aln(w.aln[Q0], a); w.have_best = 1; return; }
    if (a.score < global_cutoff(w)) return;
    for (int t = 0; t < w.have_best; ++t) if (aln_equal(w, a, w.aln[Q0 + t])) return;
    if (w.have_best < n_alt) { copy_aln(w.aln[Q0 + w.have_best], a); ++w.have_best; return; }
    // post_chain_alignments: "never skip any alignments" (:92-96) — the host chains them (chain_host.hpp); a query with more
    // than the queue holds is a capacity status, not a silently shortened list
    // (have_best = -1 marks the cause: the result writer flags the record — ReadResult::orientation = RR_CAUSE_QUEUE — so that the
    // host's capacity retry does not try to cure with larger arenas what larger arenas cannot cure)
    if (dc.post_chain) { w.status = ST_CAPACITY; w.have_best = -1; return; }
    int mn = 0;                                             // std::min_element: the first of equal minima
    for (int t = 1; t < w.have_best; ++t) if (aln_less(w.aln[Q0 + t], w.aln[Q0 + mn])) mn = t;
    if (aln_less(a, w.aln[Q0 + mn])) return;
    copy_aln(w.aln[Q0 + mn], a);
}

// ------------------------------------------------------------------------------------------------
// driver (DBGAligner<>::align_batch for one query, A/dbg_aligner.cpp:263-355,360-384,531-758)
// ------------------------------------------------------------------------------------------------
MGX_DEV SeedRef seedref_from_seed(const Wave &w, int s, int32_t idx, int32_t *sub_node_slot) {
    const DevConfig &cfg = MGX_PARAMS_OF(w).cfg;
    DevSeed sd = w.seeds[s][idx];
    SeedRef r;
    r.clipping = sd.clipping; r.qlen = sd.length; r.offset = sd.offset; r.n_nodes = sd.n_nodes;
    r.end_clipping = w.L - sd.clipping - sd.length;
    r.seq = w.q[s] + sd.clipping; r.seq_len = sd.length;
    r.nodes = (sd.offset == 0 && sd.n_nodes >= 1) ? w.nodes[s] + sd.clipping : &w.seeds[s][idx].node;
    if (sd.offset == 0 && sd.n_nodes == 1) r.nodes = w.nodes[s] + sd.clipping;
    r.orientation = s;
    // Alignment(const Seed&, config) score (alignment.hpp:160-161)
    int32_t ms = w.psum_lin[s] ? (int32_t)sd.length * w.psum_lin[s]
                               : w.psum[s][sd.clipping] - w.psum[s][sd.clipping + sd.length];
    r.score = ms + (!sd.clipping ? cfg.left_end_bonus : 0) + (!r.end_clipping ? cfg.right_end_bonus : 0);
#if MGX_WITH_LABELS
    r.lab = MGX_PARAMS_OF(w).labeled ? w.seed_lab[s][idx] : 0;
#endif
    (void)sub_node_slot;
    return r;
}

MGX_DEV int32_t min_path_score_now(const Wave &w) {          // get_min_path_score (:277-282)
    return imax(MGX_PARAMS_OF(w).cfg.min_path_score, global_cutoff(w));
}

// ---- resume records of the multi-pass extension (AlignParams::resume_*) ----
// a DevAln whose arrays live in a record
MGX_DEV DevAln resume_aln_view(const DevLimits &lim, uint8_t *p) {
    DevAln a;
    a.nodes = (uint32_t *)(p + 48);
    a.cigar = (uint32_t *)(p + 48 + align8((uint64_t)lim.max_path * 4));
    a.seq = p + 48 + 2 * align8((uint64_t)lim.max_path * 4);
    a.n_nodes = a.n_cigar = a.seq_len = 0; a.score = a.offset = a.qbegin = a.qlen = a.orientation = a.extra_score = 0;
    return a;
}
MGX_DEV void resume_save(Wave &w, uint8_t *rec) {
    const DevLimits &lim = MGX_PARAMS_OF(w).lim;
    const int n_alt = n_alt_of(w);
    int32_t *h = (int32_t *)rec;
    FOR_LANES(l) {
        if (l == 0) {
            h[0] = w.resume_phase; h[1] = w.resume_i; h[2] = w.have_best;
            h[3] = (int32_t)w.ext[0].table_cap; h[4] = (int32_t)w.ext[1].table_cap;
            h[5] = (int32_t)w.n_extensions; h[6] = (int32_t)w.n_columns; h[7] = (int32_t)w.n_fast_columns;
        }
    }
    uint8_t *al = rec + 64;
    for (int s = 0; s < 2; ++s) {
        const int32_t n = w.n_seeds[s];
        for (int32_t base = 0; base < n; base += WAVE) {
            FOR_LANES(l) { const int32_t x = base + l; if (x < n) al[(uint32_t)s * lim.max_seeds + x] = w.alive[s][x]; }
        }
    }
    uint8_t *ap = rec + 64 + align8(2ull * lim.max_seeds);
    for (int t = 0; t < w.have_best; ++t) {
        uint8_t *p = ap + (uint32_t)t * resume_aln_bytes(lim);
        DevAln v = resume_aln_view(lim, p);
        copy_aln(v, w.aln[Q0 + t]);
        int32_t *sc = (int32_t *)p;
        FOR_LANES(l) {
            if (l == 0) {
                sc[0] = v.n_nodes; sc[1] = v.n_cigar; sc[2] = v.seq_len; sc[3] = v.score; sc[4] = v.offset;
                sc[5] = v.qbegin; sc[6] = v.qlen; sc[7] = v.orientation; sc[8] = v.extra_score;
            }
        }
    }
    wave_sync();
    (void)n_alt;
}
MGX_DEV void resume_load(Wave &w, const uint8_t *rec) {
    const DevLimits &lim = MGX_PARAMS_OF(w).lim;
    const int32_t *h = (const int32_t *)rec;
    w.resume_phase = uni(h[0]); w.resume_i = uni(h[1]); w.have_best = uni(h[2]);
    w.ext[0].table_cap = (uint32_t)uni(h[3]); w.ext[1].table_cap = (uint32_t)uni(h[4]);
    w.n_extensions = (uint32_t)uni(h[5]); w.n_columns = (uint32_t)uni(h[6]); w.n_fast_columns = (uint32_t)uni(h[7]);
    const uint8_t *al = rec + 64;
    for (int s = 0; s < 2; ++s) {
        const int32_t n = w.n_seeds[s];
        for (int32_t base = 0; base < n; base += WAVE) {
            FOR_LANES(l) { const int32_t x = base + l; if (x < n) w.alive[s][x] = al[(uint32_t)s * lim.max_seeds + x]; }
        }
    }
    const uint8_t *ap = rec + 64 + align8(2ull * lim.max_seeds);
    for (int t = 0; t < w.have_best; ++t) {
        uint8_t *p = const_cast<uint8_t *>(ap) + (uint32_t)t * resume_aln_bytes(lim);
        DevAln v = resume_aln_view(lim, p);
        const int32_t *sc = (const int32_t *)p;
        v.n_nodes = uni(sc[0]); v.n_cigar = uni(sc[1]); v.seq_len = uni(sc[2]); v.score = uni(sc[3]); v.offset = uni(sc[4]);
        v.qbegin = uni(sc[5]); v.qlen = uni(sc[6]); v.orientation = uni(sc[7]); v.extra_score = uni(sc[8]);
        copy_aln(w.aln[Q0 + t], v);
    }
    wave_sync();
}
// a read that would start another seed beyond this pass's limit: true = stop here (status ST_RETRY, resume point recorded)
MGX_DEV bool pass_limit_reached(Wave &w, int32_t next_i) {
    const AlignParams &P = MGX_PARAMS_OF(w);
    if (!P.seed_limit || w.no_limit || w.seeds_done < (int32_t)P.seed_limit) return false;
    w.resume_i = next_i;
    w.status = ST_RETRY;
    return true;
}

// aln_both (:657-736): seeds of strand s; fwd extender = ext[s] on the graph, bwd extender = ext[1 - s] on RCDBG
MGX_NI_G4 void aln_both(Wave &w, int s) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    ExtenderState &F = w.ext[s];
    ExtenderState &B = w.ext[1 - s];
    // A CANONICAL-mode graph holds both strands itself (:644-655): the backward pass runs on the same graph, alignments are
    // flipped by re-mapping their reversed spelling, and an alignment on the reverse strand is reported as the forward
    // alignment it mirrors (is_reversible: orientation && !offset).
    const bool canon = P.cfg.canonical != 0;
    F.rc_view = 0;
    B.rc_view = canon ? 0 : 1;
    const int32_t n = w.n_seeds[s];
    const int32_t i0 = w.resume_i;
    w.resume_i = 0;
    for (int32_t i = i0; i < n; ++i) {
        if (!w.alive[s][i]) continue;
        if (pass_limit_reached(w, i)) return;
        ++w.seeds_done;
        const uint64_t tp0 = cycle_clock();
        SeedRef seed = seedref_from_seed(w, s, i, nullptr);
        conv_clear(w, F.conv);                                // set_seed (:90-98)
        uint64_t t0 = cycle_clock();
        w.cyc[6] += t0 - tp0;
        extend(w, s, seed, false);
        uint64_t t1 = cycle_clock();
        w.cyc[2] += t1 - t0;
        const ExtendResult er = w.er;
        if (w.status != ST_OK) return;
        int32_t mps = imax(0, P.cfg.min_cell_score);          // extend(): min_path_score = max(0, min_cell_score)
        const int n_alt = n_alt_of(w);
        int n_fwd;
        if (MGX_ABLATED(w, 4u)) { seed_as_alignment(w, seed, w.aln[0]); n_fwd = 1; }       // timing probe only
        else n_fwd = backtrack(w, s, seed, nullptr, er, mps, &w.aln[0], n_alt);
        w.cyc[3] += cycle_clock() - t1;
        if (w.status != ST_OK) return;
        // every extension of this seed goes to the aggregator; those that can be continued to the left are reversed and
        // become the seed list of ONE align_core on the backward extender (:677-729) ...
        int n_rev = 0;
        bool rev_alive[MAX_ALT];
        {
            const uint64_t tm0 = cycle_clock();
            for (int e = 0; e < n_fwd; ++e) {
                DevAln &path = w.aln[e];
                DevAln &rev = w.aln[n_alt + n_rev];
                if (canon) {
                    // the mirror image serves both purposes: what is reported for a reverse-strand alignment (:683-689) and
                    // the seed of the backward pass (:695) — the same reverse_complement(graph_, query_rc)
                    const bool reversible = path.orientation && !path.offset;
                    const bool to_left = aln_clipping(path) && !path.offset;
                    const bool good = path.score >= min_path_score_now(w);
                    bool have_rev = false;
                    if ((good && reversible) || to_left) { copy_aln(rev, path); have_rev = reverse_complement_aln_stored(w, rev); }
                    if (good) { if (reversible) { if (have_rev) add_alignment(w, rev); } else add_alignment(w, path); }
                    if (!to_left || !have_rev) continue;
                    rev_alive[n_rev++] = true;
                    continue;
                }
                if (path.score >= min_path_score_now(w)) add_alignment(w, path);
                if (!aln_clipping(path) || path.offset) continue;
                copy_aln(rev, path);
                if (!reverse_complement_aln(w, rev)) continue;
                rev_alive[n_rev++] = true;
            }
            w.cyc[7] += cycle_clock() - tm0;
        }
        // ... which extends them in order and drops a later one whose end the earlier extensions already reached with at
        // least its score (align_core :360-384: check_seed between the seeds of the list)
        const uint32_t *filt_nodes = nullptr;                 // lazily applied filter_nodes (see below)
        int32_t filt_n = 0, filt_lo = 0, filt_hi = 0;
        // filter_seed (:105-108, :731-734) for every later seed: independent look-ups into the forward extender's convergence
        // table, one seed per lane (reads of a pan-genome carry ~100 sub-k seeds: one after the other this loop was two thirds
        // of the kernel there).  With one alignment per seed nothing writes that table between the forward extension and
        // these checks (the backward pass has its own, filter_nodes is applied lazily), so the table part runs BEFORE the
        // backward pass, while the column table its alias entries point into is still the forward extension's; the filter
        // part follows the backward pass.  With several alignments per seed both run after it, on pool entries only.
        auto check_later = [&](bool with_table, bool with_filter) {
            for (int32_t base = i + 1; base < n; base += WAVE) {
                FOR_LANES(l) {
                    const int32_t j = base + l;
                    if (j < n && w.alive[s][j]) {
                        DevSeed sj = w.seeds[s][j];
                        uint32_t last_node = sj.offset == 0 ? w.nodes[s][sj.clipping + sj.n_nodes - 1] : sj.node;
                        SeedRef rj = seedref_from_seed(w, s, j, nullptr);
                        bool dead = with_table && !check_seed(w, F, last_node, rj.qlen, rj.clipping, rj.score);
                        // the deferred filter_nodes marks: position qlen + clipping - 1 of a node on the reversed backward
                        // alignment holds the maximal score (check_seed: vec[pos] < score is false)
                        const int32_t pos = rj.qlen + rj.clipping - 1;
                        if (with_filter && !dead && filt_n && pos >= filt_lo && pos < filt_hi) {
                            for (int32_t x = 0; x < filt_n; ++x) dead |= gld(filt_nodes + x) == last_node;
                        }
                        if (dead) w.alive[s][j] = 0;
                    }
                }
            }
            wave_sync();
        };
        if (n_alt == 1) check_later(true, false);
        for (int r = 0; r < n_rev; ++r) {
            if (!rev_alive[r]) continue;
            DevAln &rev = w.aln[n_alt + r];
            SeedRef rseed = seedref_from_aln(rev);
            int32_t mps2 = imax(0, min_path_score_now(w));
            conv_clear(w, B.conv);
            uint64_t t2 = cycle_clock();
            extend(w, 1 - s, rseed, true);
            uint64_t t3 = cycle_clock();
            w.cyc[2] += t3 - t2;
            const ExtendResult er2 = w.er;
            if (w.status != ST_OK) return;
            int n_bwd;
            if (MGX_ABLATED(w, 4u)) { copy_aln(w.aln[2 * n_alt], rev); n_bwd = 1; }         // timing probe only
            else n_bwd = backtrack(w, 1 - s, rseed, &rev, er2, mps2, &w.aln[2 * n_alt], n_alt);
            w.cyc[3] += cycle_clock() - t3;
            if (w.status != ST_OK) return;
            for (int b = 0; b < n_bwd; ++b) {
                DevAln &p2 = w.aln[2 * n_alt + b];
                if (canon && !(p2.orientation && !p2.offset)) { add_alignment(w, p2); continue; }     // not reversible: as it is (:711)
                if (canon ? reverse_complement_aln_stored(w, p2) : reverse_complement_aln(w, p2)) {
                    int32_t clip = aln_clipping(p2), eclip = aln_end_clipping(p2);
                    const uint64_t tf0 = cycle_clock();
                    if (n_alt == 1) {
                        // filter_nodes (:716-719) marks [clip, L - eclip) of every path node with the maximal score in the FORWARD
                        // extender's table.  Nothing reads that table before the next set_seed clears it except check_seed on
                        // the later seeds below, so with one alignment per seed (one p2, which stays in its buffer until the next
                        // seed's backtrack) the marks are not written at all: the check below tests the path's nodes directly.
                        // (Written out, the marks were 150 vectors of 150 words per backward pass.)
                        filt_nodes = p2.nodes; filt_n = p2.n_nodes; filt_lo = clip; filt_hi = w.L - eclip;
                    } else {
                        for (int32_t x = 0; x < p2.n_nodes; ++x)
                            filter_nodes(w, F, p2.nodes[x], clip, w.L - eclip);
                    }
                    w.cyc[6] += cycle_clock() - tf0;                      // (timer 6: seed pick-up + filter_nodes)
                    if (w.status != ST_OK) return;
                    add_alignment(w, p2);
                }
            }
            for (int r2 = r + 1; r2 < n_rev; ++r2) {
                if (!rev_alive[r2]) continue;
                const DevAln &o = w.aln[n_alt + r2];
                if (!check_seed(w, B, o.nodes[o.n_nodes - 1], o.qlen, aln_clipping(o), o.score)) rev_alive[r2] = false;
            }
        }
        if (n_alt == 1) { if (filt_n) check_later(false, true); }
        else check_later(true, false);
    }
}

// align_core (:360-384) with the seeds of strand 0, forward only
MGX_NI_G4 void align_core_fwd(Wave &w) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    ExtenderState &F = w.ext[0];
    F.rc_view = 0;
    const int32_t n = w.n_seeds[0];
    const int32_t i0 = w.resume_i;
    w.resume_i = 0;
    for (int32_t i = i0; i < n; ++i) {
        if (!w.alive[0][i]) continue;
        if (pass_limit_reached(w, i)) return;
        ++w.seeds_done;
        SeedRef seed = seedref_from_seed(w, 0, i, nullptr);
        int32_t mps = imax(0, min_path_score_now(w));
        conv_clear(w, F.conv);
        extend(w, 0, seed, false);
        const ExtendResult er = w.er;
        if (w.status != ST_OK) return;
        {
            const int n_fwd = backtrack(w, 0, seed, nullptr, er, mps, &w.aln[0], n_alt_of(w));
            if (w.status != ST_OK) return;
            for (int e = 0; e < n_fwd; ++e) add_alignment(w, w.aln[e]);
        }
        if (w.status != ST_OK) return;
        for (int32_t base = i + 1; base < n; base += WAVE) {
            FOR_LANES(l) {
                const int32_t j = base + l;
                if (j < n && w.alive[0][j]) {
                    DevSeed sj = w.seeds[0][j];
                    uint32_t last_node = sj.offset == 0 ? w.nodes[0][sj.clipping + sj.n_nodes - 1] : sj.node;
                    SeedRef rj = seedref_from_seed(w, 0, j, nullptr);
                    if (!check_seed(w, F, last_node, rj.qlen, rj.clipping, rj.score)) w.alive[0][j] = 0;
                }
            }
        }
        wave_sync();
        (void)P;
    }
}

#if MGX_WITH_LABELS
#include "label_driver.hpp"
#endif

#endif  // MGX_NO_EXTEND

// Predicted extension work of a read from its seeds: (number of extensions, columns), the sort key that
// lets the sub-wave groups of one wavefront work on similar reads.  Any value is correct; a better
// prediction only means less idling.
constexpr int WORK_SEGMENT_SHIFT = 21;       // 2 M reads per segment of the work-sorted order (batches up to 2^41 reads)
MGX_DEV uint32_t predicted_work(const Wave &w) {
    const int first = w.num_matching[0] >= w.num_matching[1] ? 0 : 1;
    if (!w.n_seeds[first]) return 0;
    // The forward extension starts at the first seed's first node and replays the seed, so it computes about
    // L - clipping columns; a clipped start adds the backward pass, which replays the whole forward alignment
    // before it reaches the clipped prefix (about L more columns).
    const int32_t clip = w.seeds[first][0].clipping;
    const int32_t cols = (w.L - clip) + (clip > 0 ? w.L : 0);
    return 1u + (uint32_t)imin(4094, cols);
}

// The whole per-read program; `slot` selects the arena slice.  PHASE splits it for the two-kernel
// pipeline: PH_SEED stops after build_seeders and publishes the seeds, PH_EXTEND picks them up.
template <int PHASE = PH_BOTH>
MGX_DEV void align_read(Wave &w, const AlignParams &P, uint64_t read, uint32_t slot, KernelStats *stats_accum,
                        SdustScratch *sd, const int8_t *sm_rows, uint8_t *lds, uint32_t lds_bytes,
                        const uint8_t *resume_rec = nullptr) {
#if !(defined(MGX_PARAMS_IN_LDS) && MGX_PARAMS_IN_LDS)
    w.P = &P;
    w.sm_rows = sm_rows;
#else
    (void)sm_rows;
#endif
    carve(w, P, P.arena + (uint64_t)slot * P.arena_stride, lds, lds_bytes);
    w.sd = sd ? sd : w.sd_own;
    const uint64_t off = P.offsets[read];
    w.L = (int32_t)(P.offsets[read + 1] - off);
    w.status = ST_OK;
    w.have_best = 0;
    w.seeds_done = 0;
    w.resume_phase = 0; w.resume_i = 0; w.no_limit = 0;
    int64_t retry_pos = -1;              // this read's position in the retry list once its record is written
    w.ctr.rank_lines = w.ctr.select_lines = w.ctr.bit_lines = 0;
    w.n_columns = w.n_extensions = w.n_fast_columns = 0;
    const uint64_t nb = P.node_begin[read];
    w.n_kmers = (int32_t)(P.node_begin[read + 1] - nb);
    w.nodes[0] = P.nodes_fwd + nb;
    w.nodes[1] = P.nodes_rc + nb;
    w.mlen[0] = P.mlen_fwd ? P.mlen_fwd + nb : nullptr;
    w.mlen[1] = P.mlen_rc ? P.mlen_rc + nb : nullptr;
    w.rng[0] = P.rng_fwd ? P.rng_fwd + nb : nullptr;
    w.rng[1] = P.rng_rc ? P.rng_rc + nb : nullptr;
    ReadResult rr;
    rr.status = ST_OK; rr.n_alignments = 0; rr.score = 0; rr.offset = 0; rr.n_nodes = rr.n_cigar = rr.seq_len = 0;
    rr.orientation = 0; rr.stream_off = 0;
    rr.num_matches_fwd = rr.num_matches_rc = rr.n_seeds_fwd = rr.n_seeds_rc = 0; rr.n_extensions = rr.n_columns = 0;

    for (int x = 0; x < 8; ++x) w.cyc[x] = 0;
    for (int x = 0; x < XCYC_N; ++x) w.xcyc[x] = 0;
    const uint64_t tstart = cycle_clock();
    if (w.L > (int32_t)P.lim.Lmax) {
        w.status = ST_CAPACITY;
    } else {
        prepare_query(w, P.seqs + off, (PHASE & PH_SEED) != 0, (PHASE & PH_EXTEND) != 0, read);
        w.lc_any[0] = w.lc_any[1] = -1;
        w.lc_maybe = -1;
        w.cyc[0] = cycle_clock() - tstart;
        for (int s = 0; s < 2; ++s) {
            w.ext[s].q = w.q[s];
            w.ext[s].psum = w.psum[s];
            w.ext[s].psum_lin = (PHASE & PH_EXTEND) ? w.psum_lin[s] : 0;
            w.ext[s].table_cap = 0;
            w.ext[s].rc_view = 0;
            w.ext[s].conv.n_entries = 0; w.ext[s].conv.n_recs = 0; w.ext[s].conv.pool_top = 0;
            w.ext[s].conv.cap = conv_cap0(P.lim.hash_size); w.ext[s].conv.base = 0; w.ext[s].conv.start = 0;
        }
        // generation tags make clearing the hash tables O(1); the counters persist in the arena
        // slice across the reads a wave slot processes
        for (int s = 0; s < 2; ++s) { w.ext[s].conv.gen = w.gen_store[2 * s]; w.ext[s].conv.dirty = w.gen_store[2 * s + 1]; }
        const bool have_rc = P.cfg.fwd_and_rc != 0;
        // build_seeders (:193-248)
        const uint64_t tseed = cycle_clock();
        if constexpr (PHASE & PH_SEED) {
            const bool many = (uint32_t)P.g.k >= P.cfg.max_seed_length;
            if (strand_without_seeds(w, 0)) { w.n_seeds[0] = 0; w.num_matching[0] = 0; } else
            if (many) make_seeder<true>(w, 0); else make_seeder<false>(w, 0);
            if ((double)w.L * P.cfg.min_exact_match > (double)w.num_matching[0]) { w.n_seeds[0] = 0; w.num_matching[0] = 0; }
            if (have_rc && w.status == ST_OK) {
                if (strand_without_seeds(w, 1)) { w.n_seeds[1] = 0; w.num_matching[1] = 0; } else
                if (many) make_seeder<true>(w, 1); else make_seeder<false>(w, 1);
                if ((double)w.L * P.cfg.min_exact_match > (double)w.num_matching[1]) { w.n_seeds[1] = 0; w.num_matching[1] = 0; }
            } else {
                w.n_seeds[1] = 0; w.num_matching[1] = 0;
            }
        } else {
            // seeds of the seeding kernel
            const SeedHdr h = P.seed_hdr[read];
            w.status = h.status;
            for (int s = 0; s < 2; ++s) { w.n_seeds[s] = h.n_seeds[s]; w.num_matching[s] = h.num_matching[s]; }
            if (w.status == ST_OK) {
                const DevSeed *src = P.seed_stream + h.off;
                for (int s = 0; s < 2; ++s) {
                    const int32_t n = w.n_seeds[s];
                    for (int32_t base = 0; base < n; base += WAVE) {
                        FOR_LANES(l) {
                            int32_t x = base + l;
                            if (x < n) { w.seeds[s][x] = src[x]; w.alive[s][x] = 1; }
                        }
                    }
                    src += n;
                }
            }
            wave_sync();
            if (MGX_ABLATED(w, 8u)) { w.n_seeds[0] = w.n_seeds[1] = 0; }          // timing probe: fetch + pick-up + output only
        }
#if MGX_WITH_LABELS && !defined(MGX_NO_EXTEND)
        if ((PHASE & PH_EXTEND) && P.labeled && w.status == ST_OK) {
            // LabeledAligner::build_seeders (aligner_labeled.cpp:479-558): the label filter on top of the seeders' output
            w.lab[0] = 0; w.lab_lo = 1; w.lab_hi = P.lim.lab_words;
            lab_agg_reset(w);
            lab_filter_seeds(w, 0);
            if (have_rc && w.status == ST_OK) lab_filter_seeds(w, 1);
        }
#endif
        if constexpr (PHASE == PH_SEED) {
            // publish: header, seeds, work key; the extension kernel writes the read's result record
            SeedHdr h;
            h.status = w.status; h.pad = 0; h.off = 0;
            for (int s = 0; s < 2; ++s) { h.n_seeds[s] = (uint16_t)w.n_seeds[s]; h.num_matching[s] = w.num_matching[s]; }
            const uint32_t total = (uint32_t)(w.n_seeds[0] + w.n_seeds[1]);
            if (w.status == ST_OK && total) {
                LV<uint64_t> offv;
                FOR_LANES(l) {
                    offv[l] = 0;
                    if (l == 0) {
#if MGX_WAVE_EMU
                        offv[l] = *P.seed_cursor; *P.seed_cursor += total;
#else
                        offv[l] = atomicAdd(P.seed_cursor, (unsigned long long)total);
#endif
                    }
                }
                h.off = wave_bcast(offv, 0);
                if (h.off + total > P.seed_capacity) {
                    h.status = ST_CAPACITY;
                } else {
                    DevSeed *dst = P.seed_stream + h.off;
                    for (int s = 0; s < 2; ++s) {
                        const int32_t n = w.n_seeds[s];
                        for (int32_t base = 0; base < n; base += WAVE) {
                            FOR_LANES(l) { int32_t x = base + l; if (x < n) dst[x] = w.seeds[s][x]; }
                        }
                        dst += n;
                    }
                }
            }
            // sort key: (segment of the batch, predicted work).  Ordering the whole of a large batch by work alone makes the
            // extension kernel hop across all of the batch's arrays (tens of GB at 10 M reads: measured 14 % slower per read
            // than at 5 M); within segments of WORK_SEGMENT reads the order is by work, and the segments follow each other.
            const uint32_t key = (h.status == ST_OK ? predicted_work(w) : 0) | ((uint32_t)(read >> WORK_SEGMENT_SHIFT) << 12);
            FOR_LANES(l) { if (l == 0) { P.seed_hdr[read] = h; P.work_key[read] = key; } }
        }
        rr.num_matches_fwd = w.num_matching[0]; rr.num_matches_rc = w.num_matching[1];
        rr.n_seeds_fwd = (uint32_t)w.n_seeds[0]; rr.n_seeds_rc = (uint32_t)w.n_seeds[1];
        if (P.dbg_seeds && w.status == ST_OK) {
            for (int s = 0; s < 2; ++s)
                for (int32_t i = 0; i < w.n_seeds[s]; ++i)
                    P.dbg_seeds[((uint64_t)read * 2 + s) * P.lim.max_seeds + i] = w.seeds[s][i];
        }
        wave_sync();
        w.cyc[1] = cycle_clock() - tseed;
        const uint64_t tdrv = cycle_clock();
#ifndef MGX_NO_EXTEND
        if ((PHASE & PH_EXTEND) && w.status == ST_OK) {
            if (resume_rec) resume_load(w, resume_rec);          // a later pass: the aggregator, the live seeds, where it stopped
            for (;;) {
                if (have_rc) {
                    // align_both_directions (:738-755)
                    uint32_t fm = w.num_matching[0], bm = w.num_matching[1];
                    // one call site for both orders and both passes: groups of a wavefront that are on different strands
                    // stay converged
                    const int first = fm >= bm ? 0 : 1;
                    const uint32_t m_first = first ? bm : fm, m_second = first ? fm : bm;
                    for (int ph = w.resume_phase; ph < 2 && w.status == ST_OK; ++ph) {
                        if (ph == 1 && !((double)m_second >= (double)m_first * P.cfg.rel_score_cutoff)) break;
                        if (ph != w.resume_phase) w.resume_i = 0;
                        w.resume_phase = ph;
#if MGX_WITH_LABELS
                        if (P.labeled) { lab_aln_both(w, ph == 0 ? first : 1 - first); continue; }
#endif
                        aln_both(w, ph == 0 ? first : 1 - first);
                    }
                } else {
#if MGX_WITH_LABELS
                    if (P.labeled) lab_align_core_fwd(w); else
#endif
                    align_core_fwd(w);
                }
                if (w.status != ST_RETRY || !P.resume_out) break;
                // stopped by the pass's seed limit: take a retry position; without room for a record the read just goes on
                LV<uint64_t> pv;
                FOR_LANES(l) {
                    pv[l] = 0;
                    if (l == 0) {
#if MGX_WAVE_EMU
                        pv[l] = (*P.retry_count)++;
#else
                        pv[l] = atomicAdd(P.retry_count, 1ull);
#endif
                    }
                }
                const uint64_t pos = wave_bcast(pv, 0);
                if (pos >= P.resume_cap) { w.no_limit = 1; w.status = ST_OK; continue; }
                retry_pos = (int64_t)pos;
                resume_save(w, P.resume_out + pos * P.resume_rec_bytes);
                // work key of the next pass: this read's next live seed (same prediction as the seeding kernel's)
                uint32_t key = 1;
                {
                    const int sidx = w.resume_phase == 0 ? (w.num_matching[0] >= w.num_matching[1] ? 0 : 1)
                                                         : (w.num_matching[0] >= w.num_matching[1] ? 1 : 0);
                    const int ss = have_rc ? sidx : 0;
                    if (w.resume_i < w.n_seeds[ss]) {
                        const int32_t clip = w.seeds[ss][w.resume_i].clipping;
                        key = 1u + (uint32_t)imin(4094, (w.L - clip) + (clip > 0 ? w.L : 0));
                    }
                }
                FOR_LANES(l) { if (l == 0) { P.retry_list[pos] = (uint32_t)read; P.retry_key[pos] = key; } }
                break;
            }
        }
#else
        static_assert(PHASE == PH_SEED, "this translation unit was built without the extension half");
#endif
        for (int s = 0; s < 2; ++s) { w.gen_store[2 * s] = w.ext[s].conv.gen; w.gen_store[2 * s + 1] = w.ext[s].conv.dirty; }
        wave_sync();
        w.cyc[4] = cycle_clock() - tdrv - w.cyc[2] - w.cyc[3];
    }
    if constexpr (PHASE == PH_SEED) {
        if (w.L > (int32_t)P.lim.Lmax) {
            SeedHdr h;
            h.off = 0; h.status = ST_CAPACITY; h.pad = 0;
            h.n_seeds[0] = h.n_seeds[1] = 0; h.num_matching[0] = h.num_matching[1] = 0;
            FOR_LANES(l) { if (l == 0) { P.seed_hdr[read] = h; P.work_key[read] = (uint32_t)(read >> WORK_SEGMENT_SHIFT) << 12; } }
        }
        stats_accum->rank_lines += w.ctr.rank_lines;
        stats_accum->select_lines += w.ctr.select_lines;
        stats_accum->bit_lines += w.ctr.bit_lines;
        stats_accum->seed_lines += w.ctr.rank_lines + w.ctr.select_lines + w.ctr.bit_lines;
        stats_accum->seeds += (uint32_t)(w.n_seeds[0] + w.n_seeds[1]);
#if !defined(MGX_SEED_PROBE) && !defined(MGX_CHAIN_PROBE)
        // the seeding kernel's two timers go to the spare slots of the extension breakdown, so that cyc[] is the extension
        // kernel's alone (the two kernels run different numbers of lanes per read: their cycles do not add up)
        for (int x = 0; x < 2; ++x) stats_accum->xcyc[6 + x] += w.cyc[x];
#endif
#ifdef MGX_SEED_PROBE
        for (int x = 0; x < 2; ++x) stats_accum->cyc[x] += w.cyc[x];           // probe build: the seeding kernel's own timers
        for (int x = 0; x < XCYC_N; ++x) stats_accum->xcyc[x] += w.xcyc[x];
#endif
        return;
    }
    if (w.status == ST_RETRY) {
        // this read goes on to another seed in the next pass: with a resume record (written above) or, without records,
        // untouched and from scratch
        FOR_LANES(l) {
            if (l == 0 && retry_pos < 0) {
#if MGX_WAVE_EMU
                P.retry_list[(*P.retry_count)++] = (uint32_t)read;
#else
                P.retry_list[atomicAdd(P.retry_count, 1ull)] = (uint32_t)read;
#endif
            }
        }
        stats_accum->rank_lines += w.ctr.rank_lines;
        stats_accum->select_lines += w.ctr.select_lines;
        stats_accum->bit_lines += w.ctr.bit_lines;
        for (int x = 0; x < 8; ++x) stats_accum->cyc[x] += w.cyc[x];
    for (int x = 0; x < XCYC_N; ++x) stats_accum->xcyc[x] += w.xcyc[x];
        return;
    }
    const uint64_t tout = cycle_clock();

    rr.status = w.status;
    if (w.status == ST_CAPACITY && w.have_best < 0) rr.orientation = RR_CAUSE_QUEUE;
    rr.n_extensions = w.n_extensions; rr.n_columns = w.n_columns;
#if MGX_WITH_LABELS && !defined(MGX_NO_EXTEND)
    if (P.labeled && w.status == ST_OK && (PHASE & PH_EXTEND)) {
        // the labeled aggregator's alignments; stream layout as below, with every alignment followed by its label count and
        // its labels (ascending)
        uint32_t *order = w.agg + ag_list0(P.lim) + (P.lim.lab_queues * 4 + 16);
        const int n_out = lab_get_alignments(w, order);
        uint32_t words = 0;
        for (int t = 0; t < n_out; ++t) {
            const DevAln &a = lab_pool_aln(w, order[t]);
            words += (t ? 6u : 0u) + (uint32_t)a.n_nodes + (uint32_t)a.n_cigar + ((uint32_t)a.seq_len + 3) / 4 + 1u + (a.lab ? lab_size(w, a.lab) : 0u);
        }
        if (n_out) {
            LV<uint64_t> offv;
            FOR_LANES(l) {
                offv[l] = 0;
                if (l == 0) {
#if MGX_WAVE_EMU
                    offv[l] = *P.out_cursor; *P.out_cursor += words;
#else
                    offv[l] = atomicAdd(P.out_cursor, (unsigned long long)words);
#endif
                }
            }
            const uint64_t so = wave_bcast(offv, 0);
            if (so + words > P.out_capacity) {
                rr.status = ST_CAPACITY;
            } else {
                uint32_t *dst = P.out_stream + so;
                for (int t = 0; t < n_out; ++t) {
                    const DevAln &a = lab_pool_aln(w, order[t]);
                    if (t == 0) {
                        rr.score = a.score; rr.offset = (uint32_t)a.offset;
                        rr.n_nodes = (uint32_t)a.n_nodes; rr.n_cigar = (uint32_t)a.n_cigar; rr.seq_len = (uint32_t)a.seq_len;
                        rr.orientation = (uint32_t)a.orientation; rr.stream_off = so;
                    } else {
                        dst[0] = (uint32_t)a.score; dst[1] = (uint32_t)a.offset; dst[2] = (uint32_t)a.n_nodes;
                        dst[3] = (uint32_t)a.n_cigar; dst[4] = (uint32_t)a.seq_len; dst[5] = (uint32_t)a.orientation;
                        dst += 6;
                    }
                    uint8_t *dseq = (uint8_t *)(dst + a.n_nodes + a.n_cigar);
                    const int32_t n = imax(imax(a.n_nodes, a.n_cigar), a.seq_len);
                    for (int32_t base = 0; base < n; base += WAVE) {
                        FOR_LANES(l) {
                            int32_t x = base + l;
                            if (x < a.n_nodes) dst[x] = a.nodes[x];
                            if (x < a.n_cigar) dst[a.n_nodes + x] = a.cigar[x];
                            if (x < a.seq_len) dseq[x] = a.seq[x];
                        }
                    }
                    dst += (uint32_t)a.n_nodes + (uint32_t)a.n_cigar + ((uint32_t)a.seq_len + 3) / 4;
                    const uint32_t nl = a.lab ? lab_size(w, a.lab) : 0u;
                    dst[0] = nl;
                    for (uint32_t x = 0; x < nl; ++x) dst[1 + x] = lab_at(w, a.lab, x);
                    dst += 1 + nl;
                }
                rr.n_alignments = n_out;
            }
        }
    } else
#endif
    if (w.status == ST_OK && w.have_best) {
        // get_alignments (aligner_aggregator.hpp:180-202): stable sort ascending by LocalAlignmentLess, emitted from the
        // back, empty alignments dropped
        int ord[N_ALN];                   // (post_chain_alignments: the queue may hold more than MAX_ALT)
        int n_out = 0;
        for (int t = 0; t < w.have_best; ++t) {
            int pos = t;
            while (pos > 0 && aln_less(w.aln[Q0 + t], w.aln[Q0 + ord[pos - 1]])) { ord[pos] = ord[pos - 1]; --pos; }
            ord[pos] = t;
        }
        uint32_t words = 0;
        for (int t = w.have_best - 1; t >= 0; --t) {
            const DevAln &a = w.aln[Q0 + ord[t]];
            if (!a.n_nodes) continue;
            words += (n_out ? 6u : 0u) + (uint32_t)a.n_nodes + (uint32_t)a.n_cigar + ((uint32_t)a.seq_len + 3) / 4;
            ++n_out;
        }
        if (n_out) {
            LV<uint64_t> offv;
            FOR_LANES(l) {
                offv[l] = 0;
                if (l == 0) {
#if MGX_WAVE_EMU
                    offv[l] = *P.out_cursor; *P.out_cursor += words;
#else
                    offv[l] = atomicAdd(P.out_cursor, (unsigned long long)words);
#endif
                }
            }
            uint64_t so = wave_bcast(offv, 0);
            if (so + words > P.out_capacity) {
                rr.status = ST_CAPACITY;
            } else {
                uint32_t *dst = P.out_stream + so;
                int k_out = 0;
                for (int t = w.have_best - 1; t >= 0; --t) {
                    const DevAln &a = w.aln[Q0 + ord[t]];
                    if (!a.n_nodes) continue;
                    if (k_out == 0) {
                        rr.score = a.score; rr.offset = (uint32_t)a.offset;
                        rr.n_nodes = (uint32_t)a.n_nodes; rr.n_cigar = (uint32_t)a.n_cigar; rr.seq_len = (uint32_t)a.seq_len;
                        rr.orientation = (uint32_t)a.orientation; rr.stream_off = so;
                    } else {
                        FOR_LANES(l) {
                            if (l == 0) {
                                dst[0] = (uint32_t)a.score; dst[1] = (uint32_t)a.offset; dst[2] = (uint32_t)a.n_nodes;
                                dst[3] = (uint32_t)a.n_cigar; dst[4] = (uint32_t)a.seq_len; dst[5] = (uint32_t)a.orientation;
                            }
                        }
                        dst += 6;
                    }
                    uint8_t *dseq = (uint8_t *)(dst + a.n_nodes + a.n_cigar);
                    int32_t n = imax(imax(a.n_nodes, a.n_cigar), a.seq_len);
                    for (int32_t base = 0; base < n; base += WAVE) {
                        FOR_LANES(l) {
                            int32_t x = base + l;
                            if (x < a.n_nodes) dst[x] = a.nodes[x];
                            if (x < a.n_cigar) dst[a.n_nodes + x] = a.cigar[x];
                            if (x < a.seq_len) dseq[x] = a.seq[x];
                        }
                    }
                    dst += (uint32_t)a.n_nodes + (uint32_t)a.n_cigar + ((uint32_t)a.seq_len + 3) / 4;
                    ++k_out;
                }
                rr.n_alignments = n_out;
            }
        }
    }
    FOR_LANES(l) { if (l == 0) P.results[read] = rr; }
    stats_accum->rank_lines += w.ctr.rank_lines;
    stats_accum->select_lines += w.ctr.select_lines;
    stats_accum->bit_lines += w.ctr.bit_lines;
    stats_accum->columns += w.n_columns;
    stats_accum->fast_columns += w.n_fast_columns;
    stats_accum->extensions += w.n_extensions;
    if (PHASE & PH_SEED) stats_accum->seeds += (uint32_t)(w.n_seeds[0] + w.n_seeds[1]);
    stats_accum->capacity_errors += rr.status != ST_OK;
    w.cyc[5] = cycle_clock() - tout;
    for (int x = 0; x < 8; ++x) stats_accum->cyc[x] += w.cyc[x];
    for (int x = 0; x < XCYC_N; ++x) stats_accum->xcyc[x] += w.xcyc[x];
}


#ifndef MGX_NO_EXTEND
// ------------------------------------------------------------------------------------------------
// The flat group loop (round 3): every 8-lane group of a wavefront advances its OWN read.
//
// In the per-read program above, control flow is shared by the groups of a wavefront down to the loop level: a group whose
// extension ends waits at the loop exit until the longest extension among its wave-mates ends, again after the trace
// walk, again before the next read is fetched — a quarter of the extension kernel's time on the bench batch
// (profiles/r03_ab1.txt: 315 ms for distinct reads, 230 ms with every read replicated 8 x), most of it on seed-rich
// reads.  Here the read's program is a state machine and the wavefront runs ONE loop whose iteration gives every group
// one step of whatever it is doing: a step of its extension (extend_step), up to FLAT_BT_STEPS steps of its trace walk
// (bt_step), or a driver transition (flat_drive: everything between those — seed pick, strand flip, aggregator, filters —
// run to the next extension or backtrack), or the fetch of its next read.  Groups in different activities are serialised
// by EXEC masking within the iteration only; nobody waits for a wave-mate's loop to end.
//
// The driver restates aln_both / align_core_fwd / align_read's extension half for one alignment per seed
// (num_alternative_paths == 1; the alternative-path build keeps the per-read program for N > 1), with the same calls in
// the same order per read: results are identical by construction, and the host model runs both.
// ------------------------------------------------------------------------------------------------
enum { ACT_FETCH = 0, ACT_DRIVE = 1, ACT_EXTEND = 2, ACT_BT = 3, ACT_FINISH = 4, ACT_EXIT = 5 };
enum { DS_PHASE = 0, DS_SEED = 1, DS_FWD_EXT = 2, DS_FWD_BT = 3, DS_BWD_EXT = 4, DS_BWD_BT = 5, DS_SEED_DONE = 6, DS_END = 7 };
constexpr int32_t FLAT_BT_STEPS = 12;       // trace-walk steps per iteration (about the instructions of one chain step)

MGX_DEV int flat_es(const Wave &w) { return w.fs.ds == DS_BWD_EXT || w.fs.ds == DS_BWD_BT ? 1 - (int)w.fs.s : (int)w.fs.s; }

// what the read's current backtrack works on: forward = the seed itself, into aln[0]; backward = the reversed forward
// alignment aln[1], into aln[2]
MGX_DEV SeedRef flat_bt_seed(const Wave &w) {
    if (w.fs.ds == DS_FWD_BT) return seedref_from_seed(w, (int)w.fs.s, w.fs.i, nullptr);
    return seedref_from_aln(w.aln[1]);
}

// later seeds of the strand against the forward extender's convergence table / the deferred filter_nodes marks (aln_both)
MGX_DEV void flat_check_later(Wave &w, bool with_table, bool with_filter) {
    const int s = (int)w.fs.s;
    const int32_t n = w.n_seeds[s], i = w.fs.i;
    ExtenderState &F = w.ext[s];
    const DevAln &p2 = w.aln[2];
    const uint32_t *filt_nodes = p2.nodes;
    const int32_t filt_n = with_filter ? p2.n_nodes : 0;
    const int32_t filt_lo = aln_clipping(p2), filt_hi = w.L - aln_end_clipping(p2);
    for (int32_t base = i + 1; base < n; base += WAVE) {
        FOR_LANES(l) {
            const int32_t j = base + l;
            if (j < n && w.alive[s][j]) {
                DevSeed sj = w.seeds[s][j];
                uint32_t last_node = sj.offset == 0 ? w.nodes[s][sj.clipping + sj.n_nodes - 1] : sj.node;
                SeedRef rj = seedref_from_seed(w, s, j, nullptr);
                bool dead = with_table && !check_seed(w, F, last_node, rj.qlen, rj.clipping, rj.score);
                const int32_t pos = rj.qlen + rj.clipping - 1;
                if (with_filter && !dead && filt_n && pos >= filt_lo && pos < filt_hi) {
                    for (int32_t x = 0; x < filt_n; ++x) dead |= gld(filt_nodes + x) == last_node;
                }
                if (dead) w.alive[s][j] = 0;
            }
        }
    }
    wave_sync();
}

// Driver transitions until the read needs an extension (-> ACT_EXTEND, begun), a backtrack (-> ACT_BT, begun) or is over
// (-> ACT_FINISH).  aln_both (:657-736) / align_core (:360-384) / align_both_directions (:738-755), N = 1.
MGX_DEV void flat_drive(Wave &w) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    FlatState &f = w.fs;
    const bool canon = P.cfg.canonical != 0;
    const bool have_rc = P.cfg.fwd_and_rc != 0;
    for (;;) {
        int want = 0;                       // 1: begin an extension, 2: begin a backtrack
        int es = 0;
        bool force = false;
        int32_t mps = 0;
        SeedRef seed;
        seed.nodes = nullptr; seed.seq = nullptr; seed.n_nodes = seed.seq_len = 0;
        seed.clipping = seed.end_clipping = seed.qlen = seed.offset = seed.score = seed.orientation = 0;
        const int s = (int)f.s;
        if (f.ds == DS_PHASE) {
            // align_both_directions: the strand with more matches first, the other if it is within rel_score_cutoff
            const uint32_t fm = w.num_matching[0], bm = w.num_matching[1];
            const int first = fm >= bm ? 0 : 1;
            const uint32_t m_first = first ? bm : fm, m_second = first ? fm : bm;
            const int ph = (int)f.ph;
            if (w.status != ST_OK || ph >= (have_rc ? 2 : 1)
                    || (have_rc && ph == 1 && !((double)m_second >= (double)m_first * P.cfg.rel_score_cutoff))) { f.ds = DS_END; continue; }
            if (ph != w.resume_phase) w.resume_i = 0;
            w.resume_phase = ph;
            f.s = (uint8_t)(have_rc ? (ph == 0 ? first : 1 - first) : 0);
            w.ext[f.s].rc_view = 0;
            if (have_rc) w.ext[1 - f.s].rc_view = canon ? 0 : 1;
            f.i = w.resume_i;
            w.resume_i = 0;
            f.ds = DS_SEED;
            continue;
        }
        if (f.ds == DS_SEED) {
            const int32_t n = w.n_seeds[s];
            int32_t i = f.i;
            while (i < n && !w.alive[s][i]) ++i;
            f.i = i;
            if (i >= n) { f.ph = (uint8_t)(f.ph + 1); f.ds = DS_PHASE; continue; }
            if (pass_limit_reached(w, i)) { f.ds = DS_END; continue; }
            ++w.seeds_done;
            seed = seedref_from_seed(w, s, i, nullptr);
            conv_clear(w, w.ext[s].conv);                         // set_seed (:90-98)
            f.filt = 0;
            es = s; force = false; want = 1;
            f.ds = DS_FWD_EXT;
        } else if (f.ds == DS_FWD_EXT) {
            if (w.status != ST_OK) { f.ds = DS_END; continue; }
            seed = seedref_from_seed(w, s, f.i, nullptr);
            // aln_both: extend(): min_path_score = max(0, min_cell_score); align_core: get_min_path_score
            mps = have_rc ? imax(0, P.cfg.min_cell_score) : imax(0, min_path_score_now(w));
            es = s; want = 2;
            f.ds = DS_FWD_BT;
        } else if (f.ds == DS_FWD_BT) {
            if (w.status != ST_OK) { f.ds = DS_END; continue; }
            const int n_fwd = w.bt.produced;
            bool have_rev = false;
            if (!have_rc) {
                for (int e = 0; e < n_fwd; ++e) add_alignment(w, w.aln[e]);
                if (w.status != ST_OK) { f.ds = DS_END; continue; }
                flat_check_later(w, true, false);
                f.ds = DS_SEED_DONE;
                continue;
            }
            if (n_fwd) {
                DevAln &path = w.aln[0];
                DevAln &rev = w.aln[1];
                if (canon) {
                    const bool reversible = path.orientation && !path.offset;
                    const bool to_left = aln_clipping(path) && !path.offset;
                    const bool good = path.score >= min_path_score_now(w);
                    bool got = false;
                    if ((good && reversible) || to_left) { copy_aln(rev, path); got = reverse_complement_aln_stored(w, rev); }
                    if (good) { if (reversible) { if (got) add_alignment(w, rev); } else add_alignment(w, path); }
                    have_rev = to_left && got;
                } else {
                    if (path.score >= min_path_score_now(w)) add_alignment(w, path);
                    if (aln_clipping(path) && !path.offset) {
                        copy_aln(rev, path);
                        have_rev = reverse_complement_aln(w, rev);
                    }
                }
            }
            flat_check_later(w, true, false);                     // before the backward pass overwrites the column table
            if (!have_rev) { f.ds = DS_SEED_DONE; continue; }
            seed = seedref_from_aln(w.aln[1]);
            conv_clear(w, w.ext[1 - s].conv);
            es = 1 - s; force = true; want = 1;
            f.ds = DS_BWD_EXT;
        } else if (f.ds == DS_BWD_EXT) {
            if (w.status != ST_OK) { f.ds = DS_END; continue; }
            seed = seedref_from_aln(w.aln[1]);
            mps = imax(0, min_path_score_now(w));
            es = 1 - s; want = 2;
            f.ds = DS_BWD_BT;
        } else if (f.ds == DS_BWD_BT) {
            if (w.status != ST_OK) { f.ds = DS_END; continue; }
            const int n_bwd = w.bt.produced;
            if (n_bwd) {
                DevAln &p2 = w.aln[2];
                if (canon && !(p2.orientation && !p2.offset)) {
                    add_alignment(w, p2);                          // not reversible: as it is (:711)
                } else if (canon ? reverse_complement_aln_stored(w, p2) : reverse_complement_aln(w, p2)) {
                    f.filt = 1;                                    // filter_nodes, applied lazily (flat_check_later)
                    add_alignment(w, p2);
                }
            }
            if (w.status != ST_OK) { f.ds = DS_END; continue; }
            f.ds = DS_SEED_DONE;
            continue;
        } else if (f.ds == DS_SEED_DONE) {
            if (f.filt) flat_check_later(w, false, true);
            f.filt = 0;
            f.i = f.i + 1;
            f.ds = DS_SEED;
            continue;
        } else {                                                   // DS_END
            f.act = ACT_FINISH;
            return;
        }
        if (want == 1) {
            if (!extend_begin(w, es, seed, force)) continue;      // capacity: the *_EXT state sees the status
            f.act = ACT_EXTEND;
            return;
        }
        bt_begin(w, es, seed, mps, 1);
        f.act = ACT_BT;
        return;
    }
}

// align_read's part before the driver, for the extension kernel: workspace, query, the seeds of the seeding kernel.
// false = the read is over already (status says why)
MGX_DEV bool flat_read_begin(Wave &w, const AlignParams &P, uint64_t read, uint32_t slot, uint8_t *lds, uint32_t lds_bytes,
                               const uint8_t *resume_rec) {
    MGX_ASSUME_LDS(&w);
#if !(defined(MGX_PARAMS_IN_LDS) && MGX_PARAMS_IN_LDS)
    w.P = &P;
#endif
    carve(w, P, P.arena + (uint64_t)slot * P.arena_stride, lds, lds_bytes);
    w.sd = w.sd_own;
    const uint64_t off = P.offsets[read];
    w.L = (int32_t)(P.offsets[read + 1] - off);
    w.status = ST_OK;
    w.have_best = 0;
    w.seeds_done = 0;
    w.resume_phase = 0; w.resume_i = 0; w.no_limit = 0;
    w.ctr.rank_lines = w.ctr.select_lines = w.ctr.bit_lines = 0;
    w.n_columns = w.n_extensions = w.n_fast_columns = 0;
    const uint64_t nb = P.node_begin[read];
    w.n_kmers = (int32_t)(P.node_begin[read + 1] - nb);
    w.nodes[0] = P.nodes_fwd + nb;
    w.nodes[1] = P.nodes_rc + nb;
    for (int x = 0; x < 8; ++x) w.cyc[x] = 0;
    for (int x = 0; x < XCYC_N; ++x) w.xcyc[x] = 0;
    const uint64_t tstart = cycle_clock();
    w.n_seeds[0] = w.n_seeds[1] = 0; w.num_matching[0] = w.num_matching[1] = 0;
    if (w.L > (int32_t)P.lim.Lmax) { w.status = ST_CAPACITY; return false; }
    prepare_query(w, P.seqs + off, false, true);
    w.cyc[0] = (uint32_t)(cycle_clock() - tstart);
    for (int s = 0; s < 2; ++s) {
        w.ext[s].q = w.q[s];
        w.ext[s].psum = w.psum[s];
        w.ext[s].psum_lin = w.psum_lin[s];
        w.ext[s].table_cap = 0;
        w.ext[s].rc_view = 0;
        w.ext[s].conv.n_entries = 0; w.ext[s].conv.n_recs = 0; w.ext[s].conv.pool_top = 0;
        w.ext[s].conv.cap = conv_cap0(P.lim.hash_size); w.ext[s].conv.base = 0; w.ext[s].conv.start = 0;
        w.ext[s].conv.gen = w.gen_store[2 * s]; w.ext[s].conv.dirty = w.gen_store[2 * s + 1];
    }
    const uint64_t tseed = cycle_clock();
    const SeedHdr h = P.seed_hdr[read];
    w.status = h.status;
    for (int s = 0; s < 2; ++s) { w.n_seeds[s] = h.n_seeds[s]; w.num_matching[s] = h.num_matching[s]; }
    if (w.status == ST_OK) {
        const DevSeed *src = P.seed_stream + h.off;
        for (int s = 0; s < 2; ++s) {
            const int32_t n = w.n_seeds[s];
            for (int32_t base = 0; base < n; base += WAVE) {
                FOR_LANES(l) {
                    int32_t x = base + l;
                    if (x < n) { w.seeds[s][x] = src[x]; w.alive[s][x] = 1; }
                }
            }
            src += n;
        }
    }
    wave_sync();
    if (MGX_ABLATED(w, 8u)) { w.n_seeds[0] = w.n_seeds[1] = 0; }          // timing probe: fetch + pick-up + output only
    if (P.dbg_seeds && w.status == ST_OK) {
        for (int s = 0; s < 2; ++s)
            for (int32_t i = 0; i < w.n_seeds[s]; ++i)
                P.dbg_seeds[((uint64_t)read * 2 + s) * P.lim.max_seeds + i] = w.seeds[s][i];
    }
    wave_sync();
    w.cyc[1] = (uint32_t)(cycle_clock() - tseed);
    if (w.status != ST_OK) return false;
    if (resume_rec) resume_load(w, resume_rec);          // a later pass: the aggregator, the live seeds, where it stopped
    w.fs.ph = (uint8_t)w.resume_phase; w.fs.s = 0; w.fs.i = 0; w.fs.filt = 0; w.fs.ds = DS_PHASE;
    return true;
}

// align_read's part after the driver: retry bookkeeping of the multi-pass extension, the result record, the output stream.
// true = the driver goes on with this read (no room for a resume record: the pass limit is lifted)
MGX_DEV bool flat_read_end(Wave &w, const AlignParams &P, uint64_t read, KernelStats *stats_accum) {
    MGX_ASSUME_LDS(&w);
    int64_t retry_pos = -1;
    if (w.status == ST_RETRY && P.resume_out) {
        // stopped by the pass's seed limit: take a retry position; without room for a record the read just goes on
        LV<uint64_t> pv;
        FOR_LANES(l) {
            pv[l] = 0;
            if (l == 0) {
#if MGX_WAVE_EMU
                pv[l] = (*P.retry_count)++;
#else
                pv[l] = atomicAdd(P.retry_count, 1ull);
#endif
            }
        }
        const uint64_t pos = wave_bcast(pv, 0);
        if (pos >= P.resume_cap) {
            w.no_limit = 1; w.status = ST_OK;
            w.fs.ph = (uint8_t)w.resume_phase; w.fs.ds = DS_PHASE;
            return true;
        }
        retry_pos = (int64_t)pos;
        resume_save(w, P.resume_out + pos * P.resume_rec_bytes);
        // work key of the next pass: this read's next live seed (same prediction as the seeding kernel's)
        uint32_t key = 1;
        {
            const bool have_rc = P.cfg.fwd_and_rc != 0;
            const int sidx = w.resume_phase == 0 ? (w.num_matching[0] >= w.num_matching[1] ? 0 : 1)
                                                 : (w.num_matching[0] >= w.num_matching[1] ? 1 : 0);
            const int ss = have_rc ? sidx : 0;
            if (w.resume_i < w.n_seeds[ss]) {
                const int32_t clip = w.seeds[ss][w.resume_i].clipping;
                key = 1u + (uint32_t)imin(4094, (w.L - clip) + (clip > 0 ? w.L : 0));
            }
        }
        FOR_LANES(l) { if (l == 0) { P.retry_list[pos] = (uint32_t)read; P.retry_key[pos] = key; } }
    }
    if (w.L <= (int32_t)P.lim.Lmax) {
        for (int s = 0; s < 2; ++s) { w.gen_store[2 * s] = w.ext[s].conv.gen; w.gen_store[2 * s + 1] = w.ext[s].conv.dirty; }
        wave_sync();
    }
    if (w.status == ST_RETRY) {
        // this read goes on to another seed in the next pass: with a resume record (written above) or, without records,
        // untouched and from scratch
        FOR_LANES(l) {
            if (l == 0 && retry_pos < 0) {
#if MGX_WAVE_EMU
                P.retry_list[(*P.retry_count)++] = (uint32_t)read;
#else
                P.retry_list[atomicAdd(P.retry_count, 1ull)] = (uint32_t)read;
#endif
            }
        }
        stats_accum->rank_lines += w.ctr.rank_lines;
        stats_accum->select_lines += w.ctr.select_lines;
        stats_accum->bit_lines += w.ctr.bit_lines;
        for (int x = 0; x < 8; ++x) stats_accum->cyc[x] += w.cyc[x];
        for (int x = 0; x < XCYC_N; ++x) stats_accum->xcyc[x] += w.xcyc[x];
        return false;
    }
    const uint64_t tout = cycle_clock();
    ReadResult rr;
    rr.status = w.status; rr.n_alignments = 0; rr.score = 0; rr.offset = 0; rr.n_nodes = rr.n_cigar = rr.seq_len = 0;
    rr.orientation = (w.status == ST_CAPACITY && w.have_best < 0) ? RR_CAUSE_QUEUE : 0; rr.stream_off = 0;
    rr.num_matches_fwd = w.num_matching[0]; rr.num_matches_rc = w.num_matching[1];
    rr.n_seeds_fwd = (uint32_t)w.n_seeds[0]; rr.n_seeds_rc = (uint32_t)w.n_seeds[1];
    rr.n_extensions = w.n_extensions; rr.n_columns = w.n_columns;
    if (w.status == ST_OK && w.have_best) {
        // get_alignments (aligner_aggregator.hpp:180-202), one alignment: emitted unless empty
        const DevAln &a = w.aln[Q0];
        if (a.n_nodes) {
            const uint32_t words = (uint32_t)a.n_nodes + (uint32_t)a.n_cigar + ((uint32_t)a.seq_len + 3) / 4;
            LV<uint64_t> offv;
            FOR_LANES(l) {
                offv[l] = 0;
                if (l == 0) {
#if MGX_WAVE_EMU
                    offv[l] = *P.out_cursor; *P.out_cursor += words;
#else
                    offv[l] = atomicAdd(P.out_cursor, (unsigned long long)words);
#endif
                }
            }
            const uint64_t so = wave_bcast(offv, 0);
            if (so + words > P.out_capacity) {
                rr.status = ST_CAPACITY;
            } else {
                uint32_t *dst = P.out_stream + so;
                rr.score = a.score; rr.offset = (uint32_t)a.offset;
                rr.n_nodes = (uint32_t)a.n_nodes; rr.n_cigar = (uint32_t)a.n_cigar; rr.seq_len = (uint32_t)a.seq_len;
                rr.orientation = (uint32_t)a.orientation; rr.stream_off = so;
                uint8_t *dseq = (uint8_t *)(dst + a.n_nodes + a.n_cigar);
                const int32_t n = imax(imax(a.n_nodes, a.n_cigar), a.seq_len);
                for (int32_t base = 0; base < n; base += WAVE) {
                    FOR_LANES(l) {
                        int32_t x = base + l;
                        if (x < a.n_nodes) dst[x] = a.nodes[x];
                        if (x < a.n_cigar) dst[a.n_nodes + x] = a.cigar[x];
                        if (x < a.seq_len) dseq[x] = a.seq[x];
                    }
                }
                rr.n_alignments = 1;
            }
        }
    }
    FOR_LANES(l) { if (l == 0) P.results[read] = rr; }
    stats_accum->rank_lines += w.ctr.rank_lines;
    stats_accum->select_lines += w.ctr.select_lines;
    stats_accum->bit_lines += w.ctr.bit_lines;
    stats_accum->columns += w.n_columns;
    stats_accum->fast_columns += w.n_fast_columns;
    stats_accum->extensions += w.n_extensions;
    stats_accum->capacity_errors += rr.status != ST_OK;
    w.cyc[5] = (uint32_t)(cycle_clock() - tout);
    for (int x = 0; x < 8; ++x) stats_accum->cyc[x] += w.cyc[x];
    for (int x = 0; x < XCYC_N; ++x) stats_accum->xcyc[x] += w.xcyc[x];
    return false;
}

// the read's current backtrack, advanced by one batch of walk steps (its own function: the loop around it keeps the chain
// registers of the extension live)
MGX_DEV bool flat_bt_step(Wave &w) {
    MGX_ASSUME_LDS(&w);
    const SeedRef seed = flat_bt_seed(w);
    const bool fwd = w.fs.ds == DS_FWD_BT;
    return bt_step(w, seed, fwd ? nullptr : &w.aln[1], fwd ? &w.aln[0] : &w.aln[2], FLAT_BT_STEPS);
}

#if !MGX_WAVE_EMU
// ---- the kernel's loop: rounds of [service] -> [extension loop] -> [service] -> [trace loop] ----
// Measured (profiles/r03_ab1.txt) the groups of a wavefront lose most at the joins BETWEEN phases, waiting for wave-mates
// whose read has another extension to run.  Giving every group one step per iteration of a single loop would remove all
// waiting, but then a trace step (a twelfth of a chain step) costs the wavefront a whole iteration whenever fewer than all
// groups are walking, and calls inside the step loop wreck its register allocation (1719 spilled VGPRs against 18).  So the
// phases themselves stay lock-step — every group runs ONE extension in the extension loop, ONE backtrack in the trace
// loop — and flat_service() does everything in between per group: a group whose read is finished fetches its next read
// there and joins the next extension loop with it, instead of idling through its wave-mates' second extensions.
MGX_NI_G4 void flat_service(Wave &w, uint32_t slot, KernelStats *stats_accum, uint8_t *lds, uint32_t lds_bytes, uint64_t n_items) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    FlatState &f = w.fs;
    for (;;) {
        if (f.act == ACT_FETCH) {
            LV<uint64_t> rv;
            rv.v = 0;
            if (lane_id() == 0) rv.v = atomicAdd(P.read_cursor, 1ull);
            const uint64_t item = wave_bcast(rv, 0);
            if (item >= n_items) { f.act = ACT_EXIT; break; }
            uint64_t read = P.order ? P.order[item] : item;
            const uint8_t *rec = nullptr;
            if (P.resume_in) {                                   // a later pass: `read` is a retry position of the pass before
                rec = P.resume_in + read * P.resume_rec_bytes;
                read = P.resume_reads[read];
            }
            f.read = read;
            f.act = flat_read_begin(w, P, read, slot, lds, lds_bytes, rec) ? ACT_DRIVE : ACT_FINISH;
        }
        if (f.act == ACT_DRIVE) {
            const uint64_t t0 = cycle_clock();
            flat_drive(w);
            w.cyc[4] += (uint32_t)(cycle_clock() - t0);
        }
        if (f.act == ACT_FINISH) f.act = flat_read_end(w, P, f.read, stats_accum) ? ACT_DRIVE : ACT_FETCH;
        if (f.act == ACT_EXTEND || f.act == ACT_BT) break;
    }
}
// the read's backtrack, to the end
MGX_NI_G4 void flat_bt_all(Wave &w) {
    MGX_ASSUME_LDS(&w);
    const uint64_t t0 = cycle_clock();
    const SeedRef seed = flat_bt_seed(w);
    const bool fwd = w.fs.ds == DS_FWD_BT;
    while (!bt_step(w, seed, fwd ? nullptr : &w.aln[1], fwd ? &w.aln[0] : &w.aln[2], INT32_MAX)) {}
    w.fs.act = ACT_DRIVE;
    w.cyc[3] += (uint32_t)(cycle_clock() - t0);
}
// the read's extension, to the end (its own function, like extend(): with calls in the same function the step loop got 1600 spilled VGPRs instead of 28)
MGX_NI_G3 void flat_extend_all(Wave &w) {
    MGX_ASSUME_LDS(&w);
    const uint64_t t0 = cycle_clock();
    const int es = flat_es(w);
    ChainRegs R;
    chain_regs_reset(R);
    while (extend_step(w, es, R) == XS_MORE) {}
    w.fs.act = ACT_DRIVE;
    w.cyc[2] += (uint32_t)(cycle_clock() - t0);
}
#endif

// One iteration of the group's loop: w.fs.act says what the group is doing.  `fetch(read, rec)` hands out the next read
// (false: none left).  R: the extension's chain registers, `read`: the group's current read — both loop-carried by the
// caller.
template <class Fetch>
MGX_DEV void flat_iteration(Wave &w, const AlignParams &P, uint32_t slot, KernelStats *stats_accum, uint8_t *lds, uint32_t lds_bytes,
                            ChainRegs &R, uint64_t &read, Fetch fetch) {
    FlatState &f = w.fs;
    if (f.act == ACT_FETCH) {
        const uint8_t *rec = nullptr;
        if (!fetch(read, rec)) { f.act = ACT_EXIT; return; }
        f.act = flat_read_begin(w, P, read, slot, lds, lds_bytes, rec) ? ACT_DRIVE : ACT_FINISH;
    }
    if (f.act == ACT_DRIVE) {
        const uint64_t t0 = cycle_clock();
        flat_drive(w);
        w.cyc[4] += (uint32_t)(cycle_clock() - t0);
        if (f.act == ACT_EXTEND) chain_regs_reset(R);
    }
    if (f.act == ACT_EXTEND) {
        const uint64_t t0 = cycle_clock();
        if (extend_step(w, flat_es(w), R) != XS_MORE) f.act = ACT_DRIVE;
        w.cyc[2] += (uint32_t)(cycle_clock() - t0);
    } else if (f.act == ACT_BT) {
        const uint64_t t0 = cycle_clock();
        if (flat_bt_step(w)) f.act = ACT_DRIVE;
        w.cyc[3] += (uint32_t)(cycle_clock() - t0);
    }
    if (f.act == ACT_FINISH) f.act = flat_read_end(w, P, read, stats_accum) ? ACT_DRIVE : ACT_FETCH;
}
#endif  // MGX_NO_EXTEND

} // namespace mgx
