// align_types.hpp — plain-data structures shared by the host launcher and the kernels.
#pragma once
#include "dev_graph.hpp"

namespace mgx {

constexpr int32_t NINF = INT32_MIN + 100;            // DBGAlignerConfig::ninf (aligner_config.hpp:31)
constexpr uint32_t INF_LEN = 0xFFFFFFFFu;            // "unbounded" size_t config values

enum { OP_CLIPPED = 0, OP_MISMATCH = 1, OP_MATCH = 2, OP_DELETION = 3, OP_INSERTION = 4, OP_NODE_INSERTION = 5 };

enum { ST_OK = 0, ST_CAPACITY = -5, ST_RETRY = -100 };      // ST_RETRY is internal to the two-pass extension (never reported)
// ReadResult::orientation of a record with status ST_CAPACITY: why.  0 = a per-read arena (larger limits cure it: the host's
// capacity retry), RR_CAUSE_QUEUE = more alignments than the post_chain_alignments queue holds (no limit cures it).
enum { RR_CAUSE_QUEUE = 2 };

// DBGAlignerConfig after the DBGAligner ctor clamps (dbg_aligner.cpp:33-61), narrowed for the device
struct DevConfig {
    uint32_t min_seed_length, max_seed_length;       // INF_LEN = unbounded
    uint32_t max_num_seeds_per_locus;                // INF_LEN = unbounded
    int32_t min_cell_score, min_path_score, xdrop;
    double min_exact_match, max_nodes_per_seq_char, max_ram_per_alignment, rel_score_cutoff;
    int32_t gap_open, gap_ext, left_end_bonus, right_end_bonus;
    uint32_t fwd_and_rc, allow_left_trim, seed_complexity_filter;
    uint32_t num_alt;        // num_alternative_paths
    uint32_t agg_cap;        // alignments the aggregator of a query can hold: num_alt, or — post_chain — MGX_MAX_ALTERNATIVE_PATHS
    uint32_t post_chain;     // post_chain_alignments: the aggregator never drops an alignment (aligner_aggregator.hpp:88-96)
    uint32_t canonical;      // 1: the graph is a CANONICAL-mode DBGSuccinct (both strands stored): dbg_aligner.cpp:225,644-655;
                             // 2: a PRIMARY-mode one seen through the CanonicalDBG wrapper (canon_graph.hpp), same driver flow;
                             // 3: the same with the wrapper's look-ups read from the graph's reverse-complement tables
};

struct DevLimits {
    uint32_t Lmax;           // longest query in the batch
    uint32_t max_columns;    // DP columns per extension
    uint32_t max_seeds;      // seeds per strand
    uint32_t max_path;       // nodes / cigar runs / path characters per alignment
    uint32_t max_alt;        // alternative sub-k nodes kept per read and strand
    uint32_t cell_words;     // int32 words of S/E/F storage per wave
    uint32_t hash_size;      // power of two
    uint32_t n_aln;          // alignment buffers per read: 4 * num_alternative_paths
    uint32_t conv_pool_words;    // int32 words of convergence vectors per extender (pool; see the convergence checker section of align_core.hpp)
    // label-aware alignment only (0 otherwise; see the "label sets" section of align_core.hpp)
    uint32_t lab_words;          // uint32 words of the per-read label-set arena
    uint32_t lab_ext;            // alignments one backtracking may report (one per label subset of its seed)
    uint32_t lab_pool;           // alignments the per-label aggregator may hold at a time
    uint32_t lab_queues;         // labels with alignments the aggregator may hold queues for (and labels on a read's seeds)
};

// per-read result header; variable-length parts live in the output stream
struct ReadResult {
    int32_t status;          // ST_OK / ST_CAPACITY
    int32_t n_alignments;    // 0 .. num_alternative_paths; the scalars below describe alignment 0
    int32_t score;
    uint32_t offset;
    uint32_t n_nodes, n_cigar, seq_len;
    uint32_t orientation;
    uint64_t stream_off;     // word offset into the output stream: nodes[n_nodes] (u32),
                             // cigar[n_cigar] (u32: len << 3 | op), seq bytes (padded to words); further alignments follow,
                             // each as 6 words (score, offset, n_nodes, n_cigar, seq_len, orientation) + the same arrays
    // intermediate products for parity tests
    uint32_t num_matches_fwd, num_matches_rc, n_seeds_fwd, n_seeds_rc;
    uint32_t n_extensions, n_columns;
};

struct DevSeed {             // Seed (alignment.hpp:32-98); full seeds reference the strand's node array
    uint16_t clipping, length, offset, n_nodes;
    uint32_t node;           // the single node of a sub-k seed
};

// what the seeding kernel hands to the extension kernel (split pipeline): the seeds live in seed_stream
struct SeedHdr {
    uint64_t off;            // first seed of strand 0 in the seed stream; strand 1 follows
    uint32_t num_matching[2];
    uint16_t n_seeds[2];
    int32_t status;
    uint32_t pad;
};

enum { PH_SEED = 1, PH_EXTEND = 2, PH_BOTH = 3 };

struct KernelStats {
    unsigned long long rank_lines, select_lines, bit_lines, columns, extensions, seeds, capacity_errors, map_lines;
    unsigned long long seed_lines;  // part of the three line counters issued by the seeding phase (split pipeline)
    unsigned long long xcyc[8];     // extend() breakdown
    unsigned long long cyc[8];      // shader cycles per phase: prepare, seeding, extend, backtrack, driver rest, output
    unsigned long long fast_columns; // DP columns computed by the register-resident chain path (part of `columns`)
    unsigned long long lane_lines;   // part of the line counters issued by the lane-per-read kernel
    unsigned long long lane_columns; // part of `columns` computed by the lane-per-read kernel (finished and passed-on reads)
};

struct AlignParams {
    DevGraph g;
    DevConfig cfg;
    DevLimits lim;
    const int8_t *score_matrix;          // 128 x 128, [graph char][query char]
    const char *seqs;
    const uint64_t *offsets;             // n_reads + 1
    const uint64_t *node_begin;          // n_reads + 1, k-mer slots per read
    const uint32_t *nodes_fwd, *nodes_rc;
    const uint8_t *mlen_fwd, *mlen_rc;   // optional: per k-mer position, what index() matched (graph_build.hpp MLEN_*)
    const uint2 *rng_fwd, *rng_rc;       // optional: (rl, ru) of that match where it has >= min_seed_length characters
    uint64_t n_reads;
    uint8_t *arena;                      // per-wave workspace
    uint64_t arena_stride;
    ReadResult *results;
    uint32_t *out_stream;
    uint64_t out_capacity;               // words
    unsigned long long *out_cursor;
    unsigned long long *read_cursor;
    KernelStats *stats;
    DevSeed *dbg_seeds;                  // optional: seeds dump [n_reads][2][max_seeds]
    // split pipeline (seeding kernel -> sort by predicted extension work -> extension kernel)
    SeedHdr *seed_hdr;                   // [n_reads]
    DevSeed *seed_stream;
    uint64_t seed_capacity;              // seeds
    unsigned long long *seed_cursor;
    uint32_t *work_key;                  // [n_reads], written by the seeding phase
    const uint32_t *order;               // optional: the extension phase processes read order[i] as its i-th item
    const uint32_t *seed_list;           // optional: the seeding phase processes read seed_list[i] as its i-th item (what the
                                         // lane-per-read seeder left: seed_lane.hpp; the item count then comes from *n_items_ptr)
    // two-pass extension: pass 1 stops a read that would extend a second seed (seed_limit = 1) and lists it in
    // retry_list; pass 2 re-runs the listed reads from scratch without a limit (order = retry_list, item count read
    // from *n_items_ptr on the device).  Keeps the rare multi-seed reads from stalling the 7 other reads of their
    // wavefront.
    uint32_t seed_limit;                 // 0 = unlimited
    uint32_t *retry_list;
    unsigned long long *retry_count;
    const unsigned long long *n_items_ptr;   // null: n_reads items
    // Multi-pass extension with resume records: a pass extends at most seed_limit seeds per read; a read with live seeds
    // left writes what outlives a seed — aggregator queue, live-seed flags, extender capacities, counters, where it stopped
    // — into resume_out[its retry position] and a work key for its next seed into retry_key; the next pass sorts the
    // positions by key and resumes them (order[] then holds retry positions: read = resume_reads[pos], record =
    // resume_in + pos * resume_rec_bytes).  Convergence tables never outlive a seed, so nothing else crosses passes.
    // resume_out == null keeps the from-scratch behaviour above.  A read whose position is >= resume_cap simply goes on
    // without a limit in its current pass.
    const uint8_t *resume_in;
    const uint32_t *resume_reads;
    uint8_t *resume_out;
    uint32_t *retry_key;
    uint32_t resume_rec_bytes, resume_cap;
    uint64_t n_items;                        // items of this launch (0: n_items_ptr / n_reads)
    uint32_t no_fast;                    // A/B and test switch: every column through the general (staging buffer) path
    uint32_t no_flat;                    // A/B and test switch: the per-read program instead of the flat group loop
    uint32_t no_alias;                   // A/B and test switch: every convergence-table entry gets its own vector in the pool
    uint32_t no_compact;                 // A/B and test switch: chain columns always in the two-line form (ColSlot)
    uint32_t no_bt_runs;                 // A/B and test switch: the trace walk one step at a time (no lane-parallel diagonal runs)
    uint32_t groups_per_wave;            // extension kernel: groups of a wavefront that take reads (0 = all).  A batch with fewer reads
                                         // than resident groups is spread over the wavefronts, so that a long read does not run in
                                         // lock-step with seven others
    // label-aware alignment (LabeledAligner, A/aligner_labeled.{hpp,cpp}): the row-major label matrix of mgx_annot.hip —
    // row = node - 1 (AnnotatedDBG::graph_to_anno_index); head word: count:16 | single label or offset into more[]
    uint32_t labeled;                    // bit 0: label-aware (kernels built with MGX_WITH_LABELS only); bit 1: no row of a dummy node
                                         // (W == 0) holds a label, so the "skip dummy nodes" test of the reference is the row itself
    const uint64_t *anno_head;
    const uint32_t *anno_count;
    const uint32_t *anno_more;
    uint64_t anno_rows;
    const uint32_t *anno_base;           // CANONICAL-mode graphs: node -> the representative whose row holds its labels (canon_repr_node), else null
    uint32_t ablate;                     // timing probes only (results become WRONG): bit 0 = no convergence table in the chain
                                         // step, bit 1 = no cell records / column metadata stores, bit 2 = no backtrack
    // the strands as k_pack_reads left them (k <= 32: 32 codes per 64-bit word + one invalid flag per base, word j of read r at
    // packed_word_begin(offsets[r], r) + j), or null: the seeding phase takes its 2-bit strands from here instead of encoding again
    const uint64_t *pkw[2];
    const uint32_t *ivw[2];
};

} // namespace mgx
