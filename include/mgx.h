/*
 * mgx.h — C-ABI of the MI355X-native sequence-to-graph aligner (libmgx.so).
 *
 * This is the drop-in boundary for MetaGraph's `DBGAligner` hot path.  Every entry point
 * cites the reference interface it stands in for (paths are into ratschlab/metagraph,
 * `metagraph/src/...`).  Plain C types only: pointers + sizes, no C++/torch types.
 *
 * Ownership: inputs are caller-owned and only read during the call.  Results live in a
 * library-owned arena attached to the aligner handle and stay valid until the next
 * mgx_align_batch() on the same handle or mgx_aligner_destroy().
 * Threading: one in-flight batch per aligner handle; several handles may share one graph
 * (graph is immutable after creation), mirroring `DBGAligner` instances sharing a
 * `const DeBruijnGraph&` (graph/alignment/dbg_aligner.hpp:63-64).
 * Errors: every function returns MGX_OK (0) or a negative code; mgx_last_error() gives text.
 * There is NO CPU fallback: without a HIP device every compute entry point fails with
 * MGX_ERR_NO_DEVICE.
 */
#ifndef MGX_H_
#define MGX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_ABI_VERSION 6      /* 6 (round 6, late): mgx_gather_* (RCCL gather of the device results); 5 (round 6): mgx_stats::n_seed_lane_reads / seed_lane_ms / seed_lane_left_reads, streams, coordinates,
                                * mgx_chain_seeds; 4 (round 5): mgx_stats::n_capacity_retried, mgx_chain_alignments, post_chain_alignments accepted;
                                * 3 (round 4): mgx_alignment::n_labels / labels_begin, mgx_results::labels (label-aware alignment),
                                * mgx_stats::extend_kernels / n_lane_reads / lane_ms, "key=value" options of mgx_aligner_set_pipeline;
                                * 2 (round 3): mgx_annotation_*; mgx_stats / mgx_config grew in round 2 without a bump (callers built
                                * against 1 must be rebuilt: mgx_aligner_stats writes the larger struct) */

enum {
    MGX_OK = 0,
    MGX_ERR_INVALID = -1,      /* bad argument / malformed BOSS view */
    MGX_ERR_NO_DEVICE = -2,    /* no HIP device / HIP runtime failure */
    MGX_ERR_UNSUPPORTED = -3,  /* config outside the implemented path */
    MGX_ERR_CONFIG = -4,       /* check_config_scores() failed (aligner_config.cpp:39-66);
                                  the reference throws std::runtime_error (dbg_aligner.cpp:55-56) */
    MGX_ERR_CAPACITY = -5,     /* a per-read device arena overflowed; see mgx_limits */
    MGX_ERR_OOM = -6
};

/* Graph modes (graph/representation/base/sequence_graph.hpp:160). */
enum { MGX_MODE_BASIC = 0, MGX_MODE_CANONICAL = 1, MGX_MODE_PRIMARY = 2 };

/* CIGAR operators, numeric values identical to Cigar::Operator
 * (graph/alignment/aligner_cigar.hpp:18-25); printed with "SX=DIG" (:107). */
enum { MGX_OP_CLIPPED = 0, MGX_OP_MISMATCH = 1, MGX_OP_MATCH = 2,
       MGX_OP_DELETION = 3, MGX_OP_INSERTION = 4, MGX_OP_NODE_INSERTION = 5 };

/*
 * A read-only view of a BOSS table — what `boss::BOSS` holds after load
 * (graph/representation/succinct/boss.hpp:28-631; fields W_, last_, F_, k_).
 * Edge indices are 1..n_edges; slot 0 of W/last is the unused sentinel slot
 * (boss.cpp:74-83).  A DBG node IS a BOSS edge index (dbg_succinct.cpp:43-45).
 */
typedef struct mgx_boss_view {
    uint32_t k;              /* DBG k-mer length = BOSS node length + 1 (dbg_succinct.cpp:44) */
    uint32_t sigma;          /* alphabet size incl. '$'; must be 5 ("$ACGT", kmer/alphabets.hpp:64) */
    uint64_t n_edges;        /* W/last have n_edges+1 entries */
    const uint8_t *W;        /* one byte per edge, values 0..2*sigma-1 (boss.cpp:62) */
    const uint8_t *last;     /* one byte per edge, 0/1 */
    const uint64_t *F;       /* sigma entries (boss.hpp:506-510) */
    const uint8_t *valid;    /* optional node mask, one byte per edge (dbg_succinct.cpp:934-936);
                                NULL = every edge is a node, as after reset_mask() (cli/align.cpp:337-339) */
    uint32_t mode;           /* MGX_MODE_* = DeBruijnGraph::get_mode() of the DBGSuccinct.  CANONICAL (both strands stored): the
                              * aligner always seeds both strands, runs the backward pass on the same graph and reports
                              * reverse-strand alignments as the forward alignments they mirror (dbg_aligner.cpp:225,644-655).
                              * PRIMARY (one k-mer of every pair stored): aligned through the CanonicalDBG wrapper as the reference
                              * does (dbg_aligner.cpp:52-53, canonical_dbg.cpp): node ids above n_edges are reverse complements
                              * (id - n_edges is the stored node), mgx_graph_max_index reports 2 * n_edges.  PRIMARY needs
                              * k <= 64 and 2 * n_edges < 2^32 (MGX_ERR_UNSUPPORTED otherwise). */
    uint32_t on_device;      /* 0: W/last/valid are host pointers; 1: device pointers (F always host) */
} mgx_boss_view;

/* Field-for-field mirror of DBGAlignerConfig (graph/alignment/aligner_config.hpp:18-94): same members in the same order
 * (explicit padding where the C++ struct has implicit padding), so a reference-side adapter copies member-wise. */
typedef struct mgx_config {
    uint64_t num_alternative_paths;    /* :23 */
    uint64_t min_seed_length;          /* :24  (0 -> k, dbg_aligner.cpp:37-38) */
    uint64_t max_seed_length;          /* :25  (0 -> k, dbg_aligner.cpp:43-44) */
    uint64_t max_num_seeds_per_locus;  /* :26 */
    int32_t min_cell_score;            /* :34 */
    int32_t min_path_score;            /* :35 */
    int32_t xdrop;                     /* :36 */
    int32_t _pad0;
    double min_exact_match;            /* :38 */
    double max_nodes_per_seq_char;     /* :39 */
    double max_ram_per_alignment;      /* :40 */
    double rel_score_cutoff;           /* :41 */
    int8_t gap_opening_penalty;        /* :43 */
    int8_t gap_extension_penalty;      /* :44 */
    int8_t left_end_bonus;             /* :45 */
    int8_t right_end_bonus;            /* :46 */
    uint8_t forward_and_reverse_complement; /* :48 */
    uint8_t chain_alignments;          /* :49 (must be 0.  The reference never lets a caller set it either: it exits without coordinates,
                                        * dbg_aligner.cpp:546-550, and LabeledAligner switches it on itself for an annotation with
                                        * coordinates, aligner_labeled.cpp:457-462 — such annotations are refused at
                                        * mgx_labeled_aligner_create.  Of seed chaining, aligner_chainer.cpp:47-542, the device has the
                                        * chainer's DP, mgx_chain_seeds, and k-mer coordinates, mgx_annotation_set_coordinates; the
                                        * extension between chained seeds is not built: DESIGN 3.9) */
    uint8_t post_chain_alignments;     /* :50: chain_alignments (aligner_chainer.cpp:555-720) over every query's alignments.  The
                                        * device then keeps EVERY alignment of a query (aligner_aggregator.hpp:88-96; at most
                                        * 4 x MGX_MAX_ALTERNATIVE_PATHS - 3 x num_alternative_paths = 13 with the default of
                                        * one alternative path — a query with more has status MGX_ERR_CAPACITY) and the
                                        * chaining runs on the host inside mgx_fetch_results / mgx_align_batch; results left
                                        * on the device (mgx_device_results) are the un-chained ones: decode them and call
                                        * mgx_chain_alignments.  Not with an annotation. */
    uint8_t global_xdrop;              /* :51 (must be 1) */
    uint8_t allow_left_trim;           /* :52 */
    uint8_t no_backtrack;              /* :53 (must be 0) */
    uint8_t seed_complexity_filter;    /* :54 */
    uint8_t alignment_edit_distance;          /* :56  read by mgx_config_set_scoring_matrix only, like the reference's */
    int8_t alignment_match_score;             /* :57  set_scoring_matrix(); the aligner itself reads score_matrix       */
    int8_t alignment_mm_transition_score;     /* :58  (a positive penalty, negated when the matrix is built)            */
    int8_t alignment_mm_transversion_score;   /* :59 */
    uint8_t _pad1[1];
    int8_t score_matrix[128][128];     /* :61 ScoreMatrix, [graph char][query char] */
} mgx_config;

/* Fill `c` with DBGAlignerConfig{} defaults (aligner_config.hpp:23-54); score matrix zeroed. */
void mgx_config_init_default(mgx_config *c);
/* Fill `c` with the `metagraph align` CLI defaults (cli/config/config.hpp:114-145 through
 * cli/align.cpp:33-69), for a graph with k-mer length k. */
void mgx_config_init_cli(mgx_config *c, uint32_t k);
/* DBGAlignerConfig::dna_scoring_matrix (aligner_config.cpp:164-183); penalties are negative. */
void mgx_config_set_dna_matrix(mgx_config *c, int8_t match, int8_t mm_transition, int8_t mm_transversion);
/* DBGAlignerConfig::unit_scoring_matrix over "ACGT" (aligner_config.cpp:185-204). */
void mgx_config_set_unit_matrix(mgx_config *c, int8_t match);
/* DBGAlignerConfig::set_scoring_matrix (aligner_config.cpp:128-162, DNA alphabet): from alignment_edit_distance /
 * alignment_match_score / alignment_mm_*_score; edit distance also zeroes the end bonuses. */
void mgx_config_set_scoring_matrix(mgx_config *c);

/* Largest DBGAlignerConfig::num_alternative_paths the device path keeps per query (a second instantiation of the
 * extension kernel with room for that many alignments runs when it is > 1).  More: MGX_ERR_UNSUPPORTED. */
#define MGX_MAX_ALTERNATIVE_PATHS 4u

/* Longest query the device path accepts (seed coordinates are 16-bit on the device; per-strand seed lists hold up to
 * 2 L + 64 entries).  Longer queries: MGX_ERR_UNSUPPORTED for the batch.  The reference has no such limit. */
#define MGX_MAX_QUERY_LENGTH 32704u

/* Per-read device arena limits (no reference counterpart; the reference uses the heap). */
typedef struct mgx_limits {
    uint32_t max_query_length;   /* longest read accepted in a batch (<= MGX_MAX_QUERY_LENGTH) */
    uint32_t max_columns;        /* DP-table columns per extension  */
    uint32_t max_seeds;          /* seeds per read and strand       */
    uint64_t cell_arena_bytes;   /* S/E/F storage per in-flight read */
} mgx_limits;
void mgx_limits_init_default(mgx_limits *l, uint32_t max_query_length);
/* The limits the last batch actually ran with (user limits, or what was derived from the config and the batch's
 * longest read).  A caller that sees MGX_ERR_CAPACITY statuses re-runs those reads with larger values. */
int mgx_aligner_get_limits(const struct mgx_aligner *a, mgx_limits *out);

typedef struct mgx_graph mgx_graph;
typedef struct mgx_aligner mgx_aligner;

/* One CIGAR run: Cigar::value_type (aligner_cigar.hpp:28). */
typedef struct mgx_cigar_op { uint32_t len; uint8_t op; uint8_t _pad[3]; } mgx_cigar_op;

/* One alignment: the observable state of `Alignment` (graph/alignment/alignment.hpp:132-331). */
typedef struct mgx_alignment {
    int32_t score;            /* get_score() */
    uint32_t offset;          /* get_offset() */
    uint32_t clipping;        /* get_clipping() */
    uint32_t end_clipping;    /* get_end_clipping() */
    uint32_t num_matches;     /* get_cigar().get_num_matches() */
    uint32_t n_nodes;         /* size() */
    uint32_t n_cigar;
    uint32_t seq_len;         /* get_sequence().size() */
    uint64_t nodes_begin;     /* index into mgx_results.nodes */
    uint64_t cigar_begin;     /* index into mgx_results.cigar */
    uint64_t seq_begin;       /* index into mgx_results.seqs  */
    uint8_t orientation;      /* get_orientation(): 1 = reverse complement of the query matched */
    uint8_t _pad[3];
    uint32_t n_labels;        /* label_columns.size() (alignment.hpp:327; 0 without an annotation) */
    uint64_t labels_begin;    /* index into mgx_results.labels: the alignment's label columns, ascending */
} mgx_alignment;

/* Results of one batch: AlignmentResults per query (alignment.hpp:366-406), in batch order. */
typedef struct mgx_results {
    uint64_t n_queries;
    const uint64_t *aln_begin;        /* n_queries+1 prefix offsets into `alignments` */
    const mgx_alignment *alignments;
    const uint64_t *nodes;
    const mgx_cigar_op *cigar;
    const char *seqs;
    const int32_t *status;            /* per query: MGX_OK or MGX_ERR_CAPACITY */
    const uint32_t *labels;           /* Alignment::label_columns of all alignments (label-aware alignment; else NULL) */
} mgx_results;

/* Seeding-only output (DeBruijnGraph::map_to_nodes_sequentially for both strands). */
typedef struct mgx_mapping {
    uint64_t n_queries;
    const uint64_t *node_begin;       /* n_queries+1 prefix offsets (per strand array) */
    const uint64_t *nodes_fwd;        /* BOSS::map_to_edges of the query (boss.cpp:996-1045) */
    const uint64_t *nodes_rc;         /* ... of its reverse complement (sequence_graph.cpp:563-573) */
} mgx_mapping;

/* Counters for roofline accounting: BOSS primitives actually executed. */
typedef struct mgx_stats {
    uint64_t n_reads;
    uint64_t n_rank_lines;      /* 64-B block reads issued for rank_W / W / last        */
    uint64_t n_select_lines;    /* 64-B block reads issued for select_last / select_W  */
    uint64_t n_bit_lines;       /* side-table reads (MEM terminus bits, first chars)   */
    uint64_t n_columns;         /* DP columns computed                                  */
    uint64_t n_extensions;      /* DefaultColumnExtender::extend calls                  */
    uint64_t n_seeds;
    uint64_t n_map_lines;       /* part of n_rank_lines + n_select_lines issued by the k-mer mapping kernel */
    uint64_t n_capacity_errors; /* reads whose status is MGX_ERR_CAPACITY */
    uint64_t phase_cycles[8];   /* k_align shader cycles summed over waves: query prep, seeding, extend,
                                   backtrack, driver rest, output (profiling aid) */
    uint64_t extend_cycles[8];  /* extend() breakdown: pop, stage+band, outgoing, column, scan, commit, conv, push */
    double seed_kernel_ms, align_kernel_ms;   /* HIP-event time of the k-mer mapping kernel and of the whole
                                                 alignment stage (seeding + sort + extension), last batch */
    double seeding_ms, sort_ms, extend_ms;    /* split pipeline only (0 otherwise): seeding kernel, work sort,
                                                 extension kernel */
    uint64_t n_seed_lines;      /* split pipeline: part of the line counters issued by the seeding kernel */
    uint64_t n_fast_columns;    /* part of n_columns computed by the register-resident chain path */
    uint64_t extend_kernels;    /* which extension kernels the last batch launched (MGX_KERNEL_* bits): what a parity test
                                   asserts so that it is known to have exercised the kernel it means to */
    uint64_t n_lane_reads;      /* reads the lane-per-read kernel finished on its own (the rest went on to the group kernel) */
    double lane_ms;             /* HIP-event time of the lane-per-read kernel (part of extend_ms) */
    uint64_t n_lane_lines;      /* part of the line counters issued by the lane-per-read kernel */
    uint64_t n_lane_columns;    /* part of n_columns computed by it (for the reads it passes on too) */
    uint64_t lane_bail_reads[32]; /* reads the lane-per-read kernel passed on to the group kernel, by reason (the LANE_BAIL codes of
                                   csrc/lane_read.hpp: 3 second strand, 4 many seeds, 5 invalid characters, 10 fork, 14 / 16 wide band,
                                   15 node seen before, 19 a later seed survives, 26 backward extension, 27 an equal-score batch of columns on the query,
                                   28 more parked columns than frontier slots, ...) */
    uint64_t n_capacity_retried; /* queries the last mgx_fetch_results / mgx_align_batch re-aligned with doubled limits after an
                                   MGX_ERR_CAPACITY status (they are in its results like any other query; n_capacity_errors counts
                                   them too) */
    uint64_t n_seed_lane_reads; /* reads the lane-per-read seeder (round 6, csrc/seed_lane.hpp) seeded on its own; the rest went on to the
                                   wave-per-read seeding kernel */
    double seed_lane_ms;        /* HIP-event time of the lane-per-read seeder (part of seeding_ms) */
    uint64_t seed_lane_left_reads[16]; /* reads it left to the seeding kernel, by reason (seed_lane.hpp: 1 read length, 2 k-mer positions,
                                   3 characters outside ACGT, 4 the DUST scan could mask something, 5 alternative nodes, 6 / 7 seeds) */
} mgx_stats;
enum { MGX_KERNEL_GRP8 = 1, MGX_KERNEL_GRP8_PRIM = 2, MGX_KERNEL_GRP8_ALT = 4, MGX_KERNEL_EXT64 = 8, MGX_KERNEL_LANE = 16,
       MGX_KERNEL_LAB64 = 32, MGX_KERNEL_GRP8_LAB = 64 /* the label-aware builds of the 64-lane and the 8-lane kernel */ };

int mgx_device_count(void);                 /* number of visible HIP devices (0 without a GPU) */
const char *mgx_last_error(void);
uint32_t mgx_abi_version(void);

/* Upload a BOSS table and build the device index.  Replaces BOSS::load + DBGSuccinct::load
 * (boss.cpp:338-394, dbg_succinct.cpp:690-785) as the source of W/last/F.  `device` is the HIP ordinal. */
int mgx_graph_create(const mgx_boss_view *view, int device, mgx_graph **out);
void mgx_graph_destroy(mgx_graph *g);
uint32_t mgx_graph_k(const mgx_graph *g);            /* DeBruijnGraph::get_k  (sequence_graph.hpp:176) */
uint64_t mgx_graph_max_index(const mgx_graph *g);    /* DeBruijnGraph::max_index (dbg_succinct.cpp:686-688); PRIMARY: 2 x that,
                                                      * as CanonicalDBG::max_index (canonical_dbg.hpp:96) */
uint64_t mgx_graph_device_bytes(const mgx_graph *g);
uint64_t mgx_graph_num_edges(const mgx_graph *g);    /* boss::BOSS::num_edges: the edges of the stored table (PRIMARY: not doubled) */
uint32_t mgx_graph_mode(const mgx_graph *g);         /* MGX_MODE_* the graph was created (or loaded) with */

/* DBGAligner<>::DBGAligner(graph, config) (dbg_aligner.cpp:33-61): clamps seed lengths,
 * validates scores.  `limits` may be NULL (defaults for 512-bp reads). */
int mgx_aligner_create(const mgx_graph *g, const mgx_config *config,
                       const mgx_limits *limits, mgx_aligner **out);
void mgx_aligner_destroy(mgx_aligner *a);
/* IDBGAligner::get_config() after clamping. */
int mgx_aligner_get_config(const mgx_aligner *a, mgx_config *out);

/* IDBGAligner::align_batch (dbg_aligner.hpp:32-33, dbg_aligner.cpp:251-355).
 * `seqs` holds the concatenated raw query bytes, query i = seqs[offsets[i] .. offsets[i+1]).
 * With seqs_on_device != 0, `seqs` and `offsets` are device pointers (reads already in HBM). */
int mgx_align_batch(mgx_aligner *a, const char *seqs, const uint64_t *offsets, uint64_t n_queries,
                    int seqs_on_device, mgx_results *out);

/* The two halves of mgx_align_batch: run the kernels and leave the results in HBM / copy them out.
 * mgx_align_batch(a, ...) == mgx_align_batch_device(a, ...) followed by mgx_fetch_results(a, out).
 * mgx_fetch_results re-aligns the queries whose per-read arenas overflowed (MGX_ERR_CAPACITY: the reference's tables grow on
 * the heap, it has no such status) with doubled limits, up to six times, and reads them from the batch's device buffers:
 * with seqs_on_device != 0 the caller's `seqs` / `offsets` must stay valid until it returns.  The raw device results
 * (mgx_device_results, the RCCL gather path) are what the kernels wrote: statuses included. */
int mgx_align_batch_device(mgx_aligner *a, const char *seqs, const uint64_t *offsets, uint64_t n_queries,
                           int seqs_on_device);
int mgx_fetch_results(mgx_aligner *a, mgx_results *out);
/* Device-resident results of the last batch, for gathering over RCCL without a host round trip:
 * `headers` = n_queries fixed-size records of `header_bytes` bytes each (status, n_alignments, score,
 * offset, n_nodes, n_cigar, seq_len, orientation, stream offset, ...), `stream` = `stream_words`
 * 32-bit words holding nodes / packed CIGAR runs (len << 3 | op) / path characters. */
int mgx_device_results(mgx_aligner *a, const void **headers, uint64_t *header_bytes, uint64_t *n_queries,
                       const void **stream, uint64_t *stream_words);
/* Capacity (32-bit words) of the device stream buffer behind mgx_device_results: a function of the batch shape only,
 * so equal on all ranks that run equally shaped batches — the padded RCCL gather relies on it. */
uint64_t mgx_device_stream_capacity(const mgx_aligner *a);
/* Host-only decode of raw result records (one rank's `headers` / `stream` as mgx_device_results exposes them, e.g.
 * after an RCCL gather on the root) into the mgx_results view that mgx_format_tsv prints from — the root's side of
 * cli/align.cpp:469-473.  Needs no GPU.  `*store` owns the memory behind `out`; release with mgx_raw_store_free. */
typedef struct mgx_raw_store mgx_raw_store;
int mgx_results_from_raw(const void *headers, uint64_t n_queries, const uint32_t *stream, uint64_t stream_words,
                         mgx_raw_store **store, mgx_results *out);
void mgx_raw_store_free(mgx_raw_store *store);
/* chain_alignments<LocalAlignmentLess> (aligner_chainer.cpp:555-720, called at dbg_aligner.cpp:328-332) over the decoded
 * results of a batch whose aligner had post_chain_alignments set: queries seqs[offsets[q] .. offsets[q + 1]) as they were
 * aligned, k = the graph's k (node_overlap = k - 1).  Host code, needs no GPU.  A chain's nodes hold 0 where the path has no
 * graph node, its spelling '$' at a gap, its CIGAR MGX_OP_NODE_INSERTION runs and clipping runs inside. */
int mgx_chain_alignments(const mgx_config *config, uint32_t k, const mgx_results *in, const char *seqs, const uint64_t *offsets,
                         mgx_raw_store **store, mgx_results *out);
/* Test hooks: keep and fetch the per-read seed lists (DBGAligner::build_seeders products). */
void mgx_aligner_keep_seeds(mgx_aligner *a, int keep);
int mgx_fetch_seed_info(mgx_aligner *a, uint32_t *info6, uint32_t *seeds, uint32_t *max_seeds_out);

/* Hot loop #1 only: map both strands to nodes (dbg_aligner.cpp:210,227-231): map_to_nodes_sequentially of the query and of
 * its reverse complement; on a PRIMARY graph those of the CanonicalDBG wrapper (canonical_dbg.cpp:55-146,551-560). */
int mgx_map_batch(mgx_aligner *a, const char *seqs, const uint64_t *offsets, uint64_t n_queries,
                  int seqs_on_device, mgx_mapping *out);

int mgx_aligner_stats(const mgx_aligner *a, mgx_stats *out);
/* Test hook: launches of each extension kernel since the library was loaded, out5[b] = the kernel of bit b of MGX_KERNEL_*
 * (8-lane groups, its PRIMARY build, its alternative-paths build, the 64-lane kernel, the lane-per-read kernel). */
void mgx_kernel_launch_counts(uint64_t *out5);

/* Kernel-selection switches; every setting gives the same alignments (the parity suite runs them all).  "split8" names the
 * (only) pipeline: seeding kernel with one wavefront per read, radix sort of the reads by predicted extension work, extension
 * kernel(s).  "general" / "chain": the extension's register-resident chain path off / on.  "key=value" (-1 = automatic):
 *   ext64=0|1|2          small batches on the one-read-per-wavefront 64-lane kernel (default 1) or on the 8-lane groups;
 *                        label-aware aligners: 2 = every batch on the 64-lane labeled kernel
 *   groups_per_wave=n    8-lane kernel: n = 1 .. 8 groups of a wavefront take reads, 0 = all (default: from the batch size)
 *   multi_pass=0|1       one extension per read and launch (default: automatic from the seeds per read); two_pass=1
 *   lane=0|1             the lane-per-read kernel in front of the group kernel (default: automatic)
 *   device_share=n       n handles are at work on this device at the same time (worker threads): this handle sizes its per-slot
 *                        arenas for 1 / n of the machine (default 1: all of it)
 *   no_compact / no_alias / no_bt_runs / no_flat / primary_alt_build = 1   A/B forms of the column records and loops
 * Unknown name: MGX_ERR_INVALID.  (Measurement probes that change results or occupancy exist only in -DMGX_PROBES builds.) */
int mgx_aligner_set_pipeline(mgx_aligner *a, const char *name);
/* The HIP stream (a hipStream_t, passed as void * so that this header needs no HIP header) every kernel launch, asynchronous
 * copy and device-library call of this aligner goes to; its blocking copies synchronise that stream only.  NULL (the state
 * after creation) = the legacy default stream.  With a stream of its own per handle, the aligners of several worker threads on
 * one device — the reference builds one aligner per thread-pool task, cli/align.cpp:440-475 — run side by side instead of
 * serialising on the default stream.  The stream stays the caller's (not destroyed with the aligner); device buffers handed to
 * mgx_align_batch_device / mgx_map_batch must be ready on it.  mgx_aligner_create_stream gives the aligner a non-blocking
 * stream that it owns (destroyed with it) — what host/mgx_align -p N does per worker. */
/* Device memory of destroyed aligners is kept per device for the next aligner (one aligner per task is the reference's model:
 * without this every task paid for fresh multi-GB arenas); it is released when an allocation of the library fails and by this
 * call — for a host that wants the memory back for something else. */
int mgx_device_trim(int device);
int mgx_aligner_set_stream(mgx_aligner *a, void *hip_stream);
int mgx_aligner_create_stream(mgx_aligner *a);
void *mgx_aligner_get_stream(const mgx_aligner *a);

/*
 * The gather of complete alignments over RCCL (north_star: "RCCL-over-xGMI used only to gather alignment results").
 * Replaces, for one aligner per device, the reference's output loop (cli/align.cpp:469-473: every task prints its queries under a
 * mutex): every rank's device results of a batch (mgx_device_results: n_queries records + the used words of its stream) travel to
 * `root`, which decodes them with mgx_results_from_raw[_labeled] and prints.  Two phases: an all-gather of (n_queries, used words),
 * then one group of send / receive pairs (xGMI is point-to-point).  librccl.so is opened at the first call (dlopen: libmgx does not
 * link it); without it every mgx_gather_create* returns MGX_ERR_UNSUPPORTED.  No other call of this library communicates.
 *   one process per device:  rank 0 calls mgx_gather_unique_id and hands the bytes to the other ranks (a file, MPI, a socket: the
 *                            host's business), every rank calls mgx_gather_create_rank; or mgx_gather_create_comm over an ncclComm_t
 *                            the host has already (not destroyed with the handle);
 *   one process, D devices:  mgx_gather_create_local (ncclCommInitAll) fills out[0 .. D); worker thread r uses out[r] — the calls
 *                            below are collective and block, so every device needs its own thread (host/mgx_align --rccl-gather).
 */
typedef struct mgx_gather mgx_gather;
uint64_t mgx_gather_unique_id_bytes(void);
int mgx_gather_unique_id(void *id_out, uint64_t id_bytes);
int mgx_gather_create_rank(const void *unique_id, uint64_t id_bytes, int rank, int world, int root, int device, mgx_gather **out);
int mgx_gather_create_comm(void *nccl_comm, int rank, int world, int root, int device, mgx_gather **out);
int mgx_gather_create_local(const int *devices, int n_devices, int root, mgx_gather **out /* [n_devices] */);
void mgx_gather_destroy(mgx_gather *g);
int mgx_gather_world(const mgx_gather *g);
int mgx_gather_rank(const mgx_gather *g);
/* Collective: every rank, after mgx_align_batch_device on ITS aligner (same device as the handle).  Returns when this rank's
 * transfers are enqueued on the handle's own stream; the aligner's device results must stay untouched until mgx_gather_finish. */
int mgx_gather_start(mgx_gather *g, mgx_aligner *a);
/* Waits for this rank's transfers.  On the root the caller's arrays of `world` entries receive, per rank r, the number of
 * queries, host pointers to its records and stream (valid until the next mgx_gather_start / mgx_gather_destroy) and the
 * stream's words — the arguments of mgx_results_from_raw.  On the other ranks the arrays are ignored (may be NULL). */
int mgx_gather_finish(mgx_gather *g, uint64_t *n_queries, const void **headers, const uint32_t **streams, uint64_t *stream_words);

/* Format one query's results exactly like format_alignment() (cli/align.cpp:254-285):
 * "header\tquery\t(+|-)\tpath\tscore\tnum_matches\tcigar\toffset\n", or the "*" line.
 * Returns the number of bytes needed (excluding NUL); writes at most buf_len bytes. */
size_t mgx_format_tsv(const mgx_results *res, uint64_t query_index, const char *header,
                      const char *query, size_t query_len, int32_t min_path_score,
                      char *buf, size_t buf_len);

/* Format one query's results like `metagraph align --json` (cli/align.cpp:287-305): one line per alignment, the JSON of
 * Alignment::to_json / path_json (alignment.cpp:704-963) as Json::writeString emits it with indentation "" (keys in
 * lexicographic order, no white space); a query without alignments yields {"name":...,"sequence":""}.  `k` is the graph's
 * node length (DeBruijnGraph::get_k()).  Returns the number of bytes needed (excluding NUL); writes at most buf_len bytes. */
size_t mgx_format_json(const mgx_results *res, uint64_t query_index, const char *header,
                       const char *query, size_t query_len, uint32_t k, char *buf, size_t buf_len);

/*
 * Label-aware alignment (graph/alignment/aligner_labeled.{hpp,cpp}, annotation_buffer.{hpp,cpp}; BASELINE config 3).
 * On the device: the whole LabeledAligner — filter_seeds, LabeledExtender's per-column label sets (flush / call_outgoing / the
 * label-bounded backtracking), the per-label aggregator (mgx_labeled_aligner_create below) — over a row-major label matrix
 * whose rows ARE the label sets AnnotationBuffer would fetch, plus the batched BinaryMatrix::get_rows of
 * AnnotationBuffer::fetch_queued_annotations as an entry point of its own (annotation_buffer.cpp:182; ColumnMajor::get_rows,
 * annotation/binary_matrix/column_sparse/column_major.cpp:27-44 — 1000 columns x |rows| random bit tests in the reference).
 * Not built: label coordinates and seed chaining (aligner_labeled.cpp:361-448,685-700, aligner_chainer.cpp:47-339).
 *
 * mgx_annotation_create stands in for the annotator argument of LabeledAligner<>(graph, config, annotator)
 * (aligner_labeled.hpp:125-127): the binary matrix as column bit vectors, columns[j] = ceil(n_rows / 64) words, bit r =
 * row r carries label j; row = AnnotatedDBG::graph_to_anno_index(node) = node - 1 (graph/annotated_dbg.hpp:50-52).
 * The device keeps it row-major (one 8-byte word per row; rows with several labels point into a label list).
 */
typedef struct mgx_annotation mgx_annotation;
int mgx_annotation_create(uint64_t n_rows, uint32_t n_labels, const uint64_t *const *columns, int device, mgx_annotation **out);
/* The same from the columns' set rows, the content of a ColumnCompressed annotation (one sd_vector of row indices per label):
 * rows[col_begin[j] .. col_begin[j + 1]) = the rows with label j (any order, no duplicates); col_begin: host array of
 * n_labels + 1 entries; rows: host or (on_device != 0) device pointer.  One sort on the device instead of n_labels bit vectors
 * of n_rows bits each on the host. */
int mgx_annotation_create_sparse(uint64_t n_rows, uint32_t n_labels, const uint64_t *col_begin, const uint64_t *rows,
                                 int on_device, int device, mgx_annotation **out);
void mgx_annotation_destroy(mgx_annotation *a);
uint64_t mgx_annotation_device_bytes(const mgx_annotation *a);
uint64_t mgx_annotation_num_rows(const mgx_annotation *a);
uint32_t mgx_annotation_num_labels(const mgx_annotation *a);
/* BinaryMatrix::get_rows for a batch of rows, as CSR: labels of rows[i] = out_labels[out_begin[i] .. out_begin[i + 1]),
 * ascending (what annotation_buffer.cpp:185 sorts into).  Calls on one handle are serialised inside (the handle owns the scratch
 * buffers); they run on the null stream.  out_begin has n + 1 entries; `cap` = entries out_labels holds: a
 * batch with more returns MGX_ERR_CAPACITY with the needed count in *n_labels_out (out_begin is complete, out_labels
 * untouched).  rows_on_device / out_on_device != 0: device pointers (rows already in HBM / results left in HBM). */
int mgx_annotation_get_rows(mgx_annotation *a, const uint64_t *rows, uint64_t n, int rows_on_device,
                            uint64_t *out_begin, uint32_t *out_labels, uint64_t cap, int out_on_device, uint64_t *n_labels_out);

/* k-mer coordinates (annot::ColumnCoordAnnotator: a MultiIntMatrix next to the binary matrix; AnnotationBuffer::has_coordinates,
 * annotation_buffer.cpp:26-33) for an annotation created from the same n_labels / col_begin / rows with
 * mgx_annotation_create_sparse: pair x (label j = the column whose range holds x, row rows[x]) has the ascending coordinates
 * coords[coord_begin[x] .. coord_begin[x + 1]).  Host arrays.  Round 6: the annotation and its batched lookup
 * (mgx_annotation_get_row_tuples = MultiIntMatrix::get_row_tuples, what AnnotationBuffer::fetch_queued_annotations calls,
 * annotation_buffer.cpp:166-181) are on the device; LabeledAligner's coordinate mode itself (seed chaining,
 * aligner_labeled.cpp:457-462) is not — mgx_labeled_aligner_create refuses such an annotation (MGX_ERR_UNSUPPORTED). */
int mgx_annotation_set_coordinates(mgx_annotation *a, uint32_t n_labels, const uint64_t *col_begin, const uint64_t *rows,
                                   const uint64_t *coord_begin, const int64_t *coords);
int mgx_annotation_has_coordinates(const mgx_annotation *a);
int mgx_annotation_get_row_tuples(mgx_annotation *a, const uint64_t *rows, uint64_t n, uint64_t *out_begin, uint32_t *out_labels,
                                  uint64_t label_cap, uint64_t *out_coord_begin, int64_t *out_coords, uint64_t coord_cap,
                                  uint64_t *n_labels_out, uint64_t *n_coords_out);

/* chain_seeds (aligner_chainer.cpp:341-542) — the anchor DP of seed chaining, on the device.  An anchor is one (seed, label,
 * coordinate) of a (query, strand): the reference's TableElem (aligner_chainer.cpp:23-36, same 32 bytes).  Input: the anchors of
 * n_lists lists (list l = anchors[list_begin[l] .. list_begin[l + 1])) in any order, chain_score = the seed's length
 * (seed_end - seed_clipping), query_size[l] = the length of the list's query; config->min_seed_length sets the gap cost.  Output:
 * every list sorted as the reference sorts it — (label, coordinate, seed_clipping, seed_end) descending — with the final
 * chain_score of every anchor, and backtrace_out[x] = the anchor (index relative to its list's begin, in the sorted order)
 * anchor x's best chain continues with, 0xFFFFFFFF for none.  Bit-identical to the reference's AVX2 loop incl. the float gap
 * cost (tabulated on the host with the host's libm).  At most 6144 anchors per list.  Host arrays. */
typedef struct mgx_chain_anchor {
    uint64_t label;
    int64_t coordinate;
    int32_t seed_clipping, seed_end;
    int32_t chain_score;
    uint32_t seed_index;
} mgx_chain_anchor;
int mgx_chain_seeds(const mgx_config *config, int device, const mgx_chain_anchor *anchors, const uint64_t *list_begin,
                    const uint32_t *query_size, uint64_t n_lists, mgx_chain_anchor *sorted_out, uint32_t *backtrace_out);

/* LabeledAligner<>(graph, config, annotator) (aligner_labeled.hpp:125-127; ctor aligner_labeled.cpp:450-466: DBGAligner's
 * clamps, then min / max_seed_length <= k): an aligner whose batches run label-aware — seeds filtered by label
 * (filter_seeds, aligner_labeled.cpp:612-721), every DP column carries the labels shared along its path
 * (LabeledExtender::call_outgoing / flush, :81-137,176-302), backtracking reports one alignment per label subset of the seed
 * (:304-448), the aggregator keeps num_alternative_paths alignments per label (aligner_aggregator.hpp:68-206).  Every
 * mgx_alignment of its results carries its labels (n_labels, labels_begin into mgx_results.labels, ascending).  Used with
 * mgx_align_batch / mgx_align_batch_device / mgx_fetch_results like any aligner; `annotation` must outlive it and live on
 * the graph's device, with at least as many rows as the graph has nodes (row = node - 1).
 * This round: BASIC-, PRIMARY- and CANONICAL-mode graphs (PRIMARY: through the CanonicalDBG wrapper, labels looked up by base
 * node; CANONICAL: by the k-mer's representative, the smaller BOSS index of the k-mer and its reverse complement — as
 * annotation_buffer.cpp:41-63 does), annotations without coordinates (ColumnCompressed), num_alternative_paths <= MGX_MAX_ALTERNATIVE_PATHS;
 * anything else: MGX_ERR_UNSUPPORTED.  A read whose label bookkeeping outgrows the arenas of a first run (64 labels with
 * alignments, 8 alignments per backtracking, ...) gets MGX_ERR_CAPACITY from mgx_align_batch_device; mgx_align_batch re-runs it
 * with the arenas doubled (up to 8 x) like any other capacity status (mgx_stats.n_capacity_retried).
 * One case the reference leaves UNDEFINED on CANONICAL-mode graphs: the reversed alignment that seeds a backward pass holds
 * reverse-complement nodes its AnnotationBuffer was never asked to fetch; set_seed / flush assert on them
 * (aligner_labeled.cpp:110,143-146) and a release build reads through a null pointer.  The library answers such a look-up
 * with the labels of the node's representative — what a fetch would have found; the oracle does the same and counts them
 * (orc_unfetched_label_lookups: 0 on the reference's own label tests, which is what pins this path; reads that do need
 * such a look-up agree between library and oracle but have no reference behaviour to agree with). */
int mgx_labeled_aligner_create(const mgx_graph *graph, const mgx_config *config, const mgx_limits *limits,
                               const mgx_annotation *annotation, mgx_aligner **out);
/* mgx_format_tsv for label-aware results: every alignment's fields are followed by its labels' names joined by ';'
 * (cli/align.cpp:274-281; label_names[j] = LabelEncoder::decode(j); a label without a name prints its number). */
size_t mgx_format_tsv_labeled(const mgx_results *res, uint64_t query_index, const char *header,
                              const char *query, size_t query_len, int32_t min_path_score,
                              const char *const *label_names, uint32_t n_label_names, char *buf, size_t buf_len);
/* mgx_results_from_raw for the records of a label-aware aligner (labeled != 0: every alignment's arrays are followed by
 * its label list in the stream). */
int mgx_results_from_raw_labeled(const void *headers, uint64_t n_queries, const uint32_t *stream, uint64_t stream_words,
                                 int labeled, mgx_raw_store **store, mgx_results *out);

/*
 * Files: the graph and annotation files the reference writes, read on the host and handed to
 * mgx_graph_create / mgx_annotation_create_sparse.  What a stand-alone caller (host/mgx_align) needs in place of
 * DBGSuccinct::load (dbg_succinct.cpp:690-785 over BOSS::load, boss.cpp:338-394) and ColumnCompressed::load / merge_load
 * (annotate_column_compressed.cpp:436-481,493-640); inside MetaGraph the adapter takes both from the loaded objects
 * (INTEGRATION.md 2).  Read: BOSS states SMALL, STAT and FAST (DYN: MGX_ERR_UNSUPPORTED), any alphabet in
 * mgx_boss_file_read (the device index itself is DNA only); columns stored as sd_vector, bit_vector_stat or rrr_vector<63>
 * (bit_vector_smart writes the first two), both label-encoder formats.  The `.edgemask` next
 * to a `.dbg`: mgx_edgemask_read.  Which layouts are pinned on reference-written files:
 * csrc/boss_files.hpp.  Malformed file: MGX_ERR_INVALID with the field named in mgx_last_error; the *_read calls need no GPU.
 */
typedef struct mgx_boss_file {
    uint32_t k;              /* DBG k (BOSS k + 1) */
    uint32_t sigma;          /* alphabet size incl. '$' (F's length): 5 = DNA */
    uint32_t mode;           /* MGX_MODE_* as stored behind the BOSS table (dbg_succinct.cpp:701) */
    uint32_t state;          /* BOSS::State of the file (boss.hpp:325): 1 SMALL, 3 STAT, 4 FAST */
    uint64_t n_edges;
    const uint64_t *F;       /* sigma entries */
    const uint8_t *W;        /* n_edges + 1 bytes, the layout of mgx_boss_view */
    const uint8_t *last;
    void *owner;             /* private */
} mgx_boss_file;
int mgx_boss_file_read(const char *path, mgx_boss_file *out);
void mgx_boss_file_free(mgx_boss_file *f);
/* The `.edgemask` file next to a `.dbg` (dbg_succinct.cpp:719-752) as mgx_boss_view.valid bytes: valid_out has n_edges + 1
 * entries; `state` = mgx_boss_file.state of the graph (it decides the mask's vector type).  mgx_graph_load_dbg does not apply
 * it — `metagraph align` resets the mask after loading (cli/align.cpp:337-339); a caller that keeps the mask builds the view
 * from mgx_boss_file_read + this. */
int mgx_edgemask_read(const char *path, uint32_t state, uint64_t n_edges, uint8_t *valid_out);
/* mgx_boss_file_read + mgx_graph_create: DBGSuccinct::load for the device */
int mgx_graph_load_dbg(const char *path, int device, mgx_graph **out);

typedef struct mgx_column_file mgx_column_file;
/* One or several `.column.annodbg` files over the same rows, their columns side by side in argument order (merge_load;
 * a label occurring in two files: MGX_ERR_UNSUPPORTED). */
int mgx_column_file_read(const char *const *paths, uint32_t n_paths, mgx_column_file **out);
void mgx_column_file_free(mgx_column_file *f);
uint64_t mgx_column_file_num_rows(const mgx_column_file *f);
uint32_t mgx_column_file_num_labels(const mgx_column_file *f);
const char *mgx_column_file_label(const mgx_column_file *f, uint32_t j);      /* LabelEncoder::decode(j) */
const uint64_t *mgx_column_file_col_begin(const mgx_column_file *f);          /* num_labels + 1 entries */
const uint64_t *mgx_column_file_rows(const mgx_column_file *f);               /* the arguments of mgx_annotation_create_sparse */
int mgx_annotation_create_from_file(const mgx_column_file *f, int device, mgx_annotation **out);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H_ */
