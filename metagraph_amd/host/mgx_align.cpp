// mgx_align — minimal `metagraph align`-shaped driver over HipDBGAligner (host side only: batching and
// TSV printing as in cli/align.cpp:403-480).  Graph input: a `.dbg` file written by the reference (mgx_boss_file_read:
// SMALL / STAT / FAST state, DNA; the graph's mode is the file's) or a flat BOSS dump (k, n_edges, F[5], W[], last[]).  Usage:
//   mgx_align GRAPH.{dbg,boss} READS.{fa,fq} [--align-only-forwards] [--align-min-exact-match X] [--align-min-seed-length N]
//             [-p THREADS] [--query-batch-size BASES] [--canonical | --primary (the dump is a CANONICAL- / PRIMARY-mode graph)]
//             [--time]        wall time of the align loop on stderr
//             [--devices D]   in-process multi-GPU: one graph replica per device, whole batches routed round-robin, no collective
//             [--rccl-gather] with --devices D: one worker per device, batches in rounds of D; every round's device results are
//                             gathered to device 0 over RCCL (mgx_gather_*: the C-ABI of north_star's "RCCL-over-xGMI only to
//                             gather alignment results"), decoded there (mgx_results_from_raw) and printed by that worker alone
//             [-a ANNOTATION]  label-aware alignment (metagraph align -a: LabeledAligner); every alignment is printed with its
//                              labels' names (cli/align.cpp:274-281).  ANNOTATION: `.column.annodbg` files written by the
//                              reference (-a may be repeated: their columns side by side, mgx_column_file_read), or a dump of
//                              the columns — u64 n_rows, u64 n_labels, then per label: u64 name length, the name, u64 count,
//                              count x u64 rows (row = node - 1)
//                             (the reference's unit of parallelism, cli/align.cpp:440-475: one task per batch)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>

#include "hip_dbg_aligner.hpp"

using namespace mgx::host;

static bool read_records(const std::string &path, std::vector<IDBGAligner::Query> *out) {
    std::ifstream in(path);
    if (!in) return false;
    std::string line, name, seq;
    bool fastq = false;
    auto flush = [&]() { if (!name.empty()) out->emplace_back(name, seq); name.clear(); seq.clear(); };
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        if (line[0] == '@' && (name.empty() || fastq)) {       // FASTQ record: 4 lines
            fastq = true;
            flush();
            name = line.substr(1, line.find_first_of(" \t") - 1);
            std::getline(in, seq);
            std::string plus, qual;
            std::getline(in, plus);
            std::getline(in, qual);
            flush();
        } else if (line[0] == '>') {
            flush();
            name = line.substr(1, line.find_first_of(" \t") - 1);
        } else {
            seq += line;
        }
    }
    flush();
    return true;
}

// cli/align.cpp:415-480: records are read into batches of at most `query_batch_size` bases (default 100 MB,
// cli/config/config.hpp:105), every batch is a task of a pool of `-p` workers; a task builds its own aligner over the
// shared read-only graph and prints each query's line under a mutex as it completes (so with -p 1 output is in input
// order, with more workers batches interleave — exactly the reference's behaviour).
int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s GRAPH.boss READS [--align-only-forwards] [--align-min-exact-match X] [--align-min-seed-length N]\n"
                        "          [-p THREADS] [--query-batch-size BASES] [--max-columns N (test hook: small device arena)]\n", argv[0]);
        return 2;
    }
    auto ends_with = [](const std::string &s, const char *suffix) { const size_t n = strlen(suffix); return s.size() >= n && !s.compare(s.size() - n, n, suffix); };
    uint64_t hdr[7];
    std::vector<uint8_t> W, last;
    uint32_t graph_mode = MGX_MODE_BASIC;
    if (ends_with(argv[1], ".dbg")) {             // DBGSuccinct::load (dbg_succinct.cpp:690-711)
        mgx_boss_file f;
        if (mgx_boss_file_read(argv[1], &f) != MGX_OK) { fprintf(stderr, "error: %s\n", mgx_last_error()); return 1; }
        if (f.sigma != 5) { fprintf(stderr, "error: %s: alphabet of %u characters; DNA graphs only\n", argv[1], f.sigma); mgx_boss_file_free(&f); return 1; }
        hdr[0] = f.k; hdr[1] = f.n_edges;
        for (int i = 0; i < 5; ++i) hdr[2 + i] = f.F[i];
        W.assign(f.W, f.W + f.n_edges + 1);
        last.assign(f.last, f.last + f.n_edges + 1);
        graph_mode = f.mode;
        mgx_boss_file_free(&f);
    } else {
        std::ifstream gin(argv[1], std::ios::binary);
        if (!gin) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
        gin.read((char *)hdr, sizeof(hdr));       // k, n_edges, F[0..4]
        W.resize(hdr[1] + 1); last.resize(hdr[1] + 1);
        gin.read((char *)W.data(), (std::streamsize)W.size());
        gin.read((char *)last.data(), (std::streamsize)last.size());
        if (!gin) { fprintf(stderr, "bad BOSS dump %s\n", argv[1]); return 1; }
    }
    const uint32_t k = (uint32_t)hdr[0];
    const uint64_t n = hdr[1];
    DBGAlignerConfig cfg;
    mgx_config_init_cli(&cfg, k);
    unsigned threads = 1;
    bool report_time = false;
    uint64_t batch_size = 100000000ull;
    mgx_limits lim;
    bool have_lim = false;
    int devices = 1;
    bool rccl_gather = false;
    std::vector<const char *> anno_paths;
    std::vector<std::string> kernel_options;            // --kernel-option key=value: result-preserving kernel selection (A/B runs)
    mgx_limits_init_default(&lim, 0);
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--align-only-forwards")) cfg.forward_and_reverse_complement = 0;
        else if (!strcmp(argv[i], "--align-min-exact-match") && i + 1 < argc) cfg.min_exact_match = atof(argv[++i]);
        else if (!strcmp(argv[i], "--align-min-seed-length") && i + 1 < argc) cfg.min_seed_length = std::min<uint64_t>(atoi(argv[++i]), k);
        else if (!strcmp(argv[i], "-p") && i + 1 < argc) threads = (unsigned)std::max(1, atoi(argv[++i]));
        else if (!strcmp(argv[i], "--query-batch-size") && i + 1 < argc) batch_size = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--max-columns") && i + 1 < argc) { lim.max_columns = (uint32_t)atoi(argv[++i]); have_lim = true; }
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) devices = std::max(1, atoi(argv[++i]));
        else if (!strcmp(argv[i], "-a") && i + 1 < argc) anno_paths.push_back(argv[++i]);
        else if (!strcmp(argv[i], "--kernel-option") && i + 1 < argc) kernel_options.push_back(argv[++i]);
        else if (!strcmp(argv[i], "--rccl-gather")) rccl_gather = true;
        else if (!strcmp(argv[i], "--time")) report_time = true;            // wall time of the align loop (batches -> results printed) on stderr
        else if (!strcmp(argv[i], "--canonical")) graph_mode = MGX_MODE_CANONICAL;
        else if (!strcmp(argv[i], "--primary")) graph_mode = MGX_MODE_PRIMARY;         // aligned through the CanonicalDBG wrapper
    }
    try {
        // one replica of the index per device (3.5 B/edge + the suffix-range table each); more devices than the box shows is an
        // error of the caller's, reported like any other
        if (devices > mgx_device_count() && mgx_device_count() > 0) {
            fprintf(stderr, "error: --devices %d but %d HIP device(s) visible\n", devices, mgx_device_count());
            return 1;
        }
        HipGraphSet graphs(devices, k, n, W.data(), last.data(), hdr + 2, nullptr, graph_mode);
        std::unique_ptr<HipAnnotation> annotation;
        std::vector<std::string> label_names;
        if (!anno_paths.empty() && devices != 1) { fprintf(stderr, "error: -a with --devices 1 only (one annotation replica)\n"); return 1; }
        if (!anno_paths.empty() && ends_with(anno_paths[0], ".annodbg")) {      // ColumnCompressed::merge_load
            mgx_column_file *cf = nullptr;
            if (mgx_column_file_read(anno_paths.data(), (uint32_t)anno_paths.size(), &cf) != MGX_OK) { fprintf(stderr, "error: %s\n", mgx_last_error()); return 1; }
            const uint32_t nl = mgx_column_file_num_labels(cf);
            for (uint32_t j = 0; j < nl; ++j) label_names.emplace_back(mgx_column_file_label(cf, j));
            const uint64_t *cb = mgx_column_file_col_begin(cf), *rw = mgx_column_file_rows(cf);
            const uint64_t n_rows = mgx_column_file_num_rows(cf);
            std::vector<uint64_t> col_begin(cb, cb + nl + 1), rows(rw, rw + cb[nl]);
            mgx_column_file_free(cf);
            // AnnotatedDBG::check_compatibility (annotated_dbg.cpp): one row per node
            if (n_rows != n) { fprintf(stderr, "error: the annotation has %llu rows, the graph %llu nodes\n", (unsigned long long)n_rows, (unsigned long long)n); return 1; }
            annotation = std::make_unique<HipAnnotation>(n_rows, col_begin, rows, 0);
        } else if (!anno_paths.empty()) {
            const char *anno_path = anno_paths[0];
            std::ifstream ain(anno_path, std::ios::binary);
            if (!ain) { fprintf(stderr, "cannot open %s\n", anno_path); return 1; }
            uint64_t n_rows = 0, n_labels = 0;
            ain.read((char *)&n_rows, 8); ain.read((char *)&n_labels, 8);
            std::vector<uint64_t> col_begin{ 0 }, rows;
            for (uint64_t j = 0; j < n_labels && ain; ++j) {
                uint64_t len = 0, cnt = 0;
                ain.read((char *)&len, 8);
                std::string name(len, '\0');
                ain.read(name.data(), (std::streamsize)len);
                ain.read((char *)&cnt, 8);
                const size_t at = rows.size();
                rows.resize(at + cnt);
                ain.read((char *)(rows.data() + at), (std::streamsize)(cnt * 8));
                col_begin.push_back(rows.size());
                label_names.push_back(std::move(name));
            }
            if (!ain || col_begin.size() != n_labels + 1) { fprintf(stderr, "bad annotation dump %s\n", anno_path); return 1; }
            annotation = std::make_unique<HipAnnotation>(n_rows, col_begin, rows, 0);
        }
        if ((unsigned)devices > threads) threads = (unsigned)devices;                  // at least one worker per device
        std::vector<IDBGAligner::Query> all;
        if (!read_records(argv[2], &all)) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        // batches by bases read (align.cpp:431-442: a record is added while the running total is <= batch_size)
        std::vector<std::vector<IDBGAligner::Query>> batches;
        const size_t n_queries = all.size();
        for (size_t i = 0; i < all.size();) {
            std::vector<IDBGAligner::Query> b;
            uint64_t bytes = 0;
            for (; i < all.size() && bytes <= batch_size; ++i) { bytes += all[i].second.size(); b.push_back(std::move(all[i])); }
            batches.push_back(std::move(b));
        }
        std::mutex print_mutex, err_mutex;
        std::atomic<size_t> next{ 0 };
        std::string first_error;
        auto worker = [&](unsigned worker_id) {
            try {
                const HipBOSSGraph &graph = graphs.for_worker(worker_id);               // a worker stays on its device
                for (;;) {
                    const size_t bi = next.fetch_add(1);
                    if (bi >= batches.size()) break;
                    // one aligner per task, shared graph (and annotation: LabeledAligner<>(graph, config, annotator))
                    std::unique_ptr<HipDBGAligner> aligner_p(annotation
                        ? new HipDBGAligner(graph, cfg, *annotation, have_lim ? &lim : nullptr)
                        : new HipDBGAligner(graph, cfg, have_lim ? &lim : nullptr));
                    HipDBGAligner &aligner = *aligner_p;
                    // (the workers of one device share it: every handle on its own stream, its arenas sized for its share)
                    aligner.set_device_share((threads + (unsigned)devices - 1) / (unsigned)devices);
                    for (const std::string &opt : kernel_options) aligner.set_kernel_option(opt);
                    aligner.align_batch(batches[bi], [&](const std::string &header, AlignmentResults &&paths) {
                        const std::string res = format_alignment(header, paths, cfg.min_path_score, annotation ? &label_names : nullptr);
                        std::lock_guard<std::mutex> lock(print_mutex);
                        std::cout << res;
                    });
                }
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> lock(err_mutex);
                if (first_error.empty()) first_error = e.what();
            }
        };
        // --rccl-gather: worker w = rank w = device w; round r takes batches r D .. r D + D - 1 (a rank without one aligns an empty
        // batch: the gather is collective); rank 0 receives every rank's records and stream and prints the round
        std::vector<mgx_gather *> gathers((size_t)devices, nullptr);
        auto gather_worker = [&](unsigned w) {
            const size_t D = (size_t)devices, rounds = (batches.size() + D - 1) / D;
            const HipBOSSGraph &graph = graphs.for_worker(w);
            static const std::vector<IDBGAligner::Query> no_queries;
            for (size_t r = 0; r < rounds; ++r) {
                const size_t bi = r * D + w;
                const std::vector<IDBGAligner::Query> &mine = bi < batches.size() ? batches[bi] : no_queries;
                std::unique_ptr<HipDBGAligner> aligner_p;
                bool ok = true;
                try {
                    aligner_p.reset(annotation ? new HipDBGAligner(graph, cfg, *annotation, have_lim ? &lim : nullptr)
                                               : new HipDBGAligner(graph, cfg, have_lim ? &lim : nullptr));
                    for (const std::string &opt : kernel_options) aligner_p->set_kernel_option(opt);
                    aligner_p->align_batch_device(mine);
                } catch (const std::exception &e) {
                    std::lock_guard<std::mutex> lock(err_mutex);
                    if (first_error.empty()) first_error = e.what();
                    ok = false;
                }
                if (!ok) {                            // (stay in the collective: the other ranks are waiting in it)
                    try { aligner_p.reset(new HipDBGAligner(graph, cfg, have_lim ? &lim : nullptr)); aligner_p->align_batch_device(no_queries); }
                    catch (const std::exception &) { fprintf(stderr, "error: rank %u cannot take part in the gather\n", w); std::abort(); }
                }
                std::vector<uint64_t> nq(D), words(D);
                std::vector<const void *> hdrs(D);
                std::vector<const uint32_t *> streams(D);
                if (mgx_gather_start(gathers[w], aligner_p->handle()) != MGX_OK
                    || mgx_gather_finish(gathers[w], nq.data(), hdrs.data(), streams.data(), words.data()) != MGX_OK) {
                    fprintf(stderr, "error: rank %u: %s\n", w, mgx_last_error());
                    std::abort();                    // (a broken collective cannot be left politely)
                }
                if (w != 0) continue;
                for (size_t q = 0; q < D; ++q) {
                    const size_t bq = r * D + q;
                    if (bq >= batches.size()) continue;
                    try {
                        if (nq[q] != batches[bq].size()) throw std::runtime_error("rank " + std::to_string(q) + " sent " + std::to_string(nq[q]) + " records for a batch of " + std::to_string(batches[bq].size()));
                        mgx_raw_store *store = nullptr;
                        mgx_results res{};
                        if (int rc = mgx_results_from_raw_labeled(hdrs[q], nq[q], streams[q], words[q], annotation ? 1 : 0, &store, &res))
                            throw std::runtime_error(std::string("mgx_results_from_raw: ") + mgx_last_error() + " (" + std::to_string(rc) + ")");
                        try {
                            HipDBGAligner::deliver(res, batches[bq], [&](const std::string &header, AlignmentResults &&paths) {
                                std::cout << format_alignment(header, paths, cfg.min_path_score, annotation ? &label_names : nullptr);
                            });
                        } catch (...) { mgx_raw_store_free(store); throw; }
                        mgx_raw_store_free(store);
                    } catch (const std::exception &e) {
                        std::lock_guard<std::mutex> lock(err_mutex);
                        if (first_error.empty()) first_error = e.what();
                    }
                }
            }
        };
        if (rccl_gather) {
            std::vector<int> devs((size_t)devices);
            for (int d = 0; d < devices; ++d) devs[(size_t)d] = d;
            if (int rc = mgx_gather_create_local(devs.data(), devices, 0, gathers.data())) {
                fprintf(stderr, "error: %s (%d)\n", mgx_last_error(), rc);
                return 1;
            }
            threads = (unsigned)devices;
        }
        const auto t_align0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        if (rccl_gather) {
            for (unsigned t = 1; t < threads; ++t) pool.emplace_back(gather_worker, t);
            gather_worker(0);
        } else {
            for (unsigned t = 1; t < threads; ++t) pool.emplace_back(worker, t);
            worker(0);
        }
        for (auto &t : pool) t.join();
        for (mgx_gather *g : gathers) mgx_gather_destroy(g);
        if (report_time) {
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_align0).count();
            fprintf(stderr, "mgx_align: %zu queries in %zu batches, %u worker(s), %.3f s in the align loop (%.0f queries/s)\n",
                    n_queries, batches.size(), threads, sec, sec > 0 ? (double)n_queries / sec : 0.0);
        }
        if (!first_error.empty()) { fprintf(stderr, "error: %s\n", first_error.c_str()); return 1; }
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
