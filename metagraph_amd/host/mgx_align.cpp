// mgx_align — minimal `metagraph align`-shaped driver over HipDBGAligner (host side only: batching and
// TSV printing as in cli/align.cpp:403-480).  Graph input is a flat BOSS dump (k, n_edges, F[5], W[], last[]);
// reading sdsl-serialised .dbg files is "next" (SURVEY 8f rank 2).  Usage:
//   mgx_align GRAPH.boss READS.{fa,fq} [--align-only-forwards] [--align-min-exact-match X] [--align-min-seed-length N]
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

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

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s GRAPH.boss READS [options]\n", argv[0]); return 2; }
    std::ifstream gin(argv[1], std::ios::binary);
    if (!gin) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    uint64_t hdr[7];
    gin.read((char *)hdr, sizeof(hdr));           // k, n_edges, F[0..4]
    uint32_t k = (uint32_t)hdr[0];
    uint64_t n = hdr[1];
    std::vector<uint8_t> W(n + 1), last(n + 1);
    gin.read((char *)W.data(), n + 1);
    gin.read((char *)last.data(), n + 1);
    DBGAlignerConfig cfg;
    mgx_config_init_cli(&cfg, k);
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--align-only-forwards")) cfg.forward_and_reverse_complement = 0;
        else if (!strcmp(argv[i], "--align-min-exact-match") && i + 1 < argc) cfg.min_exact_match = atof(argv[++i]);
        else if (!strcmp(argv[i], "--align-min-seed-length") && i + 1 < argc) cfg.min_seed_length = std::min<uint64_t>(atoi(argv[++i]), k);
    }
    try {
        HipBOSSGraph graph(k, n, W.data(), last.data(), hdr + 2);
        HipDBGAligner aligner(graph, cfg);
        std::vector<IDBGAligner::Query> batch;
        if (!read_records(argv[2], &batch)) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        aligner.align_batch(batch, [&](const std::string &header, AlignmentResults &&paths) {
            std::cout << format_alignment(header, paths, cfg.min_path_score);
        });
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
