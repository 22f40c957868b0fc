"""GPU parity: libmgx.so (HIP kernels through the C-ABI) against the oracle.  Needs a real MI355X."""
import json
import os
import random

import pytest

import orc
from metagraph_amd import aligner, capi
from test_emu_vs_oracle import make_world, mutate, KATS
from emu_drv import oracle_seeds_as_tuples

pytestmark = pytest.mark.gpu


def gpu_graph(g):
    W, last, F, valid = g.export()
    return aligner.Graph(g.k, W, last, F, valid)


def compare_gpu(g, G, cfg, reads, limits=None, check_seeds=True, pipeline=None):
    o = orc.AlignRun(g, cfg, reads)
    assert o.error == "", o.error
    A = aligner.Aligner(G, cfg, limits)
    if pipeline:
        for name in pipeline.split("+"):
            A.set_pipeline(name)
    A.keep_seeds(check_seeds)
    got, status = A.align_batch(reads)
    assert all(s == 0 for s in status), status
    if check_seeds:
        info = A.seed_info(len(reads))
        for strand in (0, 1):
            for q, (ss, nm) in enumerate(o.seeds(strand)):
                assert info[q]["num_matches"][strand] == nm, (q, strand)
                assert info[q]["seeds"][strand] == oracle_seeds_as_tuples(ss), (q, strand, reads[q])
    want = o.results()
    for q in range(len(reads)):
        assert got[q] == want[q], (q, reads[q], got[q], want[q])
    return A


def test_library_sees_gpu():
    assert capi.lib().mgx_device_count() >= 1


@pytest.mark.parametrize("machine", ["map_pipe=0", "map_pipe=2"])       # one chain step per lane and iteration / request-response
@pytest.mark.parametrize("k,mask", [(11, False), (11, True), (31, False), (5, False), (32, False), (3, False)])
def test_mapping(k, mask, machine):
    g, reads = make_world(10 + k, k, mask=mask, n_reads=300, read_len=150, genome_len=8000)
    reads += ["", "A", reads[0][:k], reads[1][:32], reads[2][:33], reads[3][:64], reads[4][:65], reads[5][:20] + "N" * 40 + reads[5][60:], "N" * 70]
    G = gpu_graph(g)
    cfg = capi.config_cli(k)
    want = orc.AlignRun(g, cfg, reads).mapping()
    A = aligner.Aligner(G, cfg)
    A.set_pipeline(machine)
    got = A.map_batch(reads)
    assert got == want


@pytest.mark.parametrize("k,mask,seed", [(11, False, 1), (19, False, 2), (31, False, 3), (11, True, 4), (31, True, 5)])
def test_align_cli_config(k, mask, seed, kernels):
    g, reads = make_world(100 + seed, k, mask=mask)
    compare_gpu(g, gpu_graph(g), capi.config_cli(k), reads)


PIPELINES = ["split8", "split8+general"]     # "+general": the extension's register-resident chain path off


@pytest.mark.parametrize("pipeline", PIPELINES)
def test_every_pipeline_matches_the_oracle(pipeline, kernels):
    """The product pipeline (seeding kernel, work sort, 8-lane extension kernel) gives the oracle's results with the
    extension's register-resident chain path on and off."""
    g, reads = make_world(500, 31, genome_len=6000, n_reads=150, read_len=150, n_variants=30)
    G = gpu_graph(g)
    A = compare_gpu(g, G, capi.config_cli(31), reads, pipeline=pipeline)
    st = A.stats()
    if pipeline.endswith("general"):
        assert st["n_fast_columns"] == 0
    else:
        assert st["n_fast_columns"] > 0.9 * st["n_columns"], st      # the chain path must be the one that runs
    # ragged batch, fewer reads than lanes, forward only
    g, reads = make_world(501, 13, n_reads=7)
    cfg = capi.config_cli(13)
    cfg.forward_and_reverse_complement = 0
    cfg.min_exact_match = 0.0
    compare_gpu(g, gpu_graph(g), cfg, reads + [reads[0][:40], "ACGT", ""], pipeline=pipeline)


@pytest.mark.parametrize("pipeline", ["split8", "split8+general"])
def test_aligner_reuse_across_batches_of_different_shape(pipeline):
    """One aligner handle, several batches whose longest read differs (the per-slot arena layout changes and is
    re-zeroed only then), interleaved with same-shape batches that reuse the generation-tagged tables."""
    g, reads = make_world(600, 21, genome_len=5000, n_reads=120, read_len=150, n_variants=20)
    G = gpu_graph(g)
    cfg = capi.config_cli(21)
    A = aligner.Aligner(G, cfg)
    for name in pipeline.split("+"):
        A.set_pipeline(name)
    rng = random.Random(5)
    batches = [reads[:40], [r[:90] for r in reads[40:80]], reads[80:120], [r[:60] for r in reads[:30]] + [reads[3]], reads[:40]]
    for b in batches:
        want = orc.AlignRun(g, cfg, b).results()
        got, status = A.align_batch(b)
        assert all(s == 0 for s in status)
        assert got == want
    assert rng is not None


def test_device_results_wrap_as_a_torch_tensor():
    """bench.py's N > 1 path gathers the fixed-size result records over RCCL straight from the aligner's HBM buffer
    (mgx_device_results): the buffer must wrap as a CUDA tensor and hold the same records the host fetch decodes."""
    import ctypes as C
    import numpy as np
    import torch
    g, reads = make_world(700, 21, genome_len=4000, n_reads=64, read_len=120)
    G = gpu_graph(g)
    A = aligner.Aligner(G, capi.config_cli(21))
    got, status = A.align_batch(reads)
    hp, hb, nq, sp, sw = C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_void_p(), C.c_uint64()
    assert capi.lib().mgx_device_results(A.h, C.byref(hp), C.byref(hb), C.byref(nq), C.byref(sp), C.byref(sw)) == 0
    assert nq.value == len(reads) and hb.value >= 48

    class _Ptr:
        __cuda_array_interface__ = {"shape": (hb.value * nq.value,), "typestr": "|u1", "data": (hp.value, False), "version": 2}
    t = torch.as_tensor(_Ptr(), device=torch.device("cuda", 0))
    rec = t.cpu().numpy().reshape(nq.value, hb.value)
    scores = rec[:, 8:12].copy().view(np.int32).ravel()            # ReadResult: status, n_alignments, score, ...
    n_aln = rec[:, 4:8].copy().view(np.int32).ravel()
    for q in range(len(reads)):
        assert n_aln[q] == len(got[q])
        if got[q]:
            assert scores[q] == got[q][0]["score"]


@pytest.mark.parametrize("pipeline", ["split8", "split8+general"])
def test_baseline_config0_transcripts_k12(pipeline):
    """BASELINE.json configs[0] in small: `metagraph align` of the transcripts against their own k = 12 graph
    (tests/data/transcripts_100.fa: 100 queries of 68 .. 5603 bp, every one an exact path).  Long queries put the
    per-read arenas (O(columns x length) convergence vectors) and the ragged-batch handling under load."""
    from test_oracle_kats import read_fasta, HERE
    seqs = read_fasta(os.path.join(HERE, "golden", "transcripts_100.fa"))
    g = orc.Graph.build(12, seqs, 0, False)
    cfg = capi.config_cli(12)
    want = orc.AlignRun(g, cfg, seqs, threads=8).results()
    A = aligner.Aligner(gpu_graph(g), cfg)
    for name in pipeline.split("+"):
        A.set_pipeline(name)
    got, status = A.align_batch(seqs)
    assert all(s == 0 for s in status), status
    assert got == want
    assert sum(1 for r, q in zip(got, seqs) if r and r[0]["cigar"] == "%d=" % len(q)) >= 95


def test_baseline_config0_transcripts_1000_k12():
    """BASELINE.json configs[0] in full: `metagraph align` of tests/data/transcripts_1000.fa (1000 transcripts, 1.49 Mbp,
    up to 11.7 kbp) against its own k = 12 DBGSuccinct graph, CLI defaults — every alignment equal to the oracle's."""
    from test_oracle_kats import read_fasta, HERE
    seqs = read_fasta(os.path.join(HERE, "golden", "transcripts_1000.fa"))
    g = orc.Graph.build(12, seqs, 0, False)
    cfg = capi.config_cli(12)
    want = orc.AlignRun(g, cfg, seqs, threads=os.cpu_count() or 8, validate=False).results()
    got, status = aligner.Aligner(gpu_graph(g), cfg).align_batch(seqs)
    assert all(s == 0 for s in status)
    assert got == want


@pytest.mark.parametrize("min_seed,per_locus", [(15, 1000), (13, 2), (9, 1)])
def test_sub_k_seeding_variants_on_gpu(min_seed, per_locus, kernels):
    """BASELINE configs[4] flavour (`--align-min-seed-length 15` and shorter, per-locus seed cap) on a repetitive
    genome; same construction as the CPU-model test."""
    from test_emu_vs_oracle import rand_seq
    rng = random.Random(77)
    unit = rand_seq(rng, 40)
    genome = rand_seq(rng, 600) + unit + rand_seq(rng, 300) + unit[:30] + rand_seq(rng, 5) + unit[10:] + rand_seq(rng, 600)
    g = orc.Graph.build(31, [genome], 0, False)
    reads = []
    for _ in range(30):
        p = rng.randrange(0, len(genome) - 120)
        reads.append(mutate(rng, genome[p:p + 120], 0.03))
    cfg = capi.config_cli(31)
    cfg.min_seed_length = min_seed
    cfg.max_num_seeds_per_locus = per_locus
    cfg.min_exact_match = 0.0
    compare_gpu(g, gpu_graph(g), cfg, reads)


@pytest.mark.parametrize("k", [15, 31])
def test_reads_with_invalid_and_lower_case_characters_on_gpu(k, kernels):
    from test_emu_vs_oracle import noisy_reads
    g, reads = make_world(520 + k, k, n_reads=40, read_len=120)
    cfg = capi.config_cli(k)
    cfg.min_exact_match = 0.0
    compare_gpu(g, gpu_graph(g), cfg, noisy_reads(5, reads))


def test_unknown_pipeline_is_an_error():
    g, _ = make_world(3, 9, genome_len=300, n_reads=0)
    A = aligner.Aligner(gpu_graph(g), capi.config_cli(9))
    with pytest.raises(aligner.MgxError) as e:
        A.set_pipeline("warp32")
    assert e.value.code == capi.MGX_ERR_INVALID


def test_align_forward_only_and_no_min_exact_match(kernels):
    g, reads = make_world(201, 13, n_reads=40)
    cfg = capi.config_cli(13)
    cfg.forward_and_reverse_complement = 0
    cfg.min_exact_match = 0.0
    compare_gpu(g, gpu_graph(g), cfg, reads)


def test_align_many_reads_k31(kernels):
    rng = random.Random(77)
    g, reads = make_world(1001, 31, genome_len=50000, n_reads=3000, read_len=150, n_variants=200)
    reads = [mutate(rng, r, sub=0.04, ins=0.01, dele=0.01) if i % 3 == 0 else r for i, r in enumerate(reads)]
    reads += ["", "ACGT", "N" * 150, "A" * 150, reads[0][:40]]          # empty / short / invalid / low complexity
    A = compare_gpu(g, gpu_graph(g), capi.config_cli(31), reads, check_seeds=False)
    st = A.stats()
    assert st["n_reads"] == len(reads) and st["n_columns"] > 0


BIG = capi.Limits()
BIG.max_columns = 250000
BIG.max_seeds = 2048


@pytest.mark.parametrize("case", [c for c in KATS["unit"] if not c["expect"].get("throws")], ids=lambda c: c["name"])
def test_reference_kats_on_gpu(case, kernels):
    g = orc.Graph.build(case["k"], case["graph"], 0, case["mask_dummy"])
    G = gpu_graph(g)
    for extend in (False, True):
        cfg = orc.make_config(case["config"], case["matrix"])
        if extend:
            cfg.max_seed_length = capi.UINT64_MAX
        compare_gpu(g, G, cfg, [case["query"]], limits=BIG)


def test_cli_goldens_on_gpu(kernels):
    """metagraph align goldens (integration_tests/test_align.py) byte-for-byte through mgx_format_tsv."""
    from test_oracle_kats import read_fasta, read_fastq, HERE
    cli = KATS["cli"]
    seqs = read_fasta(os.path.join(HERE, "golden", cli["graph_fasta"]))
    g = orc.Graph.build(cli["k"], seqs, 0, False)
    G = gpu_graph(g)
    reads = read_fastq(os.path.join(HERE, "golden", cli["reads_fastq"]))
    for spec in cli["runs"]:
        cfg = capi.config_cli(cli["k"])
        for key, val in spec["flags"].items():
            setattr(cfg, key, val)
        A = aligner.Aligner(G, cfg)
        blob, offs = aligner.pack_queries([r[1] for r in reads])
        import ctypes as C
        res = capi.Results()
        rc = capi.lib().mgx_align_batch(A.h, blob, offs.ctypes.data, len(reads), 0, C.byref(res))
        assert rc == 0
        lines = [A.format_tsv(res, i, reads[i][0], reads[i][1]).rstrip("\n") for i in range(len(reads))]
        for idx, want in spec["lines"].items():
            assert lines[int(idx)] == want
        for idx, fields in spec["fields"].items():
            got = lines[int(idx)].split("\t")
            for fi, fv in fields.items():
                assert got[int(fi)] == fv


@pytest.mark.parametrize("edit_distance,name", [(False, "genome_MT1.align.json"), (True, "genome_MT1.align.edit.json")])
def test_cli_json_golden_node_ids_on_gpu(edit_distance, name, kernels):
    """The reference's JSON goldens (node ids of every alignment; default and edit-distance scoring) straight
    against the HIP path."""
    from test_oracle_kats import read_fasta, read_fastq, HERE, check_against_json_golden, json_golden_config
    cli = KATS["cli"]
    g = orc.Graph.build(cli["k"], read_fasta(os.path.join(HERE, "golden", cli["graph_fasta"])), 0, False)
    reads = read_fastq(os.path.join(HERE, "golden", cli["reads_fastq"]))
    cfg = json_golden_config(cli["k"], edit_distance)
    A = aligner.Aligner(gpu_graph(g), cfg)
    got, status = A.align_batch([r[1] for r in reads])
    assert all(s == 0 for s in status)
    check_against_json_golden(got, reads, name)
    # ... and the --json lines themselves, byte for byte (mgx_format_json)
    import ctypes as C
    blob, offs = aligner.pack_queries([r[1] for r in reads])
    res = capi.Results()
    assert capi.lib().mgx_align_batch(A.h, blob, offs.ctypes.data, len(reads), 0, C.byref(res)) == 0
    want = [line for line in open(os.path.join(HERE, "golden", name)) if line.strip()]
    for i, line in enumerate(want):
        assert capi.format_json(res, i, reads[i][0].lstrip("@").split()[0], reads[i][1], cli["k"]) == line


def test_unsupported_and_bad_config_fail_loudly():
    g, _ = make_world(3, 9, genome_len=300, n_reads=0)
    G = gpu_graph(g)
    cfg = capi.config_cli(9)
    cfg.num_alternative_paths = 5                      # more than MGX_MAX_ALTERNATIVE_PATHS
    with pytest.raises(aligner.MgxError) as e:
        aligner.Aligner(G, cfg)
    assert e.value.code == capi.MGX_ERR_UNSUPPORTED
    cfg = capi.config_default()
    capi.set_dna_matrix(cfg, 2, -1, -2)
    cfg.min_cell_score = -2**31
    with pytest.raises(aligner.MgxError) as e:
        aligner.Aligner(G, cfg)
    assert e.value.code == capi.MGX_ERR_CONFIG      # the reference throws std::runtime_error (dbg_aligner.cpp:55-56)


def test_small_cell_budget_gives_capacity_statuses_not_wrong_answers():
    """mgx_limits.cell_arena_bytes sizes the S / F records of the general path and the pool of the convergence vectors: with a
    budget far too small (round 3: chain columns no longer use either, so it takes a budget below one general-path column),
    reads either come out exactly as the oracle's or carry MGX_ERR_CAPACITY — never a truncated alignment — and the same
    reads succeed under the default limits (what the adapter's retry relies on)."""
    g, reads = make_world(4242, 21, genome_len=6000, n_reads=200, read_len=150)
    cfg = capi.config_cli(21)
    want = orc.AlignRun(g, cfg, reads).results()
    G = gpu_graph(g)
    lim = capi.Limits()
    lim.cell_arena_bytes = 1600
    A = aligner.Aligner(G, cfg, lim)
    A.set_pipeline("retry_capacity=0")                      # (statuses as the kernels report them)
    got, status = A.align_batch(reads)
    assert any(s == capi.MGX_ERR_CAPACITY for s in status)
    for q, s in enumerate(status):
        assert s in (0, capi.MGX_ERR_CAPACITY)
        if s == 0:
            assert got[q] == want[q]
    got, status = aligner.Aligner(G, cfg).align_batch(reads)
    assert all(s == 0 for s in status) and got == want
    # ... and mgx_align_batch re-aligns them itself with doubled limits (round 5: for every caller, not only the C++ adapter)
    B = aligner.Aligner(G, cfg, lim)
    got, status = B.align_batch(reads)
    assert all(s == 0 for s in status) and got == want
    assert B.stats()["n_capacity_retried"] > 0
    # ... also for reads that were in HBM already (mgx_align_batch_device + mgx_fetch_results: the retry reads them back)
    import numpy as np
    import torch
    blob, offs = aligner.pack_queries(reads)
    d_seqs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    d_offs = torch.from_numpy(np.asarray(offs, dtype=np.int64)).cuda()
    D = aligner.Aligner(G, cfg, lim)
    D.align_device(d_seqs.data_ptr(), d_offs.data_ptr(), len(reads))
    res = D.fetch()
    assert all(res.status[i] == 0 for i in range(len(reads))) and capi.results_to_py(res) == want
    assert D.stats()["n_capacity_retried"] > 0


@pytest.mark.parametrize("per_wave", [0, 1, 3, 8])
def test_reads_per_wavefront_do_not_change_results(per_wave):
    """A batch with fewer reads than resident groups is spread over the wavefronts (AlignParams::groups_per_wave, chosen by the
    host from the batch size); the option groups_per_wave forces 8 (= 0), 1, 3 or 8 groups of a wavefront to take reads, with
    1 on the 8-lane kernel (ext64=0) and on the 64-lane one.  Scheduling only: the oracle's alignments either way."""
    g, reads = make_world(515, 21, genome_len=6000, n_reads=300, read_len=120)
    for ext64 in ((0, 1) if per_wave == 1 else (0,)):
        opts = "split8+ext64=%d+lane=0+groups_per_wave=%d" % (ext64, per_wave)
        A = compare_gpu(g, gpu_graph(g), capi.config_cli(21), reads, pipeline=opts)
        assert A.stats()["extend_kernels"] == (capi.KERNEL_EXT64 if ext64 and per_wave == 1 else capi.KERNEL_GRP8)
        cfg = capi.config_cli(21)
        cfg.min_seed_length = 13
        cfg.min_exact_match = 0.0
        compare_gpu(g, gpu_graph(g), cfg, reads[:120], check_seeds=False, pipeline=opts)


def test_probe_switches_are_not_in_the_product_library(monkeypatch):
    """The measurement probes of earlier rounds (MGX_ABLATE: timing ablations with WRONG results; occupancy / LDS caps) exist
    in -DMGX_PROBES builds only: the product library must ignore their environment variables."""
    g, reads = make_world(516, 21, genome_len=5000, n_reads=200, read_len=120)
    for var, val in (("MGX_ABLATE", "15"), ("MGX_EXT_GROUPS_PCT", "1"), ("MGX_NO_FAST", "1"), ("MGX_EXT64", "0"), ("MGX_SEED_LDS_CAP", "16")):
        monkeypatch.setenv(var, val)
    A = compare_gpu(g, gpu_graph(g), capi.config_cli(21), reads)
    st = A.stats()
    assert st["n_fast_columns"] > 0 and st["extend_kernels"] == capi.KERNEL_EXT64


@pytest.mark.parametrize("xdrop", [100, 300])
def test_wide_bands_on_single_read_batches(xdrop, kernels):
    """Large x-drop and indel-rich long reads make DP bands wider than a chain column's slot (32 cells); a one-read batch
    runs on the 64-lane kernel, whose register window is 256 cells wide but whose column slots are the 8-lane layout's:
    such columns must take the general path there too (round 3 wrote them past their slots)."""
    rng = random.Random(9 + xdrop)
    g, reads = make_world(910, 21, genome_len=9000, n_reads=12, read_len=600, n_variants=20)
    cfg = capi.config_cli(21)
    cfg.xdrop = xdrop
    cfg.min_exact_match = 0.0
    G = gpu_graph(g)
    for r in reads:
        compare_gpu(g, G, cfg, [mutate(rng, r, sub=0.03, ins=0.02, dele=0.02)], limits=BIG, check_seeds=False)
