"""Kernel logic (wave programs under the host model) against the oracle.  CPU only."""
import json
import os
import random

import pytest

import emu_drv
import orc
from metagraph_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "aligner_kats.json")))


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(rng, s, sub=0.03, ins=0.005, dele=0.005):
    out = []
    for ch in s:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            out.append(rng.choice([c for c in "ACGT" if c != ch]))
        else:
            out.append(ch)
        if rng.random() < ins:
            out.append(rng.choice("ACGT"))
    return "".join(out)


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def make_world(seed, k, genome_len=3000, n_reads=40, read_len=100, mask=False, n_variants=10):
    rng = random.Random(seed)
    genome = rand_seq(rng, genome_len)
    seqs = [genome]
    for _ in range(n_variants):
        p = rng.randrange(k, genome_len - k)
        alt = rng.choice([c for c in "ACGT" if c != genome[p]])
        seqs.append(genome[p - k + 1:p] + alt + genome[p + 1:p + k])
    g = orc.Graph.build(k, seqs, 0, mask)
    reads = []
    for i in range(n_reads):
        if i % 10 == 9:
            reads.append(rand_seq(rng, read_len))
            continue
        p = rng.randrange(0, genome_len - read_len)
        r = mutate(rng, genome[p:p + read_len])
        if rng.random() < 0.5:
            r = rc(r)
        if i % 7 == 3:
            r = r[:len(r) // 2] + "N" + r[len(r) // 2 + 1:]
        reads.append(r)
    return g, reads


def test_graph_primitives():
    g, _ = make_world(1, 9, genome_len=600, n_reads=0)
    eg = emu_drv.EmuGraph(g)
    Lo, Le = orc.L(), emu_drv.L()
    W, last, F, _ = g.export()
    n = g.n_edges
    for v in range(1, n + 1):
        assert Le.emu_bwd(eg.h, v) == Lo.orc_boss_bwd(g.h, v), v
        c = int(W[v]) % 5
        if c or v == 1:
            assert Le.emu_fwd(eg.h, v, c) == Lo.orc_boss_fwd(g.h, v, c), v
        assert eg.outgoing(v) == [(a, b) for a, b in g.outgoing(v) if b != "$"], v
        want_in = [(a, b) for a, b in g.outgoing(v, rc=True)]
        got_in = eg.outgoing(v, rc=True)
        assert got_in == want_in, (v, got_in, want_in)
        seq = g.node_sequence(v)
        assert "$ACGT"[Le.emu_first_char(eg.h, v)] == seq[0], (v, seq)
        term = bool(Lo.orc_graph_has_multiple_outgoing(g.h, v)) or not bool(Lo.orc_graph_has_single_incoming(g.h, v))
        assert bool(Le.emu_terminus(eg.h, v)) == term, v


def test_graph_primitives_masked():
    g, _ = make_world(2, 7, genome_len=400, n_reads=0, mask=True)
    eg = emu_drv.EmuGraph(g)
    Lo, Le = orc.L(), emu_drv.L()
    W, last, F, valid = g.export()
    for v in range(1, g.n_edges + 1):
        if not valid[v]:
            continue
        assert eg.outgoing(v) == [(a, b) for a, b in g.outgoing(v) if b != "$"], v
        assert eg.outgoing(v, rc=True) == g.outgoing(v, rc=True), v
        term = bool(Lo.orc_graph_has_multiple_outgoing(g.h, v)) or not bool(Lo.orc_graph_has_single_incoming(g.h, v))
        assert bool(Le.emu_terminus(eg.h, v)) == term, v


def test_sdust_matches_oracle():
    rng = random.Random(5)
    Lo, Le = orc.L(), emu_drv.L()
    for _ in range(400):
        n = rng.randrange(3, 64)
        mode = rng.random()
        if mode < 0.3:
            s = rand_seq(rng, n)
        elif mode < 0.6:
            unit = rand_seq(rng, rng.randrange(1, 4))
            s = (unit * 40)[:n]
        else:
            s = "".join(rng.choice("AAAT") for _ in range(n))
        if rng.random() < 0.2:
            s = s[:n // 2] + "N" + s[n // 2 + 1:]
        b = s.encode()
        assert bool(Le.emu_is_low_complexity(b, len(b))) == bool(Lo.orc_is_low_complexity(b, len(b))), s


@pytest.mark.parametrize("k,mask", [(11, False), (11, True), (31, False), (5, False)])
def test_mapping(k, mask):
    g, reads = make_world(10 + k, k, mask=mask)
    eg = emu_drv.EmuGraph(g)
    cfg = capi.config_cli(k)
    want = orc.AlignRun(g, cfg, reads).mapping()
    got = emu_drv.EmuRun(eg, cfg, reads, map_only=True).mapping()
    assert got == want


def compare_full(g, eg, cfg, reads, limits=None):
    o = orc.AlignRun(g, cfg, reads)
    assert o.error == "", o.error
    e = emu_drv.EmuRun(eg, cfg, reads, limits=limits)
    assert e.error == "", e.error
    got, status = e.results()
    assert all(s == 0 for s in status), status
    info = e.seed_info()
    for strand in (0, 1):
        for q, (ss, nm) in enumerate(o.seeds(strand)):
            assert info[q]["num_matches"][strand] == nm, (q, strand, "num_matches")
            assert info[q]["seeds"][strand] == emu_drv.oracle_seeds_as_tuples(ss), (q, strand, reads[q])
    want = o.results()
    for q in range(len(reads)):
        assert got[q] == want[q], (q, reads[q], got[q], want[q])
    return e


@pytest.mark.parametrize("k,mask,seed", [(11, False, 1), (19, False, 2), (31, False, 3), (11, True, 4), (31, True, 5)])
def test_align_cli_config(k, mask, seed):
    g, reads = make_world(100 + seed, k, mask=mask)
    eg = emu_drv.EmuGraph(g)
    cfg = capi.config_cli(k)
    compare_full(g, eg, cfg, reads)


@pytest.mark.parametrize("k,mask,seed", [(11, False, 21), (31, False, 22), (19, True, 23)])
def test_split_pipeline(k, mask, seed, monkeypatch):
    """Seeding kernel -> sort by predicted work -> extension kernel (the product's default pipeline) gives
    the results of the fused per-read program: seeds survive the hand-over and the processing order is free."""
    monkeypatch.setenv("MGX_EMU_SPLIT", "1")
    g, reads = make_world(300 + seed, k, mask=mask, n_reads=50)
    eg = emu_drv.EmuGraph(g)
    cfg = capi.config_cli(k)
    compare_full(g, eg, cfg, reads)


def test_two_pass_extension_retries_multi_seed_reads(monkeypatch):
    """Pass 1 of the split pipeline extends at most one seed per read; reads that go on to another seed are re-run
    from scratch in pass 2.  A world with chimeric reads (two distant genome pieces glued together) makes sure the
    second pass is actually taken, and the results still equal the oracle's."""
    monkeypatch.setenv("MGX_EMU_SPLIT", "1")
    g, reads = make_world(333, 15, genome_len=4000, n_reads=30, read_len=100)
    rng = random.Random(9)
    chim = [reads[i][:50] + reads[i + 1][40:90] for i in range(0, 20, 2)]
    eg = emu_drv.EmuGraph(g)
    cfg = capi.config_cli(15)
    cfg.min_exact_match = 0.0
    e = compare_full(g, eg, cfg, reads + chim)
    assert e.retried() > 0
    assert rng is not None


@pytest.mark.parametrize("cap", ["", "2"])
def test_multi_pass_extension_with_resume_records(cap, monkeypatch):
    """The product's answer to reads with many seeds: a pass extends one seed per read, what outlives a seed (aggregator
    queue, live-seed flags, extender capacities, counters, the place where the read stopped) travels to the next pass in a
    resume record, and the positions are re-sorted by the work of their next seed.  Chimeric and multi-seed reads take up
    to a dozen passes here; with room for only two records per pass the other reads simply go on without a limit.  The
    results equal the oracle's either way."""
    monkeypatch.setenv("MGX_EMU_SPLIT", "1")
    monkeypatch.setenv("MGX_EMU_MULTIPASS", "1")
    if cap:
        monkeypatch.setenv("MGX_EMU_RESUME_CAP", cap)
    g, reads = make_world(334, 15, genome_len=4000, n_reads=40, read_len=100, n_variants=60)
    chim = [reads[i][:50] + reads[i + 1][40:90] for i in range(0, 30, 2)]
    eg = emu_drv.EmuGraph(g)
    for n_alt, min_seed in ((1, 15), (2, 15), (1, 11)):
        cfg = capi.config_cli(15)
        cfg.min_exact_match = 0.0
        cfg.num_alternative_paths = n_alt
        cfg.min_seed_length = min_seed
        e = compare_full(g, eg, cfg, reads + chim)
        assert e.retried() > 0


def test_split_pipeline_unit_kats(monkeypatch):
    monkeypatch.setenv("MGX_EMU_SPLIT", "1")
    for case in KATS["unit"]:
        if case["expect"].get("throws"):
            continue
        g = orc.Graph.build(case["k"], case["graph"], 0, case["mask_dummy"])
        eg = emu_drv.EmuGraph(g)
        cfg = orc.make_config(case["config"], case["matrix"])
        compare_full(g, eg, cfg, [case["query"]], limits=BIG)


@pytest.mark.parametrize("lds_bytes", ["0", "700", "1400", "16384"])
def test_lds_placements(lds_bytes, monkeypatch):
    """carve() puts the latency-critical arrays LDS-first and the two-tier staging columns keep their first st_cap
    cells there: every split between LDS and the arena (none, partial tiers, everything) gives the same results."""
    monkeypatch.setenv("MGX_EMU_LDS", lds_bytes)
    g, reads = make_world(410, 15, n_reads=40)
    compare_full(g, emu_drv.EmuGraph(g), capi.config_cli(15), reads)
    cfg = capi.config_default()                       # x-drop off: full-width columns reach far into the arena tier
    capi.set_dna_matrix(cfg, 2, -1, -2)
    compare_full(g, emu_drv.EmuGraph(g), cfg, reads[:6], limits=BIG)


@pytest.mark.parametrize("lanes", ["16", "8"])
def test_group_sizes_in_emulation(lanes):
    """The wave programs do not depend on the lane count: the same sources modelled with 16 and 8 lanes per read
    (the sub-wave-group kernels) pass the alignment and split-pipeline checks.  Runs in a subprocess because the
    model's lane count is a compile-time constant of the loaded library."""
    import subprocess
    import sys
    env = dict(os.environ, MGX_EMU_WAVE=lanes)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.abspath(__file__), "-k",
                        "test_align_cli_config or test_split_pipeline or test_align_forward_only",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("min_seed,per_locus", [(15, 1000), (13, 2), (9, 1)])
def test_sub_k_seeding_variants(min_seed, per_locus):
    """BASELINE configs[4] flavour: `--align-min-seed-length 15` (and shorter), with the per-locus seed cap
    (aligner_seeder_methods.cpp:340) biting on a repetitive genome where suffix ranges hold several nodes."""
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
    compare_full(g, emu_drv.EmuGraph(g), cfg, reads)


def noisy_reads(seed, g_reads):
    """reads with N's, lower case and a few non-nucleotide bytes sprinkled in (AlignmentResults upper-cases and keeps
    the rest; KmerExtractorBOSS::encode maps everything else to the invalid code)"""
    rng = random.Random(seed)
    out = []
    for r in g_reads:
        r = list(r)
        for _ in range(rng.randrange(0, 4)):
            r[rng.randrange(len(r))] = rng.choice("NNNnXx-")
        if rng.random() < 0.5:
            a = rng.randrange(len(r))
            b = min(len(r), a + rng.randrange(1, 30))
            r[a:b] = [c.lower() for c in r[a:b]]
        out.append("".join(r))
    return out


@pytest.mark.parametrize("k", [15, 31])
def test_reads_with_invalid_and_lower_case_characters(k):
    g, reads = make_world(520 + k, k, n_reads=40, read_len=120)
    cfg = capi.config_cli(k)
    cfg.min_exact_match = 0.0
    compare_full(g, emu_drv.EmuGraph(g), cfg, noisy_reads(5, reads))


def test_align_cli_config_no_min_exact_match():
    g, reads = make_world(200, 15, n_reads=60)
    eg = emu_drv.EmuGraph(g)
    cfg = capi.config_cli(15)
    cfg.min_exact_match = 0.0
    compare_full(g, eg, cfg, reads)


def test_align_forward_only():
    g, reads = make_world(201, 13, n_reads=40)
    eg = emu_drv.EmuGraph(g)
    cfg = capi.config_cli(13)
    cfg.forward_and_reverse_complement = 0
    cfg.min_exact_match = 0.0
    compare_full(g, eg, cfg, reads)


BIG = capi.Limits()
BIG.max_columns = 250000
BIG.max_seeds = 2048


@pytest.mark.parametrize("case", [c for c in KATS["unit"] if not c["expect"].get("throws")], ids=lambda c: c["name"])
def test_unit_kats_through_kernels(case):
    g = orc.Graph.build(case["k"], case["graph"], 0, case["mask_dummy"])
    eg = emu_drv.EmuGraph(g)
    for extend in (False, True):
        cfg = orc.make_config(case["config"], case["matrix"])
        if extend:
            cfg.max_seed_length = capi.UINT64_MAX
        compare_full(g, eg, cfg, [case["query"]], limits=BIG)


def test_sdust_whole_read_shortcut_is_sound():
    """window_low_complexity() skips per-window sdust when the whole strand masks nothing: check the
    implication (whole clean => every window clean) on the oracle's sdust over many repeat-rich strings."""
    rng = random.Random(11)
    Lo = orc.L()
    n_clean = 0
    for t in range(600):
        mode = rng.random()
        if mode < 0.4:
            s = rand_seq(rng, 150)
        elif mode < 0.7:      # short tandem repeats embedded in random sequence
            unit = rand_seq(rng, rng.randrange(1, 7))
            rep = (unit * 30)[:rng.randrange(6, 26)]
            p = rng.randrange(0, 120)
            s = rand_seq(rng, p) + rep + rand_seq(rng, 150 - p - len(rep))
        else:                 # biased composition
            s = "".join(rng.choice("AAAAACGT") for _ in range(150))
        if rng.random() < 0.2:
            s = s[:70] + "N" + s[71:]
        b = s.encode()
        if Lo.orc_is_low_complexity(b, len(b)):
            continue
        n_clean += 1
        for wl in (19, 25, 31):
            for i in range(0, len(s) - wl + 1):
                assert not Lo.orc_is_low_complexity(b[i:i + wl], wl), (s, i, wl)
    assert n_clean > 100


def test_lane_parallel_dust_filter_is_conservative():
    """maybe_low_complexity() must never say "clean" for a strand in which sdust masks anything — on the whole strand or
    on any window the seeder may test.  Checked against the oracle's sdust on random, repeat-rich, biased and
    N-containing strings; the filter should also be tight (it rarely fires when sdust finds nothing)."""
    rng = random.Random(23)
    Lo, Le = orc.L(), emu_drv.L()
    fired = exact = loose = 0
    for t in range(1500):
        mode = rng.random()
        n = rng.choice([30, 64, 100, 150, 151, 250])
        if mode < 0.35:
            s = rand_seq(rng, n)
        elif mode < 0.7:
            unit = rand_seq(rng, rng.randrange(1, 9))
            rep = (unit * 40)[:rng.randrange(5, 40)]
            p = rng.randrange(0, max(1, n - len(rep)))
            s = (rand_seq(rng, p) + rep + rand_seq(rng, n))[:n]
        elif mode < 0.85:
            s = "".join(rng.choice("AAAAACGT") for _ in range(n))
        else:
            s = "".join(rng.choice("ACACACGT") for _ in range(n))
        if rng.random() < 0.25:
            p = rng.randrange(0, n)
            s = s[:p] + "N" + s[p + 1:]
        b = s.encode()
        maybe = bool(Le.emu_maybe_low_complexity(b, len(b)))
        whole = bool(Lo.orc_is_low_complexity(b, len(b)))
        fired += maybe
        exact += whole
        if "N" not in s:
            loose += maybe and not whole           # (strands with N are always handed to the exact algorithm)
        if not maybe:
            assert not whole, s
            for wl in (12, 19, 31):
                for i in range(0, len(s) - wl + 1, 3):
                    assert not Lo.orc_is_low_complexity(b[i:i + wl], wl), (s, i, wl)
    assert exact > 100 and fired < 1400
    assert loose <= 25, (fired, exact, loose)
