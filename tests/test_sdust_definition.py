"""sdust (github.com/lh3/sdust; third-party code absent from the reference tree, called at
graph/alignment/aligner_seeder_methods.cpp:22-29 with T = 20, W = 64) pinned at definition level.

`orc_sdust_bruteforce` enumerates every triplet interval and applies the score definition of the sdust paper — none of the
incremental machinery (sliding window, running pair counts, perfect-interval list, the L-suffix shortcut) that the
oracle's `is_low_complexity`, the kernels' exact sdust and their lane-parallel conservative pre-filter restate.  All of
them must agree with it on random, tandem-repeat, homopolymer and near-threshold strings, incl. non-ACGT characters."""
import os
import random

import pytest

import orc
import emu_drv

N_RANDOM = int(os.environ.get("MGX_SDUST_CASES", "60000"))      # (120 000 in the long runs: MGX_SDUST_CASES)


def gen_strings(seed, n):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        kind = i % 8
        L = rng.randint(3, 150)
        if kind == 0:                                   # iid
            s = "".join(rng.choice("ACGT") for _ in range(L))
        elif kind == 1:                                 # tandem repeat of a short unit with point mutations
            u = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 7)))
            s = list((u * (L // len(u) + 1))[:L])
            for _ in range(rng.randint(0, max(1, L // 6))):
                s[rng.randrange(L)] = rng.choice("ACGT")
            s = "".join(s)
        elif kind == 2:                                 # homopolymer / dinucleotide island in a random background
            s = list("".join(rng.choice("ACGT") for _ in range(L)))
            a = rng.randrange(L)
            b = min(L, a + rng.randint(4, 40))
            u = rng.choice(["A", "C", "G", "T", "AT", "CG", "AC", "GT", "AAT", "CAG"])
            for x in range(a, b):
                s[x] = u[(x - a) % len(u)]
            s = "".join(s)
        elif kind == 3:                                 # skewed composition
            wts = [rng.random() ** 3 for _ in range(4)]
            s = "".join(rng.choices("ACGT", weights=wts, k=L))
        elif kind == 4:                                 # near the threshold: few distinct triplets
            alpha = rng.sample(["ACG", "CGT", "GTA", "TAC", "AAC", "ACA", "CAA", "GGT", "TTG"], rng.randint(2, 5))
            s = "".join(rng.choice(alpha) for _ in range(L // 3 + 1))[:L]
        elif kind == 5:                                 # with non-ACGT characters (they break triplet runs)
            s = list("".join(rng.choice("ACGT") for _ in range(L)) if rng.random() < 0.5 else (rng.choice("ACGT") * L))
            for _ in range(rng.randint(1, 4)):
                s[rng.randrange(L)] = rng.choice("N$Xn-")
            s = "".join(s)
        elif kind == 6:                                 # lower case (seq_nt4_table accepts both cases)
            s = "".join(rng.choice("acgtACGT") for _ in range(L))
            if rng.random() < 0.5:
                s = s[: L // 2] + s[: L // 2]
        else:                                           # long exact windows of the seeding code: 19..64 bp
            L = rng.randint(19, 64)
            u = "".join(rng.choice("ACGT") for _ in range(rng.randint(2, 12)))
            s = (u * 40)[:L]
        out.append(s)
    return out


def test_oracle_sdust_matches_the_definition():
    L = orc.L()
    bad = []
    n_low = 0
    for s in gen_strings(11, N_RANDOM):
        b = s.encode()
        want = L.orc_sdust_bruteforce(b, len(b))
        n_low += want
        if bool(L.orc_is_low_complexity(b, len(b))) != bool(want):
            bad.append(s)
    assert not bad, bad[:5]
    assert N_RANDOM // 10 < n_low < N_RANDOM * 9 // 10          # both verdicts well represented


@pytest.mark.parametrize("lanes", ["", "16", "8"])
def test_kernel_sdust_and_prefilter_match_the_definition(lanes, monkeypatch):
    """The kernels' exact sdust, and the conservative lane-parallel pre-filter that lets them skip it (it may say
    'maybe' for a clean string, never 'clean' for a masked one), on the host model at 64 / 16 / 8 lanes."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import orc, emu_drv
        from test_sdust_definition import gen_strings
        L, E = orc.L(), emu_drv.L()
        n = 0
        for s in gen_strings(12, 40000):
            b = s.encode()
            want = bool(L.orc_sdust_bruteforce(b, len(b)))
            assert bool(E.emu_is_low_complexity(b, len(b))) == want, s
            if want: assert E.emu_maybe_low_complexity(b, len(b)), s
            n += want
        print("OK", n)
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGX_EMU_WAVE=lanes)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.startswith("OK")
