"""The device's sdust (exact algorithm + lane-parallel pre-filter inside the seeding kernel) against the definition-level
checker, on the GPU: with DBGAlignerConfig{} defaults the seeder emits one seed per matching k-mer and drops it when
sdust masks it (aligner_seeder_methods.cpp:84), so for reads that ARE one k-mer of the graph the forward strand has a
seed iff the string is not low-complexity."""
import random

import pytest

import orc
from metagraph_amd import aligner, capi

pytestmark = pytest.mark.gpu


def _strings(seed, n, L):
    rng = random.Random(seed)
    out = set()
    while len(out) < n:
        kind = rng.randrange(5)
        if kind == 0:
            s = "".join(rng.choice("ACGT") for _ in range(L))
        elif kind == 1:
            u = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 9)))
            s = list((u * (L // len(u) + 1))[:L])
            for _ in range(rng.randint(0, 6)):
                s[rng.randrange(L)] = rng.choice("ACGT")
            s = "".join(s)
        elif kind == 2:
            s = list("".join(rng.choice("ACGT") for _ in range(L)))
            a = rng.randrange(L)
            u = rng.choice(["A", "C", "G", "T", "AT", "CG", "AC", "GT", "AAT", "CAG"])
            for x in range(a, min(L, a + rng.randint(4, 30))):
                s[x] = u[(x - a) % len(u)]
            s = "".join(s)
        elif kind == 3:
            wts = [rng.random() ** 3 for _ in range(4)]
            s = "".join(rng.choices("ACGT", weights=wts, k=L))
        else:
            alpha = rng.sample(["ACG", "CGT", "GTA", "TAC", "AAC", "ACA", "CAA", "GGT", "TTG"], rng.randint(2, 5))
            s = "".join(rng.choice(alpha) for _ in range(L // 3 + 1))[:L]
        out.add(s)
    return sorted(out)


@pytest.mark.parametrize("k,n", [(40, 12000), (62, 6000), (21, 12000)])
def test_device_sdust_matches_the_definition(k, n):
    reads = _strings(1000 + k, n, k)
    g = orc.Graph.build(k, reads, 0, False)
    W, last, F, valid = g.export()
    G = aligner.Graph(k, W, last, F, valid)
    cfg = capi.config_default()
    capi.set_dna_matrix(cfg, 2, -1, -2)
    A = aligner.Aligner(G, cfg)
    A.keep_seeds(True)
    A.align_batch(reads)
    info = A.seed_info(len(reads))
    L = orc.L()
    n_low = 0
    for q, s in enumerate(reads):
        b = s.encode()
        low = bool(L.orc_sdust_bruteforce(b, len(b)))
        n_low += low
        assert (len(info[q]["seeds"][0]) == 0) == low, (s, info[q]["seeds"][0])
    assert len(reads) // 20 < n_low < len(reads) * 19 // 20
