"""BASELINE.json configs[4] (pan-genome-scale graph, `--align-min-seed-length 15`) at a reduced size as a GPU test: the
scale tool (tools/scale_test.py: divergent strains of one genome -> BOSS on the device -> 1 GPU batch with sub-k seeding ->
a sample against the oracle) with 6 strains of 4 Mbp, 120 k reads and a 4 k-read oracle sample.  What it exercises that the
other GPU tests do not: ~10 sub-k seeds and several extensions per read at this size (94 seeds / 76 extensions at 24 strains x 49 Mbp,
profiles/r02_scale_config5_1gpu.json), the per-locus seed cap, the seed- and output-stream re-runs.  Needs torch (graph construction)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scaled_pan_genome_with_sub_k_seeding_through_torch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_test.py"), "--strains", "6", "--genome", "4000000",
                        "--reads", "120000", "--sample", "4000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["parity"]["mismatches"] == 0, out["parity"]
    assert out["parity"]["capacity_errors"] == 0 and out["capacity_errors"] == 0
    assert out["seeds_per_read"] > 5 and out["extensions_per_read"] > 1.5        # several sub-k seeds and extensions per read
