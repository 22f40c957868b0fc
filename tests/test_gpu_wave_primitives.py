"""The DPP wave primitives of metagraph_amd/csrc/wave.hpp (prefix max, max, min, shift, broadcast) against a
scalar loop, on the GPU (standalone HIP program built by __graft_entry__.build())."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dpp_primitives():
    exe = os.path.join(ROOT, "metagraph_amd", "_build", "dpp_test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


def test_group_primitives():
    """wave_group.hpp (8 lanes per read): DPP butterfly reductions, masked-OR broadcast, prefix max, shifts, ballots; groups of
    one wavefront hold different data and some are masked off."""
    exe = os.path.join(ROOT, "metagraph_amd", "_build", "dpp_group_test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
