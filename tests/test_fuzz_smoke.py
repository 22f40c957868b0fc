"""A short, seeded run of tools/fuzz_emu.py (random graphs in all three modes x random aligner configurations x random reads:
the kernels' host model against the oracle) so that the fuzzer itself stays runnable; longer campaigns are run by hand."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzzer_runs_clean_for_ten_seconds():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_emu.py"), "--minutes", "0.15", "--seed", "3"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "no difference" in r.stdout


def test_deferred_chain_record_stays_inside_the_cell_arena():
    """World 504 of the lane campaign's seed 9404 (k = 7, reads of 7 to 154 bp, small arenas): a chain column that stays behind
    in the frontier writes a whole chain window's S / F record, and chain_step tested the cell arena for the smaller, window-sized
    record only — near a read's end the record ran into the first column slots (found under -fsanitize=address as a wild read in
    the trace walk).  The host model aborts on a record past the arena, whatever the arena's layout."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_emu.py"), "--seed", "9404", "--lane", "--start", "504", "--worlds", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "no difference" in r.stdout


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--labels"]], ids=["plain", "labels"])
def test_gpu_fuzzer_runs_clean_for_twenty_seconds(extra):
    """tools/fuzz_gpu_lane.py: the built library's lane kernels (forced) on random worlds against the oracle — a short seeded run, so
    that the tool stays runnable and every GPU suite run sees worlds no other test holds"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu_lane.py"), "0.3", "77000"] + extra,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "no difference" in r.stdout
