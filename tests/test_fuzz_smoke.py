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
