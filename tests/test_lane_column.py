"""lane_column() (metagraph_amd/csrc/lane_column.hpp: one column of a register chain computed by ONE lane, the core of the
lane-per-read chain kernel planned in DESIGN.md §9.1) against chain_step(), the 8-lanes-per-read implementation the product
runs: a build of the host model with -DMGX_LANE_CHECK calls it on the inputs of every chain step and aborts on the first
difference in return code, geometry, scan results, cells or flag bytes.  Runs the alignment checks of test_emu_vs_oracle in a
subprocess (the checked library is a compile-time variant), so the columns compared are those of real extensions: short reads,
sub-k seeds, x-drop off, alternative scoring."""
import os
import subprocess
import sys

import pytest


@pytest.mark.parametrize("selection", ["test_align_cli_config or test_align_forward_only or test_sub_k_seeding_variants",
                                       "test_split_pipeline or test_lds_placements or noisy"])
def test_lane_column_matches_chain_step(selection):
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MGX_EMU_LANE_CHECK="1", MGX_EMU_WAVE="8")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(here, "test_emu_vs_oracle.py"), "-k", selection,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2500:]
    assert "passed" in r.stdout
