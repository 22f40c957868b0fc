"""bench.py end to end at reduced size on one GPU, launched the way the driver launches N > 1 (torch.distributed.run,
backend nccl = RCCL): the synthetic workload generator (metagraph_amd/synth.py), the device-resident step, the
complete-alignment gather through RCCL (world size 1 executes mgx_device_results -> all_gather / gather), the
host-inclusive variant, and the oracle comparison of every read of the CPU sample."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_step_under_torchrun_nccl_single_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGX_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--reads", "30000", "--genome", "300000", "--snps", "600", "--cpu-sample", "30000", "--cpu-1t-sample", "300",
           "--host-steps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["value_device_resident"] > 0 and out["value_is"].startswith("host-inclusive")
    assert out["parity"]["sample"] == 30000 and out["parity"]["mismatches"] == 0 and out["parity"]["capacity_errors"] == 0
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"]["single_thread"]["value"] > 0


@pytest.mark.gpu
def test_bench_config3_labels_at_reduced_size():
    """BASELINE config 3 through bench.py: the annotation built on the device from the genome's mapped nodes
    (mgx_map_batch -> mgx_annotation_create_sparse), the label-aware aligner on the 8-reads-per-wavefront labeled kernel,
    alignments AND label lists of the sample against the oracle's LabeledAligner"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--labels", "40", "--steps", "1", "--warmup", "1",
           "--reads", "40000", "--genome", "400000", "--snps", "800", "--parity-sample", "1500", "--host-steps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert "label-aware" in out["config"]["workload"] and out["value"] > 0
    assert out["parity"]["sample"] == 1500 and out["parity"]["mismatches"] == 0 and out["parity"]["capacity_errors"] == 0
    assert out["parity"]["full_batch_capacity_errors"] == 0
    assert out["roofline"]["kernel_ms"]["k_extend"] > 0 and out["cpu_baseline"]["cores"] == 1


@pytest.mark.gpu
def test_bench_config3_1000_labels_at_a_million_reads():
    """BASELINE config 3 at its own graph and annotation (104 M-edge graph, 1000 labels) with 1 M reads: the lane-per-read kernel in
    front of the labeled group kernel (round 6) — most reads must finish in the lane — and every read of a 20 000-read sample
    equal to the oracle's LabeledAligner, label lists included."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--labels", "1000", "--reads", "1000000", "--steps", "1", "--warmup", "1",
           "--parity-sample", "20000", "--no-cpu-baseline", "--host-steps", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert "1000-label" in out["config"]["workload"] and out["config"]["graph_edges"] > 100_000_000
    assert out["parity"]["sample"] == 20000 and out["parity"]["mismatches"] == 0 and out["parity"]["capacity_errors"] == 0
    assert out["parity"]["full_batch_capacity_errors"] == 0
    km = out["roofline"]["kernel_ms"]
    assert km.get("k_lane", 0) > 0 and km["reads_finished_by_k_lane"] >= 0.9 * 1_000_000, km
