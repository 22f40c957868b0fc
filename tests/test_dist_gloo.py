"""World-size-2 launch path of bench.py's sharding logic on CPU (gloo): rank-dependent read shards are
disjoint, the gather of fixed-size result records lands in rank order, and max-over-ranks timing works."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from metagraph_amd import synth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cpu")
    genome = synth.random_genome(4000, 20240501, dev)          # identical graph input on every rank
    reads = synth.sample_reads(genome, 64, 50, 20240503 + rank)  # per-rank shard (seed + rank)
    # stand-in for the 64-byte per-read result records produced on the device
    rec = torch.full((64, 64), rank, dtype=torch.uint8)
    rec[:, 1] = reads[:, 0]
    out = [torch.empty_like(rec) for _ in range(world)] if rank == 0 else None
    dist.barrier()
    t0 = time.time()
    dist.gather(rec, out, dst=0)
    dist.barrier()
    el = torch.tensor([time.time() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    g = [torch.empty_like(genome) for _ in range(world)]
    dist.all_gather(g, genome)
    assert all(torch.equal(x, genome) for x in g)
    firsts = [torch.empty(64, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(firsts, reads[:, 0].contiguous())
    if rank == 0:
        assert all(int(out[r][0, 0]) == r for r in range(world))
        assert not torch.equal(firsts[0], firsts[1]), "shards must differ"
        print("OK", world, float(el))
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gather_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "OK 2" in r.stdout
