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


# ---------------------------------------------------------------------------------------------------------------------
# The complete-alignment gather (metagraph_amd/gather.py) with REAL device-layout records: every rank aligns its own
# shard with the host model of the kernels (same ReadResult headers + word stream the GPU writes), the two-phase gather
# moves headers and the variable-length stream to rank 0 over gloo, rank 0 decodes every rank's records through the
# C-ABI (mgx_results_from_raw, host only) and prints them with mgx_format_tsv; the lines must equal the oracle's TSV
# (cli/align.cpp:254-285,469-473) for every read of every rank.
WORKER_FULL = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    from metagraph_amd import capi, gather as mg
    import orc, emu_drv
    from test_emu_vs_oracle import make_world
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    k = 15
    g, all_reads = make_world(77, k, genome_len=3000, n_reads=60, read_len=90, n_variants=12)
    # ragged shards: rank 0 gets fewer and shorter reads, so stream lengths differ between ranks
    shard = all_reads[:20] if rank == 0 else all_reads[20:]
    if rank == 0:
        shard = [r[:70] for r in shard] + ["ACGTTTGA", ""]
    # equal header counts are not required by gather_raw's stream phase but dist.gather needs equal shapes: pad shards
    n_max = torch.tensor([len(shard)]); dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
    shard = shard + [""] * (int(n_max.item()) - len(shard))
    cfg = capi.config_cli(k)
    G = emu_drv.EmuGraph(g)
    run = emu_drv.EmuRun(G, cfg, shard)
    assert not run.error, run.error
    hb, sb = run.raw()
    hdr = torch.frombuffer(bytearray(hb), dtype=torch.uint8)
    used = len(sb) // 4
    cap = torch.tensor([used]); dist.all_reduce(cap, op=dist.ReduceOp.MAX)
    # stream buffers as the aligner sizes them: per rank — rank 0's is SHORTER than what rank 1 uses (the padded-temporary path)
    stream = torch.zeros(4 * (int(cap.item()) if rank else used) + 64 * rank, dtype=torch.uint8)
    stream[:len(sb)] = torch.frombuffer(bytearray(sb), dtype=torch.uint8) if sb else stream[:0]
    # the pipeline form bench.py uses: batch A's gather is launched on a snapshot, the buffers are then overwritten (the "next
    # batch"), and batch A must still arrive intact; the blocking form (gather_raw) must agree with it
    gat = mg.ResultGatherer(dist, rank, world)
    gat.start(hdr, stream, used)
    keep_h, keep_s = hdr.clone(), stream.clone()
    hdr.fill_(0xEE); stream.fill_(0xEE)
    got = [(h.clone(), s.clone()) for h, s in gat.finish()] if rank == 0 else gat.finish()
    hdr.copy_(keep_h); stream.copy_(keep_s)
    gat.start(hdr, stream, used)                   # a second batch through the same (reused) buffers
    got2 = gat.finish()
    got3 = mg.gather_raw(dist, rank, world, hdr, stream, used)
    if rank == 0:
        for r in range(world):
            for other in (got2, got3):
                assert torch.equal(got[r][0], other[r][0]) and torch.equal(got[r][1], other[r][1])
    shards = [None] * world
    dist.all_gather_object(shards, shard)
    if rank == 0:
        L = capi.lib()
        n_lines = 0
        for r in range(world):
            h, s = got[r]
            raw = mg.RawResults(h.numpy(), s.numpy())
            want = orc.AlignRun(g, cfg, shards[r]).tsv_lines()
            for qi, q in enumerate(shards[r]):
                qb = q.encode()
                n = L.mgx_format_tsv(C.byref(raw.res), qi, b"q%%d" %% qi, qb, len(qb), cfg.min_path_score, None, 0)
                buf = C.create_string_buffer(n + 1)
                L.mgx_format_tsv(C.byref(raw.res), qi, b"q%%d" %% qi, qb, len(qb), cfg.min_path_score, buf, n + 1)
                line = buf.value.decode().rstrip("\\n")
                w = want[qi].split("\\t", 1)[1]
                assert line.split("\\t", 1)[1] == w, (r, qi, line, want[qi])
                n_lines += 1
            raw.close()
        print("FULL_OK", world, n_lines)
    dist.barrier()
    dist.destroy_process_group()
""") % (ROOT, ROOT)


def test_two_rank_full_alignment_gather_gloo(tmp_path):
    script = tmp_path / "worker_full.py"
    script.write_text(WORKER_FULL)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29518", str(script)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "FULL_OK 2" in r.stdout
