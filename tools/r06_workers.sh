# round 6: mgx_align -p N on one device (per-handle streams, device_share): align-loop seconds by worker count
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python - <<'PY'
import os, random, struct, sys
sys.path.insert(0, "tests")
import orc
from test_emu_vs_oracle import rand_seq, mutate, rc
rng = random.Random(4242)
k, genome_len, n_reads, read_len = 31, 300_000, int(os.environ.get("N_READS", "120000")), 150
genome = rand_seq(rng, genome_len)
g = orc.Graph.build(k, [genome], 0, False)
W, last, F, _ = g.export()
with open("/tmp/g.boss", "wb") as f:
    f.write(struct.pack("<7Q", g.k, g.n_edges, *[int(x) for x in F])); f.write(W.tobytes()); f.write(last.tobytes())
with open("/tmp/reads.fa", "w") as f:
    for i in range(n_reads):
        p = rng.randrange(0, genome_len - read_len)
        r = mutate(rng, genome[p:p + read_len])
        if rng.random() < 0.5: r = rc(r)
        f.write(">r%d\n%s\n" % (i, r))
PY
B=$(( ${N_READS:-120000} * 150 / 8 ))
{ for rep in 1 2; do for p in 1 2 4; do
  echo "own streams, -p $p:"; metagraph_amd/_build/mgx_align /tmp/g.boss /tmp/reads.fa -p $p --query-batch-size $B --time 2>&1 >/dev/null | tail -1
  echo "default stream, -p $p:"; MGX_ADAPTER_DEFAULT_STREAM=1 metagraph_amd/_build/mgx_align /tmp/g.boss /tmp/reads.fa -p $p --query-batch-size $B --time 2>&1 >/dev/null | tail -1
done; done; } > gpurun_out/r06_workers_one_device.txt 2>&1
cat gpurun_out/r06_workers_one_device.txt
