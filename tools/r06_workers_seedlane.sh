# round 6: mgx_align -p 1 / -p 4 on one device with and without the lane-per-read seeder (the -p 4 test's scenario: 120 000
# reads in eight tasks), MGX_HOST_TIMERS-free: align-loop seconds only
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python - <<'PY'
import os, random, struct, sys
sys.path.insert(0, "tests")
import orc
from test_emu_vs_oracle import rand_seq, mutate, rc
rng = random.Random(4242)
k, genome_len, n_reads, read_len = 31, 300_000, 120000, 150
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
B=$(( 120000 * 150 / 8 ))
{ for rep in 1 2 3; do for p in 1 4; do for o in 1 0; do
  echo -n "seed_lane=$o -p $p: "; metagraph_amd/_build/mgx_align /tmp/g.boss /tmp/reads.fa -p $p --query-batch-size $B --time --kernel-option seed_lane=$o 2>&1 >/dev/null | tail -1
done; done; done; } > gpurun_out/r06_workers_seedlane.txt 2>&1
cat gpurun_out/r06_workers_seedlane.txt
