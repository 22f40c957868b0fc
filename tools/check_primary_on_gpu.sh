#!/bin/bash
# Hardware check of the PRIMARY (CanonicalDBG) kernels without python (their first run, round 2): the C++ driver over libmgx.so on BOSS dumps of
# primary graphs (genome.MT and four random worlds), compared with the oracle's TSV lines prepared on the CPU side (gpurun_in/, written by
# tools/make_primary_check_inputs.py).  Seconds of GPU time.  Writes gpurun_out/primary_check.txt and primary_check2.txt.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/primary_check.txt
: > $out
exe=metagraph_amd/_build/mgx_align
run() {   # name, expected file, args...
    name=$1; want=$2; shift 2
    timeout 40 $exe "$@" > gpurun_out/primary_$name.tsv 2> gpurun_out/primary_$name.err
    rc=$?
    if [ $rc -ne 0 ]; then echo "$name: exit code $rc: $(tail -c 300 gpurun_out/primary_$name.err)" >> $out; return; fi
    if cmp -s gpurun_out/primary_$name.tsv $want; then echo "$name: IDENTICAL ($(wc -l < $want) lines)" >> $out
    else echo "$name: DIFFERENT ($(diff gpurun_out/primary_$name.tsv $want | grep -c '^<') of $(wc -l < $want) lines)" >> $out; fi
}
run mt_a gpurun_in/mt.expect.a.tsv gpurun_in/mt.primary.boss tests/golden/genome_MT1.fq --primary --align-min-exact-match 0.0
run mt_b gpurun_in/mt.expect.b.tsv gpurun_in/mt.primary.boss tests/golden/genome_MT1.fq --primary --align-min-exact-match 0.0 --align-min-seed-length 10
run w31 gpurun_in/w31.expect.tsv gpurun_in/w31.primary.boss gpurun_in/w31.fa --primary
cat $out
if [ -f gpurun_in/w12.primary.boss ]; then     # even k (palindromic k-mers), seed-rich sub-k settings
    : > gpurun_out/primary_check2.txt; out=gpurun_out/primary_check2.txt
    run w12 gpurun_in/w12.expect.tsv gpurun_in/w12.primary.boss gpurun_in/w12.fa --primary --align-min-exact-match 0.0
    run w15 gpurun_in/w15.expect.tsv gpurun_in/w15.primary.boss gpurun_in/w15.fa --primary --align-min-exact-match 0.0 --align-min-seed-length 9
    run w20 gpurun_in/w20.expect.tsv gpurun_in/w20.primary.boss gpurun_in/w20.fa --primary --align-min-seed-length 12
    cat $out
fi
