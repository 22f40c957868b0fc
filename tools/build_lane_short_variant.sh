#!/bin/bash
# A/B 10 build: libmgx_short.so = the product's objects + the lane kernel's -DMGX_LANE_SHORT build (reads of up to 160 characters, three
# wavefronts per SIMD) + mgx.o with -DMGX_WITH_LANE_SHORT (option lane_short=1 selects it).  Run with MGX_LIB_PATH=...  Not a product build.
set -e
cd "$(dirname "$0")/.."
B=metagraph_amd/_build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
/opt/rocm/bin/hipcc $FLAGS -DMGX_LANE_SHORT "$@" -c -o $B/mgx_lane_short.o metagraph_amd/csrc/mgx_lane.hip &
/opt/rocm/bin/hipcc $FLAGS -DMGX_WITH_LANE_SHORT -c -o $B/mgx_with_short.o metagraph_amd/csrc/mgx.hip &
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o $B/libmgx_short.so $B/mgx_with_short.o $B/mgx_primary.o $B/mgx_annot.o $B/mgx_files.o $B/mgx_chain.o $B/mgx_gather.o $B/mgx_seedlane.o $B/mgx_ext64.o $B/mgx_lane.o $B/mgx_lane_short.o $B/mgx_lab64.o $B/mgx_grp8_lab.o $B/mgx_grp8.o $B/mgx_grp8_prim.o $B/mgx_grp8_alt.o
echo built $B/libmgx_short.so
