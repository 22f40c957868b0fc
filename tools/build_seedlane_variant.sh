#!/bin/bash
# A/B build of the lane-per-read seeder: tools/build_seedlane_variant.sh TAG [extra hipcc flags for mgx_seedlane.hip ...]
# -> metagraph_amd/_build/libmgx_TAG.so (run with MGX_LIB_PATH=...).  Not a product build.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
B=metagraph_amd/_build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o $B/mgx_seedlane_$tag.o metagraph_amd/csrc/mgx_seedlane.hip
/opt/rocm/bin/hipcc $FLAGS -shared -o $B/libmgx_$tag.so $B/mgx.o $B/mgx_primary.o $B/mgx_annot.o $B/mgx_files.o $B/mgx_chain.o $B/mgx_gather.o $B/mgx_seedlane_$tag.o $B/mgx_ext64.o $B/mgx_lane.o $B/mgx_lab64.o $B/mgx_grp8_lab.o $B/mgx_grp8.o $B/mgx_grp8_prim.o $B/mgx_grp8_alt.o
echo built $B/libmgx_$tag.so
