#!/bin/bash
# A/B build of the extension kernel: tools/build_variant.sh TAG [extra hipcc flags for mgx_grp.hip ...]
# -> metagraph_amd/_build/libmgx_TAG.so (run with MGX_LIB_PATH=metagraph_amd/_build/libmgx_TAG.so).  Not a product build.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
B=metagraph_amd/_build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
if [ -n "$VARIANT_ONLY_SEEDING_UNIT" ]; then     # the flags go to mgx.hip only
    cp $B/mgx_grp8.o $B/mgx_grp8_$tag.o; VARIANT_ALL_UNITS=1
else
    /opt/rocm/bin/hipcc $FLAGS -DMGX_GROUP=8 "$@" -c -o $B/mgx_grp8_$tag.o metagraph_amd/csrc/mgx_grp.hip
fi
MGX_O=$B/mgx.o
if [ -n "$VARIANT_ALL_UNITS" ]; then      # the flags also go to the seeding unit
    /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $B/mgx_$tag.o metagraph_amd/csrc/mgx.hip
    MGX_O=$B/mgx_$tag.o
fi
/opt/rocm/bin/hipcc $FLAGS -shared -o $B/libmgx_$tag.so $MGX_O $B/mgx_primary.o $B/mgx_annot.o $B/mgx_files.o $B/mgx_chain.o $B/mgx_gather.o $B/mgx_seedlane.o $B/mgx_ext64.o $B/mgx_lane.o $B/mgx_lab64.o $B/mgx_grp8_lab.o $B/mgx_grp8_$tag.o $B/mgx_grp8_prim.o $B/mgx_grp8_alt.o
echo built $B/libmgx_$tag.so
