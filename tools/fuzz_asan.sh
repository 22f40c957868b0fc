#!/bin/bash
# The fuzz campaign (tools/fuzz_emu.py) on an address-sanitized build of the kernels' host model: every array the wave programs
# index — per-read arenas, LDS models, the graph — becomes bounds-checked.  Test infrastructure, CPU only, ~3x slower.
#   tools/fuzz_asan.sh [8] [fuzz_emu.py arguments ...]      e.g.  tools/fuzz_asan.sh --minutes 60 --seed 5 --lane
#                                                                 tools/fuzz_asan.sh 8 --minutes 60 --seed 6 --labels
# (round 4: its first hour found chain_step's record running past the cell arena, tests/test_fuzz_smoke.py)
set -e
cd "$(dirname "$0")/.."
wave=""; suffix=""
if [ "$1" = "8" ] || [ "$1" = "16" ]; then wave="-DMGX_EMU_WAVE=$1"; suffix="_w$1"; lanes="--lanes $1"; shift; fi
lib=/tmp/libmgxemu_asan$suffix.so
( cd tests/emu && g++ -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -DMGX_ARENA_REDZONE=64 $wave -DMGX_MAX_ALT=4 -DMGX_WITH_PRIMARY=1 \
    -DMGX_WITH_LABELS=1 -Wno-sign-compare -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-variable -ffp-contract=off \
    -I. -I../../metagraph_amd/csrc -shared -o $lib emu_driver.cpp )
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 MGX_EMU_LIB=$lib exec python -u tools/fuzz_emu.py $lanes "$@"
