#!/bin/sh
# Builds tests/emu/libcjs_emu.so: the product's HIP sources compiled by g++ against the fake HIP
# runtime in tests/emu/hip/ (CPU logic-debug build; test infrastructure only).
set -e
cd "$(dirname "$0")/../.."
SRC=compressjs_amd/csrc
# EMU_OUT / EMU_DEFS: variant builds (e.g. a tiny bucket capacity so that the rare paths of k1_front.hip run)
OUT=${EMU_OUT:-tests/emu/libcjs_emu.so}
g++ -O2 -g -std=c++17 -fPIC -shared $EMU_DEFS -Itests/emu -I$SRC -x c++ \
    $SRC/*.hip -x c++ tests/emu/emu.cpp -o $OUT -Wall -Wno-unused-function -Wno-unknown-pragmas
echo built $OUT
