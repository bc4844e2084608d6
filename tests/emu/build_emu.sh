#!/bin/sh
# Builds tests/emu/libcjs_emu.so: the product's HIP sources compiled by g++ against the fake HIP
# runtime in tests/emu/hip/ (CPU logic-debug build; test infrastructure only).
set -e
cd "$(dirname "$0")/../.."
SRC=compressjs_amd/csrc
OUT=tests/emu/libcjs_emu.so
g++ -O2 -g -std=c++17 -fPIC -shared -Itests/emu -I$SRC -x c++ \
    $SRC/*.hip -x c++ tests/emu/emu.cpp -o $OUT -Wall -Wno-unused-function -Wno-unknown-pragmas
echo built $OUT
