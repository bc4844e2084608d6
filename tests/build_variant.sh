#!/bin/sh
# build/variants/lib_<name>.so: the product library with extra compile-time defines (A/B runs: tests/gpu_r6_ab.sh).  Build container only.
#   sh tests/build_variant.sh xcd -DK1D_BUILD_XCD=1
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Icompressjs_amd/csrc "$@" compressjs_amd/csrc/*.hip -o build/variants/lib_$name.so
echo built build/variants/lib_$name.so
