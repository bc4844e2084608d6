# K1F_TRACE build of the library (stage clocks of k1f_bsort / k1r_round) run once on $1 (default enwik) with CJS_K1_TRACE=1
cd ${GRAFT_REPO_ROOT:-.}
W=${1:-enwik}
mkdir -p /tmp/trlib && cp -r compressjs_amd /tmp/trlib/ && cp -r tests /tmp/trlib/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DK1F_TRACE ${TRACE_DEFS:-} -Icompressjs_amd/csrc compressjs_amd/csrc/*.hip -o /tmp/trlib/compressjs_amd/libcompressjs_amd.so || exit 1
cd /tmp/trlib && CJS_K1_TRACE=1 CJS_STREAMS=1 timeout 300 python tests/gpu_r2_probe.py run $W --reps 1 2>&1 | grep "^\[" | cut -c1-900
