# Round-3 A/B, second series (one gpurun call): sub-batch geometry (--batch N: blocks in flight over the streams) and compile-time
# variants of the front end (COMPRESSJS_AMD_LIB=build/lib_*.so), every run with the digest of the stream printed.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp
WL="${WL:-enwik e8sa}"
timeout 300 python tests/gpu_r2_probe.py gen $WL 2>&1 | grep -v amdgpu.ids
for b in ${BATCHES:-128 84 64 48}; do
  timeout 200 python tests/gpu_r2_probe.py run $WL --batch $b 2>&1 | grep "^\["
done
for v in ${VARS:-CJS_STREAMS=3 CJS_STREAMS=4}; do
  env $(echo $v | tr ',' ' ') timeout 200 python tests/gpu_r2_probe.py run $WL 2>&1 | grep "^\["
done
for l in ${LIBS:-}; do
  echo "lib $l"
  COMPRESSJS_AMD_LIB=$R/build/$l timeout 200 python tests/gpu_r2_probe.py run $WL 2>&1 | grep "^\["
done
