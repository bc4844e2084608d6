# baseline of the round-1 build on the round-2 workloads + batch-size sweep (one gpurun call)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/r2base
timeout 200 python tests/gpu_r2_probe.py gen enwik e8sa lcg 2>&1 | grep -v amdgpu.ids
for b in 128 64 32 16; do
  timeout 120 python tests/gpu_r2_probe.py run enwik e8sa lcg --batch $b 2>&1 | grep "^\["
done
CJS_STREAMS=2 timeout 120 python tests/gpu_r2_probe.py run enwik e8sa --batch 128 2>&1 | grep "^\["
CJS_STREAMS=2 timeout 120 python tests/gpu_r2_probe.py run enwik e8sa --batch 32 2>&1 | grep "^\["
CJS_STREAMS=4 timeout 120 python tests/gpu_r2_probe.py run enwik e8sa --batch 64 2>&1 | grep "^\["
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2base -o e8sa -- python $R/tests/gpu_r2_probe.py run e8sa --reps 3 > $R/gpurun_out/r2base/e8sa.log 2>&1
cd $R; ls gpurun_out/r2base | head
