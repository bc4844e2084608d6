set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 120 python tests/gpu_deep_probe.py gen) 2>&1 | tail -3
for v in "CJS_DEEP_ITERS=0" "CJS_DEEP_ITERS=32" "CJS_DEEP_TILE=1024"; do
  env $v CJS_K1_TRACE=1 timeout 120 python tests/gpu_deep_probe.py run enwik text --check 2>&1 | grep -v "^\[k1\] \(tile\|sparse\)" | tail -8
done 2>&1 | tee gpurun_out/deep_ab.log
(time timeout 420 python -m pytest tests -m gpu -x -q) 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 200 python bench.py 2>&1 | tail -2 | tee gpurun_out/bench_v9.json
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v9 -o v9 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --no-verify > $GRAFT_REPO_ROOT/gpurun_out/prof_v9.log 2>&1
cd $GRAFT_REPO_ROOT
(env CJS_DEEP_ITERS=16 timeout 100 python tests/gpu_deep_probe.py run enwik | tail -2
 env CJS_DEEP_ITERS=64 timeout 100 python tests/gpu_deep_probe.py run enwik | tail -2
 env CJS_DEEP_ITERS=0 timeout 100 python tests/gpu_deep_probe.py run sample5 sample3 runs | tail -4
 timeout 100 python tests/gpu_deep_probe.py run sample5 sample3 runs | tail -4) 2>&1 | tee gpurun_out/deep_ab2.log
ls gpurun_out/prof_v9 | head
