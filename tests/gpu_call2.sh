set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 120 python tests/gpu_deep_probe.py gen) 2>&1 | tail -3
(env CJS_DEEP_ITERS=0 timeout 150 python tests/gpu_deep_probe.py run enwik text sample5 sample3 runs | tail -6
 env CJS_K1_TRACE=1 timeout 150 python tests/gpu_deep_probe.py run enwik text sample5 sample3 runs 2>&1 | grep -v "^\[k1\] \(tile\|sparse\)" | sort | uniq | tail -12
 env CJS_DEEP_TILE=1024 timeout 150 python tests/gpu_deep_probe.py run enwik text sample5 | tail -4
 env CJS_DEEP_ITERS=64 timeout 150 python tests/gpu_deep_probe.py run enwik text | tail -3
 env CJS_DEEP_ITERS=8 timeout 150 python tests/gpu_deep_probe.py run enwik text | tail -3 ) 2>&1 | tee gpurun_out/deep_ab3.log
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v10 -o v10 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --no-verify > $GRAFT_REPO_ROOT/gpurun_out/prof_v10.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_v10.log | cut -c1-300
