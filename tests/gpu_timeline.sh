# Kernel timeline of the default two-stream flow (one gpurun call): rocprofv3 --kernel-trace of `gpu_r2_probe.py run <workload>`,
# kernel_trace.csv copied to gpurun_out/timeline/ for tests/timeline_report.py.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/timeline; mkdir -p $O
W=${WL:-enwik}
timeout 300 python tests/gpu_r2_probe.py gen $W 2>&1 | grep -v amdgpu.ids
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o tl_$W -- python $R/tests/gpu_r2_probe.py run $W --reps 3 > $O/tl_$W.log 2>&1
cd $R; grep "^\[" $O/tl_$W.log; ls -la $O | head
