# What the numbers in DESIGN.md section 4 / profiles/r06_* were produced with (one gpurun call, ~5 GPU-minutes):
#   GIT_REV=$(git rev-parse --short HEAD) bash tests/gpu_r6_round_end.sh      (writes under gpurun_out/r06/; copy what is to be kept into profiles/)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); O=$R/gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
# 1. the bench line (default workload; carries E8S-A in config.e8sa_*) and the secondary workloads of SURVEY.md 8(d)
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/r06_bench.json; cut -c1-300 $O/r06_bench.json; echo
for w in e8sa lcg e8sb text; do
  timeout 600 python bench.py --workload $w 2>/dev/null | tail -1 > $O/r06_bench_$w.json
  python -c "import json; j=json.load(open('$O/r06_bench_$w.json')); print('$w', j['value'], 'MB/s', j['ms_per_step'], 'ms', 'bit_exact_vs_reference', j['config']['bit_exact_vs_reference_digest'], 'pcie', j['config']['pcie_inclusive_mb_s'], 'dominant', j['roofline']['kernel'], j['roofline']['frac'])"
done
timeout 600 python bench.py --codec bwtc 2>/dev/null | tail -1 > $O/r06_bench_bwtc.json; cut -c1-200 $O/r06_bench_bwtc.json; echo
# 2. per-kernel times, one stream (every kernel has the GPU to itself), rocprofv3 --kernel-trace --stats
for w in enwik e8sa; do
  cd /tmp && CJS_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r06_e2e_$w -- python $R/tests/gpu_r2_probe.py run $w --reps 5 > $O/e2e_$w.log 2>&1
  cd $R
done
# 3. data shapes (every row with the oracle's digest next to the GPU's)
timeout 900 python tests/gpu_perf_probe.py 2>&1 | grep -v amdgpu.ids > $O/r06_shapes.log; cut -c1-150 $O/r06_shapes.log
# 4. PMC passes: bash tests/gpu_r5_traffic.sh enwik; bash tests/gpu_r5_traffic.sh e8sa  (separate call; writes gpurun_out/r5pmc/)
