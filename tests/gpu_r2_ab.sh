# A/B of the K1 front end on the round-2 workloads + per-kernel times (one gpurun call)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/r2ab
WL="${WL:-enwik e8sa lcg}"
timeout 200 python tests/gpu_r2_probe.py gen $WL 2>&1 | grep -v amdgpu.ids
for f in 0 1; do
  CJS_FRONT=$f timeout 120 python tests/gpu_r2_probe.py run $WL 2>&1 | grep "^\["
done
for w in ${PROF:-enwik}; do
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2ab -o $w -- python $R/tests/gpu_r2_probe.py run $w --reps 3 > $R/gpurun_out/r2ab/$w.log 2>&1
cd $R
python - $w <<'PY'
import csv, sys
w = sys.argv[1]
rows = list(csv.DictReader(open('gpurun_out/r2ab/%s_kernel_stats.csv' % w)))
steps = 3
print(w, 'total ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps)
for r in rows[:24]:
    print('  %-56s calls/step %5.1f ms/step %7.3f avg us %8.1f' % (r['Name'][:56], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3))
PY
done
