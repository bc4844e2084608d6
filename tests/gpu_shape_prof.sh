# per-kernel profile of shapes of tests/gpu_perf_probe.py: SHAPES="periodic ab|200k" bash tests/gpu_shape_prof.sh
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r4shape; mkdir -p $O
IFS='|' read -ra SH <<< "${SHAPES:-periodic ab}"
i=0
for s in "${SH[@]}"; do
i=$((i+1))
cd /tmp && CJS_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s$i -- python $R/tests/gpu_shape_run.py "$s" 3 2>&1 | grep -v amdgpu.ids | grep "ms \|Error\|error" | head -5
cd $R
python - $i <<'PY'
import csv, sys
i = sys.argv[1]
rows = list(csv.DictReader(open('gpurun_out/r4shape/s%s_kernel_stats.csv' % i)))
print('total ms/step %.3f' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / 3))
for r in rows[:12]:
    print('  %-56s calls/step %5.1f ms/step %7.3f avg us %8.1f' % (r['Name'][:56], int(r['Calls']) / 3, float(r['TotalDurationNs']) / 1e6 / 3, float(r['AverageNs']) / 1e3))
PY
done
