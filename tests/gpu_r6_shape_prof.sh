# Per-kernel one-stream profile of single data shapes (not a test):  SHAPES="sample5 sample3 zeros" bash tests/gpu_r6_shape_prof.sh
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r6shape; mkdir -p $O
for s in ${SHAPES:-sample5}; do
  cd /tmp && CJS_STREAMS=${STREAMS:-1} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o sh_$s -- python $R/tests/gpu_r6_shapes_ab.py $s > $O/sh_$s.log 2>&1
  cd $R
  grep -v amdgpu.ids $O/sh_$s.log | tail -2
  python - $O/sh_${s}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 5e6
print('kernel sum %.3f ms/call' % tot)
for r in rows[:14]:
    print('   %-34s calls/call %5.1f  ms/call %7.3f' % (r['Name'].split('(')[0][:34], int(r['Calls']) / 5, float(r['TotalDurationNs']) / 5e6))
PY
done
