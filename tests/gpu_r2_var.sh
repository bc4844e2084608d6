# variants of the front end built into build/lib_*.so (one gpurun call): times + per-kernel profile of each
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/r2var
WL="${WL:-enwik e8sa lcg}"
timeout 200 python tests/gpu_r2_probe.py gen $WL 2>&1 | grep -v amdgpu.ids
for lib in ${LIBS:-compressjs_amd/libcompressjs_amd.so $(ls build/lib_*.so 2>/dev/null)}; do
  echo "== $lib"
  COMPRESSJS_AMD_LIB=$R/$lib timeout 120 python tests/gpu_r2_probe.py run $WL 2>&1 | grep "^\["
  tag=$(basename $lib .so)
  for w in ${PROF:-enwik}; do
    cd /tmp && COMPRESSJS_AMD_LIB=$R/$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2var -o ${tag}_$w -- python $R/tests/gpu_r2_probe.py run $w --reps 3 > $R/gpurun_out/r2var/${tag}_$w.log 2>&1
    cd $R
    python - $tag $w <<'PY'
import csv, sys
tag, w = sys.argv[1:3]
rows = list(csv.DictReader(open('gpurun_out/r2var/%s_%s_kernel_stats.csv' % (tag, w))))
steps = 3
print(tag, w, 'total ms/step %.3f' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print('  %-56s calls/step %5.1f ms/step %7.3f avg us %8.1f' % (r['Name'][:56], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3))
PY
  done
done
