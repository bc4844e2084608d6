# Variant libraries built in the build container (build/variants/lib_<name>.so, compile-time knobs) timed on one workload: two-stream step
# and one-stream per-kernel times.  VARS="base m5 ..." WL=enwik KERN="k1r_round|k1f_bsort"
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r5lv; mkdir -p $O
WL="${WL:-enwik}"
timeout 300 python tests/gpu_r2_probe.py gen $WL 2>&1 | grep -v amdgpu.ids
for name in ${VARS:-base}; do
  if [ "$name" = base ]; then unset COMPRESSJS_AMD_LIB; else export COMPRESSJS_AMD_LIB=$R/build/variants/lib_$name.so; fi
  timeout 300 python tests/gpu_r2_probe.py run $WL --reps ${REPS:-6} 2>&1 | grep "^\[" | sed "s/^/$name /" | cut -c1-200
  cd /tmp && CJS_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${name}_$WL -- python $R/tests/gpu_r2_probe.py run $WL --reps 3 > $O/${name}_$WL.log 2>&1
  cd $R
  python - $O/${name}_${WL}_kernel_stats.csv "$name $WL" "${KERN:-k1r_round|k1f_bsort}" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 3e6
print(sys.argv[2], 'one-stream kernel sum %.3f ms/step;' % tot, '; '.join('%s %.3f' % (r['Name'].split('(')[0][:18], float(r['TotalDurationNs']) / 3e6) for r in rows if re.search(sys.argv[3], r['Name'])))
PY
done
