# A/B of variant libraries (build/variants/lib_<name>.so, built in the build container by tests/build_variant.sh) inside ONE gpurun call:
# two-stream step (median / best of REPS) per workload, and with PROF=1 the one-stream per-kernel sums under rocprofv3.
#   VARS="base xcd ..." WLS="enwik e8sa" REPS=8 PROF=1 KERN="k1d_|k1f_bsort" bash tests/gpu_r6_ab.sh
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r6ab; mkdir -p $O
WLS="${WLS:-enwik e8sa}"
timeout 300 python tests/gpu_r2_probe.py gen $WLS 2>&1 | grep -v amdgpu.ids
for pass in 1 2; do
for name in ${VARS:-base}; do
  if [ "$name" = base ]; then unset COMPRESSJS_AMD_LIB; else export COMPRESSJS_AMD_LIB=$R/build/variants/lib_$name.so; fi
  timeout 300 python tests/gpu_r2_probe.py run $WLS --reps ${REPS:-8} 2>&1 | grep "^\[" | sed "s/^/$name /" | cut -c1-220
done
done
if [ -n "$PROF" ]; then
for name in ${VARS:-base}; do
  if [ "$name" = base ]; then unset COMPRESSJS_AMD_LIB; else export COMPRESSJS_AMD_LIB=$R/build/variants/lib_$name.so; fi
  for WL in $WLS; do
  cd /tmp && CJS_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${name}_$WL -- python $R/tests/gpu_r2_probe.py run $WL --reps 3 > $O/${name}_$WL.log 2>&1
  cd $R
  python - $O/${name}_${WL}_kernel_stats.csv "$name $WL" "${KERN:-k1r_round|k1f_bsort|k1d_}" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 3e6
print(sys.argv[2], 'one-stream kernel sum %.3f ms/step;' % tot, '; '.join('%s %.3f' % (r['Name'].split('(')[0][:18], float(r['TotalDurationNs']) / 3e6) for r in rows if re.search(sys.argv[3], r['Name'])))
PY
  done
done
fi
