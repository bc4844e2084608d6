# Compile-time variants of the library, built on the GPU box and timed on one workload each:
#   VARIANTS="name:-DX=1,-DY=2 name2:..."  WL="enwik"  KERN="k1f_bsort|k1r_round" (kernels listed from the rocprof summary)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r5var; mkdir -p $O
WL="${WL:-enwik}"
timeout 300 python tests/gpu_r2_probe.py gen $WL 2>&1 | grep -v amdgpu.ids
for v in ${VARIANTS:-base:}; do
  name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
  D=/tmp/var_$name; rm -rf $D; mkdir -p $D; cp -r compressjs_amd tests $D/
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $defs -Icompressjs_amd/csrc compressjs_amd/csrc/*.hip -o $D/compressjs_amd/libcompressjs_amd.so 2>&1 | grep -E "error" ) &
done
wait
for v in ${VARIANTS:-base:}; do
  name=${v%%:*}; D=/tmp/var_$name
  for w in $WL; do
    cd $D && env $(echo ${ENVV:-CJS_NOP=0} | tr ',' ' ') timeout 300 python tests/gpu_r2_probe.py run $w --reps ${REPS:-4} 2>&1 | grep "^\[" | sed "s/^/$name /" | cut -c1-200
    cd /tmp && CJS_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${name}_$w -- python $D/tests/gpu_r2_probe.py run $w --reps 3 > $O/${name}_$w.log 2>&1
    python - $O/${name}_${w}_kernel_stats.csv "$name $w" "${KERN:-k1f_bsort}" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 3e6
print(sys.argv[2], 'one-stream kernel sum %.3f ms/step;' % tot, '; '.join('%s %.3f' % (r['Name'].split('(')[0][:14], float(r['TotalDurationNs']) / 3e6) for r in rows if re.search(sys.argv[3], r['Name'])))
PY
  done
done
