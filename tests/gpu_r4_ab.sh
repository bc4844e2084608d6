# Round-4 A/B runs (one gpurun call): VARS = list of "ENV=val,ENV=val" settings, each run on the workloads WL with the digest
# of every stream printed (tests/golden/golden_big.json has the reference's); TRACE = workloads run once with CJS_K1_TRACE=1;
# then a per-kernel rocprofv3 profile of the settings in PROF_VARS on the workloads PROF.  Output under gpurun_out/r4ab/.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r4ab; mkdir -p $O
WL="${WL:-enwik e8sa lcg text e8sb}"
timeout 300 python tests/gpu_r2_probe.py gen $WL 2>&1 | grep -v amdgpu.ids
for v in ${VARS:-CJS_NOP=0}; do
  env $(echo $v | tr ',' ' ') timeout 300 python tests/gpu_r2_probe.py run $WL 2>&1 | grep "^\[\|Error\|error\|Traceback" | head -40
done
for w in ${TRACE:-}; do
  CJS_K1_TRACE=1 CJS_STREAMS=1 timeout 200 python tests/gpu_r2_probe.py run $w --reps 1 2>&1 | grep "^\[" | cut -c1-1500
done
python - <<'PY'
import json
g = json.load(open('tests/golden/golden_big.json'))
for k, v in sorted(g.items()):
    if k.endswith(':100000000:bz2:9'): print('golden', k, v['out_sha256'][:16], v['out_len'])
PY
i=0
for v in ${PROF_VARS:-}; do
i=$((i+1))
for w in ${PROF:-enwik}; do
cd /tmp && env CJS_STREAMS=1 $(echo $v | tr ',' ' ') timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p${i}_$w -- python $R/tests/gpu_r2_probe.py run $w --reps 3 > $O/p${i}_$w.log 2>&1
cd $R
python - $i $w "$v" <<'PY'
import csv, sys
i, w, v = sys.argv[1:4]
rows = list(csv.DictReader(open('gpurun_out/r4ab/p%s_%s_kernel_stats.csv' % (i, w))))
steps = 3
print(v, w, 'total ms/step %.3f' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps))
for r in rows[:30]:
    print('  %-56s calls/step %5.1f ms/step %7.3f avg us %8.1f' % (r['Name'][:56], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3))
PY
done
done
