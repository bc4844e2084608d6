# Round-3 A/B of the fused bucket kernel (one gpurun call): in-bucket deepening iterations 0 (= round-2 flow with the K1-deep
# tile kernel) .. 32 on the bench workloads, digests against tests/golden/golden_big.json, per-kernel profile, stage clocks.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r3ab; mkdir -p $O
WL="${WL:-enwik e8sa lcg text}"
timeout 300 python tests/gpu_r2_probe.py gen $WL 2>&1 | grep -v amdgpu.ids
for it in ${ITERS:-0 32 44 56 68 92}; do
  CJS_BSORT_DEPTH=$it timeout 200 python tests/gpu_r2_probe.py run $WL 2>&1 | grep "^\["
done
python - <<'PY'
import json
g = json.load(open('tests/golden/golden_big.json'))
for k, v in sorted(g.items()):
    if k.endswith(':100000000:bz2:9'): print('golden', k, v['out_sha256'][:16], v['out_len'])
PY
for it in ${PROF_ITERS:-68}; do
for w in ${PROF:-enwik}; do
cd /tmp && CJS_STREAMS=1 CJS_BSORT_DEPTH=$it timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o it${it}_$w -- python $R/tests/gpu_r2_probe.py run $w --reps 3 > $O/it${it}_$w.log 2>&1
cd $R
python - $it $w <<'PY'
import csv, sys
it, w = sys.argv[1:3]
rows = list(csv.DictReader(open('gpurun_out/r3ab/it%s_%s_kernel_stats.csv' % (it, w))))
steps = 3
print('iters', it, w, 'total ms/step %.3f' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps))
for r in rows[:22]:
    print('  %-56s calls/step %5.1f ms/step %7.3f avg us %8.1f' % (r['Name'][:56], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3))
PY
done
done
if [ -f build/lib_trace.so ]; then
  for it in 0 68; do
  COMPRESSJS_AMD_LIB=$R/build/lib_trace.so CJS_STREAMS=1 CJS_K1_TRACE=1 CJS_BSORT_DEPTH=$it timeout 120 python tests/gpu_r2_probe.py run enwik --reps 1 2>&1 | grep "front end" | head -2
  done
fi
