# What the numbers in DESIGN.md section 4 / profiles/r02_* were produced with (one gpurun call, ~5 GPU-minutes):
#   bash tests/gpu_round_end.sh      (writes under gpurun_out/r02/; copy what is to be kept into profiles/)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); O=$R/gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
# 1. the bench line (default workload) and the secondary workloads of SURVEY.md 8(d)
timeout 300 python bench.py 2>/dev/null | tail -1 > $O/r02_bench.json; cat $O/r02_bench.json | cut -c1-400
for w in e8sa lcg e8sb text; do
  timeout 300 python bench.py --workload $w 2>/dev/null | tail -1 > $O/r02_bench_$w.json
  python -c "import json; j=json.load(open('$O/r02_bench_$w.json')); print('$w', j['value'], 'MB/s', j['ms_per_step'], 'ms', 'bit_exact_vs_reference', j['config']['bit_exact_vs_reference_digest'], 'pcie', j['config']['pcie_inclusive_mb_s'])"
done
# 2. per-kernel times, one stream (every kernel has the GPU to itself), rocprofv3 --kernel-trace --stats
for w in enwik e8sa; do
  cd /tmp && CJS_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r02_e2e_$w -- python $R/tests/gpu_r2_probe.py run $w --reps 5 > $O/e2e_$w.log 2>&1
done
cd /tmp && CJS_STREAMS=1 CJS_FRONT=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r02_e2e_enwik_lsd -- python $R/tests/gpu_r2_probe.py run enwik --reps 5 > $O/e2e_enwik_lsd.log 2>&1
# 3. HBM traffic of the initial-sort stage, new front end and the LSD passes it replaces: separate --pmc passes
for v in front:1 lsd:0; do
  tag=${v%%:*}; f=${v#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    cd /tmp && CJS_STREAMS=1 CJS_FRONT=$f timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o pmc_${tag}_$c --output-format csv -- python $R/tests/gpu_r2_probe.py run enwik --reps 2 > $O/pmc_${tag}_$c.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections, json, os
O = 'gpurun_out/r02'
def per_kernel(path):
    acc = collections.defaultdict(float); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        acc[k] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
    return {k: (acc[k] / len(seen[k]), len(seen[k])) for k in acc}
out = {}
for tag in ('front', 'lsd'):
    f = per_kernel('%s/pmc_%s_FETCH_SIZE_counter_collection.csv' % (O, tag))
    w = per_kernel('%s/pmc_%s_WRITE_SIZE_counter_collection.csv' % (O, tag))
    ent = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith('k1'): continue
        # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section)
        fb, wb = f.get(k, (0, 0))[0] * 1024 * 2, w.get(k, (0, 0))[0] * 1024
        ent[k] = dict(fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb), traffic_bytes_per_launch=round(fb + wb),
                      launches_per_step=max(1, f.get(k, w.get(k))[1] // 2))
    out[tag] = ent
stage = lambda ent, names: sum(v['traffic_bytes_per_launch'] * v['launches_per_step'] for k, v in ent.items() if k in names)
res = {'enwik:100000000': out['front'],
       'lsd_passes_enwik:100000000': out['lsd'],
       'initial_sort_stage_bytes_per_step': {
           'front_end (k1f_sample, k1f_hist, k1f_scan, k1f_scatter, k1f_bsort)': stage(out['front'], ('k1f_sample', 'k1f_hist', 'k1f_scan', 'k1f_scatter', 'k1f_bsort')),
           'lsd_passes (k1_hist, k1_scan, k1_scatter x7, k1_init_heads)': stage(out['lsd'], ('k1_hist', 'k1_scan', 'k1_scatter', 'k1_init_heads'))},
       'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of tests/gpu_r2_probe.py run enwik (CJS_STREAMS=1; 1 warm + 2 steps per pass); KiB units, FETCH_SIZE doubled (gfx950)'}
json.dump(res, open(O + '/r02_pmc_traffic.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(res['initial_sort_stage_bytes_per_step'], indent=1))
for k in ('k1f_bsort', 'k1f_scatter', 'k1f_hist'):
    print(k, out['front'].get(k))
PY
# 4. other data shapes
timeout 300 python tests/gpu_perf_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r02_shapes.log
# 5. decoder (K7-K9): kernel stats at 10^8 and 10^9 bytes, rate at 4*10^8; BWTC -9 (cfg5)
export PYTHONPATH=$R
cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r02_decode -- python $R/tests/gpu_decode_probe.py > $O/decode_1e8.log 2>&1; grep decompress $O/decode_1e8.log | tail -1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r02_decode_1e9 -- python $R/tests/gpu_decode_probe.py 1000000000 > $O/decode_1e9.log 2>&1; grep decompress $O/decode_1e9.log | tail -1
cd $R
timeout 100 python tests/gpu_decode_probe.py 400000000 2>&1 | grep decompress | tail -1
timeout 300 python bench.py --codec bwtc 2>/dev/null | tail -1 > $O/r02_bench_bwtc.json; cut -c1-160 $O/r02_bench_bwtc.json
ls $O | head -60
