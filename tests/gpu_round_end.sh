# What the numbers in DESIGN.md section 4 / profiles/r03_* were produced with (one gpurun call, ~6 GPU-minutes):
#   bash tests/gpu_round_end.sh      (writes under gpurun_out/r03/; copy what is to be kept into profiles/)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); O=$R/gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
# 1. the bench line (default workload; carries E8S-A in config.e8sa_*) and the secondary workloads of SURVEY.md 8(d)
timeout 300 python bench.py 2>/dev/null | tail -1 > $O/r03_bench.json; cut -c1-300 $O/r03_bench.json
for w in e8sa lcg e8sb text; do
  timeout 300 python bench.py --workload $w 2>/dev/null | tail -1 > $O/r03_bench_$w.json
  python -c "import json; j=json.load(open('$O/r03_bench_$w.json')); print('$w', j['value'], 'MB/s', j['ms_per_step'], 'ms', 'bit_exact_vs_reference', j['config']['bit_exact_vs_reference_digest'], 'pcie', j['config']['pcie_inclusive_mb_s'])"
done
# 2. per-kernel times, one stream (every kernel has the GPU to itself), rocprofv3 --kernel-trace --stats
for w in enwik e8sa; do
  cd /tmp && CJS_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r03_e2e_$w -- python $R/tests/gpu_r2_probe.py run $w --reps 5 > $O/e2e_$w.log 2>&1
done
cd /tmp && CJS_STREAMS=1 CJS_ROUNDS=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r03_e2e_enwik_r2flow -- python $R/tests/gpu_r2_probe.py run enwik --reps 5 > $O/e2e_enwik_r2flow.log 2>&1
# 3. PMC passes (separate --pmc runs, kernel-trace only): HBM traffic (FETCH_SIZE / WRITE_SIZE) of K1 on enwik and E8S-A,
#    and the SQ / LDS / L2 counters north_star names (LDS bank conflicts of k2_mtf; wait / issue of the sort kernels)
for w in enwik e8sa; do
  for c in FETCH_SIZE WRITE_SIZE; do
    cd /tmp && CJS_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o pmc_${w}_$c --output-format csv -- python $R/tests/gpu_r2_probe.py run $w --reps 2 > $O/pmc_${w}_$c.log 2>&1
  done
done
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  cd /tmp && CJS_STREAMS=1 timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $O -o sq$i --output-format csv -- python $R/tests/gpu_r2_probe.py run enwik --reps 2 > $O/sq$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json, os
O = 'gpurun_out/r03'
def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
    return acc, {k: len(v) for k, v in seen.items()}
res = {}
for w in ('enwik', 'e8sa'):
    f, nf = per_kernel('%s/pmc_%s_FETCH_SIZE_counter_collection.csv' % (O, w))
    wr, nw = per_kernel('%s/pmc_%s_WRITE_SIZE_counter_collection.csv' % (O, w))
    ent = {}
    for k in sorted(set(f) | set(wr)):
        if not k.startswith('k1'): continue
        steps = 2                                           # --reps 2: two compress calls per pass
        # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section: calibrated for wide
        # streaming reads only - gathers of a few bytes and Infinity-Cache hits are counted too)
        fb = f.get(k, {}).get('FETCH_SIZE', 0) * 1024 * 2 / steps
        wb = wr.get(k, {}).get('WRITE_SIZE', 0) * 1024 / steps
        ent[k] = dict(fetch_bytes_per_step=round(fb), write_bytes_per_step=round(wb), traffic_bytes_per_step=round(fb + wb),
                      launches_per_step=round(nf.get(k, nw.get(k, 0)) / steps, 1),
                      traffic_bytes_per_launch=round((fb + wb) / max(1.0, nf.get(k, nw.get(k, 1)) / steps)))
    res['%s:100000000' % w] = ent
    res['K1_bytes_per_step:%s' % w] = sum(v['traffic_bytes_per_step'] for v in ent.values())
res['note'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of tests/gpu_r2_probe.py run <workload> (CJS_STREAMS=1; 1 warm-up + 2 steps per '
               'pass); KiB units; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 - that correction is calibrated for wide streaming reads only, '
               'and Infinity-Cache hits are counted in FETCH_SIZE: for the gather-heavy kernels the figure is an upper bound of the HBM bytes')
json.dump(res, open(O + '/r03_pmc_traffic.json', 'w'), indent=1, sort_keys=True)
print({k: v for k, v in res.items() if k.startswith('K1_bytes')})
rows = []
for p in sorted(glob.glob(O + '/sq*_counter_collection.csv')):
    acc, n = per_kernel(p)
    for k in sorted(acc):
        if not (k.startswith('k1') or k.startswith('k2_mtf') or k.startswith('k34')): continue
        for c, v in sorted(acc[k].items()):
            rows.append((k, c, v / n[k], n[k]))
with open(O + '/r03_pmc_sq.csv', 'w') as fh:
    fh.write('kernel,counter,value_per_launch,launches\n')
    for r in rows: fh.write('%s,%s,%.6g,%d\n' % r)
for r in rows:
    if r[0] in ('k2_mtf', 'k1f_bsort', 'k1r_round') and r[1] in ('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES', 'SQ_INSTS_VALU'): print(r)
PY
# 4. other data shapes, with the oracle's digest next to the GPU's on every row
timeout 600 python tests/gpu_perf_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r03_shapes.log
# 5. decoder (K7-K9): kernel stats at 10^8 bytes, rate at 10^9; BWTC -9 (cfg5); gather microbenchmark
export PYTHONPATH=$R
cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r03_decode -- python $R/tests/gpu_decode_probe.py > $O/decode_1e8.log 2>&1; grep decompress $O/decode_1e8.log | tail -1
cd $R
timeout 300 python tests/gpu_decode_probe.py 1000000000 2>&1 | grep decompress | tail -1 | tee $O/r03_decode_1e9.log
timeout 300 python bench.py --codec bwtc 2>/dev/null | tail -1 > $O/r03_bench_bwtc.json; cut -c1-200 $O/r03_bench_bwtc.json
[ -x build/gather ] && ./build/gather 2>/dev/null | grep -E "^gather|^walk" > $O/r03_gather_microbench.txt
# 6. kernel timeline of the default two-stream flow (which kernels overlap, what is launch-bound): tests/timeline_report.py
cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O -o r03_tl_enwik -- python $R/tests/gpu_r2_probe.py run enwik --reps 3 > $O/tl_enwik.log 2>&1
cd $R && python tests/timeline_report.py $O/r03_tl_enwik_kernel_trace.csv --all > $O/r03_timeline_enwik.txt 2>&1; head -3 $O/r03_timeline_enwik.txt
ls $O | head -80
