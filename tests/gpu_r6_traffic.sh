# Round-6 PMC passes (separate --pmc runs, kernel-trace only) of one workload ($1, default e8sa): SQ / LDS / L2 counters and
# HBM traffic per kernel, summed per kernel name over the run.  Output under gpurun_out/r6tr/.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; W=${1:-e8sa}; O=$R/gpurun_out/r6tr; mkdir -p $O
timeout 300 python tests/gpu_r2_probe.py gen $W 2>&1 | grep -v amdgpu.ids
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  cd /tmp && CJS_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O -o ${W}_g$i --output-format csv -- python $R/tests/gpu_r2_probe.py run $W --reps 2 > $O/${W}_g$i.log 2>&1
done
cd $R
python - $W <<'PY'
import csv, glob, collections, sys
W = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in sorted(glob.glob('gpurun_out/r6tr/%s_g*_counter_collection.csv' % W)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); nd[k].add((f, r['Dispatch_Id']))
names = sorted({c for k in acc for c in acc[k]})
with open('gpurun_out/r6tr/%s_pmc_per_kernel.csv' % W, 'w') as o:
    o.write('kernel,' + ','.join(names) + '\n')
    for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0)):
        o.write(k + ',' + ','.join('%.0f' % acc[k].get(c, 0) for c in names) + '\n')
import json, os
tp = 'gpurun_out/r6tr/r06_pmc_traffic.json'
tj = json.load(open(tp)) if os.path.exists(tp) else {}
tj['build'] = os.environ.get('GIT_REV', 'unknown')
tj['note'] = ('HBM traffic per kernel and STEP (one 10^8-byte compress, CJS_STREAMS=1): (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes / steps, '
              'separate --pmc passes (tests/gpu_r6_traffic.sh); FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950, '
              'WRITE_SIZE uncalibrated; Infinity-Cache hits are counted as traffic')
tj['%s:100000000' % W] = {k.split('<')[0]: {'traffic_bytes_per_step': round((2 * acc[k].get('FETCH_SIZE', 0) + acc[k].get('WRITE_SIZE', 0)) * 1024 / 2),
                                          'fetch_kb_per_step': round(acc[k].get('FETCH_SIZE', 0) / 2), 'write_kb_per_step': round(acc[k].get('WRITE_SIZE', 0) / 2)}
                          for k in acc if acc[k].get('FETCH_SIZE', 0) + acc[k].get('WRITE_SIZE', 0) > 0 and not k.startswith('k1d_med')}
med = [k for k in acc if k.startswith('k1d_med')]
if med:
    tj['%s:100000000' % W]['k1d_med'] = {'traffic_bytes_per_step': round(sum((2 * acc[k].get('FETCH_SIZE', 0) + acc[k].get('WRITE_SIZE', 0)) for k in med) * 1024 / 2)}
json.dump(tj, open(tp, 'w'), indent=1, sort_keys=True)
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0))[:14]:
    a = acc[k]
    wc = a.get('SQ_WAVE_CYCLES', 1) or 1
    print('%-22s wait_any %4.0f%%  valu_busy/wavecyc %4.1f%%  lds_conflict/lds_active %4.0f%%  tcc_hit %4.0f%%  fetch %7.1f MB  write %7.1f MB  tcp->tcc rd %6.1fM wr %6.1fM' % (
        k[:22], 100 * a.get('SQ_WAIT_ANY', 0) / wc, 100 * a.get('SQ_ACTIVE_INST_VALU', 0) / wc,
        100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1),
        100 * a.get('TCC_HIT_sum', 0) / max(a.get('TCC_HIT_sum', 0) + a.get('TCC_MISS_sum', 0), 1),
        a.get('FETCH_SIZE', 0) / 1024 / 2, a.get('WRITE_SIZE', 0) / 1024 / 2, a.get('TCP_TCC_READ_REQ_sum', 0) / 2e6, a.get('TCP_TCC_WRITE_REQ_sum', 0) / 2e6))
PY
