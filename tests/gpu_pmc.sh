cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from compressjs_amd import synth
np.save('/tmp/enwik.npy', synth.enwik_like(100_000_000, 2025))
PY
i=0
for grp in "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o p$i --output-format csv -- python $GRAFT_REPO_ROOT/tests/gpu_deep_probe.py run enwik > $GRAFT_REPO_ROOT/gpurun_out/pmc/p$i.log 2>&1
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc/p$i.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        if not any(x in k for x in ('k1_deep', 'k1_update', 'k1_scatter', 'k1_init_heads', 'k1_finish')): continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (r['Dispatch_Id'], k) not in seen: seen.add((r['Dispatch_Id'], k)); cnt[k] += 1
    print(f)
    for k in acc:
        print('  %-42s n=%d ' % (k, cnt[k]) + ' '.join('%s=%.4g' % (c, v / cnt[k]) for c, v in acc[k].items()))
PY
