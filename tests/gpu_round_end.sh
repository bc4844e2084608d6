# What the numbers in DESIGN.md section 4 / profiles/r01_*_v9* were produced with (one gpurun call):
#   bash tests/gpu_round_end.sh      (writes under gpurun_out/)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
mkdir -p gpurun_out/v9
export TMPDIR=/tmp
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/v9/bench.json
cat gpurun_out/v9/bench.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v9 -o e2e -- python $R/bench.py --steps 5 > $R/gpurun_out/v9/e2e.log 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/v9 -o pmc$i --output-format csv -- python $R/bench.py --steps 2 --no-verify > $R/gpurun_out/v9/pmc$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
out = open('gpurun_out/v9/pmc_summary.csv', 'w')
for f in sorted(glob.glob('gpurun_out/v9/pmc*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if not k.startswith(('k1_', 'void k1_')): continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
    for k in sorted(acc):
        n = len(seen[k])
        out.write('%s,launches=%d,%s\n' % (k.replace('void ', ''), n, ','.join('%s=%.5g' % (c, v / n) for c, v in sorted(acc[k].items()))))
out.close()
print(open('gpurun_out/v9/pmc_summary.csv').read())
PY
timeout 250 python tests/gpu_perf_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/v9/shapes.log
