# PMC counters of selected kernels (one gpurun call): KERN=regex LIB=path WL=workload
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/r2pmc
WL="${WL:-enwik}"; KERN="${KERN:-k1f_}"; LIB="${LIB:-compressjs_amd/libcompressjs_amd.so}"
timeout 200 python tests/gpu_r2_probe.py gen $WL 2>&1 | grep -v amdgpu.ids
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "${EXTRA:-TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum}"; do
  i=$((i+1))
  cd /tmp && COMPRESSJS_AMD_LIB=$R/$LIB timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/r2pmc -o pmc$i --output-format csv -- python $R/tests/gpu_r2_probe.py run $WL --reps 2 > $R/gpurun_out/r2pmc/pmc$i.log 2>&1
done
cd $R
python - "$KERN" <<'PY'
import csv, glob, collections, re, sys
pat = re.compile(sys.argv[1])
for f in sorted(glob.glob('gpurun_out/r2pmc/pmc*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not pat.search(k): continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
    for k in sorted(acc):
        n = len(seen[k])
        print('%s launches=%d %s' % (k, n, ' '.join('%s=%.4g' % (c, v / n) for c, v in sorted(acc[k].items()))))
PY
