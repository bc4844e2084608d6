# Round-5 PMC pass of selected kernels ($KERN regex, default k1f_bsort) on one workload ($1, default enwik): SQ counters per launch.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); export TMPDIR=/tmp; W=${1:-enwik}; O=$R/gpurun_out/r5pmc; mkdir -p $O
timeout 300 python tests/gpu_r2_probe.py gen $W 2>&1 | grep -v amdgpu.ids
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" ${PMC_EXTRA:-}; do
  i=$((i+1))
  cd /tmp && env CJS_STREAMS=1 $(echo ${VARS:-CJS_NOP=0} | tr ',' ' ') timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O -o ${W}_g$i --output-format csv -- python $R/tests/gpu_r2_probe.py run $W --reps 1 > $O/${W}_g$i.log 2>&1
done
cd $R
python - $W "${KERN:-k1f_bsort}" <<'PY'
import csv, glob, collections, sys, re
W, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in sorted(glob.glob('gpurun_out/r5pmc/%s_g*_counter_collection.csv' % W)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); nd[k].add((f, r['Dispatch_Id']))
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0)):
    if not re.search(kern, k): continue
    a = acc[k]; wv = max(a.get('SQ_WAVES', 1), 1); wc = max(a.get('SQ_WAVE_CYCLES', 1), 1)
    print('%-20s waves %8.0f  per wave: VALU %6.0f SALU %6.0f LDS %5.0f VMEMrd %4.1f wr %4.1f | wave-cycles(quad) %7.0f wait_any %4.1f%% wait_inst %4.1f%% active %4.1f%% | LDS idx cyc/CU %9.0f conflict %4.1f%% | busy %9.0f' % (
        k[:20], wv, a.get('SQ_INSTS_VALU', 0) / wv, a.get('SQ_INSTS_SALU', 0) / wv, a.get('SQ_INSTS_LDS', 0) / wv, a.get('SQ_INSTS_VMEM_RD', 0) / wv, a.get('SQ_INSTS_VMEM_WR', 0) / wv,
        wc / wv, 100 * a.get('SQ_WAIT_ANY', 0) / wc, 100 * a.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * a.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        a.get('SQ_LDS_IDX_ACTIVE', 0) / 256, 100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1), a.get('SQ_BUSY_CYCLES', 0)))
PY
