cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python tests/gpu_deep_probe.py gen
cat > /tmp/q.py <<'PY'
import sqlite3, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/*_results.db')):
    db = sqlite3.connect(f); cur = db.cursor()
    n = cur.execute("select count(*) from kernels where name like 'k34_tables%'").fetchone()[0]
    tot = cur.execute("select sum(end-start) from kernels").fetchone()[0]
    print(f.split('/')[-1], 'steps', n, 'total ms/step %.3f' % (tot / 1e6 / n))
    for r in cur.execute("select name, count(*), sum(end-start) from kernels where name like '%k1_deep%' or name like 'k1_update%' or name like 'k1_sp_%' or name like 'k1_refine%' group by name order by 3 desc limit 5"):
        print('   %-50s calls/step %5.1f  ms/step %.3f' % (r[0][:50], r[1] / n, r[2] / 1e6 / n))
PY
timeout 150 python tests/gpu_deep_probe.py run enwik text sample5 sample3 runs 2>&1 | grep "^.iters"
rm -rf gpurun_out/prof_dbg
for ds in enwik text; do
  cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dbg -o v11_$ds -- python $GRAFT_REPO_ROOT/tests/gpu_deep_probe.py run $ds > $GRAFT_REPO_ROOT/gpurun_out/v11_$ds.log 2>&1
done
python /tmp/q.py $GRAFT_REPO_ROOT/gpurun_out/prof_dbg
