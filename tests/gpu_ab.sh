# quick A/B of the suffix-sort variants on cached synthetic streams + per-kernel times (python tests/gpu_deep_probe.py)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
timeout 120 python tests/gpu_deep_probe.py gen
timeout 150 python tests/gpu_deep_probe.py run enwik text sample5 sample3 runs 2>&1 | grep "^.iters"
rm -rf gpurun_out/prof_ab; mkdir -p gpurun_out/prof_ab
for ds in enwik; do
  cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab -o ab_$ds -- python $R/tests/gpu_deep_probe.py run $ds > $R/gpurun_out/prof_ab/$ds.log 2>&1
done
cd $R
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob('gpurun_out/prof_ab/*_results.db')):
    db = sqlite3.connect(f); cur = db.cursor()
    n = cur.execute("select count(*) from kernels where name like 'k34_tables%'").fetchone()[0]
    tot = cur.execute("select sum(end-start) from kernels").fetchone()[0]
    print(f.split('/')[-1], 'steps', n, 'total ms/step %.3f' % (tot / 1e6 / n))
    for r in cur.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc limit 14"):
        print('   %-50s calls/step %5.1f  ms/step %.3f' % (r[0][:50], r[1] / n, r[2] / 1e6 / n))
PY
