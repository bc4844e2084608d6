cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from compressjs_amd import synth
np.save('/tmp/enwik.npy', synth.enwik_like(100_000_000, 2025))
PY
cat > /tmp/q.py <<'PY'
import sqlite3, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/*_results.db')):
    db = sqlite3.connect(f); cur = db.cursor()
    n = cur.execute("select count(*) from kernels where name like 'k34_tables%'").fetchone()[0]
    tot = cur.execute("select sum(end-start) from kernels").fetchone()[0]
    r = cur.execute("select sum(end-start) from kernels where name like '%k1_deep%'").fetchone()[0]
    print(f.split('/')[-1], 'total ms/step %.3f' % (tot / 1e6 / n), 'deep %.3f' % (r / 1e6 / n))
PY
rm -rf gpurun_out/prof_dbg
for dbg in $DBGS; do
  cd /tmp && CJS_DEEP_DBG=$dbg timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dbg -o d$dbg -- python $GRAFT_REPO_ROOT/tests/gpu_deep_probe.py run enwik > $GRAFT_REPO_ROOT/gpurun_out/d$dbg.log 2>&1
done
python /tmp/q.py $GRAFT_REPO_ROOT/gpurun_out/prof_dbg
