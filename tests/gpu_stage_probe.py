import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import stagelib, oracle, cases
which=sys.argv[1]; upto=int(sys.argv[2])
L=stagelib.load(which)
for cid in sys.argv[3:]:
    d=cases.case_input(cid)
    level = 9 if d.size>120000 else 1
    cap=level*100000-19
    orc=list(oracle.block_stages(d, level))
    t=time.time(); dev=stagelib.block_stages(L,[o['T'] for o in orc],cap,upto,crcs=[o['crc'] for o in orc],level=level); dt=time.time()-t
    for i,(dv,o) in enumerate(zip(dev,orc)):
        bad=stagelib.compare_with_oracle(dv,o,upto)
        print(cid,'blk',i,'n',o['n'],'pos',o['pos'],'G',o['n_groups'],'OK' if not bad else bad, round(dt,2), flush=True)
    if upto>=5:
        ref=oracle.bz2_compress(d,level)
        print('   stream', len(dev[0]['stream']), len(ref), 'OK' if dev[0]['stream']==ref else 'STREAM MISMATCH', flush=True)
