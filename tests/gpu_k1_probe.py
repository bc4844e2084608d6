import ctypes as C, numpy as np, sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests','golden'))
import oracle
from compressjs_amd import synth, _lib
L=_lib.load()
def gpu_bwt_batch(blocks, cap, reps=0):
    nb=len(blocks); T=np.zeros(nb*cap,dtype=np.uint8); nl=np.zeros(nb,dtype=np.uint32)
    for i,b in enumerate(blocks): T[i*cap:i*cap+b.size]=b; nl[i]=b.size
    U=np.zeros(nb*cap,dtype=np.uint8); P=np.zeros(nb,dtype=np.uint32); ms=C.c_float(0)
    rc=L.cjs_dbg_bwt_batch_time(T.ctypes.data,nl.ctypes.data,nb,cap,U.ctypes.data,P.ctypes.data,reps,C.byref(ms))
    assert rc==0, rc
    return [U[i*cap:i*cap+nl[i]] for i in range(nb)], P, ms.value
def check(name, blocks, cap, reps=0):
    t=time.time(); us,ps,ms=gpu_bwt_batch(blocks,cap,reps); dt=time.time()-t
    ok=True
    for b,u,p in zip(blocks,us,ps):
        uo,po=oracle.bwt_cyclic(b)
        if po!=p or not (u==uo).all(): ok=False
    tot=sum(b.size for b in blocks)
    print(name, len(blocks), tot, 'OK' if ok else 'MISMATCH', 'wall %.2fs'%dt, 'k1 %.3f ms -> %.1f MB/s'%(ms, tot/1e3/ms if ms>0 else 0), flush=True)
cap=899981
small=[np.frombuffer(s,dtype=np.uint8) for s in [b'banana',b'a',b'abab',b'aaaa',b'bcababa']]
check('small', small, 16)
check('text100k', [synth.text_like(100000,2)], 100000)
check('periodic', [synth.periodic(100001,b'ab')], 100001)
check('runs300k', [synth.runs_mixed(300000,3)], 300000)
tx=synth.text_like(4*cap,7)
check('text4x900k', [tx[i*cap:(i+1)*cap] for i in range(4)], cap, reps=3)
lc=synth.lcg_ascii(2*cap,7)
check('lcg2x900k', [lc[i*cap:(i+1)*cap] for i in range(2)], cap, reps=3)
if len(sys.argv)>1:
    nbig=int(sys.argv[1])
    tx=synth.text_like(nbig*cap,11)
    blocks=[tx[i*cap:(i+1)*cap] for i in range(nbig)]
    us,ps,ms=gpu_bwt_batch(blocks,cap,3)
    print('text batch',nbig,'blocks: k1 %.3f ms -> %.1f MB/s'%(ms, nbig*cap/1e3/ms), flush=True)
    # verify 2 blocks only
    for i in (0,nbig-1):
        uo,po=oracle.bwt_cyclic(blocks[i]); print(' verify',i, po==ps[i] and (us[i]==uo).all(), flush=True)
