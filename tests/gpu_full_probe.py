import sys, time, hashlib, json, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import stagelib, oracle, cases
which=sys.argv[1]
L=C.CDLL(stagelib.build_emu() if which=='emu' else stagelib.REAL_SO)
L.cjs_create.restype=C.c_void_p; L.cjs_create.argtypes=[C.c_int,C.c_uint32]
L.cjs_bz2_compress.restype=C.c_int64; L.cjs_bz2_compress.argtypes=[C.c_void_p,C.c_void_p,C.c_uint64,C.c_int,C.c_void_p,C.c_uint64]
L.cjs_bz2_compress_bound.restype=C.c_int64; L.cjs_bz2_compress_bound.argtypes=[C.c_uint64]
L.cjs_last_device_ms.restype=C.c_float; L.cjs_last_device_ms.argtypes=[C.c_void_p]
ctx=L.cjs_create(0, int(sys.argv[2])); assert ctx
g=json.load(open('/root/repo/tests/golden/golden.json'))['vectors']
for spec in sys.argv[3:]:
    cid,lv=spec.split(':'); lv=int(lv)
    d=cases.case_input(cid); d=np.ascontiguousarray(d)
    cap=L.cjs_bz2_compress_bound(d.size); out=np.zeros(cap,np.uint8)
    t=time.time(); n=L.cjs_bz2_compress(ctx,d.ctypes.data,d.size,lv,out.ctypes.data,cap); dt=time.time()-t
    o=out[:max(n,0)].tobytes()
    v=g.get('%s:bz2:%d'%(cid,lv))
    ok = v is not None and n==v['out_len'] and hashlib.sha256(o).hexdigest()==v['out_sha256']
    if v is None:
        ref=oracle.bz2_compress(d,lv); ok = (o==ref)
    print(spec, d.size, n, 'OK' if ok else 'MISMATCH', 'wall %.2fs dev %.3f ms'%(dt, L.cjs_last_device_ms(ctx)), flush=True)
