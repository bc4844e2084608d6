"""Loader of the HIP shared library.  There is NO CPU fallback: if the library is missing or no
MI355X is visible the product raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcompressjs_amd.so")
_lib = None


class CompressjsAmdError(RuntimeError):
    pass


def load(path: str | None = None):
    """dlopen libcompressjs_amd.so (built by __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise CompressjsAmdError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % p)
    L = C.CDLL(p)
    vp, u32p = C.c_void_p, C.POINTER(C.c_uint32)
    L.cjs_bwt_cyclic.restype = C.c_int32
    L.cjs_bwt_cyclic.argtypes = [vp, vp, C.c_uint32, u32p]
    L.cjs_bwt_cyclic_batch.restype = C.c_int32
    L.cjs_bwt_cyclic_batch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    L.cjs_dbg_bwt_batch_time.restype = C.c_int32
    L.cjs_dbg_bwt_batch_time.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_int, C.POINTER(C.c_float)]
    if path is None:
        _lib = L
    return L


def check(rc: int, what: str = "call"):
    if rc >= 0:
        return rc
    if rc == -20:
        raise ValueError("Invalid block size multiplier")
    if rc == -23:
        raise CompressjsAmdError("%s: no HIP device visible (the product has no CPU path)" % what)
    if rc <= -100:
        raise CompressjsAmdError("%s: HIP error %d" % (what, -100 - rc))
    raise CompressjsAmdError("%s failed with code %d" % (what, rc))
