"""Loader of the HIP shared library (C ABI in include/compressjs_amd.h).

There is NO CPU fallback: if libcompressjs_amd.so is missing, or no MI355X is visible, every
product entry point raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COMPRESSJS_AMD_LIB") or os.path.join(_HERE, "libcompressjs_amd.so")   # same override as js/index.js
_lib = None

# every symbol include/compressjs_amd.h declares (tests check the .so exports all of them)
SYMBOLS = ["cjs_create", "cjs_destroy", "cjs_device_count", "cjs_lcg_ascii_device", "cjs_bz2_compress_bound", "cjs_bz2_compress",
           "cjs_bz2_compress_device", "cjs_bz2_compress_multi", "cjs_bz2_plan", "cjs_bz2_plan_block_start", "cjs_bwtc_last_times", "cjs_bz2_plan_scan", "cjs_bz2_plan_cost", "cjs_bz2_plan_phase", "cjs_bz2_plan_chain", "cjs_bz2_encode_blocks", "cjs_bwtc_compress",
           "cjs_bwtc_compress_bound",
           "cjs_last_device_ms", "cjs_last_block_count", "cjs_stream", "cjs_profile_enable",
           "cjs_profile_read", "cjs_profile_read_class", "cjs_bwt_cyclic", "cjs_bwt_cyclic_batch", "cjs_bwt_linear",
           "cjs_suffixsort", "cjs_unbwt_linear", "cjs_huff_lengths", "cjs_huff_lengths_batch",
           "cjs_bz2_decompress", "cjs_bz2_decompress_device", "cjs_bz2_decompress_block", "cjs_bz2_table",
           "cjs_bz2_last_size", "cjs_bz2_fetch", "cjs_shift_bits", "cjs_bwtc_decompress", "cjs_bwtc_last_size", "cjs_bwtc_fetch", "cjs_bz2_last_detail", "cjs_bz2_last_decode_ms",
           "cjs_dbg_bwt_batch_time", "cjs_dbg_block_stages", "cjs_dbg_k1_sparse_rounds",
           "cjs_dbg_k1_rounds", "cjs_dbg_k1_periodic_blocks", "cjs_dbg_rc_div", "cjs_dbg_multi_mallocs", "cjs_dbg_multi_fallbacks", "cjs_dbg_multi_replans"]


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
            "(the product has no CPU fallback)" % p)
    # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  Two runtimes in one
    # process cannot both own the GPU, so when torch is installed let it load its copy first; the
    # dynamic loader then binds this library to the same runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    L = C.CDLL(p)
    vp = C.c_void_p
    L.cjs_create.restype = vp
    L.cjs_create.argtypes = [C.c_int, C.c_uint32]
    L.cjs_lcg_ascii_device.restype = C.c_int32
    L.cjs_lcg_ascii_device.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint64]
    L.cjs_device_count.restype = C.c_int32
    L.cjs_device_count.argtypes = []
    L.cjs_destroy.restype = None
    L.cjs_destroy.argtypes = [vp]
    L.cjs_bz2_compress_bound.restype = C.c_int64
    L.cjs_bz2_compress_bound.argtypes = [C.c_uint64]
    L.cjs_bz2_compress.restype = C.c_int64
    L.cjs_bz2_compress.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64]
    L.cjs_bz2_compress_device.restype = C.c_int64
    L.cjs_bz2_compress_device.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64]
    L.cjs_bwtc_compress_bound.restype = C.c_int64
    L.cjs_bwtc_compress_bound.argtypes = [C.c_uint64]
    L.cjs_bwtc_compress.restype = C.c_int64
    L.cjs_bwtc_compress.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64, C.c_int64]
    L.cjs_bz2_plan.restype = C.c_int64
    L.cjs_bz2_plan.argtypes = [vp, vp, C.c_uint64, C.c_int]
    L.cjs_bz2_plan_block_start.restype = C.c_int64
    L.cjs_bz2_plan_block_start.argtypes = [vp, C.c_uint32]
    L.cjs_bwtc_last_times.restype = C.c_int
    L.cjs_bwtc_last_times.argtypes = [vp, C.POINTER(C.c_float)]
    L.cjs_bz2_plan_scan.restype = C.c_int64
    L.cjs_bz2_plan_scan.argtypes = [vp, vp, C.c_uint64, C.c_int]
    L.cjs_bz2_plan_cost.restype = C.c_int64
    L.cjs_bz2_plan_cost.argtypes = [vp, C.c_uint64]
    L.cjs_bz2_plan_phase.restype = C.c_int64
    L.cjs_bz2_plan_phase.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int]
    L.cjs_bz2_plan_chain.restype = C.c_int64
    L.cjs_bz2_plan_chain.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    L.cjs_bz2_encode_blocks.restype = C.c_int64
    L.cjs_bz2_encode_blocks.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_uint64,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.cjs_last_device_ms.restype = C.c_float
    L.cjs_last_device_ms.argtypes = [vp]
    L.cjs_last_block_count.restype = C.c_uint32
    L.cjs_last_block_count.argtypes = [vp]
    L.cjs_stream.restype = vp
    L.cjs_stream.argtypes = [vp]
    L.cjs_profile_enable.restype = C.c_int32
    L.cjs_profile_enable.argtypes = [vp, C.c_int]
    L.cjs_profile_read_class.restype = C.c_int32
    L.cjs_profile_read_class.argtypes = [vp, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.cjs_profile_read.restype = C.c_int32
    L.cjs_profile_read.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint64)]
    L.cjs_bwt_cyclic.restype = C.c_int32
    L.cjs_bwt_cyclic.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.cjs_bwt_linear.restype = C.c_int32
    L.cjs_bwt_linear.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.cjs_suffixsort.restype = C.c_int32
    L.cjs_suffixsort.argtypes = [vp, vp, C.c_uint32]
    L.cjs_unbwt_linear.restype = C.c_int32
    L.cjs_unbwt_linear.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.cjs_huff_lengths.restype = C.c_int32
    L.cjs_huff_lengths.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.cjs_huff_lengths_batch.restype = C.c_int32
    L.cjs_huff_lengths_batch.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.cjs_bwt_cyclic_batch.restype = C.c_int32
    L.cjs_bwt_cyclic_batch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    L.cjs_dbg_bwt_batch_time.restype = C.c_int32
    L.cjs_dbg_bwt_batch_time.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_int,
                                         C.POINTER(C.c_float)]
    L.cjs_bz2_decompress.restype = C.c_int64
    L.cjs_bz2_decompress.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_int]
    L.cjs_bz2_decompress_device.restype = C.c_int64
    L.cjs_bz2_decompress_device.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_int]
    L.cjs_bz2_decompress_block.restype = C.c_int64
    L.cjs_bz2_decompress_block.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, C.c_uint64]
    L.cjs_bz2_table.restype = C.c_int64
    L.cjs_bz2_table.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, vp, C.c_uint32]
    L.cjs_bz2_last_size.restype = C.c_int64
    L.cjs_bz2_last_size.argtypes = [vp]
    L.cjs_bz2_fetch.restype = C.c_int64
    L.cjs_bz2_fetch.argtypes = [vp, vp, C.c_uint64]
    L.cjs_bz2_last_detail.restype = C.c_int32
    L.cjs_bz2_last_detail.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.cjs_shift_bits.restype = C.c_int32
    L.cjs_shift_bits.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp]
    L.cjs_bz2_compress_multi.restype = C.c_int64
    L.cjs_bz2_compress_multi.argtypes = [C.POINTER(vp), C.c_uint32, vp, C.c_uint64, C.c_int, vp, C.c_uint64]
    L.cjs_bwtc_decompress.restype = C.c_int64
    L.cjs_bwtc_decompress.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_int64)]
    L.cjs_bwtc_last_size.restype = C.c_int64
    L.cjs_bwtc_last_size.argtypes = [vp]
    L.cjs_bwtc_fetch.restype = C.c_int64
    L.cjs_bwtc_fetch.argtypes = [vp, vp, C.c_uint64]
    L.cjs_bz2_last_decode_ms.restype = C.c_float
    L.cjs_bz2_last_decode_ms.argtypes = [vp]
    if path is None:
        _lib = L
    return L


# Bunzip's errors (lib/Bzip2.js:62-88): TypeError with .errorCode, message + optional detail
DECODE_MESSAGES = {-1: "Bad file checksum", -2: "Not bzip data", -3: "Unexpected input EOF", -4: "Unexpected output EOF",
                   -5: "Data error", -6: "Out of memory", -7: "Obsolete (pre 0.9.5) bzip format not supported."}


def raise_decode_error(L, h, rc: int):
    """Raise what the reference's _throw(status, optDetail) raises for a negative decoder code."""
    if rc not in DECODE_MESSAGES:
        check(rc, "cjs_bz2_decompress")
    got, want = C.c_uint32(0), C.c_uint32(0)
    d = L.cjs_bz2_last_detail(h, C.byref(got), C.byref(want))
    detail = {1: "bad magic", 2: "level out of range", 3: "initial position out of bounds",
              4: "Bad block CRC (got %x expected %x)" % (got.value, want.value),
              5: "Bad stream CRC (got %x expected %x)" % (got.value, want.value)}.get(d)
    err = TypeError(DECODE_MESSAGES[rc] + (": " + detail if detail else ""))
    err.errorCode = rc
    raise err


def check(rc: int, what: str = "call") -> int:
    """Map negative C-ABI codes to the exceptions the reference raises."""
    if rc >= 0:
        return rc
    if rc == -20:
        raise ValueError("Invalid block size multiplier")          # lib/Bzip2.js:888-890
    if rc == -21:
        raise CompressjsAmdError("%s: output buffer too small" % what)
    if rc == -30:
        raise RuntimeError("Bad magic")                             # lib/Util.js:150-152
    if rc == -31:
        raise CompressjsAmdError("%s: corrupt or truncated stream" % what)
    if rc == -24:
        raise CompressjsAmdError("%s: not supported by this build (code -24)" % what)
    if rc == -23:
        raise CompressjsAmdError("%s: no HIP device visible (the product has no CPU path)" % what)
    if rc <= -100:
        raise CompressjsAmdError("%s: HIP error %d" % (what, -100 - rc))
    raise CompressjsAmdError("%s failed with code %d" % (what, rc))
