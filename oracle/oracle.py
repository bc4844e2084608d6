"""ctypes binding of oracle/liboracle.so (bz2_oracle.c).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
MAX_SYMS = 258
MAX_GROUPS = 6


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("bz2_oracle.c", "bz2_decode_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


class BlockInfo(C.Structure):
    _fields_ = [("n", C.c_uint32), ("pidx", C.c_uint32), ("alphabet_size", C.c_uint32),
                ("pos", C.c_uint32), ("n_groups", C.c_uint32), ("n_selectors", C.c_uint32),
                ("crc", C.c_uint32), ("reserved", C.c_uint32), ("in_consumed", C.c_uint64),
                ("bit_len", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_suffixsort.restype = C.c_int
        L.orc_suffixsort.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_bwt_cyclic.restype = C.c_int32
        L.orc_bwt_cyclic.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_bwt_linear.restype = C.c_int32
        L.orc_bwt_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_unbwt_linear.restype = None
        L.orc_unbwt_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_huff_lengths.restype = None
        L.orc_huff_lengths.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_huff_lengths64.restype = None
        L.orc_huff_lengths64.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_static_huffman.restype = None
        L.orc_static_huffman.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_canonical.restype = None
        L.orc_canonical.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_read_block.restype = C.c_uint32
        L.orc_read_block.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                     C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_bz2_bound.restype = C.c_int64
        L.orc_bz2_bound.argtypes = [C.c_uint64]
        L.orc_bz2_compress.restype = C.c_int64
        L.orc_bz2_compress.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
        L.orc_bz2_block_stages.restype = C.c_int
        L.orc_bz2_block_stages.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_int,
                                           C.POINTER(BlockInfo), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        L.orc_bz2_decompress.restype = C.c_int64
        L.orc_bz2_decompress.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int,
                                         C.POINTER(C.c_int), C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_bz2_decompress_block.restype = C.c_int64
        L.orc_bz2_decompress_block.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                               C.POINTER(C.c_int)]
        _lib = L
    return _lib


class DBlock(C.Structure):
    _fields_ = [("start_bit", C.c_uint64), ("end_bit", C.c_uint64), ("n", C.c_uint32),
                ("orig_ptr", C.c_uint32), ("crc", C.c_uint32), ("pad", C.c_uint32), ("out_bytes", C.c_uint64)]


DECODE_DETAIL = {0: None, 1: "bad magic", 2: "level out of range", 3: "initial position out of bounds",
                 4: "Bad block CRC", 5: "Bad stream CRC"}


def bz2_decompress(stream, multistream: bool = False, max_blocks: int = 1 << 16):
    """Bunzip.decode (lib/Bzip2.js:454-481) -> (ret, detail, output bytes or None, [(start_bit, out_bytes)])."""
    d = _u8(stream)
    det, nb = C.c_int(0), C.c_uint32(0)
    blocks = (DBlock * max_blocks)()
    probe = np.zeros(1, dtype=np.uint8)
    n = lib().orc_bz2_decompress(_ptr(d), d.size, _ptr(probe), 0, int(multistream), C.byref(det), blocks, max_blocks, C.byref(nb))
    if n < 0:
        return int(n), det.value, None, []
    out = np.zeros(max(int(n), 1), dtype=np.uint8)
    n2 = lib().orc_bz2_decompress(_ptr(d), d.size, _ptr(out), int(n), int(multistream), C.byref(det), blocks, max_blocks, C.byref(nb))
    assert n2 == n
    tab = [(int(blocks[i].start_bit), int(blocks[i].out_bytes)) for i in range(min(nb.value, max_blocks))]
    return int(n), det.value, out[:n].tobytes(), tab


def bz2_decompress_block(stream, bitpos: int, cap: int = 1 << 26):
    """Bunzip.decodeBlock (lib/Bzip2.js:482-503) -> (ret, detail, bytes or None)."""
    d = _u8(stream)
    det = C.c_int(0)
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_bz2_decompress_block(_ptr(d), d.size, int(bitpos), _ptr(out), cap, C.byref(det))
    return (int(n), det.value, out[:n].tobytes()) if n >= 0 else (int(n), det.value, None)


def _u8(a) -> np.ndarray:
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def crc32(data) -> int:
    d = _u8(data)
    return int(lib().orc_crc32(_ptr(d), d.size))


def suffixsort(T) -> np.ndarray:
    t = _u8(T)
    sa = np.zeros(max(t.size, 1), dtype=np.int32)
    lib().orc_suffixsort(_ptr(t), _ptr(sa), t.size)
    return sa[:t.size]


def bwt_cyclic(T):
    """BWT.bwtransform2 -> (U, pidx)"""
    t = _u8(T)
    u = np.zeros(max(t.size, 1), dtype=np.uint8)
    p = lib().orc_bwt_cyclic(_ptr(t), _ptr(u), t.size)
    return u[:t.size], int(p)


def bwt_linear(T):
    """BWT.bwtransform -> (U, pidx)"""
    t = _u8(T)
    u = np.zeros(max(t.size, 1), dtype=np.uint8)
    p = lib().orc_bwt_linear(_ptr(t), _ptr(u), t.size)
    return u[:t.size], int(p)


def unbwt_linear(T, pidx: int) -> np.ndarray:
    t = _u8(T)
    u = np.zeros(max(t.size, 1), dtype=np.uint8)
    lib().orc_unbwt_linear(_ptr(t), _ptr(u), t.size, pidx)
    return u[:t.size]


def huff_lengths(sorted_freq, max_len: int):
    a = np.ascontiguousarray(sorted_freq, dtype=np.int64).copy()
    lib().orc_huff_lengths64(_ptr(a), a.size, max_len)
    return [int(x) for x in a]


def static_huffman(freq) -> np.ndarray:
    f = np.ascontiguousarray(freq, dtype=np.uint32)
    lens = np.zeros(f.size, dtype=np.uint8)
    lib().orc_static_huffman(_ptr(f), f.size, _ptr(lens))
    return lens


def canonical(lens) -> np.ndarray:
    l = np.ascontiguousarray(lens, dtype=np.uint8)
    code = np.zeros(l.size, dtype=np.uint32)
    lib().orc_canonical(_ptr(l), l.size, _ptr(code))
    return code


def read_block(data, in_pos: int, cap: int):
    """readBlock -> (block bytes, new in_pos, crc)"""
    d = _u8(data)
    blk = np.zeros(cap, dtype=np.uint8)
    ip = C.c_uint64(in_pos)
    crc = C.c_uint32(0)
    n = lib().orc_read_block(_ptr(d), d.size, C.byref(ip), _ptr(blk), cap, C.byref(crc))
    return blk[:n], int(ip.value), int(crc.value)


def bz2_compress(data, level: int = 9) -> bytes:
    d = _u8(data)
    cap = int(lib().orc_bz2_bound(d.size))
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_bz2_compress(_ptr(d), d.size, level, _ptr(out), cap)
    if n < 0:
        if n == -1:
            raise ValueError("Invalid block size multiplier")
        raise RuntimeError("oracle failure %d" % n)
    return out[:n].tobytes()


def block_stages(data, level: int = 9):
    """Yield per-block dicts {info, T, U, A, selectors, lens} for every block of the stream."""
    d = _u8(data)
    cap = level * 100000 - 19
    ip = C.c_uint64(0)
    while True:
        info = BlockInfo()
        T = np.zeros(cap, dtype=np.uint8)
        U = np.zeros(cap, dtype=np.uint8)
        A = np.zeros(cap + 1, dtype=np.uint16)
        sel = np.zeros((cap + 1) // 50 + 2, dtype=np.uint8)
        lens = np.zeros((MAX_GROUPS, MAX_SYMS), dtype=np.uint8)
        start = int(ip.value)
        rc = lib().orc_bz2_block_stages(_ptr(d), d.size, C.byref(ip), level, C.byref(info), _ptr(T),
                                        _ptr(U), _ptr(A), _ptr(sel), _ptr(lens))
        if rc == 1:
            return
        if rc != 0:
            raise RuntimeError("oracle block failure")
        yield dict(in_off=start, in_len=int(info.in_consumed), n=int(info.n), pidx=int(info.pidx),
                   alphabet_size=int(info.alphabet_size), pos=int(info.pos),
                   n_groups=int(info.n_groups), n_selectors=int(info.n_selectors),
                   crc=int(info.crc), bit_len=int(info.bit_len), T=T[:info.n], U=U[:info.n],
                   A=A[:info.pos], selectors=sel[:info.n_selectors], lens=lens)
        if info.n < cap:
            return
