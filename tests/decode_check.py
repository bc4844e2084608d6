"""Shared checker: one stream of decode_cases through the C ABI vs the reference-made golden record."""
import ctypes as C
import hashlib

import numpy as np


def check_stream(L, h, sid, s, ms, v):
    d = np.frombuffer(s, dtype=np.uint8) if len(s) else np.zeros(0, np.uint8)
    n = L.cjs_bz2_decompress(h, d.ctypes.data if d.size else None, d.size, None, 0, int(ms))
    got, want = C.c_uint32(0), C.c_uint32(0)
    det = L.cjs_bz2_last_detail(h, C.byref(got), C.byref(want))
    if n == -21:                                       # decoded; size known now
        n = L.cjs_bz2_last_size(h)
        out = np.zeros(max(n, 1), np.uint8)
        assert L.cjs_bz2_fetch(h, out.ctypes.data, n) == n, sid
        assert v["ok"] and n == v["out_len"], (sid, n, v)
        assert hashlib.sha256(out[:n].tobytes()).hexdigest() == v["out_sha256"], sid
        return
    if n == 0:
        assert v["ok"] and v["out_len"] == 0, (sid, v)
        return
    assert not v["ok"] and v["error_code"] == n, (sid, n, det, v)
    exp = {1: "bad magic", 2: "level out of range", 3: "initial position out of bounds",
           4: "Bad block CRC (got %x expected %x)" % (got.value, want.value),
           5: "Bad stream CRC (got %x expected %x)" % (got.value, want.value)}.get(det)
    if exp:
        assert v["message"].endswith(": " + exp), (sid, det, v["message"], exp)
    else:
        assert ": " not in v["message"], (sid, det, v["message"])


def check_table(L, h, sid, s, ms, v):
    d = np.frombuffer(s, dtype=np.uint8)
    cap = d.size // 8 + 16
    pos, size = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
    n = L.cjs_bz2_table(h, d.ctypes.data, d.size, int(ms), pos.ctypes.data, size.ctypes.data, cap)
    assert n == len(v["table"]), (sid, n)
    assert [[int(pos[i]), int(size[i])] for i in range(n)] == v["table"], sid


def check_block(L, h, sid, s, bitpos, v):
    d = np.frombuffer(s, dtype=np.uint8)
    n = L.cjs_bz2_decompress_block(h, d.ctypes.data, d.size, int(bitpos), None, 0)
    if n == -21:
        n = L.cjs_bz2_last_size(h)
        out = np.zeros(max(n, 1), np.uint8)
        assert L.cjs_bz2_fetch(h, out.ctypes.data, n) == n
        assert v["ok"] and n == v["out_len"] and hashlib.sha256(out[:n].tobytes()).hexdigest() == v["out_sha256"], (sid, bitpos)
    elif n == 0:
        assert v["ok"] and v["out_len"] == 0
    else:
        assert not v["ok"] and v["error_code"] == n, (sid, bitpos, n, v)
