"""ctypes driver for the debug stage entry (cjs_dbg_block_stages) of either the real HIP library
(compressjs_amd/libcompressjs_amd.so, GPU) or the CPU logic-debug build (tests/emu/libcjs_emu.so)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "emu", "libcjs_emu.so")
REAL_SO = os.path.join(ROOT, "compressjs_amd", "libcompressjs_amd.so")


class StageOut(C.Structure):
    _fields_ = [("U", C.c_void_p), ("pidx", C.c_void_p), ("A", C.c_void_p), ("pos", C.c_void_p),
                ("alpha", C.c_void_p), ("freq", C.c_void_p), ("used", C.c_void_p),
                ("sel", C.c_void_p), ("lens", C.c_void_p), ("ngroups", C.c_void_p),
                ("nsel", C.c_void_p), ("bitlen", C.c_void_p), ("crc_in", C.c_void_p),
                ("level", C.c_int32), ("stream", C.c_void_p), ("stream_cap", C.c_uint64),
                ("stream_bytes", C.c_uint64)]


def build_emu():
    srcs = [os.path.join(ROOT, "compressjs_amd", "csrc", f)
            for f in os.listdir(os.path.join(ROOT, "compressjs_amd", "csrc"))]
    srcs += [os.path.join(ROOT, "tests", "emu", "emu.cpp"),
             os.path.join(ROOT, "tests", "emu", "hip", "hip_runtime.h")]
    if not os.path.exists(EMU_SO) or os.path.getmtime(EMU_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["sh", os.path.join(ROOT, "tests", "emu", "build_emu.sh")],
                              stdout=subprocess.DEVNULL)
    return EMU_SO


def load(which: str):
    path = build_emu() if which == "emu" else REAL_SO
    if which != "emu":
        import torch  # noqa: F401  (one HIP runtime per process: see compressjs_amd/_lib.py)
    L = C.CDLL(path)
    L.cjs_dbg_block_stages.restype = C.c_int32
    L.cjs_dbg_block_stages.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                       C.POINTER(StageOut)]
    return L


def block_stages(L, blocks, cap: int, upto: int = 5, crcs=None, level: int = 9):
    """Run the device block stages on a list of RLE1 blocks (uint8 arrays, each <= cap)."""
    nb = len(blocks)
    T = np.zeros(nb * cap, dtype=np.uint8)
    nl = np.zeros(nb, dtype=np.uint32)
    for i, b in enumerate(blocks):
        T[i * cap:i * cap + b.size] = b
        nl[i] = b.size
    selp = (cap + 1) // 50 + 2
    stream_cap = nb * (cap * 2 + 8192) + 4096
    arrs = dict(U=np.zeros(nb * cap, np.uint8), pidx=np.zeros(nb, np.uint32),
                A=np.zeros(nb * (cap + 1), np.uint16), pos=np.zeros(nb, np.uint32),
                alpha=np.zeros(nb, np.uint32), freq=np.zeros(nb * 258, np.uint32),
                used=np.zeros(nb * 8, np.uint32), sel=np.zeros(nb * selp, np.uint8),
                lens=np.zeros(nb * 6 * 258, np.uint8), ngroups=np.zeros(nb, np.uint32),
                nsel=np.zeros(nb, np.uint32), bitlen=np.zeros(nb, np.uint64),
                stream=np.zeros(stream_cap, np.uint8))
    o = StageOut()
    for k, a in arrs.items():
        setattr(o, k, a.ctypes.data)
    o.stream_cap = stream_cap
    o.level = level
    crc_arr = None
    if crcs is not None:
        crc_arr = np.ascontiguousarray(crcs, dtype=np.uint32)
        o.crc_in = crc_arr.ctypes.data
    rc = L.cjs_dbg_block_stages(T.ctypes.data, nl.ctypes.data, nb, cap, upto, C.byref(o))
    if rc != 0:
        raise RuntimeError("cjs_dbg_block_stages rc=%d" % rc)
    out = []
    for i in range(nb):
        n = int(nl[i])
        pos = int(arrs["pos"][i])
        nsel = int(arrs["nsel"][i])
        out.append(dict(n=n, U=arrs["U"][i * cap:i * cap + n], pidx=int(arrs["pidx"][i]),
                        A=arrs["A"][i * (cap + 1):i * (cap + 1) + pos], pos=pos,
                        alpha=int(arrs["alpha"][i]), freq=arrs["freq"][i * 258:(i + 1) * 258],
                        used=arrs["used"][i * 8:(i + 1) * 8],
                        selectors=arrs["sel"][i * selp:i * selp + nsel],
                        lens=arrs["lens"][i * 6 * 258:(i + 1) * 6 * 258].reshape(6, 258),
                        ngroups=int(arrs["ngroups"][i]), nsel=nsel, bitlen=int(arrs["bitlen"][i])))
    out[0]["stream"] = arrs["stream"][:int(o.stream_bytes)].tobytes()
    return out


def compare_with_oracle(dev, orc, upto: int):
    """dev: one entry of block_stages(); orc: one dict of oracle.block_stages().  Returns list of
    mismatching stage names."""
    bad = []
    if dev["pidx"] != orc["pidx"] or not np.array_equal(dev["U"], orc["U"]):
        bad.append("K1:bwt")
    if upto >= 2:
        if dev["alpha"] != orc["alphabet_size"]:
            bad.append("K2:alpha")
        if dev["pos"] != orc["pos"] or not np.array_equal(dev["A"], orc["A"]):
            bad.append("K2:A")
        else:
            f = np.bincount(orc["A"], minlength=258).astype(np.uint32)
            if not np.array_equal(dev["freq"], f[:258]):
                bad.append("K2:freq")
    if upto >= 3:
        if dev["ngroups"] != orc["n_groups"] or dev["nsel"] != orc["n_selectors"]:
            bad.append("K3:counts")
        elif not np.array_equal(dev["selectors"], orc["selectors"]):
            bad.append("K4:selectors")
        S = orc["alphabet_size"] + 2
        if not np.array_equal(dev["lens"][:orc["n_groups"], :S], orc["lens"][:orc["n_groups"], :S]):
            bad.append("K3:lens")
    if upto >= 5:
        if dev["bitlen"] != orc["bit_len"]:
            bad.append("K5:bitlen")
    return bad
