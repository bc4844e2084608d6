"""Differential fuzz of the GPU decoder against the oracle's decoder (itself pinned to the reference on
138 reference-made outcomes): valid streams of random small inputs, then bit flips, truncations, byte
insertions and concatenations.  Used by the CPU debug build test and by the GPU test."""
import ctypes as C

import numpy as np

import oracle


def mutate(rng, s):
    b = bytearray(s)
    kind = rng.randint(0, 6)
    if kind == 0 and len(b) > 8:                      # flip 1..3 bits anywhere after the header
        for _ in range(rng.randint(1, 4)):
            bit = rng.randint(32, len(b) * 8)
            b[bit >> 3] ^= 0x80 >> (bit & 7)
    elif kind == 1 and len(b) > 5:                    # truncate
        del b[rng.randint(4, len(b)):]
    elif kind == 2:                                   # append garbage / a second stream
        b += bytes(rng.randint(0, 256, size=rng.randint(1, 12)).tolist())
    elif kind == 3 and len(b) > 12:                   # overwrite a byte in the block header area
        b[rng.randint(4, min(len(b), 40))] = rng.randint(0, 256)
    elif kind == 4 and len(b) > 12:                   # insert a byte
        b.insert(rng.randint(4, len(b)), rng.randint(0, 256))
    else:                                             # flip one bit in the last 10 bytes (stream CRC / end magic)
        if len(b) > 10:
            bit = rng.randint((len(b) - 10) * 8, len(b) * 8)
            b[bit >> 3] ^= 0x80 >> (bit & 7)
    return bytes(b)


def gen_input(rng):
    n = int(rng.choice([0, 1, 2, 5, 17, 60, 200, 700, 2500]))
    style = rng.randint(0, 4)
    if style == 0:
        return rng.randint(0, 256, size=n).astype(np.uint8)
    if style == 1:
        return rng.randint(97, 101, size=n).astype(np.uint8)
    if style == 2:                                    # long runs (RLE1 count bytes, RUNA/RUNB)
        out = []
        while len(out) < n:
            out += [int(rng.randint(0, 4))] * int(rng.choice([1, 2, 4, 5, 9, 255, 256, 300]))
        return np.array(out[:n], dtype=np.uint8)
    return np.frombuffer((b"the quick brown fox " * (n // 20 + 1))[:n], dtype=np.uint8).copy()


def run_gpu(L, h, s, ms):
    d = np.frombuffer(s, dtype=np.uint8) if len(s) else np.zeros(0, np.uint8)
    n = L.cjs_bz2_decompress(h, d.ctypes.data if d.size else None, d.size, None, 0, int(ms))
    det = L.cjs_bz2_last_detail(h, None, None)
    if n == -21:
        n = L.cjs_bz2_last_size(h)
        out = np.zeros(max(n, 1), np.uint8)
        assert L.cjs_bz2_fetch(h, out.ctypes.data, n) == n
        return int(n), 0, out[:n].tobytes()
    if n == 0:
        return 0, 0, b""
    return int(n), int(det), None


def fuzz(L, h, seed, cases):
    rng = np.random.RandomState(seed)
    checked = 0
    for k in range(cases):
        data = gen_input(rng)
        s = oracle.bz2_compress(data, int(rng.randint(1, 10)))
        ms = bool(rng.randint(0, 2))
        if k % 8 == 5:
            # hand-made block with stretches of 1..70 RUNA/RUNB symbols: the int32 runPos wrap of lib/Bzip2.js:314-347
            import decode_cases
            syms = "".join("".join(rng.choice(["A", "B"], size=int(rng.choice([1, 5, 30, 31, 32, 33, 63, 64, 65, 70]))).tolist()) + "L"
                           for _ in range(int(rng.randint(1, 4))))
            s = decode_cases.craft_runs(syms[:-1] if rng.randint(0, 2) else syms, bytes(rng.randint(97, 99, size=int(rng.randint(0, 3))).tolist()))
        if rng.randint(0, 4) == 0:
            s = s + oracle.bz2_compress(gen_input(rng), int(rng.randint(1, 10)))
        for _ in range(int(rng.randint(0, 3))):
            s = mutate(rng, s)
        want = oracle.bz2_decompress(s, ms)
        got = run_gpu(L, h, s, ms)
        exp = (want[0], want[1] if want[0] < 0 else 0, want[2])
        assert got == exp, ("seed %d case %d" % (seed, k), s.hex(), ms, got[:2], exp[:2])
        checked += 1
    return checked
