"""Deterministic synthetic byte streams for parity tests and bench.py.

None of these touch /root/reference (it does not exist on the GPU box).

* ``lcg_ascii``  - SURVEY.md section 8(c)/(d) "LCG(n, seed)": random printable ASCII,
  s <- s*1664525 + 1013904223 (mod 2^32), byte = 32 + ((s >>> 16) mod 95).
  This is BASELINE.json configs[3]'s stream ("Synthetic ... random-ASCII").
* ``text_like``  - "enwik8-shaped" stream for configs[2]: Zipf-distributed pseudo-words,
  wiki/XML-ish markup, and recurring boilerplate passages so that the suffix
  structure has the long-LCP tail real wiki text has (enwik8 itself is not in
  the image and there is no network).
* ``runs_mixed`` - adversarial for RLE1/block splitting: long runs of every
  length class (1..3, 4, 5, 254..260, 1000+) mixed with noise.
* ``periodic``   - short-period data (worst case for suffix sorting; equal
  rotations exercise the descending-index tie rule of BWT.js:372-417).
"""
from __future__ import annotations

import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)


def lcg_ascii(n: int, seed: int) -> np.ndarray:
    """LCG(n, seed) of SURVEY.md 8(c): vectorised closed form of the recurrence."""
    out = np.empty(n, dtype=np.uint8)
    if n == 0:
        return out
    chunk = 1 << 22
    # s_j = A^j * s0 + C * (1 + A + ... + A^(j-1))   (mod 2^32), j = 1..chunk
    with np.errstate(over="ignore"):
        apow = np.cumprod(np.full(chunk, _A, dtype=np.uint32), dtype=np.uint32)  # A^1..A^chunk
        geo = np.empty(chunk, dtype=np.uint32)
        geo[0] = 1
        geo[1:] = np.uint32(1) + np.cumsum(apow[:-1], dtype=np.uint32)           # sum_{k<j} A^k
        cterm = geo * _C
        s = np.uint32(seed & 0xFFFFFFFF)
        pos = 0
        while pos < n:
            m = min(chunk, n - pos)
            st = apow[:m] * s + cterm[:m]
            out[pos:pos + m] = (32 + ((st >> np.uint32(16)) % np.uint32(95))).astype(np.uint8)
            s = st[m - 1]
            pos += m
    return out


class _XorShift:
    """Tiny seeded generator with a fixed, version-independent stream (xorshift64*)
    used so the synthetic text does not depend on numpy's Generator internals."""

    def __init__(self, seed: int):
        self.s = np.uint64((seed * 0x9E3779B97F4A7C15 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF)
        if self.s == 0:
            self.s = np.uint64(0x2545F4914F6CDD1D)

    def u32(self, n: int) -> np.ndarray:
        """n 32-bit values: one xorshift64 stream advanced by a counter-based mix (splitmix64)."""
        with np.errstate(over="ignore"):
            base = self.s
            ctr = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + base
            z = ctr
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            self.s = np.uint64(int(base) + (n + 1) * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
        return (z >> np.uint64(32)).astype(np.uint32)


_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LETTER_W = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4,
                      2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])


def _vocab(rng: _XorShift, nwords: int):
    """Flat byte table of `nwords` pseudo-words (+ trailing separator) and start/len arrays."""
    lens = 2 + (rng.u32(nwords) % np.uint32(9)).astype(np.int64)           # 2..10 letters
    lens[:64] = 1 + (np.arange(64) % 4)                                    # frequent words are short
    cdf = np.cumsum(_LETTER_W) / _LETTER_W.sum()
    total = int(lens.sum())
    letters = _LETTERS[np.searchsorted(cdf, rng.u32(total) / 4294967296.0).clip(0, 25)]
    starts = np.zeros(nwords, dtype=np.int64)
    np.cumsum(lens[:-1], out=starts[1:])
    return letters, starts, lens


def text_like(n: int, seed: int = 1) -> np.ndarray:
    """enwik8-shaped synthetic text: n bytes, deterministic in (n, seed)."""
    rng = _XorShift(seed)
    nwords = 30000
    letters, wstart, wlen = _vocab(rng, nwords)
    # Zipf(1.0)-ish rank distribution via inverse CDF
    ranks = np.arange(1, nwords + 1, dtype=np.float64)
    zcdf = np.cumsum(1.0 / ranks)
    zcdf /= zcdf[-1]
    # separators after a word: mostly space, sometimes punctuation / markup
    seps = [b" ", b" ", b" ", b" ", b" ", b" ", b" ", b" ", b", ", b". ", b" ", b" ", b"\n",
            b" [[", b"]] ", b" ", b"'' ", b" ", b" &quot;", b"; ", b" ", b" ", b" ", b" ",
            b" <ref>", b"</ref> ", b" ", b" ", b" ", b" ", b"|", b" ", b" == ", b" ==\n",
            b" ", b" ", b" ", b" ", b" ", b": ", b" ", b" ", b" (", b") ", b" ", b" ", b" ", b" "]
    sep_flat = np.frombuffer(b"".join(seps), dtype=np.uint8)
    sep_len = np.array([len(s) for s in seps], dtype=np.int64)
    sep_start = np.zeros(len(seps), dtype=np.int64)
    np.cumsum(sep_len[:-1], out=sep_start[1:])
    # boilerplate passages that recur (long repeated substrings, like wiki templates)
    boiler = []
    for k in range(24):
        m = 40 + int(rng.u32(1)[0] % 600)
        widx = np.searchsorted(zcdf, rng.u32(m) / 4294967296.0).clip(0, nwords - 1)
        parts = [b"{{"] if k % 3 == 0 else [b"<page>\n  <title>"]
        for w in widx:
            parts.append(letters[wstart[w]:wstart[w] + wlen[w]].tobytes())
            parts.append(b" ")
        parts.append(b"}}\n" if k % 3 == 0 else b"</title>\n")
        boiler.append(np.frombuffer(b"".join(parts), dtype=np.uint8))

    out = np.empty(n, dtype=np.uint8)
    pos = 0
    words_per_chunk = 1 << 14
    while pos < n:
        r = rng.u32(3 * words_per_chunk)
        w = np.searchsorted(zcdf, r[:words_per_chunk] / 4294967296.0).clip(0, nwords - 1)
        s = (r[words_per_chunk:2 * words_per_chunk] % np.uint32(len(seps))).astype(np.int64)
        cap = (r[2 * words_per_chunk:] % np.uint32(23)) == 0               # capitalise some words
        lw = wlen[w]
        ls = sep_len[s]
        tot = lw + ls
        ends = np.cumsum(tot)
        total = int(ends[-1])
        begins = ends - tot
        buf = np.empty(total, dtype=np.uint8)
        # words
        widx = np.repeat(wstart[w] - begins, lw) + _ranges(begins, lw)
        wpos = _ranges(begins, lw)
        buf[wpos] = letters[widx]
        first = begins[cap]
        buf[first] = buf[first] - np.uint8(32)
        # separators
        sbeg = begins + lw
        spos = _ranges(sbeg, ls)
        sidx = np.repeat(sep_start[s] - sbeg, ls) + spos
        buf[spos] = sep_flat[sidx]
        # splice: chunk text, then one boilerplate passage
        m = min(total, n - pos)
        out[pos:pos + m] = buf[:m]
        pos += m
        if pos < n:
            b = boiler[int(rng.u32(1)[0] % len(boiler))]
            m = min(len(b), n - pos)
            out[pos:pos + m] = b[:m]
            pos += m
    return out


def enwik_like(n: int, seed: int = 1, reuse: float = 1.5) -> np.ndarray:
    """text_like() plus phrase reuse: `reuse` x n bytes are overwritten by copies of earlier
    passages (12..160 bytes, taken up to 700 kB back, i.e. mostly inside the same bzip2 block), which is what gives real wiki text its long repeated
    substrings.  Calibrated on the one hard number known about enwik8 without having it: bzip2 -9 brings its
    10^8 bytes to 29 008 758 (ratio 0.290); `reuse=1.5` (later copies overwrite earlier ones) gives 0.29 here (text_like alone: 0.379, the reference's
    own test/sample5||sample4 tiled: 0.199).  Deterministic in (n, seed, reuse)."""
    out = text_like(n, seed)
    if n < 4096 or reuse <= 0:
        return out
    rng = _XorShift(seed ^ 0x5EED)
    chunk = 1 << 16
    mean_len = 86.0
    per_chunk = max(1, int(reuse * chunk / mean_len))
    for c0 in range(chunk, n, chunk):
        r = rng.u32(3 * per_chunk)
        ln = 12 + (r[:per_chunk] % np.uint32(149)).astype(np.int64)
        dst = c0 + (r[per_chunk:2 * per_chunk] % np.uint32(chunk)).astype(np.int64)
        back = 64 + (r[2 * per_chunk:] % np.uint32(min(c0 - 1, 700000) - 63)).astype(np.int64) if c0 > 128 else None
        if back is None:
            continue
        for k in range(per_chunk):
            d, l = int(dst[k]), int(ln[k])
            src = d - int(back[k])
            if src < 0 or d + l > n:
                continue
            out[d:d + l] = out[src:src + l]
    return out


def _ranges(begins: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """concatenate arange(b, b+l) for each (b, l)."""
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    ends = np.cumsum(lens)
    offs = np.repeat(begins - (ends - lens), lens)
    return offs + np.arange(total, dtype=np.int64)


def runs_mixed(n: int, seed: int = 3) -> np.ndarray:
    """Runs of every RLE1 length class (Bzip2.js:636-667) mixed with literal noise."""
    rng = _XorShift(seed)
    classes = np.array([1, 2, 3, 4, 5, 6, 7, 8, 250, 251, 254, 255, 256, 257, 258, 259, 260, 261,
                        509, 510, 511, 512, 1000, 5000], dtype=np.int64)
    parts = []
    total = 0
    while total < n:
        r = rng.u32(4096)
        ln = classes[(r[:2048] % np.uint32(len(classes))).astype(np.int64)]
        # bias to short runs so the stream is not dominated by the 5000s
        short = (r[2048:] % np.uint32(4)) != 0
        ln = np.where(short, 1 + (r[2048:] >> np.uint32(8)) % np.uint32(6), ln).astype(np.int64)
        sym = (97 + (r[:2048] >> np.uint32(16)) % np.uint32(7)).astype(np.uint8)
        parts.append(np.repeat(sym, ln))
        total += int(ln.sum())
    return np.concatenate(parts)[:n]


def periodic(n: int, period: bytes = b"ab") -> np.ndarray:
    p = np.frombuffer(period, dtype=np.uint8)
    return np.tile(p, n // len(p) + 1)[:n].copy()


def all_bytes(reps: int = 40) -> np.ndarray:
    """bytes 0..255 repeated (SURVEY.md 8(c) KAT 'bytes 0..255 x40')."""
    return np.tile(np.arange(256, dtype=np.uint8), reps)
