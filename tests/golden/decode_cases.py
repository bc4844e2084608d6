"""Catalogue of bzip2 streams for the decoder parity tests.

Every stream is rebuilt deterministically at test time from a recipe:
  * `enc`  : the oracle's encoder (bit-identical to the reference's, pinned by golden.json),
  * `lib`  : libbzip2 through Python's bz2 module (a different encoder: other tables, 6 groups),
  * `fix`  : test/sample*.bz2 of the reference (libbzip2-made; staged under oracle/_ref/fixtures),
then concatenated / truncated / corrupted.  golden_decode.json (make_golden_decode.py) records the
sha256 of each stream and what the REFERENCE's Bzip2.decompressFile did with it.
"""
import bz2
import os

import numpy as np

import cases
import oracle


def _enc(cid, level):
    return oracle.bz2_compress(cases.case_input(cid), level)


def _lib(cid, level):
    return bz2.compress(cases.case_input(cid).tobytes(), level)


def _fix(name):
    for d in cases.FIXTURE_DIRS:
        p = os.path.join(d, name)
        if os.path.exists(p):
            return open(p, "rb").read()
    return None


def _flip(s, bit):
    b = bytearray(s)
    b[bit >> 3] ^= 0x80 >> (bit & 7)
    return bytes(b)


def _setbyte(s, i, v):
    b = bytearray(s)
    b[i] = v
    return bytes(b)


def craft_runs(syms, crc_of=b"b"):
    """A hand-made one-block stream over the alphabet {a, b}: `syms` is a string of 'A' (RUNA), 'B' (RUNB)
    and 'L' (the literal at MTF index 1), end-of-block is appended.  Two tables, all four codes 2 bits long.
    Exercises the reference's int32 `runPos <<= 1` wrap (lib/Bzip2.js:314-347): 32 run symbols bring runPos
    to 0, so the run is dropped or restarted.  The block/stream CRC is that of `crc_of`."""
    bits = []

    def put(n, v):
        for i in range(n - 1, -1, -1):
            bits.append((v >> i) & 1)
    for ch in b"BZh9":
        put(8, ch)
    crc = oracle.crc32(np.frombuffer(crc_of, dtype=np.uint8))
    put(48, 0x314159265359); put(32, crc); put(1, 0); put(24, 0)
    put(16, 1 << (15 - 6)); put(16, (1 << 14) | (1 << 13))           # bytes 0x61, 0x62
    nsym = len(syms) + 1
    put(3, 2); put(15, (nsym + 49) // 50)
    for _ in range((nsym + 49) // 50):
        put(1, 0)                                                     # selector MTF index 0
    for _ in range(2):
        put(5, 2)
        for _ in range(4):
            put(1, 0)
    for ch in syms:
        put(2, {"A": 0, "B": 1, "L": 2}[ch])
    put(2, 3)
    put(48, 0x177245385090); put(32, crc)
    while len(bits) % 8:
        bits.append(0)
    return np.packbits(np.array(bits, dtype=np.uint8)).tobytes()


def streams():
    """-> list of (id, bytes or None (fixture missing), multistream)"""
    out = []

    def add(i, s, ms=False):
        out.append((i, s, ms))

    # valid streams from the reference's own encoder
    for cid, lv in (("empty", 9), ("a1", 9), ("a4", 9), ("a260", 9), ("a1000", 9), ("ab500", 9), ("bytes40", 9),
                    ("text1k", 9), ("text100k", 9), ("text100k", 1), ("lcg250000", 1), ("runs300k", 9),
                    ("zeros300k", 9), ("zeros300k", 1), ("periodic_long", 9), ("periodic_ab_100k", 9), ("lcg99977_a10", 1),
                    ("lcg99976_a300", 1), ("text950k", 9), ("mary9", 9), ("abc_tie", 9)):
        if cid in cases.CASES:
            add("enc:%s:%d" % (cid, lv), _enc(cid, lv))
    # valid streams from libbzip2
    for cid, lv in (("empty", 9), ("a1", 9), ("a1000", 9), ("bytes40", 5), ("text1k", 9), ("text100k", 9),
                    ("text100k", 1), ("lcg250000", 1), ("lcg250000", 9), ("runs300k", 3), ("zeros300k", 9),
                    ("periodic_long", 2), ("text950k", 9), ("text950k", 4)):
        if cid in cases.CASES:
            add("lib:%s:%d" % (cid, lv), _lib(cid, lv))
    for k in range(5):
        add("fix:sample%d" % k, _fix("sample%d.bz2" % k))

    a = _enc("text1k", 9)
    b = _lib("a1000", 3)
    c = _enc("lcg250000", 1)
    # concatenated streams, with and without the multistream flag (lib/Bzip2.js:472-477)
    add("cat:a+b:0", a + b, False)
    add("cat:a+b:1", a + b, True)
    add("cat:b+c+a:1", b + c + a, True)
    add("cat:a+garbage:0", a + b"\x00\x01garbage", False)
    add("cat:a+garbage:1", a + b"\x00\x01garbage", True)
    add("cat:a+BZ:1", a + b"BZ", True)
    add("cat:a+BZh0:1", a + b"BZh0xxxx", True)
    add("cat:a+empty:1", a + _enc("empty", 9), True)
    # degenerate inputs
    add("raw:nothing", b"")
    add("raw:BZh9", b"BZh9")
    add("raw:BZh", b"BZh")
    add("raw:BZx9", b"BZx9" + a[4:])
    add("raw:BZh0", b"BZh0" + a[4:])
    add("raw:BZh:", b"BZh:" + a[4:])
    add("raw:level1-header-on-level9-data", b"BZh1" + _enc("text950k", 9)[4:])
    add("raw:level1-header-small", b"BZh1" + a[4:])
    # truncations: bits past EOF read as zeros (lib/BitStream.js:84)
    for n in (5, 9, 10, 14, 15, 17, 20, 40, len(a) - 11, len(a) - 10, len(a) - 9, len(a) - 5, len(a) - 1):
        add("trunc:text1k:%d" % n, a[:n])
    for n in (1000, 83137 // 2, len(c) - 4, len(c) - 12):
        add("trunc:lcg250000:%d" % n, c[:n])
    # truncation exactly at the end of a block (byte granularity): uses the block table
    _, _, _, tab = oracle.bz2_decompress(c)
    for (bitpos, _n) in tab[1:]:
        add("trunc:lcg250000:block@%d" % bitpos, c[:(bitpos + 7) >> 3])
        add("trunc:lcg250000:block@%d-1" % bitpos, c[:((bitpos + 7) >> 3) - 1])
        add("trunc:lcg250000:block@%d+6" % bitpos, c[:((bitpos + 7) >> 3) + 6])
    # corruptions
    add("flip:magic", _flip(a, 32 + 5))
    add("flip:blockcrc", _flip(a, 32 + 48 + 3))
    add("flip:randomised", _flip(a, 32 + 80))
    add("flip:origptr-high", _flip(a, 32 + 81))
    add("flip:origptr-low", _flip(a, 32 + 81 + 23))
    add("flip:streamcrc", _flip(a, len(a) * 8 - 9))
    add("flip:eos-magic", _flip(a, len(a) * 8 - 50))
    rng = np.random.RandomState(99)
    for k in range(24):
        bit = int(rng.randint(32 + 105, len(a) * 8 - 90))
        add("flip:text1k:%d" % bit, _flip(a, bit))
    for k in range(10):
        bit = int(rng.randint(32 + 105, len(c) * 8 - 90))
        add("flip:lcg250000:%d" % bit, _flip(c, bit))
    big = _lib("text100k", 9)
    for k in range(8):
        bit = int(rng.randint(32 + 105, len(big) * 8 - 90))
        add("flip:lib:text100k:%d" % bit, _flip(big, bit))
    add("setbyte:groups0", _setbyte(a, 20, 0))
    # runs of 31 / 32 / 33 / 64 / 65 RUNA-RUNB symbols: the reference's runPos wraps to 0 at 32 (lib/Bzip2.js:314-347)
    for name, syms, want in (("31A+L", "A" * 31 + "L", b"b"), ("32A+L", "A" * 32 + "L", b"b"), ("33A+L", "A" * 33 + "L", b"b"),
                             ("32A", "A" * 32, b""), ("32B+L", "B" * 32 + "L", b"b"), ("L+32A", "L" + "A" * 32, b"b"),
                             ("64A+L", "A" * 64 + "L", b"b"), ("65A+L", "A" * 65 + "L", b"b"),
                             ("33mixed+L", "AB" * 16 + "B" + "L", b"b"), ("40A+L+35B", "A" * 40 + "L" + "B" * 35, b"b"),
                             ("3A+L", "AAA" + "L", b"aaab"), ("70mixed", "ABBA" * 17 + "AB", b"b"), ("L+33A", "L" + "A" * 33, b"bb")):
        add("craft:runs:%s" % name, craft_runs(syms, want))
    return out


BLOCK_CASES = [("fix:sample0", 32), ("fix:sample2", 544888), ("fix:sample4", 32), ("fix:sample4", 1596228),
               ("fix:sample4", 2342106), ("enc:lcg250000:1", 32), ("enc:text1k:9", 32), ("enc:text1k:9", 33),
               ("enc:text1k:9", 40)]
