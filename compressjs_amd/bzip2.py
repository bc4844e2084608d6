"""Host-side mirror of the reference's module API for the bzip2 path.

    compressjs.Bzip2.compressFile(input, [output], [level])      lib/Bzip2.js:879-929
    compressjs.BWT.bwtransform2(T, U, n, [k]) -> pidx            lib/BWT.js:372-417

with the input/output coercions of lib/Util.js:9-103 translated to Python objects.  The primary
host binding for the reference's own language is js/ (N-API); this module is the same thin layer
for Python callers, bench.py and the tests.  Every call goes through the C ABI
(include/compressjs_amd.h) into the HIP kernels; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

EOF_BYTE = -1          # Stream.EOF (lib/Stream.js)


class Context:
    """One GPU context (cjs_ctx): HIP stream + HBM workspace for `batch_blocks` blocks."""

    def __init__(self, device: int = 0, batch_blocks: int = 128):
        self.L = _lib.load()
        self.h = self.L.cjs_create(device, batch_blocks)
        if not self.h:
            raise _lib.CompressjsAmdError(
                "cjs_create failed: no MI355X visible or out of HBM (the product has no CPU path)")
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.cjs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # host buffers -------------------------------------------------------------------------
    def _staging(self, cap: int) -> np.ndarray:
        """A host buffer kept across calls: a fresh np.empty() would fault in every page the D2H copy touches."""
        buf = getattr(self, "_obuf", None)
        if buf is None or buf.size < cap:
            buf = np.zeros(cap, dtype=np.uint8)
            self._obuf = buf
        return buf

    def compress(self, data: np.ndarray, level: int = 9) -> bytes:
        d = np.ascontiguousarray(data, dtype=np.uint8)
        cap = int(self.L.cjs_bz2_compress_bound(d.size))
        out = self._staging(cap)
        n = self.L.cjs_bz2_compress(self.h, d.ctypes.data, d.size, int(level), out.ctypes.data, cap)
        _lib.check(n, "cjs_bz2_compress")
        return out[:n].tobytes()

    def bwtc_compress(self, data: np.ndarray, level: int = 9, declared_size=None) -> bytes:
        d = np.ascontiguousarray(data, dtype=np.uint8)
        cap = int(self.L.cjs_bwtc_compress_bound(d.size))
        out = np.empty(cap, dtype=np.uint8)
        n = self.L.cjs_bwtc_compress(self.h, d.ctypes.data, d.size, int(level), out.ctypes.data, cap,
                                     d.size if declared_size is None else declared_size)
        _lib.check(n, "cjs_bwtc_compress")
        return out[:n].tobytes()

    def decompress(self, stream, multistream: bool = False) -> bytes:
        """Bzip2.decompressFile on host buffers; raises the reference's TypeError(.errorCode) on bad input."""
        d = np.ascontiguousarray(stream, dtype=np.uint8)
        n = self.L.cjs_bz2_decompress(self.h, d.ctypes.data, d.size, None, 0, int(bool(multistream)))
        if n == -21:                                   # size now known: fetch the retained result
            n = int(self.L.cjs_bz2_last_size(self.h))
            out = self._staging(max(n, 1))
            _lib.check(self.L.cjs_bz2_fetch(self.h, out.ctypes.data, n), "cjs_bz2_fetch")
            return out[:n].tobytes()
        if n < 0:
            _lib.raise_decode_error(self.L, self.h, int(n))
        return b""

    def decompress_block(self, stream, bitpos: int) -> bytes:
        """Bzip2.decompressBlock (lib/Bzip2.js:482-503)."""
        d = np.ascontiguousarray(stream, dtype=np.uint8)
        n = self.L.cjs_bz2_decompress_block(self.h, d.ctypes.data, d.size, int(bitpos), None, 0)
        if n == -21:
            n = int(self.L.cjs_bz2_last_size(self.h))
            out = self._staging(max(n, 1))
            _lib.check(self.L.cjs_bz2_fetch(self.h, out.ctypes.data, n), "cjs_bz2_fetch")
            return out[:n].tobytes()
        if n < 0:
            _lib.raise_decode_error(self.L, self.h, int(n))
        return b""

    def table(self, stream, multistream: bool = False):
        """Bzip2.table (lib/Bzip2.js:508-548) -> [(bit position, decoded bytes)]."""
        d = np.ascontiguousarray(stream, dtype=np.uint8)
        cap = d.size // 8 + 16
        pos = np.zeros(cap, dtype=np.uint64)
        size = np.zeros(cap, dtype=np.uint64)
        n = self.L.cjs_bz2_table(self.h, d.ctypes.data, d.size, int(bool(multistream)), pos.ctypes.data, size.ctypes.data, cap)
        if n < 0:
            _lib.raise_decode_error(self.L, self.h, int(n))
        return [(int(pos[i]), int(size[i])) for i in range(int(n))]

    def bwtc_decompress(self, stream) -> bytes:
        """BWTC.decompressFile (lib/BWTC.js:141-233): host range decoder, inverse BWT on the GPU (K6)."""
        d = np.ascontiguousarray(stream, dtype=np.uint8)
        declared = C.c_int64(-1)
        n = self.L.cjs_bwtc_decompress(self.h, d.ctypes.data, d.size, None, 0, C.byref(declared))
        if n == -21:
            n = int(self.L.cjs_bwtc_last_size(self.h))
            out = self._staging(max(n, 1))
            _lib.check(self.L.cjs_bwtc_fetch(self.h, out.ctypes.data, n), "cjs_bwtc_fetch")
            return out[:n].tobytes()
        _lib.check(n, "cjs_bwtc_decompress")
        return b""

    def decompress_device(self, d_in, d_out, multistream: bool = False) -> int:
        """Stream and output are torch uint8 CUDA tensors; returns the decoded size."""
        n = self.L.cjs_bz2_decompress_device(self.h, d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), d_out.numel(),
                                             int(bool(multistream)))
        if n < 0 and n != -21:
            _lib.raise_decode_error(self.L, self.h, int(n))
        return _lib.check(n, "cjs_bz2_decompress_device")

    @property
    def last_decode_ms(self) -> float:
        return float(self.L.cjs_bz2_last_decode_ms(self.h))

    # device-resident (torch tensors on this context's GPU) ----------------------------------
    def compress_device(self, d_in, d_out, level: int = 9) -> int:
        """d_in / d_out: torch uint8 CUDA tensors.  Returns the number of bytes written."""
        n = self.L.cjs_bz2_compress_device(self.h, d_in.data_ptr(), d_in.numel(), int(level),
                                           d_out.data_ptr(), d_out.numel())
        return _lib.check(n, "cjs_bz2_compress_device")

    def plan(self, d_in, level: int = 9) -> int:
        """RLE1/block-split pre-pass over the whole input; returns the number of blocks."""
        return _lib.check(self.L.cjs_bz2_plan(self.h, d_in.data_ptr(), d_in.numel(), int(level)),
                          "cjs_bz2_plan")

    def lcg_ascii_device(self, d_out, seed: int, first: int = 0):
        """Fill the torch uint8 CUDA tensor d_out with bytes [first, first + len) of LCG(N, seed), on the device."""
        return _lib.check(self.L.cjs_lcg_ascii_device(self.h, d_out.data_ptr(), d_out.numel(), int(seed) & 0xFFFFFFFF, int(first)),
                          "cjs_lcg_ascii_device")

    # parallel plan of a slice (compressjs_amd/dist.py: sharded_compress_parallel; include/compressjs_amd.h)
    def plan_scan(self, d_in, level: int = 9) -> int:
        """K0's scans over d_in (slice + following margin); returns the input's own RLE1 cost total."""
        return _lib.check(self.L.cjs_bz2_plan_scan(self.h, d_in.data_ptr(), d_in.numel(), int(level)), "cjs_bz2_plan_scan")

    def plan_cost(self, pos: int) -> int:
        return _lib.check(self.L.cjs_bz2_plan_cost(self.h, int(pos)), "cjs_bz2_plan_cost")

    def plan_phase(self, own_len: int, phase: int, last: bool) -> int:
        """Blocks that start in [0, own_len): their number, or -1 when the slice cannot be planned on its own (CJS_E_SPEC)."""
        n = self.L.cjs_bz2_plan_phase(self.h, int(own_len), int(phase), 1 if last else 0)
        if n == -25:
            return -1
        return _lib.check(n, "cjs_bz2_plan_phase")

    def plan_chain(self, own_len: int, t0: int, last: bool):
        """The slice's plan as a link of a chain (round 6): (blocks that start in [0, own_len), target of the first boundary at or
        beyond own_len in this input's own cost prefix), or (-1, t0) when the slice cannot be planned on its own (CJS_E_SPEC)."""
        tn = C.c_uint64(0)
        n = self.L.cjs_bz2_plan_chain(self.h, int(own_len), int(t0), 1 if last else 0, C.byref(tn))
        if n == -25:
            return -1, int(t0)
        return _lib.check(n, "cjs_bz2_plan_chain"), int(tn.value)

    def plan_block_start(self, k: int) -> int:
        """First input byte (relative to the planned input) of block k of the current plan."""
        return _lib.check(self.L.cjs_bz2_plan_block_start(self.h, int(k)), "cjs_bz2_plan_block_start")

    def encode_blocks(self, first: int, count: int, d_seg):
        """Encode blocks [first, first+count) of the planned input into d_seg (bit 0 aligned).
        Returns (bits, crc_fold, n_blocks)."""
        fold, cnt = C.c_uint32(0), C.c_uint32(0)
        bits = self.L.cjs_bz2_encode_blocks(self.h, first, count, d_seg.data_ptr(), d_seg.numel(),
                                            C.byref(fold), C.byref(cnt))
        _lib.check(bits, "cjs_bz2_encode_blocks")
        return int(bits), int(fold.value), int(cnt.value)

    def shift_bits(self, d_seg, nbytes: int, s: int, d_out):
        """d_out[:nbytes+1] = d_seg[:nbytes] shifted right by s bits (torch uint8 tensors on this context's device)."""
        return _lib.check(self.L.cjs_shift_bits(self.h, d_seg.data_ptr(), int(nbytes), int(s), d_out.data_ptr()), "cjs_shift_bits")

    @property
    def last_device_ms(self) -> float:
        return float(self.L.cjs_last_device_ms(self.h))

    @property
    def last_block_count(self) -> int:
        return int(self.L.cjs_last_block_count(self.h))


def compress_multi(contexts, data, level: int = 9) -> bytes:
    """Bzip2.compressFile over several GPUs from one process (cjs_bz2_compress_multi): `contexts` = Context objects on
    different devices; segments of about one batch of blocks go round-robin to them.  Same bytes as one context."""
    d = np.ascontiguousarray(data, dtype=np.uint8)
    L = contexts[0].L
    cap = int(L.cjs_bz2_compress_bound(d.size))
    out = contexts[0]._staging(cap)
    hs = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    n = L.cjs_bz2_compress_multi(hs, len(contexts), d.ctypes.data, d.size, int(level), out.ctypes.data, cap)
    _lib.check(n, "cjs_bz2_compress_multi")
    return out[:n].tobytes()


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0, 32)
    return _default_ctx


def _coerce_input(inp) -> np.ndarray:
    """Util.coerceInputStream (lib/Util.js:9-29): buffers are used as they are, stream objects
    are drained through readByte() until it returns -1."""
    if hasattr(inp, "readByte"):
        buf = bytearray()
        while True:
            b = inp.readByte()
            if b == EOF_BYTE:
                break
            buf.append(b & 0xFF)
        return np.frombuffer(bytes(buf), dtype=np.uint8)
    if isinstance(inp, np.ndarray):
        return np.ascontiguousarray(inp, dtype=np.uint8)
    if isinstance(inp, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(inp), dtype=np.uint8)
    return np.asarray(list(inp), dtype=np.uint8)


def _deliver(data: bytes, output):
    """Util.coerceOutputStream (lib/Util.js:85-103): None -> exact-length bytes; object with
    writeByte -> bytes pushed one at a time and the object returned; int -> expected size;
    writable buffer -> filled, size must match."""
    if output is None or output is False:
        return data
    if hasattr(output, "writeByte"):
        for b in data:
            output.writeByte(b)
        if hasattr(output, "flush"):
            output.flush()
        return output
    if isinstance(output, int):
        if output != len(data):
            raise TypeError("outputsize does not match decoded input")    # lib/Util.js:69-71
        return data
    mv = memoryview(output)
    if len(mv) != len(data):
        raise TypeError("outputsize does not match decoded input")
    mv[:] = data
    return output


class Bzip2:
    """compressjs.Bzip2 (lib/Bzip2.js:878-933), compress side on MI355X."""

    @staticmethod
    def compressFile(inStream, outStream=None, props=None):
        level = 9
        if isinstance(props, (int, float)) and not isinstance(props, bool):   # lib/Bzip2.js:884-887
            level = props
        if level < 1 or level > 9 or level != int(level):
            raise ValueError("Invalid block size multiplier")                  # :888-890
        data = _coerce_input(inStream)
        return _deliver(default_context().compress(data, int(level)), outStream)

    @staticmethod
    def decompressFile(inStream, outStream=None, multistream=False):
        """Bzip2.decompressFile = Bunzip.decode (lib/Bzip2.js:454-481) on the GPU decoder (K7-K9)."""
        data = _coerce_input(inStream)
        return _deliver(default_context().decompress(data, multistream), outStream)

    @staticmethod
    def decompressBlock(inStream, bitPos, outStream=None):
        """Bzip2.decompressBlock = Bunzip.decodeBlock (lib/Bzip2.js:482-503)."""
        data = _coerce_input(inStream)
        return _deliver(default_context().decompress_block(data, bitPos), outStream)

    @staticmethod
    def table(inStream, callback, multistream=False):
        """Bzip2.table (lib/Bzip2.js:508-548): callback(position in bits, decoded bytes) per block."""
        for pos, size in default_context().table(_coerce_input(inStream), multistream):
            callback(pos, size)


class HuffmanAllocator:
    """require('compressjs/lib/HuffmanAllocator') (lib/HuffmanAllocator.js:199-226)."""

    @staticmethod
    def allocateHuffmanCodeLengths(array, maximumLength):
        """In place, like the reference: ascending weights in, code lengths out."""
        a = np.ascontiguousarray(np.asarray(array, dtype=np.int64))
        _lib.check(_lib.load().cjs_huff_lengths(a.ctypes.data, a.size, int(maximumLength)), "cjs_huff_lengths")
        for i in range(a.size):
            array[i] = int(a[i])

    @staticmethod
    def allocate_many(arrays, maximumLength):
        """`count` independent arrays in one launch; returns a list of lists."""
        off = np.zeros(len(arrays) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(a) for a in arrays])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int64) for a in arrays]) if arrays else np.zeros(0, np.int64))
        _lib.check(_lib.load().cjs_huff_lengths_batch(flat.ctypes.data, off.ctypes.data, len(arrays), int(maximumLength)),
                   "cjs_huff_lengths_batch")
        return [flat[off[k]:off[k + 1]].tolist() for k in range(len(arrays))]


class BWTC:
    """compressjs.BWTC (lib/BWTC.js), compress side, levels 6-9: BWT + MTF/RLE2 on MI355X, the
    adaptive range coder on the host."""

    @staticmethod
    def compressFile(inStream, outStream=None, props=None):
        level = 9
        if isinstance(props, (int, float)) and not isinstance(props, bool) and 1 <= props <= 9:
            level = int(props)                                                   # lib/BWTC.js:16-19
        known = not hasattr(inStream, "readByte") or hasattr(inStream, "size")
        data = _coerce_input(inStream)
        size = data.size if known else -1                                        # lib/Util.js:119-124
        return _deliver(default_context().bwtc_compress(data, level, size), outStream)


    @staticmethod
    def decompressFile(inStream, outStream=None):
        """BWTC.decompressFile (lib/BWTC.js:141-233), levels 6-9."""
        return _deliver(default_context().bwtc_decompress(_coerce_input(inStream)), outStream)


class BWT:
    """compressjs.BWT (lib/BWT.js:302-419): the cyclic transform used by bzip2."""

    @staticmethod
    def bwtransform(T, U, A, n, alphabetSize=256):
        """BWT.bwtransform (lib/BWT.js:328-350); A is the reference's scratch array, unused here."""
        t = np.ascontiguousarray(np.asarray(T)[:n], dtype=np.uint8)
        u = np.zeros(max(n, 1), dtype=np.uint8)
        p = C.c_uint32(0)
        _lib.check(_lib.load().cjs_bwt_linear(t.ctypes.data, u.ctypes.data, n, C.byref(p)), "cjs_bwt_linear")
        U[:n] = u[:n] if isinstance(U, np.ndarray) else bytes(u[:n])
        return int(p.value)

    @staticmethod
    def unbwtransform(T, U, LF, n, pidx):
        """BWT.unbwtransform (lib/BWT.js:352-363); LF is the reference's scratch array, unused."""
        t = np.ascontiguousarray(np.asarray(T)[:n], dtype=np.uint8)
        u = np.zeros(max(n, 1), dtype=np.uint8)
        _lib.check(_lib.load().cjs_unbwt_linear(t.ctypes.data, u.ctypes.data, n, int(pidx)), "cjs_unbwt_linear")
        U[:n] = u[:n] if isinstance(U, np.ndarray) else bytes(u[:n])

    @staticmethod
    def suffixsort(T, SA, n, alphabetSize=256):
        """BWT.suffixsort (lib/BWT.js:305-321)."""
        t = np.ascontiguousarray(np.asarray(T)[:n], dtype=np.uint8)
        sa = np.zeros(max(n, 1), dtype=np.int32)
        _lib.check(_lib.load().cjs_suffixsort(t.ctypes.data, sa.ctypes.data, n), "cjs_suffixsort")
        SA[:n] = sa[:n]
        return 0

    @staticmethod
    def bwtransform2(T, U, n, alphabetSize=256):
        t = np.ascontiguousarray(np.asarray(T)[:n], dtype=np.uint8)
        u = np.zeros(max(n, 1), dtype=np.uint8)
        p = C.c_uint32(0)
        _lib.check(_lib.load().cjs_bwt_cyclic(t.ctypes.data, u.ctypes.data, n, C.byref(p)),
                   "cjs_bwt_cyclic")
        U[:n] = u[:n] if isinstance(U, np.ndarray) else bytes(u[:n])
        return int(p.value)
