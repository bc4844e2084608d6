// K5: block header / selector / code-length-table emission, Huffman bit packing, and stream
// assembly at bit granularity, for gfx950.
//
// Replaces the serial bit-at-a-time writer of the reference:
//   compressBlock header + tables   lib/Bzip2.js:740-741, 749-758, 847-867, StaticHuffman.emit :610-629
//   data emission                   lib/Bzip2.js:869-874, StaticHuffman.encode :631-633
//   stream framing                  lib/Bzip2.js:903-906, 917-919, 925-927
//   BitStream.writeBits/flush       lib/BitStream.js:52-73, 93-105 (MSB first, zero padded)
//
//   k5_header    one workgroup per block builds the header bits in LDS (selectors: parallel MTF over
//                <= 6 tables + unary codes; tables: per-symbol delta codes + scan) -> hdr[b], hbits[b]
//   k5_tilescan  code bits per tile of 4000 symbols (80 groups: the sum of the optimiser's group costs), scanned
//   k5_blockscan per block: tile offsets; across blocks: absolute bit offsets, combined CRC
//   k5_pack      every thread packs 16 consecutive symbols into 32-bit words and ORs them into the
//                stream at its absolute bit position; tile 0 also shifts the header in.
// The stream buffer is kept as big-endian 32-bit words stored byte-swapped, i.e. memory order is
// already the .bz2 byte order.  It must be zero before the first k5_pack of a stream.
#include "pipeline.h"

__device__ __forceinline__ u32 bswap32(u32 v) {
    return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24);
}

// OR `nbits` (<= 32) of `value` into an MSB-first bit array of 32-bit words at bit position `pos`.
__device__ __forceinline__ void put_bits_lds(u32* words, u32 pos, u32 nbits, u32 value) {
    if (nbits == 0) return;
    const u32 wi = pos >> 5, sh = pos & 31u;
    const u64 v = ((u64)value << (64u - nbits)) >> sh;        // left-aligned in 64 bits, then shifted
    atomicOr(&words[wi], (u32)(v >> 32));
    const u32 lo = (u32)v;
    if (lo) atomicOr(&words[wi + 1], lo);
}

// 6-entry MTF list packed 4 bits per entry (entry 0 in the low nibble)
__device__ __forceinline__ u32 mtf6_index(u32 list, u32 s) {
    u32 j = 0;
    while (((list >> (4u * j)) & 15u) != s) j++;
    return j;
}
__device__ __forceinline__ u32 mtf6_front(u32 list, u32 j, u32 s) {
    const u32 lowmask = (1u << (4u * j)) - 1u;                 // entries 0..j-1
    const u32 keep = list & ~((lowmask << 4) | 15u);           // entries above j
    return keep | ((list & lowmask) << 4) | s;
}

#define K5_MAX_SEL 18432         // (= K34_MAX_SEL)
__global__ __launch_bounds__(256) void k5_header(Pipe P) {
    const u32 b = blockIdx.x;
    const u32 n = P.nlen[b];
    const u32 tid = threadIdx.x;
    if (n == 0) { if (tid == 0) P.hbits[b] = 0; return; }
    __shared__ u32 hw[K5_HDR_WORDS];
    __shared__ u32 sh[256];
    __shared__ int lastT[6][256];
    __shared__ u32 s_cursor;
    for (u32 i = tid; i < K5_HDR_WORDS; i += 256) hw[i] = 0;
    __syncthreads();
    const u32 alpha = P.alpha[b], G = P.ngroups[b], nSel = P.nsel[b];
    const int S = (int)alpha + 2;
    // the block's selectors in LDS (k34_run: selPitch <= 18432): each thread walks its chunk of them three times, one after the other - as
    // global byte loads that was 200 dependent round trips per thread (round 6: 75 us of a step's tail for 56 workgroups)
    __shared__ __attribute__((aligned(16))) u8 sel[K5_MAX_SEL];
    {
        const u8* gsel = P.sel + (size_t)b * P.selPitch;
        const u32 nw4 = ((nSel < K5_MAX_SEL ? nSel : K5_MAX_SEL) + 3u) >> 2;
        for (u32 i = tid; i < nw4; i += 256) ((u32*)sel)[i] = ((const u32*)gsel)[i];
    }
    if (tid == 0) {
        // magic, CRC (lib/Bzip2.js:918-919); randomised bit + origPtr (:740-741)
        put_bits_lds(hw, 0, 24, 0x314159u);
        put_bits_lds(hw, 24, 24, 0x265359u);
        put_bits_lds(hw, 48, 32, P.crc[b]);
        put_bits_lds(hw, 80, 1, 0);
        put_bits_lds(hw, 81, 24, P.pidx[b]);
        // used map (:749-758): bit i of the 16-bit map = any of bytes 16i..16i+15 used
        const u32* used8 = P.used + (size_t)b * 8;
        u32 cur = 105, map = 0;
        for (int i = 0; i < 16; i++) {
            const u32 r = (used8[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
            if (r) map |= 1u << (15 - i);
        }
        put_bits_lds(hw, cur, 16, map);
        cur += 16;
        for (int i = 0; i < 16; i++) {
            const u32 r = (used8[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;   // bit j = byte 16i+j used
            if (r) {
                u32 rev = 0;                                   // written MSB first: j = 0 first
                for (int j = 0; j < 16; j++) if (r & (1u << j)) rev |= 1u << (15 - j);
                put_bits_lds(hw, cur, 16, rev);
                cur += 16;
            }
        }
        put_bits_lds(hw, cur, 3, G);                          // :847
        put_bits_lds(hw, cur + 3, 15, nSel);                  // :849
        s_cursor = cur + 18;
    }
    __syncthreads();
    u32 cursor = s_cursor;
    // ---- selectors, MTF over the table numbers, unary (:850-862)
    if (alpha >= G) {
        const u32 chunk = (nSel + 255u) / 256u;
        const u32 lo = tid * chunk < nSel ? tid * chunk : nSel;
        const u32 hi = lo + chunk < nSel ? lo + chunk : nSel;
        int lastp[6];
#pragma unroll
        for (int t = 0; t < 6; t++) lastp[t] = K5_NONE;
        for (u32 i = lo; i < hi; i++) {
            const u32 s = sel[i];
#pragma unroll
            for (int t = 0; t < 6; t++) if (s == (u32)t) lastp[t] = (int)i;
        }
#pragma unroll
        for (int t = 0; t < 6; t++) lastT[t][tid] = lastp[t];
        __syncthreads();
        if (tid < 6) {                                        // exclusive running max per table
            int cur = -1 - (int)tid;                          // initial list 0,1,2,... (:850)
            for (int k = 0; k < 256; k++) {
                const int v = lastT[tid][k];
                lastT[tid][k] = cur;
                cur = cur > v ? cur : v;
            }
        }
        __syncthreads();
        // list at the start of my chunk: tables ordered by last occurrence, most recent first
        u32 list = 0;
        {
            int Ls[6];
#pragma unroll
            for (int t = 0; t < 6; t++) Ls[t] = lastT[t][tid];
#pragma unroll
            for (int t = 0; t < 6; t++) {
                u32 r = 0;
#pragma unroll
                for (int u = 0; u < 6; u++) r += (Ls[u] > Ls[t]) ? 1u : 0u;   // all distinct
                list |= (u32)t << (4u * r);
            }
        }
        const u32 list0 = list;
        u32 bits = 0;
        for (u32 i = lo; i < hi; i++) {
            const u32 s = sel[i];
            const u32 j = mtf6_index(list, s);
            list = mtf6_front(list, j, s);
            bits += j + 1u;
        }
        u32 off = cursor + block_excl_scan_256(bits, sh);
        list = list0;
        for (u32 i = lo; i < hi; i++) {
            const u32 s = sel[i];
            const u32 j = mtf6_index(list, s);
            list = mtf6_front(list, j, s);
            put_bits_lds(hw, off, j + 1u, ((1u << j) - 1u) << 1);
            off += j + 1u;
        }
        __syncthreads();
        if (tid == 255) s_cursor = off;
        __syncthreads();
        cursor = s_cursor;
    } else {
        // The reference reuses its Uint8Array M of length alphabetSize here; with fewer than G
        // slots, stores past the end are dropped and loads past the end never match.  Mirrored
        // serially (tiny alphabets only).
        if (tid == 0) {
            int M[6];
            const int mlen = (int)alpha;
            for (int i = 0; i < mlen; i++) M[i] = i;
            u32 off = cursor;
            for (u32 i = 0; i < nSel; i++) {
                const int s = sel[i];
                int j;
                for (j = 0; j < (int)G; j++) if (j < mlen && M[j] == s) break;
                const int src = j < mlen ? M[j] : 0;
                for (int k = j; k > 0; k--) if (k < mlen) M[k] = M[k - 1];
                if (mlen > 0) M[0] = src;
                put_bits_lds(hw, off, (u32)j + 1u, ((1u << j) - 1u) << 1);
                off += (u32)j + 1u;
            }
            s_cursor = off;
        }
        __syncthreads();
        cursor = s_cursor;
    }
    // ---- code-length tables (StaticHuffman.emit :610-629): 5 bits, then per symbol delta codes
    for (u32 t = 0; t < G; t++) {
        const u8* lens = P.lens + ((size_t)b * CJS_MAX_GROUPS + t) * CJS_LEN_PITCH;
        if (tid == 0) put_bits_lds(hw, cursor, 5, lens[0]);
        // thread owns symbols 2*tid and 2*tid+1
        u32 nb0 = 0, nb1 = 0;
        const int i0 = 2 * (int)tid, i1 = i0 + 1;
        int l0 = 0, lp0 = 0, l1 = 0;
        if (i0 < S) { l0 = lens[i0]; lp0 = i0 ? lens[i0 - 1] : l0; nb0 = 2u * (u32)(l0 > lp0 ? l0 - lp0 : lp0 - l0) + 1u; }
        if (i1 < S) { l1 = lens[i1]; nb1 = 2u * (u32)(l1 > l0 ? l1 - l0 : l0 - l1) + 1u; }
        u32 off = cursor + 5u + block_excl_scan_256(nb0 + nb1, sh);
        if (i0 < S) {
            const u32 val = lp0 < l0 ? 2u : 3u;
            for (u32 d = 0; d + 1u < nb0; d += 2u) put_bits_lds(hw, off + d, 2, val);
            off += nb0;                                       // terminating 0 bit is already there
        }
        if (i1 < S) {
            const u32 val = l0 < l1 ? 2u : 3u;
            for (u32 d = 0; d + 1u < nb1; d += 2u) put_bits_lds(hw, off + d, 2, val);
            off += nb1;
        }
        __syncthreads();
        if (tid == 255) s_cursor = off;                       // thread 255 owns nothing: off = end
        __syncthreads();
        cursor = s_cursor;
    }
    const u32 nw = (cursor + 31u) >> 5;
    u32* out = P.hdr + (size_t)b * K5_HDR_WORDS;
    for (u32 i = tid; i < nw; i += 256) out[i] = hw[i];
    if (tid == 0) P.hbits[b] = cursor;
}

// per block: code bits of every tile of K5_TILE symbols = 80 groups of 50 - the sum of the groups' costs under their tables, which the optimiser's
// last k34_assign left in selCost (round 6: k5_lensum, a pass over the symbols and their selectors that looked every length up again, is gone) -
// and their exclusive scan with the header length as base -> tileBits, bitlen[b]
__global__ __launch_bounds__(256) void k5_tilescan(Pipe P) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.x, tid = threadIdx.x;
    __shared__ u32 sh[256];
    __shared__ u32 carry;
    const u32 nSel = P.nlen[b] ? P.nsel[b] : 0u;
    if (tid == 0) carry = P.nlen[b] ? P.hbits[b] : 0u;
    __syncthreads();
    const u16* cost = P.selCost + (size_t)b * P.selPitch;
    u32* cnt = P.tileBits + (size_t)b * K5_TILES(g);
    for (u32 t0 = 0; t0 < K5_TILES(g); t0 += 256) {
        const u32 t = t0 + tid;
        u32 v = 0;
        const u32 g0 = t * K5_TGROUPS;
        if (g0 + K5_TGROUPS <= nSel) {
            const uint4* c4 = (const uint4*)(cost + g0);           // 80 costs = ten 16-byte loads (g0 * 2 bytes is a multiple of 32)
#pragma unroll
            for (u32 k = 0; k < K5_TGROUPS / 8u; k++) {
                const uint4 q = c4[k];
                v += (q.x & 0xFFFFu) + (q.x >> 16) + (q.y & 0xFFFFu) + (q.y >> 16) + (q.z & 0xFFFFu) + (q.z >> 16) + (q.w & 0xFFFFu) + (q.w >> 16);
            }
        } else for (u32 gi = g0; gi < nSel; gi++) v += cost[gi];
        const u32 ex = block_excl_scan_256(v, sh);
        const u32 base = carry;
        if (t < K5_TILES(g)) cnt[t] = base + ex;
        __syncthreads();
        if (tid == 255) carry = base + ex + v;
        __syncthreads();
    }
    if (tid == 0) P.bitlen[b] = carry;
}

// across the blocks of the batch: absolute bit offsets and the combined CRC (lib/Bzip2.js:917)
__global__ __launch_bounds__(64) void k5_blockscan(Pipe P) {
    if (threadIdx.x != 0) return;
    u64 bits = P.ss->bits;
    u32 crc = P.ss->crc;
    for (u32 b = 0; b < P.g.nb; b++) {
        P.bitoff[b] = bits;
        if (P.nlen[b]) {
            bits += P.bitlen[b];
            crc = ((crc << 1) | (crc >> 31)) ^ P.crc[b];
        }
    }
    if (((bits + 80u + 7u) >> 3) + 8u > P.outCapBytes) P.ss->overflow = 1;
    P.ss->bits = bits;
    P.ss->crc = crc;
    if (P.snap) *P.snap = bits;
}

#define K5_STAGE_WORDS ((K5_TILE * 20u + 62u) / 32u + 2u)    // a tile of K5_TILE symbols, at most 20 bits each, starting anywhere in a word
__device__ __forceinline__ void or_word(u32* out, u64 wi, u32 word) {
    if (word) atomicOr(&out[wi], bswap32(word));
}

__global__ __launch_bounds__(256) void k5_pack(Pipe P) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.y, t = blockIdx.x;
    if (P.nlen[b] == 0 || P.ss->overflow) return;
    const u32 pos = P.pos[b];
    const u32 t0 = t * K5_TILE;
    const u32 tid = threadIdx.x;
    const u64 boff = P.bitoff[b];
    if (t == 0) {
        // shift the header into the stream
        const u32 hb = P.hbits[b];
        const u32 nw = (hb + 31u) >> 5;
        const u32* hdr = P.hdr + (size_t)b * K5_HDR_WORDS;
        const u32 s = (u32)(boff & 31u);
        const u64 w0 = boff >> 5;
        for (u32 k = tid; k <= nw; k += 256) {
            const u32 cur = k < nw ? hdr[k] : 0u;
            const u32 prev = k > 0 ? hdr[k - 1] : 0u;
            const u32 v = s ? ((prev << (32u - s)) | (cur >> s)) : cur;
            or_word(P.out, w0 + k, v);
        }
    }
    if (t0 >= pos) return;
    // The tile's bits are assembled in LDS and leave as whole words: only the first and the last word of a tile are shared with its neighbours
    // (or the header) and need the atomic OR - one per output word from every thread was what the kernel's time went into (round 6: 31-36 ps per word).
    __shared__ u32 codes[CJS_MAX_GROUPS][CJS_LEN_PITCH];       // code | length << 24 (lengths <= 20): ONE lookup per symbol, kept for the second walk
    __shared__ u32 sh[256];
    __shared__ u32 stage[K5_STAGE_WORDS];
    __shared__ u32 s_tot;
    const u32 G = P.ngroups[b];
    for (u32 i = tid; i < G * CJS_LEN_PITCH; i += 256)
        (&codes[0][0])[i] = P.codes[(size_t)b * CJS_MAX_GROUPS * CJS_LEN_PITCH + i] | ((u32)P.lens[(size_t)b * CJS_MAX_GROUPS * CJS_LEN_PITCH + i] << 24);
    __syncthreads();
    const u16* A = P.A + (size_t)b * g.stride;
    const u8* sel = P.sel + (size_t)b * P.selPitch;
    const u32 i0 = tid * 16u < K5_TILE ? t0 + tid * 16u : pos;         // (250 of the 256 threads own symbols)
    // The thread's 16 symbols in two 16-byte loads and its (at most two) selectors in two byte loads
    // (round 3: 16 + 16 two-byte loads 32 bytes apart from lane to lane, twice over, were 64 requests per load instruction).
    u32 aw[8];
#pragma unroll
    for (int k = 0; k < 8; k++) aw[k] = 0;
    const u32 g0 = i0 / CJS_GROUP, gb = (g0 + 1u) * CJS_GROUP;        // symbols below gb use selector g0, the others g0 + 1
    u32 s0 = 0, s1 = 0;
    if (i0 < pos) {
        if (i0 + 16u <= pos) __builtin_memcpy(aw, __builtin_assume_aligned(A + i0, 16), 32);
        else for (u32 k = 0; i0 + k < pos; k++) aw[k >> 1] |= (u32)A[i0 + k] << (16u * (k & 1u));
        s0 = sel[g0];
        if (gb < i0 + 16u && gb < pos) s1 = sel[g0 + 1u];
    }
    u32 e[16];                                                       // 0 behind the block's end: no bits
    u32 mine = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 i = i0 + k;
        e[k] = i < pos ? codes[i < gb ? s0 : s1][(aw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu] : 0u;
        mine += e[k] >> 24;
    }
    const u32 ex = block_excl_scan_256(mine, sh);
    if (tid == 255) s_tot = ex + mine;
    __syncthreads();
    const u64 tbit = boff + P.tileBits[(size_t)b * K5_TILES(g) + t];  // where the tile's first bit goes
    const u64 w0 = tbit >> 5;
    const u32 nw = ((u32)(tbit & 31u) + s_tot + 31u) >> 5;           // words the tile touches (<= K5_STAGE_WORDS)
    for (u32 j = tid; j < nw; j += 256) stage[j] = 0;
    __syncthreads();
    if (mine) {
        const u32 rel = (u32)(tbit & 31u) + ex;                      // the thread's first bit, from the top of word w0
        u32 wi = rel >> 5;
        u32 nacc = rel & 31u;                                        // bits of the current word that belong to earlier symbols
        u64 acc = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 l = e[k] >> 24;
            acc = (acc << l) | (e[k] & 0xFFFFFFu);
            nacc += l;
            if (nacc >= 32u) {
                nacc -= 32u;
                atomicOr(&stage[wi], (u32)(acc >> nacc));
                wi++;
                acc &= (1ull << nacc) - 1ull;
            }
        }
        if (nacc) atomicOr(&stage[wi], (u32)(acc << (32u - nacc)));
    }
    __syncthreads();
    for (u32 j = tid; j < nw; j += 256) {
        const u32 v = stage[j];
        if (j == 0u || j + 1u == nw) or_word(P.out, w0 + j, v);
        else P.out[w0 + j] = bswap32(v);
    }
}

// 'B','Z','h','0'+level (lib/Bzip2.js:903-906); resets the running stream state
__global__ void k5_begin(Pipe P, int level) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (level > 0) P.out[0] = bswap32(0x425A6800u | (u32)('0' + level));
        P.ss->bits = level > 0 ? 32 : 0;              // level <= 0: bare block segment (sharded encode)
        P.ss->crc = 0;
        P.ss->overflow = 0;
    }
}
// end-of-stream magic + combined CRC (lib/Bzip2.js:925-927); zero padding is already there
__global__ void k5_end(Pipe P) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || P.ss->overflow) return;
    const u64 pos = P.ss->bits;
    const u32 parts[3] = {0x177245u, 0x385090u, P.ss->crc};
    const u32 nb[3] = {24, 24, 32};
    u64 p = pos;
    for (int k = 0; k < 3; k++) {
        const u32 sh = (u32)(p & 31u);
        const u64 v = ((u64)parts[k] << (64u - nb[k])) >> sh;
        or_word(P.out, p >> 5, (u32)(v >> 32));
        or_word(P.out, (p >> 5) + 1, (u32)v);
        p += nb[k];
    }
    P.ss->bits = p;
}

int k5_stream_begin(Pipe P, int level, hipStream_t stream, bool zero) {
    if (zero) HIP_CHECK_RET(hipMemsetAsync(P.out, 0, P.outCapBytes, stream));     // (k5_pack ORs its words into the stream)
    hipLaunchKernelGGL(k5_begin, dim3(1), dim3(64), 0, stream, P, level);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

int k5_stream_end(Pipe P, hipStream_t stream) {
    hipLaunchKernelGGL(k5_end, dim3(1), dim3(64), 0, stream, P);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

// `after` (optional): event of the previous batch's k5_blockscan -- bit offsets chain across
// batches; `done` (optional) is recorded right after this batch's k5_blockscan.
int k5_run(Pipe P, u32 max_n, hipStream_t stream, hipEvent_t after, hipEvent_t done, hipEvent_t crc_ready) {
    const BatchGeom g = P.g;
    const u32 tiles = (max_n + 1 + K5_TILE - 1) / K5_TILE;
    if (crc_ready) HIP_CHECK_RET(hipStreamWaitEvent(stream, crc_ready, 0));
    hipLaunchKernelGGL(k5_header, dim3(g.nb), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(k5_tilescan, dim3(g.nb), dim3(256), 0, stream, P);
    if (after) HIP_CHECK_RET(hipStreamWaitEvent(stream, after, 0));
    hipLaunchKernelGGL(k5_blockscan, dim3(1), dim3(64), 0, stream, P);
    if (done) HIP_CHECK_RET(hipEventRecord(done, stream));
    hipLaunchKernelGGL(k5_pack, dim3(tiles, g.nb), dim3(256), 0, stream, P);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

// ---- multi-GPU seam: a rank's segment starts at bit 0; in the assembled stream it starts at bit s ----
// out[0 .. nbytes] = in[0 .. nbytes) shifted right by s (0..7) bits, MSB first (one extra byte of tail)
__global__ __launch_bounds__(256) void k5_shift_bits(const u8* in, u64 nbytes, u32 s, u8* out) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i > nbytes) return;
    const u32 cur = i < nbytes ? in[i] : 0u;
    const u32 prev = i ? in[i - 1] : 0u;
    out[i] = (u8)((cur >> s) | (prev << (8u - s)));
}
int k5_shift_bits_run(const u8* d_in, u64 nbytes, u32 s, u8* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(k5_shift_bits, dim3((u32)((nbytes + 256) / 256)), dim3(256), 0, stream, d_in, nbytes, s & 7u, d_out);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
