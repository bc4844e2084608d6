// K9: un-RLE1 of decoded bzip2 blocks + block CRC for gfx950 (the run logic of _read_bunzip,
// lib/Bzip2.js:419-436, and CRC32.updateCRCRun, lib/CRC32.js:93-99).
//
// The reference keeps (run, previous byte) while it walks the block: the byte after four equal
// bytes is a repeat count, and the byte after a count starts a fresh run.  Whether byte k is a count
// byte therefore depends on everything before it - but only through a 5-state automaton
//   C (fresh: block start or just after a count), 1, 2, 3 (equal bytes seen), 4 (next byte is a count)
// whose transition on byte k depends only on eq_k = (byte k == byte k-1).  Transition functions
// compose associatively, so the states come from a scan:
//   k9_tile_fn     composite function of every 4096-byte tile        (5 states x 3 bits in a u32)
//   k9_tile_state  per block: state entering every tile (<= 220 steps, one lane)
//   k9_tile_len    count-byte bitmap of every tile + decoded bytes per tile
//   k9_block_len   per block: exclusive tile offsets and the block's decoded size
//   k9_expand      output-driven expansion: every output byte finds its source byte by binary
//                  search over the tile's offsets in LDS (coalesced stores, any expansion ratio)
//   k9_crc         CRC of the decoded bytes of every block (1024 slices combined with x^(8m) mod P)
#include "decode.h"
#include "crc_dev.h"

#define ST_C 0u
#define FN_ID (0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12)
__device__ __forceinline__ u32 st_next(u32 s, bool eq) {
    return s == 4u ? ST_C : (s == ST_C ? 1u : (eq ? s + 1u : 1u));
}
__device__ __forceinline__ u32 fn_apply(u32 f, u32 s) { return (f >> (3u * s)) & 7u; }
__device__ __forceinline__ u32 fn_compose(u32 a, u32 b) {          // first a, then b
    u32 r = 0;
#pragma unroll
    for (u32 s = 0; s < 5; s++) r |= fn_apply(b, fn_apply(a, s)) << (3u * s);
    return r;
}

// function of this thread's 16 consecutive bytes [k0, k0+16) of the block
__device__ __forceinline__ u32 thread_fn(const u8* pre, u32 n, u32 k0) {
    u32 st[5] = {0, 1, 2, 3, 4};
    if (k0 < n) {
        u32 prev = k0 ? pre[k0 - 1] : 256u;
        for (u32 k = k0; k < k0 + 16u && k < n; k++) {
            const u32 c = pre[k];
            const bool eq = c == prev;
#pragma unroll
            for (int s = 0; s < 5; s++) st[s] = st_next(st[s], eq);
            prev = c;
        }
    }
    return st[0] | st[1] << 3 | st[2] << 6 | st[3] << 9 | st[4] << 12;
}

// inclusive scan (by composition) of one function per thread over the 256 threads of a block
__device__ __forceinline__ u32 block_fn_scan(u32 f, u32* sh) {
    const u32 tid = threadIdx.x;
    sh[tid] = f;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        const u32 a = tid >= off ? sh[tid - off] : FN_ID;
        __syncthreads();
        sh[tid] = fn_compose(a, sh[tid]);
        __syncthreads();
    }
    return sh[tid];
}

__global__ __launch_bounds__(256) void k9_tile_fn(DecBuf D) {
    const u32 slot = D.slotOf[blockIdx.y];
    const u32 n = D.res[slot].n;
    const u32 t0 = blockIdx.x * DEC_TILE;
    if (t0 >= n) return;
    __shared__ u32 sh[256];
    const u8* pre = D.pre + (size_t)slot * DEC_STRIDE;
    const u32 f = block_fn_scan(thread_fn(pre, n, t0 + threadIdx.x * 16u), sh);
    if (threadIdx.x == 255) D.tileFn[(size_t)slot * DEC_TILES + blockIdx.x] = f;
}

__global__ __launch_bounds__(64) void k9_tile_state(DecBuf D) {
    const u32 slot = D.slotOf[blockIdx.x];
    if (threadIdx.x) return;
    const u32 n = D.res[slot].n;
    const u32 ntiles = (n + DEC_TILE - 1) / DEC_TILE;
    u32 s = ST_C;
    for (u32 t = 0; t < ntiles; t++) {
        D.tileState[(size_t)slot * DEC_TILES + t] = (u8)s;
        s = fn_apply(D.tileFn[(size_t)slot * DEC_TILES + t], s);
    }
}

__global__ __launch_bounds__(256) void k9_tile_len(DecBuf D) {
    const u32 slot = D.slotOf[blockIdx.y];
    const u32 n = D.res[slot].n;
    const u32 t0 = blockIdx.x * DEC_TILE;
    if (t0 >= n) return;
    __shared__ u32 sh[256];
    __shared__ u32 tot;
    const u32 tid = threadIdx.x;
    if (tid == 0) tot = 0;
    const u8* pre = D.pre + (size_t)slot * DEC_STRIDE;
    const u32 k0 = t0 + tid * 16u;
    const u32 f = thread_fn(pre, n, k0);
    const u32 incl = block_fn_scan(f, sh);
    __syncthreads();
    sh[tid] = incl;
    __syncthreads();
    const u32 before = tid ? sh[tid - 1] : FN_ID;
    u32 s = fn_apply(before, D.tileState[(size_t)slot * DEC_TILES + blockIdx.x]);
    u32 bits = 0, len = 0;
    if (k0 < n) {
        u32 prev = k0 ? pre[k0 - 1] : 256u;
        for (u32 k = k0; k < k0 + 16u && k < n; k++) {
            const u32 c = pre[k];
            if (s == 4u) { bits |= 1u << (k - k0); len += c; } else len += 1u;
            s = st_next(s, c == prev);
            prev = c;
        }
    }
    // two threads share one 32-bit word of the bitmap
    const u32 other = __shfl_xor(bits, 1);
    if (!(tid & 1u)) D.isCount[(size_t)slot * (DEC_STRIDE / 32) + (k0 >> 5)] = bits | (other << 16);
    atomicAdd(&tot, len);
    __syncthreads();
    if (tid == 0) D.tileLen[(size_t)slot * DEC_TILES + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k9_block_len(DecBuf D) {
    const u32 slot = D.slotOf[blockIdx.x];
    const u32 n = D.res[slot].n;
    const u32 ntiles = (n + DEC_TILE - 1) / DEC_TILE;
    __shared__ u32 sh[256];
    const u32 tid = threadIdx.x;
    const u32 v = tid < ntiles ? D.tileLen[(size_t)slot * DEC_TILES + tid] : 0u;
    const u32 ex = block_excl_scan_256(v, sh);
    if (tid < ntiles) D.tileLen[(size_t)slot * DEC_TILES + tid] = ex;
    if (tid == 255) D.blkOut[slot] = ex + v;
}

__global__ __launch_bounds__(256) void k9_expand_k(DecBuf D) {
    const u32 kb = blockIdx.y;
    const u32 slot = D.slotOf[kb];
    const u32 n = D.res[slot].n;
    const u32 t0 = blockIdx.x * DEC_TILE;
    if (t0 >= n) return;
    __shared__ u32 off[DEC_TILE + 1];
    __shared__ u8 val[DEC_TILE];
    __shared__ u32 sh[256];
    const u32 tid = threadIdx.x;
    const u8* pre = D.pre + (size_t)slot * DEC_STRIDE;
    const u32* isc = D.isCount + (size_t)slot * (DEC_STRIDE / 32);
    const u32 k0 = t0 + tid * 16u;
    u32 lens[16], tl = 0;
    {
        const u32 bits = k0 < n ? (isc[k0 >> 5] >> (k0 & 31u)) & 0xffffu : 0u;
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            const u32 k = k0 + j;
            u32 l = 0, v = 0;
            if (k < n) {
                const u32 c = pre[k];
                if ((bits >> j) & 1u) { l = c; v = pre[k - 1]; } else { l = 1; v = c; }
            }
            lens[j] = l;
            val[tid * 16u + j] = (u8)v;
            tl += l;
        }
    }
    const u32 base = block_excl_scan_256(tl, sh);
    {
        u32 o = base;
#pragma unroll
        for (u32 j = 0; j < 16; j++) { off[tid * 16u + j] = o; o += lens[j]; }
        if (tid == 255) off[DEC_TILE] = o;
    }
    __syncthreads();
    const u32 total = off[DEC_TILE];
    u8* out = D.out + D.outOff[kb] + D.tileLen[(size_t)slot * DEC_TILES + blockIdx.x];
    for (u32 j = tid; j < total; j += 256u) {
        u32 lo = 0, hi = DEC_TILE;                 // largest k with off[k] <= j
        while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (off[mid] <= j) lo = mid; else hi = mid; }
        out[j] = val[lo];
    }
}

__global__ __launch_bounds__(1024) void k9_crc(DecBuf D) {
    const u32 kb = blockIdx.x;
    const u32 slot = D.slotOf[kb];
    __shared__ u32 tab[CRC_TAB_WORDS];
    __shared__ u32 pw[40];
    __shared__ u32 acc;
    const u64 s = D.outOff[kb], e = s + D.blkOut[slot];
    const u32 c = crc_range_block(D.out, s, e, tab, pw, &acc);
    if (threadIdx.x == 0) D.crcOut[kb] = c;
}

int k9_sizes(DecBuf D, u32 nvalid, hipStream_t stream) {
    if (!nvalid) return CJS_OK;
    hipLaunchKernelGGL(k9_tile_fn, dim3(DEC_TILES, nvalid), dim3(256), 0, stream, D);
    hipLaunchKernelGGL(k9_tile_state, dim3(nvalid), dim3(64), 0, stream, D);
    hipLaunchKernelGGL(k9_tile_len, dim3(DEC_TILES, nvalid), dim3(256), 0, stream, D);
    hipLaunchKernelGGL(k9_block_len, dim3(nvalid), dim3(256), 0, stream, D);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
int k9_expand(DecBuf D, u32 nvalid, hipStream_t stream) {
    if (!nvalid) return CJS_OK;
    hipLaunchKernelGGL(k9_expand_k, dim3(DEC_TILES, nvalid), dim3(256), 0, stream, D);
    hipLaunchKernelGGL(k9_crc, dim3(nvalid), dim3(1024), 0, stream, D);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
