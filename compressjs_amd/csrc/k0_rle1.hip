// K0: RLE1 + input-dependent block splitting + per-block CRC, for gfx950.
//
// Replaces readBlock (lib/Bzip2.js:636-667) and CRC32.updateCRC (lib/CRC32.js:72-103, table
// :37-70).  The reference reads one byte at a time into a block of capacity
// cap = level*100000-19 with per-block run state; semantics distilled in SURVEY.md 9.1:
//   * a run of equal input bytes is cut into sub-runs of 255: 4 literals + one count byte
//     (0..251); the count byte is appended as soon as the 4th literal is stored, unless that
//     literal filled the block; a block that fills on the count byte leaves it at 0;
//   * every block starts a fresh run, even in the middle of an input run.
//
// Per input byte j with k = j - max(run start, block start) and sub = k mod 255, the byte
// contributes c = 1 (sub < 3), 2 (sub == 3: literal + count byte) or 0 (absorbed) output bytes.
// With C(i) = sum_{j<i} c(j) over UNCUT runs (a device-wide scan over 4096-byte tiles), the
// output position of any byte of a block is a difference of C values plus a closed-form
// correction g() for the one run the block start may cut.  Block boundaries form a serial chain
// (k0_chain, one workgroup, ~10 us per block); everything else is parallel:
//
//   k0_tile_last / scan(max)  -> run start of every tile's first byte
//   k0_tile_cost / scan(sum)  -> Ctile[t] = C(4096 t)
//   k0_chain                  -> (start, end, length, correction) of every block of the input
//   k0_materialize            -> T (RLE1 output) of the blocks of one batch
//   k0_pad                    -> T_ext wrap-around padding
//   k0_crc                    -> CRC of the input bytes each block consumed: 256 independent
//                                table-driven streams per block, combined with x^(8m) mod P.
#include "pipeline.h"
#include "crc_dev.h"
#include "devutil.h"

#define K0_TILE 4096
#define K0_NONE 0ull          // boundary positions are stored +1 so that 0 means "none"

__device__ __forceinline__ u32 k0_g(u64 k) {          // output bytes before the k-th byte of a fresh run
    const u64 q = k / 255u;
    const u32 r = (u32)(k - q * 255u);
    return (u32)(5u * q) + (r < 4u ? r : 5u);
}
__device__ __forceinline__ u32 k0_c(u64 k) {
    const u32 sub = (u32)(k % 255u);
    return sub < 3u ? 1u : (sub == 3u ? 2u : 0u);
}

// ---- per-tile: last run boundary (position j with j == 0 or in[j] != in[j-1]), stored +1 -------
// 16 consecutive input bytes of a thread: one 16-byte load when the address allows it
__device__ __forceinline__ void load16(const K0Buf& K, u64 j0, u8* b) {
    if (j0 + 16u <= K.in_len && ((((uintptr_t)K.in) + j0) & 15u) == 0) {
        const uint4 v = *(const uint4*)(K.in + j0);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = (u8)(w[k >> 2] >> (8 * (k & 3)));
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = j0 + k < K.in_len ? K.in[j0 + k] : 0;
    }
}

// K0_TPW tiles per workgroup, all loads issued before anything is looked at (one tile per workgroup was 24 415 workgroups of
// one HBM round trip each, 256 lanes then queueing on ONE LDS atomicMax: 99 us for 10^8 bytes at the head of every call).
// A later thread's boundary is a later position, so the tile's last boundary is the answer of its highest lane that has one.
#define K0_TPW 4u
__global__ __launch_bounds__(256) void k0_tile_last(K0Buf K) {
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    __shared__ u64 wl[K0_TPW][4];
    u8 b[K0_TPW][16];
    u8 prev[K0_TPW];
#pragma unroll
    for (u32 q = 0; q < K0_TPW; q++) {
        const u64 j0 = ((u64)blockIdx.x * K0_TPW + q) * K0_TILE + tid * 16u;
        prev[q] = 0;
        if (j0 < K.in_len) {
            load16(K, j0, b[q]);
            if (j0) prev[q] = K.in[j0 - 1];
        }
    }
#pragma unroll
    for (u32 q = 0; q < K0_TPW; q++) {
        const u64 j0 = ((u64)blockIdx.x * K0_TPW + q) * K0_TILE + tid * 16u;
        u64 mine = K0_NONE;
        if (j0 < K.in_len) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u64 j = j0 + k;
                if (j < K.in_len && (j == 0 || b[q][k] != (k ? b[q][k - 1] : prev[q]))) mine = j + 1;
            }
        }
        const u64 has = __ballot(mine != K0_NONE);
        const u64 top = __shfl(mine, has ? 63 - __clzll((long long)has) : 0);
        if (lane == 0) wl[q][w] = has ? top : K0_NONE;
    }
    __syncthreads();
    if (tid < K0_TPW) {
        const u64 t = (u64)blockIdx.x * K0_TPW + tid;
        if (t < K.ntiles) {
            u64 last = K0_NONE;
            for (u32 ww = 0; ww < 4u; ww++) if (wl[tid][ww] != K0_NONE) last = wl[tid][ww];
            K.tileA[t] = last;
            K.tileB[t] = last;                              // (the unscanned copy k0_run_end reads: a device-to-device copy of its own until round 6)
        }
    }
}

// ---- generic 3-phase exclusive scan over u64 (sum or max), chunk = 1024 elements ------------------
template <bool MAX>
__device__ __forceinline__ u64 scan_op(u64 a, u64 b) { return MAX ? (a > b ? a : b) : a + b; }

template <bool MAX>
__global__ __launch_bounds__(256) void k0_scan_local(u64* data, u64* chunkTot, u64 n) {
    __shared__ u64 sh[256];
    const u64 c0 = (u64)blockIdx.x * 1024u;
    const u32 tid = threadIdx.x;
    u64 v[4], run = 0;
    for (int k = 0; k < 4; k++) {
        const u64 i = c0 + tid * 4u + k;
        v[k] = i < n ? data[i] : 0;
    }
    u64 tot = 0;
    for (int k = 0; k < 4; k++) tot = scan_op<MAX>(tot, v[k]);
    sh[tid] = tot;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u64 tv = 0;
        if (tid >= off) tv = sh[tid - off];
        __syncthreads();
        if (tid >= off) sh[tid] = scan_op<MAX>(sh[tid], tv);
        __syncthreads();
    }
    run = tid ? sh[tid - 1] : 0;
    for (int k = 0; k < 4; k++) {
        const u64 i = c0 + tid * 4u + k;
        if (i < n) data[i] = run;
        run = scan_op<MAX>(run, v[k]);
    }
    if (tid == 255) chunkTot[blockIdx.x] = sh[255];
}

template <bool MAX>
__global__ __launch_bounds__(1024) void k0_scan_chunks(u64* chunkTot, u64 nchunks, u64* total, u64* reset) {
    __shared__ u64 sh[1024];
    __shared__ u64 carry;
    const u32 tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (u64 c0 = 0; c0 < nchunks; c0 += 1024) {
        const u64 i = c0 + tid;
        const u64 v = i < nchunks ? chunkTot[i] : 0;
        sh[tid] = v;
        __syncthreads();
        for (u32 off = 1; off < 1024; off <<= 1) {
            u64 tv = 0;
            if (tid >= off) tv = sh[tid - off];
            __syncthreads();
            if (tid >= off) sh[tid] = scan_op<MAX>(sh[tid], tv);
            __syncthreads();
        }
        const u64 ex = scan_op<MAX>(carry, tid ? sh[tid - 1] : 0);
        if (i < nchunks) chunkTot[i] = ex;
        __syncthreads();
        if (tid == 1023) carry = scan_op<MAX>(carry, sh[1023]);
        __syncthreads();
    }
    if (tid == 0 && total) *total = carry;
    if (tid == 0 && reset) *reset = ~0ull;                  // (k0_scans: "no boundary flagged yet" for the chain kernels - a fill launch of its own until round 6)
}

template <bool MAX>
__global__ __launch_bounds__(256) void k0_scan_apply(u64* data, const u64* chunkTot, u64 n) {
    const u64 c0 = (u64)blockIdx.x * 1024u;
    const u64 base = chunkTot[blockIdx.x];
    for (int k = 0; k < 4; k++) {
        const u64 i = c0 + (u64)k * 256u + threadIdx.x;
        if (i < n) data[i] = scan_op<MAX>(base, data[i]);
    }
}

// ---- in-tile helper: every thread owns 16 consecutive bytes of tile t ----------------------------
// Returns, for the thread's first byte, the start of its run (global, uncut), given the tile's
// incoming run start rs_in.  `sh` is 256 u64 of LDS.  Also returns the bytes in b[16].
// hm: bit k = the thread's k-th byte exists (is below in_len) and starts a run.  Everything inside is tile-relative and 32 bits wide
// (round 6: 64-bit positions per byte - compares, the max-scan's shuffles - were a third of the instructions of the kernels that call it).
__device__ __forceinline__ u64 tile_runstarts(const K0Buf& K, u64 t, u64 rs_in, u8* b, u64* sh, u32& hm) {
    const u32 tid = threadIdx.x;
    const u64 j0 = t * K0_TILE + tid * 16u;
    u8 prev = 0;
    if (j0 > 0 && j0 - 1 < K.in_len) prev = K.in[j0 - 1];
    load16(K, j0, b);
    const u32 nv = j0 >= K.in_len ? 0u : (K.in_len - j0 < 16u ? (u32)(K.in_len - j0) : 16u);   // bytes of the thread that exist
    u32 heads = (j0 == 0 || b[0] != prev) ? 1u : 0u;
#pragma unroll
    for (int k = 1; k < 16; k++) heads |= b[k] != b[k - 1] ? 1u << k : 0u;
    heads &= (1u << nv) - 1u;
    hm = heads;
    // 1 + tile-relative index of the thread's last run start, 0 = none; exclusive max-scan over the 256 threads: shuffles inside a wave, one LDS hop across the 4 waves
    u32 v = heads ? tid * 16u + 32u - (u32)__clz((int)heads) : 0u;
    const u32 lane = tid & 63u, w = tid >> 6;
    u32* sh32 = (u32*)sh;
    for (u32 off = 1; off < 64; off <<= 1) {
        const u32 u = (u32)__shfl_up((int)v, off);
        if (lane >= off && u > v) v = u;
    }
    if (lane == 63u) sh32[w] = v;
    __syncthreads();
    u32 before = (u32)__shfl_up((int)v, 1u);              // inclusive of the previous lane = exclusive here
    if (lane == 0) before = 0;
    for (u32 ww = 0; ww < w; ww++) if (sh32[ww] > before) before = sh32[ww];
    __syncthreads();
    return before ? t * K0_TILE + before - 1u : rs_in;
}
__device__ __forceinline__ u64 tile_runstarts(const K0Buf& K, u64 t, u64 rs_in, u8* b, u64* sh) {
    u32 hm;
    return tile_runstarts(K, t, rs_in, b, sh, hm);
}
__device__ __forceinline__ u32 k0_mod255(u64 d) { return (d >> 32) ? (u32)(d % 255u) : (u32)d % 255u; }

// ---- per-tile cost with uncut runs ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k0_tile_cost(K0Buf K) {
    const u64 t = blockIdx.x;
    __shared__ u64 sh[256];
    __shared__ u32 tot;
    if (threadIdx.x == 0) tot = 0;
    u8 b[16];
    const u64 rsin_raw = K.tileA[t];                        // exclusive max-scan: last boundary (+1) before the tile
    const u64 rs_in = rsin_raw != K0_NONE ? rsin_raw - 1 : 0;
    u32 hm;
    const u64 rs = tile_runstarts(K, t, rs_in, b, sh, hm);
    const u64 j0 = t * K0_TILE + threadIdx.x * 16u;
    const u32 nv = j0 >= K.in_len ? 0u : (K.in_len - j0 < 16u ? (u32)(K.in_len - j0) : 16u);
    u32 c = 0, sub = 0;                                   // sub = (j - rs) mod 255, kept incrementally: one division per thread
    if (nv) {
        sub = (hm & 1u) ? 0u : k0_mod255(j0 - rs);
        c = sub < 3u ? 1u : (sub == 3u ? 2u : 0u);
#pragma unroll
        for (u32 k = 1; k < 16; k++) {
            sub = ((hm >> k) & 1u) ? 0u : (sub == 254u ? 0u : sub + 1u);
            c += k < nv ? (sub < 3u ? 1u : (sub == 3u ? 2u : 0u)) : 0u;
        }
    }
    const u32 wsum = wave_incl_scan_dpp(c);               // lane 63: the wave's sum (256 lanes on one LDS atomic queued)
    if ((threadIdx.x & 63u) == 63u) atomicAdd(&tot, wsum);
    __syncthreads();
    if (threadIdx.x == 0) K.tileC[t] = tot;
}

// ---- C(i) for an arbitrary position; workgroup-cooperative (256 threads), all threads get it ---
__device__ u64 k0_evalC(const K0Buf& K, u64 i, u64* sh, u32* sh32) {
    if (i >= K.in_len) return K.tileC[K.ntiles];           // total
    const u64 t = i / K0_TILE;
    const u64 rsin_raw = K.tileA[t];
    const u64 rs_in = rsin_raw != K0_NONE ? rsin_raw - 1 : 0;
    u8 b[16];
    u64 rs = tile_runstarts(K, t, rs_in, b, sh);
    const u64 j0 = t * K0_TILE + threadIdx.x * 16u;
    if (threadIdx.x == 0) sh32[0] = 0;
    __syncthreads();
    u32 c = 0;
    for (int k = 0; k < 16; k++) {
        const u64 j = j0 + k;
        if (j >= i) break;
        if (k == 0 ? (j == 0 || K.in[j - 1] != b[0]) : (b[k] != b[k - 1])) rs = j;
        c += k0_c(j - rs);
    }
    if (c) atomicAdd(&sh32[0], c);
    __syncthreads();
    const u64 r = K.tileC[t] + sh32[0];
    __syncthreads();
    return r;
}

// smallest i in (from, in_len] with C(i) >= target, or in_len+1 when the total is below target
__device__ u64 k0_searchC(const K0Buf& K, u64 target, u64 from, u64* sh, u32* sh32, u64* c_at) {
    const u32 tid = threadIdx.x;
    u64 lo = from / K0_TILE, hi = K.ntiles;                // tiles [lo, hi); Ctile[lo] <= C(from) < target
    while (hi - lo > 256) {
        const u64 step = (hi - lo + 255) / 256;
        const u64 p = lo + (u64)tid * step;
        if (tid == 0) sh32[0] = 0;
        __syncthreads();
        if (p < hi && K.tileC[p] < target) atomicAdd(&sh32[0], 1u);
        __syncthreads();
        const u32 cnt = sh32[0];                          // >= 1 because Ctile[lo] < target
        __syncthreads();
        const u64 nlo = lo + (u64)(cnt - 1) * step;
        hi = nlo + step < hi ? nlo + step : hi;
        lo = nlo;
    }
    if (tid == 0) sh32[0] = 0;
    __syncthreads();
    if (lo + tid < hi && K.tileC[lo + tid] < target) atomicAdd(&sh32[0], 1u);
    __syncthreads();
    const u64 t = lo + sh32[0] - 1;
    __syncthreads();
    // inside tile t
    const u64 rsin_raw = K.tileA[t];
    const u64 rs_in = rsin_raw != K0_NONE ? rsin_raw - 1 : 0;
    u8 b[16];
    u64 rs = tile_runstarts(K, t, rs_in, b, sh);
    const u64 j0 = t * K0_TILE + tid * 16u;
    u32 cs[16], mine = 0;
    for (int k = 0; k < 16; k++) {
        const u64 j = j0 + k;
        cs[k] = 0;
        if (j >= K.in_len) continue;
        if (k == 0 ? (j == 0 || K.in[j - 1] != b[0]) : (b[k] != b[k - 1])) rs = j;
        cs[k] = k0_c(j - rs);
        mine += cs[k];
    }
    // exclusive scan of `mine` over the 256 threads
    u32* s32 = (u32*)sh;
    s32[tid] = mine;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u32 tv = 0;
        if (tid >= off) tv = s32[tid - off];
        __syncthreads();
        if (tid >= off) s32[tid] += tv;
        __syncthreads();
    }
    u64 run = K.tileC[t] + (tid ? s32[tid - 1] : 0);
    __syncthreads();
    unsigned long long* best = (unsigned long long*)sh;
    if (tid == 0) best[0] = ~0ull;
    __syncthreads();
    u64 myhit = ~0ull, myrun = 0;
    for (int k = 0; k < 16; k++) {
        const u64 j = j0 + k;
        if (j >= K.in_len) break;
        run += cs[k];
        if (run >= target && j + 1 > from) {
            myhit = j + 1; myrun = run;
            atomicMin(&best[0], (unsigned long long)(j + 1));
            break;
        }
    }
    __syncthreads();
    const u64 r = best[0];
    if (r != ~0ull && myhit == r) best[1] = myrun;        // C(r), published by the thread that found it
    __syncthreads();
    *c_at = best[1];
    __syncthreads();
    return r == ~0ull ? K.in_len + 1 : r;
}

// first position > s whose byte differs from in[s] (the end of s's run), or in_len
__device__ u64 k0_run_end(const K0Buf& K, u64 s, u64* sh) {
    const u32 tid = threadIdx.x;
    const u8 c = K.in[s];
    unsigned long long* best = (unsigned long long*)sh;
    u64 t = s / K0_TILE;
    for (;;) {
        if (tid == 0) best[0] = ~0ull;
        __syncthreads();
        const u64 j0 = t * K0_TILE + tid * 16u;
        for (int k = 0; k < 16; k++) {
            const u64 j = j0 + k;
            if (j > s && j < K.in_len && K.in[j] != c) { atomicMin(&best[0], (unsigned long long)j); break; }
        }
        __syncthreads();
        const u64 r = best[0];
        __syncthreads();
        if (r != ~0ull) return r;
        // skip tiles that contain no boundary at all
        t++;
        for (;;) {
            if (t >= K.ntiles) return K.in_len;
            if (tid == 0) best[0] = ~0ull;
            __syncthreads();
            const u64 tt = t + tid;
            // tileB[tt] = last boundary (+1) inside tile tt (unscanned copy)
            if (tt < K.ntiles && K.tileB[tt] != K0_NONE) atomicMin(&best[0], (unsigned long long)tt);
            __syncthreads();
            const u64 ft = best[0];
            __syncthreads();
            if (ft != ~0ull) { t = ft; break; }
            t += 256;
        }
    }
}

// ---- the serial chain over blocks -----------------------------------------------------------------
// ---- speculative parallel chain --------------------------------------------------------------------
// If no block boundary cuts a run of 4 or more equal bytes and no boundary byte emits two output
// bytes across it, block j ends exactly where C reaches (j+1)*cap, and all boundaries can be searched
// independently.  One workgroup per boundary computes e_j = min{ i : C(i) >= j*cap } and flags the
// boundary if the assumption fails there; k0_chain then resumes serially from the first flagged
// boundary (usually there is none: 112 searches in parallel instead of a 112-step chain).
__global__ __launch_bounds__(256) void k0_chain_spec(K0Buf K, u32 cap) {
    __shared__ u64 sh[256];
    __shared__ u32 sh32[4];
    const u64 total = K.tileC[K.ntiles];
    const u64 nfull = total / cap;                        // boundaries 1..nfull
    const u64 j = (u64)blockIdx.x + 1;
    if (j > nfull) return;
    u64 ce = 0;
    const u64 e = k0_searchC(K, j * (u64)cap, 0, sh, sh32, &ce);
    if (threadIdx.x == 0) {
        bool bad = e > K.in_len || ce != j * (u64)cap;
        if (!bad && e < K.in_len && e > 0 && K.in[e - 1] == K.in[e]) {
            // the boundary lies inside a run: harmless only if the whole run is shorter than 4
            const u8 c = K.in[e];
            u32 len = 2;
            for (u64 q = e + 1; q < K.in_len && len < 4 && K.in[q] == c; q++) len++;
            for (u64 q = e - 1; q > 0 && len < 4 && K.in[q - 1] == c; q--) len++;
            bad = len >= 4;
        }
        K.specEnd[j] = e;
        K.specC[j] = ce;
        if (bad) atomicMin((unsigned long long*)K.specBad, (unsigned long long)j);
    }
}

__global__ __launch_bounds__(256) void k0_chain(K0Buf K, u32 cap) {
    __shared__ u64 sh[256];
    __shared__ u32 sh32[4];
    // resume after the speculated prefix
    const u64 total = K.tileC[K.ntiles];
    const u64 nfull = total / cap;
    u64 kb0 = *K.specBad < nfull ? *K.specBad : nfull;
    if (kb0 > K.maxBlocks) kb0 = K.maxBlocks;
    // blocks 0 .. kb0-1 straight from the speculation (k0_chain_accept, a launch of its own until round 6)
    for (u64 k = threadIdx.x; k < kb0; k += 256u) {
        const u64 s0 = k ? K.specEnd[k] : 0;
        K.blkStart[k] = s0;
        K.blkEnd[k] = K.specEnd[k + 1];
        K.blkN[k] = cap;
        K.blkAdj[k] = k * (u64)cap;                       // = C(s0)
        K.blkRe[k] = s0;
    }
    u64 s = kb0 ? K.specEnd[kb0] : 0, cnext = kb0 ? K.specC[kb0] : 0;
    bool have_cnext = true;                               // C(s) is known (C(0) = 0)
    u32 kb = (u32)kb0;
    while (s < K.in_len && kb < K.maxBlocks) {
        const bool cut = s > 0 && K.in[s - 1] == K.in[s];
        u64 e = s, adj = 0, re = s;
        u32 n = 0;
        bool done = false;
        u64 pre = 0, cbase = 0;
        if (cut) {
            re = k0_run_end(K, s, sh);
            const u64 L = re - s;
            const u64 gL = k0_g(L);
            if (gL >= cap) {
                // the block ends inside the cut run: smallest k with g(k) >= cap
                const u32 q = cap / 5u, rem = cap % 5u;
                const u64 k = (u64)q * 255u + (rem == 0 ? 0u : (rem <= 3u ? rem : 4u));
                e = s + k;
                const u32 gk = k0_g(k);
                n = gk < cap ? gk : cap;
                adj = 0;
                re = e;                                   // all of the block lies in the closed-form region
                done = true;
            } else if (re >= K.in_len) {                  // the cut run reaches EOF without filling the block
                e = K.in_len;
                n = (u32)gL;
                re = e;
                done = true;
            } else {
                pre = gL;
            }
        }
        if (!done) {
            // C(re); when the block start cuts no run this is C(s) = C(previous block's end)
            cbase = (!cut && have_cnext) ? cnext : k0_evalC(K, re, sh, sh32);
            adj = cbase - pre;                            // OB_s(i) = C(i) - adj for i >= re
            const u64 target = adj + cap;
            u64 ce = 0;
            e = k0_searchC(K, target, re, sh, sh32, &ce);
            if (e > K.in_len) {                           // EOF before the block filled
                e = K.in_len;
                n = (u32)(K.tileC[K.ntiles] - adj);
                have_cnext = false;
            } else {
                const u64 ob = ce - adj;
                n = ob < cap ? (u32)ob : cap;
                cnext = ce;
                have_cnext = true;
            }
        } else {
            have_cnext = false;
        }
        if (threadIdx.x == 0) {
            K.blkStart[kb] = s;
            K.blkEnd[kb] = e;
            K.blkN[kb] = n;
            K.blkAdj[kb] = adj;
            K.blkRe[kb] = re;
        }
        kb++;
        s = e;
        if (n < cap) break;                               // lib/Bzip2.js:922
    }
    if (threadIdx.x == 0) *K.nBlocks = kb;
}

// ---- RLE1 output of the blocks of one batch ---------------------------------------------------------
__global__ __launch_bounds__(256) void k0_materialize(K0Buf K, Pipe P, u32 first_block, u32 cap) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.y;
    const u32 kb = first_block + b;
    const u32 tid = threadIdx.x;
    if (kb >= *K.nBlocks) { if (blockIdx.x == 0 && tid == 0) P.nlen[b] = 0; return; }
    __shared__ u64 sh[256];
    const u64 s = K.blkStart[kb], e = K.blkEnd[kb], adj = K.blkAdj[kb], re = K.blkRe[kb];
    const u32 n = K.blkN[kb];
    if (blockIdx.x == 0 && tid == 0) P.nlen[b] = n;
    u8* T = P.T + (size_t)b * g.tstride;
    const u64 tfirst = s / K0_TILE, tlast = (e - 1) / K0_TILE;
    for (u64 t = tfirst + blockIdx.x; t <= tlast; t += gridDim.x) {
        const u64 rsin_raw = K.tileA[t];
        const u64 rs_in = rsin_raw != K0_NONE ? rsin_raw - 1 : 0;
        u8 by[16];
        u32 hm;
        const u64 rs = tile_runstarts(K, t, rs_in, by, sh, hm);
        const u64 j0 = t * K0_TILE + tid * 16u;
        // the thread's bytes inside the block: k in [sR, eR)
        const u64 eend = e < K.in_len ? e : K.in_len;
        const u32 sR = s <= j0 ? 0u : (s - j0 >= 16u ? 16u : (u32)(s - j0));
        const u32 eR = eend <= j0 ? 0u : (eend - j0 >= 16u ? 16u : (u32)(eend - j0));
        // output position of the thread's first byte
        u32 cs[16], mine = 0, sub = 0;
        bool plain = true;                                // all sixteen bytes are inside the block and none is the 4th .. 255th of a run
        if (sR < eR) {                                    // (j - start of the run, cut by the block start) mod 255 at the first byte inside: one division per thread
            const u32 m = hm & ((2u << sR) - 1u);
            const u64 rsa = m ? j0 + (31u - (u32)__clz((int)m)) : rs;
            const u64 rss = rsa > s ? rsa : s;
            sub = k0_mod255(j0 + sR - rss);
        }
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            cs[k] = 0;
            if (k >= sR && k < eR) {
                if (k > sR) sub = ((hm >> k) & 1u) ? 0u : (sub == 254u ? 0u : sub + 1u);
                cs[k] = sub < 3u ? 1u : (sub == 3u ? 2u : 0u);
                mine += cs[k];
            }
            plain = plain && cs[k] == 1u;
        }
        u32 tile_total;
        const u32 excl = block_excl_scan_1024(mine, (u32*)sh, &tile_total);   // wave shuffles + one LDS hop
        (void)tile_total;
        // output bytes before the first in-block byte of this tile
        const u64 tb = t * K0_TILE > s ? t * K0_TILE : s;
        const u64 ob0 = tb <= re ? (u64)k0_g(tb - s) : K.tileC[t] - adj;   // tb > re implies tile-aligned tb
        u32 ob = (u32)(ob0 + excl);                       // (inside the block: below its capacity)
        // sixteen bytes that all stand for themselves: one (unaligned) 16-byte store
        if (plain && ob + 16u <= n) __builtin_memcpy(T + ob, by, 16);
        else
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            const u64 j = j0 + k;
            if (cs[k]) {
                if (ob < n) T[ob] = by[k];
                if (cs[k] == 2 && ob + 1 < n) {
                    // count byte: equal bytes that follow inside the block, at most 251 - eight at a time (a byte-wise walk is
                    // 251 dependent loads: 125 us per tile of a long run, 7.4 of the 12.5 ms of 5*10^7 zeros)
                    const u8 v = by[k];
                    const u64 q0 = j + 1, qe = e - q0 < 251u ? e : q0 + 251u;
                    u64 q = q0;
                    bool stop = false;
                    while (q < qe && ((size_t)(K.in + q) & 7u)) {
                        if (K.in[q] != v) { stop = true; break; }
                        q++;
                    }
                    if (!stop) {
                        const u64 pat = 0x0101010101010101ull * v;
                        while (q + 8u <= qe) {
                            const u64 w = *(const u64*)(K.in + q) ^ pat;
                            if (w) { q += (u64)((__ffsll((long long)w) - 1) >> 3); stop = true; break; }
                            q += 8u;
                        }
                        if (!stop) while (q < qe && K.in[q] == v) q++;
                    }
                    T[ob + 1] = (u8)(q - q0);
                }
                ob += cs[k];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k0_pad(K0Buf K, Pipe P, u32 first_block) {
    const u32 b = blockIdx.x;
    const u32 n = P.nlen[b];
    if (n == 0) return;
    u8* T = P.T + (size_t)b * P.g.tstride;
    T[n + threadIdx.x] = T[threadIdx.x % n];
    // the CRC's constant term (init 0xffffffff carried through the block's bytes, final complement): k0_crc's parts XOR into it
    if (threadIdx.x == 0) P.crc[b] = ~gf_shift(0xffffffffu, K.blkEnd[first_block + b] - K.blkStart[first_block + b], CRC_PW.v);
}

// ---- CRC -----------------------------------------------------------------------------------------------
// gridDim.x x 1024 slices per block: a slice is a serial chain of dependent LDS look-ups, and a block of long runs consumes tens
// of megabytes of input (5*10^7 zeros are two blocks: 1.5 ms with one workgroup each).  The host picks the parts from the input
// bytes per block of the call (k0_batch): one per 2 MB, at most K0_CRC_PARTS - an ordinary block (0.9 MB) is one workgroup's work
// as before.  (Deciding per block on the device - a grid of 16 parts, 15 of which leave at once - made the kernel 2.4 times slower
// on text: 840 workgroups of 1024 threads and 33 KB of LDS are not free to place even when they do nothing.)
#define K0_CRC_PARTS 16u
__global__ __launch_bounds__(1024) void k0_crc(K0Buf K, Pipe P, u32 first_block) {
    const u32 b = blockIdx.y, kb = first_block + b;
    if (kb >= *K.nBlocks) return;
    const u64 s = K.blkStart[kb], e = K.blkEnd[kb];
    const u32 parts = gridDim.x;
    __shared__ u32 tab[CRC_TAB_WORDS];
    __shared__ u32 pw[40];
    __shared__ u32 acc;
    const u64 part = (((e - s + parts - 1u) / parts) + 15u) & ~(u64)15;
    const u64 ps = s + blockIdx.x * part < e ? s + blockIdx.x * part : e;
    const u64 pe = ps + part < e ? ps + part : e;
    if (ps >= pe) return;                                  // (uniform)
    const u32 c = crc_range_raw(K.in, ps, pe, e, tab, pw, &acc);
    if (threadIdx.x == 0) atomicXor(&P.crc[b], c);
}

// ---- host side ---------------------------------------------------------------------------------------
size_t k0_bytes(u64 in_len, u32 cap) {
    const u64 ntiles = (in_len + K0_TILE - 1) / K0_TILE;
    const u64 nchunks = (ntiles + 1 + 1023) / 1024;
    const u64 maxBlocks = in_len / (cap / 2 + 1) + 2;     // > worst-case number of blocks
    size_t tot = 0;
    tot += 3 * (((ntiles + 2) * 8 + 255) & ~(size_t)255);
    tot += ((nchunks + 1) * 8 + 255) & ~(size_t)255;
    tot += 6 * (((maxBlocks + 2) * 8 + 255) & ~(size_t)255) + ((maxBlocks * 4 + 255) & ~(size_t)255) + 512;
    return tot;
}

void k0_carve(K0Buf& K, const u8* d_in, u64 in_len, u32 cap, void* ws) {
    K.in = d_in;
    K.in_len = in_len;
    K.ntiles = (in_len + K0_TILE - 1) / K0_TILE;
    K.nchunks = (K.ntiles + 1 + 1023) / 1024;
    K.maxBlocks = (u32)(in_len / (cap / 2 + 1) + 2);
    char* p = (char*)ws;
    const size_t ta = ((K.ntiles + 2) * 8 + 255) & ~(size_t)255;
    K.tileA = (u64*)p; p += ta;
    K.tileB = (u64*)p; p += ta;
    K.tileC = (u64*)p; p += ta;
    K.chunk = (u64*)p; p += ((K.nchunks + 1) * 8 + 255) & ~(size_t)255;
    const size_t bb = ((size_t)K.maxBlocks * 8 + 255) & ~(size_t)255;
    K.blkStart = (u64*)p; p += bb;
    K.blkEnd = (u64*)p; p += bb;
    K.blkAdj = (u64*)p; p += bb;
    K.blkRe = (u64*)p; p += bb;
    K.blkN = (u32*)p; p += ((size_t)K.maxBlocks * 4 + 255) & ~(size_t)255;
    const size_t sb = (((size_t)K.maxBlocks + 2) * 8 + 255) & ~(size_t)255;
    K.specEnd = (u64*)p; p += sb;
    K.specC = (u64*)p; p += sb;
    K.nBlocks = (u32*)p; p += 256;
    K.specBad = (u64*)p;
}

// tile scans only: run starts (tileA / tileB) and the cost prefix C at every tile (tileC; tileC[ntiles] = the total)
int k0_scans(K0Buf K, hipStream_t stream) {
    const u32 nt = (u32)K.ntiles, nc = (u32)K.nchunks;
    hipLaunchKernelGGL(k0_tile_last, dim3((nt + K0_TPW - 1u) / K0_TPW), dim3(256), 0, stream, K);
    hipLaunchKernelGGL(k0_scan_local<true>, dim3(nc), dim3(256), 0, stream, K.tileA, K.chunk, K.ntiles);
    hipLaunchKernelGGL(k0_scan_chunks<true>, dim3(1), dim3(1024), 0, stream, K.chunk, K.nchunks, (u64*)nullptr, (u64*)nullptr);
    hipLaunchKernelGGL(k0_scan_apply<true>, dim3(nc), dim3(256), 0, stream, K.tileA, (const u64*)K.chunk, K.ntiles);
    hipLaunchKernelGGL(k0_tile_cost, dim3(nt), dim3(256), 0, stream, K);
    hipLaunchKernelGGL(k0_scan_local<false>, dim3(nc), dim3(256), 0, stream, K.tileC, K.chunk, K.ntiles);
    hipLaunchKernelGGL(k0_scan_chunks<false>, dim3(1), dim3(1024), 0, stream, K.chunk, K.nchunks, K.tileC + K.ntiles, K.specBad);
    hipLaunchKernelGGL(k0_scan_apply<false>, dim3(nc), dim3(256), 0, stream, K.tileC, (const u64*)K.chunk, K.ntiles);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

// whole-input pre-pass: tile scans + block chain.  After it *K.nBlocks and blk* are valid.
int k0_prepass(K0Buf K, u32 cap, hipStream_t stream) {
    if (K.in_len == 0) {
        HIP_CHECK_RET(hipMemsetAsync(K.nBlocks, 0, 4, stream));
        return CJS_OK;
    }
    const int rc = k0_scans(K, stream);
    if (rc) return rc;
    const u32 nspec = (u32)(K.in_len * 5 / 4 / cap + 2 < K.maxBlocks ? K.in_len * 5 / 4 / cap + 2 : K.maxBlocks);
    hipLaunchKernelGGL(k0_chain_spec, dim3(nspec), dim3(256), 0, stream, K, cap);
    hipLaunchKernelGGL(k0_chain, dim3(1), dim3(256), 0, stream, K, cap);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

// ---- the blocks of a SLICE of a longer stream (multi-GPU: every rank plans its own slice at once) --------------
// With G(i) the cost prefix of the WHOLE stream, block k of the stream starts at min{ i : G(i) >= k cap } wherever the
// speculation of k0_chain_spec holds.  A rank that holds bytes [lo, lo + in_len) knows G(lo) from one all_gather of per-slice
// totals and boundary runs (compressjs_amd/dist.py), and G(lo + i) = G(lo) + delta + C(i) with this input's own prefix C
// beyond its first run.  Round 6: a slice is planned from the TARGET t0 of its first boundary - the value its own prefix C has to
// reach there - instead of a phase modulo cap, so that what the speculation cannot express is CARRIED instead of refused: a block
// boundary inside a run of four or more equal bytes (ordinary text has them: indentation, rules of '=' or '-') restarts the run in
// the new block, which moves every later boundary by a few output bytes - on one device k0_chain resumes serially from such a
// boundary, and so does a slice here; what it hands to the next slice is the target of the first boundary it does not own.
//   k0_phase_spec    boundary m = min{ i : C(i) >= t0 + m cap }, one workgroup each; flags what k0_chain_spec flags
//   k0_phase_chain   one workgroup: the blocks that START before own_len (the rest of the input is the margin that completes the
//                    last of them), straight from the speculation up to the first flagged boundary and by the serial chain of
//                    k0_chain from there; *nBlocks = their number, or K0_PHASE_FAIL when the margin is too short for a block or a
//                    run fills a whole block (the caller falls back to the replicated plan); ((u64*)nBlocks)[1] = the target
//                    (in this input's own prefix C) of the first boundary at or beyond own_len - the next slice's t0 after a change of
//                    origin.  `last`: the input ends where the stream ends - the final block may be short (lib/Bzip2.js:922) or
//                    absent (:916).
__global__ __launch_bounds__(256) void k0_phase_spec(K0Buf K, u32 cap, u64 t0) {
    __shared__ u64 sh[256];
    __shared__ u32 sh32[4];
    const u64 total = K.tileC[K.ntiles];
    const u64 m = blockIdx.x;
    const u64 t = t0 + m * (u64)cap;
    if (t > total) { if (threadIdx.x == 0) { K.specEnd[m] = K.in_len + 1; K.specC[m] = 0; } return; }
    u64 ce = 0, e = 0;
    if (t) e = k0_searchC(K, t, 0, sh, sh32, &ce);
    if (threadIdx.x == 0) {
        bool bad = e > K.in_len || ce != t;
        if (!bad && e < K.in_len && e > 0 && K.in[e - 1] == K.in[e]) {
            const u8 c = K.in[e];
            u32 len = 2;
            for (u64 q = e + 1; q < K.in_len && len < 4 && K.in[q] == c; q++) len++;
            for (u64 q = e - 1; q > 0 && len < 4 && K.in[q - 1] == c; q--) len++;
            bad = len >= 4;
        }
        K.specEnd[m] = e;
        K.specC[m] = ce;
        if (bad) atomicMin((unsigned long long*)K.specBad, (unsigned long long)m);
    }
}

__global__ __launch_bounds__(256) void k0_phase_chain(K0Buf K, u32 cap, u64 t0, u64 own_len, u32 last, u32 nbound) {
    __shared__ u64 sh[256];
    __shared__ u32 sh32[4];
    __shared__ u32 nb, fail, stop;
    const u32 tid = threadIdx.x;
    if (tid == 0) { nb = 0; fail = 0; stop = nbound; }
    __syncthreads();
    const u64 total = K.tileC[K.ntiles];
    const u64 badraw = *K.specBad;
    const u32 bad = badraw < (u64)nbound ? (u32)badraw : nbound;          // first boundary the speculation does not carry past
    // ---- the speculated prefix: boundary m starts block m while m < bad, the boundary lies in the window and in front of own_len
    for (u32 m = tid; m < nbound; m += 256u) {
        const u64 t = t0 + (u64)m * cap;
        const u64 s = t > total ? K.in_len + 1 : K.specEnd[m];
        if (m >= bad || t > total || s >= own_len) { atomicMin(&stop, m); continue; }
        if (m >= K.maxBlocks) { atomicOr(&fail, 1u); continue; }
        u64 e, n = cap;
        if (t + cap <= total && m + 1u < nbound) e = K.specEnd[m + 1u];     // (boundary m + 1 <= bad: its position is right even if it is flagged)
        else if (last) { e = K.in_len; n = total - t; }                  // the stream ends inside this block
        else { atomicOr(&fail, 1u); continue; }                          // the margin does not reach the end of the block
        if (n == 0) continue;                                            // (last) nothing after the boundary: no block (lib/Bzip2.js:916)
        K.blkStart[m] = s;
        K.blkEnd[m] = e;
        K.blkN[m] = (u32)n;
        K.blkAdj[m] = t;                                                 // = C(s)
        K.blkRe[m] = s;
        atomicMax(&nb, m + 1u);
    }
    __syncthreads();
    const u32 m0 = stop;
    u64 tnext = t0 + (u64)m0 * cap;                                      // the target of the first boundary this slice does not own
    bool failed = fail != 0u;
    u32 kb = nb;
    __syncthreads();
    // ---- a flagged boundary in front of own_len: the serial chain of k0_chain from there to the end of the slice
    if (!failed && m0 < nbound && m0 == bad && tnext <= total && K.specEnd[m0] < own_len) {
        u64 s = K.specEnd[m0], cnext = K.specC[m0];
        bool have_cnext = true;
        kb = m0;
        for (;;) {
            if (s >= own_len) break;                                      // the next slice's (tnext is the target that found s)
            if (kb >= K.maxBlocks) { failed = true; break; }
            const bool cut = s > 0 && K.in[s - 1] == K.in[s];
            u64 e = s, adj = 0, re = s, pre = 0;
            u32 n = 0;
            bool ends = false;                                            // the stream ends inside this block
            if (cut) {
                re = k0_run_end(K, s, sh);
                const u64 gL = k0_g(re - s);
                if (gL >= cap) { failed = true; break; }                  // a run that fills a block: not a slice's business
                if (re >= K.in_len) {
                    if (!last) { failed = true; break; }
                    e = K.in_len; n = (u32)gL; re = e; ends = true;
                } else pre = gL;
            }
            if (!ends) {
                const u64 cbase = (!cut && have_cnext) ? cnext : k0_evalC(K, re, sh, sh32);
                adj = cbase - pre;
                const u64 target = adj + cap;
                u64 ce = 0;
                e = k0_searchC(K, target, re, sh, sh32, &ce);
                if (e > K.in_len) {
                    if (!last) { failed = true; break; }                  // the margin does not reach the end of the block
                    e = K.in_len; n = (u32)(total - adj); ends = true;
                } else {
                    n = cap;
                    cnext = ce;
                    have_cnext = true;
                    tnext = target;
                }
            }
            if (n) {
                if (tid == 0) { K.blkStart[kb] = s; K.blkEnd[kb] = e; K.blkN[kb] = n; K.blkAdj[kb] = adj; K.blkRe[kb] = re; }
                kb++;
            }
            if (ends) { tnext = total + cap; break; }                    // no boundary behind the stream's last block: a target no window reaches
            s = e;
        }
    }
    if (tid == 0) {
        *K.nBlocks = failed ? K0_PHASE_FAIL : kb;
        ((u64*)K.nBlocks)[1] = tnext;
    }
}

__global__ __launch_bounds__(256) void k0_eval_at(K0Buf K, u64 pos) {
    __shared__ u64 sh[256];
    __shared__ u32 sh32[4];
    const u64 c = k0_evalC(K, pos, sh, sh32);
    if (threadIdx.x == 0) K.specC[0] = c;
}

int k0_eval(K0Buf K, u64 pos, hipStream_t stream) {
    hipLaunchKernelGGL(k0_eval_at, dim3(1), dim3(256), 0, stream, K, pos);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

int k0_phase_plan(K0Buf K, u32 cap, u64 t0, u64 own_len, u32 last, u64 total, hipStream_t stream) {
    u64 nb = total >= t0 ? (total - t0) / cap + 2 : 1;
    if (nb > (u64)K.maxBlocks + 1) nb = (u64)K.maxBlocks + 1;
    HIP_CHECK_RET(hipMemsetAsync(K.specBad, 0xFF, 8, stream));
    hipLaunchKernelGGL(k0_phase_spec, dim3((u32)nb), dim3(256), 0, stream, K, cap, t0);
    hipLaunchKernelGGL(k0_phase_chain, dim3(1), dim3(256), 0, stream, K, cap, t0, own_len, last, (u32)nb);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

// RLE1 text + CRC of blocks [first_block, first_block + P.g.nb) into the batch P
// side / ev_pad / ev_crc (optional): the block CRCs - one workgroup per block, 90-120 us alone on the sub-batch's critical path, and nobody needs them
// before k5_header - run on `side` behind k0_pad (ev_pad) while `stream` goes on with K1; ev_crc is recorded behind them (k5_run waits for it)
int k0_batch(K0Buf K, Pipe P, u32 first_block, u32 cap, hipStream_t stream, u32 crc_parts, hipStream_t side, hipEvent_t ev_pad, hipEvent_t ev_crc) {
    if (crc_parts < 1u) crc_parts = 1u;
    if (crc_parts > K0_CRC_PARTS) crc_parts = K0_CRC_PARTS;
    const u32 gx = cap / K0_TILE + 2;
    hipLaunchKernelGGL(k0_materialize, dim3(gx, P.g.nb), dim3(256), 0, stream, K, P, first_block, cap);
    hipLaunchKernelGGL(k0_pad, dim3(P.g.nb), dim3(64), 0, stream, K, P, first_block);
    if (side && ev_pad && ev_crc) {
        HIP_CHECK_RET(hipEventRecord(ev_pad, stream));
        HIP_CHECK_RET(hipStreamWaitEvent(side, ev_pad, 0));
        hipLaunchKernelGGL(k0_crc, dim3(crc_parts, P.g.nb), dim3(1024), 0, side, K, P, first_block);
        HIP_CHECK_RET(hipEventRecord(ev_crc, side));
    } else {
        hipLaunchKernelGGL(k0_crc, dim3(crc_parts, P.g.nb), dim3(1024), 0, stream, K, P, first_block);
    }
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
