// K6: inverse of the linear BWT, BWT.unbwtransform (lib/BWT.js:352-363), for gfx950.
//
// The reference walks the LF chain serially:  t = 0; for i = n-1 .. 0: U[i] = T[t];
// t = LF[t] + C[T[t]]; if (t < pidx) t++.   Here:
//   k6_hist / k6_scan   per-tile byte counts -> C[] and the tile offsets (a stable counting sort)
//   k6_next             next(t) = LF[t] + C[T[t]] (+1 below pidx) with wave-ballot ranks, and the
//                       inverse links pred[next[t]] = t (injective except for the last position of
//                       the chain, whose successor is the sentinel row and is skipped)
//   k6_jump x log2(n)   Wyllie list ranking towards node 0 over pred: R[t] = index of t in the chain
//   k6_emit             U[n-1-R[t]] = T[t]
// One block per launch set here (the BWTC decoder that would batch it is a "next" row).
#include "pipeline.h"

#define K6_TILE 4096

__global__ __launch_bounds__(256) void k6_hist(const u8* T, u32 n, u32* tileHist) {
    __shared__ u32 h[256];
    const u32 tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const u32 t0 = blockIdx.x * K6_TILE;
    for (int k = 0; k < 16; k++) {
        const u32 i = t0 + k * 256u + tid;
        if (i < n) atomicAdd(&h[T[i]], 1u);
    }
    __syncthreads();
    tileHist[(size_t)blockIdx.x * 256 + tid] = h[tid];
}

// digit-major exclusive offsets: tileHist[t][c] <- C[c] + sum_{t' < t} count[t'][c]
__global__ __launch_bounds__(256) void k6_scan(u32* tileHist, u32 ntiles) {
    __shared__ u32 sh[256];
    const u32 c = threadIdx.x;
    u32 tot = 0;
    for (u32 t = 0; t < ntiles; t++) tot += tileHist[(size_t)t * 256 + c];
    const u32 base = block_excl_scan_256(tot, sh);
    u32 run = base;
    for (u32 t = 0; t < ntiles; t++) {
        const u32 v = tileHist[(size_t)t * 256 + c];
        tileHist[(size_t)t * 256 + c] = run;
        run += v;
    }
}

// RAW = false: inverse links pred[next(t)] = t for the list ranking.  RAW = true: the successor map itself,
// nxt[t] = LF[t] + C[T[t]] (+1 below pidx), exactly what the reference's loop computes (may be n): only for the
// serial fallback below.
template <bool RAW>
__global__ __launch_bounds__(256) void k6_next(const u8* T, u32 n, u32 pidx, const u32* tileHist, u32* pred) {
    __shared__ u32 wh[4][256];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    for (u32 i = tid; i < 1024; i += 256) (&wh[0][0])[i] = 0;
    __syncthreads();
    const u32 t0 = blockIdx.x * K6_TILE;
    const u64 lt = lanemask_lt();
    u32 rk[16], cv[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {                      // stable rank of every byte inside its wave's range
        const u32 i = t0 + w * 1024u + it * 64u + lane;
        const bool valid = i < n;
        const u32 c = valid ? T[i] : 0u;
        const u64 m = match_any(c, 8, valid);
        const u32 rank = (u32)__popcll(m & lt), cnt = (u32)__popcll(m);
        const u32 prior = valid ? wh[w][c] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wh[w][c] = prior + cnt;
        __builtin_amdgcn_wave_barrier();
        rk[it] = prior + rank;
        cv[it] = c;
    }
    __syncthreads();
    {
        u32 o = tileHist[(size_t)blockIdx.x * 256 + tid];
        for (int ww = 0; ww < 4; ww++) { const u32 c = wh[ww][tid]; wh[ww][tid] = o; o += c; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const u32 i = t0 + w * 1024u + it * 64u + lane;
        if (i < n) {
            u32 t = wh[w][cv[it]] + rk[it];                // LF[i] + C[T[i]] = (full row of the predecessor) - 1
            // Row pidx of the full (n+1)-row matrix holds the sentinel: the position that maps onto
            // it is the END of the chain (its successor is never used by the reference loop), every
            // other successor is unique.
            const bool last = t + 1u == pidx;
            if (t < pidx) t++;
            if (RAW) pred[i] = t;
            else if (!last && t < n) pred[t] = i;
        }
    }
}

// A position nothing links to (pred still ~0: only possible for a (T, pidx) pair no BWT produces) becomes a
// self-loop of rank n; ranks saturate at n, so exactly the positions on the chain that starts at node 0 end up
// with R < n, all distinct.
__global__ __launch_bounds__(256) void k6_init(u32 n, const u32* pred, u32* P, u32* R) {
    const u32 t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const u32 p = pred[t];
    P[t] = t == 0 ? 0u : (p == 0xFFFFFFFFu ? t : p);
    R[t] = t == 0 ? 0u : (p == 0xFFFFFFFFu ? n : 1u);
}
__global__ __launch_bounds__(256) void k6_jump(u32 n, const u32* Pin, const u32* Rin, u32* Pout, u32* Rout) {
    const u32 t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const u32 p = Pin[t];
    const u32 r = Rin[t] + Rin[p];
    Rout[t] = r < n ? r : n;
    Pout[t] = Pin[p];
}
__global__ __launch_bounds__(256) void k6_emit(const u8* T, u32 n, const u32* R, u8* U, u32* onChain) {
    const u32 t = blockIdx.x * 256u + threadIdx.x;
    const u32 r = t < n ? R[t] : n;
    if (r < n) U[n - 1u - r] = T[t];
    const u64 bal = __ballot(r < n);
    if ((threadIdx.x & 63u) == 0 && bal) atomicAdd(onChain, (u32)__popcll(bal));
}
// The LF mapping of a real BWT is one cycle through all n positions.  For any other (T, pidx) the reference's
// loop (lib/BWT.js:359-362) still takes n steps from t = 0: this kernel repeats it literally (one lane).  When t
// leaves [0, n) the reference's arithmetic turns to NaN and every remaining U[i] = T[NaN] stores 0.
__global__ void k6_serial(const u8* T, u32 n, const u32* nxt, u8* U) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 t = 0;
    bool dead = false;
    for (u32 i = n; i-- > 0;) {
        if (dead) { U[i] = 0; continue; }
        U[i] = T[t];
        t = nxt[t];
        if (t >= n) dead = true;
    }
}

// device pointers; ws needs 5*n*4 + ntiles*1024 + 256 bytes
int k6_unbwt_linear(const u8* dT, u8* dU, u32 n, u32 pidx, void* ws, hipStream_t stream) {
    const u32 ntiles = (n + K6_TILE - 1) / K6_TILE, nb = (n + 255) / 256;
    u32* pred = (u32*)ws;
    u32* P0 = pred + n; u32* R0 = P0 + n; u32* P1 = R0 + n; u32* R1 = P1 + n;
    u32* tileHist = R1 + n;
    u32* onChain = tileHist + (size_t)ntiles * 256;
    HIP_CHECK_RET(hipMemsetAsync(pred, 0xFF, (size_t)n * 4, stream));
    HIP_CHECK_RET(hipMemsetAsync(onChain, 0, 4, stream));
    hipLaunchKernelGGL(k6_hist, dim3(ntiles), dim3(256), 0, stream, dT, n, tileHist);
    hipLaunchKernelGGL(k6_scan, dim3(1), dim3(256), 0, stream, tileHist, ntiles);
    hipLaunchKernelGGL(k6_next<false>, dim3(ntiles), dim3(256), 0, stream, dT, n, pidx, (const u32*)tileHist, pred);
    hipLaunchKernelGGL(k6_init, dim3(nb), dim3(256), 0, stream, n, (const u32*)pred, P0, R0);
    u32 *Pi = P0, *Ri = R0, *Po = P1, *Ro = R1;
    for (u32 span = 1; span < n; span <<= 1) {
        hipLaunchKernelGGL(k6_jump, dim3(nb), dim3(256), 0, stream, n, (const u32*)Pi, (const u32*)Ri, Po, Ro);
        u32* t = Pi; Pi = Po; Po = t; t = Ri; Ri = Ro; Ro = t;
    }
    hipLaunchKernelGGL(k6_emit, dim3(nb), dim3(256), 0, stream, dT, n, (const u32*)Ri, dU, onChain);
    u32 got = 0;
    HIP_CHECK_RET(hipMemcpyAsync(&got, onChain, 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK_RET(hipStreamSynchronize(stream));
    if (got != n) {
        // not one n-cycle (corrupt BWTC stream, or a caller's arbitrary (T, pidx)): the reference's walk, literally
        hipLaunchKernelGGL(k6_next<true>, dim3(ntiles), dim3(256), 0, stream, dT, n, pidx, (const u32*)tileHist, pred);
        hipLaunchKernelGGL(k6_serial, dim3(1), dim3(64), 0, stream, dT, n, (const u32*)pred, dU);
    }
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
