// K1, special case: blocks with a small linear period (round 4).
//
// lib/BWT.js:197-300 (SA-IS) is linear on any input; prefix doubling is not: a block T[i] = P[i mod p] keeps p groups of n / p
// rotations alive for log2(n) rounds (a bzip2 block cuts a periodic input at a length that is no multiple of the period, so the
// rotations are not identical and every round peels ~2h of them off each group): `periodic ab` 50 ms per 5*10^7 bytes, the
// 44-byte period 35 ms, zeros (RLE1 makes them the 5-byte period 00 00 00 00 fb) 22 ms in rounds 1-3 and in round 4's list-driven
// rounds alike.  For p <= 64 the order of the rotations has a closed form; this file detects such blocks and writes their suffix
// arrays directly.  The blocks are then invisible to the rest of K1's sort (nfront[b] = 0): only k1_finish sees them again.
//
// With n = q p + r0, P primitive (p is the block's SMALLEST period), W_c = the infinite periodic string that starts at phase c:
//   rotation i = W_c for L = n - i characters (c = i mod p), then W_0 for i characters: periodic with ONE defect at offset L,
//   where phase r0 is replaced by phase 0.
//   * two rotations of different phase whose first segments are at least p long differ inside them (two different rotations of a
//     primitive word differ within p characters): they order as their phases do;
//   * two rotations i < j of the SAME phase agree for L_j characters; then j goes on with W_0 and i with W_r0, both for at least
//     p characters: ONE sign for the whole block - W_r0 < W_0 puts the smaller index first, else the larger (r0 = 0: the
//     rotations are identical and the larger index comes first: SURVEY.md 9.2);
//   so the "regular" rotations (L >= p) sort by (rank of their phase, then index, ascending or descending for all of them);
//   * the p - 1 rotations with L < p are placed by comparing them with regular ones segment by segment (each comparison is at
//     most a few table look-ups: first mismatch of two phases) - a binary search in every phase class, whose members a total
//     order keeps on one side of the newcomer.
// k1p_detect (smallest period <= 64 of a block, from a 2 KB prefix first: ordinary text fails there after a few dozen compares),
// k1p_tables (phase order, class bases, insertion points of the irregular rotations), k1p_fill (the suffix array, head bits).
#include "k1_bwt.h"
#include "devutil.h"

#define K1P_PREFIX 2048u
#define K1P_MINN 4096u           // shorter blocks take the general path
#define K1P_FLAGS 0
#define K1P_NREG 1
#define K1P_START 2
#define K1P_PHASE 67
#define K1P_THR 131
#define K1P_NIRR 195
#define K1P_FAIL 196

// first mismatch of W_a and W_b (a != b): offset | (W_a < W_b) << 7, in LDS
__device__ __forceinline__ u32 k1p_seg_cmp(const u8* dt, u32 p, u32 n, u32 i, u32 j) {
    // -1 / 0 / +1 as rotation i is smaller than / identical to / larger than rotation j, by walking their segments:
    // (phase, characters left in the segment); a rotation has the segments (i mod p, n - i) and (0, i)
    u32 pa = i % p, la = n - i, sa = 0, pb = j % p, lb = n - j, sb = 0;
    u32 done = 0;
    for (int step = 0; step < 8 && done < n; step++) {
        const u32 m = la < lb ? la : lb;
        if (pa != pb) {
            const u32 e = dt[pa * 64u + pb];
            const u32 d = e & 0x7Fu;
            if (d < m) return (e & 0x80u) ? 0xFFFFFFFFu : 1u;
        }
        done += m;
        pa = (pa + m) % p; pb = (pb + m) % p;
        la -= m; lb -= m;
        if (la == 0u) { if (sa) break; sa = 1; pa = 0; la = i; }
        if (lb == 0u) { if (sb) break; sb = 1; pb = 0; lb = j; }
        if (la == 0u || lb == 0u) break;                 // (rotation 0 has no second segment)
    }
    return 0u;
}

// per[b] = the smallest period <= 64 of the block's first K1P_PREFIX bytes (0: none, block too short, linear mode, switched off);
// nfront[b] = what the general sort sees of the block
__global__ __launch_bounds__(256) void k1p_detect(K1Buf B, BatchGeom g, u32 enable) {
    const u32 b = blockIdx.x, n = B.nlen[b], tid = threadIdx.x;
    const u8* T = B.T + (size_t)b * g.tstride;
    __shared__ u32 pre[K1P_PREFIX / 4u];
    __shared__ u32 bad[64];
    u32 period = 0;
    if (enable && n >= K1P_MINN && !B.linear) {              // (uniform)
        const u32* T4 = (const u32*)T;
        for (u32 i = tid; i < K1P_PREFIX / 4u; i += 256u) pre[i] = T4[i];
        if (tid < 64) bad[tid] = 0;
        __syncthreads();
        // thread t: candidate period (t & 63) + 1 on a quarter of the prefix
        const u8* pb = (const u8*)pre;
        const u32 pc = (tid & 63u) + 1u, part = tid >> 6;
        const u32 per4 = (K1P_PREFIX - 64u) / 4u;
        u32 f = 0;
        for (u32 i = part * per4; i < (part + 1u) * per4 && !f; i++) f = pb[i] != pb[i + pc] ? 1u : 0u;
        if (f) bad[pc - 1u] = 1u;
        __syncthreads();
        for (u32 k = 0; k < 64u; k++) if (!bad[k]) { period = k + 1u; break; }
    }
    if (tid == 0) {
        B.per[b] = period;
        B.nfront[b] = period ? 0u : n;
        B.ptab[(size_t)b * 256u + K1P_FAIL] = 0u;
    }
}

// the candidate against the whole block (a block with a period p' <= 64 has it on its prefix too, where the smallest period
// divides it - and then holds for the whole block: one candidate decides)
__global__ __launch_bounds__(256) void k1p_verify(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y, tid = threadIdx.x;
    const u32 p = B.per[b];
    if (!p) return;
    const u32 n = B.nlen[b];
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32 t0 = blockIdx.x * 4096u;
    u32 f = 0;
#pragma unroll
    for (u32 k = 0; k < 16u; k++) {
        const u32 i = t0 + k * 256u + tid;
        if (i + p < n) f |= (u32)(T[i] ^ T[i + p]);
    }
    if (f) B.ptab[(size_t)b * 256u + K1P_FAIL] = 1u;
}

// tables of a periodic block in ptab[b][256]:
//   [0]        flags: bit 0 = regular rotations of a phase in ascending index order
//   [1]        regular rotations (all n when r0 = 0, else n - p + 1)
//   [2..66]    start[k]: regular rank of the first rotation of the phase with rank k (k = 0..p; start[p] = their number)
//   [67..130]  phase[k]: the phase with rank k
//   [131..194] thr[u]: the irregular rotations in ascending order: regular rotations before the u-th of them
//   [195]      irregular rotations (0 or p - 1)
//   [196]      k1p_verify: the candidate period does not hold
__global__ __launch_bounds__(256) void k1p_tables(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.x, tid = threadIdx.x;
    const u32 p = B.per[b];
    if (!p) return;
    const u32 n = B.nlen[b];
    const u8* T = B.T + (size_t)b * g.tstride;
    u32* tab = B.ptab + (size_t)b * 256u;
    if (tab[K1P_FAIL]) {                                  // (uniform) not periodic after all: the general sort takes the block
        if (tid == 0) { B.per[b] = 0u; B.nfront[b] = n; }
        return;
    }
    u32* SA = B.SA + (size_t)b * g.stride;
    __shared__ u8 P[128];
    __shared__ u8 dt[64 * 64];
    __shared__ u32 ph[64], start[65], cnt[64];
    __shared__ u32 less[64][64];                          // [irregular][phase]: regular rotations of the phase below the irregular one
    __shared__ u32 ereg[64], eord[64];
    for (u32 i = tid; i < 2u * p; i += 256u) P[i] = T[i % p];
    __syncthreads();
    for (u32 e = tid; e < p * p; e += 256u) {
        const u32 a = e / p, c = e % p;
        u32 d = 0, lt = 0;
        if (a != c) {
            while (d < p && P[(a + d) % p] == P[(c + d) % p]) d++;
            lt = d < p && P[(a + d) % p] < P[(c + d) % p] ? 1u : 0u;
        }
        dt[a * 64u + c] = (u8)(d | (lt << 7));
    }
    __syncthreads();
    const u32 r0 = n % p;
    if (tid < p) {
        u32 r = 0;
        for (u32 a = 0; a < p; a++) r += (a != tid && (dt[a * 64u + tid] & 0x80u)) ? 1u : 0u;       // phases below mine
        ph[r] = tid;
        // regular rotations of phase tid: tid, tid + p, ... with n - i >= p (all of them when r0 = 0: the defect is invisible)
        cnt[tid] = r0 ? (n - p - tid) / p + 1u : n / p;
    }
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (u32 k = 0; k < p; k++) { start[k] = run; run += cnt[ph[k]]; }
        start[p] = run;
    }
    // one sign for the order inside a phase class: W_r0 < W_0 -> ascending index; r0 = 0: identical rotations, descending
    const bool asc = r0 != 0u && (dt[r0 * 64u + 0u] & 0x80u) != 0u;
    const u32 nirr = r0 ? p - 1u : 0u;
    __syncthreads();
    // irregular rotation u (start index n - p + 1 + u) against every phase class: how many of the class sort below it
    for (u32 e = tid; e < nirr * p; e += 256u) {
        const u32 u = e / p, c = e % p, j = n - p + 1u + u;
        const u32 mc = cnt[c];
        u32 lo = 0, hi = mc;                              // members [0, lo) of the class (in its order) are below j, [hi, mc) above
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            const u32 i = c + (asc ? mid : mc - 1u - mid) * p;
            if (k1p_seg_cmp(dt, p, n, i, j) == 0xFFFFFFFFu) lo = mid + 1u; else hi = mid;
        }
        less[u][c] = lo;
    }
    __syncthreads();
    if (tid < nirr) {
        u32 e = 0;
        for (u32 c = 0; c < p; c++) e += less[tid][c];
        ereg[tid] = e;
    }
    __syncthreads();
    if (tid < nirr) {
        // order among the irregular ones: by their place among the regular rotations, ties by comparing them with each other
        const u32 j = n - p + 1u + tid;
        u32 o = 0;
        for (u32 v = 0; v < nirr; v++) {
            if (v == tid) continue;
            const u32 jv = n - p + 1u + v;
            const bool below = ereg[v] < ereg[tid] || (ereg[v] == ereg[tid] && k1p_seg_cmp(dt, p, n, jv, j) == 0xFFFFFFFFu);
            o += below ? 1u : 0u;
        }
        eord[tid] = o;
    }
    __syncthreads();
    if (tid < nirr) {
        tab[K1P_THR + eord[tid]] = ereg[tid];
        SA[ereg[tid] + eord[tid]] = n - p + 1u + tid;     // its own place: regular rotations below it + irregular ones below it
    }
    if (tid <= p) tab[K1P_START + tid] = start[tid];
    if (tid < p) tab[K1P_PHASE + tid] = ph[tid];
    if (tid == 0) { tab[K1P_FLAGS] = asc ? 1u : 0u; tab[K1P_NREG] = start[p]; tab[K1P_NIRR] = nirr; }
}

// the regular rotations into their places, the head bitmap all heads
__global__ __launch_bounds__(256) void k1p_fill(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y, tid = threadIdx.x;
    const u32 p = B.per[b];
    if (!p) return;
    const u32 n = B.nlen[b];
    const u32* tab = B.ptab + (size_t)b * 256u;
    __shared__ u32 st[65], ph[64], thr[64];
    if (tid <= p) st[tid] = tab[K1P_START + tid];
    if (tid < p) ph[tid] = tab[K1P_PHASE + tid];
    const u32 nirr = tab[K1P_NIRR], nreg = tab[K1P_NREG];
    const bool asc = tab[K1P_FLAGS] & 1u;
    if (tid < nirr) thr[tid] = tab[K1P_THR + tid];
    __syncthreads();
    u32* SA = B.SA + (size_t)b * g.stride;
    u32* HN = B.HN + (size_t)b * g.hstride;
    const u32 t0 = blockIdx.x * 4096u;
    for (u32 t = t0 + tid; t < t0 + 4096u && t < nreg; t += 256u) {
        u32 lo = 0, hi = p;                               // the phase class: the last k with st[k] <= t
        while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (st[mid] <= t) lo = mid; else hi = mid; }
        const u32 c = ph[lo], idx = t - st[lo], mc = st[lo + 1u] - st[lo];
        const u32 i = c + (asc ? idx : mc - 1u - idx) * p;
        u32 a = 0, e = nirr;                              // irregular rotations in front of it: those with thr <= t
        while (a < e) { const u32 mid = (a + e) >> 1; if (thr[mid] <= t) a = mid + 1u; else e = mid; }
        SA[t + a] = i;
    }
    for (u32 w = t0 / 32u + tid; w < (t0 + 4096u) / 32u && w * 32u < n; w += 256u) HN[w] = 0xFFFFFFFFu;       // (bits at and beyond n are set already)
}

int k1_period_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream, u32 enable) {
    hipLaunchKernelGGL(k1p_detect, dim3(g.nb), dim3(256), 0, stream, B, g, enable);
    hipLaunchKernelGGL(k1p_verify, dim3((max_n + 4095u) / 4096u, g.nb), dim3(256), 0, stream, B, g);
    hipLaunchKernelGGL(k1p_tables, dim3(g.nb), dim3(256), 0, stream, B, g);
    hipLaunchKernelGGL(k1p_fill, dim3((max_n + 4095u) / 4096u, g.nb), dim3(256), 0, stream, B, g);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
