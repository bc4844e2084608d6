// K1, special case: blocks with a linear period (round 4).
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
// Longer periods (64 < p <= n / 4: tiled inputs) have no table-sized closed form, but the block need not be sorted whole: with
// n' = 3 p + (n mod p) and T' = T[0, n'), rotation i' of T' is a prefix of rotation i' + (n - n') of T and the rotations of T'
// are distinct, so the general sort of T' orders the LAST n' rotations of T; the missing ones (the first k = (n - n') / p of every
// phase c, all regular) sit next to the first member f = c + k p of their phase that T' has - before it in ascending index order
// or after it in descending order, the sign of the phase-internal order above: f and the missing members share at least
// n' - c >= 2 p + 1 characters, and a rotation of another phase that shared them would either be regular (and differ within p)
// or start in the last period with a first segment below p and then agree with W_(c + L) for p characters: the pure periodic
// rotation W_c[0, n), which lies on ONE side of all of them.  (n' = 2 p + (n mod p) is not enough: brute force over small cases
// finds rotations in between.)  n mod p = 0: T' = P, every rotation of P stands for n / p identical ones, larger index first.
// k1p_find / k1p_verify find the period, k1p_reduce hands T' to the sort (nfront[b] = n', the wrap-around bytes behind it, head
// bits), k1p_expand_* write the block's suffix array into SB afterwards.  sample3.ref (header + 30 000 x "ugh\n", 120 244 bytes) tiled:
// blocks of 900 k sort as blocks of 419 k (31 -> 13.8 ms per 5*10^7 bytes).
//
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
#define K1P_FAIL 196             // [2]: first position where the candidate of stage 0 / 1 does not hold (0xFFFFFFFF: it holds)
#define K1P_CAND 200             // [2]: candidates for a period beyond 64 (stage 0 / 1)
#define K1P_RASC 202             // reduced block: phase-internal order ascending
#define K1P_RK 203               // reduced block: periods left out
#define K1P_RN 204               // reduced block: its length n'
#define K1P_RED_MINN 16384u      // shorter blocks are not reduced
#define K1P_GRID 16u             // workgroups per block of the tile kernels (each walks its tiles: these kernels run for every batch and mostly
                                 // find nothing to do; a grid sized for the work is thousands of idle workgroups queueing behind the other stream)

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
        u32* tab = B.ptab + (size_t)b * 256u;
        tab[K1P_FAIL] = tab[K1P_FAIL + 1u] = 0xFFFFFFFFu;    // k1p_verify: the first position where the candidate does not hold
        tab[K1P_CAND] = tab[K1P_CAND + 1u] = 0xFFFFFFFFu;    // k1p_find: the smallest candidate
        B.red[b] = 0u;
    }
}

// Candidates for a longer period: positions 64 < j <= n / 4 where 32 bytes of the block come again, and 32 bytes elsewhere come
// again at the same distance; the smallest one is the candidate.  Every period passes, so a candidate that then holds for the
// whole block (k1p_verify) is the block's smallest period.
//   stage 0: the block's first 32 bytes and the 32 at n / 2.
//   stage 1: for the blocks whose first candidate (k1p_detect's or stage 0's) failed - a file of equal lines, tiled: the line
//     length matches both probes and breaks where the file ends.  k1p_verify left the first position where it broke: the 32 bytes around that place hold the
//     irregularity, and their next occurrence is one true period on (second probe: the block's first 32 bytes).
__global__ __launch_bounds__(256) void k1p_find(K1Buf B, BatchGeom g, u32 enable, u32 stage) {
    const u32 b = blockIdx.y, n = B.nlen[b], tid = threadIdx.x;
    if (!enable || B.linear || n < K1P_RED_MINN || (stage == 0u && B.per[b])) return;          // (a short candidate goes first)
    u32* tab = B.ptab + (size_t)b * 256u;
    u32 x1 = 0, x2 = n / 2u, jmax = n / 4u;
    if (stage) {
        const u32 c0 = B.per[b] ? B.per[b] : tab[K1P_CAND], m0 = tab[K1P_FAIL];                 // (k1p_detect's candidate: the lines may be short)
        if (c0 == 0xFFFFFFFFu || m0 == 0xFFFFFFFFu) return;   // no first candidate, or it holds
        x1 = m0 + c0 >= 16u ? m0 + c0 - 16u : 0u;             // T[m0] != T[m0 + c0]: around the later one
        x2 = 0u;
        if (x1 + 32u + 65u >= n) return;
        if (jmax > n - x1 - 32u) jmax = n - x1 - 32u;
    }
    const u8* T = B.T + (size_t)b * g.tstride;
    __shared__ u32 pat[8];
    __shared__ u32 tw[4096 / 4 + 12];
    if (tid < 8) { const u8* q = T + x1 + tid * 4u; pat[tid] = (u32)q[0] | (u32)q[1] << 8 | (u32)q[2] << 16 | (u32)q[3] << 24; }
    // (few workgroups, each walking its tiles: these kernels run for every batch and mostly find nothing to do - a grid sized for the
    // work would be thousands of idle workgroups that queue behind the other stream's kernels)
    for (u32 t0 = 65u + blockIdx.x * 4096u; t0 <= jmax + 3u; t0 += gridDim.x * 4096u) {      // (a tile starts up to 3 distances early, see below)
    __syncthreads();
    // The tile as ALIGNED words (byte loads of 4 KB per workgroup were most of the kernel's 30 us per 10^8 bytes of text): it starts at
    // the word that holds distance t0, so a thread's 16 consecutive distances begin on a word and slide a 64-bit window over five
    // of them with shifts known at compile time; the up to 3 distances in front of t0 belong to the tile before (checked twice).
    const u32 al = (x1 + t0) & ~3u;                          // (T + b * tstride is 128-byte aligned)
    const u32* T4 = (const u32*)(T + al);
    for (u32 i = tid; i < 4096u / 4u + 12u; i += 256u) tw[i] = al + i * 4u < n + 64u ? T4[i] : 0u;     // (k0_pad's 64 bytes behind n are readable)
    __syncthreads();
    const u8* tb = (const u8*)tw;
    const u32 p0 = pat[0];
    u32 w[5];
#pragma unroll
    for (u32 q = 0; q < 5u; q++) w[q] = tw[tid * 4u + q];
    u32 hit = 0;
#pragma unroll
    for (u32 k = 0; k < 16u; k++) {
        const u32 q = k >> 2, sft = (k & 3u) * 8u;
        const u32 v = sft ? (u32)((((u64)w[q + 1u] << 32) | w[q]) >> sft) : w[q];
        hit |= (v == p0 ? 1u : 0u) << k;
    }
    while (hit) {
        const u32 k = (u32)__builtin_ctz(hit);
        hit &= hit - 1u;
        const u32 o = tid * 16u + k, j = al + o - x1;
        if (j < 65u || j > jmax) continue;
        bool m = true;
        for (u32 ww = 1; ww < 8u && m; ww++) {
            const u8* q = tb + o + ww * 4u;
            m = ((u32)q[0] | (u32)q[1] << 8 | (u32)q[2] << 16 | (u32)q[3] << 24) == pat[ww];
        }
        for (u32 i = 0; i < 32u && m; i++) m = T[x2 + i] == T[x2 + j + i];
        if (m) atomicMin(&tab[K1P_CAND + stage], j);
    }
    }
}

// the candidate against the whole block; stage 0 also checks the short candidate of k1p_detect (a block with a period p' <= 64
// has it on its prefix too, where the smallest period divides it - and then holds for the whole block: one candidate decides)
__global__ __launch_bounds__(256) void k1p_verify(K1Buf B, BatchGeom g, u32 stage) {
    const u32 b = blockIdx.y, tid = threadIdx.x;
    u32* tab = B.ptab + (size_t)b * 256u;
    const u32 p = stage == 0u && B.per[b] ? B.per[b] : tab[K1P_CAND + stage];          // (k1p_find skips the blocks with a short candidate)
    if (p == 0xFFFFFFFFu) return;
    const u32 n = B.nlen[b];
    const u8* T = B.T + (size_t)b * g.tstride;
    u32 first = 0xFFFFFFFFu;
    for (u32 t0 = blockIdx.x * 4096u; t0 + p < n && first == 0xFFFFFFFFu; t0 += gridDim.x * 4096u)     // (tiles in ascending order: the first hit of a thread is its smallest)
        for (u32 k = 0; k < 16u; k++) {
            const u32 i = t0 + (15u - k) * 256u + tid;         // (descending inside a tile: the last hit is the smallest)
            if (i + p < n && T[i] != T[i + p]) first = i;
        }
    const u64 bal = __ballot(first != 0xFFFFFFFFu);
    if (!bal) return;
    for (u32 off = 32; off; off >>= 1) { const u32 o = __shfl_xor(first, off); first = o < first ? o : first; }
    if ((tid & 63u) == 0u) atomicMin(&tab[K1P_FAIL + stage], first);
}

// tables of a periodic block in ptab[b][256]:
//   [0]        flags: bit 0 = regular rotations of a phase in ascending index order
//   [1]        regular rotations (all n when r0 = 0, else n - p + 1)
//   [2..66]    start[k]: regular rank of the first rotation of the phase with rank k (k = 0..p; start[p] = their number)
//   [67..130]  phase[k]: the phase with rank k
//   [131..194] thr[u]: the irregular rotations in ascending order: regular rotations before the u-th of them
//   [195]      irregular rotations (0 or p - 1)
//   [196]      k1p_verify: the first position where the candidate does not hold (0xFFFFFFFF: it holds)
__global__ __launch_bounds__(256) void k1p_tables(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.x, tid = threadIdx.x;
    const u32 p = B.per[b];
    if (!p) return;
    const u32 n = B.nlen[b];
    const u8* T = B.T + (size_t)b * g.tstride;
    u32* tab = B.ptab + (size_t)b * 256u;
    if (tab[K1P_FAIL] != 0xFFFFFFFFu) {                   // (uniform) not periodic after all: the general sort takes the block
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
    for (u32 t0 = blockIdx.x * 4096u; t0 < n; t0 += gridDim.x * 4096u) {
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
}

// a verified longer period: the reduced block goes to the sort
__global__ __launch_bounds__(256) void k1p_reduce(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.x, tid = threadIdx.x;
    u32* tab = B.ptab + (size_t)b * 256u;
    const u32 n = B.nlen[b];
    u32 p = 0;                                               // the verified candidate
    if (!B.per[b]) {
        if (tab[K1P_CAND] != 0xFFFFFFFFu && tab[K1P_FAIL] == 0xFFFFFFFFu) p = tab[K1P_CAND];
        else if (tab[K1P_CAND + 1u] != 0xFFFFFFFFu && tab[K1P_FAIL + 1u] == 0xFFFFFFFFu) p = tab[K1P_CAND + 1u];
    }
    const u32 r0 = p ? n % p : 0u;
    const u32 nr = r0 ? 3u * p + r0 : p;
    // (uniform) no period found, or too little to gain
    if (!p || (u64)nr * 5u > (u64)n * 4u) return;            // (red[b] = 0 since k1p_detect)
    if (tid == 0) { B.red[b] = p; B.dred[K1_BI(B, b)] = 1u; }
    u8* T = (u8*)B.T + (size_t)b * g.tstride;
    __shared__ u32 first;
    if (tid == 0) first = 0xFFFFFFFFu;
    __syncthreads();
    // the sign of the order inside a phase: W_r0 against W_0 (they differ within p characters)
    if (r0) {
        for (u32 c0 = 0; c0 < p; c0 += 256u) {
            const u32 t = c0 + tid;
            if (t < p && T[r0 + t] != T[t]) atomicMin(&first, t);
            __syncthreads();
            const u32 seen = first;                          // (every thread reads it between the two barriers: one decision)
            __syncthreads();
            if (seen != 0xFFFFFFFFu) break;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const u32 d = first;
        tab[K1P_RASC] = r0 && d != 0xFFFFFFFFu && T[r0 + d] < T[d] ? 1u : 0u;
        tab[K1P_RK] = (n - nr) / p;
        tab[K1P_RN] = nr;
        B.nfront[b] = nr;
    }
    __syncthreads();
    // what the sort reads behind the block's end (k0_pad's 64 bytes): T'[nr + t] = T'[t]; k1p_expand_scan puts the block's own bytes back
    if (r0 && tid < 64u) T[nr + tid] = T[tid];
    // head bits at and beyond nr
    u32* HN = B.HN + (size_t)b * g.hstride;
    for (u32 w = nr / 32u + tid; w * 32u < n; w += 256u) HN[w] = w == nr / 32u ? (HN[w] | (0xFFFFFFFFu << (nr & 31u))) : 0xFFFFFFFFu;
}

// ---- after the sort: SB <- the suffix array of the whole block ----------------------------------------------
// (a block's slice of tileHist: ceil(stride / K1F_PT) x K1F_NB words, far more than its stride / 4096 tile counts)
__device__ __forceinline__ size_t k1p_th(const BatchGeom& g, u32 b) { return (size_t)b * ((g.stride + K1F_PT - 1u) / K1F_PT) * K1F_NB; }
// entry t of the reduced block's suffix array (rotation e of T') becomes rotation e + k p of T at position t + k C(t), C(t) = the
// entries before t with e < p (the first members of their phases: each brings the k missing ones along)
__global__ __launch_bounds__(256) void k1p_expand_count(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y, tid = threadIdx.x;
    const u32 p = B.red[b];
    if (!p) return;
    const u32 nr = B.ptab[(size_t)b * 256u + K1P_RN];
    const u32* SA = B.SA + (size_t)b * g.stride;
    __shared__ u32 tot;
    for (u32 tile = blockIdx.x; tile * 4096u < nr; tile += gridDim.x) {       // (uniform)
        const u32 t0 = tile * 4096u;
        u32 c = 0;
        for (u32 k = 0; k < 16u; k++) { const u32 t = t0 + k * 256u + tid; if (t < nr && SA[t] < p) c++; }
        if (tid == 0) tot = 0;
        __syncthreads();
        for (u32 off = 32; off; off >>= 1) c += __shfl_xor(c, off);
        if ((tid & 63u) == 0) atomicAdd(&tot, c);
        __syncthreads();
        if (tid == 0) B.tileHist[k1p_th(g, b) + tile] = tot;                  // (the front end's counts are long used)
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k1p_expand_scan(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.x, tid = threadIdx.x;
    const u32 p = B.red[b];
    if (!p) return;
    const u32 n = B.nlen[b];
    const u32 nr = B.ptab[(size_t)b * 256u + K1P_RN];
    const u32 nt = (nr + 4095u) / 4096u;                     // <= 176 for a bzip2 block, <= 820 for the 2^22 bytes cjs_bwt_cyclic_batch takes
    u32* th = B.tileHist + k1p_th(g, b);
    __shared__ u32 sc[256];
    u32 carry = 0;
    for (u32 c0 = 0; c0 < nt; c0 += 256u) {                  // (uniform)
        const u32 i = c0 + tid;
        const u32 v = i < nt ? th[i] : 0u;
        sc[tid] = v;
        __syncthreads();
        for (u32 off = 1; off < 256u; off <<= 1) {
            const u32 a = tid >= off ? sc[tid - off] : 0u;
            __syncthreads();
            sc[tid] += a;
            __syncthreads();
        }
        if (i < nt) th[i] = carry + sc[tid] - v;
        carry += sc[255];
        __syncthreads();
    }
    // the block's own bytes behind the reduced block again (k1_finish gathers from the whole block)
    const u32 r0 = n % p;
    u8* T = (u8*)B.T + (size_t)b * g.tstride;
    if (r0 && tid < 64u) T[nr + tid] = T[r0 + tid];
}
__global__ __launch_bounds__(256) void k1p_expand_write(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y, tid = threadIdx.x;
    const u32 p = B.red[b];
    if (!p) return;
    const u32* tab = B.ptab + (size_t)b * 256u;
    const u32 nr = tab[K1P_RN], k = tab[K1P_RK];
    const bool asc = tab[K1P_RASC] != 0u;
    const u32* SA = B.SA + (size_t)b * g.stride;
    u32* out = B.SB + (size_t)b * g.stride;
    __shared__ u32 wsum[4];
    const u32 lane = tid & 63u, w = tid >> 6;
    for (u32 tile = blockIdx.x; tile * 4096u < nr; tile += gridDim.x) {       // (uniform)
        const u32 t0 = tile * 4096u;
        u32 run = B.tileHist[k1p_th(g, b) + tile];
        for (u32 r = 0; r < 16u; r++) {                      // rows of 256 consecutive entries
            const u32 t = t0 + r * 256u + tid;
            const u32 e = t < nr ? SA[t] : 0xFFFFFFFFu;
            const bool fl = e < p;
            const u64 bal = __ballot(fl);
            if (lane == 0) wsum[w] = (u32)__popcll(bal);
            __syncthreads();
            u32 before = (u32)__popcll(bal & ((1ull << lane) - 1ull));
            for (u32 q = 0; q < w; q++) before += wsum[q];
            const u32 rowtot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            __syncthreads();
            if (t < nr) {
                const u32 base = t + k * (run + before), i = e + k * p;
                if (!fl) out[base] = i;
                else if (asc) {
                    for (u32 j = 0; j < k; j++) out[base + j] = e + j * p;
                    out[base + k] = i;
                } else {
                    out[base] = i;
                    for (u32 j = 0; j < k; j++) out[base + 1u + j] = e + (k - 1u - j) * p;
                }
            }
            run += rowtot;
        }
    }
}

int k1_period_expand(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream) {
    if (max_n < K1P_RED_MINN) return CJS_OK;                                  // (no block of the batch can have been reduced)
    hipLaunchKernelGGL(k1p_expand_count, dim3(K1P_GRID, g.nb), dim3(256), 0, stream, B, g);
    hipLaunchKernelGGL(k1p_expand_scan, dim3(g.nb), dim3(256), 0, stream, B, g);
    hipLaunchKernelGGL(k1p_expand_write, dim3(K1P_GRID, g.nb), dim3(256), 0, stream, B, g);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

int k1_period_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream, u32 enable) {
    if (max_n == 0u) max_n = 1u;                                              // (a batch of empty blocks: the grids below must not be empty)
    hipLaunchKernelGGL(k1p_detect, dim3(g.nb), dim3(256), 0, stream, B, g, enable);
    for (u32 stage = 0; stage < 2u; stage++) {
        // (round 6: 32 workgroups per block instead of 8 - a block's n / 4 candidate distances are 55 tiles, and with 7 of them per workgroup, one after
        // the other, the kernel sat 62-67 us on every sub-batch's critical path for a route text never takes)
        hipLaunchKernelGGL(k1p_find, dim3(K1P_GRID * 2u, g.nb), dim3(256), 0, stream, B, g, enable, stage);
        hipLaunchKernelGGL(k1p_verify, dim3(K1P_GRID, g.nb), dim3(256), 0, stream, B, g, stage);
    }
    hipLaunchKernelGGL(k1p_tables, dim3(g.nb), dim3(256), 0, stream, B, g);
    hipLaunchKernelGGL(k1p_fill, dim3(2u * K1P_GRID, g.nb), dim3(256), 0, stream, B, g);
    hipLaunchKernelGGL(k1p_reduce, dim3(g.nb), dim3(256), 0, stream, B, g);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
