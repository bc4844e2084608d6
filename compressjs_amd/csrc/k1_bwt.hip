// K1: batched cyclic suffix sort + BWT for gfx950 (wave64): the driver of the stages, the lane kernels and the BWT gather.
//
// Replaces BWT.bwtransform2 (lib/BWT.js:372-417), i.e. SA-IS on the doubled block plus the gather of lib/BWT.js:407-414.
// The reference's order is: rotations of the block sorted as unsigned bytes, equal rotations by DESCENDING start index
// (SURVEY.md 9.2).  Any algorithm that realises this total order yields identical bytes, so SA-IS is not ported.  For all
// blocks of a batch at once (k1_run):
//
//   1. k1_front.hip   sample-sort front end: every rotation into its place by its first 8 bytes; in cyclic mode the bucket
//                     workgroups go on in LDS (12 more bytes) and list what still ties; list-driven refinement rounds
//                     compare the text, 24 bytes per round (groups of text-like input are tiny and tie for tens of bytes);
//   2. this file      lane kernels: the pairs and groups of 3..8 that outlast the rounds are walked up to 4 KB, one lane each;
//   3. k1_dbl.hip     whatever still ties (long repeats, identical rotations; everything in linear mode): ranks and
//                     list-driven prefix doubling (Larsson-Sadakane), then the tie-break by descending start index;
//   4. this file      U[j] = T[SA[j]-1 mod n], origPtr = position of rotation 0.
//
// Rounds 1-3 also had here: seven LSD radix passes (replaced by the front end in round 2, removed in round 4: blocks of any
// size go through the front end's task levels), the K1-deep tile kernel and the medium-group rounds (text comparison inside
// tiles of the suffix array: replaced by the list-driven rounds in round 3), tile-driven and per-size-class doubling rounds
// (replaced by k1_dbl.hip in round 4).  All integer; no floating point anywhere.
#include "k1_bwt.h"
#include <stdio.h>
#include "devutil.h"
#include <stdlib.h>
#include <vector>

// ---------------------------------------------------------------------------------------------
// init: counters, head bitmap (bits at and beyond n set)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1_init(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y;
    const u32 n = B.nlen[b];
    const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 && gid < K1_STATS) B.stats[gid] = 0;
    if (b == 0) for (u32 i = gid; i < 4u * 8u * K1_DEEP_SUB; i += gridDim.x * blockDim.x) B.deepCnt[i] = 0;
    if (b == 0) for (u32 i = gid; i < (K1R_MAXR + 1u) * B.rnb8 * K1_RCS; i += gridDim.x * blockDim.x) B.rcnt[i] = 0;
    if (b == 0 && gid < K1F_LEVELS * 8u) B.bcnt[gid] = 0;
    if (b == 0) for (u32 i = gid; i < 2u * (K1D_MAXR + 2u) * B.rstride + 2u * B.rstride + (K1D_MAXR + 2u) * 4u; i += gridDim.x * blockDim.x) B.dcnt[i] = 0;   // dcnt, dchg, dtot, dbn, dred (contiguous)
    if (gid < g.hstride) {
        const u32 lo = gid * 32u;
        u32 w;
        if (lo >= n) w = 0xFFFFFFFFu;
        else if (lo + 32u > n) w = 0xFFFFFFFFu << (n - lo);
        else w = 0u;
        B.HN[(size_t)b * g.hstride + gid] = w;
    }
}

// ---------------------------------------------------------------------------------------------
// K1-deep: groups resolved by comparing the TEXT, before any rank exists.
//
// After the 8-byte radix sort most unsorted groups of text-like input are tiny (pairs, triples: a
// phrase and its copies) and tie for the next 10..200 bytes.  Prefix doubling pays one random rank
// gather and one random rank scatter per tied rotation and ROUND for them, out of a rank array that
// is far larger than L2.  Here the same tile/ownership scheme as k1_refine compacts the rotations of
// the owned groups into LDS and then iterates ENTIRELY inside the workgroup: iteration i keys every
// still-tied rotation s with the 8 text bytes at (s + 8 + 8i) mod n (the block's 900 kB of text sit
// in the XCD's L2, see xcd_block_tile), ranks it inside its group by counting, permutes, and splits the
// group where neighbouring keys differ.  No global traffic between iterations, no ranks.  Groups are
// classes of "equal first 8+8i bytes" exactly as a doubling round would produce them, only finer, so
// whatever is still tied after `iters` iterations (long repeats, identical rotations) or was too big
// (> K1_DEEP_BIG rotations after two iterations, > K1_HT at all) is left to the doubling rounds, whose
// invariant "a group at round h shares >= h bytes" holds for the finer partition as well.
// Cyclic mode only (rotations; T_ext wraps).  Replaces no reference code of its own: it is a faster
// route to the order BWT.js:372-417 defines.
// ---------------------------------------------------------------------------------------------
#define K1_DEEP_BIG 64u

// W big-endian 64-bit words of text starting at byte p of T (any alignment), through DWORD-ALIGNED loads plus
// v_alignbyte.  PMC (TCP_TOTAL_CACHE_ACCESSES): a byte-misaligned 16-byte load costs the vector L1 ~4 accesses,
// and the lane kernels below were bound by exactly that (one access per clock and CU).  Reads 8W + 4 bytes from
// p & ~3; the block slots of T_ext are 128 bytes longer than the longest block, so this stays inside the slot.
template <int W>
__device__ __forceinline__ void load_be_words(const u8* T, u32 p, u64* out) {
    const u32 sh = p & 3u;
    u32 d[2 * W + 1];
    __builtin_memcpy(d, __builtin_assume_aligned(T + (p - sh), 4), (2 * W + 1) * 4);
#pragma unroll
    for (int j = 0; j < W; j++) {
        const u32 w0 = __builtin_amdgcn_alignbyte(d[2 * j + 1], d[2 * j], sh);
        const u32 w1 = __builtin_amdgcn_alignbyte(d[2 * j + 2], d[2 * j + 1], sh);
        out[j] = ((u64)__builtin_bswap32(w0) << 32) | (u64)__builtin_bswap32(w1);
    }
}
__device__ __forceinline__ u64 load_be64(const u8* T, u32 p) {
    u64 v;
    load_be_words<1>(T, p, &v);
    return v;
}

// Phase 2 of k1_deep, one lane: the (<= M) members of a group, all known to share their first d bytes, are
// walked W*8 bytes per step with every load of the step in flight together (M*W = 16 loads), until a word
// differs between them.  On return true, keys[i] holds each member's 8 bytes at the first differing word and
// d the number of bytes that members with EQUAL keys share; false: still all equal at capd (left alone).
// T_ext wraps for K1_TPAD = 64 bytes, so a step may read [p, p + 64) for any p < n without reducing mod n.
template <int M, int W>
__device__ __forceinline__ bool deep_walk(const u8* T, u32 n, const u32* members, u32 mstride, u32 gl, u32& d, u32 capd,
                                          u64* keys, u32 kstride) {
    static_assert(W * 8 <= K1_TPAD, "one step must stay inside the wrapped tail of T_ext");
    u32 pp[M];
    const u32 dm = d < n ? d : d % n;
#pragma unroll
    for (int i = 0; i < M; i++) {
        pp[i] = 0;
        if ((u32)i < gl) { u32 q = (members[(u32)i * mstride] & K1_SMASK) + dm; if (q >= n) q -= n; pp[i] = q; }      // (the members may be packed index words)
    }
    for (;;) {
        u64 k[M][W];
#pragma unroll
        for (int i = 0; i < M; i++) {
#pragma unroll
            for (int j = 0; j < W; j++) k[i][j] = 0;
            if ((u32)i < gl) load_be_words<W>(T, pp[i], k[i]);
        }
        int jd = -1;
#pragma unroll
        for (int j = W - 1; j >= 0; j--) {
            bool eq = true;
#pragma unroll
            for (int i = 1; i < M; i++)
                if ((u32)i < gl) eq = eq && k[i][j] == k[0][j];
            if (!eq) jd = j;
        }
        if (jd >= 0) {
            d += 8u * (u32)(jd + 1);
#pragma unroll
            for (int i = 0; i < M; i++) {
                u64 x = k[i][0];
#pragma unroll
                for (int j = 1; j < W; j++) if (j == jd) x = k[i][j];
                if ((u32)i < gl) keys[(u32)i * kstride] = x;
            }
            return true;
        }
        d += 8u * W;
        if (d >= capd) return false;
#pragma unroll
        for (int i = 0; i < M; i++) { pp[i] += 8u * W; if (pp[i] >= n) pp[i] -= n; }
    }
}

#define DP_B(d) ((u32)((d) >> 52))
#define DP_START(d) ((u32)(((d) >> 26) & 0x3FFFFFFu))
#define DP_DEPTH(d) ((u32)(((d) >> 4) & 0xFFFFu))
#define DP_LEN(d) (((u32)(d) & 15u) + 1u)

// Pairs: one lane each, no LDS.  Workgroup L serves the list region of XCD L & 7 (the region holds the groups
// of the blocks whose tiles ran on that XCD, in roughly block order, so their text is in that L2).
// PASS2 = false: the lists the tile kernel and the medium rounds filled, walked up to `capd` (the short cap); what still
// ties there is listed again (second set of regions: listS[cls], counters deepCnt[2 * 8 * K1_DEEP_SUB ..]).
// PASS2 = true: those survivors with the long cap - only if there are at most `limit` of them (decided here, on the
// device: tiled / periodic inputs have ALL their rotations in such groups, and walking 4 KB for each costs more than
// the rank rounds that sort them otherwise: 200 kB of text tiled 29 -> 81 ms when tried).
// Round 3: when at most limit / 8 groups are left (one per 2048 positions), the second pass walks them up to K1_DEEP_LONG_CAP bytes
// instead of capd: a handful of long duplicated passages (E8S-B: 28 775 pairs that tie for up to 16 kB) otherwise costs the whole
// rank machinery - k1_update_ranks' 10^8 scattered stores and a dozen sparse rounds, 1.4 ms per 10^8 bytes - for 57 550 rotations.
#define K1_DEEP_LONG_CAP 60000u
template <bool PASS2>
__device__ __forceinline__ bool deep_pass2_wanted(const K1Buf& B, u32 limit, u32& capd) {
    if (!PASS2) return true;
    u32 t = 0;
    for (u32 i = threadIdx.x & 63u; i < 2u * 8u * K1_DEEP_SUB; i += 64u) t += B.deepCnt[2u * 8u * K1_DEEP_SUB + i];
    for (u32 off = 32; off > 0; off >>= 1) t += __shfl_xor(t, (int)off);
    if (t <= limit / 8u && capd < K1_DEEP_LONG_CAP) capd = K1_DEEP_LONG_CAP;
    return t <= limit;
}

template <bool PASS2>
__global__ __launch_bounds__(256) void k1_deep_pairs(K1Buf B, BatchGeom g, u32 capd, u32 limit, u32 carry) {
    if (!deep_pass2_wanted<PASS2>(B, limit, capd)) return;
    // gridDim.x is a multiple of 8 * K1_DEEP_SUB: workgroup -> (XCD region, sub-region, slice of the sub-region)
    const u32 xr = (blockIdx.x & 7u) * K1_DEEP_SUB + ((blockIdx.x >> 3) & (K1_DEEP_SUB - 1u));
    const u32 r = blockIdx.x / (8u * K1_DEEP_SUB), nr = gridDim.x / (8u * K1_DEEP_SUB);
    const u32 rcap = (PASS2 ? B.listSCap : B.listTCap) / (8u * K1_DEEP_SUB), rcap2 = B.listSCap / (8u * K1_DEEP_SUB);
    u32 cnt = B.deepCnt[(PASS2 ? 2u * 8u * K1_DEEP_SUB : 0u) + xr];
    if (cnt > rcap) cnt = rcap;
    const u64* L = (PASS2 ? B.listS[0] : B.listT[0]) + (size_t)xr * rcap;
    for (u32 gi = r * 256u + threadIdx.x; gi < cnt; gi += nr * 256u) {
        const u64 dsc = L[gi];
        const u32 b = DP_B(dsc), start = DP_START(dsc);
        u32 d = DP_DEPTH(dsc);
        const u32 n = B.nfront[b];
        const u8* T = B.T + (size_t)b * g.tstride;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u8* U = B.U + (size_t)b * g.stride + start;          // (carry mode: the BWT bytes of these positions move with the rotations)
        u32 mem[2];
        u64 keys[2];
        mem[0] = SA[0];
        mem[1] = SA[1];
        if (deep_walk<2, 8>(T, n, mem, 1u, 2u, d, capd, keys, 1u)) {
            if (keys[0] > keys[1]) {
                SA[0] = mem[1]; SA[1] = mem[0];
                if (carry) { const u8 u0 = U[0], u1 = U[1]; U[0] = u1; U[1] = u0; }
            }
            atomicOr(&B.HN[(size_t)b * g.hstride + ((start + 1u) >> 5)], 1u << ((start + 1u) & 31u));
        } else if (!PASS2) {
            const u32 idx = atomicAdd(&B.deepCnt[2u * 8u * K1_DEEP_SUB + xr], 1u);
            if (idx < rcap2) B.listS[0][(size_t)xr * rcap2 + idx] = (dsc & ~((u64)0xFFFFu << 4)) | ((u64)(d < 0xFFFFu ? d : 0xFFFFu) << 4);
        }
    }
}

// Groups of 3..K1_DEEP_LANE rotations: one lane each, members in a per-lane LDS column.  The lane walks the text
// of all members until a word differs, sorts them by that word, and goes on depth-first with every run of
// equal keys (own depth per run) until the group is resolved or a run ties up to capd (left as it is).
template <bool PASS2>
__global__ __launch_bounds__(256) void k1_deep_small(K1Buf B, BatchGeom g, u32 capd, u32 limit, u32 carry) {
    if (!deep_pass2_wanted<PASS2>(B, limit, capd)) return;
    __shared__ u64 lk[K1_DEEP_LANE * 256];
    __shared__ u32 lv[K1_DEEP_LANE * 256];
    __shared__ u16 ld[K1_DEEP_LANE * 256];
    const u32 tid = threadIdx.x;
    const u32 xr = (blockIdx.x & 7u) * K1_DEEP_SUB + ((blockIdx.x >> 3) & (K1_DEEP_SUB - 1u));
    const u32 r = blockIdx.x / (8u * K1_DEEP_SUB), nr = gridDim.x / (8u * K1_DEEP_SUB);
    const u32 rcap = (PASS2 ? B.listSCap : B.listTCap) / (8u * K1_DEEP_SUB), rcap2 = B.listSCap / (8u * K1_DEEP_SUB);
    u32 cnt = B.deepCnt[(PASS2 ? 3u : 1u) * 8u * K1_DEEP_SUB + xr];
    if (cnt > rcap) cnt = rcap;
    const u64* L = (PASS2 ? B.listS[1] : B.listT[1]) + (size_t)xr * rcap;
    u64* ck = lk + tid;
    u32* cv = lv + tid;
    u16* cd = ld + tid;
    for (u32 gi = r * 256u + tid; gi < cnt; gi += nr * 256u) {
        const u64 dsc = L[gi];
        const u32 b = DP_B(dsc), start = DP_START(dsc), gl = DP_LEN(dsc);
        const u32 n = B.nfront[b];
        const u8* T = B.T + (size_t)b * g.tstride;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u8* U = B.U + (size_t)b * g.stride + start;
        u32 mem[K1_DEEP_LANE];
#pragma unroll
        for (int i = 0; i < (int)K1_DEEP_LANE; i++) mem[i] = (u32)i < gl ? SA[i] : 0u;     // all loads in flight together
        if (carry) {                                                  // the BWT bytes ride in the index words while the group is sorted
#pragma unroll
            for (int i = 0; i < (int)K1_DEEP_LANE; i++) if ((u32)i < gl) mem[i] = K1_SPACK(mem[i], U[i]);
        }
#pragma unroll
        for (int i = 0; i < (int)K1_DEEP_LANE; i++)
            if ((u32)i < gl) { cv[(u32)i * 256u] = mem[i]; cd[(u32)i * 256u] = (u16)DP_DEPTH(dsc); }
        u32 heads = 1u, done = 0u;
        for (;;) {
            u32 a = 0, e = 0;
            while (a < gl) {                                          // first run of >= 2 that is not given up
                const u32 rest = (heads >> (a + 1u)) & ((1u << (gl - a - 1u)) - 1u);
                e = rest ? a + (u32)__ffs((int)rest) : gl;
                if (e - a >= 2u && !((done >> a) & 1u)) break;
                a = e;
            }
            if (a >= gl) break;
            const u32 len = e - a;
            u32 d = cd[a * 256u];
            const bool split = len == 2u ? deep_walk<2, 8>(T, n, cv + a * 256u, 256u, len, d, capd, ck + a * 256u, 256u)
                             : len <= 4u ? deep_walk<4, 4>(T, n, cv + a * 256u, 256u, len, d, capd, ck + a * 256u, 256u)
                                         : deep_walk<(int)K1_DEEP_LANE, 2>(T, n, cv + a * 256u, 256u, len, d, capd, ck + a * 256u, 256u);
            if (!split) {
                done |= 1u << a;
                if (!PASS2) {                                         // ties up to the short cap: listed for the second pass
                    const u32 idx = atomicAdd(&B.deepCnt[3u * 8u * K1_DEEP_SUB + xr], 1u);
                    if (idx < rcap2) B.listS[1][(size_t)xr * rcap2 + idx] = ((u64)b << 52) | ((u64)(start + a) << 26) | ((u64)(d < 0xFFFFu ? d : 0xFFFFu) << 4) | (u64)(len - 1u);
                }
                continue;
            }
            for (u32 i = a + 1u; i < e; i++) {                        // insertion sort of (ck, cv)[a .. e)
                const u64 x = ck[i * 256u];
                const u32 v = cv[i * 256u];
                u32 j = i;
                while (j > a && ck[(j - 1u) * 256u] > x) { ck[j * 256u] = ck[(j - 1u) * 256u]; cv[j * 256u] = cv[(j - 1u) * 256u]; j--; }
                ck[j * 256u] = x;
                cv[j * 256u] = v;
            }
            for (u32 i = a + 1u; i < e; i++) if (ck[i * 256u] != ck[(i - 1u) * 256u]) heads |= 1u << i;
            for (u32 i = a; i < e; i++) cd[i * 256u] = (u16)(d < 0xFFFFu ? d : 0xFFFFu);
        }
        for (u32 i = 0; i < gl; i++) {
            const u32 v = cv[i * 256u];
            SA[i] = v & K1_SMASK;
            if (carry) U[i] = (u8)(v >> 24);
        }
        const u64 bits = (u64)(heads & ~1u) << (start & 31u);
        u32* HN = B.HN + (size_t)b * g.hstride + (start >> 5);
        if ((u32)bits) atomicOr(&HN[0], (u32)bits);
        if ((u32)(bits >> 32)) atomicOr(&HN[1], (u32)(bits >> 32));
    }
}

// ---------------------------------------------------------------------------------------------
// How many rotations are still in unsorted groups under HN, per block?  (position p is settled iff p and p + 1 are heads;
// bits at and beyond n are set.)  One bitmap word per thread.  Where the answer is 0 -- random data after the first sort,
// phrase-reuse text after the text stages -- the rank pass (10^8 random 4-byte stores, 1.1 ms) and the doubling rounds are skipped.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1_count_unsorted(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y, n = B.nfront[b];
    const u32 wi = blockIdx.x * 256u + threadIdx.x;
    const u32* HN = B.HN + (size_t)b * g.hstride;
    u32 c = 0;
    if (wi * 32u < n) {
        const u32 h = HN[wi], hx = HN[wi + 1u];
        c = (u32)__popc(~(h & ((h >> 1) | (hx << 31))));
    }
    for (u32 off = 32; off; off >>= 1) c += __shfl_xor(c, off);
    __shared__ u32 part[4];
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(&B.dtot[K1_BI(B, b)], t);                  // per block (at most ~110 adds per word)
    }
}

// ---------------------------------------------------------------------------------------------
// BWT gather (lib/BWT.js:407-414)
// ---------------------------------------------------------------------------------------------
// Four suffix-array entries per thread: one 16-byte load, four text gathers in flight, one 4-byte store (round 3; one entry per
// thread was 4x the workgroups and a byte store each).
// Round 5, `carry`: the text stages wrote U next to the suffix array (the byte in front of a rotation travels with it from k1f_scatter on), so a block whose
// order they finished on their own needs no gather - only origPtr is looked up here (enwik: 0.46 -> 0.0x ms).  A block that the doubling rounds (dtot), the
// closed form (per) or the three-period reduction (red) had a hand in is gathered as before.
__global__ __launch_bounds__(256) void k1_finish(K1Buf B, BatchGeom g, u32 carry) {
    u32 b, tt;
    if (!xcd_block_tile(g.nb, b, tt)) return;
    const u32 n = B.nlen[b];
    const u32 p0 = (tt * 256u + threadIdx.x) * 4u;
    if (p0 >= n) return;
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* SA = (B.red[b] ? B.SB : B.SA) + (size_t)b * g.stride;     // (k1_period.hip: expanded from the reduced block's)
    u8* U = B.U + (size_t)b * g.stride;
    if (carry && B.dtot[K1_BI(B, b)] == 0u && B.per[b] == 0u && B.red[b] == 0u) {
        if (p0 + 4u <= n) {
            const uint4 v = *(const uint4*)(SA + p0);
            if (v.x == 0u) B.pidx[b] = p0;
            if (v.y == 0u) B.pidx[b] = p0 + 1u;
            if (v.z == 0u) B.pidx[b] = p0 + 2u;
            if (v.w == 0u) B.pidx[b] = p0 + 3u;
        } else {
            for (u32 p = p0; p < n; p++) if (SA[p] == 0u) B.pidx[b] = p;
        }
        return;
    }
    if (p0 + 4u <= n) {
        const uint4 v = *(const uint4*)(SA + p0);          // stride is a multiple of 4 entries: 16-byte aligned
        const u32 s[4] = {v.x, v.y, v.z, v.w};
        // Round 6: a block the doubling rounds finished is gathered only where they had to - a position that was a group of ONE before them
        // (head bit set, and the next one: the doubling stage never writes HN, the text stages keep it exact) was settled by someone who wrote its
        // byte next to its suffix-array entry (carry mode).  E8S-A: 57 % of the positions - 0.39 -> 0.2x ms of random one-byte gathers per step.
        u32 keep = 0, old = 0;
        if (carry && B.per[b] == 0u && B.red[b] == 0u) {
            const u32* HN = B.HN + (size_t)b * g.hstride;
            const u64 hh = ((u64)HN[p0 >> 5] | ((u64)HN[(p0 >> 5) + 1u] << 32)) >> (p0 & 31u);
            keep = (u32)(hh & (hh >> 1)) & 15u;
            if (keep) old = *(const u32*)(U + p0);
        }
        u32 out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((keep >> k) & 1u) out |= old & (0xFFu << (8 * k));
            else out |= (u32)T[s[k] == 0 ? n - 1 : s[k] - 1] << (8 * k);
            if (s[k] == 0) B.pidx[b] = p0 + (u32)k;
        }
        *(u32*)(U + p0) = out;
    } else {
        for (u32 p = p0; p < n; p++) {
            const u32 s = SA[p];
            U[p] = T[s == 0 ? n - 1 : s - 1];
            if (s == 0) B.pidx[b] = p;
        }
    }
}
__global__ __launch_bounds__(256) void k1_finish_linear(K1Buf B, BatchGeom g, int* SAout) {
    const u32 b = blockIdx.y;
    const u32 n = B.nlen[b];
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32 s = B.SA[(size_t)b * g.stride + p];
    const u32 p0 = B.ISA[(size_t)b * g.stride];          // rank of suffix 0 (all ranks are final)
    if (SAout) SAout[(size_t)b * g.stride + p] = (int)s;
    u8* U = B.U + (size_t)b * g.stride;
    if (p == 0) U[0] = T[n - 1];
    if (s == 0) B.pidx[b] = p + 1u;
    else U[p < p0 ? p + 1u : p] = T[s - 1];
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }


// one table for the sizes and the carving: (field, bytes)
template <class F>
static void k1_layout(K1Buf& B, const BatchGeom& g, F&& take) {
    const size_t e = (size_t)g.nb * g.stride;
    const u32 nb8 = (g.nb + 7u) & ~7u;
    B.rnb8 = nb8;
    B.rstride = 8u * ((nb8 / 8u + 31u) & ~31u);
    const u32 rs = B.rstride;
    B.listTCap = g.nb * (g.stride / K1D_GS);                // descriptors of groups of K1D_GS+1 .. 1024 rotations
    B.listSCap = g.nb * (g.stride / 8u);                    // lane kernels' lists (8 XCD regions x K1_DEEP_SUB sub-regions); chunks of large groups
    B.listMCap = g.nb * (g.stride / 1024u + 1u);            // descriptors of groups of 1025 .. K1_MED_MAX
    B.listLCap = g.nb * (g.stride / K1_MED_MAX + 1u);       // ... of larger ones
    B.btaskLists = g.nb < 8u ? (g.nb ? g.nb : 1u) : 8u;
    B.btaskCap = ((g.nb + 7u) / 8u) * (g.stride / 256u) * 2u;      // tasks per level AND XCD list (block mod 8): twice what its blocks' rotations make at 256 per task
    take((void**)&B.SA, e * 4);
    take((void**)&B.SB, e * 4);
    take((void**)&B.ISA, e * 4);
    take((void**)&B.KA, e * 4);
    take((void**)&B.KB, e * 4);
    take((void**)&B.HN, (size_t)g.nb * g.hstride * 4);
    take((void**)&B.tileHist, (size_t)g.nb * k1_front_tilehist_words(g) * 4);
    take((void**)&B.fsplit, (size_t)g.nb * K1F_NB * 8);
    take((void**)&B.fsub, (size_t)g.nb * K1F_NB);
    take((void**)&B.fsplit2, (size_t)g.nb * K1F_NB * 8);
    take((void**)&B.fp16, (size_t)g.nb * K1F_NB);
    take((void**)&B.fstart, (size_t)g.nb * (K1F_NB + 1) * 4);
    take((void**)&B.stats, K1_STATS * 4);
    take((void**)&B.deepCnt, 4 * 8 * K1_DEEP_SUB * 4);
    for (int k = 0; k < 2; k++) take((void**)&B.listT[k], (size_t)B.listTCap * 8);
    for (int k = 0; k < 2; k++) take((void**)&B.listS[k], (size_t)B.listSCap * 8);
    for (int k = 0; k < 2; k++) take((void**)&B.listM[k], (size_t)B.listMCap * 8);
    for (int k = 0; k < 2; k++) take((void**)&B.listL[k], (size_t)B.listLCap * 8);
    for (int k = 0; k < 2; k++) take((void**)&B.rlist[k], e * 8);
    take((void**)&B.rcnt, (size_t)(K1R_MAXR + 1) * nb8 * K1_RCS * 4);
    take((void**)&B.dcnt, ((size_t)(2u * (K1D_MAXR + 2u) + 2u) * rs + (K1D_MAXR + 2u) * 4u) * 4);      // dcnt, dchg, dtot, dbn, dred (contiguous: zeroed as one; dtot .. dred read back as one)
    take((void**)&B.btask, (size_t)K1F_LEVELS * 8u * B.btaskCap * sizeof(uint4));
    take((void**)&B.bcnt, 256);
    take((void**)&B.nfront, (size_t)nb8 * 4);
    take((void**)&B.per, (size_t)nb8 * 4);
    take((void**)&B.ptab, (size_t)g.nb * 256 * 4);
    take((void**)&B.red, (size_t)nb8 * 4);
    B.dchg = B.dcnt ? B.dcnt + (size_t)(K1D_MAXR + 2u) * rs : nullptr;
    B.dtot = B.dcnt ? B.dchg + (size_t)(K1D_MAXR + 2u) * rs : nullptr;
    B.dbn = B.dcnt ? B.dtot + rs : nullptr;
    B.dred = B.dcnt ? B.dbn + (K1D_MAXR + 2u) * 4u : nullptr;
}

size_t k1_workspace_bytes(const BatchGeom& g) {
    K1Buf B{};
    size_t tot = 0;
    k1_layout(B, g, [&](void** field, size_t bytes) { *field = nullptr; tot += al256(bytes); });
    return tot;
}

void k1_carve(K1Buf& B, const BatchGeom& g, void* ws) {
    B.hpin = nullptr; B.hpinWords = 0;
    char* p = (char*)ws;
    k1_layout(B, g, [&](void** field, size_t bytes) { *field = p; p += al256(bytes); });
}

static int g_k1_last_sparse_rounds = 0, g_k1_last_rounds = 0, g_k1_last_periodic = 0;
extern "C" int cjs_dbg_k1_periodic_blocks() { return g_k1_last_periodic; }
extern "C" int cjs_dbg_k1_sparse_rounds() { return g_k1_last_sparse_rounds; }
extern "C" int cjs_dbg_k1_rounds() { return g_k1_last_rounds; }

// The knobs of K1, read once per process (A/B runs and the variant tests; the defaults are what the numbers in DESIGN.md are for)
struct K1Knobs {
    u32 bsort_iters;   // CJS_BSORT_ITERS   1 (default): the bucket sort compares 16 bytes (the groups it lists share 16); 0: 8 bytes
    u32 text_bytes;    // CJS_TEXT_BYTES    depth up to which the refinement rounds compare the text (default 264; 0: no text stages, doubling from 8 bytes)
    u32 big_div;       // CJS_DEEP_BIG_DIV  text stages are skipped when more than 1/DIV of the rotations sit in one-key buckets (default 8)
    u32 lane_cap;      // CJS_DEEP_LANE_CAP bytes the lane kernels walk a pair / small group that outlasted the rounds (default 4096; 0: not at all)
    bool period;       // CJS_K1_PERIOD     0: no closed form for blocks with a small period (k1_period.hip): they take the general path
    u32 check_h;       // CJS_K1_CHECK_H    from the doubling round with this h on the host looks at the lists after every round and stops when they are empty (default 8192; 0: never)
    bool sync;         // CJS_K1_SYNC       0: no read-back at all (every launch of the doubling stage is enqueued whatever is left)
    bool trace;        // CJS_K1_TRACE      counters of the stages on stderr (reads them back: not for timing)
};
static const K1Knobs& k1_knobs() {
    static const K1Knobs k = []() {
        auto num = [](const char* name, u32 dflt, u32 hi) -> u32 {
            const char* e = getenv(name);
            if (!e) return dflt;
            const unsigned long v = strtoul(e, nullptr, 10);
            return v > hi ? hi : (u32)v;
        };
        K1Knobs q;
        q.bsort_iters = num("CJS_BSORT_ITERS", 1u, 1u);
        q.text_bytes = num("CJS_TEXT_BYTES", 264u, 32000u);
        q.big_div = num("CJS_DEEP_BIG_DIV", 8u, 1u << 30);
        if (!q.big_div) q.big_div = 8u;
        q.lane_cap = num("CJS_DEEP_LANE_CAP", 4096u, 60000u);
        q.period = num("CJS_K1_PERIOD", 1u, 1u) != 0u;
        q.check_h = num("CJS_K1_CHECK_H", 8192u, 1u << 22);
        q.sync = num("CJS_K1_SYNC", 1u, 1u) != 0u;
        q.trace = getenv("CJS_K1_TRACE") != nullptr;
        return q;
    }();
    return k;
}

// The whole suffix sort of a batch, enqueued on `stream`:
//   1. k1_front.hip: sample-sort front end - every rotation into its place by its first 8 bytes (k1f_sample / hist / scan /
//      scatter / bsort, task levels for what does not fit a workgroup's LDS); in cyclic mode the bucket workgroups go on in
//      LDS (CJS_BSORT_ITERS x 12 bytes) and list what still ties;
//   2. the refinement rounds over those lists (k1r_round: 24 text bytes per round) and the lane kernels for the pairs and
//      small groups that outlast them (up to CJS_DEEP_LANE_CAP bytes) - cyclic mode, and only while the predictor
//      (CJS_DEEP_BIG_DIV) expects ties to be short;
//   3. k1_dbl.hip: whatever still ties - ranks, list-driven prefix doubling from h = 8, the tie-break;
//   4. the BWT gather.
int k1_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream) {
    const K1Knobs& K = k1_knobs();
    B.btaskLists = g.nb < 8u ? (g.nb ? g.nb : 1u) : 8u;   // (the batch at hand, not the geometry the workspace was carved for)
    const u32 d0 = 8u;                                     // bytes every group shares after the front end
    const u64 total_n = (u64)g.nb * max_n;
    int rounds_with_work = 0;
    hipLaunchKernelGGL(k1_init, dim3((g.hstride + 255) / 256, g.nb), dim3(256), 0, stream, B, g);
    {
        const int rc = k1_period_run(B, g, max_n, stream, K.period ? 1u : 0u);
        if (rc) return rc;
    }
    if (K.trace) {
        std::vector<u32> per(g.nb);
        HIP_CHECK_RET(hipMemcpyAsync(per.data(), B.per, (size_t)g.nb * 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipStreamSynchronize(stream));
        std::vector<u32> red(g.nb);
        HIP_CHECK_RET(hipMemcpyAsync(red.data(), B.red, (size_t)g.nb * 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipStreamSynchronize(stream));
        g_k1_last_periodic = 0;
        int nred = 0;
        for (u32 b = 0; b < g.nb; b++) { g_k1_last_periodic += per[b] ? 1 : 0; nred += red[b] ? 1 : 0; }
        g_k1_last_periodic += nred << 16;
        fprintf(stderr, "[k1] of %u blocks: %d with a period <= 64 (closed form), %d with a longer period (sorted through their first 3 periods)\n", g.nb, g_k1_last_periodic & 0xFFFF, nred);
    }
    // the text stages: cyclic mode, blocks whose indices fit the list entries' 22 bits
    const bool fused = !B.linear && K.text_bytes > d0 && max_n < (1u << 22);
    // the byte in front of a rotation travels with it through the text stages (k1_bwt.h: K1_SPACK): blocks below 2^20 bytes (CJS_K1_CARRY=0: off)
    static const bool carry_on = []() { const char* e = getenv("CJS_K1_CARRY"); return !e || atoi(e) != 0; }();
    const u32 carry = (fused && carry_on && max_n < K1_CARRY_MAXN) ? 1u : 0u;
    {
        const int rc = k1_front_run(B, g, max_n, stream, fused ? K.bsort_iters : 0u, fused ? 1u : 0u, (u32)(total_n / K.big_div), carry);
        if (rc) return rc;
    }
    if (K.trace) {
        u32 fs[K1_STATS - K1_STAT_FRONT_BIG];
        HIP_CHECK_RET(hipMemcpyAsync(fs, B.stats + K1_STAT_FRONT_BIG, sizeof fs, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipStreamSynchronize(stream));
        u64 bg = 0;
        for (u32 i = 0; i < 8u; i++) bg += fs[K1_STAT_BIGROT - K1_STAT_FRONT_BIG + i];
        {   // task levels: tasks, the longest one and its depth
            std::vector<u32> bc(K1F_LEVELS * 8u);
            HIP_CHECK_RET(hipMemcpy(bc.data(), B.bcnt, K1F_LEVELS * 8u * 4, hipMemcpyDeviceToHost));
            fprintf(stderr, "[k1] task levels (tasks / longest / its depth):");
            const u32 cap8 = B.btaskCap;
            for (u32 lv = 0; lv < K1F_LEVELS; lv++) {
                u32 ntot = 0, ml = 0, md = 0;
                for (u32 x = 0; x < 8u; x++) {               // (one list per level and XCD)
                    const u32 nt = bc[lv * 8u + x] < cap8 ? bc[lv * 8u + x] : cap8;
                    std::vector<uint4> tk(nt);
                    if (nt) HIP_CHECK_RET(hipMemcpy(tk.data(), B.btask + (size_t)(lv * 8u + x) * cap8, (size_t)nt * sizeof(uint4), hipMemcpyDeviceToHost));
                    for (const uint4& t : tk) if (t.z > ml) { ml = t.z; md = t.w & 0x3FFFFFFFu; }
                    ntot += nt;
                }
                fprintf(stderr, " [%u] %u/%u/%u", lv, ntot, ml, md);
            }
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[k1] front end: %u oversize buckets, %llu of %llu rotations in 8-byte groups above 64, %u in one-key buckets; k1f_bsort stage clocks/256 (K1F_TRACE builds): load %u  sample %u  partition %u  rank %u  place+heads %u  flush %u\n",
                fs[0], (unsigned long long)bg, (unsigned long long)total_n, fs[K1_STAT_PUREROT - K1_STAT_FRONT_BIG], fs[1], fs[2], fs[3], fs[4], fs[5], fs[6]);
    }
    if (fused) {
        // the refinement rounds over what k1f_bsort and the task levels listed; a block ends them early when its list stops shrinking
        const u32 depth0 = K.bsort_iters ? K1F_KEYB : d0;           // bytes every listed group shares
        const int rc = k1_rounds_run(B, g, stream, depth0, K.text_bytes, carry);
        if (rc) return rc;
        if (K.trace) {
            std::vector<u32> rc2((size_t)(K1R_MAXR + 1) * B.rnb8 * K1_RCS);
            u32 rt[8];
            HIP_CHECK_RET(hipMemcpyAsync(rc2.data(), B.rcnt, rc2.size() * 4, hipMemcpyDeviceToHost, stream));
            HIP_CHECK_RET(hipMemcpyAsync(rt, B.stats + K1_STAT_RTRACE, sizeof rt, hipMemcpyDeviceToHost, stream));
            HIP_CHECK_RET(hipStreamSynchronize(stream));
            fprintf(stderr, "[k1] refinement rounds, entries per round:");
            for (u32 r = 0; r <= K1R_MAXR; r++) {
                u64 tot = 0;
                for (u32 bb = 0; bb < g.nb; bb++) tot += rc2[((size_t)r * B.rnb8 + bb) * K1_RCS];
                if (!tot) break;
                fprintf(stderr, " %llu", (unsigned long long)tot);
            }
            fprintf(stderr, "\n[k1] k1r_round stage clocks/256 (K1F_TRACE builds): load %u  keys %u  rank %u  classify %u  reserve %u  write %u\n", rt[0], rt[1], rt[2], rt[3], rt[4], rt[5]);
        }
        // lane kernels: the pairs and groups of 3..8 the last round of a block handed over are walked up to CJS_DEEP_LANE_CAP bytes
        // (boilerplate passages of text tie for up to ~3 KB) - when at most one group in 256 positions is left (decided on the device)
        const u32 capd0 = depth0 + K1R_STEP * ((K.text_bytes - depth0 + K1R_STEP - 1u) / K1R_STEP);
        if (K.lane_cap > capd0) {
            const u32 lane_unit = 8u * K1_DEEP_SUB;          // one workgroup per (XCD region, sub-region) at least
            const u32 lane_grid = g.nb * 32u <= lane_unit ? lane_unit : (g.nb * 32u >= 4096u ? 4096u : (g.nb * 32u + lane_unit - 1u) / lane_unit * lane_unit);
            const u32 limit = (u32)(total_n / 256u);
            hipLaunchKernelGGL(k1_deep_pairs<true>, dim3(lane_grid), dim3(256), 0, stream, B, g, K.lane_cap, limit, carry);
            hipLaunchKernelGGL(k1_deep_small<true>, dim3(lane_grid), dim3(256), 0, stream, B, g, K.lane_cap, limit, carry);
        }
    }
    // What still ties (long repeats, identical rotations, groups the text stages did not take; in linear mode everything beyond
    // the first 8 bytes): k1_count_unsorted finds the blocks that hold groups, k1_dbl.hip ranks their rotations and runs the
    // list-driven doubling rounds from h = 8 - every group shares at least the 8 bytes of the first sort.
    hipLaunchKernelGGL(k1_count_unsorted, dim3((g.hstride + 255u) / 256u, g.nb), dim3(256), 0, stream, B, g);
    // The one read-back K1 keeps (CJS_K1_SYNC=0: none): when no block holds a group - the text stages finished the batch, the
    // usual case on text - the ~90 launches of the doubling stage (4.6 us each when they find their lists empty: 0.35 ms per
    // 10^8 bytes, measured) are not enqueued at all.  Everything after this point is steered on the device.
    // (The same look tells whether any block was reduced by k1_period.hip: if none, the expansion's launches are left out too - empty
    // kernels are only cheap on an idle GPU: next to the other stream's k2_mtf one of them sat 190 us in the queue.)
    bool any_group = true, any_red = true;
    u32 h0 = d0;                                           // depth the doubling rounds start from
    if (K.sync && !B.linear) {
        // (ONE copy - dtot, dbn and dred lie behind one another -, into pinned memory when the caller has some: a second pageable copy was
        // issued 65 us after the first on the kernel timeline)
        const size_t words = (size_t)(B.dred - B.dtot) + B.rstride;
        std::vector<u32> pageable;
        u32* tt = B.hpin;
        if (!tt || words > B.hpinWords) { pageable.resize(words); tt = pageable.data(); }
        HIP_CHECK_RET(hipMemcpyAsync(tt, B.dtot, words * 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipStreamSynchronize(stream));
        any_group = any_red = false;
        const u32* dr = tt + (B.dred - B.dtot);
        for (u32 bb = 0; bb < B.rstride; bb++) { any_group = any_group || tt[bb] != 0u; any_red = any_red || dr[bb] != 0u; }   // (any order: K1_BI)
        // round 6: 16-byte keys in every bucket sort and nobody left a shallower group (k1_bwt.h: K1_DEEP_START): the rounds start at h = 16
        if (K1_DEEP_START && fused && K.bsort_iters && tt[&K1D_SHALLOW(B) - B.dtot] == 0u) h0 = K1F_KEYB;
    }
    if (any_group) {
        const int rc = k1_dbl_run(B, g, max_n, stream, h0, K.sync ? K.check_h : 0u);
        if (rc) return rc;
    }
    if (K.trace) {
        static thread_local u32 dc[(K1D_MAXR + 2) * 4];
        HIP_CHECK_RET(hipMemcpyAsync(dc, B.dbn, sizeof dc, hipMemcpyDeviceToHost, stream));
        std::vector<u32> cn((size_t)(K1D_MAXR + 2) * B.rstride), tt(B.rstride);
        HIP_CHECK_RET(hipMemcpyAsync(cn.data(), B.dcnt, cn.size() * 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipMemcpyAsync(tt.data(), B.dtot, tt.size() * 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipStreamSynchronize(stream));
        u64 un = 0;
        for (u32 bb = 0; bb < B.rstride; bb++) un += tt[bb];
        fprintf(stderr, "[k1] rotations in unsorted groups before the doubling rounds: %llu (of %llu), rounds from h = %u; per round entries / medium (<= 1024 + larger) / large / chunks:", (unsigned long long)un, (unsigned long long)total_n, h0);
        for (u32 r = 0; r < K1D_MAXR + 1u; r++) {
            u64 e = 0;
            for (u32 bb = 0; bb < B.rstride; bb++) e += cn[(size_t)r * B.rstride + bb];
            if (!e && !dc[r * 4] && !dc[r * 4 + 1] && !dc[r * 4 + 3]) break;
            fprintf(stderr, " [%u] %llu/%u+%u/%u/%u", r, (unsigned long long)e, dc[r * 4 + 3], dc[r * 4], dc[r * 4 + 1], dc[r * 4 + 2]);
            rounds_with_work = (int)r + 1;
        }
        fprintf(stderr, "\n");
    }
    g_k1_last_sparse_rounds = rounds_with_work;             // (known with CJS_K1_TRACE only)
    g_k1_last_rounds = rounds_with_work;
    if (!B.linear && K.period && any_red) {
        const int rc = k1_period_expand(B, g, max_n, stream);
        if (rc) return rc;
    }
    if (B.linear) hipLaunchKernelGGL(k1_finish_linear, dim3((max_n + 255) / 256, g.nb), dim3(256), 0, stream, B, g, B.SAout);
    else hipLaunchKernelGGL(k1_finish, dim3((max_n + 1023) / 1024, (g.nb + 7u) & ~7u), dim3(256), 0, stream, B, g, carry);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

int k1_prof_enable(K1Prof& p, int on) {
    if (on && !p.ev) {
        p.ev = new hipEvent_t[2 * K1_PROF_MAX];
        p.cls = new unsigned char[K1_PROF_MAX];
        for (int i = 0; i < 2 * K1_PROF_MAX; i++) HIP_CHECK_RET(hipEventCreate(&p.ev[i]));
    }
    p.enabled = on;
    if (on) {
        p.used = 0;
        p.dbl_runs = 0;
        for (int i = 0; i < K1_PROF_CLASSES; i++) p.elements[i] = 0;
    }
    return CJS_OK;
}
// totals of one class since the profile was enabled (the records stay: every class can be read)
int k1_prof_read(K1Prof& p, u32 cls, float* total_ms, u32* launches, u64* elements) {
    float tot = 0.f;
    u32 n = 0;
    const u32 used = p.used > K1_PROF_MAX ? K1_PROF_MAX : p.used;
    for (u32 i = 0; p.ev && i < used; i++) {
        if (p.cls[i] != cls) continue;
        float ms = 0.f;
        HIP_CHECK_RET(hipEventSynchronize(p.ev[2 * i + 1]));
        HIP_CHECK_RET(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
        tot += ms;
        n++;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    if (elements) *elements = cls < K1_PROF_CLASSES ? p.elements[cls] : 0;
    return CJS_OK;
}
void k1_prof_destroy(K1Prof& p) {
    if (p.ev) {
        for (int i = 0; i < 2 * K1_PROF_MAX; i++) (void)hipEventDestroy(p.ev[i]);
        delete[] p.ev;
        delete[] p.cls;
        p.ev = nullptr;
        p.cls = nullptr;
    }
}
