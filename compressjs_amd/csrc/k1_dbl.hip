// K1, last stage: list-driven prefix doubling (round 4).
//
// Replaces, together with the stages before it, the suffix sort of BWT.bwtransform2 (lib/BWT.js:372-417; SA-IS, lib/BWT.js:197-300,
// is linear whatever the repeats look like - this stage is what keeps inputs with LONG repeats, HTML-like text first of all,
// from paying for them byte by byte).  Input: a suffix array whose groups (runs of positions without a head bit in HN) are
// classes of rotations that share at least h0 bytes, in the right order group against group.  Larsson-Sadakane doubling:
// ISA[s] = position of the head of s's group; round r sorts every group by ISA[(s + h) mod n], h = h0 << r, which makes its
// sub-groups classes of equal 2h-prefixes.  What is new against rounds 1-3 (tile rounds over the whole bitmap + one kernel per
// size class over descriptor lists, 75-96 ps per rotation and round): every lane of every round works on a rotation that still
// ties, and the host never looks at a counter.
//
//   k1d_build   one pass over (SA, HN): ISA for every rotation of the blocks that still hold groups, and the lists -
//               groups of 2..K1D_GS rotations as 8-byte ENTRIES, one per rotation (K1E_MAKE: length, index in the group,
//               rotation, position; a group = consecutive entries of the block's list), larger groups as descriptors;
//   per round   k1d_round   the entry lists: a workgroup owns the groups that start in its 1024 entries (it reads K1D_GS
//                           ahead), gathers one 4-byte key per rotation, ranks inside the group by counting (keys in LDS, four
//                           per read, lanes of a group read the same address), writes the new entries in their new order
//                           and the rotations that ended up alone to the suffix array.  A software pipeline over the tiles
//                           a workgroup walks: the entries of the tile after next and the keys of the next one are in flight.
//               k1d_med     groups of K1D_GS+1 .. K1_MED_MAX: one workgroup each, bitonic network on (key, rotation) in LDS;
//               k1d_large   larger ones: one 1024-thread workgroup each - a three-way partition around the majority key when
//                           nearly all keys are equal (repetitive input), else LSD passes through global memory;
//                           both leave, per position, the head of its new sub-group (R = SB) and at the heads the length
//                           (KB), and cut their range into chunks of 1024 positions;
//               k1d_update  all reads of a round see the OLD ranks (a group whose keys mix old and new ranks can be mis-ordered),
//                           so the ranks are written by this second kernel: ISA from the new entries (only where the rank
//                           changed), survivors compacted into the next round's list; and, chunk by chunk, ISA for the big
//                           groups and their sub-groups as entries / descriptors of the next round.
//   A block in which a round split nothing holds only identical rotations (see k1d_mode): its next round is the tie-break by
//   descending start index, as is the round with h >= n.  Rounds run on whatever the lists hold: an empty
//   list costs a workgroup one load.  Linear mode (BWT.bwtransform / suffixsort): key 0 past the end, rank + 1 else.
#include "k1_bwt.h"
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include "devutil.h"

#define K1D_T 1024u                       // entries a workgroup owns per step
#define K1D_WIN 256u                      // entries / positions behind its own a workgroup looks at (the tail of its last group)
#define K1D_N (K1D_T + K1D_WIN)           // ... and looks at
#define K1D_RPT (K1D_N / 256u)            // slots per thread
#define K1D_FW (K1D_N / 32u)              // words of a slot bitmap
#define K1D_BW (K1D_N / 32u + 2u)         // head-bitmap words a build window looks at
#define K1D_INF 0x3FFFFFFF
static_assert(K1D_N % 256u == 0 && K1D_BW <= 64u && K1D_GS <= K1D_WIN && K1D_WIN <= 256u && K1_HT == K1D_T, "window geometry");

#define K1D_DESC(b, start, len) (((u64)(b) << 52) | ((u64)(start) << 26) | (u64)(len))
#define K1D_DB(d) ((u32)((d) >> 52))
#define K1D_DSTART(d) ((u32)(((d) >> 26) & 0x3FFFFFFu))
#define K1D_DLEN(d) ((u32)((d) & 0x3FFFFFFu))
// chunk of a big group's range: block, first position, positions (<= 1024), positions of the same group behind it (<= K1D_GS looked at)
#define K1D_CHUNK(b, cs, cl, ahead) (((u64)(b) << 42) | ((u64)(cs) << 20) | ((u64)(cl) << 9) | (u64)(ahead))
#define K1D_CB(d) ((u32)((d) >> 42) & 0xFFFu)
#define K1D_CS(d) ((u32)(((d) >> 20) & 0x3FFFFFu))
#define K1D_CL(d) ((u32)(((d) >> 9) & 0x7FFu))
#define K1D_CA(d) ((u32)((d) & 0x1FFu))
#define K1D_RKEEP 0xFFFFFFFFu             // R[p] of a position whose rank did not change

// The entry lists of the blocks as ONE sequence of tiles (a block's list is ceil(entries / K1D_T) tiles): the lists of a batch
// differ by an order of magnitude (HTML next to plain text), and with a fixed number of workgroups per block the longest list
// set the kernel's time (round 4 of E8S-A: 58 ps per entry against 21 in round 0).  With 16 blocks or more the sequence is per
// XCD: the workgroups the dispatcher places on XCD x (blockIdx.x & 7, as in xcd_block_tile) walk the lists of the blocks
// x, x + 8, ..., so a block's ranks are gathered through ONE L2 (round 0 of E8S-A: 0.97 ms, one sequence over all XCDs: 1.23).
#define K1D_NBW 128u
struct K1dSeq { u32 first, step, slots, wg, nwg; };        // slot i = block first + step * i; this workgroup is wg of nwg
__device__ __forceinline__ K1dSeq k1d_seq(u32 nb, u32 b0) {
    K1dSeq q;
    if (nb >= 16u && (gridDim.x & 7u) == 0u) {
        q.step = 8u; q.first = b0 + (blockIdx.x & 7u); q.wg = blockIdx.x >> 3; q.nwg = gridDim.x >> 3;
    } else {
        q.step = 1u; q.first = b0; q.wg = blockIdx.x; q.nwg = gridDim.x;
    }
    const u32 left = q.first < nb ? (nb - q.first + q.step - 1u) / q.step : 0u;
    q.slots = left < K1D_NBW ? left : K1D_NBW;
    return q;
}
// tpre[i] = tiles of the slots before slot i; returns their sum.  Every thread of the workgroup must call it.
__device__ __forceinline__ u32 k1d_tile_prefix(const K1Buf& B, const u32* cnt_row, const K1dSeq& q, u32 stride, u32* tpre, u32* scr) {
    const u32 tid = threadIdx.x;
    u32 carry = 0;
    for (u32 c0 = 0; c0 < q.slots; c0 += 256u) {
        u32 c = 0;
        if (c0 + tid < q.slots) { c = cnt_row[K1_BI(B, q.first + q.step * (c0 + tid))]; if (c > stride) c = stride; c = (c + K1D_T - 1u) / K1D_T; }
        const u32 ex = block_excl_scan_256(c, scr);
        if (c0 + tid < q.slots) tpre[c0 + tid] = carry + ex;
        __syncthreads();
        if (tid == 255u) scr[0] = carry + ex + c;
        __syncthreads();
        carry = scr[0];
        __syncthreads();
    }
    if (tid == 0) tpre[q.slots] = carry;
    __syncthreads();
    return carry;
}
// the slot that holds flat tile f < tpre[slots]: the last i with tpre[i] <= f
__device__ __forceinline__ u32 k1d_tile_block(const u32* tpre, u32 nbw, u32 f) {
    u32 lo = 0, hi = nbw;                 // invariant: tpre[lo] <= f < tpre[hi]
    while (hi - lo > 1u) {
        const u32 mid = (lo + hi) >> 1;
        if (tpre[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

// Sort key of rotation / suffix s in a doubling round (h < 2^32 * 2^K1D_MAXR: 64 bits).
__device__ __forceinline__ u32 k1d_key(const u32* ISA, u32 n, u32 s, u64 h, u32 hm, u32 mode, u32 linear) {
    if (mode) return n - 1u - s;
    if (linear) {
        const u64 x = (u64)s + h;
        return x >= n ? 0u : ISA[x] + 1u;
    }
    u32 x = s + hm;                     // hm = h mod n
    if (x >= n) x -= n;
    return ISA[x];
}

// 1: this round of block b is the tie-break (descending start index).  Either h >= n for every block of the batch, or the
// previous round split no group of the block: with P the partition into groups (it refines "equal h-prefix") a round maps it
// to P' = {s ~ s' in P and s+h ~ s'+h in P}; P' = P gives s+kh ~ s'+kh for all k, members of a group agree on h bytes, so s and
// s' agree everywhere - identical rotations (linear mode: suffixes that ran into the padding together), which no further
// doubling round can tell apart.  Periodic and tiled inputs reach that state after a few rounds instead of log2(n / h0).
// The tie-break itself needs no sort in cyclic mode: a group of identical rotations is the COMPLETE class of one rotation
// (identical rotations have equal keys in every round, so nothing ever separates them), i.e. all L = n / p rotations
// s, s + p, s + 2p, ... of a block with period p, and descending start index puts rotation s at position
// start + L - 1 - s / p.  (Rounds 1-3 sorted the keys n - 1 - s: for `periodic ab` two groups of 450 000 per block, one
// workgroup each, 20 ms.)  Linear mode (suffixes that ran into the padding together) sorts.
__device__ __forceinline__ bool k1d_tie_pos(u32 n, u32 start, u32 L, u32 s, u32& pos) {
    const u32 p = n / L;
    pos = start + L - 1u - s / p;
    return p * L == n && s / p < L;        // (always; a group that is no such class falls back to the sort)
}
__device__ __forceinline__ u32 k1d_mode(const K1Buf& B, u32 r, u32 b, u32 final_h) {
    return (final_h || (r > 0u && B.dchg[(size_t)(r - 1u) * B.rstride + K1_BI(B, b)] == 0u)) ? 1u : 0u;
}

// Barrier of the kernels below that exchange data through LDS only (k1d_window, k1d_med, k1d_update's list walk): __syncthreads()
// also waits for the workgroup's outstanding global stores - the scattered rank stores these kernels are made of - at every barrier.
#define K1D_SYNC() lds_barrier()

// descriptor of a group of more than K1D_GS rotations into round r's list of its size class (one atomic each: they are rare)
#define K1D_MED1 1024u
__device__ __forceinline__ void k1d_push_big(const K1Buf& B, u32 r, u32 b, u32 start, u32 len) {
    const u32 cls = len <= K1D_MED1 ? 3u : (len <= K1_MED_MAX ? 0u : 1u);
    const u32 idx = atomicAdd(&B.dbn[r * 4u + cls], 1u);
    u64* list = cls == 3u ? B.listT[r & 1u] : (cls == 0u ? B.listM[r & 1u] : B.listL[r & 1u]);
    const u32 cap = cls == 3u ? B.listTCap : (cls == 0u ? B.listMCap : B.listLCap);
    if (idx < cap) list[idx] = K1D_DESC(b, start, len);
}

// ---------------------------------------------------------------------------------------------
// One window of suffix-array positions [lo, lo + cl) of block b (cl <= 1024; `ahead` <= K1D_GS positions behind it are looked
// at for the tail of a group that starts inside): ISA for the window's positions, and its groups into round r's lists.
//   BITMAP  (k1d_build) heads and group lengths come from the head bitmap HN (lo is a multiple of 1024);
//   else    (k1d_update, chunks of a big group's range) from R (= SB: head position of every position) and KB (length at a head).
// Every thread of the workgroup must call it.
// ---------------------------------------------------------------------------------------------
template <bool BITMAP>
__device__ __forceinline__ void k1d_window(const K1Buf& B, const BatchGeom& g, u32 b, u32 lo, u32 cl, u32 ahead, u32 r) {
    __shared__ u32 hw[K1D_BW];
    __shared__ int wh[K1D_N];             // head of every slot, relative to lo (negative: before the window)
    __shared__ u32 wl[K1D_T];             // group length at the head slots
    __shared__ u32 fm[K1D_FW], pre[K1D_FW];
    __shared__ int prevh[64], nexth[64];
    __shared__ int inHead;
    __shared__ u32 sbase;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32* SA = B.SA + (size_t)b * g.stride;
    u32* ISA = B.ISA + (size_t)b * g.stride;
    const u32* HN = B.HN + (size_t)b * g.hstride;
    const u32 span = cl + ahead;
    K1D_SYNC();                                        // (the LDS of the previous window of this workgroup)
    u32 sv[K1D_RPT];
#pragma unroll
    for (u32 it = 0; it < K1D_RPT; it++) {
        const u32 j = it * 256u + tid;
        sv[it] = j < span ? SA[lo + j] : 0u;
    }
    if (tid < K1D_FW) fm[tid] = 0;
    if (tid == 0) inHead = (int)lo;
    if (BITMAP) {
        if (tid < K1D_BW) hw[tid] = HN[(lo >> 5) + tid];
        K1D_SYNC();
        if (w == 0) {
            const u32 word = lane < K1D_BW ? hw[lane] : 0u;
            int v = word ? (int)(lane * 32u + 31u - (u32)__clz((int)word)) : -1;
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(v, (unsigned)off);
                if ((int)lane >= off) v = v > u ? v : u;
            }
            int ex = __shfl_up(v, 1u);
            if (lane == 0) ex = -1;
            prevh[lane] = ex;                               // last head in the words before this one
            int f = word ? (int)(lane * 32u + (u32)__ffs((int)word) - 1u) : K1D_INF;
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_down(f, (unsigned)off);
                if ((int)lane + off < 64) f = f < u ? f : u;
            }
            int nx = __shfl_down(f, 1u);
            if (lane == 63u) nx = K1D_INF;
            nexth[lane] = nx;                               // first head in the words after this one
            if (!(hw[0] & 1u)) {                            // the window starts inside a group: its head, from the global bitmap
                int found = -1;
                for (int iter = 0; found < 0; iter++) {
                    const int wi = (int)(lo >> 5) - 1 - (int)lane - 64 * iter;
                    const u32 wd = wi >= 0 ? HN[wi] : 0u;
                    const u64 bal = __ballot(wd != 0u);
                    if (bal) {
                        const int src = __ffsll((long long)bal) - 1;
                        const int pos = wi * 32 + 31 - __clz((int)wd);
                        found = __shfl(pos, src);
                    } else if (wi < 0) found = 0;           // (cannot happen: position 0 is a head)
                }
                if (lane == 0) inHead = found;
            }
        }
        K1D_SYNC();
#pragma unroll
        for (u32 it = 0; it < K1D_RPT; it++) {
            const u32 j = it * 256u + tid;
            const u32 wq = j >> 5, bq = j & 31u;
            const u32 word = hw[wq];
            const u32 low = word & (0xFFFFFFFFu >> (31u - bq));
            const int ph = prevh[wq];
            wh[j] = low ? (int)(wq * 32u + 31u - (u32)__clz((int)low)) : (ph >= 0 ? ph : inHead - (int)lo);   // (negative: the head before the window)
            // length of the group that starts here: the next head - in the window's words, else in the global bitmap
            const bool head = j < cl && ((word >> bq) & 1u);
            int nx = K1D_INF;
            if (head) {
                const u32 high = bq == 31u ? 0u : (word & (0xFFFFFFFEu << bq));
                nx = high ? (int)(wq * 32u + (u32)__ffs((int)high) - 1u) : nexth[wq];
            }
            u64 far = __ballot(head && nx == K1D_INF);
            while (far) {                                   // rare: a group that reaches beyond the window's words
                const int src = __ffsll((long long)far) - 1;
                far &= far - 1;
                const u32 w0 = (lo >> 5) + K1D_BW;
                u32 endg = 0;
                bool got = false;
                for (u32 it2 = 0; !got; it2++) {
                    const u32 gw = w0 + lane + 64u * it2;
                    const u32 wd = gw < g.hstride ? HN[gw] : 0xFFFFFFFFu;
                    const u64 bal = __ballot(wd != 0u);
                    if (bal) {
                        const int fl = __ffsll((long long)bal) - 1;
                        endg = __shfl(gw * 32u + (u32)__ffs((int)wd) - 1u, fl);
                        got = true;
                    }
                }
                if ((int)lane == src) nx = (int)(endg - lo);
            }
            if (head) wl[j] = (u32)nx - j;
        }
    } else {
        const u32* R = B.SB + (size_t)b * g.stride;
        const u32* SL = B.KB + (size_t)b * g.stride;
#pragma unroll
        for (u32 it = 0; it < K1D_RPT; it++) {
            const u32 j = it * 256u + tid;
            int hp = -K1D_INF;
            if (j < span) hp = (int)R[lo + j] - (int)lo;
            wh[j] = hp;
            if (j < cl && hp == (int)j) wl[j] = SL[lo + j];
        }
    }
    K1D_SYNC();
    // ranks of the window's own positions; members of the groups that start inside it
    u32 mlen[K1D_RPT];
    int mhp[K1D_RPT];
#pragma unroll
    for (u32 it = 0; it < K1D_RPT; it++) {
        const u32 j = it * 256u + tid;
        mlen[it] = 0;
        mhp[it] = 0;
        if (j < span) {
            const int hp = wh[j];
            if (j < cl) ISA[sv[it]] = (u32)((int)lo + hp);
            if (hp >= 0 && hp < (int)cl) {
                const u32 len = wl[hp];
                if (len >= 2u && len <= K1D_GS) {
                    mlen[it] = len;
                    mhp[it] = hp;
                    atomicOr(&fm[j >> 5], 1u << (j & 31u));
                } else if (len > K1D_GS && hp == (int)j) {
                    k1d_push_big(B, r, b, lo + j, len);      // a big group: a descriptor for k1d_med / k1d_large
                }
            }
        }
    }
    K1D_SYNC();
    if (w == 0) {
        const u32 c = lane < K1D_FW ? (u32)__popc(fm[lane]) : 0u;
        const u32 inc = wave_incl_scan_u32(c);
        if (lane < K1D_FW) pre[lane] = inc - c;
        const u32 total = (u32)__shfl((int)inc, 63);
        if (lane == 0) sbase = total ? atomicAdd(&B.dcnt[(size_t)r * B.rstride + K1_BI(B, b)], total) : 0u;
    }
    K1D_SYNC();
    u64* L = B.rlist[0] + (size_t)b * g.stride;
#pragma unroll
    for (u32 it = 0; it < K1D_RPT; it++) {
        const u32 j = it * 256u + tid;
        if (mlen[it]) {
            const u32 idx = sbase + pre[j >> 5] + (u32)__popc(fm[j >> 5] & ((1u << (j & 31u)) - 1u));
            if (idx < g.stride) L[idx] = K1E_MAKE(mlen[it] - 1u, j - (u32)mhp[it], sv[it], lo + j);
        }
    }
}

// ISA for every rotation of the blocks that still hold groups (dtot[b] != 0, k1_count_unsorted), and round 0's lists
__global__ __launch_bounds__(256) void k1d_build(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y, n = B.nfront[b];
    const u32 lo = blockIdx.x * K1D_T;
    if (lo >= n || (B.dtot[K1_BI(B, b)] == 0u && !B.linear)) return;          // (linear mode: k1_finish_linear reads the rank of suffix 0)
    k1d_window<true>(B, g, b, lo, n - lo < K1D_T ? n - lo : K1D_T, K1D_WIN, 0u);
}

// ---------------------------------------------------------------------------------------------
// the entry lists of round r: keys, ranks inside the groups, new entries (same slots, new order) into rlist[1]
// ---------------------------------------------------------------------------------------------
// blocks below this size (every bzip2 block): k1d_round ranks on (key << 11 | slot) in one word - keys (ranks, + 1 in linear mode) < 2^21 - 1
#ifndef K1D_PACK_MAXN
#define K1D_PACK_MAXN ((1u << 21) - 2u)
#endif
static_assert(K1D_N <= 2048u, "11 slot bits");
#ifndef K1D_MINW
#define K1D_MINW 4
#endif
__global__ __launch_bounds__(256, K1D_MINW) void k1d_round(K1Buf B, BatchGeom g, u32 r, u64 h, u32 final_h) {
    __shared__ __attribute__((aligned(16))) u32 kk[K1D_N + 8];
    __shared__ u32 tpre[K1D_NBW + 1], scr[256];
    __shared__ u32 s_cnt[K1D_NBW], s_n[K1D_NBW], s_hm[K1D_NBW], s_mode[K1D_NBW];     // per slot (block): what a tile needs of its block
    // Round 6, blocks that k1_period.hip reduced to three periods (red[b] = p): two rotations of the SAME phase (s mod p) never need a comparison - they agree
    // until the later one wraps, and then ONE sign decides for the whole block (k1_period.hip: W_r0 against W_0, K1P_RASC).  A tiled input keeps every
    // rotation in such groups (three copies of everything) for log2(p) rounds; a group whose members all share a phase is settled by index order on the spot.
    __shared__ u32 s_per[K1D_NBW], s_asc[K1D_NBW];
    __shared__ u32 ks[K1D_N + 8], kp[K1D_N + 8];                                      // rotation / phase of every owned cell (periodic blocks only)
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32 linear = B.linear;
    const u32* cnt_row = B.dcnt + (size_t)r * B.rstride;
    for (u32 b0 = 0; b0 < g.nb; b0 += 8u * K1D_NBW) {
        const K1dSeq q = k1d_seq(g.nb, b0);
        const u32 G = q.nwg;
        const u32 ntile = k1d_tile_prefix(B, cnt_row, q, g.stride, tpre, scr);
        if (q.wg >= ntile) continue;                        // (uniform)
        if (tid < q.slots) {
            const u32 bb = q.first + q.step * tid, nn = B.nfront[bb];
            s_cnt[tid] = cnt_row[K1_BI(B, bb)] < g.stride ? cnt_row[K1_BI(B, bb)] : g.stride;
            s_n[tid] = nn;
            s_hm[tid] = nn ? (h >> 32 ? (u32)(h % nn) : (u32)h % nn) : 0u;
            s_mode[tid] = k1d_mode(B, r, bb, final_h);
            s_per[tid] = linear ? 0u : B.red[bb];
            s_asc[tid] = B.ptab[(size_t)bb * 256u + 202u];          // K1P_RASC (k1_period.hip)
        }
        __syncthreads();
        // a tile of the flat sequence: block, first entry, entries of the block's list
        struct Tile { u32 b, e0, cnt, n, hm, mode, per, asc; };
        auto tile_of = [&](u32 f) {
            Tile t;
            t.b = 0; t.e0 = 0; t.cnt = 0; t.n = 1; t.hm = 0; t.mode = 0; t.per = 0; t.asc = 0;
            if (f < ntile) {
                const u32 i = k1d_tile_block(tpre, q.slots, f);
                t.b = q.first + q.step * i;
                t.e0 = (f - tpre[i]) * K1D_T;
                t.cnt = s_cnt[i];
                t.n = s_n[i];
                t.hm = s_hm[i];
                t.mode = s_mode[i];
                t.per = s_per[i];
                t.asc = s_asc[i];
            }
            return t;
        };
        auto load_tile = [&](const Tile& t, u64 (&e)[K1D_RPT]) {
            const u64* Lin = B.rlist[0] + (size_t)t.b * g.stride;
#pragma unroll
            for (u32 it = 0; it < K1D_RPT; it++) {
                const u32 i = (it * 4u + w) * 64u + lane;
                e[it] = (t.e0 < t.cnt && i < t.cnt - t.e0) ? Lin[t.e0 + i] : 0ull;      // beyond the list: a group of one, never owned
            }
        };
        // owned: the group starts inside the tile's first K1D_T entries
        auto own = [&](u64 e, u32 i) { const u32 gs = i - K1E_IDX(e); return K1E_LEN(e) >= 2u && gs < K1D_T; };   // (i < idx wraps to a huge gs)
        auto gather = [&](const Tile& t, const u64 (&e)[K1D_RPT], u32 (&k)[K1D_RPT]) {
            const u32* ISA = B.ISA + (size_t)t.b * g.stride;
#pragma unroll
            for (u32 it = 0; it < K1D_RPT; it++) {
                const u32 i = (it * 4u + w) * 64u + lane;
                k[it] = 0;
                if (own(e[it], i)) k[it] = k1d_key(ISA, t.n, K1E_S(e[it]), h, t.hm, t.mode, linear);
            }
        };
        u64 eC[K1D_RPT], eN[K1D_RPT], eNN[K1D_RPT];
        u32 kC[K1D_RPT];
        u32 f = q.wg;
        Tile tC = tile_of(f), tN = tile_of(f + G), tNN;
        load_tile(tC, eC);
        load_tile(tN, eN);
        gather(tC, eC, kC);
        for (; f < ntile; f += G) {
            tNN = tile_of(f + 2u * G);
            load_tile(tNN, eNN);
            const bool pack = tC.n < K1D_PACK_MAXN;         // (uniform) keys and slots fit one word
#pragma unroll
            for (u32 it = 0; it < K1D_RPT; it++) {
                const u32 i = (it * 4u + w) * 64u + lane;
                if (own(eC[it], i)) {
                    kk[i] = pack ? (kC[it] << 11) | i : kC[it];
                    if (tC.per) { const u32 sx_ = K1E_S(eC[it]); ks[i] = sx_; kp[i] = sx_ % tC.per; }     // (uniform per tile)
                }
            }
            lds_barrier();
            u32 kN[K1D_RPT];
            gather(tN, eN, kN);                             // in flight while this tile is ranked
            u64* Lt = B.rlist[1] + (size_t)tC.b * g.stride;
            u32* SA = B.SA + (size_t)tC.b * g.stride;
            u32 changed = 0;
            const bool tie = tC.mode != 0u && !linear;      // (uniform) the tie-break of a cyclic block: positions by formula
#pragma unroll
            for (u32 it = 0; it < K1D_RPT; it++) {
                const u32 i = (it * 4u + w) * 64u + lane;
                u32 tpos = 0;
                if (tie && own(eC[it], i) && k1d_tie_pos(tC.n, K1E_POS(eC[it]) - K1E_IDX(eC[it]), K1E_LEN(eC[it]), K1E_S(eC[it]), tpos)) {
                    SA[tpos] = K1E_S(eC[it]);
                    Lt[tC.e0 + i] = K1E_MAKE(0u, 0u, K1E_S(eC[it]), tpos) | K1E_KEEP;      // final; k1d_update has nothing to do for it
                } else if (own(eC[it], i)) {
                    const u32 gs = i - K1E_IDX(eC[it]), ge = gs + K1E_LEN(eC[it]);
                    if (tC.per) {
                        // a group of ONE phase: index order with the block's sign, every member final
                        const u32 sx_ = K1E_S(eC[it]), myph = kp[i];
                        u32 diff = 0, lt = 0;
                        for (u32 j = gs; j < ge; j++) { diff |= kp[j] ^ myph; lt += ks[j] < sx_ ? 1u : 0u; }
                        if (diff == 0u) {
                            const u32 q = gs + (tC.asc ? lt : ge - gs - 1u - lt);
                            const u32 pos = K1E_POS(eC[it]) + q - i;
                            const bool keep = pos == K1E_POS(eC[it]) - K1E_IDX(eC[it]);
                            Lt[tC.e0 + q] = K1E_MAKE(0u, 0u, sx_, pos) | (keep ? K1E_KEEP : 0ull);
                            SA[pos] = sx_;
                            changed |= 1u;
                            continue;
                        }
                    }
                    const u32 m = kC[it];
                    u32 less = 0, eqb = 0, eqt = 0;
                    if (pack) {
                        // the cells hold (key << 11 | slot): "smaller key, or equal key in an earlier slot" is ONE unsigned compare, and
                        // the three counts (below my key, below me, below my key + 1) cost two instructions per candidate each
                        // (v_cmp + v_addc) - the kernel was bound by the VALU (PMC: 557 instructions per entry with six conditions per
                        // candidate; groups of 65..256 rotations are half of them).  Cells outside the group (the first and the last
                        // four may hold some) are replaced by ~0, which is below nothing.
                        const u32 tlo = m << 11, tme = tlo | i, thi = tlo + 2048u;
                        u32 nlo = 0, nme = 0, nhi = 0;
                        auto quad = [&](u32 j, bool edge) {
                            const uint4 c4 = *(const uint4*)&kk[j];
                            u32 c[4] = {c4.x, c4.y, c4.z, c4.w};
                            if (edge) {
#pragma unroll
                                for (u32 u = 0; u < 4u; u++) c[u] = (j + u >= gs && j + u < ge) ? c[u] : 0xFFFFFFFFu;
                            }
#pragma unroll
                            for (u32 u = 0; u < 4u; u++) {
                                nlo += c[u] < tlo ? 1u : 0u;
                                nme += c[u] < tme ? 1u : 0u;
                                nhi += c[u] < thi ? 1u : 0u;
                            }
                        };
                        u32 j = gs & ~3u;
                        quad(j, true);
                        for (j += 4u; j + 4u <= ge; j += 4u) quad(j, false);
                        if (j < ge) quad(j, true);
                        less = nlo; eqb = nme - nlo; eqt = nhi - nlo;
                    } else {
                        for (u32 j = gs & ~3u; j < ge; j += 4u) {
                            const uint4 c4 = *(const uint4*)&kk[j];
                            const u32 c[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                            for (u32 u = 0; u < 4u; u++) {
                                const bool in = j + u >= gs && j + u < ge;
                                less += (in && c[u] < m) ? 1u : 0u;
                                eqb += (in && c[u] == m && j + u < i) ? 1u : 0u;
                                eqt += (in && c[u] == m) ? 1u : 0u;
                            }
                        }
                    }
                    const u32 q = gs + less + eqb;          // new slot; positions inside a group are consecutive
                    const u32 s = K1E_S(eC[it]), pos = K1E_POS(eC[it]) + q - i;
                    const bool keep = pos - eqb == K1E_POS(eC[it]) - K1E_IDX(eC[it]);     // still under the old head: its rank stands
                    Lt[tC.e0 + q] = K1E_MAKE(eqt - 1u, eqb, s, pos) | (keep ? K1E_KEEP : 0ull);
                    if (eqt == 1u) SA[pos] = s;
                    changed |= eqt != K1E_LEN(eC[it]) ? 1u : 0u;
                }
            }
            if (__ballot(changed != 0u) && lane == 0) atomicOr(&B.dchg[(size_t)r * B.rstride + K1_BI(B, tC.b)], 1u);
            lds_barrier();
#pragma unroll
            for (u32 it = 0; it < K1D_RPT; it++) { eC[it] = eN[it]; eN[it] = eNN[it]; kC[it] = kN[it]; }
            tC = tN; tN = tNN;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// big groups: results for k1d_update are R (= SB: head position of every position of the range) and KB (length at the heads)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void k1d_push_chunks(const K1Buf& B, u32 r, u32 b, u32 start, u32 len) {
    __shared__ u32 cbase;
    const u32 nch = (len + K1D_T - 1u) / K1D_T;
    __syncthreads();
    if (threadIdx.x == 0) cbase = atomicAdd(&B.dbn[r * 4u + 2u], nch);
    __syncthreads();
    for (u32 k = threadIdx.x; k < nch; k += blockDim.x) {
        const u32 cs = k * K1D_T, cl = len - cs < K1D_T ? len - cs : K1D_T;
        const u32 rest = len - cs - cl;
        if (cbase + k < B.listSCap) B.listS[0][cbase + k] = K1D_CHUNK(b, start + cs, cl, rest < K1D_WIN ? rest : K1D_WIN);
    }
}

__device__ __forceinline__ void k1d_cmpx(u32* ck, u32* cv, u32 lo, u32 hi) {
    const u32 a = ck[lo], c = ck[hi];
    if (a > c) {
        ck[lo] = c; ck[hi] = a;
        const u32 t = cv[lo]; cv[lo] = cv[hi]; cv[hi] = t;
    }
}

// groups of K1D_GS+1 .. K1_MED_MAX rotations: one workgroup each, persistent grid.  Sorted in LDS:
//   blocks below 2^20 bytes (every bzip2 block): (key << 12 | index in the group) in ONE word, three stable 7-bit LSD passes
//   (per-wave digit counts and ranks from wave ballots: equal digits - most keys of an early round are equal - cost one LDS
//   atomic per wave and row, not one per element); the first version, a bitonic network on (key, rotation) pairs, took 66-78
//   barrier-separated stages for the typical group of 1000-2000 rotations: 64 us per group, 3.0 of the 15.4 ms of an E8S-A step;
//   larger blocks (BWT.* entry points only): that bitonic network.
// Then, from the sorted group in LDS: suffix array, ranks (R = SB, for k1d_update's copy), and the sub-groups straight into
// the next round's lists (entries: one atomic per group reserves their slots; descriptors for what is still big).
#ifndef K1D_RADIX_MAXN
#define K1D_RADIX_MAXN (1u << 20)
#endif
// CAP = 1024 (groups up to K1D_MED1: 13 KB of LDS, eight workgroups per CU - a group is a chain of dependent memory round
// trips and barrier-separated steps, 47 us alone, and what hides that is other groups) or K1_MED_MAX (49 KB, three per CU).
static_assert(K1_MED_MAX == 4096, "12 index bits next to 20 key bits");
template <u32 CAP>
__global__ __launch_bounds__(256) void k1d_med(K1Buf B, BatchGeom g, u32 r, u64 h, u32 final_h) {
    static_assert(CAP == 1024u || CAP == 4096u, "CAP / 256 positions per thread, at most 32");
    __shared__ u32 ck[CAP], cv[CAP], cx[CAP];
    __shared__ u32 wh[4][128], dsum[128];
    __shared__ u32 sbase;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u64 lt = lanemask_lt();
    const u64* list = CAP == K1D_MED1 ? B.listT[r & 1u] : B.listM[r & 1u];
    const u32 lcap = CAP == K1D_MED1 ? B.listTCap : B.listMCap;
    u32 cnt = B.dbn[r * 4u + (CAP == K1D_MED1 ? 3u : 0u)];
    if (cnt > lcap) cnt = lcap;
    for (u32 gi = blockIdx.x; gi < cnt; gi += gridDim.x) {
        const u64 d = list[gi];
        const u32 b = K1D_DB(d), start = K1D_DSTART(d), len = K1D_DLEN(d);
        const u32 n = B.nfront[b];
        const u32 mode = k1d_mode(B, r, b, final_h);
        const u32 hm = (u32)(h % n);
        const u32* ISA = B.ISA + (size_t)b * g.stride;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u32* R = B.SB + (size_t)b * g.stride + start;
        const bool radix = n < K1D_RADIX_MAXN;
        K1D_SYNC();
        if (mode && !B.linear && (n / len) * len == n) {    // (uniform) the tie-break of a cyclic block: positions by formula
            const u32 pp = n / len;
            for (u32 i = tid; i < len; i += 256) cv[i] = SA[i];
            K1D_SYNC();
            for (u32 i = tid; i < len; i += 256) {
                SA[len - 1u - cv[i] / pp] = cv[i];
                R[i] = K1D_RKEEP;
            }
            continue;
        }
        for (u32 i = tid; i < len; i += 256) {
            const u32 s = SA[i];
            cv[i] = s;
            const u32 k = k1d_key(ISA, n, s, h, hm, mode, B.linear);
            ck[i] = radix ? (k << 12) | i : k;
        }
        K1D_SYNC();
        u32* sorted = ck;                                   // radix: the packed words in order; bitonic: keys in ck, rotations in cv
        if (radix) {
            const u32 chunk = (((len + 3u) / 4u) + 63u) & ~63u;
            const u32 lo = w * chunk < len ? w * chunk : len;
            const u32 hi = lo + chunk < len ? lo + chunk : len;
            u32* src = ck;
            u32* dst = cx;
            for (u32 shift = 12; shift < 32u; shift += 7u) {
                for (u32 i = tid; i < 512u; i += 256) (&wh[0][0])[i] = 0;
                K1D_SYNC();
                for (u32 i0 = lo; i0 < hi; i0 += 64u) {
                    const u32 i = i0 + lane;
                    const bool valid = i < hi;
                    const u32 dg = valid ? (src[i] >> shift) & 127u : 0u;
                    const u64 m = match_any(dg, 7, valid);
                    if (valid && (m & lt) == 0ull) atomicAdd(&wh[w][dg], (u32)__popcll(m));
                }
                K1D_SYNC();
                if (tid < 128u) {
                    u32 run = 0;
#pragma unroll
                    for (u32 ww = 0; ww < 4u; ww++) { const u32 c = wh[ww][tid]; wh[ww][tid] = run; run += c; }
                    dsum[tid] = run;
                }
                K1D_SYNC();
                if (w == 0) {                                // exclusive scan of the 128 digit totals: two per lane
                    const u32 a0 = dsum[2u * lane], a1 = dsum[2u * lane + 1u];
                    const u32 inc = wave_incl_scan_u32(a0 + a1);
                    dsum[2u * lane] = inc - a0 - a1;
                    dsum[2u * lane + 1u] = inc - a1;
                }
                K1D_SYNC();
                for (u32 e = tid; e < 512u; e += 256) wh[e >> 7][e & 127u] += dsum[e & 127u];
                K1D_SYNC();
                for (u32 i0 = lo; i0 < hi; i0 += 64u) {
                    const u32 i = i0 + lane;
                    const bool valid = i < hi;
                    const u32 x = valid ? src[i] : 0u;
                    const u32 dg = (x >> shift) & 127u;
                    const u64 m = match_any(dg, 7, valid);
                    const u32 rank = (u32)__popcll(m & lt);
                    const u32 base = valid ? wh[w][dg] : 0u;
                    __builtin_amdgcn_wave_barrier();
                    if (valid && rank == 0) wh[w][dg] = base + (u32)__popcll(m);
                    __builtin_amdgcn_wave_barrier();
                    if (valid) dst[base + rank] = x;
                }
                K1D_SYNC();
                u32* t = src; src = dst; dst = t;
            }
            sorted = src;
        } else {
            u32 M = 128;
            while (M < len) M <<= 1;
            for (u32 k = 2; k <= M; k <<= 1) {
                const u32 hk = k >> 1;
                for (u32 i = tid; i < (M >> 1); i += 256) {
                    const u32 blk = i / hk, off = i - blk * hk;
                    const u32 lo = blk * k + off, hi = blk * k + (k - 1u - off);
                    if (hi < len) k1d_cmpx(ck, cv, lo, hi);
                }
                K1D_SYNC();
                for (u32 j = k >> 2; j > 0; j >>= 1) {
                    for (u32 i = tid; i < (M >> 1); i += 256) {
                        const u32 lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u));
                        const u32 hi = lo | j;
                        if (hi < len) k1d_cmpx(ck, cv, lo, hi);
                    }
                    K1D_SYNC();
                }
            }
        }
        // Heads, ranks and sub-group lengths from the sorted group.  Every thread takes CH consecutive positions; the last head at or
        // before a position and the first head behind it come from one max-scan and one min-scan over the threads' ranges (the first
        // version walked a head bitmap word by word for every position - in the early rounds, where most keys of a group are equal,
        // a dependent chain of up to len / 32 LDS reads per position, twice: half of the kernel's time).
        {
            constexpr u32 CH = CAP / 256u;
            const u32 ksh = radix ? 12u : 0u;
            const u32 p0 = tid * CH;
            u32 hmask = 0;                                  // heads among my positions
            u32 sv[CH];
            {
                u32 kprev = (p0 > 0u && p0 <= len) ? sorted[p0 - 1u] >> ksh : 0u;
#pragma unroll
                for (u32 c = 0; c < CH; c++) {
                    const u32 i = p0 + c;
                    sv[c] = 0;
                    if (i < len) {
                        const u32 x = sorted[i], k = x >> ksh;
                        if (i == 0u || k != kprev) hmask |= 1u << c;
                        kprev = k;
                        sv[c] = radix ? cv[x & 4095u] : cv[i];
                    }
                }
            }
            // last head (+1; 0: none) at or before the end of my range, first head (len: none) at or behind its start
            u32 lastp = hmask ? p0 + 32u - (u32)__clz((int)hmask) : 0u;
            u32 firstp = hmask ? p0 + (u32)__ffs((int)hmask) - 1u : len;
            u32 vmax = lastp, vmin = firstp;
            for (u32 off = 1; off < 64u; off <<= 1) {
                const u32 a = __shfl_up(vmax, off), c2 = __shfl_down(vmin, off);
                if (lane >= off && a > vmax) vmax = a;
                if (lane + off < 64u && c2 < vmin) vmin = c2;
            }
            K1D_SYNC();                                // (wh is free: the passes are done)
            if (lane == 63u) wh[0][w] = vmax;
            if (lane == 0u) wh[0][8u + w] = vmin;
            K1D_SYNC();
            u32 before = __shfl_up(vmax, 1u), behind = __shfl_down(vmin, 1u);      // exclusive, inside the wave
            if (lane == 0u) before = 0u;
            if (lane == 63u) behind = len;
            for (u32 ww = 0; ww < w; ww++) before = wh[0][ww] > before ? wh[0][ww] : before;
            for (u32 ww = w + 1u; ww < 4u; ww++) behind = wh[0][8u + ww] < behind ? wh[0][8u + ww] : behind;
            // per position: head hp (scan from the left), next head nh (from the right)
            u32 hpv[CH], slv[CH];
            {
                u32 run = before;                           // (+1 encoding)
#pragma unroll
                for (u32 c = 0; c < CH; c++) { if ((hmask >> c) & 1u) run = p0 + c + 1u; hpv[c] = run - 1u; }
                u32 nx = behind;
#pragma unroll
                for (u32 cc = 0; cc < CH; cc++) {
                    const u32 c = CH - 1u - cc;
                    slv[c] = nx - hpv[c];                   // length of the sub-group position p0 + c belongs to
                    if ((hmask >> c) & 1u) nx = p0 + c;
                }
            }
            // suffix array, ranks; members of sub-groups of 2..K1D_GS become entries (counted here, placed below)
            bool split = false;
            u32 mine = 0;
#pragma unroll
            for (u32 c = 0; c < CH; c++) {
                const u32 i = p0 + c;
                if (i < len) {
                    SA[i] = sv[c];
                    R[i] = hpv[c] ? start + hpv[c] : K1D_RKEEP;        // (the first sub-group keeps the group's rank)
                    split = split || hpv[c] > 0u;
                    if (slv[c] >= 2u && slv[c] <= K1D_GS) mine++;
                    else if (slv[c] > K1D_GS && hpv[c] == i) k1d_push_big(B, r + 1u, b, start + i, slv[c]);
                }
            }
            if (__ballot(split) && lane == 0u) atomicOr(&B.dchg[(size_t)r * B.rstride + K1_BI(B, b)], 1u);
            const u32 inc = wave_incl_scan_u32(mine);
            K1D_SYNC();
            if (lane == 63u) wh[1][w] = inc;
            K1D_SYNC();
            u32 slot = inc - mine;
            for (u32 ww = 0; ww < w; ww++) slot += wh[1][ww];
            if (tid == 255u) {
                const u32 total = slot + mine;
                sbase = total ? atomicAdd(&B.dcnt[(size_t)(r + 1u) * B.rstride + K1_BI(B, b)], total) : 0u;
            }
            K1D_SYNC();
            u64* L = B.rlist[0] + (size_t)b * g.stride;
            slot += sbase;
#pragma unroll
            for (u32 c = 0; c < CH; c++) {
                const u32 i = p0 + c;
                if (i < len && slv[c] >= 2u && slv[c] <= K1D_GS) {
                    if (slot < g.stride) L[slot] = K1E_MAKE(slv[c] - 1u, i - hpv[c], sv[c], start + i);
                    slot++;
                }
            }
        }
    }
}

// stable 7-bit LSD pass over (key, value) pairs of a range, through global memory (16 waves)
static __device__ void k1d_radix_pass(const u32* srcK, const u32* srcV, u32* dstK, u32* dstV, u32 L, u32 shift,
                                      u32 (*wh)[128], u32* dtot) {
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    for (u32 i = tid; i < 2048; i += 1024) (&wh[0][0])[i] = 0;
    __syncthreads();
    const u32 chunk = (((L + 15u) / 16u) + 63u) & ~63u;
    const u32 lo = w * chunk < L ? w * chunk : L;
    const u32 hi = lo + chunk < L ? lo + chunk : L;
    for (u32 i = lo + lane; i < hi; i += 64) atomicAdd(&wh[w][(srcK[i] >> shift) & 127u], 1u);
    __syncthreads();
    if (tid < 128) {
        u32 run = 0;
        for (int ww = 0; ww < 16; ww++) { const u32 c = wh[ww][tid]; wh[ww][tid] = run; run += c; }
        dtot[tid] = run;
    }
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (int d = 0; d < 128; d++) { const u32 c = dtot[d]; dtot[d] = run; run += c; }
    }
    __syncthreads();
    for (u32 e = tid; e < 2048; e += 1024) wh[e >> 7][e & 127u] += dtot[e & 127u];
    __syncthreads();
    const u64 lt = lanemask_lt();
    for (u32 i0 = lo; i0 < hi; i0 += 64) {
        const u32 i = i0 + lane;
        const bool valid = i < hi;
        const u32 k = valid ? srcK[i] : 0u, v = valid ? srcV[i] : 0u;
        const u32 d = (k >> shift) & 127u;
        const u64 m = match_any(d, 7, valid);
        const u32 rank = (u32)__popcll(m & lt), cnt = (u32)__popcll(m);
        const u32 base = valid ? wh[w][d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wh[w][d] = base + cnt;
        __builtin_amdgcn_wave_barrier();
        if (valid) { dstK[base + rank] = k; dstV[base + rank] = v; }
    }
    __syncthreads();
}

// groups of more than K1_MED_MAX rotations: one 1024-thread workgroup each
#define K1D_MAJ_SIDE 2048u
__global__ __launch_bounds__(1024) void k1d_large(K1Buf B, BatchGeom g, u32 r, u64 h, u32 final_h) {
    __shared__ u32 wh[16][128];
    __shared__ u32 dtot[128];
    __shared__ u32 s_pivot, s_side[2];
    __shared__ u32 sideK[2][K1D_MAJ_SIDE], sideV[2][K1D_MAJ_SIDE];
    __shared__ u32 scan_sh[20];
    __shared__ u32 s_carry;
    const u32 tid = threadIdx.x, lane = tid & 63u;
    u32 cnt = B.dbn[r * 4u + 1u];
    if (cnt > B.listLCap) cnt = B.listLCap;
    for (u32 gi = blockIdx.x; gi < cnt; gi += gridDim.x) {
        const u64 d = B.listL[r & 1u][gi];
        const u32 b = K1D_DB(d), start = K1D_DSTART(d), L = K1D_DLEN(d);
        const u32 n = B.nfront[b];
        const u32 mode = k1d_mode(B, r, b, final_h), linear = B.linear;
        const u32 hm = (u32)(h % n);
        const u32* ISA = B.ISA + (size_t)b * g.stride;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u32* SB = B.SB + (size_t)b * g.stride + start;
        u32* KA = B.KA + (size_t)b * g.stride + start;
        u32* KB = B.KB + (size_t)b * g.stride + start;
        __syncthreads();
        if (mode && !linear && (n / L) * L == n) {          // (uniform) the tie-break of a cyclic block: positions by formula -
            const u32 pp = n / L;                           // the members are r, r + p, ..., r + (L - 1) p: written, not permuted
            const u32 rr = SA[0] % pp;
            __syncthreads();
            for (u32 i = tid; i < L; i += 1024) SA[i] = rr + (L - 1u - i) * pp;
            continue;
        }
        // Repetitive inputs keep huge groups alive for log2(n) rounds in which all but ~2h keys of a group are equal.  A pivot
        // taken from the middle of the group is then the majority key: count the two sides while gathering the keys, and if
        // both are small do ONE stable 3-way partition pass (the sides are sorted in LDS) instead of three radix passes.
        if (tid == 0) { s_pivot = k1d_key(ISA, n, SA[L >> 1], h, hm, mode, linear); s_side[0] = 0; s_side[1] = 0; s_carry = 0; }
        __syncthreads();
        const u32 pivot = s_pivot;
        u32 myl = 0, myg = 0;
        for (u32 i = tid; i < L; i += 1024) {
            const u32 s = SA[i];
            const u32 k = k1d_key(ISA, n, s, h, hm, mode, linear);
            SB[i] = s;
            KB[i] = k;
            myl += k < pivot ? 1u : 0u;
            myg += k > pivot ? 1u : 0u;
        }
        for (u32 off = 32; off; off >>= 1) { myl += __shfl_xor(myl, off); myg += __shfl_xor(myg, off); }
        if (lane == 0) { if (myl) atomicAdd(&s_side[0], myl); if (myg) atomicAdd(&s_side[1], myg); }
        __syncthreads();
        const u32 nlt = s_side[0], ngt = s_side[1];
        if (nlt == 0u && ngt == 0u) {
            // every key equals the pivot (periodic input: every round but the last): the group stays as it is, ranks and all -
            // only its descriptor goes on to the next round
            if (tid == 0) k1d_push_big(B, r + 1u, b, start, L);
            continue;
        }
        if (n < (1u << 21) && nlt <= K1D_MAJ_SIDE && ngt <= K1D_MAJ_SIDE && (nlt + ngt) * 4u < L) {   // (key << 11 | index) needs keys < 2^21
            const u32 neq = L - nlt - ngt;
            u32 runE = 0, runL = 0, runG = 0;
            for (u32 c0 = 0; c0 < L; c0 += 1024) {
                const u32 i = c0 + tid;
                const bool valid = i < L;
                const u32 k = valid ? KB[i] : pivot, v = valid ? SB[i] : 0u;
                const bool isE = valid && k == pivot, isL = valid && k < pivot, isG = valid && k > pivot;
                u32 tot;
                const u32 ex = block_excl_scan_1024((isE ? 1u : 0u) | (isL ? 1u << 16 : 0u), dtot, &tot);
                const u32 posE = ex & 0xffffu, posL = ex >> 16, posG = tid - posE - posL;     // every earlier lane of a chunk is valid
                if (isE) { KA[nlt + runE + posE] = k; SA[nlt + runE + posE] = v; }
                if (isL) { sideK[0][runL + posL] = k; sideV[0][runL + posL] = v; }
                if (isG) { sideK[1][runG + posG] = k; sideV[1][runG + posG] = v; }
                const u32 cE = tot & 0xffffu, cL = tot >> 16, cV = L - c0 < 1024u ? L - c0 : 1024u;
                runE += cE; runL += cL; runG += cV - cE - cL;
            }
            __syncthreads();
            for (int side = 0; side < 2; side++) {
                const u32 m = side ? ngt : nlt, dst0 = side ? nlt + neq : 0u;
                if (m == 0) continue;                                   // uniform
                u32 P = 2; while (P < m) P <<= 1;
                u32* comp = sideK[side];                                // (key << 11 | arrival index): keys < 2^21
                for (u32 i = tid; i < P; i += 1024) comp[i] = i < m ? (comp[i] << 11) | i : 0xFFFFFFFFu;
                __syncthreads();
                for (u32 kk = 2; kk <= P; kk <<= 1)
                    for (u32 j = kk >> 1; j > 0; j >>= 1) {
                        for (u32 i = tid; i < P; i += 1024) {
                            const u32 x = i ^ j;
                            if (x > i) {
                                const u32 a = comp[i], bb = comp[x];
                                const bool up = (i & kk) == 0;
                                if ((a > bb) == up) { comp[i] = bb; comp[x] = a; }
                            }
                        }
                        __syncthreads();
                    }
                for (u32 i = tid; i < m; i += 1024) {
                    const u32 c = comp[i];
                    KA[dst0 + i] = c >> 11;
                    SA[dst0 + i] = sideV[side][c & 2047u];
                }
                __syncthreads();
            }
        } else {
            k1d_radix_pass(KB, SB, KA, SA, L, 0, wh, dtot);
            k1d_radix_pass(KA, SA, KB, SB, L, 7, wh, dtot);
            k1d_radix_pass(KB, SB, KA, SA, L, 14, wh, dtot);
            if (n >= (1u << 21)) {                                      // BWT.* entry points on blocks of 2^21 .. 2^22-1 bytes
                k1d_radix_pass(KA, SA, KB, SB, L, 21, wh, dtot);
                k1d_radix_pass(KB, SB, KA, SA, L, 28, wh, dtot);
            }
        }
        __threadfence_block();
        __syncthreads();
        // sorted keys in KA, rotations in SA: head position of every position into SB, sub-group lengths at the heads into KB
        // (the length of a sub-group is known when the NEXT head comes by), 1024 positions at a time
        bool split = false;
        for (u32 i0 = 0; i0 < L; i0 += 1024) {
            const u32 i = i0 + tid;
            const bool in = i < L;
            const bool head = in && (i == 0 || KA[i] != KA[i - 1]);
            const u32 v = head ? i + 1u : 0u;
            u32 m = v;                                       // inclusive max-scan (values are monotone where non-zero)
            for (u32 off = 1; off < 64; off <<= 1) {
                const u32 u = __shfl_up(m, off);
                if (lane >= off && u > m) m = u;
            }
            if (lane == 63u) scan_sh[tid >> 6] = m;
            __syncthreads();
            u32 wprev = s_carry;
            for (u32 ww = 0; ww < (tid >> 6); ww++) wprev = scan_sh[ww] > wprev ? scan_sh[ww] : wprev;
            const u32 incl = m > wprev ? m : wprev;          // last head (+1) at or before i
            u32 excl = __shfl_up(m, 1u);
            if (lane == 0) excl = 0;
            excl = excl > wprev ? excl : wprev;              // last head (+1) strictly before i
            if (in) SB[i] = start + incl - 1u;
            if (head && i > 0u) { KB[excl - 1u] = i - (excl - 1u); split = true; }
            __syncthreads();
            if (tid == 1023) s_carry = incl;
            __syncthreads();
        }
        if (tid == 0) KB[s_carry - 1u] = L - (s_carry - 1u);            // the last sub-group
        if (__ballot(split) && lane == 0) atomicOr(&B.dchg[(size_t)r * B.rstride + K1_BI(B, b)], 1u);
        __threadfence_block();
        k1d_push_chunks(B, r, b, start, L);
    }
}

// ---------------------------------------------------------------------------------------------
// the writes of round r: new ranks, the next round's lists
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1d_update(K1Buf B, BatchGeom g, u32 r) {
    __shared__ u32 sb[K1D_FW], pre[K1D_FW];
    __shared__ u32 obase;
    __shared__ u32 tpre[K1D_NBW + 1], scr[256];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32 G = gridDim.x;
    const u32* cnt_row = B.dcnt + (size_t)r * B.rstride;
    for (u32 b0 = 0; b0 < g.nb; b0 += 8u * K1D_NBW) {
        const K1dSeq q = k1d_seq(g.nb, b0);
        const u32 ntile = k1d_tile_prefix(B, cnt_row, q, g.stride, tpre, scr);
        auto load = [&](u32 f, u32& b, u32& e0, u32& cnt, u64 (&e)[K1D_RPT]) {
            b = 0; e0 = 0; cnt = 0;
            if (f < ntile) {
                const u32 i = k1d_tile_block(tpre, q.slots, f);
                b = q.first + q.step * i;
                e0 = (f - tpre[i]) * K1D_T;
                cnt = cnt_row[K1_BI(B, b)] < g.stride ? cnt_row[K1_BI(B, b)] : g.stride;
            }
            const u64* Lt = B.rlist[1] + (size_t)b * g.stride;
#pragma unroll
            for (u32 it = 0; it < K1D_RPT; it++) {
                const u32 i = (it * 4u + w) * 64u + lane;
                e[it] = (e0 < cnt && i < cnt - e0) ? Lt[e0 + i] : 0ull;
            }
        };
        u64 e[K1D_RPT], en[K1D_RPT];
        u32 b, e0, cnt, bn, e0n, cntn;
        u32 f = q.wg;
        if (f < ntile) load(f, b, e0, cnt, e);
        for (; f < ntile; f += q.nwg) {
            load(f + q.nwg, bn, e0n, cntn, en);                 // the next tile's entries: in flight while this one is written
            u32* ISA = B.ISA + (size_t)b * g.stride;
            u64* Lout = B.rlist[0] + (size_t)b * g.stride;
            bool sv[K1D_RPT];
            K1D_SYNC();
#pragma unroll
            for (u32 it = 0; it < K1D_RPT; it++) {
                const u32 i = (it * 4u + w) * 64u + lane;
                const bool valid = e0 < cnt && i < cnt - e0;
                // an entry belongs to the tile in which its (new) group starts
                const bool mine = valid && (K1E_LEN(e[it]) == 1u ? i < K1D_T : i - K1E_IDX(e[it]) < K1D_T);
                if (mine && !(e[it] & K1E_KEEP)) ISA[K1E_S(e[it])] = K1E_POS(e[it]) - K1E_IDX(e[it]);
                sv[it] = mine && K1E_LEN(e[it]) >= 2u;
                const u64 bal = __ballot(sv[it]);
                if (lane == 0) { sb[(it * 4u + w) * 2u] = (u32)bal; sb[(it * 4u + w) * 2u + 1u] = (u32)(bal >> 32); }
            }
            K1D_SYNC();
            if (w == 0) {
                const u32 c = lane < K1D_FW ? (u32)__popc(sb[lane]) : 0u;
                const u32 inc = wave_incl_scan_u32(c);
                if (lane < K1D_FW) pre[lane] = inc - c;
                const u32 total = (u32)__shfl((int)inc, 63);
                if (lane == 0) obase = total ? atomicAdd(&B.dcnt[(size_t)(r + 1u) * B.rstride + K1_BI(B, b)], total) : 0u;
            }
            K1D_SYNC();
#pragma unroll
            for (u32 it = 0; it < K1D_RPT; it++) {
                const u32 i = (it * 4u + w) * 64u + lane;
                if (sv[it]) {
                    const u32 idx = obase + pre[i >> 5] + (u32)__popc(sb[i >> 5] & ((1u << (i & 31u)) - 1u));
                    if (idx < g.stride) Lout[idx] = e[it] & ~K1E_KEEP;
                }
            }
#pragma unroll
            for (u32 it = 0; it < K1D_RPT; it++) e[it] = en[it];
            b = bn; e0 = e0n; cnt = cntn;
        }
    }
    // the ranges of this round's big groups, chunk by chunk
    u32 nch = B.dbn[r * 4u + 2u];
    if (nch > B.listSCap) nch = B.listSCap;
    for (u32 c = blockIdx.x; c < nch; c += G) {
        const u64 d = B.listS[0][c];
        const u32 b = K1D_CB(d), cs = K1D_CS(d), cl = K1D_CL(d);
        k1d_window<false>(B, g, b, cs, cl, K1D_CA(d), r + 1u);
    }
    // the medium groups of this round (k1d_med wrote their lists): the ranks of what left the group's first sub-group
    for (u32 cls = 0; cls < 2u; cls++) {
        const u64* list = cls ? B.listM[r & 1u] : B.listT[r & 1u];
        const u32 lcap = cls ? B.listMCap : B.listTCap;
        u32 ng = B.dbn[r * 4u + (cls ? 0u : 3u)];
        if (ng > lcap) ng = lcap;
        for (u32 gi = blockIdx.x; gi < ng; gi += G) {
            const u64 d = list[gi];
            const u32 b = K1D_DB(d), start = K1D_DSTART(d), len = K1D_DLEN(d);
            const u32* SA = B.SA + (size_t)b * g.stride + start;
            const u32* R = B.SB + (size_t)b * g.stride + start;
            u32* ISA = B.ISA + (size_t)b * g.stride;
            for (u32 j = tid; j < len; j += 256) {
                const u32 rk = R[j];
                if (rk != K1D_RKEEP) ISA[SA[j]] = rk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
int k1_dbl_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream, u32 h0, u32 check_h) {
    if (g.nb > 4095u || max_n >= (1u << 22)) return CJS_E_UNSUPPORTED;      // 12 block bits in the descriptors, 22 position bits in the entries
    K1Prof* pr = B.prof;
    if (pr && pr->enabled) __atomic_fetch_add(&pr->dbl_runs, 1u, __ATOMIC_RELAXED);
    u32 slot = k1_prof_begin(pr, K1P_DBUILD, stream);
    hipLaunchKernelGGL(k1d_build, dim3(g.htiles, g.nb), dim3(256), 0, stream, B, g);
    k1_prof_end(pr, slot, stream, (u64)g.nb * max_n);
    const u64 full = ((u64)g.nb * max_n + K1D_T - 1u) / K1D_T;      // tiles if every rotation were listed
    u32 bg = g.nb * 8u;                                   // workgroups of the big-group kernels
    if (bg < 64u) bg = 64u;
    if (bg > 2048u) bg = 2048u;
#ifdef CJS_CPU_DEBUG_BUILD
    if (bg > 16u) bg = 16u;
#endif
    u64 h = h0 ? h0 : 1u;
    for (u32 r = 0; r < K1D_MAXR; r++, h <<= 1) {
        const u32 final_h = h >= max_n ? 1u : 0u;
        // workgroups that walk the entry lists (each takes every G-th tile of its XCD's tile sequence): what is resident at once -
        // 1024 / 2048 / 4096 measured on E8S-A: 13.78 / 13.51 / 13.72 ms per 10^8 bytes
#ifdef CJS_CPU_DEBUG_BUILD
        u64 wgs = 24u;                                     // (the CPU logic-debug build of the tests runs the workgroups one after another)
#else
        u64 wgs = 2048u;
#endif
        if (wgs > full) wgs = full ? full : 1u;
        slot = k1_prof_begin(pr, K1P_DROUND, stream);
        hipLaunchKernelGGL(k1d_round, dim3((u32)wgs), dim3(256), 0, stream, B, g, r, h, final_h);
        k1_prof_end(pr, slot, stream, 0);
        slot = k1_prof_begin(pr, K1P_DMED, stream);
        hipLaunchKernelGGL(k1d_med<K1D_MED1>, dim3(bg * 2u), dim3(256), 0, stream, B, g, r, h, final_h);
        // (a group of 1025 .. 4096 is a chain of ~20 us for its workgroup, and E8S-A's first round has 8 500 of them, sizes all over the range: four times
        // the workgroups of round 4's first version even the shares out - 13.2 -> 12.8 ms per 10^8 bytes; 2 x .. 8 x measure alike.  Running the three
        // group kernels on a second stream NEXT TO k1d_round was tried too: they are not independent of it - the stream's digest moved - and gained 1 %)
        hipLaunchKernelGGL(k1d_med<K1_MED_MAX>, dim3(bg * 4u), dim3(256), 0, stream, B, g, r, h, final_h);
        k1_prof_end(pr, slot, stream, 0);
        slot = k1_prof_begin(pr, K1P_DLARGE, stream);
        hipLaunchKernelGGL(k1d_large, dim3(bg < 256u ? bg : 256u), dim3(1024), 0, stream, B, g, r, h, final_h);
        k1_prof_end(pr, slot, stream, 0);
        slot = k1_prof_begin(pr, K1P_DUPDATE, stream);
        hipLaunchKernelGGL(k1d_update, dim3((u32)wgs), dim3(256), 0, stream, B, g, r);
        k1_prof_end(pr, slot, stream, 0);
        if (final_h) break;
        if (check_h && h >= check_h) {
            // From h = check_h on the host looks after every round at what the next one would find (entries per block, descriptors of the three
            // group classes): inputs whose ties end within a few KB - text, HTML: E8S-A's last entries are those of the round with h = 8192 - have
            // nothing left by then, and every later round would be five empty launches (7 rounds: 0.25 ms per 10^8 bytes, on both streams at
            // once: E8S-A 12.75 -> 12.5 ms).  Lists that are empty stay empty.  Inputs that tie for longer (tiled) pay a stream sync per round.
            // (the block totals of every later round and the descriptor counts lie in one stretch - dcnt rows r + 1 .., dchg, dtot, dbn: one copy;
            // rows beyond r + 1 are still zero, dchg and dtot are skipped when summing)
            const u32* from = B.dcnt + (size_t)(r + 1u) * B.rstride;
            const size_t words = (size_t)(B.dbn + (size_t)(K1D_MAXR + 2u) * 4u - from);
            std::vector<u32> pageable;
            u32* left = B.hpin;
            if (!left || words > B.hpinWords) { pageable.resize(words); left = pageable.data(); }
            HIP_CHECK_RET(hipMemcpyAsync(left, from, words * 4, hipMemcpyDeviceToHost, stream));
            HIP_CHECK_RET(hipStreamSynchronize(stream));
            u64 any = 0;
            for (u32 i = 0; i < B.rstride; i++) any += left[i];
            const u32* dbn = left + (B.dbn - from) + (size_t)(r + 1u) * 4u;
            for (u32 i = 0; i < 4u; i++) any += dbn[i];
            if (!any) break;
        }
    }
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
