// K1 front end: the initial sort of all rotations of every block by their first 8 bytes, as a SAMPLE SORT
// (one partition pass through HBM + one in-LDS sort per bucket) instead of seven LSD radix passes over
// (key, index) pairs.
//
// Replaces steps 1 of k1_bwt.hip (k1_hist/k1_scan/k1_scatter x 7 + k1_init_heads), i.e. the first part of the
// work SA-IS does in BWT.bwtransform2 (lib/BWT.js:372-417, :197-300).  Nothing of the reference is ported: any
// algorithm that delivers "rotations ordered by their first 8 bytes, groups of equal prefixes marked" feeds the
// refinement stages (K1-deep, prefix doubling) unchanged, and the final order is the reference's.
//
//   k1f_sample   per block: K1F_S keys (8 text bytes each) at stratified, hashed positions, bitonic-sorted in
//                LDS; every K1F_OVS-th is a splitter.  A key that fills more than one quantile gets a bucket of
//                its own ([v, v+1): nothing to sort there), so runs/periodic data cannot overflow a bucket.
//   k1f_hist     per tile of K1F_PT rotations: key of every rotation from the LDS-staged text, bucket = number
//                of splitters <= key (branch-free binary search in LDS), per-tile bucket counts, bucket ids (u16).
//   k1f_scan     per block: bucket starts and per-(tile, bucket) write offsets.
//   k1f_scatter  rotation indices to their bucket (4 bytes per rotation; order inside a bucket is irrelevant).
//   k1f_bsort    one workgroup per bucket: gathers the 8-byte keys from the block's text (L2-resident, all tiles
//                of a block run on one XCD), sorts (key, index) in LDS with stable 8-bit LSD passes over the
//                bytes that actually vary inside the bucket, writes the suffix array slice and the group heads.
//
// HBM traffic per rotation: text 1 + ids 2+2 + indices 4+4 + suffix array 4 + keys gathered from L2 = ~17 bytes
// (the LSD design moved 7 x 20 = 140).  All integer work.
#include "k1_bwt.h"
#include "devutil.h"

__device__ __forceinline__ u64 k1f_load_be64(const u8* T, u32 p) {
    const u32 sh = p & 3u;
    u32 d[3];
    __builtin_memcpy(d, __builtin_assume_aligned(T + (p - sh), 4), 12);
    const u32 w0 = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
    const u32 w1 = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
    return ((u64)__builtin_bswap32(w0) << 32) | (u64)__builtin_bswap32(w1);
}

__device__ __forceinline__ u32 k1f_hash(u32 k, u32 b) {
    u32 h = k * 2654435761u ^ (b + 1u) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return h;
}

// ---------------------------------------------------------------------------------------------
// splitters
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k1f_sample(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 tid = threadIdx.x;
    u64* sp = B.fsplit + (size_t)b * K1F_NB;
    if (n <= K1F_C) {                                   // one bucket holds the whole block
        for (u32 j = tid; j < K1F_NB; j += 1024) sp[j] = ~0ull;
        return;
    }
    __shared__ u64 s[K1F_S];
    const u8* T = B.T + (size_t)b * g.tstride;
    for (u32 k = tid; k < K1F_S; k += 1024) {
        const u32 lo = (u32)((u64)k * n / K1F_S), hi = (u32)((u64)(k + 1u) * n / K1F_S);
        const u32 p = hi > lo + 1u ? lo + k1f_hash(k, b) % (hi - lo) : lo;
        s[k] = k1f_load_be64(T, p);
    }
    __syncthreads();
    for (u32 kk = 2; kk <= K1F_S; kk <<= 1) {
        for (u32 j = kk >> 1; j > 0; j >>= 1) {
            for (u32 i = tid; i < K1F_S / 2; i += 1024) {
                const u32 lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), hi = lo | j;
                const bool up = (lo & kk) == 0;
                const u64 a = s[lo], c = s[hi];
                if ((a > c) == up) { s[lo] = c; s[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (u32 j = tid; j < K1F_NB; j += 1024) {
        u64 v = ~0ull;                                  // sp[K1F_NB-1] is padding (never compared)
        if (j + 1u < K1F_NB) {
            const u64 q = s[(j + 1u) * K1F_OVS];
            v = q;
            if (j >= 1u && s[j * K1F_OVS] == q && q != ~0ull) v = q + 1u;   // heavy key: [q, q+1) becomes a bucket of its own
        }
        sp[j] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// partition: bucket ids + per-tile counts, offsets, scatter
// ---------------------------------------------------------------------------------------------
static inline u32 k1f_ptiles(const BatchGeom& g) { return (g.stride + K1F_PT - 1) / K1F_PT; }

__global__ __launch_bounds__(1024) void k1f_hist(K1Buf B, BatchGeom g, u32 ptiles) {
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 t0 = t * K1F_PT;
    const u32 tid = threadIdx.x;
    u32* th = B.tileHist + ((size_t)b * ptiles + t) * K1F_NB;
    if (t0 >= n) return;                                // k1f_scan only reads the tiles below n
    __shared__ u64 sp[K1F_NB];
    __shared__ u32 hist[K1F_NB];
    __shared__ u32 tx[K1F_PT / 4 + 4];
    const u64* gsp = B.fsplit + (size_t)b * K1F_NB;
    for (u32 d = tid; d < K1F_NB; d += 1024) { sp[d] = gsp[d]; hist[d] = 0; }
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32 avail = (g.tstride - t0) / 4u;            // dwords of this block's text slot from t0 on
    const u32* T32 = (const u32*)(T + t0);
    for (u32 i = tid; i < K1F_PT / 4 + 4; i += 1024) tx[i] = i < avail ? T32[i] : 0u;
    __syncthreads();
    u16* bid = (u16*)(B.KA + (size_t)b * g.stride);
#pragma unroll
    for (int it = 0; it < K1F_PT / 1024; it++) {
        const u32 q = (u32)it * 1024u + tid, j = t0 + q;
        if (j < n) {
            const u32 sh = q & 3u, wi = q >> 2;
            const u32 w0 = __builtin_amdgcn_alignbyte(tx[wi + 1], tx[wi], sh);
            const u32 w1 = __builtin_amdgcn_alignbyte(tx[wi + 2], tx[wi + 1], sh);
            const u64 key = ((u64)__builtin_bswap32(w0) << 32) | (u64)__builtin_bswap32(w1);
            u32 pos = 0;                                // number of splitters <= key
#pragma unroll
            for (u32 step = K1F_NB / 2; step >= 1; step >>= 1)
                if (sp[pos + step - 1u] <= key) pos += step;      // pos + step - 1 <= K1F_NB - 2
            atomicAdd(&hist[pos], 1u);
            bid[j] = (u16)pos;
        }
    }
    __syncthreads();
    for (u32 d = tid; d < K1F_NB; d += 1024) th[d] = hist[d];
}

// per block: bucket starts (fstart[0..K1F_NB], fstart[K1F_NB] = n) and tileHist[t][d] <- first write position
// of tile t in bucket d
#if K1F_NB <= 1024
__global__ __launch_bounds__(1024) void k1f_scan(K1Buf B, BatchGeom g, u32 ptiles) {
    const u32 b = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 nt = (n + K1F_PT - 1) / K1F_PT;
    constexpr u32 Q = 1024 / K1F_NB;                    // tile ranges summed in parallel
    __shared__ u32 part[Q][K1F_NB];
    __shared__ u32 sh[20];
    __shared__ u32 dbase[K1F_NB];
    const u32 tid = threadIdx.x, q = tid / K1F_NB, d = tid % K1F_NB;
    const u32 per = (nt + Q - 1) / Q;
    const u32 tlo = q * per < nt ? q * per : nt;
    const u32 thi = tlo + per < nt ? tlo + per : nt;
    u32* hist = B.tileHist + (size_t)b * ptiles * K1F_NB;
    u32 sum = 0;
#pragma unroll 8
    for (u32 t = tlo; t < thi; t++) sum += hist[(size_t)t * K1F_NB + d];
    part[q][d] = sum;
    __syncthreads();
    u32 tot = 0;
    if (tid < K1F_NB) for (u32 qq = 0; qq < Q; qq++) tot += part[qq][tid];
    u32 total;
    const u32 excl = block_excl_scan_1024(tid < K1F_NB ? tot : 0u, sh, &total);
    if (tid < K1F_NB) {
        dbase[tid] = excl;
        B.fstart[(size_t)b * (K1F_NB + 1) + tid] = excl;
        if (tid == 0) B.fstart[(size_t)b * (K1F_NB + 1) + K1F_NB] = n;
    }
    __syncthreads();
    u32 run = dbase[d];
    for (u32 qq = 0; qq < q; qq++) run += part[qq][d];
#pragma unroll 8
    for (u32 t = tlo; t < thi; t++) {
        const u32 c = hist[(size_t)t * K1F_NB + d];
        hist[(size_t)t * K1F_NB + d] = run;
        run += c;
    }
}

#else
// more buckets than threads: thread t owns buckets t, t + 1024, ...
__global__ __launch_bounds__(1024) void k1f_scan(K1Buf B, BatchGeom g, u32 ptiles) {
    const u32 b = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 nt = (n + K1F_PT - 1) / K1F_PT;
    constexpr u32 R = K1F_NB / 1024;
    __shared__ u32 sh[20];
    const u32 tid = threadIdx.x;
    u32* hist = B.tileHist + (size_t)b * ptiles * K1F_NB;
    u32 sum[R], base[R];
#pragma unroll
    for (u32 r = 0; r < R; r++) {
        const u32 d = tid + r * 1024u;
        u32 a = 0;
#pragma unroll 8
        for (u32 t = 0; t < nt; t++) a += hist[(size_t)t * K1F_NB + d];
        sum[r] = a;
    }
    u32 slab = 0;
#pragma unroll
    for (u32 r = 0; r < R; r++) {
        u32 total;
        base[r] = slab + block_excl_scan_1024(sum[r], sh, &total);
        slab += total;
        B.fstart[(size_t)b * (K1F_NB + 1) + tid + r * 1024u] = base[r];
    }
    if (tid == 0) B.fstart[(size_t)b * (K1F_NB + 1) + K1F_NB] = n;
#pragma unroll
    for (u32 r = 0; r < R; r++) {
        const u32 d = tid + r * 1024u;
        u32 run = base[r];
#pragma unroll 8
        for (u32 t = 0; t < nt; t++) {
            const u32 c = hist[(size_t)t * K1F_NB + d];
            hist[(size_t)t * K1F_NB + d] = run;
            run += c;
        }
    }
}
#endif

__global__ __launch_bounds__(1024) void k1f_scatter(K1Buf B, BatchGeom g, u32 ptiles) {
    u32 b, t;
    if (!xcd_block_tile(g.nb, b, t)) return;
    const u32 n = B.nlen[b];
    const u32 t0 = t * K1F_PT;
    if (t0 >= n) return;
    __shared__ u32 cnt[K1F_NB], base[K1F_NB];
    const u32 tid = threadIdx.x;
    const u32* th = B.tileHist + ((size_t)b * ptiles + t) * K1F_NB;
    for (u32 d = tid; d < K1F_NB; d += 1024) { cnt[d] = 0; base[d] = th[d]; }
    __syncthreads();
    const u16* bid = (const u16*)(B.KA + (size_t)b * g.stride);
    u32* SB = B.SB + (size_t)b * g.stride;
    u32 dv[K1F_PT / 1024];
#pragma unroll
    for (int it = 0; it < K1F_PT / 1024; it++) {
        const u32 j = t0 + (u32)it * 1024u + tid;
        dv[it] = j < n ? bid[j] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int it = 0; it < K1F_PT / 1024; it++) {
        const u32 j = t0 + (u32)it * 1024u + tid;
        if (dv[it] != 0xFFFFFFFFu) {
            const u32 r = atomicAdd(&cnt[dv[it]], 1u);
            SB[base[dv[it]] + r] = j;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bucket sort
// ---------------------------------------------------------------------------------------------
// Head bits of suffix-array positions [start, end) into the block's bitmap: rows of 64 positions aligned to the
// bitmap words; words that lie entirely inside the range are stored, the (at most two) edge words are OR-ed.
template <class F>
__device__ __forceinline__ void k1f_write_heads(u32* HN, u32 start, u32 end, F is_head) {
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, nw = blockDim.x >> 6;
    const u32 a0 = start & ~63u;
    for (u32 r0 = a0 + w * 64u; r0 < end; r0 += nw * 64u) {
        const u32 p = r0 + lane;
        const bool in = p >= start && p < end;
        const bool h = in && is_head(p);
        const u64 bal = __ballot(h);
        if (lane == 0 || lane == 32) {
            const u32 wp = r0 + lane;                   // first position of this 32-bit word
            const u32 bits = lane == 0 ? (u32)bal : (u32)(bal >> 32);
            if (wp >= start && wp + 32u <= end) HN[wp >> 5] = bits;
            else if (bits) atomicOr(&HN[wp >> 5], bits);
        }
    }
}

// K1F_TRACE builds: s_memtime stamps between the stages of k1f_bsort, summed (in units of 256 clocks) into
// stats[K1_STAT_FRONT_BIG+1 ..]; k1_run prints them with CJS_K1_TRACE=1.  Not in product builds.
#ifdef K1F_TRACE
#define K1F_STAMP(slot) do { if (tid == 0) { const long long now_ = clock64(); atomicAdd(&B.stats[K1_STAT_FRONT_BIG + 1 + (slot)], (u32)((now_ - tprev_) >> 8)); tprev_ = now_; } } while (0)
#else
#define K1F_STAMP(slot) do { } while (0)
#endif
#define K1F_BT 256                                      // threads of a bucket-sort workgroup
#define K1F_E (K1F_C / K1F_BT)                          // rotations per thread
// measured (10^8-byte enwik stream, k1f_bsort ms): 32 leaves x 4 samples 3.66, 32 x 2 4.06, 64 x 4 3.86, 64 x 2 3.49
#ifndef K1F_LK
#define K1F_LK 64                                       // local sub-buckets (leaves) of a bucket, at most
#endif
#ifndef K1F_LOVS
#define K1F_LOVS 2                                      // local samples per leaf
#endif
#define K1F_LS (K1F_LOVS * K1F_LK)                      // local samples, at most

// One workgroup per bucket.  The bucket's (key, index) pairs are brought into LDS and sorted by a second, LOCAL
// sample sort: up to 128 of the bucket's own keys are ranked by counting, every 4th is a local splitter, the
// rotations are partitioned into <= 32 leaves (unstable LDS counting), and every leaf is sorted by ONE WAVE by
// rank counting with the candidates broadcast through v_readlane: rank = #smaller + #equal-with-smaller-slot.
// A leaf of m <= 64 rotations costs m steps of ~9 instructions; there are five workgroup barriers per bucket (an
// 8-bit LSD sort of the same bucket took seven passes of ~20 barriers each and ran 4x longer).  Equal keys stay
// one group: its head is the member with no equal key in a smaller slot.
__global__ __launch_bounds__(K1F_BT) void k1f_bsort(K1Buf B, BatchGeom g) {
    u32 b, d;
    if (!xcd_block_tile(g.nb, b, d)) return;
    const u32 n = B.nlen[b];
    if (n == 0) return;
    const u32* fs = B.fstart + (size_t)b * (K1F_NB + 1);
    const u32 start = fs[d], end = fs[d + 1];
    if (end <= start) return;
    const u32 cnt = end - start;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* SB = B.SB + (size_t)b * g.stride + start;
    u32* SA = B.SA + (size_t)b * g.stride + start;
    u32* HN = B.HN + (size_t)b * g.hstride;
    const u64* sp = B.fsplit + (size_t)b * K1F_NB;
    // a bucket between the splitters v and v+1 holds one key only
    const bool pure = d > 0u && d < K1F_NB - 1u && sp[d] == sp[d - 1u] + 1u;
    if (pure || cnt == 1u) {
        for (u32 i = tid; i < cnt; i += K1F_BT) SA[i] = SB[i];
        k1f_write_heads(HN, start, end, [&](u32 p) { return p == start; });
        if (tid == 0 && cnt > 64u) atomicAdd(&B.stats[K1_STAT_BIGROT + (d & 7u)], cnt);     // one big group (see k1_run: K1-deep predictor)
        return;
    }
    __shared__ u64 key[K1F_C];
    __shared__ u32 idx[K1F_C];
    __shared__ u16 perm[K1F_C];                         // leaf order -> arrival slot
    __shared__ u64 smp[K1F_LS], sp2[K1F_LK];
    __shared__ u32 cnt2[K1F_LK], off2[K1F_LK + 1];
    __shared__ u32 srank[K1F_LS];
    __shared__ u32 hbits[K1F_C / 32 + 2];
    __shared__ u32 dstart[256], sh[256];
    __shared__ u32 single;
    if (cnt > K1F_C) {
        // ---- oversize bucket (unlucky sampling or a moderately heavy key): stable LSD passes through global memory,
        //      one digit byte gathered from the text per pass, ping-pong between the bucket's slices of SB and SA.
        //      Wave 0 scatters row by row (stable by construction); rare, so simple.
        if (tid == 0) { atomicAdd(&B.stats[K1_STAT_FRONT_BIG], 1u); atomicAdd(&B.stats[K1_STAT_BIGROT + (d & 7u)], cnt); }
        u32* bufs[2] = {(u32*)SB, SA};
        int cur = 0;
        for (u32 pass = 0; pass < 8u; pass++) {
            const u32* src = bufs[cur];
            u32* dst = bufs[cur ^ 1];
            dstart[tid] = 0;
            if (tid == 0) single = 0;
            __syncthreads();
            for (u32 i = tid; i < cnt; i += K1F_BT) atomicAdd(&dstart[T[src[i] + 7u - pass]], 1u);
            __syncthreads();
            const u32 c = dstart[tid];
            if (c == cnt) single = 1;
            __syncthreads();
            if (single) { __syncthreads(); continue; }  // every rotation has the same byte here (uniform)
            const u32 ex = block_excl_scan_256(c, sh);
            dstart[tid] = ex;
            __syncthreads();
            if (w == 0) {
                const u64 lt = lanemask_lt();
                for (u32 r0 = 0; r0 < cnt; r0 += 64u) {
                    const u32 i = r0 + lane;
                    const bool valid = i < cnt;
                    const u32 v = valid ? src[i] : 0u;
                    const u32 dg = valid ? T[v + 7u - pass] : 0u;
                    const u64 m = match_any(dg, 8, valid);
                    const u32 rank = (u32)__popcll(m & lt), c2 = (u32)__popcll(m);
                    const u32 bs = valid ? dstart[dg] : 0u;
                    __builtin_amdgcn_wave_barrier();
                    if (valid) dst[bs + rank] = v;
                    if (valid && rank == 0) dstart[dg] = bs + c2;
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __threadfence_block();
            __syncthreads();
            cur ^= 1;
        }
        if (cur == 0) {
            for (u32 i = tid; i < cnt; i += K1F_BT) SA[i] = SB[i];
            __threadfence_block();
            __syncthreads();
        }
        k1f_write_heads(HN, start, end, [&](u32 p) {
            return p == start || k1f_load_be64(T, SA[p - start]) != k1f_load_be64(T, SA[p - start - 1u]);
        });
        return;
    }
    // ---- the common case: everything in LDS.  Stage 0: indices, then keys (all loads of a stage in flight together)
#ifdef K1F_TRACE
    long long tprev_ = clock64();
#endif
    {
        u32 v[K1F_E];
#pragma unroll
        for (int it = 0; it < K1F_E; it++) {
            const u32 i = (u32)it * K1F_BT + tid;
            v[it] = i < cnt ? SB[i] : 0u;
        }
        u64 k[K1F_E];
#pragma unroll
        for (int it = 0; it < K1F_E; it++) {
            const u32 i = (u32)it * K1F_BT + tid;
            k[it] = i < cnt ? k1f_load_be64(T, v[it]) : 0ull;
        }
#pragma unroll
        for (int it = 0; it < K1F_E; it++) {
            const u32 i = (u32)it * K1F_BT + tid;
            if (i < cnt) { key[i] = k[it]; idx[i] = v[it]; }
        }
    }
    for (u32 i = tid; i < K1F_C / 32 + 2; i += K1F_BT) hbits[i] = 0;
    if (tid < K1F_LK) cnt2[tid] = 0;
    // leaves: ~24..48 rotations each
    K1F_STAMP(0);
    u32 K = 1u;
    while (K < K1F_LK && cnt >= 48u * K) K <<= 1;       // leaves of ~24..48 rotations (K1F_LK = 32), ~12..24 (64)
    if (K1F_LK == 64 && K > 1u && K < 64u) K <<= 1;
    __syncthreads();
    if (K > 1u) {
        // stage 1: 4K samples, ranked by counting; every 4th is a local splitter (equal neighbours: the heavy-key rule)
        const u32 LS = K1F_LOVS * K;
        // (all four waves: thread t ranks sample t % LS against one LS / (256 / LS)-th of the samples, partial ranks summed in LDS)
        const u32 parts = K1F_BT / LS < 1u ? 1u : (K1F_BT / LS > LS ? LS : K1F_BT / LS), si = tid % LS, part = tid / LS;   // powers of two
        if (tid < LS) { smp[tid] = key[(u32)((u64)tid * cnt / LS)]; srank[tid] = 0; }
        __syncthreads();
        if (part < parts) {
            const u64 mine = smp[si];
            const u32 per = LS / parts, j0 = part * per;
            u32 r = 0;
#pragma unroll 8
            for (u32 j = j0; j < j0 + per; j++) {
                const u64 o = smp[j];
                r += (o < mine || (o == mine && j < si)) ? 1u : 0u;
            }
            atomicAdd(&srank[si], r);
        }
        __syncthreads();
        u64 mine = 0;
        if (tid < LS) mine = smp[tid];
        __syncthreads();
        if (tid < LS) smp[srank[tid]] = mine;
        __syncthreads();
        if (tid < K1F_LK) {
            u64 v = ~0ull;
            if (tid + 1u < K) {
                const u64 q = smp[(tid + 1u) * K1F_LOVS];
                v = q;
                if (tid >= 1u && smp[tid * K1F_LOVS] == q && q != ~0ull) v = q + 1u;
            }
            sp2[tid] = v;
        }
        __syncthreads();
        K1F_STAMP(1);
        // stage 2: leaf of every rotation, slot inside the leaf by an LDS counter
        u32 ls[K1F_E];
#pragma unroll
        for (int it = 0; it < K1F_E; it++) {
            const u32 i = (u32)it * K1F_BT + tid;
            ls[it] = 0xFFFFFFFFu;
            if (i < cnt) {
                const u64 kk = key[i];
                u32 pos = 0;
                for (u32 step = K >> 1; step >= 1u; step >>= 1)
                    if (sp2[pos + step - 1u] <= kk) pos += step;
                ls[it] = (pos << 16) | atomicAdd(&cnt2[pos], 1u);
            }
        }
        __syncthreads();
        if (w == 0) {
            const u32 c = lane < K ? cnt2[lane] : 0u;
            const u32 inc = wave_incl_scan_u32(c);
            if (lane < K1F_LK) off2[lane] = inc - c;
            if (lane == 0) off2[K1F_LK] = cnt;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < K1F_E; it++)
            if (ls[it] != 0xFFFFFFFFu) perm[off2[ls[it] >> 16] + (ls[it] & 0xFFFFu)] = (u16)((u32)it * K1F_BT + tid);
    } else {
        for (u32 i = tid; i < cnt; i += K1F_BT) perm[i] = (u16)i;
        if (tid == 0) { cnt2[0] = cnt; off2[0] = 0; off2[1] = cnt; }
    }
    __syncthreads();
    K1F_STAMP(2);
    // stage 3: one wave per leaf.  Every member ranks itself against all members, the candidates broadcast through
    // v_readlane (no memory access in the loop): less = smaller keys, eqb = equal keys in earlier slots.
    // (handing the leaves out through an LDS counter instead of round-robin hung on the MI355X - for (;;) around a
    // lane-0 atomic + v_readfirstlane - although it ran on the CPU build; the static split costs < 5 %)
    for (u32 lf = w; lf < K; lf += K1F_BT / 64) {
        const u32 m = (u32)__builtin_amdgcn_readfirstlane((int)cnt2[lf]), o = (u32)__builtin_amdgcn_readfirstlane((int)off2[lf]);   // wave-uniform, in SGPRs
        if (m == 0) continue;
        const bool pure2 = K > 1u && lf > 0u && lf + 1u < K && sp2[lf] == sp2[lf - 1u] + 1u;    // one key only
        if (m > 64u && lane == 0) atomicAdd(&B.stats[K1_STAT_BIGROT + (d & 7u)], m);            // a leaf this big is (mostly) one key
        if (pure2 || m == 1u) {
            for (u32 j = lane; j < m; j += 64u) SA[o + j] = idx[perm[o + j]];
            if (lane == 0) atomicOr(&hbits[o >> 5], 1u << (o & 31u));
            continue;
        }
        if (m <= 64u) {
            // the common case, one row: count the smaller keys only (7 instructions per candidate); members with equal keys
            // end up with equal counts (and only they do), so their order and the group head fall out of one match_any
            const bool valid = lane < m;
            const u32 e = valid ? perm[o + lane] : 0u;
            const u64 ke = valid ? key[e] : 0ull;
            const int clo = (int)(u32)ke, chi = (int)(u32)(ke >> 32);
            u32 less = 0;
            for (u32 t = 0; t < m; t++) {
                const u64 kt = ((u64)(u32)__builtin_amdgcn_readlane(chi, (int)t) << 32) | (u64)(u32)__builtin_amdgcn_readlane(clo, (int)t);
                less += kt < ke ? 1u : 0u;
            }
            const u64 same = match_any(less, 6, valid);
            const u32 eqb = (u32)__popcll(same & lanemask_lt());
            if (valid) {
                const u32 q = o + less + eqb;
                SA[q] = idx[e];
                if (eqb == 0) atomicOr(&hbits[q >> 5], 1u << (q & 31u));
            }
            continue;
        }
        for (u32 r0 = 0; r0 < m; r0 += 64u) {
            const u32 j = r0 + lane;
            const bool valid = j < m;
            const u32 e = valid ? perm[o + j] : 0u;
            const u64 ke = valid ? key[e] : 0ull;
            u32 less = 0, eqb = 0;
            for (u32 c0 = 0; c0 < m; c0 += 64u) {
                const u32 cj = c0 + lane;
                const u64 ck = cj < m ? (c0 == r0 ? ke : key[perm[o + cj]]) : 0ull;
                const int clo = (int)(u32)ck, chi = (int)(u32)(ck >> 32);
                const u32 mm = m - c0 < 64u ? m - c0 : 64u;
                for (u32 t = 0; t < mm; t++) {
                    const u64 kt = ((u64)(u32)__builtin_amdgcn_readlane(chi, (int)t) << 32) | (u64)(u32)__builtin_amdgcn_readlane(clo, (int)t);
                    less += kt < ke ? 1u : 0u;
                    eqb += (kt == ke && c0 + t < j) ? 1u : 0u;
                }
            }
            if (valid) {
                const u32 q = o + less + eqb;
                SA[q] = idx[e];
                if (eqb == 0) atomicOr(&hbits[q >> 5], 1u << (q & 31u));
            }
        }
    }
    __syncthreads();
    K1F_STAMP(3);
    k1f_write_heads(HN, start, end, [&](u32 p) { const u32 i = p - start; return ((hbits[i >> 5] >> (i & 31u)) & 1u) != 0u; });
    K1F_STAMP(4);
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
size_t k1_front_tilehist_words(const BatchGeom& g) { return (size_t)k1f_ptiles(g) * K1F_NB; }

int k1_front_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream) {
    const u32 ptiles = k1f_ptiles(g);
    const u32 nb8 = (g.nb + 7u) & ~7u;
    hipLaunchKernelGGL(k1f_sample, dim3(g.nb), dim3(1024), 0, stream, B, g);
    hipLaunchKernelGGL(k1f_hist, dim3(ptiles, g.nb), dim3(1024), 0, stream, B, g, ptiles);
    hipLaunchKernelGGL(k1f_scan, dim3(g.nb), dim3(1024), 0, stream, B, g, ptiles);
    hipLaunchKernelGGL(k1f_scatter, dim3(ptiles, nb8), dim3(1024), 0, stream, B, g, ptiles);
    K1Prof* pr = B.prof;
    const u32 slot = pr && pr->enabled ? __atomic_fetch_add(&pr->used, 1u, __ATOMIC_RELAXED) : K1_PROF_MAX;
    const bool timed = slot < K1_PROF_MAX;
    if (timed) (void)hipEventRecord(pr->ev[2 * slot], stream);
    hipLaunchKernelGGL(k1f_bsort, dim3(K1F_NB, nb8), dim3(K1F_BT), 0, stream, B, g);
    if (timed) {
        (void)hipEventRecord(pr->ev[2 * slot + 1], stream);
        __atomic_fetch_add(&pr->elements, (u64)g.nb * max_n, __ATOMIC_RELAXED);
    }
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
