// K1: batched cyclic suffix sort + BWT for gfx950 (wave64).
//
// Replaces BWT.bwtransform2 (lib/BWT.js:372-417), i.e. SA-IS on the doubled block plus the
// gather of lib/BWT.js:407-414.  The reference's order is: rotations of the block sorted as
// unsigned bytes, equal rotations by DESCENDING start index.  Any algorithm that
// realises this total order yields identical bytes, so this file does not port SA-IS.  It runs,
// for all blocks of a batch at once:
//
//   1. LSD radix sort of rotation indices by their first 7 (CJS_SORT_BYTES = 6..8) bytes (stable 8-bit passes;
//      per-tile LDS histograms, wave-ballot ranking, bucket scatter).
//   1b. K1-deep (cyclic mode): the groups of text-like input are mostly tiny and tie for tens of bytes; they are
//      resolved by comparing the TEXT -- 8 bytes per in-LDS iteration while groups above 8 rotations are being
//      worked on (k1_deep), then one lane per group (k1_deep_pairs, k1_deep_small).  No ranks involved; what is
//      left (long repeats, identical rotations, big groups) goes on to step 2, and if nothing is left
//      (k1_count_unsorted) steps 2 and 3 are skipped.
//   2. Group refinement by prefix doubling (Larsson-Sadakane style, cyclic): positions of the
//      suffix array that still tie form "groups" marked in a head bitmap; each round sorts every
//      unsorted group by the rank of the rotation h positions ahead.  Groups of <= 2048 rotations
//      are sorted inside LDS by the workgroup owning their first position; larger groups take a
//      one-workgroup segmented radix sort through global memory.
//   3. When h >= n the remaining ties are identical rotations: one more round keyed on the
//      descending start index.
//   4. U[j] = T[SA[j]-1 mod n], origPtr = position of rotation 0.
//
// All integer; no floating point anywhere.
#include "k1_bwt.h"
#include <stdio.h>
#include "devutil.h"
#include <stdlib.h>
#include <vector>

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 load_key8(const u8* p) {
    u64 k = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) k = (k << 8) | p[i];
    return k;
}

// ---------------------------------------------------------------------------------------------
// init: stats, head bitmaps, tile flags
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1_init(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y;
    const u32 n = B.nlen[b];
    const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 && gid < K1_STATS) B.stats[gid] = 0;
    if (b == 0) for (u32 i = gid; i < 4u * 8u * K1_DEEP_SUB; i += gridDim.x * blockDim.x) B.deepCnt[i] = 0;
    if (b == 0 && gid < 4u * K1_DM_SUB) B.dmCnt[gid] = 0;
    if (b == 0) for (u32 i = gid; i < 32u * 2u * K1_SPREAD; i += gridDim.x * blockDim.x) B.spread[i] = 0;
    if (b == 0) for (u32 i = gid; i < (K1R_MAXR + 1u) * B.rstride; i += gridDim.x * blockDim.x) B.rcnt[i] = 0;
    if (b == 0 && gid < K1F_LEVELS) B.bcnt[gid] = 0;
    if (b == 0) for (u32 i = gid; i < 2u * (K1D_MAXR + 2u) * B.rstride + B.rstride + (K1D_MAXR + 2u) * 4u; i += gridDim.x * blockDim.x) B.dcnt[i] = 0;   // dcnt, dchg, dtot, dbn (contiguous)
    if (gid < g.hstride) {
        const u32 lo = gid * 32u;
        u32 w;
        if (lo >= n) w = 0xFFFFFFFFu;
        else if (lo + 32u > n) w = 0xFFFFFFFFu << (n - lo);
        else w = 0u;
        B.HN[(size_t)b * g.hstride + gid] = w;
        if (gid == 0) w |= 1u;
        B.HC[(size_t)b * g.hstride + gid] = w;
    }
    if (gid < g.htiles) {
        B.FC[(size_t)b * g.htiles + gid] = (gid * K1_HT < n) ? 3 : 0;
        B.FN[(size_t)b * g.htiles + gid] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// initial sort by the first 8 bytes: LSD radix over (key32, index) pairs, 8-bit digits.
//   stage 1 (passes 0-3): key = bytes 4..7 of the rotation, computed from T with coalesced loads;
//   the scatter of pass 3 re-keys every element with bytes 0..3 (the only random gather of T);
//   stage 2 (passes 4-7): the same four digit passes on the new key.
// Each pass = k1_hist (per-tile digit counts) -> k1_scan -> k1_scatter.  The scatter ranks
// elements with wave ballots (stable), stages the tile in LDS in digit order and writes each
// digit's run to global memory with consecutive lanes on consecutive addresses.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 load_be32(const u8* p) {
    return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
}

// Sort tile: K1_SW waves x K1_SI elements per lane per workgroup (independent of the K2/K5 tile K1_RT).
#ifndef K1_SW
#define K1_SW 8
#endif
#ifndef K1_SI
#define K1_SI 6       // measured (10^8 B text, ms/step): 4x16 16.72, 8x8 16.17, 8x6 16.13, 8x4 16.49, 16x4 16.71
#endif
#define K1_SWE (K1_SI * 64)          // elements per wave
#define K1_ST (K1_SW * K1_SWE)
#define K1_STH (K1_SW * 64)
static inline u32 k1_stiles(const BatchGeom& g) { return (g.stride + K1_ST - 1) / K1_ST; }

// PMC (profiles/r01_pmc_lds_v6.csv): with one table per wave the LDS atomics of this kernel spend 4.3x
// their active cycles on bank conflicts (text digits are skewed: a few byte values carry most of the
// mass).  K1_HREP interleaved copies per wave (copy = lane % K1_HREP, word = bin * K1_HREP + copy) put
// equal digits of neighbouring lanes on different banks.
#ifndef K1_HREP
#define K1_HREP 2       // measured ms per step: 1 copy 15.98, 2 copies 15.82, 4 copies 15.91, 8 copies 17.13 (LDS, zeroing)
#endif
template <bool FIRST>
__global__ __launch_bounds__(K1_STH) void k1_hist(K1Buf B, BatchGeom g, const u32* keys, int shift, u32 stiles) {
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 t0 = t * K1_ST;
    if (t0 >= n) return;
    __shared__ u32 wh[K1_SW][256 * K1_HREP];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, cp = lane % K1_HREP;
    for (u32 i = tid; i < K1_SW * 256 * K1_HREP; i += K1_STH) (&wh[0][0])[i] = 0;
    __syncthreads();
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* kb = keys + (size_t)b * g.stride;
#pragma unroll
    for (int it = 0; it < K1_SI; it++) {
        const u32 j = t0 + w * K1_SWE + it * 64u + lane;
        if (j < n) {
            const u32 key = FIRST ? load_be32(T + j + 4) : kb[j];
            atomicAdd(&wh[w][((key >> shift) & 255u) * K1_HREP + cp], 1u);
        }
    }
    __syncthreads();
    if (tid < 256) {
        u32 sum = 0;
#pragma unroll
        for (int ww = 0; ww < K1_SW; ww++)
#pragma unroll
            for (int c = 0; c < K1_HREP; c++) sum += wh[ww][tid * K1_HREP + c];
        B.tileHist[((size_t)b * stiles + t) * 256 + tid] = sum;
    }
}

// per block: turn per-tile digit counts into global start offsets (digit-major, tile-minor)
__global__ __launch_bounds__(1024) void k1_scan(K1Buf B, BatchGeom g, u32 stiles) {
    const u32 b = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 nt = (n + K1_ST - 1) / K1_ST;
    __shared__ u32 part[4][256];
    __shared__ u32 sh[256];
    const u32 tid = threadIdx.x, q = tid >> 8, d = tid & 255u;
    const u32 per = (nt + 3) / 4;
    const u32 tlo = q * per < nt ? q * per : nt;
    const u32 thi = tlo + per < nt ? tlo + per : nt;
    u32* hist = B.tileHist + (size_t)b * stiles * 256;
    u32 sum = 0;
#pragma unroll 8
    for (u32 t = tlo; t < thi; t++) sum += hist[(size_t)t * 256 + d];
    part[q][d] = sum;
    __syncthreads();
    const u32 tot = part[0][d] + part[1][d] + part[2][d] + part[3][d];
    const u32 excl = block_excl_scan_256(tot, sh);     // threads >= 256 pass a dummy copy; only tid<256 lands in sh
    __shared__ u32 dbase[256];
    if (tid < 256) dbase[tid] = excl;
    __syncthreads();
    u32 run = dbase[d];
    for (u32 qq = 0; qq < q; qq++) run += part[qq][d];
    for (u32 t = tlo; t < thi; t++) {
        const u32 c = hist[(size_t)t * 256 + d];
        hist[(size_t)t * 256 + d] = run;
        run += c;
    }
}

// dynamic LDS: lk[K1_ST], lv[K1_ST]
template <bool FIRST, bool REKEY>
__global__ __launch_bounds__(K1_STH) void k1_scatter(K1Buf B, BatchGeom g, const u32* kin, const u32* vin, u32* kout,
                                                     u32* vout, int shift, u32 stiles) {
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 t0 = t * K1_ST;
    if (t0 >= n) return;
    __shared__ u32 wh[K1_SW][256];
    __shared__ u32 dstart[256], gbase[256], sh[256];
    HIP_DYNAMIC_SHARED(u32, k1_dyn)
    u32* lk = k1_dyn;
    u32* lv = k1_dyn + K1_ST;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    for (u32 i = tid; i < K1_SW * 256; i += K1_STH) (&wh[0][0])[i] = 0;
    __syncthreads();
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* kb = kin + (size_t)b * g.stride;
    const u32* vb = vin + (size_t)b * g.stride;
    // One ranking pass: every lane learns its rank among the elements of its wave with the same
    // digit (elements of earlier iterations first), and the wave's per-digit counts fall out of the
    // same ballots -- no LDS atomics (text digits collide 10-way and more).
    u32 kv[K1_SI], vv[K1_SI], rk[K1_SI];
    const u64 lt = lanemask_lt();
#pragma unroll
    for (int it = 0; it < K1_SI; it++) {
        const u32 j = t0 + w * K1_SWE + it * 64u + lane;
        const bool valid = j < n;
        u32 key = 0, val = 0;
        if (valid) {
            key = FIRST ? load_be32(T + j + 4) : kb[j];
            val = FIRST ? j : vb[j];
        }
        const u32 d = (key >> shift) & 255u;
        const u64 m = match_any(d, 8, valid);
        const u32 rank = (u32)__popcll(m & lt), cnt = (u32)__popcll(m);
        const u32 prior = valid ? wh[w][d] : 0u;              // same digit, earlier iterations of this wave
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wh[w][d] = prior + cnt;
        __builtin_amdgcn_wave_barrier();
        kv[it] = key;
        vv[it] = val;
        rk[it] = prior + rank;
    }
    __syncthreads();
    u32 total = 0;
    if (tid < 256) {
        u32 o = 0;
#pragma unroll
        for (int ww = 0; ww < K1_SW; ww++) {
            const u32 c = wh[ww][tid];
            wh[ww][tid] = o;                      // offset of wave ww inside digit `tid` of this tile
            o += c;
        }
        total = o;
        gbase[tid] = B.tileHist[((size_t)b * stiles + t) * 256 + tid];
    }
    const u32 ds = block_excl_scan_256(total, sh);
    if (tid < 256) dstart[tid] = ds;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < K1_SI; it++) {
        const u32 j = t0 + w * K1_SWE + it * 64u + lane;
        if (j < n) {
            const u32 d = (kv[it] >> shift) & 255u;
            const u32 lp = dstart[d] + wh[w][d] + rk[it];
            lk[lp] = kv[it];
            lv[lp] = vv[it];
        }
    }
    __syncthreads();
    const u32 cntv = n - t0 < K1_ST ? n - t0 : K1_ST;
    u32* ko = kout + (size_t)b * g.stride;
    u32* vo = vout + (size_t)b * g.stride;
    for (u32 i = tid; i < cntv; i += K1_STH) {
        u32 key = lk[i];
        const u32 val = lv[i];
        const u32 d = (key >> shift) & 255u;
        const u32 gp = gbase[d] + (i - dstart[d]);
        if (REKEY) key = load_be32(T + val);
        ko[gp] = key;
        vo[gp] = val;
    }
}

// ---------------------------------------------------------------------------------------------
// group heads after the 8-byte sort
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1_init_heads(K1Buf B, BatchGeom g, u32 lomask) {
    u32 b, t;
    if (!xcd_block_tile(g.nb, b, t)) return;
    const u32 n = B.nlen[b];
    const u32 base = t * K1_HT;
    if (base >= n) return;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* SA = B.SA + (size_t)b * g.stride;
    const u32* KH = B.KA + (size_t)b * g.stride;     // bytes 0..3 of every rotation, in SA order
    u32* HN = B.HN + (size_t)b * g.hstride;
    for (int it = 0; it < K1_HT / 256; it++) {
        const u32 p = base + w * (K1_HT / 4u) + it * 64u + lane;
        u32 hi = 0, lo = 0;
        if (p < n) { hi = KH[p]; lo = load_be32(T + SA[p] + 4); }
        u32 phi = __shfl_up(hi, 1u), plo = __shfl_up(lo, 1u);
        if (lane == 0 && p > 0 && p < n) { phi = KH[p - 1]; plo = load_be32(T + SA[p - 1] + 4); }
        bool head = true;
        if (p < n && p > 0) head = (hi != phi) || (((lo ^ plo) & lomask) != 0u);
        const u64 bal = __ballot(head);
        if (lane == 0) {
            HN[(p >> 5)] = (u32)bal;
            HN[(p >> 5) + 1] = (u32)(bal >> 32);
        }
    }
}

__device__ __forceinline__ u64 sp_desc(u32 b, u32 start, u32 len) {
    return ((u64)b << 52) | ((u64)start << 26) | (u64)len;
}
#define SP_B(d) ((u32)((d) >> 52))
#define SP_START(d) ((u32)(((d) >> 26) & 0x3FFFFFFu))
#define SP_LEN(d) ((u32)((d) & 0x3FFFFFFu))
#define SP_TINY 8u


// Descriptors of the unsorted groups that START in tile [base, base+K1_HT) of block b, read from
// a head bitmap whose words for the tile (+ `nwords` in total) are in LDS and whose full copy is
// `Hglob`.  Staged per size class in LDS, then appended to the parity-0 lists with one global
// atomic per class and workgroup.  Every thread of the workgroup must call it.
__device__ __forceinline__ void emit_group_descriptors(const K1Buf& B, const BatchGeom& g, u32 b, u32 base, u32 n,
                                                       const u32* hwords, u32 nwords, const u32* Hglob, u32 classmask = 0xFu) {
    __shared__ u64 stT[K1_HT / 2], stS[K1_HT / 8], stM[K1_HT / 64 + 1], stL[4];
    __shared__ u32 cntc[4], basec[4];
    __shared__ u32 medrot;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    if (tid < 4) cntc[tid] = 0;
    if (tid == 0) medrot = 0;
    __syncthreads();
    for (int it = 0; it < K1_HT / 256; it++) {
        const u32 q0 = w * (K1_HT / 4u) + it * 64u;
        const u32 wi = q0 >> 5;
        const u64 m64 = (u64)hwords[wi] | ((u64)hwords[wi + 1] << 32);
        const u64 nx64 = (m64 >> 1) | ((u64)(hwords[wi + 2] & 1u) << 63);
        const u64 starts64 = m64 & ~nx64;                            // head whose successor is not a head
        if (starts64 == 0) continue;                                  // wave-uniform
        const u32 q = q0 + lane;
        const bool starts = ((starts64 >> lane) & 1u) && base + q < n;
        // end of the group = next head after q: first in the LDS words, else in the global bitmap
        u32 endq = 0;
        bool found = false;
        if (starts) {
            u32 wq = q >> 5;
            u32 mm = (q & 31u) == 31u ? 0u : (hwords[wq] & (0xFFFFFFFEu << (q & 31u)));
            while (!mm && ++wq < nwords) mm = hwords[wq];
            if (mm) { endq = wq * 32u + (u32)__ffs((int)mm) - 1u; found = true; }
        }
        u64 far = __ballot(starts && !found);
        while (far) {                                                 // rare: a group longer than the LDS window
            const int src = __ffsll((long long)far) - 1;
            far &= far - 1;
            const u32 w0 = (base >> 5) + nwords;
            u32 endg = 0;
            bool got = false;
            for (u32 it2 = 0; !got; it2++) {
                const u32 gw = w0 + lane + 64u * it2;
                const u32 wd = gw < g.hstride ? Hglob[gw] : 0xFFFFFFFFu;
                const u64 bal = __ballot(wd != 0u);
                if (bal) {
                    const int fl = __ffsll((long long)bal) - 1;
                    endg = __shfl(gw * 32u + (u32)__ffs((int)wd) - 1u, fl);
                    got = true;
                }
            }
            if ((int)lane == src) { endq = endg - base; found = true; }
        }
        if (starts) {
            const u32 len = endq - q;
            const u64 d = sp_desc(b, base + q, len);
            if (len <= 8u) { if (classmask & 1u) stT[atomicAdd(&cntc[0], 1u)] = d; }
            else if (len <= 64u) { if (classmask & 2u) stS[atomicAdd(&cntc[1], 1u)] = d; }
            else if (len <= K1_MED_MAX) { if (classmask & 4u) { stM[atomicAdd(&cntc[2], 1u)] = d; atomicAdd(&medrot, len); } }
            else if (classmask & 8u) { const u32 li = atomicAdd(&cntc[3], 1u); if (li < 4u) stL[li] = d; }
        }
    }
    __syncthreads();
    if (tid < 4 && cntc[tid]) basec[tid] = atomicAdd(&B.stats[K1_STAT_LIST + tid], cntc[tid]);
    if (tid == 0 && medrot) atomicAdd(&B.stats[K1_STAT_MEDROT], medrot);
    __syncthreads();
    for (u32 i = tid; i < cntc[0]; i += 256) if (basec[0] + i < B.listTCap) B.listT[0][basec[0] + i] = stT[i];
    for (u32 i = tid; i < cntc[1]; i += 256) if (basec[1] + i < B.listSCap) B.listS[0][basec[1] + i] = stS[i];
    for (u32 i = tid; i < cntc[2]; i += 256) if (basec[2] + i < B.listMCap) B.listM[0][basec[2] + i] = stM[i];
    for (u32 i = tid; i < cntc[3] && i < 4u; i += 256) if (basec[3] + i < B.listLCap) B.listL[0][basec[3] + i] = stL[i];
}

// ---------------------------------------------------------------------------------------------
// rank update: ISA[SA[p]] = position of p's group head under HN, for every p that was in an
// unsorted group under HC.  Also produces next round's tile flags and active-group count.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void update_ranks_tile(const K1Buf& B, const BatchGeom& g, int slot_out, u32 b, u32 t, int emit) {
    const u32 n = B.nlen[b];
    const u32 base = t * K1_HT;
    if (base >= n) return;
    const size_t fidx = (size_t)b * g.htiles + t;
    if (!(B.FC[fidx] & 2)) {
        if (threadIdx.x == 0) B.FN[fidx] = 0;
        return;
    }
    __shared__ u32 hc[K1_HT / 32 + 4], hn[K1_HT / 32 + 4];
    __shared__ int prevh[64];
    __shared__ int inHead;
    __shared__ u32 inHeadOld;                             // the head before the tile was a head in the previous round too
    __shared__ u32 red[2];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32* HC = B.HC + (size_t)b * g.hstride;
    const u32* HN = B.HN + (size_t)b * g.hstride;
    if (tid < K1_HT / 32 + 4) {
        hc[tid] = HC[(base >> 5) + tid];
        hn[tid] = HN[(base >> 5) + tid];
    }
    if (tid == 0) { red[0] = 0; red[1] = 0; }
    // suffix indices of the tile, fetched while the bitmap words are on their way
    const u32* SA = B.SA + (size_t)b * g.stride;
    u32 pre_s[K1_HT / 256];
    __syncthreads();
    // suffix indices only of the 64-position chunks that hold a position of an unsorted group: late rounds
    // touch a few percent of them (the bitmap words are in LDS now; 8 workgroups per CU hide the extra hop)
#pragma unroll
    for (int it = 0; it < K1_HT / 256; it++) {
        const u32 q0 = w * (K1_HT / 4u) + (u32)it * 64u;
        const u32 wi = q0 >> 5;
        const u64 c64 = (u64)hc[wi] | ((u64)hc[wi + 1] << 32);
        const u64 cnx = (c64 >> 1) | ((u64)(hc[wi + 2] & 1u) << 63);
        const u32 p = base + q0 + lane;
        pre_s[it] = (~(c64 & cnx)) != 0 && p < n ? SA[p] : 0u;
    }
    if (w == 0) {
        const u32 word = hn[lane];
        int v = word ? (int)(lane * 32u + 31u - (u32)__clz((int)word)) : -1;
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(v, (unsigned)off);
            if ((int)lane >= off) v = v > u ? v : u;
        }
        int ex = __shfl_up(v, 1u);
        if (lane == 0) ex = -1;
        prevh[lane] = ex;
        int found = 0;
        if (!(hn[0] & 1u)) {
            found = -1;
            for (int iter = 0; found < 0; iter++) {
                const int wi = (int)(base >> 5) - 1 - (int)lane - 64 * iter;
                const u32 wd = wi >= 0 ? HN[wi] : 0u;
                const u64 bal = __ballot(wd != 0u);
                if (bal) {
                    const int src = __ffsll((long long)bal) - 1;
                    const int pos = wi * 32 + 31 - __clz((int)wd);
                    found = __shfl(pos, src);
                }
            }
        }
        if (lane == 0) {
            inHead = found;
            inHeadOld = found >= 0 ? (HC[(u32)found >> 5] >> ((u32)found & 31u)) & 1u : 1u;
        }
    }
    __syncthreads();
    u32* ISA = B.ISA + (size_t)b * g.stride;
    u32 nstart = 0, nact = 0;
#pragma unroll
    for (int it = 0; it < K1_HT / 256; it++) {
        const u32 q0 = w * (K1_HT / 4u) + it * 64u;
        // 64 positions at once (wave-uniform): heads of this chunk and of the positions after them
        const u32 wi = q0 >> 5;
        const u64 c64 = (u64)hc[wi] | ((u64)hc[wi + 1] << 32);
        const u64 cnx = (c64 >> 1) | ((u64)(hc[wi + 2] & 1u) << 63);
        const u64 n64 = (u64)hn[wi] | ((u64)hn[wi + 1] << 32);
        const u64 nnx = (n64 >> 1) | ((u64)(hn[wi + 2] & 1u) << 63);
        // bits of positions >= n are all set, so they never count as unsorted
        if (lane == 0) {
            nact += (u32)__popcll(~(n64 & nnx));
            nstart += (u32)__popcll(n64 & ~nnx);
        }
        const u64 actc = ~(c64 & cnx);
        if (actc == 0) continue;                                      // wave-uniform
        const u32 q = q0 + lane;
        if ((actc >> lane) & 1u) {
            // Heads are only ever added.  If the head this position now belongs to was a head in the previous round as well,
            // every member of the old group already carries it as its rank (whichever member sits here now): nothing to
            // write - the scattered 4-byte stores are what this kernel waits for (one write request each).  The pass before
            // the first round (slot_out 0) builds the array and writes everything.
            const u32 wq = q >> 5;
            const u32 mask = hn[wq] & (0xFFFFFFFFu >> (31u - (q & 31u)));
            u32 r, old;
            if (mask) {
                const u32 bit = 31u - (u32)__clz((int)mask);
                r = base + wq * 32u + bit;
                old = (hc[wq] >> bit) & 1u;
            } else if (prevh[wq] >= 0) {
                const u32 pr = (u32)prevh[wq];
                r = base + pr;
                old = (hc[pr >> 5] >> (pr & 31u)) & 1u;
            } else {
                r = (u32)inHead;
                old = inHeadOld;
            }
            if (slot_out == 0 || !old) ISA[pre_s[it]] = r;
        }
    }
    if (nact) atomicAdd(&red[1], nact);
    if (nstart) atomicAdd(&red[0], nstart);
    __syncthreads();
    if (tid == 0) {
        B.FN[fidx] = (u8)((red[0] ? 1 : 0) | (red[1] ? 2 : 0));
        const u32 sp = (t * 29u + b) & (K1_SPREAD - 1u);
        if (red[0]) atomicAdd(&B.spread[((size_t)slot_out * 2 + 0) * K1_SPREAD + sp], red[0]);
        if (red[1]) atomicAdd(&B.spread[((size_t)slot_out * 2 + 1) * K1_SPREAD + sp], red[1]);
    }
    // descriptor lists for a possible switch to the sparse phase after this round (the tile's
    // bitmap words are already in LDS; a separate pass over the bitmaps cost 0.9 ms)
    if (emit && red[0]) emit_group_descriptors(B, g, b, base, n, hn, K1_HT / 32 + 4, HN);
}

// (a persistent 8-per-CU grid walking the tiles was measured 2x SLOWER than one workgroup per
// tile for k1_refine and 12 % slower here: per-tile cost varies too much for static striding)
// K1_UPT tiles per workgroup: the refinement tile is small for occupancy in k1_refine, but this
// kernel is dispatch-bound in the later (sparse) rounds, so it walks several tiles per workgroup.
#define K1_UPT 1   /* measured: 1 tile per workgroup is fastest (latency-bound, wants parallelism) */
__global__ __launch_bounds__(256) void k1_update_ranks(K1Buf B, BatchGeom g, int slot_out, int emit) {
    for (u32 k = 0; k < K1_UPT; k++) {
        const u32 t = blockIdx.x * K1_UPT + k;
        if (t < g.htiles) update_ranks_tile(B, g, slot_out, blockIdx.y, t, emit);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// refinement of groups of <= K1_HT rotations, in LDS
// mode 0: key = ISA[(s + h) mod n]        mode 1: key = n - 1 - s (identical rotations)
// ---------------------------------------------------------------------------------------------
#define K1_INF (1 << 30)

struct PosClass {
    int head;   // window-relative position of the group head (-1: before the window)
    int endp;   // window-relative position of the next head (K1_INF: beyond the window)
    bool is_head;
};

__device__ __forceinline__ PosClass classify(const u32* hw, const int* prevh, const int* nexth, u32 q) {
    PosClass c;
    const u32 wq = q >> 5, bq = q & 31u;
    const u32 word = hw[wq];
    const u32 low = word & (0xFFFFFFFFu >> (31u - bq));
    c.head = low ? (int)(wq * 32u + 31u - (u32)__clz((int)low)) : prevh[wq];
    const u32 high = bq == 31u ? 0u : (word & (0xFFFFFFFEu << bq));
    c.endp = high ? (int)(wq * 32u + (u32)__ffs((int)high) - 1u) : nexth[wq];
    c.is_head = (word >> bq) & 1u;
    return c;
}

// Sort key of rotation/suffix s in a doubling round.
//   cyclic (bzip2, BWT.bwtransform2): rank of the rotation h positions ahead, indices wrap;
//   linear (BWT.suffixsort / bwtransform, implicit smallest sentinel): past the end sorts first,
//     real ranks are shifted by one;
//   mode 1 (last round, both): descending start index -- identical rotations, and in linear mode
//     suffixes that ran into the zero padding with equal bytes, where the shorter one (larger start)
//     is a proper prefix of the longer and therefore smaller.
__device__ __forceinline__ u32 rot_key(const u32* ISA, u32 n, u32 s, u32 h, u32 hm, int mode, u32 linear) {
    if (mode) return n - 1u - s;
    if (linear) {
        const u64 x = (u64)s + h;
        return x >= n ? 0u : ISA[x] + 1u;
    }
    u32 x = s + hm;                     // hm = h mod n, hoisted by the caller
    if (x >= n) x -= n;
    return ISA[x];
}

// true when none of the 64 positions starting at window position q0 (a multiple of 64) belongs to
// an unsorted group: every position is a head and so is its successor.  Wave-uniform.
__device__ __forceinline__ bool chunk_all_sorted(const u32* hw, u32 q0) {
    const u32 wi = q0 >> 5;
    const u64 m = (u64)hw[wi] | ((u64)hw[wi + 1] << 32);
    const u64 nx = (m >> 1) | ((u64)(hw[wi + 2] & 1u) << 63);
    return (m & nx) == ~0ull;
}

// ascending compare-exchange of LDS pairs (key, value)
__device__ __forceinline__ void cmpx(u32* ck, u32* cv, u32 lo, u32 hi) {
    const u32 a = ck[lo], c2 = ck[hi];
    if (a > c2) {
        ck[lo] = c2; ck[hi] = a;
        const u32 va = cv[lo]; cv[lo] = cv[hi]; cv[hi] = va;
    }
}

#define K1_SMALL 64        // groups up to this size: enumeration sort; larger (<= K1_HT): bitonic

__device__ __forceinline__ void refine_tile(const K1Buf& B, const BatchGeom& g, u32 h, int mode, int round, u32 b, u32 t) {
    const u32 n = B.nlen[b];
    const u32 base = t * K1_HT;
    if (base >= n) return;
    if (!(B.FC[(size_t)b * g.htiles + t] & 1)) return;
    __shared__ u32 hw[K1_WW + 2];
    __shared__ int prevh[K1_WW + 2], nexth[K1_WW + 2];
    __shared__ u32 ck[K1_WIN], cv[K1_WIN];
    __shared__ u16 cp[K1_WIN], csz[K1_WIN];
    __shared__ u32 chunkoff[K1_WIN / 64 + 1];
    __shared__ u32 biglist[64];
    __shared__ u32 nbig;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32* HC = B.HC + (size_t)b * g.hstride;
    u32* HN = B.HN + (size_t)b * g.hstride;
    u32* SA = B.SA + (size_t)b * g.stride;
    const u32* ISA = B.ISA + (size_t)b * g.stride;
    const u32 wbase = base >> 5;
    if (tid < K1_WW) hw[tid] = HC[wbase + tid];
    if (tid == 0) nbig = 0;
    // The 64-position chunks of the window are dealt round-robin to the 4 waves (chunk ci = it*4+w)
    // so that all waves share the own half.  For those chunks the suffix index and its round key
    // are fetched NOW -- two dependent HBM/L2 round trips that overlap the bitmap work below
    // (in the dense first round nearly every own position is in an unsorted group).
    const u32 hm = h % n;
    u32 pre_s[K1_HT / 256], pre_k[K1_HT / 256];
#pragma unroll
    for (int it = 0; it < K1_HT / 256; it++) {
        const u32 q = ((u32)it * 4u + w) * 64u + lane;
        pre_s[it] = base + q < n ? SA[base + q] : 0u;
    }
#pragma unroll
    for (int it = 0; it < K1_HT / 256; it++) pre_k[it] = rot_key(ISA, n, pre_s[it], h, hm, mode, B.linear);
    __syncthreads();
    if (tid < K1_WW) {
        int pv = -1;
        for (int i = (int)tid - 1; i >= 0; i--) {
            const u32 wd = hw[i];
            if (wd) { pv = i * 32 + 31 - __clz((int)wd); break; }
        }
        prevh[tid] = pv;
        int nx = K1_INF;
        for (int i = (int)tid + 1; i < K1_WW; i++) {
            const u32 wd = hw[i];
            if (wd) { nx = i * 32 + __ffs((int)wd) - 1; break; }
        }
        nexth[tid] = nx;
    }
    __syncthreads();
    const u64 lt = lanemask_lt();
    // positions at or after the first head of the spill half belong to groups that start there,
    // i.e. to the next tile: chunks beyond it need no look (wave-uniform bound)
    const u32 spill_end = (hw[K1_HT / 32] & 1u) ? (u32)K1_HT
                        : (nexth[K1_HT / 32 - 1] < K1_INF ? (u32)nexth[K1_HT / 32 - 1] : (u32)K1_WIN);
    // pass 1: owned positions per chunk, register large groups
    for (int it = 0; it < K1_WIN / 256; it++) {
        const u32 ci = (u32)it * 4u + w;
        const u32 q0 = ci * 64u;
        u32 c64 = 0;
        if (q0 < spill_end && !chunk_all_sorted(hw, q0)) {            // wave-uniform
            const u32 q = q0 + lane;
            const PosClass c = classify(hw, prevh, nexth, q);
            const int size = (c.head >= 0 && c.endp < K1_INF) ? c.endp - c.head : K1_INF;
            const bool owned = c.head >= 0 && c.head < K1_HT && size >= 2 && size <= K1_HT && base + q < n;
            if (c.is_head && q < K1_HT && base + q < n && size > K1_HT) {
                const u32 idx = atomicAdd(&B.stats[K1_STAT_LARGE + round], 1u);
                if (idx < B.largeCap) B.large[idx] = make_uint2(b, base + q);
            }
            c64 = (u32)__popcll(__ballot(owned));
        }
        if (lane == 0) chunkoff[ci] = c64;
    }
    __syncthreads();
    if (tid == 0) {                                                   // exclusive scan over the chunks
        u32 run = 0;
        for (u32 ci = 0; ci < K1_WIN / 64; ci++) { const u32 c = chunkoff[ci]; chunkoff[ci] = run; run += c; }
        chunkoff[K1_WIN / 64] = run;
    }
    __syncthreads();
    const u32 m = chunkoff[K1_WIN / 64];
    if (m == 0) return;
    // pass 2: keys of owned positions into the compact arrays
#pragma unroll
    for (int it = 0; it < K1_WIN / 256; it++) {
        const u32 ci = (u32)it * 4u + w;
        const u32 q0 = ci * 64u;
        if (q0 >= spill_end || chunk_all_sorted(hw, q0)) continue;    // wave-uniform
        const u32 q = q0 + lane;
        const PosClass c = classify(hw, prevh, nexth, q);
        const int size = (c.head >= 0 && c.endp < K1_INF) ? c.endp - c.head : K1_INF;
        const bool owned = c.head >= 0 && c.head < K1_HT && size >= 2 && size <= K1_HT && base + q < n;
        const u64 bal = __ballot(owned);
        if (owned) {
            const u32 e = chunkoff[ci] + (u32)__popcll(bal & lt);
            u32 s, k;
            if (it < K1_HT / 256) { s = pre_s[it]; k = pre_k[it]; }   // own half: prefetched (static index)
            else { s = SA[base + q]; k = rot_key(ISA, n, s, h, hm, mode, B.linear); }
            ck[e] = ((u32)c.head << 22) | k;                        // head < K1_HT = 2^10, key < n < 2^22
            cv[e] = s;
            cp[e] = (u16)q;
            csz[e] = (u16)size;
            if (c.is_head && size > K1_SMALL) {
                const u32 bi = atomicAdd(&nbig, 1u);
                biglist[bi] = e | ((u32)size << 16);
            }
        }
    }
    __syncthreads();
    {
        // enumeration sort inside every small group (<= K1_SMALL elements)
        u32 nk[K1_WIN / 256], nv[K1_WIN / 256], ns[K1_WIN / 256];
#pragma unroll
        for (int it = 0; it < K1_WIN / 256; it++) {
            const u32 e = tid + (u32)it * 256u;
            ns[it] = 0xFFFFFFFFu;
            if (e < m && csz[e] <= K1_SMALL) {
                const u32 key = ck[e];
                const u32 gs = e - ((u32)cp[e] - (key >> 22));
                const u32 ge = gs + csz[e];
                u32 r = 0;
                for (u32 j = gs; j < ge; j++) {
                    const u32 kj = ck[j];
                    r += (kj < key || (kj == key && j < e)) ? 1u : 0u;
                }
                nk[it] = key; nv[it] = cv[e]; ns[it] = gs + r;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < K1_WIN / 256; it++)
            if (ns[it] != 0xFFFFFFFFu) { ck[ns[it]] = nk[it]; cv[ns[it]] = nv[it]; }
        __syncthreads();
    }
    // bigger groups: in-place bitonic network for arbitrary length (flip + half-cleaners, all
    // ascending; pairs whose upper index falls beyond the group are skipped).  One WAVE per group:
    // the waves of the workgroup sort different groups concurrently and need no block barrier.
    const u32 nb = nbig;
    for (u32 gi = w; gi < nb; gi += 4) {
        const u32 e0 = biglist[gi] & 0xFFFFu, sz = biglist[gi] >> 16;
        u32* gk = ck + e0;
        u32* gv = cv + e0;
        u32 M = 128;
        while (M < sz) M <<= 1;
        for (u32 k = 2; k <= M; k <<= 1) {
            const u32 hk = k >> 1;
            for (u32 i = lane; i < (M >> 1); i += 64) {
                const u32 blk = i / hk, off = i - blk * hk;
                const u32 lo = blk * k + off, hi = blk * k + (k - 1u - off);
                if (hi < sz) cmpx(gk, gv, lo, hi);
            }
            __builtin_amdgcn_wave_barrier();
            for (u32 j = k >> 2; j > 0; j >>= 1) {
                for (u32 i = lane; i < (M >> 1); i += 64) {
                    const u32 lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u));
                    const u32 hi = lo | j;
                    if (hi < sz) cmpx(gk, gv, lo, hi);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    // write back + new heads
    for (u32 e = tid; e < m; e += 256) {
        const u32 q = cp[e];
        const u32 p = base + q;
        SA[p] = cv[e];
        const bool newhead = (e == 0) || (ck[e] != ck[e - 1]);
        const bool cur = (hw[q >> 5] >> (q & 31u)) & 1u;
        if (newhead && !cur) atomicOr(&HN[p >> 5], 1u << (p & 31u));
    }
}

__global__ __launch_bounds__(256) void k1_refine(K1Buf B, BatchGeom g, u32 h, int mode, int round) {
    u32 b, t;
    if (!xcd_block_tile(g.nb, b, t)) return;
    refine_tile(B, g, h, mode, round, b, t);
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
        if ((u32)i < gl) { u32 q = members[(u32)i * mstride] + dm; if (q >= n) q -= n; pp[i] = q; }
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

// DHT = suffix-array positions owned by one workgroup of DNT threads (window 2*DHT).  <1024, 256> mirrors
// k1_refine; <256, 64> is one WAVE per tile: its barriers are wave-local, so a tile whose groups tie for
// 20 iterations does not stall on three other waves 20 times, and 16 independent tiles per CU overlap
// their text loads.
template <int DHT, int DNT>
__global__ __launch_bounds__(DNT, 4) void k1_deep(K1Buf B, BatchGeom g, u32 iters, u32 dbg, u32 d0, u32 bigrot_max) {
    {   // predictor (decided on the device): with this many rotations in big 8-byte groups (HTML-like input) most ties
        // are long repeats that 264 bytes of text do not settle, and the stage only costs (E8S-A: 24.5 ms with it, 22.6 without)
        u32 bg = 0;
        for (u32 i = 0; i < 8u; i++) bg += B.stats[K1_STAT_BIGROT + i];
        if (bg > bigrot_max) return;
    }
    constexpr int DWIN = 2 * DHT, DWW = DWIN / 32 + 2, DCW = DWIN / 32, NW = DNT / 64, SL = DWIN / DNT;
    static_assert(DCW <= 64 && DWW <= DNT, "one wave scans the compact bitmap");
    u32 b, t;
    if (!xcd_block_tile(g.nb, b, t)) return;
    const u32 n = B.nlen[b];
    const u32 base = t * (u32)DHT;
    if (base >= n || n < 64u) return;
    __shared__ u32 hw[DWW + 2];
    __shared__ int prevh[DWW + 2], nexth[DWW + 2];
    __shared__ u64 ck[DWIN];
    __shared__ u32 cv[DWIN];
    __shared__ u16 cp[DWIN];
    __shared__ u32 hb[DCW + 2];
    __shared__ int cprev[DCW], cnext[DCW];
    __shared__ u32 chunkoff[DWIN / 64 + 1];
    __shared__ u32 anyact[2];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32* HX = B.HX + (size_t)b * g.hstride;
    u32* HN = B.HN + (size_t)b * g.hstride;
    u32* SA = B.SA + (size_t)b * g.stride;
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32 wbase = base >> 5;
    if (tid < DWW) hw[tid] = HX[wbase + tid];
    if (tid < DCW + 2) hb[tid] = tid < DCW ? 0u : 0xFFFFFFFFu;
    if (tid < 2) anyact[tid] = 0;
    __syncthreads();
    if constexpr (DWW <= 64) {
        // previous / next head outside each bitmap word by two wave scans (the tile kernel is issue-bound:
        // PMC SQ_ACTIVE_INST_ANY x 4 waves per SIMD ~ its wave cycles; the serial per-lane walks cost more)
        if (w == 0) {
            const u32 word = lane < (u32)DWW ? hw[lane] : 0u;
            int v = word ? (int)(lane * 32u + 31u - (u32)__clz((int)word)) : -1;
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(v, (unsigned)off);
                if ((int)lane >= off) v = v > u ? v : u;
            }
            int ex = __shfl_up(v, 1u);
            if (lane == 0) ex = -1;
            if (lane < (u32)DWW) prevh[lane] = ex;
            int f = word ? (int)(lane * 32u + (u32)__ffs((int)word) - 1u) : K1_INF;
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_down(f, (unsigned)off);
                if ((int)lane + off < 64) f = f < u ? f : u;
            }
            int nx = __shfl_down(f, 1u);
            if (lane == 63u) nx = K1_INF;
            if (lane < (u32)DWW) nexth[lane] = nx;
        }
    } else if (tid < DWW) {
        int pv = -1;
        for (int i = (int)tid - 1; i >= 0; i--) {
            const u32 wd = hw[i];
            if (wd) { pv = i * 32 + 31 - __clz((int)wd); break; }
        }
        prevh[tid] = pv;
        int nx = K1_INF;
        for (int i = (int)tid + 1; i < DWW; i++) {
            const u32 wd = hw[i];
            if (wd) { nx = i * 32 + __ffs((int)wd) - 1; break; }
        }
        nexth[tid] = nx;
    }
    __syncthreads();
    const u64 lt = lanemask_lt();
    const u32 spill_end = (hw[DHT / 32] & 1u) ? (u32)DHT
                        : (nexth[DHT / 32 - 1] < K1_INF ? (u32)nexth[DHT / 32 - 1] : (u32)DWIN);
    // pass 1: owned positions per 64-position chunk (ownership as in refine_tile); their suffix indices are
    // fetched here, all chunks in flight together
    u64 obal[SL];
    u32 osa[SL], ohead = 0;
#pragma unroll
    for (int it = 0; it < SL; it++) {
        const u32 ci = (u32)it * (u32)NW + w;
        const u32 q0 = ci * 64u;
        obal[it] = 0;
        osa[it] = 0;
        if (q0 < spill_end && !chunk_all_sorted(hw, q0)) {            // wave-uniform
            const u32 q = q0 + lane;
            const PosClass c = classify(hw, prevh, nexth, q);
            const int size = (c.head >= 0 && c.endp < K1_INF) ? c.endp - c.head : K1_INF;
            const bool owned = c.head >= 0 && c.head < DHT && size >= 2 && size <= DHT && base + q < n;
            obal[it] = __ballot(owned);
            if (owned) osa[it] = SA[base + q];
            if (c.is_head) ohead |= 1u << it;
        }
        if (lane == 0) chunkoff[ci] = (u32)__popcll(obal[it]);
    }
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (u32 ci = 0; ci < DWIN / 64; ci++) { const u32 c = chunkoff[ci]; chunkoff[ci] = run; run += c; }
        chunkoff[DWIN / 64] = run;
    }
    __syncthreads();
    const u32 m = chunkoff[DWIN / 64];
    if (m == 0) return;
    // pass 2: compact the owned rotations; head bits over the compact index
#pragma unroll
    for (int it = 0; it < SL; it++) {
        if (!((obal[it] >> lane) & 1ull)) continue;
        const u32 ci = (u32)it * (u32)NW + w;
        const u32 e = chunkoff[ci] + (u32)__popcll(obal[it] & lt);
        cv[e] = osa[it];
        cp[e] = (u16)(ci * 64u + lane);
        if ((ohead >> it) & 1u) atomicOr(&hb[e >> 5], 1u << (e & 31u));
    }
    if (tid < DCW) {                                                // compact slots >= m count as sorted
        const u32 lo = tid * 32u;
        if (lo + 32u > m) atomicOr(&hb[tid], lo >= m ? 0xFFFFFFFFu : 0xFFFFFFFFu << (m - lo));
    }
    __syncthreads();
    // word-level neighbours of the compact head bitmap (one wave: DCW <= 64 words)
    auto scan_words = [&]() {
        if (w == 0) {
            const u32 word = lane < (u32)DCW ? hb[lane] : 0xFFFFFFFFu;
            int v = word ? (int)(lane * 32u + 31u - (u32)__clz((int)word)) : -1;
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(v, (unsigned)off);
                if ((int)lane >= off) v = v > u ? v : u;
            }
            int ex = __shfl_up(v, 1u);
            if (lane == 0) ex = -1;
            if (lane < (u32)DCW) cprev[lane] = ex;
            int f = word ? (int)(lane * 32u + (u32)__ffs((int)word) - 1u) : 64 * 32;
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_down(f, (unsigned)off);
                if ((int)lane + off < 64) f = f < u ? f : u;
            }
            int nx = __shfl_down(f, 1u);
            if (lane == 63u) nx = 64 * 32;
            if (lane < (u32)DCW) cnext[lane] = nx;
        }
    };
    // ---- phase 1: all groups together, 8 bytes per iteration, while a group of more than K1_DEEP_LANE
    //      rotations is still being worked on (rank by counting inside the group: any size up to DHT)
    u32 iter = 0;
    for (; iter < iters && !(dbg & 1u); iter++) {
        scan_words();
        __syncthreads();
        const u32 dm = (d0 + 8u * iter) % n;
        u64 key[SL];
        u32 val[SL], gsl[SL];                     // group start | length << 16 (0: not active)
        bool big = false;
#pragma unroll
        for (int it = 0; it < SL; it++) {
            const u32 e0 = (u32)it * (u32)DNT + w * 64u;
            gsl[it] = 0;
            if (e0 >= m || chunk_all_sorted(hb, e0)) continue;        // wave-uniform
            const u32 e = e0 + lane;
            const u32 wq = e >> 5, bq = e & 31u;
            const u32 word = hb[wq];
            const u32 low = word & (0xFFFFFFFFu >> (31u - bq));
            const int head = low ? (int)(wq * 32u + 31u - (u32)__clz((int)low)) : cprev[wq];
            const u32 high = bq == 31u ? 0u : (word & (0xFFFFFFFEu << bq));
            const int endp = high ? (int)(wq * 32u + (u32)__ffs((int)high) - 1u) : cnext[wq];
            const u32 gl = (u32)(endp - head);
            if (e < m && head >= 0 && gl >= 2u && (gl <= K1_DEEP_BIG || iter < 2u)) {
                val[it] = cv[e];
                gsl[it] = (u32)head | (gl << 16);
                big = big || gl > K1_DEEP_LANE;
            }
        }
        if (big) anyact[iter & 1u] = 1u;
        __syncthreads();
        if (!anyact[iter & 1u]) break;                                // only lane-sized groups left (or none)
        if (tid == 0) anyact[(iter + 1u) & 1u] = 0u;
        // all text loads of the iteration in flight together, then the LDS stores
#pragma unroll
        for (int it = 0; it < SL; it++) {
            key[it] = 0;
            if (gsl[it]) {
                u32 p = val[it] + dm;
                if (p >= n) p -= n;
                key[it] = load_be64(T, p);
            }
        }
#pragma unroll
        for (int it = 0; it < SL; it++)
            if (gsl[it]) ck[(u32)it * (u32)DNT + tid] = key[it];
        __syncthreads();
        u32 ns[SL];
#pragma unroll
        for (int it = 0; it < SL; it++) {
            if (!gsl[it]) continue;
            const u32 e = (u32)it * (u32)DNT + tid;
            const u32 gs = gsl[it] & 0xFFFFu, ge = gs + (gsl[it] >> 16);
            const u64 k = key[it];
            u32 r = 0;
            for (u32 j = gs; j < ge; j++) {
                const u64 kj = ck[j];
                r += (kj < k || (kj == k && j < e)) ? 1u : 0u;
            }
            ns[it] = gs + r;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < SL; it++)
            if (gsl[it]) { ck[ns[it]] = key[it]; cv[ns[it]] = val[it]; }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < SL; it++) {
            const u32 e0 = (u32)it * (u32)DNT + w * 64u;
            if (e0 >= m) continue;                                    // wave-uniform
            const u32 e = e0 + lane;
            const bool nh = gsl[it] && e != (gsl[it] & 0xFFFFu) && ck[e] != ck[e - 1u];
            const u64 bal = __ballot(nh);
            if (bal && lane == 0) {
                if ((u32)bal) atomicOr(&hb[e0 >> 5], (u32)bal);
                if ((u32)(bal >> 32)) atomicOr(&hb[(e0 >> 5) + 1u], (u32)(bal >> 32));
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // write back: the suffix indices in their new order, and the heads that are new
    for (u32 e = tid; e < m; e += DNT) {
        const u32 q = cp[e];
        const u32 p = base + q;
        SA[p] = cv[e];
        const bool nowh = (hb[e >> 5] >> (e & 31u)) & 1u;
        const bool was = (hw[q >> 5] >> (q & 31u)) & 1u;
        if (nowh && !was) atomicOr(&HN[p >> 5], 1u << (p & 31u));
    }
    // ---- phase 2 is list driven (k1_deep_pairs / k1_deep_small): descriptors of the groups of 2..K1_DEEP_LANE
    //      rotations that are left, with the depth they are known to share, appended to the list region of
    //      this block's XCD (one atomic per wave and class).  cprev/cnext are those of the last scan_words().
    const u32 depth = d0 + 8u * iter;
    if (depth >= d0 + 8u * iters || (dbg & 2u)) return;
    __syncthreads();
    scan_words();
    __syncthreads();
    // 8 XCD regions x K1_DEEP_SUB sub-regions (chosen by the tile index) per class, each with its own counter: 10^5..10^6
    // waves appending to 16 counter words serialise on them (measured: +2.5 ms for this kernel)
    const u32 rcap = B.listTCap / (8u * K1_DEEP_SUB), xr = (b & 7u) * K1_DEEP_SUB + (t & (K1_DEEP_SUB - 1u));
    u64 balc[2][SL];
    u32 glv[SL];
    u32 tot0 = 0, tot1 = 0;
#pragma unroll
    for (int it = 0; it < SL; it++) {
        const u32 e0 = (u32)it * (u32)DNT + w * 64u;
        balc[0][it] = 0; balc[1][it] = 0; glv[it] = 0;
        if (e0 >= m || chunk_all_sorted(hb, e0)) continue;            // wave-uniform
        const u32 e = e0 + lane;
        const u32 wq = e >> 5, bq = e & 31u;
        const u32 word = hb[wq];
        const bool ishead = (word >> bq) & 1u;
        const u32 high = bq == 31u ? 0u : (word & (0xFFFFFFFEu << bq));
        const int endp = high ? (int)(wq * 32u + (u32)__ffs((int)high) - 1u) : cnext[wq];
        const u32 gl = (u32)endp - e;
        const bool take = ishead && e < m && gl >= 2u && gl <= K1_DEEP_LANE;
        glv[it] = take ? gl : 0u;
        balc[0][it] = __ballot(take && gl == 2u);
        balc[1][it] = __ballot(take && gl > 2u);
        tot0 += (u32)__popcll(balc[0][it]);
        tot1 += (u32)__popcll(balc[1][it]);
    }
    u32 gb0 = 0, gb1 = 0;
    if (lane == 0) {
        if (tot0) gb0 = atomicAdd(&B.deepCnt[xr], tot0);
        if (tot1) gb1 = atomicAdd(&B.deepCnt[8u * K1_DEEP_SUB + xr], tot1);
    }
    gb0 = __shfl(gb0, 0);
    gb1 = __shfl(gb1, 0);
#pragma unroll
    for (int it = 0; it < SL; it++) {
        const u32 e = (u32)it * (u32)DNT + tid;
        if (glv[it]) {
            const int cls = glv[it] == 2u ? 0 : 1;
            const u32 idx = (cls ? gb1 : gb0) + (u32)__popcll(balc[cls][it] & lt);
            if (idx < rcap)
                B.listT[cls][(size_t)xr * rcap + idx] = ((u64)b << 52) | ((u64)(base + cp[e]) << 26) | ((u64)depth << 4) | (u64)(glv[it] - 1u);
        }
        gb0 += (u32)__popcll(balc[0][it]);
        gb1 += (u32)__popcll(balc[1][it]);
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
__global__ __launch_bounds__(256) void k1_deep_pairs(K1Buf B, BatchGeom g, u32 capd, u32 limit) {
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
        const u32 n = B.nlen[b];
        const u8* T = B.T + (size_t)b * g.tstride;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u32 mem[2];
        u64 keys[2];
        mem[0] = SA[0];
        mem[1] = SA[1];
        if (deep_walk<2, 8>(T, n, mem, 1u, 2u, d, capd, keys, 1u)) {
            if (keys[0] > keys[1]) { SA[0] = mem[1]; SA[1] = mem[0]; }
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
__global__ __launch_bounds__(256) void k1_deep_small(K1Buf B, BatchGeom g, u32 capd, u32 limit) {
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
        const u32 n = B.nlen[b];
        const u8* T = B.T + (size_t)b * g.tstride;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u32 mem[K1_DEEP_LANE];
#pragma unroll
        for (int i = 0; i < (int)K1_DEEP_LANE; i++) mem[i] = (u32)i < gl ? SA[i] : 0u;     // all loads in flight together
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
        for (u32 i = 0; i < gl; i++) SA[i] = cv[i * 256u];
        const u64 bits = (u64)(heads & ~1u) << (start & 31u);
        u32* HN = B.HN + (size_t)b * g.hstride + (start >> 5);
        if ((u32)bits) atomicOr(&HN[0], (u32)bits);
        if ((u32)(bits >> 32)) atomicOr(&HN[1], (u32)(bits >> 32));
    }
}

// ---------------------------------------------------------------------------------------------
// How many rotations are still in unsorted groups under HN?  (position p is settled iff p and p + 1 are heads;
// bits at and beyond n are set.)  One bitmap word per thread; per-workgroup sums go to round slot 31 of `spread`.
// When the answer is 0 -- random data after the radix sort, phrase-reuse text after K1-deep -- the rank pass
// (10^8 random 4-byte stores, 1.1 ms) and every doubling round are skipped.
// ---------------------------------------------------------------------------------------------
#define K1_COUNT_SLOT 31
// ---------------------------------------------------------------------------------------------
// K1-deep, medium groups: groups of 9 .. K1_MED_MAX rotations that the tile kernel leaves (it ranks by counting,
// O(length^2), and therefore gives groups above 64 rotations two iterations and groups above 256 none) are refined
// by comparing the text as well, list driven: k1_emit_medium turns the head bitmap into descriptors, every
// k1_dm_round sorts each listed group by the 8 text bytes at the round's depth (one workgroup per group, bitonic
// network on 64-bit keys in LDS), marks the new heads, hands sub-groups of 2..8 rotations to the lane kernels'
// lists and sub-groups of 9 and more to the next round's list.  On the enwik8-shaped stream this is what was left
// after K1-deep (417 groups, 118 000 rotations per 10^8): with them resolved k1_count_unsorted finds nothing and the
// whole rank machinery (k1_update_ranks: 10^8 random stores, 1.1 ms) is skipped; on HTML-like input (E8S-A) the
// medium groups are 30 % of all rotations.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sp_append_class(u64* list, u32* counter, u32 cap, bool pred, u64 d);   // below (sparse phase)

__global__ __launch_bounds__(256) void k1_emit_medium(K1Buf B, BatchGeom g) {
    u32 b, t;
    if (!xcd_block_tile(g.nb, b, t)) return;
    const u32 n = B.nlen[b];
    const u32 base = t * K1_HT;
    if (base >= n) return;
    __shared__ u32 hn[K1_HT / 32 + 4];
    const u32* HN = B.HN + (size_t)b * g.hstride;
    if (threadIdx.x < K1_HT / 32 + 4) hn[threadIdx.x] = HN[(base >> 5) + threadIdx.x];
    __syncthreads();
    emit_group_descriptors(B, g, b, base, n, hn, K1_HT / 32 + 4, HN, 4u);        // 65 .. K1_MED_MAX only: smaller groups went through
                                                                                // the tile kernel's iterations: what is left of them ties beyond its cap
}

__device__ __forceinline__ void cmpx64(u64* ck, u32* cv, u32 lo, u32 hi) {
    const u64 a = ck[lo], c2 = ck[hi];
    if (a > c2) {
        ck[lo] = c2; ck[hi] = a;
        const u32 va = cv[lo]; cv[lo] = cv[hi]; cv[hi] = va;
    }
}

// groups of 9..64 rotations listed for this round: one WAVE each, members in lanes, ranked by counting with the candidates
// broadcast through v_readlane (as the leaves of k1f_bsort): equal counts = equal keys = one sub-group
__global__ __launch_bounds__(256) void k1_dm_round_small(K1Buf B, BatchGeom g, u32 depth, u32 medrot_max, int parity) {
    if (B.stats[K1_STAT_MEDROT] > medrot_max) return;      // see k1_dm_round
    const u32 tid = threadIdx.x, lane = tid & 63u;
    // wave W serves sub-list W % K1_DM_SUB (gridDim.x * 4 is a multiple of K1_DM_SUB) and appends to the same sub-list of the next round
    const u32 W = blockIdx.x * 4u + (tid >> 6), sub = W & (K1_DM_SUB - 1u), capS = B.listSCap / K1_DM_SUB;
    u32 cs = B.dmCnt[((u32)parity * 2u + 0u) * K1_DM_SUB + sub];
    if (cs > capS) cs = capS;
    const u64* Lin = B.listS[parity] + (size_t)sub * capS;
    u64* LoutS = B.listS[parity ^ 1] + (size_t)sub * capS;
    u32* coutS = B.dmCnt + (((u32)parity ^ 1u) * 2u + 0u) * K1_DM_SUB + sub;
    const u32 rcap = B.listTCap / (8u * K1_DEEP_SUB);
    const u64 lt = lanemask_lt();
    for (u32 gi = W / K1_DM_SUB; gi < cs; gi += gridDim.x * 4u / K1_DM_SUB) {           // wave-uniform
        const u64 d = Lin[gi];
        const u32 b = SP_B(d), start = SP_START(d), len = SP_LEN(d);
        const u32 n = B.nlen[b];
        const u8* T = B.T + (size_t)b * g.tstride;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        const u32 dm = depth % n;
        const bool valid = lane < len;
        const u32 s = valid ? SA[lane] : 0u;
        u32 p = s + dm;
        if (p >= n) p -= n;
        const u64 ke = valid ? load_be64(T, p) : 0ull;
        const int clo = (int)(u32)ke, chi = (int)(u32)(ke >> 32);
        const u64 k0 = ((u64)(u32)__builtin_amdgcn_readlane(chi, 0) << 32) | (u64)(u32)__builtin_amdgcn_readlane(clo, 0);
        if (__ballot(valid && ke != k0) == 0ull) {                      // these 8 bytes tie for the whole group: nothing to sort (wave-uniform)
            sp_append_class(LoutS, coutS, capS, lane == 0, d);
            continue;
        }
        u32 less = 0;
        for (u32 t = 0; t < len; t++) {
            const u64 kt = ((u64)(u32)__builtin_amdgcn_readlane(chi, (int)t) << 32) | (u64)(u32)__builtin_amdgcn_readlane(clo, (int)t);
            less += kt < ke ? 1u : 0u;
        }
        const u64 same = match_any(less, 7, valid);
        const u32 eqb = (u32)__popcll(same & lt), sublen0 = (u32)__popcll(same);
        if (valid) SA[less + eqb] = s;
        const bool head = valid && eqb == 0;                           // this member opens the sub-group at position `less`
        const u32 sublen = head ? sublen0 : 0u;
        if (head && less) atomicOr(&B.HN[(size_t)b * g.hstride + ((start + less) >> 5)], 1u << ((start + less) & 31u));
        const u32 xr = (b & 7u) * K1_DEEP_SUB + ((start >> 10) & (K1_DEEP_SUB - 1u));
        const u64 dd = ((u64)b << 52) | ((u64)(start + less) << 26) | ((u64)(depth + 8u) << 4) | (u64)(sublen - 1u);
        sp_append_class(B.listT[0] + (size_t)xr * rcap, &B.deepCnt[xr], rcap, sublen == 2u, dd);
        sp_append_class(B.listT[1] + (size_t)xr * rcap, &B.deepCnt[8u * K1_DEEP_SUB + xr], rcap, sublen > 2u && sublen <= K1_DEEP_LANE, dd);
        sp_append_class(LoutS, coutS, capS, sublen > K1_DEEP_LANE, sp_desc(b, start + less, sublen));
    }
}

// medrot_max: with more rotations than this in listed groups (HTML-like input: 30 % of all rotations sit in groups of
// 65..4096 that mostly tie for hundreds of bytes) the text rounds cost more than the rank rounds they would replace
// (measured on E8S-A: +11 ms against -7 ms), so every kernel of the stage returns at once: decided on the device, the
// host does not wait for the count.
__global__ __launch_bounds__(256) void k1_dm_round(K1Buf B, BatchGeom g, u32 depth, u32 medrot_max, int parity, int flat) {
    if (B.stats[K1_STAT_MEDROT] > medrot_max) return;
    __shared__ u64 ck[K1_MED_MAX];
    __shared__ u32 cv[K1_MED_MAX];
    __shared__ u32 hb[K1_MED_MAX / 32 + 2];
    __shared__ u32 differs;
    const u32 tid = threadIdx.x;
    // flat: the list k1_emit_medium wrote (one counter: it appends once per tile); else the sub-lists of the previous round:
    // workgroup L serves sub-list L % K1_DM_SUB (gridDim.x is a multiple of K1_DM_SUB) and appends to the same one of the next round
    const u32 sub = blockIdx.x & (K1_DM_SUB - 1u), capS = B.listSCap / K1_DM_SUB, capM = B.listMCap / K1_DM_SUB;
    u32 cm = flat ? B.stats[K1_STAT_LIST + parity * 4 + 2] : B.dmCnt[((u32)parity * 2u + 1u) * K1_DM_SUB + sub];
    if (cm > (flat ? B.listMCap : capM)) cm = flat ? B.listMCap : capM;
    const u64* Lin = flat ? B.listM[parity] : B.listM[parity] + (size_t)sub * capM;
    u64* LoutS = B.listS[parity ^ 1] + (size_t)sub * capS;
    u64* LoutM = B.listM[parity ^ 1] + (size_t)sub * capM;
    u32* coutS = B.dmCnt + (((u32)parity ^ 1u) * 2u + 0u) * K1_DM_SUB + sub;
    u32* coutM = B.dmCnt + (((u32)parity ^ 1u) * 2u + 1u) * K1_DM_SUB + sub;
    const u32 rcap = B.listTCap / (8u * K1_DEEP_SUB);
    for (u32 gi = flat ? blockIdx.x : blockIdx.x / K1_DM_SUB; gi < cm; gi += flat ? gridDim.x : gridDim.x / K1_DM_SUB) {
        const u64 d = Lin[gi];
        const u32 b = SP_B(d), start = SP_START(d), len = SP_LEN(d);
        const u32 n = B.nlen[b];
        const u8* T = B.T + (size_t)b * g.tstride;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        const u32 dm = depth % n;
        for (u32 i = tid; i < len; i += 256) {
            const u32 s = SA[i];
            u32 p = s + dm;
            if (p >= n) p -= n;
            cv[i] = s;
            ck[i] = load_be64(T, p);
        }
        for (u32 i = tid; i < K1_MED_MAX / 32 + 2; i += 256) hb[i] = 0;
        if (tid == 0) differs = 0;
        __syncthreads();
        {
            const u64 k0 = ck[0];
            bool df = false;
            for (u32 i = tid; i < len; i += 256) df = df || ck[i] != k0;
            if (df) differs = 1;
        }
        __syncthreads();
        if (!differs) {                                        // these 8 bytes tie for the whole group: nothing to sort, next round
            if (tid < 64) sp_append_class(LoutM, coutM, capM, tid == 0, d);
            __syncthreads();
            continue;
        }
        u32 M = 128;
        while (M < len) M <<= 1;
        for (u32 k = 2; k <= M; k <<= 1) {
            const u32 hk = k >> 1;
            for (u32 i = tid; i < (M >> 1); i += 256) {
                const u32 blk = i / hk, off = i - blk * hk;
                const u32 lo = blk * k + off, hi = blk * k + (k - 1u - off);
                if (hi < len) cmpx64(ck, cv, lo, hi);
            }
            __syncthreads();
            for (u32 j = k >> 2; j > 0; j >>= 1) {
                for (u32 i = tid; i < (M >> 1); i += 256) {
                    const u32 lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u));
                    const u32 hi = lo | j;
                    if (hi < len) cmpx64(ck, cv, lo, hi);
                }
                __syncthreads();
            }
        }
        for (u32 i = tid; i <= len; i += 256) {
            const bool head = i == 0 || i == len || ck[i] != ck[i - 1];
            if (head) atomicOr(&hb[i >> 5], 1u << (i & 31u));
        }
        __syncthreads();
        u32* HN = B.HN + (size_t)b * g.hstride;
        const u32 xr = (b & 7u) * K1_DEEP_SUB + ((start >> 10) & (K1_DEEP_SUB - 1u));
        for (u32 i0 = 0; i0 < len; i0 += 256) {                // uniform trip count (wave-wide appends)
            const u32 i = i0 + tid;
            u32 sublen = 0;
            if (i < len) {
                SA[i] = cv[i];
                if ((hb[i >> 5] >> (i & 31u)) & 1u) {          // a head: the sub-group runs to the next head
                    if (i) atomicOr(&HN[(start + i) >> 5], 1u << ((start + i) & 31u));
                    u32 wj = i >> 5;
                    u32 mm = (i & 31u) == 31u ? 0u : (hb[wj] & (0xFFFFFFFEu << (i & 31u)));
                    while (!mm) mm = hb[++wj];
                    sublen = wj * 32u + (u32)__ffs((int)mm) - 1u - i;
                }
            }
            // 2..8 rotations: the lane kernels' lists (pairs / small), with the depth they are now known to share
            const u64 dd = ((u64)b << 52) | ((u64)(start + i) << 26) | ((u64)(depth + 8u) << 4) | (u64)(sublen - 1u);
            sp_append_class(B.listT[0] + (size_t)xr * rcap, &B.deepCnt[xr], rcap, sublen == 2u, dd);
            sp_append_class(B.listT[1] + (size_t)xr * rcap, &B.deepCnt[8u * K1_DEEP_SUB + xr], rcap, sublen > 2u && sublen <= K1_DEEP_LANE, dd);
            // 9 and more: the next round
            const u64 ds = sp_desc(b, start + i, sublen);
            sp_append_class(LoutS, coutS, capS, sublen > K1_DEEP_LANE && sublen <= 64u, ds);
            sp_append_class(LoutM, coutM, capM, sublen > 64u, ds);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k1_count_unsorted(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y, n = B.nlen[b];
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
        if (t) atomicAdd(&B.spread[((size_t)K1_COUNT_SLOT * 2 + 1) * K1_SPREAD + ((blockIdx.x * 29u + b) & (K1_SPREAD - 1u))], t);
        if (t) atomicAdd(&B.dtot[b], t);                  // per block, for k1d_build (at most ~110 adds per word)
    }
}

// ---------------------------------------------------------------------------------------------
// groups of > K1_HT rotations: one 1024-thread workgroup per group, 3 stable 7-bit LSD passes
// through global memory (keys < 2^20)
// ---------------------------------------------------------------------------------------------
__device__ void seg_radix_pass(const u32* srcK, const u32* srcV, u32* dstK, u32* dstV, u32 L, u32 shift,
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

#define K1_MAJ_SIDE 2048u
__global__ __launch_bounds__(1024) void k1_sort_large(K1Buf B, BatchGeom g, u32 h, int mode, int round) {
    __shared__ u32 wh[16][128];
    __shared__ u32 dtot[128];
    __shared__ u32 s_end, s_pivot, s_side[2];
    __shared__ u32 sideK[2][K1_MAJ_SIDE], sideV[2][K1_MAJ_SIDE];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    u32 nl = B.stats[K1_STAT_LARGE + round];
    if (nl > B.largeCap) nl = B.largeCap;
    for (u32 gi = blockIdx.x; gi < nl; gi += gridDim.x) {
        const u32 b = B.large[gi].x, start = B.large[gi].y;
        const u32 n = B.nlen[b];
        const u32* HC = B.HC + (size_t)b * g.hstride;
        u32* HN = B.HN + (size_t)b * g.hstride;
        if (w == 0) {
            const u32 pos0 = start + 1u;
            const u32 wi0 = pos0 >> 5;
            int found = -1;
            for (int iter = 0; found < 0; iter++) {
                const u32 wi = wi0 + lane + 64u * (u32)iter;
                u32 wd = wi < g.hstride ? HC[wi] : 0xFFFFFFFFu;
                if (iter == 0 && lane == 0) wd &= 0xFFFFFFFFu << (pos0 & 31u);
                const u64 bal = __ballot(wd != 0u);
                if (bal) {
                    const int src = __ffsll((long long)bal) - 1;
                    const int pos = (int)(wi * 32u) + __ffs((int)wd) - 1;
                    found = __shfl(pos, src);
                }
            }
            if (lane == 0) s_end = (u32)found;
        }
        __syncthreads();
        const u32 L = s_end - start;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u32* SB = B.SB + (size_t)b * g.stride + start;
        u32* KA = B.KA + (size_t)b * g.stride + start;
        u32* KB = B.KB + (size_t)b * g.stride + start;
        const u32* ISA = B.ISA + (size_t)b * g.stride;
        const u32 hm = h % n;
        // Repetitive inputs keep huge groups alive for log2(n) rounds in which all but ~2h keys of a group are
        // equal.  A pivot taken from the middle of the group is then the majority key: count the two sides while
        // gathering the keys, and if both are small do ONE stable 3-way partition pass (the sides are sorted in
        // LDS) instead of three radix passes.
        if (tid == 0) { s_pivot = rot_key(ISA, n, SA[L >> 1], h, hm, mode, B.linear); s_side[0] = 0; s_side[1] = 0; }
        __syncthreads();
        const u32 pivot = s_pivot;
        u32 myl = 0, myg = 0;
        for (u32 i = tid; i < L; i += 1024) {
            const u32 s = SA[i];
            const u32 k = rot_key(ISA, n, s, h, hm, mode, B.linear);
            SB[i] = s;
            KB[i] = k;
            myl += k < pivot ? 1u : 0u;
            myg += k > pivot ? 1u : 0u;
        }
        for (u32 off = 32; off; off >>= 1) { myl += __shfl_xor(myl, off); myg += __shfl_xor(myg, off); }
        if (lane == 0) { if (myl) atomicAdd(&s_side[0], myl); if (myg) atomicAdd(&s_side[1], myg); }
        __syncthreads();
        const u32 nlt = s_side[0], ngt = s_side[1];
        if (n < (1u << 21) && nlt <= K1_MAJ_SIDE && ngt <= K1_MAJ_SIDE && (nlt + ngt) * 4u < L) {   // (key << 11 | index) needs keys < 2^21
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
                u32* comp = sideK[side];                                // (key << 11 | arrival index): keys < 2^20
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
            seg_radix_pass(KB, SB, KA, SA, L, 0, wh, dtot);
            seg_radix_pass(KA, SA, KB, SB, L, 7, wh, dtot);
            seg_radix_pass(KB, SB, KA, SA, L, 14, wh, dtot);
            if (n >= (1u << 21)) {                                      // BWT.* entry points on blocks of 2^21 .. 2^22-1 bytes
                seg_radix_pass(KA, SA, KB, SB, L, 21, wh, dtot);
                seg_radix_pass(KB, SB, KA, SA, L, 28, wh, dtot);
            }
        }
        __syncthreads();
        for (u32 i = tid + 1; i < L; i += 1024)
            if (KA[i] != KA[i - 1]) atomicOr(&HN[(start + i) >> 5], 1u << ((start + i) & 31u));
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Sparse phase.  Once few positions remain unsorted, a round costs what the unsorted groups cost:
// groups are kept as descriptors (block, start, length) in device lists; a wave sorts one group of
// <= 64 rotations in registers, a workgroup one group of <= K1_MED_MAX in LDS.  New ranks go to
// R (= SB) and are copied into ISA by k1_sp_update after every group of the round has read its
// keys.  The head bitmaps are not maintained any more.
// ---------------------------------------------------------------------------------------------
// Append one descriptor per lane with pred set: ONE atomic per wave and class (the lists' counters
// are single words; per-lane atomics on them saturate at ~90 per microsecond).  Must be called by
// all lanes of the wave (wave-uniform control flow).
__device__ __forceinline__ void sp_append_class(u64* list, u32* counter, u32 cap, bool pred, u64 d) {
    const u64 m = __ballot(pred);
    if (m == 0) return;                                          // wave-uniform
    const int leader = __ffsll((long long)m) - 1;
    u32 base = 0;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(counter, (u32)__popcll(m));
    base = __shfl(base, leader);
    if (pred) {
        const u32 idx = base + (u32)__popcll(m & lanemask_lt());
        if (idx < cap) list[idx] = d;
    }
}
__device__ __forceinline__ void sp_append(const K1Buf& B, int parity, bool pred, u32 b, u32 start, u32 len) {
    const u64 d = sp_desc(b, start, len);
    u32* c = B.stats + K1_STAT_LIST + parity * 4;
    sp_append_class(B.listT[parity], c + 0, B.listTCap, pred && len <= SP_TINY, d);
    sp_append_class(B.listS[parity], c + 1, B.listSCap, pred && len > SP_TINY && len <= 64u, d);
    sp_append_class(B.listM[parity], c + 2, B.listMCap, pred && len > 64u && len <= K1_MED_MAX, d);
    sp_append_class(B.listL[parity], c + 3, B.listLCap, pred && len > K1_MED_MAX, d);
}

__device__ __forceinline__ u32 sp_key(const K1Buf& B, const BatchGeom& g, u32 b, u32 n, u32 s, u32 h, u32 hm, int mode) {
    return rot_key(B.ISA + (size_t)b * g.stride, n, s, h, hm, mode, B.linear);
}

// groups of <= 8 rotations: one LANE each.  Keys and values live in registers; an 8-input
// odd-even merge network (19 compare-exchanges) sorts them, absent slots carry key 0xFFFFFFFF.
#define SP_CX(i, j) { const bool sw = k##i > k##j; const u32 tk = sw ? k##j : k##i, tv = sw ? v##j : v##i; \
                      k##j = sw ? k##i : k##j; v##j = sw ? v##i : v##j; k##i = tk; v##i = tv; }
__global__ __launch_bounds__(256) void k1_sp_tiny(K1Buf B, BatchGeom g, u32 h, int mode, int parity) {
    u32 cnt = B.stats[K1_STAT_LIST + parity * 4 + 0];
    if (cnt > B.listTCap) cnt = B.listTCap;
    const u32 nthreads = gridDim.x * 256u;
    const u32 rounds = (cnt + nthreads - 1u) / nthreads;        // uniform trip count: appends are wave-wide
    for (u32 r = 0; r < rounds; r++) {
        const u32 gi = r * nthreads + blockIdx.x * 256u + threadIdx.x;
        const bool act = gi < cnt;
        const u64 d = act ? B.listT[parity][gi] : 0ull;
        const u32 b = SP_B(d), start = SP_START(d), len = act ? SP_LEN(d) : 0u;
        const u32 n = act ? B.nlen[b] : 1u;
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u32* R = B.SB + (size_t)b * g.stride + start;
        const u32 hm = h % n;
        u32 k0 = ~0u, k1 = ~0u, k2 = ~0u, k3 = ~0u, k4 = ~0u, k5 = ~0u, k6 = ~0u, k7 = ~0u;
        u32 v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, v7 = 0;
#define SP_LD(i) if (len > i) { v##i = SA[i]; k##i = sp_key(B, g, b, n, v##i, h, hm, mode); }
        SP_LD(0) SP_LD(1) SP_LD(2) SP_LD(3) SP_LD(4) SP_LD(5) SP_LD(6) SP_LD(7)
#undef SP_LD
        SP_CX(0, 1) SP_CX(2, 3) SP_CX(4, 5) SP_CX(6, 7)
        SP_CX(0, 2) SP_CX(1, 3) SP_CX(4, 6) SP_CX(5, 7)
        SP_CX(1, 2) SP_CX(5, 6)
        SP_CX(0, 4) SP_CX(1, 5) SP_CX(2, 6) SP_CX(3, 7)
        SP_CX(2, 4) SP_CX(3, 5)
        SP_CX(1, 2) SP_CX(3, 4) SP_CX(5, 6)
        // write back, ranks, and the sub-groups that are still tied (at most 4)
        u32 hp = 0, sub0 = 0, sub1 = 0, sub2 = 0, sub3 = 0, nsub = 0;
#define SP_ST(i, kprev) if (len > i) { \
            if (i > 0 && k##i != kprev) { \
                if (i - hp >= 2u) { const u32 e = hp | ((i - hp) << 8); \
                    if (nsub == 0) sub0 = e; else if (nsub == 1) sub1 = e; else if (nsub == 2) sub2 = e; else sub3 = e; nsub++; } \
                hp = i; } \
            SA[i] = v##i; R[i] = start + hp; }
        SP_ST(0, 0u) SP_ST(1, k0) SP_ST(2, k1) SP_ST(3, k2) SP_ST(4, k3) SP_ST(5, k4) SP_ST(6, k5) SP_ST(7, k6)
#undef SP_ST
        if (len - hp >= 2u && len > 0) {
            const u32 e = hp | ((len - hp) << 8);
            if (nsub == 0) sub0 = e; else if (nsub == 1) sub1 = e; else if (nsub == 2) sub2 = e; else sub3 = e;
            nsub++;
        }
        // wave-aggregated append of up to 4 descriptors per lane
        const u32 incl = wave_incl_scan_u32(nsub);
        const u32 tot = __shfl(incl, 63);
        if (tot) {
            u32 basev = 0;
            if ((threadIdx.x & 63u) == 63u) basev = atomicAdd(&B.stats[K1_STAT_LIST + (parity ^ 1) * 4 + 0], tot);
            basev = __shfl(basev, 63);
            u32 o = basev + incl - nsub;
            u64* L = B.listT[parity ^ 1];
            if (nsub > 0 && o < B.listTCap) L[o] = sp_desc(b, start + (sub0 & 255u), sub0 >> 8);
            if (nsub > 1 && o + 1 < B.listTCap) L[o + 1] = sp_desc(b, start + (sub1 & 255u), sub1 >> 8);
            if (nsub > 2 && o + 2 < B.listTCap) L[o + 2] = sp_desc(b, start + (sub2 & 255u), sub2 >> 8);
            if (nsub > 3 && o + 3 < B.listTCap) L[o + 3] = sp_desc(b, start + (sub3 & 255u), sub3 >> 8);
        }
    }
}
#undef SP_CX

// groups of 9..64 rotations: one wave each, persistent grid
__global__ __launch_bounds__(256) void k1_sp_small(K1Buf B, BatchGeom g, u32 h, int mode, int parity) {
    __shared__ u32 sk[4][64];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    u32 cnt = B.stats[K1_STAT_LIST + parity * 4 + 1];
    if (cnt > B.listSCap) cnt = B.listSCap;
    const u32 nwaves = gridDim.x * 4u;
    const u64 lt = lanemask_lt();
    for (u32 gi = blockIdx.x * 4u + w; gi < cnt; gi += nwaves) {
        const u64 d = B.listS[parity][gi];
        const u32 b = SP_B(d), start = SP_START(d), len = SP_LEN(d);
        const u32 n = B.nlen[b];
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u32* R = B.SB + (size_t)b * g.stride + start;
        const bool act = lane < len;
        const u32 s = act ? SA[lane] : 0u;
        const u32 k = act ? sp_key(B, g, b, n, s, h, h % n, mode) : 0xFFFFFFFFu;
        u32 rank = 0;
        for (u32 j = 0; j < len; j++) {                        // wave-uniform trip count
            const u32 kj = __builtin_amdgcn_readlane(k, (int)j);
            rank += (kj < k || (kj == k && j < lane)) ? 1u : 0u;
        }
        if (act) { SA[rank] = s; sk[w][rank] = k; }
        __builtin_amdgcn_wave_barrier();
        const u32 mk = act ? sk[w][lane] : 0u;                 // key of the element now at position lane
        const u32 pk = (act && lane > 0) ? sk[w][lane - 1] : 0u;
        __builtin_amdgcn_wave_barrier();
        const bool head = act && (lane == 0 || mk != pk);
        const u64 hm64 = __ballot(head);
        u32 sublen = 0;
        if (act) {
            const u64 below = hm64 & (lt | (1ull << lane));
            const u32 hp = 63u - (u32)__clzll((long long)below);
            R[lane] = start + hp;
            if (head) {
                const u64 above = hm64 & ~(lt | (1ull << lane));
                const u32 nxt = above ? (u32)__ffsll((long long)above) - 1u : len;
                sublen = nxt - lane;
            }
        }
        sp_append(B, parity ^ 1, sublen >= 2u, b, start + lane, sublen);
    }
}

// groups of 65..K1_MED_MAX rotations: one workgroup each, persistent grid
__global__ __launch_bounds__(256) void k1_sp_medium(K1Buf B, BatchGeom g, u32 h, int mode, int parity) {
    __shared__ u32 ck[K1_MED_MAX], cv[K1_MED_MAX];
    __shared__ u32 hb[K1_MED_MAX / 32 + 2];
    const u32 tid = threadIdx.x;
    u32 cnt = B.stats[K1_STAT_LIST + parity * 4 + 2];
    if (cnt > B.listMCap) cnt = B.listMCap;
    for (u32 gi = blockIdx.x; gi < cnt; gi += gridDim.x) {
        const u64 d = B.listM[parity][gi];
        const u32 b = SP_B(d), start = SP_START(d), len = SP_LEN(d);
        const u32 n = B.nlen[b];
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u32* R = B.SB + (size_t)b * g.stride + start;
        const u32 hmod = h % n;
        for (u32 i = tid; i < len; i += 256) {
            const u32 s = SA[i];
            cv[i] = s;
            ck[i] = sp_key(B, g, b, n, s, h, hmod, mode);
        }
        for (u32 i = tid; i < K1_MED_MAX / 32 + 2; i += 256) hb[i] = 0;
        __syncthreads();
        u32 M = 128;
        while (M < len) M <<= 1;
        for (u32 k = 2; k <= M; k <<= 1) {
            const u32 hk = k >> 1;
            for (u32 i = tid; i < (M >> 1); i += 256) {
                const u32 blk = i / hk, off = i - blk * hk;
                const u32 lo = blk * k + off, hi = blk * k + (k - 1u - off);
                if (hi < len) cmpx(ck, cv, lo, hi);
            }
            __syncthreads();
            for (u32 j = k >> 2; j > 0; j >>= 1) {
                for (u32 i = tid; i < (M >> 1); i += 256) {
                    const u32 lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u));
                    const u32 hi = lo | j;
                    if (hi < len) cmpx(ck, cv, lo, hi);
                }
                __syncthreads();
            }
        }
        // heads of the sorted group -> LDS bitmap (bit len is a sentinel head)
        for (u32 i = tid; i <= len; i += 256) {
            const bool head = i == 0 || i == len || ck[i] != ck[i - 1];
            if (head) atomicOr(&hb[i >> 5], 1u << (i & 31u));
        }
        __syncthreads();
        for (u32 i0 = 0; i0 < len; i0 += 256) {                // uniform trip count (wave-wide appends)
            const u32 i = i0 + tid;
            u32 sublen = 0;
            if (i < len) {
                SA[i] = cv[i];
                u32 wi = i >> 5;                               // last head <= i
                u32 m = hb[wi] & (0xFFFFFFFFu >> (31u - (i & 31u)));
                while (!m) m = hb[--wi];
                const u32 hp = wi * 32u + 31u - (u32)__clz((int)m);
                R[i] = start + hp;
                if (hp == i) {                                 // next head > i
                    u32 wj = i >> 5;
                    u32 mm = (i & 31u) == 31u ? 0u : (hb[wj] & (0xFFFFFFFEu << (i & 31u)));
                    while (!mm) mm = hb[++wj];
                    sublen = wj * 32u + (u32)__ffs((int)mm) - 1u - i;
                }
            }
            sp_append(B, parity ^ 1, sublen >= 2u, b, start + i, sublen);
        }
        __syncthreads();
    }
}

// groups of more than K1_MED_MAX rotations: one 1024-thread workgroup each; 3 stable 7-bit LSD
// passes through global memory (SB doubles as value scratch and, afterwards, as the rank array R)
__global__ __launch_bounds__(1024) void k1_sp_large(K1Buf B, BatchGeom g, u32 h, int mode, int parity) {
    __shared__ u32 wh[16][128];
    __shared__ u32 dtot[128];
    __shared__ u32 scan_sh[20];
    __shared__ u32 s_carry;
    const u32 tid = threadIdx.x;
    u32 cnt = B.stats[K1_STAT_LIST + parity * 4 + 3];
    if (cnt > B.listLCap) cnt = B.listLCap;
    for (u32 gi = blockIdx.x; gi < cnt; gi += gridDim.x) {
        const u64 d = B.listL[parity][gi];
        const u32 b = SP_B(d), start = SP_START(d), len = SP_LEN(d);
        const u32 n = B.nlen[b];
        u32* SA = B.SA + (size_t)b * g.stride + start;
        u32* SB = B.SB + (size_t)b * g.stride + start;
        u32* KA = B.KA + (size_t)b * g.stride + start;
        u32* KB = B.KB + (size_t)b * g.stride + start;
        const u32 hmod = h % n;
        for (u32 i = tid; i < len; i += 1024) {
            const u32 s = SA[i];
            SB[i] = s;
            KB[i] = sp_key(B, g, b, n, s, h, hmod, mode);
        }
        if (tid == 0) s_carry = 0;                           // (last head position + 1) so far
        __syncthreads();
        seg_radix_pass(KB, SB, KA, SA, len, 0, wh, dtot);
        seg_radix_pass(KA, SA, KB, SB, len, 7, wh, dtot);
        seg_radix_pass(KB, SB, KA, SA, len, 14, wh, dtot);
        if (n >= (1u << 21)) {
            seg_radix_pass(KA, SA, KB, SB, len, 21, wh, dtot);
            seg_radix_pass(KB, SB, KA, SA, len, 28, wh, dtot);
        }
        // ranks (into SB) and the still-tied sub-groups, 1024 positions at a time
        for (u32 i0 = 0; i0 < len; i0 += 1024) {
            const u32 i = i0 + tid;
            const bool in = i < len;
            const bool head = in && (i == 0 || KA[i] != KA[i - 1]);
            const u32 v = head ? i + 1u : 0u;
            // inclusive max-scan over the block (values are monotone where non-zero)
            u32 m = v;
            for (u32 off = 1; off < 64; off <<= 1) {
                const u32 u = __shfl_up(m, off);
                if ((tid & 63u) >= off && u > m) m = u;
            }
            if ((tid & 63u) == 63u) scan_sh[tid >> 6] = m;
            __syncthreads();
            u32 wprev = s_carry;
            for (u32 ww = 0; ww < (tid >> 6); ww++) wprev = scan_sh[ww] > wprev ? scan_sh[ww] : wprev;
            const u32 incl = m > wprev ? m : wprev;          // last head (+1) at or before i
            u32 excl = __shfl_up(m, 1u);
            if ((tid & 63u) == 0) excl = 0;
            excl = excl > wprev ? excl : wprev;              // last head (+1) strictly before i
            if (in) SB[i] = start + incl - 1u;
            const u32 sublen = (head && i > 0) ? i - (excl - 1u) : 0u;
            sp_append(B, parity ^ 1, sublen >= 2u, b, start + (excl ? excl - 1u : 0u), sublen);
            __syncthreads();
            if (tid == 1023) s_carry = incl;
            __syncthreads();
        }
        {
            const u32 lasthead = s_carry - 1u;               // the final sub-group [lasthead, len)
            const u32 sublen = len - lasthead;
            sp_append(B, parity ^ 1, tid == 0 && sublen >= 2u, b, start + lasthead, sublen);
        }
        __syncthreads();
    }
}

// ISA[SA[p]] = R[p] for every position of the groups of this round
__global__ __launch_bounds__(256) void k1_sp_update(K1Buf B, BatchGeom g, int parity) {
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32* c = B.stats + K1_STAT_LIST + parity * 4;
    const u32 ct = c[0] < B.listTCap ? c[0] : B.listTCap;
    const u32 cs = c[1] < B.listSCap ? c[1] : B.listSCap;
    const u32 cm = c[2] < B.listMCap ? c[2] : B.listMCap;
    const u32 cl = c[3] < B.listLCap ? c[3] : B.listLCap;
    for (u32 gi = blockIdx.x * 256u + tid; gi < ct; gi += gridDim.x * 256u) {
        const u64 d = B.listT[parity][gi];
        const u32 b = SP_B(d), start = SP_START(d), len = SP_LEN(d);
        const size_t o = (size_t)b * g.stride + start;
        for (u32 i = 0; i < len; i++) B.ISA[(size_t)b * g.stride + B.SA[o + i]] = B.SB[o + i];
    }
    const u32 nwaves = gridDim.x * 4u;
    for (u32 gi = blockIdx.x * 4u + w; gi < cs; gi += nwaves) {
        const u64 d = B.listS[parity][gi];
        const u32 b = SP_B(d), start = SP_START(d), len = SP_LEN(d);
        if (lane < len) {
            const size_t o = (size_t)b * g.stride + start + lane;
            B.ISA[(size_t)b * g.stride + B.SA[o]] = B.SB[o];
        }
    }
    for (u32 gi = blockIdx.x; gi < cm + cl; gi += gridDim.x) {
        const u64 d = gi < cm ? B.listM[parity][gi] : B.listL[parity][gi - cm];
        const u32 b = SP_B(d), start = SP_START(d), len = SP_LEN(d);
        for (u32 i = tid; i < len; i += 256) {
            const size_t o = (size_t)b * g.stride + start + i;
            B.ISA[(size_t)b * g.stride + B.SA[o]] = B.SB[o];
        }
    }
}

__global__ void k1_dm_reset(K1Buf B, int parity) {            // the sub-list counters of one parity (medium rounds)
    if (threadIdx.x < 2u * K1_DM_SUB && blockIdx.x == 0) B.dmCnt[(u32)parity * 2u * K1_DM_SUB + threadIdx.x] = 0;
}
__global__ void k1_sp_reset(K1Buf B, int parity) {
    if (threadIdx.x < 4 && blockIdx.x == 0) B.stats[K1_STAT_LIST + parity * 4 + threadIdx.x] = 0;
}

// ---------------------------------------------------------------------------------------------
// BWT gather (lib/BWT.js:407-414)
// ---------------------------------------------------------------------------------------------
// Four suffix-array entries per thread: one 16-byte load, four text gathers in flight, one 4-byte store (round 3; one entry per
// thread was 4x the workgroups and a byte store each).
__global__ __launch_bounds__(256) void k1_finish(K1Buf B, BatchGeom g) {
    u32 b, tt;
    if (!xcd_block_tile(g.nb, b, tt)) return;
    const u32 n = B.nlen[b];
    const u32 p0 = (tt * 256u + threadIdx.x) * 4u;
    if (p0 >= n) return;
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* SA = B.SA + (size_t)b * g.stride;
    u8* U = B.U + (size_t)b * g.stride;
    if (p0 + 4u <= n) {
        const uint4 v = *(const uint4*)(SA + p0);          // stride is a multiple of 4 entries: 16-byte aligned
        const u32 s[4] = {v.x, v.y, v.z, v.w};
        u32 out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            out |= (u32)T[s[k] == 0 ? n - 1 : s[k] - 1] << (8 * k);
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

static inline size_t k1_tilehist_words(const BatchGeom& g) {
    const size_t a = (size_t)k1_stiles(g) * 256, f = k1_front_tilehist_words(g);
    return a > f ? a : f;
}

size_t k1_workspace_bytes(const BatchGeom& g) {
    const size_t e = (size_t)g.nb * g.stride;
    size_t tot = 0;
    tot += 5 * al256(e * 4);                                   // SA SB ISA KA KB
    tot += 3 * al256((size_t)g.nb * g.hstride * 4);            // HC HN HX
    tot += 2 * al256((size_t)g.nb * g.htiles);                 // FC FN
    tot += al256((size_t)g.nb * k1_tilehist_words(g) * 4);     // tileHist
    tot += al256((size_t)g.nb * K1F_NB * 8);                   // fsplit
    tot += al256((size_t)g.nb * (K1F_NB + 1) * 4);             // fstart
    tot += al256(K1_STATS * 4);
    tot += al256(4 * 8 * K1_DEEP_SUB * 4);                     // deepCnt (two passes)
    tot += al256(2 * 2 * K1_DM_SUB * 4);                       // dmCnt
    tot += al256(32 * 2 * K1_SPREAD * 4);                      // spread
    tot += al256((size_t)g.nb * (g.htiles + 1) * sizeof(uint2));
    tot += 2 * al256((size_t)((g.nb + 7u) & ~7u) * (g.stride / 2) * 8);       // listT cur/next (also the 8 per-XCD regions of k1_deep)
    tot += 2 * al256((size_t)g.nb * (g.stride / 8) * 8);       // listS cur/next
    tot += 2 * al256((size_t)g.nb * (g.stride / 64) * 8);      // listM cur/next
    tot += 2 * al256((size_t)g.nb * (g.stride / K1_MED_MAX + 1) * 8);   // listL cur/next
    tot += 2 * al256(e * 8);                                   // rlist in/out
    tot += al256((size_t)(K1R_MAXR + 1) * ((g.nb + 7u) & ~7u) * 4);   // rcnt
    tot += al256(((size_t)(2u * (K1D_MAXR + 2u) + 1u) * ((g.nb + 7u) & ~7u) + (K1D_MAXR + 2u) * 4u) * 4);   // dcnt, dchg, dtot, dbn
    tot += al256((size_t)K1F_LEVELS * g.nb * (g.stride / 256) * sizeof(uint4)) + 256;   // btask, bcnt
    return tot;
}

void k1_carve(K1Buf& B, const BatchGeom& g, void* ws) {
    char* p = (char*)ws;
    const size_t e = (size_t)g.nb * g.stride;
    B.SA = (u32*)p; p += al256(e * 4);
    B.SB = (u32*)p; p += al256(e * 4);
    B.ISA = (u32*)p; p += al256(e * 4);
    B.KA = (u32*)p; p += al256(e * 4);
    B.KB = (u32*)p; p += al256(e * 4);
    B.HC = (u32*)p; p += al256((size_t)g.nb * g.hstride * 4);
    B.HN = (u32*)p; p += al256((size_t)g.nb * g.hstride * 4);
    B.HX = (u32*)p; p += al256((size_t)g.nb * g.hstride * 4);
    B.FC = (u8*)p; p += al256((size_t)g.nb * g.htiles);
    B.FN = (u8*)p; p += al256((size_t)g.nb * g.htiles);
    B.tileHist = (u32*)p; p += al256((size_t)g.nb * k1_tilehist_words(g) * 4);
    B.fsplit = (u64*)p; p += al256((size_t)g.nb * K1F_NB * 8);
    B.fstart = (u32*)p; p += al256((size_t)g.nb * (K1F_NB + 1) * 4);
    B.stats = (u32*)p; p += al256(K1_STATS * 4);
    B.deepCnt = (u32*)p; p += al256(4 * 8 * K1_DEEP_SUB * 4);
    B.dmCnt = (u32*)p; p += al256(2 * 2 * K1_DM_SUB * 4);
    B.spread = (u32*)p; p += al256(32 * 2 * K1_SPREAD * 4);
    B.large = (uint2*)p; p += al256((size_t)g.nb * (g.htiles + 1) * sizeof(uint2));
    B.largeCap = g.nb * (g.htiles + 1);
    B.listTCap = ((g.nb + 7u) & ~7u) * (g.stride / 2);
    B.listSCap = g.nb * (g.stride / 8);
    B.listMCap = g.nb * (g.stride / 64);
    B.listT[0] = (u64*)p; p += al256((size_t)B.listTCap * 8);
    B.listT[1] = (u64*)p; p += al256((size_t)B.listTCap * 8);
    B.listS[0] = (u64*)p; p += al256((size_t)B.listSCap * 8);
    B.listS[1] = (u64*)p; p += al256((size_t)B.listSCap * 8);
    B.listM[0] = (u64*)p; p += al256((size_t)B.listMCap * 8);
    B.listM[1] = (u64*)p; p += al256((size_t)B.listMCap * 8);
    B.listLCap = g.nb * (g.stride / K1_MED_MAX + 1);
    B.listL[0] = (u64*)p; p += al256((size_t)B.listLCap * 8);
    B.listL[1] = (u64*)p; p += al256((size_t)B.listLCap * 8);
    B.rlist[0] = (u64*)p; p += al256(e * 8);
    B.rlist[1] = (u64*)p; p += al256(e * 8);
    B.rstride = (g.nb + 7u) & ~7u;
    B.rcnt = (u32*)p; p += al256((size_t)(K1R_MAXR + 1) * B.rstride * 4);
    B.dcnt = (u32*)p;
    B.dchg = B.dcnt + (size_t)(K1D_MAXR + 2u) * B.rstride;
    B.dtot = B.dchg + (size_t)(K1D_MAXR + 2u) * B.rstride;
    B.dbn = B.dtot + B.rstride;
    p += al256(((size_t)(2u * (K1D_MAXR + 2u) + 1u) * B.rstride + (K1D_MAXR + 2u) * 4u) * 4);
    B.btaskCap = g.nb * (g.stride / 256);
    B.btask = (uint4*)p; p += al256((size_t)K1F_LEVELS * B.btaskCap * sizeof(uint4));
    B.bcnt = (u32*)p;
}

static int g_k1_last_sparse_rounds = 0, g_k1_last_rounds = 0;
extern "C" int cjs_dbg_k1_sparse_rounds() { return g_k1_last_sparse_rounds; }
extern "C" int cjs_dbg_k1_rounds() { return g_k1_last_rounds; }

int k1_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream) {
    int sparse_rounds = 0;
    const dim3 gridH(g.htiles, g.nb);
    const u32 stiles = k1_stiles(g);
    const dim3 gridS(stiles, g.nb);
    const size_t sdyn = (size_t)K1_ST * 8;                    // lk + lv of k1_scatter
    static const bool lds_ok = []() {
        bool ok = true;
        ok = ok && hipFuncSetAttribute((const void*)k1_scatter<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, K1_ST * 8) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)k1_scatter<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, K1_ST * 8) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)k1_scatter<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, K1_ST * 8) == hipSuccess;
        return ok;
    }();
    if (!lds_ok) return CJS_E_HIP;
    const dim3 gridU((g.htiles + K1_UPT - 1) / K1_UPT, g.nb);
    const dim3 gridHX(g.htiles, (g.nb + 7u) & ~7u);           // XCD-aware kernels (see xcd_block_tile)
    const u32 initx = (g.hstride + 255) / 256;
    hipLaunchKernelGGL(k1_init, dim3(initx, g.nb), dim3(256), 0, stream, B, g);
    // 8 LSD passes over (key, index) pairs; buffers alternate (KB,SB), (KA,SA), ... and end in (KA,SA)
    // CJS_SORT_BYTES = 6..8 bytes of every rotation sorted by the radix passes (default 7, measured below; linear mode always 8): with fewer, the
    // low digits of stage 1 are skipped, groups are "equal first d0 bytes", K1-deep starts at depth d0 and the doubling rounds at h = d0
    static const u32 sort_bytes = []() -> u32 { const char* e = getenv("CJS_SORT_BYTES"); const u32 v = e ? (u32)strtoul(e, nullptr, 10) : 7u; return v < 6u || v > 8u ? 7u : v; }();
    // CJS_FRONT=0: the seven LSD passes below (kept as the reference path for A/B runs); default: the sample-sort
    // front end of k1_front.hip (one partition pass + in-LDS bucket sorts), which always sorts 8 bytes
    // (blocks of more than ~1.1 million bytes - only the BWT.* entry points see them - would overflow most of its
    // K1F_NB x K1F_C bucket slots: they take the LSD passes)
    static const bool front_env = []() { const char* e = getenv("CJS_FRONT"); return !e || atoi(e) != 0; }();
    const bool front = front_env && (u64)max_n * 16u <= (u64)K1F_NB * K1F_C * 9u;
    const u32 d0 = (B.linear || front) ? 8u : sort_bytes;
    const int p0 = front ? 8 : (int)(8u - d0);
    // K1-deep: iterations of 8 text bytes each (0 switches it off; default 32 = ties up to 264 bytes)
    static const u32 deep_iters = []() -> u32 { const char* e = getenv("CJS_DEEP_ITERS"); const u32 v = e ? (u32)strtoul(e, nullptr, 10) : 32u; return v > 4000u ? 4000u : v; }();
    // CJS_BSORT_ITERS = in-bucket deepening iterations of k1f_bsort (12 bytes each; default 1: the groups it lists share 20 bytes; measured 10^8-byte enwik ms per step with 0 / 1 / 2 / 3: 11.08 / 10.99 / 11.45 / 11.7);
    // CJS_ROUNDS=0: no lists and no refinement rounds - the K1-deep tile kernel and the lane kernels of rounds 1/2 do that work
    // (kept for A/B runs).  The rounds go on up to 8 + 8 * CJS_DEEP_ITERS bytes (264), what still ties there is left to prefix doubling.
    static const u32 bsort_iters = []() -> u32 { const char* e = getenv("CJS_BSORT_ITERS"); const u32 v = e ? (u32)strtoul(e, nullptr, 10) : 1u; return v > 64u ? 64u : v; }();
    static const bool rounds_env = []() { const char* e = getenv("CJS_ROUNDS"); return !e || atoi(e) != 0; }();
    // CJS_DEEP_BIG_DIV: text comparison is skipped when more than 1/DIV of the rotations sit in big 8-byte groups (see k1f_bsort / k1_deep)
    // (measured with the round-3 flow on E8S-A, where a third of the rotations sit in such groups: text stages + task levels 32.2 ms
    // against 20.7 ms with this predictor - its ties are hundreds of bytes long, which prefix doubling settles in log steps)
    static const u32 big_div = []() -> u32 { const char* e = getenv("CJS_DEEP_BIG_DIV"); const u32 v = e ? (u32)strtoul(e, nullptr, 10) : 8u; return v ? v : 8u; }();
    const bool fused = front && !B.linear && deep_iters > 0 && rounds_env && max_n <= (1u << 20);      // (list entries hold 20-bit indices)
    if (front) {
        const int rc = k1_front_run(B, g, max_n, stream, fused ? bsort_iters : 0u, fused ? 1u : 0u, (u32)(((u64)g.nb * max_n) / big_div));
        if (rc) return rc;
        if (getenv("CJS_K1_TRACE")) {
            u32 fs[K1_STATS - K1_STAT_FRONT_BIG];
            HIP_CHECK_RET(hipMemcpyAsync(fs, B.stats + K1_STAT_FRONT_BIG, sizeof fs, hipMemcpyDeviceToHost, stream));
            HIP_CHECK_RET(hipStreamSynchronize(stream));
            u64 bg = 0;
            for (u32 i = 0; i < 8u; i++) bg += fs[K1_STAT_BIGROT - K1_STAT_FRONT_BIG + i];
            fprintf(stderr, "[k1] front end: %u oversize buckets, %llu of %llu rotations in 8-byte groups above 64, %u in one-key buckets; k1f_bsort stage clocks/256 (K1F_TRACE builds): load %u  sample %u  partition %u  leaves %u  deepen %u  flush %u\n",
                    fs[0], (unsigned long long)bg, (unsigned long long)g.nb * max_n, fs[K1_STAT_PUREROT - K1_STAT_FRONT_BIG], fs[1], fs[2], fs[3], fs[4], fs[5], fs[6]);
        }
    }
    for (int p = p0; p < 8; p++) {
        const u32* kin = (p & 1) ? B.KB : B.KA;
        const u32* vin = (p & 1) ? B.SB : B.SA;
        u32* kout = (p & 1) ? B.KA : B.KB;
        u32* vout = (p & 1) ? B.SA : B.SB;
        const int shift = 8 * (p & 3);
        if (p == p0) hipLaunchKernelGGL(k1_hist<true>, gridS, dim3(K1_STH), 0, stream, B, g, kin, shift, stiles);
        else hipLaunchKernelGGL(k1_hist<false>, gridS, dim3(K1_STH), 0, stream, B, g, kin, shift, stiles);
        hipLaunchKernelGGL(k1_scan, dim3(g.nb), dim3(1024), 0, stream, B, g, stiles);
        K1Prof* pr = B.prof;
        // slots are reserved atomically: sub-batches on different streams are driven by different host threads
        const u32 slot = pr && pr->enabled ? __atomic_fetch_add(&pr->used, 1u, __ATOMIC_RELAXED) : K1_PROF_MAX;
        const bool timed = slot < K1_PROF_MAX;
        if (timed) (void)hipEventRecord(pr->ev[2 * slot], stream);
        if (p == p0) hipLaunchKernelGGL((k1_scatter<true, false>), gridS, dim3(K1_STH), sdyn, stream, B, g, kin, vin, kout, vout, shift, stiles);
        else if (p == 3) hipLaunchKernelGGL((k1_scatter<false, true>), gridS, dim3(K1_STH), sdyn, stream, B, g, kin, vin, kout, vout, shift, stiles);
        else hipLaunchKernelGGL((k1_scatter<false, false>), gridS, dim3(K1_STH), sdyn, stream, B, g, kin, vin, kout, vout, shift, stiles);
        if (timed) {
            (void)hipEventRecord(pr->ev[2 * slot + 1], stream);
            __atomic_fetch_add(&pr->elements, (u64)g.nb * max_n, __ATOMIC_RELAXED);
        }
    }
    if (!front) hipLaunchKernelGGL(k1_init_heads, gridHX, dim3(256), 0, stream, B, g, d0 == 8u ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8u * (8u - d0)));
    const size_t hbytes = (size_t)g.nb * g.hstride * 4;
    const u64 total_n = (u64)g.nb * max_n;
    static const bool k1_trace = getenv("CJS_K1_TRACE") != nullptr;
    static const u32 deep_tile = []() -> u32 { const char* e = getenv("CJS_DEEP_TILE"); return e ? (u32)strtoul(e, nullptr, 10) : 256u; }();
    static const u32 deep_dbg = []() -> u32 { const char* e = getenv("CJS_DEEP_DBG"); return e ? (u32)strtoul(e, nullptr, 10) : 0u; }();   // timing experiments: 1 = no phase 1, 2 = no phase 2
    const bool deep = deep_iters > 0 && !B.linear;
    if (deep) {
        // K1-deep's tile kernel returns at once when more than 1/CJS_DEEP_BIG_DIV of the rotations sit in 8-byte groups of more
        // than 64 members (counted by k1f_bsort; with CJS_FRONT=0 the count is 0 and the stage always runs).  With in-bucket
        // deepening (the default) k1f_bsort has done its work already and listed the groups of 2..8 that are left.
        const u32 bigrot_max = (u32)(total_n / big_div);
        if (!fused) {
            HIP_CHECK_RET(hipMemcpyAsync(B.HX, B.HN, hbytes, hipMemcpyDeviceToDevice, stream));
            if (deep_tile == 1024u) hipLaunchKernelGGL((k1_deep<1024, 256>), gridHX, dim3(256), 0, stream, B, g, deep_iters, deep_dbg, d0, bigrot_max);
            else hipLaunchKernelGGL((k1_deep<256, 64>), dim3(g.stride / 256u, (g.nb + 7u) & ~7u), dim3(64), 0, stream, B, g, deep_iters, deep_dbg, d0, bigrot_max);
        }
        // medium groups (9 .. K1_MED_MAX rotations) by text, 8 bytes per round; what they shed goes to the lane kernels' lists.
        // CJS_DEEP_MED = rounds (8: depths d0 .. d0 + 56).  Default 0 since round 3: the task levels of k1_front.hip sort these
        // groups deeper than the stage did, and its 27 launches (an empty one is 5 us) sat on every sub-batch's critical path -
        // 10^8-byte streams, ms per step with 8 / 0 rounds: enwik 10.13 / 9.94, E8S-A 19.65 / 19.07, random ASCII 9.86 / 9.70,
        // E8S-B 12.72 / 12.60, text 9.00 / 8.72; the repetitive shapes of tests/gpu_perf_probe.py 1-25 % faster, none slower.
        static const u32 med_rounds = []() -> u32 { const char* e = getenv("CJS_DEEP_MED"); const u32 v = e ? (u32)strtoul(e, nullptr, 10) : 0u; return v > 64u ? 64u : v; }();
        // the stage (and the long walks of the lane kernels) only when at most 1/32 of the rotations sit in medium groups
        static const u32 med_div = []() -> u32 { const char* e = getenv("CJS_DEEP_MED_DIV"); const u32 v = e ? (u32)strtoul(e, nullptr, 10) : 32u; return v ? v : 32u; }();
        const u32 medrot_max = (u32)(total_n / med_div);
        if (med_rounds) {
            hipLaunchKernelGGL(k1_emit_medium, gridHX, dim3(256), 0, stream, B, g);
            u32 mgrid = g.nb * 16u < 256u ? 256u : (g.nb * 16u > 2048u ? 2048u : g.nb * 16u);
            mgrid = (mgrid + K1_DM_SUB - 1u) / K1_DM_SUB * K1_DM_SUB;
            for (u32 r = 0; r < med_rounds; r++) {
                hipLaunchKernelGGL(k1_dm_round, dim3(mgrid), dim3(256), 0, stream, B, g, d0 + 8u * r, medrot_max, (int)(r & 1u), r == 0 ? 1 : 0);
                if (r) hipLaunchKernelGGL(k1_dm_round_small, dim3(mgrid), dim3(256), 0, stream, B, g, d0 + 8u * r, medrot_max, (int)(r & 1u));
                hipLaunchKernelGGL(k1_dm_reset, dim3(1), dim3(128), 0, stream, B, (int)(r & 1u));
            }
            hipLaunchKernelGGL(k1_sp_reset, dim3(1), dim3(64), 0, stream, B, 0);   // k1_emit_medium's counters; groups the last round listed stay marked in the bitmap
            hipLaunchKernelGGL(k1_dm_reset, dim3(1), dim3(128), 0, stream, B, (int)(med_rounds & 1u));
        }
        // the refinement rounds over what k1f_bsort listed (after the medium stage: its rounds use listS[] as scratch, and the
        // last refinement round appends what outlasts it to the lane kernels' second-pass lists in there)
        if (fused) {
            const int rc = k1_rounds_run(B, g, stream, 8u + K1F_STEP * bsort_iters, 8u + 8u * deep_iters);
            if (rc) return rc;
            if (k1_trace) {
                u32 rc2[K1R_MAXR + 1], rt[8];
                u64 tot[K1R_MAXR + 1] = {0};
                for (u32 r = 0; r <= K1R_MAXR; r++) {
                    for (u32 bb = 0; bb < g.nb; bb += 1) { HIP_CHECK_RET(hipMemcpyAsync(&rc2[r], B.rcnt + (size_t)r * B.rstride + bb, 4, hipMemcpyDeviceToHost, stream)); HIP_CHECK_RET(hipStreamSynchronize(stream)); tot[r] += rc2[r]; }
                }
                HIP_CHECK_RET(hipMemcpyAsync(rt, B.stats + K1_STAT_RTRACE, sizeof rt, hipMemcpyDeviceToHost, stream));
                HIP_CHECK_RET(hipStreamSynchronize(stream));
                fprintf(stderr, "[k1] refinement rounds, entries per round:");
                for (u32 r = 0; r <= K1R_MAXR && tot[r]; r++) fprintf(stderr, " %llu", (unsigned long long)tot[r]);
                fprintf(stderr, "\n[k1] k1r_round stage clocks/256 (K1F_TRACE builds): load %u  keys %u  rank %u  classify %u  reserve %u  write %u\n", rt[0], rt[1], rt[2], rt[3], rt[4], rt[5]);
            }
        }
        // lane kernels: CJS_DEEP_LANE_CAP = bytes a pair / small group is walked before it is left to the rank rounds
        // (default 4096: boilerplate passages of the text streams tie for up to ~3 KB; 0 = no second pass)
        static const u32 lane_cap = []() -> u32 { const char* e = getenv("CJS_DEEP_LANE_CAP"); const u32 v = e ? (u32)strtoul(e, nullptr, 10) : 4096u; return v > 60000u ? 60000u : v; }();
        const u32 capd0 = d0 + 8u * deep_iters;
        const u32 lane_unit = 8u * K1_DEEP_SUB;              // one workgroup per (XCD region, sub-region) at least
        const u32 lane_grid = g.nb * 32u <= lane_unit ? lane_unit : (g.nb * 32u >= 4096u ? 4096u : (g.nb * 32u + lane_unit - 1u) / lane_unit * lane_unit);
        hipLaunchKernelGGL(k1_deep_pairs<false>, dim3(lane_grid), dim3(256), 0, stream, B, g, capd0, 0u);
        hipLaunchKernelGGL(k1_deep_small<false>, dim3(lane_grid), dim3(256), 0, stream, B, g, capd0, 0u);
        if (lane_cap > capd0) {
            // second pass over what tied up to capd0, with the long cap, when at most 1 rotation group in 256 positions is left
            const u32 limit = (u32)(total_n / 256u);
            hipLaunchKernelGGL(k1_deep_pairs<true>, dim3(lane_grid), dim3(256), 0, stream, B, g, lane_cap, limit);
            hipLaunchKernelGGL(k1_deep_small<true>, dim3(lane_grid), dim3(256), 0, stream, B, g, lane_cap, limit);
        }
    }
    // What still ties after the stages above (long repeats, identical rotations, groups the text stages did not take; in linear
    // mode and on the A/B paths everything beyond the first d0 bytes): k1_count_unsorted finds the blocks that hold groups,
    // k1_dbl.hip ranks their rotations and runs the list-driven doubling rounds from h = d0 - every group shares at least the
    // d0 bytes of the first sort.  No counter is read back: blocks without groups cost nothing but the launches.
    hipLaunchKernelGGL(k1_count_unsorted, dim3((g.hstride + 255u) / 256u, g.nb), dim3(256), 0, stream, B, g);
    // The one read-back K1 keeps (CJS_K1_SYNC=0: none): when no block holds a group - the text stages finished the batch, the
    // usual case on text - the ~90 launches of the doubling stage (4.6 us each when they find their lists empty: 0.35 ms per
    // 10^8 bytes, measured) are not enqueued at all.  Everything after this point is steered on the device.
    static const bool k1_sync = []() { const char* e = getenv("CJS_K1_SYNC"); return !e || atoi(e) != 0; }();
    bool any_group = true;
    if (k1_sync && !B.linear) {
        std::vector<u32> tt(g.nb);
        HIP_CHECK_RET(hipMemcpyAsync(tt.data(), B.dtot, (size_t)g.nb * 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipStreamSynchronize(stream));
        any_group = false;
        for (u32 bb = 0; bb < g.nb; bb++) any_group = any_group || tt[bb] != 0u;
    }
    if (any_group) {
        const int rc = k1_dbl_run(B, g, max_n, stream, d0);
        if (rc) return rc;
    }
    if (k1_trace) {
        static thread_local u32 dc[(K1D_MAXR + 2) * 4];
        HIP_CHECK_RET(hipMemcpyAsync(dc, B.dbn, sizeof dc, hipMemcpyDeviceToHost, stream));
        std::vector<u32> cn((size_t)(K1D_MAXR + 2) * B.rstride), tt(B.rstride);
        HIP_CHECK_RET(hipMemcpyAsync(cn.data(), B.dcnt, cn.size() * 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipMemcpyAsync(tt.data(), B.dtot, tt.size() * 4, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipStreamSynchronize(stream));
        u64 un = 0;
        for (u32 bb = 0; bb < g.nb; bb++) un += tt[bb];
        fprintf(stderr, "[k1] rotations in unsorted groups before the doubling rounds: %llu (of %llu); per round entries / medium (<= 1024 + larger) / large / chunks:", (unsigned long long)un, (unsigned long long)total_n);
        for (u32 r = 0; r < K1D_MAXR + 1u; r++) {
            u64 e = 0;
            for (u32 bb = 0; bb < g.nb; bb++) e += cn[(size_t)r * B.rstride + bb];
            if (!e && !dc[r * 4] && !dc[r * 4 + 1] && !dc[r * 4 + 3]) break;
            fprintf(stderr, " [%u] %llu/%u+%u/%u/%u", r, (unsigned long long)e, dc[r * 4 + 3], dc[r * 4], dc[r * 4 + 1], dc[r * 4 + 2]);
            sparse_rounds = (int)r + 1;
        }
        fprintf(stderr, "\n");
    }
    const int round = sparse_rounds;
    g_k1_last_sparse_rounds = sparse_rounds;
    g_k1_last_rounds = round;
    if (B.linear) hipLaunchKernelGGL(k1_finish_linear, dim3((max_n + 255) / 256, g.nb), dim3(256), 0, stream, B, g, B.SAout);
    else hipLaunchKernelGGL(k1_finish, dim3((max_n + 1023) / 1024, (g.nb + 7u) & ~7u), dim3(256), 0, stream, B, g);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

int k1_prof_enable(K1Prof& p, int on) {
    if (on && !p.ev) {
        p.ev = new hipEvent_t[2 * K1_PROF_MAX];
        for (int i = 0; i < 2 * K1_PROF_MAX; i++) HIP_CHECK_RET(hipEventCreate(&p.ev[i]));
    }
    p.enabled = on;
    p.used = 0;
    p.elements = 0;
    return CJS_OK;
}
int k1_prof_read(K1Prof& p, float* total_ms, u32* launches, u64* elements) {
    float tot = 0.f;
    if (p.used > K1_PROF_MAX) p.used = K1_PROF_MAX;
    for (u32 i = 0; i < p.used; i++) {
        float ms = 0.f;
        HIP_CHECK_RET(hipEventSynchronize(p.ev[2 * i + 1]));
        HIP_CHECK_RET(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = p.used;
    if (elements) *elements = p.elements;
    p.used = 0;
    p.elements = 0;
    return CJS_OK;
}
void k1_prof_destroy(K1Prof& p) {
    if (p.ev) {
        for (int i = 0; i < 2 * K1_PROF_MAX; i++) (void)hipEventDestroy(p.ev[i]);
        delete[] p.ev;
        p.ev = nullptr;
    }
}
