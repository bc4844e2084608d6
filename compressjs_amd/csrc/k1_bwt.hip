// K1: batched cyclic suffix sort + BWT for gfx950 (wave64).
//
// Replaces BWT.bwtransform2 (lib/BWT.js:372-417), i.e. SA-IS on the doubled block plus the
// gather of lib/BWT.js:407-414.  The reference's order is: rotations of the block sorted as
// unsigned bytes, equal rotations by DESCENDING start index.  Any algorithm that
// realises this total order yields identical bytes, so this file does not port SA-IS.  It runs,
// for all blocks of a batch at once:
//
//   1. LSD radix sort of rotation indices by their first 8 bytes (8 stable 8-bit passes;
//      per-tile LDS histograms, wave-ballot ranking, bucket scatter).
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
#include "devutil.h"

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

template <bool FIRST>
__global__ __launch_bounds__(256) void k1_hist(K1Buf B, BatchGeom g, const u32* keys, int shift) {
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 t0 = t * K1_RT;
    if (t0 >= n) return;
    __shared__ u32 wh[4][256];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    for (u32 i = tid; i < 1024; i += 256) (&wh[0][0])[i] = 0;
    __syncthreads();
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* kb = keys + (size_t)b * g.stride;
#pragma unroll 4
    for (int it = 0; it < 16; it++) {
        const u32 j = t0 + w * 1024u + it * 64u + lane;
        if (j < n) {
            const u32 key = FIRST ? load_be32(T + j + 4) : kb[j];
            atomicAdd(&wh[w][(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    B.tileHist[((size_t)b * g.rtiles + t) * 256 + tid] = wh[0][tid] + wh[1][tid] + wh[2][tid] + wh[3][tid];
}

// per block: turn per-tile digit counts into global start offsets (digit-major, tile-minor)
__global__ __launch_bounds__(1024) void k1_scan(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 nt = (n + K1_RT - 1) / K1_RT;
    __shared__ u32 part[4][256];
    __shared__ u32 sh[256];
    const u32 tid = threadIdx.x, q = tid >> 8, d = tid & 255u;
    const u32 per = (nt + 3) / 4;
    const u32 tlo = q * per < nt ? q * per : nt;
    const u32 thi = tlo + per < nt ? tlo + per : nt;
    u32* hist = B.tileHist + (size_t)b * g.rtiles * 256;
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

template <bool FIRST, bool REKEY>
__global__ __launch_bounds__(256) void k1_scatter(K1Buf B, BatchGeom g, const u32* kin, const u32* vin, u32* kout,
                                                  u32* vout, int shift) {
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 t0 = t * K1_RT;
    if (t0 >= n) return;
    __shared__ u32 wh[4][256];
    __shared__ u32 dstart[256], gbase[256], sh[256];
    __shared__ u32 lk[K1_RT], lv[K1_RT];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    for (u32 i = tid; i < 1024; i += 256) (&wh[0][0])[i] = 0;
    __syncthreads();
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* kb = kin + (size_t)b * g.stride;
    const u32* vb = vin + (size_t)b * g.stride;
    u32 kv[16], vv[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const u32 j = t0 + w * 1024u + it * 64u + lane;
        u32 key = 0, val = 0;
        if (j < n) {
            key = FIRST ? load_be32(T + j + 4) : kb[j];
            val = FIRST ? j : vb[j];
            atomicAdd(&wh[w][(key >> shift) & 255u], 1u);
        }
        kv[it] = key;
        vv[it] = val;
    }
    __syncthreads();
    u32 total;
    {
        u32 o = 0;
#pragma unroll
        for (int ww = 0; ww < 4; ww++) {
            const u32 c = wh[ww][tid];
            wh[ww][tid] = o;                      // offset of wave ww inside digit `tid` of this tile
            o += c;
        }
        total = o;
        gbase[tid] = B.tileHist[((size_t)b * g.rtiles + t) * 256 + tid];
    }
    const u32 ds = block_excl_scan_256(total, sh);
    dstart[tid] = ds;
    __syncthreads();
    const u64 lt = lanemask_lt();
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const u32 j = t0 + w * 1024u + it * 64u + lane;
        const bool valid = j < n;
        const u32 d = (kv[it] >> shift) & 255u;
        const u64 m = match_any(d, 8, valid);
        const u32 rank = (u32)__popcll(m & lt), cnt = (u32)__popcll(m);
        const u32 base = valid ? wh[w][d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wh[w][d] = base + cnt;
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const u32 lp = dstart[d] + base + rank;
            lk[lp] = kv[it];
            lv[lp] = vv[it];
        }
    }
    __syncthreads();
    const u32 cntv = n - t0 < K1_RT ? n - t0 : K1_RT;
    u32* ko = kout + (size_t)b * g.stride;
    u32* vo = vout + (size_t)b * g.stride;
    for (u32 i = tid; i < cntv; i += 256) {
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
__global__ __launch_bounds__(256) void k1_init_heads(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 n = B.nlen[b];
    const u32 base = t * K1_HT;
    if (base >= n) return;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* SA = B.SA + (size_t)b * g.stride;
    const u32* KH = B.KA + (size_t)b * g.stride;     // bytes 0..3 of every rotation, in SA order
    u32* HN = B.HN + (size_t)b * g.hstride;
    for (int it = 0; it < 8; it++) {
        const u32 p = base + w * 512u + it * 64u + lane;
        u32 hi = 0, lo = 0;
        if (p < n) { hi = KH[p]; lo = load_be32(T + SA[p] + 4); }
        u32 phi = __shfl_up(hi, 1u), plo = __shfl_up(lo, 1u);
        if (lane == 0 && p > 0 && p < n) { phi = KH[p - 1]; plo = load_be32(T + SA[p - 1] + 4); }
        bool head = true;
        if (p < n && p > 0) head = (hi != phi) || (lo != plo);
        const u64 bal = __ballot(head);
        if (lane == 0) {
            HN[(p >> 5)] = (u32)bal;
            HN[(p >> 5) + 1] = (u32)(bal >> 32);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rank update: ISA[SA[p]] = position of p's group head under HN, for every p that was in an
// unsorted group under HC.  Also produces next round's tile flags and active-group count.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void update_ranks_tile(const K1Buf& B, const BatchGeom& g, int slot_out, u32 b, u32 t) {
    const u32 n = B.nlen[b];
    const u32 base = t * K1_HT;
    if (base >= n) return;
    const size_t fidx = (size_t)b * g.htiles + t;
    if (!(B.FC[fidx] & 2)) {
        if (threadIdx.x == 0) B.FN[fidx] = 0;
        return;
    }
    __shared__ u32 hc[68], hn[68];
    __shared__ int prevh[64];
    __shared__ int inHead;
    __shared__ u32 red[2];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32* HC = B.HC + (size_t)b * g.hstride;
    const u32* HN = B.HN + (size_t)b * g.hstride;
    if (tid < 68) {
        hc[tid] = HC[(base >> 5) + tid];
        hn[tid] = HN[(base >> 5) + tid];
    }
    if (tid == 0) { red[0] = 0; red[1] = 0; }
    __syncthreads();
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
        if (lane == 0) inHead = found;
    }
    __syncthreads();
    const u32* SA = B.SA + (size_t)b * g.stride;
    u32* ISA = B.ISA + (size_t)b * g.stride;
    u32 nstart = 0, nact = 0;
    for (int it = 0; it < 8; it++) {
        const u32 q0 = w * 512u + it * 64u;
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
        const u32 p = base + q;
        if ((actc >> lane) & 1u) {
            const u32 wq = q >> 5;
            const u32 mask = hn[wq] & (0xFFFFFFFFu >> (31u - (q & 31u)));
            u32 r;
            if (mask) r = base + wq * 32u + 31u - (u32)__clz((int)mask);
            else if (prevh[wq] >= 0) r = base + (u32)prevh[wq];
            else r = (u32)inHead;
            ISA[SA[p]] = r;
        }
    }
    if (nact) atomicAdd(&red[1], nact);
    if (nstart) atomicAdd(&red[0], nstart);
    __syncthreads();
    if (tid == 0) {
        B.FN[fidx] = (u8)((red[0] ? 1 : 0) | (red[1] ? 2 : 0));
        if (red[0]) atomicAdd(&B.stats[K1_STAT_ACTIVE + slot_out], red[0]);
    }
}

// (a persistent 8-per-CU grid walking the tiles was measured 2x SLOWER than one workgroup per
// tile for k1_refine and 12 % slower here: per-tile cost varies too much for static striding)
__global__ __launch_bounds__(256) void k1_update_ranks(K1Buf B, BatchGeom g, int slot_out) {
    update_ranks_tile(B, g, slot_out, blockIdx.y, blockIdx.x);
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
    __shared__ u32 hw[132];
    __shared__ int prevh[132], nexth[132];
    __shared__ u32 ck[K1_WIN], cv[K1_WIN];
    __shared__ u16 cp[K1_WIN], csz[K1_WIN];
    __shared__ u32 wtot[4];
    __shared__ u32 biglist[64];
    __shared__ u32 nbig;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32* HC = B.HC + (size_t)b * g.hstride;
    u32* HN = B.HN + (size_t)b * g.hstride;
    u32* SA = B.SA + (size_t)b * g.stride;
    const u32* ISA = B.ISA + (size_t)b * g.stride;
    const u32 wbase = base >> 5;
    if (tid < 130) hw[tid] = HC[wbase + tid];
    if (tid == 0) nbig = 0;
    __syncthreads();
    if (tid < 130) {
        int pv = -1;
        for (int i = (int)tid - 1; i >= 0; i--) {
            const u32 wd = hw[i];
            if (wd) { pv = i * 32 + 31 - __clz((int)wd); break; }
        }
        prevh[tid] = pv;
        int nx = K1_INF;
        for (int i = (int)tid + 1; i < 130; i++) {
            const u32 wd = hw[i];
            if (wd) { nx = i * 32 + __ffs((int)wd) - 1; break; }
        }
        nexth[tid] = nx;
    }
    __syncthreads();
    const u64 lt = lanemask_lt();
    // pass 1: count owned positions per wave, register large groups
    u32 cnt = 0;
    for (int it = 0; it < 16; it++) {
        const u32 q0 = w * 1024u + it * 64u;
        if (chunk_all_sorted(hw, q0)) continue;                       // wave-uniform
        const u32 q = q0 + lane;
        const PosClass c = classify(hw, prevh, nexth, q);
        const int size = (c.head >= 0 && c.endp < K1_INF) ? c.endp - c.head : K1_INF;
        const bool owned = c.head >= 0 && c.head < K1_HT && size >= 2 && size <= K1_HT && base + q < n;
        if (c.is_head && q < K1_HT && base + q < n && size > K1_HT) {
            const u32 idx = atomicAdd(&B.stats[K1_STAT_LARGE + round], 1u);
            if (idx < B.largeCap) B.large[idx] = make_uint2(b, base + q);
        }
        cnt += (u32)__popcll(__ballot(owned));
    }
    if (lane == 0) wtot[w] = cnt;
    __syncthreads();
    u32 wavebase = 0;
    for (u32 i = 0; i < w; i++) wavebase += wtot[i];
    const u32 m = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (m == 0) return;
    // pass 2: gather keys of owned positions into the compact arrays
    const u32 hm = h % n;
    u32 run = wavebase;
    for (int it = 0; it < 16; it++) {
        const u32 q0 = w * 1024u + it * 64u;
        if (chunk_all_sorted(hw, q0)) continue;
        const u32 q = q0 + lane;
        const PosClass c = classify(hw, prevh, nexth, q);
        const int size = (c.head >= 0 && c.endp < K1_INF) ? c.endp - c.head : K1_INF;
        const bool owned = c.head >= 0 && c.head < K1_HT && size >= 2 && size <= K1_HT && base + q < n;
        const u64 bal = __ballot(owned);
        if (owned) {
            const u32 e = run + (u32)__popcll(bal & lt);
            const u32 s = SA[base + q];
            u32 k;
            if (mode == 0) {
                u32 x = s + hm;
                if (x >= n) x -= n;
                k = ISA[x];
            } else {
                k = n - 1u - s;
            }
            ck[e] = ((u32)c.head << 20) | k;
            cv[e] = s;
            cp[e] = (u16)q;
            csz[e] = (u16)size;
            if (c.is_head && size > K1_SMALL) {
                const u32 bi = atomicAdd(&nbig, 1u);
                biglist[bi] = e | ((u32)size << 16);
            }
        }
        run += (u32)__popcll(bal);
    }
    __syncthreads();
    {
        // enumeration sort inside every small group (<= K1_SMALL elements)
        u32 nk[16], nv[16], ns[16];
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const u32 e = tid + (u32)it * 256u;
            ns[it] = 0xFFFFFFFFu;
            if (e < m && csz[e] <= K1_SMALL) {
                const u32 key = ck[e];
                const u32 gs = e - ((u32)cp[e] - (key >> 20));
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
        for (int it = 0; it < 16; it++)
            if (ns[it] != 0xFFFFFFFFu) { ck[ns[it]] = nk[it]; cv[ns[it]] = nv[it]; }
        __syncthreads();
    }
    // bigger groups: in-place bitonic network for arbitrary length (flip + half-cleaners, all
    // ascending; pairs whose upper index falls beyond the group are skipped)
    const u32 nb = nbig;
    for (u32 gi = 0; gi < nb; gi++) {
        const u32 e0 = biglist[gi] & 0xFFFFu, sz = biglist[gi] >> 16;
        u32* gk = ck + e0;
        u32* gv = cv + e0;
        u32 M = 128;
        while (M < sz) M <<= 1;
        for (u32 k = 2; k <= M; k <<= 1) {
            const u32 hk = k >> 1;
            for (u32 i = tid; i < (M >> 1); i += 256) {
                const u32 blk = i / hk, off = i - blk * hk;
                const u32 lo = blk * k + off, hi = blk * k + (k - 1u - off);
                if (hi < sz) cmpx(gk, gv, lo, hi);
            }
            __syncthreads();
            for (u32 j = k >> 2; j > 0; j >>= 1) {
                for (u32 i = tid; i < (M >> 1); i += 256) {
                    const u32 lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u));
                    const u32 hi = lo | j;
                    if (hi < sz) cmpx(gk, gv, lo, hi);
                }
                __syncthreads();
            }
        }
    }
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
    refine_tile(B, g, h, mode, round, blockIdx.y, blockIdx.x);
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

__global__ __launch_bounds__(1024) void k1_sort_large(K1Buf B, BatchGeom g, u32 h, int mode, int round) {
    __shared__ u32 wh[16][128];
    __shared__ u32 dtot[128];
    __shared__ u32 s_end;
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
        for (u32 i = tid; i < L; i += 1024) {
            const u32 s = SA[i];
            u32 k;
            if (mode == 0) {
                u32 x = s + hm;
                if (x >= n) x -= n;
                k = ISA[x];
            } else {
                k = n - 1u - s;
            }
            SB[i] = s;
            KB[i] = k;
        }
        __syncthreads();
        seg_radix_pass(KB, SB, KA, SA, L, 0, wh, dtot);
        seg_radix_pass(KA, SA, KB, SB, L, 7, wh, dtot);
        seg_radix_pass(KB, SB, KA, SA, L, 14, wh, dtot);
        for (u32 i = tid + 1; i < L; i += 1024)
            if (KA[i] != KA[i - 1]) atomicOr(&HN[(start + i) >> 5], 1u << ((start + i) & 31u));
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// BWT gather (lib/BWT.js:407-414)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1_finish(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.y;
    const u32 n = B.nlen[b];
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32 s = B.SA[(size_t)b * g.stride + p];
    B.U[(size_t)b * g.stride + p] = T[s == 0 ? n - 1 : s - 1];
    if (s == 0) B.pidx[b] = p;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t k1_workspace_bytes(const BatchGeom& g) {
    const size_t e = (size_t)g.nb * g.stride;
    size_t tot = 0;
    tot += 5 * al256(e * 4);                                   // SA SB ISA KA KB
    tot += 2 * al256((size_t)g.nb * g.hstride * 4);            // HC HN
    tot += 2 * al256((size_t)g.nb * g.htiles);                 // FC FN
    tot += al256((size_t)g.nb * g.rtiles * 256 * 4);           // tileHist
    tot += al256(K1_STATS * 4);
    tot += al256((size_t)g.nb * (g.htiles + 1) * sizeof(uint2));
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
    B.FC = (u8*)p; p += al256((size_t)g.nb * g.htiles);
    B.FN = (u8*)p; p += al256((size_t)g.nb * g.htiles);
    B.tileHist = (u32*)p; p += al256((size_t)g.nb * g.rtiles * 256 * 4);
    B.stats = (u32*)p; p += al256(K1_STATS * 4);
    B.large = (uint2*)p;
    B.largeCap = g.nb * (g.htiles + 1);
}

int k1_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream) {
    const dim3 gridR(g.rtiles, g.nb), gridH(g.htiles, g.nb);
    const u32 initx = (g.hstride + 255) / 256;
    hipLaunchKernelGGL(k1_init, dim3(initx, g.nb), dim3(256), 0, stream, B, g);
    // 8 LSD passes over (key, index) pairs; buffers alternate (KB,SB), (KA,SA), ... and end in (KA,SA)
    for (int p = 0; p < 8; p++) {
        const u32* kin = (p & 1) ? B.KB : B.KA;
        const u32* vin = (p & 1) ? B.SB : B.SA;
        u32* kout = (p & 1) ? B.KA : B.KB;
        u32* vout = (p & 1) ? B.SA : B.SB;
        const int shift = 8 * (p & 3);
        if (p == 0) hipLaunchKernelGGL(k1_hist<true>, gridR, dim3(256), 0, stream, B, g, kin, shift);
        else hipLaunchKernelGGL(k1_hist<false>, gridR, dim3(256), 0, stream, B, g, kin, shift);
        hipLaunchKernelGGL(k1_scan, dim3(g.nb), dim3(1024), 0, stream, B, g);
        K1Prof* pr = B.prof;
        const bool timed = pr && pr->enabled && pr->used < K1_PROF_MAX;
        if (timed) (void)hipEventRecord(pr->ev[2 * pr->used], stream);
        if (p == 0) hipLaunchKernelGGL((k1_scatter<true, false>), gridR, dim3(256), 0, stream, B, g, kin, vin, kout, vout, shift);
        else if (p == 3) hipLaunchKernelGGL((k1_scatter<false, true>), gridR, dim3(256), 0, stream, B, g, kin, vin, kout, vout, shift);
        else hipLaunchKernelGGL((k1_scatter<false, false>), gridR, dim3(256), 0, stream, B, g, kin, vin, kout, vout, shift);
        if (timed) {
            (void)hipEventRecord(pr->ev[2 * pr->used + 1], stream);
            pr->used++;
            pr->elements += (u64)g.nb * max_n;
        }
    }
    hipLaunchKernelGGL(k1_init_heads, gridH, dim3(256), 0, stream, B, g);
    hipLaunchKernelGGL(k1_update_ranks, gridH, dim3(256), 0, stream, B, g, 0);
    { u32* t = B.HC; B.HC = B.HN; B.HN = t; u8* f = B.FC; B.FC = B.FN; B.FN = f; }
    const size_t hbytes = (size_t)g.nb * g.hstride * 4;
    int round = 0;
    const u32 large_grid = g.nb * 4 < 1024 ? (g.nb * 4 < 64 ? 64 : g.nb * 4) : 1024;
    for (u64 h = 8;; h <<= 1) {
        const int mode = h >= max_n ? 1 : 0;      // last round: identical rotations by descending index
        HIP_CHECK_RET(hipMemcpyAsync(B.HN, B.HC, hbytes, hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(k1_refine, gridH, dim3(256), 0, stream, B, g, (u32)h, mode, round);
        hipLaunchKernelGGL(k1_sort_large, dim3(large_grid), dim3(1024), 0, stream, B, g, (u32)h, mode, round);
        hipLaunchKernelGGL(k1_update_ranks, gridH, dim3(256), 0, stream, B, g, round + 1);
        { u32* t = B.HC; B.HC = B.HN; B.HN = t; u8* f = B.FC; B.FC = B.FN; B.FN = f; }
        round++;
        if (mode == 1) break;
    }
    hipLaunchKernelGGL(k1_finish, dim3((max_n + 255) / 256, g.nb), dim3(256), 0, stream, B, g);
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
