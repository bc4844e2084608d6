// K1 front end: the initial sort of all rotations of every block by their first 8 bytes, as a SAMPLE SORT
// (one partition pass through HBM + one in-LDS sort per bucket) instead of seven LSD radix passes over
// (key, index) pairs.
//
// Replaces steps 1 of k1_bwt.hip (k1_hist/k1_scan/k1_scatter x 7 + k1_init_heads), i.e. the first part of the
// work SA-IS does in BWT.bwtransform2 (lib/BWT.js:372-417, :197-300).  Nothing of the reference is ported: any
// algorithm that delivers "rotations ordered by their first 8 bytes, groups of equal prefixes marked" feeds the
// refinement stages (the rounds below, K1-deep, prefix doubling) unchanged, and the final order is the reference's.
//
//   k1f_sample   per block: K1F_S keys (8 text bytes each) at stratified, hashed positions, bitonic-sorted in
//                LDS; every K1F_OVS-th is a splitter.  A key that fills more than one quantile gets a bucket of
//                its own ([v, v+1): nothing to sort there), so runs/periodic data cannot overflow a bucket.  Round 6: a key that fills
//                three quantiles or more shares its rotations with the (empty) buckets behind its own, by their NEXT 8 bytes against
//                sub-splitters from the key's own samples (fsub / fsplit2 / fp16: K1F_SUBBUCKETS in k1_bwt.h).
//   k1f_hist     per tile of K1F_PT rotations: key of every rotation from the LDS-staged text, bucket = number
//                of splitters <= key (branch-free binary search in LDS), per-tile bucket counts; per rotation the bucket id
//                and (round 5) the byte in FRONT of the rotation, which travels with it from here on (K1_SPACK).
//   k1f_scan     per block: bucket starts and per-(tile, bucket) write offsets.
//   k1f_scatter  packed index words (rotation index | byte in front << 24) to their bucket; order inside a bucket is irrelevant.
//   k1f_bsort    one workgroup per bucket: (round 5) ONE in-LDS sample sort on 16-byte keys gathered from the block's text
//                (L2-resident, all tiles of a block run on one XCD) - <= 64 leaves, lane-parallel rank counting inside the leaves by a
//                carry chain -, writes the suffix-array slice, the BWT bytes and the group heads once, and one list entry per
//                rotation that still ties.
//   k1f_task     (round 3) slices beyond LDS and groups above K1F_GBIG rotations, level after level: partitioned
//                like a block, or sorted in LDS 16 bytes deeper.
//   k1r_round    (round 3) list-driven refinement: 24 more text bytes per round off every listed rotation, ranked
//                inside its group; resolved rotations are final, the others go to the next round's list.
//
// HBM traffic per rotation of the first sort: text 1 + ids 4+4 + indices 4+4 + suffix array 4 + BWT byte 1 + keys gathered from
// L2 = ~22 bytes (the LSD design moved 7 x 20 = 140).  All integer work.
#include "k1_bwt.h"
#include "devutil.h"

__device__ __forceinline__ u64 k1f_load_be64(const u8* T, u32 p) {
    const u32 sh = p & 3u;
    u32 d[3];
    __builtin_memcpy(d, __builtin_assume_aligned(T + (p - sh), 4), 12);
    const u32 sel = be_sel(sh);
    return ((u64)be32_at(d[1], d[0], sel) << 32) | (u64)be32_at(d[2], d[1], sel);
}

__device__ __forceinline__ u32 k1f_hash(u32 k, u32 b) {
    u32 h = k * 2654435761u ^ (b + 1u) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return h;
}

// ---------------------------------------------------------------------------------------------
// splitters
// ---------------------------------------------------------------------------------------------
// The K1F_S samples are sorted by a bitonic network in LDS (1024 threads).  Round 3: every thread owns K1F_SE
// consecutive cells, so the stages with j < K1F_SE run in registers (one load + one store per cell for log2(K1F_SE) stages),
// and the stages above them go two at a time (four cells per step: j and j/2 in one LDS round trip); 105 LDS stages
// with a barrier each became 11 register passes + 30 four-cell passes (165 -> 70 us, on every sub-batch's critical path).
// Cells are padded by one per 16 so that the register passes (a thread's cells are 128 bytes apart) spread over the banks.
#define K1F_SE (K1F_S / 1024u)                           // cells per thread
#define K1F_SPAD(i) ((i) + ((i) >> 4))
static_assert(K1F_SE == 4u || K1F_SE == 8u || K1F_SE == 16u, "k1f_sample: 4, 8 or 16 samples per thread");

__device__ __forceinline__ void k1f_cmpx(u64& a, u64& c, bool up) {
    if ((a > c) == up) { const u64 t = a; a = c; c = t; }
}

// less += ((c, j) < (m, i)): 8-byte keys as two big-endian dwords, ties by the slot numbers - one borrow chain (see k1f_acc_lt below)
__device__ __forceinline__ void k1f_acc_lt64(u32& less, const uint2& c, u32 j, const uint2& m, u32 i) {
#if defined(__AMDGCN__)
    u32 t;
    asm("v_cmp_lt_u32 vcc, %2, %3\n\t"
        "v_subb_co_u32 %1, vcc, %4, %5, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %6, %7, vcc\n\t"
        "v_addc_co_u32 %0, vcc, 0, %0, vcc"
        : "+v"(less), "=&v"(t)
        : "v"(j), "v"(i), "v"(c.y), "v"(m.y), "v"(c.x), "v"(m.x)
        : "vcc");
#else
    const u64 cc = ((u64)c.x << 32) | c.y, mm = ((u64)m.x << 32) | m.y;
    less += (cc < mm || (cc == mm && j < i)) ? 1u : 0u;
#endif
}
__global__ __launch_bounds__(1024) void k1f_sample(K1Buf B, BatchGeom g) {
    const u32 b = blockIdx.x;
    const u32 n = B.nfront[b];
    const u32 tid = threadIdx.x;
    u64* sp = B.fsplit + (size_t)b * K1F_NB;
    u8* fsub = B.fsub + (size_t)b * K1F_NB;
    if (n <= K1F_C) {                                   // one bucket holds the whole block
        for (u32 j = tid; j < K1F_NB; j += 1024) { sp[j] = ~0ull; fsub[j] = 0; B.fp16[(size_t)b * K1F_NB + j] = 0; }
        return;
    }
    __shared__ u64 s[K1F_S + K1F_S / 16u];
    const u8* T = B.T + (size_t)b * g.tstride;
    for (u32 k = tid; k < K1F_S; k += 1024) {
        const u32 lo = (u32)((u64)k * n / K1F_S), hi = (u32)((u64)(k + 1u) * n / K1F_S);
        const u32 p = hi > lo + 1u ? lo + k1f_hash(k, b) % (hi - lo) : lo;
        s[K1F_SPAD(k)] = k1f_load_be64(T, p);
    }
    __syncthreads();
    const u32 base = tid * K1F_SE;
    u64 r[K1F_SE];
    {   // phases kk = 2 .. K1F_SE entirely in registers
#pragma unroll
        for (u32 m = 0; m < K1F_SE; m++) r[m] = s[K1F_SPAD(base + m)];
#pragma unroll
        for (u32 kk = 2; kk <= K1F_SE; kk <<= 1)
#pragma unroll
            for (u32 j = kk >> 1; j >= 1u; j >>= 1)
#pragma unroll
                for (u32 m = 0; m < K1F_SE; m++)
                    if ((m & j) == 0u) k1f_cmpx(r[m], r[m | j], ((base + m) & kk) == 0u);
#pragma unroll
        for (u32 m = 0; m < K1F_SE; m++) s[K1F_SPAD(base + m)] = r[m];
    }
    __syncthreads();
    for (u32 kk = 2u * K1F_SE; kk <= K1F_S; kk <<= 1) {
        u32 j = kk >> 1;
        u32 nst = 0;                                      // LDS stages of this phase: j = kk/2 .. K1F_SE
        for (u32 t = j; t >= K1F_SE; t >>= 1) nst++;
        if (nst & 1u) {                                   // an odd one first, two cells per step
            for (u32 i = tid; i < K1F_S / 2; i += 1024) {
                const u32 lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), hi = lo | j;
                u64 a = s[K1F_SPAD(lo)], c = s[K1F_SPAD(hi)];
                const u64 a0 = a;
                k1f_cmpx(a, c, (lo & kk) == 0u);
                if (a != a0) { s[K1F_SPAD(lo)] = a; s[K1F_SPAD(hi)] = c; }
            }
            __syncthreads();
            j >>= 1;
        }
        for (; j >= 2u * K1F_SE; j >>= 2) {               // stages j and j/2, four cells per step
            const u32 h = j >> 1;
            for (u32 qd = tid; qd < K1F_S / 4; qd += 1024) {
                const u32 i0 = ((qd & ~(h - 1u)) << 2) | (qd & (h - 1u));
                const bool up = (i0 & kk) == 0u;
                u64 a = s[K1F_SPAD(i0)], c = s[K1F_SPAD(i0 | h)], d = s[K1F_SPAD(i0 | j)], e = s[K1F_SPAD(i0 | j | h)];
                k1f_cmpx(a, d, up); k1f_cmpx(c, e, up);   // stage j
                k1f_cmpx(a, c, up); k1f_cmpx(d, e, up);   // stage j / 2
                s[K1F_SPAD(i0)] = a; s[K1F_SPAD(i0 | h)] = c; s[K1F_SPAD(i0 | j)] = d; s[K1F_SPAD(i0 | j | h)] = e;
            }
            __syncthreads();
        }
        {   // stages K1F_SE / 2 .. 1 in registers (one direction per thread: kk > K1F_SE)
            const bool up = (base & kk) == 0u;
#pragma unroll
            for (u32 m = 0; m < K1F_SE; m++) r[m] = s[K1F_SPAD(base + m)];
#pragma unroll
            for (u32 jj = K1F_SE >> 1; jj >= 1u; jj >>= 1)
#pragma unroll
                for (u32 m = 0; m < K1F_SE; m++)
                    if ((m & jj) == 0u) k1f_cmpx(r[m], r[m | jj], up);
#pragma unroll
            for (u32 m = 0; m < K1F_SE; m++) s[K1F_SPAD(base + m)] = r[m];
        }
        __syncthreads();
    }
#if K1F_SUBBUCKETS
    constexpr u32 K1F_HK = 512u;                        // heavy keys that get sub-splitters, at most (more: the rest keep their one bucket)
#ifndef K1F_HRANK
#define K1F_HRANK 256u      // (1024: a block of a few equally heavy keys - RLE1 output of zeros - spent 1.6 ms ranking five lists of 1024: the work is quadratic; 256 entries still give every one of up to 254 sub-buckets its quantile)
#endif
    constexpr u32 K1F_HR = K1F_HRANK;                   // entries of a heavy key's list that are ranked, at most (a key with more samples is thinned)
    __shared__ u64 hk[K1F_HK];                          // the heavy keys, ascending
    __shared__ u16 hlo[K1F_HK], hcn[K1F_HK], hbk[K1F_HK], hst[K1F_HK];   // first cell / entries of the key's list (cells of the sorted sample array), its first bucket, thinning stride
    __shared__ u32 hrun[K1F_HK];                        // samples of the key seen by the second pass
    __shared__ u8 mk[K1F_NB];
    __shared__ u32 scansh[20];
    __shared__ u32 anyh;
    if (tid == 0) anyh = 0;
    for (u32 i = tid; i < K1F_HK; i += 1024) hrun[i] = 0;
    u32 myhead[K1F_NB / 1024u], mylb[K1F_NB / 1024u], mycn[K1F_NB / 1024u];
    u64 myq[K1F_NB / 1024u];
    __syncthreads();
#endif
    for (u32 j = tid; j < K1F_NB; j += 1024) {
        u64 v = ~0ull;                                  // sp[K1F_NB-1] is padding (never compared)
        if (j + 1u < K1F_NB) {
            const u64 q = s[K1F_SPAD((j + 1u) * K1F_OVS)];
            v = q;
            if (j >= 1u && s[K1F_SPAD(j * K1F_OVS)] == q && q != ~0ull) v = q + 1u;   // heavy key: [q, q+1) becomes a bucket of its own
        }
        sp[j] = v;
        // heavy keys (K1F_SUBBUCKETS, k1_bwt.h): with Q_j = the (j + 1)-th quantile of the sample, a run Q_(d-1) = Q_d = .. = q makes bucket d the bucket of the ONE
        // key q (splitters q, q + 1) and the buckets behind it, up to the run's end, empty (splitters q + 1, q + 1): bucket d is marked with their number
        // (k1f_hist spreads the key's rotations over all of them), the others as members
        u32 mark = 0;
        if (K1F_SUBBUCKETS && j >= 1u && j + 1u < K1F_NB) {
            const u64 q = s[K1F_SPAD((j + 1u) * K1F_OVS)];
            if (q != ~0ull && s[K1F_SPAD(j * K1F_OVS)] == q) {
                if (j >= 2u && s[K1F_SPAD((j - 1u) * K1F_OVS)] == q) mark = 255u;             // (not the first of the run)
                else {
                    u32 cnt = 1u;
                    while (cnt < 254u && j + cnt + 1u < K1F_NB && s[K1F_SPAD((j + cnt + 1u) * K1F_OVS)] == q) cnt++;
                    mark = cnt;
                }
            }
        }
        fsub[j] = (u8)mark;
#if K1F_SUBBUCKETS
        mk[j] = (u8)mark;
        const u32 slotj = j / 1024u;
        myhead[slotj] = 0;
        if (mark >= 2u && mark != 255u) {
            // the key's samples in the sorted array: [first >= q, first > q)
            const u64 q = s[K1F_SPAD((j + 1u) * K1F_OVS)];
            u32 lb = 0, ub = 0;
#pragma unroll
            for (u32 step = K1F_S / 2u; step >= 1u; step >>= 1) {
                if (s[K1F_SPAD(lb + step - 1u)] < q) lb += step;
                if (s[K1F_SPAD(ub + step - 1u)] <= q) ub += step;
            }
            myhead[slotj] = 1u; mylb[slotj] = lb; mycn[slotj] = ub - lb; myq[slotj] = q;
            anyh = 1u;
        }
#endif
    }
#if K1F_SUBBUCKETS
    // ---- sub-splitters of the heavy keys: quantiles of the NEXT 8 bytes of the sampled rotations that start with the key.
    u64* sp2 = B.fsplit2 + (size_t)b * K1F_NB;
    u8* fp16 = B.fp16 + (size_t)b * K1F_NB;
    __syncthreads();
    if (!anyh) {                                        // (uniform) no key fills three quantiles: nothing to spread
        for (u32 j = tid; j < K1F_NB; j += 1024) { sp2[j] = ~0ull; fp16[j] = 0; }
        return;
    }
    // the heavy keys in bucket order (= ascending): index by two block scans over the threads' heads (buckets tid, then tid + 1024)
    u32 nh = 0;
#pragma unroll
    for (u32 sl = 0; sl < K1F_NB / 1024u; sl++) {
        u32 tot;
        const u32 ex = block_excl_scan_1024(myhead[sl], scansh, &tot);
        if (myhead[sl] && nh + ex < K1F_HK) {
            const u32 i = nh + ex;
            hk[i] = myq[sl]; hlo[i] = (u16)mylb[sl]; hbk[i] = (u16)(tid + sl * 1024u);
            // a key with more samples than K1F_HR is thinned: every stride-th sample is listed
#ifndef K1F_HTHIN
#define K1F_HTHIN 1u
#endif
            u32 stride = (mycn[sl] + K1F_HR - 1u) / K1F_HR;
            if (stride < K1F_HTHIN) stride = K1F_HTHIN;
            hst[i] = (u16)stride;
            hcn[i] = (u16)((mycn[sl] + stride - 1u) / stride);
        }
        nh += tot;
    }
    if (nh > K1F_HK) nh = K1F_HK;
    // the lists take the cells of the sorted array (not needed any more): list of key i = cells [hlo[i], hlo[i] + hcn[i]); ~0 = empty
    u64* lst = s;
    for (u32 i = tid; i < K1F_S; i += 1024) lst[i] = ~0ull;
    __syncthreads();
    // second pass over the samples: a sample whose key is a heavy one puts its NEXT 8 bytes into the key's list (eight samples per step, their loads in
    // flight together: as a plain loop it was sixteen dependent pairs of gathers per thread, 40 us on every sub-batch's critical path)
    for (u32 k0 = tid; k0 < K1F_S; k0 += 1024u * 8u) {
        u32 pp[8], hi8[8];
        u64 kq[8];
#pragma unroll
        for (u32 u = 0; u < 8u; u++) {
            const u32 k = k0 + u * 1024u;
            const u32 lo = (u32)((u64)k * n / K1F_S), hi = (u32)((u64)(k + 1u) * n / K1F_S);
            pp[u] = hi > lo + 1u ? lo + k1f_hash(k, b) % (hi - lo) : lo;
        }
#pragma unroll
        for (u32 u = 0; u < 8u; u++) kq[u] = k1f_load_be64(T, pp[u]);
#pragma unroll
        for (u32 u = 0; u < 8u; u++) {
            u32 i = 0;                                  // heavy keys below the sample's key
#pragma unroll
            for (u32 step = K1F_HK / 2u; step >= 1u; step >>= 1) if (i + step <= nh && hk[i + step - 1u] < kq[u]) i += step;
            hi8[u] = (i < nh && hk[i] == kq[u]) ? i : 0xFFFFFFFFu;
        }
#pragma unroll
        for (u32 u = 0; u < 8u; u++) kq[u] = hi8[u] != 0xFFFFFFFFu ? k1f_load_be64(T, pp[u] + 8u) : 0ull;     // (p + 15 < n + 64: inside the wrap-around bytes)
#pragma unroll
        for (u32 u = 0; u < 8u; u++)
            if (hi8[u] != 0xFFFFFFFFu) {
                const u32 i = hi8[u];
                const u32 seen = atomicAdd(&hrun[i], 1u), stride = hst[i];
                if (seen % stride == 0u && seen / stride < hcn[i]) lst[hlo[i] + seen / stride] = kq[u];     // (every stride-th sample of the key, in arrival order)
            }
    }
    __syncthreads();
    // every listed entry ranks itself inside its list by counting (strict order by value, then cell); the entries whose ranks are the wanted quantiles
    // drop their values into sp2 (global scratch for now: the final thresholds below read their neighbours')
    for (u32 j = tid; j < K1F_NB; j += 1024) sp2[j] = ~0ull;
    __threadfence_block();
    __syncthreads();
    for (u32 c = tid; c < K1F_S; c += 1024) {
        const u64 v = lst[c];
        if (v == ~0ull) continue;
        u32 i = 0;                                      // the list the cell belongs to: the last one that starts at or before it
#pragma unroll
        for (u32 step = K1F_HK / 2u; step >= 1u; step >>= 1) if (i + step <= nh && hlo[i + step - 1u] <= c) i += step;
        if (i == 0u) continue;
        i--;
        const u32 lo = hlo[i], cn = hcn[i];
        if (c >= lo + cn) continue;
        // rank = cells of the list that are smaller in the strict order (value, cell): one borrow chain per candidate (k1f_acc_lt64 - 64-bit compares run
        // at a quarter of the rate), K1F_RU cells per step, their reads in flight
        // together; cells behind the list's end (other lists', spare cells) count as (all ones, all ones): below nothing
#ifndef K1F_RU
#define K1F_RU 16u                                      // cells per step of the ranking loop: a step is a dependent LDS round trip, ~100 clocks for a kernel of four waves per SIMD
#endif
        u32 r = 0;
        const uint2 mv = make_uint2((u32)(v >> 32), (u32)v);
        for (u32 e = lo; e < lo + cn; e += K1F_RU) {
            u64 x[K1F_RU];
#pragma unroll
            for (u32 u = 0; u < K1F_RU; u++) x[u] = lst[e + u];
#pragma unroll
            for (u32 u = 0; u < K1F_RU; u++) {
                const bool in = e + u < lo + cn;
                k1f_acc_lt64(r, make_uint2(in ? (u32)(x[u] >> 32) : 0xFFFFFFFFu, in ? (u32)x[u] : 0xFFFFFFFFu), in ? e + u : 0xFFFFFFFFu, mv, c);
            }
        }
        // wanted: rank (t + 1) * cn / nsub for t = 0 .. nsub - 2 (empty cells - ~0 - rank last and are never wanted while the list is mostly full)
        const u32 h = hbk[i], nsub = mk[h];
        u32 t = (r * nsub) / cn;                        // (32-bit: r, cn <= K1F_HR, nsub <= 254 - a 64-bit division is a subroutine of hundreds of instructions)
        t = t ? t - 1u : 0u;
        for (; t + 1u < nsub; t++) {
            const u32 want = ((t + 1u) * cn) / nsub;
            if (want > r) break;
            if (want == r) sp2[h + t] = v;
        }
    }
    __threadfence_block();
    __syncthreads();
    // thresholds: bucket h + t of a run of nsub buckets (head h) ends below q_t; two equal quantiles q_(t-1) = q_t = x - a continuation that is heavy itself -
    // become the thresholds x, x + 1: bucket h + t holds ONE 16-byte key (fp16)
    u64 fin[K1F_NB / 1024u];
    u32 fin16[K1F_NB / 1024u];
#pragma unroll
    for (u32 sl = 0; sl < K1F_NB / 1024u; sl++) {
        const u32 j = tid + sl * 1024u;
        u64 v2 = ~0ull;
        u32 p16 = 0;
        u32 h = j, t = 0;
        const u32 mj = mk[j];
        bool run = mj >= 2u && mj != 255u;
        if (mj == 255u) {                               // a member: its head is the nearest bucket to the left with a count
            while (h > 0u && mk[h] == 255u) h--;
            t = j - h;
            run = mk[h] >= 2u && mk[h] != 255u && t < mk[h];
        }
        if (run) {
            const u32 nsub = mk[h];
            if (t + 1u < nsub) {
                u64 qt = sp2[h + t];
                const u64 qp = t >= 1u ? sp2[h + t - 1u] : 0ull;
                if (qt < qp) qt = qp;                       // (cannot happen - ranks order the values -; the sub-buckets' order depends on ascending thresholds)
                v2 = qt;
                if (t >= 1u && qp == qt && qt != ~0ull) v2 = qt + 1u;
                // [x, x + 1): my lower threshold is an unincremented x and my upper one x + 1
                if (t >= 1u && qp == qt && qt != ~0ull) {
                    const bool lower_plain = t == 1u || sp2[h + t - 2u] != qp;
                    if (lower_plain) p16 = 1u;
                }
            }
        }
        fin[sl] = v2; fin16[sl] = p16;
    }
    __syncthreads();
#pragma unroll
    for (u32 sl = 0; sl < K1F_NB / 1024u; sl++) { sp2[tid + sl * 1024u] = fin[sl]; fp16[tid + sl * 1024u] = (u8)fin16[sl]; }
#endif
}

// ---------------------------------------------------------------------------------------------
// partition: bucket ids + per-tile counts, offsets, scatter
// ---------------------------------------------------------------------------------------------
static inline u32 k1f_ptiles(const BatchGeom& g) { return (g.stride + K1F_PT - 1) / K1F_PT; }

__global__ __launch_bounds__(1024) void k1f_hist(K1Buf B, BatchGeom g, u32 ptiles) {
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 n = B.nfront[b];
    const u32 t0 = t * K1F_PT;
    const u32 tid = threadIdx.x;
    u32* th = B.tileHist + ((size_t)b * ptiles + t) * K1F_NB;
    if (t0 >= n) return;                                // k1f_scan only reads the tiles below n
    __shared__ u64 sp[K1F_NB];
    __shared__ u32 hist[K1F_NB];
    __shared__ u32 tx[K1F_PT / 4 + 4];
    __shared__ u8 fs[K1F_NB];                           // heavy-key marks of the buckets (k1f_sample)
    __shared__ u64 sp2[K1F_NB];                         // ... and their sub-splitters (bucket order)
    // The splitters in breadth-first order (node i: children 2i, 2i + 1; level k = cells [2^k, 2^(k+1))): the sorted array
    // put the 2^k nodes of the upper levels 2^(11-k) * 8 bytes apart, i.e. all in ONE pair of LDS banks - the six upper
    // levels of every search cost 2 + 4 + .. + 64 cycles (86 % of the kernel's LDS cycles were bank conflicts).
    const u64* gsp = B.fsplit + (size_t)b * K1F_NB;
    const u8* gfs = B.fsub + (size_t)b * K1F_NB;
    for (u32 d = tid; d < K1F_NB; d += 1024) {
        hist[d] = 0;
        fs[d] = gfs[d];
        sp2[d] = B.fsplit2[(size_t)b * K1F_NB + d];
        if (d) {
            const u32 k = 31u - (u32)__clz((int)d), j = d - (1u << k);
            sp[d] = gsp[(((2u * j + 1u) << (K1F_LOG_NB - 1u - k))) - 1u];
        }
    }
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32 avail = (g.tstride - t0) / 4u;            // dwords of this block's text slot from t0 on
    const u32* T32 = (const u32*)(T + t0);
    for (u32 i = tid; i < K1F_PT / 4 + 4; i += 1024) tx[i] = i < avail ? T32[i] : 0u;
    __syncthreads();
    u32* bid = B.KA + (size_t)b * g.stride;             // bucket id | byte in front of the rotation << 16 (k1f_scatter packs it into the index word: K1_SPACK)
    const u32 prev0 = tid == 0 ? (u32)T[t0 ? t0 - 1u : n - 1u] : 0u;
#pragma unroll
    for (int it = 0; it < K1F_PT / 1024; it++) {
        const u32 q = (u32)it * 1024u + tid, j = t0 + q;
        if (j < n) {
            const u32 sh = q & 3u, wi = q >> 2;
            const u32 sel = be_sel(sh);
            const u64 key = ((u64)be32_at(tx[wi + 1], tx[wi], sel) << 32) | (u64)be32_at(tx[wi + 2], tx[wi + 1], sel);
            u32 node = 1;                               // ends at K1F_NB + the number of splitters <= key
#pragma unroll
            for (u32 l = 0; l < K1F_LOG_NB; l++) node = 2u * node + (sp[node] <= key ? 1u : 0u);
            u32 pos = node - K1F_NB;
            const u32 nsb = fs[pos];
            if (K1F_SUBBUCKETS && nsb > 1u && nsb != 255u) {
                // the bucket of ONE heavy key with nsb - 1 empty buckets behind it: which of them, by the rotation's NEXT 8 bytes against the key's own
                // sub-splitters sp2[pos .. pos + nsb - 2] (the number of them that are <= the next 8 bytes)
                const u64 key2 = ((u64)be32_at(tx[wi + 3], tx[wi + 2], sel) << 32) | (u64)be32_at(tx[wi + 4], tx[wi + 3], sel);
                u32 lo2 = 0, len2 = nsb - 1u;
                while (len2) {
                    const u32 half = len2 >> 1;
                    if (sp2[pos + lo2 + half] <= key2) { lo2 += half + 1u; len2 -= half + 1u; }
                    else len2 = half;
                }
                pos += lo2;
            }
            atomicAdd(&hist[pos], 1u);
            const u32 prev = q ? (tx[(q - 1u) >> 2] >> (((q - 1u) & 3u) * 8u)) & 0xFFu : prev0;
            bid[j] = pos | (prev << 16);
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
    const u32 n = B.nfront[b];
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
        const u64* sp = B.fsplit + (size_t)b * K1F_NB;
        const bool pure = K1F_SUBBUCKETS ? B.fsub[(size_t)b * K1F_NB + tid] != 0u : (tid > 0u && tid < K1F_NB - 1u && sp[tid] == sp[tid - 1u] + 1u);
        if ((pure && tot > 64u) || tot > K1F_C) atomicAdd(&B.stats[K1_STAT_PUREROT], tot);
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
    const u32 n = B.nfront[b];
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
    {   // rotations in buckets of one 8-byte key (or too big for LDS): k1f_bsort's predictor for in-bucket deepening
        const u64* sp = B.fsplit + (size_t)b * K1F_NB;
        u32 pr = 0;
#pragma unroll
        for (u32 r = 0; r < R; r++) {
            const u32 d = tid + r * 1024u;
            const bool pure = K1F_SUBBUCKETS ? B.fsub[(size_t)b * K1F_NB + d] != 0u : (d > 0u && d < K1F_NB - 1u && sp[d] == sp[d - 1u] + 1u);
            if ((pure && sum[r] > 64u) || sum[r] > K1F_C) pr += sum[r];
        }
        u32 total;
        (void)block_excl_scan_1024(pr, sh, &total);
        if (tid == 0 && total) atomicAdd(&B.stats[K1_STAT_PUREROT], total);
    }
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

// Round 3: the tile's rotation indices are first put in bucket order in LDS (a counting sort with the ranks the LDS atomics
// hand out), then written: the ~4 indices a tile has for a bucket are neighbours in SB and leave in one request instead of
// four (one 4-byte write request per rotation was what the kernel waited for: 10^8 requests, 0.48 ms per 10^8 bytes).
__global__ __launch_bounds__(1024) void k1f_scatter(K1Buf B, BatchGeom g, u32 ptiles) {
    u32 b, t;
    if (!xcd_block_tile(g.nb, b, t)) return;
    const u32 n = B.nfront[b];
    const u32 t0 = t * K1F_PT;
    if (t0 >= n) return;
    __shared__ u32 cnt[K1F_NB], base[K1F_NB];
    __shared__ u32 stage[K1F_PT];
    __shared__ u16 sbk[K1F_PT];                         // bucket of every staged slot (the destination follows from it: no 4-byte dst[] array - 4096 buckets fit)
    __shared__ u32 sh[20];
    const u32 tid = threadIdx.x;
    const u32* th = B.tileHist + ((size_t)b * ptiles + t) * K1F_NB;
    for (u32 d = tid; d < K1F_NB; d += 1024) { cnt[d] = 0; base[d] = th[d]; }
    __syncthreads();
    const u32* bid = B.KA + (size_t)b * g.stride;
    u32* SB = B.SB + (size_t)b * g.stride;
    u32 dv[K1F_PT / 1024], rk[K1F_PT / 1024], pb[K1F_PT / 1024];
#pragma unroll
    for (int it = 0; it < K1F_PT / 1024; it++) {
        const u32 j = t0 + (u32)it * 1024u + tid;
        const u32 v = j < n ? bid[j] : 0xFFFFFFFFu;          // bucket | byte in front of rotation j << 16 (k1f_hist): the byte rides in the index word (K1_SPACK) from here on
        dv[it] = j < n ? (v & 0xFFFFu) : 0xFFFFFFFFu;
        pb[it] = (v >> 16) & 0xFFu;
    }
#pragma unroll
    for (int it = 0; it < K1F_PT / 1024; it++)
        if (dv[it] != 0xFFFFFFFFu) rk[it] = atomicAdd(&cnt[dv[it]], 1u);
    __syncthreads();
    {   // cnt[] -> first slot of every bucket in the tile's staging order
        constexpr u32 R = K1F_NB >= 1024 ? K1F_NB / 1024 : 1;
        const u32 d0 = tid * R;
        u32 c[R], sum = 0;
#pragma unroll
        for (u32 r = 0; r < R; r++) { c[r] = d0 + r < K1F_NB ? cnt[d0 + r] : 0u; sum += c[r]; }
        u32 total;
        u32 run = block_excl_scan_1024(sum, sh, &total);
#pragma unroll
        for (u32 r = 0; r < R; r++) { if (d0 + r < K1F_NB) cnt[d0 + r] = run; run += c[r]; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < K1F_PT / 1024; it++)
        if (dv[it] != 0xFFFFFFFFu) {
            const u32 slot = cnt[dv[it]] + rk[it];
            stage[slot] = K1_SPACK(t0 + (u32)it * 1024u + tid, pb[it]);
            sbk[slot] = (u16)dv[it];
        }
    __syncthreads();
    const u32 total = n - t0 < K1F_PT ? n - t0 : K1F_PT;
    for (u32 slot = tid; slot < total; slot += 1024) {
        const u32 d = sbk[slot];
        SB[base[d] + (slot - cnt[d])] = stage[slot];
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
#ifdef K1F_TRACE
#define K1R_STAMP(slot) do { if (tid == 0) { const long long now_ = clock64(); atomicAdd(&B.stats[K1_STAT_RTRACE + (slot)], (u32)((now_ - tprev_) >> 8)); tprev_ = now_; } } while (0)
#else
#define K1R_STAMP(slot) do { } while (0)
#endif
#ifndef K1F_BT
#define K1F_BT 256                                      // threads of a bucket-sort workgroup
#endif
#define K1F_NW (K1F_BT / 64)                            // its waves
#define K1F_E (K1F_C / K1F_BT)                          // rotations per thread
static_assert(K1F_C % K1F_BT == 0 && K1F_E >= 1, "a bucket-sort workgroup's threads hold K1F_C / K1F_BT rotations each");
#ifndef K1F_LK
#define K1F_LK (K1F_C >= 1024 ? 64 : K1F_C / 16)        // local sub-buckets (leaves) of a bucket, at most
#endif
#ifndef K1F_LOVS
#define K1F_LOVS 1                                      // local samples per leaf
#endif
#ifndef K1F_LEAF
#define K1F_LEAF 12u                                    // a bucket is cut into leaves of K1F_LEAF / 2 .. K1F_LEAF rotations on average
#endif
// (round 5, ms for k1f_bsort on enwik / E8S-A / random ASCII: 2 samples per leaf and leaves of 12..24 2.14 / 1.20 / 1.99; 20 / 28 / 40 instead of 24: 2.12 / 2.20 / 2.35;
//  4 samples per leaf 2.29; ONE sample per leaf and leaves of 6..12 2.01 / 1.15 / 1.85 (8, 6: the same) - the ranking loop's trip count is what counts)
#define K1F_LS (K1F_LOVS * K1F_LK)                      // local samples, at most
static_assert(K1F_LS <= K1F_BT && K1F_LK <= 64, "one thread per local sample; leaf ids are 6 bits; the leaf counts are scanned by one wave");
#define K1F_HW (K1F_C / 32 + 2)                         // words of the in-LDS head bitmap (bits >= cnt are set: sentinel)
#ifndef K1F_MINW
#define K1F_MINW 7                                      // waves per SIMD the register allocation of k1f_bsort is held to (7 workgroups per CU: LDS)
#endif
#ifndef K1F_GBIG
#define K1F_GBIG 256u                                   // groups up to this size are listed for the refinement rounds (8-bit fields of a list entry)
#endif
#define K1F_CAP (K1F_C - 4u)                            // rotations a bucket-sort workgroup takes (four spare key cells behind them: the ranking loops read past a leaf's end)

// ---- 16-byte keys: four big-endian dwords, x the most significant -------------------------------------------------
// 16 text bytes at T + p (any alignment): one dwordx4 + one dword load, one v_perm_b32 per key dword (alignment and byte order at once).  A random gather from the L2-resident
// text costs the same at 4 and at 16 bytes (tests/microbench/gather.hip), so the bucket sort takes all 16 at once.
__device__ __forceinline__ uint4 k1f_load_be128(const u8* T, u32 p) {
    const u32 sh = p & 3u;
    u32 d[5];
    __builtin_memcpy(d, __builtin_assume_aligned(T + (p - sh), 4), 20);
    const u32 sel = be_sel(sh);
    uint4 k;
    k.x = be32_at(d[1], d[0], sel);
    k.y = be32_at(d[2], d[1], sel);
    k.z = be32_at(d[3], d[2], sel);
    k.w = be32_at(d[4], d[3], sel);
    return k;
}
__device__ __forceinline__ bool k1f_eq128(const uint4& a, const uint4& b) { return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) == 0u; }
__device__ __forceinline__ bool k1f_lt128(const uint4& a, const uint4& b) {
    const u64 ah = ((u64)a.x << 32) | a.y, al = ((u64)a.z << 32) | a.w, bh = ((u64)b.x << 32) | b.y, bl = ((u64)b.z << 32) | b.w;
    return ah < bh || (ah == bh && al < bl);
}
__device__ __forceinline__ bool k1f_ones128(const uint4& a) { return (a.x & a.y & a.z & a.w) == 0xFFFFFFFFu; }
__device__ __forceinline__ uint4 k1f_inc128(uint4 a) {
    a.w += 1u;
    if (a.w == 0u) { a.z += 1u; if (a.z == 0u) { a.y += 1u; if (a.y == 0u) a.x += 1u; } }
    return a;
}
// less += ((c, j) < (m, i)): the 128-bit keys compared as integers, ties by the slot numbers j, i - a STRICT order, so the ranks a
// set of cells gets are a permutation.  On the GPU this is one carry chain: the borrow of c - m - (j < i) is the answer (five
// VALU instructions and the add, no mask arithmetic on the scalar unit; k1f_bsort is bound by its instruction stream).
__device__ __forceinline__ void k1f_acc_lt(u32& less, const uint4& c, u32 j, const uint4& m, u32 i) {
#if defined(__AMDGCN__)
    u32 t;
    asm("v_cmp_lt_u32 vcc, %2, %3\n\t"
        "v_subb_co_u32 %1, vcc, %4, %5, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %6, %7, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %8, %9, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %10, %11, vcc\n\t"
        "v_addc_co_u32 %0, vcc, 0, %0, vcc"
        : "+v"(less), "=&v"(t)
        : "v"(j), "v"(i), "v"(c.w), "v"(m.w), "v"(c.z), "v"(m.z), "v"(c.y), "v"(m.y), "v"(c.x), "v"(m.x)
        : "vcc");
#else
    less += (k1f_lt128(c, m) || (k1f_eq128(c, m) && j < i)) ? 1u : 0u;
#endif
}

// 8-byte keys (the bucket sort without text stages behind it: linear mode, HTML-like input): two big-endian dwords in a uint2
__device__ __forceinline__ uint2 k1f_load_be64x2(const u8* T, u32 p) {
    const u32 sh = p & 3u;
    u32 d[3];
    __builtin_memcpy(d, __builtin_assume_aligned(T + (p - sh), 4), 12);
    const u32 sel = be_sel(sh);
    return make_uint2(be32_at(d[1], d[0], sel), be32_at(d[2], d[1], sel));
}
// a step of the branch-free binary search over sorted splitters: pos + step when s <= k, else pos - the borrow of k - s picks (one
// ds_read_b128 and six vector instructions; as `if (!lt(k, s)) pos += step` the compiler read the key's halves one after the other,
// the second behind a branch: two dependent LDS round trips per step)
__device__ __forceinline__ u32 k1f_step_ge(u32 pos, u32 step, const uint4& k, const uint4& s) {
#if defined(__AMDGCN__)
    u32 t, r = pos + step;
    asm("v_sub_co_u32 %1, vcc, %2, %3\n\t"
        "v_subb_co_u32 %1, vcc, %4, %5, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %6, %7, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %8, %9, vcc\n\t"
        "v_cndmask_b32 %0, %0, %10, vcc"
        : "+v"(r), "=&v"(t)
        : "v"(k.w), "v"(s.w), "v"(k.z), "v"(s.z), "v"(k.y), "v"(s.y), "v"(k.x), "v"(s.x), "v"(pos)
        : "vcc");
    return r;
#else
    return k1f_lt128(k, s) ? pos : pos + step;
#endif
}
__device__ __forceinline__ u32 k1f_step_ge64(u32 pos, u32 step, const uint2& k, const uint2& s) {
#if defined(__AMDGCN__)
    u32 t, r = pos + step;
    asm("v_sub_co_u32 %1, vcc, %2, %3\n\t"
        "v_subb_co_u32 %1, vcc, %4, %5, vcc\n\t"
        "v_cndmask_b32 %0, %0, %6, vcc"
        : "+v"(r), "=&v"(t)
        : "v"(k.y), "v"(s.y), "v"(k.x), "v"(s.x), "v"(pos)
        : "vcc");
    return r;
#else
    return ((((u64)k.x << 32) | k.y) < (((u64)s.x << 32) | s.y)) ? pos : pos + step;
#endif
}
// the two key widths of the bucket sort behind one interface
struct K1fK16 {
    typedef uint4 T;
    static __device__ __forceinline__ T load(const u8* Tx, u32 p) { return k1f_load_be128(Tx, p); }
    static __device__ __forceinline__ T zero() { return make_uint4(0u, 0u, 0u, 0u); }
    static __device__ __forceinline__ T ones() { return make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); }
    static __device__ __forceinline__ bool eq(const T& a, const T& b) { return k1f_eq128(a, b); }
    static __device__ __forceinline__ bool lt(const T& a, const T& b) { return k1f_lt128(a, b); }
    static __device__ __forceinline__ bool is_ones(const T& a) { return k1f_ones128(a); }
    static __device__ __forceinline__ T inc(const T& a) { return k1f_inc128(a); }
    static __device__ __forceinline__ void acc_lt(u32& less, const T& c, u32 j, const T& m, u32 i) { k1f_acc_lt(less, c, j, m, i); }
    static __device__ __forceinline__ u32 step_ge(u32 pos, u32 step, const T& k, const T& sp) { return k1f_step_ge(pos, step, k, sp); }
};
struct K1fK8 {
    typedef uint2 T;
    static __device__ __forceinline__ T load(const u8* Tx, u32 p) { return k1f_load_be64x2(Tx, p); }
    static __device__ __forceinline__ T zero() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ T ones() { return make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu); }
    static __device__ __forceinline__ bool eq(const T& a, const T& b) { return ((a.x ^ b.x) | (a.y ^ b.y)) == 0u; }
    static __device__ __forceinline__ bool lt(const T& a, const T& b) { return (((u64)a.x << 32) | a.y) < (((u64)b.x << 32) | b.y); }
    static __device__ __forceinline__ bool is_ones(const T& a) { return (a.x & a.y) == 0xFFFFFFFFu; }
    static __device__ __forceinline__ T inc(T a) { a.y += 1u; if (a.y == 0u) a.x += 1u; return a; }
    static __device__ __forceinline__ void acc_lt(u32& less, const T& c, u32 j, const T& m, u32 i) { k1f_acc_lt64(less, c, j, m, i); }
    static __device__ __forceinline__ u32 step_ge(u32 pos, u32 step, const T& k, const T& sp) { return k1f_step_ge64(pos, step, k, sp); }
};

__device__ __forceinline__ bool k1f_bit(const u32* bm, u32 q) { return (bm[q >> 5] >> (q & 31u)) & 1u; }

// LDS of a bucket-sort workgroup (views into the kernel's __shared__ arrays)
struct K1fL {
    uint4* key;     // [K1F_C]  16-byte keys: arrival order, then leaf order, then sorted
    u32* sx;        // [K1F_C]  packed index word (K1_SPACK) at every position; before the leaf order the local sample sort's scratch:
    uint4* smp;     //   [K1F_LS] samples (aliases sx)
    uint4* sp;      //   [K1F_LK] local splitters (aliases sx)
    u32* srank;     //   [K1F_LS] sample ranks (aliases sx)
    u32* cnt2;      // [K1F_LK]      rotations per leaf
    u32* off2;      // [K1F_LK + 1]  first position of every leaf | 1 << 31 for a leaf of ONE key
    u32* hb;        // [K1F_HW] group heads (bit 0 and bits >= cnt set)
    u32* lb;        // [K1F_HW] flush: listed positions
    u8* lf;         // [K1F_C]  leaf of every position (leaf order)
    u32* misc;      // [K1F_E * K1F_NW + 8] workgroup-wide scratch words
};

// A slice that a bucket-sort workgroup cannot finish in LDS becomes a TASK of the next level of k1f_task (below):
// x = block, y = suffix-array position of the slice, z = its length, w = depth (bytes all its rotations share) | K1F_TASK_SB
// when the indices sit in the slice of SB (not SA).  One atomic per task (they are rare).
#define K1F_TASK_SB 0x80000000u
#define K1F_TASK_PLAIN 0x40000000u   // the slice holds plain indices (k1f_flush wrote it): in carry mode the task first packs the BWT bytes of its positions into them
#define K1F_PS (K1F_C / 2)          // samples a task partition sorts in LDS
#define K1F_PB (K1F_C / 4)          // sub-buckets of a task partition, at most (<= 256: one byte per id)
static_assert(K1F_PB <= 256 && K1F_PB >= 64 && (K1F_PB & (K1F_PB - 1)) == 0 && K1F_PS % 4 == 0, "task partition geometry");
__device__ __forceinline__ void k1f_push_task(const K1Buf& B, u32 level, u32 b, u32 pos, u32 len, u32 depth_flag) {
    if (level >= K1F_LEVELS) return;
    // (round 6: one list per level AND XCD - b mod 8, where every other kernel works on block b: a task's random key gathers then hit the L2 that
    // holds the block's text; one list per level, walked by all workgroups, had 32 % L2 hits and 150 bytes of fetch traffic per rotation)
    const u32 li = level * 8u + (b & 7u) % B.btaskLists, cap8 = B.btaskCap;
    const u32 idx = atomicAdd(&B.bcnt[li], 1u);
    if (idx < cap8) B.btask[(size_t)li * cap8 + idx] = make_uint4(b, pos, len, depth_flag);
}

// Results of positions [0, cnt) of the slice that starts at suffix-array position `pos0` of block b: the suffix indices and the
// head bits; with `lists`, every group of 2..K1F_GBIG rotations goes, member by member, to the block's list of
// the first refinement round (k1r_round), and its positions are marked as heads right away: the rounds resolve them or,
// where a tie outlasts the last round, clear the bits again.  One atomic per workgroup reserves the list slots.  Groups above
// K1F_GBIG rotations stay marked as groups and become tasks of level `task_level`, `task_depth` deep.
__device__ __forceinline__ void k1f_flush(const K1fL& S, const K1Buf& B, const BatchGeom& g, u32 b, u32 pos0, u32 cnt, bool lists,
                                          u32 task_level, u32 task_depth, bool carry) {
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    u32* SA = B.SA + (size_t)b * g.stride + pos0;
    u32* HN = B.HN + (size_t)b * g.hstride;
    u32* lb = S.lb;
    const u64 lt = lanemask_lt();
    u64 bal[K1F_E];
    u32 gsv[K1F_E];                                     // listed: index in the group | (length - 1) << 8
    if (lists) {
#pragma unroll
        for (int it = 0; it < K1F_E; it++) {
            const u32 q0 = (u32)it * K1F_BT + w * 64u;
            bal[it] = 0; gsv[it] = 0;
            if (q0 >= cnt) {                            // wave-uniform
                if (lane < 2u && (q0 >> 5) + lane < K1F_HW) lb[(q0 >> 5) + lane] = 0;
                if (lane == 0) S.misc[(u32)it * K1F_NW + w] = 0;
                continue;
            }
            // The group around every position of the row from the ROW's 64 head bits (round 5; per lane it was two reads of the bitmap and two
            // searches with divergent loops - a third of this function's vector instructions were the copies their control flow needs):
            // the words are the same for all lanes, the heads in front of and behind the row are searched once per row on the scalar unit
            // (bit 0 and every bit >= cnt are set, behind them two words of ones).
            const u32 q = q0 + lane, hw = q0 >> 5;
            const u32 h0 = (u32)__builtin_amdgcn_readfirstlane((int)S.hb[hw]), h1 = (u32)__builtin_amdgcn_readfirstlane((int)S.hb[hw + 1u]);
            u32 gprev = q0, gnext;
            if (!(h0 & 1u)) {                           // the row starts inside a group (wave-uniform)
                u32 ww = hw, m;
                do { m = (u32)__builtin_amdgcn_readfirstlane((int)S.hb[--ww]); } while (!m);
                gprev = ww * 32u + 31u - (u32)__builtin_clz(m);
            }
            {
                u32 ww = hw + 2u, m;
                while (!(m = (u32)__builtin_amdgcn_readfirstlane((int)S.hb[ww]))) ww++;
                gnext = ww * 32u + (u32)__builtin_ctz(m);
            }
            const u64 H = (u64)h0 | ((u64)h1 << 32), le = (lt << 1) | 1ull;
            const u64 hb_ = H & le, ha_ = H & ~le;
            const u32 gs = hb_ ? q0 + 63u - (u32)__builtin_clzll(hb_) : gprev;
            const u32 ge_ = ha_ ? q0 + (u32)__builtin_ctzll(ha_) : gnext;
            const u32 gl = ge_ - gs;
            bool listed = false;
            if (q < cnt && gl > 1u) {
                listed = gl <= K1F_GBIG;
                gsv[it] = (q - gs) | ((gl - 1u) << 8);
                // a group too big for the lists: a task of the next level
                if (!listed && q == gs) k1f_push_task(B, task_level, b, pos0 + gs, gl, task_depth | K1F_TASK_PLAIN);
            }
            bal[it] = __ballot(listed);
            if (lane == 0) { lb[q0 >> 5] = (u32)bal[it]; lb[(q0 >> 5) + 1u] = (u32)(bal[it] >> 32); }
            if (lane == 0) S.misc[(u32)it * K1F_NW + w] = (u32)__popcll(bal[it]);     // rows in position order: it-major, wave-minor
        }
        __syncthreads();
        if (w == 0) {                                   // the rows' counts become list offsets (one wave scan); one global atomic reserves the slots
            static_assert(K1F_E * K1F_NW <= 64, "one lane per row");
            const u32 c = lane < K1F_E * K1F_NW ? S.misc[lane] : 0u;
            const u32 inc = wave_incl_scan_dpp(c);
            const u32 run = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            u32 base = 0;
            if (lane == 0 && run) base = atomicAdd(&K1_RCNT(B, 0, b), run);
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            if (lane < K1F_E * K1F_NW) S.misc[lane] = base + inc - c;
        }
    }
    // the suffix-array slice (plain indices) and, in carry mode, the BWT bytes of these positions: final where the rotation is settled here,
    // overwritten by whoever settles it later (k1r_round, the lane kernels)
    {
        u8* U = B.U + (size_t)b * g.stride + pos0;
        for (u32 i = tid; i < cnt; i += K1F_BT) {
            const u32 v = S.sx[i];
            SA[i] = v & K1_SMASK;
            if (carry) U[i] = (u8)(v >> 24);
        }
    }
    __syncthreads();
    if (lists) {
        k1f_write_heads(HN, pos0, pos0 + cnt, [&](u32 p) { return k1f_bit(S.hb, p - pos0) || k1f_bit(lb, p - pos0); });
        u64* L = B.rlist[0] + (size_t)b * g.stride;
#pragma unroll
        for (int it = 0; it < K1F_E; it++) {
            const u32 q = (u32)it * K1F_BT + tid;
            if ((bal[it] >> lane) & 1ull) {
                const u32 idx = S.misc[(u32)it * K1F_NW + w] + (u32)__popcll(bal[it] & lt);
                const u32 v = S.sx[q];
                if (idx < g.stride)
                    L[idx] = carry ? K1C_MAKE(gsv[it] >> 8, gsv[it] & 0xFFu, v >> 24, v & K1_SMASK, pos0 + q)
                                   : (((u64)gsv[it] << 44) | ((u64)(v & K1_SMASK) << 22) | (u64)(pos0 + q));
            }
        }
    } else {
        k1f_write_heads(HN, pos0, pos0 + cnt, [&](u32 p) { return k1f_bit(S.hb, p - pos0); });
    }
}

// `cnt` <= K1F_CAP rotation indices from src[] sorted in LDS by the K1F_KEYB = 16 text bytes at depth `dm` (= depth mod n; `wide` false:
// by the first 8 of them): on return S.sx[] holds them in order and S.hb[] the heads of the groups of equal keys (bit 0 and the sentinel
// bits included).  Round 5: ONE sort on 16-byte keys.  Rounds 3-4 sorted by 8 bytes and then took 12 more bytes off every group that still
// tied (77 % of the rotations of text) in a second stage with its own gathers, bitmaps, group walks and ranking loops whose trip count
// was the size of the largest 8-byte group of a wave's rows: 1 750 VALU + 1 200 SALU instructions per wave of ~110 rotations (PMC), and
// the instruction stream, not memory latency, was what the kernel's 3.1 ms were made of.
//   stage 0  indices, then keys (all loads of a stage in flight together), keys to LDS in arrival order
//   stage 1  a LOCAL sample sort: K1F_LOVS x K of the slice's own keys ranked by counting, every K1F_LOVS-th a local splitter;
//            a key that fills a whole quantile gets a leaf of its own (nothing to rank there: heavy keys cannot blow a leaf up)
//   stage 2  leaf of every rotation (binary search over the splitters), slot inside the leaf by an LDS counter; the cells move to
//            leaf order IN PLACE (every thread still holds its own in registers), the leaf id rides in the index word
//   stage 3  every POSITION ranks its cell inside its leaf by counting, all lanes at once (the leaf's cells read four at a time,
//            lanes of one leaf read the same address = broadcast; neighbouring lanes sit in the same or the next leaf, so a wave's
//            trip count is that of a handful of leaves).  (key, slot) is a strict order: the ranks are a permutation, one carry
//            chain per candidate (k1f_acc_lt).  Cells past a leaf's end belong to later leaves - strictly greater keys - and
//            behind the last rotation sit four all-ones cells: no bounds masks in the loop.
//   stage 4  cells to their final places, heads = key differs from its predecessor's (one ballot per row of 64: no atomics).
//            (Measured and dropped: a second carry chain on the keys alone in the ranking loop tells whether an equal key sits in an
//            earlier slot, i.e. which cells open a group - no key moves, one barrier and the heads pass less, but k1f_bsort 2.10 -> 2.45 ms.)
#ifdef K1F_TRACE
#define K1F_SSTAMP(slot) do { if (trs && tid == 0) { const long long now_ = clock64(); atomicAdd(&trs[K1_STAT_FRONT_BIG + 1 + (slot)], (u32)((now_ - tprev_) >> 8)); tprev_ = now_; } } while (0)
#else
#define K1F_SSTAMP(slot) do { } while (0)
#endif
template <class KT>
__device__ __forceinline__ void k1f_sortk(const K1fL& S, const u8* T, u32 n, const u32* src, u32 cnt, u32 dm, u32* trs = nullptr) {
    typedef typename KT::T KeyT;
    KeyT* key = (KeyT*)S.key;
    KeyT* smpk = (KeyT*)S.smp;
    KeyT* spk = (KeyT*)S.sp;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
#ifdef K1F_TRACE
    long long tprev_ = clock64();
#endif
    u32* sx = S.sx;
    u32 v[K1F_E];
    KeyT k[K1F_E];
#pragma unroll
    for (int it = 0; it < K1F_E; it++) {
        const u32 i = (u32)it * K1F_BT + tid;
        v[it] = i < cnt ? src[i] : 0u;
    }
#pragma unroll
    for (int it = 0; it < K1F_E; it++) {
        const u32 i = (u32)it * K1F_BT + tid;
        u32 p = (v[it] & K1_SMASK) + dm;                   // (index words are packed: K1_SPACK)
        if (p >= n) p -= n;
        k[it] = i < cnt ? KT::load(T, p) : KT::zero();
    }
#pragma unroll
    for (int it = 0; it < K1F_E; it++) {
        const u32 i = (u32)it * K1F_BT + tid;
        if (i < cnt) key[i] = k[it];
    }
    if (tid < 4u) key[cnt + tid] = KT::ones();
    if (tid < K1F_LK) S.cnt2[tid] = 0;
    u32 K = 1u;
    while (K < K1F_LK && cnt >= K1F_LEAF * K) K <<= 1;
    __syncthreads();
    K1F_SSTAMP(0);
    u32 q[K1F_E];                                       // final position of the cell this thread ranks
    KeyT kk[K1F_E];
    u32 vv[K1F_E];
    if (K > 1u) {
        // stage 1: LS = LOVS * K samples (stratified over the arrival order), ranked by counting
        const u32 LS = K1F_LOVS * K;
        KeyT mys = KT::zero();
        if (tid < LS) { mys = key[(u32)((u64)tid * cnt / LS)]; smpk[tid] = mys; S.srank[tid] = 0; }
        __syncthreads();
        {   // thread t ranks sample t % LS against one `parts`-th of the samples, partial ranks summed in LDS
            const u32 parts = K1F_BT / LS > LS ? LS : K1F_BT / LS, si = tid % LS, part = tid / LS;   // powers of two; LS <= K1F_BT
            if (part < parts) {
                const KeyT mine = smpk[si];
                const u32 per = LS / parts, j0 = part * per;
                u32 r = 0;
                if (per >= 4u) {
                    for (u32 j = j0; j < j0 + per; j += 4u) {
                        KeyT c[4];
#pragma unroll
                        for (u32 u = 0; u < 4u; u++) c[u] = smpk[j + u];
#pragma unroll
                        for (u32 u = 0; u < 4u; u++) KT::acc_lt(r, c[u], j + u, mine, si);
                    }
                } else {
                    for (u32 j = j0; j < j0 + per; j++) KT::acc_lt(r, smpk[j], j, mine, si);
                }
                atomicAdd(&S.srank[si], r);
            }
        }
        __syncthreads();
        if (tid < LS) smpk[S.srank[tid]] = mys;        // (every read of the unsorted samples is behind the barrier)
        __syncthreads();
        if (tid + 1u < K) {                             // splitters; equal neighbours: the heavy-key rule
            const KeyT qq = smpk[(tid + 1u) * K1F_LOVS];
            KeyT sv = qq;
            if (tid >= 1u && KT::eq(smpk[tid * K1F_LOVS], qq) && !KT::is_ones(qq)) sv = KT::inc(qq);
            spk[tid] = sv;
        }
        __syncthreads();
        K1F_SSTAMP(1);
        // stage 2: leaf of every rotation (number of splitters <= key), slot inside the leaf by an LDS counter
        u32 L[K1F_E];
#pragma unroll
        for (int it = 0; it < K1F_E; it++) {
            const u32 i = (u32)it * K1F_BT + tid;
            L[it] = 0xFFFFFFFFu;
            if (i < cnt) {
                u32 pos = 0;
                for (u32 step = K >> 1; step >= 1u; step >>= 1) pos = KT::step_ge(pos, step, k[it], spk[pos + step - 1u]);
                L[it] = (pos << 16) | atomicAdd(&S.cnt2[pos], 1u);
            }
        }
        __syncthreads();
        if (w == 0) {                                   // leaf starts; a leaf between the splitters v and v + 1 holds one key only
            const u32 c = lane < K ? S.cnt2[lane] : 0u;
            const u32 inc = wave_incl_scan_u32(c);
            bool pure = false;
            if (lane >= 1u && lane + 1u < K) pure = KT::eq(spk[lane], KT::inc(spk[lane - 1u]));
            __builtin_amdgcn_wave_barrier();
            if (lane < K1F_LK) S.off2[lane] = lane < K ? ((inc - c) | (pure ? 0x80000000u : 0u)) : cnt;
            if (lane == 0) S.off2[K1F_LK] = cnt;
        }
        __syncthreads();
        // the cells move to leaf order IN PLACE: nobody reads the arrival-order array any more (own cells are in registers);
        // the samples' scratch (it aliases sx[]) is dead as well
#pragma unroll
        for (int it = 0; it < K1F_E; it++)
            if (L[it] != 0xFFFFFFFFu) {
                const u32 leaf = L[it] >> 16, p = (S.off2[leaf] & 0x7FFFFFFFu) + (L[it] & 0xFFFFu);
                key[p] = k[it];
                sx[p] = v[it];
                S.lf[p] = (u8)leaf;
            }
        __syncthreads();
        K1F_SSTAMP(2);
        // stage 3: position i ranks its cell inside its leaf
#pragma unroll
        for (int it = 0; it < K1F_E; it++) {
            const u32 i = (u32)it * K1F_BT + tid;
            q[it] = 0xFFFFFFFFu;
            if (i < cnt) {
                const u32 vl = sx[i], leaf = S.lf[i];
                const u32 oo = S.off2[leaf], o = oo & 0x7FFFFFFFu, e = S.off2[leaf + 1u] & 0x7FFFFFFFu;
                const KeyT mine = key[i];
                kk[it] = mine;
                vv[it] = vl;
                u32 less = 0;
                if (!(oo >> 31)) {
                    for (u32 j = o; j < e; j += 4u) {
                        KeyT c[4];
#pragma unroll
                        for (u32 u = 0; u < 4u; u++) c[u] = key[j + u];
#pragma unroll
                        for (u32 u = 0; u < 4u; u++) KT::acc_lt(less, c[u], j + u, mine, i);
                    }
                } else {
                    less = i - o;
                }
                q[it] = o + less;
            }
        }
    } else {
        // one leaf: the whole slice (fewer than K1F_LEAF rotations), ranked where it arrived
#pragma unroll
        for (int it = 0; it < K1F_E; it++) {
            const u32 i = (u32)it * K1F_BT + tid;
            q[it] = 0xFFFFFFFFu;
            if (i < cnt) {
                kk[it] = k[it];
                vv[it] = v[it];
                u32 less = 0;
                for (u32 j = 0; j < cnt; j += 4u) {
                    KeyT c[4];
#pragma unroll
                    for (u32 u = 0; u < 4u; u++) c[u] = key[j + u];
#pragma unroll
                    for (u32 u = 0; u < 4u; u++) KT::acc_lt(less, c[u], j + u, k[it], i);
                }
                q[it] = less;
            }
        }
    }
    __syncthreads();
    K1F_SSTAMP(3);
    // stage 4: cells to their final places; heads
#pragma unroll
    for (int it = 0; it < K1F_E; it++)
        if (q[it] != 0xFFFFFFFFu) { key[q[it]] = kk[it]; sx[q[it]] = vv[it]; }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < K1F_E; it++) {
        const u32 r0 = (u32)it * K1F_BT + w * 64u, p = r0 + lane;
        bool head = true;
        if (p < cnt && p > 0u) head = !KT::eq(key[p], key[p - 1u]);
        const u64 bal = __ballot(head);
        if (lane == 0) { S.hb[r0 >> 5] = (u32)bal; S.hb[(r0 >> 5) + 1u] = (u32)(bal >> 32); }
    }
    if (tid < 2u) S.hb[K1F_C / 32u + tid] = 0xFFFFFFFFu;
    __syncthreads();
    K1F_SSTAMP(4);
}

__device__ __forceinline__ void k1f_sort128(const K1fL& S, const u8* T, u32 n, const u32* src, u32 cnt, u32 dm, bool wide, u32* trs = nullptr) {
    if (wide) k1f_sortk<K1fK16>(S, T, n, src, cnt, dm, trs);
    else k1f_sortk<K1fK8>(S, T, n, src, cnt, dm, trs);
}

// The __shared__ arrays of a bucket-sort workgroup and their views (macro: __shared__ must be declared in the kernel)
#define K1F_DECLARE_LDS(S)                                                                                                           \
    __shared__ uint4 key128[K1F_C];                                                                                                  \
    __shared__ __attribute__((aligned(16))) u32 sx[K1F_C];                                                                           \
    __shared__ u32 cnt2s[K1F_LK], off2s[K1F_LK + 1];                                                                                 \
    __shared__ u32 hbits[K1F_HW], lbits[K1F_HW];                                                                                     \
    __shared__ u8 lfs[K1F_C];                                                                                                        \
    __shared__ u32 misc[K1F_E * K1F_NW + 8];                                                                                         \
    static_assert((K1F_LS + K1F_LK) * 16 + K1F_LS * 4 <= K1F_C * 4, "the local sample sort's scratch fits sx[]");                    \
    K1fL S;                                                                                                                          \
    S.key = key128; S.sx = sx; S.smp = (uint4*)sx; S.sp = (uint4*)sx + K1F_LS; S.srank = sx + (K1F_LS + K1F_LK) * 4;                 \
    S.cnt2 = cnt2s; S.off2 = off2s; S.hb = hbits; S.lb = lbits; S.lf = lfs; S.misc = misc;

// One workgroup per bucket (a bucket is a KEY RANGE: everything that ties on its first 8 bytes, or deeper, is inside).
//   1. k1f_sort128: the bucket sorted by its first 16 bytes in LDS (`wide`; else 8), groups of equal keys marked;
//   2. k1f_flush: suffix-array slice and head bits written ONCE; what still ties goes to the refinement rounds' lists, and
//      what could not be handled here (groups above K1F_GBIG rotations) to the task levels.
// Buckets beyond LDS (cnt > K1F_CAP: an unlucky sample, a moderately heavy key) and buckets of ONE 8-byte key beyond LDS
// (HTML-like input: a quarter of all rotations; round 6: fewer and smaller - a heavy key's rotations are spread over several
// buckets by k1f_hist) are level-0 tasks, also when no text stage follows (round 6: every group is to be 16 bytes deep, the
// doubling rounds start there); a bucket of ONE 16-byte key beyond LDS (fp16) is a group as it stands.  `lists` = 0: 8-byte
// keys, no lists, no tasks but the oversize buckets (linear mode; cyclic mode with the text stages off).
// `purerot_max`: with more rotations than this in one-key buckets (counted by k1f_scan) the text stages are skipped
// altogether (CJS_DEEP_BIG_DIV = 8: HTML-like input, whose ties of hundreds of bytes prefix doubling settles faster).
// One bucket per workgroup.  (Round 5, measured and dropped, ms for the kernel on enwik against 2.08: neighbouring buckets - adjacent key ranges - sorted
// TOGETHER while they fit the LDS, pairs 3.32, fours 3.20; four buckets one after the other in a persistent workgroup 3.62: what this kernel lives on is the
// number of INDEPENDENT workgroups in flight, 7 per CU, each a short chain of dependent loads and barriers.)
__global__ __launch_bounds__(K1F_BT, K1F_MINW) void k1f_bsort(K1Buf B, BatchGeom g, u32 iters, u32 lists, u32 purerot_max, u32 carry) {
    u32 b, d;
    if (!xcd_block_tile(g.nb, b, d)) return;
    const u32 n = B.nfront[b];
    if (n == 0) return;
    const u32* fs = B.fstart + (size_t)b * (K1F_NB + 1);
    const u32 start = fs[d], end = fs[d + 1];
    if (end <= start) return;
    const u32 cnt = end - start;
    const u32 tid = threadIdx.x;
    const u8* T = B.T + (size_t)b * g.tstride;
    const u32* SB = B.SB + (size_t)b * g.stride + start;
    u32* HN = B.HN + (size_t)b * g.hstride;
    const u64* sp = B.fsplit + (size_t)b * K1F_NB;
    // a bucket between the splitters v and v+1 holds one key only
    // (K1F_SUBBUCKETS: the bucket of a heavy key and the - formerly empty - ones behind it that share its rotations, marked by k1f_sample)
    const bool pure = K1F_SUBBUCKETS ? B.fsub[(size_t)b * K1F_NB + d] != 0u : (d > 0u && d < K1F_NB - 1u && sp[d] == sp[d - 1u] + 1u);
    const bool deepen = lists != 0u && B.stats[K1_STAT_PUREROT] <= purerot_max;
    // 16-byte keys whenever the caller wants the text stages (cyclic mode) - also when the predictor then skips them (HTML-like input): the
    // doubling rounds start from h = 8 either way, but on groups that are already 16 bytes deep where a bucket fit the LDS (E8S-A: k1f_bsort
    // 1.17 -> 1.41 ms, the doubling stage 7.6 -> 6.4 ms)
    const bool wide = lists != 0u && iters != 0u;
    const bool pure16 = K1F_SUBBUCKETS && B.fp16[(size_t)b * K1F_NB + d] != 0u;     // ONE 16-byte key (a heavy continuation of a heavy key)
    if (pure16 && !deepen && wide && cnt > K1F_CAP) {
        // beyond LDS, all rotations share 16 bytes, no text stage behind this one: a single group as it stands, as deep as the doubling rounds' first step
        u32* SA = B.SA + (size_t)b * g.stride + start;
        for (u32 i = tid; i < cnt; i += K1F_BT) SA[i] = SB[i] & K1_SMASK;
        k1f_write_heads(HN, start, end, [&](u32 p) { return p == start; });
        return;
    }
    if (cnt > K1F_CAP && (deepen || !pure || (K1_DEEP_START && wide))) {
        // beyond LDS: a level-0 task (a one-key bucket starts 8 bytes deep); its indices stay in SB.  (Round 6: a one-key bucket beyond LDS also
        // without text stages behind it - partitioned by its next 8 bytes and sorted in LDS it enters the doubling rounds 16 or 24 bytes deep)
        if (tid == 0) {
            k1f_push_task(B, 0u, b, start, cnt, (pure16 ? 16u : (pure ? 8u : 0u)) | K1F_TASK_SB);
            if (!pure) atomicAdd(&B.stats[K1_STAT_FRONT_BIG], 1u);
            atomicAdd(&B.stats[K1_STAT_BIGROT + (d & 7u)], cnt);
        }
        return;
    }
    if (pure && !deepen && (!wide || (!K1_DEEP_START && cnt > K1F_CAP))) {
        // one 8-byte key, no text stage behind this one, and no 16-byte sort either (linear mode, or beyond LDS): a single group as it stands
        u32* SA = B.SA + (size_t)b * g.stride + start;
        for (u32 i = tid; i < cnt; i += K1F_BT) SA[i] = SB[i] & K1_SMASK;           // (a group: the doubling rounds finish this block, k1_finish gathers its bytes)
        k1f_write_heads(HN, start, end, [&](u32 p) { return p == start; });
        if (tid == 0 && cnt > 64u) atomicAdd(&B.stats[K1_STAT_BIGROT + (d & 7u)], cnt);         // one big group (trace only)
        return;
    }
    K1F_DECLARE_LDS(S)
    k1f_sort128(S, T, n, SB, cnt, 0u, wide, B.stats);
#ifdef K1F_TRACE
    long long tprev_ = clock64();
#endif
    k1f_flush(S, B, g, b, start, cnt, deepen, 0u, wide ? K1F_KEYB : 8u, carry != 0u);
    K1F_STAMP(5);
}

// The task levels: slices that a bucket-sort workgroup could not finish in LDS.  Level L reads the tasks level L - 1 (or
// k1f_bsort) pushed; one workgroup per task:
//   * a slice of up to K1F_CAP rotations (sharing `depth` bytes) is sorted in LDS by the 16 bytes at `depth` (k1f_sortk)
//     and flushed like a bucket: what still ties goes to the refinement rounds' lists, groups above K1F_GBIG
//     rotations become tasks of the next level, 16 bytes deeper;
//   * a longer slice is PARTITIONED like a block by the front end: K1F_PS = 512 of its own keys at `depth` sorted in LDS, every
//     k-th a splitter (a key that fills more than one quantile gets a sub-bucket of its own, 8 bytes deeper), bucket ids
//     (one byte per rotation, in the slice of KA), an LDS histogram, the indices scattered into the other of the two index
//     arrays (SB <-> SA); every sub-bucket is a task of the next level.
// HTML-like input (E8S-A) has a quarter of its rotations in 8-byte groups of more than 1024 members and another 10-15 % in
// groups of 257..1024: rounds 1-2 left all of them to prefix doubling from h = 8 (13 of 22 ms).  `last` (the last level
// launched): whatever arrives is sorted by 8 bytes at its depth and left at that (the doubling rounds take it from there):
// in LDS if it fits, else by stable LSD passes through global memory (one digit byte gathered per pass; slow, never seen).
__global__ __launch_bounds__(K1F_BT) void k1f_task(K1Buf B, BatchGeom g, u32 level, u32 iters, u32 lists, u32 purerot_max, u32 last, u32 carry) {
    K1F_DECLARE_LDS(S)
    // scratch of the partition and of the last level's LSD passes: views into the key cells (K1F_C * 16 bytes)
    u64* key = (u64*)key128;                            // [K1F_PS + K1F_PB] u64
    u32* key1 = (u32*)(key + K1F_PS + K1F_PB);          // [K1F_PS] u32
    static_assert((K1F_PS + K1F_PB) * 8 + K1F_PS * 4 <= K1F_C * 16 && 512 * 4 <= K1F_C * 16, "task scratch fits the key cells");
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    // the workgroups the dispatcher places on XCD x (blockIdx.x mod 8, as in xcd_block_tile) walk the tasks of the blocks x, x + 8, ...
    const u32 tli = level * 8u + (blockIdx.x & 7u) % B.btaskLists, cap8 = B.btaskCap;      // (batches of fewer than 8 blocks: the idle XCDs' workgroups share the lists in use)
    const u32 tshare = (8u + B.btaskLists - 1u - (blockIdx.x & 7u) % B.btaskLists) / B.btaskLists;     // XCDs that walk this list
    u32 ntask = B.bcnt[tli];
    if (ntask > cap8) ntask = cap8;
    const bool deepen = lists != 0u && B.stats[K1_STAT_PUREROT] <= purerot_max;
    for (u32 li = (blockIdx.x >> 3) * tshare + (blockIdx.x & 7u) / B.btaskLists; li < ntask; li += (gridDim.x >> 3) * tshare) {
        __syncthreads();
        const uint4 tk = B.btask[(size_t)tli * cap8 + li];
        const u32 b = tk.x, pos = tk.y, len = tk.z, depth = tk.w & ~(K1F_TASK_SB | K1F_TASK_PLAIN);
        const bool inSB = (tk.w & K1F_TASK_SB) != 0u;
        const u32 n = B.nfront[b];
        const u8* T = B.T + (size_t)b * g.tstride;
        u32* SBs = B.SB + (size_t)b * g.stride + pos;
        u32* SAs = B.SA + (size_t)b * g.stride + pos;
        u32* HN = B.HN + (size_t)b * g.hstride;
        const u32* src = inSB ? SBs : SAs;
        const u32 dm = depth % n;
        if (carry && (tk.w & K1F_TASK_PLAIN)) {
            // a group that k1f_flush left in the suffix array (plain indices, its BWT bytes in U): packed index words from here on, like every other slice
            const u8* Ub = B.U + (size_t)b * g.stride + pos;
            for (u32 i = tid; i < len; i += K1F_BT) SAs[i] = K1_SPACK(SAs[i], Ub[i]);
            __threadfence_block();
            __syncthreads();
        }
        const bool widem = lists != 0u && iters != 0u && !last;       // 16-byte keys also without lists behind them (see k1f_bsort)
        const bool wide_any = K1_DEEP_START && lists != 0u && iters != 0u;   // the doubling rounds want to start at h = 16: no group shallower than that
        if (!deepen && depth && (!widem || (len > K1F_CAP && (!wide_any || depth >= K1F_KEYB)))) {
            // no text stages behind this one (linear mode: BWT.bwtransform / suffixsort sort suffixes by the first 8 bytes only; cyclic
            // mode with the predictor on: HTML-like input, the doubling rounds take over from 8 bytes): a sub-bucket of ONE key is one
            // group, as the one-key buckets of k1f_bsort are.  (Until round 4 the cyclic case went on partitioning such slices 8 bytes
            // deeper, level after level - 0.8 ms per E8S-A sub-batch, one workgroup alone on a 50 000-rotation slice - for an order the
            // doubling rounds do not need.)
            for (u32 i = tid; i < len; i += K1F_BT) SAs[i] = src[i] & K1_SMASK;     // (stays a group: the doubling rounds finish this block)
            if (tid == 0) atomicOr(&HN[pos >> 5], 1u << (pos & 31u));
            if (tid == 0 && wide_any && depth < K1F_KEYB && len > 1u) K1D_SHALLOW(B) = 1u;
            continue;
        }
        if (len <= K1F_CAP) {
            if (len == 1u) {
                if (tid == 0) {
                    const u32 v = src[0];
                    SAs[0] = v & K1_SMASK;
                    if (carry) B.U[(size_t)b * g.stride + pos] = (u8)(v >> 24);
                    atomicOr(&HN[pos >> 5], 1u << (pos & 31u));
                }
                continue;
            }
            const bool go = deepen && !last;
            const bool wide = widem;
            if (tid == 0 && wide_any && !wide && depth + 8u < K1F_KEYB) K1D_SHALLOW(B) = 1u;     // (the last level sorts by 8 bytes)
            k1f_sort128(S, T, n, src, len, dm, wide);
            k1f_flush(S, B, g, b, pos, len, go, level + 1u, depth + (wide ? K1F_KEYB : 8u), carry != 0u);
            continue;
        }
        if (!last) {
            // ---- partition.  K1F_PS samples of the slice's keys at `depth`, ranked by counting; splitters
            u64* smp = key;                                 // [K1F_PS]
            u64* spl = key + K1F_PS;                        // [K1F_PB] sub-bucket j holds the keys in [spl[j-1], spl[j])
            u32* rk = key1;                                 // [K1F_PS] ranks; then [K1F_PB] histogram, [K1F_PB] cursors
#ifndef K1F_SUBT
#define K1F_SUBT 320u                                       // rotations per sub-bucket a partition aims at
#endif
            u32 nbk = (len + K1F_SUBT - 1u) / K1F_SUBT;
            if (nbk > K1F_PB - 1u) nbk = K1F_PB - 1u;
            // samples: ~26 per sub-bucket, at most K1F_PS (round 6: always K1F_PS = 512 until then - ranking 512 samples against 512 was most of
            // what a slice of 2 000 rotations cost, and E8S-A's one-key buckets beyond LDS are 9 900 such slices per 10^8 bytes)
            u32 ns = (len / 12u + 3u) & ~3u;
            if (ns < 64u) ns = 64u;
            if (ns > K1F_PS) ns = K1F_PS;
            for (u32 i = tid; i < ns; i += K1F_BT) {
                const u32 at = (u32)(((u64)i * len + (k1f_hash(i, pos) % len)) / ns) % len;      // stratified, hashed
                u32 p = (src[at] & K1_SMASK) + dm;
                if (p >= n) p -= n;
                smp[i] = k1f_load_be64(T, p);
            }
            __syncthreads();
            for (u32 i = tid; i < ns; i += K1F_BT) {
                const u64 mine = smp[i];
                u32 r = 0;
                for (u32 j = 0; j < ns; j += 4u) {
                    u64 c[4];
#pragma unroll
                    for (u32 u = 0; u < 4u; u++) c[u] = smp[j + u];
#pragma unroll
                    for (u32 u = 0; u < 4u; u++) r += (c[u] < mine || (c[u] == mine && j + u < i)) ? 1u : 0u;
                }
                rk[i] = r;
            }
            __syncthreads();
            u64 mine2[(K1F_PS + K1F_BT - 1) / K1F_BT];
            for (u32 i = tid, k = 0; i < ns; i += K1F_BT, k++) mine2[k] = smp[i];
            __syncthreads();
            for (u32 i = tid, k = 0; i < ns; i += K1F_BT, k++) smp[rk[i]] = mine2[k];
            __syncthreads();
            for (u32 j = tid; j < K1F_PB; j += K1F_BT) {
                u64 v = ~0ull;
                if (j + 1u < nbk) {
                    const u64 q = smp[(u32)(((u64)(j + 1u) * ns) / nbk)];
                    v = q;
                    if (j >= 1u && smp[(u32)(((u64)j * ns) / nbk)] == q && q != ~0ull) v = q + 1u;     // heavy key: [q, q+1) is a sub-bucket of its own
                }
                spl[j] = v;
            }
            __syncthreads();
            u32* hist = key1;                               // [K1F_PB]
            u32* cur = key1 + K1F_PB;                       // [K1F_PB]
            for (u32 j = tid; j < K1F_PB; j += K1F_BT) hist[j] = 0;
            __syncthreads();
            u8* ids = (u8*)(B.KA + (size_t)b * g.stride + pos);       // one byte per rotation of the slice (KA is free after k1f_scatter)
            // K1F_PU rotations per thread and step, all their loads in flight together (round 6): a slice of 50 000 rotations is ONE workgroup's
            // work, two dependent global round trips per rotation - as a plain loop 200 of them one after the other, and the launch of a level
            // lasts as long as its longest task (E8S-A, level 0: 1.30 ms for 22.8 M rotations while most of the GPU idles)
#ifndef K1F_PU
#define K1F_PU 8u
#endif
            for (u32 i0 = tid; i0 < len; i0 += K1F_BT * K1F_PU) {
                u32 pp[K1F_PU];
                u64 kq[K1F_PU];
#pragma unroll
                for (u32 u = 0; u < K1F_PU; u++) {
                    const u32 i = i0 + u * K1F_BT;
                    u32 p = (src[i < len ? i : tid] & K1_SMASK) + dm;
                    if (p >= n) p -= n;
                    pp[u] = p;
                }
#pragma unroll
                for (u32 u = 0; u < K1F_PU; u++) kq[u] = k1f_load_be64(T, pp[u]);
#pragma unroll
                for (u32 u = 0; u < K1F_PU; u++) {
                    const u32 i = i0 + u * K1F_BT;
                    if (i >= len) break;
                    u32 id = 0;
#pragma unroll
                    for (u32 step = K1F_PB / 2u; step >= 1u; step >>= 1)
                        if (id + step - 1u < K1F_PB - 1u && spl[id + step - 1u] <= kq[u]) id += step;      // spl[K1F_PB - 1] is never read
                    ids[i] = (u8)id;
                    atomicAdd(&hist[id], 1u);
                }
            }
            __syncthreads();
            {
                const u32 c = tid < K1F_PB ? hist[tid] : 0u;
                const u32 ex = block_excl_scan_256(c, (u32*)sx);
                if (tid < K1F_PB) cur[tid] = ex;
            }
            __syncthreads();
            u32* dst = inSB ? SAs : SBs;
            __threadfence_block();
            for (u32 i0 = tid; i0 < len; i0 += K1F_BT * K1F_PU) {
                u32 idv[K1F_PU], sv2[K1F_PU];
#pragma unroll
                for (u32 u = 0; u < K1F_PU; u++) {
                    const u32 i = i0 + u * K1F_BT, ic = i < len ? i : tid;
                    idv[u] = ids[ic];
                    sv2[u] = src[ic];
                }
#pragma unroll
                for (u32 u = 0; u < K1F_PU; u++) {
                    if (i0 + u * K1F_BT >= len) break;
                    const u32 at = atomicAdd(&cur[idv[u]], 1u);
                    dst[at] = sv2[u];
                }
            }
            __threadfence_block();
            __syncthreads();
            // sub-buckets -> tasks of the next level (cur[j] is now the END of sub-bucket j)
            for (u32 j = tid; j < K1F_PB; j += K1F_BT) {
                const u32 c = hist[j];
                if (!c) continue;
                const u32 e = cur[j];
                const bool heavy = j > 0u && j < K1F_PB - 1u && spl[j] == spl[j - 1u] + 1u;     // one key: 8 bytes deeper
                k1f_push_task(B, level + 1u, b, pos + e - c, c, (heavy ? depth + 8u : depth) | (inSB ? 0u : K1F_TASK_SB));
            }
            continue;
        }
        // ---- last level, beyond LDS: stable LSD passes on the 8 bytes at `depth`, heads by comparing them
        if (tid == 0 && wide_any && depth + 8u < K1F_KEYB) K1D_SHALLOW(B) = 1u;
        {
            u32* dstart = (u32*)key;                        // [256]
            u32* sh = dstart + 256;                         // [256]
            u32* single = misc;
            u32* bufs[2] = {inSB ? SBs : SAs, inSB ? SAs : SBs};
            int cur = 0;
            for (u32 pass = 0; pass < 8u; pass++) {
                const u32* sp = bufs[cur];
                u32* dp = bufs[cur ^ 1];
                const u32 dmp = (depth + 7u - pass) % n;
                if (tid < 256u) dstart[tid] = 0;
                if (tid == 0) *single = 0;
                __syncthreads();
                for (u32 i = tid; i < len; i += K1F_BT) { u32 p = (sp[i] & K1_SMASK) + dmp; if (p >= n) p -= n; atomicAdd(&dstart[T[p]], 1u); }
                __syncthreads();
                const u32 c = tid < 256u ? dstart[tid] : 0u;
                if (c == len) *single = 1;
                __syncthreads();
                if (*single) { __syncthreads(); continue; }  // every rotation has the same byte here (uniform)
                const u32 ex = block_excl_scan_256(c, sh);
                if (tid < 256u) dstart[tid] = ex;
                __syncthreads();
                if (w == 0) {
                    const u64 lt = lanemask_lt();
                    for (u32 r0 = 0; r0 < len; r0 += 64u) {
                        const u32 i = r0 + lane;
                        const bool valid = i < len;
                        const u32 v = valid ? sp[i] : 0u;
                        u32 p = (v & K1_SMASK) + dmp;
                        if (p >= n) p -= n;
                        const u32 dg = valid ? T[p] : 0u;
                        const u64 m = match_any(dg, 8, valid);
                        const u32 rank = (u32)__popcll(m & lt), c2 = (u32)__popcll(m);
                        const u32 bs = valid ? dstart[dg] : 0u;
                        __builtin_amdgcn_wave_barrier();
                        if (valid) dp[bs + rank] = v;
                        if (valid && rank == 0) dstart[dg] = bs + c2;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                __threadfence_block();
                __syncthreads();
                cur ^= 1;
            }
            {   // packed words -> plain indices in the suffix array (in place when the last pass ended there: a thread rewrites only its own cells)
                const u32* fin = bufs[cur];
                for (u32 i = tid; i < len; i += K1F_BT) {
                    const u32 v = fin[i];
                    SAs[i] = v & K1_SMASK;
                    if (carry) B.U[(size_t)b * g.stride + pos + i] = (u8)(v >> 24);
                }
                __threadfence_block();
                __syncthreads();
            }
            k1f_write_heads(HN, pos, pos + len, [&](u32 p) {
                if (p == pos) return true;
                u32 a = SAs[p - pos] + dm, c = SAs[p - pos - 1u] + dm;
                if (a >= n) a -= n;
                if (c >= n) c -= n;
                return k1f_load_be64(T, a) != k1f_load_be64(T, c);
            });
        }
    }
}

// ---------------------------------------------------------------------------------------------
// list-driven refinement rounds
// ---------------------------------------------------------------------------------------------
// What k1f_bsort leaves tied (groups of 2..K1F_GBIG = 256 rotations, sharing `depth` bytes) sits in per-block lists, one
// 8-byte entry per rotation (K1E_MAKE, k1_bwt.h):  (group length - 1) << 52 | index in the group << 44 | rotation index << 22 | suffix-array
// position;  a group = consecutive entries.  A round takes K1R_STEP = 24 more text bytes off every listed rotation: a
// workgroup owns the groups that START in its K1R_T entries (it reads K1R_W = 256 entries ahead for the tail of the last
// one; an entry knows where its group starts, so there is no bitmap), fetches the keys (a dwordx4 + a dwordx3 load per
// rotation; all tiles of a block run on one XCD: the text is in that L2), ranks every rotation inside its group by counting
// (smaller keys, equal keys before it, equal keys in all), and then
//   - a rotation that ends up alone is FINAL: its index goes to the suffix array (its position was marked a head by
//     k1f_bsort already),
//   - the others go, compacted and in their new order, to the next round's list (their slot: the survivors before them in
//     the tile, from a bitmap by new position + one atomic per workgroup).
// Every lane of every round works on a rotation that still ties - which the in-bucket iterations of k1f_bsort (rows of
// mostly idle lanes after the second iteration) and the lane kernels (one lane per group, serial) could not offer.
// The kernel is a software pipeline over the tiles a workgroup walks (t0, t0 + gridDim.x, ...): its first version ran one
// tile per workgroup with four DEPENDENT memory round trips (list -> text -> list-slot atomic -> stores), its waves parked
// 78 % of their cycles (PMC SQ_WAIT_ANY).  Now, while tile k is ranked in LDS, the entries of tile k+2 and the text of tile
// k+1 are on their way (they live in registers until their turn), and the slot reservation of tile k-1 returns: the
// survivors of a tile are written one iteration late.  Three barriers per tile.
// The LAST round (`final`) writes what still ties (long repeats, identical rotations) to the suffix array, clears the
// head bits of its non-heads, and hands groups of 2..8 to the second pass of the lane kernels (k1_deep_pairs<true> /
// k1_deep_small<true>: up to CJS_DEEP_LANE_CAP bytes); what they leave, and bigger groups, the doubling rounds of k1_run take.
// less += (c < m) for 24-byte keys held as three big-endian u64 (c0 the most significant): the borrow of c - m, one chain
__device__ __forceinline__ void k1r_acc_lt192(u32& less, u64 c0, u64 c1, u64 c2, u64 m0, u64 m1, u64 m2) {
#if defined(__AMDGCN__)
    u32 t;
    asm("v_sub_co_u32 %1, vcc, %2, %3\n\t"
        "v_subb_co_u32 %1, vcc, %4, %5, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %6, %7, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %8, %9, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %10, %11, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %12, %13, vcc\n\t"
        "v_addc_co_u32 %0, vcc, 0, %0, vcc"
        : "+v"(less), "=&v"(t)
        : "v"((u32)c2), "v"((u32)m2), "v"((u32)(c2 >> 32)), "v"((u32)(m2 >> 32)), "v"((u32)c1), "v"((u32)m1), "v"((u32)(c1 >> 32)), "v"((u32)(m1 >> 32)),
          "v"((u32)c0), "v"((u32)m0), "v"((u32)(c0 >> 32)), "v"((u32)(m0 >> 32))
        : "vcc");
#else
    less += (c0 < m0 || (c0 == m0 && (c1 < m1 || (c1 == m1 && c2 < m2)))) ? 1u : 0u;
#endif
}

#ifndef K1R_T
#define K1R_T 768u           // entries a workgroup owns per step (round 5: 768 - 24.6 KB of LDS, six workgroups per CU instead of five: the two-stream step 7.40 -> 7.24 ms (median of 8), the kernel
                                // alone 1.78 -> 1.75; round 3: 512 -> 1024: 80 % of the 1280 slots of a step are owned instead of 67 %,
#endif                          // half as many barriers and pipeline prologues per entry; 10^8-byte enwik, ms per step with 512 / 768 / 1024 / 1280 / 1792: 9.7 / 9.10 / 9.03 / 9.10 / 9.49)
#define K1R_W K1F_GBIG
#define K1R_N (K1R_T + K1R_W)
#define K1R_ROWS (K1R_N / 64u)
#define K1R_RPW (K1R_ROWS / 4u)                         // rows per wave
#define K1R_SW (K1R_N / 32u)                            // words of the survivor bitmap
static_assert(K1R_ROWS % 4u == 0 && K1R_SW <= 64u && K1F_GBIG <= 256u, "rows are dealt to four waves; one wave scans the bitmap; 8-bit group fields");
#define K1R_POS(e) K1E_POS(e)
#define K1R_S(e) K1E_S(e)
#define K1R_IDX(e) K1E_IDX(e)
#define K1R_LEN(e) K1E_LEN(e)
#ifndef K1R_MINW
#ifndef K1R_MINW
#define K1R_MINW 4                                      // waves per SIMD the register allocation is held to
#endif
#endif

// CARRY: the entries are in the K1C layout (20-bit fields and the byte in front of the rotation): whoever settles a rotation writes its BWT byte next to the suffix-array entry
template <bool CARRY>
__global__ __launch_bounds__(256, K1R_MINW) void k1r_round(K1Buf B, BatchGeom g, u32 round, u32 depth, u32 final_host) {
    auto EPOS = [](u64 e) { return CARRY ? K1C_POS(e) : K1E_POS(e); };
    auto ES = [](u64 e) { return CARRY ? K1C_S(e) : K1E_S(e); };
    auto EIDX = [](u64 e) { return CARRY ? K1C_IDX(e) : K1E_IDX(e); };
    auto ELEN = [](u64 e) { return CARRY ? K1C_LEN(e) : K1E_LEN(e); };
    u32 b, t0;
    if (!xcd_block_tile(g.nb, b, t0)) return;
    u32 cnt = K1_RCNT(B, round, b);
    if (cnt > g.stride) cnt = g.stride;
    if (t0 * K1R_T >= cnt) return;
    // The last round of a block is the one the host launched last - or the first one that finds its list (nearly) as long as
    // the round before found it: ties that 24 more bytes do not resolve are long repeats (tiled or HTML-like input: 11 rounds
    // over all of a block's rotations, 19 of 47 ms on `200k text tiled`), which the doubling rounds settle in log steps.  The
    // same counters for every workgroup of the block: one decision.
    // Only while a sizeable part of the block still ties: the last few hundred entries of a text block (boilerplate passages) are
    // cheaper to walk to the end (the lane kernels take them) than to rank the whole block for (k1d_build).
    u32 final = final_host;
    const u32 n = B.nfront[b];
    if (round >= 1u && cnt >= n / 16u) {
        const u32 prev = K1_RCNT(B, round - 1u, b);
        if ((u64)cnt * 8u > (u64)prev * 7u) final = 1u;
    }
    const u8* T = B.T + (size_t)b * g.tstride;
    const u64* Lin = B.rlist[round & 1u] + (size_t)b * g.stride;
    u64* Lout = B.rlist[(round & 1u) ^ 1u] + (size_t)b * g.stride;
    u32* ocnt = &K1_RCNT(B, round + 1u, b);
    u32* SA = B.SA + (size_t)b * g.stride;
    u8* U = B.U + (size_t)b * g.stride;
    u32* HN = B.HN + (size_t)b * g.hstride;
    __shared__ u64 kA[K1R_N + 2], kB[K1R_N + 2], kC[K1R_N + 2];   // (+2: the ranking loop reads cells in pairs, one past a group's end)
    __shared__ u32 sb[2][K1R_SW], pre[2][K1R_SW + 1];
    __shared__ u32 obase;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32 dm = depth % n;
    const u32 G = gridDim.x;
    auto load_tile = [&](u32 t, u64 (&e)[K1R_RPW]) {
        const u32 e0 = t * K1R_T;
#pragma unroll
        for (u32 it = 0; it < K1R_RPW; it++) {
            const u32 i = (it * 4u + w) * 64u + lane;
            // beyond the list: a group of one, never owned.  The load itself is UNCONDITIONAL (of entry 0 then): behind a branch the compiler
            // cannot count the loads in flight, and every s_waitcnt vmcnt of the loop became "all of them" - the entries requested at the
            // top of a step were waited for a few instructions later, at the first use of the text requested a step earlier (round 5)
            const bool in = e0 < cnt && i < cnt - e0;
            const u64 x = Lin[in ? e0 + i : 0u];
            e[it] = in ? x : 0ull;
        }
    };
    // owned: the group starts inside the tile's first K1R_T entries
    auto own = [&](u64 e, u32 i) { const u32 gs = i - EIDX(e); return ELEN(e) >= 2u && gs < K1R_T; };     // (i < idx wraps to a huge gs)
    // raw text of the owned entries of a tile (7 dwords each, decoded when the tile's turn comes)
    auto gather = [&](const u64 (&e)[K1R_RPW], u32 (&d)[K1R_RPW][7]) {
#pragma unroll
        for (u32 it = 0; it < K1R_RPW; it++) {
            const u32 i = (it * 4u + w) * 64u + lane;
            u32 p = ES(e[it]) + dm;
            if (p >= n) p -= n;
            if (!own(e[it], i)) p = 0u;                 // (unconditional as well: the block's first bytes, one line for the whole wave)
            __builtin_memcpy(d[it], __builtin_assume_aligned(T + (p & ~3u), 4), 28);
        }
    };
    u64 eC[K1R_RPW], eN[K1R_RPW], eNN[K1R_RPW];
    u32 dC[K1R_RPW][7];
    u64 pv_e[K1R_RPW];                                  // survivors of the previous tile: their new entries ...
    u32 pv_q[K1R_RPW];                                  // ... and their positions in the tile's survivor bitmap (~0: none)
    u32 abase = 0, par = 0;
    bool pv_valid = false;
    u32 tc = t0;
    // prologue: entries of the first two tiles, text of the first
    load_tile(tc, eC);
    load_tile(tc + G, eN);
#pragma unroll
    for (u32 it = 0; it < K1R_RPW; it++) { pv_e[it] = 0; pv_q[it] = 0xFFFFFFFFu; }
    gather(eC, dC);
    for (; tc * K1R_T < cnt; tc += G) {
#ifdef K1F_TRACE
        long long tprev_ = clock64();
#endif
        // 1. entries of the tile after next
        load_tile(tc + 2u * G, eNN);
        // 2. keys of this tile (gathered during the previous iteration) into LDS
#pragma unroll
        for (u32 it = 0; it < K1R_RPW; it++) {
            const u32 i = (it * 4u + w) * 64u + lane;
            if (own(eC[it], i)) {
                u32 p = ES(eC[it]) + dm;
                if (p >= n) p -= n;
                const u32 sel = be_sel(p);
                u32 x[6];
#pragma unroll
                for (int j = 0; j < 6; j++) x[j] = be32_at(dC[it][j + 1], dC[it][j], sel);
                kA[i] = ((u64)x[0] << 32) | x[1];
                kB[i] = ((u64)x[2] << 32) | x[3];
                kC[i] = ((u64)x[4] << 32) | x[5];
            }
        }
        if (tid < K1R_SW) sb[par][tid] = 0;
        if (tid == 0 && pv_valid) obase = abase;
        lds_barrier();
        K1R_STAMP(0);
        // 3. the previous tile's survivors to the next round's list: slot = survivors before it in that tile's bitmap (the
        //    other parity: counted by wave 0 after the previous step's ranking; this step's first barrier made it visible -
        //    a barrier of its own for that cost 15 % of the kernel)
        if (pv_valid && !final) {
#pragma unroll
            for (u32 it = 0; it < K1R_RPW; it++)
                if (pv_q[it] != 0xFFFFFFFFu) {
                    const u32 q = pv_q[it];
                    const u32 idx = obase + pre[par ^ 1u][q >> 5] + (u32)__popc(sb[par ^ 1u][q >> 5] & ((1u << (q & 31u)) - 1u));
                    if (idx < g.stride) Lout[idx] = pv_e[it];
                }
        }
        // 4. text of the next tile's owned entries: in flight while this tile is ranked
        u32 dN[K1R_RPW][7];
        gather(eN, dN);
        K1R_STAMP(1);
        // 5. rank inside the group: smaller keys, equal keys before, equal keys in all
        u32 qn[K1R_RPW];
#pragma unroll
        for (u32 it = 0; it < K1R_RPW; it++) {
            const u32 i = (it * 4u + w) * 64u + lane;
            qn[it] = 0xFFFFFFFFu;
            if (own(eC[it], i)) {
                // (round 5) against the OTHER members of the group, one at a time: a pair - most groups are - costs one comparison per rotation
                // where the loop over all cells in twos (ds_read2_b64) cost two, one of them with itself; the kernel is bound by its vector
                // instructions (PMC: 1 900 per wave and tile, half its clocks).  `less` is one borrow chain over the six key dwords.
                const u32 idx = EIDX(eC[it]), len = ELEN(eC[it]), gs = i - idx;
                const u64 m0 = kA[i], m1 = kB[i], m2 = kC[i];
                u32 less = 0, eqb = 0, eqt = 1;
                u32 o = gs + (idx == 0u ? 1u : 0u);
                u64 c0 = kA[o], c1 = kB[o], c2 = kC[o];
                for (u32 t = 0; t + 1u < len;) {
                    t++;
                    const u32 o2 = gs + t + (t >= idx ? 1u : 0u);          // (one past the last: a cell inside the arrays, read and not used)
                    const u64 n0 = kA[o2], n1 = kB[o2], n2 = kC[o2];
                    k1r_acc_lt192(less, c0, c1, c2, m0, m1, m2);
                    const bool eq = c0 == m0 && c1 == m1 && c2 == m2;
                    eqt += eq ? 1u : 0u;
                    eqb += (eq && t <= idx) ? 1u : 0u;                    // (this cell was member t - 1 of the others: in front of me while t - 1 < idx)
                    c0 = n0; c1 = n1; c2 = n2;
                }
                const u32 q = gs + less + eqb;
                const u32 s = ES(eC[it]), pos = EPOS(eC[it]) + q - i;     // positions inside a group are consecutive
                const bool surv = eqt > 1u;
                if (!surv || final) {
                    SA[pos] = s;
                    if (CARRY) U[pos] = (u8)K1C_BYTE(eC[it]);
                }
                if (surv) {
                    if (!final) {
                        qn[it] = q;
                        atomicOr(&sb[par][q >> 5], 1u << (q & 31u));
                        pv_e[it] = CARRY ? K1C_MAKE(eqt - 1u, eqb, K1C_BYTE(eC[it]), s, pos) : K1E_MAKE(eqt - 1u, eqb, s, pos);
                    } else if (eqb) {
                        // (not a head: its bit is cleared by the group's first member, below)
                    } else {
                        // the first member of a group that outlasted the rounds clears the head bits of the others - positions
                        // pos + 1 .. pos + eqt - 1, word by word: one or two atomics per GROUP (one per member was 37 M of them in the
                        // last round of `200k text tiled`, where every rotation survives: 12 of that round's 13 ms)
                        for (u32 p0 = pos + 1u, pe = pos + eqt; p0 < pe;) {
                            const u32 wend = (p0 | 31u) + 1u, e2 = wend < pe ? wend : pe;
                            const u32 m = (e2 - p0 == 32u ? 0xFFFFFFFFu : ((1u << (e2 - p0)) - 1u)) << (p0 & 31u);
                            atomicAnd(&HN[p0 >> 5], ~m);
                            p0 = e2;
                        }
                    }
                    if (final && !eqb && eqt <= K1_DEEP_LANE && n >= 64u && final_host) {
                        // a group that outlasted the rounds: 2..8 rotations go to the lane kernels' second pass (their walk wraps
                        // around the block at most once per step: not for blocks shorter than a step); not when the block stopped
                        // early - its ties are long repeats, which are the doubling rounds' business
                        const u32 cls = eqt == 2u ? 0u : 1u;
                        const u32 xr = (b & 7u) * K1_DEEP_SUB + ((pos >> 10) & (K1_DEEP_SUB - 1u)), rcap2 = B.listSCap / (8u * K1_DEEP_SUB);
                        const u32 idx = atomicAdd(&B.deepCnt[(2u + cls) * 8u * K1_DEEP_SUB + xr], 1u);
                        const u32 dd = depth + K1R_STEP < 0xFFFFu ? depth + K1R_STEP : 0xFFFFu;
                        if (idx < rcap2) B.listS[cls][(size_t)xr * rcap2 + idx] = ((u64)b << 52) | ((u64)pos << 26) | ((u64)dd << 4) | (u64)(eqt - 1u);
                    }
                }
            }
        }
        lds_barrier();
        K1R_STAMP(2);
        // 6. survivors before every bitmap word (one wave), slots for all of them (the result is used one iteration later)
        if (w == 0 && !final) {
            const u32 c = lane < K1R_SW ? (u32)__popc(sb[par][lane]) : 0u;
            const u32 inc = wave_incl_scan_u32(c);
            if (lane < K1R_SW) pre[par][lane] = inc - c;
            const u32 total = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            if (lane == 0) {
                // the result is first looked at a step later.  Unconditional: no select on it; and through a pointer the compiler takes for
                // divergent - for a uniform one it rewrites the atomic as a wave reduction whose v_readfirstlane waits for the result
                // (and for every load in flight) on the spot
                u32 z = 0;
                pin_vgpr(z);
                abase = atomicAdd(ocnt + z, total);
            }
        }
        K1R_STAMP(3);
#pragma unroll
        for (u32 it = 0; it < K1R_RPW; it++) pv_q[it] = qn[it];
        K1R_STAMP(4);
        // 7. rotate
#pragma unroll
        for (u32 it = 0; it < K1R_RPW; it++) {
            eC[it] = eN[it]; eN[it] = eNN[it];
#pragma unroll
            for (int j = 0; j < 7; j++) dC[it][j] = dN[it][j];
        }
        par ^= 1u;
        pv_valid = true;
    }
    // drain: the last tile's survivors
    if (pv_valid && !final) {
        if (tid == 0) obase = abase;
        lds_barrier();
#pragma unroll
        for (u32 it = 0; it < K1R_RPW; it++)
            if (pv_q[it] != 0xFFFFFFFFu) {
                const u32 q = pv_q[it];
                const u32 idx = obase + pre[par ^ 1u][q >> 5] + (u32)__popc(sb[par ^ 1u][q >> 5] & ((1u << (q & 31u)) - 1u));
                if (idx < g.stride) Lout[idx] = pv_e[it];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
size_t k1_front_tilehist_words(const BatchGeom& g) { return (size_t)k1f_ptiles(g) * K1F_NB; }

int k1_front_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream, u32 iters, u32 lists, u32 purerot_max, u32 carry) {
    const u32 ptiles = k1f_ptiles(g);
    const u32 nb8 = (g.nb + 7u) & ~7u;
    hipLaunchKernelGGL(k1f_sample, dim3(g.nb), dim3(1024), 0, stream, B, g);
    hipLaunchKernelGGL(k1f_hist, dim3(ptiles, g.nb), dim3(1024), 0, stream, B, g, ptiles);
    hipLaunchKernelGGL(k1f_scan, dim3(g.nb), dim3(1024), 0, stream, B, g, ptiles);
    hipLaunchKernelGGL(k1f_scatter, dim3(ptiles, nb8), dim3(1024), 0, stream, B, g, ptiles);
    {
        const u32 slot = k1_prof_begin(B.prof, K1P_BSORT, stream);
        hipLaunchKernelGGL(k1f_bsort, dim3(K1F_NB, nb8), dim3(K1F_BT), 0, stream, B, g, iters, lists, purerot_max, carry);
        k1_prof_end(B.prof, slot, stream, (u64)g.nb * max_n);
    }
    // the task levels: what k1f_bsort could not finish in LDS (slices beyond K1F_C, big groups), level after level; an empty
    // level costs its launch (~2 us: the workgroups read one counter and leave)
    for (u32 lv = 0; lv < K1F_LEVELS; lv++) {
        const u32 slot = k1_prof_begin(B.prof, K1P_TASK, stream);
#ifndef K1F_TASK_WGS
#define K1F_TASK_WGS 16u
#endif
        hipLaunchKernelGGL(k1f_task, dim3(((g.nb * K1F_TASK_WGS < 128u * K1F_TASK_WGS ? g.nb * K1F_TASK_WGS : 128u * K1F_TASK_WGS) + 7u) & ~7u), dim3(K1F_BT), 0, stream, B, g, lv, iters, lists, purerot_max,
                           lv + 1u == K1F_LEVELS ? 1u : 0u, carry);
        k1_prof_end(B.prof, slot, stream, 0);
    }
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

// the refinement rounds over the lists k1f_bsort filled: depth0 = what the listed groups share, up to max_depth
int k1_rounds_run(K1Buf B, const BatchGeom& g, hipStream_t stream, u32 depth0, u32 max_depth, u32 carry) {
    const u32 nb8 = (g.nb + 7u) & ~7u;
    const u32 full = (g.stride + K1R_T - 1u) / K1R_T;
    u32 rounds = max_depth > depth0 ? (max_depth - depth0 + K1R_STEP - 1u) / K1R_STEP : 1u;
    if (rounds > K1R_MAXR) rounds = K1R_MAXR;
    for (u32 r = 0; r < rounds; r++) {
        // the lists shrink from round to round (text: by a quarter to a third): later rounds launch fewer workgroups, each walks
        // its share of the tiles (an empty workgroup still costs its dispatch)
#ifndef K1R_WALK
#define K1R_WALK 4u         // (round 5, k1r_round ms on enwik with 16 / 8 / 4 / 2 / 1 tiles per workgroup: 2.07 / 1.87 / 1.79 / 1.76 / 1.78; on `text`, whose lists are short, 8: 0.27, 2: 0.32)
#endif
#ifndef K1R_SHR
#define K1R_SHR 2u
#endif
        u32 tiles = full / K1R_WALK;                  // a workgroup walks ~K1R_WALK tiles (software pipeline), later rounds launch fewer
        for (u32 q = 0; q < r; q++) tiles = K1R_SHR == 2u ? tiles / 2u : tiles * 5u / 8u;
        if (tiles < 16u) tiles = 16u;
        const u32 slot = k1_prof_begin(B.prof, K1P_RROUND, stream);
        if (carry) hipLaunchKernelGGL(k1r_round<true>, dim3(tiles, nb8), dim3(256), 0, stream, B, g, r, depth0 + K1R_STEP * r, r + 1u == rounds ? 1u : 0u);
        else hipLaunchKernelGGL(k1r_round<false>, dim3(tiles, nb8), dim3(256), 0, stream, B, g, r, depth0 + K1R_STEP * r, r + 1u == rounds ? 1u : 0u);
        k1_prof_end(B.prof, slot, stream, 0);
    }
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
