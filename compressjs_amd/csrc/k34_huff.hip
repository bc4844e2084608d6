// K3/K4: Huffman code lengths, canonical codes and the table-selection heuristic, for gfx950.
//
// Replaces, bit for bit:
//   StaticHuffman ctor            lib/Bzip2.js:551-579   (sort keys freq<<9|sym, allocator, unpermute)
//   allocateHuffmanCodeLengths    lib/HuffmanAllocator.js:199-222 (+ :52-188 helpers, incl. the
//                                 length-limit relocation branch, which ordinary files do reach)
//   computeCanonical              lib/Bzip2.js:581-600
//   cost / assignSelectors        lib/Bzip2.js:602-608, 671-684   (first minimum wins)
//   optimizeHuffmanGroups         lib/Bzip2.js:685-733   (first most-used table; STABLE sort of its
//                                 50-symbol groups by cost; upper half [len>>>1, len) moves to the
//                                 new table; recount; rebuild every table)
//   group-count rule              lib/Bzip2.js:826-830
//
// K34_SPLIT 1024-thread workgroups per bzip2 block, two kernels per optimiser iteration (see k34_assign below).
// Per-symbol code lengths of all (<= 6) tables are packed into one 64-bit LDS word (10 bits per
// table) so the cost of a 50-symbol group under every table is 50 LDS reads + 50 adds.
// The "stable sort + take upper half" is done without sorting: a cost histogram finds the
// median cost c*, and an ordered prefix count decides which groups with cost == c* stay.
#include "pipeline.h"
#include <atomic>

#define HB_PITCH 264
#ifdef K34_TRACE
#define K34_NOW() wall_clock64()
#else
#define K34_NOW() 0ll
#endif
static_assert(HB_PITCH == CJS_LEN_PITCH, "fr2's rows (pipeline.hip) have the pitch of the LDS rows");

// ---- allocator (serial, one lane) ------------------------------------------------------------
// x % len: the cells these loops look at hold extended parent pointers (tail or tail + len, :86-100), i.e. values below
// 2 len, for which the division (about forty instructions on one lane, three hundred times per table) is one subtraction;
// anything else still divides.
template <typename I>
__device__ __forceinline__ int ha_mod(I x, int len) {
    return x < (I)len ? (int)x : x < (I)(2 * len) ? (int)x - len : (int)(x % len);
}

template <typename I>
__device__ int ha_first(const I* a, int len, int i, int nodes_to_move) {             // :52-73
    const int limit = i;
    int k = len - 2;
    while (i >= nodes_to_move && ha_mod(a[i], len) > limit) { k = i; i -= (limit - i + 1); }
    if (i < nodes_to_move - 1) i = nodes_to_move - 1;
    while (k > i + 1) {
        const int t = (i + k) >> 1;
        if (ha_mod(a[t], len) > limit) k = t; else i = t;
    }
    return k;
}

template <typename I>
__device__ void ha_parents(I* a, int len) {
    // setExtendedParentPointers :79-105
    // The fronts of the two queues (vh = a[head]: the oldest unconsumed internal node, vt = a[top]: the next leaf) are
    // carried in registers: a step costs the refill of whichever front it consumed instead of four dependent reads.
    // At the first selection of a step head < tail always holds (a step consumes at most the nodes made before it).
    a[0] += a[1];
    {
        const I none = (I)(~(unsigned long long)0 >> (65 - 8 * sizeof(I)));          // larger than any weight
        int head = 0, top = 2;
        I vh = a[0];
        I vt = top < len ? a[top] : none;
        for (int tail = 1; tail < len - 1; tail++) {
            I t;
            if (top >= len || vh < vt) { t = vh; a[head++] = tail; vh = head < tail ? a[head] : none; }
            else { t = vt; top++; vt = top < len ? a[top] : none; }
            if (top >= len || (head < tail && vh < vt)) { t += vh; a[head++] = tail + len; vh = head < tail ? a[head] : none; }
            else { t += vt; top++; vt = top < len ? a[top] : none; }
            a[tail] = t;
            if (head == tail) vh = t;                     // the node just made is the only one waiting
        }
    }
}

// The rest of the allocator on one lane: findNodesToRelocate, then allocateNodeLengths or allocateNodeLengthsWithRelocation
template <typename I>
__device__ void ha_depths_serial(I* a, int len, int maxlen) {
    // findNodesToRelocate :114-124
    int reloc = len - 2;
    for (int d = 1; d < maxlen - 1 && reloc > 1; d++) reloc = ha_first(a, len, reloc - 1, 0);
    if (ha_mod(a[0], len) >= reloc) {
        // allocateNodeLengths :131-148
        int first = len - 2, next = len - 1;
        for (int depth = 1, avail = 2; avail > 0; depth++) {
            const int last = first;
            first = ha_first(a, len, last - 1, 0);
            for (int i = avail - (last - first); i > 0; i--) a[next--] = depth;
            avail = (last - first) << 1;
        }
    } else {
        // allocateNodeLengthsWithRelocation :157-188
        const int nodes_to_move = reloc;
        int fl = 0;
        for (u32 v = (u32)(reloc - 1); v; v >>= 1) fl++;                       // Util.fls
        const int insert_depth = maxlen - fl;
        int first = len - 2, next = len - 1;
        int depth = (insert_depth == 1) ? 2 : 1;
        int left = (insert_depth == 1) ? nodes_to_move - 2 : nodes_to_move;
        for (int avail = depth << 1; avail > 0; depth++) {
            const int last = first;
            first = (first <= nodes_to_move) ? first : ha_first(a, len, last - 1, nodes_to_move);
            int offset = 0;
            if (depth >= insert_depth) {
                const int capv = 1 << (depth - insert_depth);
                offset = left < capv ? left : capv;
            } else if (depth == insert_depth - 1) {
                offset = 1;
                if (a[first] == last) first++;
            }
            for (int i = avail - (last - first + offset); i > 0; i--) a[next--] = depth;
            left -= offset;
            avail = (last - first + offset) << 1;
        }
    }
}

template <typename I>
__device__ void ha_allocate(I* a, int len, int maxlen) {
    ha_parents(a, len);
    ha_depths_serial(a, len, maxlen);
}

// The same depths by the whole wave (round 3: the serial loops above - a binary search per level, one store per leaf, every
// step a dependent LDS round trip on one lane - were 27 of the 49 us a table took, five times per block on the critical path).
// After ha_parents, cells 0 .. len-3 hold parent pointers and the nodes of a level are contiguous, deeper levels first.
//   1. depth of every internal node by pointer jumping (anc / dist; log2(depth) rounds),
//   2. nodes per depth (cnt), first node of every level = nodes deeper than it (an inclusive scan),
//   3. the reference's relocation test, literally: reloc = first node of level maxlen - 2 (or of the level where that
//      index falls to <= 1), no relocation iff parent(node 0) >= reloc (:114-124, :207-212),
//   4. leaves per depth = 2 * nodes(depth - 1) - nodes(depth) (allocateNodeLengths :131-148: `avail` slots, minus the
//      internal nodes, are leaves), handed out from the END of the sorted array, smallest depth first.
//   5. a tree deeper than maxlen: the relocation branch (:157-188), level by level by the whole wave (see below).
// Returns false, with `a` untouched, only for maxlen beyond HA_DCAP - 1: the caller then runs the serial path.
// anc, dist: len ints each; cnt: 128 ints; all lanes of the wave call this.
#define HA_DCAP 63
template <typename I>
__device__ bool ha_depths_wave(I* a, int len, int maxlen, int* anc, int* dist, int* cnt) {
    const int lane = (int)(threadIdx.x & 63u);
    const int root = len - 2;
    for (int i = lane; i <= root; i += 64) { anc[i] = i < root ? ha_mod(a[i], len) : root; dist[i] = i < root ? 1 : 0; }
    cnt[lane] = 0;
    const int par0 = ha_mod(a[0], len);                    // read before any lane can be past the verdict (lane 0 then rewrites a)
    __builtin_amdgcn_wave_barrier();
    for (;;) {
        bool more = false;
        for (int i = lane; i < root; i += 64) {
            const int p = anc[i];
            if (p != root) {                              // (anc[p], dist[p]) is a consistent pair whichever lane wrote it last
                const int pa = anc[p], pd = dist[p];
                anc[i] = pa; dist[i] += pd;
                more = more || pa != root;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (!__any(more ? 1 : 0)) break;
    }
    for (int i = lane; i <= root; i += 64) atomicAdd(&cnt[dist[i] < HA_DCAP ? dist[i] : HA_DCAP], 1);
    __builtin_amdgcn_wave_barrier();
    const int c = cnt[lane];
    const int start = (len - 1) - (int)wave_incl_scan_u32((u32)c);          // first node of level `lane`
    int reloc = root;
    if (root > 1 && maxlen > 2) {
        const u64 low = __ballot(lane >= 1 && start <= 1);
        const int dstar = low ? (int)__ffsll((long long)low) - 1 : 64;
        reloc = __shfl(start, dstar < maxlen - 2 ? dstar : maxlen - 2);
    }
    int deep = dist[0];                                    // the first node made is the deepest; leaves reach deep + 1 <= maxlen
    if (par0 < reloc) {
        // allocateNodeLengthsWithRelocation :157-188 with the wave walking the levels together: the one search per level
        // (ha_first(a, len, last - 1, nodes_to_move) = the nodes whose parent lies below `last`, at least nodes_to_move)
        // is a count over all nodes, the leaves of a level are noted and handed out at the end as in the other branch.
        // Skewed alphabets (HTML, binaries: E8S-A) take this branch for most tables.
        if (maxlen > HA_DCAP - 1) return false;
        const int ntm = reloc;
        int fl = 0;
        for (u32 v = (u32)(reloc - 1); v; v >>= 1) fl++;                       // Util.fls
        const int insert_depth = maxlen - fl;
        int first = root;
        int depth = insert_depth == 1 ? 2 : 1;
        int left = insert_depth == 1 ? ntm - 2 : ntm;
        cnt[64 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int avail = depth << 1; avail > 0; depth++) {
            const int last = first;
            if (first > ntm) {
                int below = 0;
                for (int k0 = 0; k0 < root; k0 += 64) {
                    const int k = k0 + lane;
                    below += (int)__popcll(__ballot(k < root && ha_mod(a[k], len) < last));
                }
                first = below > ntm ? below : ntm;
            }
            int offset = 0;
            if (depth >= insert_depth) {
                const int capv = 1 << (depth - insert_depth);
                offset = left < capv ? left : capv;
            } else if (depth == insert_depth - 1) {
                offset = 1;
                if (a[first] == (I)last) first++;
            }
            const int fill = avail - (last - first + offset);
            if (lane == 0 && fill > 0) cnt[64 + depth] = fill;
            left -= offset;
            avail = (last - first + offset) << 1;
        }
        __builtin_amdgcn_wave_barrier();
        const int f = cnt[64 + lane];
        const int cum = (int)wave_incl_scan_u32((u32)f);
        __builtin_amdgcn_wave_barrier();
        cnt[64 + lane] = cum;
        deep = depth - 2;                                  // the loop below looks at depths 1 .. deep; the last one used is depth - 1
    } else {
        const int above = __shfl_up(c, 1);
        const int nl = lane >= 1 ? 2 * above - c : 0;
        cnt[64 + lane] = (int)wave_incl_scan_u32((u32)nl); // leaves of depth <= lane
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < len; j += 64) {
        const int r = len - j;                             // j-th smallest weight = r-th leaf from the end
        int dep = 1;
        for (int D = 1; D <= deep; D++) dep += cnt[64 + D] < r ? 1 : 0;
        a[j] = (I)dep;
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

// One wave builds one table: freq[0..S) -> lens[0..S).  keys/arr and anc/dist/cnt are per-wave LDS scratch.
__device__ void huff_build_wave(const u32* freq, int S, u8* lens, int* keys, int* arr, int* anc, int* dist, int* cnt, u32* tr = nullptr) {
    const int lane = (int)(threadIdx.x & 63u);
    long long t0_ = tr ? K34_NOW() : 0;
    for (int i = lane; i < S; i += 64) keys[i] = (int)((freq[i] << 9) | (u32)i);      // :566-568
    __builtin_amdgcn_wave_barrier();
    // ascending sort by rank counting (keys are distinct)
    {
        constexpr int M = (CJS_MAX_SYMS + 63) / 64;       // keys per lane, at most
        int k[M], r[M];
#pragma unroll
        for (int m = 0; m < M; m++) { const int i = lane + 64 * m; k[m] = i < S ? keys[i] : 0; r[m] = 0; }
        for (int j = 0; j < S; j++) {
            const int kj = keys[j];                       // one broadcast read serves all of the lane's keys
#pragma unroll
            for (int m = 0; m < M; m++) r[m] += kj < k[m] ? 1 : 0;
        }
#pragma unroll
        for (int m = 0; m < M; m++) if (lane + 64 * m < S) arr[r[m]] = k[m];
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < S; i += 64) keys[i] = arr[i];                              // sorted keys
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < S; i += 64) arr[i] = (int)((u32)keys[i] >> 9);            // :570
    __builtin_amdgcn_wave_barrier();
    if (tr && lane == 0) { const long long t1_ = K34_NOW(); atomicAdd(tr, (u32)(t1_ - t0_)); t0_ = t1_; }
    if (lane == 0) ha_parents(arr, S);                                                // :572
    __builtin_amdgcn_wave_barrier();
    if (tr && lane == 0) { const long long t1_ = K34_NOW(); atomicAdd(tr + 1, (u32)(t1_ - t0_)); t0_ = t1_; }
    if (!ha_depths_wave(arr, S, CJS_MAX_BITS, anc, dist, cnt)) {
        if (lane == 0) ha_depths_serial(arr, S, CJS_MAX_BITS);
    }
    if (tr && lane == 0) atomicAdd(tr + 2, (u32)(K34_NOW() - t0_));
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < S; i += 64) lens[keys[i] & 0x1FF] = (u8)arr[i];            // :575-578
    __builtin_amdgcn_wave_barrier();
}

// LDS plan: the 100 KB staging area for the symbol walks (dynamic LDS) is time-shared with the
// table-build scratch (keys, arr) and the cost histogram, which are never live at the same time.
// 64 groups per chunk: every lane of a wave walks a group (with 32, half of them only helped staging).
#define K34_STAGE_WORDS 1600         // 64 groups x 50 symbols x 2 B per wave
#define K34_GROUPS_PER_CHUNK 64
#define K34_PRE ((K34_STAGE_WORDS + 63) / 64)

// Walk the 50-symbol groups of A in chunks of 64 groups: the wave takes chunks c0, c0 + cstep, ..., stages each (3200
// symbols) with coalesced 4-byte loads, then every lane reads its own group back (stride 25 words: conflict free per
// half-wave).  `body(gi, sym)` is called for every symbol of group gi by the lane that owns it.
template <class Body, class Done>
__device__ __forceinline__ void walk_groups(const u16* A, u32 pos, u32 nSel, u32* stage_w, u32 c0, u32 cstep, Body body, Done done) {
    const u32 lane = threadIdx.x & 63u;
    const u32* Aw = (const u32*)A;
    const u32 nchunks = (nSel + K34_GROUPS_PER_CHUNK - 1) / K34_GROUPS_PER_CHUNK;
    // software pipeline: the words of the wave's next chunk are in flight while this one is consumed from LDS
    u32 pre[K34_PRE];
    auto fetch = [&](u32 c) {
        const u32 sym0 = c * K34_GROUPS_PER_CHUNK * CJS_GROUP;
        const u32 nsym = pos - sym0 < 2u * K34_STAGE_WORDS ? pos - sym0 : 2u * K34_STAGE_WORDS;
        const u32 nwords = (nsym + 1u) >> 1;
#pragma unroll
        for (int k = 0; k < K34_PRE; k++) {
            const u32 j = lane + 64u * (u32)k;
            pre[k] = j < nwords ? Aw[(sym0 >> 1) + j] : 0u;
        }
    };
    if (c0 < nchunks) fetch(c0);
    for (u32 c = c0; c < nchunks; c += cstep) {
        const u32 g0 = c * K34_GROUPS_PER_CHUNK;
#pragma unroll
        for (int k = 0; k < K34_PRE; k++) {
            const u32 j = lane + 64u * (u32)k;
            if (j < K34_STAGE_WORDS) stage_w[j] = pre[k];
        }
        __builtin_amdgcn_wave_barrier();
        if (c + cstep < nchunks) fetch(c + cstep);
        const u32 gi = g0 + lane;
        if (lane < K34_GROUPS_PER_CHUNK && gi < nSel) {
            const u32 cnt = pos - gi * CJS_GROUP < CJS_GROUP ? pos - gi * CJS_GROUP : CJS_GROUP;
            const u32* gw = stage_w + lane * 25u;
            if (cnt == CJS_GROUP) {
                // a full group (all but the last of a block): the 25 staged words first, then 50 independent bodies - as a
                // loop this was one dependent LDS round trip per pair of symbols (6 us per chunk)
                u32 wd[CJS_GROUP / 2];
#pragma unroll
                for (int k = 0; k < CJS_GROUP / 2; k++) wd[k] = gw[k];
#pragma unroll
                for (int k = 0; k < CJS_GROUP / 2; k++) { body(gi, wd[k] & 0xFFFFu); body(gi, wd[k] >> 16); }
            } else {
                for (u32 k = 0; k < cnt; k += 2) {
                    const u32 wd = gw[k >> 1];
                    body(gi, wd & 0xFFFFu);
                    if (k + 1 < cnt) body(gi, wd >> 16);
                }
            }
            done(gi);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The optimiser loop of lib/Bzip2.js:685-733 as K34_SPLIT workgroups per block and two kernels per iteration (round 3; rounds
// 1-2 ran the whole loop in ONE workgroup per block: 56 workgroups on 256 CUs for 1.1 ms, all of it latency):
//   k34_assign(it)  G = 2 + it tables: every workgroup of the block builds the tables from the block's frequency rows (the same
//                   serial allocator run K34_SPLIT times: latency, not work), then costs and assigns ITS chunks of groups
//                   (chunk c belongs to workgroup (c / 16) % K34_SPLIT).  G == target: workgroup 0 writes the canonical codes.
//   k34_split(it)   G < target: every workgroup finds the most-used table and the median cost over ALL groups (18 000 at most),
//                   moves the upper half to table G (in its LDS copy of the selectors), recounts its chunks into
//                   LDS rows and adds them to the block's next frequency rows (fr2, two sets used alternately).
// The kernel boundary is the barrier between the walks; nothing waits inside a kernel.
#ifndef K34_SPLIT
#define K34_SPLIT 2              // (round 3: 4, equal to 2 then; round 4, ms per 10^8 bytes with 1 / 2 / 3 / 4 / 8: enwik 8.77 / 8.53 / 8.60 / 8.58 / 8.97, E8S-A 12.68 / 12.37 / 12.39 / 12.51 / 13.18)
#endif
#define K34_MAX_SEL 18432        // selectors of a block, at most ((900000 + 19) / 50 rounded up; k34_run checks selPitch)
// -DK34_TRACE builds: 100 MHz stamps between the phases of block 0 / workgroup 0, summed over the iterations into
// k1.stats[K1_STAT_RTRACE ..]; k34_run prints them (ticks of 10 ns).  Not in product builds.
#ifdef K34_TRACE
#define K34_T0 long long tprev_ = wall_clock64();
#define K34_STAMP(slot) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) { const long long now_ = wall_clock64(); atomicAdd(&P.k1.stats[K1_STAT_RTRACE + (slot)], (u32)(now_ - tprev_)); tprev_ = now_; } } while (0)
#else
#define K34_T0
#define K34_STAMP(slot) do { } while (0)
#endif

struct K34Blk {
    u32 pos, nSel;
    int S, target;
};
__device__ __forceinline__ K34Blk k34_blk(const Pipe& P, u32 b) {
    K34Blk k;
    k.pos = P.pos[b];
    k.S = (int)P.alpha[b] + 2;
    k.nSel = (k.pos + CJS_GROUP - 1) / CJS_GROUP;                                     // :841
    k.target = k.pos >= 2400 ? 6 : k.pos >= 1200 ? 5 : k.pos >= 600 ? 4 : k.pos >= 200 ? 3 : 2;   // :826-830
    return k;
}

__global__ __launch_bounds__(1024) void k34_assign(Pipe P, u32 it) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.x / K34_SPLIT, k = blockIdx.x % K34_SPLIT;
    if (P.nlen[b] == 0) return;
    const K34Blk q = k34_blk(P, b);
    const int G = 2 + (int)it, S = q.S;
    if (G > q.target) return;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u16* A = P.A + (size_t)b * g.stride;
    u8* sel = P.sel + (size_t)b * P.selPitch;
    u16* cost = P.selCost + (size_t)b * P.selPitch;

    HIP_DYNAMIC_SHARED(u32, pool)                                                       // [16 * K34_STAGE_WORDS]
    __shared__ u8 lens[CJS_MAX_GROUPS][HB_PITCH];
    __shared__ u32 fr[CJS_MAX_GROUPS][HB_PITCH];
    __shared__ u64 lens64[HB_PITCH];
    u32* stage_w = pool + w * K34_STAGE_WORDS;
    int (*keys)[HB_PITCH] = (int (*)[HB_PITCH])pool;                                   // [6][264]   (before the walk)
    int (*arr)[HB_PITCH] = (int (*)[HB_PITCH])(pool + CJS_MAX_GROUPS * HB_PITCH);      // [6][264]
    int (*anc)[HB_PITCH] = (int (*)[HB_PITCH])(pool + 2 * CJS_MAX_GROUPS * HB_PITCH);  // [6][264]
    int (*dst)[HB_PITCH] = (int (*)[HB_PITCH])(pool + 3 * CJS_MAX_GROUPS * HB_PITCH);  // [6][264]
    int (*dcn)[128] = (int (*)[128])(pool + 4 * CJS_MAX_GROUPS * HB_PITCH);            // [6][128]

    u32* rows = P.fr2 + ((size_t)b * 2 + (it & 1u)) * CJS_MAX_GROUPS * HB_PITCH;       // written by k34_split(it - 1)
    u32* next = P.fr2 + ((size_t)b * 2 + ((it + 1u) & 1u)) * CJS_MAX_GROUPS * HB_PITCH;
    K34_T0
    if (it == 0) {
        const u32* gfreq = P.freq + (size_t)b * K2_FREQ_PITCH;
        for (u32 i = tid; i < (u32)S; i += 1024) { fr[0][i] = gfreq[i]; fr[1][i] = 1; }   // :835-837
    } else {
        for (u32 i = tid; i < (u32)G * HB_PITCH; i += 1024) (&fr[0][0])[i] = rows[i];
    }
    if (k == 0 && G < q.target)
        for (u32 i = tid; i < (u32)(CJS_MAX_GROUPS * HB_PITCH); i += 1024) next[i] = 0;
    __syncthreads();
    K34_STAMP(0);
#ifdef K34_TRACE
    if (w < (u32)G) huff_build_wave(fr[w], S, lens[w], keys[w], arr[w], anc[w], dst[w], dcn[w], blockIdx.x == 0 && w == 0 ? P.k1.stats + K1_STAT_RTRACE + 7 : nullptr);
#else
    if (w < (u32)G) huff_build_wave(fr[w], S, lens[w], keys[w], arr[w], anc[w], dst[w], dcn[w]);
#endif
    __syncthreads();
    K34_STAMP(1);
    for (u32 i = tid; i < (u32)S; i += 1024) {
        u64 v = 0;
        for (int t = 0; t < G; t++) v |= (u64)lens[t][i] << (10 * t);
        lens64[i] = v;
    }
    __syncthreads();
    {   // assignSelectors :671-684
        u64 acc = 0;
        walk_groups(A, q.pos, q.nSel, stage_w, k * 16u + w, 16u * K34_SPLIT,
            [&](u32, u32 sym) { acc += lens64[sym]; },
            [&](u32 gi) {
                u32 best = 0, bc = (u32)(acc & 1023u);
                for (int t = 1; t < G; t++) {
                    const u32 c = (u32)((acc >> (10 * t)) & 1023u);
                    if (c < bc) { best = (u32)t; bc = c; }
                }
                sel[gi] = (u8)best;
                cost[gi] = (u16)bc;
                acc = 0;
            });
    }
    K34_STAMP(2);
    if (G < q.target || k != 0) return;
    // canonical codes :581-600, one wave per table: lane l owns code length l - it counts the symbols of that length, the
    // first codes of the lengths in use chain through the wave (c <<= l - prev; next[l] = c; c += count[l]), then the lane
    // hands out its codes in symbol order.
    if (w < (u32)G) {
        const int l = (int)lane;
        const bool mine = l >= 1 && l <= CJS_MAX_BITS;
        u32 count = 0;
        if (mine) for (int i = 0; i < S; i++) count += lens[w][i] == l ? 1u : 0u;
        u32 c = 0, first = 0; int prev = 0;
        for (int ll = 1; ll <= CJS_MAX_BITS; ll++) {
            const u32 cl = (u32)__shfl((int)count, ll);
            if (!cl) continue;
            c <<= (ll - prev);
            if (ll == l) first = c;
            c += cl;
            prev = ll;
        }
        u32* codes = P.codes + ((size_t)b * CJS_MAX_GROUPS + w) * CJS_LEN_PITCH;
        u8* lout = P.lens + ((size_t)b * CJS_MAX_GROUPS + w) * CJS_LEN_PITCH;
        if (mine && count) for (int i = 0; i < S; i++) if (lens[w][i] == l) codes[i] = first++;
        for (int i = l; i < S; i += 64) lout[i] = lens[w][i];
    }
    if (tid == 0) { P.ngroups[b] = (u32)G; P.nsel[b] = q.nSel; }
    K34_STAMP(3);
}

__global__ __launch_bounds__(1024) void k34_split(Pipe P, u32 it) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.x / K34_SPLIT, k = blockIdx.x % K34_SPLIT;
    if (P.nlen[b] == 0) return;
    const K34Blk q = k34_blk(P, b);
    const int G = 2 + (int)it;
    if (G >= q.target) return;
    const u32 tid = threadIdx.x, w = tid >> 6;
    const u32 nSel = q.nSel;
    const u16* A = P.A + (size_t)b * g.stride;
    const u8* sel = P.sel + (size_t)b * P.selPitch;
    const u16* cost = P.selCost + (size_t)b * P.selPitch;

    HIP_DYNAMIC_SHARED(u32, pool)
    __shared__ u32 fr[CJS_MAX_GROUPS][HB_PITCH];
    __shared__ u32 chist[1024];
    __shared__ u32 cnt[8];
    __shared__ u32 scan_sh[20];
    __shared__ u32 s_which, s_cstar, s_keep;
    __shared__ __attribute__((aligned(16))) u8 selL[K34_MAX_SEL];
    u32* stage_w = pool + w * K34_STAGE_WORDS;
    K34_T0
    // selectors and costs of ALL groups of the block into LDS (costs in the pool, which is free until the walk): the passes
    // below touch them four times, the last one in per-thread runs; the selectors after the move stay in LDS for the recount
    u16* costL = (u16*)pool;                               // [selPitch]
    for (u32 i = tid; i < (nSel + 3u) >> 2; i += 1024) ((u32*)selL)[i] = ((const u32*)sel)[i];
    for (u32 i = tid; i < (nSel + 1u) >> 1; i += 1024) ((u32*)costL)[i] = ((const u32*)cost)[i];
    if (tid < 8) cnt[tid] = 0;
    chist[tid] = 0;
    for (u32 i = tid; i < (u32)(CJS_MAX_GROUPS * HB_PITCH); i += 1024) (&fr[0][0])[i] = 0;
    __syncthreads();
    for (u32 g0 = 0; g0 < nSel; g0 += 1024) {              // groups per table: one ballot per table and wave, one atomic per wave
        const u32 gi = g0 + tid;
        const u32 v = gi < nSel ? selL[gi] : 0xFFu;
        for (int t = 0; t < G; t++) {
            const u64 m = __ballot(v == (u32)t);
            if ((tid & 63u) == 0 && m) atomicAdd(&cnt[t], (u32)__popcll(m));
        }
    }
    __syncthreads();
    if (tid == 0) {                                   // first most-used table :699
        u32 wh = 0;
        for (int t = 1; t < G; t++) if (cnt[t] > cnt[wh]) wh = (u32)t;
        s_which = wh;
    }
    __syncthreads();
    const u32 which = s_which;
    for (u32 gi = tid; gi < nSel; gi += 1024)
        if (selL[gi] == which) atomicAdd(&chist[costL[gi]], 1u);
    __syncthreads();
    {
        // smallest c with (#groups of cost < c) + chist[c] > half: elements of stable rank >= half move (:705-716: a STABLE
        // sort by cost, the upper half [len >>> 1, len) goes to the new table)
        const u32 half = cnt[which] >> 1;             // :712
        u32 tot;
        const u32 mine = chist[tid];
        const u32 lower = block_excl_scan_1024(mine, scan_sh, &tot);
        if (lower <= half && half < lower + mine) { s_cstar = tid; s_keep = half - lower; }
    }
    __syncthreads();
    {
        const u32 cstar = s_cstar, keep = s_keep;
        const u32 chunk = (nSel + 1023u) / 1024u;
        const u32 lo = tid * chunk < nSel ? tid * chunk : nSel;
        const u32 hi = lo + chunk < nSel ? lo + chunk : nSel;
        u32 eq = 0;
        for (u32 gi = lo; gi < hi; gi++) eq += (selL[gi] == which && costL[gi] == cstar) ? 1u : 0u;
        u32 total;
        u32 rank = block_excl_scan_1024(eq, scan_sh, &total);
        for (u32 gi = lo; gi < hi; gi++) {
            u32 v = selL[gi];
            if (v == which) {
                const u32 c = costL[gi];
                if (c > cstar) v = (u32)G;
                else if (c == cstar) { if (rank >= keep) v = (u32)G; rank++; }
            }
            selL[gi] = (u8)v;
        }
    }
    __syncthreads();
    K34_STAMP(4);
    // recount :717-727.  A lane owns a group, i.e. one table row; the symbols below 8 (RUNA, RUNB and the first move-to-front
    // ranks: most of what follows a BWT) are counted in eight byte lanes of a register and added once per group - 64 lanes
    // adding 1 to the same three LDS words 50 times over was what this walk waited for.
    {
        u64 hot = 0;
        walk_groups(A, q.pos, nSel, stage_w, k * 16u + w, 16u * K34_SPLIT,
            [&](u32 gi, u32 sym) {
                if (sym < 8u) hot += 1ull << (8u * sym);
                else atomicAdd(&fr[selL[gi]][sym], 1u);
            },
            [&](u32 gi) {
                u32* row = fr[selL[gi]];
#pragma unroll
                for (u32 sy = 0; sy < 8u; sy++) {
                    const u32 c = (u32)(hot >> (8u * sy)) & 255u;
                    if (c) atomicAdd(&row[sy], c);
                }
                hot = 0;
            });
    }
    __syncthreads();
    K34_STAMP(5);
    u32* next = P.fr2 + ((size_t)b * 2 + ((it + 1u) & 1u)) * CJS_MAX_GROUPS * HB_PITCH;   // zeroed by k34_assign(it)
    for (u32 i = tid; i < (u32)(CJS_MAX_GROUPS * HB_PITCH); i += 1024) {
        const u32 v = (&fr[0][0])[i];
        if (v) atomicAdd(&next[i], v);
    }
    K34_STAMP(6);
}

int k34_run(Pipe P, hipStream_t stream) {
    const size_t dyn = (size_t)16 * K34_STAGE_WORDS * 4;
    // per device (cjs_bz2_compress_multi runs contexts on several devices from one process): set once for each device id that
    // comes by, and retried on the next call if it failed
    static std::atomic<unsigned> lds_set[64];
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !lds_set[dev].load(std::memory_order_acquire)) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)k34_assign, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * K34_STAGE_WORDS * 4)));
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)k34_split, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * K34_STAGE_WORDS * 4)));
        if (dev >= 0 && dev < 64) lds_set[dev].store(1u, std::memory_order_release);
    }
    if (P.selPitch > K34_MAX_SEL) return CJS_E_ARG;
    for (u32 it = 0; it + 2u <= CJS_MAX_GROUPS; it++) {
        hipLaunchKernelGGL(k34_assign, dim3(P.g.nb * K34_SPLIT), dim3(1024), dyn, stream, P, it);
        if (it + 3u <= CJS_MAX_GROUPS) hipLaunchKernelGGL(k34_split, dim3(P.g.nb * K34_SPLIT), dim3(1024), dyn, stream, P, it);
    }
    HIP_CHECK_RET(hipGetLastError());
#ifdef K34_TRACE
    {
        u32 h[10];
        HIP_CHECK_RET(hipMemcpyAsync(h, P.k1.stats + K1_STAT_RTRACE, sizeof h, hipMemcpyDeviceToHost, stream));
        HIP_CHECK_RET(hipStreamSynchronize(stream));
        fprintf(stderr, "k34 trace (us, block 0 / workgroup 0, summed over iterations): assign load %.1f build %.1f lens64+walk %.1f codes %.1f | split count/median/move %.1f recount %.1f flush %.1f | table 0: sort %.1f parents %.1f depths %.1f\n",
                h[0] / 100.0, h[1] / 100.0, h[2] / 100.0, h[3] / 100.0, h[4] / 100.0, h[5] / 100.0, h[6] / 100.0, h[7] / 100.0, h[8] / 100.0, h[9] / 100.0);
    }
#endif
    return CJS_OK;
}

// ---- standalone allocator entry (HuffmanAllocator.allocateHuffmanCodeLengths, :199-222) ------
// One wave per array (the same two steps as the tables of a block: parents on one lane, depths by the wave; arrays beyond
// K3_WAVE_MAX cells stay on one lane); 64-bit cells because callers of the JS function may pass weights whose sum exceeds
// 2^31 (the block pipeline above never does).
#define K3_WAVE_MAX 2048
__global__ __launch_bounds__(64) void k3_alloc_lengths(long long* arr, const u32* off, u32 count, int maxlen) {
    const u32 t = blockIdx.x;
    if (t >= count) return;
    __shared__ int anc[K3_WAVE_MAX], dist[K3_WAVE_MAX], cnt[128];
    long long* a = arr + off[t];
    const int len = (int)(off[t + 1] - off[t]);
    const bool one = threadIdx.x == 0;
    if (len == 2) { if (one) { a[0] = 1; a[1] = 1; } }                       // :201-205
    else if (len == 1) { if (one) a[0] = 1; }
    else if (len > K3_WAVE_MAX || maxlen > HA_DCAP - 1) { if (one) ha_allocate<long long>(a, len, maxlen); }
    else if (len > 2) {
        if (one) ha_parents<long long>(a, len);
        __syncthreads();
        if (!ha_depths_wave<long long>(a, len, maxlen, anc, dist, cnt)) {
            if (one) ha_depths_serial<long long>(a, len, maxlen);
        }
    }
}
int k3_alloc_lengths_run(long long* d_arr, const u32* d_off, u32 count, int maxlen, hipStream_t stream) {
    hipLaunchKernelGGL(k3_alloc_lengths, dim3(count), dim3(64), 0, stream, d_arr, d_off, count, maxlen);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
