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
// One 1024-thread workgroup per bzip2 block; the whole optimiser loop runs inside one launch.
// Per-symbol code lengths of all (<= 6) tables are packed into one 64-bit LDS word (10 bits per
// table) so the cost of a 50-symbol group under every table is 50 LDS reads + 50 adds.
// The "stable sort + take upper half" is done without sorting: a cost histogram finds the
// median cost c*, and an ordered prefix count decides which groups with cost == c* stay.
#include "pipeline.h"

#define HB_PITCH 264

// ---- allocator (serial, one lane) ------------------------------------------------------------
template <typename I>
__device__ int ha_first(const I* a, int len, int i, int nodes_to_move) {             // :52-73
    const int limit = i;
    int k = len - 2;
    while (i >= nodes_to_move && (int)(a[i] % len) > limit) { k = i; i -= (limit - i + 1); }
    if (i < nodes_to_move - 1) i = nodes_to_move - 1;
    while (k > i + 1) {
        const int t = (i + k) >> 1;
        if ((int)(a[t] % len) > limit) k = t; else i = t;
    }
    return k;
}

template <typename I>
__device__ void ha_allocate(I* a, int len, int maxlen) {
    // setExtendedParentPointers :79-105
    a[0] += a[1];
    {
        int head = 0, top = 2;
        for (int tail = 1; tail < len - 1; tail++) {
            I t;
            if (top >= len || a[head] < a[top]) { t = a[head]; a[head++] = tail; }
            else t = a[top++];
            if (top >= len || (head < tail && a[head] < a[top])) { t += a[head]; a[head++] = tail + len; }
            else t += a[top++];
            a[tail] = t;
        }
    }
    // findNodesToRelocate :114-124
    int reloc = len - 2;
    for (int d = 1; d < maxlen - 1 && reloc > 1; d++) reloc = ha_first(a, len, reloc - 1, 0);
    if ((int)(a[0] % len) >= reloc) {
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

// One wave builds one table: freq[0..S) -> lens[0..S).  keys/arr are per-wave LDS scratch.
__device__ void huff_build_wave(const u32* freq, int S, u8* lens, int* keys, int* arr) {
    const int lane = (int)(threadIdx.x & 63u);
    for (int i = lane; i < S; i += 64) keys[i] = (int)((freq[i] << 9) | (u32)i);      // :566-568
    __builtin_amdgcn_wave_barrier();
    // ascending sort by rank counting (keys are distinct)
    for (int i = lane; i < S; i += 64) {
        const int k = keys[i];
        int r = 0;
        for (int j = 0; j < S; j++) r += keys[j] < k ? 1 : 0;
        arr[r] = k;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < S; i += 64) keys[i] = arr[i];                              // sorted keys
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < S; i += 64) arr[i] = (int)((u32)keys[i] >> 9);            // :570
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) ha_allocate(arr, S, CJS_MAX_BITS);                                 // :572
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

// Walk every 50-symbol group of A: each wave stages 64 groups (3200 symbols) with coalesced
// 4-byte loads, then every lane reads its own group back (stride 25 words: conflict free per half-wave).
// `body(gi, sym)` is called for every symbol of group gi by the lane that owns it.
template <class Body, class Done>
__device__ __forceinline__ void walk_groups(const u16* A, u32 pos, u32 nSel, u32* stage_w, Body body, Done done) {
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32* Aw = (const u32*)A;
    const u32 nchunks = (nSel + K34_GROUPS_PER_CHUNK - 1) / K34_GROUPS_PER_CHUNK;
    // software pipeline: the words of chunk c+16 are in flight while chunk c is consumed from LDS
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
    if (w < nchunks) fetch(w);
    for (u32 c = w; c < nchunks; c += 16) {
        const u32 g0 = c * K34_GROUPS_PER_CHUNK;
#pragma unroll
        for (int k = 0; k < K34_PRE; k++) {
            const u32 j = lane + 64u * (u32)k;
            if (j < K34_STAGE_WORDS) stage_w[j] = pre[k];
        }
        __builtin_amdgcn_wave_barrier();
        if (c + 16 < nchunks) fetch(c + 16);
        const u32 gi = g0 + lane;
        if (lane < K34_GROUPS_PER_CHUNK && gi < nSel) {
            const u32 cnt = pos - gi * CJS_GROUP < CJS_GROUP ? pos - gi * CJS_GROUP : CJS_GROUP;
            const u32* gw = stage_w + lane * 25u;
            for (u32 k = 0; k < cnt; k += 2) {
                const u32 wd = gw[k >> 1];
                body(gi, wd & 0xFFFFu);
                if (k + 1 < cnt) body(gi, wd >> 16);
            }
            done(gi);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(1024) void k34_tables(Pipe P) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.x;
    const u32 n = P.nlen[b];
    if (n == 0) return;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32 pos = P.pos[b];
    const int S = (int)P.alpha[b] + 2;
    const u32 nSel = (pos + CJS_GROUP - 1) / CJS_GROUP;                               // :841
    const int target = pos >= 2400 ? 6 : pos >= 1200 ? 5 : pos >= 600 ? 4 : pos >= 200 ? 3 : 2;
    const u16* A = P.A + (size_t)b * g.stride;
    u8* sel = P.sel + (size_t)b * P.selPitch;
    u16* cost = P.selCost + (size_t)b * P.selPitch;

    HIP_DYNAMIC_SHARED(u32, pool)                                                       // [16 * K34_STAGE_WORDS]
    __shared__ u8 lens[CJS_MAX_GROUPS][HB_PITCH];
    __shared__ u32 fr[CJS_MAX_GROUPS][HB_PITCH];
    __shared__ u64 lens64[HB_PITCH];
    __shared__ u32 cnt[8];
    __shared__ u32 scan_sh[20];
    __shared__ u32 s_which, s_cstar, s_keep;
    u32* stage_w = pool + w * K34_STAGE_WORDS;
    int (*keys)[HB_PITCH] = (int (*)[HB_PITCH])pool;                                   // [6][264]
    int (*arr)[HB_PITCH] = (int (*)[HB_PITCH])(pool + CJS_MAX_GROUPS * HB_PITCH);      // [6][264]
    u32* chist = pool + 2 * CJS_MAX_GROUPS * HB_PITCH;                                 // [1024]

    const u32* gfreq = P.freq + (size_t)b * K2_FREQ_PITCH;
    for (u32 i = tid; i < (u32)S; i += 1024) { fr[0][i] = gfreq[i]; fr[1][i] = 1; }   // :835-837
    __syncthreads();
    int G = 2;
    for (;;) {
        // (re)build tables 0..G-1, one wave each
        if (w < (u32)G) huff_build_wave(fr[w], S, lens[w], keys[w], arr[w]);
        __syncthreads();
        for (u32 i = tid; i < (u32)S; i += 1024) {
            u64 v = 0;
            for (int t = 0; t < G; t++) v |= (u64)lens[t][i] << (10 * t);
            lens64[i] = v;
        }
        __syncthreads();
        // assignSelectors :671-684
        {
            u64 acc = 0;
            walk_groups(A, pos, nSel, stage_w,
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
        if (G >= target) break;
        __syncthreads();                                  // staging area is reused as chist below
        if (tid < 8) cnt[tid] = 0;
        chist[tid] = 0;
        __syncthreads();
        for (u32 gi = tid; gi < nSel; gi += 1024) atomicAdd(&cnt[sel[gi]], 1u);
        __syncthreads();
        if (tid == 0) {                                   // first most-used table :699
            u32 wh = 0;
            for (int t = 1; t < G; t++) if (cnt[t] > cnt[wh]) wh = (u32)t;
            s_which = wh;
        }
        __syncthreads();
        const u32 which = s_which;
        for (u32 gi = tid; gi < nSel; gi += 1024)
            if (sel[gi] == which) atomicAdd(&chist[cost[gi]], 1u);
        __syncthreads();
        {
            // smallest c with (#groups of cost < c) + chist[c] > half: elements of stable rank >= half move
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
            for (u32 gi = lo; gi < hi; gi++) eq += (sel[gi] == which && cost[gi] == cstar) ? 1u : 0u;
            u32 total;
            u32 rank = block_excl_scan_1024(eq, scan_sh, &total);
            for (u32 gi = lo; gi < hi; gi++) {
                if (sel[gi] != which) continue;
                const u32 c = cost[gi];
                if (c > cstar) sel[gi] = (u8)G;
                else if (c == cstar) { if (rank >= keep) sel[gi] = (u8)G; rank++; }
            }
        }
        G++;
        for (u32 i = tid; i < (u32)(CJS_MAX_GROUPS * HB_PITCH); i += 1024) (&fr[0][0])[i] = 0;
        __syncthreads();
        walk_groups(A, pos, nSel, stage_w,                 // recount :717-727
            [&](u32 gi, u32 sym) { atomicAdd(&fr[sel[gi]][sym], 1u); },
            [&](u32) {});
        __syncthreads();
    }
    __syncthreads();
    // canonical codes :581-600, one wave per table (lane 0 serial over <= 258 symbols)
    if (w < (u32)G && lane == 0) {
        u32 count[CJS_MAX_BITS + 2], next[CJS_MAX_BITS + 2];
        for (int l = 0; l <= CJS_MAX_BITS + 1; l++) count[l] = 0;
        for (int i = 0; i < S; i++) count[lens[w][i]]++;
        u32 c = 0; int prev = 0;
        for (int l = 1; l <= CJS_MAX_BITS; l++) {
            if (!count[l]) continue;
            c <<= (l - prev);
            next[l] = c;
            c += count[l];
            prev = l;
        }
        u32* codes = P.codes + ((size_t)b * CJS_MAX_GROUPS + w) * CJS_LEN_PITCH;
        u8* lout = P.lens + ((size_t)b * CJS_MAX_GROUPS + w) * CJS_LEN_PITCH;
        for (int i = 0; i < S; i++) { const int l = lens[w][i]; codes[i] = next[l]++; lout[i] = (u8)l; }
    }
    if (tid == 0) { P.ngroups[b] = (u32)G; P.nsel[b] = nSel; }
}

int k34_run(Pipe P, hipStream_t stream) {
    const size_t dyn = (size_t)16 * K34_STAGE_WORDS * 4;
    static const bool lds_ok = hipFuncSetAttribute((const void*)k34_tables, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * K34_STAGE_WORDS * 4)) == hipSuccess;
    if (!lds_ok) return CJS_E_HIP;
    hipLaunchKernelGGL(k34_tables, dim3(P.g.nb), dim3(1024), dyn, stream, P);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}

// ---- standalone allocator entry (HuffmanAllocator.allocateHuffmanCodeLengths, :199-222) ------
// One lane per array; 64-bit cells because callers of the JS function may pass weights whose sum
// exceeds 2^31 (the block pipeline above never does).
__global__ __launch_bounds__(64) void k3_alloc_lengths(long long* arr, const u32* off, u32 count, int maxlen) {
    const u32 t = blockIdx.x * 64u + threadIdx.x;
    if (t >= count) return;
    long long* a = arr + off[t];
    const int len = (int)(off[t + 1] - off[t]);
    if (len == 2) { a[0] = 1; a[1] = 1; }                                   // :201-205
    else if (len == 1) a[0] = 1;
    else if (len > 2) ha_allocate<long long>(a, len, maxlen);
}
int k3_alloc_lengths_run(long long* d_arr, const u32* d_off, u32 count, int maxlen, hipStream_t stream) {
    hipLaunchKernelGGL(k3_alloc_lengths, dim3((count + 63) / 64), dim3(64), 0, stream, d_arr, d_off, count, maxlen);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
