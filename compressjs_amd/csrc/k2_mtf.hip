// K2: symbol map + move-to-front + zero-run (RLE2) coding + symbol frequencies, for gfx950.
//
// Replaces lib/Bzip2.js:743-815 (used-symbol map, linear-search MTF, RUNA/RUNB run coding,
// freq[]).  The reference walks U serially with one 256-entry list; here the work is expressed
// per RUN of equal bytes in U (a run contributes one non-zero MTF index for its first byte, the
// rest of the run are zeros):
//
//   k2_count / k2_scan_tiles / k2_compact : run heads of U -> (symbol, position) arrays, and the
//       256-bit used-symbol set.
//   k2_lastocc / k2_lastscan : for every 1024-run segment, the index of the last run of each
//       symbol before the segment (symbols not seen yet get "virtual" positions -1-rank, which
//       reproduces the initial list M = used bytes ascending, lib/Bzip2.js:773-776).
//   k2_mtf : MTF index of a run head = number of symbols whose last occurrence is later than the
//       previous occurrence of the head's own symbol.  One wave per segment, 64 run heads per
//       step; last occurrences inside the step come from wave ballots, older ones from a
//       per-lane register table (4 symbols per lane), read with v_readlane.
//   k2_mtf (symbol counts per tile) / k2_scan_tiles / k2_emit : each run emits (index+1 if index>0) followed by the
//       bijective base-2 digits of its zero count (lib/Bzip2.js:783-794); exclusive scan gives
//       the output offsets; freq[] by LDS histogram.  EOB appended (lib/Bzip2.js:813-814).
#include "pipeline.h"

__device__ __forceinline__ u32 popc_below(const u32* used8, u32 s) {
    u32 r = 0;
    for (u32 w = 0; w < (s >> 5); w++) r += (u32)__popc(used8[w]);
    if (s & 31u) r += (u32)__popc(used8[s >> 5] & ((1u << (s & 31u)) - 1u));
    return r;
}

// ---- run heads -------------------------------------------------------------------------------
// A wave walks its 1024 bytes of the tile 256 at a time, FOUR per lane in one dword (round 6: two byte loads per element and pass were what k2_count and
// k2_compact spent their instructions on): the byte in front of a lane's four comes from the lane below, in front of lane 0 from the step before.
// Returns the lane's dword and, in hb, bit k = byte k exists and starts a run.
__device__ __forceinline__ u32 k2_heads4(const u8* U, u32 n, u32 i, u32 lane, u32& carry, u32& hb) {
    const u32 wv = i < n ? *(const u32*)(U + i) : 0u;
    u32 pb = (u32)__shfl_up((int)wv, 1u) >> 24;
    if (lane == 0) pb = carry;
    carry = (u32)__builtin_amdgcn_readlane((int)wv, 63) >> 24;
    const u32 x = wv ^ ((wv << 8) | pb);
    u32 h = ((x & 0xFFu) ? 1u : 0u) | ((x & 0xFF00u) ? 2u : 0u) | ((x & 0xFF0000u) ? 4u : 0u) | ((x & 0xFF000000u) ? 8u : 0u);
    if (i == 0) h |= 1u;
    const u32 nv = i >= n ? 0u : (n - i < 4u ? n - i : 4u);
    hb = h & ((1u << nv) - 1u);
    return wv;
}

__global__ __launch_bounds__(256) void k2_count(Pipe P) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 n = P.nlen[b];
    const u32 t0 = t * K1_RT;
    if (threadIdx.x == 0) P.symCnt[(size_t)b * g.rtiles + t] = 0;            // (k2_mtf adds its waves' symbol counts)
    if (t0 >= n) { if (threadIdx.x == 0) P.tileCnt[(size_t)b * g.rtiles + t] = 0; return; }
    __shared__ u32 flag[256];                             // byte value seen (plain stores of 1: the eight words of the map took one LDS atomic per run head, 64 lanes on eight addresses)
    __shared__ u32 cnt;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    flag[tid] = 0;
    if (tid == 0) cnt = 0;
    __syncthreads();
    const u8* U = P.U + (size_t)b * g.stride;
    const u32 base = t0 + w * 1024u;
    u32 carry = base ? (u32)U[base - 1u] : 0u;
    u32 c = 0;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32 i = base + it * 256u + lane * 4u;
        u32 hb;
        const u32 wv = k2_heads4(U, n, i, lane, carry, hb);
        if (hb & 1u) flag[wv & 0xFFu] = 1u;
        if (hb & 2u) flag[(wv >> 8) & 0xFFu] = 1u;
        if (hb & 4u) flag[(wv >> 16) & 0xFFu] = 1u;
        if (hb & 8u) flag[wv >> 24] = 1u;
        c += (u32)__popc(hb);
    }
    c = wave_sum_dpp(c);
    if (lane == 0) atomicAdd(&cnt, c);
    __syncthreads();
    const u64 seen = __ballot(flag[tid] != 0u);
    if (lane < 2u) {                                      // (looking first at what the map already holds costs a round trip at the end of a workgroup that lives for a few microseconds: 0.114 -> 0.244 ms)
        const u32 mine = (u32)(seen >> (32u * lane));
        if (mine) atomicOr(&P.used[(size_t)b * 8 + 2u * w + lane], mine);
    }
    if (tid == 0) P.tileCnt[(size_t)b * g.rtiles + t] = cnt;
}

// per block: exclusive scan of a per-tile count array (<= 256 tiles); total -> tot[b]
__global__ __launch_bounds__(256) void k2_scan_tiles(Pipe P, u32* cnt, u32* tot, int finish_runs) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.x, tid = threadIdx.x;
    __shared__ u32 sh[256];
    __shared__ u32 carry;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (u32 t0 = 0; t0 < g.rtiles; t0 += 256) {
        const u32 t = t0 + tid;
        const u32 v = t < g.rtiles ? cnt[(size_t)b * g.rtiles + t] : 0;
        const u32 ex = block_excl_scan_256(v, sh);
        const u32 base = carry;
        if (t < g.rtiles) cnt[(size_t)b * g.rtiles + t] = base + ex;
        __syncthreads();
        if (tid == 255) carry = base + ex + v;
        __syncthreads();
    }
    if (tid == 0) {
        tot[b] = carry + (finish_runs ? 0u : 1u);     // symbol totals include the EOB
        if (finish_runs) {
            const u32 n = P.nlen[b];
            P.RHpos[(size_t)b * (g.stride + 1) + carry] = n;    // sentinel: end of the last run
            u32 a = 0;
            for (int k = 0; k < 8; k++) a += (u32)__popc(P.used[(size_t)b * 8 + k]);
            P.alpha[b] = a;
        }
    }
}

__global__ __launch_bounds__(256) void k2_compact(Pipe P) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 n = P.nlen[b];
    const u32 t0 = t * K1_RT;
    if (t0 >= n) return;
    __shared__ u32 wtot[4];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u8* U = P.U + (size_t)b * g.stride;
    const u32 base = t0 + w * 1024u;
    u32 carry = base ? (u32)U[base - 1u] : 0u;
    u32 wv[4], hb[4], inc[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
        wv[it] = k2_heads4(U, n, base + it * 256u + lane * 4u, lane, carry, hb[it]);
        inc[it] = wave_incl_scan_dpp((u32)__popc(hb[it]));          // run heads of the lanes up to this one
    }
    if (lane == 63u) wtot[w] = inc[0] + inc[1] + inc[2] + inc[3];
    __syncthreads();
    u32 run = P.tileCnt[(size_t)b * g.rtiles + t];
    for (u32 i = 0; i < w; i++) run += wtot[i];
    u8* RHsym = P.RHsym + (size_t)b * g.stride;
    u32* RHpos = P.RHpos + (size_t)b * (g.stride + 1);
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32 i = base + it * 256u + lane * 4u;
        u32 r = run + inc[it] - (u32)__popc(hb[it]);
#pragma unroll
        for (u32 k = 0; k < 4; k++)
            if ((hb[it] >> k) & 1u) {
                RHsym[r] = (u8)(wv[it] >> (8u * k));
                RHpos[r] = i + k;
                r++;
            }
        run += (u32)__builtin_amdgcn_readlane((int)inc[it], 63);
    }
}

// ---- last-occurrence tables ------------------------------------------------------------------
#define K2_NONE (-0x40000000)

__global__ __launch_bounds__(256) void k2_lastocc(Pipe P) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.y, seg = blockIdx.x;
    const u32 nr = P.nruns[b];
    if (seg * K2_SEG >= nr) return;
    __shared__ int last[256];
    const u32 tid = threadIdx.x;
    last[tid] = K2_NONE;
    __syncthreads();
    const u8* RHsym = P.RHsym + (size_t)b * g.stride;
    for (int k = 0; k < 4; k++) {
        const u32 r = seg * K2_SEG + k * 256u + tid;
        if (r < nr) atomicMax(&last[RHsym[r]], (int)r);
    }
    __syncthreads();
    P.Ltab[((size_t)b * P.segs + seg) * 256 + tid] = last[tid];
}

__global__ __launch_bounds__(256) void k2_lastscan(Pipe P) {
    const u32 b = blockIdx.x, s = threadIdx.x;
    const u32 nr = P.nruns[b];
    const u32 nseg = (nr + K2_SEG - 1) / K2_SEG;
    const u32* used8 = P.used + (size_t)b * 8;
    const bool isused = (used8[s >> 5] >> (s & 31u)) & 1u;
    int cur = isused ? -1 - (int)popc_below(used8, s) : K2_NONE;
    int* L = P.Ltab + (size_t)b * P.segs * 256 + s;
#pragma unroll 4
    for (u32 sg = 0; sg < nseg; sg++) {
        const int v = L[(size_t)sg * 256];
        L[(size_t)sg * 256] = cur;
        cur = cur > v ? cur : v;
    }
}

__device__ __forceinline__ u32 run_symbols(u32 j, u32 len, u32& zeros) {
    zeros = len - (j ? 1u : 0u);
    const u32 nd = zeros ? 31u - (u32)__clz((int)(zeros + 1u)) : 0u;
    return (j ? 1u : 0u) + nd;
}

static_assert(K2_SEG * 4u == K1_RT, "a workgroup of k2_mtf = the runs of one symbol-count tile");
// ---- MTF index of every run head ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k2_mtf(Pipe P) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.y;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32 seg = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + w));      // (wave-uniform: base, the table updates and the loop bounds below stay scalar)
    const u32 nr = P.nruns[b];
    // whole waves leave together; no block-level barrier below
    if (seg * K2_SEG >= nr) return;
    const int* L = P.Ltab + ((size_t)b * P.segs + seg) * 256;
    int Lr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) Lr[k] = L[k * 64 + lane];
    const u32* used8 = P.used + (size_t)b * 8;
    u64 usedw[4];
#pragma unroll
    for (int k = 0; k < 4; k++) usedw[k] = (u64)used8[2 * k] | ((u64)used8[2 * k + 1] << 32);
    const u8* RHsym = P.RHsym + (size_t)b * g.stride;
    u8* J = P.J + (size_t)b * g.stride;
    const u32* RHpos = P.RHpos + (size_t)b * (g.stride + 1);
    u32 nsym = 0;
    const u64 lt = lanemask_lt();
    __shared__ u8 occ[4][256];
    for (u32 i = lane; i < 256; i += 64) occ[w][i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int it = 0; it < 16; it++) {
        const int base = (int)(seg * K2_SEG + it * 64u);
        if ((u32)base >= nr) break;                       // wave-uniform
        const u32 r = (u32)base + lane;
        const bool valid = r < nr;
        const u32 c = valid ? RHsym[r] : 0x1FFu;
        // previous occurrence of my own symbol
        const u64 own = match_any(c, 8, valid) & lt;          // (the lanes past the end are not `valid`: in nobody's mask, and their own is not used)
        int p;
        {
            const int l0 = __shfl(Lr[0], (int)(c & 63u));
            const int l1 = __shfl(Lr[1], (int)(c & 63u));
            const int l2 = __shfl(Lr[2], (int)(c & 63u));
            const int l3 = __shfl(Lr[3], (int)(c & 63u));
            const u32 k = (c >> 6) & 3u;
            const int lsel = k == 0 ? l0 : k == 1 ? l1 : k == 2 ? l2 : l3;
            p = own ? base + 63 - __clzll((long long)own) : lsel;
        }
        // which symbols occur in this 64-run step?  (LDS flags; each lane owns symbols k*64+lane)
        if (valid) occ[w][c] = 1;
        __builtin_amdgcn_wave_barrier();
        u64 occm[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u8 f = occ[w][k * 64 + lane];
            occm[k] = __ballot(f != 0);
        }
        __builtin_amdgcn_wave_barrier();
        if (valid) occ[w][c] = 0;
        u32 idx = 0;
        // Symbols that do not occur in the step: their last occurrence is the table entry, and it can only be later than p for a
        // run head whose own previous occurrence lies BEFORE the step (p >= base beats every table entry).  Those run heads are the
        // first occurrences of the step's distinct symbols - a dozen lanes: for each of them the count over all such symbols is
        // four ballots (lane l of Lr[k] holds symbol 64 k + l).  Round 3; the loop over every symbol of the alphabet that this
        // replaces was 85 (text) to 240 (binaries) iterations per step whoever needed them.
        u64 need = __ballot(valid && own == 0ull);
        u32 nrest = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) nrest += (u32)__popcll(usedw[k] & ~occm[k]);
        // (one iteration per first occurrence here costs about four of the iterations per absent symbol below: random ASCII has
        // 60 first occurrences per step and 95 symbols, text a dozen and 95, binaries two dozen and 250)
        const bool by_lane = (u32)__popcll(need) * 4u < nrest;
        if (by_lane) {
            while (need) {                                 // wave-uniform
                const int j = __builtin_ctzll(need);
                need = bitset0_b64(need, (u32)j);
                const int pj = __builtin_amdgcn_readlane(p, j);
                u32 c2 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) c2 += (u32)__popcll(__ballot(Lr[k] > pj) & usedw[k] & ~occm[k]);
                if ((int)lane == j) idx = c2;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u64 um = usedw[k] & ~occm[k];
                while (um) {                               // wave-uniform
                    const int sl = __builtin_ctzll(um);
                    um = bitset0_b64(um, (u32)sl);
                    idx += (__builtin_amdgcn_readlane(Lr[k], sl) > p) ? 1u : 0u;
                }
            }
        }
        // Symbols that occur in the step: the last occurrence of s before my lane is later than p when s occurs in my WINDOW - the lanes
        // between my own symbol's previous occurrence in this step (or the step's start) and me - or, for a first occurrence of the step
        // (p is a table entry then), when s's table entry is.  Round 5: two ANDs, an OR and two compares per symbol instead of the position
        // of the last occurrence (two v_ffbh, min3, selects: 14 vector instructions -> 9); the table entry of s is replaced by v_writelane.
        const u64 win = own ? (lt & (~0ull << (64 - __clzll((long long)own)))) : lt;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u64 um = occm[k];
            while (um) {
                const int sl = __builtin_ctzll(um);
                um = bitset0_b64(um, (u32)sl);
                const u32 s = (u32)(k * 64 + sl);
                const u64 ms = __ballot(c == s);
                const int Ls = __builtin_amdgcn_readlane(Lr[k], sl);
                idx += ((ms & win) != 0ull || Ls > p) ? 1u : 0u;
                Lr[k] = cjs_writelane(base + 63 - __builtin_clzll(ms), sl, Lr[k]);           // (ms != 0: s occurs)
            }
        }
        if (valid) {
            J[r] = (u8)idx;
            u32 z;
            nsym += run_symbols(idx, RHpos[r + 1u] - RHpos[r], z);         // (RHpos[nr] is the sentinel k2_scan_tiles wrote)
        }
    }
    // RLE2 symbols of the wave's runs into the tile's count (four waves = the 4096 runs of a tile; a kernel of its own until round 6)
    nsym = wave_sum_dpp(nsym);
    if (lane == 0) atomicAdd(&P.symCnt[(size_t)b * g.rtiles + blockIdx.x], nsym);
}

// ---- RLE2 symbol counts, offsets, emission ---------------------------------------------------

// A tile of 4096 runs emits at most 4096 * 21 symbols; the common case fits the LDS stage and is
// written out with consecutive lanes on consecutive addresses, the rest goes straight to memory.
#define K2_STAGE 12288
__global__ __launch_bounds__(256) void k2_emit(Pipe P) {
    const BatchGeom g = P.g;
    const u32 b = blockIdx.y, t = blockIdx.x;
    const u32 nr = P.nruns[b];
    const u32 t0 = t * K1_RT;
    if (t0 >= nr) return;
    __shared__ u32 sh[256];
    __shared__ u32 hist[260];
    __shared__ u16 stage[K2_STAGE];
    __shared__ u32 s_total;
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < 260; i += 256) hist[i] = 0;
    const u8* J = P.J + (size_t)b * g.stride;
    const u32* RHpos = P.RHpos + (size_t)b * (g.stride + 1);
    u16* A = P.A + (size_t)b * g.stride;
    // thread owns 16 consecutive runs
    const u32 r0 = t0 + tid * 16u;
    // MTF indices and run boundaries of the thread's 16 runs in five vector loads, kept for both walks (round 3: they were 16 + 32
    // scalar loads, 16 or 64 bytes apart from lane to lane - a request per lane and load - and read twice)
    u8 jb[16];
    u32 rp[17];
    const bool full = r0 + 16u <= nr;                                 // (RHpos[nr] is the sentinel k2_scan_tiles wrote)
    if (full) {
        __builtin_memcpy(jb, __builtin_assume_aligned(J + r0, 16), 16);
        __builtin_memcpy(rp, __builtin_assume_aligned(RHpos + r0, 4), 68);
    } else {
        for (int k = 0; k < 16; k++) { jb[k] = r0 + k < nr ? J[r0 + k] : 0; rp[k] = r0 + k <= nr ? RHpos[r0 + k] : 0u; }
        rp[16] = r0 + 16u <= nr ? RHpos[r0 + 16u] : 0u;
    }
    u32 mine = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 r = r0 + k;
        if (r < nr) { u32 z; mine += run_symbols(jb[k], rp[k + 1] - rp[k], z); }
    }
    if (r0 <= nr - 1 && nr - 1 < r0 + 16u) mine += 1u;                 // EOB
    const u32 ex = block_excl_scan_256(mine, sh);
    if (tid == 255) s_total = ex + mine;
    __syncthreads();
    const u32 total = s_total;
    const bool staged = total <= K2_STAGE;
    const u32 tile_off = P.symCnt[(size_t)b * g.rtiles + t];
    u32 off = ex;                                                      // tile-relative
    u64 hot = 0;                                                       // (a thread emits at most 16 * 21 + 1 symbols, a wave 64 times that: the fields do not carry)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 r = r0 + k;
        if (r >= nr) break;
        const u32 j = jb[k];
        u32 z;
        run_symbols(j, rp[k + 1] - rp[k], z);
        // (symbols 0 .. 3 - RUNA, RUNB and the two nearest list places, most of what a block emits - are counted in four 16-bit fields of a register and reach the
        // histogram as one sum per wave: as LDS atomics they were dozens of lanes queueing on four addresses)
#define K2_PUT(sym) do { if (staged) stage[off] = (u16)(sym); else A[tile_off + off] = (u16)(sym); off++; \
                         if ((sym) < 4u) hot += 1ull << (16u * (sym)); else atomicAdd(&hist[(sym)], 1u); } while (0)
        if (j) K2_PUT(j + 1u);
        while (z) {                                   // lib/Bzip2.js:783-794
            if (z & 1u) { K2_PUT(0u); z -= 1u; }
            else { K2_PUT(1u); z -= 2u; }
            z >>= 1;
        }
        if (r == nr - 1) {                            // end of block symbol, lib/Bzip2.js:814
            const u32 eob = P.alpha[b] + 1u;
            K2_PUT(eob);
        }
#undef K2_PUT
    }
    {
        const u32 lo = wave_sum_dpp((u32)hot), hi = wave_sum_dpp((u32)(hot >> 32));
        if ((tid & 63u) == 0u) {
            if (lo & 0xFFFFu) atomicAdd(&hist[0], lo & 0xFFFFu);
            if (lo >> 16) atomicAdd(&hist[1], lo >> 16);
            if (hi & 0xFFFFu) atomicAdd(&hist[2], hi & 0xFFFFu);
            if (hi >> 16) atomicAdd(&hist[3], hi >> 16);
        }
    }
    __syncthreads();
    if (staged) for (u32 i = tid; i < total; i += 256) A[tile_off + i] = stage[i];
    u32* freq = P.freq + (size_t)b * K2_FREQ_PITCH;
    for (u32 i = tid; i < 260; i += 256) if (hist[i]) atomicAdd(&freq[i], hist[i]);
}

int k2_run(Pipe P, u32 max_n, hipStream_t stream) {
    const BatchGeom g = P.g;
    const u32 tiles = (max_n + K1_RT - 1) / K1_RT;
    const u32 segs = (max_n + K2_SEG - 1) / K2_SEG;
    const dim3 gridT(tiles, g.nb);
    HIP_CHECK_RET(hipMemsetAsync(P.used, 0, (size_t)g.nb * 8 * 4, stream));
    HIP_CHECK_RET(hipMemsetAsync(P.freq, 0, (size_t)g.nb * K2_FREQ_PITCH * 4, stream));
    hipLaunchKernelGGL(k2_count, dim3(g.rtiles, g.nb), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(k2_scan_tiles, dim3(g.nb), dim3(256), 0, stream, P, P.tileCnt, P.nruns, 1);
    hipLaunchKernelGGL(k2_compact, gridT, dim3(256), 0, stream, P);
    hipLaunchKernelGGL(k2_lastocc, dim3(segs, g.nb), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(k2_lastscan, dim3(g.nb), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(k2_mtf, dim3((segs + 3) / 4, g.nb), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(k2_scan_tiles, dim3(g.nb), dim3(256), 0, stream, P, P.symCnt, P.pos, 0);
    hipLaunchKernelGGL(k2_emit, gridT, dim3(256), 0, stream, P);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
