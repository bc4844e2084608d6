// K8: inverse BWT of decoded bzip2 blocks for gfx950 (lib/Bzip2.js:368-397 + the pointer chase of
// _read_bunzip :405-425), without the n-step serial chase.
//
// The reference counting-sorts the last column into dbuf (T vector in the high 24 bits, the byte in
// the low 8) and then follows pos = dbuf[pos] n times from origPointer.  Here, per block:
//   k8_hist / k8_scan   per-tile byte counts -> stable counting-sort offsets (C[] + earlier tiles)
//   k8_links            word[LF(i)] = (i << 8) | L[i]:  T vector and first-column byte of every row
//   k8_walk<0>          one thread per SPLITTER (every DEC_SPLIT-th row, plus origPtr) follows T
//                       until the next splitter: sublist length and successor splitter
//   k8_rank             one workgroup per block ranks the <= 7034 splitters in LDS (pointer jumping,
//                       13 rounds) -> output index of every sublist; also proves that T is one
//                       n-cycle through origPtr (a corrupt block may not be: then k8_serial walks it
//                       exactly like the reference, repeating the short cycle)
//   k8_walk<1>          the same threads walk again and write the block's bytes in order
// Random 4-byte gathers stay inside one block's 3.6 MB `word` array.
#include "decode.h"

__global__ __launch_bounds__(256) void k8_hist(DecBuf D) {
    const u32 slot = D.slotOf[blockIdx.y];
    const u32 n = D.res[slot].n;
    const u32 t0 = blockIdx.x * DEC_TILE;
    if (t0 >= n) return;
    __shared__ u32 h[256];
    const u32 tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const u8* L = D.tt + (size_t)slot * D.ttStride;
    for (int k = 0; k < 16; k++) {
        const u32 i = t0 + k * 256u + tid;
        if (i < n) atomicAdd(&h[L[i]], 1u);
    }
    __syncthreads();
    D.tileHist[((size_t)slot * DEC_TILES + blockIdx.x) * 256 + tid] = h[tid];
}

__global__ __launch_bounds__(256) void k8_scan(DecBuf D) {
    const u32 slot = D.slotOf[blockIdx.x];
    const u32 n = D.res[slot].n;
    const u32 ntiles = (n + DEC_TILE - 1) / DEC_TILE;
    __shared__ u32 sh[256];
    const u32 c = threadIdx.x;
    u32* th = D.tileHist + (size_t)slot * DEC_TILES * 256;
    u32 tot = 0;
    for (u32 t = 0; t < ntiles; t++) tot += th[(size_t)t * 256 + c];
    u32 run = block_excl_scan_256(tot, sh);
    for (u32 t = 0; t < ntiles; t++) {
        const u32 v = th[(size_t)t * 256 + c];
        th[(size_t)t * 256 + c] = run;
        run += v;
    }
}

__global__ __launch_bounds__(256) void k8_links(DecBuf D) {
    const u32 slot = D.slotOf[blockIdx.y];
    const u32 n = D.res[slot].n;
    const u32 t0 = blockIdx.x * DEC_TILE;
    if (t0 >= n) return;
    __shared__ u32 wh[4][256];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    for (u32 i = tid; i < 1024; i += 256) (&wh[0][0])[i] = 0;
    __syncthreads();
    const u8* L = D.tt + (size_t)slot * D.ttStride;
    u32* word = D.word + (size_t)slot * DEC_STRIDE;
    const u64 lt = lanemask_lt();
    u32 rk[16], cv[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {                      // stable rank of every byte inside its wave's range
        const u32 i = t0 + w * 1024u + it * 64u + lane;
        const bool valid = i < n;
        const u32 c = valid ? L[i] : 0u;
        const u64 m = match_any(c, 8, valid);
        const u32 rank = (u32)__popcll(m & lt), cnt = (u32)__popcll(m);
        const u32 prior = valid ? wh[w][c] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wh[w][c] = prior + cnt;
        __builtin_amdgcn_wave_barrier();
        rk[it] = prior + rank;
        cv[it] = c;
    }
    __syncthreads();
    {
        u32 o = D.tileHist[((size_t)slot * DEC_TILES + blockIdx.x) * 256 + tid];
        for (int ww = 0; ww < 4; ww++) { const u32 c = wh[ww][tid]; wh[ww][tid] = o; o += c; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const u32 i = t0 + w * 1024u + it * 64u + lane;
        if (i < n) word[wh[w][cv[it]] + rk[it]] = (i << 8) | cv[it];          // :380-384
    }
}

// splitter ids of a block: id k < m-1 is row k*DEC_SPLIT; id m-1 is origPtr (the chain head)
__device__ __forceinline__ u32 spl_count(u32 n) { return (n + DEC_SPLIT - 1) / DEC_SPLIT + 1u; }
__device__ __forceinline__ bool is_spl(u32 x, u32 p0) { return (x % DEC_SPLIT) == 0u || x == p0; }
__device__ __forceinline__ u32 spl_id(u32 x, u32 p0, u32 m) { return x == p0 ? m - 1u : x / DEC_SPLIT; }

template <int WRITE>
__global__ __launch_bounds__(256) void k8_walk(DecBuf D, u32 nvalid) {
    // all splitters of one block run on the same XCD: its 3.6 MB `word` array stays in that XCD's L2
    u32 kb, bx;
    if (!xcd_block_tile(nvalid, kb, bx)) return;
    const u32 slot = D.slotOf[kb];
    const u32 n = D.res[slot].n, p0 = D.res[slot].origPtr;
    const u32 m = spl_count(n);
    const u32 id = bx * 256u + threadIdx.x;
    if (id >= m) return;
    if (WRITE && (D.flags[slot] & 1u)) return;
    const u32* word = D.word + (size_t)slot * DEC_STRIDE;
    u32 x = id == m - 1u ? p0 : id * DEC_SPLIT;
    const size_t so = (size_t)slot * DEC_MAXSPL + id;
    if (id != m - 1u && x == p0) {                         // origPtr is a multiple of DEC_SPLIT: id m-1 owns it
        if (!WRITE) { D.splSucc[so] = id; D.splLen[so] = 0; }
        return;
    }
    if (WRITE) {
        u8* pre = D.pre + (size_t)slot * DEC_STRIDE;
        u32 q = D.splOff[so];
        // the sublist's bytes are consecutive in the output: four at a time as one aligned word (a quarter of the store
        // transactions); the bytes in front of the first word boundary and behind the last one singly - those words are shared
        // with the neighbouring sublists
        u32 acc = 0, have = 0;
        do {
            const u32 e = word[x];
            const u32 b = e & 0xffu;                       // first-column byte of row x = next output byte
            x = e >> 8;
            if ((q & 3u) == 0u) { acc = b; have = 1u; }
            else if (have) { acc |= b << (8u * (q & 3u)); have++; }
            else pre[q] = (u8)b;
            q++;
            if (have == 4u) { *(u32*)(pre + q - 4u) = acc; have = 0u; }
        } while (!is_spl(x, p0));
        for (u32 k = 0; k < have; k++) pre[q - have + k] = (u8)(acc >> (8u * k));
    } else {
        u32 cnt = 0;
        do { x = word[x] >> 8; cnt++; } while (!is_spl(x, p0));
        D.splSucc[so] = spl_id(x, p0, m);
        D.splLen[so] = cnt;
    }
}

// distance-to-end ranking of the splitter list (cut in front of the head, origPtr)
__global__ __launch_bounds__(1024) void k8_rank(DecBuf D) {
    const u32 slot = D.slotOf[blockIdx.x];
    const u32 n = D.res[slot].n;
    const u32 m = spl_count(n);
    __shared__ u16 succ[DEC_MAXSPL + 1];
    __shared__ u32 dist[DEC_MAXSPL + 1];
    const u32 tid = threadIdx.x;
    const u32 END = 0xffffu;
    const size_t so = (size_t)slot * DEC_MAXSPL;
    for (u32 s = tid; s < m; s += 1024) {
        const u32 sc = D.splSucc[so + s];
        succ[s] = (u16)(sc == m - 1u ? END : sc);
        dist[s] = D.splLen[so + s];
    }
    __syncthreads();
    constexpr int PER = (DEC_MAXSPL + 1023) / 1024;
    for (u32 span = 1; span < m; span <<= 1) {
        u32 ns[PER], nd[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const u32 s = tid + k * 1024u;
            ns[k] = END; nd[k] = 0;
            if (s < m) {
                const u32 sc = succ[s];
                ns[k] = sc; nd[k] = dist[s];
                if (sc != END) { nd[k] += dist[sc]; ns[k] = succ[sc]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const u32 s = tid + k * 1024u;
            if (s < m) { succ[s] = (u16)ns[k]; dist[s] = nd[k]; }
        }
        __syncthreads();
    }
    // a single n-cycle through origPtr <=> the head reaches END having covered n rows
    const bool ok = succ[m - 1] == END && dist[m - 1] == n;
    if (tid == 0) D.flags[slot] = ok ? 0u : 1u;
    for (u32 s = tid; s < m; s += 1024) D.splOff[so + s] = ok && succ[s] == END ? n - dist[s] : 0u;
}

// the reference's own loop, for blocks whose T vector is not a single cycle (corrupt input)
__global__ __launch_bounds__(64) void k8_serial(DecBuf D) {
    const u32 slot = D.slotOf[blockIdx.x];
    if (!(D.flags[slot] & 1u) || threadIdx.x != 0) return;
    const u32 n = D.res[slot].n;
    const u32* word = D.word + (size_t)slot * DEC_STRIDE;
    u8* pre = D.pre + (size_t)slot * DEC_STRIDE;
    u32 x = D.res[slot].origPtr;
    for (u32 k = 0; k < n; k++) { const u32 e = word[x]; pre[k] = (u8)e; x = e >> 8; }
}

int k8_run(DecBuf D, u32 nvalid, hipStream_t stream) {
    if (!nvalid) return CJS_OK;
    const u32 sg = (DEC_MAXSPL + 255) / 256;
    hipLaunchKernelGGL(k8_hist, dim3(DEC_TILES, nvalid), dim3(256), 0, stream, D);
    hipLaunchKernelGGL(k8_scan, dim3(nvalid), dim3(256), 0, stream, D);
    hipLaunchKernelGGL(k8_links, dim3(DEC_TILES, nvalid), dim3(256), 0, stream, D);
    hipLaunchKernelGGL(k8_walk<0>, dim3(sg, (nvalid + 7u) & ~7u), dim3(256), 0, stream, D, nvalid);
    hipLaunchKernelGGL(k8_rank, dim3(nvalid), dim3(1024), 0, stream, D);
    hipLaunchKernelGGL(k8_walk<1>, dim3(sg, (nvalid + 7u) & ~7u), dim3(256), 0, stream, D, nvalid);
    hipLaunchKernelGGL(k8_serial, dim3(nvalid), dim3(64), 0, stream, D);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
