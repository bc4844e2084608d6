// Device helpers shared by the kernels (wave64, gfx950).
#pragma once
#include "cjs_common.h"

// v_writelane_b32: `old` with lane `lane` replaced by the wave-uniform `val` (clang has no builtin for it; the LLVM
// intrinsic is reached by its name)
extern "C" __device__ int cjs_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// m | 1 << (b & 63) in ONE scalar instruction where the compiler takes two (b is wave-uniform)
__device__ __forceinline__ u64 bitset1_b64(u64 m, u32 b) {
#if defined(__AMDGCN__)
    asm("s_bitset1_b64 %0, %1" : "+s"(m) : "s"(b));
    return m;
#else
    return m | (1ull << (b & 63u));
#endif
}

// m & ~(1 << (b & 63)) in ONE scalar instruction (b is wave-uniform; "clear the lowest set bit" as m & (m - 1) is three)
__device__ __forceinline__ u64 bitset0_b64(u64 m, u32 b) {
#if defined(__AMDGCN__)
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b));
    return m;
#else
    return m & ~(1ull << (b & 63u));
#endif
}

// The big-endian u32 made of the four bytes at byte offset sh (0..3) of the little-endian dword pair hi:lo: v_alignbyte and the byte
// swap in ONE v_perm_b32 (its selector is computed once per thread and text position: be_sel(sh)).
__device__ __forceinline__ u32 be_sel(u32 sh) {
#if defined(__AMDGCN__)
    // bytes (sh + 3, sh + 2, sh + 1, sh) = the window at byte 3 - sh of the sequence 6 5 4 3 2 1 0; v_alignbyte_b32 takes the low two bits of its
    // count, and ~sh & 3 = 3 - sh (the multiplication this replaces, v_mul_lo_u32, is a quarter-rate instruction)
    return __builtin_amdgcn_alignbyte(0x00000102u, 0x03040506u, ~sh);
#else
    return 0x00010203u + (sh & 3u) * 0x01010101u;
#endif
}
__device__ __forceinline__ u32 be32_at(u32 hi, u32 lo, u32 sel) {
#if defined(__AMDGCN__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    return __builtin_bswap32((u32)((((u64)hi << 32) | lo) >> (8u * ((sel >> 24) & 3u))));
#endif
}

// keeps the compiler from sinking the computation (typically a load) of v below this point
__device__ __forceinline__ void pin_vgpr(u32& v) {
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(v));
#endif
}

// Workgroup barrier that orders LDS only: __syncthreads() also fences global memory, i.e. the compiler puts s_waitcnt vmcnt(0)
// in front of s_barrier and every load a software pipeline keeps in flight across the barrier is drained right there.
// (The hardware does not need that: barriers do not drain VMEM.)  Data exchanged through GLOBAL memory needs __syncthreads().
__device__ __forceinline__ void lds_barrier() {
#if defined(__AMDGCN__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// mask of lanes whose low `nbits` of d equal this lane's (and are valid)
__device__ __forceinline__ u64 match_any(u32 d, int nbits, bool valid) {
    u64 m = __ballot(valid);
    for (int i = 0; i < nbits; i++) {
        const bool bit = (d >> i) & 1u;
        const u64 b = __ballot(bit);
        m &= bit ? b : ~b;
    }
    return m;
}

// exclusive scan of 256 values held one per thread (threads 0..255 of the block); every thread
// of the block must call it.  `sh` is a 256-entry LDS scratch.
__device__ __forceinline__ u32 block_excl_scan_256(u32 v, u32* sh) {
    const u32 tid = threadIdx.x;
    if (tid < 256) sh[tid] = v;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u32 t = 0;
        if (tid < 256 && tid >= off) t = sh[tid - off];
        __syncthreads();
        if (tid < 256) sh[tid] += t;
        __syncthreads();
    }
    const u32 incl = tid < 256 ? sh[tid] : 0;
    __syncthreads();
    return incl - v;
}


__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v) {
    const u32 lane = threadIdx.x & 63u;
    for (u32 off = 1; off < 64; off <<= 1) {
        const u32 u = __shfl_up(v, off);
        if (lane >= off) v += u;
    }
    return v;
}

// inclusive scan / sum over the wave in six DPP adds (rows of 16 by row_shr 1, 2, 4, 8, then row_bcast:15 and row_bcast:31):
// no LDS crossbar round trips - for a wave that runs alone on its SIMD each __shfl costs ~60 clocks, a DPP add ~8
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // lane 15 of rows 0 and 2 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // lane 31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ u32 wave_sum_dpp(u32 v) { return (u32)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(v), 63); }

// exclusive scan over the (up to 1024) threads of a block; every thread must call it.
// `sh` needs 20 u32 of LDS.  *total receives the block-wide sum.
__device__ __forceinline__ u32 block_excl_scan_1024(u32 v, u32* sh, u32* total) {
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
    const u32 nw = (blockDim.x + 63u) >> 6;
    const u32 incl = wave_incl_scan_u32(v);
    if (lane == 63u) sh[w] = incl;
    __syncthreads();
    if (w == 0) {
        const u32 t = lane < nw ? sh[lane] : 0u;
        const u32 ti = wave_incl_scan_u32(t);
        if (lane < nw) sh[lane] = ti - t;
        if (lane == 63u) sh[17] = ti;
    }
    __syncthreads();
    const u32 r = sh[w] + incl - v;
    *total = sh[17];
    __syncthreads();
    return r;
}


// XCD-aware (block, tile) mapping for grids launched as dim3(tiles, round_up(nb, 8)).
// The dispatcher places consecutive workgroup ids round-robin on the 8 XCDs (observed, used for
// speed only): with this remap all tiles of one bzip2 block run on the same XCD, so the block's
// text, suffix array and rank array stay in that XCD's 4 MB L2 while its random gathers and
// scatters are in flight.  Returns false for the padding workgroups.
__device__ __forceinline__ bool xcd_block_tile(u32 nb, u32& b, u32& t) {
    const u32 T = gridDim.x;
    const u32 L = blockIdx.x + T * blockIdx.y;
    const u32 xcd = L & 7u, r = L >> 3;
    t = r % T;
    b = xcd + 8u * (r / T);
    return b < nb;
}
