// Device helpers shared by the kernels (wave64, gfx950).
#pragma once
#include "cjs_common.h"

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

