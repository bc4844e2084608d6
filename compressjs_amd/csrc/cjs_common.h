// Shared definitions for the MI355X (gfx950) bzip2 block pipeline.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned char u8;
typedef unsigned short u16;
typedef unsigned int u32;
typedef unsigned long long u64;

// error codes returned through the C ABI (negative); the first block mirrors lib/Bzip2.js:62-72
#define CJS_OK 0
#define CJS_E_LEVEL (-20)      // 'Invalid block size multiplier' (lib/Bzip2.js:888-890)
#define CJS_E_NOSPACE (-21)    // caller's output buffer too small
#define CJS_E_ARG (-22)
#define CJS_E_NOGPU (-23)      // no HIP device: the product has no CPU path
#define CJS_E_UNSUPPORTED (-24)  // input outside what this build handles (e.g. a stream with more magic patterns than bytes / 4)
#define CJS_E_SPEC (-25)       // cjs_bz2_plan_phase: the slice cannot be planned on its own (the caller falls back)
#define CJS_E_HIP (-100)       // -100 - hipError_t

#define CJS_WAVE 64

// ---- geometry of one batch of bzip2 blocks in HBM ------------------------------------------
// Block b of a batch owns element range [b*stride, b*stride + n_b) in every per-element array.
// stride = level*100000-19 rounded up to a multiple of 4096, plus one guard tile.
#define K1_RT 4096          // radix-sort tile (elements per workgroup per pass)
#ifndef K1_HT
#define K1_HT 1024          // refinement tile (suffix-array positions owned by one workgroup)
#endif
#define K1_WIN (2 * K1_HT)  // refinement window: own tile + spill-over of the last owned group
#define K1_WW (K1_WIN / 32 + 2)  // head-bitmap words a refinement window looks at
#define K1_TPAD 64          // T_ext holds n + K1_TPAD bytes: T_ext[i] = T[i mod n]

struct BatchGeom {
    u32 nb;        // blocks in the batch
    u32 stride;    // elements per block slot (multiple of K1_RT)
    u32 tstride;   // bytes per block slot in T_ext (stride + K1_TPAD rounded to 64)
    u32 hstride;   // u32 words per block slot in the head bitmaps
    u32 rtiles;    // stride / K1_RT
    u32 htiles;    // stride / K1_HT
};

static inline BatchGeom make_geom(u32 nb, u32 cap) {
    BatchGeom g;
    g.nb = nb;
    g.stride = (cap + K1_RT - 1) / K1_RT * K1_RT;
    g.tstride = g.stride + 128;
    g.hstride = (g.stride + K1_WIN + 128) / 32;
    g.rtiles = g.stride / K1_RT;
    g.htiles = g.stride / K1_HT;
    return g;
}

#define HIP_CHECK_RET(expr)                                   \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return CJS_E_HIP - (int)e_;     \
    } while (0)

// wave-level helpers (wave64)
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << (threadIdx.x & 63u)) - 1ull; }
