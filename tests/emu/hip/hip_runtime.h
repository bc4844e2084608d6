// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A miniature stand-in for <hip/hip_runtime.h> so that the product's *unmodified* HIP sources
// (compressjs_amd/csrc/*.hip) can be compiled with g++ and executed on the CPU of the build
// container, which has no GPU.  It exists to debug kernel LOGIC (indexing, barriers, wave
// ballots, scans) before spending scarce GPU minutes; it proves nothing about the GPU build and
// no parity claim rests on it.  The product never includes this file: it is reachable only
// through `-I tests/emu` in tests/emu/build_emu.sh.
//
// Model: one workgroup at a time; every GPU thread is a fiber (hand-rolled x86-64 context
// switch); fibers yield at __syncthreads() and at wave collectives (__ballot/__shfl*/__any/__all),
// which are only legal in wave-uniform control flow (the product's kernels obey this).
// Waves are 64 lanes.  Atomics are plain read-modify-writes (fibers are cooperative).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <vector>

#define CJS_CPU_DEBUG_BUILD 1     // the fiber scheduler is single-threaded: the library keeps to one stream
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __constant__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
typedef struct emu_stream_t* hipStream_t;
typedef struct emu_event_t* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace emu {
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    dim3 tid;
};
struct State {
    std::vector<Fiber> fibers;
    int nthreads = 0, cur = 0;
    void* sched_sp = nullptr;
    dim3 bIdx, bDim, gDim;
    // block barrier
    int blk_alive = 0, blk_arrived = 0; unsigned blk_gen = 0;
    // per-wave collectives
    struct Wave { int alive = 0, arrived = 0; unsigned gen = 0; uint64_t slot[64]; uint64_t result = 0; };
    std::vector<Wave> waves;
    void (*body)(void*) = nullptr;
    void* body_arg = nullptr;
    char* dyn_shared = nullptr;
};
State& S();
void yield_();
void run_block(void (*body)(void*), void* arg, dim3 grid, dim3 block, dim3 bidx, size_t shmem);
inline int lane() { return S().cur & 63; }
inline State::Wave& wave() { return S().waves[S().cur >> 6]; }
// wave barrier; returns after all alive lanes of the wave arrived
inline void wave_sync() {
    State::Wave& w = wave();
    w.arrived++;
    if (w.arrived >= w.alive) { w.arrived = 0; w.gen++; return; }
    unsigned g = w.gen;
    while (w.gen == g) yield_();
}
}  // namespace emu

#define threadIdx (emu::S().fibers[emu::S().cur].tid)
#define blockIdx (emu::S().bIdx)
#define blockDim (emu::S().bDim)
#define gridDim (emu::S().gDim)
static const int warpSize = 64;

static inline void __syncthreads() {
    emu::State& s = emu::S();
    s.blk_arrived++;
    if (s.blk_arrived >= s.blk_alive) { s.blk_arrived = 0; s.blk_gen++; return; }
    unsigned g = s.blk_gen;
    while (s.blk_gen == g) emu::yield_();
}
static inline void __threadfence() {}
static inline void __threadfence_block() {}

static inline unsigned long long __ballot(int pred) {
    emu::State::Wave& w = emu::wave();
    w.slot[emu::lane()] = pred ? 1 : 0;
    emu::wave_sync();
    unsigned long long m = 0;
    int base = (emu::S().cur >> 6) << 6;
    for (int l = 0; l < 64; l++)
        if (base + l < emu::S().nthreads && !emu::S().fibers[base + l].done && w.slot[l]) m |= 1ull << l;
    emu::wave_sync();
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) {
    unsigned long long act = __ballot(1);
    return __ballot(pred) == act;
}
template <class T> static inline T emu_shfl_(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl width");
    emu::State::Wave& w = emu::wave();
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    w.slot[emu::lane()] = raw;
    emu::wave_sync();
    uint64_t r = w.slot[src & 63];
    emu::wave_sync();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = emu::lane();
    int s = (l & ~(width - 1)) | (src & (width - 1));
    return emu_shfl_(v, s);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = emu::lane();
    int s = ((l & (width - 1)) >= (int)d) ? l - (int)d : l;
    return emu_shfl_(v, s);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = emu::lane();
    int s = ((l & (width - 1)) + (int)d < width) ? l + (int)d : l;
    return emu_shfl_(v, s);
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) {
    (void)width;
    return emu_shfl_(v, emu::lane() ^ m);
}

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
static inline unsigned __funnelshift_r(unsigned lo, unsigned hi, unsigned sh) {
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return (unsigned)(x >> (sh & 31));
}
static inline unsigned __funnelshift_l(unsigned lo, unsigned hi, unsigned sh) {
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return (unsigned)((x << (sh & 31)) >> 32);
}

template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicXor(T* p, T v) { T o = *p; *p = o ^ v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::S().dyn_shared;

template <class... KArgs, class... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem,
                                      hipStream_t, Args... args) {
    std::tuple<KArgs...> targs(static_cast<KArgs>(args)...);
    struct Pack { void (*k)(KArgs...); std::tuple<KArgs...>* a; } pack{kernel, &targs};
    auto body = [](void* p) {
        Pack* pk = (Pack*)p;
        std::apply(pk->k, *pk->a);
    };
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++)
                emu::run_block(body, &pack, grid, block, dim3(x, y, z), shmem);
}

// ---- host API subset -------------------------------------------------------------------------
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { *fr = *tot = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0
static inline void __builtin_amdgcn_wave_barrier() { emu::wave_sync(); }
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned s) {
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8u * (s & 3u)));
}
template <class T> static inline T __builtin_amdgcn_readlane(T v, int lane) { return emu_shfl_(v, lane & 63); }   // v_readlane_b32 takes the low 6 bits
static inline long long clock64() { return 0; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline void __builtin_amdgcn_s_sleep(int) { emu::yield_(); }
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
template <class T> static inline void __hip_atomic_store(T* p, T v, int, int) { *(volatile T*)p = v; }
template <class T> static inline T __hip_atomic_load(T* p, int, int) { return *(volatile T*)p; }
// DPP (lanes of rows not in row_mask keep `old`): wave_shr:1 (0x138): lane l reads lane l-1, lane 0 keeps `old`; row_shr:n (0x111..0x11f): lane l reads lane l-n of
// its row of 16, lanes whose source falls outside the row keep `old` (bound_ctrl false) or get 0 (bound_ctrl true)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int, bool bound_ctrl) {
    const int l = emu::lane();
    const bool row_on = (row_mask >> (l >> 4)) & 1;
    if (ctrl == 0x138) {
        const int v = emu_shfl_(src, l > 0 ? l - 1 : 0);
        return l > 0 && row_on ? v : old;
    }
    if (ctrl >= 0x111 && ctrl <= 0x11f) {
        const int n = ctrl - 0x110, r = l & 15;
        const int v = emu_shfl_(src, r >= n ? l - n : l);
        if (!row_on) return old;
        return r >= n ? v : (bound_ctrl ? 0 : old);
    }
    if (ctrl == 0x142) {                                   // row_bcast:15: lane 15 of every row to the next row
        const int v = emu_shfl_(src, l >= 16 ? (l & ~15) - 1 : 0);
        return l >= 16 && row_on ? v : old;
    }
    if (ctrl == 0x143) {                                   // row_bcast:31: lane 31 to rows 2 and 3
        const int v = emu_shfl_(src, 31);
        return l >= 32 && row_on ? v : old;
    }
    abort();
}
template <class T> static inline T __builtin_amdgcn_readfirstlane(T v) {
    unsigned long long act = __ballot(1);
    return emu_shfl_(v, __builtin_ctzll(act));
}
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h,
                                          hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t r = 0; r < h; r++) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, w);
    return hipSuccess;
}
