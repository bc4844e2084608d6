// Random-gather throughput from an L2-resident region (gfx950): what bounds every "key from the text" step of K1
// (k1f_bsort's deepening iterations, the lane kernels).  Each workgroup gathers from the 1 MB region of "its" XCD
// (blockIdx % 8), like the kernels that read one bzip2 block's text.  Variants: bytes per gather, alignment.
//   hipcc --offload-arch=gfx950 -O3 gather.hip -o gather && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned char u8;

__device__ __forceinline__ u32 hash32(u32 h) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; return h; }

// W dwords per gather; ALIGN: 0 = any byte offset rounded down to a dword (what load_be_words does), 1 = rounded down to W*4 bytes
// (natural alignment for W = 1, 2, 4), 2 = rounded down to 16 bytes and W = 4 (aligned dwordx4)
template <int W, int ALIGN, int INFLIGHT>
__global__ __launch_bounds__(256) void k_gather(const u8* base, u32 region_bytes, u32 iters, u32* sink) {
    const u8* T = base + (size_t)(blockIdx.x & 7u) * region_bytes;
    u32 acc = 0;
    u32 h = hash32(blockIdx.x * 256u + threadIdx.x + 12345u);
    for (u32 it = 0; it < iters; it++) {
        u32 d[INFLIGHT][W];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) {
            h = h * 1664525u + 1013904223u;
            u32 p = hash32(h) % (region_bytes - 64u);
            p &= ALIGN == 0 ? ~3u : (ALIGN == 1 ? ~(u32)(W * 4 - 1) : ~15u);
            __builtin_memcpy(d[k], __builtin_assume_aligned(T + p, 4), W * 4);
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++)
#pragma unroll
            for (int j = 0; j < W; j++) acc ^= d[k][j];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// sequential-within-lane variant: the lane walks forward from a random start, STEP bytes per step (lane kernels' pattern)
template <int W>
__global__ __launch_bounds__(256) void k_walk(const u8* base, u32 region_bytes, u32 iters, u32 steps, u32* sink) {
    const u8* T = base + (size_t)(blockIdx.x & 7u) * region_bytes;
    u32 acc = 0;
    u32 h = hash32(blockIdx.x * 256u + threadIdx.x + 999u);
    for (u32 it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        u32 p = (hash32(h) % (region_bytes - 4096u)) & ~3u;
        for (u32 s = 0; s < steps; s++) {
            u32 d[W];
            __builtin_memcpy(d, __builtin_assume_aligned(T + p, 4), W * 4);
#pragma unroll
            for (int j = 0; j < W; j++) acc ^= d[j];
            p += W * 4;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <class F>
static float time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const u32 region = 1u << 20;
    u8* buf; u32* sink;
    hipMalloc(&buf, (size_t)8 * region + 4096);
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, (size_t)8 * region + 4096);
    const u32 iters = 64;
#define RUN(W, A, F, grid) { const u32 g_ = grid; const float ms = time_ms([&]() { hipLaunchKernelGGL((k_gather<W, A, F>), dim3(g_), dim3(256), 0, 0, buf, region, iters, sink); }); \
    const double n = (double)g_ * 256 * iters * F; printf("gather W=%d dwords align=%d inflight=%d grid=%5u: %7.3f ms  %7.1f G gathers/s  %7.1f GB/s useful\n", W, A, F, g_, ms, n / ms / 1e6, n * W * 4 / ms / 1e6); }
    for (u32 grid : {2048u, 8192u}) {
        RUN(1, 1, 4, grid) RUN(2, 1, 4, grid) RUN(2, 0, 4, grid) RUN(3, 0, 4, grid) RUN(4, 0, 4, grid) RUN(4, 2, 4, grid) RUN(8, 0, 4, grid)
        RUN(1, 1, 1, grid) RUN(3, 0, 1, grid) RUN(4, 0, 1, grid) RUN(4, 2, 1, grid)
        RUN(3, 0, 8, grid) RUN(4, 2, 8, grid)
    }
#define RUNW(W, steps) { const u32 g_ = 4096; const float ms = time_ms([&]() { hipLaunchKernelGGL((k_walk<W>), dim3(g_), dim3(256), 0, 0, buf, region, 16u, (u32)steps, sink); }); \
    const double n = (double)g_ * 256 * 16 * steps; printf("walk  W=%d dwords steps=%d: %7.3f ms  %7.1f G loads/s  %7.1f GB/s\n", W, steps, ms, n / ms / 1e6, n * W * 4 / ms / 1e6); }
    RUNW(2, 8) RUNW(3, 8) RUNW(4, 8) RUNW(4, 2) RUNW(8, 4)
    return 0;
}
