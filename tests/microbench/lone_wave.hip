// Latency of dependent instruction chains for ONE wave on an otherwise idle CU (gfx950): the numbers behind the design of
// k7_decode.  hipcc --offload-arch=gfx950 -O3 lone_wave.hip -o lone_wave && ./lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
typedef unsigned long long u64;
#define REP 4096
#define R4(x) x x x x
#define R16(x) R4(R4(x))

__global__ void k_salu(u64* out, u32* sink) {
    u32 s = (u32)__builtin_amdgcn_readfirstlane((int)sink[0]);
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) { R16(asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");) }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[1] = s; }
}
__global__ void k_valu(u64* out, u32* sink) {
    u32 v = sink[threadIdx.x];
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) { R16(asm volatile("v_add_u32 %0, %0, 3" : "+v"(v));) }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[64 + threadIdx.x] = v;
}
__global__ void k_readlane(u64* out, u32* sink) {       // s = v[s]: VALU -> SGPR -> lane select
    u32 v = (threadIdx.x * 7u + 1u) & 63u;
    u32 s = (u32)__builtin_amdgcn_readfirstlane((int)sink[0]) & 63u;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) { R16(asm volatile("s_nop 3\n\tv_readlane_b32 %0, %1, %0" : "+s"(s) : "v"(v));) }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[1] = s; }
}
__global__ void k_readlane_xor(u64* out, u32* sink) {   // the chain step of wave 0: d = v[o]; o ^= d & m
    u32 v = (threadIdx.x * 7u + 1u) & 63u;
    u32 s = (u32)__builtin_amdgcn_readfirstlane((int)sink[0]) & 63u, d = 0;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) { R16(asm volatile("s_nop 3\n\tv_readlane_b32 %1, %2, %0\n\ts_and_b32 %1, %1, 63\n\ts_xor_b32 %0, %0, %1" : "+s"(s), "+s"(d) : "v"(v) : "scc");) }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[1] = s; }
}
__global__ void k_mtf(u64* out, u32* sink) {            // the MTF step of wave 1: src = l0[idx]; l0 = shift-in
    u32 l0 = threadIdx.x, tmp = 0;
    u32 idx = 5, src = 0;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) {
        R16(asm volatile("v_readlane_b32 %1, %0, %3\n\tv_mov_b32 %2, %1\n\ts_nop 1\n\tv_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32 %0, %0, %2, vcc"
                         : "+v"(l0), "+s"(src), "+v"(tmp) : "s"(idx) : "vcc");)
    }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[64 + threadIdx.x] = l0 + src;
}
__global__ void k_rw(u64* out, u32* sink) {             // readlane -> writelane through an SGPR
    u32 v = threadIdx.x; u32 s = 0;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) { R16(asm volatile("v_readlane_b32 %1, %0, 5\n\ts_nop 3\n\tv_writelane_b32 %0, %1, 7" : "+v"(v), "+s"(s));) }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[64 + threadIdx.x] = v;
}
__global__ void k_lds(u64* out, u32* sink) {            // dependent LDS reads
    __shared__ u32 a[256];
    a[threadIdx.x] = ((threadIdx.x * 13u + 7u) & 63u) * 4u; a[threadIdx.x + 64] = 0; a[threadIdx.x + 128] = 0; a[threadIdx.x + 192] = 0;
    __syncthreads();
    u32 v = threadIdx.x * 4u;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) { R16(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(v));) }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[64 + threadIdx.x] = v + a[0];
}
__global__ void k_bperm(u64* out, u32* sink) {          // dependent ds_bpermute
    u32 v = ((threadIdx.x * 13u + 7u) & 63u) * 4u;
    u32 d = v;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) { R16(asm volatile("ds_bpermute_b32 %0, %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(d));) }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[64 + threadIdx.x] = v;
}
__global__ void k_branch(u64* out, u32* sink) {         // a taken branch per iteration
    u32 s = (u32)__builtin_amdgcn_readfirstlane((int)sink[0]);
    const u64 t0 = clock64();
    asm volatile("s_movk_i32 %0, 0x1000\n1:\n\ts_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b" : "+s"(s) : : "scc");
    const u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[1] = s; }
}
__global__ void k_indep(u64* out, u32* sink) {          // independent SALU + VALU mix (issue rate)
    u32 a = 1, b = 2, c = 3, d = 4; u32 v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) {
        R4(asm volatile("s_add_u32 %0, %0, 1\n\tv_add_u32 %4, %4, 1\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %5, %5, 1\n\ts_add_u32 %2, %2, 1\n\tv_add_u32 %6, %6, 1\n\ts_add_u32 %3, %3, 1\n\tv_add_u32 %7, %7, 1"
                        : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "scc");)
    }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[1] = a + b + c + d; }
    sink[64 + threadIdx.x] = v0 + v1 + v2 + v3;
}

// the walk step of k7_decode's wave A in four stages: o += adv[o] | + s_bitset1 | + compare and an untaken branch | masked, no branch
__global__ void k_walk1(u64* out, u32* sink) {
    u32 v = 1u + (threadIdx.x & 3u);
    u32 o = (u32)__builtin_amdgcn_readfirstlane((int)sink[0]) & 63u, a = 0;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) { R16(asm volatile("v_readlane_b32 %1, %2, %0\n\ts_add_u32 %0, %0, %1\n\ts_and_b32 %0, %0, 63" : "+s"(o), "+s"(a) : "v"(v) : "scc");) }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[1] = o; }
}
__global__ void k_walk2(u64* out, u32* sink) {
    u32 v = 1u + (threadIdx.x & 3u);
    u32 o = (u32)__builtin_amdgcn_readfirstlane((int)sink[0]) & 63u, a = 0; u64 m = 0;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) { R16(asm volatile("v_readlane_b32 %1, %3, %0\n\ts_bitset1_b64 %2, %0\n\ts_add_u32 %0, %0, %1\n\ts_and_b32 %0, %0, 63" : "+s"(o), "+s"(a), "+s"(m) : "v"(v) : "scc");) }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[1] = o + (u32)m; }
}
__global__ void k_walk3(u64* out, u32* sink) {
    u32 v = 1u + (threadIdx.x & 3u);
    u32 o = (u32)__builtin_amdgcn_readfirstlane((int)sink[0]) & 63u, a = 0; u64 m = 0;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) {
        R16(asm volatile("v_readlane_b32 %1, %3, %0\n\ts_bitset1_b64 %2, %0\n\ts_add_u32 %0, %0, %1\n\ts_and_b32 %0, %0, 63\n\ts_cmp_lt_u32 %0, 64\n\ts_cbranch_scc0 9f\n9:" : "+s"(o), "+s"(a), "+s"(m) : "v"(v) : "scc");)
    }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[1] = o + (u32)m; }
}
__global__ void k_walk4(u64* out, u32* sink) {
    u32 v = 1u + (threadIdx.x & 3u);
    u32 o = (u32)__builtin_amdgcn_readfirstlane((int)sink[0]) & 31u, a = 0, in = 0; u64 m = 0;
    const u64 t0 = clock64();
    for (int i = 0; i < REP / 16; i++) {
        R16(asm volatile("s_sub_u32 %3, %0, 64\n\tv_readlane_b32 %1, %4, %0\n\ts_ashr_i32 %3, %3, 31\n\ts_bitset1_b64 %2, %0\n\ts_and_b32 %1, %1, %3\n\ts_add_u32 %0, %0, %1\n\ts_and_b32 %0, %0, 31"
                         : "+s"(o), "+s"(a), "+s"(m), "+s"(in) : "v"(v) : "scc");)
    }
    const u64 t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[1] = o + (u32)m; }
}
#define RUN(k, n, what) do { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_sink); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_sink); \
    hipDeviceSynchronize(); u64 c; hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost); printf("%-34s %7.1f clocks per %s\n", #k, (double)c / (n), what); } while (0)
int main() {
    u64* d_out; u32* d_sink;
    hipMalloc(&d_out, 64); hipMalloc(&d_sink, 4096); hipMemset(d_sink, 0, 4096);
    RUN(k_salu, REP, "dependent s_add");
    RUN(k_valu, REP, "dependent v_add");
    RUN(k_readlane, REP, "s = readlane(v, s) (+s_nop 3)");
    RUN(k_readlane_xor, REP, "readlane + s_and + s_xor");
    RUN(k_mtf, REP, "readlane + mov + dpp + cndmask");
    RUN(k_rw, REP, "readlane -> writelane");
    RUN(k_lds, REP, "dependent ds_read_b32");
    RUN(k_bperm, REP, "dependent ds_bpermute_b32");
    RUN(k_branch, 0x1000, "3-instruction loop iteration");
    RUN(k_indep, REP / 16 * 4 * 8, "independent instruction");
    RUN(k_walk1, REP, "readlane + s_add (+ s_and)");
    RUN(k_walk2, REP, "... + s_bitset1_b64");
    RUN(k_walk3, REP, "... + s_cmp + untaken s_cbranch");
    RUN(k_walk4, REP, "masked step without a branch");
    return 0;
}
