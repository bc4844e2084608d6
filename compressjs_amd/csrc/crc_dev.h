// CRC-32 (poly 0x04c11db7, MSB first; lib/CRC32.js:37-103) of a byte range by one workgroup of 256 threads or more:
// one independent table-driven slice per thread, combined with x^(8m) mod P.  Shared by K0 (encoder: CRC of
// the input bytes a block consumed) and K9 (decoder: CRC of the bytes a block decodes to).
#pragma once
#include "cjs_common.h"

#define CRC_POLY 0x04c11db7u
__device__ __forceinline__ u32 gf_mul(u32 a, u32 b) {    // a*b mod P over GF(2), P = x^32 + CRC_POLY
    u32 r = 0;
    for (int i = 31; i >= 0; i--) {
        r = (r << 1) ^ ((r & 0x80000000u) ? CRC_POLY : 0u);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}
// v * x^(8*m) mod P using pw[k] = x^(8*2^k) mod P
__device__ __forceinline__ u32 gf_shift(u32 v, u64 m, const u32* pw) {
    for (int k = 0; m; k++, m >>= 1) if (m & 1u) v = gf_mul(v, pw[k]);
    return v;
}

// x^(8 * 2^k) mod P for k < 40, made by the compiler (one lane squaring forty times was 11 us at the head of every call)
struct CrcPw { u32 v[40]; };
constexpr u32 gf_mul_c(u32 a, u32 b) {
    u32 r = 0;
    for (int i = 31; i >= 0; i--) {
        r = (r << 1) ^ ((r & 0x80000000u) ? CRC_POLY : 0u);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}
constexpr CrcPw crc_pw_make() {
    CrcPw t{};
    u32 p = 0x100u;                                       // x^8
    for (int k = 0; k < 40; k++) { t.v[k] = p; p = gf_mul_c(p, p); }
    return t;
}
static __device__ const CrcPw CRC_PW = crc_pw_make();

// All threads of the block (>= 256) call these; tabs[CRC_TAB_WORDS], pw[40] and *acc are LDS scratch.
// crc_range_raw: what the bytes in[ps, pe) contribute to the raw remainder (init 0) of a message that ends at e; the
// contributions of the parts of a message XOR together, and crc = ~(their sum ^ 0xffffffff x^(8 len)) - crc_range_block: the
// CRC of in[s, e) (init 0xffffffff, final complement) by one workgroup.  Both return their value to every thread.
// Four bytes per step (slicing by four: table k = the effect of a byte followed by k zero bytes), every table entry eight
// times over so that the 64 look-ups of a wave spread over the LDS banks (entry v, copy lane & 7): with one 1 KB table the
// slices were a chain of one conflicted look-up per byte (128 us for 112 blocks; the look-ups, not the loads, were the time).
#define CRC_REP 8u
#define CRC_TAB_WORDS (4u * 256u * CRC_REP)
__device__ __forceinline__ u32 crc_range_raw(const u8* in, u64 s, u64 e, u64 e_msg, u32* tabs, u32* pw, u32* acc) {
    const u32 tid = threadIdx.x;
    for (u32 t = tid; t < 1024u; t += blockDim.x) {       // lib/CRC32.js:37-70, carried on through k more zero bytes
        const u32 k = t >> 8, v = t & 255u;
        u32 c = v << 24;
        for (u32 i = 0; i < 8u * (k + 1u); i++) c = (c & 0x80000000u) ? (c << 1) ^ CRC_POLY : (c << 1);
#pragma unroll
        for (u32 r = 0; r < CRC_REP; r++) tabs[t * CRC_REP + r] = c;
    }
    if (tid >= 64 && tid < 104) pw[tid - 64] = CRC_PW.v[tid - 64];
    if (tid == 0) *acc = 0;
    __syncthreads();
    const u32* T0 = tabs + (tid & (CRC_REP - 1u));
    const u32* T1 = T0 + 256u * CRC_REP;
    const u32* T2 = T1 + 256u * CRC_REP;
    const u32* T3 = T2 + 256u * CRC_REP;
    const u64 len = e - s;
    const u64 nt = blockDim.x;
    const u64 per = (((len + nt - 1) / nt) + 15) & ~(u64)15;  // multiple of 16: aligned 16-byte loads inside
    const u64 lo = s + (u64)tid * per < e ? s + (u64)tid * per : e;
    const u64 hi = lo + per < e ? lo + per : e;
    u32 crc = 0;                                          // raw remainder (init 0)
    u64 j = lo;
    for (; j < hi && (((uintptr_t)(in + j)) & 15u); j++) crc = (crc << 8) ^ T0[(((crc >> 24) ^ in[j]) & 0xffu) * CRC_REP];
    auto word = [&](u32 wd) {                             // little-endian load: lowest address in the low byte
        const u32 x = crc ^ __builtin_bswap32(wd);
        crc = T3[(x >> 24) * CRC_REP] ^ T2[((x >> 16) & 0xffu) * CRC_REP] ^ T1[((x >> 8) & 0xffu) * CRC_REP] ^ T0[(x & 0xffu) * CRC_REP];
    };
    // 64 bytes per step, the next 64 already in flight: a slice is 55 dependent steps of one 16-byte load each otherwise,
    // and every one of them waited for HBM (that wait, not the look-ups, was most of the kernel)
    uint4 nx[4];
    bool more = j + 64 <= hi;
    if (more) {
#pragma unroll
        for (int q = 0; q < 4; q++) nx[q] = *(const uint4*)(in + j + 16 * q);
    }
    while (more) {
        uint4 cu[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cu[q] = nx[q];
        j += 64;
        more = j + 64 <= hi;
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; q++) nx[q] = *(const uint4*)(in + j + 16 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { word(cu[q].x); word(cu[q].y); word(cu[q].z); word(cu[q].w); }
    }
    for (; j + 16 <= hi; j += 16) {
        const uint4 v = *(const uint4*)(in + j);
        word(v.x); word(v.y); word(v.z); word(v.w);
    }
    for (; j < hi; j++) crc = (crc << 8) ^ T0[(((crc >> 24) ^ in[j]) & 0xffu) * CRC_REP];
    if (hi > lo) {
        crc = gf_shift(crc, e_msg - hi, pw);
        atomicXor(acc, crc);
    }
    __syncthreads();
    return *acc;
}
__device__ __forceinline__ u32 crc_range_block(const u8* in, u64 s, u64 e, u32* tabs, u32* pw, u32* acc) {
    return ~(crc_range_raw(in, s, e, e, tabs, pw, acc) ^ gf_shift(0xffffffffu, e - s, pw));
}
