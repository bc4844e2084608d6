// K7: bzip2 block discovery and entropy decode for gfx950 (Bunzip._get_next_block, lib/Bzip2.js:153-366).
//
//   k7_scan_magic   every bit offset of the stream is tested for the 48-bit block magic
//                   (0x314159265359) and the end-of-stream magic (0x177245385090); hits are appended
//                   to a candidate list.  The reference never searches - it reads the next header
//                   where the previous block ended - so the host walks the real chain through the
//                   candidates afterwards (decode.hip) and ignores hits that are not on it.
//   k7_decode       FOUR WAVES per candidate block, a software pipeline through LDS rings (details at the kernel):
//                     A  header, then where the Huffman codes start: per row of 64 bit positions every lane looks up how
//                        far a step (one or two codes) starting at its bit goes; the chain through the row is five scalar
//                        instructions per step (v_readlane, s_bitset1, s_add, s_cmp, s_cbranch);
//                     B  the symbols of a group of 50 (50 lanes at once), RLE2 run lengths from ballots, MTF indices;
//                     C  the MTF recurrence (256-entry list one entry per lane, one DPP shift per literal);
//                     D  expansion into the BWT last column.
//                   At 10^8 bytes there are 112 blocks for 1024 SIMDs: every wave runs alone, an instruction costs 5-9
//                   clocks, a v_readlane -> SGPR -> use round trip 20-30, an LDS round trip 50-60 (tests/microbench/
//                   lone_wave.hip) - the design counts instructions on the serial chains, nothing else.
//                   Output: the BWT last column (dbuf low bytes) of the block, its length, origPtr,
//                   the stored CRC, the bit position where the block ends, or an Err code.
#include "decode.h"

#define WHOLEPI 0x314159265359ull
#define SQRTPI 0x177245385090ull

__global__ __launch_bounds__(256) void k7_scan_magic(const u8* in, u64 len, u64 first_bit, u64* cand, u32* ncand, u32 cap) {
    __shared__ u8 s[256 + 8];
    const u64 b0 = (u64)blockIdx.x * 256u;
    const u32 tid = threadIdx.x;
    s[tid] = b0 + tid < len ? in[b0 + tid] : 0;
    if (tid < 8) s[256 + tid] = b0 + 256 + tid < len ? in[b0 + 256 + tid] : 0;
    __syncthreads();
    if (b0 + tid >= len) return;
    u64 w = 0;
    for (int k = 0; k < 8; k++) w = (w << 8) | s[tid + k];
    for (int sft = 0; sft < 8; sft++) {
        const u64 v = (w >> (16 - sft)) & 0xFFFFFFFFFFFFull;
        const u64 bit = (b0 + tid) * 8u + sft;
        if ((v == WHOLEPI || v == SQRTPI) && bit >= first_bit) {     // bits past the end read as 0 (lib/BitStream.js:84)
            const u32 k = atomicAdd(ncand, 1u);
            if (k < cap) cand[k] = (bit << 1) | (v == SQRTPI ? 1u : 0u);
        }
    }
}

// ---- uniform bit reader over the padded stream (words beyond the stream read as 0: lib/BitStream.js:84)
// Words are appended to the window strictly in order, so chunk c+1 (64 words, one per lane) is always
// requested a whole chunk - 2048 bits - before its first word is needed; it is byte swapped when it
// becomes the current chunk, long after the load was issued.
struct BitRd {
    const u32* w;      // 4-byte aligned stream
    u64 zeroChunk;     // a chunk that lies entirely in the zero padding behind the stream
    u32 ca, cb;        // this lane's word of the current chunk (MSB first) and of the next one (raw)
    u64 win;           // next unread bits, MSB aligned
    int avail;         // valid bits in win
    u64 next;          // index of the next word to append to win
};
__device__ __forceinline__ u32 br_load(const BitRd& r, u64 chunk) {
    const u64 c = chunk < r.zeroChunk ? chunk : r.zeroChunk;      // unconditional load: nothing waits on it here
    return r.w[c * 64u + lane_id()];
}
__device__ __forceinline__ u32 br_word(BitRd& r) {
    const u32 k = (u32)r.next & 63u;
    if (k == 0) { r.ca = __builtin_bswap32(r.cb); r.cb = br_load(r, (r.next >> 6) + 1u); }
    r.next++;
    return (u32)__builtin_amdgcn_readlane((int)r.ca, (int)k);
}
__device__ __forceinline__ void br_init(BitRd& r, const u32* w, u64 zeroChunk, u64 bitpos) {
    r.w = w; r.zeroChunk = zeroChunk;
    r.next = bitpos >> 5;
    const u64 chunk = r.next >> 6;
    r.cb = br_load(r, chunk);
    r.ca = 0;
    if (r.next & 63u) { r.ca = __builtin_bswap32(r.cb); r.cb = br_load(r, chunk + 1u); }
    const u64 hi = br_word(r);
    const u64 lo = br_word(r);
    const int sk = (int)(bitpos & 31u);
    r.win = ((hi << 32) | lo) << sk;
    r.avail = 64 - sk;
}
__device__ __forceinline__ void br_consume(BitRd& r, int n) {
    r.win <<= n;
    r.avail -= n;
    if (r.avail <= 32) {
        r.win |= (u64)br_word(r) << (32 - r.avail);
        r.avail += 32;
    }
}
__device__ __forceinline__ u32 br_get(BitRd& r, int n) {          // 1 <= n <= 32
    const u32 v = (u32)(r.win >> (64 - n));
    br_consume(r, n);
    return v;
}
__device__ __forceinline__ u64 br_tell(const BitRd& r) { return r.next * 32u - (u64)r.avail; }

// set byte `idx` (0..255) of a table spread 4 bytes per lane
__device__ __forceinline__ void tab_set(u32& reg, u32 idx, u32 val) {
    if (lane_id() == (idx >> 2)) reg = (reg & ~(0xffu << (8u * (idx & 3u)))) | (val << (8u * (idx & 3u)));
}

// every cross-wave wait loop gives up after this many s_sleep(1) rounds (~64 clocks each: ~0.2 s) and reports a data error
#ifndef K7_SPIN_CAP
#define K7_SPIN_CAP (1u << 23)
#endif
#define K7_GRING 4u          // groups in flight between waves A and B
#define K7_RROWS 8u          // rows of 64 symbols in flight between waves B, C and D
#define K7_SYMS 512u
#define K7_GROWS 18u         // 50 codes of at most 20 bits: 16 rows of 64 bits and a partial one
#define K7_G_NOCODE 0x1000u  // group record: the symbol after the last one listed has no code
#define K7_G_NOSEL 0x2000u   // group record: there is no selector for this group
#define K7_WIN 512u          // stream words staged in LDS for the per-lane peeks of wave 0
#define K7_UNRES 0x100u      // "no length from the table" mark in the chain of code starts
#define K7_LTBITS 10u        // code lengths up to this many bits come from a table indexed by the next bits

// workgroup-scope publish / observe of a flag in LDS
__device__ __forceinline__ void lds_publish(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// (the value as ONE lane saw it: every lane of the wave takes the same branch on it)
__device__ __forceinline__ u32 lds_observe(u32* p) {
    return (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}

// the 32 stream bits that start at absolute bit q, from the LDS window (MSB-first words)
__device__ __forceinline__ u32 win_peek(const u32* win, u64 q) {
    const u32 w = (u32)(q >> 5), s = (u32)q & 31u;
    const u64 two = ((u64)win[w & (K7_WIN - 1u)] << 32) | win[(w + 1u) & (K7_WIN - 1u)];
    return (u32)(two >> (32u - s));
}

// RUNA/RUNB accumulation (:314-337) over a stretch of `len` run symbols given as bit masks (bit k set in sa / sb = the k-th is
// RUNA / RUNB).  N = run symbols since the run (re)started, T = its value.  The reference's int32 runPos reaches 0 after 32
// run symbols; the next one finds !runPos and restarts with t = 0, and a literal that follows a multiple of 32 run symbols
// finds runPos == 0 and flushes nothing: N counts modulo 32.
struct RunAcc { u32 N; u64 T; };
__device__ __forceinline__ void run_extend(RunAcc& r, u64 sa, u64 sb, u32 len) {
    if (r.N + len < 32u) {
        r.T += (sa << r.N) + 2ull * (sb << r.N);
        r.N += len;
    } else {
        const u32 rp = (r.N + len) & 31u, skip = len - rp;                     // only the symbols after the last restart count
        r.T = rp ? (sa >> skip) + 2ull * (sb >> skip) : 0ull;
        r.N = rp;
    }
}

// the same with the position given as a window word and a bit offset from it (32-bit arithmetic only)
__device__ __forceinline__ u32 win_peek32(const u32* win, u32 w0, u32 q) {
    const u32 w = w0 + (q >> 5), sh = q & 31u;
    const u64 two = ((u64)win[w & (K7_WIN - 1u)] << 32) | win[(w + 1u) & (K7_WIN - 1u)];
    return (u32)(two >> (32u - sh));
}

// the same for a lane that keeps the byte address of its word (not yet wrapped into the window) and 32 - (bit in the word)
__device__ __forceinline__ u32 win_peek_at(const u32* win, u32 wa, u32 shc) {
    const char* b = (const char*)win;
    const u64 two = ((u64)*(const u32*)(b + (wa & (K7_WIN * 4u - 4u))) << 32) | *(const u32*)(b + ((wa + 4u) & (K7_WIN * 4u - 4u)));
    return (u32)(two >> shc);
}

// one move-to-front step on the 256-entry list held one entry per lane in four registers (position p = register p >> 6,
// lane p & 63): returns the entry at idx and moves it to the front (mtf(), lib/Bzip2.js:53-60).  idx is wave-uniform.
__device__ __forceinline__ u32 mtf_step(u32& l0, u32& l1, u32& l2, u32& l3, u32 idx, u32 lane) {
    if (idx < 64u) {                                              // the common case: one v_readlane, one DPP shift, one select
        const u32 src = (u32)__builtin_amdgcn_readlane((int)l0, (int)idx);
        const u32 sh = (u32)__builtin_amdgcn_update_dpp((int)src, (int)l0, 0x138, 0xf, 0xf, false);      // wave_shr:1, lane 0 <- src
        l0 = lane <= idx ? sh : l0;
        return src;
    }
    if (idx < 128u) {                                             // second register (byte alphabets of ~100 symbols end here)
        const u32 r = idx - 64u;
        const u32 c0 = (u32)__builtin_amdgcn_readlane((int)l0, 63);
        const u32 src = (u32)__builtin_amdgcn_readlane((int)l1, (int)r);
        const u32 s1 = (u32)__builtin_amdgcn_update_dpp((int)c0, (int)l1, 0x138, 0xf, 0xf, false);
        l0 = (u32)__builtin_amdgcn_update_dpp((int)src, (int)l0, 0x138, 0xf, 0xf, false);
        l1 = lane <= r ? s1 : l1;
        return src;
    }
    const u32 q = idx >> 6, r = idx & 63u;
    const u32 c0 = (u32)__builtin_amdgcn_readlane((int)l0, 63), c1 = (u32)__builtin_amdgcn_readlane((int)l1, 63);
    const u32 c2 = (u32)__builtin_amdgcn_readlane((int)l2, 63);
    const u32 src = (u32)__builtin_amdgcn_readlane((int)(q == 1u ? l1 : q == 2u ? l2 : l3), (int)r);
    const u32 s0 = (u32)__builtin_amdgcn_update_dpp((int)src, (int)l0, 0x138, 0xf, 0xf, false);
    const u32 s1 = (u32)__builtin_amdgcn_update_dpp((int)c0, (int)l1, 0x138, 0xf, 0xf, false);
    const u32 s2 = (u32)__builtin_amdgcn_update_dpp((int)c1, (int)l2, 0x138, 0xf, 0xf, false);
    const u32 s3 = (u32)__builtin_amdgcn_update_dpp((int)c2, (int)l3, 0x138, 0xf, 0xf, false);
    l0 = s0;
    if (q == 1u) l1 = lane <= r ? s1 : l1;
    else if (q == 2u) { l1 = s1; l2 = lane <= r ? s2 : l2; }
    else { l1 = s1; l2 = s2; l3 = lane <= r ? s3 : l3; }
    return src;
}

// section timers of the profiling build (-DK7_PROF; s_memtime costs ~100 clocks per read, so only there)
#ifdef K7_PROF
#define K7_T(k_) do { const u64 now_ = clock64(); prof_[k_] += now_ - tl_; tl_ = now_; } while (0)
#else
#define K7_T(k_) do { } while (0)
#endif

// Four waves per block, a software pipeline through LDS rings (a group = 50 symbols of one coding table,
// lib/Bzip2.js:283-300; a row = 64 bit positions of the stream (wave A) or 64 symbols (waves B, C, D)):
//   wave A  parses the block header, then finds where the codes start.  Where the next code starts is a serial recurrence, but
//           how far a STEP - the code at a given bit and, when the 10-bit table knows it, the code after it - goes is not: every
//           lane looks that up for its own bit of the row, and the walk through the row is one v_readlane and four scalar
//           instructions per step.  Hands over, per group, the masks of step starts of its rows;
//   wave B  turns the masks into the symbols (canonical index, permute[], end-of-block, the checks of :292-300: 50 lanes at
//           once), regroups them into rows, computes the RLE2 run lengths (:318-347) and the dbufCount checks per row from
//           ballots, and compacts the literals' MTF indices;
//   wave C  runs the MTF recurrence (:53-60) over the compacted literals;
//   wave D  expands rows into the block's last column (DPP prefix sum of run length + 1, one coalesced byte store per row in
//           the common all-literal case).
// Rings: A -> B group records (K7_GRING), B -> C, D symbol rows (K7_RROWS), C -> D output bytes in the same slots.  Waits end on
// the producer's progress or on s_stop / s_abort / s_adone / s_bdone / s_cdone; wave B decides everything that depends on the
// order of symbols (which error or end-of-block comes first), exactly as the sequential reference would meet them.
__global__ __launch_bounds__(256) void k7_decode(DecBuf D, u32 first, u32 count) {
    const u32 slot = blockIdx.x;
    if (slot >= count) return;
    const u32 lane = lane_id();
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably uniform: keeps the wave's state in SGPRs
    __shared__ int s_limLA[6][32];     // left-aligned (to 20 bits) limit of every code length; -1 = length not used
    __shared__ u32 s_limit[6][32];
    __shared__ u32 s_base[6][32];
    __shared__ u16 s_perm[6][384];
    __shared__ u32 s_jeob[6];          // canonical index of the end-of-block symbol in every table
    __shared__ u8 s_len[264];
    __shared__ u16 s_lt[6][1u << K7_LTBITS];  // by the next K7_LTBITS bits: length of the code | length of the code after it << 4 (0 = not within these bits, or no code)
    __shared__ u32 s_win[K7_WIN];
    __shared__ u32 s_mtf[64];          // initial MTF list (= symToByte), 4 entries per lane
    __shared__ u64 s_gmask[K7_GRING][K7_GROWS];   // A -> B: per row of 64 bits of a group, where the steps start ...
    __shared__ u16 s_spos[64];             // B: the same as a list
    __shared__ u64 s_gP[K7_GRING];         // ... the bit it starts at ...
    __shared__ u32 s_ginfo[K7_GRING];      // ... rows | table << 8 | K7_G_* flags | bit where the step after the last one starts << 16
    __shared__ u16 s_sym[K7_SYMS];         // B: symbols regrouped from groups of 50 into rows of 64
    __shared__ u16 s_cidx[K7_RROWS][64];   // B -> C: MTF indices of the row's literals, compacted
    __shared__ u32 s_rrun[K7_RROWS][64];   // B -> D: per symbol lane the run flushed in front of it (count) | literal << 24
    __shared__ u64 s_rmL[K7_RROWS];        // B -> C, D: mask of the row's literals
    __shared__ u32 s_rinfo[K7_RROWS];      // B -> C: 1 = some index of the row is 64 or more
    __shared__ u8 s_out[K7_RROWS][64];     // C -> D: byte of the t-th literal of the row
    __shared__ u32 s_rf0[K7_RROWS];        // C -> D: front of the MTF list before the row
    __shared__ u32 s_ghead, s_gtail, s_stop, s_rhead, s_chead, s_dtail, s_adone, s_bdone, s_cdone, s_abort, s_symTotal, s_hdr;
    __shared__ u32 s_hung;             // a wait loop ran into K7_SPIN_CAP: the block is reported as a data error instead of hanging the queue
    __shared__ int s_pstat;
    __shared__ u32 s_cnt, s_origPtr, s_crc;
    __shared__ u64 s_endbit, s_nsym, s_pwait, s_cwait, s_prof[14];

    u32* gsel = D.sel + (size_t)slot * 4160u;             // [4096 + 64] words: 32768 selectors, 4 bits each, + one row of slack
    const u64 t_start = clock64();
    if (threadIdx.x == 0) { s_ghead = 0; s_gtail = 0; s_stop = 0; s_rhead = 0; s_chead = 0; s_dtail = 0; s_adone = 0; s_bdone = 0; s_cdone = 0; s_abort = 0; s_hung = 0; s_pstat = 0; s_cnt = 0; s_hdr = 0; s_nsym = 0; s_endbit = 0; s_pwait = 0; s_cwait = 0; for (int k = 0; k < 14; k++) s_prof[k] = 0; }
    __syncthreads();
    BitRd r;
    u32 nSel = 0;
    int groupCount = 0;
    if (wave == 0) {
        const u64 start = D.cand[first + slot] >> 1;
        br_init(r, D.in32, D.zeroChunk, start + 48);
        int st = 0;
        const u32 crc = br_get(r, 32);
        u32 origPtr = 0, mw = 0;
        int symTotal = 0, symCount = 0;
        do {
            if (br_get(r, 1)) { st = DEC_OBSOLETE; break; }                    // :174-175
            origPtr = br_get(r, 24);
            {                                                                  // :185-195  symToByte
                const u32 t = br_get(r, 16);
                for (int i = 0; i < 16; i++)
                    if (t & (1u << (15 - i))) {
                        const u32 k = br_get(r, 16);
                        for (int j = 0; j < 16; j++)
                            if (k & (1u << (15 - j))) { tab_set(mw, (u32)symTotal, (u32)(i * 16 + j)); symTotal++; }
                    }
            }
            groupCount = (int)br_get(r, 3);                                    // :198-200
            if (groupCount < 2 || groupCount > 6) { st = DEC_DATA_ERROR; break; }
            nSel = br_get(r, 15);                                              // :205-207
            if (nSel == 0) { st = DEC_DATA_ERROR; break; }
            {                                                                  // :209-221
                u64 list = 0;     // mtfSymbol[0..6]: entries >= groupCount are 0, as in the reference's zeroed buffer
                for (int i = 0; i < groupCount; i++) list |= (u64)i << (8 * i);
                u32 pack = 0;
                for (u32 i = 0; i < nSel; i++) {
                    const u32 v = (u32)(r.win >> 56);                 // next 8 bits
                    const int ones = __builtin_clz(~(v << 24));       // leading 1s
                    if (ones > groupCount) { st = DEC_DATA_ERROR; break; }
                    br_consume(r, ones + 1);
                    const int j = ones;
                    const u64 src = (list >> (8 * j)) & 0xffu;
                    const u64 lowmask = (1ull << (8 * j)) - 1ull;
                    list = (list & ~((1ull << (8 * (j + 1))) - 1ull)) | ((list & lowmask) << 8) | src;
                    pack |= (u32)src << (4u * (i & 7u));
                    if ((i & 7u) == 7u || i + 1 == nSel) { if (lane == 0) gsel[i >> 3] = pack; pack = 0; }
                }
                if (st) break;
            }
            symCount = symTotal + 2;
            for (int g = 0; g < groupCount && !st; g++) {                      // :226-296
                int t = (int)br_get(r, 5);
                for (int i = 0; i < symCount && !st; i++) {
                    for (;;) {
                        if (t < 1 || t > 20) { st = DEC_DATA_ERROR; break; }
                        if (!br_get(r, 1)) break;
                        if (!br_get(r, 1)) t++; else t--;
                    }
                    if (lane == 0) s_len[i] = (u8)t;
                }
                if (st) break;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    int minLen = s_len[0], maxLen = s_len[0];
                    for (int i = 1; i < symCount; i++) { const int l = s_len[i]; if (l > maxLen) maxLen = l; if (l < minLen) minLen = l; }
                    u32 temp[22];
                    for (int i = 0; i < 22; i++) temp[i] = 0;
                    for (int i = 0; i < symCount; i++) temp[s_len[i]]++;
                    u32 startp[22];
                    u32 a = 0;
                    for (int i = 0; i < 22; i++) { startp[i] = a; a += temp[i]; }
                    for (int i = 0; i < 384; i++) s_perm[g][i] = 0;
                    for (int i = 0; i < symCount; i++) {                       // ordered by (length, symbol)
                        const u32 at = startp[s_len[i]]++;
                        s_perm[g][at] = (u16)i;
                        if (i == symCount - 1) s_jeob[g] = at;
                    }
                    for (int i = 0; i < 32; i++) { s_limit[g][i] = 0; s_base[g][i] = 0; s_limLA[g][i] = -1; }
                    u32 pp = 0, tsum = 0;
                    for (int i = minLen; i < maxLen; i++) {
                        pp += temp[i];
                        s_limit[g][i] = pp - 1u;
                        pp <<= 1;
                        tsum += temp[i];
                        s_base[g][i + 1] = pp - tsum;
                    }
                    s_limit[g][maxLen] = pp + temp[maxLen] - 1u;
                    s_base[g][minLen] = 0;
                    for (int i = minLen; i <= maxLen; i++) {           // (v20 >> (20-i)) <= limit[i]  <=>  v20 <= limLA[i]
                        const u32 lim = s_limit[g][i];
                        s_limLA[g][i] = lim >= (1u << i) - 1u ? 0x7fffffff : (int)(((lim + 1u) << (20 - i)) - 1u);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } while (0);
        s_mtf[lane] = mw;
        if (lane == 0) { s_pstat = st; s_origPtr = origPtr; s_crc = crc; s_symTotal = (u32)symTotal; s_hdr = st == 0 ? 1u : 0u; }
        if (st == 0) {
            // length table: the reference takes the SMALLEST i with (next i bits) <= limit[i] (:290-297); for i <= K7_LTBITS
            // that is a function of the next K7_LTBITS bits alone - and so is the code after it when both fit
            for (int g = 0; g < groupCount; g++) {
                int lim[K7_LTBITS + 1];
                for (u32 i = 1; i <= K7_LTBITS; i++) lim[i] = s_limLA[g][i];
                for (u32 idx = lane; idx < (1u << K7_LTBITS); idx += 64u) {
                    const int v = (int)(idx << (20u - K7_LTBITS));
                    u32 L1 = 0, L2 = 0;
                    for (u32 i = K7_LTBITS; i >= 1u; i--) L1 = v <= lim[i] ? i : L1;
                    if (L1) {
                        const int v2 = (int)(((idx << L1) & ((1u << K7_LTBITS) - 1u)) << (20u - K7_LTBITS));
                        for (u32 i = K7_LTBITS; i >= 1u; i--) L2 = (i + L1 <= K7_LTBITS && v2 <= lim[i]) ? i : L2;
                    }
                    s_lt[g][idx] = (u16)(L1 | (L2 << 4));
                }
            }
        }
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)s_hdr)) {
        const u64 lt_mask = (1ull << lane) - 1ull;
        if (wave == 0) {
            // ---- wave A: where the Huffman codes of a group of 50 start -------------------------------
            u64 P = br_tell(r);            // bit where the current group starts
            u32 wl = (u32)(P >> 11) << 6;  // stream words [wl - K7_WIN, wl) are in s_win; pf = raw words of chunk wl / 64
            u32 pf = br_load(r, wl >> 6);
            u32 selrow = 0, selnext = gsel[lane];
            u32 selector = 0, gh = 0;
            u64 pwait = 0;
#ifdef K7_PROF
            u64 prof_[5] = {0, 0, 0, 0, 0}, tl_ = clock64();
#endif
            for (;;) {
                u32 i = 0, g = 0, gend = 0, gexit = 0;
                u32 flags = 0;
                // (wave B raises s_stop whenever it leaves, also on an error; it is looked at only when the ring is full: at most
                // K7_GRING groups of bits behind the block's end are walked for nothing)
                if (gh - lds_observe(&s_gtail) >= K7_GRING) {                  // a free record for this group's row masks
                    const u64 w0 = clock64();
                    u32 spins = 0;
                    while (gh - lds_observe(&s_gtail) >= K7_GRING && !lds_observe(&s_stop)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > K7_SPIN_CAP) { lds_publish(&s_hung, 1u); lds_publish(&s_abort, 1u); lds_publish(&s_stop, 1u); }
                    }
                    pwait += clock64() - w0;
                    if (lds_observe(&s_stop)) break;
                }
                if (selector >= nSel) flags = K7_G_NOSEL;                      // an error only if wave B gets this far (:286-287)
                else {
                    // selectors live in HBM (16 KB per block would cost LDS residency): a register row of 64 words =
                    // 512 selectors, the next row requested one row ahead like the stream words
                    if ((selector & 511u) == 0) { selrow = selnext; selnext = gsel[(((selector >> 9) + 1u) << 6) + lane]; }
                    g = ((u32)__builtin_amdgcn_readlane((int)selrow, (int)((selector >> 3) & 63u)) >> (4u * (selector & 7u))) & 15u;
                    selector++;
                    K7_T(0);
                    const u32 pw = (u32)(P >> 5), pb = (u32)P & 31u;            // P as a stream word and a bit in it
                    // this lane's bit of row r starts shc bits below the top of the word pair at byte wa + 8r of the window
                    const u32 shc = 32u - ((pb + lane) & 31u);
                    u32 wa = (pw + ((pb + lane) >> 5)) * 4u;
                    u32 needw = pw + 9u;                                       // words the peeks of rows 0, 1 and 2 touch
                    while ((int)(wl - needw) < 0) {
                        s_win[(wl + lane) & (K7_WIN - 1u)] = __builtin_bswap32(pf);
                        wl += 64u;
                        pf = br_load(r, wl >> 6);
                    }
                    __builtin_amdgcn_wave_barrier();
                    // Two rows ahead of the chain: the 32 bits at every bit of row r+2 and the table entries of row r+1 are requested
                    // before row r is walked - a lone wave would otherwise sit out both LDS round trips in front of every row.
                    u32 e = s_lt[g][win_peek_at(s_win, wa, shc) >> (32u - K7_LTBITS)];   // table entry of the step starting at this lane's bit
                    u32 pk = win_peek_at(s_win, wa + 8u, shc);                   // the 32 bits at this lane's bit of the NEXT row
                    wa += 16u;
                    // The chain of code starts, a row of 64 bits at a time.  One STEP takes the code at the current bit and, when
                    // the table knows it, the code after it too; all a step does is look up how far it goes (one v_readlane), mark
                    // its start in the row's mask and advance - a handful of scalar instructions.  Wave B finds the symbols from the masks.
                    u32 o = 0, rowb = 0, ns = 0, R = 0;                          // ns = symbols in the rows before this one
                    bool nocode = false;
                    // (gotos: with structured loops and their breaks the compiler spends as many instructions on exit flags as the
                    // walk itself takes; every instruction of a lone wave costs 5-9 clocks)
                    u32 l1, l2, advv, en, pkn, nsn;
                    u32 mlo = 0, mhi = 0;
                    u64 mask, twoM;
                row_top:
                    l1 = e & 15u;
                    l2 = e >> 4;
                    advv = l1 ? l1 + l2 : K7_UNRES;                            // how far a step starting at this lane's bit goes
                    en = s_lt[g][pk >> (32u - K7_LTBITS)];                    // row r+1: requested now, used after the chain
                    needw += 2u;
                    while ((int)(wl - needw) < 0) {
                        s_win[(wl + lane) & (K7_WIN - 1u)] = __builtin_bswap32(pf);
                        wl += 64u;
                        pf = br_load(r, wl >> 6);
                    }
                    __builtin_amdgcn_wave_barrier();
                    pkn = win_peek_at(s_win, wa, shc);                          // row r+2
                    wa += 8u;
                    K7_T(1);
                    mask = 0;                                                  // bit b = a step starts at bit b of the row
#define K7_STEP { const u32 a_ = (u32)__builtin_amdgcn_readlane((int)advv, (int)o); mask = bitset1_b64(mask, o); o += a_; if (o >= 64u) goto left_row; }
                walk_row:
                    K7_STEP K7_STEP K7_STEP K7_STEP K7_STEP K7_STEP K7_STEP K7_STEP
                    goto walk_row;
                left_row:
                    if (o >= K7_UNRES) {
                        const u32 off = o - K7_UNRES;                          // the code at bit `off` is longer than the table's reach,
                        const u32 u20 = win_peek32(s_win, pw, pb + rowb + off) >> 12;   // or there is no code at all
                        const int limLA = lane < 32u ? s_limLA[g][lane] : -1;
                        const u64 m = __ballot((int)u20 <= limLA);             // lane L = length L (:290-297)
                        if (m == 0) { nocode = true; mask &= ~(1ull << off); o = off; }   // i > maxLen (:292)
                        else {
                            o = off + (u32)__builtin_ctzll(m);
                            if (o < 64u) goto walk_row;
                        }
                    }
                    K7_T(2);
                    twoM = __ballot(l2 != 0u) & mask;
                    nsn = ns + (u32)__builtin_popcountll(mask) + (u32)__builtin_popcountll(twoM);
                    mlo = (u32)cjs_writelane((int)(u32)mask, (int)R, (int)mlo);          // lane R of (mhi:mlo) = the mask of row R
                    mhi = (u32)cjs_writelane((int)(u32)(mask >> 32), (int)R, (int)mhi);
                    if (nsn < 50u && !nocode) {
                        ns = nsn;
                        R++;
                        o -= 64u;
                        rowb += 64u;
                        e = en;
                        pk = pkn;
                        goto row_top;
                    }
                    if (nsn >= 50u) {
                        // the row where symbol 50 starts = where the next group starts: the step that reaches 50 (or, taking two, 51);
                        // a bit without a code behind it is the next group's business
                        nocode = false;
                        const u64 le = lt_mask | (1ull << lane);
                        const u32 inc = ns + (u32)__builtin_popcountll(mask & le) + (u32)__builtin_popcountll(twoM & le);
                        const u64 m50 = __ballot(((mask >> lane) & 1ull) && inc >= 50u);
                        const u32 kk = (u32)__builtin_ctzll(m50);
                        if ((u32)__builtin_amdgcn_readlane((int)inc, (int)kk) == 50u) {
                            const u64 above = kk == 63u ? 0ull : (mask >> (kk + 1u)) << (kk + 1u);
                            gend = rowb + (above ? (u32)__builtin_ctzll(above) : o);       // where the next step starts
                        } else gend = rowb + kk + (u32)__builtin_amdgcn_readlane((int)l1, (int)kk);   // after the first of its two codes
                    }
                    i = R + 1u;                                                // rows in the record
                    if (lane < K7_GROWS) s_gmask[gh & (K7_GRING - 1u)][lane] = ((u64)mhi << 32) | mlo;
                    gexit = rowb + o;                                          // where the step after the last marked one starts
                    if (nocode) flags = K7_G_NOCODE;
                }
                if (lane == 0) { s_gP[gh & (K7_GRING - 1u)] = P; s_ginfo[gh & (K7_GRING - 1u)] = i | (g << 8) | flags | (gexit << 16); }
                __builtin_amdgcn_wave_barrier();
                gh++;
                if (lane == 0) lds_publish(&s_ghead, gh);
                K7_T(3);
                if (flags) break;
                P += gend;
            }
#ifdef K7_PROF
            if (lane == 0) for (int k = 0; k < 5; k++) s_prof[k] = prof_[k];
#endif
            if (lane == 0) { s_pwait = pwait; lds_publish(&s_adone, 1u); }
        } else if (wave == 1) {
            // ---- wave B: the symbols of a group; per row of 64 symbols the run lengths and the literals' MTF indices ----
            // RUNA/RUNB symbols never enter the serial MTF loop of wave C: a maximal stretch of them is a bijective base-2 number
            // (:318-335) that every lane ending a stretch takes from two ballots.
            const u32 symTotal = (u32)__builtin_amdgcn_readfirstlane((int)s_symTotal);
            int st = 0;
            u32 gt = 0, np = 0, consumed = 0, rows = 0, cnt = 0;
            u32 runN = 0;                  // RUNA/RUNB symbols since the run (re)started: the reference's runPos == 1 << runN, 0 = no run pending
            u64 runT = 0;
            u64 nsym = 0, endbit = 0, cwait = 0;
            bool eob = false, finished = false;
#ifdef K7_PROF
            u64 prof_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_ = clock64();
#endif
            while (!eob && !st) {
                if (lds_observe(&s_ghead) == gt) {
                    const u64 w0 = clock64();
                    u32 spins = 0;
                    while (lds_observe(&s_ghead) == gt && !lds_observe(&s_adone)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > K7_SPIN_CAP) { lds_publish(&s_hung, 1u); break; }      // falls into the data error below
                    }
                    cwait += clock64() - w0;
                    if (lds_observe(&s_ghead) == gt) { st = DEC_DATA_ERROR; break; }   // wave A has gone (it always leaves a last record: not reached)
                }
                K7_T(5);
                const u32 slotg = gt & (K7_GRING - 1u);
                const u64 P = s_gP[slotg];
                const u32 info = (u32)__builtin_amdgcn_readfirstlane((int)s_ginfo[slotg]);
                const u32 nrows = info & 0xffu, g = (info >> 8) & 15u, gexit = info >> 16;
                if (info & K7_G_NOSEL) { st = DEC_DATA_ERROR; break; }          // more symbols than selectors (:286-287)
                const u32 jeob = (u32)__builtin_amdgcn_readfirstlane((int)s_jeob[g]);
                // where the steps of wave A start, in order: step k into lane k (the first 64 are more than the group's 50 symbols need)
                u32 K = 0;
                for (u32 rr = 0; rr < nrows; rr++) {
                    const u64 mk = s_gmask[slotg][rr];
                    const u32 rk = K + (u32)__builtin_popcountll(mk & lt_mask);
                    if (((mk >> lane) & 1ull) && rk < 64u) s_spos[rk] = (u16)(rr * 64u + lane);
                    K += (u32)__builtin_popcountll(mk);
                }
                if (lane == 0 && K < 64u) s_spos[K] = (u16)gexit;
                if (K > 63u) K = 63u;
                __builtin_amdgcn_wave_barrier();
                const u32 sk = s_spos[lane], sn = s_spos[(lane + 1u) & 63u];
                __builtin_amdgcn_wave_barrier();
                // lane k < K = step k: one code, or two when the table knew both
                const bool have = lane < K;
                const u32 tot = sn - sk;
                u32 c32 = 0, la = tot, lb = 0;
                bool two = false;
                if (have) {
                    c32 = win_peek(s_win, P + sk);
                    const u32 e = s_lt[g][c32 >> (32u - K7_LTBITS)];
                    two = (e >> 4) != 0u;
                    if (two) { la = e & 15u; lb = e >> 4; }
                }
                const u32 cntv = have ? (two ? 2u : 1u) : 0u;
                const u32 incl = wave_incl_scan_dpp(cntv);
                const u32 ia = incl - cntv, ib = ia + 1u;                      // the symbols' numbers in the group
                const u32 listed = (u32)__builtin_amdgcn_readlane((int)incl, 63);
                u32 syma = 0, symb = 0;
                bool bada = false, badb = false, eoba = false, eobb = false;
                if (have) {
                    const u32 ja = (c32 >> (32u - la)) - s_base[g][la];
                    bada = ja >= 258u;                                         // :299-300 (base <= 2^28: no wrap)
                    syma = s_perm[g][bada ? 0u : ja];
                    eoba = !bada && ja == jeob;
                    if (two) {
                        const u32 jb = ((c32 << la) >> (32u - lb)) - s_base[g][lb];
                        badb = jb >= 258u;
                        symb = s_perm[g][badb ? 0u : jb];
                        eobb = !badb && jb == jeob;
                    }
                }
                // the first end-of-block symbol and the first symbol in error, in symbol order
                const u64 mEa = __ballot(eoba), mEb = __ballot(eobb), mBa = __ballot(bada), mBb = __ballot(badb);
                u32 eIdx = 64u, bIdx = 64u, eEnd = 0;
                if (mEa | mEb) {
                    const u32 ke = (u32)__builtin_ctzll(mEa | mEb);
                    const bool first = (mEa >> ke) & 1ull;
                    eIdx = (u32)__builtin_amdgcn_readlane((int)ia, (int)ke) + (first ? 0u : 1u);
                    eEnd = first ? (u32)__builtin_amdgcn_readlane((int)(sk + la), (int)ke) : (u32)__builtin_amdgcn_readlane((int)sn, (int)ke);
                }
                if (mBa | mBb) {
                    const u32 kb = (u32)__builtin_ctzll(mBa | mBb);
                    bIdx = (u32)__builtin_amdgcn_readlane((int)ia, (int)kb) + (((mBa >> kb) & 1ull) ? 0u : 1u);
                }
                if ((info & K7_G_NOCODE) && listed < bIdx) bIdx = listed;       // the symbol after the listed ones has no code
                if (eIdx > 49u) eIdx = 64u;                                    // symbols 50.. belong to the next group: wave A reads them again
                if (bIdx > 49u) bIdx = 64u;
                u32 cntg = 50u;
                if (bIdx < eIdx) st = DEC_DATA_ERROR;
                else if (eIdx < 64u) { eob = true; cntg = eIdx + 1u; endbit = P + eEnd; }
                __builtin_amdgcn_wave_barrier();                               // the record (and the stream words under it) may go now
                gt++;
                if (lane == 0) { lds_publish(&s_gtail, gt); if (eob || st) lds_publish(&s_stop, 1u); }
                if (st) break;
                nsym += cntg;
                if (have && ia < cntg) s_sym[(np + ia) & (K7_SYMS - 1u)] = (u16)syma;
                if (two && ib < cntg) s_sym[(np + ib) & (K7_SYMS - 1u)] = (u16)symb;
                np += cntg;
                __builtin_amdgcn_wave_barrier();
                K7_T(6);
                // whole rows of 64 symbols (the last one of the block as it is)
                while (!st && (np - consumed >= 64u || (eob && np > consumed))) {
                    const u32 hi = np - consumed < 64u ? np - consumed : 64u;
                    const bool valid = lane < hi;
                    const u32 symv = valid ? s_sym[(consumed + lane) & (K7_SYMS - 1u)] : 0u;
                    const u32 idxv = symv - 1u;                                // MTF index of a literal
                    const u64 mE = __ballot(valid && symv > symTotal);         // end of block (:349-350)
                    const u32 eobAt = mE ? (u32)__builtin_ctzll(mE) : 64u;
                    const bool inrow = valid && lane <= eobAt;
                    const bool isL = inrow && symv >= 2u && lane != eobAt;
                    const u64 mA = __ballot(inrow && symv == 0u), mB = __ballot(inrow && symv == 1u), mL = __ballot(isL);
                    const u64 nonrun = mL | (mE ? 1ull << eobAt : 0ull);
                    const bool isN = isL || lane == eobAt;
                    // the run symbols in front of this lane: [start, lane)
                    const u64 below = nonrun & lt_mask;
                    const u32 start = below ? 64u - (u32)__builtin_clzll(below) : 0u;
                    const u32 len2 = lane - start;
                    const u64 seg = (1ull << len2) - 1ull;                     // len2 <= 63
                    RunAcc ra;
                    ra.N = below ? 0u : runN;
                    ra.T = below ? 0ull : runT;
                    run_extend(ra, (mA >> start) & seg, (mB >> start) & seg, len2);
                    const bool fl = isN && ra.N != 0u;                         // :340-347: a pending run is flushed in front of this symbol
                    const u32 flc = fl ? (ra.T > (u64)DEC_CAP ? DEC_CAP + 1u : (u32)ra.T) : 0u;
                    const u32 rowsum = wave_sum_dpp(flc + (isL ? 1u : 0u));
                    if (cnt + rowsum > DEC_CAP) { st = DEC_DATA_ERROR; break; }   // :342, :351
                    cnt += rowsum;
                    // the row's literals, compacted: entry t = MTF index of the t-th literal
                    const u32 rank = (u32)__builtin_popcountll(mL & lt_mask);
                    const bool far = __ballot(isL && idxv >= 64u) != 0ull;
                    const u32 slotr = rows & (K7_RROWS - 1u);
                    if (rows - lds_observe(&s_dtail) >= K7_RROWS) {
                        const u64 w0 = clock64();
                        u32 spins = 0;
                        while (rows - lds_observe(&s_dtail) >= K7_RROWS && !lds_observe(&s_abort)) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > K7_SPIN_CAP) { lds_publish(&s_hung, 1u); lds_publish(&s_abort, 1u); }
                        }
                        cwait += clock64() - w0;
                    }
                    if (isL) s_cidx[slotr][rank] = (u16)idxv;
                    s_rrun[slotr][lane] = flc | (isL ? 1u << 24 : 0u);          // flc < 2^24
                    if (lane == 0) { s_rmL[slotr] = mL; s_rinfo[slotr] = far ? 1u : 0u; }
                    __builtin_amdgcn_wave_barrier();
                    rows++;
                    if (lane == 0) lds_publish(&s_rhead, rows);
                    consumed += hi;
                    if (mE) { finished = true; break; }
                    // the run symbols behind the row's last literal carry over
                    const u32 st0 = nonrun ? 64u - (u32)__builtin_clzll(nonrun) : 0u;
                    const u32 ln = hi - st0;
                    const u64 sg = ln == 64u ? ~0ull : ((1ull << ln) - 1ull);
                    RunAcc c;
                    c.N = nonrun ? 0u : runN;
                    c.T = nonrun ? 0ull : runT;
                    run_extend(c, st0 < 64u ? (mA >> st0) & sg : 0ull, st0 < 64u ? (mB >> st0) & sg : 0ull, ln);
                    runN = c.N;
                    runT = c.T;
                }
                K7_T(7);
            }
            if (st == 0 && !finished) st = DEC_DATA_ERROR;                     // unreachable for st == 0 (a block ends with its end-of-block symbol)
#ifdef K7_PROF
            if (lane == 0) for (int k = 5; k < 10; k++) s_prof[k] = prof_[k];
#endif
            if (lane == 0) {
                s_pstat = st; s_cnt = cnt; s_cwait = cwait; s_endbit = endbit; s_nsym = nsym;
                if (st) lds_publish(&s_abort, 1u);
                lds_publish(&s_stop, 1u);
                lds_publish(&s_bdone, 1u);
            }
        } else if (wave == 2) {
            // ---- wave C: the MTF recurrence over the literals of a row ------------------------------------
            u32 l0, l1, l2, l3;            // the MTF list in the byte domain (entries are symToByte values), one per lane
            {
                const u8* m8 = (const u8*)s_mtf;
                l0 = m8[lane]; l1 = m8[64u + lane]; l2 = m8[128u + lane]; l3 = m8[192u + lane];
            }
            u32 rr = 0, spins = 0;
#ifdef K7_PROF
            u64 prof_[14] = {0}, tl_ = clock64();
#endif
            for (;;) {
                if (lds_observe(&s_rhead) == rr) {
                    if (lds_observe(&s_abort)) break;
                    if (lds_observe(&s_bdone)) { if (lds_observe(&s_rhead) == rr) break; }
                    else {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > K7_SPIN_CAP) { lds_publish(&s_hung, 1u); lds_publish(&s_abort, 1u); }
                        continue;
                    }
                }
                spins = 0;
                K7_T(11);
                const u32 slotr = rr & (K7_RROWS - 1u);
                const u32 cidx = s_cidx[slotr][lane];
                const u32 nlit = (u32)__builtin_popcountll(s_rmL[slotr]);
                const bool far = __builtin_amdgcn_readfirstlane((int)s_rinfo[slotr]) != 0;
                const u32 f0 = (u32)__builtin_amdgcn_readlane((int)l0, 0);       // front of the list before the row's first literal
                u32 outc = 0;                                                  // lane t = byte of the t-th literal
                if (!far) {
                    // mtf(mtfSymbol, sym - 1) :53-60 for an index below 64: one v_readlane, one DPP shift, one select; straight-line
                    // steps, eight per loop branch
#define K7_MTF_STEP(t_) do {                                                                                       \
                        const u32 idx_ = (u32)__builtin_amdgcn_readlane((int)cidx, (int)(t_));                     \
                        const u32 src_ = (u32)__builtin_amdgcn_readlane((int)l0, (int)idx_);                       \
                        const u32 sh_ = (u32)__builtin_amdgcn_update_dpp((int)src_, (int)l0, 0x138, 0xf, 0xf, false); \
                        l0 = lane <= idx_ ? sh_ : l0;                                                              \
                        outc = (u32)cjs_writelane((int)src_, (int)(t_), (int)outc);                                \
                    } while (0)
                    u32 t = 0;
                    for (; t + 8u <= nlit; t += 8u) {
                        K7_MTF_STEP(t); K7_MTF_STEP(t + 1u); K7_MTF_STEP(t + 2u); K7_MTF_STEP(t + 3u);
                        K7_MTF_STEP(t + 4u); K7_MTF_STEP(t + 5u); K7_MTF_STEP(t + 6u); K7_MTF_STEP(t + 7u);
                    }
                    if (nlit & 4u) { K7_MTF_STEP(t); K7_MTF_STEP(t + 1u); K7_MTF_STEP(t + 2u); K7_MTF_STEP(t + 3u); t += 4u; }
                    if (nlit & 2u) { K7_MTF_STEP(t); K7_MTF_STEP(t + 1u); t += 2u; }
                    if (nlit & 1u) K7_MTF_STEP(t);
                } else {
#define K7_MTF_ANY(t_) do {                                                                                        \
                        const u32 idx_ = (u32)__builtin_amdgcn_readlane((int)cidx, (int)(t_));                     \
                        const u32 src_ = mtf_step(l0, l1, l2, l3, idx_, lane);                                     \
                        outc = (u32)cjs_writelane((int)src_, (int)(t_), (int)outc);                                \
                    } while (0)
                    u32 t = 0;
                    for (; t + 4u <= nlit; t += 4u) { K7_MTF_ANY(t); K7_MTF_ANY(t + 1u); K7_MTF_ANY(t + 2u); K7_MTF_ANY(t + 3u); }
                    for (; t < nlit; t++) K7_MTF_ANY(t);
                }
                s_out[slotr][lane] = (u8)outc;
                if (lane == 0) s_rf0[slotr] = f0;
                __builtin_amdgcn_wave_barrier();
                rr++;
                if (lane == 0) lds_publish(&s_chead, rr);
                K7_T(10);
            }
#ifdef K7_PROF
            if (lane == 0) { s_prof[10] = prof_[10]; s_prof[11] = prof_[11]; }
#endif
            if (lane == 0) lds_publish(&s_cdone, 1u);
        } else {
            // ---- wave D: rows -> bytes of the last column ----------------------------------------------------
            u8* out = D.tt + (size_t)slot * D.ttStride;
            u32 rr = 0, opos = 0, spins = 0;
#ifdef K7_PROF
            u64 prof_[14] = {0}, tl_ = clock64();
#endif
            for (;;) {
                if (lds_observe(&s_chead) == rr) {
                    if (lds_observe(&s_abort)) break;
                    if (lds_observe(&s_cdone)) { if (lds_observe(&s_chead) == rr) break; }
                    else {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > K7_SPIN_CAP) { lds_publish(&s_hung, 1u); lds_publish(&s_abort, 1u); }
                        continue;
                    }
                }
                spins = 0;
                K7_T(13);
                const u32 slotr = rr & (K7_RROWS - 1u);
                const u32 rinfo = s_rrun[slotr][lane];
                const u64 mL = s_rmL[slotr];
                const u32 outc = s_out[slotr][lane];
                const u32 f0 = s_rf0[slotr];
                // back to the row's lanes: own byte for a literal, the byte of the literal before for a run
                const u32 rank = (u32)__builtin_popcountll(mL & lt_mask);
                const u32 own = __shfl(outc, (int)(rank & 63u));
                const u32 prevb = __shfl(outc, (int)((rank - 1u) & 63u));
                const u32 rb = rank ? prevb : f0;
                const u32 c = rinfo & 0xffffffu, f = (rinfo >> 24) & 1u;
                const u32 tot = c + f;
                const u32 incl = wave_incl_scan_dpp(tot);
                const u32 at = opos + incl - tot;
                if (f) out[at + c] = (u8)own;                                  // the common case: one coalesced store per row
                if (c && c < 64u) for (u32 i = 0; i < c; i++) out[at + i] = (u8)rb;
                u64 big = __ballot(c >= 64u);                                  // long runs: the whole wave fills each of them
                while (big) {
                    const int l = __builtin_ctzll(big);
                    big &= big - 1;
                    const u32 bc = (u32)__builtin_amdgcn_readlane((int)c, l), bat = (u32)__builtin_amdgcn_readlane((int)at, l);
                    const u32 bb = (u32)__builtin_amdgcn_readlane((int)rb, l);
                    for (u32 i = lane; i < bc; i += 64u) out[bat + i] = (u8)bb;
                }
                opos += (u32)__builtin_amdgcn_readlane((int)incl, 63);
                rr++;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) lds_publish(&s_dtail, rr);
                K7_T(12);
            }
#ifdef K7_PROF
            if (lane == 0) { s_prof[12] = prof_[12]; s_prof[13] = prof_[13]; }
#endif
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        DecResult res;
        int st = s_pstat;
        if (s_hung && st == 0) st = DEC_DATA_ERROR;                              // a bounded wait gave up (never seen; a guard against hanging the GPU queue)
        if (st == 0 && s_origPtr >= s_cnt) st = DEC_DATA_ERROR;               // :372
        res.endbit = s_endbit;
        res.status = st;
        res.n = s_cnt;
        res.origPtr = s_origPtr;
        res.crc = s_crc;
        res.cycles = clock64() - t_start;
        res.symbols = s_nsym;
        res.pwait = s_pwait;
        for (int k = 0; k < 14; k++) res.prof[k] = s_prof[k];
        res.cwait = s_cwait;
        D.res[slot] = res;
    }
}

int k7_scan(const u8* d_in, u64 len, u64 first_bit, u64* d_cand, u32* d_ncand, u32 cap, hipStream_t stream) {
    HIP_CHECK_RET(hipMemsetAsync(d_ncand, 0, 4, stream));
    if (len) hipLaunchKernelGGL(k7_scan_magic, dim3((u32)((len + 255) / 256)), dim3(256), 0, stream, d_in, len, first_bit, d_cand, d_ncand, cap);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
int k7_run(DecBuf D, u32 first, u32 count, hipStream_t stream) {
    hipLaunchKernelGGL(k7_decode, dim3(count), dim3(256), 0, stream, D, first, count);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
