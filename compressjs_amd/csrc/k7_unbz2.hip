// K7: bzip2 block discovery and entropy decode for gfx950 (Bunzip._get_next_block, lib/Bzip2.js:153-366).
//
//   k7_scan_magic   every bit offset of the stream is tested for the 48-bit block magic
//                   (0x314159265359) and the end-of-stream magic (0x177245385090); hits are appended
//                   to a candidate list.  The reference never searches - it reads the next header
//                   where the previous block ended - so the host walks the real chain through the
//                   candidates afterwards (decode.hip) and ignores hits that are not on it.
//   k7_decode       two WAVES per candidate block (code boundaries | symbols, RLE2, MTF), pipelined
//                   through an LDS ring.  The 64 lanes of a wave execute one uniform instruction
//                   stream (the Huffman/MTF recurrences are serial) and use the vector registers as
//                   tables indexed with v_readlane instead of going to LDS for every symbol:
//                     - 2 x 64 stream words prefetched per lane (coalesced 256-byte loads),
//                     - limit[] / base[] of the current coding table: lane L holds length L; the
//                       code length is one ballot of "prefix(L) <= limit[L]" (:290-297),
//                     - permute[] of the current table: 3 registers x 64 lanes (two u16 each),
//                     - the 256-entry MTF list, kept in the byte domain (entries are symToByte values):
//                       4 bytes per lane, shifted with one DPP wave_shr,
//                     - output: one (byte, count) token per literal/run, expanded 64 tokens at a time.
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

#define K7_RING 4096u        // u16 symbols in flight between waves 0 and 1
#define K7_TROWS 8u          // token rows in flight between waves 1 and 2
#define K7_WIN 256u          // stream words staged in LDS for the per-lane peeks of wave 0
#define K7_UNRES 0x100u      // "no length from the table" mark in the chain of code starts
#define K7_LTBITS 10u        // code lengths up to this many bits come from a table indexed by the next bits

// workgroup-scope publish / observe of a flag in LDS
__device__ __forceinline__ void lds_publish(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ u32 lds_observe(u32* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }

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

// one move-to-front step on the 256-entry list held one entry per lane in four registers (position p = register p >> 6,
// lane p & 63): returns the entry at idx and moves it to the front (mtf(), lib/Bzip2.js:53-60).  idx is wave-uniform.
__device__ __forceinline__ u32 mtf_step(u32& l0, u32& l1, u32& l2, u32& l3, u32 idx, u32 lane) {
    if (idx < 64u) {                                              // the common case: one v_readlane, one DPP shift, one select
        const u32 src = (u32)__builtin_amdgcn_readlane((int)l0, (int)idx);
        const u32 sh = (u32)__builtin_amdgcn_update_dpp((int)src, (int)l0, 0x138, 0xf, 0xf, false);      // wave_shr:1, lane 0 <- src
        l0 = lane <= idx ? sh : l0;
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

// Three waves per block, a software pipeline through two LDS rings:
//   wave 0  parses the block header, then decodes the Huffman symbols one GROUP of 50 at a time (one coding table per
//           group, lib/Bzip2.js:283-300).  Where the next code starts is a serial recurrence, but how long the code
//           starting at a given bit is, is not: every lane looks the length up for its own bit offset (64 offsets per
//           row, a table indexed by the next 10 bits), the chain of code starts is then followed with one v_readlane
//           per symbol, and the 50 symbols themselves (canonical index, permute[], end-of-block, the checks of
//           :292-300) are extracted by 50 lanes at once;
//   wave 1  undoes RLE2 and MTF and emits one (byte, count) token per literal or run (:305-366);
//   wave 2  expands 64 tokens per step into the block's last column (prefix sum of the counts, one
//           coalesced byte store per row in the common all-literal case).
__global__ __launch_bounds__(192) void k7_decode(DecBuf D, u32 first, u32 count) {
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
    __shared__ u8 s_lt[6][1u << K7_LTBITS];   // code length by the next K7_LTBITS bits; 0 = longer than that, or no code
    __shared__ u32 s_win[K7_WIN];
    __shared__ u16 s_ring[K7_RING];
    __shared__ u32 s_mtf[64];          // initial MTF list (= symToByte), 4 entries per lane
    __shared__ u32 s_head, s_tail, s_done, s_abort, s_symTotal, s_hdr;
    __shared__ u32 s_trun[K7_TROWS * 64u];   // per symbol lane: the run flushed in front of it (count | byte << 24) ...
    __shared__ u32 s_tlit[K7_TROWS * 64u];   // ... and its own literal (byte | 1 << 8), rows in flight between waves 1 and 2
    __shared__ u32 s_cidx[64];
    __shared__ u32 s_thead, s_ttail, s_tdone;
    __shared__ int s_pstat, s_cstat;
    __shared__ u32 s_cnt, s_origPtr, s_crc;
    __shared__ u64 s_endbit, s_nsym, s_pwait, s_cwait;

    u32* gsel = D.sel + (size_t)slot * 4160u;             // [4096 + 64] words: 32768 selectors, 4 bits each, + one row of slack
    const u64 t_start = clock64();
    if (threadIdx.x == 0) { s_thead = 0; s_ttail = 0; s_tdone = 0; s_head = 0; s_tail = 0; s_done = 0; s_abort = 0; s_pstat = 0; s_cstat = 0; s_cnt = 0; s_hdr = 0; s_nsym = 0; s_endbit = 0; s_pwait = 0; s_cwait = 0; }
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
            // that is a function of the next K7_LTBITS bits alone
            for (int g = 0; g < groupCount; g++) {
                int lim[K7_LTBITS + 1];
                for (u32 i = 1; i <= K7_LTBITS; i++) lim[i] = s_limLA[g][i];
                for (u32 idx = lane; idx < (1u << K7_LTBITS); idx += 64u) {
                    const int v = (int)(idx << (20u - K7_LTBITS));
                    u32 L = 0;
                    for (u32 i = K7_LTBITS; i >= 1u; i--) L = v <= lim[i] ? i : L;
                    s_lt[g][idx] = (u8)L;
                }
            }
        }
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)s_hdr)) {
        if (wave == 0) {
            // ---- wave 0: Huffman symbols, one group of 50 per step ------------------------------------
            int st = 0;
            u64 P = br_tell(r);            // bit where the current group starts
            u64 wl = (P >> 11) << 6;       // stream words [wl - K7_WIN, wl) are in s_win; pf = raw words of chunk wl / 64
            u32 pf = br_load(r, wl >> 6);
            u32 selrow = 0, selnext = gsel[lane];
            u32 selector = 0, np = 0;
            u64 nsym = 0, pwait = 0;
            bool eob = false;
            while (!eob) {
                if (selector >= nSel) { st = DEC_DATA_ERROR; break; }
                // selectors live in HBM (16 KB per block would cost LDS residency): a register row of 64 words =
                // 512 selectors, the next row requested one row ahead like the stream words
                if ((selector & 511u) == 0) { selrow = selnext; selnext = gsel[(((selector >> 9) + 1u) << 6) + lane]; }
                const u32 g = ((u32)__builtin_amdgcn_readlane((int)selrow, (int)((selector >> 3) & 63u)) >> (4u * (selector & 7u))) & 15u;
                selector++;
                const int limLA = lane < 32u ? s_limLA[g][lane] : -1;
                const u32 jeob = (u32)__builtin_amdgcn_readfirstlane((int)s_jeob[g]);
                // code starts of the group: pos[i] (relative to P) in lane i of posv, i = 0..50 (pos[50] = end of the group)
                u32 posv = 0, i = 0, o = 0, rowb = 0;
                bool nocode = false;
                for (;;) {
                    const u64 rowbit = P + rowb;
                    const u64 need = ((rowbit + 63u) >> 5) + 2u;           // words the peeks of this row (and of the group so far) touch
                    while (wl < need) {
                        s_win[((u32)wl + lane) & (K7_WIN - 1u)] = __builtin_bswap32(pf);
                        wl += 64u;
                        pf = br_load(r, wl >> 6);
                    }
                    __builtin_amdgcn_wave_barrier();
                    const u32 v20 = win_peek(s_win, rowbit + lane) >> 12;
                    const u32 L = s_lt[g][v20 >> (20u - K7_LTBITS)];
                    const u32 dxv = ((L ? L : K7_UNRES) + lane) ^ lane;    // (next code start if a code started at this lane's bit) ^ lane
                    // the chain of code starts through this row: straight-line steps (a taken branch costs a lone wave more than
                    // the step itself), steps past the row's end or past symbol 50 change nothing that is read later
                    for (;;) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            posv = (u32)cjs_writelane((int)(rowb + o), (int)i, (int)posv);
                            const u32 in = (u32)((int)(o - 64u) >> 31);                    // all ones while o is inside the row
                            const u32 dx = (u32)__builtin_amdgcn_readlane((int)dxv, (int)o);
                            i -= in;                                                       // i += 1
                            o ^= dx & in;                                                  // o = next code start
                        }
                        if (o < 64u && i < 50u) continue;
                        if (o >= K7_UNRES && i <= 50u) {                       // symbol i-1 (at bit o - K7_UNRES of the row): longer than the
                            const u32 off = o - K7_UNRES;                      // table's reach, or no code at all
                            const u32 u20 = win_peek(s_win, rowbit + off) >> 12;
                            const u64 m = __ballot((int)u20 <= limLA);         // lane L = length L (:290-297)
                            if (m == 0) { nocode = true; i -= 1u; o = off; break; }   // i > maxLen (:292)
                            o = off + (u32)__builtin_ctzll(m);
                            if (o < 64u && i < 50u) continue;
                        }
                        break;
                    }
                    if (nocode || i >= 50u) break;
                    o -= 64u;
                    rowb += 64u;
                }
                posv = (u32)cjs_writelane((int)(rowb + o), (int)i, (int)posv);
                if (i > 50u) i = 50u;
                // lanes < i hold the start of a code whose end is the next lane's start
                const u32 pn = __shfl_down(posv, 1);
                const u32 len = pn - posv;
                const bool have = lane < i;
                u32 sym = 0;
                bool badj = false, iseob = false;
                if (have) {
                    const u32 c20 = win_peek(s_win, P + posv) >> 12;
                    const u32 j = (c20 >> (20u - len)) - s_base[g][len];
                    badj = j >= 258u;                                          // :299-300 (base <= 2^28: no wrap)
                    sym = s_perm[g][badj ? 0u : j];
                    iseob = !badj && j == jeob;
                }
                u64 mBad = __ballot(badj);
                if (nocode) mBad |= 1ull << i;
                const u64 mEob = __ballot(iseob);
                const u32 e = mEob ? (u32)__builtin_ctzll(mEob) : 64u, b = mBad ? (u32)__builtin_ctzll(mBad) : 64u;
                u32 cntg = 50u;
                if (b < e) { st = DEC_DATA_ERROR; cntg = b; }
                else if (e < 64u) { eob = true; cntg = e + 1u; }
                nsym += cntg;
                if (np + cntg - lds_observe(&s_tail) > K7_RING) {
                    const u64 w0 = clock64();
                    while (np + cntg - lds_observe(&s_tail) > K7_RING && !lds_observe(&s_abort)) __builtin_amdgcn_s_sleep(2);
                    pwait += clock64() - w0;
                }
                if (lane < cntg) s_ring[(np + lane) & (K7_RING - 1u)] = (u16)sym;
                np += cntg;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) lds_publish(&s_head, np);
                P += (u32)__builtin_amdgcn_readlane((int)pn, (int)(cntg ? cntg - 1u : 0u));       // end of the last symbol taken
                if (st || lds_observe(&s_abort)) break;
            }
            if (lane == 0) { s_pstat = st; s_endbit = P; s_nsym = nsym; s_pwait = pwait; lds_publish(&s_done, 1u); }
        } else if (wave == 1) {
            // ---- wave 1: symbols -> RLE2 -> MTF -> per row of 64 symbols a (run, literal) token pair per lane --------
            // Only the MTF recurrence over the row's literals is serial.  RUNA/RUNB symbols never enter it: a maximal stretch
            // of them is a bijective base-2 number (:318-335) that every lane ending a stretch takes from two ballots, and the
            // byte a run repeats is the one the literal before it produced.
            const u32 symTotal = (u32)__builtin_amdgcn_readfirstlane((int)s_symTotal);
            u32 l0, l1, l2, l3;            // the MTF list in the byte domain (entries are symToByte values), one per lane
            {
                const u8* m8 = (const u8*)s_mtf;
                l0 = m8[lane]; l1 = m8[64u + lane]; l2 = m8[128u + lane]; l3 = m8[192u + lane];
            }
            u32 cnt = 0, trow = 0;
            int st = 0;
            u32 runN = 0;                  // RUNA/RUNB symbols since the run (re)started: the reference's runPos == 1 << runN, 0 = no run pending
            u64 runT = 0;
            u32 consumed = 0;              // always a multiple of 64 while wave 0 runs
            u64 cwait = 0;
            bool finished = false;
            const u64 lt_mask = (1ull << lane) - 1ull;
            while (!finished) {
                const u32 b0 = consumed;
                u32 head = lds_observe(&s_head);
                if (head - b0 < 64u) {                                         // wait for the whole row unless wave 0 has stopped
                    if (lds_observe(&s_done)) {
                        head = lds_observe(&s_head);
                        if (head == b0) break;                                 // wave 0 stopped without end-of-block
                    } else { const u64 w0 = clock64(); __builtin_amdgcn_s_sleep(2); cwait += clock64() - w0; continue; }
                }
                const u32 hi = head - b0 < 64u ? head - b0 : 64u;              // symbols [b0, b0+hi) are ready
                const bool valid = lane < hi;
                const u32 symv = valid ? s_ring[(b0 + lane) & (K7_RING - 1u)] : 0u;
                const u32 idxv = symv - 1u;                                    // MTF index of a literal
                const u64 mE = __ballot(valid && symv > symTotal);             // end of block (:349-350)
                const u32 eobAt = mE ? (u32)__builtin_ctzll(mE) : 64u;
                const bool inrow = valid && lane <= eobAt;
                const bool isL = inrow && symv >= 2u && lane != eobAt;
                const u64 mA = __ballot(inrow && symv == 0u), mB = __ballot(inrow && symv == 1u), mL = __ballot(isL);
                const u64 nonrun = mL | (mE ? 1ull << eobAt : 0ull);
                const bool isN = isL || lane == eobAt;
                // the run symbols in front of this lane: [start, lane)
                const u64 below = nonrun & lt_mask;
                const u32 start = below ? 64u - (u32)__builtin_clzll(below) : 0u;
                const u32 len = lane - start;
                const u64 seg = (1ull << len) - 1ull;                          // len <= 63
                RunAcc ra;
                ra.N = below ? 0u : runN;
                ra.T = below ? 0ull : runT;
                run_extend(ra, (mA >> start) & seg, (mB >> start) & seg, len);
                const bool fl = isN && ra.N != 0u;                             // :340-347: a pending run is flushed in front of this symbol
                const u32 flc = fl ? (ra.T > (u64)DEC_CAP ? DEC_CAP + 1u : (u32)ra.T) : 0u;
                u32 rowsum = flc + (isL ? 1u : 0u);
                for (int off = 32; off > 0; off >>= 1) rowsum += __shfl_xor(rowsum, off);
                rowsum = (u32)__builtin_amdgcn_readfirstlane((int)rowsum);
                if (cnt + rowsum > DEC_CAP) { st = DEC_DATA_ERROR; if (lane == 0) lds_publish(&s_abort, 1u); break; }   // :342, :351
                cnt += rowsum;
                // the row's literals, compacted: lane t = MTF index of the t-th literal
                const u32 rank = (u32)__builtin_popcountll(mL & lt_mask);
                const u32 nlit = (u32)__builtin_popcountll(mL);
                const bool far = __ballot(isL && idxv >= 64u) != 0ull;
                const u32 f0 = (u32)__builtin_amdgcn_readlane((int)l0, 0);       // front of the list before the row's first literal
                if (isL) s_cidx[rank] = idxv;
                __builtin_amdgcn_wave_barrier();
                const u32 cidx = s_cidx[lane];
                __builtin_amdgcn_wave_barrier();
                u32 outc = 0;                                                  // lane t = byte of the t-th literal
                if (!far) {
                    // mtf(mtfSymbol, sym - 1) :53-60 for an index below 64: one v_readlane, one DPP shift, one select; straight-line
                    // steps, four per loop branch
#define K7_MTF_STEP(t_) do {                                                                                       \
                        const u32 idx_ = (u32)__builtin_amdgcn_readlane((int)cidx, (int)(t_));                     \
                        const u32 src_ = (u32)__builtin_amdgcn_readlane((int)l0, (int)idx_);                       \
                        const u32 sh_ = (u32)__builtin_amdgcn_update_dpp((int)src_, (int)l0, 0x138, 0xf, 0xf, false); \
                        l0 = lane <= idx_ ? sh_ : l0;                                                              \
                        outc = (u32)cjs_writelane((int)src_, (int)(t_), (int)outc);                                \
                    } while (0)
                    u32 t = 0;
                    for (; t + 4u <= nlit; t += 4u) { K7_MTF_STEP(t); K7_MTF_STEP(t + 1u); K7_MTF_STEP(t + 2u); K7_MTF_STEP(t + 3u); }
                    if (nlit & 2u) { K7_MTF_STEP(t); K7_MTF_STEP(t + 1u); t += 2u; }
                    if (nlit & 1u) K7_MTF_STEP(t);
                } else {
                    for (u32 t = 0; t < nlit; t++) {
                        const u32 idx = (u32)__builtin_amdgcn_readlane((int)cidx, (int)t);
                        const u32 src = mtf_step(l0, l1, l2, l3, idx, lane);
                        outc = (u32)cjs_writelane((int)src, (int)t, (int)outc);
                    }
                }
                // back to the row's lanes: own byte for a literal, the byte of the literal before for a run
                const u32 nb = (u32)__builtin_popcountll(mL & lt_mask);        // literals in front of this lane
                const u32 own = __shfl(outc, (int)(rank & 63u));
                const u32 prevb = __shfl(outc, (int)((nb - 1u) & 63u));
                const u32 rb = nb ? prevb : f0;
                const u32 tr = fl && ra.T ? (flc | (rb << 24)) : 0u;            // flc < 2^24
                const u32 tl = isL ? (own | 0x100u) : 0u;
                while (trow - lds_observe(&s_ttail) >= K7_TROWS && !lds_observe(&s_abort)) __builtin_amdgcn_s_sleep(1);
                s_trun[(trow & (K7_TROWS - 1u)) * 64u + lane] = tr;
                s_tlit[(trow & (K7_TROWS - 1u)) * 64u + lane] = tl;
                __builtin_amdgcn_wave_barrier();
                trow++;
                if (lane == 0) lds_publish(&s_thead, trow);
                if (mE) { finished = true; break; }
                // the run symbols behind the row's last literal carry over
                {
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
                consumed = b0 + hi;
                if (lane == 0) lds_publish(&s_tail, consumed);
                if (lds_observe(&s_abort)) break;
            }
            if (!st && !finished) st = -1;                                     // wave 0 reports why it stopped
            if (lane == 0) { s_cstat = st; s_cnt = cnt; s_cwait = cwait; if (st) lds_publish(&s_abort, 1u); lds_publish(&s_tdone, 1u); }
        } else {
            // ---- wave 2: token rows -> bytes of the last column ------------------------------------------
            u8* out = D.tt + (size_t)slot * D.ttStride;
            u32 taken = 0, opos = 0;
            for (;;) {
                u32 head = lds_observe(&s_thead);
                if (head == taken) {
                    if (lds_observe(&s_tdone)) {
                        head = lds_observe(&s_thead);
                        if (head == taken) break;
                    } else { __builtin_amdgcn_s_sleep(2); continue; }
                }
                const u32 tr = s_trun[(taken & (K7_TROWS - 1u)) * 64u + lane], tl = s_tlit[(taken & (K7_TROWS - 1u)) * 64u + lane];
                const u32 c = tr & 0xffffffu, rb = tr >> 24, f = (tl >> 8) & 1u;
                const u32 tot = c + f;
                const u32 incl = wave_incl_scan_u32(tot);
                const u32 at = opos + incl - tot;
                if (f) out[at + c] = (u8)tl;                                   // the common case: one coalesced store per row
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
                taken++;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) lds_publish(&s_ttail, taken);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        DecResult res;
        int st = s_pstat;
        if (st == 0 && s_hdr) st = s_cstat == -1 ? DEC_DATA_ERROR : s_cstat;
        if (st == 0 && s_origPtr >= s_cnt) st = DEC_DATA_ERROR;               // :372
        res.endbit = s_endbit;
        res.status = st;
        res.n = s_cnt;
        res.origPtr = s_origPtr;
        res.crc = s_crc;
        res.cycles = clock64() - t_start;
        res.symbols = s_nsym;
        res.pwait = s_pwait;
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
    hipLaunchKernelGGL(k7_decode, dim3(count), dim3(192), 0, stream, D, first, count);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
