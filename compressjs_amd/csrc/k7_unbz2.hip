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

#define K7_RING 4096u        // u16 records in flight between waves 0 and 1
#define K7_TRING 1024u       // u32 tokens in flight between waves 1 and 2

// workgroup-scope publish / observe of a flag in LDS
__device__ __forceinline__ void lds_publish(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ u32 lds_observe(u32* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Three waves per block, a software pipeline through two LDS rings:
//   wave 0  parses the block header, then only finds code boundaries: one (table, canonical index)
//           record per Huffman symbol (lib/Bzip2.js:283-300);
//   wave 1  turns records into symbols (permute[] gather, 64 at a time), undoes RLE2 and MTF and emits
//           one (byte, count) token per literal or run (:305-366);
//   wave 2  expands 64 tokens per step into the block's last column (prefix sum of the counts, one
//           coalesced byte store per row in the common all-literal case).
// The two recurrences are serial; splitting them shortens the dependent instruction chain per symbol.
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
    __shared__ u16 s_ring[K7_RING];
    __shared__ u32 s_mtf[64];          // initial MTF list (= symToByte), 4 entries per lane
    __shared__ u32 s_head, s_tail, s_done, s_abort, s_symTotal, s_hdr;
    __shared__ u32 s_tring[K7_TRING];  // (byte | count << 8) tokens between waves 1 and 2
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
    if (wave == 0) {
        const u64 start = D.cand[first + slot] >> 1;
        br_init(r, D.in32, D.zeroChunk, start + 48);
        int st = 0;
        const u32 crc = br_get(r, 32);
        u32 origPtr = 0, mw = 0;
        int symTotal = 0, groupCount = 0, symCount = 0;
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
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)s_hdr)) {
        if (wave == 0) {
            // ---- wave 0: code boundaries only --------------------------------------------------------
            int limLA = -1, st = 0;
            u32 base = 0, g = 0, jeob = 0, recv = 0;
            u32 selrow = 0, selnext = gsel[lane];
            int left = 0;                 // symbols left in the current group of 50
            u32 selector = 0, np = 0;
            u64 nsym = 0, pwait = 0;
            bool eob = false;
            for (;;) {
                if (left == 0) {
                    left = 50;
                    if (selector >= nSel) { st = DEC_DATA_ERROR; break; }
                    // selectors live in HBM (16 KB per block would cost LDS residency): a register row of 64 words =
                    // 512 selectors, the next row requested one row ahead like the stream words
                    if ((selector & 511u) == 0) { selrow = selnext; selnext = gsel[(((selector >> 9) + 1u) << 6) + lane]; }
                    g = ((u32)__builtin_amdgcn_readlane((int)selrow, (int)((selector >> 3) & 63u)) >> (4u * (selector & 7u))) & 15u;
                    selector++;
                    limLA = lane < 32u ? s_limLA[g][lane] : -1;
                    base = s_base[g][lane & 31u];
                    jeob = (u32)__builtin_amdgcn_readfirstlane((int)s_jeob[g]);
                }
                left--;
                nsym++;
                const u32 v20 = (u32)(r.win >> 44);
                const u64 m = __ballot((int)v20 <= limLA);
                if (m == 0) { st = DEC_DATA_ERROR; break; }                    // i > maxLen (:292)
                const int len = __builtin_ctzll(m);
                br_consume(r, len);
                const u32 j = (v20 >> (20 - len)) - (u32)__builtin_amdgcn_readlane((int)base, len);
                if (j >= 258u) { st = DEC_DATA_ERROR; break; }                 // :299-300 (base <= 2^28: no wrap)
                recv = lane == (np & 63u) ? (j | (g << 9)) : recv;
                np++;
                eob = j == jeob;
                if (eob || !(np & 63u)) {
                    const u32 b0 = (np - 1u) & ~63u;                           // first record of this batch
                    if (b0 + 64u - lds_observe(&s_tail) > K7_RING) {
                        const u64 w0 = clock64();
                        while (b0 + 64u - lds_observe(&s_tail) > K7_RING && !lds_observe(&s_abort)) __builtin_amdgcn_s_sleep(2);
                        pwait += clock64() - w0;
                    }
                    if (b0 + lane < np) s_ring[(b0 + lane) & (K7_RING - 1u)] = (u16)recv;
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) lds_publish(&s_head, np);
                    if (eob || lds_observe(&s_abort)) break;
                }
            }
            if (lane == 0) { s_pstat = st; s_endbit = br_tell(r); s_nsym = nsym; s_pwait = pwait; lds_publish(&s_done, 1u); }
        } else if (wave == 1) {
            // ---- wave 1: records -> symbols -> RLE2 -> MTF -> (byte, count) tokens ---------------------
            const u32 symTotal = (u32)__builtin_amdgcn_readfirstlane((int)s_symTotal);
            u32 mw = s_mtf[lane], cnt = 0, tokv = 0, nt = 0;
            const bool is0 = lane == 0;
            int st = 0;
            u32 runN = 0;                  // RUNA/RUNB symbols since the run (re)started: the reference's runPos == 1 << runN, 0 = no run pending
            long long runT = 0;
            u32 consumed = 0;
            u64 cwait = 0;
            bool finished = false;
            // one token per literal or run; 64 tokens are gathered in a register row and handed over at once
#define K7_TOKEN(byte_, count_) do {                                                              \
                tokv = lane == (nt & 63u) ? ((u32)(byte_) | ((u32)(count_) << 8)) : tokv;                 \
                nt++;                                                                                     \
                if (!(nt & 63u)) {                                                                        \
                    while (nt - lds_observe(&s_ttail) > K7_TRING && !lds_observe(&s_abort)) __builtin_amdgcn_s_sleep(1); \
                    s_tring[(nt - 64u + lane) & (K7_TRING - 1u)] = tokv;                                  \
                    __builtin_amdgcn_wave_barrier();                                                      \
                    if (lane == 0) lds_publish(&s_thead, nt);                                             \
                }                                                                                         \
            } while (0)
            while (!finished) {
                u32 head = lds_observe(&s_head);
                if (head == consumed) {
                    if (lds_observe(&s_done)) {
                        head = lds_observe(&s_head);
                        if (head == consumed) break;                           // wave 0 stopped without end-of-block
                    } else { const u64 w0 = clock64(); __builtin_amdgcn_s_sleep(2); cwait += clock64() - w0; continue; }
                }
                const u32 b0 = consumed & ~63u;
                const u32 hi = head - b0 < 64u ? head - b0 : 64u;              // records [consumed, b0+hi) are ready
                u32 symv = 0;
                if (lane < hi) {
                    const u32 rec = s_ring[(b0 + lane) & (K7_RING - 1u)];
                    symv = s_perm[rec >> 9][rec & 511u];
                }
                // RUNA/RUNB symbols never enter the serial loop: a maximal stretch of them is a bijective
                // base-2 number (:318-335) whose value falls out of two ballots; the loop below only visits
                // literals and the end-of-block symbol
                const u32 lo = consumed - b0;
                const u64 inb = (hi == 64u ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
                const u64 mA = __ballot(symv == 0u) & inb, mB = __ballot(symv == 1u) & inb;
                u64 lit = inb & ~(mA | mB);
                u32 cur = lo;
                for (;;) {
                    const u32 k = lit ? (u32)__builtin_ctzll(lit) : hi;       // next literal, or the end of the row
                    if (k > cur) {                                             // run symbols [cur, k) of the row
                        // The reference's int32 runPos (:314-337) reaches 0 after 32 run symbols; the next one finds
                        // !runPos and restarts with t = 0, and a literal that follows a multiple of 32 run symbols finds
                        // runPos == 0 and flushes nothing.  runN = run symbols since the last (re)start, 0..31.
                        const u32 len = k - cur;
                        const u64 seg = len == 64u ? ~0ull : ((1ull << len) - 1ull);
                        const u64 sa = (mA >> cur) & seg, sb = (mB >> cur) & seg;
                        if (runN + len < 32u) {
                            runT += (long long)(sa << runN) + 2ll * (long long)(sb << runN);
                            runN += len;
                        } else {
                            const u32 rp = (runN + len) & 31u, skip = len - rp;   // only the symbols after the last restart count
                            runT = rp ? (long long)(sa >> skip) + 2ll * (long long)(sb >> skip) : 0ll;
                            runN = rp;
                        }
                    }
                    if (!lit) break;
                    lit &= lit - 1;
                    cur = k + 1u;
                    const u32 sym = (u32)__builtin_amdgcn_readlane((int)symv, (int)k);
                    if (runN) {                                                // :340-347
                        runN = 0;
                        if ((long long)cnt + runT > (long long)DEC_CAP) { st = DEC_DATA_ERROR; break; }
                        if (runT) {
                            const u32 uc = (u32)__builtin_amdgcn_readlane((int)mw, 0) & 0xffu;
                            K7_TOKEN(uc, (u32)runT);
                            cnt += (u32)runT;
                        }
                        runT = 0;
                    }
                    if (sym > symTotal) { finished = true; break; }            // :349-350
                    if (cnt >= DEC_CAP) { st = DEC_DATA_ERROR; break; }
                    {                                                          // mtf(mtfSymbol, sym - 1) :53-60
                        const u32 idx = sym - 1u, ql = idx >> 2, sh8 = 8u * (idx & 3u);
                        const u32 src = ((u32)__builtin_amdgcn_readlane((int)mw, (int)ql) >> sh8) & 0xffu;
                        const u32 lowmask = (1u << sh8) - 1u, keepmask = (~lowmask) << 8;
                        if (ql == 0) {
                            const u32 nv = (mw & keepmask) | ((mw & lowmask) << 8) | src;
                            mw = is0 ? nv : mw;
                        } else {
                            const u32 up = (u32)__builtin_amdgcn_update_dpp(0, (int)mw, 0x138, 0xf, 0xf, false);   // wave_shr:1
                            const u32 carry = is0 ? src : (up >> 24);
                            const u32 full = (mw << 8) | carry;
                            const u32 part = (mw & keepmask) | ((mw & lowmask) << 8) | carry;
                            mw = lane < ql ? full : (lane == ql ? part : mw);
                        }
                        K7_TOKEN(src, 1u);
                        cnt++;
                    }
                }
                if (st) { if (lane == 0) lds_publish(&s_abort, 1u); break; }
                consumed = b0 + hi;
                if (lane == 0) lds_publish(&s_tail, consumed);
            }
            if (!st && !finished) st = -1;                                     // wave 0 reports why it stopped
            if (st == 0 && (nt & 63u)) {                                       // the last, partial row of tokens
                const u32 t0 = nt & ~63u;
                while (t0 + 64u - lds_observe(&s_ttail) > K7_TRING && !lds_observe(&s_abort)) __builtin_amdgcn_s_sleep(1);
                if (t0 + lane < nt) s_tring[(t0 + lane) & (K7_TRING - 1u)] = tokv;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) lds_publish(&s_thead, nt);
            }
            if (lane == 0) { s_cstat = st; s_cnt = cnt; s_cwait = cwait; if (st) lds_publish(&s_abort, 1u); lds_publish(&s_tdone, 1u); }
        } else {
            // ---- wave 2: tokens -> bytes of the last column, 64 tokens per step -------------------------
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
                const u32 hi = head - taken < 64u ? head - taken : 64u;        // rows are 64-aligned: taken % 64 == 0
                const u32 tok = lane < hi ? s_tring[(taken + lane) & (K7_TRING - 1u)] : 0u;
                const u32 c = tok >> 8, byte = tok & 0xffu;
                const u32 incl = wave_incl_scan_u32(c);
                const u32 at = opos + incl - c;
                if (c == 1u) out[at] = (u8)byte;                               // the common case: one coalesced store per row
                else if (c && c < 64u) for (u32 i = 0; i < c; i++) out[at + i] = (u8)byte;
                u64 big = __ballot(c >= 64u);                                  // long runs: the whole wave fills each of them
                while (big) {
                    const int l = __builtin_ctzll(big);
                    big &= big - 1;
                    const u32 bc = (u32)__builtin_amdgcn_readlane((int)c, l), bat = (u32)__builtin_amdgcn_readlane((int)at, l);
                    const u32 bb = (u32)__builtin_amdgcn_readlane((int)byte, l);
                    for (u32 i = lane; i < bc; i += 64u) out[bat + i] = (u8)bb;
                }
                opos += (u32)__builtin_amdgcn_readlane((int)incl, 63);
                taken += hi;
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
