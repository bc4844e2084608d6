// Host side of the GPU bzip2 decoder: Bunzip.decode / decodeBlock / table (lib/Bzip2.js:454-548).
//
// The reference is strictly sequential: block k+1's header is read where block k's last symbol
// ended.  Here every occurrence of the block magic is a *candidate* (k7_scan_magic), candidates are
// entropy-decoded in parallel batches (k7_decode), and this file then walks the real chain through
// the results exactly in the reference's order - stream header, block, block, ..., end-of-stream
// magic, stream CRC, next stream if `multistream` - so that the outcome (bytes, or which error is
// raised first) is the reference's.  Blocks on the chain go through K8 (inverse BWT) and K9
// (un-RLE1, CRC) in the same batches.  There is no CPU decode path.
#include "decode.h"
#include "decode_host.h"
#include <algorithm>
#include <vector>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#define WHOLEPI 0x314159265359ull
#define SQRTPI 0x177245385090ull

struct DecState {
    u32 slots;
    u8* d_in; size_t in_cap;
    u64* d_cand; u32 cand_cap; u32* d_ncand;
    void* slab;                      // per-slot arrays
    DecBuf D;
    u64* d_bcand; u32* d_slotOf; u64* d_outOff; u32* d_crcOut;
    u8* d_out; size_t out_cap; u64 out_size;
    int detail; u32 crc_got, crc_want;
    std::vector<u64> tab_pos, tab_size;
};

#define TRYH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return CJS_E_HIP - (int)e_; } while (0)

static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

static int dec_alloc(DecState& S, u32 slots) {
    if (S.slab) { (void)hipFree(S.slab); S.slab = nullptr; }
    S.slots = 0;                                  // nothing usable until the new slab exists
    const size_t n = slots;
    size_t tot = 0;
    const size_t o_tt = tot;        tot += al(n * DEC_STRIDE);
    const size_t o_res = tot;       tot += al(n * sizeof(DecResult));
    const size_t o_sel = tot;       tot += al(n * 4160 * 4);
    const size_t o_word = tot;      tot += al(n * DEC_STRIDE * 4);
    const size_t o_hist = tot;      tot += al(n * DEC_TILES * 256 * 4);
    const size_t o_succ = tot;      tot += al(n * DEC_MAXSPL * 4);
    const size_t o_len = tot;       tot += al(n * DEC_MAXSPL * 4);
    const size_t o_off = tot;       tot += al(n * DEC_MAXSPL * 4);
    const size_t o_flags = tot;     tot += al(n * 4);
    const size_t o_pre = tot;       tot += al(n * DEC_STRIDE);
    const size_t o_fn = tot;        tot += al(n * DEC_TILES * 4);
    const size_t o_state = tot;     tot += al(n * DEC_TILES);
    const size_t o_isc = tot;       tot += al(n * (DEC_STRIDE / 32) * 4);
    const size_t o_tlen = tot;      tot += al(n * DEC_TILES * 4);
    const size_t o_bout = tot;      tot += al(n * 4);
    const size_t o_bcand = tot;     tot += al(n * 8);
    const size_t o_slotof = tot;    tot += al(n * 4);
    const size_t o_outoff = tot;    tot += al(n * 8);
    const size_t o_crc = tot;       tot += al(n * 4);
    TRYH(hipMalloc(&S.slab, tot));
    S.slots = slots;
    u8* b = (u8*)S.slab;
    { const u32* keep_in = S.D.in32; const u64 keep_zc = S.D.zeroChunk; memset(&S.D, 0, sizeof S.D); S.D.in32 = keep_in; S.D.zeroChunk = keep_zc; }
    S.D.tt = b + o_tt; S.D.ttStride = DEC_STRIDE;
    S.D.res = (DecResult*)(b + o_res);
    S.D.sel = (u32*)(b + o_sel);
    S.D.word = (u32*)(b + o_word);
    S.D.tileHist = (u32*)(b + o_hist);
    S.D.splSucc = (u32*)(b + o_succ); S.D.splLen = (u32*)(b + o_len); S.D.splOff = (u32*)(b + o_off);
    S.D.flags = (u32*)(b + o_flags);
    S.D.pre = b + o_pre;
    S.D.tileFn = (u32*)(b + o_fn); S.D.tileState = b + o_state;
    S.D.isCount = (u32*)(b + o_isc); S.D.tileLen = (u32*)(b + o_tlen); S.D.blkOut = (u32*)(b + o_bout);
    S.d_bcand = (u64*)(b + o_bcand); S.d_slotOf = (u32*)(b + o_slotof);
    S.d_outOff = (u64*)(b + o_outoff); S.d_crcOut = (u32*)(b + o_crc);
    S.D.cand = S.d_bcand; S.D.slotOf = S.d_slotOf; S.D.outOff = S.d_outOff; S.D.crcOut = S.d_crcOut;
    if (!S.d_ncand) TRYH(hipMalloc((void**)&S.d_ncand, 256));
    return CJS_OK;
}

// slots for the blocks of this stream: as many as it has candidates, within [min_slots, DEC_MAX_SLOTS].
// One k7_decode wave set per block is latency-bound, so throughput comes from the number of blocks in flight
// (4 per CU fit); 5.7 MB of HBM per slot.
#define DEC_MAX_SLOTS 4096u
// The slab never shrinks while the context lives, so its size is capped: CJS_DEC_MAX_SLOTS (env), else what fits
// a quarter of the HBM that is free right now (a 400 MB level-1 stream, or a small stream stuffed with fake
// block magics, has thousands of candidates: 4096 slots would pin 23 GB).  Fewer slots only mean more batches.
static u32 dec_slot_limit(u32 min_slots) {
    const u32 env_cap = []() -> u32 { const char* e = getenv("CJS_DEC_MAX_SLOTS"); return e ? (u32)strtoul(e, nullptr, 10) : 0u; }();   // (read per call)
    u32 lim = DEC_MAX_SLOTS;
    if (env_cap) lim = env_cap;
    else {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr) {
            const size_t per = (size_t)DEC_STRIDE * 6 + 4160 * 4 + DEC_TILES * 1040 + DEC_MAXSPL * 12 + 4096;
            const size_t fit = fr / 4 / per;
            if (fit < lim) lim = (u32)fit;
        }
    }
    return lim < min_slots ? min_slots : lim;
}
static int dec_fit_slots(DecState& S, u32 min_slots, size_t ncand) {
    u32 want = (u32)std::min<size_t>(dec_slot_limit(min_slots), ncand);
    if (want < min_slots) want = min_slots;
    if (want <= S.slots) return CJS_OK;
    want = (want + 63u) & ~63u;
    for (;;) {                                   // out of memory: halve, down to the minimum the caller can work with
        const int rc = dec_alloc(S, want);
        if (rc == CJS_OK || want <= ((min_slots + 63u) & ~63u)) return rc;
        want = ((want / 2 > min_slots ? want / 2 : min_slots) + 63u) & ~63u;
    }
}

void dec_free(DecState* S) {
    if (!S) return;
    (void)hipFree(S->d_in); (void)hipFree(S->d_cand); (void)hipFree(S->d_ncand); (void)hipFree(S->slab); (void)hipFree(S->d_out);
    delete S;
}

static int dec_get(DecState** ps, u32 slots) {
    if (*ps) return CJS_OK;
    DecState* S = new DecState();
    S->slots = 0; S->d_in = nullptr; S->in_cap = 0; S->d_cand = nullptr; S->cand_cap = 0; S->d_ncand = nullptr;
    S->slab = nullptr; S->d_out = nullptr; S->out_cap = 0; S->out_size = 0; S->detail = 0; S->crc_got = S->crc_want = 0;
    const int rc = dec_alloc(*S, slots);
    if (rc) { dec_free(S); return rc; }
    *ps = S;
    return CJS_OK;
}

static int ensure_out(DecState& S, u64 need, hipStream_t st) {
    if (need <= S.out_cap) return CJS_OK;
    size_t cap = S.out_cap ? S.out_cap * 2 : (size_t)64 << 20;
    if (cap < need) cap = need;
    cap = (cap + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
    u8* p = nullptr;
    TRYH(hipMalloc((void**)&p, cap));
    if (S.out_size) {
        hipError_t e = hipMemcpyAsync(p, S.d_out, S.out_size, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { (void)hipFree(p); return CJS_E_HIP - (int)e; }
    }
    (void)hipFree(S.d_out);
    S.d_out = p; S.out_cap = cap;
    return CJS_OK;
}

// the stream, padded with zeros (bits past EOF read as 0, lib/BitStream.js:84), resident in HBM
static int stage_input(DecState& S, const u8* in, u64 len, bool in_dev, hipStream_t st) {
    const size_t need = ((len + 255) & ~(size_t)255) + 1024;
    if (need > S.in_cap) {
        (void)hipFree(S.d_in); S.d_in = nullptr; S.in_cap = 0;
        TRYH(hipMalloc((void**)&S.d_in, need));
        S.in_cap = need;
    }
    TRYH(hipMemsetAsync(S.d_in + (len & ~(u64)255), 0, need - (len & ~(u64)255), st));
    if (len) TRYH(hipMemcpyAsync(S.d_in, in, len, in_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    const u32 ccap = (u32)std::min<u64>(len / 4 + 16, 1u << 27);      // a 48-bit pattern cannot occur more often than every 6 bytes
    if (ccap > S.cand_cap) {
        (void)hipFree(S.d_cand); S.d_cand = nullptr; S.cand_cap = 0;
        TRYH(hipMalloc((void**)&S.d_cand, (size_t)ccap * 8));
        S.cand_cap = ccap;
    }
    S.D.in32 = (const u32*)S.d_in;
    S.D.zeroChunk = ((len + 255) >> 8) + 1;          // [zeroChunk*256, +256) is inside the 1024 zero bytes
    return CJS_OK;
}

struct Walker {
    DecState& S;
    hipStream_t st;
    const u8* host_in;     // non-null when the caller's buffer is host memory
    u64 len;
    int rd_err;
    // up to 64 bits at absolute bit position p, zero extended
    u64 rd(u64 p, int n) {
        u8 b[16];
        memset(b, 0, sizeof b);
        const u64 byte = p >> 3;
        if (byte < len) {
            const size_t k = (size_t)std::min<u64>(16, len - byte);
            if (host_in) memcpy(b, host_in + byte, k);
            else {
                hipError_t e = hipMemcpyAsync(b, S.d_in + byte, k, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e != hipSuccess) rd_err = CJS_E_HIP - (int)e;
            }
        }
        u64 v = 0;
        for (int i = 0; i < n; i++) {
            const u64 q = (p & 7u) + (u64)i;
            v = (v << 1) | ((b[q >> 3] >> (7 - (q & 7u))) & 1u);
        }
        return v;
    }
    bool eof(u64 p) const { return ((p + 7) >> 3) >= len; }          // lib/Util.js:30 on the coerced buffer stream
};

static int fail(DecState& S, int code, int detail, u32 got = 0, u32 want = 0) {
    S.detail = detail; S.crc_got = got; S.crc_want = want;
    return code;
}

// reads the 4-byte stream header at byte `base` (lib/Bzip2.js:137-152); 0 or an error
static int read_header(DecState& S, Walker& W, u64 base, u32* dbufSize) {
    if (W.len < base + 4) return fail(S, DEC_NOT_BZIP, DEC_DETAIL_BAD_MAGIC);
    const u64 h = W.rd(base * 8, 32);
    if ((h >> 8) != 0x425a68u) return fail(S, DEC_NOT_BZIP, DEC_DETAIL_BAD_MAGIC);
    const int level = (int)(h & 0xffu) - '0';
    if (level < 1 || level > 9) return fail(S, DEC_NOT_BZIP, DEC_DETAIL_LEVEL);
    *dbufSize = 100000u * (u32)level;
    return 0;
}

// the reference's per-block checks, in its order, on a k7_decode result
static int block_error(DecState& S, const DecResult& r, u32 dbufSize) {
    if (r.status == DEC_OBSOLETE) return fail(S, DEC_OBSOLETE, DEC_DETAIL_NONE);                // :174-175
    if (r.origPtr > dbufSize) return fail(S, DEC_DATA_ERROR, DEC_DETAIL_ORIGPTR);                // :177-178
    if (r.status) return fail(S, r.status, DEC_DETAIL_NONE);
    if (r.n > dbufSize) return fail(S, DEC_DATA_ERROR, DEC_DETAIL_NONE);                         // :342,:360 with this stream's dbufSize
    return 0;
}

struct ValidBlk { u32 slot; u64 pos; u32 crc; };

// K8 + K9 of the chain blocks of one batch; appends to S.d_out.  Returns 0 or an error.
static int process_valid(DecState& S, hipStream_t st, std::vector<ValidBlk>& valid, const std::vector<DecResult>& res) {
    const u32 nv = (u32)valid.size();
    if (!nv) return 0;
    std::vector<u32> slotOf(nv);
    for (u32 k = 0; k < nv; k++) slotOf[k] = valid[k].slot;
    TRYH(hipMemcpyAsync(S.d_slotOf, slotOf.data(), nv * 4, hipMemcpyHostToDevice, st));
    int rc = k8_run(S.D, nv, st);
    if (rc) return rc;
    rc = k9_sizes(S.D, nv, st);
    if (rc) return rc;
    std::vector<u32> blkOut(S.slots);
    TRYH(hipMemcpyAsync(blkOut.data(), S.D.blkOut, S.slots * 4, hipMemcpyDeviceToHost, st));
    TRYH(hipStreamSynchronize(st));
    std::vector<u64> outOff(nv);
    u64 o = S.out_size;
    for (u32 k = 0; k < nv; k++) { outOff[k] = o; o += blkOut[valid[k].slot]; }
    rc = ensure_out(S, o + 64, st);
    if (rc) return rc;
    S.D.out = S.d_out;
    TRYH(hipMemcpyAsync(S.d_outOff, outOff.data(), nv * 8, hipMemcpyHostToDevice, st));
    rc = k9_expand(S.D, nv, st);
    if (rc) return rc;
    std::vector<u32> crc(nv);
    TRYH(hipMemcpyAsync(crc.data(), S.d_crcOut, nv * 4, hipMemcpyDeviceToHost, st));
    TRYH(hipStreamSynchronize(st));
    for (u32 k = 0; k < nv; k++) {
        // the reference has written the block before it compares the CRC (:437-445)
        S.out_size = outOff[k] + blkOut[valid[k].slot];
        if (crc[k] != valid[k].crc) return fail(S, DEC_DATA_ERROR, DEC_DETAIL_BLOCK_CRC, crc[k], valid[k].crc);
        S.tab_pos.push_back(valid[k].pos);
        S.tab_size.push_back(blkOut[valid[k].slot]);
    }
    (void)res;
    valid.clear();
    return 0;
}

static int decode_batch(DecState& S, hipStream_t st, const u64* pos, u32 count, std::vector<DecResult>& res) {
    std::vector<u64> enc(count);
    for (u32 i = 0; i < count; i++) enc[i] = pos[i] << 1;
    TRYH(hipMemcpyAsync(S.d_bcand, enc.data(), (size_t)count * 8, hipMemcpyHostToDevice, st));
    const int rc = k7_run(S.D, 0, count, st);
    if (rc) return rc;
    res.resize(count);
    TRYH(hipMemcpyAsync(res.data(), S.D.res, (size_t)count * sizeof(DecResult), hipMemcpyDeviceToHost, st));
    TRYH(hipStreamSynchronize(st));
    if (getenv("CJS_DEC_TRACE")) {
        u64 cy = 0, sy = 0, by = 0, pw = 0, cw = 0;
        for (u32 i = 0; i < count; i++) { cy += res[i].cycles; sy += res[i].symbols; by += res[i].n; pw += res[i].pwait; cw += res[i].cwait; }
        fprintf(stderr, "[k7] %u blocks: %.1f Mcycles/block, %.0f symbols/block, %.0f bytes/block, %.1f cycles/symbol (boundary wave waits %.1f, symbol wave waits %.1f)\n",
                count, cy / 1e6 / count, (double)sy / count, (double)by / count, sy ? (double)cy / sy : 0.0,
                sy ? (double)pw / sy : 0.0, sy ? (double)cw / sy : 0.0);
        u64 pf[14] = {0};
        for (u32 i = 0; i < count; i++) for (int k = 0; k < 14; k++) pf[k] += res[i].prof[k];
        if (pf[2] && sy)                                           // a -DK7_PROF build
            fprintf(stderr, "[k7] clocks/symbol  A: group set-up %.1f, row fill + row end %.1f, walk %.1f, group end + hand-over %.1f | B: wait %.1f, symbols %.1f, "
                    "rows %.1f | C: MTF %.1f, wait %.1f | D: expand %.1f, wait %.1f\n", (double)pf[0] / sy, (double)pf[1] / sy, (double)pf[2] / sy, (double)pf[3] / sy,
                    (double)pf[5] / sy, (double)pf[6] / sy, (double)pf[7] / sy, (double)pf[10] / sy, (double)pf[11] / sy, (double)pf[12] / sy, (double)pf[13] / sy);
    }
    return 0;
}

int64_t dec_stream(DecState** ps, u32 slots, hipStream_t st, const u8* in, u64 len, bool in_dev, int multistream,
                   bool check_stream_crc) {
    int rc = dec_get(ps, slots);
    if (rc) return rc;
    DecState& S = **ps;
    S.out_size = 0; S.detail = 0; S.crc_got = S.crc_want = 0;
    S.tab_pos.clear(); S.tab_size.clear();
    Walker W = {S, st, in_dev ? nullptr : in, len, 0};
    rc = stage_input(S, in, len, in_dev, st);
    if (rc) return rc;
    u32 dbufSize = 0;
    rc = read_header(S, W, 0, &dbufSize);                                        // :137-152, before anything else
    if (rc) return rc;
    // candidates
    rc = k7_scan(S.d_in, len, 32, S.d_cand, S.d_ncand, S.cand_cap, st);
    if (rc) return rc;
    u32 nc = 0;
    TRYH(hipMemcpyAsync(&nc, S.d_ncand, 4, hipMemcpyDeviceToHost, st));
    TRYH(hipStreamSynchronize(st));
    if (nc > S.cand_cap) return CJS_E_UNSUPPORTED;                               // only when the 2^27 cap on candidates is hit
    std::vector<u64> cand(nc);
    if (nc) TRYH(hipMemcpy(cand.data(), S.d_cand, (size_t)nc * 8, hipMemcpyDeviceToHost));
    std::sort(cand.begin(), cand.end());
    std::vector<u64> bpos;                                                       // block-magic positions only
    for (u64 c : cand) if (!(c & 1u)) bpos.push_back(c >> 1);
    rc = dec_fit_slots(S, slots, bpos.size());
    if (rc) return rc;

    std::vector<DecResult> res;
    std::vector<ValidBlk> valid;
    u32 bfirst = 0, bcount = 0;
    u64 p = 32;
    u32 streamCRC = 0;
    for (;;) {
        int term = 0;          // 0: stream finished, <0: error, 1: need a batch starting at `need`
        u32 need = 0;
        for (;;) {
            if (W.eof(p)) break;                                                 // :462
            const u64 key = p << 1;
            auto it = std::lower_bound(cand.begin(), cand.end(), key);
            if (it == cand.end() || (*it >> 1) != p) { term = fail(S, DEC_NOT_BZIP, DEC_DETAIL_NONE); break; }     // :160-161
            if (*it & 1u) {                                                      // end of stream :157-159,:465-477
                const u32 target = (u32)W.rd(p + 48, 32);
                if (check_stream_crc && target != streamCRC) { term = fail(S, DEC_DATA_ERROR, DEC_DETAIL_STREAM_CRC, streamCRC, target); break; }
                if (multistream && !W.eof(p + 80)) {
                    const u64 base = (p + 80 + 7) >> 3;
                    term = read_header(S, W, base, &dbufSize);
                    if (term) break;
                    p = (base + 4) * 8;
                    streamCRC = 0;
                    continue;
                }
                break;
            }
            const u32 idx = (u32)(std::lower_bound(bpos.begin(), bpos.end(), p) - bpos.begin());
            if (idx < bfirst || idx >= bfirst + bcount) { term = 1; need = idx; break; }
            const DecResult& r = res[idx - bfirst];
            streamCRC = r.crc ^ ((streamCRC << 1) | (streamCRC >> 31));          // :163-164
            term = block_error(S, r, dbufSize);
            if (term) break;
            valid.push_back({idx - bfirst, p, r.crc});
            p = r.endbit;
        }
        if (W.rd_err) return W.rd_err;
        const int saved_detail = S.detail; const u32 sg = S.crc_got, sw = S.crc_want;
        rc = process_valid(S, st, valid, res);                                   // earlier blocks' CRC errors come first
        if (rc) return rc;
        if (term < 0) { S.detail = saved_detail; S.crc_got = sg; S.crc_want = sw; return term; }
        if (term == 0) break;
        bfirst = need;
        bcount = (u32)std::min<size_t>(S.slots, bpos.size() - need);
        rc = decode_batch(S, st, bpos.data() + bfirst, bcount, res);
        if (rc) return rc;
    }
    return (int64_t)S.out_size;
}

// Bunzip.decodeBlock (lib/Bzip2.js:482-503)
int64_t dec_block(DecState** ps, u32 slots, hipStream_t st, const u8* in, u64 len, u64 bitpos) {
    int rc = dec_get(ps, slots);
    if (rc) return rc;
    DecState& S = **ps;
    S.out_size = 0; S.detail = 0; S.crc_got = S.crc_want = 0;
    S.tab_pos.clear(); S.tab_size.clear();
    Walker W = {S, st, in, len, 0};
    u32 dbufSize = 0;
    rc = read_header(S, W, 0, &dbufSize);
    if (rc) return rc;
    const u64 h = W.rd(bitpos, 48);
    if (h == SQRTPI) return 0;
    if (h != WHOLEPI) return fail(S, DEC_NOT_BZIP, DEC_DETAIL_NONE);
    rc = stage_input(S, in, len, false, st);
    if (rc) return rc;
    std::vector<DecResult> res;
    rc = decode_batch(S, st, &bitpos, 1, res);
    if (rc) return rc;
    rc = block_error(S, res[0], dbufSize);
    if (rc) return rc;
    std::vector<ValidBlk> valid;
    valid.push_back({0, bitpos, res[0].crc});
    rc = process_valid(S, st, valid, res);
    if (rc) return rc;
    return (int64_t)S.out_size;
}

const u8* dec_output(DecState* S, u64* size) { *size = S ? S->out_size : 0; return S ? S->d_out : nullptr; }
void dec_error_info(DecState* S, int* detail, u32* got, u32* want) {
    *detail = S ? S->detail : 0; *got = S ? S->crc_got : 0; *want = S ? S->crc_want : 0;
}
u32 dec_table(DecState* S, const u64** pos, const u64** size) {
    if (!S) return 0;
    *pos = S->tab_pos.data(); *size = S->tab_size.data();
    return (u32)S->tab_pos.size();
}
