/*
 * oracle/bz2_decode_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the compressjs bzip2 DECODER (Bunzip, lib/Bzip2.js:91-548), the
 * "next" row of SURVEY.md 8(f)-1 / row a8.  Used only as the checker for the GPU decoder.
 *
 * Parity status: PINNED against the reference itself (node 12): tests/golden/golden_decode.json
 * (made by tests/golden/make_golden_decode.py) holds, for ~80 valid, truncated, concatenated and
 * corrupted streams, what Bzip2.decompressFile returned or threw; tests/test_oracle_decode.py
 * replays them here.
 *
 * One deliberate difference: a run of more than 31 RUNA/RUNB symbols makes the reference's
 * int32 `runPos <<= 1` wrap (lib/Bzip2.js:330-334) and can drive `t` negative, after which its
 * `while (t--)` loop (:345) never terminates.  This file returns DATA_ERROR for a negative `t`.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define D_OK 0
#define D_NOT_BZIP (-2)      /* lib/Bzip2.js:62-72 */
#define D_DATA_ERROR (-5)
#define D_OBSOLETE (-7)

/* detail codes (the optDetail strings of _throw) */
#define DD_NONE 0
#define DD_BAD_MAGIC 1        /* :142 */
#define DD_LEVEL 2            /* :146 */
#define DD_ORIGPTR 3          /* :178 */
#define DD_BLOCK_CRC 4        /* :441-445 */
#define DD_STREAM_CRC 5       /* :467-471 */

#define MAX_BITS 20
#define MAX_SYMS 258
#define GROUP 50

uint32_t orc_crc32(const uint8_t *p, uint64_t n);   /* bz2_oracle.c */

typedef struct {
    const uint8_t *in;
    uint64_t len;
    uint64_t p;          /* absolute bit position of the next unread bit */
} bitr;

/* BitStream.readBit / readBits (lib/BitStream.js:8-19,78-92): bits past EOF are zeros */
static inline uint32_t rd_bit(bitr *r) {
    uint64_t b = r->p >> 3;
    uint32_t v = b < r->len ? (r->in[b] >> (7 - (r->p & 7))) & 1u : 0u;
    r->p++;
    return v;
}
static uint64_t rd_bits(bitr *r, int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v = (v << 1) | rd_bit(r);
    return v;
}
/* the coerced input stream's eof() (lib/Util.js:30): every byte has been pulled into the reader */
static inline int at_eof(const bitr *r) { return ((r->p + 7) >> 3) >= r->len; }

typedef struct {
    uint16_t permute[MAX_SYMS];
    uint32_t limit[MAX_BITS + 2];
    uint32_t base[MAX_BITS + 1];
    int minLen, maxLen;
} hgroup;

typedef struct {
    uint64_t start_bit;   /* of the 48-bit block magic */
    uint64_t end_bit;     /* first bit after the block's last symbol */
    uint32_t n;           /* dbufCount: bytes before un-RLE1 */
    uint32_t orig_ptr;
    uint32_t crc;         /* targetBlockCRC */
    uint64_t out_bytes;   /* bytes this block decodes to */
} orc_dblock;

static uint8_t mtf_u8(uint8_t *a, int idx) {            /* lib/Bzip2.js:53-60 */
    uint8_t src = a[idx];
    for (int i = idx; i > 0; i--) a[i] = a[i - 1];
    a[0] = src;
    return src;
}

/* _get_next_block after the magic and CRC have been read (lib/Bzip2.js:170-397) followed by
 * _read_bunzip (:405-448).  dbuf holds dbufSize u32.  Appends to out (bounded by cap, counting on). */
static int decode_block(bitr *r, uint32_t dbufSize, uint32_t *dbuf, uint32_t target_crc, uint8_t *out,
                        uint64_t cap, uint64_t *opos, int *detail, orc_dblock *info) {
    if (rd_bit(r)) return D_OBSOLETE;                                        /* :174-175 */
    uint32_t origPointer = (uint32_t)rd_bits(r, 24);
    if (origPointer > dbufSize) { *detail = DD_ORIGPTR; return D_DATA_ERROR; }
    uint32_t t = (uint32_t)rd_bits(r, 16);                                   /* :185-195 */
    uint8_t symToByte[256];
    int symTotal = 0;
    memset(symToByte, 0, sizeof symToByte);
    for (int i = 0; i < 16; i++)
        if (t & (1u << (15 - i))) {
            uint32_t k = (uint32_t)rd_bits(r, 16);
            for (int j = 0; j < 16; j++)
                if (k & (1u << (15 - j))) symToByte[symTotal++] = (uint8_t)(i * 16 + j);
        }
    int groupCount = (int)rd_bits(r, 3);                                     /* :198-200 */
    if (groupCount < 2 || groupCount > 6) return D_DATA_ERROR;
    uint32_t nSelectors = (uint32_t)rd_bits(r, 15);                          /* :205-207 */
    if (nSelectors == 0) return D_DATA_ERROR;
    uint8_t mtfSymbol[256];
    memset(mtfSymbol, 0, sizeof mtfSymbol);
    for (int i = 0; i < groupCount; i++) mtfSymbol[i] = (uint8_t)i;
    uint8_t *selectors = (uint8_t *)malloc(nSelectors);
    for (uint32_t i = 0; i < nSelectors; i++) {                              /* :215-221 */
        int j;
        for (j = 0; rd_bit(r); j++)
            if (j >= groupCount) { free(selectors); return D_DATA_ERROR; }
        selectors[i] = mtf_u8(mtfSymbol, j);
    }
    int symCount = symTotal + 2;
    hgroup groups[6];
    for (int g = 0; g < groupCount; g++) {                                   /* :226-296 */
        uint8_t length[MAX_SYMS];
        uint16_t temp[MAX_BITS + 1];
        int tt = (int)rd_bits(r, 5);
        for (int i = 0; i < symCount; i++) {
            for (;;) {
                if (tt < 1 || tt > MAX_BITS) { free(selectors); return D_DATA_ERROR; }
                if (!rd_bit(r)) break;
                if (!rd_bit(r)) tt++; else tt--;
            }
            length[i] = (uint8_t)tt;
        }
        int minLen = length[0], maxLen = length[0];
        for (int i = 1; i < symCount; i++) {
            if (length[i] > maxLen) maxLen = length[i];
            else if (length[i] < minLen) minLen = length[i];
        }
        hgroup *h = &groups[g];
        memset(h, 0, sizeof *h);
        h->minLen = minLen; h->maxLen = maxLen;
        memset(temp, 0, sizeof temp);
        uint32_t pp = 0;
        for (int i = minLen; i <= maxLen; i++) {
            temp[i] = 0; h->limit[i] = 0;
            for (int s = 0; s < symCount; s++)
                if (length[s] == i) h->permute[pp++] = (uint16_t)s;
        }
        for (int i = 0; i < symCount; i++) temp[length[i]]++;
        pp = 0;
        uint32_t tsum = 0;
        for (int i = minLen; i < maxLen; i++) {
            pp += temp[i];
            h->limit[i] = pp - 1;
            pp <<= 1;
            tsum += temp[i];
            h->base[i + 1] = pp - tsum;
        }
        h->limit[maxLen] = pp + temp[maxLen] - 1;
        h->base[minLen] = 0;
        /* limit[maxLen+1] = MAX_VALUE sentinel: handled by the i > maxLen test below */
    }
    uint32_t byteCount[256];                                                 /* :301-366 */
    memset(byteCount, 0, sizeof byteCount);
    for (int i = 0; i < 256; i++) mtfSymbol[i] = (uint8_t)i;
    int32_t runPos = 0;
    int64_t run_t = 0;
    uint32_t dbufCount = 0, selector = 0;
    int sc = 0;
    hgroup *h = NULL;
    for (;;) {
        if (!(sc--)) {
            sc = GROUP - 1;
            if (selector >= nSelectors) { free(selectors); return D_DATA_ERROR; }
            h = &groups[selectors[selector++]];
        }
        int i = h->minLen;
        int64_t j = (int64_t)rd_bits(r, i);
        for (;; i++) {
            if (i > h->maxLen) { free(selectors); return D_DATA_ERROR; }
            if (j <= (int64_t)h->limit[i]) break;
            j = (j << 1) | rd_bit(r);
        }
        j -= (int64_t)h->base[i];
        if (j < 0 || j >= MAX_SYMS) { free(selectors); return D_DATA_ERROR; }
        int nextSym = h->permute[j];
        if (nextSym == 0 || nextSym == 1) {                                  /* :318-335 */
            if (!runPos) { runPos = 1; run_t = 0; }
            if (nextSym == 0) run_t += runPos; else run_t += 2 * (int64_t)runPos;
            runPos = (int32_t)((uint32_t)runPos << 1);
            continue;
        }
        if (runPos) {                                                        /* :340-347 */
            runPos = 0;
            if (run_t < 0) { free(selectors); return D_DATA_ERROR; }         /* see header */
            if ((int64_t)dbufCount + run_t > (int64_t)dbufSize) { free(selectors); return D_DATA_ERROR; }
            uint8_t uc = symToByte[mtfSymbol[0]];
            byteCount[uc] += (uint32_t)run_t;
            while (run_t--) dbuf[dbufCount++] = uc;
        }
        if (nextSym > symTotal) break;                                       /* :349-350 */
        if (dbufCount >= dbufSize) { free(selectors); return D_DATA_ERROR; }
        uint8_t uc = symToByte[mtf_u8(mtfSymbol, nextSym - 1)];
        byteCount[uc]++;
        dbuf[dbufCount++] = uc;
    }
    free(selectors);
    if (origPointer >= dbufCount) return D_DATA_ERROR;                       /* :372 */
    uint32_t jsum = 0;                                                       /* :374-378 */
    for (int i = 0; i < 256; i++) { uint32_t k = jsum + byteCount[i]; byteCount[i] = jsum; jsum = k; }
    for (uint32_t i = 0; i < dbufCount; i++) {                               /* :380-384 */
        uint8_t uc = (uint8_t)(dbuf[i] & 0xff);
        dbuf[byteCount[uc]] |= (i << 8);
        byteCount[uc]++;
    }
    if (info) { info->end_bit = r->p; info->n = dbufCount; info->orig_ptr = origPointer; info->crc = target_crc; }
    /* _read_bunzip :405-448 */
    uint32_t pos = dbuf[origPointer];
    int current = (int)(pos & 0xff), previous;
    pos >>= 8;
    int run = -1;
    uint32_t left = dbufCount;
    uint32_t crc = 0xffffffffu;
    static uint32_t tab[256];
    static int tab_ready = 0;
    if (!tab_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i << 24;
            for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04c11db7u : (c << 1);
            tab[i] = c;
        }
        tab_ready = 1;
    }
    uint64_t o = *opos, o0 = *opos;
    while (left) {
        left--;
        previous = current;
        pos = dbuf[pos];
        current = (int)(pos & 0xff);
        pos >>= 8;
        int copies, outbyte;
        if (run++ == 3) { copies = current; outbyte = previous; current = -1; }
        else { copies = 1; outbyte = current; }
        while (copies--) {
            crc = (crc << 8) ^ tab[((crc >> 24) ^ (uint32_t)outbyte) & 0xff];
            if (o < cap) out[o] = (uint8_t)outbyte;
            o++;
        }
        if (current != previous) run = 0;
    }
    *opos = o;
    if (info) info->out_bytes = o - o0;
    if ((~crc) != target_crc) { *detail = DD_BLOCK_CRC; return D_DATA_ERROR; }
    return D_OK;
}

/* Bunzip.decode (lib/Bzip2.js:454-481).  Returns the decoded size (may exceed cap: nothing is
 * written past cap) or a negative Err code; *detail gets a DD_* code.  blocks/max_blocks: optional
 * table of the blocks walked (what Bzip2.table reports, :508-548), *n_blocks their number. */
int64_t orc_bz2_decompress(const uint8_t *in, uint64_t len, uint8_t *out, uint64_t cap, int multistream,
                           int *detail, orc_dblock *blocks, uint32_t max_blocks, uint32_t *n_blocks) {
    int dd = DD_NONE;
    uint32_t nb = 0;
    int64_t ret;
    uint32_t *dbuf = NULL;
    uint64_t o = 0;
    uint64_t base = 0;           /* byte offset of the current stream header */
    bitr r = { in, len, 0 };
    uint32_t streamCRC = 0, dbufSize = 0;
#define FAIL(code, d) do { ret = (code); dd = (d); goto done; } while (0)
start_stream:
    if (len - base < 4 || in[base] != 'B' || in[base + 1] != 'Z' || in[base + 2] != 'h') FAIL(D_NOT_BZIP, DD_BAD_MAGIC);
    {
        int level = in[base + 3] - '0';
        if (level < 1 || level > 9) FAIL(D_NOT_BZIP, DD_LEVEL);
        dbufSize = 100000u * (uint32_t)level;
    }
    r.p = (base + 4) * 8;
    streamCRC = 0;
    free(dbuf);
    dbuf = (uint32_t *)malloc((size_t)dbufSize * 4);
    for (;;) {
        if (at_eof(&r)) break;                                               /* :462 */
        uint64_t start = r.p;
        uint64_t h = rd_bits(&r, 48);                                        /* :156-161 */
        if (h == 0x177245385090ull) {
            uint32_t target = (uint32_t)rd_bits(&r, 32);                     /* :466-471 */
            if (target != streamCRC) FAIL(D_DATA_ERROR, DD_STREAM_CRC);
            if (multistream && !at_eof(&r)) {                                /* :472-477 */
                base = (r.p + 7) >> 3;
                goto start_stream;
            }
            break;
        }
        if (h != 0x314159265359ull) FAIL(D_NOT_BZIP, DD_NONE);
        uint32_t target = (uint32_t)rd_bits(&r, 32);
        streamCRC = target ^ ((streamCRC << 1) | (streamCRC >> 31));         /* :163-164 */
        orc_dblock info;
        memset(&info, 0, sizeof info);
        info.start_bit = start;
        int d2 = DD_NONE;
        int rc = decode_block(&r, dbufSize, dbuf, target, out, cap, &o, &d2, &info);
        if (rc) FAIL(rc, d2);
        if (blocks && nb < max_blocks) blocks[nb] = info;
        nb++;
    }
    ret = (int64_t)o;
done:
    free(dbuf);
    if (detail) *detail = dd;
    if (n_blocks) *n_blocks = nb;
    return ret;
#undef FAIL
}

/* Bunzip.decodeBlock (lib/Bzip2.js:482-503): one block whose magic starts at bit `pos`. */
int64_t orc_bz2_decompress_block(const uint8_t *in, uint64_t len, uint64_t pos, uint8_t *out, uint64_t cap, int *detail) {
    int dd = DD_NONE;
    if (detail) *detail = DD_NONE;
    if (len < 4 || in[0] != 'B' || in[1] != 'Z' || in[2] != 'h') { if (detail) *detail = DD_BAD_MAGIC; return D_NOT_BZIP; }
    int level = in[3] - '0';
    if (level < 1 || level > 9) { if (detail) *detail = DD_LEVEL; return D_NOT_BZIP; }
    uint32_t dbufSize = 100000u * (uint32_t)level;
    bitr r = { in, len, pos };
    uint64_t h = rd_bits(&r, 48);
    if (h == 0x177245385090ull) return 0;                                    /* moreBlocks false: nothing written */
    if (h != 0x314159265359ull) return D_NOT_BZIP;
    uint32_t target = (uint32_t)rd_bits(&r, 32);
    uint32_t *dbuf = (uint32_t *)malloc((size_t)dbufSize * 4);
    uint64_t o = 0;
    int rc = decode_block(&r, dbufSize, dbuf, target, out, cap, &o, &dd, NULL);
    free(dbuf);
    if (detail) *detail = dd;
    return rc ? rc : (int64_t)o;
}
