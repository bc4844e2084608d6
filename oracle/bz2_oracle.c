/*
 * oracle/bz2_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the compressjs bzip2 block pipeline (the path BASELINE.json's
 * north_star names), written from the reference's behaviour, used ONLY as
 * the checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * compressjs_amd/ may include, link or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function here against golden
 * vectors produced by running the reference itself (node 12, tests/golden/make_golden.py):
 * the reference's own KATs (test/bwtest.js:38-90, test/huffman.js:15-77), whole-stream .bz2
 * digests for test/sample0..5.ref at -1/-9 and ~40 crafted / synthetic inputs.
 *
 * Known limit, inherited from the reference: orc_huff_lengths64 follows lib/HuffmanAllocator.js statement by statement,
 * and that algorithm indexes outside its array for weight vectors with many zeros when maxLen is far below what the
 * pipeline uses (e.g. 16 symbols at 6 bits).  The bzip2 path always calls it with maxLen = 20 (and the fuzz test,
 * tests/test_allocator_fuzz.py, stays at 15 bits and above), where it is in range.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 * The one deliberate algorithmic difference: the reference builds its suffix array with SA-IS
 * (lib/BWT.js:197-300); a suffix array is unique, so this file uses plain prefix doubling with
 * counting sorts on the same (doubled) string and obtains the identical array.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_SYMS 258   /* lib/Bzip2.js:41 MAX_SYMBOLS */
#define ORC_MAX_BITS 20    /* lib/Bzip2.js:40 MAX_HUFCODE_BITS */
#define ORC_MAX_GROUPS 6   /* lib/Bzip2.js:45 */
#define ORC_GROUP 50       /* lib/Bzip2.js:46 GROUP_SIZE */

/* ------------------------------------------------------------------------------------------
 * CRC (lib/CRC32.js:37-70 table, :89-91 update, :82-84 final complement)
 * poly 0x04c11db7, MSB first, init 0xffffffff, xorout 0xffffffff
 * ------------------------------------------------------------------------------------------ */
static uint32_t crc_table[256];
static int crc_ready = 0;
static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i << 24;
        for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04c11db7u : (c << 1);
        crc_table[i] = c;
    }
    crc_ready = 1;
}
static inline uint32_t crc_update(uint32_t crc, uint8_t v) {
    return (crc << 8) ^ crc_table[((crc >> 24) ^ v) & 0xff];
}
uint32_t orc_crc32(const uint8_t *p, uint64_t n) {
    if (!crc_ready) crc_init();
    uint32_t c = 0xffffffffu;
    for (uint64_t i = 0; i < n; i++) c = crc_update(c, p[i]);
    return ~c;
}

/* ------------------------------------------------------------------------------------------
 * readBlock: RLE1 while filling one block + CRC of the bytes consumed (lib/Bzip2.js:636-667)
 * returns the block length; *in_pos advances by the number of input bytes consumed.
 * ------------------------------------------------------------------------------------------ */
uint32_t orc_read_block(const uint8_t *in, uint64_t in_len, uint64_t *in_pos, uint8_t *block,
                        uint32_t cap, uint32_t *crc_out) {
    if (!crc_ready) crc_init();
    uint32_t pos = 0, crc = 0xffffffffu;
    int last = -1, run = 0;
    uint64_t ip = *in_pos;
    while (pos < cap) {
        if (run == 4) {                 /* :641-644 count byte, may fill the block */
            block[pos++] = 0;
            if (pos >= cap) break;
        }
        if (ip >= in_len) break;        /* :646 EOF */
        int ch = in[ip++];
        crc = crc_update(crc, (uint8_t)ch);
        if (ch != last) {
            last = ch; run = 1;
        } else {
            run++;
            if (run > 4) {
                if (run < 256) { block[pos - 1]++; continue; }   /* :656-658 */
                run = 1;                                         /* :660 */
            }
        }
        block[pos++] = (uint8_t)ch;
    }
    *in_pos = ip;
    *crc_out = ~crc;
    return pos;
}

/* ------------------------------------------------------------------------------------------
 * Suffix array of T[0..n) with the implicit-smallest-sentinel order SA-IS produces
 * (lib/BWT.js:197-300, entry points :305-321).  Prefix doubling; unique result.
 * ------------------------------------------------------------------------------------------ */
int orc_suffixsort(const uint8_t *T, int32_t *SA, uint32_t n) {
    if (n == 0) return 0;
    if (n == 1) { SA[0] = 0; return 0; }
    uint32_t m = n > 256 ? n : 256;
    int32_t *rank = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (m + 2));
    if (!rank || !tmp || !cnt) { free(rank); free(tmp); free(cnt); return -1; }
    memset(cnt, 0, sizeof(int32_t) * 258);
    for (uint32_t i = 0; i < n; i++) cnt[T[i] + 1]++;
    for (int c = 0; c < 256; c++) cnt[c + 1] += cnt[c];
    for (uint32_t i = 0; i < n; i++) SA[cnt[T[i]]++] = (int32_t)i;
    int32_t r = 0;
    rank[SA[0]] = 0;
    for (uint32_t j = 1; j < n; j++) {
        if (T[SA[j]] != T[SA[j - 1]]) r++;
        rank[SA[j]] = r;
    }
    for (uint32_t h = 1; (uint32_t)r + 1 < n; h <<= 1) {
        /* order by second key rank[i+h] (past the end = smallest) */
        uint32_t p = 0;
        for (uint32_t i = n - (h < n ? h : n); i < n; i++) tmp[p++] = (int32_t)i;
        for (uint32_t j = 0; j < n; j++) if ((uint32_t)SA[j] >= h) tmp[p++] = SA[j] - (int32_t)h;
        /* stable counting sort by first key */
        memset(cnt, 0, sizeof(int32_t) * ((uint32_t)r + 2));
        for (uint32_t i = 0; i < n; i++) cnt[rank[i] + 1]++;
        for (int32_t c = 0; c <= r; c++) cnt[c + 1] += cnt[c];
        for (uint32_t j = 0; j < n; j++) SA[cnt[rank[tmp[j]]]++] = tmp[j];
        int32_t r2 = 0;
        tmp[SA[0]] = 0;
        for (uint32_t j = 1; j < n; j++) {
            uint32_t a = (uint32_t)SA[j - 1], b = (uint32_t)SA[j];
            int32_t ka = a + h < n ? rank[a + h] : -1, kb = b + h < n ? rank[b + h] : -1;
            if (rank[a] != rank[b] || ka != kb) r2++;
            tmp[b] = r2;
        }
        int32_t *sw = rank; rank = tmp; tmp = sw;
        r = r2;
        if (h >= n) break;
    }
    free(rank); free(tmp); free(cnt);
    return 0;
}

/* BWT.bwtransform2 (lib/BWT.js:372-417): cyclic BWT through the suffix array of T||T. */
int32_t orc_bwt_cyclic(const uint8_t *T, uint8_t *U, uint32_t n) {
    if (n <= 1) { if (n == 1) U[0] = T[0]; return 0; }       /* :376-379 */
    uint8_t *TT = (uint8_t *)malloc(2 * (size_t)n);
    int32_t *A = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
    if (!TT || !A) { free(TT); free(A); return -1; }
    memcpy(TT, T, n); memcpy(TT + n, T, n);                    /* :390-403 */
    orc_suffixsort(TT, A, 2 * n);                              /* :405-406 */
    int32_t pidx = 0; uint32_t j = 0;
    for (uint32_t i = 0; i < 2 * n; i++) {                     /* :407-414 */
        int32_t s = A[i];
        if ((uint32_t)s < n) {
            if (s == 0) pidx = (int32_t)j;
            if (--s < 0) s = (int32_t)n - 1;
            U[j++] = T[s];
        }
    }
    free(TT); free(A);
    return pidx;
}

/* BWT.bwtransform (lib/BWT.js:328-350 + computeBWT :153-192): BWT of T$ */
int32_t orc_bwt_linear(const uint8_t *T, uint8_t *U, uint32_t n) {
    if (n <= 1) { if (n == 1) U[0] = T[0]; return (int32_t)n; }   /* :332-335 */
    int32_t *SA = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!SA) return -1;
    orc_suffixsort(T, SA, n);
    uint32_t j = 0; int32_t pidx = 0;
    U[j++] = T[n - 1];                                              /* :346 */
    for (uint32_t r = 0; r < n; r++) {
        if (SA[r] == 0) { pidx = (int32_t)r; continue; }
        U[j++] = T[SA[r] - 1];
    }
    free(SA);
    return pidx + 1;                                                /* :349 */
}

/* BWT.unbwtransform (lib/BWT.js:352-363) */
void orc_unbwt_linear(const uint8_t *T, uint8_t *U, uint32_t n, uint32_t pidx) {
    uint32_t C[256], t; int32_t i;
    uint32_t *LF = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    memset(C, 0, sizeof C);
    for (uint32_t k = 0; k < n; k++) LF[k] = C[T[k]]++;
    t = 0;
    for (int c = 0; c < 256; c++) { t += C[c]; C[c] = t - C[c]; }
    for (i = (int32_t)n - 1, t = 0; i >= 0; i--) {
        if (t >= n) { U[i] = 0; continue; }                         /* JS: T[n] is undefined, t turns NaN, U[i] = T[NaN] stores 0 */
        U[i] = T[t];
        t = LF[t] + C[U[i]];
        t += (t < pidx) ? 1 : 0;
    }
    free(LF);
}

/* ------------------------------------------------------------------------------------------
 * allocateHuffmanCodeLengths (lib/HuffmanAllocator.js:199-222), in place on int64 values
 * ------------------------------------------------------------------------------------------ */
static int fls32(uint32_t v) { int r = 0; while (v) { r++; v >>= 1; } return r; }   /* lib/Util.js:301-317 */

static int ha_first(const int64_t *a, int len, int i, int nodes_to_move) {           /* :52-73 */
    int limit = i, k = len - 2;
    while (i >= nodes_to_move && (a[i] % len) > limit) { k = i; i -= (limit - i + 1); }
    if (i < nodes_to_move - 1) i = nodes_to_move - 1;
    while (k > i + 1) {
        int t = (i + k) >> 1;
        if ((a[t] % len) > limit) k = t; else i = t;
    }
    return k;
}
static void ha_set_parents(int64_t *a, int len) {                                     /* :79-105 */
    a[0] += a[1];
    int head = 0, tail = 1, top = 2;
    for (; tail < len - 1; tail++) {
        int64_t t;
        if (top >= len || a[head] < a[top]) { t = a[head]; a[head++] = tail; }
        else t = a[top++];
        if (top >= len || (head < tail && a[head] < a[top])) { t += a[head]; a[head++] = tail + len; }
        else t += a[top++];
        a[tail] = t;
    }
}
static int ha_find_relocate(const int64_t *a, int len, int maxlen) {                  /* :114-124 */
    int cur = len - 2;
    for (int d = 1; d < maxlen - 1 && cur > 1; d++) cur = ha_first(a, len, cur - 1, 0);
    return cur;
}
static void ha_alloc(int64_t *a, int len) {                                           /* :131-148 */
    int first = len - 2, next = len - 1;
    for (int depth = 1, avail = 2; avail > 0; depth++) {
        int last = first;
        first = ha_first(a, len, last - 1, 0);
        for (int i = avail - (last - first); i > 0; i--) a[next--] = depth;
        avail = (last - first) << 1;
    }
}
static void ha_alloc_reloc(int64_t *a, int len, int nodes_to_move, int insert_depth) { /* :157-188 */
    int first = len - 2, next = len - 1;
    int depth = (insert_depth == 1) ? 2 : 1;
    int left = (insert_depth == 1) ? nodes_to_move - 2 : nodes_to_move;
    for (int avail = depth << 1; avail > 0; depth++) {
        int last = first;
        first = (first <= nodes_to_move) ? first : ha_first(a, len, last - 1, nodes_to_move);
        int offset = 0;
        if (depth >= insert_depth) {
            int cap = 1 << (depth - insert_depth);
            offset = left < cap ? left : cap;
        } else if (depth == insert_depth - 1) {
            offset = 1;
            if (a[first] == last) first++;
        }
        for (int i = avail - (last - first + offset); i > 0; i--) a[next--] = depth;
        left -= offset;
        avail = (last - first + offset) << 1;
    }
}
void orc_huff_lengths64(int64_t *a, int len, int maxlen) {
    if (len == 2) { a[1] = 1; a[0] = 1; return; }     /* :200-206 */
    if (len == 1) { a[0] = 1; return; }
    if (len <= 0) return;
    ha_set_parents(a, len);
    int reloc = ha_find_relocate(a, len, maxlen);
    if ((a[0] % len) >= reloc) ha_alloc(a, len);       /* :216 */
    else ha_alloc_reloc(a, len, reloc, maxlen - fls32((uint32_t)(reloc - 1)));
}
/* int32 convenience wrapper (the C-ABI's cjs_huff_lengths has this shape) */
void orc_huff_lengths(int32_t *arr, uint32_t n, uint32_t maxlen) {
    int64_t *a = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) a[i] = arr[i];
    orc_huff_lengths64(a, (int)n, (int)maxlen);
    for (uint32_t i = 0; i < n; i++) arr[i] = (int32_t)a[i];
    free(a);
}

/* StaticHuffman constructor (lib/Bzip2.js:551-579): code lengths for freq[0..S) */
static int cmp_i64(const void *x, const void *y) {
    int64_t a = *(const int64_t *)x, b = *(const int64_t *)y;
    return a < b ? -1 : a > b;
}
void orc_static_huffman(const uint32_t *freq, int S, uint8_t *lens) {
    int64_t merged[ORC_MAX_SYMS], sorted[ORC_MAX_SYMS];
    for (int i = 0; i < S; i++) merged[i] = ((int64_t)freq[i] << 9) | i;   /* :566-568 */
    qsort(merged, S, sizeof(int64_t), cmp_i64);                              /* keys are distinct */
    for (int i = 0; i < S; i++) sorted[i] = merged[i] >> 9;
    orc_huff_lengths64(sorted, S, ORC_MAX_BITS);
    for (int i = 0; i < S; i++) lens[merged[i] & 0x1ff] = (uint8_t)sorted[i];
}
/* computeCanonical (lib/Bzip2.js:581-600) */
void orc_canonical(const uint8_t *lens, int S, uint32_t *code) {
    int64_t merged[ORC_MAX_SYMS];
    for (int i = 0; i < S; i++) merged[i] = ((int64_t)lens[i] << 9) | i;
    qsort(merged, S, sizeof(int64_t), cmp_i64);
    uint32_t c = 0; int prev = 0;
    for (int i = 0; i < S; i++) {
        int cur = (int)(merged[i] >> 9), sym = (int)(merged[i] & 0x1ff);
        c <<= (cur - prev);
        code[sym] = c++;
        prev = cur;
    }
}

/* ------------------------------------------------------------------------------------------
 * MSB-first bit writer (lib/BitStream.js:52-73,93-105)
 * ------------------------------------------------------------------------------------------ */
typedef struct { uint8_t *out; uint64_t cap, bits; int overflow; } bitw;
static inline void bw_put(bitw *w, int n, uint64_t v) {
    for (int i = n - 1; i >= 0; i--) {
        uint64_t byte = w->bits >> 3;
        if (byte >= w->cap) { w->overflow = 1; w->bits++; continue; }
        if ((w->bits & 7) == 0) w->out[byte] = 0;
        if ((v >> i) & 1) w->out[byte] |= (uint8_t)(0x80 >> (w->bits & 7));
        w->bits++;
    }
}

/* ------------------------------------------------------------------------------------------
 * Stage outputs of one block, for differential tests of the HIP stages.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t n;              /* block length after RLE1                      */
    uint32_t pidx;           /* origPtr                                      */
    uint32_t alphabet_size;  /* number of distinct bytes in the block        */
    uint32_t pos;            /* MTF/RLE2 symbols incl. EOB                   */
    uint32_t n_groups;
    uint32_t n_selectors;
    uint32_t crc;
    uint32_t reserved;
    uint64_t in_consumed;
    uint64_t bit_len;        /* bits from the block magic to the last code   */
} orc_block_info;

static int cost_of(const uint8_t *lens, const uint16_t *A, uint32_t off, uint32_t len) {  /* :602-608 */
    int c = 0;
    for (uint32_t i = 0; i < len; i++) c += lens[A[off + i]];
    return c;
}
static void assign_selectors(uint8_t *sel, uint8_t lens[][ORC_MAX_SYMS], int G, const uint16_t *A,
                             uint32_t pos) {                                               /* :671-684 */
    uint32_t k = 0;
    for (uint32_t i = 0; i < pos; i += ORC_GROUP) {
        uint32_t gs = pos - i < ORC_GROUP ? pos - i : ORC_GROUP;
        int best = 0, bc = cost_of(lens[0], A, i, gs);
        for (int j = 1; j < G; j++) {
            int c = cost_of(lens[j], A, i, gs);
            if (c < bc) { best = j; bc = c; }
        }
        sel[k++] = (uint8_t)best;
    }
}

/* compressBlock (lib/Bzip2.js:735-876).  block[0..n) is the RLE1 output.  If stage pointers are
 * non-NULL they receive U[n], A[pos], selectors, lens[6][258].  Bits are appended to w. */
static int compress_block(const uint8_t *block, uint32_t n, bitw *w, orc_block_info *info,
                          uint8_t *U_out, uint16_t *A_out, uint8_t *sel_out, uint8_t *lens_out) {
    uint8_t *U = (uint8_t *)malloc(n ? n : 1);
    uint16_t *A = (uint16_t *)malloc(sizeof(uint16_t) * ((size_t)n + 1));
    uint32_t nsel_cap = (n + 1 + ORC_GROUP - 1) / ORC_GROUP + 1;
    uint8_t *sel = (uint8_t *)malloc(nsel_cap);
    if (!U || !A || !sel) { free(U); free(A); free(sel); return -1; }
    int32_t pidx = orc_bwt_cyclic(block, U, n);                        /* :739 */
    bw_put(w, 1, 0); bw_put(w, 24, (uint32_t)pidx);                    /* :740-741 */
    int used[256], compact[16];
    memset(used, 0, sizeof used); memset(compact, 0, sizeof compact);
    for (uint32_t i = 0; i < n; i++) { used[block[i]] = 1; compact[block[i] >> 4] = 1; }
    for (int i = 0; i < 16; i++) bw_put(w, 1, compact[i]);             /* :749-751 */
    for (int i = 0; i < 16; i++) if (compact[i])
        for (int j = 0; j < 16; j++) bw_put(w, 1, used[(i << 4) | j]); /* :752-758 */
    int alpha = 0;
    for (int i = 0; i < 256; i++) alpha += used[i];
    int eob = alpha + 1, S = alpha + 2;
    uint32_t freq[ORC_MAX_SYMS];
    memset(freq, 0, sizeof freq);
    uint8_t M[256];
    for (int i = 0, j = 0; i < 256; i++) if (used[i]) M[j++] = (uint8_t)i;
    uint32_t pos = 0, run = 0;
#define EMIT(c) do { A[pos++] = (uint16_t)(c); freq[(c)]++; } while (0)
#define FLUSH_RUN() do { while (run) { if (run & 1) { EMIT(0); run -= 1; } else { EMIT(1); run -= 2; } run >>= 1; } } while (0)
    for (uint32_t i = 0; i < n; i++) {                                 /* :795-812 */
        uint8_t c = U[i];
        int j = 0;
        while (M[j] != c) j++;
        for (int k = j; k > 0; k--) M[k] = M[k - 1];
        M[0] = c;
        if (j == 0) run++;
        else { FLUSH_RUN(); EMIT(j + 1); run = 0; }
    }
    FLUSH_RUN();
    EMIT(eob);                                                          /* :813-814 */
    int target = pos >= 2400 ? 6 : pos >= 1200 ? 5 : pos >= 600 ? 4 : pos >= 200 ? 3 : 2;  /* :826-830 */
    uint8_t lens[ORC_MAX_GROUPS][ORC_MAX_SYMS];
    int G = 0;
    orc_static_huffman(freq, S, lens[G++]);                            /* :835 */
    for (int i = 0; i < S; i++) freq[i] = 1;
    orc_static_huffman(freq, S, lens[G++]);                            /* :836-837 */
    uint32_t nsel = (pos + ORC_GROUP - 1) / ORC_GROUP;                 /* :841 */
    /* optimizeHuffmanGroups :685-733 */
    uint32_t *split_idx = (uint32_t *)malloc(sizeof(uint32_t) * (nsel + 1));
    uint32_t *split_tmp = (uint32_t *)malloc(sizeof(uint32_t) * (nsel + 1));
    uint32_t (*gfreq)[ORC_MAX_SYMS] = malloc(sizeof(uint32_t) * ORC_MAX_GROUPS * ORC_MAX_SYMS);
    while (G < target) {
        assign_selectors(sel, lens, G, A, pos);
        uint32_t counts[ORC_MAX_GROUPS] = {0};
        for (uint32_t i = 0; i < nsel; i++) counts[sel[i]]++;
        int which = 0;
        for (int i = 1; i < G; i++) if (counts[i] > counts[which]) which = i;   /* first max :699 */
        /* stable sort of the groups using `which` by cost (counting sort, cost <= 50*20) :701-710 */
        uint32_t bucket[ORC_GROUP * ORC_MAX_BITS + 2];
        memset(bucket, 0, sizeof bucket);
        uint32_t m = 0;
        for (uint32_t i = 0; i < nsel; i++) if (sel[i] == which) {
            uint32_t start = i * ORC_GROUP, end = start + ORC_GROUP < pos ? start + ORC_GROUP : pos;
            split_tmp[m] = (uint32_t)cost_of(lens[which], A, start, end - start);
            split_idx[m++] = i;
            bucket[split_tmp[m - 1] + 1]++;
        }
        for (int c = 0; c <= ORC_GROUP * ORC_MAX_BITS; c++) bucket[c + 1] += bucket[c];
        uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * (m + 1));
        for (uint32_t k = 0; k < m; k++) order[bucket[split_tmp[k]]++] = split_idx[k];
        for (uint32_t k = m >> 1; k < m; k++) sel[order[k]] = (uint8_t)G;       /* :712-714 */
        free(order);
        G++;
        memset(gfreq, 0, sizeof(uint32_t) * ORC_MAX_GROUPS * ORC_MAX_SYMS);      /* :717-727 */
        for (uint32_t i = 0, j = 0; i < pos;) {
            uint32_t *f = gfreq[sel[j++]];
            for (int k = 0; k < ORC_GROUP && i < pos; k++) f[A[i++]]++;
        }
        for (int g = 0; g < G; g++) orc_static_huffman(gfreq[g], S, lens[g]);    /* :729-731 */
    }
    free(split_idx); free(split_tmp); free(gfreq);
    assign_selectors(sel, lens, G, A, pos);                             /* :843 */
    bw_put(w, 3, (uint32_t)G);                                          /* :847 */
    bw_put(w, 15, nsel);                                                /* :849 */
    {   /* :850-862.  The reference reuses the Uint8Array M of length `alpha` here: writes past
           its end are dropped and reads past its end are `undefined` (never equal). */
        int Ms[ORC_MAX_GROUPS + 1];
        int mlen = alpha < G ? alpha : G;
        for (int i = 0; i < mlen; i++) Ms[i] = i;
        for (uint32_t i = 0; i < nsel; i++) {
            int s = sel[i], j;
            for (j = 0; j < G; j++) if (j < alpha && Ms[j] == s) break;
            int src = j < alpha ? Ms[j] : 0;                  /* undefined -> 0 on store */
            for (int k = j; k > 0; k--) if (k < alpha) Ms[k] = (k - 1 < alpha) ? Ms[k - 1] : 0;
            if (alpha > 0) Ms[0] = src;
            for (; j > 0; j--) bw_put(w, 1, 1);
            bw_put(w, 1, 0);
        }
    }
    uint32_t codes[ORC_MAX_GROUPS][ORC_MAX_SYMS];
    for (int g = 0; g < G; g++) {                                       /* :864-867, emit :610-629 */
        int cur = lens[g][0];
        bw_put(w, 5, (uint32_t)cur);
        for (int i = 0; i < S; i++) {
            int cl = lens[g][i];
            int val = cur < cl ? 2 : 3, delta = cur < cl ? cl - cur : cur - cl;
            while (delta-- > 0) bw_put(w, 2, (uint32_t)val);
            bw_put(w, 1, 0);
            cur = cl;
        }
        orc_canonical(lens[g], S, codes[g]);
    }
    for (uint32_t i = 0, k = 0; i < pos;) {                             /* :869-874 */
        int g = sel[k++];
        for (int j = 0; j < ORC_GROUP && i < pos; j++, i++) bw_put(w, lens[g][A[i]], codes[g][A[i]]);
    }
    if (info) {
        info->n = n; info->pidx = (uint32_t)pidx; info->alphabet_size = (uint32_t)alpha;
        info->pos = pos; info->n_groups = (uint32_t)G; info->n_selectors = nsel;
    }
    if (U_out) memcpy(U_out, U, n);
    if (A_out) memcpy(A_out, A, sizeof(uint16_t) * pos);
    if (sel_out) memcpy(sel_out, sel, nsel);
    if (lens_out) { memset(lens_out, 0, ORC_MAX_GROUPS * ORC_MAX_SYMS); for (int g = 0; g < G; g++) memcpy(lens_out + g * ORC_MAX_SYMS, lens[g], S); }
    free(U); free(A); free(sel);
    return 0;
}

/* Worst-case output size for a given input length (not a reference function). */
int64_t orc_bz2_bound(uint64_t in_len) { return (int64_t)(in_len + in_len / 2 + 4096); }

/* Bzip2.compressFile (lib/Bzip2.js:879-929) on a whole buffer; returns bytes written or <0 */
int64_t orc_bz2_compress(const uint8_t *in, uint64_t in_len, int level, uint8_t *out, uint64_t out_cap) {
    if (level < 1 || level > 9) return -1;                              /* :888-890 */
    uint32_t cap = (uint32_t)level * 100000u - 19u;                     /* :892-900 */
    uint8_t *block = (uint8_t *)malloc(cap);
    if (!block) return -2;
    bitw w = { out, out_cap, 0, 0 };
    bw_put(&w, 8, 'B'); bw_put(&w, 8, 'Z'); bw_put(&w, 8, 'h'); bw_put(&w, 8, (uint32_t)('0' + level));
    uint32_t stream_crc = 0, length;
    uint64_t ip = 0;
    do {                                                                /* :913-922 */
        uint32_t crc;
        length = orc_read_block(in, in_len, &ip, block, cap, &crc);
        if (length > 0) {
            stream_crc = ((stream_crc << 1) | (stream_crc >> 31)) ^ crc;
            bw_put(&w, 48, 0x314159265359ull);
            bw_put(&w, 32, crc);
            if (compress_block(block, length, &w, NULL, NULL, NULL, NULL, NULL)) { free(block); return -2; }
        }
    } while (length == cap);
    bw_put(&w, 48, 0x177245385090ull);                                  /* :925-927 */
    bw_put(&w, 32, stream_crc);
    while (w.bits & 7) bw_put(&w, 1, 0);
    free(block);
    if (w.overflow) return -3;
    return (int64_t)(w.bits >> 3);
}

/* Stage dump of the block that starts at input offset *in_pos (for differential tests).
 * T_out must hold level*100000-19 bytes, U_out the same, A_out one more uint16 than that,
 * sel_out (cap+1)/50+2 bytes, lens_out 6*258 bytes.  Returns 0, or 1 when no block remains. */
int orc_bz2_block_stages(const uint8_t *in, uint64_t in_len, uint64_t *in_pos, int level,
                         orc_block_info *info, uint8_t *T_out, uint8_t *U_out, uint16_t *A_out,
                         uint8_t *sel_out, uint8_t *lens_out) {
    uint32_t cap = (uint32_t)level * 100000u - 19u;
    uint64_t ip = *in_pos;
    uint32_t crc;
    uint32_t n = orc_read_block(in, in_len, &ip, T_out, cap, &crc);
    if (n == 0) { *in_pos = ip; return 1; }
    uint64_t scratch_cap = (uint64_t)n * 2 + 8192;
    uint8_t *scratch = (uint8_t *)malloc(scratch_cap);
    bitw w = { scratch, scratch_cap, 0, 0 };
    bw_put(&w, 48, 0x314159265359ull);
    bw_put(&w, 32, crc);
    int rc = compress_block(T_out, n, &w, info, U_out, A_out, sel_out, lens_out);
    info->crc = crc;
    info->in_consumed = ip - *in_pos;
    info->bit_len = w.bits;
    *in_pos = ip;
    free(scratch);
    return rc ? -1 : 0;
}
