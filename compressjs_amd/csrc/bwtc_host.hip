// BWTC -6..-9 (BASELINE.json configs[4]): the GPU produces, per 100000*level-byte block, the
// linear BWT (K1, sentinel mode), the used-symbol set and the MTF/RLE2 symbol stream (K2); this
// file is the serial entropy tail -- the range coder's state crosses blocks, so it cannot shard
// ("IN, CPU-side, serial by construction").  Host code only.
//
// Restates, on 32-bit unsigned integers:
//   Util.compressFileHelper      lib/Util.js:105-142    magic, varint(size+1), last byte withheld
//   RangeCoder (encoder)         lib/RangeCoder.js:27-140
//   NoModel.encode               lib/NoModel.js:15-21   bits through encodeShift(1,b,1)
//   LogDistanceModel.encode      lib/LogDistanceModel.js:24-36
//   FenwickModel                 lib/FenwickModel.js:13-32,47-87,137-172
//   BWTC.compressFile body       lib/BWTC.js:12-139
// and the decoder sides: RangeCoder :146-226, NoModel.decode :22-29, LogDistanceModel.decode :37-44,
// FenwickModel._decode/decode :88-136, Util.decompressFileHelper lib/Util.js:143-166, BWTC.decompressFile :141-233.
#include "bwtc_host.h"
#include <string.h>
#include <vector>

namespace {

struct Out {
    uint8_t* p; uint64_t cap, n; bool overflow;
    void put(uint32_t b) { if (n < cap) p[n] = (uint8_t)b; else overflow = true; n++; }
};

const uint32_t TOP = 0x80000000u;            // Top_value   lib/RangeCoder.js:15
const uint32_t BOTTOM = TOP >> 8;            // Bottom_value :18
const int SHIFT_BITS = 23;                   // :16

// floor(range / tot) of lib/RangeCoder.js:81 without the division: the coder is a serial chain (every call needs the range the
// previous one left), and the 32-bit division was most of its ~7.5 ns per call.  For tot <= 0xFFFF and range < 2^32,
//   floor(range / tot) == (range * M[tot]) >> 48   with   M[tot] = ceil(2^48 / tot):
// M = 2^48 / tot + e, 0 <= e < 1, so range * M / 2^48 = range / tot + e * range / 2^48 < range / tot + 2^-16, and the
// fractional part of range / tot is at most 1 - 1 / tot <= 1 - 1 / 65535 < 1 - 2^-16: the floor cannot move.  (tests check
// every tot against the multiples of tot around 2^k and at the ends of the range.)  The table entry does not depend on the
// chain, so its load (512 KB table: L2) is off the critical path; bigger totals (none in BWTC: max_prob = 0xFF00) divide.
static const uint64_t* rc_recip_table() {
    static uint64_t* tab = []() {
        uint64_t* t = new uint64_t[65536];
        t[0] = 0;
        for (uint32_t d = 1; d < 65536; d++) t[d] = (uint64_t)((((unsigned __int128)1 << 48) + d - 1) / d);
        return t;
    }();
    return tab;
}
static const uint64_t* const g_rc_recip = rc_recip_table();
static inline uint32_t rc_div(uint32_t range, uint32_t tot) {
    if (tot < 65536u) return (uint32_t)(((unsigned __int128)range * g_rc_recip[tot]) >> 48);
    return range / tot;
}

struct RangeEnc {                            // lib/RangeCoder.js:27-140
    uint32_t low, range, buffer, help, bytecount;
    Out* o;
    void start(uint32_t c, uint32_t initlength) { low = 0; range = TOP; buffer = c; help = 0; bytecount = initlength; }
    // (a branch-free first step - conditional moves on `range <= BOTTOM`, true for ~45 % of the calls - was measured SLOWER:
    // 6.6 vs 5.3 ns per call on the EPYC 9575F of the GPU box; the coder is bound by its dependency chain, not by mispredictions)
    void normalize() {                       // :38-60
        while (range <= BOTTOM) {
            if (low < (0xFFu << SHIFT_BITS)) {
                o->put(buffer);
                for (; help; help--) o->put(0xFF);
                buffer = (low >> SHIFT_BITS) & 0xFF;
            } else if (low & TOP) {
                o->put(buffer + 1);
                for (; help; help--) o->put(0x00);
                buffer = (low >> SHIFT_BITS) & 0xFF;
            } else {
                help++;
            }
            range <<= 8;
            low = (low << 8) & (TOP - 1);
            bytecount++;
        }
    }
    void encodeFreq(uint32_t sy_f, uint32_t lt_f, uint32_t tot_f) {   // :79-89
        normalize();
        const uint32_t r = rc_div(range, tot_f);
        const uint32_t tmp = r * lt_f;
        low += tmp;
        if (lt_f + sy_f < tot_f) range = r * sy_f; else range -= tmp;
    }
    void encodeShift(uint32_t sy_f, uint32_t lt_f, int shift) {       // :90-100
        normalize();
        const uint32_t r = range >> shift;
        const uint32_t tmp = r * lt_f;
        low += tmp;
        if ((lt_f + sy_f) >> shift) range -= tmp; else range = r * sy_f;
    }
    void encodeBit(uint32_t b) { encodeShift(1, b ? 1 : 0, 1); }      // :102-104
    void encodeByte(uint32_t b) { encodeShift(1, b, 8); }             // :106-108
    void finish() {                                                   // :116-140
        normalize();
        bytecount += 5;
        uint32_t tmp = low >> SHIFT_BITS;
        if ((low & (BOTTOM - 1)) >= ((bytecount & 0xFFFFFF) >> 1)) tmp++;
        if (tmp > 0xFF) { o->put(buffer + 1); for (; help; help--) o->put(0x00); }
        else { o->put(buffer); for (; help; help--) o->put(0xFF); }
        o->put(tmp & 0xFF);
        o->put((bytecount >> 16) & 0xFF);
        o->put((bytecount >> 8) & 0xFF);
        o->put(bytecount & 0xFF);
    }
};

int fls(uint32_t v) { int r = 0; while (v) { r++; v >>= 1; } return r; }     // lib/Util.js:301

void nomodel_encode(RangeEnc& rc, int bits, uint32_t sym) {                  // lib/NoModel.js:15-21
    for (int i = bits - 1; i >= 0; i--) rc.encodeBit((sym >> i) & 1);
}

struct LogDistance {                                                         // lib/LogDistanceModel.js
    int lgbits;                              // NoModel(1 + bits): fls(bits) bits
    void init(uint32_t size) { lgbits = fls((uint32_t)(1 + fls(size - 1)) - 1); }
    void encode(RangeEnc& rc, uint32_t d) {
        if (d < 2) { nomodel_encode(rc, lgbits, d); return; }
        const int lg = fls(d);
        nomodel_encode(rc, lgbits, (uint32_t)lg);
        // distanceModel[lg] = NoModel(1 << (lg-1)): fls((1<<(lg-1))-1) = lg-1 bits
        nomodel_encode(rc, lg - 1, d & ((1u << (lg - 1)) - 1));
    }
};

struct Fenwick {                                                             // lib/FenwickModel.js
    uint32_t numSyms, increment, max_prob;
    uint32_t tree[2 * 260];
    void init(uint32_t size, uint32_t maxp, uint32_t inc) {                  // :13-32
        numSyms = size + 1; increment = inc; max_prob = maxp;
        memset(tree, 0, sizeof tree);
        for (uint32_t i = 0; i < size; i++) tree[numSyms + i] = 1u;          // escape prob 1, sym prob 0
        tree[numSyms + size] = increment << 16;                              // the escape symbol
        sum();
    }
    void sum() { for (uint32_t i = numSyms - 1; i > 0; i--) tree[i] = tree[2 * i] + tree[2 * i + 1]; }   // :167-172
    void encode(RangeEnc& rc, uint32_t symbol) {                             // :47-87
        uint32_t i = numSyms + symbol;
        uint32_t sy_f = tree[i];
        uint32_t mask = 0xFFFF0000u; int shift = 16;
        uint32_t update = increment << 16;
        if ((sy_f & 0xFFFF0000u) == 0) {                                     // escape first
            encode(rc, numSyms - 1);
            mask = 0x0000FFFFu; update -= 1u; shift = 0;
        } else if (symbol == numSyms - 1 && (tree[1] & 0xFFFFu) == 1u) {
            update = 0u - tree[i];                                           // last escape: zero it out
        }
        uint32_t lt_f = 0;
        while (i > 1) {
            const uint32_t parent = i >> 1;
            if (i & 1u) lt_f += tree[2 * parent];
            tree[i] += update;
            i = parent;
        }
        uint32_t tot_f = tree[1];
        tree[1] += update;
        sy_f = (sy_f & mask) >> shift;
        lt_f = (lt_f & mask) >> shift;
        tot_f = (tot_f & mask) >> shift;
        rc.encodeFreq(sy_f, lt_f, tot_f);
        if (((tree[1] & 0xFFFF0000u) >> 16) >= max_prob) rescale();
    }
    void rescale() {                                                         // :137-166
        bool noEscape = true;
        uint32_t i;
        for (i = 0; i < numSyms - 1; i++) {
            uint32_t prob = tree[numSyms + i];
            if (prob & 0xFFFFu) { noEscape = false; continue; }
            prob = (prob & 0xFFFEFFFEu) >> 1;
            if (prob == 0) { prob = 1u; noEscape = false; }
            tree[numSyms + i] = prob;
        }
        uint32_t prob = tree[numSyms + i];
        prob = (prob & 0xFFFEFFFEu) >> 1;
        if (noEscape) prob = 0; else if (prob == 0) prob = 1u << 16;
        tree[numSyms + i] = prob;
        sum();
    }
};


struct RangeDec;
// DefSumModel (lib/DefSumModel.js:11-139): the model of BWTC levels 1-5.  prob[] / escape[] are cumulative.
struct DefSum {
    uint32_t numSyms, updateCount, updateThresh;
    uint16_t prob[304], escape[304], update[304];
    uint16_t probToSym[256], escProbToSym[304];
    bool dec;
    void init(uint32_t size, bool isDecoder) {                               // :11-37
        numSyms = size; dec = isDecoder;
        memset(prob, 0, sizeof prob); memset(escape, 0, sizeof escape); memset(update, 0, sizeof update);
        prob[size + 1] = 256;
        for (uint32_t i = 0; i <= size; i++) escape[i] = (uint16_t)i;
        updateCount = 0;
        updateThresh = 256 - 128;
        if (!dec) return;
        for (int i = 0; i < 256; i++) probToSym[i] = (uint16_t)size;
        for (uint32_t i = 0; i < size; i++) escProbToSym[i] = (uint16_t)i;
    }
    void upd(uint32_t symbol) {                                              // _update :41-97
        if (symbol == numSyms) {
            if (update[symbol] >= 40) return;
            if (updateCount >= updateThresh - 1) return;
        }
        update[symbol]++;
        updateCount++;
        if (updateCount < updateThresh) return;
        uint32_t cumProb = 0, cumEscProb = 0, odd = 0, i;
        escape[0] = 0; prob[0] = 0;
        for (i = 0; i < numSyms + 1; i++) {
            const uint32_t newProb = ((uint32_t)(prob[i + 1] - prob[i]) >> 1) + update[i];
            if (newProb) {
                prob[i] = (uint16_t)cumProb;
                cumProb += newProb;
                if (newProb & 1u) odd++;
                escape[i] = (uint16_t)cumEscProb;
            } else {
                prob[i] = (uint16_t)cumProb;
                escape[i] = (uint16_t)cumEscProb;
                cumEscProb++;
            }
        }
        prob[i] = (uint16_t)cumProb;
        updateThresh = 256 - (cumProb - odd) / 2;
        for (i = 0; i < numSyms + 1; i++) update[i] = 0;
        update[numSyms] = 1;
        updateCount = 1;
        if (!dec) return;
        uint32_t j = 0, k = 0;
        for (i = 0; i < numSyms + 1; i++) {
            const uint32_t probLimit = prob[i + 1];
            for (; j < probLimit && j < 256; j++) probToSym[j] = (uint16_t)i;
            if (i + 1 <= numSyms) {                                          // escape[numSyms+1] is past the reference's array
                const uint32_t escProbLimit = escape[i + 1];
                for (; k < escProbLimit && k < 304; k++) escProbToSym[k] = (uint16_t)i;
            }
        }
    }
    void encode(RangeEnc& rc, uint32_t symbol) {                             // :98-115
        uint32_t lt_f = prob[symbol];
        uint32_t sy_f = prob[symbol + 1] - lt_f;
        if (sy_f) { rc.encodeShift(sy_f, lt_f, 8); upd(symbol); return; }
        encode(rc, numSyms);
        lt_f = escape[symbol];
        sy_f = escape[symbol + 1] - lt_f;
        rc.encodeFreq(sy_f, lt_f, escape[numSyms]);
        upd(symbol);
    }
};

}  // namespace

// (tests) the range coder's division by reciprocal
extern "C" uint32_t cjs_dbg_rc_div(uint32_t range, uint32_t tot) { return rc_div(range, tot); }

struct bwtc_coder {
    Out out; RangeEnc rc; LogDistance len; uint32_t blockSize; int level;
};

extern "C" uint64_t bwtc_bound(uint64_t in_len) { return in_len + in_len / 4 + 4096; }

bwtc_coder* bwtc_begin(uint8_t* out, uint64_t cap, int64_t file_size, int level) {
    bwtc_coder* c = new bwtc_coder();
    c->out = Out{out, cap, 0, false};
    c->level = level;
    c->blockSize = (uint32_t)level * 100000u;
    // Util.compressFileHelper (lib/Util.js:105-142): magic, varint(size + 1) without its last byte
    const char* magic = "bwtc";
    for (int i = 0; i < 4; i++) c->out.put((uint8_t)magic[i]);
    uint8_t v[12]; int nv = 0;
    uint64_t n = (uint64_t)(file_size + 1);
    do { v[nv++] = (uint8_t)(n & 0x7F); n >>= 7; } while (n);                // writeUnsignedNumber :194-206
    v[0] |= 0x80;
    for (int i = nv - 1; i >= 1; i--) c->out.put(v[i]);
    c->rc.o = &c->out;
    c->rc.start(v[0], 1);                                                    // lib/BWTC.js:13-14
    c->rc.encodeByte((uint32_t)level);                                       // :20
    c->len.init(c->blockSize);                                               // :37-39
    return c;
}

// block header: short-block flag + length, primary index, used-symbol tree (lib/BWTC.js:42-79); returns alphabetSize
static uint32_t bwtc_block_header(bwtc_coder* c, uint32_t length, uint32_t pidx, const uint32_t* used8) {
    RangeEnc& rc = c->rc;
    if (length == c->blockSize) rc.encodeFreq(1, 0, 3);                      // :47-49
    else { rc.encodeFreq(1, 1, 3); c->len.encode(rc, length); }              // :51-53
    c->len.encode(rc, pidx);                                                 // :56
    uint16_t useTree[512];
    memset(useTree, 0, sizeof useTree);
    uint32_t alphabetSize = 0;
    for (int s = 0; s < 256; s++) if ((used8[s >> 5] >> (s & 31)) & 1u) { useTree[256 + s] = 1; alphabetSize++; }
    for (int i = 255; i > 0; i--) useTree[i] = (uint16_t)(useTree[2 * i] + useTree[2 * i + 1]);
    useTree[0] = 1;
    for (int i = 1; i < 512; i++) {                                          // :66-79
        const int parent = i >> 1;
        const int full = 1 << (9 - fls((uint32_t)i));
        if (useTree[parent] == 0 || useTree[parent] == full * 2) continue;
        if (i >= 256) rc.encodeBit(useTree[i]);
        else {
            const int v = useTree[i];
            rc.encodeFreq(1, v == 0 ? 0 : (v == full ? 2 : 1), 3);
        }
    }
    return alphabetSize;
}

// one block: `used8` = 256-bit set of byte values present, `sym`/`nsym` = RUNA(0)/RUNB(1)/index+1
// stream of the MTF'd BWT output (no end-of-block symbol), lib/BWTC.js:42-135
void bwtc_block(bwtc_coder* c, uint32_t length, uint32_t pidx, const uint32_t* used8, const uint16_t* sym, uint32_t nsym) {
    RangeEnc& rc = c->rc;
    const uint32_t alphabetSize = bwtc_block_header(c, length, pidx, used8);
    if (c->level <= 5) {                                                     // :107 `fast`
        static thread_local DefSum dm;
        dm.init(alphabetSize + 1, false);
        for (uint32_t i = 0; i < nsym; i++) dm.encode(rc, sym[i]);
        return;
    }
    Fenwick m;
    m.init(alphabetSize + 1, 0xFF00, 0x0100);                                // :105-106
    for (uint32_t i = 0; i < nsym; i++) m.encode(rc, sym[i]);                // :109-133
}

// levels 6..9 with the FenwickModel run on the GPU (k10_bwtc_model.hip): what is left is RangeCoder.encodeFreq
// (lib/RangeCoder.js:79-89) over the model's (sy_f, lt_f, tot_f) triples, in order
void bwtc_block_triples(bwtc_coder* c, uint32_t length, uint32_t pidx, const uint32_t* used8, const uint32_t* sylt,
                        const uint32_t* tot, uint32_t ntri) {
    RangeEnc& rc = c->rc;
    (void)bwtc_block_header(c, length, pidx, used8);
    for (uint32_t i = 0; i < ntri; i++) {
        if (i + 16u < ntri) __builtin_prefetch(&g_rc_recip[tot[i + 16u] & 0xFFFFu]);
        rc.encodeFreq(sylt[i] & 0xFFFFu, sylt[i] >> 16, tot[i]);
    }
}

int64_t bwtc_end(bwtc_coder* c) {
    c->rc.encodeFreq(1, 2, 3);                                               // "no more blocks" :137
    c->rc.finish();                                                          // :138
    const int64_t n = c->out.overflow ? -21 : (int64_t)c->out.n;
    delete c;
    return n;
}

// ---------------------------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------------------------
namespace {

struct RangeDec {                            // lib/RangeCoder.js:146-226 (EXTRA_BITS = 7)
    const uint8_t* p; uint64_t n, pos;
    uint32_t low, range, help; int32_t buffer;
    bool eof;
    int32_t readByte() { if (pos < n) return p[pos++]; eof = true; return -1; }
    void start() { buffer = readByte(); low = (uint32_t)buffer >> 1; range = 1u << 7; help = 0; }   // decodeStart(true)
    void normalize() {                       // :157-166
        while (range <= BOTTOM) {
            low = (low << 8) | (((uint32_t)buffer << 7) & 0xFFu);
            buffer = readByte();
            low |= (uint32_t)buffer >> 1;
            range <<= 8;
        }
    }
    uint32_t culFreq(uint32_t tot) {         // :173-178
        normalize();
        help = range / tot;
        if (help == 0) { eof = true; return 0; }
        const uint32_t tmp = low / help;
        return tmp >= tot ? tot - 1 : tmp;
    }
    uint32_t culShift(int shift) {           // :179-185
        normalize();
        help = range >> shift;
        if (help == 0) { eof = true; return 0; }
        const uint32_t tmp = low / help;
        return (tmp >> shift) ? (1u << shift) - 1u : tmp;
    }
    void update(uint32_t sy_f, uint32_t lt_f, uint32_t tot_f) {     // :192-200
        const uint32_t tmp = help * lt_f;
        low -= tmp;
        if (lt_f + sy_f < tot_f) range = help * sy_f; else range -= tmp;
    }
    uint32_t bit() { const uint32_t t = culShift(1); update(1, t, 2); return t; }           // :203-207
    uint32_t byte() { const uint32_t t = culShift(8); update(1, t, 256); return t; }        // :209-213
};

uint32_t nomodel_decode(RangeDec& rc, int bits) {                            // lib/NoModel.js:22-29
    uint32_t r = 0;
    for (int i = bits - 1; i >= 0; i--) r = (r << 1) | rc.bit();
    return r;
}

uint32_t logdist_decode(RangeDec& rc, int lgbits) {                          // lib/LogDistanceModel.js:37-44
    const uint32_t lg = nomodel_decode(rc, lgbits);
    if (lg < 2) return lg;
    if (lg > 32) { rc.eof = true; return 0; }
    return (1u << (lg - 1)) + nomodel_decode(rc, (int)lg - 1);
}

uint32_t fenwick_decode1(Fenwick& m, RangeDec& rc, bool isEscape) {          // lib/FenwickModel.js:88-128
    uint32_t mask = 0xFFFF0000u; int shift = 16;
    uint32_t update = m.increment << 16;
    if (isEscape) { mask = 0x0000FFFFu; update -= 1u; shift = 0; }
    const uint32_t tot_f = (m.tree[1] & mask) >> shift;
    if (tot_f == 0) { rc.eof = true; return 0; }
    const uint32_t prob = rc.culFreq(tot_f);
    uint32_t i = 1, lt_f = 0;
    while (i < m.numSyms) {
        m.tree[i] += update;
        const uint32_t leftProb = (m.tree[2 * i] & mask) >> shift;
        i *= 2;
        if ((prob - lt_f) >= leftProb) { lt_f += leftProb; i++; }
    }
    const uint32_t symbol = i - m.numSyms;
    const uint32_t sy_f = (m.tree[i] & mask) >> shift;
    m.tree[i] += update;
    rc.update(sy_f, lt_f, tot_f);
    if (symbol == m.numSyms - 1 && (m.tree[1] & 0xFFFFu) == 1u) {            // the last escape: zero it out
        update = 0u - m.tree[i];
        while (i >= 1) { m.tree[i] += update; i >>= 1; }
    }
    if (((m.tree[1] & 0xFFFF0000u) >> 16) >= m.max_prob) m.rescale();
    return symbol;
}
uint32_t defsum_decode(DefSum& m, RangeDec& rc) {                            // lib/DefSumModel.js:116-137
    uint32_t prob = rc.culShift(8);
    uint32_t symbol = m.probToSym[prob & 255u];
    uint32_t lt_f = m.prob[symbol];
    uint32_t sy_f = m.prob[symbol + 1] - lt_f;
    rc.update(sy_f, lt_f, 256);
    m.upd(symbol);
    if (symbol != m.numSyms) return symbol;
    const uint32_t tot_f = m.escape[m.numSyms];
    if (tot_f == 0) { rc.eof = true; return 0; }
    prob = rc.culFreq(tot_f);
    symbol = m.escProbToSym[prob < 304 ? prob : 303];
    lt_f = m.escape[symbol];
    sy_f = m.escape[symbol + 1] - lt_f;
    rc.update(sy_f, lt_f, tot_f);
    m.upd(symbol);
    return symbol;
}
uint32_t fenwick_decode(Fenwick& m, RangeDec& rc) {                          // :129-136
    uint32_t s = fenwick_decode1(m, rc, false);
    if (s == m.numSyms - 1) s = fenwick_decode1(m, rc, true);
    return s;
}

}  // namespace

int bwtc_decode(const uint8_t* in, uint64_t len, int64_t* declared_size, void* user, bwtc_block_fn on_block) {
    // Util.decompressFileHelper (lib/Util.js:143-166)
    const char* magic = "bwtc";
    for (int i = 0; i < 4; i++) if ((uint64_t)i >= len || in[i] != (uint8_t)magic[i]) return BWTC_E_MAGIC;
    uint64_t pos = 4;
    double n = 0;                                                            // readUnsignedNumber :211-220
    for (;;) {
        if (pos >= len) return BWTC_E_CORRUPT;
        const uint32_t c = in[pos++];
        if (c & 0x80u) { n += (double)(c & 0x7Fu); break; }
        n = (n + (double)c) * 128.0;
    }
    if (declared_size) *declared_size = (int64_t)n - 1;
    RangeDec rc;
    rc.p = in; rc.n = len; rc.pos = pos; rc.eof = false;
    rc.start();                                                              // lib/BWTC.js:142-143
    uint32_t blockSize = rc.byte();                                          // :144
    if (rc.eof || blockSize < 1 || blockSize > 9) return BWTC_E_CORRUPT;
    const bool fast = blockSize <= 5;                                        // :146 DefSumModel levels
    blockSize *= 100000u;
    const int lgbits = fls((uint32_t)(1 + fls(blockSize - 1)) - 1);          // LogDistanceModel(blockSize, 0, NoModel, NoModel)
    std::vector<uint8_t> b(blockSize);
    uint8_t M[256];
    for (;;) {
        const uint32_t ind = rc.culFreq(3);                                  // :158-159
        rc.update(1, ind, 3);
        if (rc.eof) return BWTC_E_CORRUPT;
        uint32_t length;
        if (ind == 0) length = blockSize;
        else if (ind == 1) length = logdist_decode(rc, lgbits);
        else break;                                                          // :166-167
        const uint32_t pidx = logdist_decode(rc, lgbits);                    // :170
        if (rc.eof || length > blockSize || pidx > length) return BWTC_E_CORRUPT;
        uint16_t useTree[512];                                               // :172-186
        useTree[0] = 1;
        for (int i = 1; i < 512; i++) {
            const int parent = i >> 1;
            const int full = 1 << (9 - fls((uint32_t)i));
            if (useTree[parent] == 0 || useTree[parent] == full * 2) useTree[i] = useTree[parent] >> 1;
            else if (i >= 256) useTree[i] = (uint16_t)rc.bit();
            else {
                const uint32_t v = rc.culFreq(3);
                rc.update(1, v, 3);
                useTree[i] = (uint16_t)(v == 2 ? full : v);
            }
        }
        uint32_t alphabetSize = 0;                                           // :188-193
        for (int i = 0; i < 256; i++) if (useTree[256 + i]) M[alphabetSize++] = (uint8_t)i;
        if (rc.eof) return BWTC_E_CORRUPT;
        Fenwick model;
        static thread_local DefSum dmodel;
        if (fast) dmodel.init(alphabetSize + 1, true);                       // :198
        else model.init(alphabetSize + 1, 0xFF00, 0x0100);                   // :196-197
        uint64_t val = 1;
        for (uint64_t i = 0; i < length;) {                                  // :200-212
            const uint32_t c = fast ? defsum_decode(dmodel, rc) : fenwick_decode(model, rc);
            if (rc.eof) return BWTC_E_CORRUPT;
            if (c <= 1) {
                const uint64_t cnt = val * (c + 1);
                if (i + cnt > length) return BWTC_E_CORRUPT;                 // the reference would write past the block
                memset(b.data() + i, 0, cnt);
                i += cnt;
                val *= 2;
            } else {
                val = 1;
                if (c - 1 >= alphabetSize) return BWTC_E_CORRUPT;
                b[i++] = (uint8_t)(c - 1);
            }
        }
        for (uint32_t i = 0; i < length; i++) {                              // MTF decode :214-222
            uint32_t j = b[i];
            if (j >= alphabetSize) return BWTC_E_CORRUPT;
            const uint8_t c = M[j];
            b[i] = c;
            for (; j > 0; j--) M[j] = M[j - 1];
            M[0] = c;
        }
        const int rc2 = on_block(user, b.data(), length, pidx);
        if (rc2) return rc2;
    }
    rc.normalize();                                                          // decodeFinish :216-219
    return 0;
}
