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
#include "bwtc_host.h"
#include <string.h>

namespace {

struct Out {
    uint8_t* p; uint64_t cap, n; bool overflow;
    void put(uint32_t b) { if (n < cap) p[n] = (uint8_t)b; else overflow = true; n++; }
};

const uint32_t TOP = 0x80000000u;            // Top_value   lib/RangeCoder.js:15
const uint32_t BOTTOM = TOP >> 8;            // Bottom_value :18
const int SHIFT_BITS = 23;                   // :16

struct RangeEnc {                            // lib/RangeCoder.js:27-140
    uint32_t low, range, buffer, help, bytecount;
    Out* o;
    void start(uint32_t c, uint32_t initlength) { low = 0; range = TOP; buffer = c; help = 0; bytecount = initlength; }
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
        const uint32_t r = range / tot_f;
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

}  // namespace

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

// one block: `used8` = 256-bit set of byte values present, `sym`/`nsym` = RUNA(0)/RUNB(1)/index+1
// stream of the MTF'd BWT output (no end-of-block symbol), lib/BWTC.js:42-135
void bwtc_block(bwtc_coder* c, uint32_t length, uint32_t pidx, const uint32_t* used8, const uint16_t* sym, uint32_t nsym) {
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
    Fenwick m;
    m.init(alphabetSize + 1, 0xFF00, 0x0100);                                // :105-106
    for (uint32_t i = 0; i < nsym; i++) m.encode(rc, sym[i]);                // :109-133
}

int64_t bwtc_end(bwtc_coder* c) {
    c->rc.encodeFreq(1, 2, 3);                                               // "no more blocks" :137
    c->rc.finish();                                                          // :138
    const int64_t n = c->out.overflow ? -21 : (int64_t)c->out.n;
    delete c;
    return n;
}
