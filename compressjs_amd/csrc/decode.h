// GPU bzip2 decoder (Bunzip, lib/Bzip2.js:91-548): shared declarations of K7 (entropy decode),
// K8 (inverse BWT by splitter list ranking) and K9 (un-RLE1 + CRC).
#pragma once
#include "cjs_common.h"
#include "devutil.h"

#define DEC_CAP 900000u            // largest dbufSize (level 9, lib/Bzip2.js:151)
#define DEC_TILE 4096u
#define DEC_TILES ((DEC_CAP + DEC_TILE - 1) / DEC_TILE)      // 220
#define DEC_STRIDE (DEC_TILES * DEC_TILE + DEC_TILE)          // per-slot element stride (guard tile)
#define DEC_SPLIT 128u             // one splitter every DEC_SPLIT positions of the T vector
#define DEC_MAXSPL (DEC_CAP / DEC_SPLIT + 3)                  // splitters per block incl. origPtr

// Err codes of the reference (lib/Bzip2.js:62-72), returned negated through the ABI as they are
#define DEC_OK 0
#define DEC_NOT_BZIP (-2)
#define DEC_DATA_ERROR (-5)
#define DEC_OBSOLETE (-7)
// detail of the last decoder error (the optDetail strings of _throw)
#define DEC_DETAIL_NONE 0
#define DEC_DETAIL_BAD_MAGIC 1
#define DEC_DETAIL_LEVEL 2
#define DEC_DETAIL_ORIGPTR 3
#define DEC_DETAIL_BLOCK_CRC 4
#define DEC_DETAIL_STREAM_CRC 5

struct DecResult {
    u64 endbit;      // first bit after the block's end-of-block symbol
    int status;      // DEC_OK or an Err code
    u32 n;           // dbufCount
    u32 origPtr;
    u32 crc;         // targetBlockCRC
    u64 cycles;      // shader clocks k7_decode spent on the block (s_memtime)
    u64 symbols;     // Huffman symbols decoded
    u64 pwait, cwait; // clocks the boundary wave / the symbol wave spent waiting for each other
    u64 prof[14];     // -DK7_PROF builds: clocks per section of the four waves (k7_unbz2.hip)
};

struct DecBuf {
    const u32* in32;     // stream, zero padded, 4-byte aligned
    u64 zeroChunk;       // index of a 256-byte chunk that lies entirely in the zero padding
    const u64* cand;     // [slots] (bit position of a block magic) << 1
    u8* tt;              // [slots][ttStride]   BWT last column (dbuf low bytes)
    u32 ttStride;
    DecResult* res;      // [slots]
    u32* sel;            // [slots][4160]  selectors of the block being decoded, 4 bits each (k7 scratch)
    // K8 / K9, indexed by slot
    u32* word;           // [slots][DEC_STRIDE]  (T[p] << 8) | F[p]
    u32* tileHist;       // [slots][DEC_TILES][256]
    u32* splSucc;        // [slots][DEC_MAXSPL]  next splitter id
    u32* splLen;         // [slots][DEC_MAXSPL]  nodes owned by the splitter
    u32* splOff;         // [slots][DEC_MAXSPL]  output index of the splitter's first node
    u32* flags;          // [slots] bit0: the T vector is not one cycle -> serial walk
    u8* pre;             // [slots][DEC_STRIDE]  block before un-RLE1
    u32* tileFn;         // [slots][DEC_TILES]   K9: composite run-state function of a tile
    u8* tileState;       // [slots][DEC_TILES]   K9: run state entering a tile
    u32* isCount;        // [slots][DEC_STRIDE/32] K9: bit k set = byte k is an RLE1 count byte
    u32* tileLen;        // [slots][DEC_TILES]   K9: decoded bytes of a tile, then exclusive offsets
    u32* blkOut;         // [slots]              decoded bytes of the block
    // per launch set
    const u32* slotOf;   // [nvalid] slot of the k-th valid block
    const u64* outOff;   // [nvalid] offset of its decoded bytes in `out`
    u8* out;
    u32* crcOut;         // [nvalid] CRC of the decoded bytes
};

int k7_scan(const u8* d_in, u64 len, u64 first_bit, u64* d_cand, u32* d_ncand, u32 cap, hipStream_t stream);
int k7_run(DecBuf D, u32 first, u32 count, hipStream_t stream);
int k8_run(DecBuf D, u32 nvalid, hipStream_t stream);
int k9_sizes(DecBuf D, u32 nvalid, hipStream_t stream);
int k9_expand(DecBuf D, u32 nvalid, hipStream_t stream);
