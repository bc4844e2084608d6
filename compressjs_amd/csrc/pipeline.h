// Device-side view of one batch of bzip2 blocks moving through K0..K5.
#pragma once
#include "cjs_common.h"
#include "devutil.h"
#include "k1_bwt.h"

#define K2_SEG 1024          // run heads per MTF segment (one wave)
#define K2_FREQ_PITCH 260    // u32 per block in freq[]
#define CJS_MAX_SYMS 258     // lib/Bzip2.js:41
#define CJS_MAX_GROUPS 6     // lib/Bzip2.js:45
#define CJS_GROUP 50         // lib/Bzip2.js:46
#define CJS_MAX_BITS 20      // lib/Bzip2.js:40
#define CJS_LEN_PITCH 264    // bytes per table in lens[], u32 per table in codes[]

#define K5_NONE (-0x40000000)
#define K5_TGROUPS 80u       // groups of 50 symbols per K5 tile: a tile's code bits are a sum of the optimiser's group costs
#define K5_TILE (K5_TGROUPS * CJS_GROUP)
#define K5_TILES(g) ((g).stride / K5_TILE + 2u)              // >= ceil((symbols of a block + EOB) / K5_TILE)
#define K5_HDR_WORDS 6144     // >= worst-case header: 18001 selectors x 6 bits + 6 tables x 10067 bits

struct StreamState {
    u64 bits;          // bits written so far (next block starts here)
    u32 crc;           // combined CRC so far (lib/Bzip2.js:917)
    u32 overflow;      // set when the stream would not fit outCapBytes
};

struct Pipe {
    BatchGeom g;
    u32 segs;          // stride / K2_SEG
    // ---- block text (K0 output / K1 input)
    u8* T;             // [nb][tstride]  T_ext
    u32* nlen;         // [nb]
    u32* crc;          // [nb]           block CRCs
    // ---- K1
    K1Buf k1;
    u8* U;             // [nb][stride]
    u32* pidx;         // [nb]
    // ---- K2
    u32* used;         // [nb][8]        256-bit used-symbol set
    u32* alpha;        // [nb]           alphabetSize
    u32* nruns;        // [nb]
    u32* tileCnt;      // [nb][rtiles]
    u32* symCnt;       // [nb][rtiles]
    u8* RHsym;         // [nb][stride]
    u32* RHpos;        // [nb][stride+1]
    int* Ltab;         // [nb][segs][256]
    u8* J;             // [nb][stride]
    u16* A;            // [nb][stride]   MTF/RLE2 symbols incl. EOB
    u32* pos;          // [nb]           number of symbols in A
    u32* freq;         // [nb][K2_FREQ_PITCH]
    // ---- K3/K4
    u8* sel;           // [nb][selPitch] selectors
    u32 selPitch;
    u8* lens;          // [nb][6][CJS_LEN_PITCH]
    u32* codes;        // [nb][6][CJS_LEN_PITCH]
    u32* ngroups;      // [nb]
    u32* nsel;         // [nb]
    u16* selCost;      // [nb][selPitch] best cost of every 50-symbol group (optimiser scratch)
    u32* fr2;          // [nb][2][6][CJS_LEN_PITCH] frequency rows of the optimiser's tables, two sets used alternately
    // ---- K5
    u32* hdr;          // [nb][K5_HDR_WORDS] block header bits (magic .. code-length tables)
    u32* hbits;        // [nb]           header length in bits
    u32* tileBits;     // [nb][K5_TILES] code bits per tile of K5_TILE symbols -> exclusive offsets
    u64* bitlen;       // [nb]           bits of the encoded block
    u64* bitoff;       // [nb]           absolute bit offset of the block in the stream
    StreamState* ss;   // running stream state (bit cursor, combined CRC), device resident
    u32* out;          // the .bz2 stream being assembled (byte order = memory order)
    u64 outCapBytes;
    u64* snap;         // optional (pinned host memory): k5_blockscan leaves the bit cursor behind this batch's blocks here - everything before it is final once the batch is packed
};

// K0 (whole-input pre-pass) ----------------------------------------------------------------------
struct K0Buf {
    const u8* in;      // input bytes (device)
    u64 in_len;
    u64 ntiles;        // 4096-byte tiles
    u64 nchunks;       // scan chunks of 1024 tiles
    u32 maxBlocks;
    u64* tileA;        // [ntiles+2] last run boundary (+1) before the tile   (exclusive max-scan)
    u64* tileB;        // [ntiles+2] last run boundary (+1) inside the tile
    u64* tileC;        // [ntiles+2] C(4096 t): RLE1 output bytes before the tile with uncut runs; [ntiles] = total
    u64* chunk;        // [nchunks+1] scan scratch
    u64* blkStart;     // [maxBlocks] first input byte of block k
    u64* blkEnd;       // [maxBlocks] one past its last input byte
    u64* blkAdj;       // [maxBlocks] output position of input byte i >= blkRe is C(i) - blkAdj
    u64* blkRe;        // [maxBlocks] end of the run the block start cuts (== blkStart if none)
    u32* blkN;         // [maxBlocks] block length after RLE1
    u32* nBlocks;      // [1]
    u64* specEnd;      // [maxBlocks+2] speculative chain: boundary j = min{ i : C(i) >= j*cap }
    u64* specC;        // [maxBlocks+2] C at that boundary
    u64* specBad;      // [1] first boundary where the speculation does not hold (UINT64_MAX: none)
};
size_t k0_bytes(u64 in_len, u32 cap);
void k0_carve(K0Buf& K, const u8* d_in, u64 in_len, u32 cap, void* ws);
int k0_prepass(K0Buf K, u32 cap, hipStream_t stream);
// multi-GPU parallel plan (see k0_rle1.hip): scans only; C at a position (-> K.specC[0]); the blocks that start in [0, own_len)
#define K0_PHASE_FAIL 0xFFFFFFFFu
int k0_scans(K0Buf K, hipStream_t stream);
int k0_eval(K0Buf K, u64 pos, hipStream_t stream);
int k0_phase_plan(K0Buf K, u32 cap, u64 t0, u64 own_len, u32 last, u64 total, hipStream_t stream);   // t0: target of the slice's first boundary (round 6: carried from slice to slice; *K.nBlocks, ((u64*)K.nBlocks)[1] = blocks, the next slice's target)

int k6_unbwt_linear(const u8* dT, u8* dU, u32 n, u32 pidx, void* ws, hipStream_t stream);
int k2_run(Pipe P, u32 max_n, hipStream_t stream);
// BWTC levels 6..9: FenwickModel of every block on the GPU -> (sy_f | lt_f << 16, tot_f) per encodeFreq call, [nb][stride] each
int k10_model_run(Pipe P, u32* sylt, u32* tot, u32* ntri, u32 ostride, u32 ocap, hipStream_t stream);
int k34_run(Pipe P, hipStream_t stream);
int k3_alloc_lengths_run(long long* d_arr, const u32* d_off, u32 count, int maxlen, hipStream_t stream);
int k5_run(Pipe P, u32 max_n, hipStream_t stream, hipEvent_t after = nullptr, hipEvent_t done = nullptr, hipEvent_t crc_ready = nullptr);   // crc_ready: the block CRCs were computed on another stream (k0_batch)
int k5_stream_begin(Pipe P, int level, hipStream_t stream, bool zero = true);   // zero = false: the caller has zeroed P.out (on another stream, next to the pre-pass)
int k5_stream_end(Pipe P, hipStream_t stream);
int k5_shift_bits_run(const u8* d_in, u64 nbytes, u32 s, u8* d_out, hipStream_t stream);
int k0_batch(K0Buf K, Pipe P, u32 first_block, u32 cap, hipStream_t stream, u32 crc_parts, hipStream_t side = nullptr, hipEvent_t ev_pad = nullptr, hipEvent_t ev_crc = nullptr);   // crc_parts: workgroups per block of k0_crc (1 .. 16)
size_t pipe_bytes(const BatchGeom& g);
void pipe_carve(Pipe& P, const BatchGeom& g, void* base);
int pipe_run_block_stages(Pipe& P, u32 max_n, hipStream_t stream, int upto, hipEvent_t after = nullptr,
                          hipEvent_t done = nullptr);
