/* compressjs_amd.h -- C ABI of libcompressjs_amd.so (MI355X-native bzip2 block pipeline).
 * (state: round 5; the full entry-point list with the reference binding of each is in INTEGRATION.md)
 */
#ifndef COMPRESSJS_AMD_H
#define COMPRESSJS_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Opaque per-GPU context: one HIP stream + the HBM workspace for `batch_blocks` bzip2 blocks in
 * flight (about 31 B per block byte; 128 blocks of 900 kB = 3.6 GB).  Not thread safe; one
 * context per thread/GPU.  Returns NULL when no MI355X is visible (there is no CPU path). */
typedef struct cjs_ctx cjs_ctx;
cjs_ctx* cjs_create(int device, uint32_t batch_blocks);
void cjs_destroy(cjs_ctx* ctx);

/* Upper bound of the .bz2 size for in_len input bytes (any level). */
int64_t cjs_bz2_compress_bound(uint64_t in_len);

/* = Bzip2.compressFile(input, null, level)            (reference: lib/Bzip2.js:879-929)
 * Host buffers in, complete .bz2 stream out.  Returns the number of bytes written, or < 0:
 *   CJS_E_LEVEL (-20) 'Invalid block size multiplier' (lib/Bzip2.js:888-890),
 *   CJS_E_NOSPACE (-21) out_cap too small, CJS_E_ARG (-22), CJS_E_NOGPU (-23), -100-hipError_t. */
int64_t cjs_bz2_compress(cjs_ctx* ctx, const uint8_t* in, uint64_t in_len, int level, uint8_t* out,
                         uint64_t out_cap);
/* The same with input and output resident in HBM (device pointers, d_out 4-byte aligned). */
int64_t cjs_bz2_compress_device(cjs_ctx* ctx, const void* d_in, uint64_t in_len, int level, void* d_out,
                                uint64_t out_cap);
/* = BWTC.compressFile(input, null, level) for level 6..9   (reference: lib/BWTC.js:12-139,
 * lib/Util.js:105-142).  GPU: BWT.bwtransform + MTF + RLE2 per 100000*level-byte block; host: the
 * adaptive Fenwick model + range coder (serial by construction: lib/RangeCoder.js, lib/FenwickModel.js).
 * declared_size = the size written into the header (input length, or -1 for streams of unknown
 * size, lib/Util.js:119-124).  Levels 1-5 use DefSumModel (lib/DefSumModel.js), 6-9 FenwickModel. */
int64_t cjs_bwtc_compress_bound(uint64_t in_len);
int64_t cjs_bwtc_compress(cjs_ctx* ctx, const uint8_t* in, uint64_t in_len, int level, uint8_t* out,
                          uint64_t out_cap, int64_t declared_size);
#define CJS_E_UNSUPPORTED (-24)
/* Phases of the last cjs_bwtc_compress (bench.py's BWTC line): out5[0] = ms of the first K10 launch (FenwickModel on the GPU),
 * [1] = ms until every (sy_f, lt_f, tot_f) triple was on the host, [2] = ms the serial range coder (lib/RangeCoder.js:79-89) was busy,
 * [3] = ms of the whole call, [4] = encodeFreq calls. */
int cjs_bwtc_last_times(cjs_ctx* ctx, float* out5);

/* Sharded encoding for multi-GPU runs (blocks are independent once the RLE1 split is known):
 * cjs_bz2_plan   = the readBlock chain of lib/Bzip2.js:913-922 over the whole (device) input;
 *                  returns the number of blocks and keeps the split in the context, in a workspace of
 *                  its own: it stays valid across other calls on the context until the next
 *                  cjs_bz2_plan.  d_in must stay alive and unmodified until the last
 *                  cjs_bz2_encode_blocks that uses the plan (the plan points into it).
 * cjs_bz2_encode_blocks = compressBlock (lib/Bzip2.js:735-876) for blocks [first, first+count):
 *                  bare bit stream in d_seg starting at bit 0 (no "BZh" header, no trailer);
 *                  returns its length in BITS; *crc_fold = XOR_i rotl^(count-1-i)(blockCRC_i).
 *                  count is clamped to the blocks that exist (0xFFFFFFFF = "all remaining"). */
int64_t cjs_bz2_plan(cjs_ctx* ctx, const void* d_in, uint64_t in_len, int level);
/* first input byte, relative to the planned input, of block k of the current plan (k == block count: its length).  A
 * driver that holds only a slice of the stream plans from a block start, drops the last, incomplete block and tells the
 * next slice where that block started (compressjs_amd/dist.py, cjs_bz2_compress_multi). */
int64_t cjs_bz2_plan_block_start(cjs_ctx* ctx, uint32_t k);
/* Parallel plan of one slice of a longer stream (round 3; replaces the rank-to-rank chain of lib/Bzip2.js:913-922's serial
 * `do { readBlock } while` across GPUs).  With G(i) the RLE1 output length of stream bytes [0, i) under uncut runs, block k of
 * the stream starts where G reaches k * cap (cap = level * 100000 - 19, lib/Bzip2.js:636-667) unless that boundary falls
 * into a run of four or more equal bytes.  A rank that holds stream bytes [lo, lo + in_len) - its own slice [lo, lo + own_len)
 * and a margin of what follows - learns G(lo) from one all_gather of per-slice totals and boundary runs (compressjs_amd/dist.py):
 *   cjs_bz2_plan_scan   K0's scans over d_in; returns the input's own cost total;
 *   cjs_bz2_plan_cost   the input's own cost prefix at byte pos (pos = own_len: the slice's total);
 *   cjs_bz2_plan_phase  plans the blocks that START in [0, own_len), boundaries where the own prefix reaches phase + m * cap
 *                       (phase = (-(G(lo) + head-run correction)) mod cap; last != 0: d_in ends where the stream ends, the final
 *                       block may be short or absent, lib/Bzip2.js:916,922).  Returns their number - they are blocks 0 .. n-1
 *                       for cjs_bz2_encode_blocks - or CJS_E_SPEC when the slice cannot be planned on its own (a boundary in a
 *                       long run, a block longer than the margin): the caller falls back to cjs_bz2_plan on more of the stream.
 *   cjs_bz2_plan_chain  (round 6) the same plan as a link of a chain: t0 = the value the own prefix reaches at the slice's FIRST
 *                       block boundary (the slice before it hands it on; phase when nothing in front of the slice moved a
 *                       boundary).  A boundary inside a run of four or more equal bytes - ordinary text has them - restarts the
 *                       run in the new block (lib/Bzip2.js:636-667) and moves every later boundary by a few bytes: the slice goes
 *                       on serially from there instead of refusing, and *t_next = the target of the first boundary at or beyond
 *                       own_len carries the shift to the slices behind it (the next slice's t0 = *t_next + G(lo) - G(lo') under
 *                       its own origin, not below 0).  CJS_E_SPEC only for a block longer than the margin or a run that fills a
 *                       block. */
#define CJS_E_SPEC (-25)
int64_t cjs_bz2_plan_scan(cjs_ctx* ctx, const void* d_in, uint64_t in_len, int level);
int64_t cjs_bz2_plan_cost(cjs_ctx* ctx, uint64_t pos);
int64_t cjs_bz2_plan_phase(cjs_ctx* ctx, uint64_t own_len, uint64_t phase, int last);
int64_t cjs_bz2_plan_chain(cjs_ctx* ctx, uint64_t own_len, uint64_t t0, int last, uint64_t* t_next);
int64_t cjs_bz2_encode_blocks(cjs_ctx* ctx, uint32_t first, uint32_t count, void* d_seg,
                              uint64_t seg_cap, uint32_t* crc_fold, uint32_t* n_done);
/* Device time (HIP events on the context's stream) and block count of the last compress call. */
float cjs_last_device_ms(const cjs_ctx* ctx);
uint32_t cjs_last_block_count(const cjs_ctx* ctx);
void* cjs_stream(const cjs_ctx* ctx);   /* the hipStream_t the library launches on */
/* HIP-event timing of every launch of the suffix sort's main kernels (event pairs on the library's own streams), for
 * bench.py's roofline leg - which names the kernel with the largest total for the workload at hand.  Classes: 0 k1f_bsort
 * (in-LDS bucket sort of the sample-sort front end), 1 k1r_round (text refinement rounds), 2 k1d_build, 3 k1d_round,
 * 4 k1d_med, 5 k1d_large, 6 k1d_update (prefix-doubling stage), 7 k1f_task.  `elements`: rotations (classes 0, 2), list
 * entries walked (3, 6: read back from the last sub-batch's counters - exact for a one-stream pass over equal batches).
 * cjs_profile_enable(ctx, 1) clears the records; cjs_profile_read = class 0.  Test / bench instrumentation: no reference
 * counterpart. */
int32_t cjs_profile_enable(cjs_ctx* ctx, int on);
int32_t cjs_profile_read_class(cjs_ctx* ctx, uint32_t cls, float* total_ms, uint32_t* launches, uint64_t* elements);
int32_t cjs_profile_read(cjs_ctx* ctx, float* total_ms, uint32_t* launches, uint64_t* elements);

#define CJS_E_LEVEL (-20)
#define CJS_E_NOSPACE (-21)
#define CJS_E_ARG (-22)
#define CJS_E_NOGPU (-23)

/* The four BWT.* entry points below handle ONE block per call with n <= 2^22 - 1 = 4 194 303 bytes (CJS_E_ARG
 * above that: ranks are packed into 22 bits by the refinement kernels; bzip2 blocks are <= 900 000 and the
 * reference's own tests go up to test/sample5.ref = 2 130 640).  The reference has no such limit; js/index.js
 * hands larger n to the reference package when it is installed. */
/* = BWT.bwtransform2(T, U, n, 256) -> pidx            (reference: lib/BWT.js:372-417) */
int32_t cjs_bwt_cyclic(const uint8_t* T, uint8_t* U, uint32_t n, uint32_t* pidx);
/* = BWT.bwtransform(T, U, A, n, 256) -> pidx           (reference: lib/BWT.js:328-350, :153-192) */
int32_t cjs_bwt_linear(const uint8_t* T, uint8_t* U, uint32_t n, uint32_t* pidx);
/* = BWT.suffixsort(T, SA, n, 256)                      (reference: lib/BWT.js:305-321) */
int32_t cjs_suffixsort(const uint8_t* T, int32_t* SA, uint32_t n);
/* = BWT.unbwtransform(T, U, LF, n, pidx)                (reference: lib/BWT.js:352-363); list ranking */
int32_t cjs_unbwt_linear(const uint8_t* T, uint8_t* U, uint32_t n, uint32_t pidx);

/* = require('compressjs/lib/HuffmanAllocator').allocateHuffmanCodeLengths(array, maxLength)
 *   (reference: lib/HuffmanAllocator.js:199-222, exercised by test/huffman.js): in place, ascending
 *   weights in, code lengths out.  64-bit cells carry any integer weight a JS caller can pass.
 *   The batch form runs `count` independent arrays, array k = arr[off[k] .. off[k+1]). */
int32_t cjs_huff_lengths(int64_t* arr, uint32_t n, uint32_t max_len);
int32_t cjs_huff_lengths_batch(int64_t* arr, const uint32_t* off, uint32_t count, uint32_t max_len);
/* cjs_bwt_cyclic for nb independent blocks laid out at a fixed pitch `cap` (host pointers) */
int32_t cjs_bwt_cyclic_batch(const uint8_t* T, const uint32_t* nlen, uint32_t nb, uint32_t cap,
                             uint8_t* U, uint32_t* pidx);

/* ---- debug/test entry points (not part of the drop-in surface) ------------------------------ */
typedef struct cjs_dbg_stage_out {   /* every pointer may be NULL; pitches in elements */
    uint8_t* U;          /* [nb][cap]        BWT output                           */
    uint32_t* pidx;      /* [nb]                                                   */
    uint16_t* A;         /* [nb][cap+1]      MTF/RLE2 symbols incl. EOB            */
    uint32_t* pos;       /* [nb]                                                   */
    uint32_t* alpha;     /* [nb]             alphabetSize                          */
    uint32_t* freq;      /* [nb][258]                                              */
    uint32_t* used;      /* [nb][8]          256-bit used-symbol set               */
    uint8_t* sel;        /* [nb][(cap+1)/50+2]                                     */
    uint8_t* lens;       /* [nb][6][258]                                           */
    uint32_t* ngroups;   /* [nb]                                                   */
    uint32_t* nsel;      /* [nb]                                                   */
    uint64_t* bitlen;    /* [nb]                                                   */
    const uint32_t* crc_in; /* [nb] block CRCs to put in the headers (K0 is bypassed here) */
    int32_t level;       /* level digit for the stream header                      */
    uint8_t* stream;     /* whole .bz2 stream of the batch (header, blocks, trailer) */
    uint64_t stream_cap;
    uint64_t stream_bytes; /* out */
} cjs_dbg_stage_out;
int32_t cjs_dbg_bwt_batch_time(const uint8_t* T, const uint32_t* nlen, uint32_t nb, uint32_t cap,
                               uint8_t* U, uint32_t* pidx, int reps, float* ms);
int cjs_dbg_k1_sparse_rounds(void);   /* rounds of the last K1 run that used the sparse phase */
int cjs_dbg_k1_rounds(void);
uint32_t cjs_dbg_rc_div(uint32_t range, uint32_t tot);   /* bwtc_host.hip: the range coder's range / tot by reciprocal (boundary test in tests/test_host_api.py) */
int cjs_dbg_multi_replans(void);      /* segments of cjs_bz2_compress_multi planned a second time: the chain carried a boundary shift to them (round 6: such calls used to take the replicated plan) */
int cjs_dbg_multi_fallbacks(void);    /* calls of cjs_bz2_compress_multi that took the REPLICATED plan (a segment that cannot be planned on its own: every device plans the whole input, encodes its share) */
int cjs_dbg_multi_mallocs(void);      /* hipMalloc calls cjs_bz2_compress_multi has made for its per-device segment buffers (grow-only pools: none after warm-up) */
int cjs_dbg_k1_periodic_blocks(void); /* k1_period.hip, last K1 run (counted under CJS_K1_TRACE only): blocks with a period <= 64 (closed form) | blocks sorted through a reduced block << 16 */
int32_t cjs_dbg_block_stages(const uint8_t* T, const uint32_t* nlen, uint32_t nb, uint32_t cap,
                             int upto, cjs_dbg_stage_out* out);

/* ---- decoder --------------------------------------------------------
 * = Bzip2.decompressFile(input, output, multistream) = Bunzip.decode   (reference: lib/Bzip2.js:454-481,
 *   _start_bunzip :137-152, _get_next_block :153-398, _read_bunzip :405-448).
 * Returns the decoded size or a negative code: the reference's own Err values -2 NOT_BZIP_DATA,
 * -5 DATA_ERROR, -7 OBSOLETE_INPUT (lib/Bzip2.js:62-72; which one, and when, follows the reference's
 * sequential order), or -21 when `out_cap` is too small (the result stays in HBM: cjs_bz2_last_size /
 * cjs_bz2_fetch), -22/-23/-100-e as above.  cjs_bz2_last_detail gives the optDetail of the error. */
int64_t cjs_bz2_decompress(cjs_ctx* ctx, const uint8_t* in, uint64_t in_len, uint8_t* out, uint64_t out_cap,
                           int multistream);
/* the same with the stream and the output resident in HBM */
int64_t cjs_bz2_decompress_device(cjs_ctx* ctx, const uint8_t* d_in, uint64_t in_len, uint8_t* d_out,
                                  uint64_t out_cap, int multistream);
/* = Bzip2.decompressBlock(input, bitPos, output) = Bunzip.decodeBlock   (reference: lib/Bzip2.js:482-503) */
int64_t cjs_bz2_decompress_block(cjs_ctx* ctx, const uint8_t* in, uint64_t in_len, uint64_t bitpos, uint8_t* out,
                                 uint64_t out_cap);
/* = Bzip2.table(input, callback, multistream)   (reference: lib/Bzip2.js:508-548): positions[i] (bits)
 *   and sizes[i] (decoded bytes) of block i, at most `cap` of them; returns the number of blocks */
int64_t cjs_bz2_table(cjs_ctx* ctx, const uint8_t* in, uint64_t in_len, int multistream, uint64_t* positions,
                      uint64_t* sizes, uint32_t cap);
int64_t cjs_bz2_last_size(cjs_ctx* ctx);
int64_t cjs_bz2_fetch(cjs_ctx* ctx, uint8_t* out, uint64_t out_cap);
int32_t cjs_bz2_last_detail(cjs_ctx* ctx, uint32_t* crc_got, uint32_t* crc_expected);
float cjs_bz2_last_decode_ms(cjs_ctx* ctx);

/* = BWTC.decompressFile(input, output)   (reference: lib/BWTC.js:141-233 via Util.decompressFileHelper
 *   lib/Util.js:143-166; decoder sides of lib/RangeCoder.js:146-226, lib/FenwickModel.js:88-136,
 *   lib/LogDistanceModel.js:37-44, lib/NoModel.js:22-29).  Range decoder on the host (serial by
 *   construction), BWT.unbwtransform of every block on the GPU.  Returns the decoded size, -30 'Bad
 *   magic', -31 corrupt/truncated stream (undefined behaviour in the reference),
 *   -21 when out_cap is too small (then cjs_bwtc_last_size / cjs_bwtc_fetch).  *declared_size gets the
 *   size recorded in the header (-1: unknown). */
int64_t cjs_bwtc_decompress(cjs_ctx* ctx, const uint8_t* in, uint64_t in_len, uint8_t* out, uint64_t out_cap,
                            int64_t* declared_size);
int64_t cjs_bwtc_last_size(cjs_ctx* ctx);
int64_t cjs_bwtc_fetch(cjs_ctx* ctx, uint8_t* out, uint64_t out_cap);

/* Workload helper (not a compression entry point): bytes [first, first + n) of LCG(N, seed) - SURVEY.md 8(c), BASELINE.json
 * configs[3] "random printable ASCII" - written straight into HBM, so that every GPU of a multi-GPU run fills its own slice. */
int32_t cjs_lcg_ascii_device(cjs_ctx* ctx, uint8_t* d_out, uint64_t n, uint32_t seed, uint64_t first);
/* HIP devices visible to the process (0: none; the product has no CPU path). */
int32_t cjs_device_count(void);
/* = Bzip2.compressFile over SEVERAL GPUs of one node from one process (the N-API addon's path to N
 * devices, the reference's single entry point lib/Bzip2.js:879 staying the only call): ctxs[i] = contexts created
 * on different devices (the same device twice is allowed: tests).  The input is cut into segments of about one batch of
 * blocks, segment k is uploaded to, planned and encoded on device k mod n, the segments' bit streams are shifted to
 * their offsets on the devices and copied side by side into `out` (parallel D2H); same bytes as cjs_bz2_compress. */
int64_t cjs_bz2_compress_multi(cjs_ctx** ctxs, uint32_t n, const uint8_t* in, uint64_t in_len, int level,
                               uint8_t* out, uint64_t out_cap);

/* multi-GPU seam helper: d_out[0 .. nbytes] = d_in[0 .. nbytes) shifted right by s (0..7) bits (MSB first).
 * Used when a rank's bit-0-aligned segment (cjs_bz2_encode_blocks) is placed at its offset in the stream. */
int32_t cjs_shift_bits(cjs_ctx* ctx, const uint8_t* d_in, uint64_t nbytes, uint32_t s, uint8_t* d_out);

#ifdef __cplusplus
}
#endif
#endif
