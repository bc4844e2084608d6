/* compressjs_amd.h -- C ABI of libcompressjs_amd.so (MI355X-native bzip2 block pipeline).
 * (round-1 work in progress; the full entry-point list is in INTEGRATION.md)
 */
#ifndef COMPRESSJS_AMD_H
#define COMPRESSJS_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* = BWT.bwtransform2(T, U, n, 256) -> pidx            (reference: lib/BWT.js:372-417) */
int32_t cjs_bwt_cyclic(const uint8_t* T, uint8_t* U, uint32_t n, uint32_t* pidx);
/* the same for nb independent blocks laid out at a fixed pitch `cap` (host pointers) */
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
int32_t cjs_dbg_block_stages(const uint8_t* T, const uint32_t* nlen, uint32_t nb, uint32_t cap,
                             int upto, cjs_dbg_stage_out* out);

#ifdef __cplusplus
}
#endif
#endif
