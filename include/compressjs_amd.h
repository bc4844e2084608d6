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

#ifdef __cplusplus
}
#endif
#endif
