// Serial entropy tail of the BWTC codec (host side).  See bwtc_host.hip.
#pragma once
#include <stdint.h>
struct bwtc_coder;
extern "C" uint64_t bwtc_bound(uint64_t in_len);
bwtc_coder* bwtc_begin(uint8_t* out, uint64_t cap, int64_t file_size, int level);
void bwtc_block(bwtc_coder* c, uint32_t length, uint32_t pidx, const uint32_t* used8, const uint16_t* sym, uint32_t nsym);
void bwtc_block_triples(bwtc_coder* c, uint32_t length, uint32_t pidx, const uint32_t* used8, const uint32_t* sylt,
                        const uint32_t* tot, uint32_t ntri);
int64_t bwtc_end(bwtc_coder* c);

// Decoder side (BWTC.decompressFile, lib/BWTC.js:141-233): serial range decoder + models on the host;
// every block is handed to `on_block` as the BWT string (MTF and zero-run coding already undone)
// with its primary index; the caller inverts the BWT on the GPU (K6).
// Returns 0, BWTC_E_MAGIC ('Bad magic', lib/Util.js:150-152), BWTC_E_CORRUPT (input ends early or is
// inconsistent: the reference has no defined behaviour there), or the non-zero value on_block returned.
#define BWTC_E_MAGIC (-30)
#define BWTC_E_CORRUPT (-31)
typedef int (*bwtc_block_fn)(void* user, const uint8_t* T, uint32_t length, uint32_t pidx);
int bwtc_decode(const uint8_t* in, uint64_t len, int64_t* declared_size, void* user, bwtc_block_fn on_block);
