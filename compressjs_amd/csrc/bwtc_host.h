// Serial entropy tail of the BWTC codec (host side).  See bwtc_host.hip.
#pragma once
#include <stdint.h>
struct bwtc_coder;
extern "C" uint64_t bwtc_bound(uint64_t in_len);
bwtc_coder* bwtc_begin(uint8_t* out, uint64_t cap, int64_t file_size, int level);
void bwtc_block(bwtc_coder* c, uint32_t length, uint32_t pidx, const uint32_t* used8, const uint16_t* sym, uint32_t nsym);
int64_t bwtc_end(bwtc_coder* c);
