// Host entry points of the GPU bzip2 decoder (decode.hip), used by the C ABI (cjs_abi.hip).
#pragma once
#include "cjs_common.h"

struct DecState;
void dec_free(DecState* S);
// Bunzip.decode: returns the decoded size (bytes stay in the decoder's HBM buffer) or an Err code
int64_t dec_stream(DecState** ps, u32 slots, hipStream_t st, const u8* in, u64 len, bool in_dev, int multistream,
                   bool check_stream_crc);
// Bunzip.decodeBlock on a host buffer
int64_t dec_block(DecState** ps, u32 slots, hipStream_t st, const u8* in, u64 len, u64 bitpos);
const u8* dec_output(DecState* S, u64* size);
void dec_error_info(DecState* S, int* detail, u32* got, u32* want);
u32 dec_table(DecState* S, const u64** pos, const u64** size);
