// Host-side orchestration of the block pipeline: HBM layout of a batch and stage sequencing.
#include "pipeline.h"
#include <string.h>

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// HBM layout of one batch: every array is SoA and block-major (block b at b*pitch); see
// DESIGN.md "Data layout in HBM".  Nothing is aliased in round 1 (about 31 B per block byte).
size_t pipe_bytes(const BatchGeom& g) {
    size_t tot = 0;
    tot += al256((size_t)g.nb * g.tstride);                 // T
    tot += al256((size_t)g.nb * g.stride);                  // U
    tot += al256((size_t)g.nb * (g.stride + 1) * 4);        // RHpos
    tot += 2 * al256((size_t)g.nb * g.stride);              // RHsym, J
    tot += al256((size_t)g.nb * (g.stride / K2_SEG) * 256 * 4);   // Ltab
    tot += al256((size_t)g.nb * g.stride * 2);              // A
    tot += 2 * al256((size_t)g.nb * g.rtiles * 4);          // tileCnt symCnt
    tot += 16 * al256((size_t)g.nb * 8 * 4);                // small per-block arrays
    tot += al256((size_t)g.nb * K2_FREQ_PITCH * 4);         // freq
    const u32 selPitch = (g.stride / CJS_GROUP + 64) & ~63u;
    tot += al256((size_t)g.nb * selPitch);                  // selectors
    tot += al256((size_t)g.nb * CJS_MAX_GROUPS * CJS_LEN_PITCH);       // lens
    tot += al256((size_t)g.nb * CJS_MAX_GROUPS * CJS_LEN_PITCH * 4);   // codes
    tot += al256((size_t)g.nb * selPitch * 2);              // selCost
    tot += al256((size_t)g.nb * 2 * CJS_MAX_GROUPS * CJS_LEN_PITCH * 4);   // fr2
    tot += al256((size_t)g.nb * K5_HDR_WORDS * 4);          // hdr
    tot += al256((size_t)g.nb * K5_TILES(g) * 4);           // tileBits
    tot += 4 * al256((size_t)g.nb * 8);                     // hbits bitlen bitoff ss
    tot += k1_workspace_bytes(g);
    return tot;
}

void pipe_carve(Pipe& P, const BatchGeom& g, void* base) {
    memset(&P, 0, sizeof P);
    P.g = g;
    P.segs = g.stride / K2_SEG;
    char* p = (char*)base;
#define TAKE(field, type, bytes) P.field = (type)p; p += al256(bytes)
    TAKE(T, u8*, (size_t)g.nb * g.tstride);
    TAKE(U, u8*, (size_t)g.nb * g.stride);
    TAKE(RHpos, u32*, (size_t)g.nb * (g.stride + 1) * 4);
    TAKE(RHsym, u8*, (size_t)g.nb * g.stride);
    TAKE(J, u8*, (size_t)g.nb * g.stride);
    TAKE(Ltab, int*, (size_t)g.nb * P.segs * 256 * 4);
    TAKE(A, u16*, (size_t)g.nb * g.stride * 2);
    TAKE(tileCnt, u32*, (size_t)g.nb * g.rtiles * 4);
    TAKE(symCnt, u32*, (size_t)g.nb * g.rtiles * 4);
    TAKE(nlen, u32*, (size_t)g.nb * 4);
    TAKE(crc, u32*, (size_t)g.nb * 4);
    TAKE(pidx, u32*, (size_t)g.nb * 4);
    TAKE(used, u32*, (size_t)g.nb * 8 * 4);
    TAKE(alpha, u32*, (size_t)g.nb * 4);
    TAKE(nruns, u32*, (size_t)g.nb * 4);
    TAKE(pos, u32*, (size_t)g.nb * 4);
    TAKE(ngroups, u32*, (size_t)g.nb * 4);
    TAKE(nsel, u32*, (size_t)g.nb * 4);
    TAKE(bitlen, u64*, (size_t)g.nb * 8);
    TAKE(bitoff, u64*, (size_t)(g.nb + 1) * 8);
    TAKE(freq, u32*, (size_t)g.nb * K2_FREQ_PITCH * 4);
    P.selPitch = (g.stride / CJS_GROUP + 64) & ~63u;
    TAKE(sel, u8*, (size_t)g.nb * P.selPitch);
    TAKE(lens, u8*, (size_t)g.nb * CJS_MAX_GROUPS * CJS_LEN_PITCH);
    TAKE(codes, u32*, (size_t)g.nb * CJS_MAX_GROUPS * CJS_LEN_PITCH * 4);
    TAKE(selCost, u16*, (size_t)g.nb * P.selPitch * 2);
    TAKE(fr2, u32*, (size_t)g.nb * 2 * CJS_MAX_GROUPS * CJS_LEN_PITCH * 4);
    TAKE(hdr, u32*, (size_t)g.nb * K5_HDR_WORDS * 4);
    TAKE(tileBits, u32*, (size_t)g.nb * K5_TILES(g) * 4);
    TAKE(hbits, u32*, (size_t)g.nb * 4);
    TAKE(ss, StreamState*, sizeof(StreamState));
#undef TAKE
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    k1_carve(P.k1, g, p);
    P.k1.T = P.T;
    P.k1.nlen = P.nlen;
    P.k1.U = P.U;
    P.k1.pidx = P.pidx;
}

int pipe_run_block_stages(Pipe& P, u32 max_n, hipStream_t stream, int upto, hipEvent_t after, hipEvent_t done) {
    int rc = k1_run(P.k1, P.g, max_n, stream);
    if (rc || upto <= 1) return rc;
    rc = k2_run(P, max_n, stream);
    if (rc || upto <= 2) return rc;
    rc = k34_run(P, stream);
    if (rc || upto <= 4) return rc;
    return k5_run(P, max_n, stream, after, done);
}
