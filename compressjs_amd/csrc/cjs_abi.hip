// C ABI of libcompressjs_amd.so (declared in include/compressjs_amd.h).
#include "../../include/compressjs_amd.h"
#include "cjs_common.h"
#include "k1_bwt.h"
#include "pipeline.h"
#include <vector>
#include <string.h>

static int ensure_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CJS_E_NOGPU;
    return CJS_OK;
}

// Cyclic BWT of several independent blocks (block i = T + i*cap, length nlen[i] <= cap).
static int32_t bwt_batch_impl(const uint8_t* T, const uint32_t* nlen, uint32_t nb, uint32_t cap,
                              uint8_t* U, uint32_t* pidx, int reps, float* ms_out) {
    if (!T || !U || !nlen || !pidx || nb == 0 || cap == 0 || cap > (1u << 20) - 1) return CJS_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    BatchGeom g = make_geom(nb, cap);
    u32 max_n = 0;
    std::vector<u8> text((size_t)nb * g.tstride, 0);
    std::vector<u32> lens(nb);
    for (u32 b = 0; b < nb; b++) {
        const u32 n = nlen[b];
        if (n > cap) return CJS_E_ARG;
        lens[b] = n;
        if (n > max_n) max_n = n;
        u8* dst = text.data() + (size_t)b * g.tstride;
        const u8* src = T + (size_t)b * cap;
        if (n) {
            memcpy(dst, src, n);
            for (u32 i = 0; i < K1_TPAD; i++) dst[n + i] = dst[i % n] ;
        }
    }
    // blocks of length 0/1 never reach the kernels (lib/BWT.js:376-379)
    K1Buf B;
    memset(&B, 0, sizeof B);
    u8 *dT = nullptr, *dU = nullptr; u32 *dN = nullptr, *dP = nullptr; void* ws = nullptr;
    const size_t wsb = k1_workspace_bytes(g);
    hipStream_t st = nullptr;
    hipError_t e;
#define TRY(x) if ((e = (x)) != hipSuccess) { rc = CJS_E_HIP - (int)e; goto done; }
    TRY(hipStreamCreate(&st));
    TRY(hipMalloc((void**)&dT, text.size()));
    TRY(hipMalloc((void**)&dU, (size_t)nb * g.stride));
    TRY(hipMalloc((void**)&dN, nb * 4));
    TRY(hipMalloc((void**)&dP, nb * 4));
    TRY(hipMalloc(&ws, wsb));
    TRY(hipMemcpyAsync(dT, text.data(), text.size(), hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(dN, lens.data(), nb * 4, hipMemcpyHostToDevice, st));
    TRY(hipMemsetAsync(dP, 0, nb * 4, st));
    k1_carve(B, g, ws);
    B.T = dT; B.nlen = dN; B.U = dU; B.pidx = dP;
    if (max_n >= 2) {
        hipEvent_t e0, e1;
        TRY(hipEventCreate(&e0)); TRY(hipEventCreate(&e1));
        rc = k1_run(B, g, max_n, st);          // warm-up / the result
        if (rc) goto done;
        TRY(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r++) { rc = k1_run(B, g, max_n, st); if (rc) goto done; }
        TRY(hipEventRecord(e1, st));
        TRY(hipStreamSynchronize(st));
        float ms = 0.f;
        TRY(hipEventElapsedTime(&ms, e0, e1));
        if (ms_out) *ms_out = reps > 0 ? ms / reps : 0.f;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    TRY(hipStreamSynchronize(st));
    {
        std::vector<u8> hu((size_t)nb * g.stride);
        TRY(hipMemcpy(hu.data(), dU, hu.size(), hipMemcpyDeviceToHost));
        TRY(hipMemcpy(pidx, dP, nb * 4, hipMemcpyDeviceToHost));
        for (u32 b = 0; b < nb; b++) {
            if (lens[b] >= 2) memcpy(U + (size_t)b * cap, hu.data() + (size_t)b * g.stride, lens[b]);
            else { if (lens[b] == 1) U[(size_t)b * cap] = T[(size_t)b * cap]; pidx[b] = 0; }
        }
    }
done:
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(dT); (void)hipFree(dU); (void)hipFree(dN); (void)hipFree(dP); (void)hipFree(ws);
    return rc;
#undef TRY
}

extern "C" int32_t cjs_bwt_cyclic_batch(const uint8_t* T, const uint32_t* nlen, uint32_t nb, uint32_t cap,
                                        uint8_t* U, uint32_t* pidx) {
    return bwt_batch_impl(T, nlen, nb, cap, U, pidx, 0, nullptr);
}
// debug/bench helper: same, re-running the device pipeline `reps` times and reporting ms per run
extern "C" int32_t cjs_dbg_bwt_batch_time(const uint8_t* T, const uint32_t* nlen, uint32_t nb, uint32_t cap,
                                          uint8_t* U, uint32_t* pidx, int reps, float* ms) {
    return bwt_batch_impl(T, nlen, nb, cap, U, pidx, reps, ms);
}

extern "C" int32_t cjs_bwt_cyclic(const uint8_t* T, uint8_t* U, uint32_t n, uint32_t* pidx) {
    if (n == 0) { if (pidx) *pidx = 0; return CJS_OK; }
    return cjs_bwt_cyclic_batch(T, &n, 1, n, U, pidx);
}

// ---------------------------------------------------------------------------------------------
// debug / test entry: run the block stages on host-supplied RLE1 blocks and copy every
// intermediate array back, so tests can compare stage by stage with the oracle.
// ---------------------------------------------------------------------------------------------
extern "C" int32_t cjs_dbg_block_stages(const uint8_t* T, const uint32_t* nlen, uint32_t nb, uint32_t cap,
                                        int upto, cjs_dbg_stage_out* o) {
    void* dout = nullptr;
    if (!T || !nlen || !o || nb == 0 || cap == 0 || cap > (1u << 20) - 1) return CJS_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    BatchGeom g = make_geom(nb, cap);
    u32 max_n = 0;
    std::vector<u8> text((size_t)nb * g.tstride, 0);
    for (u32 b = 0; b < nb; b++) {
        const u32 n = nlen[b];
        if (n > cap || n == 0) return CJS_E_ARG;
        if (n > max_n) max_n = n;
        u8* dst = text.data() + (size_t)b * g.tstride;
        memcpy(dst, T + (size_t)b * cap, n);
        for (u32 i = 0; i < K1_TPAD; i++) dst[n + i] = dst[i % n];
    }
    void* ws = nullptr;
    hipStream_t st = nullptr;
    hipError_t e;
    Pipe P;
#define TRY(x) if ((e = (x)) != hipSuccess) { rc = CJS_E_HIP - (int)e; goto done; }
    TRY(hipStreamCreate(&st));
    TRY(hipMalloc(&ws, pipe_bytes(g)));
    pipe_carve(P, g, ws);
    TRY(hipMemcpyAsync(P.T, text.data(), text.size(), hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(P.nlen, nlen, nb * 4, hipMemcpyHostToDevice, st));
    TRY(hipMemsetAsync(P.pidx, 0, nb * 4, st));
    if (o->crc_in) { TRY(hipMemcpyAsync(P.crc, o->crc_in, nb * 4, hipMemcpyHostToDevice, st)); }
    else { TRY(hipMemsetAsync(P.crc, 0, nb * 4, st)); }
    if (upto >= 5) {
        P.outCapBytes = ((size_t)nb * ((size_t)cap * 2 + 8192) + 4096 + 255) & ~(size_t)255;
        TRY(hipMalloc(&dout, P.outCapBytes));
        P.out = (u32*)dout;
        rc = k5_stream_begin(P, o->level ? o->level : 9, st);
        if (rc) goto done;
    }
    rc = pipe_run_block_stages(P, max_n, st, upto);
    if (rc) goto done;
    if (upto >= 5) { rc = k5_stream_end(P, st); if (rc) goto done; }
    TRY(hipStreamSynchronize(st));
#define BACK2D(dst, src, rowbytes, srcpitch, dstpitch)                                            \
    if (dst) for (u32 b = 0; b < nb; b++)                                                        \
        TRY(hipMemcpy((char*)(dst) + (size_t)b * (dstpitch), (const char*)(src) + (size_t)b * (srcpitch), \
                      (rowbytes), hipMemcpyDeviceToHost));
    BACK2D(o->U, P.U, max_n, g.stride, cap);
    BACK2D(o->pidx, P.pidx, 4, 4, 4);
    if (upto >= 2) {
        BACK2D(o->A, P.A, ((size_t)max_n + 1) * 2, (size_t)g.stride * 2, ((size_t)cap + 1) * 2);
        BACK2D(o->pos, P.pos, 4, 4, 4);
        BACK2D(o->alpha, P.alpha, 4, 4, 4);
        BACK2D(o->freq, P.freq, 258 * 4, K2_FREQ_PITCH * 4, 258 * 4);
        BACK2D(o->used, P.used, 32, 32, 32);
    }
    if (upto >= 3) {
        BACK2D(o->sel, P.sel, (max_n + 1) / 50 + 2, P.selPitch, (cap + 1) / 50 + 2);
        for (int t = 0; t < 6; t++) {
            BACK2D(o->lens ? o->lens + t * 258 : nullptr, P.lens + t * CJS_LEN_PITCH, 258,
                   CJS_MAX_GROUPS * CJS_LEN_PITCH, 6 * 258);
        }
        BACK2D(o->ngroups, P.ngroups, 4, 4, 4);
        BACK2D(o->nsel, P.nsel, 4, 4, 4);
    }
    if (upto >= 5) {
        BACK2D(o->bitlen, P.bitlen, 8, 8, 8);
        StreamState hs;
        TRY(hipMemcpy(&hs, P.ss, sizeof hs, hipMemcpyDeviceToHost));
        if (hs.overflow) { rc = CJS_E_NOSPACE; goto done; }
        o->stream_bytes = (hs.bits + 7) >> 3;
        if (o->stream && o->stream_cap >= o->stream_bytes) {
            TRY(hipMemcpy(o->stream, P.out, o->stream_bytes, hipMemcpyDeviceToHost));
        }
    }
done:
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(ws);
    (void)hipFree(dout);
    return rc;
#undef TRY
#undef BACK2D
}
