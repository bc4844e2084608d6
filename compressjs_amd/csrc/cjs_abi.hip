// C ABI of libcompressjs_amd.so (declared in include/compressjs_amd.h).
#include "../../include/compressjs_amd.h"
#include "cjs_common.h"
#include "k1_bwt.h"
#include "pipeline.h"
#include "bwtc_host.h"
#include "decode_host.h"
#include <vector>
#include <atomic>
#include <chrono>
#include <thread>
#include <functional>
#include <mutex>
#include <condition_variable>
#include <string.h>
#include <stdlib.h>

static int ensure_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CJS_E_NOGPU;
    return CJS_OK;
}

// Cyclic BWT of several independent blocks (block i = T + i*cap, length nlen[i] <= cap).
static int32_t bwt_batch_impl(const uint8_t* T, const uint32_t* nlen, uint32_t nb, uint32_t cap,
                              uint8_t* U, uint32_t* pidx, int reps, float* ms_out, int linear = 0,
                              int32_t* SAout = nullptr) {
    if (!T || !U || !nlen || !pidx || nb == 0 || cap == 0 || cap > (1u << 22) - 1) return CJS_E_ARG;
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
            for (u32 i = 0; i < K1_TPAD; i++) dst[n + i] = linear ? 0 : dst[i % n];
        }
    }
    // blocks of length 0/1 never reach the kernels (lib/BWT.js:376-379)
    K1Buf B;
    memset(&B, 0, sizeof B);
    u8 *dT = nullptr, *dU = nullptr; u32 *dN = nullptr, *dP = nullptr; void* ws = nullptr;
    int* dSA = nullptr;
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
    B.linear = linear ? 1u : 0u;
    if (SAout) { TRY(hipMalloc((void**)&dSA, (size_t)nb * g.stride * 4)); B.SAout = dSA; }
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
            else { if (lens[b] == 1) U[(size_t)b * cap] = T[(size_t)b * cap]; pidx[b] = linear ? lens[b] : 0; }   // lib/BWT.js:332-335,376-379
        }
        if (SAout) for (u32 b = 0; b < nb; b++) {
            if (lens[b] >= 2) { TRY(hipMemcpy(SAout + (size_t)b * cap, dSA + (size_t)b * g.stride, (size_t)lens[b] * 4, hipMemcpyDeviceToHost)); }
            else if (lens[b] == 1) SAout[(size_t)b * cap] = 0;
        }
    }
done:
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(dT); (void)hipFree(dU); (void)hipFree(dN); (void)hipFree(dP); (void)hipFree(ws); (void)hipFree(dSA);
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

// = BWT.bwtransform(T, U, A, n, 256) -> pidx   (lib/BWT.js:328-350): BWT of T$ with implicit sentinel
extern "C" int32_t cjs_bwt_linear(const uint8_t* T, uint8_t* U, uint32_t n, uint32_t* pidx) {
    if (n == 0) { if (pidx) *pidx = 0; return CJS_OK; }
    return bwt_batch_impl(T, &n, 1, n, U, pidx, 0, nullptr, 1, nullptr);
}
// = BWT.suffixsort(T, SA, n, 256)               (lib/BWT.js:305-321)
extern "C" int32_t cjs_suffixsort(const uint8_t* T, int32_t* SA, uint32_t n) {
    if (n == 0) return CJS_OK;
    if (!SA) return CJS_E_ARG;
    std::vector<u8> u(n);
    u32 p = 0;
    return bwt_batch_impl(T, &n, 1, n, u.data(), &p, 0, nullptr, 1, SA);
}

// = BWT.unbwtransform(T, U, LF, n, pidx)          (lib/BWT.js:352-363): inverse of cjs_bwt_linear
extern "C" int32_t cjs_unbwt_linear(const uint8_t* T, uint8_t* U, uint32_t n, uint32_t pidx) {
    if (n == 0) return CJS_OK;
    if (!T || !U || pidx > n) return CJS_E_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    u8 *dT = nullptr, *dU = nullptr; void* ws = nullptr;
    hipStream_t st = nullptr;
    hipError_t e;
    const size_t wsb = (size_t)n * 20 + ((size_t)(n + 4095) / 4096) * 1024 + 256;
#define TRY(x) if ((e = (x)) != hipSuccess) { rc = CJS_E_HIP - (int)e; goto done; }
    TRY(hipStreamCreate(&st));
    TRY(hipMalloc((void**)&dT, n));
    TRY(hipMalloc((void**)&dU, n));
    TRY(hipMalloc(&ws, wsb));
    TRY(hipMemcpyAsync(dT, T, n, hipMemcpyHostToDevice, st));
    TRY(hipMemsetAsync(dU, 0, n, st));
    rc = k6_unbwt_linear(dT, dU, n, pidx, ws, st);
    if (rc) goto done;
    TRY(hipStreamSynchronize(st));
    TRY(hipMemcpy(U, dU, n, hipMemcpyDeviceToHost));
done:
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(dT); (void)hipFree(dU); (void)hipFree(ws);
    return rc;
#undef TRY
}

// = allocateHuffmanCodeLengths(array, maxLength)   (lib/HuffmanAllocator.js:199-222), `count`
// independent arrays at once: array k is arr[off[k] .. off[k+1]), ascending weights in, lengths out.
extern "C" int32_t cjs_huff_lengths_batch(int64_t* arr, const uint32_t* off, uint32_t count, uint32_t max_len) {
    if (count == 0) return CJS_OK;
    if (!arr || !off || max_len < 1 || max_len > 62) return CJS_E_ARG;
    for (u32 k = 0; k < count; k++) {
        if (off[k + 1] < off[k]) return CJS_E_ARG;
        const u64 len = off[k + 1] - off[k];
        if (len > (1ull << max_len)) return CJS_E_ARG;          // no prefix code that short exists
    }
    const u32 total = off[count];
    if (total == 0) return CJS_OK;
    int rc = ensure_device();
    if (rc) return rc;
    long long* d_arr = nullptr; u32* d_off = nullptr;
    hipError_t e;
#define TRY(x) if ((e = (x)) != hipSuccess) { rc = CJS_E_HIP - (int)e; goto done; }
    TRY(hipMalloc((void**)&d_arr, (size_t)total * 8));
    TRY(hipMalloc((void**)&d_off, (size_t)(count + 1) * 4));
    TRY(hipMemcpy(d_arr, arr, (size_t)total * 8, hipMemcpyHostToDevice));
    TRY(hipMemcpy(d_off, off, (size_t)(count + 1) * 4, hipMemcpyHostToDevice));
    rc = k3_alloc_lengths_run(d_arr, d_off, count, (int)max_len, nullptr);
    if (rc) goto done;
    TRY(hipDeviceSynchronize());
    TRY(hipMemcpy(arr, d_arr, (size_t)total * 8, hipMemcpyDeviceToHost));
done:
    (void)hipFree(d_arr); (void)hipFree(d_off);
    return rc;
#undef TRY
}
extern "C" int32_t cjs_huff_lengths(int64_t* arr, uint32_t n, uint32_t max_len) {
    const uint32_t off[2] = {0, n};
    return cjs_huff_lengths_batch(arr, off, 1, max_len);
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
    if (!T || !nlen || !o || nb == 0 || cap == 0 || cap > (1u << 22) - 1) return CJS_E_ARG;
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

// =============================================================================================
// Product entry points
// =============================================================================================
#define CJS_PIN_WORDS 8192u  // pinned words per stream for K1's read-backs
#define CJS_NSTREAMS 4       // sub-batches in flight: latency-bound stages of one sub-batch
                             // (Huffman optimiser, scans, sparse sort rounds) overlap the
                             // bandwidth-bound stages of the others

// BWTC: what the coder thread needs of one group of sub-batches (kept in the context: fresh 100 MB vectors per call would
// be page-faulted in by the D2H copies)
struct BwtcBlockJob { u32 len, pidx, nsym, ntri; u32 used[8]; size_t off; bool host; size_t soff; };   // host: K10 gave up (triples beyond its rows): model on the host
// triples of a group of blocks on the host: PINNED (hipHostMalloc, grow-only): the 350 MB of a 10^8-byte input took ~100 ms to
// arrive in pageable vectors, more than K1 + K2 together
struct PinnedU32 {
    u32* p = nullptr; size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return CJS_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        n += n / 8 + 1024;
        const hipError_t e = hipHostMalloc((void**)&p, n * sizeof(u32), hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return CJS_E_HIP - (int)e; }
        cap = n;
        return CJS_OK;
    }
    u32* data() { return p; }
    ~PinnedU32() { if (p) (void)hipHostFree(p); }
};
struct BwtcGroupJob { std::vector<BwtcBlockJob> blocks; PinnedU32 a, t; std::vector<u16> sym; bool busy = false; size_t nready = 0; };   // nready: leading blocks whose data has landed

// One persistent helper thread per extra stream: a sub-batch is issued from a host thread of its own (K1's read-backs block the thread that
// waits for them), and a thread made per call pays for its creation and its first HIP call (30-170 us before its first launch, seen on the
// kernel timeline) every time.
struct CjsHelper {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, busy = false, quit = false;
    void loop() {
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&]() { return has_job || quit; });
                if (!has_job) return;
                j = std::move(job);
                has_job = false;
            }
            j();
            {
                std::lock_guard<std::mutex> g(mu);
                busy = false;
            }
            cv.notify_all();
        }
    }
    void post(std::function<void()> j) {
        {
            std::lock_guard<std::mutex> g(mu);
            job = std::move(j);
            has_job = true;
            busy = true;
            if (!th.joinable()) th = std::thread([this]() { loop(); });
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&]() { return !busy; });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> g(mu);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

struct cjs_ctx {
    int device;
    hipStream_t stream;        // stream 0: pre-pass, framing, timing events
    hipStream_t sub[CJS_NSTREAMS];
    hipEvent_t evReady;        // pre-pass + output zeroing done
    hipEvent_t evScan[CJS_NSTREAMS];   // k5_blockscan of the last sub-batch issued on that stream
    hipEvent_t evDone[CJS_NSTREAMS];   // everything issued on that stream
    u32* pin[CJS_NSTREAMS];            // pinned host memory for K1's small read-backs of that stream's sub-batch (CJS_PIN_WORDS u32)
    CjsHelper* helper[CJS_NSTREAMS];   // [i]: the host thread that issues stream i's sub-batches (i >= 1; stream 0's are issued by the caller)
    u32 nstreams;              // streams in use (<= CJS_NSTREAMS; env CJS_STREAMS overrides)
    u32 batch_blocks;          // blocks in flight over all streams
    u32 sub_blocks;            // blocks per sub-batch, at most (what every stream's workspace is sized for)
    u32 share[CJS_NSTREAMS];   // per mille of a batch that stream i's sub-batch takes (sums to 1000)
    void* ws[CJS_NSTREAMS];    // block-pipeline workspaces (level-9 geometry)
    size_t ws_bytes;
    StreamState* d_ss;         // stream cursor + combined CRC, shared by all sub-batches
    void* k0ws;                // K0 workspace (grows with the input length)
    size_t k0ws_bytes;
    void* planws;              // K0 workspace of cjs_bz2_plan: its own, so that a later compress call on this
    size_t planws_bytes;       // context cannot re-carve or free what `plan` points into
    void* din;  size_t din_bytes;      // staging for the host-buffer entry point
    void* dout; size_t dout_bytes;
    hipEvent_t ev0, ev1;
    float last_ms;             // device time of the last compress call (HIP events on ctx->stream)
    u32 last_blocks;
    // optional per-kernel timing of the dominant kernel (bench.py roofline leg)
    K1Prof prof;
    BatchGeom prof_g;          // geometry of the last sub-batch issued (where cjs_profile_read_class finds K1's counters)
    // state of cjs_bz2_plan
    K0Buf plan;
    int plan_level;
    u32 plan_blocks;
    float bwtc_times[5];       // last cjs_bwtc_compress: K10 launch ms (first stream), ms until all triples were on the host, coder busy ms, total ms, encodeFreq calls
    hipEvent_t evK10[2];
    int scan_level;            // cjs_bz2_plan_scan ran for this level (0: no scan)
    uint64_t scan_total;
    // decoder state (allocated by the first decompress call)
    DecState* dec;
    float dec_ms;
    std::vector<u8>* bwtc_out;  // result of the last cjs_bwtc_decompress
    BwtcGroupJob* bwtc_jobs[2]; // double buffer between the GPU stages and the coder thread of cjs_bwtc_compress
    // the overlapped host path (compress_overlapped): persistent helper threads, copy streams, per-slice events and cursors
    hipStream_t side;           // the block CRCs of every sub-batch (k0_crc) run here, next to K1
    hipEvent_t evPad[CJS_NSTREAMS], evCrc[CJS_NSTREAMS];
    CjsHelper* io[3];           // [0] uploader, [1] downloader, [2] issues the slices of stream 0 (the caller's thread plans)
    hipStream_t sIn, sOut;      // copy streams (non-blocking)
    std::vector<hipEvent_t>* evPool;   // per slice: its k5_blockscan, everything of it
    u64* snapPin;               // [CJS_SNAP_SLOTS] pinned: bit cursor behind every slice (written by k5_blockscan)
    void* k0sl; size_t k0sl_bytes;     // K0 workspaces of the slices of one call (one each: no reuse within a call)
    std::vector<std::pair<void*, size_t>>* segpool;   // cjs_bz2_compress_multi: this device's segment buffers, one per wave of a call, grow-only and kept across calls
};
static std::atomic<int> g_multi_mallocs{0};        // hipMalloc calls of cjs_bz2_compress_multi's segment buffers (cjs_dbg_multi_mallocs: none after warm-up)
extern "C" int cjs_dbg_multi_mallocs(void) { return g_multi_mallocs.load(); }
static std::atomic<int> g_multi_replans{0};        // segments of cjs_bz2_compress_multi planned a second time (the chain carried a shift to them)
extern "C" int cjs_dbg_multi_replans(void) { return g_multi_replans.load(); }
static std::atomic<int> g_multi_fallbacks{0};      // calls of cjs_bz2_compress_multi that took the replicated plan (a segment that cannot be planned on its own)
extern "C" int cjs_dbg_multi_fallbacks(void) { return g_multi_fallbacks.load(); }
#define CJS_SNAP_SLOTS 4096u

extern "C" void cjs_destroy(cjs_ctx* c);

extern "C" int32_t cjs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 0) return 0;
    return n;
}

extern "C" cjs_ctx* cjs_create(int device, uint32_t batch_blocks) {
    if (ensure_device()) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    cjs_ctx* c = new cjs_ctx();
    memset(c, 0, sizeof *c);
    c->device = device;
    c->batch_blocks = batch_blocks ? batch_blocks : 128;
    // Two streams by default, each driven by its own host thread (issue_blocks): the latency-bound kernels of one half of
    // the batch (K3/K4: four workgroups per block, the sparse rounds, the small scans, K1's read-backs) overlap with
    // the bandwidth-bound ones of the other.  Measured, 10^8 bytes, ms per step with 1 / 2 / 3 / 4 streams: enwik
    // 14.86 / 14.35 / 14.73 / 17.72, E8S-A 23.94 / 21.45 / 21.90 / 27.48, random ASCII 12.81 / 12.07 / 12.58 / 15.35 (round 2;
    // round 3, enwik: one stream 9.4, two 8.7, three 8.7, four 11.6 - the main stream and four more share four hardware queues).
    // CJS_STREAMS=1 gives every kernel the GPU to itself: what per-kernel timings (bench.py's roofline leg, rocprof
    // summaries) are taken with.
    c->nstreams = batch_blocks && batch_blocks < 32 ? 1 : 2;
    if (const char* ev = getenv("CJS_STREAMS")) {
        const int v = atoi(ev);
        if (v >= 1 && v <= CJS_NSTREAMS) c->nstreams = (u32)v;
    }
#ifdef CJS_CPU_DEBUG_BUILD
    c->nstreams = 1;                             // the CPU logic-debug build runs kernels as fibers of one thread
#endif
    for (u32 i = 0; i < c->nstreams; i++) c->share[i] = 1000u / c->nstreams + (i < 1000u % c->nstreams ? 1u : 0u);
    if (const char* ev = getenv("CJS_SHARES")) {        // "300,700": unequal sub-batches (the streams then leave K1 at different times)
        u32 v[CJS_NSTREAMS], n = 0, sum = 0;
        for (const char* q = ev; *q && n < CJS_NSTREAMS; ) {
            v[n] = (u32)strtoul(q, (char**)&q, 10); sum += v[n++];
            if (*q == ',' || *q == ':') q++;
        }
        bool good = n >= 1 && sum == 1000u;
        for (u32 i = 0; i < n; i++) good = good && v[i] >= 50u;
#ifndef CJS_CPU_DEBUG_BUILD
        if (good) { c->nstreams = n; for (u32 i = 0; i < n; i++) c->share[i] = v[i]; }
#endif
    }
    {
        u32 mx = 0;
        for (u32 i = 0; i < c->nstreams; i++) mx = c->share[i] > mx ? c->share[i] : mx;
        c->sub_blocks = (u32)(((u64)c->batch_blocks * mx + 999u) / 1000u);
    }
    bool ok = hipStreamCreate(&c->stream) == hipSuccess;
    BatchGeom g = make_geom(c->sub_blocks, 9u * 100000u - 19u);
    c->ws_bytes = pipe_bytes(g);
    for (u32 i = 0; ok && i < c->nstreams; i++) {
        ok = ok && hipStreamCreate(&c->sub[i]) == hipSuccess;
        ok = ok && hipMalloc(&c->ws[i], c->ws_bytes) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->evScan[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->evDone[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&c->pin[i], CJS_PIN_WORDS * 4) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->evPad[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->evCrc[i], hipEventDisableTiming) == hipSuccess;
    }
#ifndef CJS_CPU_DEBUG_BUILD
    if (ok && !getenv("CJS_NO_SIDE_CRC")) ok = hipStreamCreate(&c->side) == hipSuccess;
#endif
    ok = ok && hipEventCreateWithFlags(&c->evReady, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_ss, 256) == hipSuccess;
    ok = ok && hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess;
    if (!ok) { cjs_destroy(c); return nullptr; }
    return c;
}

extern "C" void cjs_destroy(cjs_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < CJS_NSTREAMS; i++) {
        (void)hipFree(c->ws[i]);
        if (c->evScan[i]) (void)hipEventDestroy(c->evScan[i]);
        if (c->evDone[i]) (void)hipEventDestroy(c->evDone[i]);
        if (c->pin[i]) (void)hipHostFree(c->pin[i]);
        if (c->evPad[i]) (void)hipEventDestroy(c->evPad[i]);
        if (c->evCrc[i]) (void)hipEventDestroy(c->evCrc[i]);
        if (c->helper[i]) { c->helper[i]->stop(); delete c->helper[i]; }
        if (c->sub[i]) (void)hipStreamDestroy(c->sub[i]);
    }
    (void)hipFree(c->d_ss); (void)hipFree(c->k0ws); (void)hipFree(c->planws); (void)hipFree(c->din); (void)hipFree(c->dout);
    if (c->evReady) (void)hipEventDestroy(c->evReady);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (int i = 0; i < 2; i++) if (c->evK10[i]) (void)hipEventDestroy(c->evK10[i]);
    k1_prof_destroy(c->prof);
    dec_free(c->dec);
    delete c->bwtc_out;
    delete c->bwtc_jobs[0];
    delete c->bwtc_jobs[1];
    for (int i = 0; i < 3; i++) if (c->io[i]) { c->io[i]->stop(); delete c->io[i]; }
    if (c->evPool) { for (hipEvent_t ev : *c->evPool) (void)hipEventDestroy(ev); delete c->evPool; }
    if (c->snapPin) (void)hipHostFree(c->snapPin);
    (void)hipFree(c->k0sl);
    if (c->segpool) { for (auto& pr : *c->segpool) (void)hipFree(pr.first); delete c->segpool; }
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->sIn) (void)hipStreamDestroy(c->sIn);
    if (c->sOut) (void)hipStreamDestroy(c->sOut);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int64_t cjs_bz2_compress_bound(uint64_t in_len) {
    // RLE1 expands by at most 5/4, Huffman codes are at most 20 bits per symbol; headers per block
    return (int64_t)(in_len + in_len / 2 + (in_len / 100000 + 2) * 24576 + 4096);
}

static int grow(void** p, size_t* have, size_t need) {
    if (*have >= need) return CJS_OK;
    (void)hipFree(*p);
    *p = nullptr; *have = 0;
    need = (need + (1u << 20)) & ~(size_t)((1u << 20) - 1);
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) return CJS_E_HIP - (int)e;
    *have = need;
    return CJS_OK;
}

// Issue blocks [first, first+count) of the planned input K as sub-batches round-robin over the
// context's streams.  Stream 0 must already hold the pre-pass + output zeroing (evReady).
// One sub-batch = up to sub_blocks blocks through K0 (materialise) .. K5 on one stream.  The only
// cross-stream dependency is the stream cursor: k5_blockscan of sub-batch j needs the cursor left by
// sub-batch j-1 (an event); everything before it overlaps freely.
static int run_sub_batch(cjs_ctx* c, const K0Buf& K, const BatchGeom& g, u32 cap, u32 f, u32 nb, u32 si, void* d_out,
                         uint64_t out_cap, Pipe& P, u32 total_blocks) {
    hipStream_t st = c->sub[si];
    pipe_carve(P, g, c->ws[si]);
    P.ss = c->d_ss;
    P.out = (u32*)d_out;
    P.outCapBytes = out_cap & ~(uint64_t)3;
    P.k1.prof = c->prof.enabled ? &c->prof : nullptr;
    P.k1.hpin = c->pin[si]; P.k1.hpinWords = CJS_PIN_WORDS;
    P.g.nb = nb;
    // (CRC workgroups per block: one per 2 MB of input a block of this call consumes on average - long runs make that tens of megabytes;
    // total_blocks = the blocks the planned input K holds, so that windows with a margin and sub-ranges count their own bytes)
    const u64 per_block = K.in_len / (total_blocks ? total_blocks : 1u);
    int rc = k0_batch(K, P, f, cap, st, (u32)(1u + per_block / (2u << 20)), c->side, c->side ? c->evPad[si] : nullptr, c->side ? c->evCrc[si] : nullptr);
    if (rc) return rc;
    return pipe_run_block_stages(P, cap, st, 4);
}

static int issue_blocks(cjs_ctx* c, const K0Buf& K, u32 cap, u32 first, u32 count, void* d_out, uint64_t out_cap) {
    BatchGeom g = make_geom(c->sub_blocks, cap);
    if (c->prof.enabled) c->prof_g = g;                   // (once, before the workers start: they only read it)
    // The blocks go out in batches of batch_blocks; a batch is cut into one sub-batch per stream by the streams' shares
    // (when it has at least 16 blocks per stream: below that one stream takes it whole, or sub_blocks at a time).
    std::vector<u32> sfirst, scount;
    for (u32 f = 0; f < count; ) {
        const u32 wave = count - f < c->batch_blocks ? count - f : c->batch_blocks;
        if (c->nstreams > 1 && wave >= 16 * c->nstreams) {
            u32 acc = 0, done = 0;
            for (u32 i = 0; i < c->nstreams; i++) {
                acc += c->share[i];
                const u32 upto = i + 1 == c->nstreams ? wave : (u32)(((u64)wave * acc + 500u) / 1000u);
                if (upto > done) { sfirst.push_back(first + f + done); scount.push_back(upto - done); }
                done = upto;
            }
            f += wave;
        } else {
            const u32 take = wave < c->sub_blocks ? wave : c->sub_blocks;
            sfirst.push_back(first + f); scount.push_back(take);
            f += take;
        }
    }
    const u32 nsub = (u32)sfirst.size();
    const u32 ns = c->nstreams < nsub ? c->nstreams : nsub;
    for (u32 i = 0; i < ns; i++) HIP_CHECK_RET(hipStreamWaitEvent(c->sub[i], c->evReady, 0));
    // K1 steers itself with small read-backs (stream syncs), so every stream gets its own host thread:
    // while one sub-batch waits for its counters the other keeps the GPU fed.  The cursor events are
    // recorded and waited for in sub-batch order (`recorded` hands the order from thread to thread).
    std::atomic<u32> recorded(0);
    std::atomic<int> err(0);
    auto worker = [&](u32 si) {
        if (hipSetDevice(c->device) != hipSuccess) { int z = 0; err.compare_exchange_strong(z, CJS_E_NOGPU); }   // (it still walks its slots below, work skipped: the hand-off chain must complete - ADVICE r4)
        for (u32 j = si; j < nsub; j += ns) {
            const u32 f = sfirst[j], nb = scount[j];
            Pipe P;
            int rc = err.load() ? err.load() : run_sub_batch(c, K, g, cap, f, nb, si, d_out, out_cap, P, first + count);
            while (recorded.load(std::memory_order_acquire) != j) std::this_thread::yield();
            if (!rc && !err.load()) rc = k5_run(P, cap, c->sub[si], j ? c->evScan[(j - 1) % ns] : nullptr, c->evScan[si], c->side ? c->evCrc[si] : nullptr);
            if (rc) { int z = 0; err.compare_exchange_strong(z, rc); }
            recorded.store(j + 1, std::memory_order_release);
        }
    };
    if (ns <= 1) worker(0);
    else {
        for (u32 i = 1; i < ns; i++) {
            if (!c->helper[i]) c->helper[i] = new CjsHelper();
            c->helper[i]->post([&worker, i]() { worker(i); });
        }
        worker(0);
        for (u32 i = 1; i < ns; i++) c->helper[i]->wait();
    }
    if (err.load()) {
        // (the block CRCs run on c->side and only k5_run joins them back: on an error between the two, k0_crc may still be reading the input and
        // XOR-ing into the workspace when the call returns - ADVICE r5)
        if (c->side) (void)hipStreamSynchronize(c->side);
        return err.load();
    }
    for (u32 i = 0; i < ns; i++) {
        HIP_CHECK_RET(hipEventRecord(c->evDone[i], c->sub[i]));
        HIP_CHECK_RET(hipStreamWaitEvent(c->stream, c->evDone[i], 0));
    }
    return CJS_OK;
}

extern "C" int64_t cjs_bz2_compress_device(cjs_ctx* c, const void* d_in, uint64_t in_len, int level,
                                           void* d_out, uint64_t out_cap) {
    if (!c || (!d_in && in_len) || !d_out) return CJS_E_ARG;
    if (level < 1 || level > 9) return CJS_E_LEVEL;                  // lib/Bzip2.js:888-890
    if (out_cap < 64 || ((uintptr_t)d_out & 3)) return CJS_E_ARG;
    hipError_t e;
    int rc;
#define TRYR(x) if ((e = (x)) != hipSuccess) return CJS_E_HIP - (int)e
    TRYR(hipSetDevice(c->device));
    const u32 cap = (u32)level * 100000u - 19u;                      // lib/Bzip2.js:892-900
    hipStream_t st = c->stream;
    rc = grow(&c->k0ws, &c->k0ws_bytes, k0_bytes(in_len, cap));
    if (rc) return rc;
    K0Buf K;
    k0_carve(K, (const u8*)d_in, in_len, cap, c->k0ws);
    Pipe P0;
    memset(&P0, 0, sizeof P0);
    P0.ss = c->d_ss;
    P0.out = (u32*)d_out;
    P0.outCapBytes = out_cap & ~(uint64_t)3;
    TRYR(hipEventRecord(c->ev0, st));
    // the output is zeroed (k5_pack ORs into it) on a sub-batch stream NEXT TO the pre-pass instead of behind it (23 us per 10^8 bytes at the
    // head of every call); k5_begin and evReady come behind both
    TRYR(hipStreamWaitEvent(c->sub[0], c->ev0, 0));
    TRYR(hipMemsetAsync(P0.out, 0, P0.outCapBytes, c->sub[0]));
    TRYR(hipEventRecord(c->evDone[0], c->sub[0]));
    rc = k0_prepass(K, cap, st);
    if (rc) { (void)hipStreamSynchronize(c->sub[0]); return rc; }     // (the zeroing of the caller's buffer is still in flight - ADVICE r4)
    TRYR(hipStreamWaitEvent(st, c->evDone[0], 0));        // (k5_begin writes the stream header into the zeroed output)
    rc = k5_stream_begin(P0, level, st, false);
    if (rc) { (void)hipStreamSynchronize(c->sub[0]); return rc; }
    TRYR(hipEventRecord(c->evReady, st));
    // (both small read-backs of the call go through the context's pinned memory - stream 0's, which no sub-batch is using at either time:
    // a pageable copy costs tens of microseconds more, with the GPU idle behind the first one)
    TRYR(hipMemcpyAsync(c->pin[0], K.nBlocks, 4, hipMemcpyDeviceToHost, st));
    TRYR(hipStreamSynchronize(st));
    const u32 nblocks = c->pin[0][0];
    rc = issue_blocks(c, K, cap, 0, nblocks, d_out, out_cap);
    if (rc) return rc;
    rc = k5_stream_end(P0, st);
    if (rc) return rc;
    TRYR(hipEventRecord(c->ev1, st));
    TRYR(hipMemcpyAsync(c->pin[0], c->d_ss, sizeof(StreamState), hipMemcpyDeviceToHost, st));
    TRYR(hipStreamSynchronize(st));
    StreamState hs;
    memcpy(&hs, c->pin[0], sizeof hs);
    TRYR(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    c->last_blocks = nblocks;
    if (hs.overflow) return CJS_E_NOSPACE;
    return (int64_t)((hs.bits + 7) >> 3);
#undef TRYR
}

// Host buffers in, host buffers out, input longer than ~1.5 batches: the input is cut into SEGMENTS of about one batch
// of blocks.  A segment is planned as an input of its own (a bzip2 block always starts with a fresh RLE1 state,
// lib/Bzip2.js:636-667, so planning from a block start is exact); all of its blocks but the last are encoded, and
// the next segment starts where that last, possibly incomplete, block started.  Two helper threads move data on
// their own non-blocking streams: the uploader keeps H2D a segment ahead of the encoder, the downloader copies the
// bytes of the stream that are final (everything before the bit cursor) while the next segment is being encoded.
namespace {
struct SegCopy {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t uploaded = 0;          // input bytes resident in HBM
    uint64_t want = 0;              // downloader: copy the stream up to this byte
    bool quit = false;
    int err = 0;
};
}

static int64_t compress_segmented(cjs_ctx* c, const uint8_t* in, uint64_t in_len, int level, uint8_t* out, uint64_t out_cap,
                                  uint64_t seg_bytes) {
    hipError_t e;
#define TRYR(x) if ((e = (x)) != hipSuccess) { rc = CJS_E_HIP - (int)e; goto done; }
    const u32 cap = (u32)level * 100000u - 19u;
    int64_t rc = 0;
    SegCopy sc;
    hipStream_t sIn = nullptr, sOut = nullptr;
    std::thread up, down;
    uint64_t copied = 0;            // bytes of the stream already in `out` (downloader's own)
    const int dev = c->device;
    u8* din = (u8*)c->din;
    u8* dout = (u8*)c->dout;
    const uint64_t dout_cap = c->dout_bytes;
    float total_ms = 0.f;
    u32 total_blocks = 0;
    Pipe P0;
    memset(&P0, 0, sizeof P0);
    P0.ss = c->d_ss;
    P0.out = (u32*)dout;
    P0.outCapBytes = dout_cap & ~(uint64_t)3;
    if (hipStreamCreateWithFlags(&sIn, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&sOut, hipStreamNonBlocking) != hipSuccess) {
        if (sIn) (void)hipStreamDestroy(sIn);
        return CJS_E_HIP;
    }
    up = std::thread([&]() {
        if (hipSetDevice(dev) != hipSuccess) { std::lock_guard<std::mutex> g(sc.mu); sc.err = CJS_E_NOGPU; sc.cv.notify_all(); return; }
        const uint64_t chunk = (uint64_t)32 << 20;
        for (uint64_t off = 0; off < in_len; off += chunk) {
            const uint64_t len = in_len - off < chunk ? in_len - off : chunk;
            hipError_t e2 = hipMemcpyAsync(din + off, in + off, len, hipMemcpyHostToDevice, sIn);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(sIn);
            std::lock_guard<std::mutex> g(sc.mu);
            if (e2 != hipSuccess) { sc.err = CJS_E_HIP - (int)e2; sc.cv.notify_all(); return; }
            sc.uploaded = off + len;
            sc.cv.notify_all();
            if (sc.quit) return;
        }
    });
    down = std::thread([&]() {
        if (hipSetDevice(dev) != hipSuccess) return;
        for (;;) {
            uint64_t to;
            {
                std::unique_lock<std::mutex> g(sc.mu);
                sc.cv.wait(g, [&]() { return sc.quit || sc.want > copied; });
                if (sc.want <= copied) return;          // quit and nothing left
                to = sc.want;
            }
            if (to > out_cap) { std::lock_guard<std::mutex> g(sc.mu); sc.err = CJS_E_NOSPACE; return; }
            hipError_t e2 = hipMemcpyAsync(out + copied, dout + copied, to - copied, hipMemcpyDeviceToHost, sOut);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(sOut);
            if (e2 != hipSuccess) { std::lock_guard<std::mutex> g(sc.mu); sc.err = CJS_E_HIP - (int)e2; return; }
            copied = to;
        }
    });
    {
        hipStream_t st = c->stream;
        int rci = k5_stream_begin(P0, level, st);
        if (rci) { rc = rci; goto done; }
        uint64_t s = 0;
        u32 seg = 0;
        while (s < in_len || (in_len == 0 && seg == 0)) {
            uint64_t eoff = s + seg_bytes < in_len ? s + seg_bytes : in_len;
            {   // wait until the uploader has brought the segment in
                std::unique_lock<std::mutex> g(sc.mu);
                sc.cv.wait(g, [&]() { return sc.err || sc.uploaded >= eoff; });
                if (sc.err) { rc = sc.err; goto done; }
            }
            void** kws = seg & 1 ? &c->planws : &c->k0ws;            // alternate: the previous segment's tables may still be read
            size_t* kwb = seg & 1 ? &c->planws_bytes : &c->k0ws_bytes;
            if (seg & 1) { c->plan_level = 0; c->plan_blocks = 0; c->scan_level = 0; }  // a cjs_bz2_plan / _plan_scan result does not survive this call
            rci = grow(kws, kwb, k0_bytes(eoff - s, cap));
            if (rci) { rc = rci; goto done; }
            K0Buf K;
            k0_carve(K, din + s, eoff - s, cap, *kws);
            TRYR(hipEventRecord(c->ev0, st));
            rci = k0_prepass(K, cap, st);
            if (rci) { rc = rci; goto done; }
            TRYR(hipEventRecord(c->evReady, st));
            u32 nblocks = 0;
            TRYR(hipMemcpyAsync(&nblocks, K.nBlocks, 4, hipMemcpyDeviceToHost, st));
            TRYR(hipStreamSynchronize(st));
            u32 keep = nblocks;
            uint64_t next = in_len;
            if (eoff < in_len) {
                if (nblocks < 2) { seg_bytes *= 2; continue; }       // run-heavy input: one block swallowed the segment
                keep = nblocks - 1;
                uint64_t bs = 0;
                TRYR(hipMemcpy(&bs, K.blkStart + keep, 8, hipMemcpyDeviceToHost));
                next = s + bs;
            }
            rci = issue_blocks(c, K, cap, 0, keep, dout, dout_cap);
            if (rci) { rc = rci; goto done; }
            TRYR(hipEventRecord(c->ev1, st));
            StreamState hs;
            TRYR(hipMemcpyAsync(&hs, c->d_ss, sizeof hs, hipMemcpyDeviceToHost, st));
            TRYR(hipStreamSynchronize(st));
            float ms = 0.f;
            TRYR(hipEventElapsedTime(&ms, c->ev0, c->ev1));
            total_ms += ms;
            total_blocks += keep;
            if (hs.overflow) { rc = CJS_E_NOSPACE; goto done; }
            {   // everything before the bit cursor is final
                std::lock_guard<std::mutex> g(sc.mu);
                if (sc.err) { rc = sc.err; goto done; }
                sc.want = hs.bits >> 3;
                sc.cv.notify_all();
            }
            s = next;
            seg++;
            if (in_len == 0) break;
        }
        rci = k5_stream_end(P0, st);
        if (rci) { rc = rci; goto done; }
        StreamState hs;
        TRYR(hipMemcpyAsync(&hs, c->d_ss, sizeof hs, hipMemcpyDeviceToHost, st));
        TRYR(hipStreamSynchronize(st));
        if (hs.overflow) { rc = CJS_E_NOSPACE; goto done; }
        rc = (int64_t)((hs.bits + 7) >> 3);
        if ((uint64_t)rc > out_cap) { rc = CJS_E_NOSPACE; goto done; }
        {
            std::lock_guard<std::mutex> g(sc.mu);
            sc.want = (uint64_t)rc;
            sc.cv.notify_all();
        }
    }
done:
    {
        std::lock_guard<std::mutex> g(sc.mu);
        sc.quit = true;
        sc.cv.notify_all();
    }
    if (up.joinable()) up.join();
    if (down.joinable()) down.join();
    (void)hipStreamDestroy(sIn);
    (void)hipStreamDestroy(sOut);
    if (rc >= 0 && sc.err) rc = sc.err;
    c->last_ms = total_ms;
    c->last_blocks = total_blocks;
    return rc;
#undef TRYR
}


#ifndef CJS_CPU_DEBUG_BUILD
static int64_t compress_overlapped(cjs_ctx* c, const uint8_t* in, uint64_t in_len, int level, uint8_t* out, uint64_t out_cap, u32 slice_blocks);
#endif
#define CJS_OV_NOT_STARTED (-1000)   // compress_overlapped: the input does not cut into slices - nothing was uploaded (internal: never returned by the ABI)
extern "C" int64_t cjs_bz2_compress(cjs_ctx* c, const uint8_t* in, uint64_t in_len, int level, uint8_t* out,
                                    uint64_t out_cap) {
    if (!c || (!in && in_len) || !out) return CJS_E_ARG;
    if (level < 1 || level > 9) return CJS_E_LEVEL;
    hipError_t e;
#define TRYR(x) if ((e = (x)) != hipSuccess) return CJS_E_HIP - (int)e
    TRYR(hipSetDevice(c->device));
    const uint64_t need = (uint64_t)cjs_bz2_compress_bound(in_len);
    int rc = grow(&c->din, &c->din_bytes, in_len + 64);
    if (rc) return rc;
    rc = grow(&c->dout, &c->dout_bytes, need);
    if (rc) return rc;
    // one batch of blocks per segment; CJS_SEG_BYTES (tests) overrides.  Inputs of up to a batch and a half go in one
    // piece: half-size batches cost more (16.9 vs 14.2 ms per 10^8 bytes) than overlapping their copies would save.
    const uint64_t seg_env = []() -> uint64_t { const char* ev = getenv("CJS_SEG_BYTES"); return ev ? strtoull(ev, nullptr, 10) : 0; }();   // (read per call)
    const uint64_t seg_bytes = seg_env ? seg_env : (uint64_t)c->batch_blocks * ((u32)level * 100000u - 19u);
    bool resident = false;
#ifndef CJS_CPU_DEBUG_BUILD
    {
        // the overlapped path: slices of CJS_SLICE_BLOCKS blocks - any input of at least two slices.  OFF by default (0): measured on the MI355X
        // (10^8 bytes of enwik, device-resident step 7.5 ms, one piece 9.92 ms host to host): 2 streams x slices of 28 / 40 / 56 blocks 10.9 /
        // 10.0 / 10.2 ms, 3 streams x 40 9.65, explicit 64+48 10.0, 4 streams 10.4 - a slice of 28 blocks takes 3.5-4.5 ms from its first
        // launch to its last however few blocks it has (a chain of ~160 launches, K1's read-back in the middle; tests/gpu_host_path_probe.py
        // with CJS_OV_TRACE=1 prints the timeline), so what the copies gain the shorter sub-batches lose.  Same bytes either way.
        static const u32 slice_blocks = []() -> u32 { const char* ev = getenv("CJS_SLICE_BLOCKS"); return ev ? (u32)strtoul(ev, nullptr, 10) : 0u; }();
        const u32 cap = (u32)level * 100000u - 19u;
        if (slice_blocks && !seg_env && in_len >= (uint64_t)slice_blocks * cap * 3 / 2) {
            const int64_t r = compress_overlapped(c, in, in_len, level, out, out_cap, slice_blocks);
            if (r != CJS_E_SPEC && r != CJS_OV_NOT_STARTED) return r;
            resident = r == CJS_E_SPEC;                     // a slice refused its plan (rare): the uploader has run to its end, the input is in HBM - one piece from here
        }
    }
#endif
    if (!resident && in_len > seg_bytes + seg_bytes / 2) return compress_segmented(c, in, in_len, level, out, out_cap, seg_bytes);
    if (!resident && in_len) TRYR(hipMemcpyAsync(c->din, in, in_len, hipMemcpyHostToDevice, c->stream));
    const int64_t n = cjs_bz2_compress_device(c, c->din, in_len, level, c->dout, c->dout_bytes);
    if (n < 0) return n;
    if ((uint64_t)n > out_cap) return CJS_E_NOSPACE;
    TRYR(hipMemcpy(out, c->dout, (size_t)n, hipMemcpyDeviceToHost));
    return n;
#undef TRYR
}

// ---- Bzip2.compressFile over several GPUs of one node, from ONE process (the Node addon's path to N devices) -----
// The input is cut at the nominal segment ends E_k = (k+1) * seg_bytes; segment k goes to device k mod n.  Round 4: the plan of
// a segment no longer waits for the plan of the segment before it (the reference's `do { readBlock } while`, lib/Bzip2.js:913-922,
// is serial; rounds 2-3 kept it serial across the segments).  Segments are taken in waves of n (one per device), and per wave:
//   A. every device uploads its window - the segment and the margin BEHIND it, [E_(k-1), E_k + W) - and scans it (K0's RLE1 cost
//      prefix, cjs_bz2_plan_scan): the segment's cost total; the host reads the segment's first and last run off the input;
//   B. from those summaries (the integer arithmetic of compressjs_amd/dist.py:plan_bases, restated below) every segment gets the
//      phase of the block boundaries inside it, plans the blocks that START in it (cjs_bz2_plan_phase) and encodes them from bit 0
//      into a buffer of its own, the margin completing the last one.
// When all bit lengths are known the segments are shifted on their devices to their bit offsets and copied side by side into the
// caller's buffer (parallel D2H, no peer traffic); the bytes two segments share are OR-ed on the host, which also folds the
// combined CRC (linear over GF(2)) and writes header and trailer.  A segment that cannot be planned on its own (a block boundary
// inside a run of four or more bytes that straddles a cut, a block longer than the margin, a boundary run beyond 4 KB) makes the
// call fall back to one device.  = SURVEY.md 8(e) without torch.distributed.
namespace {
struct MSeg {
    uint64_t lo = 0, e = 0;         // the segment's bytes [lo, e) (absolute input offsets)
    uint64_t cost = 0;              // RLE1 cost of its bytes, scanned as an input of its own
    u32 phase = 0;                  // block boundaries lie where the slice's own cost prefix reaches phase + m * cap - while no boundary in front of the segment moved
    uint64_t base = 0;              // origin of the segment's own cost prefix in the stream's: G(lo + i) = base + C_own(i) beyond the head run
    bool has_span = false;          // a run of four or more bytes straddles the segment's start: its cost span [c0, c1] from the run's first byte
    uint64_t c0 = 0, c1 = 0;        //   (the own prefix is wrong inside it: a boundary target in there cannot be planned by this segment)
    uint64_t t0 = 0, tnext = 0;     // round 6, the chained plan: the target the segment was planned from, the one it hands on
    int64_t nb = 0;                 // blocks that start in the segment
    bool ok = true;                 // can be planned on its own
    uint64_t bits = 0, off = 0;
    u32 fold = 0, count = 0;
    u8* dseg = nullptr;             // device: the segment's bit stream from bit 0
    uint64_t dseg_cap = 0;
    u8 first = 0, last = 0;         // seam bytes (shifted)
};
static inline u32 rotl32(u32 v, u32 k) { k &= 31u; return k ? (v << k) | (v >> (32u - k)) : v; }
// RLE1 output bytes of the first k bytes of a fresh run (k0_g in k0_rle1.hip)
static inline uint64_t rle1_g(uint64_t k) { const uint64_t q = k / 255u, r = k % 255u; return 5u * q + (r < 4u ? r : 5u); }
// first and last run of in[lo, e), from its first / last 4096 bytes; long_run: a boundary run that reaches beyond them
struct EdgeRuns { int hb = -1, tb = -1; uint64_t lh = 0, lt = 0; bool long_run = false; };
static EdgeRuns edge_runs(const uint8_t* in, uint64_t lo, uint64_t e) {
    EdgeRuns r;
    const uint64_t n = e - lo, EDGE = 4096;
    if (!n) return r;
    const uint64_t hn = n < EDGE ? n : EDGE;
    r.hb = in[lo]; r.tb = in[e - 1];
    while (r.lh < hn && in[lo + r.lh] == (uint8_t)r.hb) r.lh++;
    while (r.lt < hn && in[e - 1 - r.lt] == (uint8_t)r.tb) r.lt++;
    r.long_run = (r.lh == hn && n > hn) || (r.lt == hn && n > hn);
    return r;
}
// compressjs_amd/dist.py:plan_bases for one more segment: G = cost prefix of the stream at the segment's start, (inb, ink) = the
// run that reaches it from the left
struct PlanChain { uint64_t G = 0; int inb = -1; uint64_t ink = 0; };
static void plan_base(PlanChain& P, MSeg& g, const EdgeRuns& r, u32 cap) {
    const uint64_t n = g.e - g.lo;
    g.ok = !r.long_run;
    int64_t delta = 0;
    if (n && P.inb == r.hb && P.ink > 0) {
        delta = (int64_t)rle1_g(P.ink + r.lh) - (int64_t)rle1_g(P.ink) - (int64_t)rle1_g(r.lh);      // the head run costs what the tail of a longer run costs
        if (P.ink + r.lh >= 4) {
            // a run of four or more bytes straddles the cut: no block boundary may fall into its cost span, measured from the run's first byte
            // (round 6: checked against the target the chain actually carries to this segment - seg_target - not against multiples of cap)
            g.has_span = true;
            g.c0 = P.G - rle1_g(P.ink);
            g.c1 = P.G + rle1_g(P.ink + r.lh) - rle1_g(P.ink);
        }
    }
    const uint64_t base = (uint64_t)((int64_t)P.G + delta);
    g.base = base;
    g.phase = (u32)((cap - base % cap) % cap);
    if (n) {
        if (r.lh == n && P.inb == r.hb && P.ink > 0) P.ink += n;       // the whole segment continues the incoming run
        else if (r.lh == n) { P.inb = r.hb; P.ink = n; }
        else { P.inb = r.tb; P.ink = r.lt; }
    }
    P.G = (uint64_t)((int64_t)P.G + (int64_t)g.cost + delta);
}
// The chained plan (compressjs_amd/dist.py: chain_step): tau = the stream-wide target of the first block boundary at or behind the
// segment's start (what the segment before it handed on; 0 for the first).  Sets g.t0, the same target under the segment's own origin;
// false when the boundary falls into a run that straddles the segment's start (the segment cannot plan it: the caller falls back).
static bool seg_target(MSeg& g, uint64_t tau) {
    if (g.has_span && tau >= g.c0 && tau <= g.c1) return false;
    g.t0 = tau > g.base ? tau - g.base : 0;
    return true;
}
}

// ---- host buffers in, host buffers out, ONE call, copies under the kernels (round 5) ----------------------------------------------
// lib/Bzip2.js:879-929 + lib/Util.js:9-103 define the product as host Buffer -> host Buffer.  Until round 4 a call that fits a batch and a
// half was upload-everything, encode, download-everything: 2.6 ms of PCIe around an 8.4 ms step (0.78 of the device-resident rate), and
// cutting it into the segments of compress_segmented cost more than their overlap saved (every segment a pre-pass, three read-backs and a
// drained GPU of its own).  Here the input is cut into SLICES of a few dozen blocks that flow through the context's streams without a gap:
//   uploader    (thread, own stream) brings the input in, in order, 8 MB at a time;
//   planner     (the caller's thread) - as soon as a slice and the margin behind it are resident: K0's tile scans over that window, the
//               slice's RLE1 cost, and from the cost prefix of the stream so far (plan_base: the arithmetic of the multi-device plan) the
//               phase of the block boundaries inside it: the blocks that START in the slice (k0_phase_plan); no chain through earlier plans;
//   encoders    (one thread per stream) run slice after slice as ONE sub-batch each, alternating streams; the only dependency between
//               slices is the bit cursor (k5_blockscan of slice j waits for that of slice j - 1: an event), as between sub-batches before;
//   downloader  (thread, own stream) copies every byte in front of the cursor a finished slice left (k5_blockscan drops it into pinned
//               memory) while later slices are encoded.
// A slice that cannot be planned on its own (a boundary inside a run that straddles a cut, a block longer than the margin) sends the
// call down the one-piece path.  Same bytes as cjs_bz2_compress_device on the whole input (tests/test_gpu_parity.py).
namespace {
struct OvSlice {
    uint64_t lo = 0, e = 0, wend = 0;   // bytes [lo, e), window end (margin included)
    K0Buf K;
    u32 nblocks = 0;
    hipEvent_t evScan = nullptr, evDone = nullptr;
    float t_up = 0, t_plan = 0, t_iss0 = 0, t_iss1 = 0, t_done = 0, t_copied = 0;   // CJS_OV_TRACE: ms since the call began
};
}
#ifndef CJS_CPU_DEBUG_BUILD
static int64_t compress_overlapped(cjs_ctx* c, const uint8_t* in, uint64_t in_len, int level, uint8_t* out, uint64_t out_cap,
                                   u32 slice_blocks) {
    const u32 cap = (u32)level * 100000u - 19u;
    const uint64_t W = (uint64_t)4 * (cap + 19u);
    if (slice_blocks > c->sub_blocks) slice_blocks = c->sub_blocks;
    // the first slice is half a slice: the encoders start after 1/8 ... of the upload instead of 1/4
    std::vector<OvSlice> S;
    {
        const uint64_t sb = (uint64_t)slice_blocks * cap;
        uint64_t lo = 0, len = sb / 2 > W ? sb / 2 : sb;
        // CJS_SLICE_LIST="64,48": explicit slice lengths in blocks (experiments); the last one takes the rest
        std::vector<uint64_t> lens;
        if (const char* ev = getenv("CJS_SLICE_LIST"))
            for (const char* q = ev; *q;) { lens.push_back((uint64_t)strtoul(q, (char**)&q, 10) * cap); if (*q == ',') q++; }
        size_t li = 0;
        if (!lens.empty()) len = lens[li++];
        while (lo < in_len) {
            OvSlice sl;
            sl.lo = lo;
            sl.e = lo + len < in_len ? lo + len : in_len;
            if (lens.empty() && in_len - sl.e < sb / 4) sl.e = in_len;          // no stub at the end
            sl.wend = sl.e + W < in_len ? sl.e + W : in_len;
            S.push_back(sl);
            lo = sl.e;
            len = lens.empty() ? sb : (li < lens.size() ? lens[li++] : in_len);
        }
    }
    const u32 ns = (u32)S.size();
    if (ns < 2 || ns > CJS_SNAP_SLOTS) return CJS_OV_NOT_STARTED;      // (nothing uploaded: the caller must not take the input for resident - ADVICE r5)
    hipError_t e;
#define TRYR(x) if ((e = (x)) != hipSuccess) return CJS_E_HIP - (int)e
    if (!c->sIn) TRYR(hipStreamCreateWithFlags(&c->sIn, hipStreamNonBlocking));
    if (!c->sOut) TRYR(hipStreamCreateWithFlags(&c->sOut, hipStreamNonBlocking));
    if (!c->snapPin) TRYR(hipHostMalloc((void**)&c->snapPin, CJS_SNAP_SLOTS * 8));
    if (!c->evPool) c->evPool = new std::vector<hipEvent_t>();
    while (c->evPool->size() < 2u * ns) {
        hipEvent_t ev;
        TRYR(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c->evPool->push_back(ev);
    }
    for (int i = 0; i < 3; i++) if (!c->io[i]) c->io[i] = new CjsHelper();
    for (u32 i = 1; i < c->nstreams; i++) if (!c->helper[i]) c->helper[i] = new CjsHelper();
    size_t k0each = 0;
    for (u32 k = 0; k < ns; k++) { const size_t b = (k0_bytes(S[k].wend - S[k].lo, cap) + 255) & ~(size_t)255; k0each = b > k0each ? b : k0each; }
    int rc = grow(&c->k0sl, &c->k0sl_bytes, k0each * ns);
    if (rc) return rc;
    c->plan_level = 0; c->plan_blocks = 0; c->scan_level = 0;
    u8* din = (u8*)c->din;
    u8* dout = (u8*)c->dout;
    const uint64_t dout_cap = c->dout_bytes;
    const int dev = c->device;
    const u32 nst = c->nstreams;
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_ms = [&]() { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    static const bool ov_trace = getenv("CJS_OV_TRACE") != nullptr;

    std::mutex mu;
    std::condition_variable cv;
    uint64_t uploaded = 0;
    u32 planned = 0;                    // slices whose plan is ready
    u32 issued = 0;                     // slices whose launches are all enqueued (evDone recorded)
    bool stop = false;                  // error or fall-back: everybody winds down
    int err = 0;
    auto fail = [&](int code) { std::lock_guard<std::mutex> g(mu); if (!err) err = code; stop = true; cv.notify_all(); };

    // ---- uploader
    c->io[0]->post([&]() {
        if (hipSetDevice(dev) != hipSuccess) { fail(CJS_E_NOGPU); return; }
        const uint64_t chunk = (uint64_t)8 << 20;
        for (uint64_t off = 0; off < in_len; off += chunk) {
            const uint64_t len = in_len - off < chunk ? in_len - off : chunk;
            hipError_t e2 = hipMemcpyAsync(din + off, in + off, len, hipMemcpyHostToDevice, c->sIn);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->sIn);
            if (e2 != hipSuccess) { fail(CJS_E_HIP - (int)e2); return; }
            std::lock_guard<std::mutex> g(mu);
            uploaded = off + len;
            cv.notify_all();
            if (err) return;                               // (a fall-back still wants the whole input resident)
        }
    });
    // ---- downloader: the bytes in front of the cursor of every finished slice
    uint64_t copied = 0;
    int64_t result = -1;
    c->io[1]->post([&]() {
        if (hipSetDevice(dev) != hipSuccess) { fail(CJS_E_NOGPU); return; }
        for (u32 k = 0; k < ns; k++) {
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&]() { return stop || issued > k; });
                if (stop) return;
            }
            hipError_t e2 = hipEventSynchronize(S[k].evDone);
            if (e2 != hipSuccess) { fail(CJS_E_HIP - (int)e2); return; }
            S[k].t_done = now_ms();
            const uint64_t to = c->snapPin[k] >> 3;
            if (to > out_cap) { fail(CJS_E_NOSPACE); return; }
            if (to > copied) {
                e2 = hipMemcpyAsync(out + copied, dout + copied, to - copied, hipMemcpyDeviceToHost, c->sOut);
                if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->sOut);
                if (e2 != hipSuccess) { fail(CJS_E_HIP - (int)e2); return; }
                copied = to;
            }
            S[k].t_copied = now_ms();
        }
    });
    // ---- encoders: stream si takes slices si, si + nst, ...; the cursor events are recorded and waited for in slice order
    std::atomic<u32> recorded(0);
    const BatchGeom g = make_geom(c->sub_blocks, cap);
    auto encoder = [&](u32 si) {
        if (hipSetDevice(dev) != hipSuccess) { fail(CJS_E_NOGPU); return; }
        for (u32 k = si; k < ns; k += nst) {
            {
                std::unique_lock<std::mutex> gq(mu);
                cv.wait(gq, [&]() { return stop || planned > k; });
                if (stop) return;
            }
            OvSlice& sl = S[k];
            int r2 = CJS_OK;
            Pipe P;
            sl.t_iss0 = now_ms();
            if (sl.nblocks) {
                if (k < nst) r2 = hipStreamWaitEvent(c->sub[si], c->evReady, 0) == hipSuccess ? CJS_OK : CJS_E_HIP;
                if (!r2) r2 = run_sub_batch(c, sl.K, g, cap, 0, sl.nblocks, si, dout, dout_cap, P, sl.nblocks);
            }
            while (recorded.load(std::memory_order_acquire) != k) {
                std::this_thread::yield();
                std::lock_guard<std::mutex> gq(mu);
                if (stop) return;
            }
            if (!r2) {
                if (sl.nblocks) {
                    P.snap = c->snapPin + k;
                    r2 = k5_run(P, cap, c->sub[si], k ? S[k - 1].evScan : nullptr, sl.evScan, c->side ? c->evCrc[si] : nullptr);
                } else {
                    // (a slice in which no block starts: the cursor stays where it is)
                    // (the cursor is carried IN STREAM ORDER behind the previous slice's k5_blockscan: read on the host at issue time, slot k - 1
                    // may not have been written yet, and the downloader would take stale bytes for final - ADVICE r5)
                    if (k && hipStreamWaitEvent(c->sub[si], S[k - 1].evScan, 0) != hipSuccess) r2 = CJS_E_HIP;
                    if (!k) c->snapPin[0] = 32u;
                    else if (!r2 && hipMemcpyAsync(c->snapPin + k, c->snapPin + k - 1, 8, hipMemcpyHostToHost, c->sub[si]) != hipSuccess) r2 = CJS_E_HIP;
                    if (!r2 && hipEventRecord(sl.evScan, c->sub[si]) != hipSuccess) r2 = CJS_E_HIP;
                }
            }
            if (!r2 && hipEventRecord(sl.evDone, c->sub[si]) != hipSuccess) r2 = CJS_E_HIP;
            sl.t_iss1 = now_ms();
            recorded.store(k + 1, std::memory_order_release);
            if (r2) { fail(r2); return; }
            std::lock_guard<std::mutex> gq(mu);
            if (k + 1 > issued) issued = k + 1;             // (slices finish issuing in any order; the downloader walks them in order and waits per slice)
            cv.notify_all();
        }
    };
    // (issued > k implies that slice k's evDone is recorded: a later slice only gets past `recorded` behind it)
    c->io[2]->post([&]() { encoder(0); });
    for (u32 i = 1; i < nst; i++) c->helper[i]->post([&, i]() { encoder(i); });

    // ---- planner (this thread)
    hipStream_t st = c->stream;
    bool fallback = false;
    {
        Pipe P0;
        memset(&P0, 0, sizeof P0);
        P0.ss = c->d_ss;
        P0.out = (u32*)dout;
        P0.outCapBytes = dout_cap & ~(uint64_t)3;
        rc = k5_stream_begin(P0, level, st, true);
        if (!rc && hipEventRecord(c->evReady, st) != hipSuccess) rc = CJS_E_HIP;
        if (rc) fail(rc);
        PlanChain pc;
        uint64_t tau = 0;                                   // stream-wide target of the next block boundary (round 6: carried from slice to slice)
        for (u32 k = 0; k < ns && !rc; k++) {
            OvSlice& sl = S[k];
            sl.evScan = (*c->evPool)[2 * k];
            sl.evDone = (*c->evPool)[2 * k + 1];
            {
                std::unique_lock<std::mutex> gq(mu);
                cv.wait(gq, [&]() { return stop || uploaded >= sl.wend; });
                if (stop) break;
            }
            sl.t_up = now_ms();
            const uint64_t wlen = sl.wend - sl.lo, own = sl.e - sl.lo;
            k0_carve(sl.K, din + sl.lo, wlen, cap, (char*)c->k0sl + (size_t)k * k0each);
            rc = k0_scans(sl.K, st);
            if (!rc && own < wlen) rc = k0_eval(sl.K, own, st);
            if (rc) break;
            u64 tot[2] = {0, 0};                            // [0] cost of the window, [1] of the slice
            if (hipMemcpyAsync(&tot[0], sl.K.tileC + sl.K.ntiles, 8, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = CJS_E_HIP; break; }
            if (own < wlen && hipMemcpyAsync(&tot[1], sl.K.specC, 8, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = CJS_E_HIP; break; }
            if (hipStreamSynchronize(st) != hipSuccess) { rc = CJS_E_HIP; break; }
            MSeg ms;
            ms.lo = sl.lo; ms.e = sl.e;
            ms.cost = own < wlen ? tot[1] : tot[0];
            plan_base(pc, ms, edge_runs(in, sl.lo, sl.e), cap);
            if (!ms.ok || !seg_target(ms, tau)) { fallback = true; break; }
            rc = k0_phase_plan(sl.K, cap, ms.t0, own, sl.wend >= in_len ? 1u : 0u, tot[0], st);
            if (rc) break;
            u64 nbt[2] = {0, 0};                            // blocks, the target handed on (k0_phase_chain)
            if (hipMemcpyAsync(nbt, sl.K.nBlocks, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = CJS_E_HIP; break; }
            const u32 nb = (u32)nbt[0];
            if (nb == K0_PHASE_FAIL || nb > c->sub_blocks) { fallback = true; break; }
            tau = ms.base + nbt[1];                         // the slices are planned in order: the chain needs no speculation here
            sl.nblocks = nb;
            sl.t_plan = now_ms();
            std::lock_guard<std::mutex> gq(mu);
            planned = k + 1;
            cv.notify_all();
        }
        if (rc) fail(rc);
        if (fallback) { std::lock_guard<std::mutex> gq(mu); stop = true; cv.notify_all(); }
    }
    // ---- wind down
    c->io[2]->wait();
    for (u32 i = 1; i < nst; i++) c->helper[i]->wait();
    c->io[1]->wait();
    if (!stop) {
        // trailer behind the last slice, then the rest of the stream
        Pipe P0;
        memset(&P0, 0, sizeof P0);
        P0.ss = c->d_ss;
        P0.out = (u32*)dout;
        P0.outCapBytes = dout_cap & ~(uint64_t)3;
        for (u32 i = 0; i < nst && i < ns; i++) if (hipStreamWaitEvent(st, S[ns - 1 - i].evDone, 0) != hipSuccess) rc = CJS_E_HIP;
        if (!rc) rc = k5_stream_end(P0, st);
        StreamState hs;
        memset(&hs, 0, sizeof hs);
        if (!rc && (hipMemcpyAsync(&hs, c->d_ss, sizeof hs, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) rc = CJS_E_HIP;
        if (!rc && hs.overflow) rc = CJS_E_NOSPACE;
        if (!rc) {
            const uint64_t total = (hs.bits + 7) >> 3;
            if (total > out_cap) rc = CJS_E_NOSPACE;
            else if (total > copied && hipMemcpy(out + copied, dout + copied, total - copied, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
            result = (int64_t)total;
        }
        if (rc) fail(rc);
    }
    c->io[0]->wait();                                       // (the uploader reads the caller's buffer: never leave it running)
    (void)hipDeviceSynchronize();
    u32 blocks = 0;
    for (u32 k = 0; k < ns; k++) blocks += S[k].nblocks;
    c->last_blocks = blocks;
    c->last_ms = now_ms();
    if (ov_trace) {
        fprintf(stderr, "[ov] %u slices, %.2f ms:", ns, c->last_ms);
        for (u32 k = 0; k < ns; k++) fprintf(stderr, " [%u: %u blk up %.2f plan %.2f issue %.2f-%.2f done %.2f copied %.2f]", k, S[k].nblocks, S[k].t_up, S[k].t_plan, S[k].t_iss0, S[k].t_iss1, S[k].t_done, S[k].t_copied);
        fprintf(stderr, "\n");
    }
    if (err) return err;
    if (fallback || stop) return CJS_E_SPEC;
    return result;
#undef TRYR
}
#endif


extern "C" int64_t cjs_bz2_compress_multi(cjs_ctx** ctxs, uint32_t n, const uint8_t* in, uint64_t in_len, int level,
                                          uint8_t* out, uint64_t out_cap) {
    if (!ctxs || n == 0 || !ctxs[0] || (!in && in_len) || !out) return CJS_E_ARG;
    if (level < 1 || level > 9) return CJS_E_LEVEL;
    for (u32 i = 0; i < n; i++) if (!ctxs[i]) return CJS_E_ARG;
    const u32 cap = (u32)level * 100000u - 19u;
    const uint64_t seg_env = []() -> uint64_t { const char* ev = getenv("CJS_SEG_BYTES"); return ev ? strtoull(ev, nullptr, 10) : 0; }();   // (read per call)
    const uint64_t seg_bytes = seg_env ? seg_env : (uint64_t)ctxs[0]->batch_blocks * cap;
    uint64_t nseg = (in_len + seg_bytes - 1) / seg_bytes;
    if (n == 1 || nseg <= 1) return cjs_bz2_compress(ctxs[0], in, in_len, level, out, out_cap);
    const uint64_t W = (uint64_t)4 * (cap + 19u);
    std::vector<MSeg> S(nseg);
    for (uint64_t k = 0; k < nseg; k++) {
        S[k].lo = k * seg_bytes;
        S[k].e = (k + 1) * seg_bytes < in_len ? (k + 1) * seg_bytes : in_len;
    }
    std::mutex mu;
    const auto t_begin = std::chrono::steady_clock::now();
    std::atomic<int> err{0};        // first error; 1 = take the replicated plan (the workers read it without the lock)
    auto fail = [&](int code) { std::lock_guard<std::mutex> g(mu); if (!err.load()) err.store(code); };
    // phase A of one segment on its context: window into HBM, cost scan
    auto scan_seg = [&](uint64_t k) {
        cjs_ctx* c = ctxs[k % n];
        MSeg& g = S[k];
        if (hipSetDevice(c->device) != hipSuccess) { fail(CJS_E_NOGPU); return; }
        const uint64_t we = g.e + W < in_len ? g.e + W : in_len, wlen = we - g.lo;
        int rc = grow(&c->din, &c->din_bytes, wlen + 64);
        if (rc) { fail(rc); return; }
        hipError_t e2 = hipMemcpyAsync(c->din, in + g.lo, wlen, hipMemcpyHostToDevice, c->stream);
        if (e2 != hipSuccess) { fail(CJS_E_HIP - (int)e2); return; }
        int64_t v = cjs_bz2_plan_scan(c, c->din, wlen, level);
        if (v >= 0) v = cjs_bz2_plan_cost(c, g.e - g.lo);
        if (v < 0) { (void)hipStreamSynchronize(c->stream); fail((int)v); return; }     // (the window's upload reads the caller's buffer: not left pending - ADVICE r4)
        g.cost = (uint64_t)v;
    };
    // phase B1: the blocks that START in the segment, planned from the target g.t0 (round 6: a link of the chain - cjs_bz2_plan_chain;
    // the segments of a wave plan at once from speculative targets, the calling thread validates the chain and plans a segment whose
    // target was wrong again)
    auto plan_seg = [&](uint64_t k) {
        cjs_ctx* c = ctxs[k % n];
        MSeg& g = S[k];
        if (hipSetDevice(c->device) != hipSuccess) { fail(CJS_E_NOGPU); return; }
        const uint64_t we = g.e + W < in_len ? g.e + W : in_len;
        uint64_t tn = g.t0;
        const int64_t nb = cjs_bz2_plan_chain(c, g.e - g.lo, g.t0, we >= in_len ? 1 : 0, &tn);
        if (nb == CJS_E_SPEC) { fail(1); return; }
        if (nb < 0) { fail((int)nb); return; }
        g.nb = nb; g.tnext = tn;
    };
    // phase B2: encoded from bit 0
    auto encode_seg = [&](uint64_t k) {
        cjs_ctx* c = ctxs[k % n];
        MSeg& g = S[k];
        if (g.nb == 0) return;
        if (hipSetDevice(c->device) != hipSuccess) { fail(CJS_E_NOGPU); return; }
        if (!g.dseg) { fail(CJS_E_HIP - (int)hipErrorOutOfMemory); return; }
        u32 fold = 0, cnt = 0;
        const int64_t bits = cjs_bz2_encode_blocks(c, 0, (u32)g.nb, g.dseg, g.dseg_cap, &fold, &cnt);
        if (bits < 0) { fail((int)bits); return; }
        g.bits = (uint64_t)bits; g.fold = fold; g.count = cnt;
    };
    // phase 2: shift to the bit offset, copy the inner bytes into place, fetch the two seam bytes
    auto place_seg = [&](uint64_t k) {
        cjs_ctx* c = ctxs[k % n];
        MSeg& g = S[k];
        if (!g.bits) return;
        if (hipSetDevice(c->device) != hipSuccess) { fail(CJS_E_NOGPU); return; }
        const uint64_t nbytes = (g.bits + 7) >> 3, sh = g.off & 7u;
        const uint64_t fb = g.off >> 3, lb = (g.off + g.bits - 1) >> 3, span = lb - fb + 1;
        int rc = grow(&c->dout, &c->dout_bytes, nbytes + 64);
        if (rc) { fail(rc); return; }
        rc = cjs_shift_bits(c, g.dseg, nbytes, (u32)sh, (u8*)c->dout);
        if (rc) { fail(rc); return; }
        hipError_t e2 = hipMemcpy(&g.first, c->dout, 1, hipMemcpyDeviceToHost);
        if (e2 == hipSuccess) e2 = hipMemcpy(&g.last, (u8*)c->dout + span - 1, 1, hipMemcpyDeviceToHost);
        if (e2 == hipSuccess && span > 2) e2 = hipMemcpy(out + fb + 1, (u8*)c->dout + 1, span - 2, hipMemcpyDeviceToHost);
        if (e2 != hipSuccess) fail(CJS_E_HIP - (int)e2);
    };
    // one PERSISTENT worker thread per device (the context's io[0] helper: created at its first job, kept until cjs_destroy) - until
    // round 5 a std::thread per device per phase per wave
    auto run_range = [&](uint64_t k0, uint64_t k1, const std::function<void(uint64_t)>& fn) {
#ifdef CJS_CPU_DEBUG_BUILD
        for (uint64_t k = k0; k < k1 && !err; k++) fn(k);             // the CPU logic-debug build runs kernels on one thread
#else
        const u32 nw = k1 - k0 < n ? (u32)(k1 - k0) : n;
        for (u32 d = 1; d < nw; d++) {
            cjs_ctx* c = ctxs[(k0 + d) % n];
            if (!c->io[0]) c->io[0] = new CjsHelper();
            c->io[0]->post([&, d]() { for (uint64_t k = k0 + d; k < k1 && !err; k += n) fn(k); });
        }
        for (uint64_t k = k0; k < k1 && !err; k += n) fn(k);          // (the calling thread takes the first device's share)
        for (u32 d = 1; d < nw; d++) ctxs[(k0 + d) % n]->io[0]->wait();
#endif
    };
    auto run_all = [&](const std::function<void(uint64_t)>& fn) { run_range(0, nseg, fn); };
    {
        PlanChain pc;
        uint64_t tau = 0;                                             // stream-wide target of the next block boundary (the chain)
        int64_t shift = 0;                                            // what the boundaries in front of the wave have moved by so far (a guess for its later segments)
        for (uint64_t k0 = 0; k0 < nseg && !err; k0 += n) {           // a wave: one segment per device
            const uint64_t k1 = k0 + n < nseg ? k0 + n : nseg;
            run_range(k0, k1, scan_seg);
            if (err) break;
            for (uint64_t k = k0; k < k1; k++) {
                plan_base(pc, S[k], edge_runs(in, S[k].lo, S[k].e), cap);
                if (!S[k].ok) fail(1);
                const int64_t guess = (int64_t)S[k].phase + shift;
                S[k].t0 = (uint64_t)(guess < 0 ? guess + (int64_t)cap : guess);
            }
            if (!err && !seg_target(S[k0], tau)) fail(1);             // (the wave's first segment: its target is known)
            if (err) break;
            run_range(k0, k1, plan_seg);
            if (err) break;
            // the chain through the wave's segments: a segment planned from a wrong target (a boundary inside a run of four or more equal
            // bytes in a segment before it: ordinary text has them) is planned again, here and now - 0.1 ms, one job in some dozens
            for (uint64_t k = k0; k < k1 && !err; k++) {
                const uint64_t planned = S[k].t0;
                if (!seg_target(S[k], tau)) { fail(1); break; }
                if (S[k].t0 != planned) { g_multi_replans++; plan_seg(k); }
                if (err) break;
                shift = (int64_t)S[k].t0 - (int64_t)S[k].phase;
                tau = S[k].base + S[k].tnext;
            }
            if (err) break;
            // Every device's buffer for its segment of this wave, from the context's pool (grow-only, kept across calls: a hipMalloc per
            // segment and a hipFree - a device-wide sync - per segment in every call was what rounds 2-4 did).  Taken HERE, on the calling
            // thread and for every segment of the wave: inside the workers it depended on which of them saw another one's fallback flag
            // first, and a slot skipped in one call was a hipMalloc in the next (round 5: the test that counts them failed one run in eight).
            for (uint64_t k = k0; k < k1 && !err; k++) {
                cjs_ctx* c = ctxs[k % n];
                MSeg& g = S[k];
                g.dseg_cap = ((uint64_t)cjs_bz2_compress_bound(g.e - g.lo + W) + 3) & ~(uint64_t)3;
                if (!c->segpool) c->segpool = new std::vector<std::pair<void*, size_t>>();
                const size_t slot = (size_t)(k / n);
                if (c->segpool->size() <= slot) c->segpool->resize(slot + 1, std::pair<void*, size_t>(nullptr, 0));
                std::pair<void*, size_t>& pr = (*c->segpool)[slot];
                if (pr.second < g.dseg_cap) {
                    if (hipSetDevice(c->device) != hipSuccess) { fail(CJS_E_NOGPU); break; }
                    (void)hipFree(pr.first);
                    pr = std::pair<void*, size_t>(nullptr, 0);
                    void* pnew = nullptr;
                    g_multi_mallocs++;
                    if (hipMalloc(&pnew, g.dseg_cap + (g.dseg_cap >> 3)) != hipSuccess) { fail(CJS_E_HIP - (int)hipErrorOutOfMemory); break; }
                    pr = std::pair<void*, size_t>(pnew, g.dseg_cap + (g.dseg_cap >> 3));
                }
                g.dseg = (u8*)pr.first;
            }
            if (err) break;
            run_range(k0, k1, encode_seg);
        }
    }
    bool replicated = false;
    if (err == 1) {
        // A segment that cannot be planned on its own - a block longer than the margin, a run that fills a block or reaches beyond 4 KB at a
        // cut, a boundary inside a run that straddles a cut (round 5 also: any block boundary inside a run of four or more equal bytes - one
        // call in twenty-five at 7*10^7 bytes of text; round 6 carries those through the chain): the REPLICATED plan.  Every device takes
        // the whole input and plans it with the serial chain (cjs_bz2_plan: 0.5 ms per 10^8 bytes; the uploads run on the devices' own
        // links), then encodes its share of the blocks - what compressjs_amd/dist.py::sharded_compress does across processes.
        err.store(0);
        replicated = true;
        g_multi_fallbacks++;
        std::vector<int64_t> nbv(n, 0);
        run_range(0, n, [&](uint64_t d) {
            cjs_ctx* c = ctxs[d];
            if (hipSetDevice(c->device) != hipSuccess) { fail(CJS_E_NOGPU); return; }
            int rc = grow(&c->din, &c->din_bytes, in_len + 64);
            if (rc) { fail(rc); return; }
            hipError_t e2 = hipMemcpyAsync(c->din, in, in_len, hipMemcpyHostToDevice, c->stream);
            if (e2 != hipSuccess) { fail(CJS_E_HIP - (int)e2); return; }
            nbv[d] = cjs_bz2_plan(c, c->din, in_len, level);               // (synchronises the stream)
            if (nbv[d] < 0) { (void)hipStreamSynchronize(c->stream); fail((int)nbv[d]); }
        });
        for (u32 d = 1; d < n && !err; d++) if (nbv[d] != nbv[0]) fail(CJS_E_ARG);      // (cannot happen: the same bytes, the same plan)
        if (!err) {
            const uint64_t nb = (uint64_t)nbv[0];
            nseg = n;
            S.assign(nseg, MSeg());
            std::vector<u32> first(n), cnt(n);
            for (u32 d = 0; d < n && !err; d++) {
                cjs_ctx* c = ctxs[d];
                first[d] = (u32)(nb * d / n);
                cnt[d] = (u32)(nb * (d + 1) / n) - first[d];
                if (!cnt[d]) continue;
                MSeg& g = S[d];
                g.dseg_cap = ((uint64_t)cjs_bz2_compress_bound((uint64_t)cnt[d] * (cap + 19u)) + 3) & ~(uint64_t)3;
                if (!c->segpool) c->segpool = new std::vector<std::pair<void*, size_t>>();
                if (c->segpool->empty()) c->segpool->resize(1, std::pair<void*, size_t>(nullptr, 0));
                std::pair<void*, size_t>& pr = (*c->segpool)[0];
                if (pr.second < g.dseg_cap) {
                    if (hipSetDevice(c->device) != hipSuccess) { fail(CJS_E_NOGPU); break; }
                    (void)hipFree(pr.first);
                    pr = std::pair<void*, size_t>(nullptr, 0);
                    void* pnew = nullptr;
                    g_multi_mallocs++;
                    if (hipMalloc(&pnew, g.dseg_cap + (g.dseg_cap >> 3)) != hipSuccess) { fail(CJS_E_HIP - (int)hipErrorOutOfMemory); break; }
                    pr = std::pair<void*, size_t>(pnew, g.dseg_cap + (g.dseg_cap >> 3));
                }
                g.dseg = (u8*)pr.first;
            }
            if (!err)
                run_range(0, n, [&](uint64_t d) {
                    if (!cnt[d]) return;
                    cjs_ctx* c = ctxs[d];
                    MSeg& g = S[d];
                    u32 fold = 0, done = 0;
                    const int64_t bits = cjs_bz2_encode_blocks(c, first[d], cnt[d], g.dseg, g.dseg_cap, &fold, &done);
                    if (bits < 0) { fail((int)bits); return; }
                    g.bits = (uint64_t)bits; g.fold = fold; g.count = done;
                });
        }
    }
    // the replicated plan needs the whole input and its K0 workspace on EVERY device: when that does not fit (a HIP / out-of-memory
    // error), the call degrades to one device's segmented path, whose device memory is bounded
    if (replicated && err.load() <= CJS_E_HIP) return cjs_bz2_compress(ctxs[0], in, in_len, level, out, out_cap);
    int64_t result = 0;
    if (!err) {
        uint64_t pos = 32;
        u32 crc = 0;
        for (uint64_t k = 0; k < nseg; k++) {
            S[k].off = pos;
            pos += S[k].bits;
            crc = rotl32(crc, S[k].count) ^ S[k].fold;                // lib/Bzip2.js:917 over the segment's blocks
        }
        const uint64_t end = pos + 80, total = (end + 7) >> 3;
        if (total > out_cap) err = CJS_E_NOSPACE;
        else {
            for (uint64_t k = 0; k < nseg; k++)
                if (S[k].bits) { out[S[k].off >> 3] = 0; out[(S[k].off + S[k].bits - 1) >> 3] = 0; }
            for (uint64_t b = pos >> 3; b < total; b++) out[b] = 0;
            run_all(place_seg);
            if (!err) {
                out[0] = 'B'; out[1] = 'Z'; out[2] = 'h'; out[3] = (u8)('0' + level);       // lib/Bzip2.js:903-906
                for (uint64_t k = 0; k < nseg; k++)
                    if (S[k].bits) { out[S[k].off >> 3] |= S[k].first; out[(S[k].off + S[k].bits - 1) >> 3] |= S[k].last; }
                // end-of-stream magic + combined CRC at bit `pos` (lib/Bzip2.js:925-926), zero-padded to a byte
                const u8 tr[10] = {0x17, 0x72, 0x45, 0x38, 0x50, 0x90, (u8)(crc >> 24), (u8)(crc >> 16), (u8)(crc >> 8), (u8)crc};
                const u32 sh = (u32)(pos & 7u);
                for (u32 i = 0; i < 10; i++) {
                    out[(pos >> 3) + i] |= (u8)(tr[i] >> sh);
                    if (sh) out[(pos >> 3) + i + 1] |= (u8)(tr[i] << (8u - sh));
                }
                result = (int64_t)total;
            }
        }
    }
    if (err) return err;
    u32 blocks = 0;
    for (uint64_t k = 0; k < nseg; k++) blocks += S[k].count;
    ctxs[0]->last_blocks = blocks;
    // (cjs_last_device_ms: this path has no single device interval; it reports the wall time of the call)
    ctxs[0]->last_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return result;
}

// ---- sharded encoding (multi-GPU): plan once, then encode a block range bit-aligned at 0 ----------
// cjs_bz2_plan runs the K0 pre-pass over the whole input (kept in the context) and returns the
// number of bzip2 blocks.  cjs_bz2_encode_blocks encodes blocks [first, first+count) into d_seg
// WITHOUT stream header/trailer, first block starting at bit 0; returns the number of bits and the
// partial combined-CRC fold  P = XOR_i rotl^(count-1-i)(crc_i)  so that the caller can chain
// S' = rotl^count(S) ^ P across shards (lib/Bzip2.js:917 is linear over GF(2)).
extern "C" int64_t cjs_bz2_plan(cjs_ctx* c, const void* d_in, uint64_t in_len, int level) {
    if (!c || (!d_in && in_len)) return CJS_E_ARG;
    if (level < 1 || level > 9) return CJS_E_LEVEL;
    hipError_t e;
#define TRYR(x) if ((e = (x)) != hipSuccess) return CJS_E_HIP - (int)e
    TRYR(hipSetDevice(c->device));
    const u32 cap = (u32)level * 100000u - 19u;
    c->plan_level = 0;                               // no valid plan while this one is being made
    c->plan_blocks = 0;
    c->scan_level = 0;                               // c->plan is re-carved: an earlier cjs_bz2_plan_scan's tables are gone
    int rc = grow(&c->planws, &c->planws_bytes, k0_bytes(in_len, cap));
    if (rc) return rc;
    k0_carve(c->plan, (const u8*)d_in, in_len, cap, c->planws);
    rc = k0_prepass(c->plan, cap, c->stream);
    if (rc) return rc;
    u32 nblocks = 0;
    TRYR(hipMemcpyAsync(&nblocks, c->plan.nBlocks, 4, hipMemcpyDeviceToHost, c->stream));
    TRYR(hipStreamSynchronize(c->stream));
    c->plan_blocks = nblocks;
    c->plan_level = level;
    return (int64_t)nblocks;
#undef TRYR
}

// (bench.py) phases of the last cjs_bwtc_compress: out[0] = ms of the first K10 launch, [1] = ms until every triple was on the host,
// [2] = ms the range coder was busy, [3] = ms of the whole call, [4] = encodeFreq calls (model symbols + escapes)
extern "C" int cjs_bwtc_last_times(cjs_ctx* c, float* out5) {
    if (!c || !out5) return CJS_E_ARG;
    for (int i = 0; i < 5; i++) out5[i] = c->bwtc_times[i];
    return CJS_OK;
}

// ---- parallel plan of a slice (multi-GPU): see k0_rle1.hip "the blocks of a SLICE of a longer stream" and compressjs_amd/dist.py ----
// cjs_bz2_plan_scan: K0's tile scans over d_in (the rank's slice followed by the margin it holds of what comes after); returns
// the input's own RLE1 cost total.  cjs_bz2_plan_cost: that cost prefix at byte `pos`.  cjs_bz2_plan_phase: the blocks that
// START in [0, own_len), given that boundaries lie where the prefix reaches phase + m * cap; returns their number (they are
// then blocks 0 .. n-1 for cjs_bz2_encode_blocks), or CJS_E_SPEC when the slice cannot be planned on its own (a boundary
// inside a long run, a block longer than the margin): the caller falls back to the chained / replicated plan.
extern "C" int64_t cjs_bz2_plan_scan(cjs_ctx* c, const void* d_in, uint64_t in_len, int level) {
    if (!c || (!d_in && in_len)) return CJS_E_ARG;
    if (level < 1 || level > 9) return CJS_E_LEVEL;
    hipError_t e;
#define TRYR(x) if ((e = (x)) != hipSuccess) return CJS_E_HIP - (int)e
    TRYR(hipSetDevice(c->device));
    const u32 cap = (u32)level * 100000u - 19u;
    c->plan_level = 0;
    c->plan_blocks = 0;
    c->scan_level = 0;
    int rc = grow(&c->planws, &c->planws_bytes, k0_bytes(in_len, cap));
    if (rc) return rc;
    k0_carve(c->plan, (const u8*)d_in, in_len, cap, c->planws);
    uint64_t total = 0;
    if (in_len) {
        rc = k0_scans(c->plan, c->stream);
        if (rc) return rc;
        TRYR(hipMemcpyAsync(&total, c->plan.tileC + c->plan.ntiles, 8, hipMemcpyDeviceToHost, c->stream));
        TRYR(hipStreamSynchronize(c->stream));
    }
    c->scan_level = level;
    c->scan_total = total;
    return (int64_t)total;
#undef TRYR
}

extern "C" int64_t cjs_bz2_plan_cost(cjs_ctx* c, uint64_t pos) {
    if (!c || !c->scan_level) return CJS_E_ARG;
    if (pos >= c->plan.in_len) return (int64_t)c->scan_total;
    if (pos == 0) return 0;
    hipError_t e;
#define TRYR(x) if ((e = (x)) != hipSuccess) return CJS_E_HIP - (int)e
    TRYR(hipSetDevice(c->device));
    const int rc = k0_eval(c->plan, pos, c->stream);
    if (rc) return rc;
    uint64_t v = 0;
    TRYR(hipMemcpyAsync(&v, c->plan.specC, 8, hipMemcpyDeviceToHost, c->stream));
    TRYR(hipStreamSynchronize(c->stream));
    return (int64_t)v;
#undef TRYR
}

// Round 6: the slice's plan as a link of a CHAIN.  t0 = the value this input's own cost prefix reaches at the slice's first block boundary
// (the previous slice's *t_next after a change of origin; phase = (-G(lo)) mod cap when nothing in front of the slice moved a boundary).
// A block boundary inside a run of four or more equal bytes no longer refuses: the slice goes on serially from there (k0_phase_chain)
// and *t_next - the target of the first boundary at or beyond own_len - carries the shift to the slices behind it.  CJS_E_SPEC is left
// for what a slice cannot do on its own: a block longer than the margin, a run that fills a block.
extern "C" int64_t cjs_bz2_plan_chain(cjs_ctx* c, uint64_t own_len, uint64_t t0, int last, uint64_t* t_next) {
    if (!c || !c->scan_level || own_len > c->plan.in_len) return CJS_E_ARG;
    hipError_t e;
#define TRYR(x) if ((e = (x)) != hipSuccess) return CJS_E_HIP - (int)e
    TRYR(hipSetDevice(c->device));
    const u32 cap = (u32)c->scan_level * 100000u - 19u;
    c->plan_level = 0;
    c->plan_blocks = 0;
    if (t_next) *t_next = t0;
    if (c->plan.in_len == 0) { c->plan_level = c->scan_level; return 0; }
    const int rc = k0_phase_plan(c->plan, cap, t0, own_len, last ? 1u : 0u, c->scan_total, c->stream);
    if (rc) return rc;
    // (pinned: a pageable read-back costs tens of microseconds more; pin[0] is free here - no sub-batch is running on this context)
    TRYR(hipMemcpyAsync(c->pin[0], c->plan.nBlocks, 16, hipMemcpyDeviceToHost, c->stream));
    TRYR(hipStreamSynchronize(c->stream));
    const u32 nblocks = c->pin[0][0];
    uint64_t tn = 0;
    memcpy(&tn, c->pin[0] + 2, 8);
    if (nblocks == K0_PHASE_FAIL) return CJS_E_SPEC;
    if (t_next) *t_next = tn;
    c->plan_blocks = nblocks;
    c->plan_level = c->scan_level;
    return (int64_t)nblocks;
#undef TRYR
}

// The round-3 entry: boundaries where the prefix reaches phase + m * cap and NOTHING carried - a slice whose plan moves the
// boundaries behind it (a boundary inside a long run) is refused, as before, unless the stream ends in it.
extern "C" int64_t cjs_bz2_plan_phase(cjs_ctx* c, uint64_t own_len, uint64_t phase, int last) {
    if (!c || !c->scan_level) return CJS_E_ARG;
    const u32 cap = (u32)c->scan_level * 100000u - 19u;
    if (phase >= cap) return CJS_E_ARG;
    uint64_t tn = 0;
    const int64_t nb = cjs_bz2_plan_chain(c, own_len, phase, last, &tn);
    if (nb < 0) return nb;
    if (!last && tn >= phase && (tn - phase) % cap != 0) { c->plan_level = 0; c->plan_blocks = 0; return CJS_E_SPEC; }
    return nb;
}

// first input byte (relative to the planned input) of block k of the current plan; k == number of blocks: the input length
extern "C" int64_t cjs_bz2_plan_block_start(cjs_ctx* c, uint32_t k) {
    if (!c || !c->plan_level || k > c->plan_blocks) return CJS_E_ARG;
    if (k == c->plan_blocks) return (int64_t)c->plan.in_len;
    if (hipSetDevice(c->device) != hipSuccess) return CJS_E_NOGPU;
    uint64_t v = 0;
    const hipError_t e = hipMemcpy(&v, c->plan.blkStart + k, 8, hipMemcpyDeviceToHost);
    return e == hipSuccess ? (int64_t)v : CJS_E_HIP - (int)e;
}

extern "C" int64_t cjs_bz2_encode_blocks(cjs_ctx* c, uint32_t first, uint32_t count, void* d_seg,
                                         uint64_t seg_cap, uint32_t* crc_fold, uint32_t* n_done) {
    if (!c || !d_seg || !c->plan_level || ((uintptr_t)d_seg & 3) || seg_cap < 64) return CJS_E_ARG;
    if (first > c->plan_blocks) return CJS_E_ARG;
    if (count > c->plan_blocks - first) count = c->plan_blocks - first;   // no uint32 wrap for count = 0xFFFFFFFF ("all remaining")
    hipError_t e;
    int rc;
#define TRYR(x) if ((e = (x)) != hipSuccess) return CJS_E_HIP - (int)e
    TRYR(hipSetDevice(c->device));
    const u32 cap = (u32)c->plan_level * 100000u - 19u;
    hipStream_t st = c->stream;
    Pipe P0;
    memset(&P0, 0, sizeof P0);
    P0.ss = c->d_ss;
    P0.out = (u32*)d_seg;
    P0.outCapBytes = seg_cap & ~(uint64_t)3;
    TRYR(hipEventRecord(c->ev0, st));
    rc = k5_stream_begin(P0, -1, st);                // level < 0: no "BZh" header, cursor at bit 0
    if (rc) return rc;
    TRYR(hipEventRecord(c->evReady, st));
    rc = issue_blocks(c, c->plan, cap, first, count, d_seg, seg_cap);
    if (rc) return rc;
    TRYR(hipEventRecord(c->ev1, st));
    StreamState hs;
    TRYR(hipMemcpyAsync(&hs, c->d_ss, sizeof hs, hipMemcpyDeviceToHost, st));
    TRYR(hipStreamSynchronize(st));
    TRYR(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    c->last_blocks = count;
    if (hs.overflow) return CJS_E_NOSPACE;
    if (crc_fold) *crc_fold = hs.crc;                // k5 folded from 0: exactly P
    if (n_done) *n_done = count;
    return (int64_t)hs.bits;
#undef TRYR
}

// ---------------------------------------------------------------------------------------------
// BWTC.compressFile(input, null, level), level 6..9 (lib/BWTC.js:12-139): BWT + MTF/RLE2 on the
// GPU per 100000*level-byte block, adaptive range coder on the host (serial by construction).
// Levels 1-5 use DefSumModel (lib/BWTC.js:107), levels 6-9 FenwickModel; both coders run on the host.
// ---------------------------------------------------------------------------------------------
extern "C" int64_t cjs_bwtc_compress_bound(uint64_t in_len) { return (int64_t)bwtc_bound(in_len); }

extern "C" int64_t cjs_bwtc_compress(cjs_ctx* c, const uint8_t* in, uint64_t in_len, int level, uint8_t* out,
                                     uint64_t out_cap, int64_t declared_size) {
    if (!c || (!in && in_len) || !out) return CJS_E_ARG;
    if (level < 1 || level > 9) level = 9;                         // lib/BWTC.js:16-19: bad props -> 9
    hipError_t e;
    int rc;
#define TRYR(x) if ((e = (x)) != hipSuccess) { if (coder) (void)bwtc_end(coder); return CJS_E_HIP - (int)e; }
    bwtc_coder* coder = nullptr;
    TRYR(hipSetDevice(c->device));
    const u32 bs = (u32)level * 100000u;
    rc = grow(&c->din, &c->din_bytes, in_len + 64);
    if (rc) return rc;
    hipStream_t st = c->stream;
    if (in_len) TRYR(hipMemcpyAsync(c->din, in, in_len, hipMemcpyHostToDevice, st));
    const u64 nblocks = (in_len + bs - 1) / bs;
    BatchGeom g = make_geom(c->sub_blocks, bs);
    coder = bwtc_begin(out, out_cap, declared_size, level);
    // levels 6..9: the adaptive FenwickModel of every block runs on the GPU too (K10, one wave per block: a serial
    // recurrence, ~170 ms per launch whatever the number of blocks), the host keeps the range coder.
    // CJS_BWTC_GPU_MODEL=0: model on the host as in round 1 (A/B runs).
    const bool gpu_model = []() { const char* ev = getenv("CJS_BWTC_GPU_MODEL"); return !ev || atoi(ev) != 0; }();          // (read per call)
    // rows of K10's triples: 2 x stride per block in the round lists of K1 (free in linear mode); CJS_K10_CAP (tests) shrinks them
    const u32 k10_ostride = 2u * make_geom(c->sub_blocks, (u32)level * 100000u).stride;
    const u32 k10_cap = []() -> u32 { const char* ev = getenv("CJS_K10_CAP"); return ev ? (u32)strtoul(ev, nullptr, 10) : 0xFFFFFFFFu; }() < k10_ostride
                            ? (u32)strtoul(getenv("CJS_K10_CAP"), nullptr, 10) : k10_ostride;
    const bool tri = gpu_model && level >= 6;
    // Sub-batches are processed in GROUPS of one per stream: their GPU stages are issued back to back on different
    // streams (the K10 launches of a group overlap), then everything the coder needs is copied to host vectors and
    // handed to the coder thread, which works through group g while the GPU runs group g + 1.
    typedef BwtcBlockJob BlockJob;
    typedef BwtcGroupJob GroupJob;
    for (int i = 0; i < 2; i++) { if (!c->bwtc_jobs[i]) c->bwtc_jobs[i] = new GroupJob(); c->bwtc_jobs[i]->busy = false; }
    std::mutex mu;
    std::condition_variable cv;
    std::vector<GroupJob*> queue;
    bool done_issuing = false;
    const bool btrace = getenv("CJS_BWTC_TRACE") != nullptr;
    const auto tb0 = std::chrono::steady_clock::now();
    auto msnow = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(); };
    double coder_busy = 0, coder_first = 0, ncalls_total = 0;
    std::thread coder_thread([&]() {
        for (;;) {
            GroupJob* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return done_issuing || !queue.empty(); });
                if (queue.empty()) return;
                job = queue.front();
                queue.erase(queue.begin());
            }
            const double tj0 = msnow();
            if (coder_first == 0) coder_first = tj0;
            for (size_t bi = 0; bi < job->blocks.size(); bi++) {
                {   // the copies of a group land stream by stream: start on what is there
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&]() { return bi < job->nready || done_issuing; });
                    if (bi >= job->nready) return;                     // (error path: the issuer gave up)
                }
                const BlockJob& bj = job->blocks[bi];
                if (tri && !bj.host) bwtc_block_triples(coder, bj.len, bj.pidx, bj.used, job->a.data() + bj.off, job->t.data() + bj.off, bj.ntri);
                else bwtc_block(coder, bj.len, bj.pidx, bj.used, job->sym.data() + (tri ? bj.soff : bj.off), bj.nsym);
            }
            coder_busy += msnow() - tj0;
            { std::lock_guard<std::mutex> lk(mu); job->busy = false; cv.notify_all(); }
        }
    });
    auto stop_coder = [&]() {
        { std::lock_guard<std::mutex> lk(mu); done_issuing = true; cv.notify_all(); }
        if (coder_thread.joinable()) coder_thread.join();
    };
#undef TRYR
#define TRYR(x) if ((e = (x)) != hipSuccess) { stop_coder(); (void)bwtc_end(coder); return CJS_E_HIP - (int)e; }
    const u32 ns = c->nstreams;
    std::vector<u32> nl((size_t)c->sub_blocks * ns), hpos((size_t)c->sub_blocks * ns), hpidx((size_t)c->sub_blocks * ns),
        hused((size_t)c->sub_blocks * ns * 8), hntri((size_t)c->sub_blocks * ns);
    TRYR(hipStreamSynchronize(st));                                // the input is resident
    TRYR(hipEventRecord(c->ev0, st));
    float gpu_ms = 0.f;
    try {
    for (u64 first = 0; first < nblocks; first += (u64)c->sub_blocks * ns) {
        Pipe Ps[CJS_NSTREAMS];
        u32 nbs[CJS_NSTREAMS] = {0, 0, 0, 0};
        for (u32 si = 0; si < ns; si++) {
            const u64 f = first + (u64)si * c->sub_blocks;
            if (f >= nblocks) break;
            const u32 nb = (u32)(nblocks - f < c->sub_blocks ? nblocks - f : c->sub_blocks);
            nbs[si] = nb;
            hipStream_t ss = c->sub[si];
            Pipe& P = Ps[si];
            pipe_carve(P, g, c->ws[si]);
            P.g.nb = nb;
                    P.k1.linear = 1;
            u32* nls = nl.data() + (size_t)si * c->sub_blocks;
            u32 max_n = 0;
            for (u32 b = 0; b < nb; b++) {
                const u64 off = (f + b) * bs;
                nls[b] = (u32)(in_len - off < bs ? in_len - off : bs);
                if (nls[b] > max_n) max_n = nls[b];
            }
            // T_ext rows: block bytes followed by zeros (linear mode pads with the smallest symbol)
            TRYR(hipMemsetAsync(P.T, 0, (size_t)nb * g.tstride, ss));
            const u32 full = (nls[nb - 1] == bs) ? nb : nb - 1;
            if (full) TRYR(hipMemcpy2DAsync(P.T, g.tstride, (const u8*)c->din + f * bs, bs, bs, full, hipMemcpyDeviceToDevice, ss));
            if (full < nb) TRYR(hipMemcpyAsync(P.T + (size_t)full * g.tstride, (const u8*)c->din + (f + full) * bs, nls[nb - 1], hipMemcpyDeviceToDevice, ss));
            TRYR(hipMemcpyAsync(P.nlen, nls, nb * 4, hipMemcpyHostToDevice, ss));
            rc = k1_run(P.k1, P.g, max_n, ss);
            if (!rc) rc = k2_run(P, max_n, ss);
            if (!rc && tri) {
                if (si == 0 && first == 0) { if (!c->evK10[0]) { TRYR(hipEventCreate(&c->evK10[0])); TRYR(hipEventCreate(&c->evK10[1])); } TRYR(hipEventRecord(c->evK10[0], ss)); }
                rc = k10_model_run(P, (u32*)P.k1.rlist[0], (u32*)P.k1.rlist[1], P.ngroups, k10_ostride, k10_cap, ss);
                if (si == 0 && first == 0) TRYR(hipEventRecord(c->evK10[1], ss));
            }
            if (rc) { stop_coder(); (void)bwtc_end(coder); return rc; }
        }
        // the per-block results, only now: a device-to-host copy into pageable memory blocks the HOST until the stream has
        // drained, i.e. for the 132 ms of that stream's K10 - issued inside the loop above it kept the next stream's K1 / K2 /
        // K10 from even being launched (kernel trace of round 2: the two K10 launches ran back to back, 264 ms)
        for (u32 si = 0; si < ns && nbs[si]; si++) {
            hipStream_t ss = c->sub[si];
            Pipe& P = Ps[si];
            const u32 nb = nbs[si];
            const size_t o = (size_t)si * c->sub_blocks;
            if (tri) TRYR(hipMemcpyAsync(hntri.data() + o, P.ngroups, nb * 4, hipMemcpyDeviceToHost, ss));
            TRYR(hipMemcpyAsync(hpos.data() + o, P.pos, nb * 4, hipMemcpyDeviceToHost, ss));
            TRYR(hipMemcpyAsync(hpidx.data() + o, P.pidx, nb * 4, hipMemcpyDeviceToHost, ss));
            TRYR(hipMemcpyAsync(hused.data() + o * 8, P.used, (size_t)nb * 32, hipMemcpyDeviceToHost, ss));
        }
        GroupJob* job = nullptr;
        {   // a buffer the coder thread is done with
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return !c->bwtc_jobs[0]->busy || !c->bwtc_jobs[1]->busy; });
            job = !c->bwtc_jobs[0]->busy ? c->bwtc_jobs[0] : c->bwtc_jobs[1];
            job->busy = true;
        }
        if (btrace) fprintf(stderr, "[bwtc] group issued at %.1f ms\n", msnow());
        job->blocks.clear();
        size_t total = 0, stotal = 0;
        for (u32 si = 0; si < ns && nbs[si]; si++) {
            TRYR(hipStreamSynchronize(c->sub[si]));
            const size_t o = (size_t)si * c->sub_blocks;
            for (u32 b = 0; b < nbs[si]; b++) {
                BlockJob bj;
                bj.len = nl[o + b]; bj.pidx = hpidx[o + b];
                bj.nsym = hpos[o + b] - 1;                         // K2 appends bzip2's EOB; BWTC has none
                bj.ntri = tri ? hntri[o + b] : 0;
                bj.host = tri && bj.ntri > k10_cap;                 // K10_OVERFLOW (or anything beyond the row: never copied)
                bj.soff = 0;
                if (bj.host) { bj.ntri = 0; bj.soff = stotal; stotal += (size_t)bj.nsym + 1; }
                memcpy(bj.used, hused.data() + (o + b) * 8, 32);
                bj.off = total;
                total += (tri ? bj.ntri : bj.nsym) + 1;
                ncalls_total += tri ? bj.ntri : bj.nsym;
                job->blocks.push_back(bj);
            }
        }
        if (btrace) fprintf(stderr, "[bwtc] K1 + K2 + K10 of the group done at %.1f ms\n", msnow());
        if (tri) {
            rc = job->a.reserve(total);
            if (!rc) rc = job->t.reserve(total);
            if (rc) { stop_coder(); (void)bwtc_end(coder); return rc; }
            if (job->sym.size() < stotal) job->sym.resize(stotal);
        }
        else if (job->sym.size() < total) job->sym.resize(total);
        size_t k = 0;
        for (u32 si = 0; si < ns && nbs[si]; si++)
            for (u32 b = 0; b < nbs[si]; b++, k++) {
                const BlockJob& bj = job->blocks[k];
                Pipe& P = Ps[si];
                if (tri && bj.host) {
                    if (bj.nsym) TRYR(hipMemcpyAsync(job->sym.data() + bj.soff, P.A + (size_t)b * g.stride, (size_t)bj.nsym * 2, hipMemcpyDeviceToHost, c->sub[si]));
                } else if (tri && bj.ntri) {
                    TRYR(hipMemcpyAsync(job->a.data() + bj.off, (u32*)P.k1.rlist[0] + (size_t)b * k10_ostride, (size_t)bj.ntri * 4, hipMemcpyDeviceToHost, c->sub[si]));
                    TRYR(hipMemcpyAsync(job->t.data() + bj.off, (u32*)P.k1.rlist[1] + (size_t)b * k10_ostride, (size_t)bj.ntri * 4, hipMemcpyDeviceToHost, c->sub[si]));
                } else if (!tri && bj.nsym) {
                    TRYR(hipMemcpyAsync(job->sym.data() + bj.off, P.A + (size_t)b * g.stride, (size_t)bj.nsym * 2, hipMemcpyDeviceToHost, c->sub[si]));
                }
            }
        if (btrace) fprintf(stderr, "[bwtc] copies issued at %.1f ms\n", msnow());
        { std::lock_guard<std::mutex> lk(mu); job->nready = 0; queue.push_back(job); cv.notify_all(); }
        for (u32 si = 0; si < ns && nbs[si]; si++) {
            TRYR(hipStreamSynchronize(c->sub[si]));
            { std::lock_guard<std::mutex> lk(mu); job->nready += nbs[si]; cv.notify_all(); }
        }
    }
    } catch (const std::exception&) {                              // e.g. std::bad_alloc while sizing a job: the coder thread must be joined
        stop_coder();
        (void)bwtc_end(coder);
        return CJS_E_NOSPACE;
    }
    const double t_issued = msnow();
    stop_coder();
    {
        float k10ms = 0.f;
        if (tri && nblocks && c->evK10[0]) (void)hipEventElapsedTime(&k10ms, c->evK10[0], c->evK10[1]);
        c->bwtc_times[0] = k10ms; c->bwtc_times[1] = (float)t_issued; c->bwtc_times[2] = (float)coder_busy; c->bwtc_times[3] = (float)msnow();
        c->bwtc_times[4] = (float)ncalls_total;
    }
    if (btrace) fprintf(stderr, "[bwtc] %llu blocks: GPU stages + copies issued and done at %.1f ms, coder started at %.1f ms, busy %.1f ms, all done at %.1f ms\n",
                        (unsigned long long)nblocks, t_issued, coder_first, coder_busy, msnow());
    TRYR(hipEventRecord(c->ev1, st));
    TRYR(hipStreamSynchronize(st));
    TRYR(hipEventElapsedTime(&gpu_ms, c->ev0, c->ev1));
    c->last_ms = gpu_ms;
    c->last_blocks = (u32)nblocks;
    {
        bwtc_coder* cc = coder;
        coder = nullptr;
        return bwtc_end(cc);
    }
#undef TRYR
}

// ---------------------------------------------------------------------------------------------
// Workload generator of BASELINE.json configs[3] / SURVEY.md 8(c) on the device: bytes [first, first + n) of LCG(N, seed),
//   s <- s * 1664525 + 1013904223 (mod 2^32), byte = 32 + ((s >>> 16) mod 95),
// so that every rank of a multi-GPU run fills its own slice in HBM instead of receiving it from the host.  Jump-ahead:
// s_j = A^j s_0 + C (A^(j-1) + ... + 1); each thread squares its way to the state of its first byte (32 steps), then
// walks 16 bytes.  Same bytes as compressjs_amd.synth.lcg_ascii (tests).  Not a compression entry point.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lcg_fill(u8* out, u64 n, u32 seed, u64 first) {
    const u64 t = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 i0 = t * 16u;
    if (i0 >= n) return;
    // state after j = first + i0 steps: affine map x -> a x + c composed j times by binary powering
    u64 j = first + i0;
    u32 ra = 1u, rc = 0u;                    // result map (identity)
    u32 ba = 1664525u, bc = 1013904223u;     // current power of the base map
    while (j) {
        if (j & 1u) { rc = ba * rc + bc; ra = ba * ra; }
        bc = ba * bc + bc; ba = ba * ba;
        j >>= 1;
    }
    u32 st = ra * seed + rc;
    const u64 m = n - i0 < 16u ? n - i0 : 16u;
    for (u64 k = 0; k < m; k++) {
        st = st * 1664525u + 1013904223u;
        out[i0 + k] = (u8)(32u + ((st >> 16) % 95u));
    }
}

extern "C" int32_t cjs_lcg_ascii_device(cjs_ctx* c, uint8_t* d_out, uint64_t n, uint32_t seed, uint64_t first) {
    if (!c || (!d_out && n)) return CJS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return CJS_E_NOGPU;
    if (n) hipLaunchKernelGGL(k_lcg_fill, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, c->stream, d_out, n, seed, first);
    const hipError_t e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? CJS_OK : CJS_E_HIP - (int)e;
}

extern "C" float cjs_last_device_ms(const cjs_ctx* c) { return c ? c->last_ms : 0.f; }
extern "C" uint32_t cjs_last_block_count(const cjs_ctx* c) { return c ? c->last_blocks : 0; }
extern "C" void* cjs_stream(const cjs_ctx* c) { return c ? (void*)c->stream : nullptr; }

// Per-kernel timing of K1's main kernels with HIP events on the library's own streams, for bench.py's roofline leg
// (K1P_* classes of k1_bwt.h: 0 k1f_bsort, 1 k1r_round, 2 k1d_build, 3 k1d_round, 4 k1d_med, 5 k1d_large, 6 k1d_update,
// 7 k1f_task).  cjs_profile_enable(1) clears the records; the classes can be read in any order while they stand.
extern "C" int32_t cjs_profile_enable(cjs_ctx* c, int on) {
    if (!c) return CJS_E_ARG;
    return k1_prof_enable(c->prof, on);
}
extern "C" int32_t cjs_profile_read_class(cjs_ctx* c, uint32_t cls, float* total_ms, uint32_t* launches, uint64_t* elements) {
    if (!c || cls >= K1_PROF_CLASSES) return CJS_E_ARG;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return CJS_E_HIP - (int)e;
    u64 el = 0;
    const int rc = k1_prof_read(c->prof, cls, total_ms, launches, &el);
    if (rc) return rc;
    if (cls == K1P_DROUND || cls == K1P_DUPDATE) {
        // list entries the doubling rounds walked: the counters of the LAST sub-batch of stream 0 (what a one-stream pass over
        // equal batches leaves), times the k1_dbl_run calls recorded
        const BatchGeom g = c->prof_g;
        if (!g.nb) { if (elements) *elements = 0; return CJS_OK; }
        Pipe P;
        pipe_carve(P, g, c->ws[0]);
        std::vector<u32> cn((size_t)(K1D_MAXR + 2) * P.k1.rstride);
        e = hipMemcpy(cn.data(), P.k1.dcnt, cn.size() * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return CJS_E_HIP - (int)e;
        u64 tot = 0;
        for (u32 v : cn) tot += v;
        el = tot * c->prof.dbl_runs;
    }
    if (elements) *elements = el;
    return CJS_OK;
}
extern "C" int32_t cjs_profile_read(cjs_ctx* c, float* total_ms, uint32_t* launches, uint64_t* elements) {
    return cjs_profile_read_class(c, K1P_BSORT, total_ms, launches, elements);
}

// ---------------------------------------------------------------------------------------------
// decoder: Bzip2.decompressFile / decompressBlock / table (lib/Bzip2.js:454-548)
// ---------------------------------------------------------------------------------------------
static u32 dec_slots(const cjs_ctx* c) { return c->batch_blocks < 16 ? 16u : c->batch_blocks; }

static int64_t dec_deliver(cjs_ctx* c, int64_t n, uint8_t* out, uint64_t out_cap, bool out_dev) {
    if (n < 0) return n;
    if ((uint64_t)n > out_cap) return CJS_E_NOSPACE;             // result stays fetchable (cjs_bz2_fetch)
    if (n == 0) return 0;
    if (!out) return CJS_E_ARG;
    u64 sz = 0;
    const u8* d = dec_output(c->dec, &sz);
    hipError_t e = hipMemcpyAsync(out, d, (size_t)n, out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? n : (int64_t)(CJS_E_HIP - (int)e);
}

static int64_t dec_timed(cjs_ctx* c, const uint8_t* in, uint64_t in_len, bool in_dev, int multistream, bool check_crc) {
    if (!c || (!in && in_len)) return CJS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return CJS_E_NOGPU;
    (void)hipEventRecord(c->ev0, c->stream);
    const int64_t n = dec_stream(&c->dec, dec_slots(c), c->stream, in, in_len, in_dev, multistream, check_crc);
    (void)hipEventRecord(c->ev1, c->stream);
    (void)hipStreamSynchronize(c->stream);
    c->dec_ms = 0.f;
    (void)hipEventElapsedTime(&c->dec_ms, c->ev0, c->ev1);
    return n;
}

extern "C" int64_t cjs_bz2_decompress(cjs_ctx* c, const uint8_t* in, uint64_t in_len, uint8_t* out, uint64_t out_cap,
                                      int multistream) {
    return dec_deliver(c, dec_timed(c, in, in_len, false, multistream, true), out, out_cap, false);
}
extern "C" int64_t cjs_bz2_decompress_device(cjs_ctx* c, const uint8_t* d_in, uint64_t in_len, uint8_t* d_out,
                                             uint64_t out_cap, int multistream) {
    return dec_deliver(c, dec_timed(c, d_in, in_len, true, multistream, true), d_out, out_cap, true);
}
extern "C" int64_t cjs_bz2_decompress_block(cjs_ctx* c, const uint8_t* in, uint64_t in_len, uint64_t bitpos,
                                            uint8_t* out, uint64_t out_cap) {
    if (!c || (!in && in_len)) return CJS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return CJS_E_NOGPU;
    return dec_deliver(c, dec_block(&c->dec, dec_slots(c), c->stream, in, in_len, bitpos), out, out_cap, false);
}
// Bzip2.table: callback(position in bits, decoded bytes) per block -> two arrays; returns the block count
extern "C" int64_t cjs_bz2_table(cjs_ctx* c, const uint8_t* in, uint64_t in_len, int multistream, uint64_t* positions,
                                 uint64_t* sizes, uint32_t cap) {
    const int64_t n = dec_timed(c, in, in_len, false, multistream, false);     // the stream CRC is ignored (:536)
    if (n < 0) return n;
    const u64 *p = nullptr, *z = nullptr;
    const u32 nb = dec_table(c->dec, &p, &z);
    for (u32 i = 0; i < nb && i < cap; i++) { if (positions) positions[i] = p[i]; if (sizes) sizes[i] = z[i]; }
    return (int64_t)nb;
}
// size / bytes of the last successful decode (for callers that learn the size from the first call)
extern "C" int64_t cjs_bz2_last_size(cjs_ctx* c) {
    u64 sz = 0;
    if (!c || !c->dec) return 0;
    (void)dec_output(c->dec, &sz);
    return (int64_t)sz;
}
extern "C" int64_t cjs_bz2_fetch(cjs_ctx* c, uint8_t* out, uint64_t out_cap) {
    if (!c || !c->dec) return CJS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return CJS_E_NOGPU;
    return dec_deliver(c, cjs_bz2_last_size(c), out, out_cap, false);
}
// detail of the last decoder error: 1 'bad magic', 2 'level out of range', 3 'initial position out of
// bounds', 4 'Bad block CRC (got .. expected ..)', 5 'Bad stream CRC (got .. expected ..)'
extern "C" int32_t cjs_bz2_last_detail(cjs_ctx* c, uint32_t* crc_got, uint32_t* crc_expected) {
    int d = 0; u32 g = 0, w = 0;
    if (c && c->dec) dec_error_info(c->dec, &d, &g, &w);
    if (crc_got) *crc_got = g;
    if (crc_expected) *crc_expected = w;
    return d;
}
extern "C" float cjs_bz2_last_decode_ms(cjs_ctx* c) { return c ? c->dec_ms : 0.f; }

// ---------------------------------------------------------------------------------------------
// BWTC.decompressFile (lib/BWTC.js:141-233): host range decoder (serial), inverse BWT of every block on
// the GPU (K6).  The decoded bytes are retained for cjs_bwtc_fetch when `out_cap` is too small.
// ---------------------------------------------------------------------------------------------
struct BwtcSink {
    cjs_ctx* c; u8 *dT, *dU; void* ws; std::vector<u8>* out; int err;
};
static int bwtc_on_block(void* user, const uint8_t* T, uint32_t length, uint32_t pidx) {
    BwtcSink* k = (BwtcSink*)user;
    if (length == 0) return 0;
    hipStream_t st = k->c->stream;
    hipError_t e = hipMemcpyAsync(k->dT, T, length, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return CJS_E_HIP - (int)e;
    const int rc = k6_unbwt_linear(k->dT, k->dU, length, pidx, k->ws, st);     // BWT.unbwtransform :224
    if (rc) return rc;
    const size_t at = k->out->size();
    k->out->resize(at + length);
    e = hipMemcpyAsync(k->out->data() + at, k->dU, length, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e == hipSuccess ? 0 : CJS_E_HIP - (int)e;
}
extern "C" int64_t cjs_bwtc_decompress(cjs_ctx* c, const uint8_t* in, uint64_t in_len, uint8_t* out, uint64_t out_cap,
                                       int64_t* declared_size) {
    if (!c || (!in && in_len)) return CJS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return CJS_E_NOGPU;
    if (!c->bwtc_out) c->bwtc_out = new std::vector<u8>();
    c->bwtc_out->clear();
    BwtcSink k = {c, nullptr, nullptr, nullptr, c->bwtc_out, 0};
    const u32 bs = 900000u;
    const size_t wsb = (size_t)bs * 20 + ((size_t)(bs + 4095) / 4096) * 1024 + 256;
    int rc = CJS_OK;
    hipError_t e;
    if ((e = hipMalloc((void**)&k.dT, bs)) != hipSuccess || (e = hipMalloc((void**)&k.dU, bs)) != hipSuccess ||
        (e = hipMalloc(&k.ws, wsb)) != hipSuccess) rc = CJS_E_HIP - (int)e;
    if (!rc) rc = bwtc_decode(in, in_len, declared_size, &k, bwtc_on_block);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(k.dT); (void)hipFree(k.dU); (void)hipFree(k.ws);
    if (rc) { c->bwtc_out->clear(); return rc; }
    const u64 n = c->bwtc_out->size();
    if (n > out_cap) return CJS_E_NOSPACE;
    if (n) memcpy(out, c->bwtc_out->data(), n);
    return (int64_t)n;
}
extern "C" int64_t cjs_bwtc_last_size(cjs_ctx* c) { return c && c->bwtc_out ? (int64_t)c->bwtc_out->size() : 0; }
extern "C" int64_t cjs_bwtc_fetch(cjs_ctx* c, uint8_t* out, uint64_t out_cap) {
    if (!c || !c->bwtc_out) return CJS_E_ARG;
    const u64 n = c->bwtc_out->size();
    if (n > out_cap) return CJS_E_NOSPACE;
    if (n) memcpy(out, c->bwtc_out->data(), n);
    return (int64_t)n;
}

// Seam helper of the multi-GPU assembly (compressjs_amd/dist.py): d_out[0 .. nbytes] = the nbytes of
// d_in shifted right by `s` (0..7) bits.  Device pointers; d_out needs nbytes + 1 bytes.
extern "C" int32_t cjs_shift_bits(cjs_ctx* c, const uint8_t* d_in, uint64_t nbytes, uint32_t s, uint8_t* d_out) {
    if (!c || !d_in || !d_out || s > 7) return CJS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return CJS_E_NOGPU;
    const int rc = k5_shift_bits_run(d_in, nbytes, s, d_out, c->stream);
    if (rc) return rc;
    const hipError_t e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? CJS_OK : CJS_E_HIP - (int)e;
}
