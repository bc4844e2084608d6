// N-API addon: the thin binding between Node.js and the C ABI of libcompressjs_amd.so
// (include/compressjs_amd.h).  Built by plain g++ (no node-gyp): see __graft_entry__.build_addon().
//
//   compress(input: Buffer|Uint8Array, level: number) -> Buffer      = Bzip2.compressFile hot path
//   bwtransform2(T: Uint8Array, U: Uint8Array, n: number) -> pidx     = BWT.bwtransform2
//
// The shared library is dlopen()ed at require() time from ../compressjs_amd/ (or
// $COMPRESSJS_AMD_LIB); a missing library or a missing GPU surfaces as a thrown Error -- there is
// no JavaScript/CPU fallback for the accelerated path.
#include <node_api.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <mutex>
#include <sys/mman.h>

typedef struct cjs_ctx cjs_ctx;
static cjs_ctx* (*p_create)(int, uint32_t);
static void (*p_destroy)(cjs_ctx*);
static int64_t (*p_bound)(uint64_t);
static int64_t (*p_compress)(cjs_ctx*, const uint8_t*, uint64_t, int, uint8_t*, uint64_t);
static int64_t (*p_compress_multi)(cjs_ctx**, uint32_t, const uint8_t*, uint64_t, int, uint8_t*, uint64_t);
static int32_t (*p_device_count)(void);
static int32_t (*p_bwt)(const uint8_t*, uint8_t*, uint32_t, uint32_t*);
static int32_t (*p_bwtlin)(const uint8_t*, uint8_t*, uint32_t, uint32_t*);
static int32_t (*p_sufsort)(const uint8_t*, int32_t*, uint32_t);
static int32_t (*p_unbwt)(const uint8_t*, uint8_t*, uint32_t, uint32_t);
static int32_t (*p_hufflen)(int64_t*, uint32_t, uint32_t);
static int64_t (*p_dec)(cjs_ctx*, const uint8_t*, uint64_t, uint8_t*, uint64_t, int);
static int64_t (*p_decblk)(cjs_ctx*, const uint8_t*, uint64_t, uint64_t, uint8_t*, uint64_t);
static int64_t (*p_table)(cjs_ctx*, const uint8_t*, uint64_t, int, uint64_t*, uint64_t*, uint32_t);
static int64_t (*p_lastsize)(cjs_ctx*);
static int64_t (*p_fetch)(cjs_ctx*, uint8_t*, uint64_t);
static int32_t (*p_detail)(cjs_ctx*, uint32_t*, uint32_t*);
static int64_t (*p_bwtc_dec)(cjs_ctx*, const uint8_t*, uint64_t, uint8_t*, uint64_t, int64_t*);
static int64_t (*p_bwtc_lastsize)(cjs_ctx*);
static int64_t (*p_bwtc_fetch)(cjs_ctx*, uint8_t*, uint64_t);
static int64_t (*p_bwtc_bound)(uint64_t);
static int64_t (*p_bwtc)(cjs_ctx*, const uint8_t*, uint64_t, int, uint8_t*, uint64_t, int64_t);
static void* g_lib;
static cjs_ctx* g_ctx;                       // context on the first configured device: every single-device entry point
static std::vector<cjs_ctx*> g_ctxs;         // one context per configured device (g_ctxs[0] == g_ctx)
// configuration: configure({devices, blocksInFlight}) of js/index.js, or COMPRESSJS_AMD_DEVICES ("all", "0,1,2,3")
// and COMPRESSJS_AMD_BLOCKS in the environment; default: device 0, 128 blocks in flight (what bench.py uses)
static std::vector<int> g_devices;
static uint32_t g_blocks = 0;
static std::string g_err;

static bool load_lib(const char* path) {
    if (g_lib) return true;
    g_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) { g_err = dlerror(); return false; }
    p_create = (cjs_ctx * (*)(int, uint32_t)) dlsym(g_lib, "cjs_create");
    p_destroy = (void (*)(cjs_ctx*))dlsym(g_lib, "cjs_destroy");
    p_bound = (int64_t(*)(uint64_t))dlsym(g_lib, "cjs_bz2_compress_bound");
    p_compress = (int64_t(*)(cjs_ctx*, const uint8_t*, uint64_t, int, uint8_t*, uint64_t))dlsym(g_lib, "cjs_bz2_compress");
    p_compress_multi = (int64_t(*)(cjs_ctx**, uint32_t, const uint8_t*, uint64_t, int, uint8_t*, uint64_t))dlsym(g_lib, "cjs_bz2_compress_multi");
    p_device_count = (int32_t(*)(void))dlsym(g_lib, "cjs_device_count");
    p_bwt = (int32_t(*)(const uint8_t*, uint8_t*, uint32_t, uint32_t*))dlsym(g_lib, "cjs_bwt_cyclic");
    p_bwtlin = (int32_t(*)(const uint8_t*, uint8_t*, uint32_t, uint32_t*))dlsym(g_lib, "cjs_bwt_linear");
    p_sufsort = (int32_t(*)(const uint8_t*, int32_t*, uint32_t))dlsym(g_lib, "cjs_suffixsort");
    p_unbwt = (int32_t(*)(const uint8_t*, uint8_t*, uint32_t, uint32_t))dlsym(g_lib, "cjs_unbwt_linear");
    p_hufflen = (int32_t(*)(int64_t*, uint32_t, uint32_t))dlsym(g_lib, "cjs_huff_lengths");
    p_dec = (int64_t(*)(cjs_ctx*, const uint8_t*, uint64_t, uint8_t*, uint64_t, int))dlsym(g_lib, "cjs_bz2_decompress");
    p_decblk = (int64_t(*)(cjs_ctx*, const uint8_t*, uint64_t, uint64_t, uint8_t*, uint64_t))dlsym(g_lib, "cjs_bz2_decompress_block");
    p_table = (int64_t(*)(cjs_ctx*, const uint8_t*, uint64_t, int, uint64_t*, uint64_t*, uint32_t))dlsym(g_lib, "cjs_bz2_table");
    p_lastsize = (int64_t(*)(cjs_ctx*))dlsym(g_lib, "cjs_bz2_last_size");
    p_fetch = (int64_t(*)(cjs_ctx*, uint8_t*, uint64_t))dlsym(g_lib, "cjs_bz2_fetch");
    p_detail = (int32_t(*)(cjs_ctx*, uint32_t*, uint32_t*))dlsym(g_lib, "cjs_bz2_last_detail");
    p_bwtc_dec = (int64_t(*)(cjs_ctx*, const uint8_t*, uint64_t, uint8_t*, uint64_t, int64_t*))dlsym(g_lib, "cjs_bwtc_decompress");
    p_bwtc_lastsize = (int64_t(*)(cjs_ctx*))dlsym(g_lib, "cjs_bwtc_last_size");
    p_bwtc_fetch = (int64_t(*)(cjs_ctx*, uint8_t*, uint64_t))dlsym(g_lib, "cjs_bwtc_fetch");
    p_bwtc_bound = (int64_t(*)(uint64_t))dlsym(g_lib, "cjs_bwtc_compress_bound");
    p_bwtc = (int64_t(*)(cjs_ctx*, const uint8_t*, uint64_t, int, uint8_t*, uint64_t, int64_t))dlsym(g_lib, "cjs_bwtc_compress");
    if (!p_compress_multi || !p_device_count || !p_create || !p_destroy || !p_bound || !p_compress || !p_bwt || !p_bwtlin || !p_sufsort || !p_unbwt || !p_hufflen || !p_dec || !p_decblk || !p_table || !p_lastsize || !p_fetch || !p_detail || !p_bwtc_dec || !p_bwtc_lastsize || !p_bwtc_fetch || !p_bwtc || !p_bwtc_bound) { g_err = "missing symbols"; return false; }
    return true;
}

static napi_value throw_code(napi_env env, int64_t rc, const char* what) {
    char msg[160];
    if (rc == -20) snprintf(msg, sizeof msg, "Invalid block size multiplier");      // lib/Bzip2.js:889
    else if (rc == -24) snprintf(msg, sizeof msg, "%s: not supported by this build (code -24)", what);
    else if (rc == -23) snprintf(msg, sizeof msg, "%s: no HIP device visible (compressjs_amd has no CPU path)", what);
    else snprintf(msg, sizeof msg, "%s failed with code %lld", what, (long long)rc);
    napi_throw_error(env, nullptr, msg);
    return nullptr;
}

static bool get_bytes(napi_env env, napi_value v, uint8_t** data, size_t* len) {
    bool is = false;
    if (napi_is_buffer(env, v, &is) == napi_ok && is)
        return napi_get_buffer_info(env, v, (void**)data, len) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type t; napi_value ab; size_t off;
        if (napi_get_typedarray_info(env, v, &t, len, (void**)data, &ab, &off) != napi_ok) return false;
        return t == napi_uint8_array || t == napi_int8_array || t == napi_uint8_clamped_array;
    }
    return false;
}

static napi_value Load(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    char path[4096]; size_t n = 0;
    napi_get_value_string_utf8(env, argv[0], path, sizeof path, &n);
    napi_value r;
    napi_get_boolean(env, load_lib(path), &r);
    return r;
}

static napi_value LastError(napi_env env, napi_callback_info) {
    napi_value r;
    napi_create_string_utf8(env, g_err.c_str(), NAPI_AUTO_LENGTH, &r);
    return r;
}

static void drop_ctxs() {
    for (cjs_ctx* c : g_ctxs) p_destroy(c);
    g_ctxs.clear();
    g_ctx = nullptr;
}

static void config_from_env() {
    if (g_devices.empty()) {
        const char* e = getenv("COMPRESSJS_AMD_DEVICES");
        if (e && !strcmp(e, "all")) { const int n = p_device_count(); for (int i = 0; i < n; i++) g_devices.push_back(i); }
        else if (e) { for (const char* p = e; *p;) { char* q; const long v = strtol(p, &q, 10); if (q == p) break; g_devices.push_back((int)v); p = *q == ',' ? q + 1 : q; } }
        if (g_devices.empty()) g_devices.push_back(0);
    }
    if (!g_blocks) {
        const char* e = getenv("COMPRESSJS_AMD_BLOCKS");
        const long v = e ? strtol(e, nullptr, 10) : 0;
        g_blocks = v >= 1 && v <= 4096 ? (uint32_t)v : 128u;
    }
}

static bool ensure_ctx(napi_env env) {
    if (!g_lib) { napi_throw_error(env, nullptr, ("libcompressjs_amd.so not loaded: " + g_err).c_str()); return false; }
    if (g_ctx) return true;
    config_from_env();
    for (int d : g_devices) {
        cjs_ctx* c = p_create(d, g_blocks);
        if (!c) { drop_ctxs(); throw_code(env, -23, "cjs_create"); return false; }
        g_ctxs.push_back(c);
    }
    g_ctx = g_ctxs[0];
    return true;
}

// configure(devices: number[] | null, blocksInFlight: number | 0): takes effect for the next call (contexts are rebuilt)
static napi_value Configure(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    if (!g_lib) { napi_throw_error(env, nullptr, ("libcompressjs_amd.so not loaded: " + g_err).c_str()); return nullptr; }
    std::vector<int> devs;
    bool isarr = false;
    if (argc >= 1 && napi_is_array(env, argv[0], &isarr) == napi_ok && isarr) {
        uint32_t n = 0; napi_get_array_length(env, argv[0], &n);
        for (uint32_t i = 0; i < n; i++) { napi_value v; int32_t d = 0; napi_get_element(env, argv[0], i, &v); napi_get_value_int32(env, v, &d); devs.push_back(d); }
    }
    int32_t blocks = 0;
    if (argc >= 2) napi_get_value_int32(env, argv[1], &blocks);
    const int have = p_device_count();
    for (int d : devs) if (d < 0 || d >= have) { napi_throw_range_error(env, nullptr, "configure: no such HIP device"); return nullptr; }
    if (blocks < 0 || blocks > 4096) { napi_throw_range_error(env, nullptr, "configure: blocksInFlight must be 1..4096"); return nullptr; }
    drop_ctxs();
    if (!devs.empty()) g_devices = devs;
    if (blocks) g_blocks = (uint32_t)blocks;
    napi_value r; napi_create_int32(env, have, &r);
    return r;
}

static napi_value DeviceCount(napi_env env, napi_callback_info) {
    napi_value r; napi_create_int32(env, g_lib ? p_device_count() : 0, &r);
    return r;
}

// result blocks of Compress: a pool of at most four (what is beyond that is freed when its Buffer is collected).  The pool is shared by
// every environment of the process (worker_threads load the same addon): guarded by a mutex.
struct StageBlock { uint8_t* data; uint64_t cap; uint64_t size; int64_t accounted; uint64_t hw; };   // size: bytes of address space; accounted: bytes V8 was told an external Buffer over this block holds; hw: bytes ever written (resident pages)
static std::vector<StageBlock*> g_stage_pool;
static std::mutex g_stage_mu;
static StageBlock* stage_take(uint64_t cap) {
    {
        std::lock_guard<std::mutex> g(g_stage_mu);
        for (size_t i = 0; i < g_stage_pool.size(); i++)
            if (g_stage_pool[i]->cap >= cap) { StageBlock* b = g_stage_pool[i]; g_stage_pool.erase(g_stage_pool.begin() + (long)i); return b; }
    }
    // 2 MB-aligned and advised for transparent huge pages: a fresh block's pages are faulted in by the copy that brings the result home
    // (7 200 small pages for the 29 MB of a 10^8-byte text; results that JavaScript still holds cannot be reused)
    const uint64_t sz = ((cap ? cap : 1) + ((2u << 20) - 1)) & ~(uint64_t)((2u << 20) - 1);
    void* mem = nullptr;
    if (posix_memalign(&mem, 2u << 20, sz) != 0) return nullptr;
    (void)madvise(mem, sz, MADV_HUGEPAGE);
    StageBlock* b = new StageBlock{(uint8_t*)mem, cap, sz, 0, 0};
    return b;
}
static void stage_give(StageBlock* b) {
    {
        std::lock_guard<std::mutex> g(g_stage_mu);
        if (g_stage_pool.size() < 4) { g_stage_pool.push_back(b); return; }
    }
    free(b->data);
    delete b;
}
static void stage_finalize(napi_env env, void*, void* hint) {
    StageBlock* b = (StageBlock*)hint;
    int64_t now;
    if (b->accounted) { napi_adjust_external_memory(env, -b->accounted, &now); b->accounted = 0; }
    stage_give(b);
}
// A result of at least 1 MB goes to JavaScript as an EXTERNAL Buffer over its block (below that a copy is microseconds, and a few KB must not pin a
// block).  A block is sized for the worst case (1.5 x the input + 24 KB per 100 KB: 175 MB of ADDRESS SPACE for a 10^8-byte input), but only the
// pages a result was written into are resident; what an earlier, larger result left resident behind the current one is given back to the system
// before the Buffer is handed over (madvise DONTNEED), and V8 is told the resident size (a retained 29 MB result holds 30 MB, not 175).
static bool stage_external(const StageBlock*, uint64_t n) { return n >= (1u << 20); }
static uint64_t stage_trim(StageBlock* b, uint64_t n) {
    const uint64_t keep = (n + ((2u << 20) - 1)) & ~(uint64_t)((2u << 20) - 1);
    if (b->hw > keep) (void)madvise(b->data + keep, (size_t)(b->hw - keep), MADV_DONTNEED);
    b->hw = keep < b->size ? keep : b->size;
    return b->hw;
}

static napi_value Compress(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t* in; size_t len;
    if (argc < 2 || !get_bytes(env, argv[0], &in, &len)) { napi_throw_type_error(env, nullptr, "compress(bytes, level)"); return nullptr; }
    int32_t level = 9;
    napi_get_value_int32(env, argv[1], &level);
    if (level < 1 || level > 9) return throw_code(env, -20, "compress");
    if (!ensure_ctx(env)) return nullptr;
    const uint64_t cap = (uint64_t)p_bound(len);
    // The result is handed to JavaScript as an EXTERNAL Buffer over the block the library wrote into - no copy (round 5; the copy into a
    // fresh Buffer was 4 of the 14 ms of a 10^8-byte call).  Blocks come from a small pool and go back to it when the Buffer is collected:
    // a fresh malloc would fault in every page the D2H copy touches.
    StageBlock* blk = stage_take(cap);
    if (!blk) { napi_throw_error(env, nullptr, "out of memory"); return nullptr; }
    // several devices configured: segments of the input go round-robin to them (cjs_bz2_compress_multi)
    const int64_t n = g_ctxs.size() > 1 ? p_compress_multi(g_ctxs.data(), (uint32_t)g_ctxs.size(), in, len, level, blk->data, cap)
                                        : p_compress(g_ctx, in, len, level, blk->data, cap);
    if (n < 0) { stage_give(blk); return throw_code(env, n, "cjs_bz2_compress"); }
    napi_value out;
    if ((uint64_t)n > blk->hw) blk->hw = (uint64_t)n;
    if (!stage_external(blk, (uint64_t)n) || napi_create_external_buffer(env, (size_t)n, blk->data, stage_finalize, blk, &out) != napi_ok) {
        void* dst;                                           // (small results, and embedders without external buffers: a copy)
        napi_create_buffer_copy(env, (size_t)n, blk->data, &dst, &out);
        stage_give(blk);
    } else {
        // V8 sees a Buffer object of a few dozen bytes: tell it what hangs on it - the block's resident pages -, or results pile up
        // uncollected (every call a fresh block whose pages the D2H copy has to fault in) until the heap of small objects happens to fill
        int64_t now;
        blk->accounted = (int64_t)stage_trim(blk, (uint64_t)n);
        napi_adjust_external_memory(env, blk->accounted, &now);
    }
    return out;
}

static napi_value Bwt2(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t *T, *U; size_t tl, ul; uint32_t n = 0;
    if (argc < 3 || !get_bytes(env, argv[0], &T, &tl) || !get_bytes(env, argv[1], &U, &ul)) { napi_throw_type_error(env, nullptr, "bwtransform2(T, U, n)"); return nullptr; }
    napi_get_value_uint32(env, argv[2], &n);
    if (n > tl || n > ul) { napi_throw_range_error(env, nullptr, "n exceeds the arrays"); return nullptr; }
    if (!g_lib) { napi_throw_error(env, nullptr, "libcompressjs_amd.so not loaded"); return nullptr; }
    uint32_t pidx = 0;
    const int32_t rc = p_bwt(T, U, n, &pidx);
    if (rc < 0) return throw_code(env, rc, "cjs_bwt_cyclic");
    napi_value r;
    napi_create_uint32(env, pidx, &r);
    return r;
}

// bwtcCompress(bytes, level, declaredSize) -> Buffer            = BWTC.compressFile hot path
static napi_value BwtcCompress(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t* in; size_t len;
    if (argc < 3 || !get_bytes(env, argv[0], &in, &len)) { napi_throw_type_error(env, nullptr, "bwtcCompress(bytes, level, size)"); return nullptr; }
    int32_t level = 9; int64_t declared = -1;
    napi_get_value_int32(env, argv[1], &level);
    napi_get_value_int64(env, argv[2], &declared);
    if (!ensure_ctx(env)) return nullptr;
    const uint64_t cap = (uint64_t)p_bwtc_bound(len);
    uint8_t* tmp = (uint8_t*)malloc(cap);
    if (!tmp) { napi_throw_error(env, nullptr, "out of memory"); return nullptr; }
    const int64_t n = p_bwtc(g_ctx, in, len, level, tmp, cap, declared);
    if (n < 0) { free(tmp); return throw_code(env, n, "cjs_bwtc_compress"); }
    napi_value out; void* dst;
    napi_create_buffer_copy(env, (size_t)n, tmp, &dst, &out);
    free(tmp);
    return out;
}

// bwtransform(T, U, n) -> pidx                                    = BWT.bwtransform
static napi_value BwtLinear(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t *T, *U; size_t tl, ul; uint32_t n = 0;
    if (argc < 3 || !get_bytes(env, argv[0], &T, &tl) || !get_bytes(env, argv[1], &U, &ul)) { napi_throw_type_error(env, nullptr, "bwtransform(T, U, n)"); return nullptr; }
    napi_get_value_uint32(env, argv[2], &n);
    if (n > tl || n > ul) { napi_throw_range_error(env, nullptr, "n exceeds the arrays"); return nullptr; }
    if (!g_lib) { napi_throw_error(env, nullptr, "libcompressjs_amd.so not loaded"); return nullptr; }
    uint32_t pidx = 0;
    const int32_t rc = p_bwtlin(T, U, n, &pidx);
    if (rc < 0) return throw_code(env, rc, "cjs_bwt_linear");
    napi_value r;
    napi_create_uint32(env, pidx, &r);
    return r;
}

// unbwtransform(T, U, n, pidx)                                   = BWT.unbwtransform
static napi_value UnBwtLinear(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t *T, *U; size_t tl, ul; uint32_t n = 0, pidx = 0;
    if (argc < 4 || !get_bytes(env, argv[0], &T, &tl) || !get_bytes(env, argv[1], &U, &ul)) { napi_throw_type_error(env, nullptr, "unbwtransform(T, U, n, pidx)"); return nullptr; }
    napi_get_value_uint32(env, argv[2], &n);
    napi_get_value_uint32(env, argv[3], &pidx);
    if (n > tl || n > ul) { napi_throw_range_error(env, nullptr, "n exceeds the arrays"); return nullptr; }
    if (!g_lib) { napi_throw_error(env, nullptr, "libcompressjs_amd.so not loaded"); return nullptr; }
    const int32_t rc = p_unbwt(T, U, n, pidx);
    if (rc < 0) return throw_code(env, rc, "cjs_unbwt_linear");
    napi_value r;
    napi_get_undefined(env, &r);
    return r;
}

// huffLengths(a: Float64Array, maxLen)  in place                 = allocateHuffmanCodeLengths
static napi_value HuffLengths(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    napi_typedarray_type tt; size_t n; void* data; napi_value ab; size_t off; uint32_t maxlen = 0;
    if (argc < 2 || napi_get_typedarray_info(env, argv[0], &tt, &n, &data, &ab, &off) != napi_ok || tt != napi_float64_array) {
        napi_throw_type_error(env, nullptr, "huffLengths(Float64Array, maxLen)"); return nullptr;
    }
    napi_get_value_uint32(env, argv[1], &maxlen);
    if (!g_lib) { napi_throw_error(env, nullptr, "libcompressjs_amd.so not loaded"); return nullptr; }
    std::vector<int64_t> a(n);
    double* d = (double*)data;
    for (size_t i = 0; i < n; i++) a[i] = (int64_t)d[i];
    const int32_t rc = p_hufflen(a.data(), (uint32_t)n, maxlen);
    if (rc < 0) return throw_code(env, rc, "cjs_huff_lengths");
    for (size_t i = 0; i < n; i++) d[i] = (double)a[i];
    napi_value r;
    napi_get_undefined(env, &r);
    return r;
}

// Bunzip's _throw(status, optDetail) (lib/Bzip2.js:82-88): TypeError with .errorCode and the reference's text
static napi_value throw_decode(napi_env env, int64_t rc) {
    const char* base = rc == -2 ? "Not bzip data" : rc == -5 ? "Data error" : rc == -7 ? "Obsolete (pre 0.9.5) bzip format not supported." : nullptr;
    if (!base) return throw_code(env, rc, "cjs_bz2_decompress");
    uint32_t got = 0, want = 0;
    const int d = p_detail(g_ctx, &got, &want);
    char msg[200];
    if (d == 1) snprintf(msg, sizeof msg, "%s: bad magic", base);
    else if (d == 2) snprintf(msg, sizeof msg, "%s: level out of range", base);
    else if (d == 3) snprintf(msg, sizeof msg, "%s: initial position out of bounds", base);
    else if (d == 4) snprintf(msg, sizeof msg, "%s: Bad block CRC (got %x expected %x)", base, got, want);
    else if (d == 5) snprintf(msg, sizeof msg, "%s: Bad stream CRC (got %x expected %x)", base, got, want);
    else snprintf(msg, sizeof msg, "%s", base);
    napi_value m, e, code;
    napi_create_string_utf8(env, msg, NAPI_AUTO_LENGTH, &m);
    napi_create_type_error(env, nullptr, m, &e);
    napi_create_int32(env, (int32_t)rc, &code);
    napi_set_named_property(env, e, "errorCode", code);
    napi_throw(env, e);
    return nullptr;
}
static napi_value fetch_result(napi_env env, int64_t n) {
    if (n == -21) n = p_lastsize(g_ctx);                       // decoded; the size is known now
    else if (n < 0) return throw_decode(env, n);
    napi_value out; void* dst = nullptr;
    if (n >= (1 << 20)) {
        // a large result: fetched into a pooled, huge-page-advised block handed over as an external Buffer (as Compress does) - a fresh
        // Node Buffer of 10^8 bytes is 24 400 small pages faulted in by the copy
        StageBlock* blk = stage_take((uint64_t)n);
        if (blk) {
            const int64_t m = p_fetch(g_ctx, blk->data, (uint64_t)n);
            if (m < 0) { stage_give(blk); return throw_code(env, m, "cjs_bz2_fetch"); }
            if ((uint64_t)n > blk->hw) blk->hw = (uint64_t)n;
            if (napi_create_external_buffer(env, (size_t)n, blk->data, stage_finalize, blk, &out) == napi_ok) {
                int64_t now;
                blk->accounted = (int64_t)stage_trim(blk, (uint64_t)n);
                napi_adjust_external_memory(env, blk->accounted, &now);
                return out;
            }
            napi_create_buffer_copy(env, (size_t)n, blk->data, &dst, &out);
            stage_give(blk);
            return out;
        }
    }
    napi_create_buffer(env, (size_t)n, &dst, &out);
    if (n > 0) {
        const int64_t m = p_fetch(g_ctx, (uint8_t*)dst, (uint64_t)n);
        if (m < 0) return throw_code(env, m, "cjs_bz2_fetch");
    }
    return out;
}
// decompress(bytes, multistream) -> Buffer                        = Bzip2.decompressFile
static napi_value Decompress(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t* in; size_t len; bool ms = false;
    if (argc < 1 || !get_bytes(env, argv[0], &in, &len)) { napi_throw_type_error(env, nullptr, "decompress(bytes, multistream)"); return nullptr; }
    if (argc > 1) napi_get_value_bool(env, argv[1], &ms);
    if (!ensure_ctx(env)) return nullptr;
    return fetch_result(env, p_dec(g_ctx, in, len, nullptr, 0, ms ? 1 : 0));
}
// decompressBlock(bytes, bitPos) -> Buffer                        = Bzip2.decompressBlock
static napi_value DecompressBlock(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t* in; size_t len; double pos = 0;
    if (argc < 2 || !get_bytes(env, argv[0], &in, &len)) { napi_throw_type_error(env, nullptr, "decompressBlock(bytes, bitPos)"); return nullptr; }
    napi_get_value_double(env, argv[1], &pos);
    if (!ensure_ctx(env)) return nullptr;
    return fetch_result(env, p_decblk(g_ctx, in, len, (uint64_t)pos, nullptr, 0));
}
// table(bytes, multistream) -> Float64Array [pos0, size0, pos1, size1, ...]   = Bzip2.table
static napi_value Table(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t* in; size_t len; bool ms = false;
    if (argc < 1 || !get_bytes(env, argv[0], &in, &len)) { napi_throw_type_error(env, nullptr, "table(bytes, multistream)"); return nullptr; }
    if (argc > 1) napi_get_value_bool(env, argv[1], &ms);
    if (!ensure_ctx(env)) return nullptr;
    const uint32_t cap = (uint32_t)(len / 8 + 16);
    std::vector<uint64_t> pos(cap), size(cap);
    const int64_t n = p_table(g_ctx, in, len, ms ? 1 : 0, pos.data(), size.data(), cap);
    if (n < 0) return throw_decode(env, n);
    napi_value ab, arr; void* data = nullptr;
    napi_create_arraybuffer(env, (size_t)n * 16, &data, &ab);
    double* d = (double*)data;
    for (int64_t i = 0; i < n; i++) { d[2 * i] = (double)pos[i]; d[2 * i + 1] = (double)size[i]; }
    napi_create_typedarray(env, napi_float64_array, (size_t)n * 2, ab, 0, &arr);
    return arr;
}

// bwtcDecompress(bytes) -> Buffer                                = BWTC.decompressFile (levels 6-9)
static napi_value BwtcDecompress(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t* in; size_t len;
    if (argc < 1 || !get_bytes(env, argv[0], &in, &len)) { napi_throw_type_error(env, nullptr, "bwtcDecompress(bytes)"); return nullptr; }
    if (!ensure_ctx(env)) return nullptr;
    int64_t declared = -1;
    int64_t n = p_bwtc_dec(g_ctx, in, len, nullptr, 0, &declared);
    if (n == -21) n = p_bwtc_lastsize(g_ctx);
    else if (n == -30) { napi_throw_error(env, nullptr, "Bad magic"); return nullptr; }      // lib/Util.js:150-152
    else if (n < 0) return throw_code(env, n, "cjs_bwtc_decompress");
    napi_value out; void* dst = nullptr;
    napi_create_buffer(env, (size_t)n, &dst, &out);
    if (n > 0 && p_bwtc_fetch(g_ctx, (uint8_t*)dst, (uint64_t)n) < 0) return throw_code(env, -22, "cjs_bwtc_fetch");
    return out;
}

// suffixsort(T, SA: Int32Array, n)                                = BWT.suffixsort
static napi_value SuffixSort(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    uint8_t* T; size_t tl; uint32_t n = 0;
    if (argc < 3 || !get_bytes(env, argv[0], &T, &tl)) { napi_throw_type_error(env, nullptr, "suffixsort(T, SA, n)"); return nullptr; }
    napi_typedarray_type tt; size_t sl; void* sa; napi_value ab; size_t off;
    if (napi_get_typedarray_info(env, argv[1], &tt, &sl, &sa, &ab, &off) != napi_ok || tt != napi_int32_array) { napi_throw_type_error(env, nullptr, "SA must be an Int32Array"); return nullptr; }
    napi_get_value_uint32(env, argv[2], &n);
    if (n > tl || n > sl) { napi_throw_range_error(env, nullptr, "n exceeds the arrays"); return nullptr; }
    if (!g_lib) { napi_throw_error(env, nullptr, "libcompressjs_amd.so not loaded"); return nullptr; }
    const int32_t rc = p_sufsort(T, (int32_t*)sa, n);
    if (rc < 0) return throw_code(env, rc, "cjs_suffixsort");
    napi_value r;
    napi_create_int32(env, 0, &r);
    return r;
}

static napi_value Init(napi_env env, napi_value exports) {
    napi_property_descriptor d[] = {
        {"load", nullptr, Load, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"lastError", nullptr, LastError, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"compress", nullptr, Compress, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"bwtransform2", nullptr, Bwt2, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"bwtransform", nullptr, BwtLinear, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"unbwtransform", nullptr, UnBwtLinear, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"decompress", nullptr, Decompress, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"decompressBlock", nullptr, DecompressBlock, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"table", nullptr, Table, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"bwtcDecompress", nullptr, BwtcDecompress, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"huffLengths", nullptr, HuffLengths, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"suffixsort", nullptr, SuffixSort, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"bwtcCompress", nullptr, BwtcCompress, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"configure", nullptr, Configure, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"deviceCount", nullptr, DeviceCount, nullptr, nullptr, nullptr, napi_default, nullptr},
    };
    napi_define_properties(env, exports, sizeof d / sizeof d[0], d);
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
