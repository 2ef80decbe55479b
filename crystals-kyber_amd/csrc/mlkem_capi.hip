// mlkem_capi.hip — C-ABI of libmlkem_amd.so (include/mlkem_batch.h) over the gfx950 kernels.
//
// No CPU fallback exists here by design: every entry point either enqueues HIP kernels or fails with
// MLKEM_ERR_NO_DEVICE / a HIP error.  The CPU oracle under oracle/ is test infrastructure and is never
// linked or loaded by this library.
#include "mlkem_pipeline.hpp"
#include "mlkem_selftest.hpp"

#include "../../include/mlkem_batch.h"

#include <sys/random.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace mlkem;

namespace {

thread_local std::string g_last_hip_error;

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    g_last_hip_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        if (!hip_ok((expr), #expr)) return MLKEM_ERR_NO_DEVICE; \
    } while (0)

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

size_t default_chunk() {
    if (const char* e = getenv("MLKEM_CHUNK_ITEMS")) {
        long long v = atoll(e);
        if (v > 0) return (size_t)v;
    }
    return (size_t)1 << 18;
}

}   // namespace

struct mlkem_ctx {
    int device = 0;
    size_t chunk = 0;
    size_t scratch_bytes = 0;
    void* scratch = nullptr;
    Workspace ws;
    // side stream of one-chunk calls (SideFork, mlkem_pipeline.hpp).  Created on the first call that can use it, not with the
    // context: an idle fourth stream in the process costs the streaming front-end 10 % of its pinned-buffer rate (the runtime
    // multiplexes streams onto a few hardware queues; profiles/r03_batch_sweep.txt).  The front-end lends its idle copy stream
    // to its own context instead (stream_op).
    bool side_allowed = true;
    hipStream_t own_side = nullptr;
};

// called by the *_dev entry points that fork: gives the context its side stream the first time a call fits one chunk
static void ctx_arm_side(mlkem_ctx* ctx, size_t n) {
    if (!ctx->side_allowed || !ctx->ws.side_on || ctx->ws.side || n == 0 || n > ctx->ws.cap) return;
    if (hipStreamCreateWithFlags(&ctx->own_side, hipStreamNonBlocking) == hipSuccess) {
        ctx->ws.side = ctx->own_side;
    } else {
        ctx->own_side = nullptr;
        ctx->side_allowed = false;   // the context works without the overlap
        (void)hipGetLastError();
    }
}

// a context is bound to the device it was created on: its scratch lives there and `*_dev` calls launch on the CALLER's
// current device, so a call made while another device is current is rejected instead of faulting on foreign memory
static bool ctx_ok(const mlkem_ctx* ctx) {
    int cur = -1;
    return ctx && hipGetDevice(&cur) == hipSuccess && cur == ctx->device;
}

extern "C" {

int mlkem_sizes(int set, unsigned* ek_len, unsigned* dk_len, unsigned* c_len) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (ek_len) *ek_len = p.ek_len;
    if (dk_len) *dk_len = p.dk_len;
    if (c_len) *c_len = p.c_len;
    return MLKEM_OK;
}

int mlkem_params(int set, int out[5]) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!out) return MLKEM_ERR_ARG;
    out[0] = p.k; out[1] = p.eta1; out[2] = p.eta2; out[3] = p.du; out[4] = p.dv;
    return MLKEM_OK;
}

int mlkem_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mlkem_strerror(int code) {
    switch (code) {
    case MLKEM_OK: return "ok";
    case MLKEM_ERR_PARAM_SET: return "invalid parameter set (reference ml_errno -1)";
    case MLKEM_ERR_RNG: return "random bit generation failed (reference ml_errno -2)";
    case MLKEM_ERR_LENGTH: return "type check failed: wrong ek/dk/c length (reference ml_errno -3)";
    case MLKEM_ERR_MODULUS: return "modulus check failed (reference ml_errno -4; unreachable)";
    case MLKEM_ERR_HASH: return "decapsulation key hash check failed (reference ml_errno -5)";
    case MLKEM_ERR_NO_DEVICE: return "no usable HIP device / HIP runtime error (no CPU fallback exists)";
    case MLKEM_ERR_ARG: return "bad argument (NULL or misaligned pointer)";
    case MLKEM_ERR_ALLOC: return "allocation failed";
    default: return "unknown error";
    }
}

const char* mlkem_last_hip_error(void) { return g_last_hip_error.c_str(); }

int mlkem_ctx_create(mlkem_ctx** out, int device, size_t chunk_items) {
    if (!out) return MLKEM_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (!hip_ok(hipGetDeviceCount(&ndev), "hipGetDeviceCount") || ndev <= 0) return MLKEM_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return MLKEM_ERR_ARG;
    HIP_TRY(hipSetDevice(device));
    mlkem_ctx* c = new (std::nothrow) mlkem_ctx();
    if (!c) return MLKEM_ERR_ALLOC;
    c->device = device;
    c->chunk = chunk_items ? chunk_items : default_chunk();
    const size_t n = c->chunk;
    // hash stages run over h-chunks of 8 chunks (>= 2^16 items keeps 1024 SIMDs busy with one sponge per lane)
    size_t hn = n * 8;
    if (const char* e = getenv("MLKEM_HCHUNK_ITEMS")) {
        long long v = atoll(e);
        if (v > 0) hn = (size_t)v;
    }
    if (hn < n) hn = n;
    // carve one allocation: A | prf | leftover | resume | r | rho | m | Kp | Kbar, each 256-byte aligned
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t rcap = n * 2 < 64 ? 64 : n * 2;   // resume records: 1/8 of the chunk's (up to 16 n) sponges, expected 0.8 %
    const size_t szA = up(n * 16 * 512), szP = up(n * 9 * 192), szL = up((n * 16 + 2) * 4), szR = up(rcap * RESUME_WORDS * 4), sz32 = up(hn * 32);
    c->scratch_bytes = szA + szP + szL + szR + 5 * sz32;
    if (!hip_ok(hipMalloc(&c->scratch, c->scratch_bytes), "hipMalloc(scratch)")) {
        delete c;
        return MLKEM_ERR_ALLOC;
    }
    uint8_t* base = static_cast<uint8_t*>(c->scratch);
    c->ws.A = reinterpret_cast<uint16_t*>(base);
    c->ws.prf = base + szA;
    c->ws.leftover = reinterpret_cast<uint32_t*>(base + szA + szP);
    c->ws.resume = reinterpret_cast<uint32_t*>(base + szA + szP + szL);
    c->ws.resume_cap = (uint32_t)rcap;
    c->ws.r = base + szA + szP + szL + szR;
    c->ws.rho = c->ws.r + sz32;
    c->ws.m = c->ws.rho + sz32;
    c->ws.Kp = c->ws.m + sz32;
    c->ws.Kbar = c->ws.Kp + sz32;
    c->ws.cap = n;
    c->ws.hcap = hn;
    if (const char* e = getenv("MLKEM_WIDE_HASH_ITEMS")) {   // 0: always the lane-sliced hash kernels
        long long v = atoll(e);
        if (v >= 0) c->ws.wide_max = c->ws.wide_max_k[0] = c->ws.wide_max_k[1] = c->ws.wide_max_k[2] = (size_t)v;
    }
    if (const char* e = getenv("MLKEM_SMALL_ITEMS")) {       // 0: never the one-workgroup-per-item kernels
        long long v = atoll(e);
        if (v >= 0) c->ws.small_max_k[0] = c->ws.small_max_k[1] = c->ws.small_max_k[2] = (size_t)v;
    }
    if (const char* e = getenv("MLKEM_SMALL_WIDE_ITEMS")) {      // Decaps calls up to this size: twelve waves per item (k >= 3)
        long long v = atoll(e);
        if (v >= 0) c->ws.small_wide_max = (size_t)v;
    }
    if (const char* e = getenv("MLKEM_SMALL_LATENCY_ITEMS")) {   // small calls above this size: four waves per item instead of eight
        long long v = atoll(e);
        if (v >= 0) c->ws.small_lat_max = (size_t)v;
    }
    // fork / join events of one-chunk calls (SideFork, mlkem_pipeline.hpp); the side stream itself comes with the first such call
    // (ctx_arm_side).  MLKEM_SIDE_STREAM=0 keeps every call on the caller's stream; failing to create the events does the same.
    const char* se = getenv("MLKEM_SIDE_STREAM");
    c->side_allowed = !(se && atoi(se) == 0);
    if (c->side_allowed && (hipEventCreateWithFlags(&c->ws.ev_fork, hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&c->ws.ev_join, hipEventDisableTiming) != hipSuccess)) {
        if (c->ws.ev_fork) (void)hipEventDestroy(c->ws.ev_fork);
        if (c->ws.ev_join) (void)hipEventDestroy(c->ws.ev_join);
        c->ws.ev_fork = c->ws.ev_join = nullptr;
        c->side_allowed = false;
        (void)hipGetLastError();
    }
    *out = c;
    return MLKEM_OK;
}

void mlkem_ctx_destroy(mlkem_ctx* ctx) {
    if (!ctx) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(ctx->device);
    if (ctx->own_side) {
        (void)hipStreamSynchronize(ctx->own_side);
        (void)hipStreamDestroy(ctx->own_side);
    }
    if (ctx->ws.ev_fork) (void)hipEventDestroy(ctx->ws.ev_fork);
    if (ctx->ws.ev_join) (void)hipEventDestroy(ctx->ws.ev_join);
    if (ctx->scratch) {
        (void)hipMemset(ctx->scratch, 0, ctx->scratch_bytes);   // r, m', K', K-bar, PRF output: secret-dependent intermediates
        (void)hipFree(ctx->scratch);
    }
    delete ctx;
    if (prev >= 0) (void)hipSetDevice(prev);
}

size_t mlkem_ctx_scratch_bytes(const mlkem_ctx* ctx) { return ctx ? ctx->scratch_bytes : 0; }

// ---- per-kernel timing (measurement aid; HIP events on the launch stream) ----------------------------------
int mlkem_timing_begin(void) {
    if (launch_recorder()) return MLKEM_ERR_ARG;
    launch_recorder() = new LaunchRecorder();
    return MLKEM_OK;
}
// Synchronises, then writes up to `max` rows "label\0" (32 bytes each), total milliseconds and launch counts,
// aggregated per kernel label; returns the number of rows (or a negative error) and uninstalls the recorder.
int mlkem_timing_end(char* labels, double* total_ms, int* counts, int max) {
    LaunchRecorder* rec = launch_recorder();
    if (!rec) return MLKEM_ERR_ARG;
    launch_recorder() = nullptr;
    const bool failed = rec->failed;
    int rows = 0;
    for (auto& r : rec->recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess) (void)hipEventElapsedTime(&ms, r.a, r.b);
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
        int k = 0;
        for (; k < rows; k++)
            if (strncmp(labels + 32 * k, r.label, 31) == 0) break;
        if (k == rows) {
            if (rows >= max) continue;
            memset(labels + 32 * k, 0, 32);
            strncpy(labels + 32 * k, r.label, 31);
            total_ms[k] = 0;
            counts[k] = 0;
            rows++;
        }
        total_ms[k] += ms;
        counts[k] += 1;
    }
    delete rec;
    return failed ? MLKEM_ERR_NO_DEVICE : rows;   // incomplete rows are worse than none
}

// ---- device-pointer KEM ------------------------------------------------------------------------------

int mlkem_keygen_dev(mlkem_ctx* ctx, int set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk, void* stream) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!ctx_ok(ctx) || (n && (!d || !z || !ek || !dk))) return MLKEM_ERR_ARG;
    if (!aligned16(d) || !aligned16(z) || !aligned16(ek) || !aligned16(dk)) return MLKEM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    keygen_dispatch(st, set, n, d, z, ek, dk, ctx->ws);   // chunks internally through ctx->ws
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}

// ---- shared-key batches ------------------------------------------------------------------------------------------
int mlkem_encaps_shared_dev(mlkem_ctx* ctx, int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K, void* stream) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!ctx_ok(ctx) || (n && (!ek || !m || !c || !K))) return MLKEM_ERR_ARG;
    if (!aligned16(ek) || !aligned16(m) || !aligned16(c) || !aligned16(K)) return MLKEM_ERR_ARG;
    encaps_shared_dispatch(static_cast<hipStream_t>(stream), set, n, ek, m, c, K, ctx->ws);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_decaps_shared_dev(mlkem_ctx* ctx, int set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status,
                            void* stream) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!ctx_ok(ctx) || (n && (!dk || !c || !K))) return MLKEM_ERR_ARG;
    if (!aligned16(dk) || !aligned16(c) || !aligned16(K) || (reinterpret_cast<uintptr_t>(status) & 3u)) return MLKEM_ERR_ARG;
    decaps_shared_dispatch(static_cast<hipStream_t>(stream), set, n, dk, c, K, status, ctx->ws);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}

// ---- K-PKE on its own (ml_kem.c:651 / :776 / :942; static in the reference, reachable there through KeyGen_internal /
// Encaps_internal / Decaps_internal and through the oracle's harness) ------------------------------------------------
int mlkem_pke_keygen_dev(mlkem_ctx* ctx, int set, size_t n, const uint8_t* d, uint8_t* ek, uint8_t* dk_pke, void* stream) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!ctx_ok(ctx) || (n && (!d || !ek || !dk_pke))) return MLKEM_ERR_ARG;
    if (!aligned16(d) || !aligned16(ek) || !aligned16(dk_pke)) return MLKEM_ERR_ARG;
    if (n) pke_keygen_dispatch(static_cast<hipStream_t>(stream), set, n, d, ek, dk_pke, ctx->ws);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_pke_encrypt_dev(mlkem_ctx* ctx, int set, size_t n, const uint8_t* ek, const uint8_t* m, const uint8_t* r, uint8_t* c,
                          void* stream) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!ctx_ok(ctx) || (n && (!ek || !m || !r || !c))) return MLKEM_ERR_ARG;
    if (!aligned16(ek) || !aligned16(m) || !aligned16(r) || !aligned16(c)) return MLKEM_ERR_ARG;
    if (n) pke_encrypt_dispatch(static_cast<hipStream_t>(stream), set, n, ek, m, r, c, ctx->ws);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_pke_decrypt_dev(mlkem_ctx* ctx, int set, size_t n, const uint8_t* dk_pke, const uint8_t* c, uint8_t* m, void* stream) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!ctx_ok(ctx) || (n && (!dk_pke || !c || !m))) return MLKEM_ERR_ARG;
    if (!aligned16(dk_pke) || !aligned16(c) || !aligned16(m)) return MLKEM_ERR_ARG;
    pke_decrypt_dispatch(static_cast<hipStream_t>(stream), set, n, dk_pke, c, m);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}

int mlkem_encaps_status_dev(mlkem_ctx* ctx, int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K,
                            int32_t* status, void* stream) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!ctx_ok(ctx) || (n && (!ek || !m || !c || !K))) return MLKEM_ERR_ARG;
    if (!aligned16(ek) || !aligned16(m) || !aligned16(c) || !aligned16(K) || (reinterpret_cast<uintptr_t>(status) & 3u)) return MLKEM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (status && !ctx->ws.fips) {   // reference mode: the reference's modulus check can never fail (ml_kem.c:1273-1291, F3)
        HIP_TRY(hipMemsetAsync(status, 0, n * sizeof(int32_t), st));
        status = nullptr;
    }
    ctx_arm_side(ctx, n);
    encaps_dispatch(st, set, n, ek, m, c, K, status, ctx->ws);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_encaps_dev(mlkem_ctx* ctx, int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K, void* stream) {
    return mlkem_encaps_status_dev(ctx, set, n, ek, m, c, K, nullptr, stream);
}

int mlkem_ctx_set_conformance(mlkem_ctx* ctx, int mode) {
    if (!ctx_ok(ctx) || (mode != MLKEM_CONFORMANCE_REFERENCE && mode != MLKEM_CONFORMANCE_FIPS203)) return MLKEM_ERR_ARG;
    ctx->ws.fips = mode == MLKEM_CONFORMANCE_FIPS203;
    return MLKEM_OK;
}

int mlkem_ctx_debug_stages(mlkem_ctx* ctx, unsigned mask) {
    if (!ctx_ok(ctx) || mask > 15u) return MLKEM_ERR_ARG;
    ctx->ws.stages = mask;
    return MLKEM_OK;
}

int mlkem_decaps_dev(mlkem_ctx* ctx, int set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status, void* stream) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!ctx_ok(ctx) || (n && (!dk || !c || !K))) return MLKEM_ERR_ARG;
    if (!aligned16(dk) || !aligned16(c) || !aligned16(K) || (reinterpret_cast<uintptr_t>(status) & 3u)) return MLKEM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ctx_arm_side(ctx, n);
    decaps_dispatch(st, set, n, dk, c, K, status, status != nullptr, ctx->ws);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}

// ---- device-pointer primitives -------------------------------------------------------------------------

int mlkem_ntt_dev(mlkem_ctx* ctx, size_t n, const uint16_t* f, uint16_t* fh, void* stream) {
    if (!ctx_ok(ctx) || (n && (!f || !fh)) || !aligned16(f) || !aligned16(fh)) return MLKEM_ERR_ARG;
    ntt_launch(static_cast<hipStream_t>(stream), false, n, f, fh);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_intt_dev(mlkem_ctx* ctx, size_t n, const uint16_t* fh, uint16_t* f, void* stream) {
    if (!ctx_ok(ctx) || (n && (!f || !fh)) || !aligned16(f) || !aligned16(fh)) return MLKEM_ERR_ARG;
    ntt_launch(static_cast<hipStream_t>(stream), true, n, fh, f);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_multiply_ntts_dev(mlkem_ctx* ctx, size_t n, const uint16_t* a, const uint16_t* b, uint16_t* h, void* stream) {
    if (!ctx_ok(ctx) || (n && (!a || !b || !h)) || !aligned16(a) || !aligned16(b) || !aligned16(h)) return MLKEM_ERR_ARG;
    basemul_launch(static_cast<hipStream_t>(stream), n, a, b, h);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_vector_multiply_dev(mlkem_ctx* ctx, int k, size_t n, const uint16_t* u, const uint16_t* v, uint16_t* w, void* stream) {
    if (!ctx_ok(ctx) || k < 1 || k > 4 || (n && (!u || !v || !w)) || !aligned16(u) || !aligned16(v) || !aligned16(w)) return MLKEM_ERR_ARG;
    if (n) vecmul_launch(static_cast<hipStream_t>(stream), k, n, u, v, w);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_poly_add_dev(mlkem_ctx* ctx, size_t n_values, const uint16_t* u, const uint16_t* v, uint16_t* z, void* stream) {
    if (!ctx_ok(ctx) || (n_values && (!u || !v || !z)) || !aligned16(u) || !aligned16(v) || !aligned16(z)) return MLKEM_ERR_ARG;
    if (n_values) poly_addsub_launch(static_cast<hipStream_t>(stream), false, n_values, u, v, z);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_poly_sub_dev(mlkem_ctx* ctx, size_t n_values, const uint16_t* u, const uint16_t* v, uint16_t* z, void* stream) {
    if (!ctx_ok(ctx) || (n_values && (!u || !v || !z)) || !aligned16(u) || !aligned16(v) || !aligned16(z)) return MLKEM_ERR_ARG;
    if (n_values) poly_addsub_launch(static_cast<hipStream_t>(stream), true, n_values, u, v, z);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_sample_ntt_retries_dev(mlkem_ctx* ctx, size_t n, const uint8_t* seeds34, uint16_t* a, uint8_t* retries, void* stream) {
    if (!ctx_ok(ctx) || (n && (!seeds34 || !a)) || !aligned16(a)) return MLKEM_ERR_ARG;
    if (n) sample_ntt_launch(static_cast<hipStream_t>(stream), n, seeds34, a, ctx->ws.wide_max, retries);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_sample_ntt_dev(mlkem_ctx* ctx, size_t n, const uint8_t* seeds34, uint16_t* a, void* stream) {
    return mlkem_sample_ntt_retries_dev(ctx, n, seeds34, a, nullptr, stream);
}
int mlkem_sample_cbd_dev(mlkem_ctx* ctx, int eta, size_t n, const uint8_t* bytes, uint16_t* f, void* stream) {
    if (!ctx_ok(ctx) || (n && (!bytes || !f)) || !aligned16(bytes) || !aligned16(f)) return MLKEM_ERR_ARG;
    if (cbd_launch(static_cast<hipStream_t>(stream), eta, n, bytes, f)) return MLKEM_ERR_ARG;
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_compress_encode_dev(mlkem_ctx* ctx, int d, size_t n, const uint16_t* f, uint8_t* bytes, void* stream) {
    if (!ctx_ok(ctx) || (n && (!f || !bytes)) || !aligned16(f) || (reinterpret_cast<uintptr_t>(bytes) & 3u)) return MLKEM_ERR_ARG;
    if (d != 1 && d != 4 && d != 5 && d != 10 && d != 11 && d != 12) return MLKEM_ERR_ARG;
    if (n) codec_launch(static_cast<hipStream_t>(stream), true, d, n, f, bytes);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_decode_decompress_dev(mlkem_ctx* ctx, int d, size_t n, const uint8_t* bytes, uint16_t* f, void* stream) {
    if (!ctx_ok(ctx) || (n && (!f || !bytes)) || !aligned16(f) || (reinterpret_cast<uintptr_t>(bytes) & 3u)) return MLKEM_ERR_ARG;
    if (d != 1 && d != 4 && d != 5 && d != 10 && d != 11 && d != 12) return MLKEM_ERR_ARG;
    if (n) codec_launch(static_cast<hipStream_t>(stream), false, d, n, bytes, f);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_compress_dev(mlkem_ctx* ctx, int d, size_t n, const uint16_t* x, uint16_t* y, void* stream) {
    if (!ctx_ok(ctx) || (n && (!x || !y)) || !aligned16(x) || !aligned16(y)) return MLKEM_ERR_ARG;
    if (compress_values_launch(static_cast<hipStream_t>(stream), false, d, n, x, y)) return MLKEM_ERR_ARG;
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_decompress_dev(mlkem_ctx* ctx, int d, size_t n, const uint16_t* y, uint16_t* x, void* stream) {
    if (!ctx_ok(ctx) || (n && (!x || !y)) || !aligned16(x) || !aligned16(y)) return MLKEM_ERR_ARG;
    if (compress_values_launch(static_cast<hipStream_t>(stream), true, d, n, y, x)) return MLKEM_ERR_ARG;
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
// On-device exhaustive self-test of the exact fp32-pipe arithmetic (mlkem_selftest.hpp).  Runs sweep `which`
// (0 .. mlkem_selftest_count()-1) on the context's device, synchronises, and stores the number of violations.
int mlkem_selftest_count(void) { return SELFTEST_SWEEPS; }
int mlkem_selftest(mlkem_ctx* ctx, int which, unsigned long long* violations) {
    if (!ctx_ok(ctx) || !violations || which < 0 || which >= SELFTEST_SWEEPS) return MLKEM_ERR_ARG;
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof *d));
    int rc = MLKEM_OK;
    if (!hip_ok(hipMemset(d, 0, sizeof *d), "hipMemset")) rc = MLKEM_ERR_NO_DEVICE;
    if (rc == MLKEM_OK) {
        k_selftest<<<dim3(256 * 8), dim3(256), 0, nullptr>>>(which, d);
        if (!hip_ok(hipGetLastError(), "k_selftest") || !hip_ok(hipMemcpy(violations, d, sizeof *d, hipMemcpyDeviceToHost), "hipMemcpy"))
            rc = MLKEM_ERR_NO_DEVICE;
    }
    (void)hipFree(d);
    return rc;
}
int mlkem_prf_dev(mlkem_ctx* ctx, int eta, size_t n, const uint8_t* in33, uint8_t* out, void* stream) {
    if (!ctx_ok(ctx) || (n && (!in33 || !out)) || !aligned16(out)) return MLKEM_ERR_ARG;
    if (n && prf_launch(static_cast<hipStream_t>(stream), eta, n, in33, out)) return MLKEM_ERR_ARG;
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_hash_dev(mlkem_ctx* ctx, int kind, size_t n, const uint8_t* msg, unsigned len, size_t stride, uint8_t* out, void* stream) {
    if (!ctx_ok(ctx) || (n && (!msg || !out)) || !aligned16(msg) || !aligned16(out)) return MLKEM_ERR_ARG;
    if (n && hash_launch(static_cast<hipStream_t>(stream), kind, n, msg, len, stride, out)) return MLKEM_ERR_ARG;
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_keccak_sponge_dev(mlkem_ctx* ctx, unsigned rate, size_t n, const uint8_t* padded, unsigned nblocks, uint8_t* out,
                            unsigned outlen, size_t out_stride, void* stream) {
    if (!ctx_ok(ctx) || (n && (!padded || !out)) || !aligned16(padded) || !aligned16(out)) return MLKEM_ERR_ARG;
    if (n && sponge_raw_launch(static_cast<hipStream_t>(stream), rate, n, padded, nblocks, out, outlen, out_stride, ctx->ws.wide_max)) return MLKEM_ERR_ARG;
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}

int mlkem_cells_to_bytes_dev(mlkem_ctx* ctx, size_t n, const uint32_t* cells, uint8_t* bytes, void* stream) {
    if (!ctx_ok(ctx) || (n && (!cells || !bytes)) || !aligned16(cells) || !aligned16(bytes)) return MLKEM_ERR_ARG;
    if (n) cells_launch(static_cast<hipStream_t>(stream), true, n, cells, bytes);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}
int mlkem_bytes_to_cells_dev(mlkem_ctx* ctx, size_t n, const uint8_t* bytes, uint32_t* cells, void* stream) {
    if (!ctx_ok(ctx) || (n && (!cells || !bytes)) || !aligned16(cells) || !aligned16(bytes)) return MLKEM_ERR_ARG;
    if (n) cells_launch(static_cast<hipStream_t>(stream), false, n, bytes, cells);
    HIP_TRY(hipGetLastError());
    return MLKEM_OK;
}

// Pure host helpers (no device work): message bits (one per byte) + SHA-3 suffix + pad10*1 -> whole rate blocks.
// sha3.c:408-436 (the caller's suffix bits appended verbatim: two or four of them there) and :226-240 (pad), without the
// reference's latent bug for (n + suffix + 2) == 0 mod r.  Returns the number of blocks, or a negative error.
int mlkem_sha3_pad_suffix(const uint8_t* msg_bits, size_t nbits, const uint8_t* sfx_bits, unsigned nsfx, unsigned rate,
                          uint8_t* padded, size_t padded_cap) {
    if (rate == 0 || rate >= 200 || nsfx > 8 || (nsfx && !sfx_bits)) return MLKEM_ERR_ARG;   // any byte rate: capacity 8 .. 1592 bits
    const size_t sbits = nsfx, rbits = (size_t)rate * 8;
    const size_t nblocks = (nbits + sbits + 2 + rbits - 1) / rbits;
    if (nblocks * rate > padded_cap || !padded || (nbits && !msg_bits)) return MLKEM_ERR_ARG;
    memset(padded, 0, nblocks * rate);
    auto setbit = [&](size_t pos) { padded[pos >> 3] |= (uint8_t)(1u << (pos & 7)); };
    for (size_t i = 0; i < nbits; i++)
        if (msg_bits[i] & 1) setbit(i);
    for (size_t i = 0; i < sbits; i++)
        if (sfx_bits[i] & 1) setbit(nbits + i);
    setbit(nbits + sbits);               // first pad bit
    setbit(nblocks * rbits - 1);         // last pad bit
    return (int)nblocks;
}
int mlkem_sha3_pad_bits(const uint8_t* msg_bits, size_t nbits, int xof, unsigned rate, uint8_t* padded, size_t padded_cap) {
    static const uint8_t hash_sfx[2] = {0, 1}, xof_sfx[4] = {1, 1, 1, 1};
    return mlkem_sha3_pad_suffix(msg_bits, nbits, xof ? xof_sfx : hash_sfx, xof ? 4u : 2u, rate, padded, padded_cap);
}

}   // extern "C"

// ---- per-device state of the host-pointer entry points -------------------------------------------------------------
// Everything the host-pointer calls keep between calls is keyed by the HIP device that is current when the call is made:
// a small context for the primitives (NTT, SampleNTT, SamplePolyCBD, sponge: they use no scratch) and a streaming engine
// (streams, events, device + pinned staging buffers, a context sized for one chunk).  Two host threads working on two
// devices therefore share nothing and take different locks.  mlkem_host_release() wipes and frees all of it.
namespace {

// makes `dev` current for the duration of a call and restores the caller's device afterwards
struct DeviceGuard {
    int orig = -1;
    bool enter(int dev) {   // may be called repeatedly; the device current at the first call is restored at the end
        if (orig < 0 && hipGetDevice(&orig) != hipSuccess) orig = -1;
        return hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (orig >= 0) (void)hipSetDevice(orig); }
};

struct DevBuf {   // device staging of a host-pointer primitive call; zeroed before it is freed
    void* p = nullptr;
    size_t size = 0;
    ~DevBuf() {
        if (!p) return;
        (void)hipMemset(p, 0, size);
        (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        size = bytes ? bytes : 16;
        return hip_ok(hipMalloc(&p, size), "hipMalloc") ? MLKEM_OK : MLKEM_ERR_ALLOC;
    }
    template <class T> T* as() { return static_cast<T*>(p); }
};

// ---- streaming engine: host-resident batches of any size ----------------------------------------
// Three in-order streams (H2D copies / kernels / D2H copies) and NSETS sets of device buffers ordered by events:
//   kernels(i) wait for H2D(i) and for D2H(i - NSETS) (output buffers free); D2H(i) waits for kernels(i);
//   H2D(i) waits for kernels(i - NSETS) (input buffers free).
// PCIe is full duplex, so in steady state H2D(i+1), kernels(i) and D2H(i-1) overlap.  A caller buffer that is pinned
// (hipHostMalloc / hipHostRegister / mlkem_host_register) is used by the DMA engines directly and the host never blocks
// between chunks; a pageable buffer goes through the set's pinned staging buffer with threaded memcpy.
struct Span { const void* in; void* out; size_t bytes; };   // per-item bytes; exactly one of in/out is set

constexpr int NSETS = 3;

void par_memcpy(void* dst, const void* src, size_t bytes) {
    constexpr size_t MIN_PER_THREAD = (size_t)2 << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = bytes / MIN_PER_THREAD;
    if (nt > 8) nt = 8;
    if (hw && nt > hw) nt = hw;
    if (nt <= 1) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t per = ((bytes + nt - 1) / nt + 63) & ~(size_t)63;   // slice t covers [t*per, min((t+1)*per, bytes))
    std::vector<std::thread> th;
    for (size_t t = 1; t * per < bytes; t++) {
        const size_t off = t * per, len = bytes - off < per ? bytes - off : per;
        th.emplace_back([=]() { memcpy(static_cast<uint8_t*>(dst) + off, static_cast<const uint8_t*>(src) + off, len); });
    }
    memcpy(dst, src, per < bytes ? per : bytes);
    for (auto& x : th) x.join();
}

// is [p, p + bytes) host memory the DMA engines can address directly (pinned or registered)?  The first AND the last byte
// must be pinned host memory, and where the runtime reports the allocation's range the whole span must lie inside it: a
// buffer that is only partly registered is staged like a pageable one.
bool host_pinned_byte(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // pageable memory: "invalid value" on this runtime; not an error of the call
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
bool host_pinned(const void* p, size_t bytes) {
    if (!p || bytes == 0) return false;
    const uint8_t* b = static_cast<const uint8_t*>(p);
    if (!host_pinned_byte(b) || !host_pinned_byte(b + bytes - 1)) return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void*>(p)) == hipSuccess) {
        // the range decides only when it is a host range around p: for hipHostRegister'ed memory the runtime may report no
        // base, a device-side alias or a size of its own; then the two end bytes decide
        const uint8_t* lo = static_cast<const uint8_t*>(base);
        if (lo && size && b >= lo && b < lo + size) return b + bytes <= lo + size;
        return true;
    }
    (void)hipGetLastError();       // no range for this kind of pointer: the two end bytes decide
    return true;
}
// bit j set: span j of the calling thread's last streaming call went through the engine's pinned staging buffer (pageable or
// partly registered caller memory); -1 before the first call.  Lets a test see a silent demotion of a registered buffer.
thread_local int t_last_staged = -1;

struct StageBuf {
    void* dev = nullptr;
    void* pin = nullptr;   // allocated on first use by a pageable caller buffer
    size_t cap = 0;
};
struct BufSet {
    std::vector<StageBuf> buf;
    hipEvent_t evH = nullptr, evK = nullptr, evD = nullptr;
    size_t pending = 0, pending_off = 0;   // items whose outputs have not been handed to the caller yet
};
struct StreamEngine {
    std::mutex mu;
    int device = -1;
    hipStream_t h2d = nullptr, k = nullptr, d2h = nullptr;
    mlkem_ctx* ctx = nullptr;
    size_t ctx_chunk = 0;
    BufSet set[NSETS];
};

void stage_free(StageBuf& b) {   // staging buffers carry seeds / keys / shared secrets: zero, then free
    if (b.dev) { (void)hipMemset(b.dev, 0, b.cap); (void)hipFree(b.dev); b.dev = nullptr; }
    if (b.pin) { explicit_bzero(b.pin, b.cap); (void)hipHostFree(b.pin); b.pin = nullptr; }
    b.cap = 0;
}

void engine_release(StreamEngine& e) {   // caller holds e.mu (or owns e exclusively)
    if (e.device < 0) return;
    DeviceGuard g;
    (void)g.enter(e.device);
    for (hipStream_t s : {e.h2d, e.k, e.d2h})
        if (s) (void)hipStreamSynchronize(s);
    for (BufSet& s : e.set) {
        for (StageBuf& b : s.buf) stage_free(b);
        s.buf.clear();
        for (hipEvent_t* ev : {&s.evH, &s.evK, &s.evD})
            if (*ev) { (void)hipEventDestroy(*ev); *ev = nullptr; }
        s.pending = 0;
    }
    if (e.ctx) mlkem_ctx_destroy(e.ctx);
    e.ctx = nullptr;
    e.ctx_chunk = 0;
    for (hipStream_t* s : {&e.h2d, &e.k, &e.d2h})
        if (*s) { (void)hipStreamDestroy(*s); *s = nullptr; }
    e.device = -1;
}

int engine_prepare(StreamEngine& e, int dev, size_t chunk, const std::vector<Span>& spans, const std::vector<char>& staged, int nsets) {
    if (e.device >= 0 && e.device != dev) engine_release(e);
    e.device = dev;
    for (hipStream_t* s : {&e.h2d, &e.k, &e.d2h})
        if (!*s && !hip_ok(hipStreamCreateWithFlags(s, hipStreamNonBlocking), "hipStreamCreate")) return MLKEM_ERR_NO_DEVICE;
    if (!e.ctx || e.ctx_chunk < chunk) {   // a context serves any batch size; its chunk capacity only grows
        if (e.ctx) {
            (void)hipStreamSynchronize(e.k);
            mlkem_ctx_destroy(e.ctx);
        }
        e.ctx = nullptr;
        int rc = mlkem_ctx_create(&e.ctx, dev, chunk);
        if (rc) return rc;
        e.ctx_chunk = chunk;
    }
    for (int k = 0; k < nsets; k++) {
        BufSet& s = e.set[k];
        for (hipEvent_t* ev : {&s.evH, &s.evK, &s.evD})
            if (!*ev && !hip_ok(hipEventCreateWithFlags(ev, hipEventDisableTiming), "hipEventCreate")) return MLKEM_ERR_NO_DEVICE;
        if (s.buf.size() < spans.size()) s.buf.resize(spans.size());
        for (size_t j = 0; j < spans.size(); j++) {
            StageBuf& b = s.buf[j];
            const size_t need = chunk * spans[j].bytes;
            if (b.cap < need) {
                for (hipStream_t st : {e.h2d, e.k, e.d2h}) (void)hipStreamSynchronize(st);
                stage_free(b);
                if (!hip_ok(hipMalloc(&b.dev, need), "hipMalloc")) return MLKEM_ERR_ALLOC;
                b.cap = need;
            }
            if (staged[j] && !b.pin && !hip_ok(hipHostMalloc(&b.pin, b.cap, hipHostMallocDefault), "hipHostMalloc")) return MLKEM_ERR_ALLOC;
        }
        s.pending = 0;
    }
    return MLKEM_OK;
}

size_t default_stream_chunk() {
    if (const char* e = getenv("MLKEM_STREAM_CHUNK_ITEMS")) {
        long long v = atoll(e);
        if (v > 0) return (size_t)v;
    }
    return (size_t)1 << 15;   // measured best on MI355X for pageable and pinned buffers (tools/stream_bench.py, profiles/r02_stream_bench.json)
}

// runs on the calling thread; the engine's device must be current
// pre_locked: the caller already holds e.mu (kem_stream_current picks a free engine by try_lock)
template <class Launch>
int stream_op(StreamEngine& e, size_t n, size_t chunk, const std::vector<Span>& spans, Launch launch, bool pre_locked = false) {
    if (n == 0) return MLKEM_OK;
    if (chunk == 0) chunk = default_stream_chunk();
    if (chunk > n) chunk = n;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MLKEM_ERR_NO_DEVICE;
    std::unique_lock<std::mutex> lock(e.mu, std::defer_lock);
    if (!pre_locked) lock.lock();
    const size_t nchunks = (n + chunk - 1) / chunk;
    const int nsets = nchunks < (size_t)NSETS ? (int)nchunks : NSETS;
    std::vector<char> staged(spans.size());
    // a tiny call (a few KB per operand: the ml_kem.h shim's one-item calls) copies its operands through the pinned staging
    // buffers without asking the runtime what kind of memory the caller handed in: eight pointer-attribute queries cost more
    // than the memcpy of 3 KB
    size_t widest = 0;
    for (const Span& sp : spans) widest = sp.bytes > widest ? sp.bytes : widest;
    const bool tiny = n * widest <= 16384;
    for (size_t j = 0; j < spans.size(); j++)
        staged[j] = tiny || !host_pinned(spans[j].in ? spans[j].in : spans[j].out, n * spans[j].bytes);
    t_last_staged = 0;
    for (size_t j = 0; j < spans.size() && j < 31; j++) t_last_staged |= staged[j] ? (1 << j) : 0;
    int rc = engine_prepare(e, dev, chunk, spans, staged, nsets);
    auto drain = [&](BufSet& s) -> int {   // wait for the set's D2H copies and hand staged outputs to the caller
        if (!s.pending) return MLKEM_OK;
        if (!hip_ok(hipEventSynchronize(s.evD), "hipEventSynchronize")) return MLKEM_ERR_NO_DEVICE;
        for (size_t j = 0; j < spans.size(); j++)
            if (spans[j].out && staged[j])
                par_memcpy(static_cast<uint8_t*>(spans[j].out) + s.pending_off * spans[j].bytes, s.buf[j].pin, s.pending * spans[j].bytes);
        s.pending = 0;
        return MLKEM_OK;
    };
    bool any_staged = false;
    for (char c : staged) any_staged = any_staged || c;
    std::vector<void*> devp(spans.size());
    // One chunk: nothing to overlap, so copy-in, kernels and copy-out go down the kernel stream in order and the call saves the
    // two cross-stream event hops (host-pointer Encaps + Decaps of <= 64 items: 431 -> see profiles/r03_host_latency.txt).
    // Every call ends with its streams drained, so the choice is per call.
    const bool single = nchunks == 1;
    const hipStream_t sh = single ? e.k : e.h2d, sd = single ? e.k : e.d2h;
    if (rc == MLKEM_OK && e.ctx) {
        // side-stream sampling only for a stand-alone chunk, and then on the front-end's own idle copy stream
        e.ctx->ws.side_on = single && e.ctx->side_allowed;
        if (e.ctx->ws.side_on) e.ctx->ws.side = e.h2d;
    }
    // Small call (what the ml_kem.h shim makes: one item): no copy commands at all.  The kernels of mlkem_small.hpp read their
    // inputs from, and write their outputs to, PINNED HOST memory over PCIe -- the caller's own buffers where they are pinned or
    // registered (and 16-byte aligned), the set's pinned staging buffer otherwise -- so that the call is: memcpy in, ONE launch,
    // stream synchronise, memcpy out.  (Per operation and item the kernels read 1.2-3.5 KB and write 0.03-1.1 KB; four copy
    // commands and an event cost more than the PCIe round trips inside the kernel: profiles/r04_host_latency.txt.)
    // MLKEM_ZERO_COPY=0 keeps the copy commands.  Measured up to 512 items; larger small calls keep the copy commands.
    constexpr size_t ZERO_COPY_MAX = 512;
    static const bool zero_copy = [] { const char* z = getenv("MLKEM_ZERO_COPY"); return !(z && atoi(z) == 0); }();
    if (rc == MLKEM_OK && single && zero_copy && e.ctx && n <= e.ctx->ws.small_max(4) && n <= ZERO_COPY_MAX) {   // (k = 4 has the lowest limit, above ZERO_COPY_MAX by default)
        BufSet& s = e.set[0];
        bool ok = true;
        for (size_t j = 0; j < spans.size() && ok; j++) {
            void* host = spans[j].in ? const_cast<void*>(spans[j].in) : spans[j].out;
            const bool direct = !staged[j] && (reinterpret_cast<uintptr_t>(host) & 15u) == 0;
            if (!direct && !s.buf[j].pin) ok = hip_ok(hipHostMalloc(&s.buf[j].pin, s.buf[j].cap, hipHostMallocDefault), "hipHostMalloc");
            if (!ok) break;
            void* h = direct ? host : s.buf[j].pin;
            if (!direct && spans[j].in) memcpy(h, spans[j].in, n * spans[j].bytes);
            void* d = nullptr;
            ok = hip_ok(hipHostGetDevicePointer(&d, h, 0), "hipHostGetDevicePointer") && d;
            devp[j] = d;
            staged[j] = !direct;
        }
        if (ok) {
            rc = launch(e.ctx, n, devp, e.k);
            if (rc == MLKEM_OK && !hip_ok(hipStreamSynchronize(e.k), "hipStreamSynchronize")) rc = MLKEM_ERR_NO_DEVICE;
            if (rc == MLKEM_OK)
                for (size_t j = 0; j < spans.size(); j++)
                    if (spans[j].out && staged[j]) memcpy(spans[j].out, s.buf[j].pin, n * spans[j].bytes);
            if (rc != MLKEM_OK) engine_release(e);
            return rc;
        }
        (void)hipGetLastError();   // no device view of a buffer: take the copy path below
        for (size_t j = 0; j < spans.size(); j++) staged[j] = !host_pinned(spans[j].in ? spans[j].in : spans[j].out, n * spans[j].bytes);
    }
    size_t i = 0;
    for (size_t off = 0; off < n && rc == MLKEM_OK; off += chunk, i++) {
        BufSet& s = e.set[i % nsets];
        const size_t cnt = n - off < chunk ? n - off : chunk;
        if (any_staged && (rc = drain(s)) != MLKEM_OK) break;        // frees the set's pinned buffers (and implies its H2D is done)
        if (!single) (void)hipStreamWaitEvent(sh, s.evK, 0);         // the set's previous kernels have consumed its inputs
        for (size_t j = 0; j < spans.size() && rc == MLKEM_OK; j++) {
            devp[j] = s.buf[j].dev;
            if (!spans[j].in) continue;
            const uint8_t* src = static_cast<const uint8_t*>(spans[j].in) + off * spans[j].bytes;
            if (staged[j]) {
                par_memcpy(s.buf[j].pin, src, cnt * spans[j].bytes);
                src = static_cast<const uint8_t*>(s.buf[j].pin);
            }
            if (!hip_ok(hipMemcpyAsync(s.buf[j].dev, src, cnt * spans[j].bytes, hipMemcpyHostToDevice, sh), "H2D")) rc = MLKEM_ERR_NO_DEVICE;
        }
        if (rc != MLKEM_OK) break;
        if (!single) {
            (void)hipEventRecord(s.evH, sh);
            (void)hipStreamWaitEvent(e.k, s.evH, 0);
            (void)hipStreamWaitEvent(e.k, s.evD, 0);                 // the set's previous outputs have left the device
        }
        rc = launch(e.ctx, cnt, devp, e.k);
        if (rc != MLKEM_OK) break;
        if (!single) {
            (void)hipEventRecord(s.evK, e.k);
            (void)hipStreamWaitEvent(sd, s.evK, 0);
        }
        for (size_t j = 0; j < spans.size() && rc == MLKEM_OK; j++) {
            if (!spans[j].out) continue;
            void* dst = staged[j] ? s.buf[j].pin : static_cast<void*>(static_cast<uint8_t*>(spans[j].out) + off * spans[j].bytes);
            if (!hip_ok(hipMemcpyAsync(dst, s.buf[j].dev, cnt * spans[j].bytes, hipMemcpyDeviceToHost, sd), "D2H")) rc = MLKEM_ERR_NO_DEVICE;
        }
        (void)hipEventRecord(s.evD, sd);
        s.pending = cnt;
        s.pending_off = off;
    }
    // hand over in issue order: the oldest pending set first
    for (int k = 0; k < nsets && rc == MLKEM_OK; k++) rc = drain(e.set[(i + k) % nsets]);
    if (rc == MLKEM_OK && !hip_ok(hipStreamSynchronize(sd), "hipStreamSynchronize")) rc = MLKEM_ERR_NO_DEVICE;
    if (rc != MLKEM_OK) engine_release(e);   // leave nothing half-done behind
    return rc;
}

// Ownership: the registry and every call in flight hold a shared_ptr.  mlkem_host_release() takes the entries out of the
// registry, wipes and frees their contents under the entry's own locks (so it waits for calls in flight) and drops its
// reference; a call that had already looked its entry up keeps the object alive, finds it empty, rebuilds what it needs in
// the orphaned entry, and the destructor wipes and frees that when the call returns.  Release is therefore safe to call
// from any thread at any time.
size_t host_lanes_max() {   // env MLKEM_HOST_LANES (0: one engine per device, calls of other threads queue)
    static const size_t v = [] {
        const char* e = getenv("MLKEM_HOST_LANES");
        long long x = e ? atoll(e) : 7;
        return (size_t)(x < 0 ? 0 : (x > 63 ? 63 : x));
    }();
    return v;
}
// Calls of a few items (the ml_kem.h shim makes calls of one) that arrive while other such calls of the same operation and
// parameter set are in flight are COMBINED: the first caller that finds fewer than host_combine_leaders() batches in flight becomes a
// leader, takes everything that has queued up, runs it as ONE call (one launch + one synchronise for the whole batch) and hands the
// results back; the others sleep until their request is done.  A lone caller is its own leader at once: no waiting is added.
// One launch + synchronise costs the host ~50 us whatever it carries, so a host with many threads is limited by calls per
// second, not by the GPU (profiles/r04_host_threads.txt).  MLKEM_HOST_COMBINE=0 turns it off.
struct CombineReq {
    const uint8_t *a, *b;
    uint8_t *x, *y;
    size_t n;
    int rc = MLKEM_OK;
    bool taken = false;      // a leader has it in its batch: the owner must wait for `done`, not lead
    bool done = false, combined = false;
};
struct Combiner {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<CombineReq*> waiting;
    int leaders = 0;
};
constexpr size_t COMBINE_MAX_ITEMS = 4;    // per request
constexpr size_t COMBINE_MAX_BATCH = 256;  // items per combined call: one eight-wave workgroup per CU
int host_combine_leaders() {   // batches in flight per operation and parameter set; env MLKEM_HOST_COMBINE (0: no combining)
    static const int v = [] {
        const char* e = getenv("MLKEM_HOST_COMBINE");
        const int x = e ? atoi(e) : 2;   // 2: best from 16 host threads up; 3-4 is ~10 % better at four (profiles/r04_host_threads.txt)
        return x < 0 ? 0 : (x > 16 ? 16 : x);
    }();
    return v;
}
bool host_combine_enabled() { return host_combine_leaders() > 0; }
struct HostState {
    std::mutex mu;              // serialises the host-pointer primitives of one device
    int device = 0;
    mlkem_ctx* ctx = nullptr;   // chunk capacity 1: the primitives use no scratch
    StreamEngine eng;
    // Further engines ("lanes") for host threads that call while `eng` is busy: every lane has its own streams, staging buffers
    // and context, so that one-item calls of a multi-threaded host (the ml_kem.h shim under a server) run side by side on the
    // GPU instead of queueing on one mutex.  Opened on demand, at most host_lanes_max(); the objects live as long as the entry.
    std::mutex lanes_mu;
    std::vector<std::unique_ptr<StreamEngine>> lanes;
    Combiner comb[9];           // [operation][parameter set]
    void release_engines() {
        {
            std::lock_guard<std::mutex> l2(eng.mu);
            engine_release(eng);
        }
        std::vector<StreamEngine*> all;
        {
            std::lock_guard<std::mutex> lg(lanes_mu);
            for (auto& l : lanes) all.push_back(l.get());
        }
        for (StreamEngine* l : all) {   // waits for a call in flight on that lane
            std::lock_guard<std::mutex> l2(l->mu);
            engine_release(*l);
        }
    }
    void wipe() {               // zero + free everything cached; the entry stays usable (contents are rebuilt on demand)
        release_engines();
        std::lock_guard<std::mutex> l3(mu);
        if (ctx) mlkem_ctx_destroy(ctx);
        ctx = nullptr;
    }
    ~HostState() { wipe(); }
};
using HostRef = std::shared_ptr<HostState>;
std::mutex g_reg_mu;
// leaked on purpose: a static map would run ~HostState (HIP calls) during static destruction, after the runtime is gone
std::map<int, HostRef>& g_host = *new std::map<int, HostRef>();

HostRef host_state_current(int* dev_out = nullptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev_out) *dev_out = dev;
    std::lock_guard<std::mutex> lock(g_reg_mu);
    HostRef& hs = g_host[dev];
    if (!hs) {
        hs.reset(new (std::nothrow) HostState());
        if (hs) hs->device = dev;
    }
    return hs;
}
std::vector<HostRef> host_state_snapshot(bool take) {   // take: the registry forgets them
    std::lock_guard<std::mutex> lock(g_reg_mu);
    std::vector<HostRef> v;
    for (auto& kv : g_host)
        if (kv.second) v.push_back(kv.second);
    if (take) g_host.clear();
    return v;
}
// caller holds hs->mu
int host_ctx(HostState* hs, mlkem_ctx** out) {
    if (!hs->ctx) {
        int rc = mlkem_ctx_create(&hs->ctx, hs->device, 1);
        if (rc) return rc;
    }
    *out = hs->ctx;
    return MLKEM_OK;
}

// No C++ exception may cross the C boundary: allocation failures inside the host paths (staging threads, vectors) come back
// as MLKEM_ERR_ALLOC.
template <class Fn>
int guarded(Fn fn) {
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        return MLKEM_ERR_ALLOC;
    } catch (...) {   // std::system_error from std::thread / std::mutex: resources exhausted
        return MLKEM_ERR_ALLOC;
    }
}

// the three KEM operations as (spans, launch) pairs for stream_op
int kem_stream(StreamEngine& e, int op, int set, size_t n, const void* a, const void* b, void* x, void* y, size_t chunk, bool pre_locked = false) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (n && (!a || !b || !x || !y)) return MLKEM_ERR_ARG;
    if (op == 0) {   // keygen: d, z -> ek, dk
        std::vector<Span> sp = {{a, nullptr, 32}, {b, nullptr, 32}, {nullptr, x, p.ek_len}, {nullptr, y, p.dk_len}};
        return stream_op(e, n, chunk, sp, [&](mlkem_ctx* ctx, size_t cnt, const std::vector<void*>& v, hipStream_t st) {
            return mlkem_keygen_dev(ctx, set, cnt, (const uint8_t*)v[0], (const uint8_t*)v[1], (uint8_t*)v[2], (uint8_t*)v[3], st);
        }, pre_locked);
    }
    if (op == 1) {   // encaps: ek, m -> c, K
        std::vector<Span> sp = {{a, nullptr, p.ek_len}, {b, nullptr, 32}, {nullptr, x, p.c_len}, {nullptr, y, 32}};
        return stream_op(e, n, chunk, sp, [&](mlkem_ctx* ctx, size_t cnt, const std::vector<void*>& v, hipStream_t st) {
            return mlkem_encaps_dev(ctx, set, cnt, (const uint8_t*)v[0], (const uint8_t*)v[1], (uint8_t*)v[2], (uint8_t*)v[3], st);
        }, pre_locked);
    }
    // decaps: dk, c -> K, status
    std::vector<Span> sp = {{a, nullptr, p.dk_len}, {b, nullptr, p.c_len}, {nullptr, x, 32}, {nullptr, y, 4}};
    return stream_op(e, n, chunk, sp, [&](mlkem_ctx* ctx, size_t cnt, const std::vector<void*>& v, hipStream_t st) {
        return mlkem_decaps_dev(ctx, set, cnt, (const uint8_t*)v[0], (const uint8_t*)v[1], (uint8_t*)v[2], (int32_t*)v[3], st);
    }, pre_locked);
}

int kem_on_free_engine(HostState* hs, int op, int set, size_t n, const void* a, const void* b, void* x, void* y, size_t chunk);

// see Combiner
int kem_combined(HostState* hs, int op, int set, size_t n, const void* a, const void* b, void* x, void* y) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!a || !b || !x || !y) return MLKEM_ERR_ARG;
    const size_t sz[3][4] = {{32, 32, p.ek_len, p.dk_len}, {p.ek_len, 32, p.c_len, 32}, {p.dk_len, p.c_len, 32, 4}};
    const size_t* w = sz[op];
    Combiner& cb = hs->comb[op * 3 + (set == 512 ? 0 : set == 768 ? 1 : 2)];
    CombineReq me{static_cast<const uint8_t*>(a), static_cast<const uint8_t*>(b), static_cast<uint8_t*>(x), static_cast<uint8_t*>(y), n};
    std::unique_lock<std::mutex> lk(cb.mu);
    cb.waiting.push_back(&me);
    for (;;) {
        if (me.done) {
            if (me.combined) t_last_staged = 15;   // every operand went through the leader's batch buffers
            return me.rc;
        }
        if (!me.taken && cb.leaders < host_combine_leaders()) break;
        cb.cv.wait(lk);
    }
    // leader: my own request (still queued, not taken) and whatever else fits
    std::vector<CombineReq*> batch, rest;
    size_t total = me.n;
    try {
        batch.reserve(cb.waiting.size());
        rest.reserve(cb.waiting.size());
    } catch (...) {   // nothing has been taken yet: withdraw my request
        cb.waiting.erase(std::find(cb.waiting.begin(), cb.waiting.end(), &me));
        return MLKEM_ERR_ALLOC;
    }
    batch.push_back(&me);
    for (CombineReq* r : cb.waiting) {
        if (r == &me) continue;
        if (total + r->n <= COMBINE_MAX_BATCH) { batch.push_back(r); total += r->n; r->taken = true; }
        else rest.push_back(r);
    }
    cb.waiting.swap(rest);
    cb.leaders++;
    lk.unlock();
    int rc = MLKEM_OK;
    try {
        if (batch.size() == 1) {
            rc = kem_on_free_engine(hs, op, set, me.n, me.a, me.b, me.x, me.y, 0);
        } else {
            std::vector<uint8_t> A(total * w[0]), B(total * w[1]), X(total * w[2]), Y(total * w[3]);
            size_t off = 0;
            for (CombineReq* r : batch) {
                memcpy(A.data() + off * w[0], r->a, r->n * w[0]);
                memcpy(B.data() + off * w[1], r->b, r->n * w[1]);
                off += r->n;
            }
            rc = kem_on_free_engine(hs, op, set, total, A.data(), B.data(), X.data(), Y.data(), 0);
            off = 0;
            for (CombineReq* r : batch) {
                if (rc == MLKEM_OK) {
                    memcpy(r->x, X.data() + off * w[2], r->n * w[2]);
                    memcpy(r->y, Y.data() + off * w[3], r->n * w[3]);
                }
                off += r->n;
            }
            explicit_bzero(A.data(), A.size()); explicit_bzero(B.data(), B.size());   // seeds, keys, shared secrets
            explicit_bzero(X.data(), X.size()); explicit_bzero(Y.data(), Y.size());
        }
    } catch (...) {   // allocation or thread-resource failure: every request of the batch gets the error
        rc = MLKEM_ERR_ALLOC;
    }
    lk.lock();
    const bool combined = batch.size() > 1;
    for (CombineReq* r : batch) { r->rc = rc; r->combined = combined; r->done = true; }   // a waiter may return (its request gone) from here on
    cb.leaders--;
    cb.cv.notify_all();
    if (combined) t_last_staged = 15;
    return rc;
}

int kem_stream_current(int op, int set, size_t n, const void* a, const void* b, void* x, void* y, size_t chunk) {
    return guarded([&]() -> int {
        HostRef hs = host_state_current();
        if (!hs) return MLKEM_ERR_NO_DEVICE;
        if (n && n <= COMBINE_MAX_ITEMS && chunk == 0 && host_combine_enabled()) return kem_combined(hs.get(), op, set, n, a, b, x, y);
        return kem_on_free_engine(hs.get(), op, set, n, a, b, x, y, chunk);
    });
}

int kem_on_free_engine(HostState* hs, int op, int set, size_t n, const void* a, const void* b, void* x, void* y, size_t chunk) {
    {
        // the first free engine: the device's own, else a lane (opened on demand); all busy: queue on one of them, spread by thread
        StreamEngine* e = &hs->eng;
        std::unique_lock<std::mutex> held(e->mu, std::try_to_lock);
        if (!held.owns_lock()) {
            std::lock_guard<std::mutex> lg(hs->lanes_mu);
            for (auto& l : hs->lanes) {
                held = std::unique_lock<std::mutex>(l->mu, std::try_to_lock);
                if (held.owns_lock()) { e = l.get(); break; }
            }
            if (!held.owns_lock() && hs->lanes.size() < host_lanes_max()) {
                hs->lanes.emplace_back(new StreamEngine());
                e = hs->lanes.back().get();
                held = std::unique_lock<std::mutex>(e->mu);
            }
            if (!held.owns_lock()) {
                const size_t pick = std::hash<std::thread::id>()(std::this_thread::get_id()) % (hs->lanes.size() + 1);
                if (pick) e = hs->lanes[pick - 1].get();
            }
        }
        if (!held.owns_lock()) held = std::unique_lock<std::mutex>(e->mu);
        return kem_stream(*e, op, set, n, a, b, x, y, chunk, /*pre_locked=*/true);
    }
}

}   // namespace

extern "C" {

// The host-pointer KEM calls are the streaming front-end with its default chunking: the engine of the current device is
// cached between calls (no allocation per call after the first).
int mlkem_keygen(int set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk) {
    return mlkem_keygen_stream(set, n, d, z, ek, dk, 0);
}
int mlkem_encaps(int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K) {
    return mlkem_encaps_stream(set, n, ek, m, c, K, 0);
}
int mlkem_decaps(int set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status) {
    if (status || n == 0) return mlkem_decaps_stream(set, n, dk, c, K, status, 0);
    return guarded([&]() -> int {
        std::vector<int32_t> st(n);   // the caller did not ask for the hash-check codes; K does not depend on them
        return mlkem_decaps_stream(set, n, dk, c, K, st.data(), 0);
    });
}
int mlkem_keygen_stream(int set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk, size_t chunk_items) {
    return kem_stream_current(0, set, n, d, z, ek, dk, chunk_items);
}
int mlkem_encaps_stream(int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K, size_t chunk_items) {
    return kem_stream_current(1, set, n, ek, m, c, K, chunk_items);
}
int mlkem_decaps_stream(int set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status, size_t chunk_items) {
    return kem_stream_current(2, set, n, dk, c, K, status, chunk_items);
}

int mlkem_stream_last_staged(void) { return t_last_staged; }

// pins caller memory so that the streaming calls hand it to the DMA engines directly (no staging copy)
int mlkem_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return MLKEM_ERR_ARG;
    return hip_ok(hipHostRegister(p, bytes, hipHostRegisterPortable), "hipHostRegister") ? MLKEM_OK : MLKEM_ERR_ALLOC;
}
int mlkem_host_unregister(void* p) {
    if (!p) return MLKEM_ERR_ARG;
    return hip_ok(hipHostUnregister(p), "hipHostUnregister") ? MLKEM_OK : MLKEM_ERR_ARG;
}

void mlkem_stream_release(void) {
    for (HostRef& hs : host_state_snapshot(false)) hs->release_engines();
}
// wipes and frees everything the host-pointer entry points cached, on every device they were used on; safe against calls
// in flight on other threads (see HostState)
void mlkem_host_release(void) {
    for (HostRef& hs : host_state_snapshot(true)) hs->wipe();
}

// ---- host-pointer primitives ----------------------------------------------------------------------------------
#define MLKEM_HOST_PROLOGUE()                                   \
    HostRef hs = host_state_current();                          \
    if (!hs) return MLKEM_ERR_NO_DEVICE;                        \
    std::lock_guard<std::mutex> lock(hs->mu);                   \
    mlkem_ctx* ctx;                                             \
    int rc = host_ctx(hs.get(), &ctx);                          \
    if (rc) return rc;

static int host_ntt(bool inverse, size_t n, const uint16_t* in, uint16_t* out) {
    if (n && (!in || !out)) return MLKEM_ERR_ARG;
    MLKEM_HOST_PROLOGUE()
    if (n == 0) return MLKEM_OK;
    DevBuf bi, bo;
    if ((rc = bi.alloc(n * 512)) || (rc = bo.alloc(n * 512))) return rc;
    HIP_TRY(hipMemcpy(bi.p, in, n * 512, hipMemcpyHostToDevice));
    rc = inverse ? mlkem_intt_dev(ctx, n, bi.as<uint16_t>(), bo.as<uint16_t>(), nullptr)
                 : mlkem_ntt_dev(ctx, n, bi.as<uint16_t>(), bo.as<uint16_t>(), nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, bo.p, n * 512, hipMemcpyDeviceToHost));
    return MLKEM_OK;
}
int mlkem_ntt(size_t n, const uint16_t* f, uint16_t* fh) { return host_ntt(false, n, f, fh); }
int mlkem_intt(size_t n, const uint16_t* fh, uint16_t* f) { return host_ntt(true, n, fh, f); }

int mlkem_sample_ntt_retries(size_t n, const uint8_t* seeds34, uint16_t* a_hat, uint8_t* retries) {
    if (n && (!seeds34 || !a_hat)) return MLKEM_ERR_ARG;
    MLKEM_HOST_PROLOGUE()
    if (n == 0) return MLKEM_OK;
    DevBuf bi, bo, br;
    if ((rc = bi.alloc(n * 34)) || (rc = bo.alloc(n * 512)) || (retries && (rc = br.alloc(n)))) return rc;
    HIP_TRY(hipMemcpy(bi.p, seeds34, n * 34, hipMemcpyHostToDevice));
    if ((rc = mlkem_sample_ntt_retries_dev(ctx, n, bi.as<uint8_t>(), bo.as<uint16_t>(), retries ? br.as<uint8_t>() : nullptr, nullptr))) return rc;
    HIP_TRY(hipMemcpy(a_hat, bo.p, n * 512, hipMemcpyDeviceToHost));
    if (retries) HIP_TRY(hipMemcpy(retries, br.p, n, hipMemcpyDeviceToHost));
    return MLKEM_OK;
}
int mlkem_sample_ntt(size_t n, const uint8_t* seeds34, uint16_t* a_hat) { return mlkem_sample_ntt_retries(n, seeds34, a_hat, nullptr); }
int mlkem_sample_cbd(int eta, size_t n, const uint8_t* bytes, uint16_t* f) {
    if ((eta != 2 && eta != 3) || (n && (!bytes || !f))) return MLKEM_ERR_ARG;
    MLKEM_HOST_PROLOGUE()
    if (n == 0) return MLKEM_OK;
    DevBuf bi, bo;
    if ((rc = bi.alloc(n * 64 * (size_t)eta)) || (rc = bo.alloc(n * 512))) return rc;
    HIP_TRY(hipMemcpy(bi.p, bytes, n * 64 * (size_t)eta, hipMemcpyHostToDevice));
    if ((rc = mlkem_sample_cbd_dev(ctx, eta, n, bi.as<uint8_t>(), bo.as<uint16_t>(), nullptr))) return rc;
    HIP_TRY(hipMemcpy(f, bo.p, n * 512, hipMemcpyDeviceToHost));
    return MLKEM_OK;
}

static int host_keccak_sponge(unsigned rate, size_t n, const uint8_t* padded, unsigned nblocks, uint8_t* out, unsigned outlen) {
    if (n && (!padded || !out)) return MLKEM_ERR_ARG;
    MLKEM_HOST_PROLOGUE()
    if (n == 0) return MLKEM_OK;
    const size_t in_bytes = n * (size_t)nblocks * rate, ostride = ((size_t)outlen + 3) & ~(size_t)3;
    DevBuf bi, bo;
    if ((rc = bi.alloc(in_bytes)) || (rc = bo.alloc(n * ostride))) return rc;
    HIP_TRY(hipMemcpy(bi.p, padded, in_bytes, hipMemcpyHostToDevice));
    rc = mlkem_keccak_sponge_dev(ctx, rate, n, bi.as<uint8_t>(), nblocks, bo.as<uint8_t>(), outlen, ostride, nullptr);
    if (rc) return rc;
    std::vector<uint8_t> tmp(n * ostride);
    HIP_TRY(hipMemcpy(tmp.data(), bo.p, n * ostride, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) memcpy(out + i * outlen, tmp.data() + i * ostride, outlen);
    return MLKEM_OK;
}
int mlkem_keccak_sponge(unsigned rate, size_t n, const uint8_t* padded, unsigned nblocks, uint8_t* out, unsigned outlen) {
    return guarded([&]() -> int { return host_keccak_sponge(rate, n, padded, nblocks, out, outlen); });
}
// Compress_d / Decompress_d for any d in 1..12 (ml_kem.c:83-119; d = 12 is the identity there) over n values
int mlkem_compress(int d, size_t n, const uint16_t* x, uint16_t* y) {
    if (d < 1 || d > 12 || (n && (!x || !y))) return MLKEM_ERR_ARG;
    MLKEM_HOST_PROLOGUE()
    if (n == 0) return MLKEM_OK;
    DevBuf b;
    if ((rc = b.alloc(n * 2))) return rc;
    HIP_TRY(hipMemcpy(b.p, x, n * 2, hipMemcpyHostToDevice));
    if ((rc = mlkem_compress_dev(ctx, d, n, b.as<uint16_t>(), b.as<uint16_t>(), nullptr))) return rc;
    HIP_TRY(hipMemcpy(y, b.p, n * 2, hipMemcpyDeviceToHost));
    return MLKEM_OK;
}
int mlkem_decompress(int d, size_t n, const uint16_t* y, uint16_t* x) {
    if (d < 1 || d > 12 || (n && (!x || !y))) return MLKEM_ERR_ARG;
    MLKEM_HOST_PROLOGUE()
    if (n == 0) return MLKEM_OK;
    DevBuf b;
    if ((rc = b.alloc(n * 2))) return rc;
    HIP_TRY(hipMemcpy(b.p, y, n * 2, hipMemcpyHostToDevice));
    if ((rc = mlkem_decompress_dev(ctx, d, n, b.as<uint16_t>(), b.as<uint16_t>(), nullptr))) return rc;
    HIP_TRY(hipMemcpy(x, b.p, n * 2, hipMemcpyDeviceToHost));
    return MLKEM_OK;
}

// ---- randomised wrappers: KEM_KeyGen / KEM_Encaps semantics at batch scale (ml_kem.c:458-478, :1233, :1257) --
static bool fill_random(uint8_t* p, size_t n) {
    while (n) {
        ssize_t got = getrandom(p, n > 256 ? 256 : n, 0);
        if (got <= 0) return false;
        p += got;
        n -= (size_t)got;
    }
    return true;
}

int mlkem_keygen_random(int set, size_t n, uint8_t* ek, uint8_t* dk) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    return guarded([&]() -> int {
        std::vector<uint8_t> d(n * 32 + 1), z(n * 32 + 1);
        int rc = MLKEM_ERR_RNG;
        if (fill_random(d.data(), n * 32) && fill_random(z.data(), n * 32)) rc = mlkem_keygen(set, n, d.data(), z.data(), ek, dk);
        explicit_bzero(d.data(), d.size());
        explicit_bzero(z.data(), z.size());
        return rc;
    });
}

int mlkem_encaps_random(int set, size_t n, const uint8_t* ek, unsigned ek_len, uint8_t* c, uint8_t* K) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (ek_len != p.ek_len) return MLKEM_ERR_LENGTH;   // ml_kem.c:1267-1271; the modulus check that follows is a no-op (F3)
    return guarded([&]() -> int {
        std::vector<uint8_t> m(n * 32 + 1);
        int rc = MLKEM_ERR_RNG;
        if (fill_random(m.data(), n * 32)) rc = mlkem_encaps(set, n, ek, m.data(), c, K);
        explicit_bzero(m.data(), m.size());
        return rc;
    });
}

}   // extern "C"

// ---- in-process sharding over several devices (BASELINE configs[4]) ---------------------------------------
// An mlkem_multi is a list of MEMBERS, each bound to one HIP device (a device may appear more than once: that is how
// the sharded path is rehearsed on a single GPU).  A batch of n items is cut into contiguous ranges, member r takes
// items [start_r, stop_r) = mlkem_shard_range(n, r, R); there is no exchange between members and no collective.
//   *_multi      host-resident batch: one host thread per member drives that member's streaming engine (own streams,
//                events, pinned staging) on its device; the call returns when every member has finished
//   *_multi_dev  device-resident shards: shard r already lives on member r's device; the work is enqueued on the member's
//                stream and the call returns without synchronising (mlkem_multi_sync waits for all members)
// One persistent host thread per member: started on the member's first host-resident call, bound to the member's device
// once, fed one job at a time through a mutex + condition variable, joined by mlkem_multi_destroy.  (A std::thread per
// call cost a thread creation + hipSetDevice per member and call.)
namespace {
struct MemberWorker {
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = false, quit = false;
    std::thread th;
    void loop(int device) {
        const bool dev_ok = hipSetDevice(device) == hipSuccess;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            std::function<void()> j = std::move(job);
            has_job = false;
            lk.unlock();
            if (dev_ok) j();
            lk.lock();
            job_ok = dev_ok;
            done = true;
            cv.notify_all();
        }
    }
    bool job_ok = true;   // false: the worker could not select its device and did not run the job
    void start(int device) {
        if (!th.joinable()) th = std::thread([this, device] { loop(device); });
    }
    void post(std::function<void()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    bool wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return job_ok;
    }
    void stop() {
        if (!th.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        th.join();
    }
};
}   // namespace

struct mlkem_multi {
    struct Member {
        int device = 0;
        mlkem_ctx* ctx = nullptr;       // *_multi_dev: created on first use (a context owns ~10 KiB x chunk of HBM)
        hipStream_t st = nullptr;
        StreamEngine eng;               // *_multi
        MemberWorker worker;            // *_multi: members 1 .. R-1 (member 0 runs on the calling thread)
    };
    std::vector<Member*> mem;
    size_t chunk = 0;
    std::mutex call_mu;                 // one host-resident call at a time per mlkem_multi (the workers take one job each)
};

namespace {

// Member streams come from a per-device pool that lives as long as the process (like the stream pools of the frameworks
// that sit on top of this library): a stream handed out through mlkem_multi_stream() may have been recorded in a caller's
// allocator (torch's record_stream) or events, and those outlive the mlkem_multi.  mlkem_multi_destroy() synchronises the
// stream and puts it back; nothing ever destroys it.
struct StreamPool {
    std::mutex mu;
    std::map<int, std::vector<std::pair<hipStream_t, bool>>> per_dev;   // (stream, in use)
};
StreamPool& stream_pool() {
    static StreamPool* p = new StreamPool();   // leaked on purpose: no HIP calls during static destruction
    return *p;
}
hipStream_t pool_acquire(int dev) {   // `dev` is current
    StreamPool& P = stream_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    auto& v = P.per_dev[dev];
    for (auto& e : v)
        if (!e.second) { e.second = true; return e.first; }
    hipStream_t s = nullptr;
    if (!hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate")) return nullptr;
    v.emplace_back(s, true);
    return s;
}
void pool_release(int dev, hipStream_t s) {
    StreamPool& P = stream_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    for (auto& e : P.per_dev[dev])
        if (e.first == s) e.second = false;
}

int member_ready(mlkem_multi* mm, mlkem_multi::Member& m) {   // m.device is current
    if (!m.st && !(m.st = pool_acquire(m.device))) return MLKEM_ERR_NO_DEVICE;
    if (!m.ctx) return mlkem_ctx_create(&m.ctx, m.device, mm->chunk);
    return MLKEM_OK;
}

// fn(member index) runs with the member's device current: member 0 on the calling thread, the others on their workers
template <class Fn>
int multi_run_threads(mlkem_multi* mm, Fn fn) {
    const size_t R = mm->mem.size();
    std::lock_guard<std::mutex> call_lock(mm->call_mu);
    std::vector<int> rcs(R, MLKEM_OK);
    std::vector<std::string> errs(R);
    // the jobs reference this frame: whatever fails below, every posted job is waited for before the frame is left
    size_t posted = 1;
    int post_rc = MLKEM_OK;
    try {
        for (size_t r = 1; r < R; r++) mm->mem[r]->worker.start(mm->mem[r]->device);
        for (size_t r = 1; r < R; r++, posted++)
            mm->mem[r]->worker.post([&, r] {
                rcs[r] = guarded([&]() -> int { return fn(r); });
                if (rcs[r]) errs[r] = g_last_hip_error;
            });
    } catch (...) {
        post_rc = MLKEM_ERR_ALLOC;
    }
    if (post_rc == MLKEM_OK) {
        DeviceGuard g;
        if (!g.enter(mm->mem[0]->device)) rcs[0] = MLKEM_ERR_NO_DEVICE;
        else {
            rcs[0] = guarded([&]() -> int { return fn(0); });
            if (rcs[0]) errs[0] = g_last_hip_error;
        }
    }
    for (size_t r = 1; r < posted; r++)
        if (!mm->mem[r]->worker.wait() && rcs[r] == MLKEM_OK) rcs[r] = MLKEM_ERR_NO_DEVICE;
    if (post_rc != MLKEM_OK) return post_rc;
    for (size_t r = 0; r < R; r++)
        if (rcs[r]) {
            g_last_hip_error = errs[r];
            return rcs[r];
        }
    return MLKEM_OK;
}

}   // namespace

extern "C" {

int mlkem_shard_range(size_t n, int member, int n_members, size_t* start, size_t* stop) {
    if (n_members <= 0 || member < 0 || member >= n_members || !start || !stop) return MLKEM_ERR_ARG;
    const size_t base = n / (size_t)n_members, rem = n % (size_t)n_members, r = (size_t)member;
    *start = r * base + (r < rem ? r : rem);
    *stop = *start + base + (r < rem ? 1 : 0);
    return MLKEM_OK;
}

int mlkem_multi_create(mlkem_multi** out, int n_members, const int* devices, size_t chunk_items) {
    if (!out) return MLKEM_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (!hip_ok(hipGetDeviceCount(&ndev), "hipGetDeviceCount") || ndev <= 0) return MLKEM_ERR_NO_DEVICE;
    if (n_members == 0 && !devices) n_members = ndev;   // every visible device once
    if (n_members <= 0 || n_members > 1024) return MLKEM_ERR_ARG;
    for (int r = 0; r < n_members; r++) {
        const int d = devices ? devices[r] : r;
        if (d < 0 || d >= ndev) return MLKEM_ERR_ARG;
    }
    mlkem_multi* mm = new (std::nothrow) mlkem_multi();
    if (!mm) return MLKEM_ERR_ALLOC;
    mm->chunk = chunk_items;
    for (int r = 0; r < n_members; r++) {
        auto* m = new (std::nothrow) mlkem_multi::Member();
        if (!m) { mlkem_multi_destroy(mm); return MLKEM_ERR_ALLOC; }
        m->device = devices ? devices[r] : r;
        mm->mem.push_back(m);
    }
    *out = mm;
    return MLKEM_OK;
}

void mlkem_multi_destroy(mlkem_multi* mm) {
    if (!mm) return;
    DeviceGuard g;
    for (auto* m : mm->mem) {
        m->worker.stop();
        (void)g.enter(m->device);
        if (m->st) (void)hipStreamSynchronize(m->st);
        if (m->ctx) mlkem_ctx_destroy(m->ctx);
        if (m->st) pool_release(m->device, m->st);   // synchronised above; the stream itself lives on (see StreamPool)
        engine_release(m->eng);
        delete m;
    }
    delete mm;
}

int mlkem_multi_members(const mlkem_multi* mm) { return mm ? (int)mm->mem.size() : 0; }
int mlkem_multi_device(const mlkem_multi* mm, int member) {
    if (!mm || member < 0 || member >= (int)mm->mem.size()) return MLKEM_ERR_ARG;
    return mm->mem[member]->device;
}

// ---- host-resident batch, sharded over the members ------------------------------------------------------------------
static int multi_host(mlkem_multi* mm, int op, int set, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* x, uint8_t* y,
                      size_t chunk_items) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!mm || mm->mem.empty() || (n && (!a || !b || !x || !y))) return MLKEM_ERR_ARG;
    const size_t la = op == 0 ? 32 : op == 1 ? p.ek_len : p.dk_len, lb = op == 2 ? p.c_len : 32;
    const size_t lx = op == 0 ? p.ek_len : op == 1 ? p.c_len : 32, ly = op == 0 ? p.dk_len : op == 1 ? 32 : 4;
    const int R = (int)mm->mem.size();
    return guarded([&]() -> int {
        return multi_run_threads(mm, [&](size_t r) -> int {
            size_t lo, hi;
            mlkem_shard_range(n, (int)r, R, &lo, &hi);
            return kem_stream(mm->mem[r]->eng, op, set, hi - lo, a + lo * la, b + lo * lb, x + lo * lx, y + lo * ly, chunk_items);
        });
    });
}
int mlkem_keygen_multi(mlkem_multi* mm, int set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk, size_t chunk_items) {
    return multi_host(mm, 0, set, n, d, z, ek, dk, chunk_items);
}
int mlkem_encaps_multi(mlkem_multi* mm, int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K, size_t chunk_items) {
    return multi_host(mm, 1, set, n, ek, m, c, K, chunk_items);
}
int mlkem_decaps_multi(mlkem_multi* mm, int set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status, size_t chunk_items) {
    if (status || n == 0) return multi_host(mm, 2, set, n, dk, c, K, reinterpret_cast<uint8_t*>(status), chunk_items);
    return guarded([&]() -> int {
        std::vector<int32_t> st(n);
        return multi_host(mm, 2, set, n, dk, c, K, reinterpret_cast<uint8_t*>(st.data()), chunk_items);
    });
}

// ---- device-resident shards ------------------------------------------------------------------------------------------
// Arrays of length mlkem_multi_members(): shard r (n_shard[r] items) lives on member r's device.  Enqueued from the
// calling thread (a launch costs microseconds, a shard milliseconds), each member on its own stream; not ordered after
// any other stream: the inputs must be complete when the call is made.
static int multi_dev(mlkem_multi* mm, int op, int set, const size_t* n_shard, const uint8_t* const* a, const uint8_t* const* b,
                     uint8_t* const* x, void* const* y) {
    ParamSet p;
    if (!param_set(set, p)) return MLKEM_ERR_PARAM_SET;
    if (!mm || mm->mem.empty() || !n_shard || !a || !b || !x) return MLKEM_ERR_ARG;
    DeviceGuard g;
    int rc = MLKEM_OK;
    for (size_t r = 0; r < mm->mem.size() && rc == MLKEM_OK; r++) {
        auto& m = *mm->mem[r];
        if (!g.enter(m.device)) return MLKEM_ERR_NO_DEVICE;
        if ((rc = member_ready(mm, m))) break;
        const size_t n = n_shard[r];
        if (op == 0) rc = mlkem_keygen_dev(m.ctx, set, n, a[r], b[r], x[r], static_cast<uint8_t*>(y ? y[r] : nullptr), m.st);
        else if (op == 1) rc = mlkem_encaps_dev(m.ctx, set, n, a[r], b[r], x[r], static_cast<uint8_t*>(y ? y[r] : nullptr), m.st);
        else rc = mlkem_decaps_dev(m.ctx, set, n, a[r], b[r], x[r], static_cast<int32_t*>(y ? y[r] : nullptr), m.st);
    }
    return rc;
}
int mlkem_keygen_multi_dev(mlkem_multi* mm, int set, const size_t* n_shard, const uint8_t* const* d, const uint8_t* const* z,
                           uint8_t* const* ek, uint8_t* const* dk) {
    return multi_dev(mm, 0, set, n_shard, d, z, ek, reinterpret_cast<void* const*>(dk));
}
int mlkem_encaps_multi_dev(mlkem_multi* mm, int set, const size_t* n_shard, const uint8_t* const* ek, const uint8_t* const* m,
                           uint8_t* const* c, uint8_t* const* K) {
    return multi_dev(mm, 1, set, n_shard, ek, m, c, reinterpret_cast<void* const*>(K));
}
int mlkem_decaps_multi_dev(mlkem_multi* mm, int set, const size_t* n_shard, const uint8_t* const* dk, const uint8_t* const* c,
                           uint8_t* const* K, int32_t* const* status) {
    return multi_dev(mm, 2, set, n_shard, dk, c, K, reinterpret_cast<void* const*>(status));
}
// the HIP stream member `member` enqueues its device-resident work on (taken from the process-lifetime pool on first use):
// lets a caller order that stream after the producers of its inputs (hipStreamWaitEvent) and time or consume the member's
// work with events; the handle stays a valid stream after mlkem_multi_destroy
void* mlkem_multi_stream(mlkem_multi* mm, int member) {
    if (!mm || member < 0 || member >= (int)mm->mem.size()) return nullptr;
    auto& m = *mm->mem[member];
    DeviceGuard g;
    if (!g.enter(m.device)) return nullptr;
    if (!m.st) m.st = pool_acquire(m.device);
    return m.st;
}
int mlkem_multi_sync(mlkem_multi* mm) {
    if (!mm) return MLKEM_ERR_ARG;
    DeviceGuard g;
    for (auto* m : mm->mem) {
        if (!m->st) continue;
        if (!g.enter(m->device) || !hip_ok(hipStreamSynchronize(m->st), "hipStreamSynchronize")) return MLKEM_ERR_NO_DEVICE;
    }
    return MLKEM_OK;
}

}   // extern "C"
