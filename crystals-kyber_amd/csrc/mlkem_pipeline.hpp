// mlkem_pipeline.hpp — kernel sequencing for the batched ML-KEM operations.
//
// The same sequencing is compiled twice: by hipcc into libmlkem_amd.so (real launches on a HIP stream)
// and by g++ into the test-only wave emulator (tests/emu), which runs the identical kernel bodies on
// 64 host threads per wave so that kernel logic can be checked without a GPU.  There is no CPU fallback
// in the product: the emulator is never built into, or loaded by, libmlkem_amd.so.
#pragma once
#include "mlkem_kernels.hpp"
#ifndef MLKEM_EMU
#include <vector>
#endif

namespace mlkem {

#ifdef MLKEM_EMU
using stream_t = void*;
#else
using stream_t = hipStream_t;
#endif

#ifndef MLKEM_EMU
// Optional per-kernel timing (bench.py's roofline leg): when a recorder is installed for the calling thread,
// every launch is bracketed by HIP events on the launch stream.
struct LaunchRecorder {
    struct Rec { const char* label; hipEvent_t a, b; };
    std::vector<Rec> recs;
};
inline LaunchRecorder*& launch_recorder() {
    static thread_local LaunchRecorder* r = nullptr;
    return r;
}
#endif

template <class... KArgs, class... Args>
inline void launch(const char* label, void (*kfn)(KArgs...), size_t grid, unsigned block, stream_t st, Args... args) {
    if (grid == 0) return;
#ifdef MLKEM_EMU
    (void)st; (void)label;
    emu::launch((unsigned)grid, block, [=] { kfn(args...); });
#else
    LaunchRecorder* rec = launch_recorder();
    if (rec) {
        LaunchRecorder::Rec r{label, nullptr, nullptr};
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, st);
        kfn<<<dim3((unsigned)grid), dim3(block), 0, st>>>(args...);
        (void)hipEventRecord(r.b, st);
        rec->recs.push_back(r);
    } else {
        kfn<<<dim3((unsigned)grid), dim3(block), 0, st>>>(args...);
    }
#endif
}

struct ParamSet {
    int set, k, eta1, eta2, du, dv;
    unsigned ek_len, dk_len, c_len;
};
// ml_kem.c:1363-1395 (init) + ml_kem.h:52-59 (lengths)
inline bool param_set(int set, ParamSet& p) {
    switch (set) {
    case 512: p = {512, 2, 3, 2, 10, 4, 0, 0, 0}; break;
    case 768: p = {768, 3, 2, 2, 10, 4, 0, 0, 0}; break;
    case 1024: p = {1024, 4, 2, 2, 11, 5, 0, 0, 0}; break;
    default: return false;
    }
    p.ek_len = 384u * p.k + 32;
    p.dk_len = 768u * p.k + 96;
    p.c_len = 32u * (p.du * p.k + p.dv);
    return true;
}

// Per-chunk scratch in HBM (bytes per item for parameter k, worst case over the three operations):
//   A    k*k*512   sampled matrix, uint16 coefficients, natural order
//   prf  (2k+1)*PS raw PRF output rows (PS = 192 for eta1 = 3, else 128)
//   r, rho, m, Kp, Kbar : 32 each
struct Workspace {
    uint16_t* A = nullptr;
    uint8_t *prf = nullptr, *r = nullptr, *rho = nullptr, *m = nullptr, *Kp = nullptr, *Kbar = nullptr;
    size_t cap_items = 0;   // capacity in items at k = 4
    static size_t bytes_per_item() { return 16 * 512 + 9 * 192 + 5 * 32; }
};

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

inline void launch_sample(stream_t st, const ParamSet& p, size_t n, const uint8_t* rho, size_t rho_stride, int transpose,
                          const uint8_t* r, int prf_per_item, int n_eta1, const Workspace& ws) {
    SampleArgs a{};
    a.n_xof = n * (size_t)(p.k * p.k);
    a.rho = rho; a.rho_stride = rho_stride; a.K = p.k; a.transpose = transpose; a.A = ws.A;
    a.xof_blocks = (unsigned)ceil_div(a.n_xof, WAVE);
    a.n_prf = n * (size_t)prf_per_item;
    a.r = r; a.per_item = prf_per_item; a.n_eta1 = n_eta1; a.eta1 = p.eta1; a.prf = ws.prf;
    a.prf_stride = p.eta1 == 3 ? 192 : 128;
    launch("k_sample", k_sample, a.xof_blocks + ceil_div(a.n_prf, WAVE), WAVE, st, a);
}

// ---- ML-KEM.KeyGen_internal (ml_kem.c:1034-1084) -------------------------------------------------
template <int K, int ETA1>
inline void keygen_chunk(stream_t st, const ParamSet& p, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk,
                         const Workspace& ws) {
    launch("k_hash_keygen_seed", k_hash_keygen_seed<K>, ceil_div(n, WAVE), WAVE, st, n, d, ws.rho, ws.r);
    launch_sample(st, p, n, ws.rho, 32, /*transpose=*/0, ws.r, 2 * K, 2 * K, ws);
    launch("k_keygen", k_keygen<K, ETA1>, ceil_div(n, ARITH_WAVES), WAVE * ARITH_WAVES, st, n, (const uint16_t*)ws.A, (const uint8_t*)ws.prf,
           (const uint8_t*)ws.rho, ek, dk);
    launch("k_hash_keygen_fin", k_hash_keygen_fin<K>, ceil_div(n, WAVE), WAVE, st, n, (const uint8_t*)ek, z, dk);
}

// ---- ML-KEM.Encaps_internal (ml_kem.c:1093-1130) -------------------------------------------------
template <int K, int ETA1, int DU, int DV>
inline void encaps_chunk(stream_t st, const ParamSet& p, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* Kout,
                         const Workspace& ws) {
    launch("k_hash_encaps", k_hash_encaps<K>, ceil_div(n, WAVE), WAVE, st, n, ek, m, Kout, ws.r);
    launch_sample(st, p, n, ek + 384 * K, p.ek_len, /*transpose=*/1, ws.r, 2 * K + 1, K, ws);
    launch("k_encrypt", k_encrypt<K, ETA1, DU, DV, false>, ceil_div(n, ARITH_WAVES), WAVE * ARITH_WAVES, st, n, ek, (size_t)p.ek_len, m,
           (const uint16_t*)ws.A, (const uint8_t*)ws.prf, c, (const uint8_t*)nullptr, (const uint8_t*)nullptr,
           (const uint8_t*)nullptr, (uint8_t*)nullptr);
}

// ---- KEM_Decaps / Decaps_internal (ml_kem.c:1310-1359, :1136-1225) -------------------------------
// hash_check = true reproduces the public KEM_Decaps: status[i] = -5 when H(dk.ek) != dk.h (K[i] is then
// still the Decaps_internal result; the host shim discards it like the reference does).
template <int K, int ETA1, int DU, int DV>
inline void decaps_chunk(stream_t st, const ParamSet& p, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* Kout,
                         int32_t* status, bool hash_check, const Workspace& ws) {
    constexpr int CLEN = 32 * (DU * K + DV);
    launch("k_decrypt", k_decrypt<K, DU, DV>, ceil_div(n, ARITH_WAVES), WAVE * ARITH_WAVES, st, n, dk, (size_t)p.dk_len, c, ws.m);
    if (hash_check)
        launch("k_hash_decaps", k_hash_decaps<K, CLEN, true>, ceil_div(n, WAVE), WAVE, st, n, dk, c, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, status);
    else
        launch("k_hash_decaps", k_hash_decaps<K, CLEN, false>, ceil_div(n, WAVE), WAVE, st, n, dk, c, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, status);
    launch_sample(st, p, n, dk + 768 * K, p.dk_len, /*transpose=*/1, ws.r, 2 * K + 1, K, ws);
    launch("k_encrypt_cmp", k_encrypt<K, ETA1, DU, DV, true>, ceil_div(n, ARITH_WAVES), WAVE * ARITH_WAVES, st, n, dk + 384 * K, (size_t)p.dk_len,
           (const uint8_t*)ws.m, (const uint16_t*)ws.A, (const uint8_t*)ws.prf, (uint8_t*)nullptr, c, (const uint8_t*)ws.Kp,
           (const uint8_t*)ws.Kbar, Kout);
}

inline int keygen_dispatch(stream_t st, int set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk,
                           const Workspace& ws) {
    ParamSet p;
    if (!param_set(set, p)) return -1;
    switch (set) {
    case 512: keygen_chunk<2, 3>(st, p, n, d, z, ek, dk, ws); break;
    case 768: keygen_chunk<3, 2>(st, p, n, d, z, ek, dk, ws); break;
    default: keygen_chunk<4, 2>(st, p, n, d, z, ek, dk, ws); break;
    }
    return 0;
}
inline int encaps_dispatch(stream_t st, int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K,
                           const Workspace& ws) {
    ParamSet p;
    if (!param_set(set, p)) return -1;
    switch (set) {
    case 512: encaps_chunk<2, 3, 10, 4>(st, p, n, ek, m, c, K, ws); break;
    case 768: encaps_chunk<3, 2, 10, 4>(st, p, n, ek, m, c, K, ws); break;
    default: encaps_chunk<4, 2, 11, 5>(st, p, n, ek, m, c, K, ws); break;
    }
    return 0;
}
inline int decaps_dispatch(stream_t st, int set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status,
                           bool hash_check, const Workspace& ws) {
    ParamSet p;
    if (!param_set(set, p)) return -1;
    switch (set) {
    case 512: decaps_chunk<2, 3, 10, 4>(st, p, n, dk, c, K, status, hash_check, ws); break;
    case 768: decaps_chunk<3, 2, 10, 4>(st, p, n, dk, c, K, status, hash_check, ws); break;
    default: decaps_chunk<4, 2, 11, 5>(st, p, n, dk, c, K, status, hash_check, ws); break;
    }
    return 0;
}

// ---- stand-alone primitives ------------------------------------------------------------------------
inline size_t poly_grid(size_t n) {
    size_t g = ceil_div(n, ARITH_WAVES);
    const size_t cap = 256 * 16;   // 256 CUs x a few workgroups, grid-stride for the rest
    return g < cap ? g : cap;
}
inline void ntt_launch(stream_t st, bool inverse, size_t n, const uint16_t* in, uint16_t* out) {
    if (inverse) launch("k_intt_batch", k_ntt_batch<true>, poly_grid(n), WAVE * ARITH_WAVES, st, n, in, out);
    else launch("k_ntt_batch", k_ntt_batch<false>, poly_grid(n), WAVE * ARITH_WAVES, st, n, in, out);
}
inline void basemul_launch(stream_t st, size_t n, const uint16_t* a, const uint16_t* b, uint16_t* h) {
    launch("k_basemul_batch", k_basemul_batch, poly_grid(n), WAVE * ARITH_WAVES, st, n, a, b, h);
}
inline int cbd_launch(stream_t st, int eta, size_t n, const uint8_t* bytes, uint16_t* out) {
    if (eta == 2) launch("k_cbd_batch", k_cbd_batch<2>, poly_grid(n), WAVE * ARITH_WAVES, st, n, bytes, out);
    else if (eta == 3) launch("k_cbd_batch", k_cbd_batch<3>, poly_grid(n), WAVE * ARITH_WAVES, st, n, bytes, out);
    else return -1;
    return 0;
}
inline void sample_ntt_launch(stream_t st, size_t n, const uint8_t* seeds34, uint16_t* out) {
    SampleArgs a{};
    a.n_xof = n; a.rho = seeds34; a.rho_stride = 34; a.K = 0; a.A = out;
    a.xof_blocks = (unsigned)ceil_div(n, WAVE);
    launch("k_sample_xof", k_sample, (size_t)a.xof_blocks, WAVE, st, a);
}
inline int prf_launch(stream_t st, int eta, size_t n, const uint8_t* in33, uint8_t* out) {
    if (eta != 2 && eta != 3) return -1;
    SampleArgs a{};
    a.n_prf = n; a.r = in33; a.per_item = 0; a.eta1 = eta; a.prf = out; a.prf_stride = 64u * (unsigned)eta;
    launch("k_sample_prf", k_sample, ceil_div(n, WAVE), WAVE, st, a);
    return 0;
}
inline int hash_launch(stream_t st, int kind, size_t n, const uint8_t* msg, unsigned len, size_t stride, uint8_t* out) {
    if (kind < 0 || kind > 2 || (stride & 3) || stride < len) return -1;   // rows must start 4-byte aligned
    launch("k_hash_batch", k_hash_batch, ceil_div(n, WAVE), WAVE, st, n, kind, msg, len, stride, out);
    return 0;
}

}   // namespace mlkem
