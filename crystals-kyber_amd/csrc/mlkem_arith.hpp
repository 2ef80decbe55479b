// mlkem_arith.hpp — wave-per-instance polynomial kernels: K-PKE KeyGen / Encrypt / Decrypt (ml_kem.c:651-1023) after
// hashing and sampling, plus the stand-alone NTT / MultiplyNTTs / CBD primitives.
//
// One wavefront owns one KEM instance (or one polynomial); each lane holds 4 coefficients (NAT layout: 4l..4l+3,
// which is also the base-case-multiply pair layout and the HBM layout).  Modular arithmetic runs exactly on the
// fp32 pipe (mlkem_fntt.hpp); Compress / ByteEncode / ByteDecode are integer work on a wave-private LDS byte buffer.
// No workgroup barrier is used: the 4 waves of a workgroup are independent.
//
// Memory-latency structure (profiles/r01_pmc_sq_arith_before_prefetch.txt: 57 % of a wave's life was s_waitcnt): the wave-LDS fences are
// compiler barriers, so a load written next to its use is issued next to its use.  Every kernel therefore issues ALL
// of its prologue loads (PRF bytes, packed key / ciphertext polynomials, message bits) before the first NTT, and the
// rows of the sampled matrix are double-buffered: row a+1 is requested before row a is consumed.
#pragma once
#include "mlkem_fntt.hpp"

namespace mlkem {

#ifndef MLKEM_ARITH_WAVES
#define MLKEM_ARITH_WAVES 4
#endif
constexpr int ARITH_WAVES = MLKEM_ARITH_WAVES;   // waves per workgroup (each fully independent)
// __launch_bounds__ second argument of the K-PKE kernels = resident waves per SIMD the register allocator aims at.  The
// kernels hide their LDS-exchange latency with occupancy (A/B on one box: 6 waves beat 5 by 3-4 % in k_encrypt<3>, 7 beat 6
// by another 2-3 %, profiles/r03_kpke_experiments.txt; a loop-over-items form with 13 % fewer instructions but 4 waves was
// slower): 7 where the LDS block allows it (k = 2, 3: 16-22 KB per workgroup; the build uses -fno-slp-vectorize, under which
// k_encrypt<3> needs 66-69 VGPRs), 5 for k = 4 (28 KB).  MLKEM_ARITH_MINWAVES overrides for experiments.
#ifdef MLKEM_ARITH_MINWAVES
constexpr int arith_minwaves(int) { return MLKEM_ARITH_MINWAVES; }
#else
constexpr int arith_minwaves(int k) { return k == 4 ? 5 : 7; }
#endif

template <int K>
struct __attribute__((aligned(16))) ArithLds {
    float xch[256];         // NTT exchange buffer; between transforms it doubles as the codec byte buffer (cbuf())
    float vhat[K][256];     // NTT-domain vector (y-hat or s-hat), reduced
    float vgam[K][128];     // its odd coefficients times gamma (ml_kem.c:402-403)
    // The codec byte buffer lives in the exchange buffer: a codec step never overlaps a transform of the same wave, both sides
    // fence their LDS traffic, and a wave's DS operations execute in issue order.  5.5 instead of 5.9 KB per wave at k = 3, so
    // that a seventh workgroup fits the CU's 160 KB.
    __device__ __forceinline__ uint32_t* cbuf() { return reinterpret_cast<uint32_t*>(xch); }
};
static_assert(CODEC_BUF_WORDS * 4 <= 256 * 4, "the codec buffer must fit the exchange buffer");

__device__ __forceinline__ void store_poly_nat(uint16_t* p, const int (&x)[4]) {
    uint2 v;
    v.x = ((uint32_t)x[0] & 0xFFFFu) | ((uint32_t)x[1] << 16);
    v.y = ((uint32_t)x[2] & 0xFFFFu) | ((uint32_t)x[3] << 16);
    *reinterpret_cast<uint2*>(p + 4 * lane_id()) = v;
}
__device__ __forceinline__ uint2 load_poly_raw(const uint16_t* p) { return stream_load8(p + 4 * lane_id()); }
__device__ __forceinline__ void poly_raw_to_f(const uint2 v, float (&x)[4]) {
    x[0] = (float)(v.x & 0xFFFFu); x[1] = (float)(v.x >> 16);
    x[2] = (float)(v.y & 0xFFFFu); x[3] = (float)(v.y >> 16);
}

// ---- SamplePolyCBD (ml_kem.c:253-275), split into the load of the lane's 8*ETA bits and their evaluation --------
template <int ETA>
__device__ __forceinline__ uint32_t cbd_load(const uint8_t* prf) {
    const int l = lane_id();
    if constexpr (ETA == 2) return *reinterpret_cast<const uint16_t*>(prf + 2 * l);   // 4 coefficients = 16 bits
    else return (uint32_t)prf[3 * l] | ((uint32_t)prf[3 * l + 1] << 8) | ((uint32_t)prf[3 * l + 2] << 16);   // 24 bits
}
template <int ETA>
__device__ __forceinline__ void cbd_eval_f(uint32_t t, float (&x)[4]) {
    if constexpr (ETA == 2) {
        // coefficient m = (a0 + a1) - (b0 + b1) of nibble m = popcount(nibble ^ 0b1100) - 2.  The four nibbles are spread
        // to the four bytes of one register (order n0, n2, n1, n3), counted in parallel, and converted with
        // v_cvt_f32_ubyteN: 19 instructions for the 4 coefficients, 7 of them slow-class (the per-coefficient bit-field
        // form took 27, most of them slow-class).
        uint32_t u = ((t | (t << 12)) & 0x0F0F0F0Fu) ^ 0x0C0C0C0Cu;
        u = (u & 0x05050505u) + ((u >> 1) & 0x05050505u);
        u = (u & 0x03030303u) + ((u >> 2) & 0x03030303u);
        float c0 = (float)(u & 0xFFu), c2 = (float)((u >> 8) & 0xFFu), c1 = (float)((u >> 16) & 0xFFu), c3 = (float)(u >> 24);
#ifndef MLKEM_EMU   // keep the byte -> float conversions as they are (v_cvt_f32_ubyteN): folding the "- 2" into an
                    // integer subtract per byte costs a second slow-class instruction per coefficient
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
#endif
        x[0] = c0 - 2.0f; x[1] = c1 - 2.0f; x[2] = c2 - 2.0f; x[3] = c3 - 2.0f;
    } else {
        const uint32_t d = (t & 0x249249u) + ((t >> 1) & 0x249249u) + ((t >> 2) & 0x249249u);   // 3-bit group sums
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = (float)((int)((d >> (6 * m)) & 7u) - (int)((d >> (6 * m + 3)) & 7u));
    }
}

// ---- packed-polynomial bytes held in registers between the (early) load and the (late) decode -------------------
template <int D>
struct CodecRegs {
    static constexpr int NW = (8 * D + 63) / 64;   // dwords per lane: 32*D bytes = 8*D dwords over 64 lanes
    uint32_t w[NW];
};
template <int D>
__device__ __forceinline__ void codec_fetch(const uint8_t* g, CodecRegs<D>& r) {
    const int l = lane_id();
    const uint32_t* gw = reinterpret_cast<const uint32_t*>(g);
#pragma unroll
    for (int i = 0; i < CodecRegs<D>::NW; i++) r.w[i] = (l + 64 * i < 8 * D) ? stream_load4(gw + l + 64 * i) : 0u;
}
// ByteDecode_D (+ optional Decompress_D) (ml_kem.c:153-177, :104-119) of pre-fetched bytes -> 4 floats per lane
template <int D, bool DECOMPRESS>
__device__ __forceinline__ void decode_regs(uint32_t* cbuf, const CodecRegs<D>& r, float (&x)[4]) {
    const int l = lane_id();
#pragma unroll
    for (int i = 0; i < CodecRegs<D>::NW; i++)
        if (l + 64 * i < 8 * D + 4) cbuf[l + 64 * i] = r.w[i];   // the 4 words past the end are zero (fetch masks them)
    if (8 * D + 4 > 64 * CodecRegs<D>::NW && l < 4) cbuf[8 * D + l] = 0;
    wave_lds_fence();
    unsigned v[4];
    codec_decode<D>(cbuf, v);
    wave_lds_fence();
#pragma unroll
    for (int m = 0; m < 4; m++) {
        if constexpr (DECOMPRESS) x[m] = (float)decompress_d<D>(v[m]);
        else x[m] = (float)v[m];   // raw D-bit value: for D = 12 no reduction mod q (ml_kem.c:170, F3)
    }
}

// Compress_D + ByteEncode_D (ml_kem.c:83-97, :125-145) of x (any representative, |x| <= 2^24), then either store the
// 32*D bytes or compare them with the pre-fetched reference bytes
template <int D, bool COMPARE>
__device__ __forceinline__ uint32_t emit_compressed(uint32_t* cbuf, const float (&x)[4], uint8_t* out, const CodecRegs<D>& ref) {
    unsigned v[4];
#pragma unroll
    for (int m = 0; m < 4; m++) v[m] = compress_f<D>(x[m]);
    codec_zero<D>(cbuf);
    wave_lds_fence();
    codec_encode<D>(cbuf, v);
    wave_lds_fence();
    uint32_t diff = 0;
    if constexpr (COMPARE) {
        const int l = lane_id();
#pragma unroll
        for (int i = 0; i < CodecRegs<D>::NW; i++)
            if (l + 64 * i < 8 * D) diff |= cbuf[l + 64 * i] ^ ref.w[i];
    } else {
        codec_store_bytes<D>(cbuf, out);
    }
    wave_lds_fence();
    return diff;
}
// ByteEncode_12 (ml_kem.c:736-756) of x (any representative) to one or two destinations
__device__ __forceinline__ void emit_encode12(uint32_t* cbuf, const float (&x)[4], uint8_t* out0, uint8_t* out1) {
    unsigned v[4];
#pragma unroll
    for (int m = 0; m < 4; m++) v[m] = (unsigned)fcanon(x[m]);
    codec_zero<12>(cbuf);
    wave_lds_fence();
    codec_encode<12>(cbuf, v);
    wave_lds_fence();
    codec_store_bytes<12>(cbuf, out0);
    if (out1) codec_store_bytes<12>(cbuf, out1);
    wave_lds_fence();
}

// ------------------------------------------------------------------------------------------------
// k_encrypt — K-PKE.Encrypt (ml_kem.c:776-936) given A^T (sampler, XOF role) and the PRF bytes (PRF role).
//   COMPARE = false : write c                                   (Encaps_internal, ml_kem.c:1127)
//   COMPARE = true  : compare c' with c, K = (c == c') ? K' : Kbar (Decaps_internal, ml_kem.c:1206-1215)
// ------------------------------------------------------------------------------------------------
template <int K, int ETA1, int DU, int DV, bool COMPARE>
__global__ void __launch_bounds__(WAVE * ARITH_WAVES, arith_minwaves(K))
k_encrypt(size_t n, const uint8_t* __restrict__ ek, size_t ek_stride, const uint8_t* __restrict__ msg, const uint16_t* __restrict__ A,
          const uint8_t* __restrict__ prf, uint8_t* __restrict__ c_out, const uint8_t* __restrict__ c_in, const uint8_t* __restrict__ Kp,
          const uint8_t* __restrict__ Kbar, uint8_t* __restrict__ Kout, int32_t* __restrict__ mod_status, size_t a_stride) {
    // a_stride: uint16 elements between the matrices of consecutive items (K*K*256), or 0 when every item uses the same
    // key and therefore the same matrix (shared-key batches: ek_stride is 0 as well)
    // mod_status (optional): per-item result of the FIPS 203 encapsulation-key modulus check, 0 or -4.  The reference's
    // own check can never fail (ml_kem.c:1273-1291, F3), so its callers pass nullptr.
    __shared__ ArithLds<K> lds_all[ARITH_WAVES];
    const int wv = wave_id(), l = lane_id();
    const size_t item = (size_t)blockIdx.x * ARITH_WAVES + wv;
    if (item >= n) return;
    ArithLds<K>& L = lds_all[wv];
    constexpr unsigned PS = (ETA1 == 3) ? 192 : 128, CLEN = 32 * (DU * K + DV);
    const uint8_t* my_prf = prf + item * (size_t)((2 * K + 1) * PS);
    const uint8_t* my_ek = ek + item * ek_stride;
    const uint16_t* my_A = A + item * a_stride;
    uint8_t* my_c = COMPARE ? nullptr : c_out + item * CLEN;
    const uint8_t* my_cin = COMPARE ? c_in + item * CLEN : nullptr;

    // ---- prologue: the loads the first phases need (PRF bytes, row 0 of A^T, message bits); the packed t-hat row and the
    //      reference ciphertext rows follow one matrix row ahead of their use (fewer live registers: 6 waves per SIMD) ----
    uint32_t raw_y[K], raw_e1[K], raw_e2;
#pragma unroll
    for (int b = 0; b < K; b++) raw_y[b] = cbd_load<ETA1>(my_prf + b * PS);
    uint2 a_next[K];
#pragma unroll
    for (int b = 0; b < K; b++) a_next[b] = load_poly_raw(my_A + b * 256);   // row 0 of A^T
#pragma unroll
    for (int a = 0; a < K; a++) raw_e1[a] = cbd_load<2>(my_prf + (K + a) * PS);
    raw_e2 = cbd_load<2>(my_prf + (2 * K) * PS);
    const unsigned mb = msg[item * 32 + (l >> 1)] >> (4 * (l & 1));   // the lane's 4 message bits
    CodecRegs<DU> cu_ref;   // the ciphertext row the current u row is compared with (fetched one row ahead)
    CodecRegs<DV> cv_ref;
    if constexpr (COMPARE) codec_fetch<DU>(my_cin, cu_ref);
    NttTwiddlesF tw;
    load_twiddles_f(tw);

    uint32_t diff = 0;
    float x[4];
    CodecRegs<12> that[K];
    // y-hat = NTT(CBD_eta1(PRF(r, n)))  n = 0..K-1   (ml_kem.c:826-836)
#pragma unroll
    for (int b = 0; b < K; b++) {
        cbd_eval_f<ETA1>(raw_y[b], x);
        wave_ntt_f(x, L.xch, tw);
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = fred(x[m]);
        stash_vhat_f(L.vhat[b], L.vgam[b], x, tw);
    }
    wave_lds_fence();
    // u[a] = InverseNTT(sum_b A^T[a][b] o y-hat[b]) + e1[a]  ->  Compress_du, ByteEncode_du   (ml_kem.c:854-896)
#pragma unroll
    for (int a = 0; a < K; a++) {
        uint2 a_cur[K];
#pragma unroll
        for (int b = 0; b < K; b++) a_cur[b] = a_next[b];
        if (a + 1 < K) {
#pragma unroll
            for (int b = 0; b < K; b++) a_next[b] = load_poly_raw(my_A + ((a + 1) * K + b) * 256);   // prefetch the next row
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < K; b++) {
            float av[4];
            poly_raw_to_f(a_cur[b], av);
            basemul_acc_f(acc, av, L.vhat[b], L.vgam[b]);
        }
        wave_intt_f(acc, L.xch, tw);
        float e[4];
        cbd_eval_f<2>(raw_e1[a], e);
#pragma unroll
        for (int m = 0; m < 4; m++) acc[m] += e[m];
        diff |= emit_compressed<DU, COMPARE>(L.cbuf(), acc, COMPARE ? nullptr : my_c + a * 32 * DU, cu_ref);
        if constexpr (COMPARE) {
            if (a + 1 < K) codec_fetch<DU>(my_cin + (a + 1) * 32 * DU, cu_ref);
            else codec_fetch<DV>(my_cin + K * 32 * DU, cv_ref);
        }
        if (a == K - 2 || K == 1) {   // t-hat is needed after the last row: fetch it one row ahead
#pragma unroll
            for (int b = 0; b < K; b++) codec_fetch<12>(my_ek + 384 * b, that[b]);
        }
    }
    // v = InverseNTT(t-hat . y-hat) + e2 + Decompress_1(m)  ->  Compress_dv, ByteEncode_dv   (ml_kem.c:867-904)
    {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        bool over = false;
#pragma unroll
        for (int b = 0; b < K; b++) {
            float tv[4];
            decode_regs<12, false>(L.cbuf(), that[b], tv);   // raw 12-bit values (F3)
#pragma unroll
            for (int m = 0; m < 4; m++) over = over || (tv[m] >= F_Q);
            basemul_acc_f(acc, tv, L.vhat[b], L.vgam[b]);
        }
        if (mod_status) {
            const bool bad = __ballot(over) != 0;
            if (l == 0) mod_status[item] = bad ? -4 : 0;
        }
        wave_intt_f(acc, L.xch, tw);
        float e[4];
        cbd_eval_f<2>(raw_e2, e);
#pragma unroll
        for (int m = 0; m < 4; m++) acc[m] += e[m] + (((mb >> m) & 1u) ? 1665.0f : 0.0f);   // Decompress_1(1) = 1665
        diff |= emit_compressed<DV, COMPARE>(L.cbuf(), acc, COMPARE ? nullptr : my_c + K * 32 * DU, cv_ref);
    }
    if constexpr (COMPARE) {
        // both candidates are read and blended by mask: neither a branch nor an address depends on whether the
        // ciphertext matched (the reference's early-exit compare, ml_kem.c:1206-1215, leaks it; implicit rejection
        // is meant to hide it)
        const uint32_t reject = __ballot(diff != 0) != 0 ? 0xFFFFFFFFu : 0u;
        if (l < 8) {
            const uint32_t kp = reinterpret_cast<const uint32_t*>(Kp + item * 32)[l];
            const uint32_t kb = reinterpret_cast<const uint32_t*>(Kbar + item * 32)[l];
            reinterpret_cast<uint32_t*>(Kout + item * 32)[l] = (kp & ~reject) | (kb & reject);
        }
    }
}

// K-PKE.Decrypt (ml_kem.c:942-1023) lives in mlkem_kpke4.hpp: four items per wave, all in registers.

// ------------------------------------------------------------------------------------------------
// k_keygen — K-PKE.KeyGen after G and sampling (ml_kem.c:696-756) + the plain copies of KeyGen_internal
// (ml_kem.c:1054-1062): ek = Enc12(t-hat) || rho ; dk = Enc12(s-hat) || ek || [H(ek)] || [z]
// ------------------------------------------------------------------------------------------------
// KEM_DK = true : dk rows are the ML-KEM decapsulation keys (768k+96 bytes: ŝ ‖ ek ‖ H(ek) ‖ z; this kernel fills ŝ ‖ ek)
// KEM_DK = false: K-PKE.KeyGen on its own (ml_kem.c:651-769): dk rows are the 384k bytes of ŝ only
template <int K, int ETA1, bool KEM_DK>
__global__ void __launch_bounds__(WAVE * ARITH_WAVES, arith_minwaves(K))
k_keygen(size_t n, const uint16_t* __restrict__ A, const uint8_t* __restrict__ prf, const uint8_t* __restrict__ rho,
         uint8_t* __restrict__ ek, uint8_t* __restrict__ dk) {
    __shared__ ArithLds<K> lds_all[ARITH_WAVES];
    const int wv = wave_id(), l = lane_id();
    const size_t item = (size_t)blockIdx.x * ARITH_WAVES + wv;
    if (item >= n) return;
    ArithLds<K>& L = lds_all[wv];
    constexpr unsigned PS = (ETA1 == 3) ? 192 : 128, EK = 384 * K + 32, DK = KEM_DK ? 768 * K + 96 : 384 * K;
    const uint8_t* my_prf = prf + item * (size_t)(2 * K * PS);
    const uint16_t* my_A = A + item * (size_t)(K * K * 256);
    uint8_t* my_ek = ek + item * EK;
    uint8_t* my_dk = dk + item * DK;
    uint32_t raw_s[K], raw_e[K];
#pragma unroll
    for (int b = 0; b < K; b++) raw_s[b] = cbd_load<ETA1>(my_prf + b * PS);
    uint2 a_next[K];
#pragma unroll
    for (int b = 0; b < K; b++) a_next[b] = load_poly_raw(my_A + b * 256);
#pragma unroll
    for (int a = 0; a < K; a++) raw_e[a] = cbd_load<ETA1>(my_prf + (K + a) * PS);
    const uint32_t rho_w = reinterpret_cast<const uint32_t*>(rho + item * 32)[l & 7];
    NttTwiddlesF tw;
    load_twiddles_f(tw);
    float x[4];
    // s-hat (ml_kem.c:696-706), dk_pke = ByteEncode_12(s-hat) (ml_kem.c:750-756)
#pragma unroll
    for (int b = 0; b < K; b++) {
        cbd_eval_f<ETA1>(raw_s[b], x);
        wave_ntt_f(x, L.xch, tw);
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = fred(x[m]);
        stash_vhat_f(L.vhat[b], L.vgam[b], x, tw);
        emit_encode12(L.cbuf(), x, my_dk + 384 * b, nullptr);
    }
    wave_lds_fence();
    // t-hat[a] = sum_b A[a][b] o s-hat[b] + e-hat[a] (ml_kem.c:710-727), ek = ByteEncode_12(t-hat) || rho
#pragma unroll
    for (int a = 0; a < K; a++) {
        uint2 a_cur[K];
#pragma unroll
        for (int b = 0; b < K; b++) a_cur[b] = a_next[b];
        if (a + 1 < K) {
#pragma unroll
            for (int b = 0; b < K; b++) a_next[b] = load_poly_raw(my_A + ((a + 1) * K + b) * 256);
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < K; b++) {
            float av[4];
            poly_raw_to_f(a_cur[b], av);
            basemul_acc_f(acc, av, L.vhat[b], L.vgam[b]);
        }
        float e[4];
        cbd_eval_f<ETA1>(raw_e[a], e);
        wave_ntt_f(e, L.xch, tw);
#pragma unroll
        for (int m = 0; m < 4; m++) acc[m] += e[m];   // <= 1665 + 6660
        emit_encode12(L.cbuf(), acc, my_ek + 384 * a, KEM_DK ? my_dk + 384 * K + 384 * a : (uint8_t*)nullptr);
    }
    if (l < 8) {
        reinterpret_cast<uint32_t*>(my_ek + 384 * K)[l] = rho_w;
        if (KEM_DK) reinterpret_cast<uint32_t*>(my_dk + 768 * K)[l] = rho_w;
    }
}

// ================================================================================================
// stand-alone primitives (C-ABI: mlkem_ntt / mlkem_intt / mlkem_multiply_ntts / mlkem_sample_cbd)
// ================================================================================================
// (the stand-alone NTT / InverseNTT kernels live in mlkem_rntt.hpp: four polynomials per wave in registers)

// MultiplyNTTs (ml_kem.c:415-442): h = a o b, one polynomial pair per wave; inputs may be any 12-bit value
__global__ void __launch_bounds__(WAVE * ARITH_WAVES) k_basemul_batch(size_t n, const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                                      uint16_t* __restrict__ h) {
    __shared__ __attribute__((aligned(16))) float vh_all[ARITH_WAVES][256];
    __shared__ __attribute__((aligned(16))) float vg_all[ARITH_WAVES][128];
    const int wv = (int)(threadIdx.x >> 6);
    NttTwiddlesF tw;
    load_twiddles_f(tw);
    const size_t stride = (size_t)gridDim.x * ARITH_WAVES;
    for (size_t p = (size_t)blockIdx.x * ARITH_WAVES + wv; p < n; p += stride) {
        float av[4], bv[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
        load_poly_nat_f12(a + p * 256, av);
        load_poly_nat_f12(b + p * 256, bv);
#pragma unroll
        for (int m = 0; m < 4; m++) bv[m] = fred(bv[m]);
        stash_vhat_f(vh_all[wv], vg_all[wv], bv, tw);
        wave_lds_fence();
        basemul_acc_f(acc, av, vh_all[wv], vg_all[wv]);
        wave_lds_fence();
        int o[4];
#pragma unroll
        for (int m = 0; m < 4; m++) o[m] = fcanon(acc[m]);
        store_poly_nat(h + p * 256, o);
    }
}

// VectorMultiply (ml_kem.c:618-638): w = sum_{i<k} MultiplyNTTs(u[i], v[i]), every partial sum reduced like the
// reference's PolyAddition does (ml_kem.c:580-592) -- the k-term accumulate the K-PKE kernels run fused, as a stand-alone
// entry for row-level parity.  u, v : [n][k][256], any 12-bit values; w : [n][256] canonical.
__global__ void __launch_bounds__(WAVE * ARITH_WAVES) k_vecmul_batch(size_t n, int k, const uint16_t* __restrict__ u, const uint16_t* __restrict__ v,
                                                                     uint16_t* __restrict__ w) {
    __shared__ __attribute__((aligned(16))) float vh_all[ARITH_WAVES][256];
    __shared__ __attribute__((aligned(16))) float vg_all[ARITH_WAVES][128];
    const int wv = (int)(threadIdx.x >> 6);
    NttTwiddlesF tw;
    load_twiddles_f(tw);
    const size_t stride = (size_t)gridDim.x * ARITH_WAVES;
    for (size_t p = (size_t)blockIdx.x * ARITH_WAVES + wv; p < n; p += stride) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < k; i++) {
            float av[4], bv[4];
            load_poly_nat_f12(u + (p * (size_t)k + (size_t)i) * 256, av);
            load_poly_nat_f12(v + (p * (size_t)k + (size_t)i) * 256, bv);
#pragma unroll
            for (int m = 0; m < 4; m++) bv[m] = fred(bv[m]);
            stash_vhat_f(vh_all[wv], vg_all[wv], bv, tw);
            wave_lds_fence();
            basemul_acc_f(acc, av, vh_all[wv], vg_all[wv]);   // acc stays reduced: |acc| <= 1665
            wave_lds_fence();
        }
        int o[4];
#pragma unroll
        for (int m = 0; m < 4; m++) o[m] = fcanon(acc[m]);
        store_poly_nat(w + p * 256, o);
    }
}

// PolyAddition / PolySubtraction (ml_kem.c:580-592 / :599-613) coefficient by coefficient over n uint16 values, inputs
// taken mod 2^12 like the reference's `union integer.t`:
//   add: (u + v) % Q                         (u + v <= 8190: at most two subtractions of Q)
//   sub: u < v ? Q - (v - u) : u - v         stored into the 12-bit field, i.e. & 0xFFF (only differs for v - u > Q)
// Streaming kernel, 8 values (16 B) per lane and iteration; in-place allowed.
__device__ __forceinline__ unsigned poly_add1(unsigned a, unsigned b) {
    unsigned s = (a & 0xFFFu) + (b & 0xFFFu);
    s -= s >= (unsigned)KQ ? (unsigned)KQ : 0u;
    s -= s >= (unsigned)KQ ? (unsigned)KQ : 0u;
    return s;
}
__device__ __forceinline__ unsigned poly_sub1(unsigned a, unsigned b) {
    a &= 0xFFFu; b &= 0xFFFu;
    return (a < b ? (unsigned)KQ - (b - a) : a - b) & 0xFFFu;
}
template <bool SUB>
__global__ void __launch_bounds__(256) k_poly_addsub(size_t n, const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ out) {
    const size_t groups = n / 8, stride = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto f = [](unsigned x, unsigned y) { return SUB ? poly_sub1(x, y) : poly_add1(x, y); };
    auto f2 = [&f](uint32_t x, uint32_t y) { return f(x & 0xFFFFu, y & 0xFFFFu) | (f(x >> 16, y >> 16) << 16); };
    for (size_t g = t; g < groups; g += stride) {
        const uint4 x = reinterpret_cast<const uint4*>(a)[g], y = reinterpret_cast<const uint4*>(b)[g];
        uint4 o;
        o.x = f2(x.x, y.x); o.y = f2(x.y, y.y); o.z = f2(x.z, y.z); o.w = f2(x.w, y.w);
        reinterpret_cast<uint4*>(out)[g] = o;
    }
    for (size_t i = groups * 8 + t; i < n; i += stride) out[i] = (uint16_t)f(a[i], b[i]);
}

// SamplePolyCBD (ml_kem.c:253-275): bytes [n][64*eta] -> canonical uint16 polynomials
template <int ETA>
__global__ void __launch_bounds__(WAVE * ARITH_WAVES) k_cbd_batch(size_t n, const uint8_t* __restrict__ bytes, uint16_t* __restrict__ out) {
    const int wv = (int)(threadIdx.x >> 6);
    const size_t stride = (size_t)gridDim.x * ARITH_WAVES;
    for (size_t p = (size_t)blockIdx.x * ARITH_WAVES + wv; p < n; p += stride) {
        int x[4];
        cbd_nat<ETA>(bytes + p * 64 * ETA, x);
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] += (x[m] >> 31) & KQ;
        store_poly_nat(out + p * 256, x);
    }
}

// Compress_D + ByteEncode_D (ml_kem.c:83-97, :125-145; D = 12: ByteEncode_12 alone) and ByteDecode_D + Decompress_D
// (ml_kem.c:153-177, :104-119; D = 12: raw 12-bit values, no reduction -- F3) over n polynomials: the codec helpers the
// K-PKE kernels use, as stand-alone primitives for component parity against the reference's full Compress tables.
template <int D>
__global__ void __launch_bounds__(WAVE * ARITH_WAVES) k_encode_batch(size_t n, const uint16_t* __restrict__ f, uint8_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t cbuf_all[ARITH_WAVES][CODEC_BUF_WORDS];
    const int wv = (int)(threadIdx.x >> 6);
    uint32_t* cbuf = cbuf_all[wv];
    const size_t stride = (size_t)gridDim.x * ARITH_WAVES;
    for (size_t p = (size_t)blockIdx.x * ARITH_WAVES + wv; p < n; p += stride) {
        float x[4];
        load_poly_nat_f12(f + p * 256, x);
        if constexpr (D == 12) {
            emit_encode12(cbuf, x, out + p * 384, nullptr);
        } else {
            CodecRegs<D> unused{};
            emit_compressed<D, false>(cbuf, x, out + p * (32 * D), unused);
        }
    }
}
template <int D>
__global__ void __launch_bounds__(WAVE * ARITH_WAVES) k_decode_batch(size_t n, const uint8_t* __restrict__ in, uint16_t* __restrict__ f) {
    __shared__ __attribute__((aligned(16))) uint32_t cbuf_all[ARITH_WAVES][CODEC_BUF_WORDS];
    const int wv = (int)(threadIdx.x >> 6);
    uint32_t* cbuf = cbuf_all[wv];
    const size_t stride = (size_t)gridDim.x * ARITH_WAVES;
    for (size_t p = (size_t)blockIdx.x * ARITH_WAVES + wv; p < n; p += stride) {
        CodecRegs<D> r;
        codec_fetch<D>(in + p * (32 * D), r);
        float x[4];
        decode_regs<D, D != 12>(cbuf, r, x);
        int xi[4];
#pragma unroll
        for (int m = 0; m < 4; m++) xi[m] = (int)x[m];
        store_poly_nat(f + p * 256, xi);
    }
}

// Compress_d / Decompress_d (ml_kem.c:83-97 / :104-119) value by value for ANY d in 1..12 and ANY 12-bit input, with the
// reference's field widths: inputs are taken mod 2^12 (`union integer.t`), the dividend lives in 24 bits (it always fits),
// the rounded quotient wraps at 12 bits, d = 12 is the identity.  The K-PKE kernels only ever need d in {1, 4, 5, 10, 11}
// on reduced inputs (compress_f / decompress_d above); this is the general entry the reference's CompressDecompress test
// (d = 1..12) exercises.  Element-wise streaming kernel, 8 values (16 B) per lane and iteration.
__device__ __forceinline__ unsigned compress_any(unsigned x, int d) {
    x &= 0xFFFu;
    if (d >= 12) return x;
    const unsigned num = x << d, quo = div_q(num), rem = num - quo * (unsigned)KQ;   // num < 2^23
    return (quo + (rem > (unsigned)(KQ / 2) ? 1u : 0u)) & ((1u << d) - 1u);
}
__device__ __forceinline__ unsigned decompress_any(unsigned y, int d) {
    y &= 0xFFFu;
    if (d >= 12) return y;
    const unsigned t = (unsigned)KQ * y;                                            // < 2^24
    return ((t >> d) + ((t & ((1u << d) - 1u)) >= (1u << (d - 1)) ? 1u : 0u)) & 0xFFFu;
}
template <bool DECOMPRESS>
__global__ void __launch_bounds__(256) k_compress_values(size_t n, int d, const uint16_t* __restrict__ in, uint16_t* __restrict__ out) {
    const size_t groups = n / 8, stride = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto f = [d](unsigned v) { return DECOMPRESS ? decompress_any(v, d) : compress_any(v, d); };
    for (size_t g = t; g < groups; g += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(in)[g];
        uint4 o;
        o.x = f(v.x & 0xFFFFu) | (f(v.x >> 16) << 16); o.y = f(v.y & 0xFFFFu) | (f(v.y >> 16) << 16);
        o.z = f(v.z & 0xFFFFu) | (f(v.z >> 16) << 16); o.w = f(v.w & 0xFFFFu) | (f(v.w >> 16) << 16);
        reinterpret_cast<uint4*>(out)[g] = o;
    }
    for (size_t i = groups * 8 + t; i < n; i += stride) out[i] = (uint16_t)f(in[i]);
}

// ================================================================================================
// layout converters: the reference keeps every "byte" in a 4-byte `union byte` cell (value in bits
// 0-7, upper 24 bits undefined: ml_kem.h:35-38).  Pure streaming kernels, 16 cells (64 B in / 16 B out, or
// the reverse) per lane and iteration; the n % 16 tail is handled cell by cell.
// ================================================================================================
__global__ void __launch_bounds__(256) k_cells_to_bytes(size_t n, const uint32_t* __restrict__ cells, uint8_t* __restrict__ bytes) {
    const size_t groups = n / 16, stride = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t g = t; g < groups; g += stride) {
        const uint4* src = reinterpret_cast<const uint4*>(cells) + 4 * g;
        const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
        uint4 o;
        o.x = (a.x & 0xFFu) | ((a.y & 0xFFu) << 8) | ((a.z & 0xFFu) << 16) | (a.w << 24);
        o.y = (b.x & 0xFFu) | ((b.y & 0xFFu) << 8) | ((b.z & 0xFFu) << 16) | (b.w << 24);
        o.z = (c.x & 0xFFu) | ((c.y & 0xFFu) << 8) | ((c.z & 0xFFu) << 16) | (c.w << 24);
        o.w = (d.x & 0xFFu) | ((d.y & 0xFFu) << 8) | ((d.z & 0xFFu) << 16) | (d.w << 24);
        reinterpret_cast<uint4*>(bytes)[g] = o;
    }
    for (size_t i = groups * 16 + t; i < n; i += stride) bytes[i] = (uint8_t)cells[i];
}
__global__ void __launch_bounds__(256) k_bytes_to_cells(size_t n, const uint8_t* __restrict__ bytes, uint32_t* __restrict__ cells) {
    const size_t groups = n / 16, stride = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t g = t; g < groups; g += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(bytes)[g];
        uint4* dst = reinterpret_cast<uint4*>(cells) + 4 * g;
        uint4 o;
        o.x = v.x & 0xFFu; o.y = (v.x >> 8) & 0xFFu; o.z = (v.x >> 16) & 0xFFu; o.w = v.x >> 24; dst[0] = o;
        o.x = v.y & 0xFFu; o.y = (v.y >> 8) & 0xFFu; o.z = (v.y >> 16) & 0xFFu; o.w = v.y >> 24; dst[1] = o;
        o.x = v.z & 0xFFu; o.y = (v.z >> 8) & 0xFFu; o.z = (v.z >> 16) & 0xFFu; o.w = v.z >> 24; dst[2] = o;
        o.x = v.w & 0xFFu; o.y = (v.w >> 8) & 0xFFu; o.z = (v.w >> 16) & 0xFFu; o.w = v.w >> 24; dst[3] = o;
    }
    for (size_t i = groups * 16 + t; i < n; i += stride) cells[i] = bytes[i];
}

}   // namespace mlkem
