// mlkem_kpke4.hpp — K-PKE.Decrypt (ml_kem.c:942-1023) in the FOUR-ITEMS-PER-WAVE register form of mlkem_rntt.hpp:
// one item per 16-lane row, 16 coefficients of a polynomial per lane, no LDS at all.  The transforms cost what they cost
// in the one-item-per-wave LDS form K-PKE.Encrypt / KeyGen still use (mlkem_arith.hpp), but everything around them — byte
// codecs, address arithmetic, per-wave set-up, exchange fences — is shared by four items or disappears: 681 instead of 810
// VALU instructions per item at k = 3 (520 / 642 at k = 2, 858 / 979 at k = 4), 98-128 VGPRs = 4 waves per SIMD.
// Measured against the LDS form it replaced, alternating runs on one box: k_decrypt -4.8 % (768), -8.7 % (512), -11 % (1024).
//
// Layout (mlkem_rntt.hpp): lane = 16 p + code(m); row p works on item 4 q + p of the wave's quad q; the lane holds
// lo[r] = coefficient 8 m + r and hi[r] = coefficient 128 + 8 m + r of whatever polynomial is being processed.  In a
// ByteEncode_d polynomial (32 d bytes) those are the d bytes at byte d m and the d bytes at byte 16 d + d m: a "piece".
#pragma once
#include "mlkem_rntt.hpp"

namespace mlkem {

#ifndef MLKEM_KPKE4_WAVES
#define MLKEM_KPKE4_WAVES 4
#endif
constexpr int KPKE4_WAVES = MLKEM_KPKE4_WAVES;
// waves per SIMD the register allocator aims at: 4 (128 VGPRs) for k = 2, 3; the packed-fp32 form of k = 4 needs 162 VGPRs
// and would spill 160 B per lane at 128 (measured: 1.87 instead of 1.39 ms per 2^20 items), so it is compiled for 3 waves
#ifdef MLKEM_KPKE4_MINWAVES
constexpr int kpke4_minwaves(int) { return MLKEM_KPKE4_MINWAVES; }
#else
constexpr int kpke4_minwaves(int k) { return k == 4 ? 3 : 4; }
#endif

// low 32 bits of {hi, lo} >> s, 0 <= s < 32 (v_alignbit_b32)
__device__ __forceinline__ uint32_t funnel_shr(uint32_t hi, uint32_t lo, unsigned s) {
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u));
}

// dwords that cover a piece from the dword boundary at or below its first byte (pieces of d = 4, 12 start on one)
template <int D>
struct PieceCfg {
    static_assert(D == 4 || D == 5 || D == 10 || D == 11 || D == 12, "ciphertext / key encodings of ML-KEM");
    static constexpr int NW = D == 4 ? 1 : D == 5 ? 2 : D == 11 ? 4 : 3;
    static constexpr bool ALIGNED = D == 4 || D == 12;
};
template <int D>
struct Piece {
    uint32_t w[PieceCfg<D>::NW];
    unsigned shift;   // bits between the dword boundary and the piece (0, 8, 16, 24)
};
// `poly`: the 32 D bytes of one encoded polynomial, dword-aligned
template <int D>
__device__ __forceinline__ void piece_load(const uint8_t* poly, int m, int half, Piece<D>& pc) {
    const unsigned B = (unsigned)(D * (16 * half + m)), off = PieceCfg<D>::ALIGNED ? 0u : (B & 3u);
    pc.shift = 8u * off;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(poly + (B - off));
#pragma unroll
    for (int i = 0; i < PieceCfg<D>::NW; i++) {
        if (i < 3 || off >= 2u) pc.w[i] = stream_load4(p + i);   // d = 11: the 4th dword only where the piece reaches into it
        else pc.w[i] = 0u;                                       //         (the last piece of a polynomial must not read past it)
    }
}
// the 8 D-bit fields of a piece (ml_kem.c:153-177: little-endian bit order; d = 12 is NOT reduced mod q, F3)
template <int D>
__device__ __forceinline__ void piece_fields(const Piece<D>& pc, unsigned (&c)[8]) {
    constexpr int NW = PieceCfg<D>::NW;
    uint32_t v[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if constexpr (PieceCfg<D>::ALIGNED) v[i] = pc.w[i];
        else v[i] = (i + 1 < NW) ? funnel_shr(pc.w[i + 1], pc.w[i], pc.shift) : (pc.w[i] >> pc.shift);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        constexpr unsigned mask = (1u << D) - 1u;
        const int bit = D * j, q = bit >> 5, r = bit & 31;
        c[j] = ((r + D <= 32) ? (v[q] >> r) : funnel_shr(v[q + 1 < NW ? q + 1 : q], v[q], (unsigned)r)) & mask;
    }
}
template <int D, bool DECOMPRESS>
__device__ __forceinline__ void piece_to_f(const Piece<D>& pc, float (&x)[8]) {
    unsigned c[8];
    piece_fields<D>(pc, c);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if constexpr (DECOMPRESS) x[j] = (float)decompress_d<D>(c[j]);
        else x[j] = (float)c[j];
    }
}

// acc += a o y over the lane's 8 coefficient pairs (ml_kem.c:395-442): a raw 12-bit, y reduced, g[j] = gamma_j * y[2j+1]
__device__ __forceinline__ void basemul_acc8(float (&acc)[8], const float (&a)[8], const float (&y)[8], const float (&g)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        acc[2 * j] = basemul_term(acc[2 * j], a[2 * j], y[2 * j], a[2 * j + 1], g[j]);
        acc[2 * j + 1] = basemul_term(acc[2 * j + 1], a[2 * j], y[2 * j + 1], a[2 * j + 1], y[2 * j]);
    }
}
// gamma of the pair (coefficients 2i, 2i+1) is +zeta_{64 + i/2} for even i and -zeta_{64 + i/2} for odd i (ml_kem.c:426-433);
// for this lane's pairs j = 0..3 of half v that is +-z1[v][j >> 1] of the forward twiddle set (mlkem_rntt.hpp)
__device__ __forceinline__ void gamma_products(const float (&y)[8], const Tw z10, const Tw z11, float (&g)[4]) {
    g[0] = fmulmod_shoup(z10, y[1]);
    g[1] = fmulmod_shoup(tw_neg(z10), y[3]);
    g[2] = fmulmod_shoup(z11, y[5]);
    g[3] = fmulmod_shoup(tw_neg(z11), y[7]);
}

// ------------------------------------------------------------------------------------------------
// k_decrypt4 — m' = ByteEncode_1(Compress_1(v - InverseNTT(s-hat . NTT(u)))) for four items per wave
// ------------------------------------------------------------------------------------------------
// decrypt4_body: the work of one wave = items 4 quad .. 4 quad + 3 of the n the pointers describe
template <int K, int DU, int DV>
__device__ __forceinline__ void decrypt4_body(size_t quad, size_t n, const uint8_t* __restrict__ dk, size_t dk_stride,
                                              const uint8_t* __restrict__ c, uint8_t* __restrict__ m_out) {
    const RnttLane a = rntt_lane();
    // rows beyond n (n % 4 != 0) redo item n - 1 and store nothing
    const size_t item_raw = 4 * quad + (size_t)a.p, item = item_raw < n ? item_raw : n - 1;
    constexpr unsigned CLEN = 32 * (DU * K + DV);
    const uint8_t* my_c = c + item * CLEN;
    const uint8_t* my_dk = dk + item * dk_stride;

    Piece<DU> u_lo, u_hi;
    piece_load<DU>(my_c, a.m, 0, u_lo); piece_load<DU>(my_c, a.m, 1, u_hi);
    RnttTw tw;
    rntt_load_twiddles_fwd(tw);

    float acc_lo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc_hi[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < K; b++) {
        float lo[8], hi[8];
        piece_to_f<DU, true>(u_lo, lo);    // u = Decompress_du(ByteDecode_du(c1))   (ml_kem.c:978-987)
        piece_to_f<DU, true>(u_hi, hi);
        // in flight during the transform: this polynomial's s-hat pieces (still packed: 7 registers, not 16) and the next u
        Piece<12> s_lo, s_hi;
        piece_load<12>(my_dk + b * 384, a.m, 0, s_lo); piece_load<12>(my_dk + b * 384, a.m, 1, s_hi);
        if (b + 1 < K) { piece_load<DU>(my_c + (b + 1) * 32 * DU, a.m, 0, u_lo); piece_load<DU>(my_c + (b + 1) * 32 * DU, a.m, 1, u_hi); }
        wave4_ntt_r(lo, hi, tw);
#pragma unroll
        for (int r = 0; r < 8; r++) { lo[r] = fred(lo[r]); hi[r] = fred(hi[r]); }
        float g[4], sv[8];
        gamma_products(lo, tw.z1[0][0], tw.z1[0][1], g);
        piece_to_f<12, false>(s_lo, sv);   // s-hat = ByteDecode_12(dk_pke), raw (ml_kem.c:996-998, F3)
        basemul_acc8(acc_lo, sv, lo, g);
        gamma_products(hi, tw.z1[1][0], tw.z1[1][1], g);
        piece_to_f<12, false>(s_hi, sv);
        basemul_acc8(acc_hi, sv, hi, g);
    }
    Piece<DV> v_lo, v_hi;                  // in flight during the inverse transform
    piece_load<DV>(my_c + K * 32 * DU, a.m, 0, v_lo); piece_load<DV>(my_c + K * 32 * DU, a.m, 1, v_hi);
    rntt_load_twiddles_inv(tw);            // the forward set is dead: its registers take the inverse set
    wave4_intt_r(acc_lo, acc_hi, tw);
    float vl[8], vh[8];
    piece_to_f<DV, true>(v_lo, vl);        // v = Decompress_dv(ByteDecode_dv(c2))   (ml_kem.c:990-993)
    piece_to_f<DV, true>(v_hi, vh);
    unsigned b_lo = 0, b_hi = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {          // ml_kem.c:1003-1011
        b_lo |= compress_f<1>(vl[r] - acc_lo[r]) << r;
        b_hi |= compress_f<1>(vh[r] - acc_hi[r]) << r;
    }
    // ByteEncode_1: coefficients 8 m .. 8 m + 7 are byte m of the message, 128 + 8 m .. byte 16 + m
    if (item_raw < n) {
        m_out[item * 32 + (size_t)a.m] = (uint8_t)b_lo;
        m_out[item * 32 + 16 + (size_t)a.m] = (uint8_t)b_hi;
    }
}
template <int K, int DU, int DV>
__global__ void __launch_bounds__(64 * KPKE4_WAVES, kpke4_minwaves(K)) k_decrypt4(size_t n, const uint8_t* __restrict__ dk, size_t dk_stride,
                                                                const uint8_t* __restrict__ c, uint8_t* __restrict__ m_out) {
    const size_t quad = (size_t)blockIdx.x * KPKE4_WAVES + wave_id();
    if (4 * quad >= n) return;
    decrypt4_body<K, DU, DV>(quad, n, dk, dk_stride, c, m_out);
}

}   // namespace mlkem
