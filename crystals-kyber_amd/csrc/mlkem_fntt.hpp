// mlkem_fntt.hpp — EXACT mod-3329 arithmetic on the fp32 pipe: the reduction, twiddle product, canonicalisation, Compress and
// base-case-multiply helpers every transform and K-PKE kernel is built from (ml_kem.c:83-97, :287-442), and the zeta table.
//
// Why fp32: on gfx950 every integer multiply (v_mul_*_i24/u24, v_mul_lo_u32, v_mad_*24), shift and bit-field op
// issues at ~0.55x the rate of v_fma_f32 / v_mul_f32 / v_add_f32 (tools/valu_ubench2.hip, profiles/r01_valu_ubench.txt).
// All values are integers of magnitude < 2^24, for which fp32 add / mul / fma are exact, so the results are
// bit-identical to the reference's `% Q` arithmetic:
//   product  p = zeta * b           exact while |p| <= 2^24  (|zeta| <= 1664 centred, |b| <= 10082)
//   Barrett  k = rint(p / q)        via  k = fma(p, 1/q, 1.5*2^23) - 1.5*2^23   (round-to-nearest integer)
//            r = fma(k, -q, p)      exact, |r| <= 1665
// i.e. the reduction IS Barrett's (multiply by a precomputed reciprocal, round, subtract), evaluated in fp32.
// Lazy bounds are tracked in the comments; `fred` is applied exactly where a bound would exceed 10082 (= 2^24/1664).
#pragma once
#include "mlkem_device.hpp"

namespace mlkem {

constexpr float F_Q = 3329.0f;
constexpr float F_INVQ = 1.0f / 3329.0f;
constexpr float F_MAGIC = 12582912.0f;   // 1.5 * 2^23: adding and subtracting rounds to the nearest integer

// centred representative of x mod q, |result| <= 1665, for integer |x| <= 2^24
__device__ __forceinline__ float fred(float x) {
    const float k = __builtin_fmaf(x, F_INVQ, F_MAGIC) - F_MAGIC;
    return __builtin_fmaf(k, -F_Q, x);
}
// zeta * b mod q (centred), |zeta * b| <= 2^24
__device__ __forceinline__ float fmulmod(float zeta, float b) { return fred(zeta * b); }
// The same for a CONSTANT multiplier whose quotient zq = fl(zeta / q) is precomputed (Shoup's trick on the fp32 pipe):
//   kM  = fma(b, zq, M)          = M + k,  k = rint(b zeta / q) up to the usual off-by-one near .5   (exact: |k| < 2^22)
//   nkq = fma(kM, -q, M q)       = -k q    exact: M q = 9987 * 2^22 is a float, |k q| <= 2^24 (k q is even if it is larger)
//   t   = fma(b, zeta, nkq)      = b zeta - k q,  |t| <= 1665
// Three dependent FMAs instead of mul + 3 (fred): one VALU instruction less per butterfly (fmulmod_shoup).
constexpr float F_MAGIC_Q = 12582912.0f * 3329.0f;   // exactly representable (9987 * 2^22)
struct Tw {
    float z, zq;   // zeta (centred) and fl(zeta / q)
};
__device__ __forceinline__ float fmulmod_shoup(Tw w, float b) {
    const float km = __builtin_fmaf(b, w.zq, F_MAGIC);
    const float nkq = __builtin_fmaf(km, -F_Q, F_MAGIC_Q);
    return __builtin_fmaf(b, w.z, nkq);
}
// mul + Barrett form of the same product (4 instead of 3 operations, no quotient half needed); the transforms use the 3-FMA
// form, the stand-alone base-case products this one.
__device__ __forceinline__ float fmulmod(Tw w, float b) { return fred(w.z * b); }
constexpr Tw tw_const(int zeta_centred) { return Tw{(float)zeta_centred, (float)((double)zeta_centred / 3329.0)}; }
__device__ __forceinline__ Tw tw_neg(Tw w) { return Tw{-w.z, -w.zq}; }
// canonical representative in [0, q) as an integer, for |x| <= 2^24
__device__ __forceinline__ int fcanon(float x) {
    float r = fred(x);
    r = r < 0.0f ? r + F_Q : r;
    return (int)r;
}

// canonical representative in [0, q) as a float, for integer |x| <= 2^20, in three full-rate instructions:
//   k = floor(x fl(1/q) + 2^-13) = floor(x / q) exactly: the bias exceeds the evaluation error (<= 2 * 315 * 2^-24 < 4e-5)
//   and is smaller than the distance 1/q = 3.0e-4 of any non-multiple's quotient from the next integer
//   r = x - k q                  exact
__device__ __forceinline__ float fcanon_floor(float x) {
    const float k = __builtin_floorf(__builtin_fmaf(x, F_INVQ, 0.0001220703125f));
    return __builtin_fmaf(k, -F_Q, x);
}

// Compress_d (ml_kem.c:83-97) of ANY representative x (integer, |x| <= 4095) without canonicalising it first:
// round(2^d x / q) mod 2^d is invariant under x -> x + q, and q is odd, so there are no ties.
//   t  = 2^d x                         exact (|t| < 2^23.1)
//   k0 = rint(t fl(1/q))               off by one only when frac(t/q) is within 1.2e-4 of 1/2
//   r  = t - k0 q                      exact; |r| <= 1665, and |r| = 1665 exactly in the off-by-one case
//   k  = k0 + rint(r fl(1/q))          rint(+-0.50015) = +-1 repairs it; otherwise the term is 0
// Seven full-rate fp32 instructions + one conversion + one AND, against canonicalise (5) + convert + shift/add + 32-bit
// mul_hi + shift + AND (five of them slow-class) in the integer form.  Verified exhaustively (tests/emu: every
// |x| <= 4095, d in {1, 4, 5, 10, 11}, against the integer form on the canonical representative).
template <int D>
__device__ __forceinline__ unsigned compress_f(float x) {
    static_assert(D >= 1 && D <= 11, "d");
    const float t = x * (float)(1 << D);
    const float k0 = __builtin_fmaf(t, F_INVQ, F_MAGIC) - F_MAGIC;
    const float r = __builtin_fmaf(k0, -F_Q, t);
    const float k = k0 + (__builtin_fmaf(r, F_INVQ, F_MAGIC) - F_MAGIC);
    return (unsigned)(int)k & ((1u << D) - 1u);
}

struct ZetaTableF {
    Tw z[128];
    constexpr ZetaTableF() : z{} {
        for (int i = 0; i < 128; i++) z[i] = tw_const(cx_centered(cx_pow17(cx_bitrev7(i))));
    }
};
__device__ const ZetaTableF ZETA_F = ZetaTableF();   // zeta_i = 17^BitRev7(i) mod q, centred (ml_kem.c:300-307)

constexpr Tw FZ1 = tw_const(cx_centered(cx_pow17(cx_bitrev7(1))));
constexpr Tw FZ2 = tw_const(cx_centered(cx_pow17(cx_bitrev7(2))));
constexpr Tw FZ3 = tw_const(cx_centered(cx_pow17(cx_bitrev7(3))));
constexpr Tw F_INV128 = tw_const(INV128 - KQ);   // 128^-1 = 3303 = -26 mod q (ml_kem.c:378-381)
constexpr Tw F_INV128_Z1 = tw_const(cx_centered((INV128 * cx_pow17(cx_bitrev7(1))) % KQ));   // zeta_1 / 128 mod q

// 4 consecutive coefficients of lane l as one 16-byte LDS access
__device__ __forceinline__ void fx_write4(float* buf, int slot, const float (&x)[4]) {
    float4 v;
    v.x = x[0]; v.y = x[1]; v.z = x[2]; v.w = x[3];
    *reinterpret_cast<float4*>(buf + slot) = v;
}
__device__ __forceinline__ void fx_read4(const float* buf, int slot, float (&x)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(buf + slot);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
}

// ---- base-case multiply-accumulate (ml_kem.c:395-442, :618-638) ------------------------------------------------
// An NTT-domain polynomial v (reduced, |v| <= 1665) is kept in LDS together with its odd coefficients times gamma
// (gamma_{2l} = zeta_{64+l}, gamma_{2l+1} = -gamma_{2l}).
__device__ __forceinline__ void stash_vhat_f(float* vh, float* vg, const float (&x)[4], const Tw fD) {
    const int l = lane_id();
    fx_write4(vh, 4 * l, x);
    float2 g;
    g.x = fmulmod(fD, x[1]);
    g.y = fmulmod(tw_neg(fD), x[3]);
    *reinterpret_cast<float2*>(vg + 2 * l) = g;
}
// acc = (acc + a o v) reduced : for 0 <= a <= 4095 (raw 12-bit, F3) and |v|, |acc| <= 1665 the exact sum is
// <= 1665 + 2 * 4095 * 1665 < 2^24, so the fp32 evaluation is exact and one `fred` per coefficient suffices.
__device__ __forceinline__ float basemul_term(float acc, float a0, float y0, float a1, float y1) {
    return fred(__builtin_fmaf(a1, y1, __builtin_fmaf(a0, y0, acc)));   // swept over its whole domain on the device (mlkem_selftest.hpp)
}
__device__ __forceinline__ void basemul_acc_f(float (&acc)[4], const float (&a)[4], const float* vh, const float* vg) {
    const int l = lane_id();
    float y[4];
    fx_read4(vh, 4 * l, y);
    const float2 g = *reinterpret_cast<const float2*>(vg + 2 * l);
    acc[0] = basemul_term(acc[0], a[0], y[0], a[1], g.x);
    acc[1] = basemul_term(acc[1], a[0], y[1], a[1], y[0]);
    acc[2] = basemul_term(acc[2], a[2], y[2], a[3], g.y);
    acc[3] = basemul_term(acc[3], a[2], y[3], a[3], y[2]);
}

// uint16 polynomial in HBM (natural order) -> 4 floats per lane (values 0..65535 -> exact)
__device__ __forceinline__ void load_poly_nat_f(const uint16_t* p, float (&x)[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + 4 * lane_id());
    x[0] = (float)(v.x & 0xFFFFu); x[1] = (float)(v.x >> 16);
    x[2] = (float)(v.y & 0xFFFFu); x[3] = (float)(v.y >> 16);
}

// same, taking the inputs mod 2^12 like the reference's 12-bit `union integer` fields
__device__ __forceinline__ void load_poly_nat_f12(const uint16_t* p, float (&x)[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + 4 * lane_id());
    x[0] = (float)(v.x & 0xFFFu); x[1] = (float)((v.x >> 16) & 0xFFFu);
    x[2] = (float)(v.y & 0xFFFu); x[3] = (float)((v.y >> 16) & 0xFFFu);
}

}   // namespace mlkem
