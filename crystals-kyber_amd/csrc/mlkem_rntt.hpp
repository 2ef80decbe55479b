// mlkem_rntt.hpp — NTT / InverseNTT (ml_kem.c:287-384) of FOUR polynomials per wavefront held entirely in registers:
// 16 coefficients per lane, one polynomial per 16-lane row, no LDS.  This is the form the stand-alone batch kernels use
// (BASELINE configs[1]): HBM is read and written as 16 bytes per lane, fully coalesced, and the seven butterfly layers
// cost VALU issue slots only.
//
// Index bits.  A polynomial's coefficient index is idx[7:0]; the seven layers act on idx7 (len 128) ... idx1 (len 2), idx0
// is never a butterfly bit (the transform stops at degree-1 residues).  Lane l = 16 p + (l3 l2 l1 l0) works on polynomial
// p of the wave's four and holds 16 registers x[r3 r2 r1 r0]:
//
//   registers  (r3, r2, r1, r0) = (idx7, idx2, idx1, idx0)   two 16-byte loads: coefficients 8m .. 8m+7 and 128+8m .. 128+8m+7
//   row lanes  m = idx[6:3] with  idx6 = l3,  idx5 = l2,  idx4 = l1 ^ l2,  idx3 = l0 ^ l2
//
// A layer whose bit is a REGISTER bit (idx7, idx2, idx1) is an ordinary register butterfly.  A layer whose bit is a LANE
// bit is a cross-lane butterfly through DPP: with the lane code above one DPP modifier reaches the butterfly partner —
// flipping idx3 alone is lane ^ 1 (quad_perm [1,0,3,2]), idx4 is lane ^ 2 (quad_perm [2,3,0,1]), idx5 is lane ^ 7
// (row_half_mirror) and idx6 is lane ^ 8 (row_ror:8).  The lane code only permutes which 16-byte piece of its polynomial
// a lane loads: the 16 lanes of a row still cover 256 contiguous bytes per load instruction.
//
// Cross-lane Cooley-Tukey layer (a' = a + zeta b, b' = a - zeta b; lower lane holds a, upper lane b):
//   w  = x * ze  mod q      ze = 1 in the lower lane (w = a reduced), ze = -zeta in the upper lane (w = -zeta b)
//   x' = w + c * w(partner) c = -1 in the lower lane (a + zeta b),    c = +1 in the upper lane (a - zeta b)
// i.e. one 3-FMA modular product (fmulmod_shoup) and one v_fmac_f32_dpp per coefficient, and EVERY value comes out
// reduced (|x'| <= 2 * 1668), so the forward transform needs no separate reductions.
// Cross-lane Gentleman-Sande layer (a' = a + b, b' = zeta (b - a)):
//   d  = x + c * x(partner) c = +1 lower (a + b), c = -1 upper (b - a)
//   x' = d * ze  mod q      ze = 1 lower (reduction), ze = zeta upper
//
// Why this shape (measured on MI355X, tools/valu_ubench3.hip / tools/ntt_ubench.hip, profiles/r02_*): the transform is
// VALU-bound, not HBM-bound (a copy with the same loop runs at 6.4-6.7 TB/s, the transform's arithmetic alone takes longer
// than its memory traffic); a DPP-modified VALU instruction costs 1.6 plain ones and a v_permlane32/16_swap 3.2, while an LDS
// exchange costs a ds_write + ds_read per register and LDS bandwidth.  Four polynomials per wave put four butterfly bits
// inside a row, where DPP reaches them, and need no lane-to-register transposition at all.
//
// Exactness: all values are integers; a product zeta * b is exact in the FMA while |zeta b| <= 2^24, i.e. |b| <= 10082
// for |zeta| <= 1664 (no bound for zeta = 1); bounds are tracked in the comments below for inputs up to 4095 (raw 12-bit
// values, as the reference's `union integer.t` admits).  The 3-FMA product leaves |t| <= 1668.  Swept exhaustively on the
// device by mlkem_selftest (sweeps 1 and 7) and through the emulator in the CPU tier.
#pragma once
#include "mlkem_fntt.hpp"

namespace mlkem {

constexpr int XL_IDX3 = 1, XL_IDX4 = 2, XL_IDX5 = 7, XL_IDX6 = 8;   // lane xor that flips idx3 / idx4 / idx5 / idx6 alone

#ifdef MLKEM_EMU
__device__ __forceinline__ float emu_shfl_f(float v, int src) {
    union { float f; int i; } a, b;
    a.f = v;
    b.i = __shfl(a.i, src);
    return b.f;
}
template <int XOR>
__device__ __forceinline__ void xlane_fmac8(float (&v)[8], float c) {
    const int l = lane_id();
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(emu_shfl_f(v[i], l ^ XOR), c, v[i]);
}
#else
// v[i] += v[i](lane ^ XOR) * c for 8 registers: eight v_fmac_f32_dpp (the DPP operand is the accumulator register itself,
// as in the classic wave-reduction idiom).  A VALU write of a VGPR followed by a DPP read of it needs two wait states,
// which the compiler cannot see inside inline assembly: the block opens with s_nop 1.
#define MLKEM_XL8(CTRL)                                                                       \
    asm volatile("s_nop 1\n\t"                                                                \
                 "v_fmac_f32_dpp %0, %0, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
                 "v_fmac_f32_dpp %1, %1, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
                 "v_fmac_f32_dpp %2, %2, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
                 "v_fmac_f32_dpp %3, %3, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
                 "v_fmac_f32_dpp %4, %4, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
                 "v_fmac_f32_dpp %5, %5, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
                 "v_fmac_f32_dpp %6, %6, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
                 "v_fmac_f32_dpp %7, %7, %8 " CTRL " row_mask:0xf bank_mask:0xf"               \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])   \
                 : "v"(c))
template <int XOR>
__device__ __forceinline__ void xlane_fmac8(float (&v)[8], float c) {
    static_assert(XOR == 1 || XOR == 2 || XOR == 7 || XOR == 8, "lane xor reachable by one DPP modifier inside a row");
    if constexpr (XOR == 1) MLKEM_XL8("quad_perm:[1,0,3,2]");
    else if constexpr (XOR == 2) MLKEM_XL8("quad_perm:[2,3,0,1]");
    else if constexpr (XOR == 7) MLKEM_XL8("row_half_mirror");
    else MLKEM_XL8("row_ror:8");
}
#undef MLKEM_XL8
#endif

// index bits of this lane's two 16-byte pieces
struct RnttLane {
    int p, m, idx3, idx4, idx5, idx6;   // polynomial of the wave's four; block m = idx[6:3]
};
__device__ __forceinline__ RnttLane rntt_lane() {
    const int l = lane_id();
    RnttLane a;
    a.p = l >> 4;
    a.idx6 = (l >> 3) & 1;
    a.idx5 = (l >> 2) & 1;
    a.idx4 = ((l >> 1) ^ a.idx5) & 1;
    a.idx3 = (l ^ a.idx5) & 1;
    a.m = (a.idx6 << 3) | (a.idx5 << 2) | (a.idx4 << 1) | a.idx3;
    return a;
}

// Twiddles of one lane.  v = idx7 = register bit r3 (lo[] / hi[]).  Kept in registers: moving the 14 pairs to an LDS table
// read one layer ahead (ds_read_b64 per use) was measured 15 % SLOWER (tools/ntt_ubench.hip, profiles/r02_ntt_design.txt):
// the scheduler keeps as many values live either way, and the LDS reads add latency in front of every layer.
struct RnttTw {
    Tw e6[2], e5[2], e4[2], e3[2];   // cross-lane layers: effective multiplier (1 in the lower lane, -+zeta in the upper)
    float c6, c5, c4, c3;            // +-1: sign of the partner's contribution
    Tw z2[2], z1[2][2];              // register layers len 4 and len 2 ([idx7][idx2])
};
constexpr Tw TW_ONE = Tw{1.0f, F_INVQ};

// FORWARD: the zeta index of layer `len` is 128/len + idx[7 : log2(2 len)], i.e. 1, 2+idx7, 4+idx[7:6], 8+idx[7:5],
// 16+idx[7:4], 32+idx[7:3], 64+idx[7:2] (ml_kem.c:296-324)
__device__ __forceinline__ void rntt_load_twiddles_fwd(RnttTw& t) {
    const RnttLane a = rntt_lane();
    const Tw* z = ZETA_F.z;
#pragma unroll
    for (int v = 0; v < 2; v++) {
        const Tw w6 = z[2 + v], w5 = z[4 + 2 * v + a.idx6], w4 = z[8 + 4 * v + 2 * a.idx6 + a.idx5],
                 w3 = z[16 + 8 * v + 4 * a.idx6 + 2 * a.idx5 + a.idx4];
        t.e6[v] = a.idx6 ? tw_neg(w6) : TW_ONE;
        t.e5[v] = a.idx5 ? tw_neg(w5) : TW_ONE;
        t.e4[v] = a.idx4 ? tw_neg(w4) : TW_ONE;
        t.e3[v] = a.idx3 ? tw_neg(w3) : TW_ONE;
        t.z2[v] = z[32 + 16 * v + a.m];
        t.z1[v][0] = z[64 + 32 * v + 2 * a.m];
        t.z1[v][1] = z[65 + 32 * v + 2 * a.m];
    }
    t.c6 = a.idx6 ? 1.0f : -1.0f; t.c5 = a.idx5 ? 1.0f : -1.0f; t.c4 = a.idx4 ? 1.0f : -1.0f; t.c3 = a.idx3 ? 1.0f : -1.0f;
}
// INVERSE: the same table walked backwards, 127 - idx[7:2], 63 - idx[7:3], ... , 3 - idx7, 1 (ml_kem.c:345-373)
__device__ __forceinline__ void rntt_load_twiddles_inv(RnttTw& t) {
    const RnttLane a = rntt_lane();
    const Tw* z = ZETA_F.z;
#pragma unroll
    for (int v = 0; v < 2; v++) {
        const Tw w6 = z[3 - v], w5 = z[7 - (2 * v + a.idx6)], w4 = z[15 - (4 * v + 2 * a.idx6 + a.idx5)],
                 w3 = z[31 - (8 * v + 4 * a.idx6 + 2 * a.idx5 + a.idx4)];
        t.e6[v] = a.idx6 ? w6 : TW_ONE;
        t.e5[v] = a.idx5 ? w5 : TW_ONE;
        t.e4[v] = a.idx4 ? w4 : TW_ONE;
        t.e3[v] = a.idx3 ? w3 : TW_ONE;
        t.z2[v] = z[63 - (16 * v + a.m)];
        t.z1[v][0] = z[127 - (32 * v + 2 * a.m)];
        t.z1[v][1] = z[126 - (32 * v + 2 * a.m)];
    }
    t.c6 = a.idx6 ? -1.0f : 1.0f; t.c5 = a.idx5 ? -1.0f : 1.0f; t.c4 = a.idx4 ? -1.0f : 1.0f; t.c3 = a.idx3 ? -1.0f : 1.0f;
}

// ---- packed fp32 ---------------------------------------------------------------------------------------------------------
// idx0 is never a butterfly bit, so the coefficients 2i and 2i + 1 of a lane go through identical operations with identical
// twiddles in every layer: modular products and register butterflies work on float2 pairs (v_pk_fma_f32 / v_pk_add_f32)
// without any shuffle; only the DPP accumulate of the cross-lane layers has no packed form.  Against the scalar form of
// round 2 (profiles/r03_kpke_experiments.txt, one box): k_ntt4_batch 614 -> 492 / 656 -> 525 VALU instructions and
// -4.9 % / -5.5 % kernel time, k_decrypt4<3> 2723 -> 2180 and -9.4 %.  The library is built with -fno-slp-vectorize: where
// pairs have to be FORMED (the LDS transforms of mlkem_fntt.hpp) packing costs more than it saves.
#ifdef MLKEM_EMU
struct v2f { float x, y; };
__device__ __forceinline__ v2f operator+(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ v2f operator-(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return v2f{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
#else
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#endif
__device__ __forceinline__ v2f splat2(float s) { return v2f{s, s}; }
// zeta * b mod q on both halves: fmulmod_shoup (mlkem_fntt.hpp), |zeta b| <= 2^24, |result| <= 1668
__device__ __forceinline__ v2f fmulmod_shoup2(Tw w, v2f b) {
    const v2f km = fma2(b, splat2(w.zq), splat2(F_MAGIC));
    const v2f nkq = fma2(km, splat2(-F_Q), splat2(F_MAGIC_Q));
    return fma2(b, splat2(w.z), nkq);
}
__device__ __forceinline__ v2f fred2(v2f x) {
    const v2f k = fma2(x, splat2(F_INVQ), splat2(F_MAGIC)) - splat2(F_MAGIC);
    return fma2(k, splat2(-F_Q), x);
}
// register butterflies (any twiddle, register or literal)
__device__ __forceinline__ void ct_bfly_p(v2f& a, v2f& b, Tw zeta) {
    const v2f t = fmulmod_shoup2(zeta, b);
    b = a - t;
    a = a + t;
}
__device__ __forceinline__ void gs_bfly_p(v2f& a, v2f& b, Tw zeta) {
    const v2f t = a;
    a = t + b;
    b = fmulmod_shoup2(zeta, b - t);
}
template <int XOR>
__device__ __forceinline__ void xlane_fmac_p(v2f (&v)[4], float c) {
    float t[8] = {v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y};   // the same registers: no moves
    xlane_fmac8<XOR>(t, c);
    v[0] = v2f{t[0], t[1]}; v[1] = v2f{t[2], t[3]}; v[2] = v2f{t[4], t[5]}; v[3] = v2f{t[6], t[7]};
}
template <int XOR>
__device__ __forceinline__ void ct_xlane(v2f (&lo)[4], v2f (&hi)[4], Tw e0, Tw e1, float c) {
#pragma unroll
    for (int i = 0; i < 4; i++) { lo[i] = fmulmod_shoup2(e0, lo[i]); hi[i] = fmulmod_shoup2(e1, hi[i]); }
    xlane_fmac_p<XOR>(lo, c);
    xlane_fmac_p<XOR>(hi, c);
}
template <int XOR>
__device__ __forceinline__ void gs_xlane(v2f (&lo)[4], v2f (&hi)[4], Tw e0, Tw e1, float c) {
    xlane_fmac_p<XOR>(lo, c);
    xlane_fmac_p<XOR>(hi, c);
#pragma unroll
    for (int i = 0; i < 4; i++) { lo[i] = fmulmod_shoup2(e0, lo[i]); hi[i] = fmulmod_shoup2(e1, hi[i]); }
}

// Forward NTT of the wave's four polynomials.  lo[i] = coefficients (8 m + 2 i, 8 m + 2 i + 1), hi[i] = the pair 128 further.
// In: 0 <= x <= 4095.  Out: the same positions, |x| <= 6672 (lazy; canonicalise with fcanon_floor).
__device__ __forceinline__ void wave4_ntt_p(v2f (&lo)[4], v2f (&hi)[4], const RnttTw& t) {
#pragma unroll
    for (int r = 0; r < 4; r++) ct_bfly_p(lo[r], hi[r], FZ1);               // len 128 : <= 4095 + 1668
    ct_xlane<XL_IDX6>(lo, hi, t.e6[0], t.e6[1], t.c6);                       // len 64  : multiplicand <= 5763, out <= 3336
    ct_xlane<XL_IDX5>(lo, hi, t.e5[0], t.e5[1], t.c5);                       // len 32
    ct_xlane<XL_IDX4>(lo, hi, t.e4[0], t.e4[1], t.c4);                       // len 16
    ct_xlane<XL_IDX3>(lo, hi, t.e3[0], t.e3[1], t.c3);                       // len 8   : <= 3336
#pragma unroll
    for (int r = 0; r < 2; r++) { ct_bfly_p(lo[r], lo[r + 2], t.z2[0]); ct_bfly_p(hi[r], hi[r + 2], t.z2[1]); }   // len 4 : <= 5004
    ct_bfly_p(lo[0], lo[1], t.z1[0][0]); ct_bfly_p(lo[2], lo[3], t.z1[0][1]);                                    // len 2 : <= 6672
    ct_bfly_p(hi[0], hi[1], t.z1[1][0]); ct_bfly_p(hi[2], hi[3], t.z1[1][1]);
}
// Inverse NTT including the multiplication by 128^-1 (ml_kem.c:336-384).  In: 0 <= x <= 4095.  Out: |x| <= 1668.
__device__ __forceinline__ void wave4_intt_p(v2f (&lo)[4], v2f (&hi)[4], const RnttTw& t) {
    gs_bfly_p(lo[0], lo[1], t.z1[0][0]); gs_bfly_p(lo[2], lo[3], t.z1[0][1]);   // len 2 : sums <= 8190, products <= 1668
    gs_bfly_p(hi[0], hi[1], t.z1[1][0]); gs_bfly_p(hi[2], hi[3], t.z1[1][1]);
#pragma unroll
    for (int r = 0; r < 2; r++) { gs_bfly_p(lo[r], lo[r + 2], t.z2[0]); gs_bfly_p(hi[r], hi[r + 2], t.z2[1]); }   // len 4 : multiplicand <= 9858
    lo[0] = fred2(lo[0]); hi[0] = fred2(hi[0]);                              // the sums of sums (<= 16380); the rest <= 3336
    gs_xlane<XL_IDX3>(lo, hi, t.e3[0], t.e3[1], t.c3);                       // len 8  : |x +- x'| <= 6672, out <= 1668
    gs_xlane<XL_IDX4>(lo, hi, t.e4[0], t.e4[1], t.c4);                       // len 16
    gs_xlane<XL_IDX5>(lo, hi, t.e5[0], t.e5[1], t.c5);                       // len 32
    gs_xlane<XL_IDX6>(lo, hi, t.e6[0], t.e6[1], t.c6);                       // len 64
#pragma unroll
    for (int r = 0; r < 4; r++) {                                            // len 128 with the final scaling folded in:
        const v2f sm = lo[r] + hi[r], d = hi[r] - lo[r];                     //   a' = 128^-1 (a + b), b' = (zeta_1 128^-1)(b - a)
        lo[r] = fmulmod_shoup2(F_INV128, sm);                                //   |s|, |d| <= 3336
        hi[r] = fmulmod_shoup2(F_INV128_Z1, d);
    }
}
// float[8] views (lo[r] = coefficient 8 m + r, hi[r] = coefficient 128 + 8 m + r): what the codecs around the transforms use
#define MLKEM_V2F_IN(l, h, lo, hi)                                                                              \
    v2f l[4] = {v2f{lo[0], lo[1]}, v2f{lo[2], lo[3]}, v2f{lo[4], lo[5]}, v2f{lo[6], lo[7]}};                    \
    v2f h[4] = {v2f{hi[0], hi[1]}, v2f{hi[2], hi[3]}, v2f{hi[4], hi[5]}, v2f{hi[6], hi[7]}};
#define MLKEM_V2F_OUT(l, h, lo, hi)                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; i++) { lo[2 * i] = l[i].x; lo[2 * i + 1] = l[i].y; hi[2 * i] = h[i].x; hi[2 * i + 1] = h[i].y; }
__device__ __forceinline__ void wave4_ntt_r(float (&lo)[8], float (&hi)[8], const RnttTw& t) {
    MLKEM_V2F_IN(l, h, lo, hi)
    wave4_ntt_p(l, h, t);
    MLKEM_V2F_OUT(l, h, lo, hi)
}
__device__ __forceinline__ void wave4_intt_r(float (&lo)[8], float (&hi)[8], const RnttTw& t) {
    MLKEM_V2F_IN(l, h, lo, hi)
    wave4_intt_p(l, h, t);
    MLKEM_V2F_OUT(l, h, lo, hi)
}
#undef MLKEM_V2F_IN
#undef MLKEM_V2F_OUT

// 8 coefficients (16 bytes) <-> registers; inputs are taken mod 2^12 like the reference's 12-bit `union integer.t`
__device__ __forceinline__ void rntt_unpack(const uint4 v, float (&x)[8]) {
    const uint32_t w[4] = {v.x & 0x0FFF0FFFu, v.y & 0x0FFF0FFFu, v.z & 0x0FFF0FFFu, v.w & 0x0FFF0FFFu};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        x[2 * i] = (float)(w[i] & 0xFFFFu);      // (an SDWA conversion that picks its 16-bit half was measured 4 % slower)
        x[2 * i + 1] = (float)(w[i] >> 16);
    }
}
__device__ __forceinline__ uint4 rntt_pack_canonical(const float (&x)[8]) {
    uint32_t c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = (uint32_t)fcanon_floor(x[i]);
    uint4 o;
    o.x = c[0] | (c[1] << 16); o.y = c[2] | (c[3] << 16); o.z = c[4] | (c[5] << 16); o.w = c[6] | (c[7] << 16);
    return o;
}

// ---- the reference's forward NTT on coefficients >= q --------------------------------------------------------------
// FIPS 203 only defines the transform on [0, q), and every call site of the reference feeds it such values, but NTT()
// itself accepts any 12-bit `union integer.t` and its arithmetic is not modular there: ml_kem.c:317 stores f[j] - t
// UNREDUCED when f[j] >= t, so a coefficient >= q can travel through the layers and come out >= q (NTT of 4095 x^0 has
// 4095 at index 254).  That behaviour is well defined (no overflow; both builds of the reference and the oracle agree), so
// the batch entry reproduces it: a wave whose four polynomials contain any coefficient >= q (v_max3_f32 over the unpacked
// inputs and one compare) does them with the reference's integer steps, layer by layer, in 4 KB of wave-private LDS.
// The path is cold; the register transform above is used otherwise.
// (InverseNTT is different: for inputs >= q the reference's ml_kem.c:364-367 overflows a signed int -- the -O0 and -O2
// builds of the reference disagree with each other, tests/test_oracle_vs_reference.py -- so there is nothing to match;
// the inverse entry transforms the input reduced mod q.)
__device__ __forceinline__ bool rntt_any_noncanonical(const float (&lo)[8], const float (&hi)[8]) {   // the unpacked 12-bit inputs
    float m = __builtin_fmaxf(lo[0], hi[0]);
#pragma unroll
    for (int i = 1; i < 8; i++) m = __builtin_fmaxf(__builtin_fmaxf(m, lo[i]), hi[i]);   // v_max3_f32
    return __ballot(m >= F_Q) != 0;
}
struct RnttOut { uint4 lo, hi; };
// noinline on purpose: inlined (or with more live state) the cold path raises the hot loop's register count past the
// 104-VGPR step or makes it spill (measured: 0.26 ms instead of 0.21 ms per 2^20 polynomials)
__device__ __attribute__((noinline)) RnttOut wave4_ntt_reference_steps(uint32_t* w, const RnttLane a, const uint4 vlo, const uint4 vhi) {
    // w: 1024 words of wave-private LDS, [polynomial of the four][coefficient]
    const int l = lane_id();
    uint32_t* mine = w + a.p * 256 + a.m * 8;
    const uint32_t in[8] = {vlo.x, vlo.y, vlo.z, vlo.w, vhi.x, vhi.y, vhi.z, vhi.w};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int o = (i >> 2) * 128 + 2 * (i & 3);
        mine[o] = in[i] & 0xFFFu;
        mine[o + 1] = (in[i] >> 16) & 0xFFFu;
    }
    wave_lds_fence();
#pragma unroll 1
    for (int len = 128; len >= 2; len >>= 1) {
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            const int t = l + 64 * k, u = t & 127, blk = u / len;
            uint32_t* f = w + (t >> 7) * 256 + blk * 2 * len + (u - blk * len);
            const int zc = (int)ZETA_F.z[128 / len + blk].z;                 // centred table -> canonical zeta
            const uint32_t zeta = (uint32_t)(zc < 0 ? zc + KQ : zc);
            const uint32_t fa = f[0], tt = (zeta * f[len]) % (uint32_t)KQ;    // ml_kem.c:313 (3328 * 4095 < 2^24)
            f[len] = fa >= tt ? fa - tt : (uint32_t)KQ - (tt - fa);           // ml_kem.c:317-318, NOT reduced
            f[0] = (fa + tt) % (uint32_t)KQ;                                  // ml_kem.c:322-323
        }
        wave_lds_fence();
    }
    RnttOut o;
    o.lo.x = mine[0] | (mine[1] << 16); o.lo.y = mine[2] | (mine[3] << 16);
    o.lo.z = mine[4] | (mine[5] << 16); o.lo.w = mine[6] | (mine[7] << 16);
    o.hi.x = mine[128] | (mine[129] << 16); o.hi.y = mine[130] | (mine[131] << 16);
    o.hi.z = mine[132] | (mine[133] << 16); o.hi.w = mine[134] | (mine[135] << 16);
    wave_lds_fence();
    return o;
}

// streaming 16-byte accesses: every byte is touched once, so loads and stores carry the non-temporal hint (measured with
// tools/ntt_ubench.hip: -3 % kernel time; a plain copy with this loop reaches 6.4-6.7 TB/s with it, 5.8-6.2 without)
__device__ __forceinline__ uint4 load16_stream(const uint16_t* p) {
#ifdef MLKEM_EMU
    return *reinterpret_cast<const uint4*>(p);
#else
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
    uint4 r;
    r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w;
    return r;
#endif
}
__device__ __forceinline__ void store16_stream(uint16_t* p, const uint4 o) {
#ifdef MLKEM_EMU
    *reinterpret_cast<uint4*>(p) = o;
#else
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 t;
    t.x = o.x; t.y = o.y; t.z = o.z; t.w = o.w;
    __builtin_nontemporal_store(t, reinterpret_cast<v4*>(p));
#endif
}

// Stand-alone NTT / InverseNTT over n polynomials, four per wave and iteration, grid-stride with the next four's 2 x 16
// bytes requested before the current four are transformed.  uint16 in / out; canonical output (forward: except where the
// reference itself yields a value >= q, see above).
#ifndef MLKEM_RNTT_WAVES
#define MLKEM_RNTT_WAVES 4
#endif
constexpr int RNTT_WAVES = MLKEM_RNTT_WAVES;
#ifndef MLKEM_RNTT_MINWAVES
#define MLKEM_RNTT_MINWAVES 1   // register budget knob: 1 = the compiler's own choice (96-98 VGPRs, no spills); forcing >= 5 waves spills and is 20-40 % slower
#endif
#ifndef MLKEM_RNTT_PREFETCH
#define MLKEM_RNTT_PREFETCH 1   // iterations of input kept in flight per wave (x 2 KB); 2 was measured slower (registers)
#endif
constexpr int RNTT_PF = MLKEM_RNTT_PREFETCH;
template <bool INVERSE>
__global__ void __launch_bounds__(64 * RNTT_WAVES, MLKEM_RNTT_MINWAVES) k_ntt4_batch(size_t n, const uint16_t* __restrict__ in, uint16_t* __restrict__ out) {
    __shared__ uint32_t exact_all[INVERSE ? 1 : RNTT_WAVES][INVERSE ? 1 : 1024];
    const int wv = (int)(threadIdx.x >> 6);
    const RnttLane a = rntt_lane();
    RnttTw tw;
    if constexpr (INVERSE) rntt_load_twiddles_inv(tw);
    else rntt_load_twiddles_fwd(tw);
    const size_t nquads = (n + 3) / 4, stride = (size_t)gridDim.x * RNTT_WAVES;
    size_t q = (size_t)blockIdx.x * RNTT_WAVES + wv;
    if (q >= nquads) return;
    // rows whose polynomial lies beyond n (n % 4 != 0) re-read polynomial n - 1 and store nothing
    const size_t lane_off = (size_t)a.m * 8;   // uint16 elements inside the polynomial; the second piece is 128 further
    auto src = [&](size_t quad) {
        const size_t pi = 4 * quad + (size_t)a.p;
        return in + (pi < n ? pi : n - 1) * 256 + lane_off;
    };
    uint4 b_lo[RNTT_PF], b_hi[RNTT_PF];
#pragma unroll
    for (int d = 0; d < RNTT_PF; d++)
        if (q + d * stride < nquads) {
            const uint16_t* s = src(q + d * stride);
            b_lo[d] = load16_stream(s);
            b_hi[d] = load16_stream(s + 128);
        }
    for (; q < nquads; q += stride) {
        const uint4 v_lo = b_lo[0], v_hi = b_hi[0];
#pragma unroll
        for (int d = 0; d + 1 < RNTT_PF; d++) { b_lo[d] = b_lo[d + 1]; b_hi[d] = b_hi[d + 1]; }
        if (q + RNTT_PF * stride < nquads) {
            const uint16_t* s = src(q + RNTT_PF * stride);
            b_lo[RNTT_PF - 1] = load16_stream(s);
            b_hi[RNTT_PF - 1] = load16_stream(s + 128);
        }
        float lo[8], hi[8];
        rntt_unpack(v_lo, lo);
        rntt_unpack(v_hi, hi);
        uint4 o_lo, o_hi;
        bool exact = false;
        if constexpr (!INVERSE) exact = rntt_any_noncanonical(lo, hi);
        if (exact) {
            if constexpr (!INVERSE) {
                const RnttOut o = wave4_ntt_reference_steps(exact_all[wv], a, v_lo, v_hi);
                o_lo = o.lo;
                o_hi = o.hi;
            }
        } else {
            if constexpr (INVERSE) wave4_intt_r(lo, hi, tw);
            else wave4_ntt_r(lo, hi, tw);
            o_lo = rntt_pack_canonical(lo);
            o_hi = rntt_pack_canonical(hi);
        }
        const size_t pi = 4 * q + (size_t)a.p;
        if (pi < n) {
            uint16_t* d = out + pi * 256 + lane_off;
            store16_stream(d, o_lo);
            store16_stream(d + 128, o_hi);
        }
    }
}

}   // namespace mlkem
