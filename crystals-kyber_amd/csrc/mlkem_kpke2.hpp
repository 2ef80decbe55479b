// mlkem_kpke2.hpp — K-PKE.Encrypt and K-PKE.KeyGen (ml_kem.c:776-936, :651-769) with TWO items per wavefront and every
// fp32 operation PACKED: 32 lanes per polynomial, 8 coefficients = 4 idx0-pairs per lane.
//
// Why (profiles/r03_kpke_experiments.txt): the one lever that removed instructions AND time in round 3 was packed fp32
// where the pairing is free — coefficients 2i and 2i+1 go through identical operations with identical twiddles in every
// layer (idx0 is never a butterfly bit).  The one-polynomial-per-wave transform of mlkem_fntt.hpp keeps idx0 pairs in
// registers in only one of its four layouts; here every layout does: a lane's four float2 registers are indexed by two
// butterfly bits (hi, lo), so each layout runs two layers, every butterfly, base-case product and Compress step is a
// v_pk_fma_f32 / v_pk_add_f32, exchanges move 8 bytes per LDS access, and the 10- / 4- / 12-bit output pieces of 8
// coefficients are assembled in registers (one DPP move merges the dword two lanes share) instead of through LDS atomics.
//
// Index bits of a coefficient: idx7..idx0.  pair = idx7..idx1 (128 pairs per polynomial).  Lane t = (t4 t3 t2 t1 t0) of the
// item's half-wave and register j (0..3) hold pair:
//   LA : j = (idx7, idx6)   t = (idx5, idx4, idx3, idx2, idx1)   layers len 128, 64   (forward input: CBD evaluates here)
//   LB : j = (idx5, idx4)   t = (idx7, idx6, idx3, idx2, idx1)   layers len 32, 16
//   LC : j = (idx3, idx2)   t = (idx5, idx7, idx6, idx4, idx1)   layers len 8, 4
//   NAT: j = (idx2, idx1)   t = (idx5, idx7, idx6, idx4, idx3)   layer  len 2; the lane owns block k2_blk(t) = idx7..idx3 of the
//                                                                 HBM order (8 consecutive coefficients), base-case products, codecs
// LDS exchange buffer: 128 float2 slots per polynomial, slot = pair ^ (idx7 ? 0x15 : 0) ^ (idx6 ? 0x0A : 0).  The LDS serves a
// ds_read_b64 in two groups of 32 lanes over 64 banks and a ds_write_b64 (and the ds_read2 forms) in four groups of 16
// contiguous lanes over 32 banks (MI355X_MICROARCH.md).  With this swizzle the map from the five lane bits to the low five
// slot bits is a bijection in all four layouts (reads: every bank once per group), and because idx5 -- the one index bit that
// only reaches slot bit 4 -- is lane bit t4 wherever it is a lane bit, the four lane bits that vary inside a 16-lane group map
// bijectively to the low four slot bits (writes: every bank once per group).  (With idx7 as t4 in LC and NAT a third of the
// LDS cycles of the first build were bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.32.)
//
// Lazy bounds (integers, exact below 2^24; a twiddle product needs |b| <= 10082): forward transforms only ever see CBD
// samples (|x| <= 3), the bound grows by 1668 per layer, so the multiplicand of layer 7 is <= 3 + 6 * 1668 = 10011 and no
// intermediate reduction is needed; inverse transforms reduce the two sum-path pairs after every two-layer stage exactly
// like wave_intt_f (inputs of a stage <= 2520).
#pragma once
#include "mlkem_arith.hpp"
#include "mlkem_rntt.hpp"
#include <type_traits>

namespace mlkem {

#ifndef MLKEM_KPKE2_WAVES
#define MLKEM_KPKE2_WAVES 1   // waves per workgroup: 1 measured best (profiles/r03_kpke_experiments.txt)
#endif
constexpr int KPKE2_WAVES = MLKEM_KPKE2_WAVES;
#ifdef MLKEM_KPKE2_MINWAVES
constexpr int kpke2_minwaves(int) { return MLKEM_KPKE2_MINWAVES; }
#else
constexpr int kpke2_minwaves(int k) { return k == 2 ? 4 : k == 3 ? 3 : 2; }   // k = 3: 140-157 VGPRs; two, three waves measured equal, four spill
#endif

// exchange buffers of one wave: NP polynomials in flight per item (their transforms are interleaved instruction by
// instruction: independent work between an LDS read and its use, and between the dependent packed FMAs of a butterfly)
template <int NP>
struct __attribute__((aligned(16))) K2Lds {
    float2 xch[NP][2][128];   // [polynomial in flight][half = item of the wave][swizzled pair slot]
};

__device__ __forceinline__ int k2_slot(int pair) { return pair ^ ((pair & 64) ? 0x15 : 0) ^ ((pair & 32) ? 0x0A : 0); }
enum K2Layout { K2_LA = 0, K2_LB = 1, K2_LC = 2, K2_NAT = 3 };
// block of 8 consecutive coefficients the lane owns in NAT layout: idx7..idx3 = (t3, t2, t4, t1, t0); lanes 4m..4m+3 own
// consecutive blocks (the piece codecs' lane pairs and quads)
__device__ __forceinline__ int k2_blk(int t) { return (((t >> 2) & 3) << 3) | ((t >> 4) << 2) | (t & 3); }
// idx7..idx4 of the lane in LC layout = (t3, t2, t4, t1)
__device__ __forceinline__ int k2_lc_hi(int t) { return (((t >> 2) & 3) << 2) | ((t >> 4) << 1) | ((t >> 1) & 1); }
template <int LAYOUT>
__device__ __forceinline__ int k2_pair(int t, int j) {
    if constexpr (LAYOUT == K2_LA) return (j << 5) | t;
    else if constexpr (LAYOUT == K2_LB) return ((t >> 3) << 5) | (j << 3) | (t & 7);
    else if constexpr (LAYOUT == K2_LC) return (k2_lc_hi(t) << 3) | (j << 1) | (t & 1);
    else return (k2_blk(t) << 2) | j;
}
template <int LAYOUT>
__device__ __forceinline__ void k2_write(float2* xh, int t, const v2f (&p)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float2 v;
        v.x = p[j].x; v.y = p[j].y;
        xh[k2_slot(k2_pair<LAYOUT>(t, j))] = v;
    }
}
template <int LAYOUT>
__device__ __forceinline__ void k2_read(const float2* xh, int t, v2f (&p)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float2 v = xh[k2_slot(k2_pair<LAYOUT>(t, j))];   // (forcing single ds_read_b64 through volatile accesses spilled: 3x slower)
        p[j] = v2f{v.x, v.y};
    }
}
// NP polynomials of the item change layout together: all writes, one fence, all reads
template <int FROM, int TO, int NP>
__device__ __forceinline__ void k2_exchange(float2 (*xch)[2][128], int h, int t, v2f (&p)[NP][4]) {
#pragma unroll
    for (int q = 0; q < NP; q++) k2_write<FROM>(xch[q][h], t, p[q]);
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < NP; q++) k2_read<TO>(xch[q][h], t, p[q]);
    wave_lds_fence();
}

// per-lane twiddles of one direction: stage B (len 32 | 16, 16), stage C (len 8 | 4, 4), stage D (len 2, 2)
struct K2Tw {
    Tw b0, b1, b2, c0, c1, c2, d0, d1;
};
// forward: zeta index of layer `len` = 128/len + (coefficient index >> log2(2 len))   (ml_kem.c:296-324)
__device__ __forceinline__ void k2_twiddles_fwd(K2Tw& w, int t) {
    const Tw* z = ZETA_F.z;
    const int b = t >> 3, c = k2_lc_hi(t), n = k2_blk(t);   // (idx7, idx6) in LB ; idx7..idx4 in LC ; idx7..idx3 in NAT
    w.b0 = z[4 + b]; w.b1 = z[8 + 2 * b]; w.b2 = z[9 + 2 * b];
    w.c0 = z[16 + c]; w.c1 = z[32 + 2 * c]; w.c2 = z[33 + 2 * c];
    w.d0 = z[64 + 2 * n]; w.d1 = z[65 + 2 * n];
}
// inverse: the same table walked backwards (ml_kem.c:345-373)
__device__ __forceinline__ void k2_twiddles_inv(K2Tw& w, int t) {
    const Tw* z = ZETA_F.z;
    const int b = t >> 3, c = k2_lc_hi(t), n = k2_blk(t);
    w.d0 = z[127 - 2 * n]; w.d1 = z[126 - 2 * n];
    w.c1 = z[63 - 2 * c]; w.c2 = z[62 - 2 * c]; w.c0 = z[31 - c];
    w.b1 = z[15 - 2 * b]; w.b2 = z[14 - 2 * b]; w.b0 = z[7 - b];
}

// ---- batched packed arithmetic ------------------------------------------------------------------------------------------
// A packed FMA that consumes the result of the packed instruction right before it costs a wait state on gfx950 (the
// compiler fills it with s_nop), and a butterfly is a chain of five such instructions.  Every primitive therefore works on N
// independent operands STEP BY STEP (step 1 of all, then step 2 of all, ...): the chains of the 2 NP butterflies of a layer
// interleave, and so do the NP polynomials whose transforms run together.
template <int N>
__device__ __forceinline__ void k2_shoup_n(const Tw (&z)[N], v2f (&b)[N]) {   // b[i] = z[i] * b[i] mod q, |result| <= 1668 (fmulmod_shoup)
    v2f km[N], nk[N];
#pragma unroll
    for (int i = 0; i < N; i++) km[i] = fma2(b[i], splat2(z[i].zq), splat2(F_MAGIC));
#pragma unroll
    for (int i = 0; i < N; i++) nk[i] = fma2(km[i], splat2(-F_Q), splat2(F_MAGIC_Q));
#pragma unroll
    for (int i = 0; i < N; i++) b[i] = fma2(b[i], splat2(z[i].z), nk[i]);
}
template <int N>
__device__ __forceinline__ void k2_fred_n(v2f (&x)[N]) {                       // centred representative, |x| <= 2^24 (fred)
    v2f k[N];
#pragma unroll
    for (int i = 0; i < N; i++) k[i] = fma2(x[i], splat2(F_INVQ), splat2(F_MAGIC));
#pragma unroll
    for (int i = 0; i < N; i++) k[i] = k[i] - splat2(F_MAGIC);
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = fma2(k[i], splat2(-F_Q), x[i]);
}
// N butterflies (a[i], b[i]) with twiddles z[i]: Cooley-Tukey a' = a + z b, b' = a - z b ; Gentleman-Sande a' = a + b, b' = z (b - a)
template <int N, bool GS>
__device__ __forceinline__ void k2_bfly_n(v2f (&a)[N], v2f (&b)[N], const Tw (&z)[N]) {
    if constexpr (!GS) {
        k2_shoup_n<N>(z, b);
#pragma unroll
        for (int i = 0; i < N; i++) { const v2f tt = b[i]; b[i] = a[i] - tt; a[i] = a[i] + tt; }
    } else {
        v2f d[N];
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = b[i] - a[i];
#pragma unroll
        for (int i = 0; i < N; i++) a[i] = a[i] + b[i];
        k2_shoup_n<N>(z, d);
#pragma unroll
        for (int i = 0; i < N; i++) b[i] = d[i];
    }
}
// one layer over NP polynomials: the "hi" register bit pairs registers (0, 2) and (1, 3) under one twiddle, the "lo" bit
// pairs (0, 1) under z01 and (2, 3) under z23
template <int NP, bool GS>
__device__ __forceinline__ void k2_layer_hi(v2f (&p)[NP][4], Tw z) {
    v2f a[2 * NP], b[2 * NP];
    Tw zz[2 * NP];
#pragma unroll
    for (int q = 0; q < NP; q++) { a[2 * q] = p[q][0]; b[2 * q] = p[q][2]; a[2 * q + 1] = p[q][1]; b[2 * q + 1] = p[q][3]; zz[2 * q] = z; zz[2 * q + 1] = z; }
    k2_bfly_n<2 * NP, GS>(a, b, zz);
#pragma unroll
    for (int q = 0; q < NP; q++) { p[q][0] = a[2 * q]; p[q][2] = b[2 * q]; p[q][1] = a[2 * q + 1]; p[q][3] = b[2 * q + 1]; }
}
template <int NP, bool GS>
__device__ __forceinline__ void k2_layer_lo(v2f (&p)[NP][4], Tw z01, Tw z23) {
    v2f a[2 * NP], b[2 * NP];
    Tw zz[2 * NP];
#pragma unroll
    for (int q = 0; q < NP; q++) { a[2 * q] = p[q][0]; b[2 * q] = p[q][1]; a[2 * q + 1] = p[q][2]; b[2 * q + 1] = p[q][3]; zz[2 * q] = z01; zz[2 * q + 1] = z23; }
    k2_bfly_n<2 * NP, GS>(a, b, zz);
#pragma unroll
    for (int q = 0; q < NP; q++) { p[q][0] = a[2 * q]; p[q][1] = b[2 * q]; p[q][2] = a[2 * q + 1]; p[q][3] = b[2 * q + 1]; }
}
// reduce registers R0 and R1 of every polynomial (the sum paths of an inverse stage)
template <int NP, int R0, int R1>
__device__ __forceinline__ void k2_fred_pairs(v2f (&p)[NP][4]) {
    v2f x[2 * NP];
#pragma unroll
    for (int q = 0; q < NP; q++) { x[2 * q] = p[q][R0]; x[2 * q + 1] = p[q][R1]; }
    k2_fred_n<2 * NP>(x);
#pragma unroll
    for (int q = 0; q < NP; q++) { p[q][R0] = x[2 * q]; p[q][R1] = x[2 * q + 1]; }
}

// Forward NTT (ml_kem.c:287-329) of NP polynomials.  In: LA layout, |x| <= 3.  Out: NAT layout, |x| <= 3 + 7 * 1668 (lazy).
template <int NP>
__device__ __forceinline__ void k2_ntt(v2f (&p)[NP][4], float2 (*xch)[2][128], int h, int t, const K2Tw& w) {
    k2_layer_hi<NP, false>(p, FZ1);                     // len 128
    k2_layer_lo<NP, false>(p, FZ2, FZ3);                // len 64
    k2_exchange<K2_LA, K2_LB, NP>(xch, h, t, p);
    k2_layer_hi<NP, false>(p, w.b0);                    // len 32
    k2_layer_lo<NP, false>(p, w.b1, w.b2);              // len 16
    k2_exchange<K2_LB, K2_LC, NP>(xch, h, t, p);
    k2_layer_hi<NP, false>(p, w.c0);                    // len 8
    k2_layer_lo<NP, false>(p, w.c1, w.c2);              // len 4
    k2_exchange<K2_LC, K2_NAT, NP>(xch, h, t, p);
    k2_layer_lo<NP, false>(p, w.d0, w.d1);              // len 2 : multiplicand <= 10011
}
// Inverse NTT incl. the multiplication by 128^-1 (ml_kem.c:336-384) of NP polynomials.  In: NAT layout, |x| <= 2520.
// Out: NAT layout, |x| <= 1668.
template <int NP>
__device__ __forceinline__ void k2_intt(v2f (&p)[NP][4], float2 (*xch)[2][128], int h, int t, const K2Tw& w) {
    k2_layer_lo<NP, true>(p, w.d0, w.d1);               // len 2 : sums <= 5040, products <= 1668
    k2_fred_pairs<NP, 0, 2>(p);
    k2_exchange<K2_NAT, K2_LC, NP>(xch, h, t, p);
    k2_layer_lo<NP, true>(p, w.c1, w.c2);               // len 4
    k2_layer_hi<NP, true>(p, w.c0);                     // len 8 : p0 <= 6672, p1 <= 3336, p2, p3 <= 1668
    k2_fred_pairs<NP, 0, 1>(p);
    k2_exchange<K2_LC, K2_LB, NP>(xch, h, t, p);
    k2_layer_lo<NP, true>(p, w.b1, w.b2);               // len 16
    k2_layer_hi<NP, true>(p, w.b0);                     // len 32
    k2_fred_pairs<NP, 0, 1>(p);
    k2_exchange<K2_LB, K2_LA, NP>(xch, h, t, p);
    k2_layer_lo<NP, true>(p, FZ3, FZ2);                 // len 64 : sums <= 3336
    {                                                   // len 128 with the final x 128^-1 folded in (ml_kem.c:378-381):
        v2f sm[2 * NP], df[2 * NP];                     //   a' = 128^-1 (a + b), b' = (zeta_1 128^-1)(b - a)
        Tw zs[2 * NP], zd[2 * NP];
#pragma unroll
        for (int q = 0; q < NP; q++)
#pragma unroll
            for (int m = 0; m < 2; m++) {
                sm[2 * q + m] = p[q][m] + p[q][m + 2];
                df[2 * q + m] = p[q][m + 2] - p[q][m];
                zs[2 * q + m] = F_INV128; zd[2 * q + m] = F_INV128_Z1;
            }
        k2_shoup_n<2 * NP>(zs, sm);
        k2_shoup_n<2 * NP>(zd, df);
#pragma unroll
        for (int q = 0; q < NP; q++)
#pragma unroll
            for (int m = 0; m < 2; m++) { p[q][m] = sm[2 * q + m]; p[q][m + 2] = df[2 * q + m]; }
    }
    k2_exchange<K2_LA, K2_NAT, NP>(xch, h, t, p);
}

// ---- SamplePolyCBD (ml_kem.c:253-275) ------------------------------------------------------------------------------------
// raw bits of the lane's four pairs; LA layout: pairs t, t + 32, t + 64, t + 96
template <int ETA>
struct K2CbdRaw {
    uint32_t w[ETA == 2 ? 1 : 2];
};
template <int ETA>
__device__ __forceinline__ void k2_cbd_load_la(const uint8_t* prf, int t, K2CbdRaw<ETA>& r) {
    if constexpr (ETA == 2) {   // a pair = 8 bits = byte `pair` of the 128-byte row
        r.w[0] = (uint32_t)prf[t] | ((uint32_t)prf[t + 32] << 8) | ((uint32_t)prf[t + 64] << 16) | ((uint32_t)prf[t + 96] << 24);
    } else {                    // a pair = 12 bits at bit 12 * pair of the 192-byte row
        uint32_t f[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int o = ((3 * t) >> 1) + 48 * j;
            f[j] = (((uint32_t)prf[o] | ((uint32_t)prf[o + 1] << 8)) >> (4 * (t & 1))) & 0xFFFu;
        }
        r.w[0] = f[0] | (f[1] << 12);
        r.w[1] = f[2] | (f[3] << 12);
    }
}
// NAT layout (coefficients 8 t .. 8 t + 7), eta = 2: the dword at byte 4 t
__device__ __forceinline__ uint32_t k2_cbd_load_nat2(const uint8_t* prf, int t) { return reinterpret_cast<const uint32_t*>(prf)[t]; }
// CBD_2 of 8 coefficients = the 8 nibbles of one dword, all nibbles at once: per nibble (a0 a1 b0 b1) the coefficient is
// (a0 + a1) - (b0 + b1).  s = pairwise bit sums (2-bit fields), d = low field + 2 - high field in 0..4 per nibble (no borrow
// crosses a nibble); even nibbles are the low nibbles of the four bytes = the .x halves of the lane's four pairs, odd nibbles
// the .y halves, so the conversion is v_cvt_f32_ubyteN and the "- 2" four packed adds: 24 instructions for 8 coefficients
// (cbd_eval_f<2> twice: 38).
// BIASED: the coefficients + 2 (0..4), for consumers that fold the -2 into an FMA they issue anyway (k2_compress4<D, true>)
template <bool BIASED = false>
__device__ __forceinline__ void k2_cbd2_eval8(uint32_t t, v2f (&p)[4]) {
    const uint32_t s = (t & 0x55555555u) + ((t >> 1) & 0x55555555u);
    const uint32_t d = ((s & 0x33333333u) + 0x22222222u) - ((s >> 2) & 0x33333333u);
    const uint32_t lo = d & 0x0F0F0F0Fu, hi = (d >> 4) & 0x0F0F0F0Fu;
    float fl[4] = {(float)(lo & 0xFFu), (float)((lo >> 8) & 0xFFu), (float)((lo >> 16) & 0xFFu), (float)(lo >> 24)};
    float fh[4] = {(float)(hi & 0xFFu), (float)((hi >> 8) & 0xFFu), (float)((hi >> 16) & 0xFFu), (float)(hi >> 24)};
#ifndef MLKEM_EMU   // keep the byte -> float conversions as they are (v_cvt_f32_ubyteN)
    asm volatile("" : "+v"(fl[0]), "+v"(fl[1]), "+v"(fl[2]), "+v"(fl[3]), "+v"(fh[0]), "+v"(fh[1]), "+v"(fh[2]), "+v"(fh[3]));
#endif
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if constexpr (BIASED) p[j] = v2f{fl[j], fh[j]};
        else p[j] = v2f{fl[j], fh[j]} - splat2(2.0f);
    }
}
template <int ETA>
__device__ __forceinline__ void k2_cbd_eval(const K2CbdRaw<ETA>& r, v2f (&p)[4]) {
    if constexpr (ETA == 2) {
        k2_cbd2_eval8<false>(r.w[0], p);
    } else {
        float a[4], b[4];
        cbd_eval_f<3>(r.w[0], a);
        cbd_eval_f<3>(r.w[1], b);
        p[0] = v2f{a[0], a[1]}; p[1] = v2f{a[2], a[3]}; p[2] = v2f{b[0], b[1]}; p[3] = v2f{b[2], b[3]};
    }
}

// ---- base-case multiply-accumulate (ml_kem.c:395-442, :618-638), NAT layout ------------------------------------------------
// gamma of the lane's pairs 4 t + j: +zeta_{64 + 2t}, -zeta_{64 + 2t}, +zeta_{65 + 2t}, -zeta_{65 + 2t} = +-(d0, d1) of the
// forward set; yg[j] = y[2j + 1] * gamma_j
__device__ __forceinline__ void k2_gamma(const v2f (&y)[4], Tw d0, Tw d1, float (&yg)[4]) {
    yg[0] = fmulmod_shoup(d0, y[0].y);
    yg[1] = fmulmod_shoup(tw_neg(d0), y[1].y);
    yg[2] = fmulmod_shoup(d1, y[2].y);
    yg[3] = fmulmod_shoup(tw_neg(d1), y[3].y);
}
// acc = (acc + a o y) reduced: for 0 <= a <= 4095 and |y|, |yg|, |acc| <= 1668 the exact value stays below 2^24 (basemul_term)
__device__ __forceinline__ void k2_basemul_acc(v2f (&acc)[4], const v2f (&a)[4], const v2f (&y)[4], const float (&yg)[4]) {
    v2f c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = fma2(splat2(a[j].x), y[j], acc[j]);     // (a0 y0, a0 y1) + acc
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c[j].x = __builtin_fmaf(a[j].y, yg[j], c[j].x);                          // + a1 y1 gamma
        c[j].y = __builtin_fmaf(a[j].y, y[j].x, c[j].y);                         // + a1 y0
    }
    k2_fred_n<4>(c);
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = c[j];
}

// 8 uint16 coefficients (16 bytes, NAT) -> pairs; values are < 2^12 (sampled matrix: < q)
__device__ __forceinline__ void k2_unpack16(const uint4 v, v2f (&p)[4]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) p[j] = v2f{(float)(w[j] & 0xFFFFu), (float)(w[j] >> 16)};
}
// ByteDecode_12 (ml_kem.c:153-177, no reduction: F3) of the lane's 8 coefficients = 3 dwords at dword 3 t
__device__ __forceinline__ void k2_decode12(uint32_t w0, uint32_t w1, uint32_t w2, v2f (&p)[4]) {
    const uint32_t c0 = w0 & 0xFFFu, c1 = (w0 >> 12) & 0xFFFu, c2 = __builtin_amdgcn_alignbit(w1, w0, 24) & 0xFFFu, c3 = (w1 >> 4) & 0xFFFu;
    const uint32_t c4 = (w1 >> 16) & 0xFFFu, c5 = __builtin_amdgcn_alignbit(w2, w1, 28) & 0xFFFu, c6 = (w2 >> 8) & 0xFFFu, c7 = w2 >> 20;
    p[0] = v2f{(float)c0, (float)c1}; p[1] = v2f{(float)c2, (float)c3}; p[2] = v2f{(float)c4, (float)c5}; p[3] = v2f{(float)c6, (float)c7};
}

// Compress_D (ml_kem.c:83-97) of any representative |x| <= 4095 on both halves of a pair: compress_f (mlkem_fntt.hpp) packed.
// BIASED: x holds (value + 2) -- a CBD_2 sample that kept its bias (k2_cbd2_eval8<true>) was added; the -2 rides in the first FMA.
// The quotient k0 = round(2^D x / q) and its repair k in {-1, 0, 1} both leave their FMA as MAGIC + integer (MAGIC = 1.5 * 2^23),
// i.e. as bit patterns bits(MAGIC) + integer; 2 * bits(MAGIC) = 0x96800000 has its low 23 bits clear, so
// (bits(MAGIC + k0) + bits(MAGIC + k)) mod 2^D = (k0 + k) mod 2^D: one integer add and one mask per coefficient replace two packed
// subtractions / additions and a float -> int conversion (proven over the whole domain by sweep 8 of mlkem_selftest).
__device__ __forceinline__ uint32_t f_bits(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    return v.u;
}
template <int D, bool BIASED = false>
__device__ __forceinline__ void k2_compress4(const v2f (&x)[4], unsigned (&c)[8]) {
    v2f tt[4], k0m[4], k0[4], r[4], km[4];
#pragma unroll
    for (int j = 0; j < 4; j++) tt[j] = fma2(x[j], splat2((float)(1 << D)), splat2(BIASED ? -(float)(2 << D) : 0.0f));
#pragma unroll
    for (int j = 0; j < 4; j++) k0m[j] = fma2(tt[j], splat2(F_INVQ), splat2(F_MAGIC));
#pragma unroll
    for (int j = 0; j < 4; j++) k0[j] = k0m[j] - splat2(F_MAGIC);
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = fma2(k0[j], splat2(-F_Q), tt[j]);
#pragma unroll
    for (int j = 0; j < 4; j++) km[j] = fma2(r[j], splat2(F_INVQ), splat2(F_MAGIC));
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c[2 * j] = (f_bits(k0m[j].x) + f_bits(km[j].x)) & ((1u << D) - 1u);
        c[2 * j + 1] = (f_bits(k0m[j].y) + f_bits(km[j].y)) & ((1u << D) - 1u);
    }
}

// value of lane `lane ^ 1` (the other lane of an even / odd pair) / of the previous lane of the quad (lane 0 of a quad: itself)
__device__ __forceinline__ uint32_t k2_swap1(uint32_t v) {
#ifdef MLKEM_EMU
    return (uint32_t)emu_quad_shfl((int)v, (int)((threadIdx.x & 3) ^ 1));
#else
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm:[1,0,3,2]
#endif
}
__device__ __forceinline__ uint32_t k2_prev_in_quad(uint32_t v) {
#ifdef MLKEM_EMU
    const int l = (int)(threadIdx.x & 3);
    return (uint32_t)emu_quad_shfl((int)v, l ? l - 1 : l);
#else
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x90, 0xF, 0xF, true);   // quad_perm:[0,0,1,2]
#endif
}
// (hi:lo) << s, bits 32..63, for s in {0, 8, 16, 24} (lane-dependent)
__device__ __forceinline__ uint32_t k2_shl_hi(uint32_t hi, uint32_t lo, unsigned s) {
    return s ? __builtin_amdgcn_alignbit(hi, lo, 32u - s) : hi;
}

// ByteEncode_D of the lane's 8 D-bit values (ml_kem.c:125-145): the lane owns bytes [D t, D t + D) of the polynomial's 32 D
// bytes.  A piece = the dwords this lane stores / compares.  Where neighbouring lanes share a dword, ONE of them owns it and
// receives the other's bits by a DPP move:
//   D = 4, 12 : 1 / 3 dwords per lane, dword aligned
//   D = 10    : 80 bits; two lanes cover 5 dwords: the even lane stores dwords 0, 1 and the shared dword 2, the odd lane 3, 4
//   D = 11    : 88 bits; four lanes cover 11 dwords; lane j of the quad starts at bit 88 j = dword {0, 2, 5, 8} + {0, 24, 16, 8}
//               bits; a lane owns the dword it STARTS in (lanes 1-3 merge the previous lane's last bits): 2, 3, 3, 3 dwords
//   D = 5     : 40 bits; four lanes cover 5 dwords; lane j starts at dword j + 8 j bits; lanes own 1, 1, 1, 2 dwords
template <int D>
struct K2Piece {
    static_assert(D == 4 || D == 5 || D == 10 || D == 11 || D == 12, "ciphertext / key encodings of ML-KEM");
    static constexpr int NW = D == 4 ? 1 : D == 5 ? 2 : 3;
    uint32_t w[NW];
    // dword index of w[0] inside the polynomial, and how many of the lane's dwords are valid
    static __device__ __forceinline__ int first(int t) {
        if constexpr (D == 4) return t;
        else if constexpr (D == 12) return 3 * t;
        else if constexpr (D == 10) return 5 * (t >> 1) + ((t & 1) ? 3 : 0);
        else if constexpr (D == 11) return 11 * (t >> 2) + ((88 * (t & 3)) >> 5);
        else return 5 * (t >> 2) + (t & 3);
    }
    static __device__ __forceinline__ int count(int t) {
        if constexpr (D == 4) return 1;
        else if constexpr (D == 12) return 3;
        else if constexpr (D == 10) return (t & 1) ? 2 : 3;
        else if constexpr (D == 11) return (t & 3) ? 3 : 2;
        else return (t & 3) == 3 ? 2 : 1;
    }
};
template <int D>
__device__ __forceinline__ void k2_encode(const unsigned (&c)[8], int t, K2Piece<D>& o) {
    if constexpr (D == 4) {
        o.w[0] = c[0] | (c[1] << 4) | (c[2] << 8) | (c[3] << 12) | (c[4] << 16) | (c[5] << 20) | (c[6] << 24) | (c[7] << 28);
    } else if constexpr (D == 12) {
        o.w[0] = c[0] | (c[1] << 12) | (c[2] << 24);
        o.w[1] = (c[2] >> 8) | (c[3] << 4) | (c[4] << 16) | (c[5] << 28);
        o.w[2] = (c[5] >> 4) | (c[6] << 8) | (c[7] << 20);
    } else if constexpr (D == 10) {
        // 80 bits: v0 = bits 0..31, v1 = bits 32..63, v2 = bits 64..79
        const uint32_t v0 = c[0] | (c[1] << 10) | (c[2] << 20) | (c[3] << 30);
        const uint32_t v1 = (c[3] >> 2) | (c[4] << 8) | (c[5] << 18) | (c[6] << 28);
        const uint32_t v2 = (c[6] >> 4) | (c[7] << 6);
        const uint32_t other0 = k2_swap1(v0);
        const unsigned sh = (t & 1) ? 16u : 0u;
        o.w[0] = __builtin_amdgcn_alignbit(v1, v0, sh);             // even: v0          odd: bits 16..47
        o.w[1] = __builtin_amdgcn_alignbit(v2, v1, sh);             // even: v1          odd: bits 48..79
        o.w[2] = v2 | (other0 << 16);                               // even: shared dword (odd lanes do not use it)
    } else if constexpr (D == 11) {
        // 88 bits: v0, v1, v2 (24 bits), shifted to the lane's bit offset s inside its first dword
        const uint32_t v0 = c[0] | (c[1] << 11) | (c[2] << 22);
        const uint32_t v1 = (c[2] >> 10) | (c[3] << 1) | (c[4] << 12) | (c[5] << 23);
        const uint32_t v2 = (c[5] >> 9) | (c[6] << 2) | (c[7] << 13);
        const int j = t & 3;
        const unsigned sh = (88u * (unsigned)j) & 31u;              // 0, 24, 16, 8
        const uint32_t e0 = v0 << sh, e1 = k2_shl_hi(v1, v0, sh), e2 = k2_shl_hi(v2, v1, sh), e3 = k2_shl_hi(0u, v2, sh);
        const uint32_t tail = j == 0 ? e2 : e3;                     // the bits in the dword the NEXT lane owns
        const uint32_t prev = k2_prev_in_quad(tail);
        o.w[0] = e0 | (j ? prev : 0u);
        o.w[1] = e1;
        o.w[2] = e2;                                                // lane 0 of the quad: not stored (count = 2)
    } else {
        // 40 bits: v0, v1 (8 bits)
        const uint32_t v0 = c[0] | (c[1] << 5) | (c[2] << 10) | (c[3] << 15) | (c[4] << 20) | (c[5] << 25) | (c[6] << 30);
        const uint32_t v1 = (c[6] >> 2) | (c[7] << 3);
        const int j = t & 3;
        const unsigned sh = 8u * (unsigned)j;
        const uint32_t e0 = v0 << sh, e1 = k2_shl_hi(v1, v0, sh);
        const uint32_t prev = k2_prev_in_quad(e1);
        o.w[0] = e0 | (j ? prev : 0u);
        o.w[1] = e1;                                                // lane 3 of the quad only (count = 2)
    }
}
template <int D>
__device__ __forceinline__ void k2_piece_load(const uint8_t* poly, int t, K2Piece<D>& o) {   // reference bytes, same shape
    const uint32_t* g = reinterpret_cast<const uint32_t*>(poly) + K2Piece<D>::first(t);
#pragma unroll
    for (int i = 0; i < K2Piece<D>::NW; i++) o.w[i] = i < K2Piece<D>::count(t) ? stream_load4(g + i) : 0u;
}
template <int D>
__device__ __forceinline__ void k2_piece_store(uint8_t* poly, int t, const K2Piece<D>& o) {
    uint32_t* g = reinterpret_cast<uint32_t*>(poly) + K2Piece<D>::first(t);
#pragma unroll
    for (int i = 0; i < K2Piece<D>::NW; i++)
        if (i < K2Piece<D>::count(t)) g[i] = o.w[i];
}
template <int D>
__device__ __forceinline__ uint32_t k2_piece_diff(int t, const K2Piece<D>& a, const K2Piece<D>& ref) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < K2Piece<D>::NW; i++)
        if (i < K2Piece<D>::count(t)) d |= a.w[i] ^ ref.w[i];
    return d;
}
// Compress_D + ByteEncode_D of the lane's pairs, then store or compare
template <int D, bool COMPARE, bool BIASED = false>
__device__ __forceinline__ uint32_t k2_emit(const v2f (&p)[4], int t, uint8_t* out, const K2Piece<D>& ref, bool store) {
    unsigned c[8];
    k2_compress4<D, BIASED>(p, c);
    K2Piece<D> o;
    k2_encode<D>(c, t, o);
    if constexpr (COMPARE) return k2_piece_diff<D>(t, o, ref);
    if (store) k2_piece_store<D>(out, t, o);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// k_encrypt2 / encrypt2_body — K-PKE.Encrypt (ml_kem.c:776-936) for two items per wave, given A^T and the PRF bytes of the sampler.
//   COMPARE = false : write c                                   (Encaps_internal, ml_kem.c:1127)
//   COMPARE = true  : compare c' with c, K = (c == c') ? K' : Kbar (Decaps_internal, ml_kem.c:1206-1215)
// ek: packed keys, ek_stride bytes apart (0: one shared key) ; A: matrices, a_stride uint16 apart (0: one shared matrix) ;
// prf: (2K+1) rows of PS bytes per item ; mod_status (may be null): -4 where a t-hat coefficient is >= q (FIPS 203 mode).
// ------------------------------------------------------------------------------------------------
// encrypt2_body: the work of ONE wave -- items item0 and item0 + 1 of the n the pointers describe; `xch` = the wave's exchange
// buffers (K2Lds<K + 1>::xch).  Called by k_encrypt2 (batches) and by the one-workgroup-per-item kernels of mlkem_small.hpp.
template <int K, int ETA1, int DU, int DV, bool COMPARE>
__device__ __forceinline__ void
encrypt2_body(float2 (*xch)[2][128], size_t item0, size_t n, const uint8_t* __restrict__ ek, size_t ek_stride, const uint8_t* __restrict__ msg,
              const uint16_t* __restrict__ A, const uint8_t* __restrict__ prf, uint8_t* __restrict__ c_out, const uint8_t* __restrict__ c_in,
              const uint8_t* __restrict__ Kp, const uint8_t* __restrict__ Kbar, uint8_t* __restrict__ Kout, int32_t* __restrict__ mod_status,
              size_t a_stride) {
    const int l = lane_id(), h = l >> 5, t = l & 31, nb = k2_blk(t);   // nb: the lane's block of 8 coefficients in NAT layout
    const bool valid = item0 + (size_t)h < n;          // n odd: the upper half of the last wave redoes item n - 1, stores nothing
    const unsigned hh = valid ? (unsigned)h : 0u;      // the half's item = item0 + hh; per-lane offsets are 32-bit
    constexpr unsigned PS = (ETA1 == 3) ? 192 : 128, CLEN = 32 * (DU * K + DV), PRFLEN = (2 * K + 1) * PS;
    // wave-uniform bases (SGPRs) + 32-bit per-lane offsets: the upper half's item lies one stride further
    const uint8_t* my_prf = prf + item0 * (size_t)PRFLEN + (size_t)(hh * PRFLEN);
    const uint8_t* my_ek = ek + item0 * ek_stride + (size_t)(hh * (unsigned)ek_stride);
    const uint16_t* my_A = A + item0 * a_stride + (size_t)(hh * (unsigned)a_stride);
    uint8_t* my_c = COMPARE ? nullptr : c_out + item0 * CLEN + (size_t)(hh * CLEN);
    const uint8_t* my_cin = COMPARE ? c_in + item0 * CLEN + (size_t)(hh * CLEN) : nullptr;
    const size_t item = item0 + hh;

    // ---- prologue loads ----
    K2CbdRaw<ETA1> raw_y[K];
#pragma unroll
    for (int b = 0; b < K; b++) k2_cbd_load_la<ETA1>(my_prf + b * PS, t, raw_y[b]);
    // The whole matrix A^T (K x K pieces of 16 bytes per lane) and the t-hat pieces (3 dwords per lane) are requested HERE: with
    // two or three waves per SIMD nothing else hides the HBM latency, and they have the CBD evaluation and the K forward
    // transforms (thousands of cycles) to arrive.  (Requested one row ahead of their use, each row waited for its loads.)
    uint4 a_all[K + 1][K];
#pragma unroll
    for (int a = 0; a < K; a++)
#pragma unroll
        for (int b = 0; b < K; b++) a_all[a][b] = *reinterpret_cast<const uint4*>(my_A + (a * K + b) * 256 + 8 * nb);
#pragma unroll
    for (int b = 0; b < K; b++) {                      // the "row" after the matrix is t-hat: ByteEncode_12 pieces
        const uint32_t* gp = reinterpret_cast<const uint32_t*>(my_ek + 384 * b) + 3 * nb;
        a_all[K][b].x = stream_load4(gp); a_all[K][b].y = stream_load4(gp + 1); a_all[K][b].z = stream_load4(gp + 2); a_all[K][b].w = 0u;
    }
    uint32_t raw_e[K + 1];                             // e1[0..K-1], e2: CBD_2 in NAT layout
#pragma unroll
    for (int a = 0; a <= K; a++) raw_e[a] = k2_cbd_load_nat2(my_prf + (K + a) * PS, nb);
    const unsigned mb = msg[item * 32 + nb];           // the lane's 8 message bits

    // ---- y-hat = NTT(CBD_eta1(PRF(r, n)))  n = 0..K-1   (ml_kem.c:826-836): K transforms interleaved, result in registers ----
    v2f yh[K][4];
    float yg[K][4];                                    // y[2j+1] * gamma_j of every y-hat polynomial stays in registers (4 per polynomial)
    {
        K2Tw tw;
        k2_twiddles_fwd(tw, t);
#pragma unroll
        for (int b = 0; b < K; b++) k2_cbd_eval<ETA1>(raw_y[b], yh[b]);
        k2_ntt<K>(yh, xch, h, t, tw);
#pragma unroll
        for (int b = 0; b < K; b++) {
            k2_fred_n<4>(yh[b]);
            k2_gamma(yh[b], tw.d0, tw.d1, yg[b]);
        }
    }
    // ---- the K + 1 output polynomials (ml_kem.c:854-904):
    //        acc[a] = sum_b A^T[a][b] o y-hat[b] (a < K),  acc[K] = t-hat . y-hat ; the K + 1 inverse transforms interleaved ;
    //        u[a] = . + e1[a] -> Compress_du, ByteEncode_du ;  v = . + e2 + Decompress_1(m) -> Compress_dv, ByteEncode_dv
    //      (running them two at a time to save accumulators keeps y-hat live across a transform and needs MORE registers:
    //       156 instead of 138 at k = 3, profiles/r03_kpke_experiments.txt) ----
    uint32_t diff = 0;
    bool over = false;
    constexpr int NG = K + 1;
    v2f acc[NG][4];
    K2Piece<DU> cu_ref[NG];
    K2Piece<DV> cv_ref;
#pragma unroll
    for (int a = 0; a < NG; a++) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc[a][j] = splat2(0.f);
#pragma unroll
        for (int b = 0; b < K; b++) {
            v2f av[4];
            if (a < K) {
                k2_unpack16(a_all[a][b], av);
            } else {
                k2_decode12(a_all[a][b].x, a_all[a][b].y, a_all[a][b].z, av);   // raw 12-bit values (F3)
#pragma unroll
                for (int j = 0; j < 4; j++) over = over || (av[j].x >= F_Q) || (av[j].y >= F_Q);
            }
            k2_basemul_acc(acc[a], av, yh[b], yg[b]);
        }
    }
    if constexpr (COMPARE) {                           // reference ciphertext pieces: in flight during the inverse transforms
#pragma unroll
        for (int a = 0; a < NG; a++) {
            if (a < K) k2_piece_load<DU>(my_cin + a * 32 * DU, nb, cu_ref[a]);
            else k2_piece_load<DV>(my_cin + K * 32 * DU, nb, cv_ref);
        }
    }
    {
        K2Tw twi;                                      // loaded here, not before the base-case products: 16 registers less at the peak
        k2_twiddles_inv(twi, t);
        k2_intt<NG>(acc, xch, h, t, twi);
    }
#pragma unroll
    for (int a = 0; a < NG; a++) {
        v2f e[4];
        k2_cbd2_eval8<true>(raw_e[a], e);              // e1 / e2 + 2: the bias leaves in Compress's first FMA
#pragma unroll
        for (int j = 0; j < 4; j++) acc[a][j] = acc[a][j] + e[j];
        if (a < K) {
            diff |= k2_emit<DU, COMPARE, true>(acc[a], nb, COMPARE ? nullptr : my_c + a * 32 * DU, cu_ref[a], valid);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)                // Decompress_1(1) = 1665
                acc[a][j] = acc[a][j] + v2f{((mb >> (2 * j)) & 1u) ? 1665.0f : 0.0f, ((mb >> (2 * j + 1)) & 1u) ? 1665.0f : 0.0f};
            diff |= k2_emit<DV, COMPARE, true>(acc[a], nb, COMPARE ? nullptr : my_c + K * 32 * DU, cv_ref, valid);
        }
    }
    if (mod_status) {
        const unsigned long long bal = __ballot(over);
        const bool bad = ((bal >> (32 * h)) & 0xFFFFFFFFull) != 0;
        if (t == 0 && valid) mod_status[item] = bad ? -4 : 0;
    }
    if constexpr (COMPARE) {
        // both candidates are read and blended by mask: neither a branch nor an address depends on whether the ciphertext
        // matched (ml_kem.c:1206-1215 exits at the first mismatch; implicit rejection is meant to hide it)
        const unsigned long long bal = __ballot(diff != 0);
        const uint32_t reject = ((bal >> (32 * h)) & 0xFFFFFFFFull) != 0 ? 0xFFFFFFFFu : 0u;
        if (t < 8 && valid) {
            const uint32_t kp = reinterpret_cast<const uint32_t*>(Kp + item * 32)[t];
            const uint32_t kb = reinterpret_cast<const uint32_t*>(Kbar + item * 32)[t];
            reinterpret_cast<uint32_t*>(Kout + item * 32)[t] = (kp & ~reject) | (kb & reject);
        }
    }
}
// ------------------------------------------------------------------------------------------------
// encrypt1_body — K-PKE.Encrypt of ONE item on one wave: the two half-waves share the item instead of carrying one each.
// Used by the one-workgroup-per-item kernels (mlkem_small.hpp), where a wave's second half would otherwise redo the first
// one's item.  Both halves run the k forward transforms (each needs all of y-hat in its registers); the k + 1 output rows
// -- u[0..k-1] and v -- are dealt out alternately: half h computes rows 2 i + h, i.e. the base-case products, inverse transforms,
// Compress / ByteEncode (or the compare) of ceil((k + 1) / 2) rows instead of k + 1.  Where the two halves' rows differ in kind
// (u: matrix row, d_u ; v: t-hat row, message bits, d_v) the wave runs both code paths one after the other under the halves'
// EXEC masks; for even k the last v row is computed by both halves and stored by the lower one.
// Pointers address the item itself (no item index); `xch` = K2Lds<(K + 2) / 2 or more>::xch of the wave.
// ------------------------------------------------------------------------------------------------
template <int K, int ETA1, int DU, int DV, bool COMPARE>
__device__ __forceinline__ void
encrypt1_body(float2 (*xch)[2][128], const uint8_t* __restrict__ ek, const uint8_t* __restrict__ msg, const uint16_t* __restrict__ A,
              const uint8_t* __restrict__ prf, uint8_t* __restrict__ c_out, const uint8_t* __restrict__ c_in, const uint8_t* __restrict__ Kp,
              const uint8_t* Kbar, uint8_t* __restrict__ Kout, int32_t* __restrict__ mod_status,
              const uint32_t* kbar_flag = nullptr) {   // kbar_flag: counter in LDS that reaches 1 once Kbar is written (hence Kbar not __restrict__)
    const int l = lane_id(), h = l >> 5, t = l & 31, nb = k2_blk(t);
    constexpr unsigned PS = (ETA1 == 3) ? 192 : 128;
    constexpr int NG = K + 1, NL = (NG + 1) / 2;       // rows in all, rows per half
    // kind of local row i: 0 = both halves a matrix row ; 1 = lower half a matrix row, upper half v ; 2 = v in both halves (even k)
    auto kind = [](int i) constexpr { return 2 * i + 1 < K ? 0 : (2 * i < K ? 1 : 2); };

    K2CbdRaw<ETA1> raw_y[K];
#pragma unroll
    for (int b = 0; b < K; b++) k2_cbd_load_la<ETA1>(prf + b * PS, t, raw_y[b]);
    uint4 a_all[NL][K];
    uint32_t raw_e[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int a = kind(i) == 2 ? K : 2 * i + h;    // this half's row
        const bool is_v = kind(i) == 2 || (kind(i) == 1 && h == 1);
#pragma unroll
        for (int b = 0; b < K; b++) {
            if (is_v) {                                // t-hat: ByteEncode_12 pieces
                const uint32_t* gp = reinterpret_cast<const uint32_t*>(ek + 384 * b) + 3 * nb;
                a_all[i][b].x = gp[0]; a_all[i][b].y = gp[1]; a_all[i][b].z = gp[2]; a_all[i][b].w = 0u;
            } else {
                a_all[i][b] = *reinterpret_cast<const uint4*>(A + (a * K + b) * 256 + 8 * nb);
            }
        }
        raw_e[i] = k2_cbd_load_nat2(prf + (K + a) * PS, nb);
    }
    const unsigned mb = msg[nb];                       // the lane's 8 message bits

    v2f yh[K][4];
    float yg[K][4];
    {
        K2Tw tw;
        k2_twiddles_fwd(tw, t);
#pragma unroll
        for (int b = 0; b < K; b++) k2_cbd_eval<ETA1>(raw_y[b], yh[b]);
        k2_ntt<K>(yh, xch, h, t, tw);
#pragma unroll
        for (int b = 0; b < K; b++) {
            k2_fred_n<4>(yh[b]);
            k2_gamma(yh[b], tw.d0, tw.d1, yg[b]);
        }
    }
    uint32_t diff = 0;
    bool over = false;
    v2f acc[NL][4];
    K2Piece<DU> cu_ref[NL];
    K2Piece<DV> cv_ref;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const bool is_v = kind(i) == 2 || (kind(i) == 1 && h == 1);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = splat2(0.f);
#pragma unroll
        for (int b = 0; b < K; b++) {
            v2f av[4];
            if (is_v) {
                k2_decode12(a_all[i][b].x, a_all[i][b].y, a_all[i][b].z, av);   // raw 12-bit values (F3)
#pragma unroll
                for (int j = 0; j < 4; j++) over = over || (av[j].x >= F_Q) || (av[j].y >= F_Q);
            } else {
                k2_unpack16(a_all[i][b], av);
            }
            k2_basemul_acc(acc[i], av, yh[b], yg[b]);
        }
    }
    if constexpr (COMPARE) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int a = kind(i) == 2 ? K : 2 * i + h;
            const bool is_v = kind(i) == 2 || (kind(i) == 1 && h == 1);
            if (is_v) k2_piece_load<DV>(c_in + K * 32 * DU, nb, cv_ref);
            else k2_piece_load<DU>(c_in + a * 32 * DU, nb, cu_ref[i]);
        }
    }
    {
        K2Tw twi;
        k2_twiddles_inv(twi, t);
        k2_intt<NL>(acc, xch, h, t, twi);
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int a = kind(i) == 2 ? K : 2 * i + h;
        const bool is_v = kind(i) == 2 || (kind(i) == 1 && h == 1);
        const bool store = kind(i) != 2 || h == 0;     // even k: the v row exists in both halves
        v2f e[4];
        k2_cbd2_eval8<true>(raw_e[i], e);              // e1 / e2 + 2: the bias leaves in Compress's first FMA
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = acc[i][j] + e[j];
        if (is_v) {
#pragma unroll
            for (int j = 0; j < 4; j++)                // Decompress_1(1) = 1665
                acc[i][j] = acc[i][j] + v2f{((mb >> (2 * j)) & 1u) ? 1665.0f : 0.0f, ((mb >> (2 * j + 1)) & 1u) ? 1665.0f : 0.0f};
            diff |= k2_emit<DV, COMPARE, true>(acc[i], nb, COMPARE ? nullptr : c_out + K * 32 * DU, cv_ref, store);
        } else {
            diff |= k2_emit<DU, COMPARE, true>(acc[i], nb, COMPARE ? nullptr : c_out + a * 32 * DU, cu_ref[i], store);
        }
    }
    if (mod_status) {
        const bool bad = __ballot(over) != 0;
        if (l == 0) mod_status[0] = bad ? -4 : 0;
    }
    if constexpr (COMPARE) {
        // both candidates are read and blended by mask: neither a branch nor an address depends on whether the ciphertext matched
        const uint32_t reject = __ballot(diff != 0) != 0 ? 0xFFFFFFFFu : 0u;
        if (kbar_flag) flag_wait(kbar_flag, 1u);
        if (l < 8) {
            const uint32_t kp = reinterpret_cast<const uint32_t*>(Kp)[l];
            const uint32_t kb = reinterpret_cast<const uint32_t*>(Kbar)[l];
            reinterpret_cast<uint32_t*>(Kout)[l] = (kp & ~reject) | (kb & reject);
        }
    }
}

template <int K, int ETA1, int DU, int DV, bool COMPARE>
__global__ void __launch_bounds__(WAVE * KPKE2_WAVES, kpke2_minwaves(K))
k_encrypt2(size_t n, const uint8_t* __restrict__ ek, size_t ek_stride, const uint8_t* __restrict__ msg, const uint16_t* __restrict__ A,
           const uint8_t* __restrict__ prf, uint8_t* __restrict__ c_out, const uint8_t* __restrict__ c_in, const uint8_t* __restrict__ Kp,
           const uint8_t* __restrict__ Kbar, uint8_t* __restrict__ Kout, int32_t* __restrict__ mod_status, size_t a_stride) {
    __shared__ K2Lds<K + 1> lds_all[KPKE2_WAVES];   // K forward, then K + 1 inverse transforms in flight
    const int wv = wave_id();
    const size_t item0 = 2 * ((size_t)blockIdx.x * KPKE2_WAVES + wv);     // wave-uniform: item bases live in SGPRs
    if (item0 >= n) return;
    encrypt2_body<K, ETA1, DU, DV, COMPARE>(lds_all[wv].xch, item0, n, ek, ek_stride, msg, A, prf, c_out, c_in, Kp, Kbar, Kout, mod_status, a_stride);
}

// canonical representatives in [0, q) of the lane's pairs (any |x| <= 2^24), as integers
__device__ __forceinline__ void k2_canon(const v2f (&p)[4], unsigned (&c)[8]) {
    v2f r[4] = {p[0], p[1], p[2], p[3]};
    k2_fred_n<4>(r);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int a = (int)r[j].x, b = (int)r[j].y;
        c[2 * j] = (unsigned)(a + ((a >> 31) & KQ));
        c[2 * j + 1] = (unsigned)(b + ((b >> 31) & KQ));
    }
}

// ------------------------------------------------------------------------------------------------
// k_keygen2 — K-PKE.KeyGen after G and sampling (ml_kem.c:696-756) + the plain copies of KeyGen_internal
// (ml_kem.c:1054-1062) for two items per wave: ek = Enc12(t-hat) || rho ; dk = Enc12(s-hat) || ek || [H(ek)] || [z]
// A: the sampled matrix (K*K polynomials per item, uint16, natural order) ; prf: 2K rows of PS bytes per item (s then e) ;
// rho: 32 bytes per item.
// ------------------------------------------------------------------------------------------------
// keygen2_body: the work of one wave (items item0, item0 + 1); `xch` = the wave's K2Lds<K>::xch
template <int K, int ETA1, bool KEM_DK>
__device__ __forceinline__ void
keygen2_body(float2 (*xch)[2][128], size_t item0, size_t n, const uint16_t* __restrict__ A, const uint8_t* __restrict__ prf,
             const uint8_t* __restrict__ rho, uint8_t* __restrict__ ek, uint8_t* __restrict__ dk) {
    const int l = lane_id(), h = l >> 5, t = l & 31, nb = k2_blk(t);
    const bool valid = item0 + (size_t)h < n;
    const unsigned hh = valid ? (unsigned)h : 0u;
    constexpr unsigned PS = (ETA1 == 3) ? 192 : 128, EK = 384 * K + 32, DK = KEM_DK ? 768 * K + 96 : 384 * K, PRFLEN = 2 * K * PS;
    const uint8_t* my_prf = prf + item0 * (size_t)PRFLEN + (size_t)(hh * PRFLEN);
    const uint16_t* my_A = A + item0 * (size_t)(K * K * 256) + (size_t)(hh * (unsigned)(K * K * 256));
    uint8_t* my_ek = ek + item0 * EK + (size_t)(hh * EK);
    uint8_t* my_dk = dk + item0 * DK + (size_t)(hh * DK);
    const size_t item = item0 + hh;

    K2CbdRaw<ETA1> raw_s[K], raw_e[K];
#pragma unroll
    for (int b = 0; b < K; b++) k2_cbd_load_la<ETA1>(my_prf + b * PS, t, raw_s[b]);
    constexpr bool PF = K >= 3;                        // k = 2 keeps four waves per SIMD instead (measured: prefetch there costs 5 %)
    uint4 a_all[K][K];                                 // the whole matrix, requested up front (see k_encrypt2)
    if constexpr (PF) {
#pragma unroll
        for (int a = 0; a < K; a++)
#pragma unroll
            for (int b = 0; b < K; b++) a_all[a][b] = *reinterpret_cast<const uint4*>(my_A + (a * K + b) * 256 + 8 * nb);
    }
#pragma unroll
    for (int a = 0; a < K; a++) k2_cbd_load_la<ETA1>(my_prf + (K + a) * PS, t, raw_e[a]);
    const uint32_t rho_w = reinterpret_cast<const uint32_t*>(rho + item * 32)[t & 7];
    K2Tw tw;
    k2_twiddles_fwd(tw, t);
    // s-hat (ml_kem.c:696-706), dk_pke = ByteEncode_12(s-hat) (ml_kem.c:750-756)
    v2f sh[K][4];
    float sg[K][4];
#pragma unroll
    for (int b = 0; b < K; b++) k2_cbd_eval<ETA1>(raw_s[b], sh[b]);
    k2_ntt<K>(sh, xch, h, t, tw);
#pragma unroll
    for (int b = 0; b < K; b++) {
        k2_fred_n<4>(sh[b]);
        k2_gamma(sh[b], tw.d0, tw.d1, sg[b]);
        unsigned c[8];
        k2_canon(sh[b], c);
        K2Piece<12> o;
        k2_encode<12>(c, nb, o);
        if (valid) k2_piece_store<12>(my_dk + 384 * b, nb, o);
    }
    // e-hat (ml_kem.c:710-716): lazy, |x| <= 3 + 7 * 1668
    v2f eh[K][4];
#pragma unroll
    for (int a = 0; a < K; a++) k2_cbd_eval<ETA1>(raw_e[a], eh[a]);
    k2_ntt<K>(eh, xch, h, t, tw);
    // t-hat[a] = sum_b A[a][b] o s-hat[b] + e-hat[a] (ml_kem.c:717-727), ek = ByteEncode_12(t-hat) || rho
#pragma unroll
    for (int a = 0; a < K; a++) {
        v2f acc[4] = {splat2(0.f), splat2(0.f), splat2(0.f), splat2(0.f)};
#pragma unroll
        for (int b = 0; b < K; b++) {
            v2f av[4];
            if constexpr (!PF) a_all[a][b] = *reinterpret_cast<const uint4*>(my_A + (a * K + b) * 256 + 8 * nb);
            k2_unpack16(a_all[a][b], av);
            k2_basemul_acc(acc, av, sh[b], sg[b]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = acc[j] + eh[a][j];   // <= 1665 + 11679
        unsigned c[8];
        k2_canon(acc, c);
        K2Piece<12> o;
        k2_encode<12>(c, nb, o);
        if (valid) {
            k2_piece_store<12>(my_ek + 384 * a, nb, o);
            if constexpr (KEM_DK) k2_piece_store<12>(my_dk + 384 * K + 384 * a, nb, o);
        }
    }
    if (t < 8 && valid) {
        reinterpret_cast<uint32_t*>(my_ek + 384 * K)[t] = rho_w;
        if constexpr (KEM_DK) reinterpret_cast<uint32_t*>(my_dk + 768 * K)[t] = rho_w;
    }
}
template <int K, int ETA1, bool KEM_DK>
__global__ void __launch_bounds__(WAVE * KPKE2_WAVES, K == 2 ? kpke2_minwaves(2) : kpke2_minwaves(K + 1))   // s-hat AND e-hat stay in registers
k_keygen2(size_t n, const uint16_t* __restrict__ A, const uint8_t* __restrict__ prf, const uint8_t* __restrict__ rho,
          uint8_t* __restrict__ ek, uint8_t* __restrict__ dk) {
    __shared__ K2Lds<K> lds_all[KPKE2_WAVES];
    const int wv = wave_id();
    const size_t item0 = 2 * ((size_t)blockIdx.x * KPKE2_WAVES + wv);
    if (item0 >= n) return;
    keygen2_body<K, ETA1, KEM_DK>(lds_all[wv].xch, item0, n, A, prf, rho, ek, dk);
}

}   // namespace mlkem
