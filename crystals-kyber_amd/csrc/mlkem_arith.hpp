// mlkem_arith.hpp — stand-alone polynomial primitives of the C-ABI (MultiplyNTTs, VectorMultiply, PolyAddition /
// PolySubtraction, SamplePolyCBD, Compress / Decompress, ByteEncode / ByteDecode, cell converters) and the CBD evaluation
// shared with the K-PKE kernels.  The K-PKE kernels themselves live in mlkem_kpke2.hpp (KeyGen, Encrypt: two items per wave,
// packed fp32) and mlkem_kpke4.hpp (Decrypt: four items per wave); the stand-alone transforms in mlkem_rntt.hpp.
//
// Here one wavefront owns one polynomial, each lane holds 4 coefficients (NAT layout: 4l..4l+3, which is also the base-case
// multiply pair layout and the HBM layout).  Modular arithmetic runs exactly on the fp32 pipe (mlkem_fntt.hpp); the byte
// codecs work on a wave-private LDS byte buffer.  No workgroup barrier is used: the 4 waves of a workgroup are independent.
#pragma once
#include "mlkem_fntt.hpp"

namespace mlkem {

#ifndef MLKEM_ARITH_WAVES
#define MLKEM_ARITH_WAVES 4
#endif
constexpr int ARITH_WAVES = MLKEM_ARITH_WAVES;   // waves per workgroup (each fully independent)
__device__ __forceinline__ void store_poly_nat(uint16_t* p, const int (&x)[4]) {
    uint2 v;
    v.x = ((uint32_t)x[0] & 0xFFFFu) | ((uint32_t)x[1] << 16);
    v.y = ((uint32_t)x[2] & 0xFFFFu) | ((uint32_t)x[3] << 16);
    *reinterpret_cast<uint2*>(p + 4 * lane_id()) = v;
}
// ---- SamplePolyCBD (ml_kem.c:253-275): evaluation of 4 coefficients from their 8*ETA bits ------------------------------
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
    const Tw fD = ZETA_F.z[64 + lane_id()];   // gamma of the lane's pairs = +-zeta_{64 + lane}
    const size_t stride = (size_t)gridDim.x * ARITH_WAVES;
    for (size_t p = (size_t)blockIdx.x * ARITH_WAVES + wv; p < n; p += stride) {
        float av[4], bv[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
        load_poly_nat_f12(a + p * 256, av);
        load_poly_nat_f12(b + p * 256, bv);
#pragma unroll
        for (int m = 0; m < 4; m++) bv[m] = fred(bv[m]);
        stash_vhat_f(vh_all[wv], vg_all[wv], bv, fD);
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
    const Tw fD = ZETA_F.z[64 + lane_id()];   // gamma of the lane's pairs = +-zeta_{64 + lane}
    const size_t stride = (size_t)gridDim.x * ARITH_WAVES;
    for (size_t p = (size_t)blockIdx.x * ARITH_WAVES + wv; p < n; p += stride) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < k; i++) {
            float av[4], bv[4];
            load_poly_nat_f12(u + (p * (size_t)k + (size_t)i) * 256, av);
            load_poly_nat_f12(v + (p * (size_t)k + (size_t)i) * 256, bv);
#pragma unroll
            for (int m = 0; m < 4; m++) bv[m] = fred(bv[m]);
            stash_vhat_f(vh_all[wv], vg_all[wv], bv, fD);
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
// (no __restrict__: out may be a or b -- every element is read by the thread that then writes it)
__global__ void __launch_bounds__(256) k_poly_addsub(size_t n, const uint16_t* a, const uint16_t* b, uint16_t* out) {
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
__global__ void __launch_bounds__(256) k_compress_values(size_t n, int d, const uint16_t* in, uint16_t* out) {   // out may be in
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
