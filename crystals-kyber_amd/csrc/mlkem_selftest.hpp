// mlkem_selftest.hpp — on-device exhaustive checks of the exact-integer arithmetic the kernels run on the fp32 pipe
// (mlkem_fntt.hpp, mlkem_arith.hpp), against plain integer `% q` forms (ml_kem.c:83-97, :253-275, :287-442).
//
// The exactness argument (all intermediates are integers below 2^24, so v_fma_f32 / v_mul_f32 / v_add_f32 are exact; the
// magic-number rint; the reciprocal-multiply quotients) is proven on the host by the emulator tests.  These kernels prove it
// on gfx950 itself — the hardware FMA, v_cvt_f32_ubyteN, hipcc's constant folding and contraction choices — over the FULL
// input domain of each helper.  Every sweep adds its violations to one 64-bit counter; zero means the property holds.
//   sweep 0  fred(x)                                   every integer |x| <= 2^24
//   sweep 1  fmulmod_shoup / fmulmod (both forms)      (+-128 zetas, +-128^-1, +-zeta_1/128) x every |b| <= 10082
//   sweep 2  compress_f<D>, D = 1..11                  every |x| <= 4095 (any representative)
//   sweep 3  cbd_eval_f<2>                             all 2^16 lane inputs
//   sweep 4  cbd_eval_f<3>                             all 2^24 lane inputs
//   sweep 5  basemul_term (the body of basemul_acc_f)  every a in [0,4095] x every |y| <= 1665 with the other product
//                                                      and the accumulator pinned at each corner of their ranges
//   sweep 6  fcanon(x), fcanon_floor(x)                every |x| <= 2^20 (outputs canonical in [0, q))
//   sweep 7  fmulmod_shoup by 1 (the reduction of the cross-lane butterflies, mlkem_rntt.hpp)   every |x| <= 2^22
//   sweep 8  the PACKED forms (v_pk_fma_f32 / v_pk_add_f32: mlkem_rntt.hpp, mlkem_kpke2.hpp) against the integer forms, both
//            halves of every pair: fred2 every |x| <= 2^24 ; fmulmod_shoup2 over the 260 multipliers x every |b| <= 10082 ;
//            k2_compress4<D>, D in {4, 5, 10, 11}, every |x| <= 4095 ; k2_canon every |x| <= 2^20 ;
//            k2_basemul_acc at the corners of its bound over every (a0, y0) ; k2_cbd2_eval8 over every 16-bit half pattern in
//            both halves of the dword (a nibble's result does not depend on the other nibbles: 2^16 x 2 positions x 3 fillers)
#pragma once
#include "mlkem_arith.hpp"
#include "mlkem_rntt.hpp"
#include "mlkem_kpke2.hpp"

namespace mlkem {

__device__ __forceinline__ int imod_q(long long v) {   // canonical v mod q for any sign
    int r = (int)(v % KQ);
    return r < 0 ? r + KQ : r;
}
__device__ __forceinline__ bool centred_ok(float t, long long exact) {   // t integral, |t| <= 1665, t == exact mod q
    const int ti = (int)t;
    return (float)ti == t && ti <= 1665 && ti >= -1665 && imod_q(exact - ti) == 0;
}

template <int D>
__device__ __forceinline__ unsigned compress_ref(int x) {   // ml_kem.c:83-97 on the canonical representative
    const unsigned c = (unsigned)imod_q(x);
    const unsigned num = c << D, quo = num / (unsigned)KQ, rem = num % (unsigned)KQ;
    return (quo + (rem > (unsigned)(KQ / 2) ? 1u : 0u)) & ((1u << D) - 1u);
}
template <int D>
__device__ __forceinline__ unsigned compress_bad(int x) { return compress_f<D>((float)x) != compress_ref<D>(x) ? 1u : 0u; }

__global__ void __launch_bounds__(256) k_selftest(int sweep, unsigned long long* __restrict__ bad_out) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    if (sweep == 0) {
        for (long long i = (long long)tid; i <= (1ll << 25); i += (long long)nthreads) {
            const long long x = i - (1ll << 24);
            if (!centred_ok(fred((float)x), x)) bad++;
        }
    } else if (sweep == 1) {
        const long long per = 2 * 10082 + 1, total = 260 * per;
        for (long long i = (long long)tid; i < total; i += (long long)nthreads) {
            const int wi = (int)(i / per), b = (int)(i % per) - 10082;
            Tw w = (wi >> 1) < 128 ? ZETA_F.z[wi >> 1] : (wi >> 1) == 128 ? F_INV128 : F_INV128_Z1;
            if (wi & 1) w = tw_neg(w);
            const long long exact = (long long)(int)w.z * b;
            if (!centred_ok(fmulmod_shoup(w, (float)b), exact)) bad++;
            if (!centred_ok(fmulmod(w.z, (float)b), exact)) bad++;
            if (!centred_ok(fmulmod(w, (float)b), exact)) bad++;
        }
    } else if (sweep == 2) {
        for (long long i = (long long)tid; i < 8191; i += (long long)nthreads) {
            const int x = (int)i - 4095;
            bad += compress_bad<1>(x) + compress_bad<2>(x) + compress_bad<3>(x) + compress_bad<4>(x) + compress_bad<5>(x) +
                   compress_bad<6>(x) + compress_bad<7>(x) + compress_bad<8>(x) + compress_bad<9>(x) + compress_bad<10>(x) +
                   compress_bad<11>(x);
        }
    } else if (sweep == 3) {
        for (uint32_t t = (uint32_t)tid; t < 65536u; t += (uint32_t)nthreads) {
            float x[4];
            cbd_eval_f<2>(t, x);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t n = (t >> (4 * m)) & 15u;
                const int want = (int)((n & 1) + ((n >> 1) & 1)) - (int)(((n >> 2) & 1) + ((n >> 3) & 1));
                if (x[m] != (float)want) bad++;
            }
        }
    } else if (sweep == 4) {
        for (size_t t = tid; t < ((size_t)1 << 24); t += nthreads) {
            float x[4];
            cbd_eval_f<3>((uint32_t)t, x);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t n = ((uint32_t)t >> (6 * m)) & 63u;
                const int want = (int)__builtin_popcount(n & 7u) - (int)__builtin_popcount(n >> 3);
                if (x[m] != (float)want) bad++;
            }
        }
    } else if (sweep == 5) {
        // r = fred(a1 * g + a0 * y + acc): every (a0, y); (a1, g, acc) at the 8 corners {0, 4095} x {-1665, 1665} x {-1665, 1665}
        const long long ny = 2 * 1665 + 1, total = 4096 * ny * 8;
        for (long long i = (long long)tid; i < total; i += (long long)nthreads) {
            const int corner = (int)(i & 7);
            const long long j = i >> 3;
            const int a0 = (int)(j / ny), y = (int)(j % ny) - 1665;
            const int a1 = (corner & 1) ? 4095 : 0, g = (corner & 2) ? 1665 : -1665, acc = (corner & 4) ? 1665 : -1665;
            const long long exact = (long long)a1 * g + (long long)a0 * y + acc;
            if (!centred_ok(basemul_term((float)acc, (float)a0, (float)y, (float)a1, (float)g), exact)) bad++;
            // and with the roles of the two products exchanged
            if (!centred_ok(basemul_term((float)acc, (float)a1, (float)g, (float)a0, (float)y), exact)) bad++;
        }
    } else if (sweep == 6) {
        for (long long i = (long long)tid; i <= (1ll << 21); i += (long long)nthreads) {
            const long long x = i - (1ll << 20);
            const int want = imod_q(x);
            if (fcanon((float)x) != want) bad++;
            const float c = fcanon_floor((float)x);
            if (c != (float)want) bad++;
        }
    }
    else if (sweep == 7) {
        for (long long i = (long long)tid; i <= (1ll << 23); i += (long long)nthreads) {
            const long long x = i - (1ll << 22);
            if (!centred_ok(fmulmod_shoup(TW_ONE, (float)x), x)) bad++;
        }
    }
    else if (sweep == 8) {
        // a pair carries (x, -x) or two neighbouring operands, so both halves of the packed instruction see the whole domain
        for (long long i = (long long)tid; i <= (1ll << 24); i += (long long)nthreads) {
            const v2f r = fred2(v2f{(float)i, (float)-i});
            if (!centred_ok(r.x, i) || !centred_ok(r.y, -i)) bad++;
        }
        const long long per = 2 * 10082 + 1, total = 260 * per;
        for (long long i = (long long)tid; i < total; i += (long long)nthreads) {
            const int wi = (int)(i / per), b = (int)(i % per) - 10082;
            Tw w = (wi >> 1) < 128 ? ZETA_F.z[wi >> 1] : (wi >> 1) == 128 ? F_INV128 : F_INV128_Z1;
            if (wi & 1) w = tw_neg(w);
            const v2f r = fmulmod_shoup2(w, v2f{(float)b, (float)-b});
            if (!centred_ok(r.x, (long long)(int)w.z * b) || !centred_ok(r.y, -(long long)(int)w.z * b)) bad++;
        }
        for (long long i = (long long)tid; i <= 8190; i += (long long)nthreads) {
            const int x = (int)i - 4095;
            const v2f in[4] = {v2f{(float)x, (float)-x}, v2f{(float)x, (float)x}, v2f{(float)-x, (float)x}, v2f{0.f, (float)x}};
            unsigned c[8];
            k2_compress4<4>(in, c);
            if (c[0] != compress_ref<4>(x) || c[1] != compress_ref<4>(-x) || c[2] != c[0] || c[3] != c[0] || c[4] != c[1] || c[7] != c[0]) bad++;
            k2_compress4<5>(in, c);
            if (c[0] != compress_ref<5>(x) || c[1] != compress_ref<5>(-x) || c[5] != c[0]) bad++;
            k2_compress4<10>(in, c);
            if (c[0] != compress_ref<10>(x) || c[1] != compress_ref<10>(-x) || c[3] != c[0]) bad++;
            k2_compress4<11>(in, c);
            if (c[0] != compress_ref<11>(x) || c[1] != compress_ref<11>(-x) || c[4] != c[1]) bad++;
            // the biased form (input = value + 2, the -2 folded into the first FMA): what K-PKE.Encrypt runs
            const v2f inb[4] = {v2f{(float)(x + 2), (float)(2 - x)}, v2f{(float)(x + 2), (float)(x + 2)}, v2f{(float)(2 - x), (float)(x + 2)}, v2f{2.f, (float)(x + 2)}};
            k2_compress4<4, true>(inb, c);
            if (c[0] != compress_ref<4>(x) || c[1] != compress_ref<4>(-x) || c[2] != c[0] || c[4] != c[1] || c[6] != 0u || c[7] != c[0]) bad++;
            k2_compress4<5, true>(inb, c);
            if (c[0] != compress_ref<5>(x) || c[1] != compress_ref<5>(-x) || c[5] != c[0]) bad++;
            k2_compress4<10, true>(inb, c);
            if (c[0] != compress_ref<10>(x) || c[1] != compress_ref<10>(-x) || c[3] != c[0]) bad++;
            k2_compress4<11, true>(inb, c);
            if (c[0] != compress_ref<11>(x) || c[1] != compress_ref<11>(-x) || c[4] != c[1]) bad++;
        }
        for (long long i = (long long)tid; i <= (1ll << 20); i += (long long)nthreads) {
            const v2f in[4] = {v2f{(float)i, (float)-i}, v2f{(float)-i, (float)i}, v2f{(float)i, (float)i}, v2f{0.f, (float)-i}};
            unsigned c[8];
            k2_canon(in, c);
            const unsigned p = (unsigned)imod_q(i), m = (unsigned)imod_q(-i);
            if (c[0] != p || c[1] != m || c[2] != m || c[3] != p || c[4] != p || c[5] != p || c[6] != 0u || c[7] != m) bad++;
        }
        for (long long i = (long long)tid; i < (1ll << 16) * 6; i += (long long)nthreads) {
            const uint32_t v = (uint32_t)(i & 0xFFFF), fill = (uint32_t)((i >> 16) % 3) * 0x5A5Au + ((i >> 16) % 3 == 2 ? 0xFFFFu - 2 * 0x5A5Au : 0u);
            const bool upper = (i >> 16) >= 3;
            const uint32_t t = upper ? (v << 16) | fill : v | (fill << 16);
            v2f p[4], pb[4];
            k2_cbd2_eval8<false>(t, p);
            k2_cbd2_eval8<true>(t, pb);
            for (int j = 0; j < 4; j++)
                if (pb[j].x != p[j].x + 2.0f || pb[j].y != p[j].y + 2.0f) bad++;
            const float got[8] = {p[0].x, p[0].y, p[1].x, p[1].y, p[2].x, p[2].y, p[3].x, p[3].y};
            for (int c = 0; c < 8; c++) {
                const uint32_t nib = (t >> (4 * c)) & 0xFu;
                const int want = (int)((nib & 1u) + ((nib >> 1) & 1u)) - (int)(((nib >> 2) & 1u) + ((nib >> 3) & 1u));
                if (got[c] != (float)want) bad++;
            }
        }
        const long long na = 4096, ny = 2 * 1665 + 1;
        for (long long i = (long long)tid; i < na * ny * 8; i += (long long)nthreads) {
            const int corner = (int)(i / (na * ny));
            const long long r = i % (na * ny);
            const int a0 = (int)(r / ny), y0 = (int)(r % ny) - 1665;
            const int a1 = (corner & 1) ? 4095 : 0, g = (corner & 2) ? 1665 : -1665, ac = (corner & 4) ? 1665 : -1665;
            // pair j: (acc0, acc1) + (a0, a1) o (y0, y1) with y1 * gamma = g:  c0 = a0 y0 + a1 g + acc ; c1 = a0 y1 + a1 y0 + acc
            v2f acc[4], a[4], y[4];
            float yg[4];
            for (int j = 0; j < 4; j++) { acc[j] = v2f{(float)ac, (float)ac}; a[j] = v2f{(float)a0, (float)a1}; y[j] = v2f{(float)y0, (float)g}; yg[j] = (float)g; }
            k2_basemul_acc(acc, a, y, yg);
            const long long e0 = (long long)a0 * y0 + (long long)a1 * g + ac, e1 = (long long)a0 * g + (long long)a1 * y0 + ac;
            if (!centred_ok(acc[0].x, e0) || !centred_ok(acc[0].y, e1) || !centred_ok(acc[3].x, e0) || !centred_ok(acc[3].y, e1)) bad++;
        }
    }
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(bad_out, bad);
}

constexpr int SELFTEST_SWEEPS = 9;

}   // namespace mlkem
