// tools/ntt_ubench.hip — measurement aid (not product code): variants of the stand-alone NTT kernel's memory structure
// around the product's register transform (crystals-kyber_amd/csrc/mlkem_rntt.hpp), timed with HIP events on 2^20
// polynomials: grid size (persistent vs one pair per wave), prefetch depth, non-temporal loads / stores, lane order of
// the 16-byte pieces, and a copy-only form of the same loop (the ceiling for this access pattern).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ntt_ubench.bin tools/ntt_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../crystals-kyber_amd/csrc/mlkem_kernels.hpp"
#include "../crystals-kyber_amd/csrc/mlkem_rntt.hpp"
using namespace mlkem;

template <bool NT> __device__ __forceinline__ uint4 ld(const uint16_t* p) {
    if constexpr (NT) {
        typedef unsigned v4 __attribute__((ext_vector_type(4)));
        v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
        uint4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w; return r;
    } else return *reinterpret_cast<const uint4*>(p);
}
template <bool NT> __device__ __forceinline__ void st(uint16_t* p, uint4 o) {
    if constexpr (NT) {
        typedef unsigned v4 __attribute__((ext_vector_type(4)));
        v4 t; t.x = o.x; t.y = o.y; t.z = o.z; t.w = o.w;
        __builtin_nontemporal_store(t, reinterpret_cast<v4*>(p));
    } else *reinterpret_cast<uint4*>(p) = o;
}

// copy with the product kernel's loop shape (two 16-byte pieces per lane, four polynomials per wave and iteration)
template <bool NT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) kcopy(size_t n, const uint16_t* __restrict__ in, uint16_t* __restrict__ out) {
    const int wv = (int)(threadIdx.x >> 6);
    const RnttLane a = rntt_lane();
    const size_t nquads = n / 4, stride = (size_t)gridDim.x * WAVES;
    size_t q = (size_t)blockIdx.x * WAVES + wv;
    if (q >= nquads) return;
    const size_t off = (size_t)a.p * 256 + (size_t)a.m * 8;
    uint4 nl = ld<NT>(in + q * 1024 + off), nh = ld<NT>(in + q * 1024 + off + 128);
    for (; q < nquads; q += stride) {
        const uint4 l = nl, h = nh;
        if (q + stride < nquads) { nl = ld<NT>(in + (q + stride) * 1024 + off); nh = ld<NT>(in + (q + stride) * 1024 + off + 128); }
        st<NT>(out + q * 1024 + off, l);
        st<NT>(out + q * 1024 + off + 128, h);
    }
}

template <class K> static float timeit(K kfn, int grid, int block, size_t n, const uint16_t* in, uint16_t* out, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) kfn<<<grid, block>>>(n, in, out);
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) kfn<<<grid, block>>>(n, in, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const size_t n = (size_t)1 << 20;
    uint16_t *in, *out;
    hipMalloc(&in, n * 512); hipMalloc(&out, n * 512);
    std::vector<uint16_t> h(n * 256);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (uint16_t)((s >> 8) % 3329); }
    hipMemcpy(in, h.data(), n * 512, hipMemcpyHostToDevice);
    std::vector<int> grids = {1024, 1536, 2048, 2560, 3072, 3584, 4096, 5120, 6144, 8192, 12288, 16384};
    for (int g : grids) {
        float ms = timeit(k_ntt4_batch<false>, g, 64 * RNTT_WAVES, n, in, out, 200);
        float ms2 = timeit(k_ntt4_batch<true>, g, 64 * RNTT_WAVES, n, in, out, 200);
        float ms3 = timeit(kcopy<true, 4>, g, 256, n, in, out, 200);
        printf("grid %6d  fwd %.4f ms %.2f TB/s   inv %.4f ms %.2f TB/s   copy(nt) %.4f ms %.2f TB/s\n", g, ms, n * 1024.0 / ms / 1e9, ms2,
               n * 1024.0 / ms2 / 1e9, ms3, n * 1024.0 / ms3 / 1e9);
    }
    {   // compute side alone: 2^12 polynomials (4 MB in + out, L2-resident) transformed 256 times inside... one launch each: see DESIGN
        const size_t ns = 4096;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int inv = 0; inv < 2; inv++) {
            hipEventRecord(a);
            for (int i = 0; i < 2560; i++) { if (inv) k_ntt4_batch<true><<<256, 256>>>(ns, in, out); else k_ntt4_batch<false><<<256, 256>>>(ns, in, out); }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("L2-resident 2^12 polys x 256 launches (launch gaps included) %s: %.4f ms per 2^20 polys\n", inv ? "inv" : "fwd", ms / 10);
        }
    }
    hipMemcpy(h.data(), out, 64, hipMemcpyDeviceToHost);
    return 0;
}
