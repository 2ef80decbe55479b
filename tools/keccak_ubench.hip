// tools/keccak_ubench.hip — measurement aid: cost of the lane-sliced Keccak-f[1600] of mlkem_device.hpp per round and
// per wave as a function of the number of resident waves per SIMD (set through a dynamic-LDS request per 64-lane block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../crystals-kyber_amd/csrc/mlkem_device.hpp"
using namespace mlkem;

#ifndef UB_MINWAVES
#define UB_MINWAVES 1
#endif
__global__ void __launch_bounds__(64, UB_MINWAVES) k_perm(uint32_t* out, int perms) {
    extern __shared__ uint32_t dyn[];
    KeccakState s;
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = threadIdx.x * 2654435761u + i; s.hi[i] = blockIdx.x * 40503u + i; }
    for (int p = 0; p < perms; p++) keccak_f1600(s);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 25; i++) acc ^= s.lo[i] ^ s.hi[i];
    if (perms < 0) dyn[threadIdx.x] = acc;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// shader clock actually delivered: cycles of s_memtime (shader clock) per tick of s_memrealtime (100 MHz) over the kernel
__global__ void __launch_bounds__(64, UB_MINWAVES) k_perm_clk(uint32_t* out, int perms, unsigned long long* clk) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    KeccakState s;
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = threadIdx.x * 2654435761u + i; s.hi[i] = blockIdx.x * 40503u + i; }
    for (int p = 0; p < perms; p++) keccak_f1600(s);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 25; i++) acc ^= s.lo[i] ^ s.hi[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
    uint32_t* out;
    const int max_blocks = 1024 * 8 * 4;   // largest launch below: 8 waves per SIMD x 4 rounds
    hipMalloc(&out, (size_t)max_blocks * 64 * 4);
    hipFuncSetAttribute((const void*)k_perm, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int perms = 64;
    for (int wps = 1; wps <= 8; wps++) {
        // waves per CU = 4 * wps  ->  LDS per block = 160 KB / (4 * wps) (minus a little so exactly that many fit)
        const int lds = (160 * 1024) / (4 * wps) - 64;
        const int rounds_of_waves = 4;   // every SIMD gets wps resident waves, 4 times over
        const int blocks = 1024 * wps * rounds_of_waves;
        if (blocks > max_blocks) return 1;
        k_perm<<<blocks, 64, lds>>>(out, 2);
        hipDeviceSynchronize();
        float best = 1e9;
        for (int t = 0; t < 3; t++) {
            hipEventRecord(a);
            k_perm<<<blocks, 64, lds>>>(out, perms);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        const double wave_rounds_per_simd = (double)blocks / 1024.0 * perms * 24;
        printf("waves/SIMD %d (lds %6d B/wave): %.3f ms  -> %.1f ns per wave-round per SIMD, %.2f us per permutation-wave\n", wps, lds, best,
               best * 1e6 / wave_rounds_per_simd, best * 1e3 / wave_rounds_per_simd * 24);
    }
    // sustained load: ~4 s of back-to-back launches at 6 waves per SIMD; time per launch and the delivered shader clock
    unsigned long long* clk;
    (void)hipMallocManaged(&clk, 16);
    const int blocks = 1024 * 6 * 4;
    for (int rep = 0; rep < 40; rep++) {
        float ms_sum = 0;
        for (int t = 0; t < 10; t++) {
            (void)hipEventRecord(a);
            k_perm_clk<<<blocks, 64>>>(out, perms, clk);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            ms_sum += ms;
        }
        const double wave_rounds_per_simd = (double)blocks / 1024.0 * perms * 24 * 10;
        if (rep % 4 == 0)
            printf("sustained rep %2d: %.1f ns per wave-round per SIMD, shader clock %.0f MHz\n", rep, ms_sum * 1e6 / wave_rounds_per_simd,
                   (double)clk[0] / (double)clk[1] * 100.0);
    }
    return 0;
}
