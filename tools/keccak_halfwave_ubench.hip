// tools/keccak_halfwave_ubench.hip -- measurement aid: latency of the lane-sliced Keccak-f[1600] for a LONE wave per SIMD when the
// wave has 64, 32 or 16 active lanes (does the SIMD-32 skip the pass of an all-inactive half?): decides whether small batches
// should be spread over more, narrower waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../crystals-kyber_amd/csrc/mlkem_device.hpp"
using namespace mlkem;

__global__ void __launch_bounds__(64, 6) k_perm(uint32_t* out, int perms, int active) {
    if ((int)threadIdx.x >= active) return;
    KeccakState s;
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = threadIdx.x * 2654435761u + i; s.hi[i] = blockIdx.x * 40503u + i; }
    for (int p = 0; p < perms; p++) keccak_f1600(s);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 25; i++) acc ^= s.lo[i] ^ s.hi[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    uint32_t* out;
    (void)hipMalloc(&out, 4096 * 64 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int perms = 64;
    for (int blocks : {256, 1024, 4096})
        for (int active : {64, 32, 16}) {
            k_perm<<<blocks, 64>>>(out, 2, active);
            (void)hipDeviceSynchronize();
            float best = 1e9;
            for (int t = 0; t < 3; t++) {
                (void)hipEventRecord(a);
                k_perm<<<blocks, 64>>>(out, perms, active);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("blocks %5d (%.2f waves per SIMD) active lanes %2d: %.3f ms for %d permutations -> %.2f us per permutation\n",
                   blocks, blocks / 1024.0, active, best, perms, best * 1e3 / perms);
        }
    return 0;
}
