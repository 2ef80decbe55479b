// tools/valu_ubench.hip — measurement aid (not part of the product): per-SIMD issue rate of the VALU
// instructions the Keccak round is made of (v_bitop3_b32, v_alignbit_b32) against v_xor_b32 / v_add_u32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o gpurun_out/valu_ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256) bench(uint32_t* out, int iters, unsigned long long* cyc) {
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) r[i] = r[i] ^ r[(i + 5) & 15];                                           // v_xor_b32 (VOP2)
                if (KIND == 1) r[i] = __builtin_amdgcn_bitop3_b32(r[i], r[(i + 5) & 15], r[(i + 9) & 15], 0x96);   // xor3
                if (KIND == 2) r[i] = __builtin_amdgcn_alignbit(r[i], r[(i + 5) & 15], 7);             // alignbit, const shift
                if (KIND == 3) r[i] = r[i] + r[(i + 5) & 15];                                           // v_add_u32
                if (KIND == 4) r[i] = __builtin_amdgcn_bitop3_b32(r[i], r[(i + 4) & 15], r[(i + 8) & 15], 0xD2);   // same-bank operands (i, i+4, i+8)
                if (KIND == 5) r[i] = __mul24((int)r[i], (int)r[(i + 5) & 15]);                        // v_mul_i32_i24
                if (KIND == 6) r[i] = r[i] * r[(i + 5) & 15];                                           // v_mul_lo_u32
                if (KIND == 7) r[i] = __builtin_amdgcn_bitop3_b32(r[i], r[(i + 5) & 15], 0x5555aaaau, 0x96);       // 2 VGPR + literal
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
    const int iters = 2000, blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves = 1 per SIMD
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&cyc, (size_t)blocks * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    bench<KIND><<<blocks, 256>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    bench<KIND><<<blocks, 256>>>(out, iters, cyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
    const double insts_per_wave = (double)iters * 64;
    // s_memtime ticks at 100 MHz on this part? report both raw ticks and wall-derived numbers
    double wave_insts_total = insts_per_wave * blocks * 4;
    double per_simd = wave_insts_total / 1024.0;
    printf("%-28s waves/SIMD=%d  wall=%.3f ms  wave-instr/SIMD=%.0f  => %.3f ns per wave-instr per SIMD  (memtime ticks/wave %.0f)\n",
           name, waves_per_simd, ms, per_simd, ms * 1e6 / per_simd, avg);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_xor_b32", w);
        run<1>("v_bitop3_b32 xor3", w);
        run<2>("v_alignbit_b32", w);
        run<3>("v_add_u32", w);
        run<4>("v_bitop3 same-bank srcs", w);
        run<5>("v_mul_i32_i24", w);
        run<6>("v_mul_lo_u32", w);
        run<7>("v_bitop3 2vgpr+literal", w);
    }
    return 0;
}
