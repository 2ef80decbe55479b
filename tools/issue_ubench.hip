// tools/issue_ubench.hip — measurement aid: VALU issue cost per wave-instruction per SIMD on gfx950 as a function of
// (a) resident waves per SIMD (1..8, set through the dynamic-LDS request of a 64-lane block) and (b) the distance between
// dependent instructions in one wave's stream (DEP = number of independent accumulators: 1 = a serial chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int DEP, int OP>
__global__ void __launch_bounds__(64) k_issue(uint32_t* out, int iters) {
    extern __shared__ uint32_t dyn[];
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 64 / DEP; rep++) {
#pragma unroll
            for (int i = 0; i < DEP; i++) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(r[15]));
                if (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[i]) : "v"(r[15]), "v"(r[14]));
                if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r[i]) : "v"(r[15]));
                if (OP == 3) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(r[15]) : "vcc");
                if (OP == 4) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(r[15]) : "vcc");
                if (OP == 5) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(r[15]) : "vcc");
                if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(r[15]) : "vcc");
                if (OP == 7) asm volatile("v_addc_co_u32_e64 %0, s[20:21], %0, 0, s[22:23]" : "+v"(r[i]) : : "s20", "s21", "s22", "s23");
                if (OP == 8) asm volatile("v_sub_co_u32_e64 %0, s[20:21], %0, %1" : "+v"(r[i]) : "v"(r[15]) : "s20", "s21");
                if (OP == 9) asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1" : : "v"(r[i]), "v"(r[15]) : "s20", "s21");
                if (OP == 10) asm volatile("v_cndmask_b32_e64 %0, 0, 2, s[22:23]" : "+v"(r[i]) : : "s22", "s23");
            }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= r[i];
    if (iters < 0) dyn[threadIdx.x] = acc;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int DEP, int OP>
void run(uint32_t* out, const char* name) {
    auto k = k_issue<DEP, OP>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    printf("%-10s dep-distance %2d :", name, DEP);
    for (int wps = 1; wps <= 8; wps++) {
        const int lds = (160 * 1024) / (4 * wps) - 64, blocks = 1024 * wps * 2, iters = 400;
        k<<<blocks, 64, lds>>>(out, 2);
        (void)hipDeviceSynchronize();
        float best = 1e9;
        for (int t = 0; t < 3; t++) {
            (void)hipEventRecord(a);
            k<<<blocks, 64, lds>>>(out, iters);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf(" %5.2f", best * 1e6 / ((double)blocks / 1024.0 * iters * 64));
    }
    printf("   ns per wave-instruction per SIMD at 1..8 waves/SIMD\n");
}

int main() {
    uint32_t* out;
    (void)hipMalloc(&out, (size_t)1024 * 8 * 2 * 64 * 4);
    run<1, 0>(out, "v_xor"); run<2, 0>(out, "v_xor"); run<4, 0>(out, "v_xor"); run<8, 0>(out, "v_xor"); run<16, 0>(out, "v_xor");
    run<1, 1>(out, "v_bitop3"); run<2, 1>(out, "v_bitop3"); run<4, 1>(out, "v_bitop3"); run<8, 1>(out, "v_bitop3");
    run<1, 2>(out, "v_alignbit"); run<2, 2>(out, "v_alignbit"); run<4, 2>(out, "v_alignbit"); run<8, 2>(out, "v_alignbit");
    run<8, 3>(out, "v_sub_co"); run<8, 4>(out, "v_addc_co"); run<8, 5>(out, "v_cmp_gt"); run<8, 6>(out, "v_cndmask");
    run<8, 7>(out, "addc_e64"); run<8, 8>(out, "sub_co_e64"); run<8, 9>(out, "cmp_e64"); run<8, 10>(out, "cndmask_e64");
    return 0;
}
