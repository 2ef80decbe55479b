// tools/valu_ubench3.hip — measurement aid: per-SIMD issue cost of the cross-lane instructions of mlkem_rntt.hpp
// (v_permlane32/16_swap, v_fmac_f32_dpp with the three DPP modifiers used, s_nop) next to plain VALU, 8 waves per SIMD,
// 16 independent accumulators, inline asm so the opcode is exact.   hipcc --offload-arch=gfx950 -O3 -o tools/valu_ubench3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define BODY(ASM)                                                                                           \
    for (int it = 0; it < iters; it++) {                                                                    \
        _Pragma("unroll") for (int rep = 0; rep < 4; rep++) {                                               \
            _Pragma("unroll") for (int i = 0; i < 16; i++) {                                                \
                asm volatile(ASM : "+v"(r[i]), "+v"(r[(i + 5) & 15]) : "v"(r[(i + 9) & 15]));               \
            }                                                                                               \
        }                                                                                                   \
    }
#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                                    \
        float r[16];                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 16; i++) r[i] = (float)((threadIdx.x * 7 + i * 13 + blockIdx.x) & 1023) * 1e-3f; \
        BODY(ASM)                                                                                           \
        float acc = 0;                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 16; i++) acc += r[i];                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                   \
    }
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_fmac_dpp_q1, "v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_fmac_dpp_q2, "v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
KERNEL(k_fmac_dpp_hm, "v_fmac_f32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf")
KERNEL(k_fmac_dpp_self, "v_fmac_f32_dpp %0, %0, %2 row_half_mirror row_mask:0xf bank_mask:0xf")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_add_dpp, "v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_perm32, "v_permlane32_swap_b32 %0, %1")
KERNEL(k_perm16, "v_permlane16_swap_b32 %0, %1")
KERNEL(k_nop_fma, "s_nop 1\n\tv_fma_f32 %0, %0, %1, %2")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_floor, "v_floor_f32 %0, %0")
KERNEL(k_cvt_u32, "v_cvt_u32_f32 %0, %0")
KERNEL(k_cvt_f32, "v_cvt_f32_u32 %0, %0")
KERNEL(k_fract, "v_fract_f32 %0, %0")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_cmp, "v_cmp_ge_f32 vcc, %0, %1")
KERNEL(k_perm_b32, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1")
KERNEL(k_cvt_pk_u16_u32, "v_cvt_pk_u16_u32 %0, %0, %1")
KERNEL(k_cvt_pknorm_u16, "v_cvt_pknorm_u16_f32 %0, %0, %1")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 16, %1")

typedef void (*kfn_t)(float*, int);
double run(const char* name, kfn_t k, double ref_ns) {
    const int iters = 1000, blocks = 256 * 8;
    static float* out = nullptr;
    if (!out) hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int t = 0; t < 3; t++) {
        hipEventRecord(a);
        k<<<blocks, 256>>>(out, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double per_simd = (double)iters * 64 * blocks * 4 / 1024.0;
    const double ns = best * 1e6 / per_simd;
    printf("%-22s %.3f ns/wave-instr/SIMD  = %.2f x v_fma_f32\n", name, ns, ref_ns > 0 ? ns / ref_ns : 1.0);
    return ns;
}
#define RUN(k) run(#k, k, ref)
int main() {
    double ref = run("k_fma", k_fma, 0);
    RUN(k_fmac); RUN(k_fmac_dpp_q1); RUN(k_fmac_dpp_q2); RUN(k_fmac_dpp_hm); RUN(k_fmac_dpp_self); RUN(k_mov_dpp); RUN(k_add_dpp);
    RUN(k_perm32); RUN(k_perm16); RUN(k_nop_fma); RUN(k_max3); RUN(k_floor); RUN(k_cvt_u32); RUN(k_cvt_f32); RUN(k_fract); RUN(k_med3);
    RUN(k_cmp); RUN(k_perm_b32); RUN(k_ldexp); RUN(k_cvt_pk_u16_u32); RUN(k_cvt_pknorm_u16); RUN(k_and_or); RUN(k_lshl_or);
    run("k_fma(again)", k_fma, 0);
    return 0;
}
