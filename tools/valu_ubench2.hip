// tools/valu_ubench2.hip — measurement aid: per-SIMD issue cost (ns and cycles at the measured clock) of single
// VALU instructions on gfx950, 8 waves per SIMD, 16 independent accumulators, inline asm so the opcode is exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define BODY(ASM)                                                                                           \
    for (int it = 0; it < iters; it++) {                                                                    \
        _Pragma("unroll") for (int rep = 0; rep < 4; rep++) {                                               \
            _Pragma("unroll") for (int i = 0; i < 16; i++) {                                                \
                asm volatile(ASM : "+v"(r[i]) : "v"(r[(i + 5) & 15]), "v"(r[(i + 9) & 15]));                \
            }                                                                                               \
        }                                                                                                   \
    }

#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) {                                 \
        uint32_t r[16];                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x; \
        BODY(ASM)                                                                                           \
        uint32_t acc = 0;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 16; i++) acc ^= r[i];                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                   \
    }

KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(k_alignbit_c, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL(k_alignbit_v, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 7, %0")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 7, %1")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_bfe_u, "v_bfe_u32 %0, %0, 3, 12")
KERNEL(k_bfe_i, "v_bfe_i32 %0, %0, 0, 16")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL(k_min, "v_min_u32 %0, %0, %1")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 16, %0")
KERNEL(k_and, "v_and_b32 %0, 0xfff, %0")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mul_i24, "v_mul_i32_i24 %0, %0, %1")
KERNEL(k_mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
KERNEL(k_mad_u16, "v_mad_u16 %0, %0, %1, %2")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
KERNEL(k_pk_ashr_i16, "v_pk_ashrrev_i16 %0, 15, %0")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL(k_rndne_f32, "v_rndne_f32 %0, %0")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL(k_mad_i32_i16, "v_mad_i32_i16 %0, %0, %1, %2")
KERNEL(k_dot2_i32_i16, "v_dot2_i32_i16 %0, %0, %1, %2")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_sad, "v_sad_u32 %0, %0, %1, %2")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_mul_hi_i32, "v_mul_hi_i32 %0, %0, %1")
KERNEL(k_fmamk, "v_fmamk_f32 %0, %0, 0x39a02c79, %1")
KERNEL(k_fmaak, "v_fmaak_f32 %0, %0, %1, 0x4b400000")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
#define KERNEL64(NAME, ASM)                                                                                 \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) {                                 \
        unsigned long long r[16];                                                                           \
        _Pragma("unroll") for (int i = 0; i < 16; i++) r[i] = (threadIdx.x * 2654435761ull + i * 40503u + blockIdx.x) * 0x100000001ull; \
        BODY(ASM)                                                                                           \
        unsigned long long acc = 0;                                                                         \
        _Pragma("unroll") for (int i = 0; i < 16; i++) acc ^= r[i];                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));                         \
    }
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL(k_fma_sgpr, "v_fma_f32 %0, %0, s4, %1")
KERNEL(k_add_lit, "v_add_f32 %0, 0x4b400000, %0")
KERNEL(k_cvt_sdwa, "v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
KERNEL(k_cmp_cnd, "v_cmp_gt_f32 vcc, 0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL(k_min_f32, "v_min_f32 %0, %0, %1")
KERNEL(k_floor_f32, "v_floor_f32 %0, %0")
KERNEL(k_xor_lit, "v_xor_b32 %0, 0x12345678, %0")
KERNEL(k_and_or2, "v_and_b32 %0, %0, %1\n\tv_or_b32 %0, %0, %2")

typedef void (*kfn_t)(uint32_t*, int);
double run(const char* name, kfn_t k, double ref_ns) {
    const int iters = 1000, blocks = 256 * 8;
    static uint32_t* out = nullptr;
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
    printf("%-22s %.3f ns/wave-instr/SIMD  = %.2f x v_xor\n", name, ns, ref_ns > 0 ? ns / ref_ns : 1.0);
    return ns;
}
#define RUN(k) run(#k, k, ref)
int main() {
    double ref = run("k_xor", k_xor, 0);
    RUN(k_bitop3); RUN(k_alignbit_c); RUN(k_alignbit_v); RUN(k_alignbyte); RUN(k_perm); RUN(k_lshl); RUN(k_lshl_or); RUN(k_and_or);
    RUN(k_or3); RUN(k_bfe_u); RUN(k_bfe_i); RUN(k_lshl_add); RUN(k_add3); RUN(k_add); RUN(k_sub); RUN(k_min); RUN(k_ashr); RUN(k_and);
    RUN(k_mul_u24); RUN(k_mul_i24); RUN(k_mul_hi_u24); RUN(k_mad_u24); RUN(k_mad_i24); RUN(k_mul_lo); RUN(k_mul_hi); RUN(k_mul_hi_i32);
    RUN(k_mul_lo_u16); RUN(k_mad_u16); RUN(k_pk_add_u16); RUN(k_pk_mul_lo_u16); RUN(k_pk_mad_u16); RUN(k_pk_sub_i16); RUN(k_pk_min_i16);
    RUN(k_pk_ashr_i16); RUN(k_fma_f32); RUN(k_mul_f32); RUN(k_add_f32); RUN(k_rndne_f32); RUN(k_cvt_f32_i32); RUN(k_cvt_i32_f32);
    RUN(k_mad_i32_i16); RUN(k_dot2_i32_i16); RUN(k_mov); RUN(k_sad); RUN(k_xad);
    RUN(k_fmamk); RUN(k_fmaak); RUN(k_fmac); RUN(k_pk_fma_f32); RUN(k_pk_add_f32); RUN(k_pk_mul_f32); RUN(k_fma_sgpr); RUN(k_add_lit);
    RUN(k_cvt_sdwa); RUN(k_cmp_cnd); RUN(k_max_f32); RUN(k_min_f32); RUN(k_floor_f32); RUN(k_xor_lit); RUN(k_and_or2);
    ref = run("k_xor(again)", k_xor, 0); RUN(k_add); RUN(k_fma_f32); RUN(k_bitop3); RUN(k_alignbit_c);
    return 0;
}
