// tools/power_ubench.hip — measurement aid: socket power and delivered shader clock while one instruction class runs at
// 8 waves per SIMD for ~3 s (run under tools/clock_watch.sh; phases are matched to the rocm-smi samples by wall time).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(256) k_pow(uint32_t* out, int iters) {
    __shared__ uint32_t lds[256 * 5];
    uint32_t r[16];
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { r[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x; f[i] = (float)(r[i] & 0xFFFF) - 30000.0f; }
    lds[threadIdx.x] = r[0];
    const uint32_t a4 = threadIdx.x * 4, a16 = threadIdx.x * 16;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 5) & 15]));
                if (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[i]) : "v"(r[(i + 5) & 15]), "v"(r[(i + 9) & 15]));
                if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r[i]) : "v"(r[(i + 5) & 15]));
                if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 5) & 15]), "v"(f[(i + 9) & 15]));
                if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 5) & 15]));
                if (OP == 5) { asm volatile("ds_write_b32 %0, %1" ::"v"(a4), "v"(r[i]) : "memory"); asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r[i]) : "v"(a4) : "memory"); }
                if (OP == 6) { uint4 v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a16) : "memory"); r[i] ^= v.x; }
            }
        }
        if (OP == 3 || OP == 4) {   // keep the fp values finite and busy
#pragma unroll
            for (int i = 0; i < 16; i++) f[i] = f[i] * 1e-3f + 1234.5f;
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= r[i] ^ __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc ^ lds[(threadIdx.x + 1) & 255];
}

template <int OP>
void run(uint32_t* out, const char* name) {
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const auto t0 = std::chrono::system_clock::now();
    double ms_total = 0;
    int launches = 0;
    while (std::chrono::duration<double>(std::chrono::system_clock::now() - t0).count() < 3.0) {
        (void)hipEventRecord(a);
        k_pow<OP><<<blocks, 256>>>(out, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        ms_total += ms;
        launches++;
    }
    const auto t1 = std::chrono::system_clock::now();
    const double per_simd = (double)iters * 64 * blocks * 4 / 1024.0 * launches;
    printf("PHASE %-22s start %.3f end %.3f  %.3f ns per wave-op per SIMD\n", name,
           std::chrono::duration<double>(t0.time_since_epoch()).count(), std::chrono::duration<double>(t1.time_since_epoch()).count(),
           ms_total * 1e6 / per_simd);
    fflush(stdout);
    (void)hipDeviceSynchronize();
    struct timespec ts = {1, 0};
    nanosleep(&ts, nullptr);
}

int main() {
    uint32_t* out;
    (void)hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
    run<0>(out, "v_xor_b32"); run<1>(out, "v_bitop3_b32"); run<2>(out, "v_alignbit_b32"); run<3>(out, "v_fma_f32"); run<4>(out, "v_add_f32");
    run<5>(out, "ds_write+read_b32"); run<6>(out, "ds_read_b128");
    return 0;
}
