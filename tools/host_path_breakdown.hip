// tools/host_path_breakdown.hip -- measurement aid: where the host-pointer n = 1 call spends its time.
//   (a) an empty kernel: launch + hipStreamSynchronize           (the floor of any launch-based call)
//   (b) mlkem_encaps_dev / mlkem_decaps_dev on device buffers + synchronise
//   (c) mlkem_encaps / mlkem_decaps on host buffers (the shim's path: zero-copy small call)
// build: hipcc --offload-arch=gfx950 -O2 -Iinclude -o tools/host_path_breakdown.bin tools/host_path_breakdown.hip -Lcrystals-kyber_amd -lmlkem_amd -Wl,-rpath,'$ORIGIN/../crystals-kyber_amd'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "mlkem_batch.h"

__global__ void k_empty() {}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int R = 2000;
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int i = 0; i < 100; i++) { k_empty<<<1, 64, 0, st>>>(); (void)hipStreamSynchronize(st); }
    double t0 = now_us();
    for (int i = 0; i < R; i++) { k_empty<<<1, 64, 0, st>>>(); (void)hipStreamSynchronize(st); }
    printf("(a) empty kernel launch + hipStreamSynchronize:            %6.1f us per call\n", (now_us() - t0) / R);
    unsigned ekl, dkl, cl;
    mlkem_sizes(768, &ekl, &dkl, &cl);
    std::vector<uint8_t> d(32, 1), z(32, 2), m(32, 3), ek(ekl), dk(dkl), c(cl), K(32), K2(32);
    int32_t status = 1;
    if (mlkem_keygen(768, 1, d.data(), z.data(), ek.data(), dk.data())) return 1;
    uint8_t *dek, *ddk, *dm, *dc, *dK;
    int32_t* dst;
    (void)hipMalloc(&dek, ekl); (void)hipMalloc(&ddk, dkl); (void)hipMalloc(&dm, 32); (void)hipMalloc(&dc, cl); (void)hipMalloc(&dK, 32); (void)hipMalloc(&dst, 4);
    (void)hipMemcpy(dek, ek.data(), ekl, hipMemcpyHostToDevice); (void)hipMemcpy(ddk, dk.data(), dkl, hipMemcpyHostToDevice); (void)hipMemcpy(dm, m.data(), 32, hipMemcpyHostToDevice);
    mlkem_ctx* ctx;
    if (mlkem_ctx_create(&ctx, 0, 1024)) return 1;
    for (int i = 0; i < 50; i++) { mlkem_encaps_dev(ctx, 768, 1, dek, dm, dc, dK, st); mlkem_decaps_dev(ctx, 768, 1, ddk, dc, dK, dst, st); (void)hipStreamSynchronize(st); }
    t0 = now_us();
    for (int i = 0; i < R; i++) { mlkem_encaps_dev(ctx, 768, 1, dek, dm, dc, dK, st); (void)hipStreamSynchronize(st); }
    const double e_dev = (now_us() - t0) / R;
    t0 = now_us();
    for (int i = 0; i < R; i++) { mlkem_decaps_dev(ctx, 768, 1, ddk, dc, dK, dst, st); (void)hipStreamSynchronize(st); }
    const double d_dev = (now_us() - t0) / R;
    printf("(b) device buffers + synchronise:        encaps %6.1f us   decaps %6.1f us\n", e_dev, d_dev);
    for (int i = 0; i < 50; i++) { mlkem_encaps(768, 1, ek.data(), m.data(), c.data(), K.data()); mlkem_decaps(768, 1, dk.data(), c.data(), K2.data(), &status); }
    t0 = now_us();
    for (int i = 0; i < R; i++) mlkem_encaps(768, 1, ek.data(), m.data(), c.data(), K.data());
    const double e_host = (now_us() - t0) / R;
    t0 = now_us();
    for (int i = 0; i < R; i++) mlkem_decaps(768, 1, dk.data(), c.data(), K2.data(), &status);
    const double d_host = (now_us() - t0) / R;
    printf("(c) host buffers (the shim's path):      encaps %6.1f us   decaps %6.1f us   K match %d status %d\n", e_host, d_host, !memcmp(K.data(), K2.data(), 32), status);
    mlkem_ctx_destroy(ctx);
    return memcmp(K.data(), K2.data(), 32) != 0;
}
