// tools/small_stamps.hip -- measurement aid: where the time of k_encaps_small<3> (one item, lone workgroup) goes.  A copy of the
// kernel's sequence (mlkem_small.hpp) with s_memtime stamps of wave 0 between its steps, and of the other waves when they run out of
// jobs; the stamps go to a buffer of their own.  Cycle counts are shader cycles; the wall clock of the same spans comes from
// s_memrealtime (100 MHz).  Both workgroup shapes: eight waves (latency form) and four (dense form).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude -o tools/small_stamps.bin tools/small_stamps.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../crystals-kyber_amd/csrc/mlkem_pipeline.hpp"
using namespace mlkem;

__device__ __forceinline__ void stamp(unsigned long long* s, int k) {
    if (lane_id() == 0) { s[2 * k] = __builtin_amdgcn_s_memtime(); s[2 * k + 1] = __builtin_amdgcn_s_memrealtime(); }
}

template <int K, int ETA1, int DU, int DV, int NW>
__global__ void __launch_bounds__(WAVE * NW)
k_encaps_stamped(const uint8_t* __restrict__ ek, const uint8_t* __restrict__ m, uint8_t* __restrict__ c, uint8_t* __restrict__ Kout,
                 unsigned long long* stamps) {
    __shared__ K2Lds<K + 1> xl;
    __shared__ SmallHand<K, ETA1> hand;
    __shared__ uint32_t sq[NW][XOF_LDS_WORDS];
    __shared__ uint2 rc_tables[NW][WK_RC_ENTRIES];
    __shared__ SmallSync sy;
    constexpr unsigned EK = 384 * K + 32;
    const int wv = wave_id();
    unsigned long long* st = stamps + 32 * wv;
    stamp(st, 0);
    WkLane cst;
    wk_lane_init(cst, rc_tables[wave_id()]);
    if (threadIdx.x == 0) { sy.next_job = 0; sy.jobs_done = 0; sy.r_ready = 0; sy.kbar_ready = 0; }
    block_barrier();
    stamp(st, 1);
    if (wv == 0) {
        const int i = wk_index();
        WkState a;
        wk_H(a, cst, ek, EK);
        stamp(st, 2);
        uint2 mv;
        mv.x = 0; mv.y = 0;
        if (i >= 0 && i < 4) mv = reinterpret_cast<const uint2*>(m)[i];
        wk_G_of_x_and_digest(a, cst, mv);
        uint2 o;
        o.x = a.lo; o.y = a.hi;
        if (wk_primary() && i < 4) reinterpret_cast<uint2*>(Kout)[i] = o;
        else if (wk_primary() && i < 8) reinterpret_cast<uint2*>(hand.r)[i - 4] = o;
        flag_signal(&sy.r_ready);
        stamp(st, 3);
    }
    small_jobs<K, ETA1>(sy, cst, ek + 384 * K, true, hand.A, hand.r, hand.prf, 2 * K + 1, 168u, sq[wv]);
    stamp(st, 4);                       // this wave found no job left
    if (wv != 0) return;
    flag_wait(&sy.jobs_done, (uint32_t)(K * K + 2 * K + 1));
    stamp(st, 5);
    encrypt1_body<K, ETA1, DU, DV, false>(xl.xch, ek, m, hand.A, hand.prf, c, nullptr, nullptr, nullptr, nullptr, nullptr);
    stamp(st, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(st, 7);
}

template <int NW> int run() {
    constexpr int K = 3;
    std::vector<uint8_t> ek(1184), m(32), c(1088), Kk(32);
    for (size_t i = 0; i < ek.size(); i++) ek[i] = (uint8_t)(i * 7 + 1);
    for (int b = 0; b < K; b++)
        for (int i = 0; i < 384; i += 3) { ek[384 * b + i + 1] &= 0x0F; ek[384 * b + i + 2] &= 0x7F; }   // keep the 12-bit fields below q-ish
    uint8_t *dek, *dm, *dc, *dK;
    unsigned long long* ds;
    (void)hipMalloc(&dek, 1184); (void)hipMalloc(&dm, 32); (void)hipMalloc(&dc, 1088); (void)hipMalloc(&dK, 32); (void)hipMalloc(&ds, 8 * 32 * 8);
    (void)hipMemcpy(dek, ek.data(), 1184, hipMemcpyHostToDevice); (void)hipMemcpy(dm, m.data(), 32, hipMemcpyHostToDevice);
    unsigned long long h[8 * 32] = {0};
    double acc[10] = {0}, accr[10] = {0}, xof_done = 0, xof_first = 0;
    const int R = 200;
    for (int r = 0; r < R + 20; r++) {
        k_encaps_stamped<3, 2, 10, 4, NW><<<1, WAVE * NW>>>(dek, dm, dc, dK, ds);
        (void)hipDeviceSynchronize();
        if (r < 20) continue;
        (void)hipMemcpy(h, ds, sizeof h, hipMemcpyDeviceToHost);
        for (int k = 1; k <= 7; k++) { acc[k] += (double)(h[2 * k] - h[2 * (k - 1)]); accr[k] += (double)(h[2 * k + 1] - h[2 * (k - 1) + 1]); }
        unsigned long long last = 0, first = ~0ull;
        for (int w = 1; w < NW; w++) {                  // realtime stamp 4 of the other waves: no job left for them
            last = h[32 * w + 2 * 4 + 1] > last ? h[32 * w + 2 * 4 + 1] : last;
            first = h[32 * w + 2 * 4 + 1] < first ? h[32 * w + 2 * 4 + 1] : first;
        }
        xof_done += (double)(last - h[1]);
        xof_first += (double)(first - h[1]);
    }
    const char* name[8] = {"", "lane constants, iota table, counters, first barrier", "H(ek): 9 blocks absorbed (9 permutations)", "G(m || h) + stores + r_ready",
                           "jobs wave 0 still found (SampleNTT / PRF rows)", "wait for the last job (jobs_done)", "K-PKE.Encrypt (encrypt1_body)",
                           "outstanding stores drained"};
    printf("k_encaps_small<3, ..., %d waves> stage by stage, wave 0 of a lone workgroup, mean of %d launches (shader cycles ; us by the 100 MHz clock)\n", NW, R);
    double tc = 0, tr = 0;
    for (int k = 1; k <= 7; k++) { printf("  %-52s %9.0f cycles  %6.2f us\n", name[k], acc[k] / R, accr[k] / R / 100.0); tc += acc[k]; tr += accr[k]; }
    printf("  %-52s %9.0f cycles  %6.2f us   (clock %.0f MHz)\n", "total", tc / R, tr / R / 100.0, tc / tr * 100.0);
    printf("  the other waves run out of jobs (SampleNTT, then PRF rows behind r_ready) %.2f .. %.2f us after the start\n", xof_first / R / 100.0, xof_done / R / 100.0);
    return 0;
}
int main() { run<SMALL_WAVES>(); run<SMALL_WAVES_DENSE>(); return 0; }
