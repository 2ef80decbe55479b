// tools/keccak_wave_ubench.hip -- measurement aid: latency of ONE Keccak-f[1600] for a lone wave in three forms
//   lane   : lane-sliced (one sponge per SIMD lane, mlkem_device.hpp: keccak_f1600)
//   half   : one sponge per half-wave, 18 ds_bpermute per round in three dependent groups (round-3 form, kept here as `hw_*`)
//   wave   : one sponge per wave, theta by DPP + v_permlane swaps, ONE dependent group of 6 ds_bpermute per round, iota's
//            constant from an LDS table (mlkem_wkeccak.hpp: wk_permute; the first form of the round -- scalar loads + selects for
//            iota, three instructions per word for D -- ran 3.43 us)
// and a correctness check of the wave forms against the lane-sliced permutation on random states.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../crystals-kyber_amd/csrc/mlkem_wkeccak.hpp"
using namespace mlkem;

// ---- round-3 half-wave form (two sponges per wave) ----
struct HwLane { uint32_t col[4], xm, xp, src[3], sh; bool swp, lane0; };
__device__ __forceinline__ uint32_t hw_fetch(uint32_t byte_addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_addr, (int)v); }
__device__ __forceinline__ void hw_lane_init(HwLane& c) {
    const unsigned hl = (unsigned)lane_id() & 31u, half = (unsigned)lane_id() >> 5;
    const int L = hl < 25 ? (int)hl : 0;
    const uint32_t base = 128u * half;
    const int x = L % 5, y = L / 5;
    for (int k = 0; k < 4; k++) c.col[k] = base + 4u * (uint32_t)(x + 5 * ((y + k + 1) % 5));
    c.xm = base + 4u * (uint32_t)((x + 4) % 5 + 5 * y);
    c.xp = base + 4u * (uint32_t)((x + 1) % 5 + 5 * y);
    for (int k = 0; k < 3; k++) { const int xd = (x + k) % 5; c.src[k] = base + 4u * (uint32_t)((xd + 3 * y) % 5 + 5 * xd); }
    const unsigned r = WK_RHO[L];
    c.swp = r >= 32 || r == 0;
    c.sh = r == 0 ? 0u : (32u - (r & 31u)) & 31u;
    c.lane0 = hl == 0;
}
__device__ __forceinline__ void hw_permute(uint32_t& lo, uint32_t& hi, const HwLane& c) {
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        uint32_t tl[4], th[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { tl[k] = hw_fetch(c.col[k], lo); th[k] = hw_fetch(c.col[k], hi); }
        const uint32_t cl = MLKEM_XOR3(MLKEM_XOR3(lo, tl[0], tl[1]), tl[2], tl[3]);
        const uint32_t ch = MLKEM_XOR3(MLKEM_XOR3(hi, th[0], th[1]), th[2], th[3]);
        const uint32_t ml = hw_fetch(c.xm, cl), mh = hw_fetch(c.xm, ch), pl = hw_fetch(c.xp, cl), ph = hw_fetch(c.xp, ch);
        uint32_t rl, rh;
        rotl64<1>(pl, ph, rl, rh);
        const uint32_t al = MLKEM_XOR3(lo, ml, rl), ah = MLKEM_XOR3(hi, mh, rh);
        const uint32_t a0 = c.swp ? ah : al, a1 = c.swp ? al : ah;
        const uint32_t bl = __builtin_amdgcn_alignbit(a0, a1, c.sh), bh = __builtin_amdgcn_alignbit(a1, a0, c.sh);
        const uint32_t b0l = hw_fetch(c.src[0], bl), b1l = hw_fetch(c.src[1], bl), b2l = hw_fetch(c.src[2], bl);
        const uint32_t b0h = hw_fetch(c.src[0], bh), b1h = hw_fetch(c.src[1], bh), b2h = hw_fetch(c.src[2], bh);
        lo = MLKEM_CHI(b0l, b1l, b2l);
        hi = MLKEM_CHI(b0h, b1h, b2h);
        lo ^= c.lane0 ? KECCAK_RC[2 * round] : 0u;
        hi ^= c.lane0 ? KECCAK_RC[2 * round + 1] : 0u;
    }
}

__global__ void __launch_bounds__(64, 6) k_lane(uint32_t* out, int perms) {
    KeccakState s;
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = threadIdx.x * 2654435761u + i; s.hi[i] = blockIdx.x * 40503u + i; }
    for (int p = 0; p < perms; p++) keccak_f1600(s);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 25; i++) acc ^= s.lo[i] ^ s.hi[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(64) k_half(uint32_t* out, int perms) {
    HwLane c;
    hw_lane_init(c);
    uint32_t lo = threadIdx.x * 2654435761u, hi = blockIdx.x * 40503u + threadIdx.x;
    for (int p = 0; p < perms; p++) hw_permute(lo, hi, c);
    out[blockIdx.x * 64 + threadIdx.x] = lo ^ hi;
}
__global__ void __launch_bounds__(64) k_wave(uint32_t* out, int perms) {
    __shared__ uint2 rc_table[WK_RC_ENTRIES];
    WkLane c;
    wk_lane_init(c, rc_table);
    WkState a;
    a.lo = threadIdx.x * 2654435761u; a.hi = blockIdx.x * 40503u + threadIdx.x;
    wk_canon(a, c);
    for (int p = 0; p < perms; p++) wk_permute(a, c);
    out[blockIdx.x * 64 + threadIdx.x] = a.lo ^ a.hi;
}
// correctness: state words in[50] (lo/hi of Keccak lane i at in[2i], in[2i+1]) -> out[50] after `perms` permutations
__global__ void __launch_bounds__(64) k_wave_check(const uint32_t* in, uint32_t* out, int perms) {
    __shared__ uint2 rc_table[WK_RC_ENTRIES];
    WkLane c;
    wk_lane_init(c, rc_table);
    WkState a;
    const int i = wk_index();   // Keccak lane x + 5 y this SIMD lane holds, or -1
    a.lo = i >= 0 ? in[2 * i] : 0u;
    a.hi = i >= 0 ? in[2 * i + 1] : 0u;
    for (int p = 0; p < perms; p++) wk_permute(a, c);
    if (wk_primary() && i >= 0) { out[2 * i] = a.lo; out[2 * i + 1] = a.hi; }
}
__global__ void __launch_bounds__(64) k_lane_check(const uint32_t* in, uint32_t* out, int perms) {
    KeccakState s;
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = in[2 * i]; s.hi[i] = in[2 * i + 1]; }
    for (int p = 0; p < perms; p++) keccak_f1600(s);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 25; i++) { out[2 * i] = s.lo[i]; out[2 * i + 1] = s.hi[i]; }
    }
}


template <class F>
static float time_ms(F launch) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(2);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int t = 0; t < 5; t++) {
        (void)hipEventRecord(a);
        launch(64);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t *out, *din, *d1, *d2;
    (void)hipMalloc(&out, 16384 * 64 * 4);
    (void)hipMalloc(&din, 200); (void)hipMalloc(&d1, 200); (void)hipMalloc(&d2, 200);
    // correctness
    int bad = 0;
    srand(7);
    for (int trial = 0; trial < 20; trial++) {
        uint32_t h[50], r1[50], r2[50];
        for (int i = 0; i < 50; i++) h[i] = trial == 0 ? 0u : ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        (void)hipMemcpy(din, h, 200, hipMemcpyHostToDevice);
        const int perms = 1 + trial % 3;
        k_lane_check<<<1, 64>>>(din, d1, perms);
        k_wave_check<<<1, 64>>>(din, d2, perms);
        (void)hipMemcpy(r1, d1, 200, hipMemcpyDeviceToHost);
        (void)hipMemcpy(r2, d2, 200, hipMemcpyDeviceToHost);
        for (int i = 0; i < 50; i++) bad += r1[i] != r2[i];
    }
    printf("wave form vs lane-sliced permutation on 20 states: %d mismatching words\n", bad);
    for (int blocks : {1, 256, 1024, 4096, 16384}) {
        const float tl = time_ms([&](int p) { k_lane<<<blocks, 64>>>(out, p); });
        const float th = time_ms([&](int p) { k_half<<<blocks, 64>>>(out, p); });
        const float tw = time_ms([&](int p) { k_wave<<<blocks, 64>>>(out, p); });
        printf("waves %5d (%.2f per SIMD): us per permutation (per wave)  lane-sliced %.2f (64 sponges)  half-wave %.2f (2 sponges)  wave %.2f (1 sponge)\n",
               blocks, blocks / 1024.0, tl * 1e3 / 64, th * 1e3 / 64, tw * 1e3 / 64);
    }
    return bad != 0;
}
