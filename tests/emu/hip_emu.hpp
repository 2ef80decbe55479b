// tests/emu/hip_emu.hpp — TEST INFRASTRUCTURE ONLY.
//
// A minimal wave64 emulator: compiles the product's kernel bodies (crystals-kyber_amd/csrc/mlkem_kernels.hpp)
// with g++ and runs every wavefront as 64 host threads that meet at a barrier wherever the device code
// orders wave-private LDS traffic (wave_lds_fence) or exchanges data across lanes (__shfl, __ballot).
// It exists so kernel logic (indexing, layouts, codecs, sponge padding, rejection sampling) can be checked
// against the oracle in the CPU test tier; it is never part of, nor a fallback for, libmlkem_amd.so.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <thread>
#include <vector>

#define MLKEM_EMU 1
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __constant__
#define __shared__ static
#define __launch_bounds__(...)

struct emu_dim3 { unsigned x = 0, y = 0, z = 0; };
static thread_local emu_dim3 threadIdx, blockIdx, gridDim, blockDim;

struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };

namespace emu {
struct WaveCtx {
    pthread_barrier_t bar;
    pthread_barrier_t qbar[16];   // one per quad of four lanes: DPP quad_perm exchanges, which the device code also issues inside
    uint64_t qslot[2][64];        // lane-divergent branches (only the lanes of the quad have to arrive)
    uint64_t slot[2][64];   // two exchange buffers used alternately: ONE barrier per cross-lane operation (a lane can run at
                            // most one operation ahead of the slowest lane, and then writes the other buffer)
};
static thread_local unsigned qphase = 0;
static thread_local unsigned xphase = 0;   // every lane of a wave executes the same sequence of cross-lane operations
static thread_local WaveCtx* wave = nullptr;
static thread_local pthread_barrier_t* block_bar = nullptr;
inline void wave_barrier() { pthread_barrier_wait(&wave->bar); }
inline void block_barrier() { pthread_barrier_wait(block_bar); }   // __syncthreads(): every thread of the workgroup must arrive

template <class F>
inline void launch(unsigned grid, unsigned block, F body) {
    const unsigned nwaves = (block + 63) / 64;
    for (unsigned b = 0; b < grid; b++) {
        std::vector<WaveCtx> ctx(nwaves);
        for (unsigned w = 0; w < nwaves; w++) {
            unsigned lanes = block - 64 * w < 64 ? block - 64 * w : 64;
            pthread_barrier_init(&ctx[w].bar, nullptr, lanes);
            for (int q = 0; q < 16; q++) pthread_barrier_init(&ctx[w].qbar[q], nullptr, 4);
        }
        pthread_barrier_t bbar;
        pthread_barrier_init(&bbar, nullptr, block);
        std::vector<std::thread> th;
        th.reserve(block);
        for (unsigned t = 0; t < block; t++)
            th.emplace_back([&, t, b] {
                threadIdx.x = t; blockIdx.x = b; gridDim.x = grid; blockDim.x = block;
                wave = &ctx[t / 64];
                block_bar = &bbar;
                xphase = 0;
                qphase = 0;
                body();
            });
        for (auto& x : th) x.join();
        pthread_barrier_destroy(&bbar);
        for (unsigned w = 0; w < nwaves; w++) {
            pthread_barrier_destroy(&ctx[w].bar);
            for (int q = 0; q < 16; q++) pthread_barrier_destroy(&ctx[w].qbar[q]);
        }
    }
}
}   // namespace emu

// ---- builtins used by the device code ---------------------------------------------------------------
inline uint32_t __builtin_amdgcn_bitop3_b32(uint32_t a, uint32_t b, uint32_t c, unsigned lut) {
    uint32_t r = 0;
    for (int i = 0; i < 8; i++)
        if (lut & (1u << i)) {
            uint32_t ta = (i & 4) ? a : ~a, tb = (i & 2) ? b : ~b, tc = (i & 1) ? c : ~c;
            r |= ta & tb & tc;
        }
    return r;
}
inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31));
}
#define __builtin_amdgcn_fence(order, scope) emu::wave_barrier()
inline int __mul24(int a, int b) {
    int sa = (a << 8) >> 8, sb = (b << 8) >> 8;   // low 24 bits, sign-extended
    return (int)((int64_t)sa * sb);
}
inline unsigned __umul24(unsigned a, unsigned b) { return (unsigned)((uint64_t)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)); }
inline unsigned long long __ballot(int pred) {
    const int l = threadIdx.x & 63;
    uint64_t* slot = emu::wave->slot[emu::xphase++ & 1];
    slot[l] = pred ? 1 : 0;
    emu::wave_barrier();
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m |= (unsigned long long)(slot[i] & 1) << i;
    return m;
}
inline int __shfl(int v, int src) {
    const int l = threadIdx.x & 63;
    uint64_t* slot = emu::wave->slot[emu::xphase++ & 1];
    slot[l] = (uint64_t)(uint32_t)v;
    emu::wave_barrier();
    return (int)(uint32_t)slot[src & 63];
}
// value of lane `src` of the caller's own quad (DPP quad_perm): synchronises the four lanes of the quad only
inline int emu_quad_shfl(int v, int src) {
    const int l = threadIdx.x & 63;
    uint64_t* slot = emu::wave->qslot[emu::qphase++ & 1];
    slot[l] = (uint64_t)(uint32_t)v;
    pthread_barrier_wait(&emu::wave->qbar[l >> 2]);
    return (int)(uint32_t)slot[(l & ~3) | (src & 3)];
}
inline int __shfl_xor(int v, int mask) { return __shfl(v, (int)((threadIdx.x & 63) ^ (unsigned)mask)); }
inline uint32_t atomicOr(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
