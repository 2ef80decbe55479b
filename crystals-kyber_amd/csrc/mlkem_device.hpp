// mlkem_device.hpp — gfx950 device building blocks for the batched ML-KEM engine.
//
// Everything here is wave64-native:
//   * Keccak-f[1600] is lane-sliced (one sponge per lane, 25 lanes x 2 x u32 VGPRs) and built from
//     v_bitop3_b32 (3-input LUT: xor3 for theta, a^(~b&c) for chi) and v_alignbit_b32 (64-bit rotates).
//   * NTT / InverseNTT run two items per wavefront through LDS inside K-PKE.KeyGen / Encrypt (mlkem_kpke2.hpp) and four
//     polynomials per wavefront in registers with DPP butterflies elsewhere (mlkem_rntt.hpp), always on float2 pairs; the
//     mod-3329 arithmetic is exact (integers < 2^24 on the fp32 pipe, Barrett reduction by reciprocal multiply, mlkem_fntt.hpp),
//     so results are canonical in [0, q) and bit-identical to the reference's `% Q` arithmetic (ml_kem.c:287-442).
//
// Reference behaviour that is reproduced on purpose: PRF and J are SHAKE128
// (ml_kem.c:508, :546), ByteDecode_12 does not reduce mod q (ml_kem.c:170).
#pragma once
#ifndef MLKEM_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace mlkem {

constexpr int KQ = 3329;
constexpr int INV128 = 3303;    // 128^-1 mod q (ml_kem.c:378-381)

// ----------------------------------------------------------------------------------------------
// compile-time helpers for the twiddle tables: zeta_i = 17^BitRev7(i) (ml_kem.c:300-307)
// ----------------------------------------------------------------------------------------------
constexpr int cx_bitrev7(int r) {
    int o = 0;
    for (int i = 0; i < 7; i++) o |= ((r >> i) & 1) << (6 - i);
    return o;
}
constexpr int cx_pow17(int e) {
    int r = 1;
    for (int i = 0; i < e; i++) r = (r * 17) % KQ;
    return r;
}
constexpr int cx_centered(int x) { return x > KQ / 2 ? x - KQ : x; }
// ----------------------------------------------------------------------------------------------
// integer helpers for Compress / Decompress
// ----------------------------------------------------------------------------------------------
// floor(num / q) for 0 <= num < 2^23 (exact: see DESIGN.md "division by q")
__device__ __forceinline__ unsigned div_q(unsigned num) {
    return (unsigned)(((uint64_t)num * 10321340ull) >> 35);
}
// Compress_d (ml_kem.c:83-97): round(2^d x / q) mod 2^d, ties cannot occur (q odd). x in [0, q).
template <int D>
__device__ __forceinline__ unsigned compress_d(unsigned x) {
    static_assert(D >= 1 && D <= 11, "d");
    return div_q((x << D) + (KQ / 2)) & ((1u << D) - 1);
}
// Decompress_d (ml_kem.c:104-119): floor(q y / 2^d) + (remainder >= 2^(d-1))
template <int D>
__device__ __forceinline__ unsigned decompress_d(unsigned y) {
    static_assert(D >= 1 && D <= 11, "d");
    return (KQ * y + (1u << (D - 1))) >> D;
}

// ----------------------------------------------------------------------------------------------
// Keccak-f[1600], lane-sliced: state = 25 lanes x (lo, hi) u32.   sha3.c:15-216
// ----------------------------------------------------------------------------------------------
__constant__ uint32_t KECCAK_RC[48] = {   // (lo, hi) pairs of the 24 round constants (sha3.c:148-201)
    0x00000001u, 0x00000000u, 0x00008082u, 0x00000000u, 0x0000808au, 0x80000000u, 0x80008000u, 0x80000000u,
    0x0000808bu, 0x00000000u, 0x80000001u, 0x00000000u, 0x80008081u, 0x80000000u, 0x00008009u, 0x80000000u,
    0x0000008au, 0x00000000u, 0x00000088u, 0x00000000u, 0x80008009u, 0x00000000u, 0x8000000au, 0x00000000u,
    0x8000808bu, 0x00000000u, 0x0000008bu, 0x80000000u, 0x00008089u, 0x80000000u, 0x00008003u, 0x80000000u,
    0x00008002u, 0x80000000u, 0x00000080u, 0x80000000u, 0x0000800au, 0x00000000u, 0x8000000au, 0x80000000u,
    0x80008081u, 0x80000000u, 0x00008080u, 0x80000000u, 0x80000001u, 0x00000000u, 0x80008008u, 0x80000000u};

#define MLKEM_XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#define MLKEM_CHI(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xD2)   // a ^ (~b & c)

struct KeccakState {
    uint32_t lo[25], hi[25];
};

// 64-bit rotate-left by compile-time N of (lo, hi) via v_alignbit_b32
template <int N>
__device__ __forceinline__ void rotl64(uint32_t lo, uint32_t hi, uint32_t& olo, uint32_t& ohi) {
    if constexpr (N == 0) {
        olo = lo; ohi = hi;
    } else if constexpr (N < 32) {
        olo = __builtin_amdgcn_alignbit(lo, hi, 32 - N);
        ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - N);
    } else if constexpr (N == 32) {
        olo = hi; ohi = lo;
    } else {
        olo = __builtin_amdgcn_alignbit(hi, lo, 64 - N);
        ohi = __builtin_amdgcn_alignbit(lo, hi, 64 - N);
    }
}

// theta-apply + rho + pi of one lane: A ^ C[x-1] ^ rotl(C[x+1], 1) as one v_bitop3 per half-lane (C and rotl(C), 20
// registers, stay live through rho / pi: 180 VALU per round, 78 VGPRs for the bare permutation; materialising D first needs
// 10 more instructions per round and saved no registers in practice)
#define MLKEM_RHOPI(dst, src, rot, dx)                                                       \
    {                                                                                        \
        uint32_t tl = MLKEM_XOR3(s.lo[src], cl[(dx + 4) % 5], rl[(dx + 1) % 5]);             \
        uint32_t th = MLKEM_XOR3(s.hi[src], ch[(dx + 4) % 5], rh[(dx + 1) % 5]);             \
        rotl64<rot>(tl, th, bl[dst], bh[dst]);                                               \
    }

__device__ __forceinline__ void keccak_f1600(KeccakState& s) {
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        uint32_t cl[5], ch[5], rl[5], rh[5], bl[25], bh[25];
        // theta (sha3.c:15-49): column parities, D[x] = C[x-1] ^ rotl(C[x+1], 1) folded into the xor3 below
#pragma unroll
        for (int x = 0; x < 5; x++) {
            cl[x] = MLKEM_XOR3(MLKEM_XOR3(s.lo[x], s.lo[x + 5], s.lo[x + 10]), s.lo[x + 15], s.lo[x + 20]);
            ch[x] = MLKEM_XOR3(MLKEM_XOR3(s.hi[x], s.hi[x + 5], s.hi[x + 10]), s.hi[x + 15], s.hi[x + 20]);
        }
#pragma unroll
        for (int x = 0; x < 5; x++) rotl64<1>(cl[x], ch[x], rl[x], rh[x]);
        // theta-apply + rho (sha3.c:53-84) + pi (sha3.c:88-112): B[y, 2x+3y] = rotl(A[x, y] ^ D[x], r[x, y])
        MLKEM_RHOPI(0, 0, 0, 0)   MLKEM_RHOPI(10, 1, 1, 1)  MLKEM_RHOPI(20, 2, 62, 2) MLKEM_RHOPI(5, 3, 28, 3)  MLKEM_RHOPI(15, 4, 27, 4)
        MLKEM_RHOPI(16, 5, 36, 0) MLKEM_RHOPI(1, 6, 44, 1)  MLKEM_RHOPI(11, 7, 6, 2)  MLKEM_RHOPI(21, 8, 55, 3) MLKEM_RHOPI(6, 9, 20, 4)
        MLKEM_RHOPI(7, 10, 3, 0)  MLKEM_RHOPI(17, 11, 10, 1) MLKEM_RHOPI(2, 12, 43, 2) MLKEM_RHOPI(12, 13, 25, 3) MLKEM_RHOPI(22, 14, 39, 4)
        MLKEM_RHOPI(23, 15, 41, 0) MLKEM_RHOPI(8, 16, 45, 1) MLKEM_RHOPI(18, 17, 15, 2) MLKEM_RHOPI(3, 18, 21, 3) MLKEM_RHOPI(13, 19, 8, 4)
        MLKEM_RHOPI(14, 20, 18, 0) MLKEM_RHOPI(24, 21, 2, 1) MLKEM_RHOPI(9, 22, 61, 2) MLKEM_RHOPI(19, 23, 56, 3) MLKEM_RHOPI(4, 24, 14, 4)
        // chi (sha3.c:116-140)
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
#pragma unroll
            for (int x = 0; x < 5; x++) {
                s.lo[y + x] = MLKEM_CHI(bl[y + x], bl[y + (x + 1) % 5], bl[y + (x + 2) % 5]);
                s.hi[y + x] = MLKEM_CHI(bh[y + x], bh[y + (x + 1) % 5], bh[y + (x + 2) % 5]);
            }
        }
        // iota (sha3.c:182-201)
        s.lo[0] ^= KECCAK_RC[2 * round];
        s.hi[0] ^= KECCAK_RC[2 * round + 1];
    }
}

__device__ __forceinline__ void keccak_zero(KeccakState& s) {
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = 0; s.hi[i] = 0; }
}

// xor one byte into the state at compile-time byte position POS (pad / domain bytes)
template <int POS>
__device__ __forceinline__ void keccak_xor_byte(KeccakState& s, uint32_t byte) {
    constexpr int w = POS / 4, sh = 8 * (POS % 4);
    if constexpr (w % 2 == 0) s.lo[w / 2] ^= byte << sh;
    else s.hi[w / 2] ^= byte << sh;
}
// state viewed as 50 dwords: dword w = (w even ? lo : hi)[w/2]
template <int W>
__device__ __forceinline__ uint32_t& keccak_word(KeccakState& s) {
    if constexpr (W % 2 == 0) return s.lo[W / 2];
    else return s.hi[W / 2];
}

// ----------------------------------------------------------------------------------------------
// wave-level helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// Single-use streams of the K-PKE kernels: the sampled matrix A-hat (written once by the sampler, read once, 1.2-2.5 GB per
// chunk: far beyond the 4 MB L2 of an XCD) and the packed key / ciphertext rows.  MLKEM_A_NT is a bit set: 1 = A-hat loads,
// 4 = packed-row loads carry the non-temporal hint (A/B on one box, tools/ab_run.py: -0.4..-0.7 % step time, at the edge of
// the noise); 2 = the sampler's A-hat stores as well: those are 32-byte pieces of a lane's own row, and as non-temporal
// stores they are no longer merged in L2 - k_sample_main 1.12 -> 3.73 ms.  Default 5.
#ifndef MLKEM_A_NT
#define MLKEM_A_NT 5
#endif
__device__ __forceinline__ void stream_store16(void* p, const uint4 v) {
#if (MLKEM_A_NT & 2) && !defined(MLKEM_EMU)
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    v4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<v4*>(p));
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}
__device__ __forceinline__ uint32_t stream_load4(const uint32_t* p) {
#if (MLKEM_A_NT & 4) && !defined(MLKEM_EMU)
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint2 stream_load8(const void* p) {
#if (MLKEM_A_NT & 1) && !defined(MLKEM_EMU)
    typedef unsigned v2 __attribute__((ext_vector_type(2)));
    const v2 t = __builtin_nontemporal_load(reinterpret_cast<const v2*>(p));
    uint2 r;
    r.x = t.x; r.y = t.y;
    return r;
#else
    return *reinterpret_cast<const uint2*>(p);
#endif
}
// index of the wave inside its workgroup as a wave-uniform (scalar) value: everything derived from it -- the item index,
// the item's base pointers, the wave's LDS block -- then lives in SGPRs and its arithmetic runs on the scalar unit
__device__ __forceinline__ int wave_id() {
#ifdef MLKEM_EMU
    return (int)(threadIdx.x >> 6);
#else
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
}
// lane id the compiler cannot merge with an earlier lane_id(): values derived from it are recomputed where they are
// used instead of being kept in registers across a permutation
__device__ __forceinline__ int lane_id_fresh() {
    int l = lane_id();
#ifndef MLKEM_EMU
    asm volatile("" : "+v"(l));
#endif
    return l;
}
// Hand-over between the waves of ONE workgroup through a counter in LDS, for stages that not every wave takes part in (a workgroup
// barrier would make all of them wait for the slowest): the producer's earlier writes (LDS or global) are visible to a wave that
// has seen the counter reach its value.  Waves of a workgroup are co-resident, so a waiting wave cannot keep its producer from running.
__device__ __forceinline__ void flag_signal(uint32_t* flag) {
#ifdef MLKEM_EMU
    emu::wave_barrier();
    if ((threadIdx.x & 63) == 0) __atomic_fetch_add(flag, 1u, __ATOMIC_SEQ_CST);
    emu::wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // lane_id_fresh(): the condition must not be a loop invariant the compiler can see.  With `lane_id() == 0` here and in take_job
    // the job loop was unswitched on it -- lane 0 and lanes 1..63 of ONE wave in two copies of the loop -- and the copy without
    // lane 0 read job number 0 from readfirstlane for ever (the first GPU run of this code hung; the emulator cannot show it)
    if (lane_id_fresh() == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
__device__ __forceinline__ void flag_wait(const uint32_t* flag, uint32_t want) {
#ifdef MLKEM_EMU
    while (__atomic_load_n(flag, __ATOMIC_SEQ_CST) < want) sched_yield();
    emu::wave_barrier();
#else
    // the value is read into an SGPR (wave-uniform: a scalar loop); the poll count is bounded so that a logic error ends in a trap
    // (the call fails) instead of a wave that never finishes
    unsigned polls = 0;
    while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < want) {
        __builtin_amdgcn_s_sleep(4);
        if (++polls > (1u << 22)) __builtin_trap();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}
// Order wave-private LDS traffic between lanes of ONE wave (DS ops of a wave execute in issue order;
// this only stops the compiler from moving them).
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// stop the instruction scheduler from moving anything across this point (no code is emitted): used to bound how many
// LDS reads are in flight, i.e. how many VGPRs a staging step may hold next to the 50-register Keccak state
__device__ __forceinline__ void sched_fence() {
#ifndef MLKEM_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// LDS-DMA (global_load_lds_dword): every enabled lane fetches the 4 bytes at uniform_base + its lane_offset and the
// memory pipe writes them to dword `lane` of the 256-byte LDS chunk `lds_chunk` (wave-uniform) without passing
// through VGPRs; completion is tracked by vmcnt.  lds_dma_begin() orders the issue after the wave's earlier LDS
// reads of the same buffer; lds_dma_wait() must precede the first LDS read of the fetched data.
__device__ __forceinline__ void lds_dma_dword(const uint8_t* uniform_base, unsigned lane_offset, uint32_t* lds_chunk) {
#ifdef MLKEM_EMU
    lds_chunk[lane_id()] = *reinterpret_cast<const uint32_t*>(uniform_base + lane_offset);
#else
    // wave-uniform 64-bit base + zero-extended 32-bit per-lane offset: selects the SGPR-base + VGPR-offset addressing
    // form, so a DMA instruction needs one VGPR instead of a 64-bit per-lane address
    typedef const __attribute__((address_space(1))) uint8_t* gptr_t;
    gptr_t g = (gptr_t)uniform_base;
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(g + lane_offset),
        reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(reinterpret_cast<uintptr_t>(lds_chunk)), 4, 0, 0);
#endif
}
__device__ __forceinline__ void lds_dma_begin() {
#ifndef MLKEM_EMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void lds_dma_wait() {
#ifndef MLKEM_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ----------------------------------------------------------------------------------------------
// bit-packed codecs (ml_kem.c:125-177) through a wave-private LDS byte buffer.
// Lane l owns coefficients 4l..4l+3, i.e. bits [4 l d, 4 l d + 4 d) of the stream.
// `buf` must hold 32*d bytes rounded up to a dword multiple plus 8 bytes of slack.
// ----------------------------------------------------------------------------------------------
constexpr int CODEC_BUF_WORDS = 96 + 4;   // 384 B (d = 12) + slack

// stage 32*D bytes from global (dword-aligned) into LDS
template <int D>
__device__ __forceinline__ void codec_load_bytes(uint32_t* buf, const uint8_t* g) {
    const int l = lane_id();
    const uint32_t* gw = reinterpret_cast<const uint32_t*>(g);
#pragma unroll
    for (int w = l; w < 8 * D; w += 64) buf[w] = gw[w];
    if (l < 4) buf[8 * D + l] = 0;
}
// ByteDecode_D: 4 values per lane (raw D-bit fields; no reduction even for D = 12: ml_kem.c:170, F3)
template <int D>
__device__ __forceinline__ void codec_decode(const uint32_t* buf, unsigned (&v)[4]) {
    const int l = lane_id();
    const unsigned bit = 4u * D * (unsigned)l, w = bit >> 5, sh = bit & 31;
    uint32_t d0 = buf[w], d1 = buf[w + 1], d2 = buf[w + 2];
    uint64_t lo = ((uint64_t)d1 << 32) | d0;
    uint64_t val = sh ? ((lo >> sh) | ((uint64_t)d2 << (64 - sh))) : lo;
#pragma unroll
    for (int m = 0; m < 4; m++) v[m] = (unsigned)(val >> (m * D)) & ((1u << D) - 1);
}
// ByteEncode_D: each lane ORs its 4*D bits into a zeroed buffer
template <int D>
__device__ __forceinline__ void codec_zero(uint32_t* buf) {
    const int l = lane_id();
#pragma unroll
    for (int w = l; w < 8 * D + 4; w += 64) buf[w] = 0;
}
template <int D>
__device__ __forceinline__ void codec_encode(uint32_t* buf, const unsigned (&v)[4]) {
    const int l = lane_id();
    uint64_t val = 0;
#pragma unroll
    for (int m = 0; m < 4; m++) val |= (uint64_t)(v[m] & ((1u << D) - 1)) << (m * D);
    const unsigned bit = 4u * D * (unsigned)l, w = bit >> 5, sh = bit & 31;
    uint64_t lo = val << sh;
    uint32_t hi = sh ? (uint32_t)(val >> (64 - sh)) : 0u;
    atomicOr(&buf[w], (uint32_t)lo);
    atomicOr(&buf[w + 1], (uint32_t)(lo >> 32));
    if (4 * D + 31 > 64) atomicOr(&buf[w + 2], hi);
}
// write 32*D bytes LDS -> global (dword-aligned)
template <int D>
__device__ __forceinline__ void codec_store_bytes(const uint32_t* buf, uint8_t* g) {
    const int l = lane_id();
    uint32_t* gw = reinterpret_cast<uint32_t*>(g);
#pragma unroll
    for (int w = l; w < 8 * D; w += 64) gw[w] = buf[w];
}
// compare 32*D bytes LDS vs global; returns nonzero in some lane iff any dword differs
template <int D>
__device__ __forceinline__ uint32_t codec_diff_bytes(const uint32_t* buf, const uint8_t* g) {
    const int l = lane_id();
    const uint32_t* gw = reinterpret_cast<const uint32_t*>(g);
    uint32_t diff = 0;
#pragma unroll
    for (int w = l; w < 8 * D; w += 64) diff |= buf[w] ^ gw[w];
    return diff;
}

// ----------------------------------------------------------------------------------------------
// SamplePolyCBD (ml_kem.c:253-275) for the lane's 4 coefficients (NAT layout), centred output in [-eta, eta]
// `prf` points at the 64*eta PRF bytes of this polynomial in global memory.
// ----------------------------------------------------------------------------------------------
template <int ETA>
__device__ __forceinline__ void cbd_nat(const uint8_t* prf, int (&x)[4]) {
    const int l = lane_id();
    if constexpr (ETA == 2) {
        // 4 coefficients = 16 bits = bytes 2l, 2l+1
        unsigned t = *reinterpret_cast<const uint16_t*>(prf + 2 * l);
        unsigned d = (t & 0x5555u) + ((t >> 1) & 0x5555u);   // pairwise bit sums
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = (int)((d >> (4 * m)) & 3u) - (int)((d >> (4 * m + 2)) & 3u);
    } else {
        // 4 coefficients = 24 bits = bytes 3l..3l+2
        unsigned t = (unsigned)prf[3 * l] | ((unsigned)prf[3 * l + 1] << 8) | ((unsigned)prf[3 * l + 2] << 16);
        unsigned d = (t & 0x249249u) + ((t >> 1) & 0x249249u) + ((t >> 2) & 0x249249u);   // 3-bit group sums
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = (int)((d >> (6 * m)) & 7u) - (int)((d >> (6 * m + 3)) & 7u);
    }
}

}   // namespace mlkem
