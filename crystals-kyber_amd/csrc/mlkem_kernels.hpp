// mlkem_kernels.hpp — the gfx950 kernels of the batched ML-KEM engine (kernel bodies only; launch code and
// the C-ABI live in mlkem_capi.hip).
//
// Kernel families:
//   k_hash_*      lane = one KEM instance; serial sponges H / G / J with inputs staged through LDS so that
//                 the packed per-item byte strings are read from HBM in coalesced runs
//   k_sample      lane = one SHAKE128 sponge; role XOF -> SampleNTT rejection sampling (ml_kem.c:189) with
//                 an LDS ring per lane flushed in aligned 128-byte chunks; role PRF -> raw PRF bytes (ml_kem.c:496)
//                 (its three-block fast path lives in mlkem_sampler.hpp)
// The polynomial kernels: K-PKE.KeyGen / Encrypt two items per wave in mlkem_kpke2.hpp (k_keygen2, k_encrypt2), K-PKE.Decrypt four
// items per wave in mlkem_kpke4.hpp (k_decrypt4), the stand-alone transforms in mlkem_rntt.hpp, the other primitives in mlkem_arith.hpp;
// one-workgroup-per-item kernels for small calls in mlkem_small.hpp (on the one-sponge-per-wave Keccak of mlkem_wkeccak.hpp).
//
// No kernel uses a workgroup barrier: every wave is independent, LDS is carved per wave.
#pragma once
#include "mlkem_device.hpp"

namespace mlkem {

constexpr int WAVE = 64;

// Resident waves per SIMD the Keccak kernels are compiled for (second __launch_bounds__ argument = register budget
// 512 / N per lane).  The lane-sliced permutation is a long dependent VALU stream: measured with tools/keccak_ubench.hip
// it costs 357 / 313 / 288 / 270 ns per round per wave at 3 / 4 / 5 / 6 resident waves, so occupancy is worth more than
// anything the compiler buys with extra registers.
#ifndef MLKEM_KECCAK_MINWAVES
#define MLKEM_KECCAK_MINWAVES 6
#endif

// ================================================================================================
// LDS-staged absorption: 64 lanes each own one sponge.  For every rate block the wave reads the 64 rows
// (one per item) x RATE bytes of a virtual message made of two segments (seg0 then seg1, each with its own
// base / per-item stride) as RATE/8 fully coalesced 8-byte-per-lane loads (flat index f = it*64 + lane ->
// row f / NQ, qword f % NQ), stores them linearly into LDS, and every lane then XORs its own row into its
// state.  NQ = RATE/8 is odd for all three rates (17, 21, 9), so the per-lane ds_read_b64 row reads are
// bank-conflict free without padding.  Bytes past the end of the message are staged as zeros.
// Requirements: bases and strides 8-byte aligned; len0 % 8 == 0 whenever len1 != 0.
// ================================================================================================
struct MsgView {
    const uint8_t* p0; size_t stride0; unsigned len0;   // segment 0 (may have len0 = 0)
    const uint8_t* p1; size_t stride1; unsigned len1;   // segment 1
};

__device__ __forceinline__ uint2 load_qword_tail(const uint8_t* p, unsigned avail) {
    // p is 8-byte aligned; `avail` (>= 1) bytes are readable
    if (avail >= 8) return *reinterpret_cast<const uint2*>(p);
    uint2 v;
    v.x = 0; v.y = 0;
    for (unsigned i = 0; i < avail; i++) {
        const uint32_t b = p[i];
        if (i < 4) v.x |= b << (8 * i);
        else v.y |= b << (8 * (i - 4));
    }
    return v;
}

// The stage holds STAGE_ROWS rows at a time (a rate block of the wave is staged in 64 / STAGE_ROWS parts): 32 rows keep
// the LDS footprint at 5.3 KB per wave so that occupancy is set by registers, not by LDS.
#ifndef MLKEM_STAGE_ROWS
#define MLKEM_STAGE_ROWS 32
#endif
#ifndef MLKEM_STAGE_DMA
#define MLKEM_STAGE_DMA 1
#endif
constexpr int STAGE_ROWS = MLKEM_STAGE_ROWS;
constexpr int STAGE_GROUP = MLKEM_STAGE_DMA ? 2 : 7;
constexpr int STAGE_PARTS = WAVE / STAGE_ROWS;

template <int RATE>
__device__ __forceinline__ void wave_stage_block(uint2* stage, const MsgView& mv, size_t item0, size_t n_items, unsigned voff,
                                                 int part) {
    constexpr int NQ = RATE / 8, TOT = STAGE_ROWS * NQ, ITERS = (TOT + WAVE - 1) / WAVE;
    const unsigned l = (unsigned)lane_id(), total = mv.len0 + mv.len1;
    // loads are issued in groups of STAGE_GROUP (x 8 B per lane in flight).  With the LDS-DMA path compiled in, this
    // synchronous path only serves ragged last waves and odd message shapes, so it is kept register-lean (groups of 2);
    // without it, groups of 7 give enough memory-level parallelism next to the 50-register Keccak state.
#pragma unroll 1
    for (int it0 = 0; it0 < ITERS; it0 += STAGE_GROUP)
#pragma unroll
    for (int it = it0; it < it0 + STAGE_GROUP; it++) {
        if (it >= ITERS) break;
        const unsigned f = (unsigned)it * WAVE + l;
        if (TOT % WAVE != 0 && f >= (unsigned)TOT) break;
        const unsigned row = f / NQ, col = f - row * NQ;
        size_t item = item0 + (size_t)(part * STAGE_ROWS) + row;
        if (item >= n_items) item = n_items - 1;   // rows beyond the batch are computed but never stored
        const unsigned pos = voff + 8u * col;
        uint2 v;
        v.x = 0; v.y = 0;
        if (pos < total) {
            if (pos < mv.len0) v = load_qword_tail(mv.p0 + item * mv.stride0 + pos, mv.len0 - pos);
            else v = load_qword_tail(mv.p1 + item * mv.stride1 + (pos - mv.len0), total - pos);
        }
        stage[f] = v;
    }
}

// XOR the staged row (RATE bytes) of every lane that belongs to part `part` into its state
template <int RATE>
__device__ __forceinline__ void lane_xor_row(KeccakState& s, const uint2* stage, int part) {
    constexpr int NQ = RATE / 8;
    const int l = lane_id();
    if (l / STAGE_ROWS == part) {
        const uint2* row = stage + (l % STAGE_ROWS) * NQ;
#pragma unroll
        for (int w = 0; w < NQ; w++) {
            const uint2 v = row[w];
            s.lo[w] ^= v.x;
            s.hi[w] ^= v.y;
        }
    }
}
// stage + XOR one rate block (or the final partial block) for all 64 lanes
template <int RATE>
__device__ __forceinline__ void wave_absorb_block(KeccakState& s, uint2* stage, const MsgView& mv, size_t item0, size_t n_items,
                                                  unsigned voff) {
#pragma unroll 1
    for (int part = 0; part < STAGE_PARTS; part++) {
        wave_stage_block<RATE>(stage, mv, item0, n_items, voff, part);
        wave_lds_fence();
        lane_xor_row<RATE>(s, stage, part);
        wave_lds_fence();
    }
}

// ------------------------------------------------------------------------------------------------
// Asynchronous variant of the staging above (MLKEM_STAGE_DMA, default on): the rate block is fetched with LDS-DMA
// loads (mlkem_device.hpp: lds_dma_dword) that cost neither VGPRs nor issue slots for LDS stores.  The wave's 64 rows
// are staged as two halves of 32 rows through ONE 32-row buffer (4.5 KB at rate 136, 5.5 KB at rate 168, so that six
// waves per SIMD stay resident): half 0 of block b+1 is issued right after block b has been XORed into the states and
// is in flight during the permutation of block b; half 1 follows as soon as half 0 has been consumed.
// One DMA instruction moves 16 rows x 16 bytes (lane -> row lane/4, dword lane%4: 16-byte runs per row, one base per
// row group); the buffer is laid out [row group g = 0..1][16-byte column j][row 0..15][4 dwords], so lane L later reads
// its row (group (L/16)%2, row L%16) with one conflict-free ds_read_b128 per column.
// Preconditions (checked per sponge, wave-uniform; otherwise the synchronous path above runs): a full wave of items,
// message length % 4 == 0, len0 % 16 == 0 when there are two segments, strides < 2^27.
// ------------------------------------------------------------------------------------------------
#ifndef MLKEM_STAGE_DMA
#define MLKEM_STAGE_DMA 1
#endif
template <int RATE>
struct DmaStage {
    static constexpr int NW = RATE / 4;            // dwords per row
    static constexpr int NJ = (NW + 3) / 4;        // 16-byte columns per row
    static constexpr int DWORDS = 2 * NJ * WAVE;   // 2 row groups x NJ chunks x 64 dwords
};

// issue the loads of rows 32*half .. 32*half+31 of the rate block at message offset voff;
// valid = message bytes in this block (<= RATE, % 4 == 0)
template <int RATE>
__device__ __forceinline__ void wave_dma_half(uint32_t* stage, const MsgView& mv, size_t item0, unsigned voff, unsigned valid,
                                              int half) {
    using D = DmaStage<RATE>;
    const unsigned l = (unsigned)lane_id_fresh(), rr = l >> 2, c4 = 4u * (l & 3u);
    const unsigned off0 = rr * (unsigned)mv.stride0 + c4, off1 = rr * (unsigned)mv.stride1 + c4;
    lds_dma_begin();
#pragma unroll
    for (int j = 0; j < D::NJ; j++) {
        const unsigned pos = 16u * (unsigned)j;
        const bool seg0 = voff + pos < mv.len0;
        const uint8_t* base = seg0 ? mv.p0 + (voff + pos) : mv.p1 + (voff + pos - mv.len0);
        const size_t stride = seg0 ? mv.stride0 : mv.stride1;
        const unsigned off = seg0 ? off0 : off1;
        if (pos + c4 < valid) {
#pragma unroll
            for (int g = 0; g < 2; g++)
                lds_dma_dword(base + (item0 + (size_t)(32 * half + 16 * g)) * stride, off, stage + (g * D::NJ + j) * WAVE);
        }
    }
}

// XOR the first `valid` bytes of the staged row into the state of every lane of half `half`.  Branch-free across lanes:
// every lane reads the row slot (L % 32) and folds it in with s ^= (v & m), m = all-ones in the owning half (one
// v_bitop3 per dword) -- a lane-divergent `if` around 2 x 25 state registers makes the compiler keep two copies of the
// state.  FULL = the whole rate block is message (no per-dword length tests).
template <int RATE, bool FULL>
__device__ __forceinline__ void lane_xor_dma(KeccakState& s, const uint32_t* stage, unsigned valid, int half) {
    using D = DmaStage<RATE>;
    const int l = lane_id_fresh();
    const uint32_t m = (l >> 5) == half ? 0xFFFFFFFFu : 0u;
    const uint32_t* row = stage + ((l >> 4) & 1) * (D::NJ * WAVE) + (l & 15) * 4;
#pragma unroll
    for (int j = 0; j < D::NJ; j++) {
        if (FULL || 16u * (unsigned)j < valid) {   // wave-uniform
            const uint4 v = *reinterpret_cast<const uint4*>(row + WAVE * j);
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int w = 4 * j + k;
                if (w < D::NW && (FULL || 4u * (unsigned)w < valid)) {
                    if (w & 1) s.hi[w >> 1] = __builtin_amdgcn_bitop3_b32(s.hi[w >> 1], d[k], m, 0x78);   // a ^ (b & c)
                    else s.lo[w >> 1] = __builtin_amdgcn_bitop3_b32(s.lo[w >> 1], d[k], m, 0x78);
                }
            }
        }
    }
}
// Make the first NQ state lanes "opaque" at this point (no code): the XORs that produced them cannot be sunk past it.
// Without it the compiler keeps the 34-42 staged dwords of half 0 in registers, fetches half 1, and folds both in at
// once -- twice the staging registers next to the 50-register state.
template <int NQ>
__device__ __forceinline__ void pin_state(KeccakState& s) {
#ifndef MLKEM_EMU
#pragma unroll
    for (int w = 0; w < NQ; w++) {
        asm volatile("" : "+v"(s.lo[w]));
        asm volatile("" : "+v"(s.hi[w]));
    }
#endif
}
// both halves of one rate block: half 0 is already in flight on entry
template <int RATE, bool FULL>
__device__ __forceinline__ void wave_absorb_dma(KeccakState& s, uint32_t* st32, const MsgView& mv, size_t item0, unsigned voff,
                                                unsigned valid) {
    lds_dma_wait();
    wave_lds_fence();
    lane_xor_dma<RATE, FULL>(s, st32, valid, 0);
    pin_state<RATE / 8>(s);
    wave_lds_fence();
    wave_dma_half<RATE>(st32, mv, item0, voff, valid, 1);
    lds_dma_wait();
    wave_lds_fence();
    lane_xor_dma<RATE, FULL>(s, st32, valid, 1);
    pin_state<RATE / 8>(s);
    wave_lds_fence();
}

// xor `byte` at runtime byte position `pos` (wave-uniform) — used for the domain/pad byte of a
// message whose length is only known at run time
__device__ __forceinline__ void keccak_xor_byte_rt(KeccakState& s, unsigned pos, uint32_t byte) {
    const unsigned w = pos >> 2;
    const uint32_t v = byte << (8 * (pos & 3));
#define MLKEM_XB(W) keccak_word<W>(s) ^= (w == W) ? v : 0u;   // select, not `if`: an if-chain is turned into a dynamic state index (scratch)
    MLKEM_XB(0) MLKEM_XB(1) MLKEM_XB(2) MLKEM_XB(3) MLKEM_XB(4) MLKEM_XB(5) MLKEM_XB(6) MLKEM_XB(7)
    MLKEM_XB(8) MLKEM_XB(9) MLKEM_XB(10) MLKEM_XB(11) MLKEM_XB(12) MLKEM_XB(13) MLKEM_XB(14) MLKEM_XB(15)
    MLKEM_XB(16) MLKEM_XB(17) MLKEM_XB(18) MLKEM_XB(19) MLKEM_XB(20) MLKEM_XB(21) MLKEM_XB(22) MLKEM_XB(23)
    MLKEM_XB(24) MLKEM_XB(25) MLKEM_XB(26) MLKEM_XB(27) MLKEM_XB(28) MLKEM_XB(29) MLKEM_XB(30) MLKEM_XB(31)
    MLKEM_XB(32) MLKEM_XB(33) MLKEM_XB(34) MLKEM_XB(35) MLKEM_XB(36) MLKEM_XB(37) MLKEM_XB(38) MLKEM_XB(39)
    MLKEM_XB(40) MLKEM_XB(41)
#undef MLKEM_XB
}

// Full sponge absorb of a (two-segment) message of wave-uniform length; leaves the state after the
// final permutation (ready to squeeze).  sha3.c:257-291 + :408-436; SUFFIX = 0x06 (SHA-3) / 0x1F (SHAKE).
template <int RATE, int SUFFIX>
__device__ __forceinline__ void wave_sponge_absorb(KeccakState& s, uint2* stage, const MsgView& mv, size_t item0,
                                                   size_t n_items) {
    const unsigned total = mv.len0 + mv.len1;
    keccak_zero(s);
#if MLKEM_STAGE_DMA
    if (item0 + WAVE <= n_items && total != 0 && total % 4 == 0 && (mv.len1 == 0 || mv.len0 % 16 == 0) &&
        mv.stride0 < (1u << 27) && mv.stride1 < (1u << 27)) {
        uint32_t* st32 = reinterpret_cast<uint32_t*>(stage);
        const unsigned nfull = total / (unsigned)RATE, rem = total - nfull * (unsigned)RATE;
        wave_dma_half<RATE>(st32, mv, item0, 0, nfull ? (unsigned)RATE : rem, 0);
#pragma unroll 1
        for (unsigned b = 0; b < nfull; b++) {
            wave_absorb_dma<RATE, true>(s, st32, mv, item0, b * RATE, RATE);
            // half 0 of the next block (or of the partial tail) is in flight during the permutation
            if (b + 1 < nfull) wave_dma_half<RATE>(st32, mv, item0, (b + 1) * RATE, RATE, 0);
            else if (rem) wave_dma_half<RATE>(st32, mv, item0, nfull * RATE, rem, 0);
            keccak_f1600(s);
        }
        if (rem) wave_absorb_dma<RATE, false>(s, st32, mv, item0, nfull * RATE, rem);
        keccak_xor_byte_rt(s, rem, SUFFIX);
        keccak_xor_byte<RATE - 1>(s, 0x80);
        keccak_f1600(s);
        return;
    }
#endif
    unsigned voff = 0;
    while (total - voff >= (unsigned)RATE) {
        wave_absorb_block<RATE>(s, stage, mv, item0, n_items, voff);
        keccak_f1600(s);
        voff += RATE;
    }
    const unsigned rem = total - voff;
    if (rem) wave_absorb_block<RATE>(s, stage, mv, item0, n_items, voff);
    keccak_xor_byte_rt(s, rem, SUFFIX);
    keccak_xor_byte<RATE - 1>(s, 0x80);
    keccak_f1600(s);
}

// stage size per wave: the synchronous path needs STAGE_ROWS rows of the largest rate (5376 bytes at 32 rows); the DMA
// path needs 32 rows in 16-byte columns (DmaStage).  Kernels that only hash at rate 136 use the smaller figure.
constexpr int stage_qwords(int max_rate) {
    const int sync_q = STAGE_ROWS * (max_rate / 8);
    const int dma_q = MLKEM_STAGE_DMA ? 2 * ((max_rate / 4 + 3) / 4) * WAVE / 2 : 0;
    return sync_q > dma_q ? sync_q : dma_q;
}
constexpr int STAGE_QWORDS = stage_qwords(168);

// store / load 8 dwords (32 bytes) of per-item data: row `item` of a [n][32]-byte array
__device__ __forceinline__ void store32(uint8_t* base, size_t stride, size_t item, const uint32_t (&w)[8]) {
    uint32_t* p = reinterpret_cast<uint32_t*>(base + item * stride);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = w[i];
}
__device__ __forceinline__ void load32(const uint8_t* base, size_t stride, size_t item, uint32_t (&w)[8]) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(base + item * stride);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = p[i];
}
#define MLKEM_STATE_WORDS8(s, FIRST, w)                                                             \
    {                                                                                               \
        w[0] = keccak_word<FIRST + 0>(s); w[1] = keccak_word<FIRST + 1>(s); w[2] = keccak_word<FIRST + 2>(s); \
        w[3] = keccak_word<FIRST + 3>(s); w[4] = keccak_word<FIRST + 4>(s); w[5] = keccak_word<FIRST + 5>(s); \
        w[6] = keccak_word<FIRST + 6>(s); w[7] = keccak_word<FIRST + 7>(s);                         \
    }
#define MLKEM_SET_WORDS8(s, FIRST, w)                                                               \
    {                                                                                               \
        keccak_word<FIRST + 0>(s) = w[0]; keccak_word<FIRST + 1>(s) = w[1]; keccak_word<FIRST + 2>(s) = w[2]; \
        keccak_word<FIRST + 3>(s) = w[3]; keccak_word<FIRST + 4>(s) = w[4]; keccak_word<FIRST + 5>(s) = w[5]; \
        keccak_word<FIRST + 6>(s) = w[6]; keccak_word<FIRST + 7>(s) = w[7];                         \
    }

// G(x || y) for two 32-byte halves already in registers: SHA3-512, one permutation (ml_kem.c:559-572)
__device__ __forceinline__ void lane_G64(KeccakState& s, const uint32_t (&x)[8], const uint32_t (&y)[8]) {
    keccak_zero(s);
    MLKEM_SET_WORDS8(s, 0, x)
    MLKEM_SET_WORDS8(s, 8, y)
    keccak_xor_byte<64>(s, 0x06);
    keccak_xor_byte<71>(s, 0x80);
    keccak_f1600(s);
}

// ------------------------------------------------------------------------------------------------
// k_hash_encaps — Encaps_internal's hashing (ml_kem.c:1108-1124): h = H(ek); (K, r) = G(m || h)
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(WAVE, MLKEM_KECCAK_MINWAVES) k_hash_encaps(size_t n, const uint8_t* __restrict__ ek, const uint8_t* __restrict__ m,
                                                      uint8_t* __restrict__ Kout, uint8_t* __restrict__ r_ws) {
    __shared__ __attribute__((aligned(16))) uint2 stage[stage_qwords(136)];
    constexpr unsigned EK = 384 * K + 32;
    const size_t item0 = (size_t)blockIdx.x * WAVE, item = item0 + lane_id();
    KeccakState s;
    MsgView mv{ek, EK, EK, ek, EK, 0};
    wave_sponge_absorb<136, 0x06>(s, stage, mv, item0, n);
    uint32_t h[8], mm[8], w[8];
    MLKEM_STATE_WORDS8(s, 0, h)
    const size_t it = item < n ? item : n - 1;
    load32(m, 32, it, mm);
    lane_G64(s, mm, h);
    if (item < n) {
        MLKEM_STATE_WORDS8(s, 0, w)
        store32(Kout, 32, item, w);
        MLKEM_STATE_WORDS8(s, 8, w)
        store32(r_ws, 32, item, w);
    }
}

// ------------------------------------------------------------------------------------------------
// Shared-key batches (one ek / dk for all n items): H(ek) is computed once (k_hash_batch<0>, one item) and
//   k_hash_g_shared  : (K_i, r_i) = G(m_i || h) per item (ml_kem.c:1113-1124) with the broadcast h
//   k_status_fill    : KEM_Decaps' hash check (ml_kem.c:1336-1350) evaluated once: status[i] = (h_calc == h_stored) ? 0 : -5
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE, MLKEM_KECCAK_MINWAVES) k_hash_g_shared(size_t n, const uint8_t* __restrict__ m, const uint8_t* __restrict__ h,
                                                                               uint8_t* __restrict__ Kout, uint8_t* __restrict__ r_ws) {
    const size_t item = (size_t)blockIdx.x * WAVE + lane_id();
    const size_t it = item < n ? item : n - 1;
    uint32_t mm[8], hh[8], w[8];
    load32(m, 32, it, mm);
    load32(h, 0, 0, hh);
    KeccakState s;
    lane_G64(s, mm, hh);
    if (item < n) {
        MLKEM_STATE_WORDS8(s, 0, w)
        store32(Kout, 32, item, w);
        MLKEM_STATE_WORDS8(s, 8, w)
        store32(r_ws, 32, item, w);
    }
}
__global__ void __launch_bounds__(256) k_status_fill(size_t n, const uint8_t* __restrict__ h_calc, const uint8_t* __restrict__ h_stored,
                                                     int32_t* __restrict__ status) {
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) diff |= reinterpret_cast<const uint32_t*>(h_calc)[i] ^ reinterpret_cast<const uint32_t*>(h_stored)[i];
    const int32_t v = diff ? -5 : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) status[i] = v;
}

// ------------------------------------------------------------------------------------------------
// k_hash_decaps — KEM_Decaps hash check (ml_kem.c:1336-1350) and Decaps_internal's hashing
// (ml_kem.c:1187-1202): status = (H(dk.ek) == dk.h) ? 0 : -5 ; (K', r') = G(m' || dk.h) ; Kbar = J(dk.z || c)
// J is SHAKE128 in the reference (F2): JRATE = 168; the FIPS 203 mode uses SHAKE256: JRATE = 136.
// The three sponges of an item are independent.  With HASH_CHECK the grid holds TWO waves per 64 items: blocks [0, nb) run
// H(ek) (9 permutations for k = 3), blocks [nb, 2 nb) run J and G (7 + 1) -- the same permutations in all, but a chain of 9
// instead of 17 per lane: below ~2^16 items the GPU is not full and the kernel's time is the length of that chain
// (0.174 -> 0.10 ms at 64..16384 items), at 2^20 it is neutral (profiles/r03_batch_sweep.txt).
// ------------------------------------------------------------------------------------------------
template <int K, int CLEN, bool HASH_CHECK, int JRATE = 168>
__global__ void __launch_bounds__(WAVE, MLKEM_KECCAK_MINWAVES) k_hash_decaps(size_t n, const uint8_t* __restrict__ dk, const uint8_t* __restrict__ c,
                                                      const uint8_t* __restrict__ m_ws, uint8_t* __restrict__ Kp_ws,
                                                      uint8_t* __restrict__ r_ws, uint8_t* __restrict__ Kbar_ws,
                                                      int32_t* __restrict__ status, size_t dk_stride) {
    // dk_stride: bytes between the decapsulation keys of consecutive items (768k+96), or 0 for a shared-key batch
    __shared__ __attribute__((aligned(16))) uint2 stage[STAGE_QWORDS];
    constexpr unsigned EK = 384 * K + 32;
    const size_t DK = dk_stride;
    const size_t nb = (n + WAVE - 1) / WAVE;
    const bool check_role = HASH_CHECK && blockIdx.x < nb;
    const size_t item0 = (size_t)(HASH_CHECK && !check_role ? blockIdx.x - nb : blockIdx.x) * WAVE;
    KeccakState s;
    uint32_t h[8], w[8];
    // The stored h = dk[768k+32 ..] is loaded only where it is used (after the sponge that precedes its use), and the lane's
    // item index is recomputed from a fresh lane id there: nothing but the 50 state registers lives across the permutations
    // (with h and the row pointers held across them the kernel spilled 60-92 bytes per lane).
    auto my_item = [&]() {
        const size_t i2 = item0 + (size_t)lane_id_fresh();
        return i2 < n ? i2 : n - 1;
    };
    if (check_role) {
        MsgView mv{dk + 384 * K, DK, EK, dk, DK, 0};
        wave_sponge_absorb<136, 0x06>(s, stage, mv, item0, n);
        MLKEM_STATE_WORDS8(s, 0, w)
        load32(dk + 768 * K + 32, DK, my_item(), h);
        uint32_t diff = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) diff |= w[i] ^ h[i];
        const size_t i2 = item0 + (size_t)lane_id_fresh();
        if (i2 < n && status) status[i2] = diff ? -5 : 0;
        return;
    }
    // Kbar = J(z || c)
    {
        MsgView mv{dk + 768 * K + 64, DK, 32, c, CLEN, CLEN};
        wave_sponge_absorb<JRATE, 0x1F>(s, stage, mv, item0, n);
        const size_t i2 = item0 + (size_t)lane_id_fresh();
        if (i2 < n) {
            MLKEM_STATE_WORDS8(s, 0, w)
            store32(Kbar_ws, 32, i2, w);
        }
    }
    // (K', r') = G(m' || h)
    uint32_t mm[8];
    const size_t it = my_item();
    load32(m_ws, 32, it, mm);
    load32(dk + 768 * K + 32, DK, it, h);
    lane_G64(s, mm, h);
    const size_t item = item0 + (size_t)lane_id_fresh();
    if (item < n) {
        MLKEM_STATE_WORDS8(s, 0, w)
        store32(Kp_ws, 32, item, w);
        MLKEM_STATE_WORDS8(s, 8, w)
        store32(r_ws, 32, item, w);
        if (!HASH_CHECK && status) status[item] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// k_hash_keygen_seed — K-PKE.KeyGen's (rho, sigma) = G(d || k) (ml_kem.c:674-681)
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(WAVE, MLKEM_KECCAK_MINWAVES) k_hash_keygen_seed(size_t n, const uint8_t* __restrict__ d, uint8_t* __restrict__ rho_ws,
                                                           uint8_t* __restrict__ sigma_ws) {
    const size_t item = (size_t)blockIdx.x * WAVE + lane_id();
    const size_t it = item < n ? item : n - 1;
    KeccakState s;
    uint32_t dd[8], w[8];
    load32(d, 32, it, dd);
    keccak_zero(s);
    MLKEM_SET_WORDS8(s, 0, dd)
    keccak_xor_byte<32>(s, (uint32_t)K);
    keccak_xor_byte<33>(s, 0x06);
    keccak_xor_byte<71>(s, 0x80);
    keccak_f1600(s);
    if (item < n) {
        MLKEM_STATE_WORDS8(s, 0, w)
        store32(rho_ws, 32, item, w);
        MLKEM_STATE_WORDS8(s, 8, w)
        store32(sigma_ws, 32, item, w);
    }
}

// ------------------------------------------------------------------------------------------------
// k_hash_keygen_fin — KeyGen_internal's dk tail (ml_kem.c:1065-1077): dk[768k+32 ..] = H(ek) ; dk[768k+64 ..] = z
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(WAVE, MLKEM_KECCAK_MINWAVES) k_hash_keygen_fin(size_t n, const uint8_t* __restrict__ ek, const uint8_t* __restrict__ z,
                                                          uint8_t* __restrict__ dk) {
    __shared__ __attribute__((aligned(16))) uint2 stage[stage_qwords(136)];
    constexpr unsigned EK = 384 * K + 32, DK = 768 * K + 96;
    const size_t item0 = (size_t)blockIdx.x * WAVE, item = item0 + lane_id();
    KeccakState s;
    MsgView mv{ek, EK, EK, ek, EK, 0};
    wave_sponge_absorb<136, 0x06>(s, stage, mv, item0, n);
    if (item < n) {
        uint32_t w[8];
        MLKEM_STATE_WORDS8(s, 0, w)
        store32(dk + 768 * K + 32, DK, item, w);
        load32(z, 32, item, w);
        store32(dk + 768 * K + 64, DK, item, w);
    }
}

// ------------------------------------------------------------------------------------------------
// k_hash_batch — stand-alone H / G / J over equal-length messages (parity tests of the staged sponge)
//   KIND 0: H = SHA3-256 (32 B out), 1: G = SHA3-512 (64 B out), 2: J = SHAKE128 (32 B out)
// ------------------------------------------------------------------------------------------------
// (runtime message length and stride: compiled for 4 waves per SIMD = 128 VGPRs; at the hot kernels' 6 waves = 80 VGPRs the
// run-time MsgView spilled 56 bytes per lane inside the absorb loop)
template <int KIND>
__global__ void __launch_bounds__(WAVE, 4) k_hash_batch(size_t n, const uint8_t* __restrict__ msg, unsigned len,
                                                                            size_t stride, uint8_t* __restrict__ out) {
    constexpr int RATE = KIND == 0 ? 136 : KIND == 1 ? 72 : 168;
    __shared__ __attribute__((aligned(16))) uint2 stage[stage_qwords(RATE)];
    const size_t item0 = (size_t)blockIdx.x * WAVE;
    KeccakState s;
    MsgView mv{msg, stride, len, msg, stride, 0};
    uint32_t w[8];
    wave_sponge_absorb<RATE, KIND == 2 ? 0x1F : 0x06>(s, stage, mv, item0, n);
    const size_t item = item0 + (size_t)lane_id_fresh();
    if (item < n) {
        MLKEM_STATE_WORDS8(s, 0, w)
        store32(out, KIND == 1 ? 64 : 32, item, w);
        if (KIND == 1) {
            MLKEM_STATE_WORDS8(s, 8, w)
            store32(out + 32, 64, item, w);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_sponge_raw — the bare Keccak sponge (sha3.c:257-317) over n already-padded messages: absorb `nblocks` rate
// blocks per lane, squeeze `outlen` bytes.  Suffix bits and pad10*1 are applied by the caller (bit-granular lengths:
// the sha3.h front-ends in the drop-in shim).  Message i at msg + i*nblocks*RATE, output i at
// out + i*out_stride (out_stride % 4 == 0).
// ------------------------------------------------------------------------------------------------
template <int RATE>
__global__ void __launch_bounds__(WAVE) k_sponge_raw(size_t n, const uint8_t* __restrict__ msg, unsigned nblocks, uint8_t* __restrict__ out,
                                                     unsigned outlen, size_t out_stride) {
    __shared__ __attribute__((aligned(16))) uint2 stage[STAGE_QWORDS];
    const size_t item0 = (size_t)blockIdx.x * WAVE, item = item0 + lane_id();
    KeccakState s;
    const unsigned total = nblocks * RATE;
    MsgView mv{msg, total, total, msg, total, 0};
    keccak_zero(s);
    for (unsigned voff = 0; voff < total; voff += RATE) {
        wave_absorb_block<RATE>(s, stage, mv, item0, n, voff);
        keccak_f1600(s);
    }
    uint8_t* o = out + (item < n ? item : n - 1) * out_stride;
    for (unsigned done = 0; done < outlen; done += RATE) {
        if (done) keccak_f1600(s);
        const unsigned take = outlen - done < (unsigned)RATE ? outlen - done : (unsigned)RATE;
        if (item < n) {
#define MLKEM_SQ(W)                                                                                         \
            if constexpr (W < RATE / 4) {                                                                   \
                if (4u * W + 4u <= take) *reinterpret_cast<uint32_t*>(o + done + 4 * W) = keccak_word<W>(s); \
                else if (4u * W < take) {                                                                   \
                    const uint32_t v = keccak_word<W>(s);                                                   \
                    for (unsigned b = 0; 4u * W + b < take; b++) o[done + 4 * W + b] = (uint8_t)(v >> (8 * b)); \
                }                                                                                           \
            }
            MLKEM_SQ(0) MLKEM_SQ(1) MLKEM_SQ(2) MLKEM_SQ(3) MLKEM_SQ(4) MLKEM_SQ(5) MLKEM_SQ(6) MLKEM_SQ(7)
            MLKEM_SQ(8) MLKEM_SQ(9) MLKEM_SQ(10) MLKEM_SQ(11) MLKEM_SQ(12) MLKEM_SQ(13) MLKEM_SQ(14) MLKEM_SQ(15)
            MLKEM_SQ(16) MLKEM_SQ(17) MLKEM_SQ(18) MLKEM_SQ(19) MLKEM_SQ(20) MLKEM_SQ(21) MLKEM_SQ(22) MLKEM_SQ(23)
            MLKEM_SQ(24) MLKEM_SQ(25) MLKEM_SQ(26) MLKEM_SQ(27) MLKEM_SQ(28) MLKEM_SQ(29) MLKEM_SQ(30) MLKEM_SQ(31)
            MLKEM_SQ(32) MLKEM_SQ(33) MLKEM_SQ(34) MLKEM_SQ(35) MLKEM_SQ(36) MLKEM_SQ(37) MLKEM_SQ(38) MLKEM_SQ(39)
            MLKEM_SQ(40) MLKEM_SQ(41)
#undef MLKEM_SQ
        }
    }
}

// ================================================================================================
// k_sample — lane-sliced SHAKE128 sponges.
//   blocks [0, xof_blocks)            : XOF role, sponge g = block*64 + lane  -> SampleNTT (ml_kem.c:189-245)
//   blocks [xof_blocks, +prf_blocks)  : PRF role, sponge g                    -> PRF bytes  (ml_kem.c:496-515)
//
// XOF seeds:  mode MATRIX : g -> (item, row a, col b);  seed = rho[item] || i0 || i1 with (i0, i1) = (a, b)
//                           for Encrypt's A^T (ml_kem.c:817-823) and (b, a) for KeyGen's A (ml_kem.c:686-693);
//                           output polynomial index g  (= item*K*K + a*K + b)
//             mode SEEDS  : g -> explicit 34-byte seed (stand-alone SampleNTT primitive)
// PRF seeds:  g -> (item, counter n) ; input = r[item] || n ; the first `n_eta1` counters produce 64*ETA1
//             bytes, the rest 128 bytes; all rows are PRF_STRIDE bytes apart.
//
// Rejection sampling writes accepted coefficients into a per-lane LDS ring of 128 coefficients and the
// wave flushes every completed, 128-byte aligned 64-coefficient chunk with 8 lanes x 16 B per polynomial.
// ================================================================================================
constexpr int RING = 128;                 // coefficients per lane ring
constexpr int RING_STRIDE = RING + 8;     // int16 units: 272 B rows (16-byte aligned, 4-bank skew)

constexpr int RESUME_WORDS = 52;
struct SampleArgs {
    // XOF role
    size_t n_xof;             // number of XOF sponges
    const uint8_t* rho;       // MATRIX: per-item 32-byte seed base ; SEEDS: 34-byte seeds
    size_t rho_stride;
    int K;                    // MATRIX: matrix dimension ; SEEDS: 0
    int transpose;            // MATRIX: 1 -> seed bytes (a, b) [Encrypt], 0 -> (b, a) [KeyGen]
    uint16_t* A;              // output polynomials [n_xof][256]
    unsigned xof_blocks;
    // PRF role
    size_t n_prf;             // number of PRF sponges
    const uint8_t* r;         // per-item 32-byte seed, stride 32 ; or (PRF primitive) 33-byte inputs
    int per_item;             // counters per item (0 -> explicit 33-byte inputs, stride 33)
    int n_eta1;               // counters < n_eta1 use eta1
    int eta1;
    uint8_t* prf;             // output rows
    unsigned prf_stride;
    // sponges that need more than three squeeze blocks (0.8 %), handed from k_sample_main to the leftover passes:
    //   leftover[1] = number of RESUME records in `resume` (k_sample_resume continues them from the saved sponge state)
    //   leftover[0] = number of entries of the restart list leftover[2..] (sponge indices k_sample in list mode redoes from
    //                 the seed: records beyond resume_cap, the ring forms of k_sample_main, and a resumed sponge that is
    //                 still short after its fifth block, i.e. the reference's seed-mutation retry, ml_kem.c:237-242)
    uint32_t* leftover;
    uint32_t* resume;         // records of RESUME_WORDS dwords: sponge index, accepted coefficients so far, 50 state dwords
    uint32_t resume_cap;
    int list_mode;            // k_sample only: take the sponge indices from `leftover` (grid-stride)
    int prf_rate;             // 0 / 168: PRF on SHAKE128 like the reference (F2) ; 136: SHAKE256 (FIPS 203 mode)
    uint8_t* retries;         // XOF role, direct mode (may be null): per sponge, how often the seed was mutated (ml_kem.c:237-242)
};

// PRF with eta = 3 squeezes 192 bytes: dwords 32 .. RATE/4-1 of the first block are still unwritten, then one more
// permutation supplies dwords RATE/4 .. 47.
// `out_of_lane()` yields the lane's output row; it is called again after the permutation so that a caller can hand in a
// recomputation (from lane_id_fresh() and wave-uniform values) instead of keeping a 64-bit pointer alive across it.
// `rate` (136 or 168, wave-uniform) is a run-time value so that there is ONE permutation in the code path.
template <class OutFn>
__device__ __forceinline__ void prf_eta3_tail(KeccakState& s, unsigned rate, OutFn out_of_lane, bool mine) {
    const unsigned nw = rate / 4;
    if (mine) {
        uint32_t* out = out_of_lane();
#define MLKEM_PT(W) if (W < nw) out[W] = keccak_word<W>(s);
        MLKEM_PT(32) MLKEM_PT(33) MLKEM_PT(34) MLKEM_PT(35) MLKEM_PT(36) MLKEM_PT(37) MLKEM_PT(38) MLKEM_PT(39) MLKEM_PT(40) MLKEM_PT(41)
#undef MLKEM_PT
    }
    keccak_f1600(s);
    if (mine) {
        uint32_t* out = out_of_lane() + nw;
#define MLKEM_PT(W) if (nw + W < 48u) out[W] = keccak_word<W>(s);
        MLKEM_PT(0) MLKEM_PT(1) MLKEM_PT(2) MLKEM_PT(3) MLKEM_PT(4) MLKEM_PT(5) MLKEM_PT(6) MLKEM_PT(7)
        MLKEM_PT(8) MLKEM_PT(9) MLKEM_PT(10) MLKEM_PT(11) MLKEM_PT(12) MLKEM_PT(13)
#undef MLKEM_PT
    }
}

// flush every lane-ring chunk that has become complete; ring rows live in `ring` (this wave's slice).
// `g` is this lane's sponge index (lanes need not hold consecutive sponges), `valid` = lane owns a real sponge.
__device__ __forceinline__ void ring_flush(const int16_t* ring, uint16_t* A, size_t g, bool valid, int cnt, int& flushed) {
    const int l = lane_id();
    const int has = (cnt - flushed >= 64) && valid;
    if (__ballot(has) == 0) return;
    const int grp = l >> 3, sub = l & 7;
#pragma unroll 1
    for (int step = 0; step < 8; step++) {
        const int p = step * 8 + grp;                       // lane (= polynomial of this wave) being flushed
        const int p_has = __shfl(has, p);
        const int p_flushed = __shfl(flushed, p);
        const size_t gp = (size_t)(uint32_t)__shfl((int)(uint32_t)g, p);   // sponge index of lane p (< 2^32)
        if (p_has) {
            const int16_t* src = ring + p * RING_STRIDE + (p_flushed & (RING - 1)) + sub * 8;
            uint4 v = *reinterpret_cast<const uint4*>(src);
            *reinterpret_cast<uint4*>(A + gp * 256 + p_flushed + sub * 8) = v;
        }
    }
    if (has) flushed += 64;
}

// The reference squeezes 280 triples and gives up when the triple that would be number 279 comes up (ml_kem.c:221-229): the
// usable triples are 0 .. SAMPLE_CAP - 1 = 277, i.e. four full blocks of 56 and the first 54 of the fifth; a polynomial still
// short after them restarts from the mutated seed (ml_kem.c:237-242).  The acceptance bound QB and the cap CAP are template
// parameters of every sampler so that the CPU tier can run these branches (a fifth block is otherwise reached with
// probability ~ e^-40 per sponge, the retry with ~ e^-220): tests/test_emulated_kernels.py instantiates the wave emulator with
// lower values; the product instantiates (KQ, SAMPLE_CAP) only.
constexpr int SAMPLE_CAP = 278;
// triples of the fifth block's group G (four triples per three dwords) that may be used under cap CAP
constexpr int sample_nt5(int cap, int g) { return cap - 224 - 4 * g >= 4 ? 4 : (cap - 224 - 4 * g <= 0 ? 0 : cap - 224 - 4 * g); }

// one candidate of SampleNTT (ml_kem.c:211-219)
#define MLKEM_CAND(d)                                                      \
    {                                                                      \
        const bool ok = ((d) < (unsigned)QB) && (cnt < 256);               \
        if (ok) myring[cnt & (RING - 1)] = (int16_t)(d);                   \
        cnt += ok ? 1 : 0;                                                 \
    }
// four 3-byte groups out of three state dwords; NT limits how many of the four are consumed
#define MLKEM_GROUP4(W0, NT)                                                                      \
    {                                                                                             \
        const uint32_t w0 = keccak_word<W0>(s), w1 = keccak_word<W0 + 1>(s), w2 = keccak_word<W0 + 2>(s); \
        const uint32_t v0 = w0 & 0xFFFFFFu, v1 = (w0 >> 24) | ((w1 & 0xFFFFu) << 8);              \
        const uint32_t v2 = (w1 >> 16) | ((w2 & 0xFFu) << 16), v3 = w2 >> 8;                      \
        if (NT > 0) { MLKEM_CAND(v0 & 0xFFFu) MLKEM_CAND(v0 >> 12) }                              \
        if (NT > 1) { MLKEM_CAND(v1 & 0xFFFu) MLKEM_CAND(v1 >> 12) }                              \
        if (NT > 2) { MLKEM_CAND(v2 & 0xFFFu) MLKEM_CAND(v2 >> 12) }                              \
        if (NT > 3) { MLKEM_CAND(v3 & 0xFFFu) MLKEM_CAND(v3 >> 12) }                              \
    }
// group G of a block: all four triples, except in the fifth block (blk == 4) where the cap may cut it short
#define MLKEM_GROUP4_CAP(W0, G)                                                                   \
    if constexpr (sample_nt5(CAP, G) == 4) { MLKEM_GROUP4(W0, 4) }                                \
    else { if (blk == 4) MLKEM_GROUP4(W0, sample_nt5(CAP, G)) else MLKEM_GROUP4(W0, 4) }

template <int QB = KQ, int CAP = SAMPLE_CAP>
__global__ void __launch_bounds__(WAVE) k_sample(SampleArgs a) {
    __shared__ __attribute__((aligned(16))) int16_t ring[WAVE * RING_STRIDE];
    const int l = lane_id();
    KeccakState s;
    if (blockIdx.x < a.xof_blocks) {
        // ---------------- XOF role: SampleNTT (general form: up to 5 blocks + the reference's retry) ----------------
        // direct mode: one pass, sponge = block*64 + lane.  list mode: sponges come from the leftover list written by
        // k_sample_main; the grid strides over the list (its length is only known on the device).
        const size_t limit = a.list_mode ? (size_t)a.leftover[0] : a.n_xof;
        const size_t stride = a.list_mode ? (size_t)gridDim.x * WAVE : ~(size_t)0 / 2;
      for (size_t base = (size_t)blockIdx.x * WAVE; base < limit; base += stride) {
        const size_t slot = base + l;
        const bool valid = slot < limit;
        const size_t slot_c = valid ? slot : limit - 1;
        const size_t g = a.list_mode ? (size_t)a.leftover[2 + slot_c] : slot_c;
        const size_t gc = g;
        uint32_t seed[8];
        unsigned i0, i1;
        if (a.K) {
            const size_t kk = (size_t)(a.K * a.K), item = gc / kk;
            const unsigned e = (unsigned)(gc - item * kk), ra = e / (unsigned)a.K, cb = e - ra * (unsigned)a.K;
            load32(a.rho, a.rho_stride, item, seed);
            i0 = a.transpose ? ra : cb;
            i1 = a.transpose ? cb : ra;
        } else {
            const uint8_t* sp = a.rho + gc * a.rho_stride;
#pragma unroll
            for (int i = 0; i < 8; i++)
                seed[i] = (uint32_t)sp[4 * i] | ((uint32_t)sp[4 * i + 1] << 8) | ((uint32_t)sp[4 * i + 2] << 16) |
                          ((uint32_t)sp[4 * i + 3] << 24);
            i0 = sp[32];
            i1 = sp[33];
        }
        int16_t* myring = ring + l * RING_STRIDE;
        unsigned nretry = 0;
        int cnt = 0, flushed = 0;
        bool fresh = true;      // (re)initialise the sponge: first pass, or the reference's seed-mutation retry
        int blk = 0;            // squeeze blocks consumed since the last (re)initialisation
        while (true) {
            if (fresh) {
                keccak_zero(s);
                MLKEM_SET_WORDS8(s, 0, seed)
                keccak_xor_byte<32>(s, i0 & 0xFFu);
                keccak_xor_byte<33>(s, i1 & 0xFFu);
                keccak_xor_byte<34>(s, 0x1F);
                keccak_xor_byte<167>(s, 0x80);
                cnt = 0; flushed = 0; blk = 0; fresh = false;
            }
            keccak_f1600(s);
            // first half of the block: triples 0..27  (dwords 0..20)
            MLKEM_GROUP4_CAP(0, 0) MLKEM_GROUP4_CAP(3, 1) MLKEM_GROUP4_CAP(6, 2) MLKEM_GROUP4_CAP(9, 3)
            MLKEM_GROUP4_CAP(12, 4) MLKEM_GROUP4_CAP(15, 5) MLKEM_GROUP4_CAP(18, 6)
            wave_lds_fence();
            ring_flush(ring, a.A, g, valid, cnt, flushed);
            wave_lds_fence();
            // second half: triples 28..55 (dwords 21..41); in the 5th block the reference never uses
            // triples 278, 279 (ml_kem.c:223-227: the 279th triple only trips the iteration limit)
            MLKEM_GROUP4_CAP(21, 7) MLKEM_GROUP4_CAP(24, 8) MLKEM_GROUP4_CAP(27, 9) MLKEM_GROUP4_CAP(30, 10)
            MLKEM_GROUP4_CAP(33, 11) MLKEM_GROUP4_CAP(36, 12) MLKEM_GROUP4_CAP(39, 13)
            wave_lds_fence();
            ring_flush(ring, a.A, g, valid, cnt, flushed);
            wave_lds_fence();
            blk++;
            if (blk == 5 && cnt < 256) {   // ml_kem.c:237-242: B[32]++, B[33]++ and start over
                i0 = (i0 + 1) & 0xFFu;
                i1 = (i1 + 1) & 0xFFu;
                fresh = true;
                nretry++;
            }
            if (__ballot(cnt < 256) == 0) break;
        }
        if (a.retries && valid && !a.list_mode) a.retries[g] = (uint8_t)(nretry < 255u ? nretry : 255u);
        wave_lds_fence();
      }
    } else {
        // ---------------- PRF role ----------------
        const size_t g = (size_t)(blockIdx.x - a.xof_blocks) * WAVE + l;
        const size_t gc = g < a.n_prf ? g : a.n_prf - 1;
        uint32_t seed[8];
        unsigned ctr, eta;
        if (a.per_item) {
            const size_t item = gc / (size_t)a.per_item;
            ctr = (unsigned)(gc - item * (size_t)a.per_item);
            load32(a.r, 32, item, seed);
            eta = (int)ctr < a.n_eta1 ? (unsigned)a.eta1 : 2u;
        } else {
            const uint8_t* sp = a.r + gc * 33;
#pragma unroll
            for (int i = 0; i < 8; i++)
                seed[i] = (uint32_t)sp[4 * i] | ((uint32_t)sp[4 * i + 1] << 8) | ((uint32_t)sp[4 * i + 2] << 16) |
                          ((uint32_t)sp[4 * i + 3] << 24);
            ctr = sp[32];
            eta = (unsigned)a.eta1;
        }
        keccak_zero(s);
        MLKEM_SET_WORDS8(s, 0, seed)
        keccak_xor_byte<32>(s, ctr & 0xFFu);
        keccak_xor_byte<33>(s, 0x1F);
        if (a.prf_rate == 136) keccak_xor_byte<135>(s, 0x80);   // SHAKE256 (FIPS 203 mode)
        else keccak_xor_byte<167>(s, 0x80);                     // SHAKE128 (the reference: ml_kem.c:508)
        keccak_f1600(s);
        uint32_t* out = reinterpret_cast<uint32_t*>(a.prf + gc * a.prf_stride);
        if (g < a.n_prf) {
#define MLKEM_OW(W) out[W] = keccak_word<W>(s);
            MLKEM_OW(0) MLKEM_OW(1) MLKEM_OW(2) MLKEM_OW(3) MLKEM_OW(4) MLKEM_OW(5) MLKEM_OW(6) MLKEM_OW(7)
            MLKEM_OW(8) MLKEM_OW(9) MLKEM_OW(10) MLKEM_OW(11) MLKEM_OW(12) MLKEM_OW(13) MLKEM_OW(14) MLKEM_OW(15)
            MLKEM_OW(16) MLKEM_OW(17) MLKEM_OW(18) MLKEM_OW(19) MLKEM_OW(20) MLKEM_OW(21) MLKEM_OW(22) MLKEM_OW(23)
            MLKEM_OW(24) MLKEM_OW(25) MLKEM_OW(26) MLKEM_OW(27) MLKEM_OW(28) MLKEM_OW(29) MLKEM_OW(30) MLKEM_OW(31)
        }
#undef MLKEM_OW
        if (__ballot(eta == 3) != 0) {   // eta = 3 needs 192 bytes: the rest of this block + the head of the next
            const bool mine = g < a.n_prf && eta == 3;
            auto out_fn = [out]() { return out; };
            prf_eta3_tail(s, a.prf_rate == 136 ? 136u : 168u, out_fn, mine);
        }
    }
}

}   // namespace mlkem
