// mlkem_wkeccak.hpp — one sponge per HALF-WAVE (two per wavefront): the hash kernels of small batches.
//
// The lane-sliced kernels (mlkem_kernels.hpp: one sponge per SIMD lane) are built for throughput; a call of a few thousand
// items does not fill the GPU with them and waits for the dependency chain of ONE lane: 9-10 permutations of 4320
// instructions, ~10.5 us each for a lone wave.  Here a wave carries TWO sponges, one per 32-lane half; Keccak lane (x, y) of a
// state lives in SIMD lane x + 5 y of its half (25 of 32 lanes; one 64-bit Keccak lane = two VGPRs per SIMD lane), a round is
// ~40 instructions, and theta / pi / chi fetch their operands from other lanes with ds_bpermute_b32 (18 per round, in three
// dependent groups):
//   theta   C[x] = xor of the column: four fetches from (x, y + k) ; D needs C[x - 1], C[x + 1]: two more ;
//   rho     a per-lane 64-bit rotate (two v_alignbit with the lane's own shift, halves pre-swapped for offsets >= 32) ;
//   pi+chi  lane (x', y') fetches the rotated lanes that pi maps to (x', y'), (x' + 1, y'), (x' + 2, y') and combines them ;
//   iota    lane 0.
// 5.3 us per permutation of a lone wave instead of 10.5: H(ek) -> G 0.107 -> 0.053 ms, the Decaps sponges 0.095 -> 0.047 ms
// at 64 items; faster up to 2048 items per call, slower from 4096 (profiles/r03_batch_sweep.txt), hence Workspace::wide_max.
// Message bytes map 1:1: lane L of a half owns bytes [8 L, 8 L + 8) of every rate block, so absorbing is one 8-byte load per
// lane and block.  All message lengths on this path are multiples of 8 (ek, z || c, m || h).  Follows sha3.c:15-216
// (permutation), :257-330 (sponge) of the reference.
#pragma once
#include "mlkem_kernels.hpp"

namespace mlkem {

#ifdef MLKEM_EMU
__device__ __forceinline__ uint32_t wk_fetch(uint32_t byte_addr, uint32_t v) { return (uint32_t)__shfl((int)v, (int)(byte_addr >> 2)); }
#else
__device__ __forceinline__ uint32_t wk_fetch(uint32_t byte_addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_addr, (int)v); }
#endif

__constant__ uint8_t WK_RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

// per-lane constants of the round: byte addresses (lane * 4) of the lanes to fetch from, the rho rotation
struct WkLane {
    uint32_t col[4];      // (x, y + k), k = 1..4
    uint32_t xm, xp;      // (x - 1, y), (x + 1, y)
    uint32_t src[3];      // pi sources of (x, y), (x + 1, y), (x + 2, y)
    uint32_t sh;          // v_alignbit shift of rho
    bool swp;             // rho offset >= 32 (or 0): halves swapped before the shift
    bool lane0;
};
__device__ __forceinline__ unsigned wk_lane() { return (unsigned)lane_id() & 31u; }   // lane within the sponge's half
__device__ __forceinline__ unsigned wk_half() { return (unsigned)lane_id() >> 5; }
__device__ __forceinline__ void wk_lane_init(WkLane& c) {
    const int L = wk_lane() < 25 ? (int)wk_lane() : 0;   // idle lanes mirror lane 0 (their values are never stored)
    const uint32_t base = 128u * wk_half();              // byte address of the half's lane 0
    const int x = L % 5, y = L / 5;
#pragma unroll
    for (int k = 0; k < 4; k++) c.col[k] = base + 4u * (uint32_t)(x + 5 * ((y + k + 1) % 5));
    c.xm = base + 4u * (uint32_t)((x + 4) % 5 + 5 * y);
    c.xp = base + 4u * (uint32_t)((x + 1) % 5 + 5 * y);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int xd = (x + k) % 5;                    // destination (xd, y) <- source (xd + 3 y, xd)   (B[y', 2x'+3y'] = A[x', y'])
        c.src[k] = base + 4u * (uint32_t)((xd + 3 * y) % 5 + 5 * xd);
    }
    const unsigned r = WK_RHO[L];
    c.swp = r >= 32 || r == 0;                         // r = 0: swapped halves and a zero shift return the lane unchanged
    c.sh = r == 0 ? 0u : (32u - (r & 31u)) & 31u;
    c.lane0 = wk_lane() == 0;
}

struct WkState { uint32_t lo, hi; };

__device__ __forceinline__ void wk_permute(WkState& a, const WkLane& c) {
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        // theta
        uint32_t tl[4], th[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { tl[k] = wk_fetch(c.col[k], a.lo); th[k] = wk_fetch(c.col[k], a.hi); }
        const uint32_t cl = MLKEM_XOR3(MLKEM_XOR3(a.lo, tl[0], tl[1]), tl[2], tl[3]);
        const uint32_t ch = MLKEM_XOR3(MLKEM_XOR3(a.hi, th[0], th[1]), th[2], th[3]);
        const uint32_t ml = wk_fetch(c.xm, cl), mh = wk_fetch(c.xm, ch), pl = wk_fetch(c.xp, cl), ph = wk_fetch(c.xp, ch);
        uint32_t rl, rh;
        rotl64<1>(pl, ph, rl, rh);
        const uint32_t al = MLKEM_XOR3(a.lo, ml, rl), ah = MLKEM_XOR3(a.hi, mh, rh);
        // rho (this lane's offset)
        const uint32_t a0 = c.swp ? ah : al, a1 = c.swp ? al : ah;
        const uint32_t bl = __builtin_amdgcn_alignbit(a0, a1, c.sh), bh = __builtin_amdgcn_alignbit(a1, a0, c.sh);
        // pi + chi
        const uint32_t b0l = wk_fetch(c.src[0], bl), b1l = wk_fetch(c.src[1], bl), b2l = wk_fetch(c.src[2], bl);
        const uint32_t b0h = wk_fetch(c.src[0], bh), b1h = wk_fetch(c.src[1], bh), b2h = wk_fetch(c.src[2], bh);
        a.lo = MLKEM_CHI(b0l, b1l, b2l);
        a.hi = MLKEM_CHI(b0h, b1h, b2h);
        // iota
        const uint32_t rcl = KECCAK_RC[2 * round], rch = KECCAK_RC[2 * round + 1];
        a.lo ^= c.lane0 ? rcl : 0u;
        a.hi ^= c.lane0 ? rch : 0u;
    }
}

// Absorb a message of two segments (seg0 then seg1; len0, total multiples of 8) with the pad10*1 padding and the domain suffix
// SUFFIX, one wave, RATE bytes per block; the state is left after the last permutation (squeeze = read lanes 0..).
template <int RATE, unsigned SUFFIX>
__device__ __forceinline__ void wk_absorb(WkState& a, const WkLane& c, const uint8_t* p0, unsigned len0, const uint8_t* p1, unsigned total) {
    constexpr unsigned NQ = RATE / 8;
    const unsigned L = wk_lane();
    a.lo = 0; a.hi = 0;
    const unsigned nblocks = total / RATE + 1;
    uint2 v;
    // the first block's load is issued before the loop, every next one before the permutation it hides behind
    auto fetch = [&](unsigned blk) {
        const unsigned o = blk * RATE + 8u * L;
        uint2 w;
        w.x = 0; w.y = 0;
        if (L < NQ && o < total) w = *reinterpret_cast<const uint2*>(o < len0 ? p0 + o : p1 + (o - len0));
        return w;
    };
    v = fetch(0);
#pragma unroll 1
    for (unsigned blk = 0; blk < nblocks; blk++) {
        a.lo ^= v.x; a.hi ^= v.y;
        if (blk + 1 == nblocks) {   // pad10*1: suffix byte at message position `total`, 0x80 at the block's last byte
            const unsigned pos = total - blk * RATE;           // < RATE, multiple of 8
            if (L == pos / 8) a.lo ^= SUFFIX;
            if (L == NQ - 1) a.hi ^= 0x80000000u;
        } else {
            v = fetch(blk + 1);
        }
        wk_permute(a, c);
    }
}

// ------------------------------------------------------------------------------------------------
// k_hash_encaps_w — Encaps_internal's hashing (ml_kem.c:1108-1124), two items per wave (one per half): h = H(ek); (K, r) = G(m || h)
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(WAVE) k_hash_encaps_w(size_t n, const uint8_t* __restrict__ ek, const uint8_t* __restrict__ m,
                                                        uint8_t* __restrict__ Kout, uint8_t* __restrict__ r_ws) {
    constexpr unsigned EK = 384 * K + 32;
    const size_t slot = 2 * (size_t)blockIdx.x + wk_half();
    const bool valid = slot < n;
    const size_t item = valid ? slot : n - 1;            // the odd half of the last wave repeats the last item and stores nothing
    const unsigned L = wk_lane();
    WkLane c;
    wk_lane_init(c);
    WkState a;
    wk_absorb<136, 0x06>(a, c, ek + item * EK, EK, ek, EK);
    // G(m || h): lanes 0..3 = m, lanes 4..7 = h (the digest sits in lanes 0..3), SHA3-512: rate 72, suffix at byte 64
    const uint32_t from = 128u * wk_half() + 4u * ((L + 28u) & 31u);   // lane L - 4 of the same half
    const uint32_t hl = wk_fetch(from, a.lo), hh = wk_fetch(from, a.hi);
    uint2 mv;
    mv.x = 0; mv.y = 0;
    if (L < 4) mv = reinterpret_cast<const uint2*>(m + item * 32)[L];
    a.lo = L < 4 ? mv.x : (L < 8 ? hl : 0u);
    a.hi = L < 4 ? mv.y : (L < 8 ? hh : 0u);
    if (L == 8) { a.lo = 0x06u; a.hi = 0x80000000u; }
    wk_permute(a, c);
    uint2 o;
    o.x = a.lo; o.y = a.hi;
    if (valid && L < 4) reinterpret_cast<uint2*>(Kout + item * 32)[L] = o;
    else if (valid && L < 8) reinterpret_cast<uint2*>(r_ws + item * 32)[L - 4] = o;
}

// ------------------------------------------------------------------------------------------------
// k_hash_keygen_fin_w — KeyGen_internal's dk tail (ml_kem.c:1065-1077), two items per wave: dk[768k+32 ..] = H(ek) ; dk[768k+64 ..] = z
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(WAVE) k_hash_keygen_fin_w(size_t n, const uint8_t* __restrict__ ek, const uint8_t* __restrict__ z,
                                                            uint8_t* __restrict__ dk) {
    constexpr unsigned EK = 384 * K + 32, DK = 768 * K + 96;
    const size_t slot = 2 * (size_t)blockIdx.x + wk_half();
    const bool valid = slot < n;
    const size_t item = valid ? slot : n - 1;
    const unsigned L = wk_lane();
    WkLane c;
    wk_lane_init(c);
    WkState a;
    wk_absorb<136, 0x06>(a, c, ek + item * EK, EK, ek, EK);
    uint2 o;
    o.x = a.lo; o.y = a.hi;
    if (L >= 4 && L < 8) o = reinterpret_cast<const uint2*>(z + item * 32)[L - 4];
    if (valid && L < 8) reinterpret_cast<uint2*>(dk + item * DK + 768 * K + 32)[L] = o;   // h (lanes 0..3) then z (lanes 4..7): 64 contiguous bytes
}

// ------------------------------------------------------------------------------------------------
// k_hash_decaps_w — KEM_Decaps' hash check and Decaps_internal's hashing (ml_kem.c:1336-1350, :1187-1202), one sponge per
// half-wave: blocks [0, nb) run J(z || c) and then G(m' || h), nb = ceil(n / 2); with HASH_CHECK blocks [nb, 2 nb) run H(dk.ek) against dk.h.
// Arguments as k_hash_decaps.
// ------------------------------------------------------------------------------------------------
template <int K, int CLEN, bool HASH_CHECK, int JRATE>
__global__ void __launch_bounds__(WAVE) k_hash_decaps_w(size_t n, const uint8_t* __restrict__ dk, const uint8_t* __restrict__ c,
                                                        const uint8_t* __restrict__ m_ws, uint8_t* __restrict__ Kp_ws,
                                                        uint8_t* __restrict__ r_ws, uint8_t* __restrict__ Kbar_ws,
                                                        int32_t* __restrict__ status, size_t dk_stride) {
    constexpr unsigned EK = 384 * K + 32;
    const size_t nb = (n + 1) / 2;                        // waves per role: two items per wave
    const bool check_role = HASH_CHECK && blockIdx.x >= nb;
    const size_t slot = 2 * (size_t)(check_role ? blockIdx.x - nb : blockIdx.x) + wk_half();
    const bool valid = slot < n;
    const size_t item = valid ? slot : n - 1;
    const unsigned L = wk_lane();
    const uint8_t* my_dk = dk + item * dk_stride;
    WkLane cst;
    wk_lane_init(cst);
    WkState a;
    if (check_role) {
        wk_absorb<136, 0x06>(a, cst, my_dk + 384 * K, EK, my_dk, EK);
        uint2 h;
        h.x = 0; h.y = 0;
        if (L < 4) h = reinterpret_cast<const uint2*>(my_dk + 768 * K + 32)[L];
        const bool differ = L < 4 && (h.x != a.lo || h.y != a.hi);
        const unsigned long long bal = __ballot(differ);
        const bool bad = ((bal >> (32 * wk_half())) & 0xFFFFFFFFull) != 0;
        if (valid && L == 0 && status) status[item] = bad ? -5 : 0;
        return;
    }
    // Kbar = J(z || c)
    wk_absorb<JRATE, 0x1F>(a, cst, my_dk + 768 * K + 64, 32, c + item * (size_t)CLEN, 32 + CLEN);
    uint2 o;
    o.x = a.lo; o.y = a.hi;
    if (valid && L < 4) reinterpret_cast<uint2*>(Kbar_ws + item * 32)[L] = o;
    // (K', r') = G(m' || h)
    uint2 v;
    v.x = 0; v.y = 0;
    if (L < 4) v = reinterpret_cast<const uint2*>(m_ws + item * 32)[L];
    else if (L < 8) v = reinterpret_cast<const uint2*>(my_dk + 768 * K + 32)[L - 4];
    a.lo = v.x; a.hi = v.y;
    if (L == 8) { a.lo = 0x06u; a.hi = 0x80000000u; }
    wk_permute(a, cst);
    o.x = a.lo; o.y = a.hi;
    if (valid && L < 4) reinterpret_cast<uint2*>(Kp_ws + item * 32)[L] = o;
    else if (valid && L < 8) reinterpret_cast<uint2*>(r_ws + item * 32)[L - 4] = o;
    if (!HASH_CHECK && valid && L == 0 && status) status[item] = 0;
}

}   // namespace mlkem
