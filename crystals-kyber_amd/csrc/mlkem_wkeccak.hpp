// mlkem_wkeccak.hpp — one sponge per WAVEFRONT: the Keccak of small calls (the regime of the ml_kem.h drop-in API).
//
// The lane-sliced kernels (mlkem_kernels.hpp: one sponge per SIMD lane) are built for throughput; a call of a few hundred items
// does not fill the GPU with them and waits for the dependency chain of ONE lane: 9-10 permutations of 4320 instructions,
// ~10.5 us each for a lone wave.  Here the 25 Keccak lanes of ONE state are spread over the SIMD lanes of a wave and a round is
// ~30 instructions with a single LDS round trip:
//   layout  Keccak lane (x, y) lives in SIMD lane 8 y + 1 + x (its "primary"); SIMD lanes 8 y and 8 y + 6 hold COPIES of
//           (4, y) and (0, y), so that x - 1 and x + 1 are always the neighbouring SIMD lane of the same 8-lane group.
//           40 of the 64 lanes carry state; the others are kept at zero.  One 64-bit Keccak lane = two VGPRs.
//   theta   column parities without LDS: rows y = 2r, 2r + 1 share DPP row r (v_xor with row_ror:8), the three occupied rows
//           are folded with v_permlane32_swap / v_permlane16_swap (the lo and hi words travel together: the first swap pairs
//           lo of the lower half-wave with hi of the upper one), and C[x - 1], rot(C[x + 1]) arrive by DPP row_shr:1 / row_shl:1
//           folded into the two v_xor that apply D;
//   rho     a per-lane 64-bit rotate (two v_alignbit with the lane's own shift, halves pre-swapped for offsets >= 32);
//   pi+chi  lane (x', y') fetches the rotated lanes that pi maps to (x', y'), (x' + 1, y'), (x' + 2, y') with six ds_bpermute_b32
//           -- the round's only LDS trip; copies fetch what their primary fetches, which is what keeps them valid;
//   iota    the lanes holding (0, 0), from a 200-byte table of the wave in LDS.
// 24 VALU + 7 DS instructions per round, 2.6 us per permutation of a lone wave.  Round 3's form (one sponge per half-wave, 18
// ds_bpermute per round in three dependent groups) ran 4.8; tools/keccak_wave_ubench.hip measures all forms
// (profiles/r04_keccak_wave_ubench.txt).
// Message bytes map 1:1: Keccak lane i = x + 5 y owns bytes [8 i, 8 i + 8) of every rate block, so absorbing is one 8-byte
// load per lane and block.  All message lengths on this path are multiples of 8 (ek, z || c, m || h) or fit one block.
// Follows sha3.c:15-216 (permutation), :257-330 (sponge) of the reference.
#pragma once
#include "mlkem_kernels.hpp"

namespace mlkem {

// ---- cross-lane primitives (device: DPP / permlane / LDS crossbar; emulator: shuffles) -------------------------------------
#ifdef MLKEM_EMU
__device__ __forceinline__ uint32_t wk_fetch(uint32_t byte_addr, uint32_t v) { return (uint32_t)__shfl((int)v, (int)(byte_addr >> 2)); }
// value of the lane 8 further round the 16-lane row
__device__ __forceinline__ uint32_t wk_row_ror8(uint32_t v) { return (uint32_t)__shfl((int)v, (int)((threadIdx.x & 63) ^ 8)); }
// value of lane - 1 / lane + 1 of the same row (0 at the row's end)
__device__ __forceinline__ uint32_t wk_row_shr1(uint32_t v) {
    const int l = (int)(threadIdx.x & 63);
    const uint32_t r = (uint32_t)__shfl((int)v, (l & 15) ? l - 1 : l);
    return (l & 15) ? r : 0u;
}
__device__ __forceinline__ uint32_t wk_row_shl1(uint32_t v) {
    const int l = (int)(threadIdx.x & 63);
    const uint32_t r = (uint32_t)__shfl((int)v, (l & 15) != 15 ? l + 1 : l);
    return (l & 15) != 15 ? r : 0u;
}
// v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b ; v_permlane16_swap: odd rows of a <-> even rows of b
__device__ __forceinline__ void wk_swap32(uint32_t& a, uint32_t& b) {
    const int l = (int)(threadIdx.x & 63);
    const uint32_t a_other = (uint32_t)__shfl((int)a, l ^ 32), b_other = (uint32_t)__shfl((int)b, l ^ 32);
    if (l >= 32) a = b_other; else b = a_other;
}
__device__ __forceinline__ void wk_swap16(uint32_t& a, uint32_t& b) {
    const int l = (int)(threadIdx.x & 63);
    const uint32_t a_other = (uint32_t)__shfl((int)a, l ^ 16), b_other = (uint32_t)__shfl((int)b, l ^ 16);
    if (l & 16) a = b_other; else b = a_other;
}
#else
__device__ __forceinline__ uint32_t wk_fetch(uint32_t byte_addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_addr, (int)v); }
__device__ __forceinline__ uint32_t wk_row_ror8(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t wk_row_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t wk_row_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true); }
__device__ __forceinline__ void wk_swap32(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void wk_swap16(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
#endif

__constant__ uint8_t WK_RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

// ---- the lane map ----------------------------------------------------------------------------------------------------------
// SIMD lane of the primary of Keccak lane i = x + 5 y
__device__ __forceinline__ unsigned wk_lane_of(unsigned i) { return 8u * (i / 5u) + 1u + i % 5u; }
// Keccak lane index x + 5 y this SIMD lane holds (primaries and copies), -1 for the idle lanes
__device__ __forceinline__ int wk_index() {
    const unsigned l = (unsigned)lane_id(), g = l >> 3, p = l & 7u;
    if (g >= 5u || p == 7u) return -1;
    return (int)(5u * g + (p + 4u) % 5u);
}
__device__ __forceinline__ bool wk_primary() {
    const unsigned l = (unsigned)lane_id(), p = l & 7u;
    return (l >> 3) < 5u && p >= 1u && p <= 5u;
}

// per-lane constants of the round
constexpr int WK_RC_ENTRIES = 25;   // the wave's table in LDS: the 24 round constants and a zero entry
struct WkLane {
    uint32_t src[3];      // byte addresses (lane * 4) of the pi sources of (x, y), (x + 1, y), (x + 2, y)
    uint32_t sh;          // v_alignbit shift of rho
    uint32_t keep;        // all-ones in the lanes that carry state
    bool swp;             // rho offset >= 32 (or 0): halves swapped before the shift
    bool is00;            // holds Keccak lane (0, 0): iota
    const uint2* rc;      // iota: the lanes holding (0, 0) read entry `round` of the wave's table, all others its zero entry --
                          // one ds_read_b64 per round instead of two scalar loads + two selects
};
// `rc_table`: WK_RC_ENTRIES entries of LDS owned by the calling wave (filled here)
__device__ __forceinline__ void wk_lane_init(WkLane& c, uint2* rc_table) {
    const unsigned l = (unsigned)lane_id();
    if (l < 24u) {
        uint2 v;
        v.x = KECCAK_RC[2 * l]; v.y = KECCAK_RC[2 * l + 1];
        rc_table[l] = v;
    } else if (l == 24u) {
        uint2 v;
        v.x = 0; v.y = 0;
        rc_table[24] = v;
    }
    wave_lds_fence();
    c.rc = rc_table;
    const int i = wk_index();
    const int ii = i < 0 ? 0 : i;                       // idle lanes run lane 0's constants; `keep` zeroes what they compute
    const int x = ii % 5, y = ii / 5;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int xd = (x + k) % 5;                    // destination (xd, y) <- source (xd + 3 y, xd)   (B[y', 2x'+3y'] = A[x', y'])
        c.src[k] = 4u * wk_lane_of((unsigned)((xd + 3 * y) % 5 + 5 * xd));
    }
    const unsigned r = WK_RHO[ii];
    c.swp = r >= 32 || r == 0;                         // r = 0: swapped halves and a zero shift return the lane unchanged
    c.sh = r == 0 ? 0u : (32u - (r & 31u)) & 31u;
    c.keep = i < 0 ? 0u : 0xFFFFFFFFu;
    c.is00 = i == 0;
}

struct WkState { uint32_t lo, hi; };

// make a freshly loaded state well-formed: idle lanes zero (copies were loaded like their primaries by the caller)
__device__ __forceinline__ void wk_canon(WkState& a, const WkLane& c) {
    const uint32_t from = 4u * wk_lane_of((unsigned)(wk_index() < 0 ? 0 : wk_index()));
    a.lo = wk_fetch(from, a.lo) & c.keep;
    a.hi = wk_fetch(from, a.hi) & c.keep;
}

__device__ __forceinline__ void wk_permute(WkState& a, const WkLane& c) {
    const uint2* rcp = c.is00 ? c.rc : c.rc + 24;
    // four rounds per loop iteration: the loop's scalar bookkeeping and the taken branch cost a lone wave ~25 cycles per round
    // (2.84 -> 2.59 us per permutation; unrolled 8 / 24 times: 2.56 / 2.52, not worth the code)
#pragma unroll 4
    for (int round = 0; round < 24; round++) {
        const uint2 rc = *rcp;                  // issued first: long since there when iota needs it
        rcp += c.is00 ? 1 : 0;
        // theta: column parity C[x] in every lane of the wave
        uint32_t tl = a.lo ^ wk_row_ror8(a.lo), th = a.hi ^ wk_row_ror8(a.hi);   // rows y = 2r, 2r + 1 folded, both 8-lane groups
        wk_swap32(tl, th);                  // tl = [lo rows 0 1 | hi rows 0 1], th = [lo rows 2 3 | hi rows 2 3]
        uint32_t u = tl ^ th, v = u;        // lower half-wave: lo folded over the halves ; upper: hi
        wk_swap16(u, v);
        uint32_t cl = u ^ v, ch = cl;       // lanes 0..31: C.lo, lanes 32..63: C.hi
        wk_swap32(cl, ch);                  // cl = C.lo everywhere, ch = C.hi everywhere
        // D[x] = C[x - 1] ^ rotl(C[x + 1], 1), applied: two v_xor_b32_dpp per word (left to itself the compiler moves one
        // operand with v_mov_b32_dpp first: three instructions per word)
        const uint32_t rl = __builtin_amdgcn_alignbit(cl, ch, 31), rh = __builtin_amdgcn_alignbit(ch, cl, 31);
        uint32_t al, ah;
#ifdef MLKEM_EMU
        al = (a.lo ^ wk_row_shr1(cl)) ^ wk_row_shl1(rl);
        ah = (a.hi ^ wk_row_shr1(ch)) ^ wk_row_shl1(rh);
#else
        // (DPP hazards: cl / ch were written at least two instructions -- the two v_alignbit above, whose results this block
        // needs -- before the first DPP read, rl / rh at least two before theirs)
        asm volatile("v_xor_b32_dpp %0, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_xor_b32_dpp %1, %3, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_xor_b32_dpp %0, %6, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_xor_b32_dpp %1, %7, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "=&v"(al), "=&v"(ah) : "v"(cl), "v"(ch), "v"(a.lo), "v"(a.hi), "v"(rl), "v"(rh));
#endif
        // rho (this lane's offset)
        const uint32_t a0 = c.swp ? ah : al, a1 = c.swp ? al : ah;
        const uint32_t bl = __builtin_amdgcn_alignbit(a0, a1, c.sh), bh = __builtin_amdgcn_alignbit(a1, a0, c.sh);
        // pi + chi
        const uint32_t b0l = wk_fetch(c.src[0], bl), b1l = wk_fetch(c.src[1], bl), b2l = wk_fetch(c.src[2], bl);
        const uint32_t b0h = wk_fetch(c.src[0], bh), b1h = wk_fetch(c.src[1], bh), b2h = wk_fetch(c.src[2], bh);
        // iota in the lanes holding (0, 0), zero in the idle lanes: (chi & keep) ^ rc as one v_bitop3
        a.lo = __builtin_amdgcn_bitop3_b32(MLKEM_CHI(b0l, b1l, b2l), c.keep, rc.x, 0x6A);
        a.hi = __builtin_amdgcn_bitop3_b32(MLKEM_CHI(b0h, b1h, b2h), c.keep, rc.y, 0x6A);
    }
}

// Absorb a message of two segments (seg0 then seg1; len0, total multiples of 8) with the pad10*1 padding and the domain suffix
// SUFFIX, one wave, RATE bytes per block; the state is left after the last permutation (squeeze = read Keccak lanes 0..).
template <int RATE, unsigned SUFFIX>
__device__ __forceinline__ void wk_absorb(WkState& a, const WkLane& c, const uint8_t* p0, unsigned len0, const uint8_t* p1, unsigned total) {
    constexpr int NQ = RATE / 8;
    const int i = wk_index();
    a.lo = 0; a.hi = 0;
    const unsigned nblocks = total / RATE + 1;
    uint2 v;
    // the first block's load is issued before the loop, every next one before the permutation it hides behind
    auto fetch = [&](unsigned blk) {
        const unsigned o = blk * RATE + 8u * (unsigned)(i < 0 ? 0 : i);
        uint2 w;
        w.x = 0; w.y = 0;
        if (i >= 0 && i < NQ && o < total) w = *reinterpret_cast<const uint2*>(o < len0 ? p0 + o : p1 + (o - len0));
        return w;
    };
    v = fetch(0);
#pragma unroll 1
    for (unsigned blk = 0; blk < nblocks; blk++) {
        a.lo ^= v.x; a.hi ^= v.y;
        // The next block's load is issued HERE and lands during the permutation.  It must come after the XOR above (the old value
        // dead, so that the load writes the same registers and nothing waits for it before the loop) and must not sit in a branch:
        // the first build fetched under `if (not the last block)`, the compiler resolved the loop-carried value with a register
        // copy right after the load, and every block waited for its HBM round trip before permuting (0.35 us per block).  The last
        // iteration re-reads its own block (harmless).
        sched_fence();
        const bool last = blk + 1 == nblocks;
        v = fetch(last ? blk : blk + 1);
        if (last) {                 // pad10*1: suffix byte at message position `total`, 0x80 at the block's last byte
            const unsigned pos = total - blk * RATE;           // < RATE, multiple of 8
            if (i == (int)(pos / 8)) a.lo ^= SUFFIX;
            if (i == NQ - 1) a.hi ^= 0x80000000u;
        }
        wk_permute(a, c);
    }
}

// G(x || y) (SHA3-512 of 64 bytes, one permutation; ml_kem.c:559-572): Keccak lanes 0..3 = x (`xv` in the lanes of index 0..3),
// lanes 4..7 = y = lanes 0..3 of the state `a` on entry (a digest).  On exit lanes 0..3 = first half of the output, 4..7 = second.
__device__ __forceinline__ void wk_G_of_x_and_digest(WkState& a, const WkLane& c, uint2 xv) {
    const int i = wk_index();
    const uint32_t from = 4u * wk_lane_of((unsigned)(i >= 4 && i < 8 ? i - 4 : 0));
    const uint32_t hl = wk_fetch(from, a.lo), hh = wk_fetch(from, a.hi);
    a.lo = (i >= 0 && i < 4) ? xv.x : ((i >= 4 && i < 8) ? hl : 0u);
    a.hi = (i >= 0 && i < 4) ? xv.y : ((i >= 4 && i < 8) ? hh : 0u);
    if (i == 8) { a.lo = 0x06u; a.hi = 0x80000000u; }
    wk_permute(a, c);
}

// ------------------------------------------------------------------------------------------------
// k_hash_encaps_w — Encaps_internal's hashing (ml_kem.c:1108-1124), one item per wave: h = H(ek); (K, r) = G(m || h)
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(WAVE) k_hash_encaps_w(size_t n, const uint8_t* __restrict__ ek, const uint8_t* __restrict__ m,
                                                        uint8_t* __restrict__ Kout, uint8_t* __restrict__ r_ws) {
    constexpr unsigned EK = 384 * K + 32;
    const size_t item = blockIdx.x;
    if (item >= n) return;
    __shared__ uint2 rc_table[WK_RC_ENTRIES];
    const int i = wk_index();
    WkLane c;
    wk_lane_init(c, rc_table);
    WkState a;
    wk_absorb<136, 0x06>(a, c, ek + item * EK, EK, ek, EK);
    uint2 mv;
    mv.x = 0; mv.y = 0;
    if (i >= 0 && i < 4) mv = reinterpret_cast<const uint2*>(m + item * 32)[i];
    wk_G_of_x_and_digest(a, c, mv);
    uint2 o;
    o.x = a.lo; o.y = a.hi;
    if (wk_primary() && i < 4) reinterpret_cast<uint2*>(Kout + item * 32)[i] = o;
    else if (wk_primary() && i < 8) reinterpret_cast<uint2*>(r_ws + item * 32)[i - 4] = o;
}

// ------------------------------------------------------------------------------------------------
// k_hash_keygen_fin_w — KeyGen_internal's dk tail (ml_kem.c:1065-1077), one item per wave: dk[768k+32 ..] = H(ek) ; dk[768k+64 ..] = z
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(WAVE) k_hash_keygen_fin_w(size_t n, const uint8_t* __restrict__ ek, const uint8_t* __restrict__ z,
                                                            uint8_t* __restrict__ dk) {
    constexpr unsigned EK = 384 * K + 32, DK = 768 * K + 96;
    const size_t item = blockIdx.x;
    if (item >= n) return;
    __shared__ uint2 rc_table[WK_RC_ENTRIES];
    const int i = wk_index();
    WkLane c;
    wk_lane_init(c, rc_table);
    WkState a;
    wk_absorb<136, 0x06>(a, c, ek + item * EK, EK, ek, EK);
    uint2 o;
    o.x = a.lo; o.y = a.hi;
    if (i >= 4 && i < 8) o = reinterpret_cast<const uint2*>(z + item * 32)[i - 4];
    if (wk_primary() && i < 8) reinterpret_cast<uint2*>(dk + item * DK + 768 * K + 32)[i] = o;   // h (lanes 0..3) then z (lanes 4..7): 64 contiguous bytes
}

// ------------------------------------------------------------------------------------------------
// k_hash_decaps_w — KEM_Decaps' hash check and Decaps_internal's hashing (ml_kem.c:1336-1350, :1187-1202), one sponge per
// wave: blocks [0, n) run J(z || c) and then G(m' || h); with HASH_CHECK blocks [n, 2 n) run H(dk.ek) against dk.h.
// Arguments as k_hash_decaps.
// ------------------------------------------------------------------------------------------------
template <int K, int CLEN, bool HASH_CHECK, int JRATE>
__global__ void __launch_bounds__(WAVE) k_hash_decaps_w(size_t n, const uint8_t* __restrict__ dk, const uint8_t* __restrict__ c,
                                                        const uint8_t* __restrict__ m_ws, uint8_t* __restrict__ Kp_ws,
                                                        uint8_t* __restrict__ r_ws, uint8_t* __restrict__ Kbar_ws,
                                                        int32_t* __restrict__ status, size_t dk_stride) {
    constexpr unsigned EK = 384 * K + 32;
    const bool check_role = HASH_CHECK && blockIdx.x >= n;
    const size_t item = check_role ? blockIdx.x - n : blockIdx.x;
    if (item >= n) return;
    const int i = wk_index();
    const bool prim = wk_primary();
    const uint8_t* my_dk = dk + item * dk_stride;
    __shared__ uint2 rc_table[WK_RC_ENTRIES];
    WkLane cst;
    wk_lane_init(cst, rc_table);
    WkState a;
    if (check_role) {
        wk_absorb<136, 0x06>(a, cst, my_dk + 384 * K, EK, my_dk, EK);
        uint2 h;
        h.x = 0; h.y = 0;
        if (prim && i < 4) h = reinterpret_cast<const uint2*>(my_dk + 768 * K + 32)[i];
        const bool differ = prim && i < 4 && (h.x != a.lo || h.y != a.hi);
        const bool bad = __ballot(differ) != 0;
        if (lane_id() == 0 && status) status[item] = bad ? -5 : 0;
        return;
    }
    // Kbar = J(z || c)
    wk_absorb<JRATE, 0x1F>(a, cst, my_dk + 768 * K + 64, 32, c + item * (size_t)CLEN, 32 + CLEN);
    uint2 o;
    o.x = a.lo; o.y = a.hi;
    if (prim && i < 4) reinterpret_cast<uint2*>(Kbar_ws + item * 32)[i] = o;
    // (K', r') = G(m' || h)
    uint2 v;
    v.x = 0; v.y = 0;
    if (i >= 0 && i < 4) v = reinterpret_cast<const uint2*>(m_ws + item * 32)[i];
    else if (i >= 4 && i < 8) v = reinterpret_cast<const uint2*>(my_dk + 768 * K + 32)[i - 4];
    a.lo = v.x; a.hi = v.y;
    if (i == 8) { a.lo = 0x06u; a.hi = 0x80000000u; }
    wk_permute(a, cst);
    o.x = a.lo; o.y = a.hi;
    if (prim && i < 4) reinterpret_cast<uint2*>(Kp_ws + item * 32)[i] = o;
    else if (prim && i < 8) reinterpret_cast<uint2*>(r_ws + item * 32)[i - 4] = o;
    if (!HASH_CHECK && lane_id() == 0 && status) status[item] = 0;
}

}   // namespace mlkem
