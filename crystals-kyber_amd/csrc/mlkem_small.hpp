// mlkem_small.hpp — ONE WORKGROUP PER ITEM: ML-KEM KeyGen / Encaps / Decaps for small calls, each a single launch.
//
// A call of one to a few hundred items -- every call of the ml_kem.h drop-in API -- cannot fill the GPU; its time is the
// length of its dependency chain: H(ek) (9 sequential permutations at k = 3) -> G -> PRF -> Encrypt for Encaps.  The batch path
// spreads that chain over four kernels on two streams (lane-sliced or wave-wide sponges, sampler, arithmetic) and pays a launch
// gap between each.  Here the item's whole operation is one workgroup of 4, 8 or 12 wavefronts that run the item's
// independent sponges SIDE BY SIDE, each on the one-sponge-per-wave Keccak of mlkem_wkeccak.hpp (2.6 us per permutation for a
// lone wave instead of 8.8 lane-sliced), and hand the sampled matrix and the PRF rows to the
// arithmetic (keygen2_body / decrypt4_body as the batch kernels run them; encrypt1_body, where the two half-waves share the item's
// k + 1 output rows) through LDS (SmallHand):
//   k_encaps_small  wave 0: h = H(ek), (K, r) = G(m || h), r_ready | every wave, when it has nothing else to do: jobs = the k^2
//                   SampleNTT(rho || j || i), then the 2k + 1 PRF(r, n) (these wait for r_ready) | wave 0, when all jobs are
//                   done: K-PKE.Encrypt                                                       (ml_kem.c:1093-1130, :776-936)
//   k_decaps_small  wave 0: m' = K-PKE.Decrypt, (K', r') = G(m' || h), r_ready | wave 1: Kbar = J(z || c), kbar_ready | jobs as
//                   above | wave 0: c' = Encrypt, then -- waiting for kbar_ready only here -- K = c == c' ? K' : Kbar
//                   (status = H(dk.ek) == dk.h ? 0 : -5 in a workgroup of its own: nothing else waits for that chain)
//                                                                                             (ml_kem.c:1310-1359, :1136-1225)
//   k_keygen_small  wave 0: (rho, sigma) = G(d || k) ; barrier ; jobs: A-hat and the 2k PRF rows ; barrier ;
//                   wave 0: K-PKE.KeyGen, then H(ek) and the dk tail                          (ml_kem.c:1034-1084, :651-769)
// Hand-overs are counters in LDS (flag_signal / flag_wait, mlkem_device.hpp), not workgroup barriers: Encaps and Decaps have one
// barrier, right after the counters are zeroed.  A barrier makes every wave wait for the slowest; here only the consumer of a value
// waits for its producer, so that J (7 permutations at k = 3, needed by the last instruction of Decaps) delays nothing.
// SampleNTT on a wave: the squeezed block (168 bytes in 21 SIMD lanes) goes through 176 bytes of LDS so that SIMD lane t
// reads triple t (ml_kem.c:208-219: 3 bytes -> two 12-bit candidates); the accepted candidates of the 56 lanes are put in order
// with two ballots and a prefix count and stored straight to the polynomial.  The triple limit and the seed-mutation retry of
// the reference (ml_kem.c:221-242) are reproduced: template parameters QB (acceptance bound) and CAP (usable triples) exist so
// that the CPU tier can drive this branch (tests/test_emulated_kernels.py); the product instantiates (KQ, SAMPLE_CAP) only.
#pragma once
#include "mlkem_wkeccak.hpp"
#include "mlkem_kpke2.hpp"
#include "mlkem_kpke4.hpp"

namespace mlkem {

// Waves per workgroup (template parameter NW of the three kernels).  8 is the latency form: a lone item's nine SampleNTT sponges
// run on seven waves while wave 0 walks H(ek).  4 is the dense form for calls that fill the chip: half the registers per item,
// so twice the items per CU, and the item's sponges are dealt out by a counter in LDS (take_job) so that the waves of the serial
// roles join the sampling when they are through -- four waves then finish about when wave 0's chain does.
constexpr int SMALL_WAVES = 8, SMALL_WAVES_DENSE = 4, SMALL_WAVES_WIDE = 12;

// The next job number of the workgroup's counter (0, 1, 2, ... in the order the waves ask), the same value in all lanes.
// (readfirstlane: the compiler must KNOW the number is wave-uniform, or it wraps the job loop -- ballots, shuffles and spin-waits
// inside -- into EXEC-mask bookkeeping for a divergence that never happens)
__device__ __forceinline__ unsigned take_job(uint32_t* counter) {
    unsigned j = 0;
    if (lane_id_fresh() == 0) j = atomicAdd(counter, 1u);   // _fresh: see flag_signal
#ifdef MLKEM_EMU
    return (unsigned)__shfl((int)j, 0);
#else
    return (unsigned)__builtin_amdgcn_readfirstlane((int)j);
#endif
}

#ifdef MLKEM_EMU
__device__ __forceinline__ void block_barrier() { emu::block_barrier(); }
__device__ __forceinline__ void wave_global_fence() { emu::wave_barrier(); }
#else
__device__ __forceinline__ void block_barrier() { __syncthreads(); }
// a wave's own global stores (made by other lanes) before its later loads of the same bytes
__device__ __forceinline__ void wave_global_fence() { __threadfence_block(); }
#endif

constexpr int XOF_LDS_WORDS = 44;   // 168 squeezed bytes + the dword the last triple's unaligned read touches

// ------------------------------------------------------------------------------------------------
// SampleNTT (ml_kem.c:189-245), one sponge on the calling wave: seed = rho[32] || i0 || i1 -> poly[256] (uint16, < QB)
// ------------------------------------------------------------------------------------------------
// `seed`: bytes [8 i, 8 i + 8) of rho in the SIMD lanes of Keccak index i < 4 (copies included), anything elsewhere
template <int QB = KQ, int CAP = SAMPLE_CAP>
// Returns the number of seed mutations (retries) it took: 0 for every real SHAKE stream.
__device__ __forceinline__ unsigned wk_sample_ntt_seed(const WkLane& c, uint2 seed, unsigned i0, unsigned i1, uint16_t* poly, uint32_t* sq) {
    static_assert(CAP > 224 && CAP <= 280, "the limit falls into the fifth squeeze block");
    const int i = wk_index();
    const bool prim = wk_primary();
    const unsigned l = (unsigned)lane_id();
    if (!(i >= 0 && i < 4)) { seed.x = 0; seed.y = 0; }
    const unsigned bo = 3u * l, w = bo >> 2, sh = 8u * (bo & 3u);   // triple l of a block: bytes 3 l .. 3 l + 2
    const unsigned long long below = (1ull << l) - 1ull;
    unsigned retries = 0;
    for (;;) {
        WkState a;
        a.lo = seed.x; a.hi = seed.y;
        if (i == 4) a.lo = (i0 & 0xFFu) | ((i1 & 0xFFu) << 8) | (0x1Fu << 16);   // bytes 32, 33 and the SHAKE suffix
        if (i == 20) a.hi = 0x80000000u;                                            // pad10*1 ends at byte 167
        unsigned cnt = 0;
#pragma unroll 1
        for (int blk = 0; blk < 5 && cnt < 256u; blk++) {
            wk_permute(a, c);
            if (prim && i < 21) { sq[2 * i] = a.lo; sq[2 * i + 1] = a.hi; }
            wave_lds_fence();
            uint32_t v = 0;
            if (l < 56u) v = __builtin_amdgcn_alignbit(sq[w + 1], sq[w], sh) & 0xFFFFFFu;
            wave_lds_fence();
            // the fifth block: the triple that would be number CAP + 1 only trips the reference's limit (ml_kem.c:223-227)
            const unsigned ntr = blk == 4 ? (unsigned)(CAP - 224) : 56u;
            const unsigned d1 = v & 0xFFFu, d2 = v >> 12;
            const bool ok1 = l < ntr && d1 < (unsigned)QB, ok2 = l < ntr && d2 < (unsigned)QB;
            const unsigned long long m1 = __ballot(ok1), m2 = __ballot(ok2);
            const unsigned pos1 = cnt + (unsigned)__builtin_popcountll(m1 & below) + (unsigned)__builtin_popcountll(m2 & below);
            const unsigned pos2 = pos1 + (ok1 ? 1u : 0u);
            if (ok1 && pos1 < 256u) poly[pos1] = (uint16_t)d1;
            if (ok2 && pos2 < 256u) poly[pos2] = (uint16_t)d2;
            cnt += (unsigned)__builtin_popcountll(m1) + (unsigned)__builtin_popcountll(m2);
        }
        if (cnt >= 256u) break;
        i0 = (i0 + 1u) & 0xFFu;    // ml_kem.c:237-242: B[32]++, B[33]++ and start over
        i1 = (i1 + 1u) & 0xFFu;
        retries++;
    }
    return retries;
}

// rho: 32 bytes, 8-byte aligned
template <int QB = KQ, int CAP = SAMPLE_CAP>
__device__ __forceinline__ void wk_sample_ntt(const WkLane& c, const uint8_t* rho, unsigned i0, unsigned i1, uint16_t* poly, uint32_t* sq) {
    const int i = wk_index();
    uint2 seed;
    seed.x = 0; seed.y = 0;
    if (i >= 0 && i < 4) seed = reinterpret_cast<const uint2*>(rho)[i];
    wk_sample_ntt_seed<QB, CAP>(c, seed, i0, i1, poly, sq);
}
// k_sample_ntt_w — stand-alone SampleNTT over explicit 34-byte seeds (any alignment), one sponge per wave
// retries (may be null): per seed, how often the reference would have bumped B[32], B[33] (ml_kem.c:237-242), saturating at 255
template <int QB = KQ, int CAP = SAMPLE_CAP>
__global__ void __launch_bounds__(WAVE) k_sample_ntt_w(size_t n, const uint8_t* __restrict__ seeds34, uint16_t* __restrict__ out, uint8_t* __restrict__ retries) {
    __shared__ uint32_t sq[XOF_LDS_WORDS];
    __shared__ uint2 rc_table[WK_RC_ENTRIES];
    const size_t g = blockIdx.x;
    if (g >= n) return;
    const uint8_t* sp = seeds34 + g * 34;
    const int i = wk_index();
    uint2 seed;
    seed.x = 0; seed.y = 0;
    if (i >= 0 && i < 4) {
        const uint8_t* b = sp + 8 * i;
        seed.x = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        seed.y = (uint32_t)b[4] | ((uint32_t)b[5] << 8) | ((uint32_t)b[6] << 16) | ((uint32_t)b[7] << 24);
    }
    WkLane c;
    wk_lane_init(c, rc_table);
    const unsigned r = wk_sample_ntt_seed<QB, CAP>(c, seed, sp[32], sp[33], out + g * 256, sq);
    if (retries && lane_id() == 0) retries[g] = (uint8_t)(r < 255u ? r : 255u);
}

// k_sponge_raw_w — the bare Keccak sponge (sha3.c:257-317) over n already-padded messages, one sponge per wave, for ANY rate of
// 1..199 bytes (the reference's Sponge takes any capacity; sha3_b(..., c, ...) with (1600 - c) % 8 == 0 lands here when the rate is
// not one of the five SHA-3 / SHAKE rates, and every small call does, whatever its rate).  Message i at msg + i * nblocks * rate,
// output i at out + i * out_stride; bytes are moved one by one (rates and strides need not be aligned to anything).
__global__ void __launch_bounds__(WAVE) k_sponge_raw_w(size_t n, const uint8_t* __restrict__ msg, unsigned rate, unsigned nblocks,
                                                       uint8_t* __restrict__ out, unsigned outlen, size_t out_stride) {
    __shared__ uint2 rc_table[WK_RC_ENTRIES];
    const size_t g = blockIdx.x;
    if (g >= n) return;
    WkLane c;
    wk_lane_init(c, rc_table);
    const int i = wk_index();
    const unsigned base = 8u * (unsigned)(i < 0 ? 0 : i);   // first byte of the rate block this lane's Keccak lane covers
    const uint8_t* m = msg + g * (size_t)nblocks * rate;
    WkState a;
    a.lo = 0; a.hi = 0;
    for (unsigned blk = 0; blk < nblocks; blk++) {
        if (i >= 0) {
#pragma unroll
            for (unsigned b = 0; b < 8; b++)
                if (base + b < rate) {
                    const uint32_t v = m[(size_t)blk * rate + base + b];
                    if (b < 4) a.lo ^= v << (8 * b);
                    else a.hi ^= v << (8 * (b - 4));
                }
        }
        wk_permute(a, c);
    }
    uint8_t* o = out + g * out_stride;
    for (unsigned done = 0; done < outlen; done += rate) {
        if (done) wk_permute(a, c);
        const unsigned take = outlen - done < rate ? outlen - done : rate;
        if (wk_primary()) {
#pragma unroll
            for (unsigned b = 0; b < 8; b++)
                if (base + b < take) o[done + base + b] = (uint8_t)((b < 4 ? a.lo >> (8 * b) : a.hi >> (8 * (b - 4))) & 0xFFu);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// PRF (ml_kem.c:496-515): SHAKE(r[32] || ctr) -> 64 eta bytes at `out`; rate 168 = SHAKE128 like the reference (F2), 136 = SHAKE256
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wk_prf(const WkLane& c, const uint8_t* r32, unsigned ctr, unsigned eta, unsigned rate, uint8_t* out) {
    const int i = wk_index();
    const bool prim = wk_primary();
    WkState a;
    a.lo = 0; a.hi = 0;
    if (i >= 0 && i < 4) {
        const uint2 v = reinterpret_cast<const uint2*>(r32)[i];
        a.lo = v.x; a.hi = v.y;
    }
    if (i == 4) a.lo = (ctr & 0xFFu) | (0x1Fu << 8);
    const int nq = (int)(rate / 8u), want = (int)(8u * eta);   // qwords per block, qwords wanted (16 or 24)
    if (i == nq - 1) a.hi ^= 0x80000000u;
    wk_permute(a, c);
    uint2 o;
    o.x = a.lo; o.y = a.hi;
    if (prim && i < (want < nq ? want : nq)) reinterpret_cast<uint2*>(out)[i] = o;
    if (want > nq) {                                            // eta = 3: the head of the second block
        wk_permute(a, c);
        o.x = a.lo; o.y = a.hi;
        if (prim && i < want - nq) reinterpret_cast<uint2*>(out)[nq + i] = o;
    }
}

// What the waves of an item's workgroup hand to each other: the sampled matrix, the PRF rows and the 32-byte values.  It lives in
// LDS -- a producer's ds_write, the workgroup barrier, the consumer's ds_read -- instead of the call's scratch in HBM, where every
// hand-over cost a store acknowledgement before the barrier and an L2 round trip after it (3-4 us per operation).
template <int K, int ETA1>
struct __attribute__((aligned(16))) SmallHand {
    uint16_t A[K * K * 256];
    uint8_t prf[(2 * K + 1) * (ETA1 == 3 ? 192 : 128)];
    uint8_t r[32], m[32], Kp[32], Kbar[32], rho[32];
};

// The workgroup's hand-over counters (flag_signal / flag_wait, mlkem_device.hpp).  Only wave 0 runs an operation from end to end;
// the others take jobs (take_job) and leave, so after the first barrier nothing makes every wave wait for the slowest one.
struct SmallSync {
    uint32_t next_job;     // next job to hand out
    uint32_t jobs_done;    // jobs finished (their outputs are in SmallHand)
    uint32_t r_ready;      // 1: the 32-byte PRF key (r / r' / sigma) is in SmallHand
    uint32_t kbar_ready;   // 1: Decaps' K-bar = J(z || c) is in SmallHand
};

// The job loop every wave of an item's workgroup runs when it has nothing else to do: jobs 0 .. k^2 - 1 are the matrix entries
// (entry s = SampleNTT(rho || s / k || s % k) at A + 256 s with `transpose`, rho || s % k || s / k without), the next `rows` jobs
// the PRF rows (counter n = row; eta1 for n < k, ETA_REST after), which need the key behind sy.r_ready.
template <int K, int ETA1, int ETA_REST = 2>
__device__ __forceinline__ void small_jobs(SmallSync& sy, const WkLane& cst, const uint8_t* rho, bool transpose, uint16_t* A, const uint8_t* key,
                                           uint8_t* prf, int rows, unsigned prf_rate, uint32_t* sq) {
    constexpr unsigned PS = (ETA1 == 3) ? 192 : 128;
    for (unsigned j; (j = take_job(&sy.next_job)) < (unsigned)(K * K + rows);) {
        if (j < (unsigned)(K * K)) {
            wk_sample_ntt(cst, rho, transpose ? j / K : j % K, transpose ? j % K : j / K, A + j * 256, sq);
        } else {
            const unsigned row = j - K * K;
            flag_wait(&sy.r_ready, 1u);
            wk_prf(cst, key, row, row < (unsigned)K ? (unsigned)ETA1 : (unsigned)ETA_REST, prf_rate, prf + row * PS);
        }
        flag_signal(&sy.jobs_done);
    }
}

// H(msg) for a message of `len` bytes (multiple of 8) at `msg`: digest in Keccak lanes 0..3 of `a`
__device__ __forceinline__ void wk_H(WkState& a, const WkLane& c, const uint8_t* msg, unsigned len) {
    wk_absorb<136, 0x06>(a, c, msg, len, msg, len);
}

// ------------------------------------------------------------------------------------------------
// k_encaps_small — ML-KEM.Encaps_internal (ml_kem.c:1093-1130), one workgroup per item
// ------------------------------------------------------------------------------------------------
template <int K, int ETA1, int DU, int DV, int NW>
__global__ void __launch_bounds__(WAVE * NW)
k_encaps_small(size_t n, const uint8_t* __restrict__ ek, const uint8_t* __restrict__ m, uint8_t* __restrict__ c, uint8_t* __restrict__ Kout,
               int32_t* __restrict__ mod_status, int prf_rate) {
    __shared__ K2Lds<K + 1> xl;
    __shared__ SmallHand<K, ETA1> hand;
    __shared__ uint32_t sq[NW][XOF_LDS_WORDS];
    __shared__ uint2 rc_tables[NW][WK_RC_ENTRIES];
    __shared__ SmallSync sy;
    constexpr unsigned EK = 384 * K + 32, CLEN = 32 * (DU * K + DV);
    const int wv = wave_id();
    const size_t item = blockIdx.x;
    if (item >= n) return;
    const uint8_t* my_ek = ek + item * EK;
    uint16_t* my_A = hand.A;
    uint8_t *my_prf = hand.prf, *my_r = hand.r;
    WkLane cst;
    wk_lane_init(cst, rc_tables[wave_id()]);
    if (threadIdx.x == 0) { sy.next_job = 0; sy.jobs_done = 0; sy.r_ready = 0; sy.kbar_ready = 0; }
    block_barrier();
    if (wv == 0) {                                   // h = H(ek) ; (K, r) = G(m || h)
        const int i = wk_index();
        WkState a;
        wk_H(a, cst, my_ek, EK);
        uint2 mv;
        mv.x = 0; mv.y = 0;
        if (i >= 0 && i < 4) mv = reinterpret_cast<const uint2*>(m + item * 32)[i];
        wk_G_of_x_and_digest(a, cst, mv);
        uint2 o;
        o.x = a.lo; o.y = a.hi;
        if (wk_primary() && i < 4) reinterpret_cast<uint2*>(Kout + item * 32)[i] = o;
        else if (wk_primary() && i < 8) reinterpret_cast<uint2*>(my_r)[i - 4] = o;
        flag_signal(&sy.r_ready);
    }
    // jobs 0 .. k^2 - 1: A-hat^T[a][b] = SampleNTT(rho || a || b) (ml_kem.c:817-823) ; then the 2k + 1 PRF rows, which wait for r
    small_jobs<K, ETA1>(sy, cst, my_ek + 384 * K, /*transpose=*/true, my_A, my_r, my_prf, 2 * K + 1, (unsigned)prf_rate, sq[wv]);
    if (wv != 0) return;
    flag_wait(&sy.jobs_done, (uint32_t)(K * K + 2 * K + 1));
    encrypt1_body<K, ETA1, DU, DV, false>(xl.xch, my_ek, m + item * 32, my_A, my_prf, c + item * CLEN, nullptr, nullptr, nullptr, nullptr,
                                          mod_status ? mod_status + item : nullptr);
}

// ------------------------------------------------------------------------------------------------
// k_decaps_small — KEM_Decaps / Decaps_internal (ml_kem.c:1310-1359, :1136-1225), one workgroup per item
// ------------------------------------------------------------------------------------------------
template <int K, int ETA1, int DU, int DV, bool HASH_CHECK, int JRATE, int NW>
__global__ void __launch_bounds__(WAVE * NW)   // (forced to 128 VGPRs -- two eight-wave workgroups per CU -- k = 3 spills 88 bytes: not done)
k_decaps_small(size_t n, const uint8_t* __restrict__ dk, const uint8_t* __restrict__ c, uint8_t* __restrict__ Kout, int32_t* __restrict__ status,
               int prf_rate) {
    __shared__ K2Lds<K + 1> xl;
    __shared__ SmallHand<K, ETA1> hand;
    __shared__ uint32_t sq[NW][XOF_LDS_WORDS];
    __shared__ uint2 rc_tables[NW][WK_RC_ENTRIES];
    __shared__ SmallSync sy;
    constexpr unsigned EK = 384 * K + 32, DK = 768 * K + 96, CLEN = 32 * (DU * K + DV);
    const int wv = wave_id();
    // KEM_Decaps' check of the stored H(ek) (9 permutations at k = 3: the longest chain of the operation, and nothing but the
    // status depends on it) runs in workgroups of its own, one live wave each.  They are blocks [0, n): dispatched first, they
    // shrink to one wave at once, and the eight-wave workgroup of the item still fits the same CU beside it (a check workgroup
    // dispatched AFTER the main ones would wait for a whole CU's worth of registers, i.e. for a main workgroup to finish)
    const bool check_block = HASH_CHECK && blockIdx.x < n;
    const size_t item = HASH_CHECK && !check_block ? blockIdx.x - n : blockIdx.x;
    if (item >= n || (check_block && wv != 0)) return;
    const uint8_t* my_dk = dk + item * DK;
    const uint8_t* my_c = c + item * CLEN;
    uint16_t* my_A = hand.A;
    uint8_t *my_prf = hand.prf, *my_r = hand.r, *my_m = hand.m, *my_Kp = hand.Kp, *my_Kbar = hand.Kbar;
    WkLane cst;
    wk_lane_init(cst, rc_tables[wave_id()]);
    const int i = wk_index();
    const bool prim = wk_primary();
    if (check_block) {
        WkState a;
        wk_H(a, cst, my_dk + 384 * K, EK);
        uint2 h;
        h.x = 0; h.y = 0;
        if (prim && i < 4) h = reinterpret_cast<const uint2*>(my_dk + 768 * K + 32)[i];
        const bool differ = prim && i < 4 && (h.x != a.lo || h.y != a.hi);
        const bool bad = __ballot(differ) != 0;
        if (lane_id() == 0 && status) status[item] = bad ? -5 : 0;
        return;
    }
    if (threadIdx.x == 0) { sy.next_job = 0; sy.jobs_done = 0; sy.r_ready = 0; sy.kbar_ready = 0; }
    block_barrier();
    if (wv == 0) {                                   // m' = K-PKE.Decrypt(dk_pke, c) ; (K', r') = G(m' || h)
        decrypt4_body<K, DU, DV>(0, 1, my_dk, (size_t)DK, my_c, my_m);
        wave_global_fence();
        uint2 v;
        v.x = 0; v.y = 0;
        if (i >= 0 && i < 4) v = reinterpret_cast<const uint2*>(my_m)[i];
        else if (i >= 4 && i < 8) v = reinterpret_cast<const uint2*>(my_dk + 768 * K + 32)[i - 4];
        WkState a;
        a.lo = v.x; a.hi = v.y;
        if (i == 8) { a.lo = 0x06u; a.hi = 0x80000000u; }
        wk_permute(a, cst);
        uint2 o;
        o.x = a.lo; o.y = a.hi;
        if (prim && i < 4) reinterpret_cast<uint2*>(my_Kp)[i] = o;
        else if (prim && i < 8) reinterpret_cast<uint2*>(my_r)[i - 4] = o;
        if (!HASH_CHECK && lane_id() == 0 && status) status[item] = 0;
        flag_signal(&sy.r_ready);
    } else if (wv == 1) {                            // Kbar = J(z || c): only the final select waits for it
        WkState a;
        wk_absorb<JRATE, 0x1F>(a, cst, my_dk + 768 * K + 64, 32, my_c, 32 + CLEN);
        uint2 o;
        o.x = a.lo; o.y = a.hi;
        if (prim && i < 4) reinterpret_cast<uint2*>(my_Kbar)[i] = o;
        flag_signal(&sy.kbar_ready);
    }
    // jobs: A-hat^T of the re-encryption (rho sits in dk.ek), then the 2k + 1 PRF rows, which wait for r'
    small_jobs<K, ETA1>(sy, cst, my_dk + 768 * K, /*transpose=*/true, my_A, my_r, my_prf, 2 * K + 1, (unsigned)prf_rate, sq[wv]);
    if (wv != 0) return;
    flag_wait(&sy.jobs_done, (uint32_t)(K * K + 2 * K + 1));
    encrypt1_body<K, ETA1, DU, DV, true>(xl.xch, my_dk + 384 * K, my_m, my_A, my_prf, nullptr, my_c, my_Kp, my_Kbar, Kout + item * 32, nullptr,
                                         &sy.kbar_ready);
}

// ------------------------------------------------------------------------------------------------
// k_keygen_small — ML-KEM.KeyGen_internal (ml_kem.c:1034-1084) ; KEM_DK = false: K-PKE.KeyGen alone (dk = the 384k bytes of s-hat)
// ------------------------------------------------------------------------------------------------
template <int K, int ETA1, bool KEM_DK, int NW>
__global__ void __launch_bounds__(WAVE * NW)
k_keygen_small(size_t n, const uint8_t* __restrict__ d, const uint8_t* __restrict__ z, uint8_t* ek, uint8_t* dk, int prf_rate) {
    __shared__ K2Lds<K> xl;
    __shared__ SmallHand<K, ETA1> hand;          // prf: 2k rows of it ; r: sigma
    __shared__ uint32_t sq[NW][XOF_LDS_WORDS];
    __shared__ uint2 rc_tables[NW][WK_RC_ENTRIES];
    __shared__ SmallSync sy;
    constexpr unsigned EK = 384 * K + 32, DK = KEM_DK ? 768 * K + 96 : 384 * K;
    const int wv = wave_id();
    const size_t item = blockIdx.x;
    if (item >= n) return;
    uint8_t *my_ek = ek + item * EK, *my_dk = dk + item * DK;
    uint16_t* my_A = hand.A;
    uint8_t *my_prf = hand.prf, *my_rho = hand.rho, *my_sigma = hand.r;
    WkLane cst;
    wk_lane_init(cst, rc_tables[wave_id()]);
    const int i = wk_index();
    const bool prim = wk_primary();
    if (wv == 0) {                                   // (rho, sigma) = G(d || k)   (ml_kem.c:674-681)
        WkState a;
        a.lo = 0; a.hi = 0;
        if (i >= 0 && i < 4) {
            const uint2 v = reinterpret_cast<const uint2*>(d + item * 32)[i];
            a.lo = v.x; a.hi = v.y;
        }
        if (i == 4) a.lo = (unsigned)K | (0x06u << 8);
        if (i == 8) a.hi = 0x80000000u;
        wk_permute(a, cst);
        uint2 o;
        o.x = a.lo; o.y = a.hi;
        if (prim && i < 4) reinterpret_cast<uint2*>(my_rho)[i] = o;
        else if (prim && i < 8) reinterpret_cast<uint2*>(my_sigma)[i - 4] = o;
    }
    if (threadIdx.x == 0) { sy.next_job = 0; sy.jobs_done = 0; sy.r_ready = 1; sy.kbar_ready = 0; }   // sigma is ready behind the barrier
    block_barrier();
    // A-hat[a][b] = SampleNTT(rho || b || a) (ml_kem.c:686-693) and the 2k PRF rows (s: n = 0..k-1, e: n = k..2k-1, all eta1),
    // dealt out longest first
    small_jobs<K, ETA1, ETA1>(sy, cst, my_rho, /*transpose=*/false, my_A, my_sigma, my_prf, 2 * K, (unsigned)prf_rate, sq[wv]);
    block_barrier();
    if (wv == 0) {
        keygen2_body<K, ETA1, KEM_DK>(xl.xch, 0, 1, my_A, my_prf, my_rho, my_ek, my_dk);
        if constexpr (KEM_DK) {                      // dk tail: H(ek) || z   (ml_kem.c:1065-1077)
            wave_global_fence();
            WkState a;
            wk_H(a, cst, my_ek, EK);
            uint2 o;
            o.x = a.lo; o.y = a.hi;
            if (i >= 4 && i < 8) o = reinterpret_cast<const uint2*>(z + item * 32)[i - 4];
            if (prim && i < 8) reinterpret_cast<uint2*>(my_dk + 768 * K + 32)[i] = o;
        }
    }
}

}   // namespace mlkem
