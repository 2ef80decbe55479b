// mlkem_sampler.hpp — the fast path of SampleNTT / PRF sampling (ml_kem.c:189-245, :496-515).
//
// k_sample_main : every XOF lane squeezes exactly THREE SHAKE128 blocks (336 candidates, mean 273 accepted);
//                 the ~0.8 % of sponges that still miss coefficients are handed over with their Keccak state and
//                 finished by k_sample_resume (one more permutation) in a second, tiny launch.  Without this split
//                 ~40 % of the waves would run a 4th permutation for the sake of one or two lanes.
//                 Blocks 0 and 1 can never complete a polynomial (2 x 112 < 256), so their rejection loop
//                 carries no `count < 256` test.
//                 PRF lanes: one permutation (two for eta = 3), raw bytes out.
// Earlier forms of the staging (per-lane LDS rings of 32 / 64 / 128 coefficients, cooperative flushes) are in the git
// history and their measurements in profiles/r01_*, profiles/r02_sampler_experiments.txt.
#pragma once
#include "mlkem_kernels.hpp"

namespace mlkem {

// ---- linear staging buffer + EXEC-masked acceptance --------------------------------------------------------------------
// A lane's `pos` IS the LDS byte address of its next coefficient in a linear 144-byte buffer, and a candidate is accepted
// by masking: v_cmpx_gt_u32 leaves EXEC = lanes whose candidate is < q, the ds_write_b16 and the v_add_u32 pos += 2 run under that mask, s_mov_b64 restores EXEC
// (4.3 units: one VOPC + one plain VALU; the DS and scalar instructions issue on their own ports).  Twice per block the
// complete 32-byte pieces go to HBM and the < 32-byte remainder moves to the front of the buffer (at most 30 + 112 bytes
// are ever staged: 56 candidates between flush points).
constexpr int LIN_STRIDE = 144;   // bytes per lane: 9 x 16, rows 16-byte aligned, 36-dword skew
#ifdef MLKEM_EMU
#define MLKEM_LIN_ACCEPT(lds0, pos, d) { if ((d) < (uint32_t)QB) { *reinterpret_cast<int16_t*>((lds0) + (pos)) = (int16_t)(d); (pos) += 2u; } }
#define MLKEM_LIN_ACCEPT_LIM(lds0, pos, d, lim) { if ((d) < (uint32_t)QB && (pos) < (lim)) { *reinterpret_cast<int16_t*>((lds0) + (pos)) = (int16_t)(d); (pos) += 2u; } }
#else
// all 64 lanes are active in the XOF role (sponges beyond n_xof are clamped duplicates), so EXEC is restored to -1
#define MLKEM_LIN_ACCEPT(lds0, pos, d)                                              \
    asm volatile("v_cmpx_gt_u32 vcc, 0xd01, %1\n\t"                                 \
                 "ds_write_b16 %0, %1\n\t"                                          \
                 "v_add_u32 %0, 2, %0\n\t"                                          \
                 "s_mov_b64 exec, -1"                                               \
                 : "+v"(pos) : "v"(d) : "vcc", "memory")
#define MLKEM_LIN_ACCEPT_LIM(lds0, pos, d, lim)                                     \
    asm volatile("v_cmpx_gt_u32 vcc, 0xd01, %1\n\t"                                 \
                 "v_cmpx_gt_u32 vcc, %2, %0\n\t"                                    \
                 "ds_write_b16 %0, %1\n\t"                                          \
                 "v_add_u32 %0, 2, %0\n\t"                                          \
                 "s_mov_b64 exec, -1"                                               \
                 : "+v"(pos) : "v"(d), "v"(lim) : "vcc", "memory")
#endif
// the 8 candidates of a 12-byte group of the squeezed block (ml_kem.c:208-219), FIRST..LAST-1 of its 4 triples
#define MLKEM_LG(ACC, W0, FIRST, LAST)                                                                         \
    {                                                                                                           \
        const uint32_t w0 = keccak_word<W0>(s), w1 = keccak_word<W0 + 1>(s), w2 = keccak_word<W0 + 2>(s);       \
        if (FIRST <= 0 && LAST > 0) { ACC(w0 & 0xFFFu) ACC((w0 >> 12) & 0xFFFu) }                               \
        if (FIRST <= 1 && LAST > 1) { ACC(__builtin_amdgcn_alignbit(w1, w0, 24) & 0xFFFu) ACC((w1 >> 4) & 0xFFFu) } \
        if (FIRST <= 2 && LAST > 2) { ACC((w1 >> 16) & 0xFFFu) ACC(__builtin_amdgcn_alignbit(w2, w1, 28) & 0xFFFu) } \
        if (FIRST <= 3 && LAST > 3) { ACC((w2 >> 8) & 0xFFFu) ACC(w2 >> 20) }                                   \
    }
#define MLKEM_LBLOCK(ACC, FLUSH)                                                                               \
    MLKEM_LG(ACC, 0, 0, 4) MLKEM_LG(ACC, 3, 0, 4) MLKEM_LG(ACC, 6, 0, 4) MLKEM_LG(ACC, 9, 0, 4)                 \
    MLKEM_LG(ACC, 12, 0, 4) MLKEM_LG(ACC, 15, 0, 4) MLKEM_LG(ACC, 18, 0, 4)                                     \
    FLUSH                                                                                                       \
    MLKEM_LG(ACC, 21, 0, 4) MLKEM_LG(ACC, 24, 0, 4) MLKEM_LG(ACC, 27, 0, 4) MLKEM_LG(ACC, 30, 0, 4)             \
    MLKEM_LG(ACC, 33, 0, 4) MLKEM_LG(ACC, 36, 0, 4) MLKEM_LG(ACC, 39, 0, 4)                                     \
    FLUSH

// complete 32-byte pieces of the lane's staging buffer -> HBM (the polynomial's next bytes), remainder to the front.
// 32 bytes = one HBM sector: flushing 16-byte pieces was measured to write 12 % more than the algorithmic bytes
// (WRITE_SIZE 1767 MB against 1443 MB per launch: partial-sector writes), with 32-byte pieces the remainder (< 32 bytes)
// plus the 112 bytes of 56 candidates still fit the 144-byte row.
// `buf` = the lane's buffer, `pos0` = its LDS byte address; pos / lim are LDS addresses, flushed counts bytes in HBM.
__device__ __forceinline__ void lin_flush(char* buf, uint32_t pos0, uint32_t& pos, uint32_t& lim, uint32_t& flushed, char* dst_poly, bool valid) {
    const uint32_t n32 = (pos - pos0) >> 5;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
        if (i < n32 && valid) {
            const uint4 a = *reinterpret_cast<const uint4*>(buf + 32u * i), b = *reinterpret_cast<const uint4*>(buf + 32u * i + 16u);
            stream_store16(dst_poly + flushed + 32u * i, a);
            stream_store16(dst_poly + flushed + 32u * i + 16u, b);
        }
    if (n32) {   // remainder (< 32 bytes; < 16 when four pieces left, so that the row's 144 bytes are never exceeded)
        const uint4 a = *reinterpret_cast<const uint4*>(buf + 32u * n32);
        *reinterpret_cast<uint4*>(buf) = a;
        if (n32 < 4) {
            const uint4 b = *reinterpret_cast<const uint4*>(buf + 32u * n32 + 16u);
            *reinterpret_cast<uint4*>(buf + 16u) = b;
        }
    }
    pos -= 32u * n32;
    lim -= 32u * n32;
    flushed += 32u * n32;
}

template <int QB = KQ>
__global__ void __launch_bounds__(WAVE, 4) k_sample_main(SampleArgs a) {   // 9 KB of LDS per wave: 4 waves per SIMD
#ifndef MLKEM_EMU
    static_assert(QB == 0xd01, "the acceptance bound is a literal of the inline assembly (MLKEM_LIN_ACCEPT)");
#endif
    static_assert(WAVE * LIN_STRIDE >= 32 * 33 * 4, "the PRF role stages 32 rows x 33 dwords in the same buffer");
    __shared__ __attribute__((aligned(16))) char stage[WAVE * LIN_STRIDE];
    const int l = lane_id();
    KeccakState s;
    if (blockIdx.x < a.xof_blocks) {
        // ---------------- XOF role: three blocks, no data-dependent control flow ----------------
        const size_t g = (size_t)blockIdx.x * WAVE + l;
        const size_t gc = g < a.n_xof ? g : a.n_xof - 1;
        uint32_t seed[8];
        const size_t kk = (size_t)(a.K * a.K), item = gc / kk;
        const unsigned e = (unsigned)(gc - item * kk), ra = e / (unsigned)a.K, cb = e - ra * (unsigned)a.K;
        load32(a.rho, a.rho_stride, item, seed);
        const unsigned i0 = a.transpose ? ra : cb, i1 = a.transpose ? cb : ra;
        keccak_zero(s);
        MLKEM_SET_WORDS8(s, 0, seed)
        keccak_xor_byte<32>(s, i0);
        keccak_xor_byte<33>(s, i1);
        keccak_xor_byte<34>(s, 0x1F);
        keccak_xor_byte<167>(s, 0x80);
        bool unfinished;
        uint32_t resume_cnt = 0;   // coefficients of this polynomial already in HBM when the sponge is handed over
        {
            char* lbuf = stage + l * LIN_STRIDE;   // this lane's staging buffer
#ifdef MLKEM_EMU
            char* const lds0 = lbuf;                 // emulator: positions are offsets into the lane's own buffer
            const uint32_t pos0 = 0;
#else
            const uint32_t pos0 = (uint32_t)reinterpret_cast<uintptr_t>(lbuf);   // low 32 bits of a generic LDS address = the LDS byte address
#endif
            uint32_t pos = pos0, lim = pos0 + 512u, flushed = 0;   // lim: address at which the 256th coefficient would land (moves with the flushes)
            char* dst_poly = reinterpret_cast<char*>(a.A + gc * 256);
            const bool valid = g < a.n_xof;
#define MLKEM_ACC_FAST(d) { const uint32_t dd = (d); MLKEM_LIN_ACCEPT(lds0, pos, dd); }
#define MLKEM_ACC_LIM(d) { const uint32_t dd = (d); MLKEM_LIN_ACCEPT_LIM(lds0, pos, dd, lim); }
#define MLKEM_LFLUSH { wave_lds_fence(); lin_flush(lbuf, pos0, pos, lim, flushed, dst_poly, valid); wave_lds_fence(); }
            keccak_f1600(s);
            MLKEM_LBLOCK(MLKEM_ACC_FAST, MLKEM_LFLUSH)     // blocks 1 and 2 cannot complete a polynomial (2 x 112 < 256): no limit test
            keccak_f1600(s);
            MLKEM_LBLOCK(MLKEM_ACC_FAST, MLKEM_LFLUSH)
            keccak_f1600(s);
            MLKEM_LBLOCK(MLKEM_ACC_LIM, MLKEM_LFLUSH)
#undef MLKEM_ACC_FAST
#undef MLKEM_ACC_LIM
#undef MLKEM_LFLUSH
            unfinished = flushed < 512u;
            if (unfinished && valid) {   // the staged remainder (< 32 bytes of coefficients, then don't-care bytes) joins the flushed part
                *reinterpret_cast<uint4*>(dst_poly + flushed) = *reinterpret_cast<const uint4*>(lbuf);
                *reinterpret_cast<uint4*>(dst_poly + flushed + 16u) = *reinterpret_cast<const uint4*>(lbuf + 16);
            }
            resume_cnt = (flushed + (pos - pos0)) >> 1;
        }
        if (unfinished && g < a.n_xof) {   // ~0.8 % of sponges: finished by the leftover passes
            const uint32_t idx = atomicAdd(&a.leftover[1], 1u);
            if (idx < a.resume_cap) {   // hand over the sponge as it stands: the fourth block costs one permutation there, not four
                uint32_t* e = a.resume + (size_t)idx * RESUME_WORDS;
                e[0] = (uint32_t)g;
                e[1] = resume_cnt;
#define MLKEM_SV(W) e[2 + W] = keccak_word<W>(s);
                MLKEM_SV(0) MLKEM_SV(1) MLKEM_SV(2) MLKEM_SV(3) MLKEM_SV(4) MLKEM_SV(5) MLKEM_SV(6) MLKEM_SV(7) MLKEM_SV(8) MLKEM_SV(9)
                MLKEM_SV(10) MLKEM_SV(11) MLKEM_SV(12) MLKEM_SV(13) MLKEM_SV(14) MLKEM_SV(15) MLKEM_SV(16) MLKEM_SV(17) MLKEM_SV(18) MLKEM_SV(19)
                MLKEM_SV(20) MLKEM_SV(21) MLKEM_SV(22) MLKEM_SV(23) MLKEM_SV(24) MLKEM_SV(25) MLKEM_SV(26) MLKEM_SV(27) MLKEM_SV(28) MLKEM_SV(29)
                MLKEM_SV(30) MLKEM_SV(31) MLKEM_SV(32) MLKEM_SV(33) MLKEM_SV(34) MLKEM_SV(35) MLKEM_SV(36) MLKEM_SV(37) MLKEM_SV(38) MLKEM_SV(39)
                MLKEM_SV(40) MLKEM_SV(41) MLKEM_SV(42) MLKEM_SV(43) MLKEM_SV(44) MLKEM_SV(45) MLKEM_SV(46) MLKEM_SV(47) MLKEM_SV(48) MLKEM_SV(49)
#undef MLKEM_SV
            } else {
                const uint32_t j = atomicAdd(&a.leftover[0], 1u);
                a.leftover[2 + j] = (uint32_t)g;
            }
        }
    } else {
        // ---------------- PRF role (ml_kem.c:496-515; SHAKE128 in the reference) ----------------
        const size_t g = (size_t)(blockIdx.x - a.xof_blocks) * WAVE + l;
        const size_t gc = g < a.n_prf ? g : a.n_prf - 1;
        uint32_t seed[8];
        const size_t item = gc / (size_t)a.per_item;
        const unsigned ctr = (unsigned)(gc - item * (size_t)a.per_item);
        load32(a.r, 32, item, seed);
        const unsigned eta = (int)ctr < a.n_eta1 ? (unsigned)a.eta1 : 2u;
        keccak_zero(s);
        MLKEM_SET_WORDS8(s, 0, seed)
        keccak_xor_byte<32>(s, ctr);
        keccak_xor_byte<33>(s, 0x1F);
        if (a.prf_rate == 136) keccak_xor_byte<135>(s, 0x80);   // SHAKE256 (FIPS 203 mode)
        else keccak_xor_byte<167>(s, 0x80);                     // SHAKE128 (the reference: ml_kem.c:508)
        keccak_f1600(s);
        // stage the first 128 bytes of every lane in LDS, 32 lanes at a time (32 rows x 33 dwords: odd stride,
        // conflict-free, 4.2 KB), and write them out as 16 B per lane, 8 lanes per row
        uint32_t* st32 = reinterpret_cast<uint32_t*>(stage);
        const size_t g0 = g - (size_t)l;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            if ((l >> 5) == half) {
                uint32_t* mine = st32 + (l & 31) * 33;
#define MLKEM_SW(W) mine[W] = keccak_word<W>(s);
                MLKEM_SW(0) MLKEM_SW(1) MLKEM_SW(2) MLKEM_SW(3) MLKEM_SW(4) MLKEM_SW(5) MLKEM_SW(6) MLKEM_SW(7)
                MLKEM_SW(8) MLKEM_SW(9) MLKEM_SW(10) MLKEM_SW(11) MLKEM_SW(12) MLKEM_SW(13) MLKEM_SW(14) MLKEM_SW(15)
                MLKEM_SW(16) MLKEM_SW(17) MLKEM_SW(18) MLKEM_SW(19) MLKEM_SW(20) MLKEM_SW(21) MLKEM_SW(22) MLKEM_SW(23)
                MLKEM_SW(24) MLKEM_SW(25) MLKEM_SW(26) MLKEM_SW(27) MLKEM_SW(28) MLKEM_SW(29) MLKEM_SW(30) MLKEM_SW(31)
#undef MLKEM_SW
            }
            wave_lds_fence();
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int f = it * WAVE + l, row = f >> 3, q = f & 7;   // row = staged lane (0..31), q = 16-byte piece
                const size_t gi = g0 + (size_t)(32 * half + row);
                if (gi < a.n_prf) {
                    uint4 v;
                    v.x = st32[row * 33 + 4 * q]; v.y = st32[row * 33 + 4 * q + 1];
                    v.z = st32[row * 33 + 4 * q + 2]; v.w = st32[row * 33 + 4 * q + 3];
                    *reinterpret_cast<uint4*>(a.prf + gi * a.prf_stride + 16 * q) = v;
                }
            }
            wave_lds_fence();
        }
        if (__ballot(eta == 3) != 0) {   // eta = 3: the rest of this block + the head of the next (192 bytes in all)
            const bool mine = g < a.n_prf && eta == 3;
            // the output row is recomputed from a fresh lane id after the permutation: nothing but the state is live
            // across it, which keeps the kernel inside its register budget (MLKEM_KECCAK_MINWAVES)
            auto out_fn = [&a]() {
                const size_t g2 = (size_t)(blockIdx.x - a.xof_blocks) * WAVE + lane_id_fresh();
                return reinterpret_cast<uint32_t*>(a.prf + (g2 < a.n_prf ? g2 : a.n_prf - 1) * a.prf_stride);
            };
            prf_eta3_tail(s, a.prf_rate == 136 ? 136u : 168u, out_fn, mine);
        }
    }
}

// k_sample_resume — the fourth (and, if ever needed, fifth) squeeze block of the sponges k_sample_main handed over:
// load the sponge state, permute, and append accepted candidates straight to the polynomial in HBM (on average a dozen
// 2-byte stores per sponge).  One permutation instead of the four a restart from the seed costs; the pass is latency-bound
// (one wave per CU), so this is wall time.  A sponge still short after block five follows the reference's retry
// (ml_kem.c:221-242: 279-triple limit, seed mutation): it goes to the restart list for k_sample.
template <int QB = KQ, int CAP = SAMPLE_CAP>
__global__ void __launch_bounds__(WAVE) k_sample_resume(SampleArgs a) {
    const int l = lane_id();
    const size_t limit = (size_t)(a.leftover[1] < a.resume_cap ? a.leftover[1] : a.resume_cap);
    for (size_t base = (size_t)blockIdx.x * WAVE; base < limit; base += (size_t)gridDim.x * WAVE) {
        const size_t slot = base + l;
        const bool valid = slot < limit;
        const uint32_t* e = a.resume + (valid ? slot : limit - 1) * RESUME_WORDS;
        const size_t g = e[0];
        uint32_t cnt = valid ? e[1] : 256u;
        KeccakState s;
#define MLKEM_LD(W) keccak_word<W>(s) = e[2 + W];
        MLKEM_LD(0) MLKEM_LD(1) MLKEM_LD(2) MLKEM_LD(3) MLKEM_LD(4) MLKEM_LD(5) MLKEM_LD(6) MLKEM_LD(7) MLKEM_LD(8) MLKEM_LD(9)
        MLKEM_LD(10) MLKEM_LD(11) MLKEM_LD(12) MLKEM_LD(13) MLKEM_LD(14) MLKEM_LD(15) MLKEM_LD(16) MLKEM_LD(17) MLKEM_LD(18) MLKEM_LD(19)
        MLKEM_LD(20) MLKEM_LD(21) MLKEM_LD(22) MLKEM_LD(23) MLKEM_LD(24) MLKEM_LD(25) MLKEM_LD(26) MLKEM_LD(27) MLKEM_LD(28) MLKEM_LD(29)
        MLKEM_LD(30) MLKEM_LD(31) MLKEM_LD(32) MLKEM_LD(33) MLKEM_LD(34) MLKEM_LD(35) MLKEM_LD(36) MLKEM_LD(37) MLKEM_LD(38) MLKEM_LD(39)
        MLKEM_LD(40) MLKEM_LD(41) MLKEM_LD(42) MLKEM_LD(43) MLKEM_LD(44) MLKEM_LD(45) MLKEM_LD(46) MLKEM_LD(47) MLKEM_LD(48) MLKEM_LD(49)
#undef MLKEM_LD
        uint16_t* poly = a.A + g * 256;
#define MLKEM_RC(d) { const uint32_t dd = (d); if (dd < (uint32_t)QB && cnt < 256u) { poly[cnt] = (uint16_t)dd; cnt++; } }
// group G of the block; in the fifth block the reference never uses the triples from the cap on (ml_kem.c:223-227: with the
// product's CAP = 278 that is triples 278, 279 -- the 279th only trips the limit)
#define MLKEM_RG(W0, G)                                                                                         \
            if constexpr (sample_nt5(CAP, G) == 4) { MLKEM_LG(MLKEM_RC, W0, 0, 4) }                             \
            else { if (blk == 4) { MLKEM_LG(MLKEM_RC, W0, 0, sample_nt5(CAP, G)) } else { MLKEM_LG(MLKEM_RC, W0, 0, 4) } }
#pragma unroll 1
        for (int blk = 3; blk < 5; blk++) {
            keccak_f1600(s);
            MLKEM_RG(0, 0) MLKEM_RG(3, 1) MLKEM_RG(6, 2) MLKEM_RG(9, 3) MLKEM_RG(12, 4) MLKEM_RG(15, 5) MLKEM_RG(18, 6)
            MLKEM_RG(21, 7) MLKEM_RG(24, 8) MLKEM_RG(27, 9) MLKEM_RG(30, 10) MLKEM_RG(33, 11) MLKEM_RG(36, 12) MLKEM_RG(39, 13)
            if (__ballot(cnt < 256u) == 0) break;
        }
#undef MLKEM_RG
#undef MLKEM_RC
        if (cnt < 256u) {   // valid lanes only (the others start at 256): restart from the mutated seed, as the reference does
            const uint32_t j = atomicAdd(&a.leftover[0], 1u);
            a.leftover[2 + j] = (uint32_t)g;
        }
    }
}

}   // namespace mlkem
