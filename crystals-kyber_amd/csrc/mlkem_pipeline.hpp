// mlkem_pipeline.hpp — kernel sequencing for the batched ML-KEM operations.
//
// The same sequencing is compiled twice: by hipcc into libmlkem_amd.so (real launches on a HIP stream)
// and by g++ into the test-only wave emulator (tests/emu), which runs the identical kernel bodies on
// 64 host threads per wave so that kernel logic can be checked without a GPU.  There is no CPU fallback
// in the product: the emulator is never built into, or loaded by, libmlkem_amd.so.
//
// Two granularities:
//   hash stages (lane = item, 32-byte results)   run over "h-chunks" of up to Workspace::hcap items, so that even
//                                                 a serial 9-permutation sponge per lane fills 1024 SIMDs;
//   sampler + polynomial arithmetic               run over chunks of Workspace::cap items, because their
//                                                 intermediates (A-hat, PRF bytes: ~5.5 KB per item) live in scratch.
#pragma once
#include "mlkem_kernels.hpp"
#include "mlkem_sampler.hpp"
#include "mlkem_arith.hpp"
#include "mlkem_rntt.hpp"
#include "mlkem_kpke4.hpp"
#include "mlkem_kpke2.hpp"
#include "mlkem_wkeccak.hpp"
#include "mlkem_small.hpp"
#include <stdlib.h>
#ifndef MLKEM_EMU
#include <vector>
#endif

namespace mlkem {

#ifdef MLKEM_EMU
using stream_t = void*;
#else
using stream_t = hipStream_t;
#endif

#ifndef MLKEM_EMU
// Optional per-kernel timing (bench.py's roofline leg): when a recorder is installed for the calling thread,
// every launch is bracketed by HIP events on the launch stream.
struct LaunchRecorder {
    struct Rec { const char* label; hipEvent_t a, b; };
    std::vector<Rec> recs;
    bool failed = false;   // an event could not be created / recorded: the rows would be incomplete
};
inline LaunchRecorder*& launch_recorder() {
    static thread_local LaunchRecorder* r = nullptr;
    return r;
}
#endif

template <class... KArgs, class... Args>
inline void launch(const char* label, void (*kfn)(KArgs...), size_t grid, unsigned block, stream_t st, Args... args) {
    if (grid == 0) return;
#ifdef MLKEM_EMU
    (void)st; (void)label;
    emu::launch((unsigned)grid, block, [=] { kfn(args...); });
#else
    LaunchRecorder* rec = launch_recorder();
    LaunchRecorder::Rec r{label, nullptr, nullptr};
    // the kernel is launched whatever happens to the measurement: an event that cannot be created or recorded drops this
    // launch from the timing and marks the recorder failed (mlkem_timing_end reports it instead of a silent zero)
    bool timed = rec && hipEventCreate(&r.a) == hipSuccess && hipEventCreate(&r.b) == hipSuccess && hipEventRecord(r.a, st) == hipSuccess;
    kfn<<<dim3((unsigned)grid), dim3(block), 0, st>>>(args...);
    if (timed) timed = hipEventRecord(r.b, st) == hipSuccess;
    if (timed) {
        rec->recs.push_back(r);
    } else if (rec) {
        rec->failed = true;
        if (r.a) (void)hipEventDestroy(r.a);
        if (r.b) (void)hipEventDestroy(r.b);
    }
#endif
}

inline void zero_u32x2(stream_t st, uint32_t* p) {   // the two counters in front of the restart list
#ifdef MLKEM_EMU
    (void)st;
    p[0] = 0; p[1] = 0;
#else
    (void)hipMemsetAsync(p, 0, 2 * sizeof(uint32_t), st);
#endif
}

struct ParamSet {
    int set, k, eta1, eta2, du, dv;
    unsigned ek_len, dk_len, c_len;
};
// ml_kem.c:1363-1395 (init) + ml_kem.h:52-59 (lengths)
inline bool param_set(int set, ParamSet& p) {
    switch (set) {
    case 512: p = {512, 2, 3, 2, 10, 4, 0, 0, 0}; break;
    case 768: p = {768, 3, 2, 2, 10, 4, 0, 0, 0}; break;
    case 1024: p = {1024, 4, 2, 2, 11, 5, 0, 0, 0}; break;
    default: return false;
    }
    p.ek_len = 384u * p.k + 32;
    p.dk_len = 768u * p.k + 96;
    p.c_len = 32u * (p.du * p.k + p.dv);
    return true;
}

// Scratch in HBM.  Per chunk item (worst case k = 4):
//   A        k*k*512   sampled matrix, uint16 coefficients, natural order
//   prf      (2k+1)*PS raw PRF output rows (PS = 192 for eta1 = 3, else 128)
//   leftover 4*(k*k)   sponge indices needing a restart from the seed (+2 counter words)
//   resume   208*(k*k)/8   saved sponges (index, count, Keccak state) that need a 4th squeeze block: room for 1/8 of all
//                      sponges (expected: 0.8 %), the overflow goes to the restart list
// Per h-chunk item: r, rho, m, Kp, Kbar : 32 bytes each.
struct Workspace {
    uint16_t* A = nullptr;      // sampled matrices of one chunk
    uint8_t *prf = nullptr, *r = nullptr, *rho = nullptr, *m = nullptr, *Kp = nullptr, *Kbar = nullptr;
    uint32_t* leftover = nullptr;
    uint32_t* resume = nullptr;
    uint32_t resume_cap = 0;    // records `resume` has room for
    size_t cap = 0;    // chunk capacity (items) of A / prf / leftover
    size_t hcap = 0;   // h-chunk capacity (items) of the 32-byte arrays
    size_t wide_max = 2048;   // stand-alone SampleNTT / sponge calls of at most this many inputs run one sponge per wavefront
    // KEM calls of at most wide_max_k[k - 2] items (and more than small_max) hash with one sponge per WAVEFRONT (mlkem_wkeccak.hpp) and
    // sample in direct mode (one launch).  ms per ML-KEM-768 pair, this form against lane-sliced: 0.257 / 0.297 at 2304 items,
    // 0.273 / 0.301 at 2816, 0.302 / 0.295 at 3328; ML-KEM-1024 triple 0.454 / 0.594 at 2304, 0.579 / 0.611 at 3840, 0.699 / 0.663 at
    // 4608; ML-KEM-512 0.397 / 0.410 at 3840, 0.433 / 0.422 at 4608 (profiles/r04_small_limits.txt).  env MLKEM_WIDE_HASH_ITEMS sets
    // all of them (0: always the lane-sliced hash kernels)
    size_t wide_max_k[3] = {4096, 3072, 4096};
    size_t wide_kem(int k) const { return wide_max_k[k - 2]; }
    // calls of at most small_max_k[k - 2] items run as ONE launch, one workgroup per item (mlkem_small.hpp).  Per parameter set: the
    // limit is what the chip takes in ONE round of four-wave workgroups -- 6 / 3 / 2 per CU at k = 2 / 3 / 4 (91-122 / 118-178 /
    // 152-226 VGPRs) x 256 CUs; a second round loses to the batch path (small against batch, ms per call triple or pair:
    // ML-KEM-512 0.281 / 0.305 at 1536 items, 0.318 / 0.315 at 1792; ML-KEM-768 0.156 / 0.193 at 768, 0.199 / 0.193 at 896;
    // ML-KEM-1024 0.278 / 0.313 at 512, 0.387 / 0.326 at 576; profiles/r04_small_limits.txt).  env MLKEM_SMALL_ITEMS sets all three (0: never)
    size_t small_max_k[3] = {1536, 768, 512};
    size_t small_max(int k) const { return small_max_k[k - 2]; }
    size_t small_lat_max = 256;   // ... of which calls of at most this many items use eight waves per item (shortest chain: 0.070
                                  // against 0.083 ms per ML-KEM-768 pair at one item, 0.085 against 0.090 at 256), larger ones four:
                                  // beyond one workgroup per CU the eight-wave form needs a second round in every parameter set
                                  // (768: 0.118 against 0.110 at 288; 1024: 0.323 against 0.268; 512: a tie); env MLKEM_SMALL_LATENCY_ITEMS
    size_t small_wide_max = 128;  // ... and Decaps calls of at most this many items twelve (k >= 3); env MLKEM_SMALL_WIDE_ITEMS
    // measurement aid (mlkem_ctx_debug_stages, tools/energy_probe.py): which kernel families the batch path launches; the
    // outputs of a call with stages missing are meaningless.  1 = hash kernels, 2 = sampler, 4 = K-PKE.Encrypt / KeyGen, 8 = Decrypt
    unsigned stages = 15;
    int fips = 0;      // 0: bit-identical to the reference (PRF, J on SHAKE128; no-op modulus check)
                       // 1: FIPS 203 conformant (PRF, J on SHAKE256; encaps reports status -4 for t-hat coefficients >= q)
#ifndef MLKEM_EMU
    // side stream of the context + fork / join events: calls that fit ONE chunk sample the matrix there while the hash
    // kernels run on the caller's stream (SideFork below); null = everything on the caller's stream
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_on = true;   // the streaming front-end turns it off while several chunks are in flight: its copy streams already
                           // keep the GPU busy, and a third compute stream cost 15 % of the pinned-buffer streaming rate
#endif
};
// A call of more than one chunk runs on the caller's stream alone, chunk after chunk: sampler -> leftover passes ->
// arithmetic.  A two-stream pipeline over chunks (arithmetic of chunk i beside the sampler of chunk i + 1, also with the
// arithmetic confined to a subset of the CUs) and a phase-separated schedule were measured and do not pay at 2^20 items: the
// pass is energy-limited (profiles/r02_sampler_experiments.txt, profiles/r03_power_schedule.txt).
// A call that fits one chunk does not fill the GPU for long (one sponge per lane: 2^14 items are 256 waves of hashing on 1024
// SIMDs) and its time is the LENGTH of its dependency chain: H(ek) -> G -> PRF -> Encrypt on one side, SampleNTT (which needs
// only rho) on the other.  SideFork runs the matrix sampling on the context's side stream from the start of the call and
// joins before the arithmetic kernel (profiles/r03_batch_sweep.txt).
struct SideFork {
    stream_t main, side = nullptr;
#ifndef MLKEM_EMU
    hipEvent_t join = nullptr;
#endif
    // active only when the whole call is one chunk: the side stream then owns ws.A / ws.leftover / ws.resume until join()
    SideFork(const Workspace& ws, stream_t st, size_t n) : main(st) {
#ifndef MLKEM_EMU
        if (ws.side && ws.side_on && n <= ws.cap && hipEventRecord(ws.ev_fork, st) == hipSuccess && hipStreamWaitEvent(ws.side, ws.ev_fork, 0) == hipSuccess) {
            side = ws.side;
            join = ws.ev_join;
        }
#else
        (void)ws; (void)n;
#endif
    }
    bool active() const { return side != nullptr; }
    stream_t xof_stream() const { return side ? side : main; }
    void join_main() {   // everything launched on the side stream so far precedes what follows on the caller's stream
#ifndef MLKEM_EMU
        if (side) {
            // a failed record / wait leaves no ordering: fall back to a full wait for the side stream
            if (hipEventRecord(join, side) != hipSuccess || hipStreamWaitEvent(main, join, 0) != hipSuccess) (void)hipStreamSynchronize(side);
            side = nullptr;
        }
#endif
    }
};

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }
inline size_t min_sz(size_t a, size_t b) { return a < b ? a : b; }

// SampleNTT for the k x k matrix + PRF rows of `n` items: the three-block main kernel, then the leftover passes
// (ml_kem.c:189-245, :496-515)
// n_xof_items / n_prf_items: items whose matrix / PRF rows are produced (equal except for shared-key batches, where the
// matrix is sampled once and the PRF rows per item)
// QB, CAP: acceptance bound and triple cap of SampleNTT (mlkem_kernels.hpp: the product uses the defaults; the CPU tier also
// instantiates lower values to run the fifth squeeze block and the seed-mutation retry)
template <int QB = KQ, int CAP = SAMPLE_CAP>
inline void launch_sample_split(stream_t st, const ParamSet& p, size_t n_xof_items, size_t n_prf_items, const uint8_t* rho, size_t rho_stride,
                                int transpose, const uint8_t* r, int prf_per_item, int n_eta1, const Workspace& ws) {
    SampleArgs a{};
    a.n_xof = n_xof_items * (size_t)(p.k * p.k);
    a.rho = rho; a.rho_stride = rho_stride; a.K = p.k; a.transpose = transpose; a.A = ws.A;
    a.xof_blocks = (unsigned)ceil_div(a.n_xof, WAVE);
    a.n_prf = n_prf_items * (size_t)prf_per_item;
    a.r = r; a.per_item = prf_per_item; a.n_eta1 = n_eta1; a.eta1 = p.eta1; a.prf = ws.prf;
    a.prf_stride = p.eta1 == 3 ? 192 : 128;
    a.leftover = ws.leftover;
    a.resume = ws.resume;
    a.resume_cap = ws.resume ? ws.resume_cap : 0;
    a.list_mode = 0;
    a.prf_rate = ws.fips ? 136 : 168;
    const size_t grid = a.xof_blocks + ceil_div(a.n_prf, WAVE);
    if (grid == 0) return;
    if (n_xof_items <= ws.wide_kem(p.k) && n_prf_items <= ws.wide_kem(p.k)) {
        // small call: the general sampler finishes every sponge itself (a wave runs a fourth permutation when one of its lanes
        // needs it) -- one launch instead of three, 0.045 instead of 0.08 ms on the call's critical path; at full batches the
        // three-block kernel + hand-over is 13 % cheaper (mlkem_sampler.hpp)
        launch("k_sample_direct", k_sample<QB, CAP>, grid, WAVE, st, a);
        return;
    }
    if (a.n_xof) zero_u32x2(st, ws.leftover);   // the counters belong to the stream that samples the matrix
    launch("k_sample_main", k_sample_main<QB>, grid, WAVE, st, a);
    if (a.n_xof == 0) return;
    // leftovers: expected 0.8 % of the sponges; the grids cover 1/16 of them and stride over the rest if ever needed.
    // First the sponges handed over with their state (one more permutation each), then the restart list (normally empty).
    SampleArgs t = a;
    t.list_mode = 1;
    t.n_prf = 0;
    t.xof_blocks = (unsigned)(ceil_div(a.n_xof, WAVE * 16) + 1);
    if (a.resume_cap) {
        launch("k_sample_tail", k_sample_resume<QB, CAP>, (size_t)t.xof_blocks, WAVE, st, t);
        t.xof_blocks = (unsigned)(ceil_div(a.n_xof, WAVE * 256) + 1);
    }
    launch("k_sample_restart", k_sample<QB, CAP>, (size_t)t.xof_blocks, WAVE, st, t);
}
template <int QB = KQ, int CAP = SAMPLE_CAP>
inline void launch_sample(stream_t st, const ParamSet& p, size_t n, const uint8_t* rho, size_t rho_stride, int transpose,
                          const uint8_t* r, int prf_per_item, int n_eta1, const Workspace& ws) {
    launch_sample_split<QB, CAP>(st, p, n, n, rho, rho_stride, transpose, r, prf_per_item, n_eta1, ws);
}

// K-PKE.Encrypt launch: two items per wave (mlkem_kpke2.hpp)
template <int K, int ETA1, int DU, int DV, bool CMP, class... Args>
inline void encrypt_launch(const char* label, stream_t st, size_t n, Args... args) {
    launch(label, k_encrypt2<K, ETA1, DU, DV, CMP>, ceil_div(ceil_div(n, 2), KPKE2_WAVES), WAVE * KPKE2_WAVES, st, n, args...);
}

// ---- ML-KEM.KeyGen_internal (ml_kem.c:1034-1084) ; z == nullptr: K-PKE.KeyGen alone (ml_kem.c:651-769, dk = 384k-byte ŝ) ----
template <int K, int ETA1>
inline void keygen_run(stream_t st, const ParamSet& p, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk,
                       const Workspace& ws) {
    const bool kem = z != nullptr;
    const size_t dk_len = kem ? (size_t)p.dk_len : (size_t)(384 * K);
    if (n && n <= ws.small_max(K)) {   // small call: one launch, one workgroup per item (its intermediates live in LDS: no scratch)
        const int rate = ws.fips ? 136 : 168;
#define MLKEM_KS(KEM, NW) launch("k_keygen_small", k_keygen_small<K, ETA1, KEM, NW>, n, WAVE * NW, st, n, d, z, ek, dk, rate)
        const bool lat = n <= ws.small_lat_max;
        if (kem && lat) MLKEM_KS(true, SMALL_WAVES);
        else if (kem) MLKEM_KS(true, SMALL_WAVES_DENSE);
        else if (lat) MLKEM_KS(false, SMALL_WAVES);
        else MLKEM_KS(false, SMALL_WAVES_DENSE);
#undef MLKEM_KS
        return;
    }
    for (size_t h0 = 0; h0 < n; h0 += ws.hcap) {
        const size_t hn = min_sz(ws.hcap, n - h0);
        launch("k_hash_keygen_seed", k_hash_keygen_seed<K>, ceil_div(hn, WAVE), WAVE, st, hn, d + h0 * 32, ws.rho, ws.r);
        for (size_t c0 = 0; c0 < hn; c0 += ws.cap) {
            const size_t cn = min_sz(ws.cap, hn - c0), i0 = h0 + c0;
            launch_sample(st, p, cn, ws.rho + c0 * 32, 32, /*transpose=*/0, ws.r + c0 * 32, 2 * K, 2 * K, ws);
            const size_t kgrid = ceil_div(ceil_div(cn, 2), KPKE2_WAVES);
            if (kem)
                launch("k_keygen", k_keygen2<K, ETA1, true>, kgrid, WAVE * KPKE2_WAVES, st, cn, (const uint16_t*)ws.A, (const uint8_t*)ws.prf,
                       (const uint8_t*)(ws.rho + c0 * 32), ek + i0 * p.ek_len, dk + i0 * dk_len);
            else
                launch("k_keygen", k_keygen2<K, ETA1, false>, kgrid, WAVE * KPKE2_WAVES, st, cn, (const uint16_t*)ws.A, (const uint8_t*)ws.prf,
                       (const uint8_t*)(ws.rho + c0 * 32), ek + i0 * p.ek_len, dk + i0 * dk_len);
        }
        if (kem && n <= ws.wide_kem(K))   // small call: one sponge per wave (mlkem_wkeccak.hpp)
            launch("k_hash_keygen_fin", k_hash_keygen_fin_w<K>, hn, WAVE, st, hn, (const uint8_t*)(ek + h0 * p.ek_len), z + h0 * 32, dk + h0 * p.dk_len);
        else if (kem)
            launch("k_hash_keygen_fin", k_hash_keygen_fin<K>, ceil_div(hn, WAVE), WAVE, st, hn, (const uint8_t*)(ek + h0 * p.ek_len),
                   z + h0 * 32, dk + h0 * p.dk_len);
    }
}

// ---- ML-KEM.Encaps_internal (ml_kem.c:1093-1130) -------------------------------------------------
template <int K, int ETA1, int DU, int DV>
// r_user != nullptr: K-PKE.Encrypt alone (ml_kem.c:776-936) with the caller's randomness; no hashing, Kout unused
inline void encaps_run(stream_t st, const ParamSet& p, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* Kout,
                       int32_t* mod_status, const Workspace& ws, const uint8_t* r_user = nullptr) {
    if (n == 0) return;
    if (!r_user && n <= ws.small_max(K)) {   // small call: one launch, one workgroup per item
        const int rate = ws.fips ? 136 : 168;
        if (n <= ws.small_lat_max)
            launch("k_encaps_small", k_encaps_small<K, ETA1, DU, DV, SMALL_WAVES>, n, WAVE * SMALL_WAVES, st, n, ek, m, c, Kout, mod_status, rate);
        else
            launch("k_encaps_small", k_encaps_small<K, ETA1, DU, DV, SMALL_WAVES_DENSE>, n, WAVE * SMALL_WAVES_DENSE, st, n, ek, m, c, Kout, mod_status, rate);
        return;
    }
    SideFork fork(ws, st, r_user ? (size_t)-1 : n);   // one chunk: A-hat^T (needs rho alone) is sampled beside H(ek) and G
    if (fork.active()) launch_sample_split(fork.xof_stream(), p, n, 0, ek + 384 * K, p.ek_len, /*transpose=*/1, nullptr, 2 * K + 1, K, ws);
    for (size_t h0 = 0; h0 < n; h0 += ws.hcap) {
        const size_t hn = min_sz(ws.hcap, n - h0);
        const uint8_t* r_h = r_user ? r_user + h0 * 32 : (const uint8_t*)ws.r;
        if (!(ws.stages & 1u)) {
        } else if (!r_user && n <= ws.wide_kem(K))   // mid-size call: one sponge per wave (mlkem_wkeccak.hpp)
            launch("k_hash_encaps", k_hash_encaps_w<K>, hn, WAVE, st, hn, ek + h0 * p.ek_len, m + h0 * 32, Kout + h0 * 32, ws.r);
        else if (!r_user)
            launch("k_hash_encaps", k_hash_encaps<K>, ceil_div(hn, WAVE), WAVE, st, hn, ek + h0 * p.ek_len, m + h0 * 32, Kout + h0 * 32, ws.r);
        for (size_t c0 = 0; c0 < hn; c0 += ws.cap) {
            const size_t cn = min_sz(ws.cap, hn - c0), i0 = h0 + c0;
            const uint8_t* eki = ek + i0 * p.ek_len;
            if (fork.active()) {   // PRF rows on the caller's stream (they need r), then wait for the matrix
                launch_sample_split(st, p, 0, cn, nullptr, 0, 1, r_h + c0 * 32, 2 * K + 1, K, ws);
                fork.join_main();
            } else if (ws.stages & 2u) {
                launch_sample(st, p, cn, eki + 384 * K, p.ek_len, /*transpose=*/1, r_h + c0 * 32, 2 * K + 1, K, ws);
            }
            if (ws.stages & 4u)
            encrypt_launch<K, ETA1, DU, DV, false>("k_encrypt", st, cn, eki, (size_t)p.ek_len, m + i0 * 32, (const uint16_t*)ws.A, (const uint8_t*)ws.prf,
                   c + i0 * p.c_len, (const uint8_t*)nullptr, (const uint8_t*)nullptr, (const uint8_t*)nullptr, (uint8_t*)nullptr,
                   mod_status ? mod_status + i0 : (int32_t*)nullptr, (size_t)(K * K * 256));
        }
    }
}

// ---- KEM_Decaps / Decaps_internal (ml_kem.c:1310-1359, :1136-1225) -------------------------------
// hash_check = true reproduces the public KEM_Decaps: status[i] = -5 when H(dk.ek) != dk.h (K[i] is then
// still the Decaps_internal result; the host shim discards it like the reference does).
template <int K, int DU, int DV>
inline void decrypt_launch(stream_t st, size_t n, const uint8_t* dk, size_t dk_stride, const uint8_t* c, uint8_t* m) {
    // four items per wave (mlkem_kpke4.hpp)
    launch("k_decrypt", k_decrypt4<K, DU, DV>, ceil_div(ceil_div(n, 4), KPKE4_WAVES), 64 * KPKE4_WAVES, st, n, dk, dk_stride, c, m);
}

template <int K, int ETA1, int DU, int DV>
inline void decaps_run(stream_t st, const ParamSet& p, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* Kout,
                       int32_t* status, bool hash_check, const Workspace& ws) {
    constexpr int CLEN = 32 * (DU * K + DV);
    if (n && n <= ws.small_max(K)) {   // small call: one launch, one workgroup per item
        const int rate = ws.fips ? 136 : 168;
        int32_t* sts = hash_check ? status : (int32_t*)nullptr;
#define MLKEM_DS(HC, JR, NW) launch("k_decaps_small", k_decaps_small<K, ETA1, DU, DV, HC, JR, NW>, (HC) ? 2 * n : n, WAVE * NW, st, n, dk, c, Kout, sts, rate)
        constexpr int LW = K >= 3 ? SMALL_WAVES_WIDE : SMALL_WAVES;   // k^2 = 9 / 16 SampleNTT jobs: twelve waves while a workgroup has a CU to itself
#define MLKEM_DS2(HC, JR) do { if (n <= ws.small_wide_max) MLKEM_DS(HC, JR, LW); else if (n <= ws.small_lat_max) MLKEM_DS(HC, JR, SMALL_WAVES); else MLKEM_DS(HC, JR, SMALL_WAVES_DENSE); } while (0)
        if (hash_check && !ws.fips) MLKEM_DS2(true, 168);
        else if (!ws.fips) MLKEM_DS2(false, 168);
        else if (hash_check) MLKEM_DS2(true, 136);
        else MLKEM_DS2(false, 136);
#undef MLKEM_DS2
#undef MLKEM_DS
        return;
    }
    for (size_t h0 = 0; h0 < n; h0 += ws.hcap) {
        const size_t hn = min_sz(ws.hcap, n - h0);
        const uint8_t* dkh = dk + h0 * p.dk_len;
        const uint8_t* ch = c + h0 * p.c_len;
        SideFork fork(ws, st, n);   // one chunk: A-hat^T of the re-encryption is sampled beside Decrypt and the three sponges
        if (fork.active()) launch_sample_split(fork.xof_stream(), p, n, 0, dk + 768 * K, p.dk_len, /*transpose=*/1, nullptr, 2 * K + 1, K, ws);
        if (ws.stages & 8u) decrypt_launch<K, DU, DV>(st, hn, dkh, (size_t)p.dk_len, ch, ws.m);
        int32_t* sth = (hash_check && status) ? status + h0 : (int32_t*)nullptr;
        const size_t hgrid = ceil_div(hn, WAVE);
        if (!(ws.stages & 1u)) {
        } else if (n <= ws.wide_kem(K)) {   // small call: one sponge per wave (mlkem_wkeccak.hpp)
            if (hash_check && !ws.fips)
                launch("k_hash_decaps", k_hash_decaps_w<K, CLEN, true, 168>, 2 * hn, WAVE, st, hn, dkh, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, sth, (size_t)p.dk_len);
            else if (!ws.fips)
                launch("k_hash_decaps", k_hash_decaps_w<K, CLEN, false, 168>, hn, WAVE, st, hn, dkh, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, sth, (size_t)p.dk_len);
            else if (hash_check)
                launch("k_hash_decaps", k_hash_decaps_w<K, CLEN, true, 136>, 2 * hn, WAVE, st, hn, dkh, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, sth, (size_t)p.dk_len);
            else
                launch("k_hash_decaps", k_hash_decaps_w<K, CLEN, false, 136>, hn, WAVE, st, hn, dkh, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, sth, (size_t)p.dk_len);
        } else
        if (hash_check && !ws.fips)
            launch("k_hash_decaps", k_hash_decaps<K, CLEN, true, 168>, 2 * hgrid, WAVE, st, hn, dkh, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, sth, (size_t)p.dk_len);
        else if (!ws.fips)
            launch("k_hash_decaps", k_hash_decaps<K, CLEN, false, 168>, hgrid, WAVE, st, hn, dkh, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, sth, (size_t)p.dk_len);
        else if (hash_check)
            launch("k_hash_decaps", k_hash_decaps<K, CLEN, true, 136>, 2 * hgrid, WAVE, st, hn, dkh, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, sth, (size_t)p.dk_len);
        else
            launch("k_hash_decaps", k_hash_decaps<K, CLEN, false, 136>, hgrid, WAVE, st, hn, dkh, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, sth, (size_t)p.dk_len);
        for (size_t c0 = 0; c0 < hn; c0 += ws.cap) {
            const size_t cn = min_sz(ws.cap, hn - c0), i0 = h0 + c0;
            const uint8_t* dki = dk + i0 * p.dk_len;
            if (fork.active()) {
                launch_sample_split(st, p, 0, cn, nullptr, 0, 1, ws.r + c0 * 32, 2 * K + 1, K, ws);
                fork.join_main();
            } else if (ws.stages & 2u) {
                launch_sample(st, p, cn, dki + 768 * K, p.dk_len, /*transpose=*/1, ws.r + c0 * 32, 2 * K + 1, K, ws);
            }
            if (ws.stages & 4u)
            encrypt_launch<K, ETA1, DU, DV, true>("k_encrypt_cmp", st, cn, dki + 384 * K, (size_t)p.dk_len, (const uint8_t*)(ws.m + c0 * 32), (const uint16_t*)ws.A,
                   (const uint8_t*)ws.prf, (uint8_t*)nullptr, c + i0 * p.c_len, (const uint8_t*)(ws.Kp + c0 * 32),
                   (const uint8_t*)(ws.Kbar + c0 * 32), Kout + i0 * 32, (int32_t*)nullptr, (size_t)(K * K * 256));
        }
    }
}

// ---- shared-key batches: ONE encapsulation key (encaps) or ONE decapsulation key (decaps) for all n items -----------
// Same bytes as the per-item calls on replicated keys, but H(ek), the dk hash check and the k x k matrix (9 of the 44 /
// 36 of the 51 Keccak-f per item at k = 3 ... plus H: 35 / 36) are computed once instead of n times.
template <int K, int ETA1, int DU, int DV>
inline void encaps_shared_run(stream_t st, const ParamSet& p, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* Kout,
                              const Workspace& ws) {
    const Workspace& w = ws;
    uint8_t* h = ws.rho;                                                          // 32 bytes, unused by Encaps otherwise
    launch("k_hash_batch", k_hash_batch<0>, (size_t)1, WAVE, st, (size_t)1, ek, (unsigned)p.ek_len, (size_t)p.ek_len, h);
    launch_sample_split(st, p, 1, 0, ek + 384 * K, p.ek_len, /*transpose=*/1, nullptr, 0, 0, w);   // A^T once
    for (size_t h0 = 0; h0 < n; h0 += ws.hcap) {
        const size_t hn = min_sz(ws.hcap, n - h0);
        launch("k_hash_g_shared", k_hash_g_shared, ceil_div(hn, WAVE), WAVE, st, hn, m + h0 * 32, (const uint8_t*)h, Kout + h0 * 32, ws.r);
        for (size_t c0 = 0; c0 < hn; c0 += ws.cap) {
            const size_t cn = min_sz(ws.cap, hn - c0), i0 = h0 + c0;
            launch_sample_split(st, p, 0, cn, nullptr, 0, 1, ws.r + c0 * 32, 2 * K + 1, K, w);      // PRF rows per item
            encrypt_launch<K, ETA1, DU, DV, false>("k_encrypt", st, cn, ek, (size_t)0,
                   m + i0 * 32, (const uint16_t*)w.A, (const uint8_t*)w.prf, c + i0 * p.c_len, (const uint8_t*)nullptr,
                   (const uint8_t*)nullptr, (const uint8_t*)nullptr, (uint8_t*)nullptr, (int32_t*)nullptr, (size_t)0);
        }
    }
}
template <int K, int ETA1, int DU, int DV>
inline void decaps_shared_run(stream_t st, const ParamSet& p, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* Kout,
                              int32_t* status, const Workspace& ws) {
    constexpr int CLEN = 32 * (DU * K + DV);
    const Workspace& w = ws;
    if (status) {   // KEM_Decaps' hash check, once
        launch("k_hash_batch", k_hash_batch<0>, (size_t)1, WAVE, st, (size_t)1, dk + 384 * K, (unsigned)p.ek_len, (size_t)p.dk_len, ws.rho);
        launch("k_status_fill", k_status_fill, min_sz(ceil_div(n, 256), 1024), 256u, st, n, (const uint8_t*)ws.rho, dk + 768 * K + 32, status);
    }
    launch_sample_split(st, p, 1, 0, dk + 768 * K, p.dk_len, /*transpose=*/1, nullptr, 0, 0, w);   // A^T once (rho sits in dk.ek)
    for (size_t h0 = 0; h0 < n; h0 += ws.hcap) {
        const size_t hn = min_sz(ws.hcap, n - h0);
        const uint8_t* ch = c + h0 * p.c_len;
        decrypt_launch<K, DU, DV>(st, hn, dk, (size_t)0, ch, ws.m);
        const size_t hgrid = ceil_div(hn, WAVE);
        if (!ws.fips)
            launch("k_hash_decaps", k_hash_decaps<K, CLEN, false, 168>, hgrid, WAVE, st, hn, dk, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, (int32_t*)nullptr, (size_t)0);
        else
            launch("k_hash_decaps", k_hash_decaps<K, CLEN, false, 136>, hgrid, WAVE, st, hn, dk, ch, (const uint8_t*)ws.m, ws.Kp, ws.r, ws.Kbar, (int32_t*)nullptr, (size_t)0);
        for (size_t c0 = 0; c0 < hn; c0 += ws.cap) {
            const size_t cn = min_sz(ws.cap, hn - c0), i0 = h0 + c0;
            launch_sample_split(st, p, 0, cn, nullptr, 0, 1, ws.r + c0 * 32, 2 * K + 1, K, w);
            encrypt_launch<K, ETA1, DU, DV, true>("k_encrypt_cmp", st, cn, dk + 384 * K,
                   (size_t)0, (const uint8_t*)(ws.m + c0 * 32), (const uint16_t*)w.A, (const uint8_t*)w.prf, (uint8_t*)nullptr, c + i0 * p.c_len,
                   (const uint8_t*)(ws.Kp + c0 * 32), (const uint8_t*)(ws.Kbar + c0 * 32), Kout + i0 * 32, (int32_t*)nullptr, (size_t)0);
        }
    }
}
inline int encaps_shared_dispatch(stream_t st, int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K, const Workspace& ws) {
    ParamSet p;
    if (!param_set(set, p)) return -1;
    if (n == 0) return 0;
    switch (set) {
    case 512: encaps_shared_run<2, 3, 10, 4>(st, p, n, ek, m, c, K, ws); break;
    case 768: encaps_shared_run<3, 2, 10, 4>(st, p, n, ek, m, c, K, ws); break;
    default: encaps_shared_run<4, 2, 11, 5>(st, p, n, ek, m, c, K, ws); break;
    }
    return 0;
}
inline int decaps_shared_dispatch(stream_t st, int set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status,
                                  const Workspace& ws) {
    ParamSet p;
    if (!param_set(set, p)) return -1;
    if (n == 0) return 0;
    switch (set) {
    case 512: decaps_shared_run<2, 3, 10, 4>(st, p, n, dk, c, K, status, ws); break;
    case 768: decaps_shared_run<3, 2, 10, 4>(st, p, n, dk, c, K, status, ws); break;
    default: decaps_shared_run<4, 2, 11, 5>(st, p, n, dk, c, K, status, ws); break;
    }
    return 0;
}

inline int keygen_dispatch(stream_t st, int set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk,
                           const Workspace& ws) {
    ParamSet p;
    if (!param_set(set, p)) return -1;
    switch (set) {
    case 512: keygen_run<2, 3>(st, p, n, d, z, ek, dk, ws); break;
    case 768: keygen_run<3, 2>(st, p, n, d, z, ek, dk, ws); break;
    default: keygen_run<4, 2>(st, p, n, d, z, ek, dk, ws); break;
    }
    return 0;
}
inline int encaps_dispatch(stream_t st, int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K,
                           int32_t* mod_status, const Workspace& ws) {
    ParamSet p;
    if (!param_set(set, p)) return -1;
    switch (set) {
    case 512: encaps_run<2, 3, 10, 4>(st, p, n, ek, m, c, K, mod_status, ws); break;
    case 768: encaps_run<3, 2, 10, 4>(st, p, n, ek, m, c, K, mod_status, ws); break;
    default: encaps_run<4, 2, 11, 5>(st, p, n, ek, m, c, K, mod_status, ws); break;
    }
    return 0;
}
// K-PKE on its own: PKE_KeyGen / PKE_Encrypt / PKE_Decrypt
inline int pke_keygen_dispatch(stream_t st, int set, size_t n, const uint8_t* d, uint8_t* ek, uint8_t* dk_pke, const Workspace& ws) {
    return keygen_dispatch(st, set, n, d, nullptr, ek, dk_pke, ws);
}
inline int pke_encrypt_dispatch(stream_t st, int set, size_t n, const uint8_t* ek, const uint8_t* m, const uint8_t* r, uint8_t* c,
                                const Workspace& ws) {
    ParamSet p;
    if (!param_set(set, p) || !r) return -1;
    switch (set) {
    case 512: encaps_run<2, 3, 10, 4>(st, p, n, ek, m, c, nullptr, nullptr, ws, r); break;
    case 768: encaps_run<3, 2, 10, 4>(st, p, n, ek, m, c, nullptr, nullptr, ws, r); break;
    default: encaps_run<4, 2, 11, 5>(st, p, n, ek, m, c, nullptr, nullptr, ws, r); break;
    }
    return 0;
}
inline int pke_decrypt_dispatch(stream_t st, int set, size_t n, const uint8_t* dk_pke, const uint8_t* c, uint8_t* m) {
    ParamSet p;
    if (!param_set(set, p)) return -1;
    if (n == 0) return 0;
    switch (set) {
    case 512: decrypt_launch<2, 10, 4>(st, n, dk_pke, (size_t)(384 * 2), c, m); break;
    case 768: decrypt_launch<3, 10, 4>(st, n, dk_pke, (size_t)(384 * 3), c, m); break;
    default: decrypt_launch<4, 11, 5>(st, n, dk_pke, (size_t)(384 * 4), c, m); break;
    }
    return 0;
}
inline int decaps_dispatch(stream_t st, int set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status,
                           bool hash_check, const Workspace& ws) {
    ParamSet p;
    if (!param_set(set, p)) return -1;
    switch (set) {
    case 512: decaps_run<2, 3, 10, 4>(st, p, n, dk, c, K, status, hash_check, ws); break;
    case 768: decaps_run<3, 2, 10, 4>(st, p, n, dk, c, K, status, hash_check, ws); break;
    default: decaps_run<4, 2, 11, 5>(st, p, n, dk, c, K, status, hash_check, ws); break;
    }
    return 0;
}

// ---- stand-alone primitives ------------------------------------------------------------------------
inline size_t poly_grid(size_t n) {
    size_t g = ceil_div(n, ARITH_WAVES);
    const size_t cap = 256 * 16;   // 256 CUs x a few workgroups, grid-stride for the rest
    return g < cap ? g : cap;
}
// persistent grid of the register NTT (measured with tools/ntt_ubench.hip, profiles/r02_ntt_design.txt: 12288-16384
// workgroups are best, 2^20 polynomials = 5.3 wave-iterations each; below 2048 the tail dominates; MLKEM_NTT_GRID overrides)
inline size_t ntt_grid_cap() {
    static const size_t v = [] {
        const char* e = getenv("MLKEM_NTT_GRID");
        const long long x = e ? atoll(e) : 0;
        return x > 0 ? (size_t)x : (size_t)(256 * 48);
    }();
    return v;
}
// four polynomials per wave and iteration, all in registers (mlkem_rntt.hpp)
inline void ntt_launch(stream_t st, bool inverse, size_t n, const uint16_t* in, uint16_t* out) {
    size_t grid = ceil_div(ceil_div(n, 4), RNTT_WAVES);
    const size_t cap = ntt_grid_cap();
    if (grid > cap) grid = cap;
    if (inverse) launch("k_intt_batch", k_ntt4_batch<true>, grid, WAVE * RNTT_WAVES, st, n, in, out);
    else launch("k_ntt_batch", k_ntt4_batch<false>, grid, WAVE * RNTT_WAVES, st, n, in, out);
}
inline void basemul_launch(stream_t st, size_t n, const uint16_t* a, const uint16_t* b, uint16_t* h) {
    launch("k_basemul_batch", k_basemul_batch, poly_grid(n), WAVE * ARITH_WAVES, st, n, a, b, h);
}
// VectorMultiply over n items of k polynomial pairs each; PolyAddition / PolySubtraction over n_values coefficients
inline int vecmul_launch(stream_t st, int k, size_t n, const uint16_t* u, const uint16_t* v, uint16_t* w) {
    if (k < 1 || k > 4) return -1;
    launch("k_vecmul_batch", k_vecmul_batch, poly_grid(n), WAVE * ARITH_WAVES, st, n, k, u, v, w);
    return 0;
}
inline void poly_addsub_launch(stream_t st, bool sub, size_t n_values, const uint16_t* a, const uint16_t* b, uint16_t* out) {
    size_t grid = ceil_div(ceil_div(n_values, 8), 256);
    if (grid > 256 * 8) grid = 256 * 8;
    if (sub) launch("k_poly_sub", k_poly_addsub<true>, grid, 256u, st, n_values, a, b, out);
    else launch("k_poly_add", k_poly_addsub<false>, grid, 256u, st, n_values, a, b, out);
}
inline int cbd_launch(stream_t st, int eta, size_t n, const uint8_t* bytes, uint16_t* out) {
    if (eta == 2) launch("k_cbd_batch", k_cbd_batch<2>, poly_grid(n), WAVE * ARITH_WAVES, st, n, bytes, out);
    else if (eta == 3) launch("k_cbd_batch", k_cbd_batch<3>, poly_grid(n), WAVE * ARITH_WAVES, st, n, bytes, out);
    else return -1;
    return 0;
}
// Compress+ByteEncode / ByteDecode+Decompress for the d values ML-KEM uses
inline int codec_launch(stream_t st, bool encode, int d, size_t n, const void* in, void* out) {
#define MLKEM_CODEC_CASE(D)                                                                                                    \
    case D:                                                                                                                    \
        if (encode) launch("k_encode_batch", k_encode_batch<D>, poly_grid(n), WAVE * ARITH_WAVES, st, n, (const uint16_t*)in, (uint8_t*)out); \
        else launch("k_decode_batch", k_decode_batch<D>, poly_grid(n), WAVE * ARITH_WAVES, st, n, (const uint8_t*)in, (uint16_t*)out);        \
        return 0;
    switch (d) {
        MLKEM_CODEC_CASE(1) MLKEM_CODEC_CASE(4) MLKEM_CODEC_CASE(5) MLKEM_CODEC_CASE(10) MLKEM_CODEC_CASE(11) MLKEM_CODEC_CASE(12)
    default: return -1;
    }
#undef MLKEM_CODEC_CASE
}
// Compress_d / Decompress_d value by value, any d in 1..12 (in-place allowed)
inline int compress_values_launch(stream_t st, bool decompress, int d, size_t n, const uint16_t* in, uint16_t* out) {
    if (d < 1 || d > 12) return -1;
    size_t grid = ceil_div(ceil_div(n, 8), 256);
    if (grid > 256 * 8) grid = 256 * 8;
    if (grid == 0) return 0;
    if (decompress) launch("k_decompress_values", k_compress_values<true>, grid, 256u, st, n, d, in, out);
    else launch("k_compress_values", k_compress_values<false>, grid, 256u, st, n, d, in, out);
    return 0;
}
// stand-alone SampleNTT over explicit 34-byte seeds: the general kernel in direct mode
// (calls of at most `wave_max` seeds: one sponge per wave, mlkem_small.hpp -- the ml_kem.h shim's SampleNTT is a call of one)
template <int QB = KQ, int CAP = SAMPLE_CAP>
inline void sample_ntt_launch(stream_t st, size_t n, const uint8_t* seeds34, uint16_t* out, size_t wave_max = 0, uint8_t* retries = nullptr) {
    if (n <= wave_max) {
        launch("k_sample_xof", k_sample_ntt_w<QB, CAP>, n, WAVE, st, n, seeds34, out, retries);
        return;
    }
    SampleArgs a{};
    a.n_xof = n; a.rho = seeds34; a.rho_stride = 34; a.K = 0; a.A = out; a.retries = retries;
    a.xof_blocks = (unsigned)ceil_div(n, WAVE);
    launch("k_sample_xof", k_sample<QB, CAP>, (size_t)a.xof_blocks, WAVE, st, a);
}
inline int prf_launch(stream_t st, int eta, size_t n, const uint8_t* in33, uint8_t* out) {
    if (eta != 2 && eta != 3) return -1;
    SampleArgs a{};
    a.n_prf = n; a.r = in33; a.per_item = 0; a.eta1 = eta; a.prf = out; a.prf_stride = 64u * (unsigned)eta;
    launch("k_sample_prf", k_sample<>, ceil_div(n, WAVE), WAVE, st, a);
    return 0;
}
inline int hash_launch(stream_t st, int kind, size_t n, const uint8_t* msg, unsigned len, size_t stride, uint8_t* out) {
    if (kind < 0 || kind > 2 || (stride & 7) || stride < len) return -1;   // rows must start 8-byte aligned
    const size_t grid = ceil_div(n, WAVE);
    if (kind == 0) launch("k_hash_batch", k_hash_batch<0>, grid, WAVE, st, n, msg, len, stride, out);
    else if (kind == 1) launch("k_hash_batch", k_hash_batch<1>, grid, WAVE, st, n, msg, len, stride, out);
    else launch("k_hash_batch", k_hash_batch<2>, grid, WAVE, st, n, msg, len, stride, out);
    return 0;
}

// 4-byte `union byte` cells <-> packed bytes
inline void cells_launch(stream_t st, bool to_bytes, size_t n, const void* in, void* out) {
    size_t grid = ceil_div(ceil_div(n, 16), 256);
    if (grid > 256 * 8) grid = 256 * 8;
    if (grid == 0) grid = 1;
    if (to_bytes) launch("k_cells_to_bytes", k_cells_to_bytes, grid, 256u, st, n, (const uint32_t*)in, (uint8_t*)out);
    else launch("k_bytes_to_cells", k_bytes_to_cells, grid, 256u, st, n, (const uint8_t*)in, (uint32_t*)out);
}

// bare sponge over pre-padded messages (sha3.h front-ends); rate in bytes: 72 / 104 / 136 / 144 / 168
// (calls of at most `wave_max` messages, and every rate that is not one of the five: one sponge per wave, any rate of 1..199 bytes)
inline int sponge_raw_launch(stream_t st, unsigned rate, size_t n, const uint8_t* msg, unsigned nblocks, uint8_t* out, unsigned outlen,
                             size_t out_stride, size_t wave_max = 0) {
    if (nblocks == 0 || out_stride < outlen || rate == 0 || rate >= 200) return -1;
    const bool standard = rate == 72 || rate == 104 || rate == 136 || rate == 144 || rate == 168;
    if (n <= wave_max || !standard) {
        launch("k_sponge_raw", k_sponge_raw_w, n, WAVE, st, n, msg, rate, nblocks, out, outlen, out_stride);
        return 0;
    }
    if (out_stride & 3) return -1;
    const size_t grid = ceil_div(n, WAVE);
    switch (rate) {
    case 72: launch("k_sponge_raw", k_sponge_raw<72>, grid, WAVE, st, n, msg, nblocks, out, outlen, out_stride); break;
    case 104: launch("k_sponge_raw", k_sponge_raw<104>, grid, WAVE, st, n, msg, nblocks, out, outlen, out_stride); break;
    case 136: launch("k_sponge_raw", k_sponge_raw<136>, grid, WAVE, st, n, msg, nblocks, out, outlen, out_stride); break;
    case 144: launch("k_sponge_raw", k_sponge_raw<144>, grid, WAVE, st, n, msg, nblocks, out, outlen, out_stride); break;
    case 168: launch("k_sponge_raw", k_sponge_raw<168>, grid, WAVE, st, n, msg, nblocks, out, outlen, out_stride); break;
    default: return -1;
    }
    return 0;
}

}   // namespace mlkem
