"""GPU tier (-m gpu), round-2 additions: on-device exhaustive arithmetic sweeps, the general Compress / Decompress entry,
sha3_b with caller-chosen suffix bits, in-process sharding over several engine contexts (BASELINE configs[4] rehearsed on
one GPU), per-device host state, the streaming front-end with pinned caller buffers, output validation."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import __graft_entry__ as ge
from conftest import seeds
from oracle.loader import SIZES

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tier needs a HIP device"
    return torch


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.load_library()
    return p


@pytest.fixture(scope="module")
def eng(pkg, torch):
    e = pkg.MLKEM(768, device=0, chunk_items=4096)
    yield e
    e.close()


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


# ---- the fp32 exactness argument, proven on gfx950 itself -------------------------------------------------------------
def test_fp32_arithmetic_exhaustive_on_device(pkg, eng):
    """fred over |x| <= 2^24; fmulmod / fmulmod_shoup over 258 twiddles x |b| <= 10082; compress_f<1..11>
    over |x| <= 4095; cbd_eval_f<2> / <3> over all 2^16 / 2^24 lane inputs; the base-case multiply-accumulate at the corners
    of its bound over every (a, y); canonicalisation over |x| <= 2^20 -- each against integer `% q` arithmetic computed on
    the device (ml_kem.c:83-97, :253-275, :287-442).  Zero violations."""
    assert pkg.load_library().mlkem_selftest_count() == 9
    assert eng.selftest() == [0] * 9


def test_selftest_counter_actually_counts(pkg, eng):
    """the sweeps are not vacuous: an out-of-range sweep index is rejected, and the violation counter starts from a
    poisoned host value that the call must overwrite"""
    lib = pkg.load_library()
    v = C.c_ulonglong(777)
    assert lib.mlkem_selftest(eng._ctx, 99, C.byref(v)) == -101 and v.value == 777
    assert lib.mlkem_selftest(eng._ctx, 3, C.byref(v)) == 0 and v.value == 0


# ---- the register NTT (four polynomials per wave, DPP cross-lane butterflies) on the real lanes -------------------------
def test_register_ntt_extremes_and_reference_behaviour_above_q(pkg, eng, torch, oracle):
    """mlkem_rntt.hpp on gfx950: v_fmac_f32_dpp (quad_perm, row_half_mirror, row_ror) only exists on the device, so the layouts are
    checked here against the oracle on inputs built to hit every lane / register position and the lazy bounds: unit
    vectors at every coefficient index, all-maximum polynomials, alternating extremes, random canonical and random raw
    12-bit polynomials (forward: the reference's non-modular arithmetic above q is reproduced exactly, e.g. NTT of
    4095 x^0 has 4095 at index 254; inverse: inputs >= q are reduced first, the reference's result there is
    compiler-dependent), every batch size mod 4."""
    rng = np.random.default_rng(77)
    unit = np.zeros((256, 256), np.uint16)
    unit[np.arange(256), np.arange(256)] = 3328
    ext = np.zeros((8, 256), np.uint16)
    ext[0] = 3328; ext[1] = 4095; ext[2, ::2] = 3328; ext[3, 1::2] = 4095
    for b in range(4):
        ext[4 + b] = np.where(np.arange(256) & (1 << (2 * b + 1)), 3328, 0)
    one = np.zeros((1, 256), np.uint16)
    one[0, 0] = 4095
    cases = [unit, unit[:255], unit[:254], unit[:253], ext, one, rng.integers(0, 3329, (1001, 256)).astype(np.uint16),
             rng.integers(0, 4096, (1000, 256)).astype(np.uint16)]
    for a in cases:
        da = dev(torch, a.view(np.int16))
        want_f = oracle.ntt(a)
        want_i = oracle.intt(a % 3329)
        assert (host(eng.ntt(da)).view(np.uint16) == want_f).all(), a.shape
        assert (host(eng.intt(da)).view(np.uint16) == want_i).all(), a.shape
    assert oracle.ntt(one)[0, 254] == 4095
    f = rng.integers(0, 3329, (513, 256)).astype(np.uint16)
    fh = eng.ntt(dev(torch, f.view(np.int16)))
    assert (host(eng.intt(fh)).view(np.uint16) == f).all()


# ---- Compress / Decompress for every d (Test_Archive/CompressDecompress_test04.c) ------------------------------------
def test_compress_decompress_every_d_whole_field(pkg, eng, torch, golden_npz, oracle):
    lib = pkg.load_library()
    x = np.arange(4096, dtype=np.uint16)
    junk = (x | ((x.astype(np.uint32) * 7919) & 0xF000).astype(np.uint16)).astype(np.uint16)   # bits 12..15 must be ignored
    for d in range(1, 13):
        got_c = host(eng.compress(dev(torch, junk.view(np.int16)), d)).view(np.uint16)
        got_d = host(eng.decompress(dev(torch, junk.view(np.int16)), d)).view(np.uint16)
        assert (got_c == golden_npz["g4_compress_full"][d - 1]).all(), d
        assert (got_d == golden_npz["g4_decompress_full"][d - 1]).all(), d
        # test04's property: Compress(Decompress(y, d), d) == y for y < 2^d
        y = np.arange(1 << d, dtype=np.uint16)
        back = host(eng.compress(eng.decompress(dev(torch, y.view(np.int16)), d), d)).view(np.uint16)
        assert (back == y).all(), d
    # host-pointer entry, ragged length
    xs = np.random.default_rng(4).integers(0, 4096, 1003).astype(np.uint16)
    out = np.zeros_like(xs)
    assert lib.mlkem_compress(10, xs.size, xs.ctypes.data, out.ctypes.data) == 0
    assert out.tolist() == [oracle.compress(int(v), 10) for v in xs]
    assert lib.mlkem_decompress(5, xs.size, xs.ctypes.data, out.ctypes.data) == 0
    assert out.tolist() == [oracle.decompress(int(v), 5) for v in xs]
    assert lib.mlkem_compress(0, 1, xs.ctypes.data, out.ctypes.data) == -101
    assert lib.mlkem_compress(13, 1, xs.ctypes.data, out.ctypes.data) == -101


# ---- sha3_b: caller's suffix bits verbatim (sha3.c:414-429) -----------------------------------------------------------
def test_sha3_b_suffix_bits_verbatim_through_shim_and_binding(pkg, eng, golden, oracle):
    shim = C.CDLL(pkg.SHIM_PATH)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    cell = C.c_uint
    shim.sha3_b.restype = C.POINTER(cell)
    shim.sha3_b.argtypes = [C.POINTER(cell), C.c_uint, C.c_uint, C.c_uint, C.POINTER(cell)]
    for g in golden["G9_sha3_suffix"]:
        bits = [int(ch) for ch in g["msg_bits"]]
        want = bytes.fromhex(g["out"])
        cells = (cell * max(1, len(bits)))(*[b | 0x5A5A0000 for b in bits])
        sfx = (cell * 4)(*[b | 0xFFFFFF00 for b in g["sfx"]])   # only bit 0 of a suffix cell counts
        r = shim.sha3_b(cells, len(bits), 8 * len(want), g["cap"], sfx)
        assert r, g
        got = np.packbits(np.array([r[i] & 1 for i in range(8 * len(want))], np.uint8), bitorder="little")
        libc.free(r)
        assert bytes(got) == want, (g["sfx"], len(bits))
        nsfx = 4 if g["sfx"][2] == 1 else 2
        got2 = host(eng.sha3_bits([bits, bits, bits], None, g["rate_bytes"], len(want), suffix=g["sfx"][:nsfx]))
        assert bytes(got2[2]) == want
    # RawSHAKE128 of a byte string against the oracle's bit-granular sponge
    msg = np.unpackbits(np.frombuffer(b"rawshake", np.uint8), bitorder="little")
    want = oracle.sponge_bits_sfx(168, [1, 1], msg, 64)
    assert (host(eng.sha3_bits([msg], None, 168, 64, suffix=(1, 1)))[0] == want).all()


# ---- in-process sharding: BASELINE configs[4] rehearsed with several contexts on device 0 ------------------------------
@pytest.mark.parametrize("pset,members", ((768, 4), (512, 3), (1024, 2)))
def test_multi_member_shards_equal_single_context(pkg, torch, oracle, pset, members):
    """A 4-member in-process shard (every member on device 0, each with its own host thread, streams, engine context and
    staging) reproduces the single-context bytes: host-resident and device-resident forms, ragged split."""
    ekl, dkl, cl = SIZES[pset]
    n = 1003   # not divisible by the member count
    d, z, m = seeds("multi-d", n, pset), seeds("multi-z", n, pset), seeds("multi-m", n, pset)
    one = pkg.MLKEM(pset, device=0, chunk_items=256)
    ek1, dk1 = one.keygen(dev(torch, d), dev(torch, z))
    c1, K1 = one.encaps(ek1, dev(torch, m))
    cb = host(c1).copy()
    cb[::50, 11] ^= 0x08
    dkb = host(dk1).copy()
    dkb[7, dkl - 40] ^= 1
    Kd1, st1 = one.decaps(dev(torch, dkb), dev(torch, cb))
    torch.cuda.synchronize()
    mm = pkg.MLKEMMulti(pset, devices=[0] * members, chunk_items=256)
    assert mm.ranges(n)[0][0] == 0 and mm.ranges(n)[-1][1] == n
    # host-resident batch, 64-item streaming chunks (several chunks per member, all three buffer sets in use)
    ek, dk = mm.keygen(d, z, chunk_items=64)
    c, K = mm.encaps(ek, m, chunk_items=64)
    Kd, st = mm.decaps(dkb, cb, chunk_items=64)
    assert (ek == host(ek1)).all() and (dk == host(dk1)).all() and (c == host(c1)).all() and (K == host(K1)).all()
    assert (Kd == host(Kd1)).all() and (st == host(st1)).all() and st[7] == -5 and (np.delete(st, 7) == 0).all()
    # oracle on a subset straddling the shard boundaries
    idx = sorted({0, n - 1} | {b for a, b in mm.ranges(n)[:-1]} | {b - 1 for a, b in mm.ranges(n)[:-1]})
    ek_o, dk_o = oracle.keygen(pset, d[idx], z[idx])
    c_o, K_o = oracle.encaps(pset, ek_o, m[idx])
    assert (ek[idx] == ek_o).all() and (dk[idx] == dk_o).all() and (c[idx] == c_o).all() and (K[idx] == K_o).all()
    # device-resident shards
    rg = mm.ranges(n)
    ds, zs, ms = ([dev(torch, a[lo:hi]) for lo, hi in rg] for a in (d, z, m))
    torch.cuda.synchronize()
    eks, dks = mm.keygen_dev(ds, zs)
    cs, Ks = mm.encaps_dev(eks, ms)          # same member streams: ordered behind keygen
    Kds, sts = mm.decaps_dev(dks, cs)
    mm.sync()
    assert (np.concatenate([host(t) for t in eks]) == ek).all() and (np.concatenate([host(t) for t in dks]) == dk).all()
    assert (np.concatenate([host(t) for t in cs]) == c).all() and (np.concatenate([host(t) for t in Ks]) == K).all()
    assert (np.concatenate([host(t) for t in Kds]) == K).all() and (np.concatenate([host(t) for t in sts]) == 0).all()
    with pytest.raises(pkg.MLKEMError):
        mm.encaps_dev(eks[:-1], ms[:-1])     # one shard per member
    with pytest.raises(pkg.MLKEMError):
        mm.encaps_dev([t.cpu() for t in eks], ms)
    mm.close()
    one.close()


def test_multi_empty_and_tiny_batches(pkg, torch, oracle):
    mm = pkg.MLKEMMulti(768, devices=[0, 0, 0])
    d, z, m = seeds("mt-d", 2, 1), seeds("mt-z", 2, 1), seeds("mt-m", 2, 1)
    ek, dk = mm.keygen(d, z)                 # 2 items over 3 members: one member gets nothing
    c, K = mm.encaps(ek, m)
    ek_o, dk_o = oracle.keygen(768, d, z)
    c_o, K_o = oracle.encaps(768, ek_o, m)
    assert (ek == ek_o).all() and (dk == dk_o).all() and (c == c_o).all() and (K == K_o).all()
    e0 = np.zeros((0, 32), np.uint8)
    ek0, dk0 = mm.keygen(e0, e0)
    assert ek0.shape == (0, 1184) and dk0.shape == (0, 2400)
    mm.close()


# ---- per-device host state -----------------------------------------------------------------------------------------
def test_host_state_release_and_reuse(pkg, oracle):
    lib = pkg.load_library()
    f = np.random.default_rng(3).integers(0, 3329, (5, 256)).astype(np.uint16)
    fh = np.zeros_like(f)
    assert lib.mlkem_ntt(5, f.ctypes.data, fh.ctypes.data) == 0 and (fh == oracle.ntt(f)).all()
    n = 40
    d, z = seeds("hs-d", n, 1), seeds("hs-z", n, 1)
    ek, dk = np.zeros((n, 1184), np.uint8), np.zeros((n, 2400), np.uint8)
    assert lib.mlkem_keygen(768, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data) == 0
    lib.mlkem_host_release()                  # wipes + frees the primitive context and the streaming engine
    lib.mlkem_host_release()                  # idempotent
    fh2 = np.zeros_like(f)
    assert lib.mlkem_ntt(5, f.ctypes.data, fh2.ctypes.data) == 0 and (fh2 == fh).all()    # recreated on demand
    ek2, dk2 = np.zeros_like(ek), np.zeros_like(dk)
    assert lib.mlkem_keygen(768, n, d.ctypes.data, z.ctypes.data, ek2.ctypes.data, dk2.ctypes.data) == 0
    assert (ek2 == ek).all() and (dk2 == dk).all()
    ek_o, dk_o = oracle.keygen(768, d, z)
    assert (ek == ek_o).all() and (dk == dk_o).all()
    lib.mlkem_host_release()


def test_host_calls_from_two_threads_do_not_interfere(pkg, oracle):
    """two host threads through the host-pointer API at the same time (same device here; the state is keyed per device and
    locked per engine): both get the right bytes"""
    import threading
    lib = pkg.load_library()
    out = {}

    def work(tag, pset, n):
        ekl, dkl, cl = SIZES[pset]
        d, z, m = seeds(tag + "d", n, pset), seeds(tag + "z", n, pset), seeds(tag + "m", n, pset)
        ek, dk = np.zeros((n, ekl), np.uint8), np.zeros((n, dkl), np.uint8)
        c, K = np.zeros((n, cl), np.uint8), np.zeros((n, 32), np.uint8)
        rc1 = lib.mlkem_keygen_stream(pset, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data, 50)
        rc2 = lib.mlkem_encaps_stream(pset, n, ek.ctypes.data, m.ctypes.data, c.ctypes.data, K.ctypes.data, 50)
        out[tag] = (rc1, rc2, d, z, m, ek, dk, c, K)

    ts = [threading.Thread(target=work, args=("ta", 768, 333)), threading.Thread(target=work, args=("tb", 512, 207))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for tag, pset in (("ta", 768), ("tb", 512)):
        rc1, rc2, d, z, m, ek, dk, c, K = out[tag]
        assert rc1 == 0 and rc2 == 0
        ek_o, dk_o = oracle.keygen(pset, d, z)
        c_o, K_o = oracle.encaps(pset, ek_o, m)
        assert (ek == ek_o).all() and (dk == dk_o).all() and (c == c_o).all() and (K == K_o).all()
    lib.mlkem_stream_release()


# ---- streaming front-end: pinned caller buffers go to the DMA engines directly --------------------------------------
def test_streaming_pinned_and_pageable_buffers_give_equal_bytes(pkg, torch, oracle):
    lib = pkg.load_library()
    n = 3000
    d, z, m = seeds("pin-d", n, 2), seeds("pin-z", n, 2), seeds("pin-m", n, 2)
    ek_o, dk_o = oracle.keygen(768, d[:64], z[:64])
    # pageable path
    ek, dk = np.zeros((n, 1184), np.uint8), np.zeros((n, 2400), np.uint8)
    assert lib.mlkem_keygen_stream(768, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data, 256) == 0
    assert (ek[:64] == ek_o).all() and (dk[:64] == dk_o).all()
    # pinned path: torch pinned tensors and an mlkem_host_register'ed numpy array, mixed with pageable operands
    ekp = torch.from_numpy(ek).pin_memory()
    mp = torch.from_numpy(m).pin_memory()
    cp = torch.empty((n, 1088), dtype=torch.uint8).pin_memory()
    Kreg = np.zeros((n + 64, 32), np.uint8)   # registered numpy buffer (page-aligned registration is the runtime's business)
    assert lib.mlkem_host_register(Kreg.ctypes.data, Kreg.nbytes) == 0
    try:
        assert lib.mlkem_encaps_stream(768, n, ekp.data_ptr(), mp.data_ptr(), cp.data_ptr(), Kreg.ctypes.data, 256) == 0
        assert lib.mlkem_stream_last_staged() == 0, "a pinned or registered operand was silently staged like pageable memory"
        c, K = np.zeros((n, 1088), np.uint8), np.zeros((n, 32), np.uint8)
        assert lib.mlkem_encaps_stream(768, n, ek.ctypes.data, m.ctypes.data, c.ctypes.data, K.ctypes.data, 256) == 0
        assert lib.mlkem_stream_last_staged() == 0b1111
        assert lib.mlkem_encaps_stream(768, n, ekp.data_ptr(), m.ctypes.data, cp.data_ptr(), K.ctypes.data, 256) == 0
        assert lib.mlkem_stream_last_staged() == 0b1010   # mixed: m and K pageable
        assert (cp.numpy() == c).all() and (Kreg[:n] == K).all() and not Kreg[n:].any()
        c_o, K_o = oracle.encaps(768, ek_o, m[:64])
        assert (c[:64] == c_o).all() and (K[:64] == K_o).all()
        # decaps: pinned inputs, pageable outputs
        dkp = torch.from_numpy(dk).pin_memory()
        Kd, st = np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
        assert lib.mlkem_decaps_stream(768, n, dkp.data_ptr(), cp.data_ptr(), Kd.ctypes.data, st.ctypes.data, 500) == 0
        assert (Kd == K).all() and (st == 0).all()
    finally:
        assert lib.mlkem_host_unregister(Kreg.ctypes.data) == 0
    lib.mlkem_stream_release()


# ---- output validation on the device --------------------------------------------------------------------------------
def test_caller_outputs_are_validated_on_gpu(pkg, eng, torch):
    n = 8
    d = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    for bad in (torch.empty((n, 1184), dtype=torch.uint8),                              # host tensor
                torch.empty((n - 1, 1184), dtype=torch.uint8, device="cuda"),           # too small
                torch.empty((n, 1184), dtype=torch.int8, device="cuda"),                # wrong dtype
                torch.empty((n, 2368), dtype=torch.uint8, device="cuda")[:, ::2]):      # strided
        with pytest.raises(pkg.MLKEMError):
            eng.keygen(d, d, ek=bad)
    ek, dk = eng.keygen(d, d)
    c, K = eng.encaps(ek, d)
    with pytest.raises(pkg.MLKEMError):
        eng.decaps(dk, c, status=torch.empty(n, dtype=torch.int64, device="cuda"))
    with pytest.raises(pkg.MLKEMError):
        eng.encaps(ek, d, K=torch.empty((n, 16), dtype=torch.uint8, device="cuda"))
    K2, st = eng.decaps(dk, c, K=torch.empty((n, 32), dtype=torch.uint8, device="cuda"),
                        status=torch.empty(n, dtype=torch.int32, device="cuda"))
    assert torch.equal(K2, K) and int(st.abs().max()) == 0


# ---- bench.py --gpus 2 rehearsed on the one GPU ------------------------------------------------------------------------
def test_bench_two_rank_rehearsal_prints_a_valid_line(tmp_path):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank); with one GPU
    visible the ranks share it and synchronise over gloo.  Small batch: this checks the N > 1 path end to end on the HIP
    engine (sharded seeds, barrier, max-over-ranks, rank-0 line), not performance."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "8192", "--rehearse"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["correct"] is True and j["scaling"] == "weak" and j["steps"] == 2
    assert "REHEARSAL" in j["config"]["parallelism"] and j["value"] > 0 and j["unit"] == "pairs/s"
    assert abs(j["value"] - 2 * 8192 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-6   # whole-job aggregate over both ranks


def test_bench_rccl_group_with_one_rank(tmp_path):
    """The RCCL leg of the N > 1 path (process group bound to the rank's device, barrier and MAX all-reduce on the device)
    on a one-GPU box: torch.distributed.run with a single rank makes bench.py initialise the group exactly as it does for
    one rank per GPU on the 8-GPU node."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_TRACE_DIST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--batch", "8192", "--no-cpu", "--no-also"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["correct"] is True and "REHEARSAL" not in j["config"]["parallelism"]
    assert "dist backend nccl" in r.stderr   # the group really was RCCL


def test_batch_whose_buffers_cross_4GiB(pkg, torch, oracle):
    """A 288 GB device holds batches whose buffers are larger than 4 GiB (ML-KEM-1024 at 3 x 2^20 + 5 items: dk 10 GB, ek and c
    4.9 GB each), so every byte offset must be 64-bit: K_encaps == K_decaps for all items, tampered ciphertexts rejected, and
    the items that straddle each buffer's 4 GiB line (and the ragged tail) are compared with the oracle."""
    pset, n = 1024, 3 * (1 << 20) + 5
    ekl, dkl, cl = SIZES[pset]
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip("needs 40 GB of free HBM")
    e = pkg.MLKEM(pset, device=0)
    g = torch.Generator(device="cuda").manual_seed(4096)
    d, z, m = (torch.randint(0, 256, (n, 32), generator=g, device="cuda", dtype=torch.uint8) for _ in range(3))
    ek, dk = e.keygen(d, z)
    c, K = e.encaps(ek, m)
    assert dk.numel() > 1 << 32 and ek.numel() > 1 << 32 and c.numel() > 1 << 32
    ct = c.clone()
    idx = torch.arange(0, n, 4096, device="cuda")
    ct[idx, (idx * 11) % cl] ^= 4
    Kd, st = e.decaps(dk, ct)
    same = (Kd == K).all(dim=1)
    assert int(st.abs().max()) == 0 and not bool(same[idx].any()) and int(same.sum()) == n - idx.numel()
    cross = sorted({(1 << 32) // L + k for L in (ekl, dkl, cl) for k in (-1, 0, 1)} | {0, n - 2, n - 1})
    sub = torch.tensor(cross, device="cuda")
    ek_o, dk_o = oracle.keygen(pset, host(d[sub]), host(z[sub]))
    c_o, K_o = oracle.encaps(pset, ek_o, host(m[sub]))
    assert (host(ek[sub]) == ek_o).all() and (host(dk[sub]) == dk_o).all()
    assert (host(c[sub]) == c_o).all() and (host(K[sub]) == K_o).all()
    Kd_o, st_o = oracle.decaps(pset, dk_o, host(ct[sub]))
    assert (host(Kd[sub]) == Kd_o).all() and (st_o == 0).all()
    del ek, dk, c, ct, K, Kd
    e.close()
    torch.cuda.empty_cache()


def test_ntt_buffers_beyond_4GiB(pkg, eng, torch, oracle):
    """2^23 + 3 polynomials = 4.3 GB per buffer: round trip over the whole batch, oracle on the polynomials around the 4 GiB
    line and on the ragged last quad."""
    n = (1 << 23) + 3
    free, _ = torch.cuda.mem_get_info()
    if free < 20 << 30:
        pytest.skip("needs 20 GB of free HBM")
    g = torch.Generator(device="cuda").manual_seed(23)
    f = torch.randint(0, 3329, (n, 256), generator=g, device="cuda", dtype=torch.int16)
    fh = eng.ntt(f)
    back = eng.intt(fh)
    assert torch.equal(back, f)
    line = (1 << 32) // 512
    sub = torch.tensor([0, line - 2, line - 1, line, line + 1, n - 4, n - 3, n - 2, n - 1], device="cuda")
    assert (host(fh[sub]).view(np.uint16) == oracle.ntt(host(f[sub]).view(np.uint16))).all()
    del f, fh, back
    torch.cuda.empty_cache()


@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_decrypt_four_items_per_wave_random_bytes(pkg, torch, oracle, pset):
    """mlkem_kpke4.hpp on the device (the DPP butterflies only exist there): K-PKE.Decrypt of random ciphertext BYTES under
    random key bytes (raw 12-bit coefficients up to 4095, F3) — every field value and piece alignment, every batch size
    mod 4 — against the oracle."""
    ekl, dkl, cl = SIZES[pset]
    k = {512: 2, 768: 3, 1024: 4}[pset]
    e = pkg.MLKEM(pset, device=0, chunk_items=256)
    rng = np.random.default_rng(pset + 4)
    for n in (1, 2, 3, 4, 5, 1003):
        c = rng.integers(0, 256, (n, cl)).astype(np.uint8)
        dkp = rng.integers(0, 256, (n, 384 * k)).astype(np.uint8)
        m = host(e.PKE_Decrypt(dev(torch, dkp), dev(torch, c)))
        idx = range(n) if n < 10 else list(range(0, n, 37)) + [n - 3, n - 2, n - 1]
        for i in idx:
            assert (m[i] == oracle.pke_decrypt(pset, dkp[i], c[i])).all(), (n, i)
    e.close()
