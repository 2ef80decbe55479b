"""GPU tier (-m gpu), round-3 additions: BASELINE configs[4] at its stated shape on the one GPU (2^23 items, 8 members,
item -> member i >> 20), the in-process and per-rank forms of bench.py's N > 1 line, row-level parity for SURVEY 8a rows
a12 / a13 (PolyAddition / PolySubtraction / VectorMultiply), release of the cached host state while calls are in flight,
stream ordering of the device-resident multi-member calls."""
import json
import os
import socket
import subprocess
import sys
import threading
import time

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


def as_u16(t):
    return host(t).view(np.uint16)


# ---- SURVEY 8a rows a12 / a13 as their own entries ----------------------------------------------------------------------
def test_poly_add_sub_golden_and_raw_12bit(eng, torch, oracle, golden_npz):
    """PolyAddition / PolySubtraction (ml_kem.c:580-613) against the goldens the real reference generated (rand_ab_add,
    rand_ab_sub) and, on raw 12-bit inputs with junk in bits 12..15 and a length that is not a multiple of 8, the oracle."""
    a, b = golden_npz["rand_a"], golden_npz["rand_b"]
    da, db = dev(torch, a.view(np.int16)), dev(torch, b.view(np.int16))
    assert (as_u16(eng.poly_add(da, db)) == golden_npz["rand_ab_add"]).all()
    assert (as_u16(eng.poly_sub(da, db)) == golden_npz["rand_ab_sub"]).all()
    na, nb = golden_npz["nc_a"], golden_npz["nc_b"]
    junk = (na | 0xF000).astype(np.uint16)
    want_add = np.stack([oracle.poly_add(na[i], nb[i]) for i in range(na.shape[0])])
    want_sub = np.stack([oracle.poly_sub(na[i], nb[i]) for i in range(na.shape[0])])
    assert (as_u16(eng.poly_add(dev(torch, junk.view(np.int16)), dev(torch, nb.view(np.int16)))) == want_add).all()
    assert (as_u16(eng.poly_sub(dev(torch, junk.view(np.int16)), dev(torch, nb.view(np.int16)))) == want_sub).all()
    # C-ABI directly, ragged value count (scalar tail), in place
    lib = eng.lib
    x = dev(torch, na.reshape(-1)[:1003].copy().view(np.int16))
    y = dev(torch, nb.reshape(-1)[:1003].copy().view(np.int16))
    torch.cuda.synchronize()
    assert lib.mlkem_poly_sub_dev(eng._ctx, 1003, x.data_ptr(), y.data_ptr(), x.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert (as_u16(x) == want_sub.reshape(-1)[:1003]).all()
    x2 = dev(torch, na.reshape(-1)[:1003].copy().view(np.int16))
    torch.cuda.synchronize()
    assert lib.mlkem_poly_add_dev(eng._ctx, 1003, x2.data_ptr(), y.data_ptr(), y.data_ptr(), None) == 0   # out == second operand
    torch.cuda.synchronize()
    assert (as_u16(y) == want_add.reshape(-1)[:1003]).all()
    y = dev(torch, nb.reshape(-1)[:1003].copy().view(np.int16))
    assert lib.mlkem_poly_add_dev(eng._ctx, 8, None, y.data_ptr(), x.data_ptr(), None) == -101
    # the extremes of the 12-bit field
    e = np.zeros((3, 256), np.uint16)
    f = np.zeros((3, 256), np.uint16)
    e[0], f[0] = 0, 4095
    e[1], f[1] = 4095, 4095
    e[2], f[2] = 4095, 0
    for fn, orc in ((eng.poly_add, oracle.poly_add), (eng.poly_sub, oracle.poly_sub)):
        got = as_u16(fn(dev(torch, e.view(np.int16)), dev(torch, f.view(np.int16))))
        assert (got == np.stack([orc(e[i], f[i]) for i in range(3)])).all()


def test_vector_multiply_every_k(pkg, eng, torch, oracle, golden_npz):
    """VectorMultiply (ml_kem.c:618-638) for k = 1..4 against the reference's own composition MultiplyNTTs + PolyAddition
    (oracle), canonical and raw 12-bit operands; k = 1 on the golden pair equals the MultiplyNTTs golden."""
    a, b = golden_npz["rand_a"], golden_npz["rand_b"]
    got = as_u16(eng.vector_multiply(dev(torch, a.view(np.int16)[:, None, :]), dev(torch, b.view(np.int16)[:, None, :])))
    assert (got == golden_npz["rand_ab_mul"]).all()
    rng = np.random.default_rng(618)
    for k in (1, 2, 3, 4):
        for hi in (3329, 4096):
            n = 37
            u = rng.integers(0, hi, (n, k, 256)).astype(np.uint16)
            v = rng.integers(0, hi, (n, k, 256)).astype(np.uint16)
            want = np.zeros((n, 256), np.uint16)
            for i in range(n):
                w = oracle.multiply_ntts(u[i, 0], v[i, 0])
                for j in range(1, k):
                    w = oracle.poly_add(w, oracle.multiply_ntts(u[i, j], v[i, j]))
                want[i] = w
            got = as_u16(eng.vector_multiply(dev(torch, u.view(np.int16)), dev(torch, v.view(np.int16))))
            assert (got == want).all(), (k, hi)
    with pytest.raises(pkg.MLKEMError):
        eng.vector_multiply(torch.zeros((1, 5, 256), dtype=torch.int16), torch.zeros((1, 5, 256), dtype=torch.int16))


# ---- BASELINE configs[4] at its stated shape, on the one GPU --------------------------------------------------------------
def test_config4_shape_2p23_items_8_members(pkg, torch, oracle):
    """configs[4]: ML-KEM-768, batch 2^23 sharded over 8 members, item i -> member i >> 20, no exchange between members.
    All eight members live on device 0 here (each with its own stream and engine context: the in-process form of the shard;
    on the 8-GPU node member r sits on device r).  Checked over the whole batch: K_encaps == K_decaps for every item, status 0,
    one tampered ciphertext per 1024 rejected; byte for byte against the oracle: the items on both sides of every member
    boundary and the two ends of the batch."""
    import bench
    free, _ = torch.cuda.mem_get_info()
    if free < 110 << 30:
        pytest.skip("needs 110 GB of free HBM")
    members, per = 8, 1 << int(os.environ.get("MLKEM_TEST_CFG4_LOG2", "20"))   # the env knob only exists to bisect a failure
    n = members * per
    mm = pkg.MLKEMMulti(768, devices=[0] * members)
    assert mm.ranges(n) == [(r * per, (r + 1) * per) for r in range(members)]          # i -> member i >> 20
    dv = torch.device("cuda", 0)
    d, z, m = ([bench.device_seeds(lbl, r * per, per, dv) for r in range(members)] for lbl in ("mlkem-bench-d", "mlkem-bench-z", "mlkem-bench-m"))
    ek, dk = mm.keygen_dev(d, z)
    c, K = mm.encaps_dev(ek, m)
    Kd, st = mm.decaps_dev(dk, c)
    mm.sync()
    for r in range(members):
        assert torch.equal(K[r], Kd[r]) and int(st[r].abs().max()) == 0, r
    # implicit rejection: one tampered ciphertext per 1024 items, in every member
    idx = torch.arange(0, per, 1024, device=dv)
    ct = []
    for r in range(members):
        t = c[r].clone()
        t[idx, (idx * 13 + r) % mm.c_len] ^= 1 << (r % 8)
        ct.append(t)
    Kt, stt = mm.decaps_dev(dk, ct)
    mm.sync()
    for r in range(members):
        same = (Kt[r] == K[r]).all(dim=1)
        assert not bool(same[idx].any()) and int(same.sum()) == per - idx.numel() and int(stt[r].abs().max()) == 0, r
    # oracle at every member boundary: last item of member r and first item of member r + 1, plus both ends of the batch
    edge = torch.tensor([0, 1, 1024, per - 2, per - 1], device=dv)   # items 0 and 1024 carry a tampered ciphertext
    for r in range(members):
        dh, zh, mh = host(d[r][edge]), host(z[r][edge]), host(m[r][edge])
        ek_o, dk_o = oracle.keygen(768, dh, zh)
        c_o, K_o = oracle.encaps(768, ek_o, mh)
        assert (host(ek[r][edge]) == ek_o).all() and (host(dk[r][edge]) == dk_o).all(), r
        assert (host(c[r][edge]) == c_o).all() and (host(K[r][edge]) == K_o).all(), r
        Kt_o, st_o = oracle.decaps(768, dk_o, host(ct[r][edge]))
        assert (host(Kt[r][edge]) == Kt_o).all() and (st_o == 0).all(), r
        assert (Kt_o[0] != K_o[0]).any() and (Kt_o[2] != K_o[2]).any() and (Kt_o[[1, 3, 4]] == K_o[[1, 3, 4]]).all()
    # the shards are the global batch: member r's first items are the documented expander at global index r * 2^20 + i
    for r in (0, 3, 7):
        want = np.frombuffer(b"".join(bench.expand("mlkem-bench-m", r * per + i) for i in range(4)), np.uint8).reshape(4, 32)
        assert (host(m[r][:4]) == want).all()
    mm.close()
    del ek, dk, c, K, Kd, st, ct, Kt, stt, d, z, m
    torch.cuda.empty_cache()


def test_multi_dev_calls_are_ordered_behind_torch_producers(pkg, torch, oracle):
    """the members enqueue on their own streams.  The wrapper makes those wait for torch's current stream, so
    a device-resident call may follow the torch ops that produce its inputs WITHOUT a synchronize, and outputs dropped
    before sync() stay allocated for the member stream."""
    n = 4096
    mm = pkg.MLKEMMulti(512, devices=[0, 0, 0], chunk_items=512)
    rg = mm.ranges(n)
    base = dev(torch, seeds("ord-d", n, 512))
    torch.cuda.synchronize()
    for trial in range(3):
        # inputs are produced by a chain of torch kernels immediately before the call
        big = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
        for _ in range(20):
            big = big + 1
        ds = [(base[lo:hi] ^ (big[lo:hi] - 20 + trial)).contiguous() for lo, hi in rg]
        zs = [(t ^ 0x5A).contiguous() for t in ds]
        eks, dks = mm.keygen_dev(ds, zs)
        cs, Ks = mm.encaps_dev(eks, zs)
        del eks                                        # dropped before sync(): must not be recycled under the member
        junk = [torch.full((hi - lo, 800), 0xEE, dtype=torch.uint8, device="cuda") for lo, hi in rg]
        Kd, st = mm.decaps_dev(dks, cs)
        mm.sync()
        d_h = host(base) ^ np.uint8(trial)
        ek_o, dk_o = oracle.keygen(512, d_h[:64], d_h[:64] ^ 0x5A)
        c_o, K_o = oracle.encaps(512, ek_o, d_h[:64] ^ 0x5A)
        assert (host(dks[0][:64]) == dk_o).all() and (host(cs[0][:64]) == c_o).all() and (host(Ks[0][:64]) == K_o).all()
        assert all(torch.equal(a, b) for a, b in zip(Ks, Kd)) and all(int(s.abs().max()) == 0 for s in st)
        del junk
    assert len(mm.streams()) == 3
    mm.close()


# ---- release of the cached host state while calls are in flight ------------------------------------------------------------
def test_host_release_races_with_calls_in_flight(pkg, oracle):
    """Two threads keep calling host-pointer entry points (streaming KEM + a primitive) while a third keeps calling
    mlkem_host_release() / mlkem_stream_release(): every call returns 0 with the right bytes (the state is reference-counted;
    a release waits on the per-device locks and an overlapping call rebuilds what it needs)."""
    lib = pkg.load_library()
    stop = threading.Event()
    bad = []
    n = 257
    d, z = seeds("race-d", n, 3), seeds("race-z", n, 3)
    ek_o, dk_o = oracle.keygen(768, d[:16], z[:16])
    f = np.random.default_rng(9).integers(0, 3329, (9, 256)).astype(np.uint16)
    fh_o = oracle.ntt(f)

    def kem_caller():
        for _ in range(25):
            ek, dk = np.zeros((n, 1184), np.uint8), np.zeros((n, 2400), np.uint8)
            rc = lib.mlkem_keygen_stream(768, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data, 64)
            if rc != 0 or not (ek[:16] == ek_o).all() or not (dk[:16] == dk_o).all():
                bad.append(("keygen", rc))

    def prim_caller():
        for _ in range(60):
            fh = np.zeros_like(f)
            rc = lib.mlkem_ntt(9, f.ctypes.data, fh.ctypes.data)
            if rc != 0 or not (fh == fh_o).all():
                bad.append(("ntt", rc))

    def releaser():
        while not stop.is_set():
            lib.mlkem_host_release()
            lib.mlkem_stream_release()

    ts = [threading.Thread(target=kem_caller), threading.Thread(target=prim_caller)]
    rel = threading.Thread(target=releaser)
    rel.start()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop.set()
    rel.join()
    assert not bad, bad[:5]
    lib.mlkem_host_release()


@pytest.mark.parametrize("combine", ("2", "0"))
def test_concurrent_host_threads_are_combined_or_get_engines_of_their_own(combine):
    """Twelve host threads make one- and three-item host-pointer calls at once (what a multi-threaded host of the ml_kem.h shim
    does).  Default: calls of the same operation that arrive while others are in flight are combined into one launch by a leader
    (Combiner, mlkem_capi.hip); MLKEM_HOST_COMBINE=0: every concurrent caller gets an engine of its own (HostState::lanes) or
    queues on one.  Either way every thread's keys, ciphertexts and shared secrets equal the oracle's, with a tampered
    ciphertext per round, while a further thread keeps releasing the cached engines."""
    code = r'''
import sys, threading, time, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import __graft_entry__ as ge
from conftest import seeds
from oracle.loader import Oracle
pkg = ge.load_package(); lib = pkg.load_library(); oracle = Oracle()
bad, stop = [], threading.Event()
def caller(tid):
    n = 1 if tid %% 2 == 0 else 3
    d, z, m = seeds("lane-d%%d" %% tid, n, 5), seeds("lane-z%%d" %% tid, n, 5), seeds("lane-m%%d" %% tid, n, 5)
    ek_o, dk_o = oracle.keygen(768, d, z)
    c_o, K_o = oracle.encaps(768, ek_o, m)
    cb = c_o.copy()
    cb[n - 1, 3 + tid] ^= 0x40
    Kd_o, st_o = oracle.decaps(768, dk_o, cb)
    for rnd in range(30):
        ek, dk = np.zeros((n, 1184), np.uint8), np.zeros((n, 2400), np.uint8)
        c, K, Kd, st = np.zeros((n, 1088), np.uint8), np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
        rc = lib.mlkem_keygen(768, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data)
        rc |= lib.mlkem_encaps(768, n, ek.ctypes.data, m.ctypes.data, c.ctypes.data, K.ctypes.data)
        rc |= lib.mlkem_decaps(768, n, dk.ctypes.data, cb.ctypes.data, Kd.ctypes.data, st.ctypes.data)
        if rc or not ((ek == ek_o).all() and (dk == dk_o).all() and (c == c_o).all() and (K == K_o).all()
                      and (Kd == Kd_o).all() and (st == st_o).all()):
            bad.append((tid, rnd, rc))
def releaser():
    while not stop.is_set():
        lib.mlkem_stream_release()
        time.sleep(0.002)
ts = [threading.Thread(target=caller, args=(t,)) for t in range(12)]
rel = threading.Thread(target=releaser)
rel.start()
[t.start() for t in ts]
[t.join() for t in ts]
stop.set(); rel.join()
lib.mlkem_host_release()
print("bad", bad[:5])
sys.exit(1 if bad else 0)
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MLKEM_HOST_COMBINE=combine), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_small_call_kernels_soak_ten_seconds():
    """tools/soak_small.py for ten seconds: keygen -> encaps -> decaps calls of random sizes 1..896 and random parameter sets on two
    streams, every round checked, under a watchdog that ends the process if a round does not finish within 5 s (the small-call
    kernels hand values over by spin-waiting on counters in LDS: a logic error there shows as a wave that never finishes)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_small.py"), "10"], capture_output=True, text=True, timeout=200, cwd=ROOT)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout + r.stderr[-2000:]


# ---- bench.py: the N > 1 line carries what configs[4] is defined by --------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _one_json_line(r):
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_two_rank_line_carries_per_gpu(tmp_path):
    """`bench.py --gpus 2` as the driver launches it: per_gpu has one entry per rank (own rate, own ms/step, device, clock and
    power sampled on that rank), `correct` is the AND over the ranks, the aggregate is the MAX-elapsed figure."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "8192", "--rehearse"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    j = _one_json_line(r)
    assert j["n_gpus"] == 2 and j["correct"] is True and len(j["per_gpu"]) == 2
    for rank, p in enumerate(j["per_gpu"]):
        assert p["rank"] == rank and p["device"] == 0 and p["correct"] is True and p["value"] > 0 and p["ms_per_step"] > 0
        assert set(p) >= {"rank", "device", "name", "value", "ms_per_step", "sclk_mhz", "socket_w", "correct"}
        assert p["ms_per_step"] <= j["ms_per_step"] * 1.001          # a rank's own time is inside the MAX-over-ranks region
    assert abs(j["value"] - 2 * 8192 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-6


def test_bench_inproc_members_line(tmp_path):
    """`bench.py --inproc --gpus 3`: one process, three members through mlkem_{encaps,decaps}_multi_dev (all on device 0
    here), per-member times from events on the members' own streams."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--inproc", "--gpus", "3", "--steps", "2", "--warmup", "1",
                        "--batch", "8192"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    j = _one_json_line(r)
    assert j["n_gpus"] == 3 and j["correct"] is True and len(j["per_gpu"]) == 3 and j["scaling"] == "weak"
    assert "inproc" in j["config"]["parallelism"] and "REHEARSAL" in j["config"]["parallelism"]
    assert j["config"]["member_devices"] == [0, 0, 0]
    assert all(p["correct"] and p["value"] > 0 for p in j["per_gpu"])
    assert abs(j["value"] - 3 * 8192 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-6


def test_bench_default_line_has_three_cpu_legs(tmp_path):
    """the N = 1 line: cpu_baseline carries reference -O2, reference -O0 (the reference makefile's flags) and the port, each
    with its core count and a byte comparison against the GPU; the roofline is the dominant kernel's by the algorithmic-bytes
    rule with the whole-pass figure beside it.  Small batch, no `also` legs: this checks the plumbing, not the numbers."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "16384", "--no-also"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    j = _one_json_line(r)
    cb = j["cpu_baseline"]
    assert cb["outputs_match_gpu"] is True and cb["cores"] >= 1
    legs = cb["legs"]
    assert "port" in legs and legs["port"]["kind"] == "port" and legs["port"]["outputs_match_gpu"] is True
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libmlkem_ref.so")):
        assert cb["kind"] == "reference" and legs["reference_O2"]["outputs_match_gpu"] is True
        assert legs["reference_O0"]["outputs_match_gpu"] is True and "-g" in legs["reference_O0"]["flags"]
        assert legs["reference_O0"]["per_core"] < legs["reference_O2"]["per_core"] < legs["port"]["per_core"]
    rf = j["roofline"]
    assert rf["dominant_kernel"] and rf["whole_pass"]["frac"] <= rf["frac"] < 1 and len(j["per_gpu"]) == 1


# ---- the re-encryption compare looks at every byte ------------------------------------------------------------------------
@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_every_ciphertext_byte_takes_part_in_the_compare(pkg, torch, oracle, pset):
    """mlkem_kpke2.hpp compares the re-encrypted ciphertext as register pieces (dwords that two or four lanes share are owned by
    one of them).  Item i of the batch carries a ciphertext with one bit flipped in byte i, for EVERY byte of c: all of them must
    come back with the implicit-rejection key (oracle on a spread subset), the untouched batch with K."""
    ekl, dkl, cl = SIZES[pset]
    e = pkg.MLKEM(pset, device=0, chunk_items=512)
    n = cl
    d, z, m = seeds("cmp-d", n, pset), seeds("cmp-z", n, pset), seeds("cmp-m", n, pset)
    ek, dk = e.keygen(dev(torch, d), dev(torch, z))
    c, K = e.encaps(ek, dev(torch, m))
    ch = host(c)
    cb = ch.copy()
    cb[np.arange(n), np.arange(n)] ^= (1 << (np.arange(n) % 8)).astype(np.uint8)
    Kd, st = e.decaps(dk, dev(torch, cb))
    K0, st0 = e.decaps(dk, c)
    torch.cuda.synchronize()
    Kh, Kdh = host(K), host(Kd)
    assert (host(st) == 0).all() and (host(st0) == 0).all() and (host(K0) == Kh).all()
    assert not (Kdh == Kh).all(axis=1).any(), np.nonzero((Kdh == Kh).all(axis=1))[0][:10]
    sub = np.arange(0, n, 53)
    Ko, sto = oracle.decaps(pset, host(dk)[sub], cb[sub])
    assert (Kdh[sub] == Ko).all() and (sto == 0).all()
    e.close()


# ---- calls of one chunk: matrix sampling on the context's side stream ------------------------------------------------------
@pytest.mark.parametrize("env", ({}, {"MLKEM_SMALL_ITEMS": "0"}, {"MLKEM_SMALL_ITEMS": "0", "MLKEM_SIDE_STREAM": "0"},
                                 {"MLKEM_SMALL_ITEMS": "0", "MLKEM_WIDE_HASH_ITEMS": "0"}, {"MLKEM_SMALL_ITEMS": "0", "MLKEM_WIDE_HASH_ITEMS": "100000"},
                                 {"MLKEM_SMALL_ITEMS": "100000", "MLKEM_SMALL_LATENCY_ITEMS": "100000", "MLKEM_SMALL_WIDE_ITEMS": "0"},
                                 {"MLKEM_SMALL_ITEMS": "100000", "MLKEM_SMALL_LATENCY_ITEMS": "100000", "MLKEM_SMALL_WIDE_ITEMS": "100000"},
                                 {"MLKEM_SMALL_ITEMS": "100000", "MLKEM_SMALL_LATENCY_ITEMS": "0", "MLKEM_SMALL_WIDE_ITEMS": "0"}),
                         ids=("default", "side-stream", "one-stream", "lane-sliced-hashes", "one-sponge-per-wave-hashes", "one-workgroup-per-item-8-waves",
                              "one-workgroup-per-item-12-wave-decaps", "one-workgroup-per-item-4-waves"))
@pytest.mark.parametrize("pset,n", ((768, 1000), (512, 3), (1024, 130)))
def test_single_chunk_calls_keep_their_results_with_and_without_the_side_stream(pkg, torch, oracle, env, monkeypatch, pset, n):
    """A call that fits one chunk samples A-hat on the context's side stream while H(ek) / G (encaps) or Decrypt and the
    three sponges (decaps) run on the caller's stream (SideFork, mlkem_pipeline.hpp).  Six rounds of keygen -> encaps ->
    decaps with different data are queued back to back WITHOUT a synchronisation in between: a matrix sampled too early
    (before the previous call's arithmetic has read the scratch) or joined too late would change bytes.  Same bytes with the
    side stream disabled, with either family of hash kernels forced for every size, and with the one-workgroup-per-item
    kernels forced on (in their eight-wave and their four-wave form) and off (defaults: calls of at most Workspace::small_max items run as one launch per operation,
    mlkem_small.hpp; up to Workspace::wide_max items the hash kernels carry one sponge per wavefront, mlkem_wkeccak.hpp; larger
    calls hash with one sponge per lane)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = pkg.MLKEM(pset, device=0, chunk_items=1024)
    outs = []
    for rnd in range(6):
        d, z, m = seeds("sf-d%d" % rnd, n, pset), seeds("sf-z%d" % rnd, n, pset), seeds("sf-m%d" % rnd, n, pset)
        ek, dk = e.keygen(dev(torch, d), dev(torch, z))
        c, K = e.encaps(ek, dev(torch, m))
        cb = c.clone()
        cb[::7, rnd] ^= 0x20                                  # every 7th ciphertext tampered: implicit rejection
        Kd, st = e.decaps(dk, cb)
        outs.append((d, z, m, ek, dk, c, K, cb, Kd, st))
    torch.cuda.synchronize()
    for d, z, m, ek, dk, c, K, cb, Kd, st in outs:
        ek_o, dk_o = oracle.keygen(pset, d, z)
        c_o, K_o = oracle.encaps(pset, ek_o, m)
        Kd_o, st_o = oracle.decaps(pset, dk_o, host(cb))
        assert (host(ek) == ek_o).all() and (host(dk) == dk_o).all() and (host(c) == c_o).all() and (host(K) == K_o).all()
        assert (host(Kd) == Kd_o).all() and (host(st) == st_o).all()
    e.close()


@pytest.mark.parametrize("pset,n", tuple((768, n) for n in (1, 2, 128, 129, 256, 257, 320, 321, 512, 767, 768, 769, 2048, 2049, 3071, 3072, 3073))
                         + ((512, 1536), (512, 1537), (512, 4096), (512, 4097), (1024, 512), (1024, 513), (1024, 4096), (1024, 4097)))
def test_default_path_switches_at_their_boundaries(pkg, torch, oracle, pset, n):
    """The sizes either side of every switch of the default path, one chunk (chunk_items 8192), default environment: Workspace::
    small_wide_max (128 | 129: Decaps with twelve | eight waves per item), small_lat_max (256 | 257: one workgroup of eight | of four
    waves per item; 320, 321, 512: inside the four-wave range), small_max (ML-KEM-768: 768 | 769, ML-KEM-512: 1536 | 1537,
    ML-KEM-1024: 512 | 513: one workgroup per item | batch kernels) and wide_max_k (ML-KEM-768: 3072 | 3073, ML-KEM-512 and 1024:
    4096 | 4097: one sponge per wave + direct sampler | lane-sliced hashes + three-block sampler; 2048 | 2049: the limit of the
    stand-alone primitives, inside the range): keygen -> encaps -> decaps with tampered ciphertexts and one corrupted stored hash."""
    k = {512: 2, 768: 3, 1024: 4}[pset]
    e = pkg.MLKEM(pset, device=0, chunk_items=8192)
    d, z, m = seeds("bd-d", n, pset), seeds("bd-z", n, pset), seeds("bd-m", n, pset)
    ek, dk = e.keygen(dev(torch, d), dev(torch, z))
    c, K = e.encaps(ek, dev(torch, m))
    cb = c.clone()
    cb[::5, 17] ^= 0x10
    dkb = dk.clone()
    dkb[n // 2, 768 * k + 33] ^= 4
    Kd, st = e.decaps(dkb, cb)
    torch.cuda.synchronize()
    sub = np.unique(np.concatenate([np.arange(0, n, max(1, n // 64)), [n // 2, n - 1]]))
    ek_o, dk_o = oracle.keygen(pset, d[sub], z[sub])
    c_o, K_o = oracle.encaps(pset, ek_o, m[sub])
    Kd_o, st_o = oracle.decaps(pset, host(dkb)[sub], host(cb)[sub])
    assert (host(ek)[sub] == ek_o).all() and (host(dk)[sub] == dk_o).all() and (host(c)[sub] == c_o).all() and (host(K)[sub] == K_o).all()
    assert (host(st)[sub] == st_o).all() and host(st)[n // 2] == -5 and (np.delete(host(st), n // 2) == 0).all()
    ok = st_o == 0
    assert (host(Kd)[sub][ok] == Kd_o[ok]).all()
    same = (host(Kd) == host(K)).all(axis=1)
    tam = np.zeros(n, bool)
    tam[::5] = True
    keep = np.arange(n) != n // 2
    assert (same[keep] == ~tam[keep]).all()
    e.close()


def test_sample_ntt_reports_zero_retries_and_the_shim_leaves_real_seeds_alone(pkg, torch, oracle):
    """mlkem_sample_ntt_retries: the polynomial of every seed equals the oracle's and the retry count is 0 (a real SHAKE128 stream
    never exhausts 278 triples), for the lane-sliced form (3000 seeds) and the one-sponge-per-wave form (40 seeds); the counts are
    what the shim adds to the caller's B[32], B[33] as the reference does (ml_kem.c:237-242; the non-zero case runs in the CPU
    tier with a lowered acceptance bound)."""
    import ctypes as C
    lib = pkg.load_library()
    lib.mlkem_sample_ntt_retries.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    for n in (3000, 40):
        s34 = np.concatenate([seeds("rt-s", n, 7), seeds("rt-t", n, 8)[:, :2]], axis=1).astype(np.uint8)
        out, rt = np.zeros((n, 256), np.uint16), np.full(n, 5, np.uint8)
        assert lib.mlkem_sample_ntt_retries(n, s34.ctypes.data, out.ctypes.data, rt.ctypes.data) == 0
        assert not rt.any()
        for i in range(0, n, max(1, n // 200)):
            assert (out[i] == oracle.sample_ntt(s34[i])).all(), i
    out2 = np.zeros((40, 256), np.uint16)
    assert lib.mlkem_sample_ntt_retries(40, s34.ctypes.data, out2.ctypes.data, None) == 0 and (out2 == out).all()


def test_sha3_b_of_the_shim_at_any_byte_aligned_capacity(pkg, oracle):
    """sha3_b(bstr, n, d, c, sfx) through libml_kem.so for capacities other than the SHA-3 / SHAKE ones (the reference's Sponge
    takes any, sha3.c:257-317): rate = (1600 - c) / 8 bytes, 1..199, on the one-sponge-per-wave kernel; against the oracle (pinned
    against the live reference at these rates in the CPU tier).  A capacity that is not a multiple of 8 bits is refused."""
    import ctypes as C
    shim = C.CDLL(pkg.SHIM_PATH)
    shim.sha3_b.restype = C.c_void_p
    shim.sha3_b.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_void_p]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(77)
    for rate in (1, 13, 100, 137, 199, 136):
        for nbits in (0, 8 * rate - 3, 1203):
            bits = rng.integers(0, 2, nbits).astype(np.uint32) | 0xABCD0000      # 4-byte cells, junk above bit 0
            sfx = np.array([1, 1, 1, 1], np.uint32) | 0x55AA0000
            d = 8 * (2 * rate + 3)
            p = shim.sha3_b(bits.ctypes.data, nbits, d, 1600 - 8 * rate, sfx.ctypes.data)
            assert p, (rate, nbits)
            got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), (d,)).copy() & 1
            libc.free(p)
            want = oracle.sponge_bits_sfx(rate, np.ones(4, np.uint8), (bits & 1).astype(np.uint8), d // 8)
            assert (np.packbits(got.astype(np.uint8), bitorder="little") == want).all(), (rate, nbits)
    sfx = np.array([0, 1, 0, 0], np.uint32)
    assert not shim.sha3_b(None, 0, 256, 1600 - 1001, sfx.ctypes.data)             # 1001-bit rate: not byte-aligned


@pytest.mark.parametrize("zero_copy", ("1", "0"))
def test_small_host_pointer_calls_pinned_registered_and_pageable(pkg, torch, oracle, monkeypatch, zero_copy):
    """Host-pointer calls of <= Workspace::small_max items issue no copy commands (MLKEM_ZERO_COPY, default on): the kernels
    read and write pinned host memory -- the caller's buffers where they are pinned (torch) or registered (mlkem_host_register)
    and the operands exceed 16 KB, the engine's pinned staging otherwise.  64 items (ek 75 KB: direct) and 3 items (staged):
    same bytes as the oracle for every mix of pageable / pinned / registered operands, with the path on and off."""
    import subprocess
    code = r'''
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import __graft_entry__ as ge
from conftest import seeds
from oracle.loader import Oracle
pkg = ge.load_package(); lib = pkg.load_library(); orc = Oracle()
for n in (64, 3, 700):      # 700: small kernels, but above the zero-copy size: copy commands
    d, z, m = seeds("zc-d", n, 768), seeds("zc-z", n, 768), seeds("zc-m", n, 768)
    ek, dk = np.zeros((n, 1184), np.uint8), np.zeros((n, 2400), np.uint8)
    assert lib.mlkem_keygen(768, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data) == 0
    ek_o, dk_o = orc.keygen(768, d, z)
    assert (ek == ek_o).all() and (dk == dk_o).all()
    c_o, K_o = orc.encaps(768, ek_o, m)
    ekp, mp = torch.from_numpy(ek).pin_memory(), torch.from_numpy(m).pin_memory()
    cp = torch.zeros((n, 1088), dtype=torch.uint8).pin_memory()
    Kreg = np.zeros((n + 8, 32), np.uint8)
    assert lib.mlkem_host_register(Kreg.ctypes.data, Kreg.nbytes) == 0
    assert lib.mlkem_encaps(768, n, ekp.data_ptr(), mp.data_ptr(), cp.data_ptr(), Kreg.ctypes.data) == 0          # all pinned / registered
    staged = lib.mlkem_stream_last_staged()
    assert (cp.numpy() == c_o).all() and (Kreg[:n] == K_o).all() and not Kreg[n:].any(), n
    c2, K2 = np.zeros((n, 1088), np.uint8), np.zeros((n, 32), np.uint8)
    assert lib.mlkem_encaps(768, n, ekp.data_ptr(), m.ctypes.data, c2.ctypes.data, Kreg.ctypes.data) == 0           # mixed
    assert (c2 == c_o).all() and (Kreg[:n] == K_o).all()
    cb = c_o.copy(); cb[n - 1, 7] ^= 1
    dkp = torch.from_numpy(dk).pin_memory()
    Kd, st = np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
    assert lib.mlkem_decaps(768, n, dkp.data_ptr(), cb.ctypes.data, Kd.ctypes.data, st.ctypes.data) == 0
    Kd_o, st_o = orc.decaps(768, dk_o, cb)
    assert (Kd == Kd_o).all() and (st == st_o).all() and (Kd[n - 1] != K_o[n - 1]).any()
    assert lib.mlkem_host_unregister(Kreg.ctypes.data) == 0
    print("n=%%d staged mask of the all-pinned call: %%d" %% (n, staged))
''' % (ROOT, ROOT)
    env = dict(os.environ, MLKEM_ZERO_COPY=zero_copy)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("n=")]
    assert lines[0].endswith(": 0") and lines[1].endswith(": 15"), lines     # 64 items: in place ; 3 items (<= 16 KB per operand): always staged
