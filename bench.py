#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X:
ML-KEM-768 encaps+decaps pairs per second at batch 2^20 (BASELINE configs[2]), inputs resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload kem768|kem1024|kem512|ntt] [--rehearse]

One "step" = one pass of the hot path over one batch: Encaps_internal over 2^20 (ek, m) followed by KEM_Decaps
(hash check included, as the reference's public API does) over the 2^20 (dk, c) it produced.  Keys come from the
engine's own batch KeyGen on seeds d_i, z_i, m_i = SHAKE128(label || LE64(i) || LE64(0xC0FFEE))[:32] (untimed).
N > 1: one process per GPU (torchrun), every rank runs the same per-GPU batch on its own item range (weak
scaling, no collective in the data path); the timed region is bracketed by barrier + synchronize and the MAX
over ranks is used.  Rank 0 prints ONE JSON line.  When the node has fewer GPUs than ranks (or with --rehearse) the
ranks share the visible GPU(s) and the barrier runs over gloo instead of RCCL: a rehearsal of the N-rank path on a
one-GPU box (the line says so in config.parallelism; its value is not a scaling result).

At N = 1 the line also carries `also`: BASELINE configs[1] (NTT-only) and configs[3] (ML-KEM-1024 KeyGen+Encaps+Decaps)
measured in the same process right after the headline, each with value / ms_per_step / correct / roofline.

Extra objects on the line:
  roofline      whole-pass algorithmic bytes (5856 B per pair) / pass time vs the 8 TB/s HBM peak, the
                dominant kernel by HIP-event time, the integer-VALU view that actually binds this path, and
                `clock_power`: shader clock and socket power sampled with rocm-smi while the same steps keep running
                (the path runs into the socket power limit; nominal clock 2400 MHz)
  cpu_baseline  the real reference (oracle/_ref, `kind: reference`) or the oracle restatement (`kind: port`) timed
                on this box's host cores on a bounded sample of the same items, outputs cross-checked against the GPU
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD-32 x 2.4 GHz (one 32-bit integer op per lane-clock)
BENCH_SEED = 0xC0FFEE
ALGO_BYTES = {"kem768": 5856, "kem1024": 12768, "kem512": 6560, "ntt": 2048,
              "kem768_shared": 32 + 1088 + 32 + 1088 + 32}   # per pair: m, c, K (encaps) + c, K (decaps); the one key is read once   # SURVEY 8d / BASELINE.md section 4
# 32-bit VALU lane-operations per unit, from the ISA of this build
KECCAK_PERM_LANE_OPS = 24 * 180
TIMING_PASSES = 3   # passes of the per-kernel HIP-event timing leg


def expand(label, i):
    return hashlib.shake_128(label.encode() + int(i).to_bytes(8, "little") + BENCH_SEED.to_bytes(8, "little")).digest(32)


def device_seeds(label, start, n, device):
    """Per-item 32-byte seeds.  Item i of the global batch always gets the same seed regardless of sharding: the
    first 4096 items of a rank use the documented SHAKE128 expander exactly; the bulk is filled by a counter-mode
    torch generator keyed by (label, start) — synthetic data, generated on the device."""
    g = torch.Generator(device=device).manual_seed((int.from_bytes(hashlib.sha256(label.encode()).digest()[:6], "little") + start) & (2**62 - 1))
    out = torch.randint(0, 256, (n, 32), generator=g, device=device, dtype=torch.uint8)
    head = min(n, 4096)
    exact = np.frombuffer(b"".join(expand(label, start + i) for i in range(head)), np.uint8).reshape(head, 32)
    out[:head] = torch.from_numpy(exact.copy()).to(device)
    return out


SHARED_GPU = False   # set by dist_setup: ranks share a GPU (rehearsal), the barrier runs over gloo


def dist_setup(n_gpus, rehearse=False):
    """Returns (rank, world, device index).  One rank per GPU over RCCL when the node has a GPU per rank; otherwise (or
    with rehearse=True) the ranks are dealt round-robin over the visible GPUs and synchronise over gloo."""
    global SHARED_GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = world > 1 or all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_PORT"))
    if launched:
        # also for a single rank started by torch.distributed.run: the RCCL group, barrier and max-reduce of the N > 1
        # path then run on a one-GPU box as well (tests/test_gpu_round2.py)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()   # counting devices does not initialise the GPU
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        SHARED_GPU = bool(rehearse) or (0 < ndev < local_world)
        if SHARED_GPU:
            local = local % max(ndev, 1)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        if os.environ.get("BENCH_TRACE_DIST"):
            print("dist backend %s, rank %d of %d, device %d" % (dist.get_backend(), rank, world, local), file=sys.stderr)
    elif n_gpus > 1:
        raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    return rank, world, local


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def barrier(world):
    dist = _dist()
    if dist is not None:
        dist.barrier()


def max_over_ranks(x, world, device):
    dist = _dist()
    if dist is None:
        return x
    on_cpu = SHARED_GPU or dist.get_backend() == "gloo"
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if on_cpu else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def host_cores():
    """Host cores this process can really use: the affinity mask, capped by the cgroup CPU quota when there is one (a GPU
    box hands a job a share of its CPUs through cpu.max while the affinity mask still shows every core)."""
    n = max(1, len(os.sched_getaffinity(0)))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def clock_power_probe(step, device, seconds=1.2):
    """Shader clock and socket power while the workload runs (untimed, after the timed region): the hot path runs into
    the socket power limit, and the line should carry the evidence.  Read from the amdgpu hwmon files of the device
    (freq1_input = sclk in Hz, power1_input = socket power in uW) by a helper thread while this thread keeps the GPU
    busy: plain file reads, no child process.  Returns None where sysfs does not offer them."""
    import glob
    import threading
    try:
        p = torch.cuda.get_device_properties(device)
        hw = glob.glob("/sys/bus/pci/devices/%04x:%02x:%02x.0/hwmon/hwmon*" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))
        f_clk, f_pw, f_cap = (os.path.join(hw[0], n) for n in ("freq1_input", "power1_input", "power1_cap"))
        cap_w = int(open(f_cap).read()) / 1e6
        int(open(f_clk).read()), int(open(f_pw).read())
    except (OSError, ValueError, IndexError, AttributeError):
        return None
    samples, stop = [], threading.Event()

    def poll():
        time.sleep(0.3)   # let the clock settle under load first
        while not stop.is_set():
            try:
                samples.append((int(open(f_clk).read()) // 1000000, int(open(f_pw).read()) / 1e6))
            except (OSError, ValueError):
                break
            time.sleep(0.1)

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        step()
        torch.cuda.synchronize(device)
    stop.set()
    th.join(timeout=2)
    if not samples:
        return None
    clk, pw = sorted(s[0] for s in samples), sorted(s[1] for s in samples)
    return {"sclk_mhz": {"min": clk[0], "median": clk[len(clk) // 2], "max": clk[-1]},
            "socket_w": {"min": pw[0], "median": pw[len(pw) // 2], "max": pw[-1]}, "samples": len(samples),
            "power_cap_w": cap_w, "sclk_nominal_mhz": 2400,
            "how": "amdgpu hwmon (freq1_input, power1_input) read every 0.1 s while the same steps keep running after the timed region"}


def cpu_baseline(pset, ek, dk, m, c_gpu, K_gpu, want_seconds=20.0):
    """Time the reference (or the port) on the host cores over a bounded sample of the SAME items."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import loader
    cores = host_cores()   # every host core this process may use (affinity mask and cgroup quota); the count is reported
    n_all = m.shape[0]
    use_ref = loader.Ref.available()
    if use_ref:   # calibrate on one pair (30-80 ms per pair per core at -O2), then size for ~want_seconds of CPU work
        t1 = loader.Ref().time_encaps_decaps(pset, ek[:1].cpu().numpy(), dk[:1].cpu().numpy(), m[:1].cpu().numpy())[0]
        per_core = max(2, min(512, int(round(want_seconds / max(t1, 1e-3) / cores))))
    else:
        per_core = 2000   # the port runs ~0.5 ms per pair
    per_core = max(1, min(per_core, n_all // cores))
    take = per_core * cores
    ekh, dkh, mh = ek[:take].cpu().numpy(), dk[:take].cpu().numpy(), m[:take].cpu().numpy()
    if use_ref:
        ref = loader.Ref()
        kind = "reference"

        def work(t):
            sl = slice(t * per_core, (t + 1) * per_core)
            return ref.time_encaps_decaps(pset, ekh[sl], dkh[sl], mh[sl])
        label = "reference ml_kem.c+sha3.c (gcc -O2) Encaps_internal + KEM_Decaps"
    else:
        orc = loader.Oracle()
        kind = "port"

        def work(t):
            sl = slice(t * per_core, (t + 1) * per_core)
            t0 = time.perf_counter()
            c, K = orc.encaps(pset, ekh[sl], mh[sl])
            K2, st = orc.decaps(pset, dkh[sl], c)
            return time.perf_counter() - t0, c, K, int(((K2 == K).all(axis=1) & (st == 0)).sum())
        label = "oracle/mlkem_oracle.c restatement (gcc -O2) encaps + decaps"
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(work, range(cores)))
    wall = time.perf_counter() - t0
    pairs = per_core * cores
    cg, Kg = c_gpu[:pairs].cpu().numpy(), K_gpu[:pairs].cpu().numpy()
    c_cpu = np.concatenate([r[1] for r in res])
    K_cpu = np.concatenate([r[2] for r in res])
    agree = sum(r[3] for r in res)
    matches_gpu = bool((c_cpu == cg).all() and (K_cpu == Kg).all() and agree == pairs)
    return {"value": pairs / wall, "unit": "pairs/s", "cores": cores, "kind": kind,
            "sample": f"{pairs} pairs ({per_core} per thread x {cores} threads) of the benched batch, {label}; "
                      f"wall {wall:.1f}s; per-core {pairs / sum(r[0] for r in res):.2f} pairs/s",
            "outputs_match_gpu": matches_gpu}


def run_kem(args, pset, rank, world, device):
    pkg = ge.load_package()
    n = args.batch
    eng = pkg.MLKEM(pset, device=device.index, chunk_items=args.chunk)
    start = rank * n
    d, z, m = (device_seeds(lbl, start, n, device) for lbl in ("mlkem-bench-d", "mlkem-bench-z", "mlkem-bench-m"))
    ek, dk = eng.keygen(d, z)          # untimed (configs[2]); configs[3] times it too (workload kem1024)
    c = torch.empty((n, eng.c_len), dtype=torch.uint8, device=device)
    K = torch.empty((n, 32), dtype=torch.uint8, device=device)
    K2 = torch.empty((n, 32), dtype=torch.uint8, device=device)
    st = torch.empty(n, dtype=torch.int32, device=device)
    timed_keygen = args.workload in ("kem1024", "kem512")
    shared = args.workload == "kem768_shared"
    lib, ctx = eng.lib, eng._ctx

    def step():
        if shared:   # one server key for the whole batch (keys of item 0)
            s = eng._stream()
            eng._check(lib.mlkem_encaps_shared_dev(ctx, pset, n, ek.data_ptr(), m.data_ptr(), c.data_ptr(), K.data_ptr(), s))
            eng._check(lib.mlkem_decaps_shared_dev(ctx, pset, n, dk.data_ptr(), c.data_ptr(), K2.data_ptr(), st.data_ptr(), s))
            return
        if timed_keygen:
            eng.keygen(d, z, ek=ek, dk=dk)
        eng.encaps(ek, m, c=c, K=K)
        eng.decaps(dk, c, K=K2, status=st)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, device)

    # correctness gate (outside the timed region)
    ok = bool(torch.equal(K, K2)) and int(st.abs().max()) == 0
    ct = c.clone()
    idx = torch.arange(0, n, 1024, device=device)
    ct[idx, (idx * 13) % eng.c_len] ^= 2
    Kt, stt = eng.decaps_shared(dk[0], ct) if shared else eng.decaps(dk, ct)
    same = (Kt == K).all(dim=1)
    ok = ok and not bool(same[idx].any()) and int(same.sum()) == n - idx.numel() and int(stt.abs().max()) == 0

    extra = {}
    if rank == 0:
        # per-kernel HIP-event timing of TIMING_PASSES more passes (not part of `value`); totals are per pass
        with pkg.kernel_timing() as kt:
            for _ in range(TIMING_PASSES):
                step()
            torch.cuda.synchronize(device)
        rows = {k: {"ms_total": v[0] / TIMING_PASSES, "launches": v[1] // TIMING_PASSES, "ms_avg": v[0] / max(v[1], 1)}
                for k, v in kt.rows.items()}
        extra["kernels"] = rows
        extra["chunk_items"] = args.chunk or int(os.environ.get("MLKEM_CHUNK_ITEMS", 1 << 18))
        if world == 1:
            extra["clock_power"] = clock_power_probe(step, device)
        if world == 1 and not args.no_cpu:
            if shared:   # the reference has no shared-key path: every pair pays for the full key handling
                extra["cpu_baseline"] = cpu_baseline(pset, ek[:1].expand(n, -1), dk[:1].expand(n, -1), m, c, K)
            else:
                extra["cpu_baseline"] = cpu_baseline(pset, ek, dk, m, c, K)
    eng.close()
    return elapsed, ok, extra


def run_ntt(args, rank, world, device):
    pkg = ge.load_package()
    n = args.batch
    eng = pkg.MLKEM(768, device=device.index, chunk_items=1024)
    g = torch.Generator(device=device).manual_seed(BENCH_SEED + rank)
    f = torch.randint(0, 3329, (n, 256), generator=g, device=device, dtype=torch.int16)
    fh = torch.empty_like(f)
    f2 = torch.empty_like(f)
    lib, ctx = eng.lib, eng._ctx

    def step():
        s = eng._stream()
        eng._check(lib.mlkem_ntt_dev(ctx, n, f.data_ptr(), fh.data_ptr(), s))
        eng._check(lib.mlkem_intt_dev(ctx, n, fh.data_ptr(), f2.data_ptr(), s))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, device)
    ok = bool(torch.equal(f, f2)) and int(fh.min()) >= 0 and int(fh.max()) < 3329
    extra = {}
    if rank == 0:
        with pkg.kernel_timing() as kt:
            for _ in range(TIMING_PASSES):
                step()
            torch.cuda.synchronize(device)
        extra["kernels"] = {k: {"ms_total": v[0] / TIMING_PASSES, "launches": v[1] // TIMING_PASSES, "ms_avg": v[0] / max(v[1], 1)}
                            for k, v in kt.rows.items()}
        if world == 1:
            extra["clock_power"] = clock_power_probe(step, device)
        if world == 1 and not args.no_cpu:
            from oracle import loader
            orc = loader.Oracle()
            sample = f[:20000].cpu().numpy().view(np.uint16)
            t0 = time.perf_counter()
            h = orc.ntt(sample)
            back = orc.intt(h)
            dt = time.perf_counter() - t0
            extra["cpu_baseline"] = {"value": sample.shape[0] / dt, "unit": "polys/s", "cores": 1, "kind": "port",
                                     "sample": "20000 polynomials NTT+InverseNTT, oracle/mlkem_oracle.c (gcc -O2), 1 thread",
                                     "outputs_match_gpu": bool((h == fh[:20000].cpu().numpy().view(np.uint16)).all() and (back == sample).all())}
    eng.close()
    return elapsed, ok, extra


def source_id():
    """sha256 over the CODE of the kernel / C-ABI sources (comments and white space stripped): ties a committed PMC traffic
    file to the build it was measured on (there is no .git on the GPU box, so the commit hash itself cannot be checked
    there); a comment edit does not invalidate a measurement."""
    import re
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "crystals-kyber_amd", "csrc")
    for f in sorted(x for x in os.listdir(csrc) if x.endswith((".hpp", ".hip", ".c"))):   # headers under include/ only declare
        with open(os.path.join(csrc, f), "r", encoding="utf-8", errors="replace") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # block comments
        text = re.sub(r"//[^\n]*", " ", text)                    # line comments (the sources hold no '//' inside string literals)
        h.update(f.encode() + b"\0" + " ".join(text.split()).encode())
    return h.hexdigest()[:16]


WORKLOAD_META = {
    "ntt": ("batched forward+inverse NTT polynomials/sec at batch 2^20", "polys/s",
            "configs[1]: batched forward+inverse NTT only, %d polynomials per GPU",
            "f32 (exact integer arithmetic < 2^24), u16 I/O"),
    "kem768": ("ML-KEM-768 encaps+decaps/sec at batch 2^20; achieved HBM GB/s vs peak", "pairs/s",
               "configs[2]: ML-KEM-768 full Encaps+Decaps (KEM_Decaps incl. dk hash check), batch %d per GPU, keys from batch KeyGen (untimed)", None),
    "kem1024": ("ML-KEM-1024 keygen+encaps+decaps/sec at batch 2^20", "triples/s", "configs[3]: ML-KEM-1024 KeyGen+Encaps+Decaps, batch %d per GPU", None),
    "kem512": ("ML-KEM-512 keygen+encaps+decaps/sec at batch 2^20", "triples/s", "ML-KEM-512 KeyGen+Encaps+Decaps, batch %d per GPU", None),
    "kem768_shared": ("ML-KEM-768 encaps+decaps/sec at batch 2^20, ONE key pair for the whole batch", "pairs/s",
                      "extra (not a BASELINE config): ML-KEM-768 Encaps+Decaps of %d items to / under one key (mlkem_*_shared_dev)", None),
}
KEM_DTYPE = "u32+f32 (64-bit Keccak lanes as 2 x u32; mod-3329 arithmetic exact on integers < 2^24 in the fp32 pipe; u8/u16 I/O)"


def run_workload(workload, args, rank, world, device):
    if workload == "ntt":
        return run_ntt(args, rank, world, device)
    pset = {"kem768": 768, "kem1024": 1024, "kem512": 512, "kem768_shared": 768}[workload]
    return run_kem(args, pset, rank, world, device)


def entry(workload, args, elapsed, ok, extra, world):
    """value / ms_per_step / roofline / kernels of one measured workload (the headline and each `also` member)."""
    units = args.batch * world * args.steps
    value = units / elapsed
    ms_step = 1e3 * elapsed / args.steps
    algo = ALGO_BYTES[workload]
    achieved = value / world * algo / 1e9          # per-GPU algorithmic GB/s
    kernels = extra.get("kernels", {})
    dom = max(kernels.items(), key=lambda kv: kv[1]["ms_total"])[0] if kernels else None
    if workload == "ntt" and dom:
        # NTT-only: each of the two kernels (forward, inverse) handles a whole unit half: 1024 algorithmic bytes per
        # polynomial and launch; the roofline entry is the dominant kernel's, from its own HIP-event duration.
        per_launch = 1024.0 * args.batch
        achieved = per_launch / (kernels[dom]["ms_avg"] * 1e-3) / 1e9
        scope = "dominant kernel %s: 1024 B/polynomial x %d polynomials per launch / its average HIP-event duration; " \
                "whole step (fwd+inv, 2048 B/poly) = %.1f GB/s" % (dom, args.batch, value / world * algo / 1e9)
    else:
        scope = "whole pass (all kernels of one step, no single kernel covers a unit); algorithmic bytes = %d B/unit x %d units " \
                "per step / step time" % (algo, args.batch)
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "scope": scope,
                "dominant_kernel": dom,
                "dominant_kernel_ms_avg": kernels[dom]["ms_avg"] if dom else None,
                "dominant_kernel_share": kernels[dom]["ms_total"] / sum(k["ms_total"] for k in kernels.values()) if dom else None}
    # HBM traffic from the PMC counters is collected off-line (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot run inside
    # this process).  The committed summary is attached only when it was measured on THIS build (same source hash), for
    # this workload, batch and chunking; otherwise traffic stays null.
    tpath = os.path.join(ROOT, "profiles", "r02_pmc_traffic_%s.json" % workload)
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        if t.get("source_id") == source_id() and t.get("batch") == args.batch:
            roofline["traffic"] = t["hbm_bytes_per_step_corrected"]
            roofline["traffic_note"] = "bytes per step from %s (git %s, source_id %s): (2*FETCH_SIZE + WRITE_SIZE)*1024; raw = %.3g" % (
                os.path.relpath(tpath, ROOT), t.get("git_head", "?"), t["source_id"], t["hbm_bytes_per_step_raw"])
        else:
            roofline["traffic_note"] = "%s was measured on another build (source_id %s != %s) or batch: not attached" % (
                os.path.relpath(tpath, ROOT), t.get("source_id"), source_id())
    if workload in ("kem512", "kem768", "kem1024") and dom == "k_sample_main":
        # the dominant kernel's own HBM bytes per launch: per item it reads rho and r (32 B each) and
        # writes the k x k matrix (A_POLY_BYTES per polynomial) and the PRF rows (128 B, 192 B for eta = 3)
        k = {"kem512": 2, "kem768": 3, "kem1024": 4}[workload]
        per_item = 64 + extra.get("a_poly_bytes", 512) * k * k + (2 * k + 1) * 128 + (k * 64 if k == 2 else 0)
        chunk = min(extra.get("chunk_items") or (1 << 18), args.batch)
        roofline["dominant_kernel_bytes_per_launch"] = per_item * chunk
        roofline["dominant_kernel_GBps"] = per_item * chunk / (kernels[dom]["ms_avg"] * 1e-3) / 1e9
        roofline["dominant_kernel_frac"] = roofline["dominant_kernel_GBps"] / HBM_PEAK_GBS
    if workload == "kem768":
        # the bound that actually binds: 95 Keccak-f per pair x 24 rounds x 180 VALU (122 full-rate + 58
        # v_alignbit_b32 at ~0.58x rate, profiles/r01_valu_ubench.txt), + NTT / codec / sampling work
        keccak_ops = 95 * KECCAK_PERM_LANE_OPS
        roofline["binding"] = {"bound": "valu-int32", "keccak_lane_ops_per_pair": keccak_ops,
                               "keccak_lane_ops_per_s": value / world * keccak_ops,
                               "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                               "frac_keccak_only": value / world * keccak_ops / VALU_PEAK_LANE_OPS,
                               "note": "the pass runs into the socket power limit (rocm-smi: ~1340 W, shader clock ~2.07 GHz instead of "
                                       "2.4 GHz; DESIGN.md section 5, profiles/r01_clock_power_watch.txt, profiles/r01_power_ubench.txt)"}
    if extra.get("clock_power"):
        roofline["clock_power"] = extra["clock_power"]
    return {"value": value, "ms_per_step": ms_step, "correct": ok, "roofline": roofline, "kernels": kernels}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 20: ~0.4 s, long enough for the power-capped clock to settle; 200 for the "
                         "sub-millisecond NTT workload)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kem768", choices=("kem768", "kem1024", "kem512", "ntt", "kem768_shared"))
    ap.add_argument("--batch", type=int, default=1 << 20, help="items per GPU per step (BASELINE: 2^20)")
    ap.add_argument("--chunk", type=int, default=0, help="engine chunk size in items (0 = library default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-also", action="store_true", help="skip the configs[1] / configs[3] legs of the default run")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 ranks on fewer GPUs: ranks share the visible GPU(s), barrier over gloo (automatic when the node "
                         "has fewer GPUs than ranks)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.workload == "ntt" else 20

    rank, world, local = dist_setup(args.gpus, args.rehearse)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    elapsed, ok, extra = run_workload(args.workload, args, rank, world, device)
    metric, unit, wl, dtype = WORKLOAD_META[args.workload]
    dtype = dtype or KEM_DTYPE

    also = {}
    if world == 1 and args.workload == "kem768" and not args.no_also:
        # BASELINE configs[1] and configs[3] in the same process, so that they are driver-observed too (not the headline:
        # `value` above is configs[2]); same timing discipline, their own step counts
        for wl2, steps2 in (("ntt", 200), ("kem1024", 10)):
            a2 = argparse.Namespace(**vars(args))
            a2.workload, a2.steps, a2.warmup, a2.no_cpu = wl2, steps2, 2, True
            el2, ok2, ex2 = run_workload(wl2, a2, rank, world, device)
            e2 = entry(wl2, a2, el2, ok2, ex2, world)
            m2 = WORKLOAD_META[wl2]
            e2.update({"metric": m2[0], "unit": m2[1], "steps": steps2, "warmup": 2, "config": {"workload": m2[2] % a2.batch}})
            also[wl2] = e2
            ok = ok and ok2
            torch.cuda.empty_cache()

    if _dist() is not None:
        _dist().destroy_process_group()
    if rank != 0:
        return
    e = entry(args.workload, args, elapsed, ok, extra, world)
    par = "shard%d (no collectives)" % world
    if SHARED_GPU:
        par += "; REHEARSAL: %d ranks share %d GPU(s), barrier over gloo - not a scaling measurement" % (world, torch.cuda.device_count())
    line = {"metric": metric, "value": e["value"], "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": e["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic", "config": {"workload": wl % args.batch, "batch_per_gpu": args.batch, "parallelism": par,
                                            "chunk_items": extra.get("chunk_items")},
            "correct": ok, "roofline": e["roofline"], "kernels": e["kernels"]}
    if "cpu_baseline" in extra:
        line["cpu_baseline"] = extra["cpu_baseline"]
    if also:
        line["also"] = also
    print(json.dumps(line))
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
