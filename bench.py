#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X:
ML-KEM-768 encaps+decaps pairs per second at batch 2^20 (BASELINE configs[2]), inputs resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload kem768|kem1024|kem512|ntt] [--rehearse] [--inproc]

One "step" = one pass of the hot path over one batch: Encaps_internal over 2^20 (ek, m) followed by KEM_Decaps
(hash check included, as the reference's public API does) over the 2^20 (dk, c) it produced.  Keys come from the
engine's own batch KeyGen on seeds d_i, z_i, m_i = SHAKE128(label || LE64(i) || LE64(0xC0FFEE))[:32] (untimed).
N > 1: one process per GPU (torchrun), every rank runs the same per-GPU batch on its own item range (weak
scaling, no collective in the data path); the timed region is bracketed by barrier + synchronize and the MAX
over ranks is used.  Rank 0 prints ONE JSON line.  When the node has fewer GPUs than ranks (or with --rehearse) the
ranks share the visible GPU(s) and the barrier runs over gloo instead of RCCL: a rehearsal of the N-rank path on a
one-GPU box (the line says so in config.parallelism; its value is not a scaling result).

N > 1 lines carry `per_gpu`: one entry per rank (device, its own rate and ms/step from its own clock between the opening
barrier and its own synchronize, shader clock and socket power sampled on that rank while all ranks keep running) next to
the aggregate; `correct` is the AND over all ranks and every rank exits non-zero when it is false.
--inproc: the same workload in ONE process through mlkem_{encaps,decaps}_multi_dev, one member per visible device (members
repeat devices when there are fewer devices than --gpus: a rehearsal, labelled as such) -- the in-process form of the shard.

At N = 1 the line also carries `also`: BASELINE configs[1] (NTT-only), configs[3] (ML-KEM-1024 KeyGen+Encaps+Decaps), small_calls (1 / 64 / 768-item call pairs)
measured in the same process right after the headline, each with value / ms_per_step / correct / roofline.

Extra objects on the line:
  roofline      whole-pass algorithmic bytes (5856 B per pair) / pass time vs the 8 TB/s HBM peak, the
                dominant kernel by HIP-event time, the integer-VALU view that actually binds this path, and
                `clock_power`: shader clock and socket power sampled with rocm-smi while the same steps keep running
                (the path runs into the socket power limit; nominal clock 2400 MHz)
  cpu_baseline  BASELINE.md section 3's three legs, each timed on this box's host cores on a bounded sample of the same items
                and cross-checked against the GPU bytes: `legs.reference_O2` (ml_kem.c + sha3.c, gcc -O2), `legs.reference_O0`
                (the reference makefile's own flags, -Wall -g) and `legs.port` (the multi-threaded restatement
                oracle/mlkem_oracle.c); the top-level fields repeat the -O2 reference (the port when oracle/_ref is absent)
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


def kernel_label(kernel_name):
    """rocprofv3 kernel name -> the label bench.py's `kernels` table uses (the labels the library's launch() calls carry)."""
    import re
    short = re.sub(r"^void ", "", kernel_name).replace("mlkem::", "")
    base = short.split("<")[0].split("(")[0]
    if base == "k_sample":
        return "k_sample_restart"
    if base == "k_ntt4_batch":
        return "k_intt_batch" if "<true>" in short else "k_ntt_batch"
    if base in ("k_encrypt", "k_encrypt2"):
        return "k_encrypt_cmp" if re.search(r",\s*true>", short) else "k_encrypt"
    return {"k_keygen2": "k_keygen", "k_decrypt4": "k_decrypt", "k_sample_resume": "k_sample_tail"}.get(base, base)


def issue_floors(workload):
    """{label: floor SIMD cycles per VALU instruction} of this workload's kernels from profiles/r04_isa_floor.json: the instruction
    mix of the library's own ISA priced at 2 cycles per full-rate wave64 VALU instruction, 4 per half-rate one (v_pk_*,
    v_alignbit, conversions ...), 3.2 per DPP one (tools/isa_floor.py; Keccak kernels: their round loop).  Empty when the file
    belongs to another build."""
    path = os.path.join(ROOT, "profiles", "r04_isa_floor.json")
    if not os.path.exists(path):
        return {}
    d = json.load(open(path))
    if d.get("source_id") != source_id():
        return {}
    k = {"kem512": 2, "kem768": 3, "kem1024": 4, "kem768_shared": 3}.get(workload)
    out = {}
    for name, row in d["kernels"].items():
        if k is not None and "<" in name and not name.split("<")[1].startswith(("%d," % k, "%d>" % k, "true", "false", "3329")):
            continue
        out.setdefault(row["label"], row["floor_cycles_per_valu_instr"])
    return out


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


def all_ranks_ok(ok, device):
    """AND of the per-rank correctness gates (MIN of int(ok)): a wrong byte on ANY rank fails the whole line."""
    dist = _dist()
    if dist is None:
        return bool(ok)
    on_cpu = SHARED_GPU or dist.get_backend() == "gloo"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if on_cpu else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def gather_per_gpu(entry):
    """One dict per rank, in rank order (control plane, outside the timed region)."""
    dist = _dist()
    if dist is None:
        return [entry]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, entry)
    return out


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


def _threaded(work, cores):
    """work(t) for t in range(cores) on a thread pool (ctypes releases the GIL inside the C calls); returns (results, wall s)."""
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(work, range(cores)))
    return res, time.perf_counter() - t0


def _leg(kind, flags, unit, done, wall, cpu_s, cores, per_core, matches, what):
    return {"value": done / wall, "unit": unit, "cores": cores, "kind": kind, "flags": flags, "per_core": done / cpu_s,
            "sample": f"{done} {unit.split('/')[0]} ({per_core} per thread x {cores} threads) of the benched batch, {what}; "
                      f"wall {wall:.1f}s", "outputs_match_gpu": bool(matches)}


def cpu_baseline(pset, ek, dk, m, c_gpu, K_gpu, triples=None, budget=(14.0, 8.0, 4.0)):
    """the reference at -O2, the reference at its makefile's own flags (-Wall -g, i.e. -O0) and the
    multi-threaded restatement, each over a bounded sample of the SAME items (sized for ~budget[i] seconds of CPU work in all)
    and each compared byte for byte with what the GPU produced.  `triples` = (d, z): time KeyGen_internal as well (configs[3])."""
    from oracle import loader
    cores = host_cores()   # every host core this process may use (affinity mask and cgroup quota); the count is reported
    n_all = m.shape[0]
    unit = "triples/s" if triples else "pairs/s"
    ops = "KeyGen_internal + Encaps_internal + KEM_Decaps" if triples else "Encaps_internal + KEM_Decaps"
    legs = {}

    def host_rows(t, take):
        return t[:take].cpu().numpy()

    def check(take, res):   # res[t] = (seconds, [ek, dk,] c, K, agree)
        c_cpu = np.concatenate([r[-3] for r in res])
        K_cpu = np.concatenate([r[-2] for r in res])
        ok = (c_cpu == host_rows(c_gpu, take)).all() and (K_cpu == host_rows(K_gpu, take)).all() and sum(r[-1] for r in res) == take
        if triples:
            ok = ok and (np.concatenate([r[1] for r in res]) == host_rows(ek, take)).all() and \
                (np.concatenate([r[2] for r in res]) == host_rows(dk, take)).all()
        return ok

    for name, o0, flags, want in (("reference_O2", False, "gcc -O2", budget[0]), ("reference_O0", True, "gcc -Wall -g (the reference makefile's flags, /root/reference/makefile:2)", budget[1])):
        if want <= 0 or not loader.Ref.available(o0):
            continue
        ref = loader.Ref(o0)
        if triples and not hasattr(ref.lib, "ref_time_triples"):
            continue
        one = (lambda sl: ref.time_triples(pset, dz[0][sl], dz[1][sl], mh[sl])) if triples else \
              (lambda sl: ref.time_encaps_decaps(pset, ekh[sl], dkh[sl], mh[sl]))
        # calibrate on one item (30-800 ms per item per core), then size the sample
        mh = host_rows(m, 1)
        ekh, dkh = host_rows(ek, 1), host_rows(dk, 1)
        dz = (host_rows(triples[0], 1), host_rows(triples[1], 1)) if triples else None
        t1 = one(slice(0, 1))[0]
        per_core = max(1, min(512, int(round(want / max(t1, 1e-3) / cores)), n_all // cores))
        take = per_core * cores
        mh, ekh, dkh = host_rows(m, take), host_rows(ek, take), host_rows(dk, take)
        dz = (host_rows(triples[0], take), host_rows(triples[1], take)) if triples else None
        res, wall = _threaded(lambda t: one(slice(t * per_core, (t + 1) * per_core)), cores)
        legs[name] = _leg("reference", flags, unit, take, wall, sum(r[0] for r in res), cores, per_core, check(take, res),
                          f"reference ml_kem.c+sha3.c ({flags.split(' (')[0]}) {ops}")

    orc = loader.Oracle()
    per_core = max(1, min(int(budget[2] * (400 if triples else 1200)), n_all // cores))   # the port runs ~0.5 ms per pair
    take = per_core * cores
    mh, ekh, dkh = host_rows(m, take), host_rows(ek, take), host_rows(dk, take)
    dz = (host_rows(triples[0], take), host_rows(triples[1], take)) if triples else None

    def port_work(t):
        sl = slice(t * per_core, (t + 1) * per_core)
        t0 = time.perf_counter()
        if triples:
            ek_c, dk_c = orc.keygen(pset, dz[0][sl], dz[1][sl])
        else:
            ek_c, dk_c = ekh[sl], dkh[sl]
        c, K = orc.encaps(pset, ek_c, mh[sl])
        K2, st = orc.decaps(pset, dk_c, c)
        return time.perf_counter() - t0, ek_c, dk_c, c, K, int(((K2 == K).all(axis=1) & (st == 0)).sum())
    res, wall = _threaded(port_work, cores)
    legs["port"] = _leg("port", "gcc -O2", unit, take, wall, sum(r[0] for r in res), cores, per_core, check(take, res),
                        f"oracle/mlkem_oracle.c restatement (gcc -O2), one thread per core, {ops}")

    head = dict(legs.get("reference_O2") or legs["port"])
    head["legs"] = legs
    head["outputs_match_gpu"] = all(v["outputs_match_gpu"] for v in legs.values())
    return head


def probe_medians(cp):
    return (cp["sclk_mhz"]["median"], cp["socket_w"]["median"]) if cp else (None, None)


def per_gpu_entry(rank, device, units, steps, own_s, ok, cp, solo=None):
    clk, pw = probe_medians(cp)
    e = {"rank": rank, "device": device.index, "name": torch.cuda.get_device_name(device), "value": units * steps / own_s,
         "ms_per_step": 1e3 * own_s / steps, "correct": bool(ok), "sclk_mhz": clk, "socket_w": pw}
    if solo is not None:
        e["solo_value"] = solo
    return e


SOLO_STEPS = 5


def solo_anchor(step, sync, rank, world, units):
    """The N = 1 anchor INSIDE the N > 1 job: before the group-timed region every rank in turn runs SOLO_STEPS steps of the same
    workload ALONE on its GPU while the others idle at a barrier.  aggregate / sum(solo values) is then a scaling efficiency
    between runs of one job on one set of devices, instead of against an N = 1 run on another box (box-to-box spread: +-4 %).
    Returns this rank's solo units/s (None when the job has one rank)."""
    if world <= 1:
        return None
    mine = None
    for r in range(world):
        barrier(world)
        if r == rank:
            sync()
            t0 = time.perf_counter()
            for _ in range(SOLO_STEPS):
                step()
            sync()
            mine = units * SOLO_STEPS / (time.perf_counter() - t0)
    barrier(world)
    return mine


def scaling_anchor(value, per_gpu):
    """aggregate vs the in-job solo runs (see solo_anchor); None when a rank has no solo value."""
    if not per_gpu or any(g.get("solo_value") is None for g in per_gpu) or len(per_gpu) < 2:
        return None
    total = sum(g["solo_value"] for g in per_gpu)
    return {"solo_steps": SOLO_STEPS, "sum_solo_value": total, "efficiency": value / total,
            "per_gpu_efficiency": [g["value"] / g["solo_value"] for g in per_gpu],
            "note": "solo_value = the same step run by that rank alone while the other ranks wait at a barrier, in this job, "
                    "before the group-timed region; efficiency = aggregate value / sum of solo values"}


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
    solo = solo_anchor(step, lambda: torch.cuda.synchronize(device), rank, world, n)
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    own = time.perf_counter() - t0      # this rank's own time: per_gpu
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, device)

    # shader clock and socket power on EVERY rank, right behind the timed region while all ranks keep stepping together
    clock_power = clock_power_probe(step, device)
    barrier(world)

    # correctness gate (outside the timed region), on every rank; reduced over the ranks by the caller
    ok = bool(torch.equal(K, K2)) and int(st.abs().max()) == 0
    ct = c.clone()
    idx = torch.arange(0, n, 1024, device=device)
    ct[idx, (idx * 13) % eng.c_len] ^= 2
    Kt, stt = eng.decaps_shared(dk[0], ct) if shared else eng.decaps(dk, ct)
    same = (Kt == K).all(dim=1)
    ok = ok and not bool(same[idx].any()) and int(same.sum()) == n - idx.numel() and int(stt.abs().max()) == 0

    extra = {"per_gpu": per_gpu_entry(rank, device, n, args.steps, own, ok, clock_power, solo), "clock_power": clock_power}
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
        if world == 1 and not args.no_cpu:
            if shared:   # the reference has no shared-key path: every pair pays for the full key handling
                extra["cpu_baseline"] = cpu_baseline(pset, ek[:1].expand(n, -1), dk[:1].expand(n, -1), m, c, K)
            else:
                extra["cpu_baseline"] = cpu_baseline(pset, ek, dk, m, c, K, triples=(d, z) if timed_keygen else None,
                                                     budget=args.cpu_budget)
    eng.close()
    return elapsed, ok, extra


def run_kem_inproc(args, pset, members):
    """The same per-member workload in ONE process: mlkem_{encaps,decaps}_multi_dev, member r on device r % ndev with its own
    2^20-item shard (items r*batch ...), own stream and engine context; no exchange between members.  Per-member times come
    from events on the members' own streams."""
    pkg = ge.load_package()
    n = args.batch
    ndev = torch.cuda.device_count()
    devices = [r % ndev for r in range(members)]
    mm = pkg.MLKEMMulti(pset, devices=devices, chunk_items=args.chunk)
    devs = [torch.device("cuda", dv) for dv in devices]
    d, z, m = ([device_seeds(lbl, r * n, n, devs[r]) for r in range(members)] for lbl in ("mlkem-bench-d", "mlkem-bench-z", "mlkem-bench-m"))
    ek, dk = mm.keygen_dev(d, z)       # untimed
    c = [torch.empty((n, mm.c_len), dtype=torch.uint8, device=dv) for dv in devs]
    K, K2 = ([torch.empty((n, 32), dtype=torch.uint8, device=dv) for dv in devs] for _ in range(2))
    st = [torch.empty(n, dtype=torch.int32, device=dv) for dv in devs]

    def step():
        mm.encaps_dev(ek, m, c=c, K=K)
        mm.decaps_dev(dk, c, K=K2, status=st)

    for _ in range(args.warmup):
        step()
    mm.sync()
    # the N = 1 anchor inside the job (solo_anchor's rule): each member in turn runs alone, the others get empty shards
    solos = [None] * members
    if members > 1:
        for r in range(members):
            only = lambda ts: [t if i == r else t[:0] for i, t in enumerate(ts)]   # noqa: E731
            a = [only(x) for x in (ek, m, c, K, dk, K2, st)]
            mm.sync()
            t0 = time.perf_counter()
            for _ in range(SOLO_STEPS):
                mm.encaps_dev(a[0], a[1], c=a[2], K=a[3])
                mm.decaps_dev(a[4], a[2], K=a[5], status=a[6])
            mm.sync()
            solos[r] = n * SOLO_STEPS / (time.perf_counter() - t0)
    ext = mm.streams()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(members)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(members)]
    t0 = time.perf_counter()
    for r in range(members):
        ev0[r].record(ext[r])
    for _ in range(args.steps):
        step()
    for r in range(members):
        ev1[r].record(ext[r])
    mm.sync()
    elapsed = time.perf_counter() - t0
    own = [ev0[r].elapsed_time(ev1[r]) * 1e-3 for r in range(members)]
    cps = {}
    for dv in sorted(set(devices)):   # one probe per distinct device while ALL members keep running
        cps[dv] = clock_power_probe(lambda: (step(), mm.sync()), torch.device("cuda", dv))
    ok_all, per_gpu = True, []
    for r in range(members):
        ok = bool(torch.equal(K[r], K2[r])) and int(st[r].abs().max()) == 0
        per_gpu.append(per_gpu_entry(r, devs[r], n, args.steps, own[r], ok, cps[devices[r]], solos[r]))
        ok_all = ok_all and ok
    ct = [t.clone() for t in c]
    idx = [torch.arange(0, n, 1024, device=dv) for dv in devs]
    for r in range(members):
        ct[r][idx[r], (idx[r] * 13) % mm.c_len] ^= 2
    Kt, stt = mm.decaps_dev(dk, ct)
    mm.sync()
    for r in range(members):
        same = (Kt[r] == K[r]).all(dim=1)
        ok = not bool(same[idx[r]].any()) and int(same.sum()) == n - idx[r].numel() and int(stt[r].abs().max()) == 0
        per_gpu[r]["correct"] = per_gpu[r]["correct"] and ok
        ok_all = ok_all and ok
    with pkg.kernel_timing() as kt:
        step()
        mm.sync()
    kernels = {k: {"ms_total": v[0], "launches": v[1], "ms_avg": v[0] / max(v[1], 1)} for k, v in kt.rows.items()}
    mm.close()
    return elapsed, ok_all, {"per_gpu_all": per_gpu, "clock_power": cps[devices[0]], "kernels": kernels, "devices": devices,
                             "kernel_table_units": n * members,   # the kernel table covers one step of ALL members
                             "chunk_items": args.chunk or int(os.environ.get("MLKEM_CHUNK_ITEMS", 1 << 18))}


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
    solo = solo_anchor(step, lambda: torch.cuda.synchronize(device), rank, world, n)
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    own = time.perf_counter() - t0
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, device)
    clock_power = clock_power_probe(step, device)
    barrier(world)
    ok = bool(torch.equal(f, f2)) and int(fh.min()) >= 0 and int(fh.max()) < 3329
    extra = {"per_gpu": per_gpu_entry(rank, device, n, args.steps, own, ok, clock_power, solo), "clock_power": clock_power}
    if rank == 0:
        with pkg.kernel_timing() as kt:
            for _ in range(TIMING_PASSES):
                step()
            torch.cuda.synchronize(device)
        extra["kernels"] = {k: {"ms_total": v[0] / TIMING_PASSES, "launches": v[1] // TIMING_PASSES, "ms_avg": v[0] / max(v[1], 1)}
                            for k, v in kt.rows.items()}
        if world == 1 and not args.no_cpu:
            from oracle import loader
            orc = loader.Oracle()
            cores = host_cores()
            per_core = 4000
            sample = f[:per_core * cores].cpu().numpy().view(np.uint16)

            def work(t):
                sl = slice(t * per_core, (t + 1) * per_core)
                t0 = time.perf_counter()
                h = orc.ntt(sample[sl])
                back = orc.intt(h)
                return time.perf_counter() - t0, h, back
            res, wall = _threaded(work, cores)
            h = np.concatenate([r[1] for r in res])
            back = np.concatenate([r[2] for r in res])
            done = sample.shape[0]
            extra["cpu_baseline"] = {"value": done / wall, "unit": "polys/s", "cores": cores, "kind": "port", "flags": "gcc -O2",
                                     "per_core": done / sum(r[0] for r in res),
                                     "sample": f"{done} polynomials ({per_core} per thread x {cores} threads) NTT+InverseNTT, "
                                               f"oracle/mlkem_oracle.c (the reference's NTT spends half its time recomputing zeta: "
                                               f"SURVEY a8); wall {wall:.1f}s",
                                     "outputs_match_gpu": bool((h == fh[:done].cpu().numpy().view(np.uint16)).all() and (back == sample).all())}
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
    whole = value / world * algo / 1e9             # per-GPU algorithmic GB/s of the whole pass
    kernels = extra.get("kernels", {})
    dom = max(kernels.items(), key=lambda kv: kv[1]["ms_total"])[0] if kernels else None
    achieved, scope = whole, "whole pass: algorithmic bytes = %d B/unit x %d units per step / step time" % (algo, args.batch)
    if dom:
        # The dominant kernel by the algorithmic-bytes rule: SURVEY 8d's per-unit bytes x the units one launch stands for / the
        # kernel's average launch duration (HIP events on its launch stream).  NTT-only: each of the two kernels handles one
        # direction = 1024 B per polynomial, all `batch` polynomials per launch.  Full KEM: a step issues `launches` launches
        # of the dominant kernel over the batch, so one launch stands for batch / launches units.
        launches = max(1, kernels[dom]["launches"])
        per_unit = 1024.0 if workload == "ntt" else float(algo)
        units_per_launch = args.batch if workload == "ntt" else extra.get("kernel_table_units", args.batch) / launches
        achieved = per_unit * units_per_launch / (kernels[dom]["ms_avg"] * 1e-3) / 1e9
        scope = "dominant kernel %s: %g B/unit x %g units per launch (%d launches per step) / its average HIP-event duration %.4f ms" % (
            dom, per_unit, units_per_launch, launches, kernels[dom]["ms_avg"])
    # `bound` names what binds: the NTT-only pass streams at 0.6 of the HBM peak; the full-KEM passes are VALU- and socket-power-bound
    # (~1 M lane-ops per pair against 5.9 KB) -- achieved / peak / frac stay the HBM figures the contract asks for
    roofline = {"bound": "hbm" if workload == "ntt" else "valu+power", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "scope": scope,
                "whole_pass": {"achieved": whole, "frac": whole / HBM_PEAK_GBS,
                               "scope": "all kernels of one step: %d B/unit x %d units / step time" % (algo, args.batch)},
                "dominant_kernel": dom,
                "dominant_kernel_ms_avg": kernels[dom]["ms_avg"] if dom else None,
                "dominant_kernel_share": kernels[dom]["ms_total"] / sum(k["ms_total"] for k in kernels.values()) if dom else None}
    # HBM traffic from the PMC counters is collected off-line (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot run inside
    # this process).  The committed summary is attached only when it was measured on THIS build (same source hash), for
    # this workload, batch and chunking; otherwise traffic stays null.
    tpath = next((q for q in (os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (rnd, workload)) for rnd in ("r04", "r03", "r02"))
                  if os.path.exists(q)), None)
    if tpath:
        t = json.load(open(tpath))
        if t.get("source_id") == source_id() and t.get("batch") == args.batch:
            # `traffic` is per LAUNCH of the dominant kernel, like `achieved`; the all-kernel figure of one step stands beside it
            pl = t.get("per_label", {}).get(dom) if dom else None
            roofline["traffic"] = pl["bytes_per_launch_corrected"] if pl else None
            roofline["traffic_raw"] = pl["bytes_per_launch_raw"] if pl else None
            roofline["traffic_exact"] = pl.get("bytes_per_launch_exact") if pl else None
            roofline["traffic_per_step"] = t["hbm_bytes_per_step_corrected"]
            roofline["traffic_per_step_raw"] = t["hbm_bytes_per_step_raw"]
            roofline["traffic_per_step_exact"] = t.get("hbm_bytes_per_step_exact")
            roofline["traffic_over_algorithmic"] = t["hbm_bytes_per_step_corrected"] / (float(algo) * args.batch)
            roofline["traffic_note"] = ("traffic = HBM bytes per launch of %s, traffic_per_step = all kernels of one step; from %s (git %s, "
                                        "source_id %s): (2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md prescribes; *_raw = (FETCH_SIZE + "
                                        "WRITE_SIZE)*1024 as reported; *_exact = request-size weighted TCC_EA0_RDREQ*/WRREQ* counts (all read requests "
                                        "are 128-byte requests: profiles/r04_traffic_calibration.txt)" % (
                                            dom, os.path.relpath(tpath, ROOT), t.get("git_head", "?"), t["source_id"]))
        else:
            roofline["traffic_note"] = "%s was measured on another build (source_id %s != %s) or batch: not attached" % (
                os.path.relpath(tpath, ROOT), t.get("source_id"), source_id())
    # measured issue fraction of the hot kernels: floor cycles per VALU instruction of the kernel's mix (ISSUE_FLOOR) / the cycles
    # per VALU instruction the SQ counters of the committed PMC pass show for this build (tools/pmc_summary.py --json)
    spath = next((q for q in (os.path.join(ROOT, "profiles", "%s_pmc_sq_%s.json" % (rnd, workload)) for rnd in ("r04",)) if os.path.exists(q)), None)
    if spath and tpath and json.load(open(tpath)).get("source_id") == source_id():
        sq = json.load(open(spath))
        floors = issue_floors(workload)
        issue = {}
        for name, row in sq.items():
            lbl = kernel_label(name)
            if lbl in kernels and lbl in floors and row.get("cycles_per_valu_instr"):
                issue[lbl] = {"cycles_per_valu_instr": row["cycles_per_valu_instr"], "floor": floors[lbl],
                              "issue_frac": floors[lbl] / row["cycles_per_valu_instr"],
                              "resident_waves_per_simd": row.get("resident_waves_per_simd")}
        if issue:
            roofline["issue"] = {"dominant_kernel": issue.get(dom), "kernels": issue,
                                 "note": "issue_frac = floor SIMD cycles per VALU instruction of the kernel's instruction mix "
                                         "(profiles/r04_isa_floor.json, tools/isa_floor.py) / measured (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs / "
                                         "SQ_INSTS_VALU), from %s of the same source_id" % os.path.relpath(spath, ROOT)}
    if workload in ("kem512", "kem768", "kem1024") and dom == "k_sample_main":
        # the dominant kernel's own HBM bytes per launch: per item it reads rho and r (32 B each) and
        # writes the k x k matrix (512 B per polynomial) and the PRF rows (128 B, 192 B for eta = 3)
        k = {"kem512": 2, "kem768": 3, "kem1024": 4}[workload]
        per_item = 64 + 512 * k * k + (2 * k + 1) * 128 + (k * 64 if k == 2 else 0)
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
                               "note": "the pass runs into the socket power limit (clock_power below: shader clock well under the "
                                       "nominal 2.4 GHz at ~1.33 kW; DESIGN.md section 5)"}
    energy = None
    if extra.get("clock_power"):
        roofline["clock_power"] = extra["clock_power"]
        # socket energy per unit: median socket power while the same steps keep running x step time / units per step and GPU.
        # The pass sits on the socket power limit, where time = dynamic joules / (limit - idle) (profiles/r04_energy.txt: the
        # kernel families' stand-alone joules predict the step within 3 %), so joules per unit is the figure of merit.
        w = extra["clock_power"]["socket_w"]["median"]
        energy = {"joules_per_unit": w * ms_step * 1e-3 / args.batch, "socket_w": w,
                  "how": "median socket power (hwmon power1_input) x ms_per_step / batch_per_gpu; includes the idle share (~290 W)"}
    return {"value": value, "ms_per_step": ms_step, "correct": ok, "roofline": roofline, "kernels": kernels, "energy": energy}


def small_calls(device):
    """The regime every call of the ml_kem.h drop-in API lives in, measured in the driver-observed run: ML-KEM-768 Encaps + Decaps
    CALL PAIRS of 1 / 64 / 768 items -- device-resident buffers, calls queued back to back (launch overhead included), and the
    host-pointer path of one item (what the shim calls: mlkem_encaps / mlkem_decaps on host memory, synchronous).  Every pair is
    checked (K == K', status 0); a tampered ciphertext must be rejected."""
    import ctypes as C
    pkg = ge.load_package()
    lib = pkg.load_library()
    eng = pkg.MLKEM(768, device=device.index or 0, chunk_items=4096)
    out = {"unit": "ms per Encaps + Decaps call pair (ML-KEM-768)", "device_resident": {}, "correct": True}
    for n in (1, 64, 768):
        d, z, m = (device_seeds(lbl, 0, n, device) for lbl in ("mlkem-bench-d", "mlkem-bench-z", "mlkem-bench-m"))
        ek, dk = eng.keygen(d, z)
        c = torch.empty((n, eng.c_len), dtype=torch.uint8, device=device)
        K, K2 = (torch.empty((n, 32), dtype=torch.uint8, device=device) for _ in range(2))
        st = torch.empty(n, dtype=torch.int32, device=device)
        reps = 300

        def pair():
            eng.encaps(ek, m, c=c, K=K)
            eng.decaps(dk, c, K=K2, status=st)
        for _ in range(10):
            pair()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            pair()
        torch.cuda.synchronize(device)
        out["device_resident"][str(n)] = 1e3 * (time.perf_counter() - t0) / reps
        good = bool(torch.equal(K, K2)) and int(st.abs().sum()) == 0
        cb = c.clone()
        cb[n - 1, 5] ^= 2
        Kb, _ = eng.decaps(dk, cb)
        good = good and not bool(torch.equal(Kb[n - 1], K[n - 1])) and bool(torch.equal(Kb[: n - 1], K[: n - 1]))
        out["correct"] = out["correct"] and good
    # host pointers, one item
    ek_h, dk_h, m_h = (t.cpu().numpy().copy() for t in (ek[:1], dk[:1], m[:1]))
    c_h, K_h, K2_h = np.zeros((1, eng.c_len), np.uint8), np.zeros((1, 32), np.uint8), np.zeros((1, 32), np.uint8)
    st_h = np.ones(1, np.int32)
    for f in (lib.mlkem_encaps, lib.mlkem_decaps):
        f.argtypes = [C.c_int, C.c_size_t] + [C.c_void_p] * 4

    def hpair():
        rc = lib.mlkem_encaps(768, 1, ek_h.ctypes.data, m_h.ctypes.data, c_h.ctypes.data, K_h.ctypes.data)
        return rc | lib.mlkem_decaps(768, 1, dk_h.ctypes.data, c_h.ctypes.data, K2_h.ctypes.data, st_h.ctypes.data)
    rc = 0
    for _ in range(10):
        rc |= hpair()
    t0 = time.perf_counter()
    for _ in range(300):
        rc |= hpair()
    out["host_pointer"] = {"1": 1e3 * (time.perf_counter() - t0) / 300}
    out["correct"] = out["correct"] and rc == 0 and bool((K_h == K2_h).all()) and int(st_h[0]) == 0 and bool((K_h == K[:1].cpu().numpy()).all())
    out["note"] = ("one launch per operation, one workgroup per item (mlkem_small.hpp); host_pointer = mlkem_encaps + mlkem_decaps on host memory "
                   "through ctypes, no copy commands (the kernels read and write pinned host memory)")
    eng.close()
    lib.mlkem_host_release()
    return out


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
    ap.add_argument("--inproc", action="store_true",
                    help="one process, --gpus members through mlkem_*_multi_dev (member r on device r %% visible devices)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.workload == "ntt" else 20
    args.cpu_budget = (14.0, 8.0, 4.0)   # CPU-seconds of work for the reference -O2 / reference -O0 / port legs

    if args.inproc:
        return main_inproc(args)
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
        # `value` above is configs[2]); same timing discipline, their own step counts, their own CPU legs
        for wl2, steps2 in (("ntt", 200), ("kem1024", 10)):
            a2 = argparse.Namespace(**vars(args))
            a2.workload, a2.steps, a2.warmup = wl2, steps2, 2
            a2.cpu_budget = (8.0, 0.0, 2.0)   # configs[3]: reference -O2 and port legs (the -O0 leg is on the headline)
            el2, ok2, ex2 = run_workload(wl2, a2, rank, world, device)
            e2 = entry(wl2, a2, el2, ok2, ex2, world)
            m2 = WORKLOAD_META[wl2]
            e2.update({"metric": m2[0], "unit": m2[1], "steps": steps2, "warmup": 2, "config": {"workload": m2[2] % a2.batch}})
            if "cpu_baseline" in ex2:
                e2["cpu_baseline"] = ex2["cpu_baseline"]
            also[wl2] = e2
            ok = ok and ok2
            torch.cuda.empty_cache()
        also["small_calls"] = small_calls(device)
        ok = ok and also["small_calls"]["correct"]

    # the gate of the whole job: every rank's bytes must be right; per-rank results travel to rank 0 over the control plane
    ok = all_ranks_ok(ok, device)
    per_gpu = gather_per_gpu(extra["per_gpu"])
    if _dist() is not None:
        _dist().destroy_process_group()
    if rank != 0:
        sys.exit(0 if ok else 3)
    e = entry(args.workload, args, elapsed, ok, extra, world)
    par = "shard%d (no collectives)" % world
    if SHARED_GPU:
        par += "; REHEARSAL: %d ranks share %d GPU(s), barrier over gloo - not a scaling measurement" % (world, torch.cuda.device_count())
    line = {"metric": metric, "value": e["value"], "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": e["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic", "config": {"workload": wl % args.batch, "batch_per_gpu": args.batch, "parallelism": par,
                                            "chunk_items": extra.get("chunk_items")},
            "correct": ok, "per_gpu": per_gpu, "roofline": e["roofline"], "kernels": e["kernels"]}
    anchor = scaling_anchor(e["value"], per_gpu)
    if anchor:
        line["scaling_anchor"] = anchor
    if e.get("energy"):
        line["joules_per_unit"] = e["energy"]["joules_per_unit"]
        line["energy"] = e["energy"]
    if "cpu_baseline" in extra:
        line["cpu_baseline"] = extra["cpu_baseline"]
    if also:
        line["also"] = also
    print(json.dumps(line))
    if not ok:
        sys.exit(3)


def main_inproc(args):
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    if args.workload != "kem768":
        raise SystemExit("--inproc runs BASELINE configs[4]'s workload (kem768)")
    members = max(1, args.gpus)
    elapsed, ok, extra = run_kem_inproc(args, 768, members)
    metric, unit, wl, _ = WORKLOAD_META["kem768"]
    ndev = len(set(extra["devices"]))
    e = entry("kem768", args, elapsed, ok, extra, members)
    par = "inproc shard%d: one process, %d members on %d device(s) through mlkem_*_multi_dev (no collectives)" % (members, members, ndev)
    if ndev < members:
        par += "; REHEARSAL: members share devices - not a scaling measurement"
    line = {"metric": metric, "value": e["value"], "unit": unit, "n_gpus": members, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": e["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": KEM_DTYPE,
            "data": "synthetic", "config": {"workload": wl % args.batch, "batch_per_gpu": args.batch, "parallelism": par,
                                            "chunk_items": extra.get("chunk_items"), "member_devices": extra["devices"]},
            "correct": ok, "per_gpu": extra["per_gpu_all"], "roofline": e["roofline"], "kernels": e["kernels"]}
    anchor = scaling_anchor(e["value"], extra["per_gpu_all"])
    if anchor:
        line["scaling_anchor"] = anchor
    print(json.dumps(line))
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
