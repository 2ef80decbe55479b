#!/usr/bin/env python3
"""tools/energy_probe.py — does "time = dynamic energy / (P_limit - P_idle)" describe the full-KEM pass?  (round-4 check of the
model DESIGN.md stopped on.)

ML-KEM-768, 2^20 items, default chunking.  Every kernel family of Encaps + Decaps is looped ALONE for a few seconds at exactly the
shapes it has inside the step (mlkem_ctx_debug_stages restricts the batch path to one family; the scratch keeps the data of the
last full pass), while a side thread reads the socket power (amdgpu hwmon power1_input, uW) and the shader clock every 20 ms:
   joules per step-share = mean power x time per pass of that family ; dynamic = (mean power - idle power) x time.
Then the whole step is looped the same way and compared with
   (a) sum of the families' stand-alone times                      -- what issue alone predicts,
   (b) sum of the families' DYNAMIC joules / (P_cap - P_idle)      -- what the energy model predicts,
   (c) per family max(stand-alone time, dynamic joules / (P_cap - P_idle)) summed.
Output: a table + one JSON line (joules_per_pair feeds bench.py's `joules_per_unit` via profiles/r04_energy.json)."""
import glob
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

SECONDS = float(os.environ.get("ENERGY_SECONDS", "4"))
dev = torch.device("cuda", 0)
p = torch.cuda.get_device_properties(dev)
hw = glob.glob("/sys/bus/pci/devices/%04x:%02x:%02x.0/hwmon/hwmon*" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))[0]
f_clk, f_pw, f_cap = (os.path.join(hw, n) for n in ("freq1_input", "power1_input", "power1_cap"))
cap_w = int(open(f_cap).read()) / 1e6


class Sampler:
    def __init__(self):
        self.s, self.stop = [], threading.Event()
        self.th = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop.is_set():
            self.s.append((time.perf_counter(), int(open(f_pw).read()) / 1e6, int(open(f_clk).read()) / 1e6))
            time.sleep(0.02)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        self.th.join(timeout=2)

    def mean(self, t0, t1):
        rows = [(w, c) for (t, w, c) in self.s if t0 <= t <= t1]
        return (sum(r[0] for r in rows) / len(rows), sum(r[1] for r in rows) / len(rows), len(rows)) if rows else (float("nan"), float("nan"), 0)


pkg = ge.load_package()
n = 1 << 20
eng = pkg.MLKEM(768, device=0)
d, z, m = (bench.device_seeds(lbl, 0, n, dev) for lbl in ("mlkem-bench-d", "mlkem-bench-z", "mlkem-bench-m"))
ek, dk = eng.keygen(d, z)
c = torch.empty((n, eng.c_len), dtype=torch.uint8, device=dev)
K = torch.empty((n, 32), dtype=torch.uint8, device=dev)
K2 = torch.empty((n, 32), dtype=torch.uint8, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
c_scratch, K_scratch = torch.empty_like(c), torch.empty_like(K)
lib, ctx = eng.lib, eng._ctx


def full():
    eng.encaps(ek, m, c=c, K=K)
    eng.decaps(dk, c, K=K2, status=st)


def enc_only(mask):
    def f():
        lib.mlkem_ctx_debug_stages(ctx, mask)
        eng.encaps(ek, m, c=c_scratch, K=K_scratch)
        lib.mlkem_ctx_debug_stages(ctx, 15)
    return f


def dec_only(mask):
    def f():
        lib.mlkem_ctx_debug_stages(ctx, mask)
        eng.decaps(dk, c, K=K_scratch, status=st)
        lib.mlkem_ctx_debug_stages(ctx, 15)
    return f


full()
torch.cuda.synchronize()
assert torch.equal(K, K2)
families = [("encaps: k_hash_encaps", enc_only(1)), ("encaps: sampler (k_sample_main + tails) x4", enc_only(2)),
            ("encaps: k_encrypt x4", enc_only(4)), ("decaps: k_decrypt", dec_only(8)), ("decaps: k_hash_decaps", dec_only(1)),
            ("decaps: sampler x4", dec_only(2)), ("decaps: k_encrypt_cmp x4", dec_only(4)), ("WHOLE STEP (encaps + decaps)", full)]
res = {}
with Sampler() as smp:
    time.sleep(2.5)
    t_idle0, t_idle1 = time.perf_counter() - 2.0, time.perf_counter()
    for name, fn in families:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = 0
        while time.perf_counter() - t0 < SECONDS:
            for _ in range(4):
                fn()
            it += 4
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        res[name] = (it, t0, t1)
        time.sleep(1.5)   # cool-down between families: each starts from the same thermal state
    t_idle2, t_idle3 = time.perf_counter() - 1.2, time.perf_counter()
p_idle = smp.mean(t_idle0, t_idle1)[0]
p_idle_end = smp.mean(t_idle2, t_idle3)[0]
rows = {}
print("ML-KEM-768, 2^20 items per pass; socket power cap %.0f W; idle %.0f W before, %.0f W after (%d hwmon samples at 20 ms)" % (
    cap_w, p_idle, p_idle_end, len(smp.s)))
print("%-46s %9s %8s %8s %9s %9s" % ("kernel family looped alone", "ms/pass", "W", "MHz", "J/pass", "dyn J"))
for name, (it, t0, t1) in res.items():
    w, clk, ns = smp.mean(t0 + 0.6, t1)   # skip the ramp
    ms = 1e3 * (t1 - t0) / it
    rows[name] = {"ms_per_pass": ms, "watts": w, "sclk_mhz": clk, "joules_per_pass": w * ms * 1e-3, "dyn_joules_per_pass": (w - p_idle) * ms * 1e-3,
                  "samples": ns}
    print("%-46s %9.3f %8.0f %8.0f %9.2f %9.2f" % (name, ms, w, clk, rows[name]["joules_per_pass"], rows[name]["dyn_joules_per_pass"]))
whole = rows["WHOLE STEP (encaps + decaps)"]
fam = [v for k, v in rows.items() if not k.startswith("WHOLE")]
sum_t = sum(v["ms_per_pass"] for v in fam)
sum_dyn = sum(v["dyn_joules_per_pass"] for v in fam)
head = cap_w - p_idle
pred_energy = 1e3 * sum_dyn / head
pred_max = sum(max(v["ms_per_pass"], 1e3 * v["dyn_joules_per_pass"] / head) for v in fam)
kpke = sum(v["dyn_joules_per_pass"] for k, v in rows.items() if "encrypt" in k or "decrypt" in k)
print("whole step measured                         %.3f ms at %.0f W, %.0f MHz: %.2f J per 2^20 pairs = %.2f uJ per pair (dynamic %.2f J)" % (
    whole["ms_per_pass"], whole["watts"], whole["sclk_mhz"], whole["joules_per_pass"], whole["joules_per_pass"] / n * 1e6, whole["dyn_joules_per_pass"]))
print("(a) sum of stand-alone times                %.3f ms  (%+.1f %% vs measured)" % (sum_t, 100 * (sum_t / whole["ms_per_pass"] - 1)))
print("(b) sum of dynamic joules / (cap - idle)    %.3f ms  (%+.1f %%)   [%.2f J / %.0f W]" % (pred_energy, 100 * (pred_energy / whole["ms_per_pass"] - 1), sum_dyn, head))
print("(c) sum of max(time alone, joules / head)   %.3f ms  (%+.1f %%)" % (pred_max, 100 * (pred_max / whole["ms_per_pass"] - 1)))
print("sum of the families' dynamic joules %.2f vs the whole step's %.2f (%+.1f %%); K-PKE kernels' share of the dynamic joules: %.1f %%" % (
    sum_dyn, whole["dyn_joules_per_pass"], 100 * (sum_dyn / whole["dyn_joules_per_pass"] - 1), 100 * kpke / sum_dyn))
print(json.dumps({"source_id": bench.source_id(), "batch": n, "power_cap_w": cap_w, "idle_w": p_idle, "families": rows,
                  "joules_per_pair": whole["joules_per_pass"] / n, "dynamic_joules_per_pair": whole["dyn_joules_per_pass"] / n,
                  "pred_ms": {"sum_alone": sum_t, "energy_model": pred_energy, "max_model": pred_max}, "measured_ms": whole["ms_per_pass"],
                  "kpke_dynamic_share": kpke / sum_dyn}))
eng.close()
