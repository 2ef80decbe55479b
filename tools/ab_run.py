#!/usr/bin/env python3
"""tools/ab_run.py — alternate several configurations of bench.py on ONE box (box-to-box spread is +-2-5 %, larger than most of
the effects looked for) and print one line per run: step time, shader clock and socket power sampled by bench.py right behind
the timed region, and the per-kernel HIP-event averages.

  tools/ab_run.py [--workload kem768] [--rounds 3] [--steps 20] name[:KEY=VALUE,...] ...

KEY = lib (path of a libmlkem_amd.so build -> MLKEM_LIB_PATH), chunk (bench.py --chunk), anything else = environment variable.
Raw lines go to gpurun_out/ab/<name>.<round>.json."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="kem768")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    out = os.path.join(ROOT, "gpurun_out", "ab")
    os.makedirs(out, exist_ok=True)
    cfgs = []
    for c in a.configs:
        name, _, rest = c.partition(":")
        kv = dict(x.split("=", 1) for x in rest.split(",") if x)
        cfgs.append((name, kv))
    for r in range(1, a.rounds + 1):
        for name, kv in cfgs:
            env = dict(os.environ)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", a.workload, "--no-cpu", "--no-also", "--steps", str(a.steps)]
            for k, v in kv.items():
                if k == "lib":
                    env["MLKEM_LIB_PATH"] = os.path.join(ROOT, v)
                elif k == "chunk":
                    cmd += ["--chunk", v]
                else:
                    env[k] = v
            p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                print("%-12s %d FAILED rc=%d %s" % (name, r, p.returncode, p.stderr[-300:].replace("\n", " | ")), flush=True)
                continue
            open(os.path.join(out, "%s.%d.json" % (name, r)), "w").write(lines[-1])
            d = json.loads(lines[-1])
            cp = d["roofline"].get("clock_power") or {}
            clk = (cp.get("sclk_mhz") or {}).get("median")
            pw = (cp.get("socket_w") or {}).get("median")
            ks = " ".join("%s=%.4f" % (n.replace("k_", ""), v["ms_avg"]) for n, v in d["kernels"].items() if v["ms_avg"] > 0.05)
            print("%-12s %d ms/step %.4f %s sclk %s MHz %s W | %s" % (name, r, d["ms_per_step"], "ok" if d["correct"] else "WRONG", clk, pw, ks), flush=True)


if __name__ == "__main__":
    main()
