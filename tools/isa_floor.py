#!/usr/bin/env python3
"""tools/isa_floor.py — the ISSUE FLOOR of every hot kernel from the library's own ISA: SIMD cycles per VALU instruction its
instruction mix costs when nothing stalls.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o capi.s crystals-kyber_amd/csrc/mlkem_capi.hip
  python tools/isa_floor.py capi.s [--out profiles/r04_isa_floor.json]

Prices (MI355X_MICROARCH.md 'Per-instruction cycle constants' + tools/valu_ubench*.hip, profiles/r01_valu_ubench.txt): a wave64
VALU instruction occupies its SIMD for 2 cycles; the half-rate classes (v_pk_*, v_alignbit, conversions, 64-bit shifts,
32-bit integer multiplies, v_max3 / v_floor ...) for 4; a DPP-modified instruction for 3.2 (1.6 plain ones).  The K-PKE kernels
and the register NTT are straight-line code, so the static mix IS the dynamic mix; the Keccak kernels spend > 90 % of their
instructions in the 24-round loop, so their mix is that loop's (the largest innermost loop of the kernel).
bench.py divides these floors by the cycles per VALU instruction the SQ counters measure (roofline.issue)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HALF = re.compile(r'^(v_pk_|v_alignbit|v_cvt_|v_mul_lo|v_mul_hi|v_lshl_add_u64|v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64|v_mad_u64|v_mad_i64|'
                  r'v_permlane|v_readlane|v_readfirstlane|v_mbcnt|v_floor|v_max3|v_min3|v_med3|v_rndne|v_trunc|v_fract)')
LOOP_KERNELS = ("k_sample_main", "k_hash_", "k_sponge_raw", "k_sample")   # mix = the Keccak round loop


def price(op):
    return 4.0 if HALF.match(op) else 3.2 if op.endswith("_dpp") else 2.0


def main():
    text = open(sys.argv[1]).read()
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    names = re.findall(r'^(_Z\w+):', text, re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    rows = {}
    for name, full in zip(names, dem):
        short = full.replace("mlkem::", "").split("(")[0]
        if not short.startswith(("void k_", "k_")):
            continue
        short = short.replace("void ", "")
        m = re.search(r'^' + re.escape(name) + r':(.*?)\.Lfunc_end', text, re.S | re.M)
        lines = [l.strip() for l in m.group(1).split('\n')]
        scope = "whole kernel (straight-line)"
        if short.startswith(LOOP_KERNELS):
            label_at = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r'(\.LBB\d+_\d+):', l)] if mm}
            loops = []
            for i, l in enumerate(lines):
                mm = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.match(r's_branch (\.LBB\d+_\d+)', l)
                if mm and mm.group(1) in label_at and label_at[mm.group(1)] < i:
                    loops.append((label_at[mm.group(1)], i))
            inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
            if inner:
                a, b = max(inner, key=lambda lp: lp[1] - lp[0])
                lines = lines[a:b + 1]
                scope = "largest innermost loop (the Keccak round)"
        c = collections.Counter(l.split()[0] for l in lines if l.startswith("v_"))
        tot = sum(c.values())
        if not tot:
            continue
        rows[short] = {"label": bench.kernel_label(full), "valu": tot, "half_rate": sum(v for k, v in c.items() if HALF.match(k)),
                       "dpp": sum(v for k, v in c.items() if k.endswith("_dpp")),
                       "floor_cycles_per_valu_instr": sum(price(k) * v for k, v in c.items()) / tot, "scope": scope}
        print("%-58s %-20s VALU %5d half-rate %3.0f %% dpp %4d floor %.2f  [%s]" % (short[:58], rows[short]["label"], tot,
              100.0 * rows[short]["half_rate"] / tot, rows[short]["dpp"], rows[short]["floor_cycles_per_valu_instr"], scope))
    if out_path:
        json.dump({"source_id": bench.source_id(), "prices": {"full_rate": 2.0, "half_rate_classes": 4.0, "dpp": 3.2}, "kernels": rows},
                  open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
