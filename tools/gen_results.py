#!/usr/bin/env python3
"""tools/gen_results.py <tag> — every measured number DESIGN.md / README.md quote for a round, generated from the committed
files under profiles/ (so that prose and profiles cannot drift apart):

  profiles/<tag>_bench_default.json      the default `python bench.py` line (headline + also.ntt + also.kem1024)
  profiles/<tag>_bench_kem512.json, <tag>_bench_shared.json
  profiles/<tag>_<wl>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --workload <wl> --no-cpu --no-also`
  profiles/<tag>_pmc_traffic_<wl>.json   FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py)
  profiles/<tag>_pmc_sq_<wl>.txt         SQ counters (tools/pmc_summary.py)
  profiles/<tag>_rehearsal_gpus2.json, <tag>_inproc8.json   the N > 1 forms of the bench line on the one GPU
  profiles/<tag>_gpu_tier.log            `pytest -m gpu` on the GPU box (test count)
  profiles/<tag>_batch_sweep_head.txt, <tag>_host_latency.txt, <tag>_stream_bench.json   batch-size sweep, host-pointer latency and streaming rate

Writes profiles/<tag>_RESULTS.md -- the ONE generated results block of a round; README.md and DESIGN.md link to it."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(ROOT, "profiles")
HBM = 8000.0
ALGO = {"kem768": 5856, "kem1024": 12768, "kem512": 6560, "ntt": 2048}


def load(name):
    path = os.path.join(P, "%s_%s" % (TAG, name))
    return json.load(open(path)) if os.path.exists(path) else None


def short(kernel_name):
    s = re.sub(r"^void ", "", kernel_name).replace("mlkem::", "")
    base = s.split("(")[0]
    m = re.match(r"(k_\w+)<(.*)>$", base)
    if not m:
        return base
    name, targs = m.group(1), m.group(2)
    if name in ("k_encrypt", "k_encrypt2"):
        return "k_encrypt_cmp" if re.search(r",\s*true$", targs) else "k_encrypt"
    if name == "k_keygen2":
        return "k_keygen"
    if name == "k_ntt4_batch":
        return "k_intt_batch" if targs == "true" else "k_ntt_batch"
    if name == "k_decrypt4":
        return "k_decrypt"
    return name


def stats(wl):
    """label -> (calls, avg ms) from the rocprofv3 stats csv; k_sample (list mode) is the bench's k_sample_restart"""
    path = os.path.join(P, "%s_%s_kernel_stats.csv" % (TAG, wl))
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        if "mlkem::" not in n:
            continue
        lab = short(n)
        lab = {"k_sample": "k_sample_restart", "k_sample_resume": "k_sample_tail"}.get(lab, lab)
        calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
        c0, t0 = out.get(lab, (0, 0.0))
        out[lab] = (c0 + calls, t0 + tot)
    return {k: (c, t / c / 1e6) for k, (c, t) in out.items()}


def fmt(x, d=3):
    return ("%%.%df" % d) % x


def bench_table(j, wl, st):
    k = j["kernels"]
    tot = sum(v["ms_total"] for v in k.values())
    rows = ["| kernel | launches / step | HIP-event avg (ms) | ms / step | share | rocprofv3 avg (ms) over calls |", "|---|---|---|---|---|---|"]
    for name, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_total"]):
        rp = st.get(name)
        rows.append("| `%s` | %d | %s | %s | %.1f %% | %s |" % (name, v["launches"], fmt(v["ms_avg"], 4), fmt(v["ms_total"]), 100 * v["ms_total"] / tot,
                                                                   ("%s over %d" % (fmt(rp[1], 4), rp[0])) if rp else "—"))
    return "\n".join(rows)


def roofline_lines(j, wl, traffic):
    rf = j["roofline"]
    out = ["* roofline (dominant kernel by the algorithmic-bytes rule): **%.1f GB/s = %.3f of the 8 TB/s HBM peak** — %s" % (rf["achieved"], rf["frac"], rf["scope"])]
    if "whole_pass" in rf:
        out.append("* whole pass: %.1f GB/s = %.4f of peak (%s)" % (rf["whole_pass"]["achieved"], rf["whole_pass"]["frac"], rf["whole_pass"]["scope"]))
    if traffic:
        algo = ALGO[wl] * traffic["batch"]
        out.append("* HBM traffic by PMC (`profiles/%s_pmc_traffic_%s.json`, git %s, source_id %s): %.2f GB per step corrected (raw %.2f GB) against %.2f GB algorithmic = **%.2f x**" % (
            TAG, wl, traffic["git_head"], traffic["source_id"], traffic["hbm_bytes_per_step_corrected"] / 1e9, traffic["hbm_bytes_per_step_raw"] / 1e9,
            algo / 1e9, traffic["hbm_bytes_per_step_corrected"] / algo))
        if traffic.get("hbm_bytes_per_step_exact"):
            out.append("  * exact, from the request-size counters (128 B x RDREQ_128B + ..., 64 B x WRREQ_64B + ...): %.2f GB per step — every read "
                       "request is a 128-byte request, so the doubled FETCH_SIZE is the truth (`profiles/r04_traffic_calibration.txt`)" % (
                           traffic["hbm_bytes_per_step_exact"] / 1e9))
        dom = rf.get("dominant_kernel")
        pl = traffic.get("per_label", {}).get(dom)
        if pl:
            out.append("  * dominant kernel `%s`: %.3f GB per launch corrected (raw %.3f GB), %d launches per step" % (
                dom, pl["bytes_per_launch_corrected"] / 1e9, pl["bytes_per_launch_raw"] / 1e9, pl["launches_per_step"]))
        of = [(k, v) for k, v in traffic.get("per_kernel", {}).items() if v.get("over_fetch")]
        if of:
            out.append("  * read bytes / bytes of the kernel's input arguments: " + ", ".join("`%s` %.2f" % (k.split("<")[0], v["over_fetch"]) for k, v in of))
    if rf.get("issue"):
        out.append("* issue fraction (floor cycles per VALU instruction of the kernel's ISA mix / measured; `bound: %s`): " % rf.get("bound") + ", ".join(
            "`%s` %.2f (%.2f / %.2f at %.1f waves)" % (k, v["issue_frac"], v["floor"], v["cycles_per_valu_instr"], v.get("resident_waves_per_simd") or 0)
            for k, v in sorted(rf["issue"]["kernels"].items(), key=lambda kv: -kv[1]["issue_frac"])))
    cp = rf.get("clock_power")
    if cp:
        out.append("* clock / power behind the timed region: shader clock median %s MHz (nominal %s), socket power median %s W (cap %s W)" % (
            cp["sclk_mhz"]["median"], cp["sclk_nominal_mhz"], cp["socket_w"]["median"], cp["power_cap_w"]))
    return "\n".join(out)


def cpu_lines(cb):
    if not cb:
        return ""
    out = []
    legs = cb.get("legs") or {"(single leg)": cb}
    for name, v in legs.items():
        out.append("  * `%s`: %s %s on %d cores (%.2f per core), outputs match GPU: %s — %s" % (
            name, ("%.4g" % v["value"]), v["unit"], v["cores"], v.get("per_core", float("nan")), v["outputs_match_gpu"], v["sample"]))
    return "\n".join(out)


def sq_lines(wl):
    path = os.path.join(P, "%s_pmc_sq_%s.txt" % (TAG, wl))
    if not os.path.exists(path):
        return ""
    rows = ["| kernel | cycles per VALU instruction | resident waves / SIMD |", "|---|---|---|"]
    for ln in open(path):
        m = re.search(r"^(.+?) dispatches=.*cycles_per_valu_instr=([\d.]+) resident_waves_per_simd=([\d.]+)", ln)
        if m:
            rows.append("| `%s` | %s | %s |" % (m.group(1), m.group(2), m.group(3)))
    return "\n".join(rows) if len(rows) > 2 else ""


def main():
    d = load("bench_default.json")
    if d is None:
        raise SystemExit("profiles/%s_bench_default.json missing" % TAG)
    L = []
    L.append("_Generated by `tools/gen_results.py %s` from the files under `profiles/%s_*`; do not edit by hand._\n" % (TAG, TAG))
    L.append("**Headline (BASELINE configs[2])**: `python bench.py` → **%.3g %s**, %.2f ms per step of 2^20 pairs, `correct: %s` (`profiles/%s_bench_default.json`).\n" % (
        d["value"], d["unit"], d["ms_per_step"], d["correct"], TAG))
    if d.get("joules_per_unit"):
        L.append("Socket energy: **%.1f uJ per pair** (%.0f W median x step time / 2^20; `profiles/%s_energy.txt` splits it by kernel family).\n" % (
            1e6 * d["joules_per_unit"], d["energy"]["socket_w"], TAG))
    st = stats("kem768")
    L.append(bench_table(d, "kem768", st))
    L.append("")
    L.append(roofline_lines(d, "kem768", load("pmc_traffic_kem768.json")))
    L.append("* CPU baseline legs (same items, bytes compared with the GPU's):\n" + cpu_lines(d.get("cpu_baseline")))
    L.append("")
    for wl, title in (("ntt", "configs[1]: NTT fwd+inv"), ("kem1024", "configs[3]: ML-KEM-1024 keygen+encaps+decaps")):
        e = d.get("also", {}).get(wl)
        if not e:
            continue
        L.append("**%s** (same process, `also.%s`): **%.3g %s**, %.3f ms per step, `correct: %s`\n" % (title, wl, e["value"], e["unit"], e["ms_per_step"], e["correct"]))
        L.append(bench_table(e, wl, stats(wl)))
        L.append("")
        L.append(roofline_lines(e, wl, load("pmc_traffic_%s.json" % wl)))
        if e.get("cpu_baseline"):
            L.append("* CPU baseline:\n" + cpu_lines(e["cpu_baseline"]))
        L.append("")
    sc = d.get("also", {}).get("small_calls")
    if sc:
        L.append("**Small calls** (same process, `also.small_calls`; %s): device-resident %s; host pointers, one item: %.3f; `correct: %s`\n" % (
            sc["unit"], ", ".join("%s items: %.4f" % (k, v) for k, v in sc["device_resident"].items()), sc["host_pointer"]["1"], sc["correct"]))
    for name, title in (("bench_kem512.json", "ML-KEM-512 keygen+encaps+decaps"), ("bench_shared.json", "ML-KEM-768, one key pair for the whole batch (extra workload)")):
        j = load(name)
        if j:
            L.append("**%s**: %.3g %s, %.2f ms per step, `correct: %s` (`profiles/%s_%s`)\n" % (title, j["value"], j["unit"], j["ms_per_step"], j["correct"], TAG, name))
    sq = sq_lines("kem768")
    if sq:
        L.append("**Issue utilisation, ML-KEM-768** (`profiles/%s_pmc_sq_kem768.txt`; 2.0 cycles = a full-rate wave64 instruction on a SIMD-32):\n" % TAG)
        L.append(sq)
        L.append("")
    for name, title in (("rehearsal_gpus2.json", "`bench.py --gpus 2 --rehearse` under torch.distributed.run (two ranks share the one GPU, gloo barrier)"),
                        ("rehearsal_gpus4.json", "`bench.py --gpus 4 --rehearse` (four ranks share the one GPU)"),
                        ("inproc8.json", "`bench.py --inproc --gpus 8` (one process, eight members through `mlkem_*_multi_dev`, all on the one GPU)")):
        j = load(name)
        if not j:
            continue
        L.append("**N > 1 line on the one GPU — %s**: aggregate %.3g %s, %.2f ms per step, `correct: %s`; `per_gpu`:\n" % (title, j["value"], j["unit"], j["ms_per_step"], j["correct"]))
        L.append("| rank | device | %s | ms / step | solo %s (in-job N = 1 anchor) | sclk MHz | socket W | correct |" % (j["unit"], j["unit"]))
        L.append("|---|---|---|---|---|---|---|---|")
        for p in j["per_gpu"]:
            L.append("| %d | %d | %.3g | %.2f | %s | %s | %s | %s |" % (p["rank"], p["device"], p["value"], p["ms_per_step"],
                     ("%.3g" % p["solo_value"]) if p.get("solo_value") else "—", p["sclk_mhz"], p["socket_w"], p["correct"]))
        a = j.get("scaling_anchor")
        if a:
            L.append("\n`scaling_anchor`: efficiency = aggregate / sum of solo values = **%.3f** (ranks sharing ONE GPU: 1/N is the expected figure), per rank %s" % (
                a["efficiency"], ", ".join("%.3f" % x for x in a["per_gpu_efficiency"])))
        L.append("")
    sweep = os.path.join(P, "%s_batch_sweep_head.txt" % TAG)
    if os.path.exists(sweep):
        L.append("**Batch size, device-resident ML-KEM-768 encaps + decaps** (`tools/batch_sweep.sh`, `profiles/%s_batch_sweep_head.txt`; the steps that led here: `profiles/%s_batch_sweep.txt`):\n" % (TAG, TAG))
        L.append("| items per call | pairs/s | ms per encaps + decaps |")
        L.append("|---|---|---|")
        for ln in open(sweep):
            m = re.match(r"n=2\^(\d+)\s+([\d.e+]+) pairs/s\s+([\d.]+) ms/step", ln)
            if m:
                L.append("| 2^%s | %s | %s |" % (m.group(1), m.group(2), m.group(3)))
        L.append("")
    hl = os.path.join(P, "%s_host_latency.txt" % TAG)
    sb = load("stream_bench.json")
    if os.path.exists(hl) or sb:
        L.append("**Host-pointer entry points** (`mlkem_encaps` / `mlkem_decaps` on host buffers, PCIe inside the call):\n")
        if os.path.exists(hl):
            for ln in open(hl):
                if ln.startswith("n="):
                    L.append("* " + ln.strip() + " (`profiles/%s_host_latency.txt`)" % TAG)
        if sb:
            best = lambda kind: max((v, k) for k, v in sb.items() if k.startswith(kind + "_pairs_per_s_chunk_"))
            for kind in ("pinned", "pageable"):
                v, k = best(kind)
                L.append("* streaming %d items, %s buffers: %.3g pairs/s at chunk %s (`profiles/%s_stream_bench.json`)" % (sb["items"], kind, v, k.rsplit("_", 1)[1], TAG))
        L.append("")
    for name, title in (("soak.txt", "Sustained run (`tools/soak.py`)"),
                        ("soak_small.txt", "Sustained run of the small-call kernels, random sizes and parameter sets on two streams (`tools/soak_small.py`)"),
                        ("small_kernel_stats.txt", "Small-call kernels under rocprofv3 (`tools/small_profile.sh`)"),
                        ("small_stamps.txt", "Where a one-item Encaps spends its time, stage by stage (`tools/small_stamps.hip`)"),
                        ("keygen_latency.txt", "Device-resident call latency per parameter set and operation (`tools/keygen_latency.py`)"),
                        ("host_path_breakdown.txt", "Host-pointer call of one item from C: launch floor, device buffers, host buffers (`tools/host_path_breakdown.hip`)"),
                        ("host_threads.txt", "One-item host-pointer pairs from T host threads at once, with and without engine lanes (`tools/host_threads_c.cpp`, `tools/host_threads.py`)"),
                        ("small_limits.txt", "Small-call limit per parameter set: one workgroup per item against the batch path"),
                        ("small_sweep.txt", "Small calls: one workgroup per item (eight / four waves) against the batch path (`tools/small_sweep.sh`)"),
                        ("energy.txt", "Energy by kernel family, each looped alone at its 2^20 shapes (`tools/energy_probe.py`)"),
                        ("keccak_wave_ubench.txt", "Keccak-f[1600] of a lone wave: lane-sliced / half-wave (round 3) / wave-wide (`tools/keccak_wave_ubench.hip`)")):
        path = os.path.join(P, "%s_%s" % (TAG, name))
        if os.path.exists(path):
            body = [ln.rstrip()[:230] for ln in open(path) if not ln.startswith("{")]
            L.append("**%s** (`profiles/%s_%s`):\n\n```\n%s\n```\n" % (title, TAG, name, "\n".join(body)))
    log = os.path.join(P, "%s_gpu_tier.log" % TAG)
    if os.path.exists(log):
        m = re.search(r"(\d+) passed", open(log).read())
        if m:
            L.append("**GPU tier**: %s tests passed on MI355X (`profiles/%s_gpu_tier.log`).\n" % (m.group(1), TAG))
    text = "\n".join(L).rstrip() + "\n"
    open(os.path.join(P, "%s_RESULTS.md" % TAG), "w").write(text)
    print("wrote profiles/%s_RESULTS.md (%d lines)" % (TAG, text.count("\n")))


if __name__ == "__main__":
    main()
