#!/bin/bash
# Turn the raw output of tools/profile_round.sh <tag> (gpurun_out/prof_<tag>/) into the committed summaries of the round
#   profiles/<tag>_<wl>_kernel_stats.csv, profiles/<tag>_pmc_traffic_<wl>.json (tied to the kernel sources by source_id and
#   stamped with the commit), profiles/<tag>_pmc_sq_<wl>.txt, profiles/<tag>_bench_*.json, profiles/<tag>_rehearsal_gpus2.json,
#   profiles/<tag>_inproc8.json -- and regenerate the tables of DESIGN.md / README.md from them (tools/gen_results.py).
# Refuses to run on a dirty kernel tree: a profile of uncommitted sources cannot be reproduced from a commit.
# Usage: tools/refresh_profiles.sh <tag>
T=${1:?tag}; D=gpurun_out/prof_$T
if [ -n "$(git status --porcelain -- crystals-kyber_amd/csrc include bench.py)" ]; then
  echo "kernel sources / bench.py differ from HEAD: commit first, then profile and refresh" >&2; exit 1
fi
# the raw data must come from THIS build: profile_round.sh leaves the hash of the kernel sources it ran on
want=$(python -c "import bench; print(bench.source_id())")
have=$(cat $D/source_id.txt 2>/dev/null)
if [ "$want" != "$have" ]; then
  echo "gpurun_out/prof_$T was measured on source_id '$have', the tree is '$want': run tools/profile_round.sh $T on the GPU first" >&2; exit 1
fi
for wl in kem768 ntt kem1024; do
  cp $D/kt_$wl/${wl}_kernel_stats.csv profiles/${T}_${wl}_kernel_stats.csv || exit 1
  python tools/pmc_traffic.py --fetch $D/fetch_$wl/${wl}_counter_collection.csv --write $D/write_$wl/${wl}_counter_collection.csv \
      --rdreq $D/rdreq_$wl/${wl}_counter_collection.csv --wrreq $D/wrreq_$wl/${wl}_counter_collection.csv \
      --bench $D/fetch_$wl.bench.json --workload $wl --out profiles/${T}_pmc_traffic_$wl.json | tail -1 || exit 1
  python tools/pmc_summary.py $D/sq_$wl/${wl}_counter_collection.csv --filter k_ --json profiles/${T}_pmc_sq_$wl.json > profiles/${T}_pmc_sq_$wl.txt || exit 1
done
cp $D/bench_default.json profiles/${T}_bench_default.json
cp $D/bench_kem512.json profiles/${T}_bench_kem512.json
cp $D/bench_kem768_shared.json profiles/${T}_bench_shared.json
cp $D/rehearsal_gpus2.json profiles/${T}_rehearsal_gpus2.json
cp $D/inproc8.json profiles/${T}_inproc8.json
cp $D/host_latency.txt profiles/${T}_host_latency.txt
cp $D/stream_bench.json profiles/${T}_stream_bench.json
cp $D/batch_sweep.txt profiles/${T}_batch_sweep_head.txt
cp $D/small_sweep.txt profiles/${T}_small_sweep.txt
cp $D/keygen_latency.txt profiles/${T}_keygen_latency.txt
cp $D/keccak_wave_ubench.txt profiles/${T}_keccak_wave_ubench.txt
cp $D/energy.txt profiles/${T}_energy.txt
cp $D/small_stamps.txt profiles/${T}_small_stamps.txt
cp $D/host_threads.txt profiles/${T}_host_threads.txt
cp $D/soak_small.txt profiles/${T}_soak_small.txt
cp $D/host_path_breakdown.txt profiles/${T}_host_path_breakdown.txt
cp $D/small_kernel_stats.txt profiles/${T}_small_kernel_stats.txt
cp $D/gpu_tier.log profiles/${T}_gpu_tier.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o /tmp/capi_floor.s crystals-kyber_amd/csrc/mlkem_capi.hip 2>/dev/null && \
  python tools/isa_floor.py /tmp/capi_floor.s --out profiles/${T}_isa_floor.json > /dev/null || exit 1
python - $T <<'P'
import json, sys, bench
t = sys.argv[1]
for w in ("kem768", "ntt", "kem1024"):
    d = json.load(open("profiles/%s_pmc_traffic_%s.json" % (t, w)))
    print(w, "source_id", d["source_id"], "git", d["git_head"], "matches build:", d["source_id"] == bench.source_id())
P
python tools/gen_results.py $T
