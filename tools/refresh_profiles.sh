#!/bin/bash
# Turn the raw output of tools/profile_round.sh <tag> (gpurun_out/prof_<tag>/) into the committed round-2 summaries:
#   profiles/r02_<wl>_kernel_stats.csv, profiles/r02_pmc_traffic_<wl>.json (tied to the kernel sources by source_id),
#   profiles/r02_pmc_sq_<wl>.txt.     Usage: tools/refresh_profiles.sh <tag>
T=${1:?tag}; D=gpurun_out/prof_$T
for wl in kem768 ntt kem1024; do
  cp $D/kt_$wl/${wl}_kernel_stats.csv profiles/r02_${wl}_kernel_stats.csv || exit 1
  python tools/pmc_traffic.py --fetch $D/fetch_$wl/${wl}_counter_collection.csv --write $D/write_$wl/${wl}_counter_collection.csv \
      --bench $D/fetch_$wl.bench.json --workload $wl --out profiles/r02_pmc_traffic_$wl.json | tail -1 || exit 1
  python tools/pmc_summary.py $D/sq_$wl/${wl}_counter_collection.csv --filter k_ > profiles/r02_pmc_sq_$wl.txt || exit 1
done
python - <<'P'
import json, bench
for w in ("kem768", "ntt", "kem1024"):
    d = json.load(open("profiles/r02_pmc_traffic_%s.json" % w))
    print(w, "source_id", d["source_id"], "matches build:", d["source_id"] == bench.source_id())
P
