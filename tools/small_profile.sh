#!/bin/bash
# tools/small_profile.sh — rocprofv3 evidence for the small-call kernels (mlkem_small.hpp): kernel-trace stats of Encaps + Decaps at
# 1, 64, 512 and 768 items per call (bench.py --batch N), and an SQ pass at 64 items (waves, VALU / LDS instructions, waiting).
# Output: gpurun_out/prof_small/ ; summaries go to profiles/r04_small_kernel_stats.txt.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1 64 512 768; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$n -o s -- python3 $ROOT/bench.py --batch $n --steps 300 --warmup 10 --no-cpu --no-also > $OUT/kt_$n.log 2>&1 || { echo FAILED kt_$n; exit 1; }
done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
  -d $OUT/sq_64 -o s -- python3 $ROOT/bench.py --batch 64 --steps 50 --warmup 5 --no-cpu --no-also > $OUT/sq_64.log 2>&1 || { echo FAILED sq; exit 1; }
find $OUT -name '*_kernel_trace.csv' -delete
cd $ROOT
{
  echo "Small-call kernels under rocprofv3 (tools/small_profile.sh): ML-KEM-768 Encaps + Decaps of N items per call, 310 calls each."
  for n in 1 64 512 768; do
    echo "== N = $n: --kernel-trace --stats (name, calls, average ns, min, max)"
    python3 - $OUT/kt_$n/s_kernel_stats.csv <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mlkem::" in r["Name"]:
        print("  %-62s calls %5s avg %9.0f ns min %9s max %9s" % (r["Name"].replace("void mlkem::", "")[:62], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
P
  done
  echo "== N = 64: SQ counters per dispatch (tools/pmc_summary.py)"
  python3 tools/pmc_summary.py $OUT/sq_64/s_counter_collection.csv --filter small | cut -c1-600
} > $OUT/summary.txt
cat $OUT/summary.txt
