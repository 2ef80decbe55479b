#!/bin/bash
# tools/profile_round.sh <tag> — the rocprofv3 passes a round's profiles/ are made from, run on the GPU box:
#   kernel-trace + stats of bench.py for kem768 / ntt / kem1024, and two PMC passes each (FETCH_SIZE, WRITE_SIZE; separate
#   runs, as MI355X_MICROARCH.md prescribes) whose CSVs tools/pmc_traffic.py turns into profiles/rNN_pmc_traffic_*.json.
# Output: gpurun_out/prof_<tag>/.   The program after `--` is python3 itself (no env / bash -c hop under the profiler).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # name, rocprof args..., -- bench args
    local name=$1; shift
    echo "== $name" >&2
    timeout -k 10 400 rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name (see $OUT/$name.log)" >&2; return 1; }
    grep -h '^{' "$OUT/$name.log" | tail -1 > "$OUT/$name.bench.json"
}
for wl in kem768 ntt kem1024; do
    steps=10; [ $wl = ntt ] && steps=100
    run kt_$wl --kernel-trace --stats --output-format csv -d "$OUT/kt_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also --steps $steps --warmup 2 || exit 1
    run fetch_$wl --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also --steps 3 --warmup 1 || exit 1
    run write_$wl --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also --steps 3 --warmup 1 || exit 1
    # SQ view of the same command: issue utilisation (VALU instructions per SIMD-cycle), waiting, LDS use
    run sq_$wl --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d "$OUT/sq_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also --steps 3 --warmup 1 || exit 1
done
find "$OUT" -name '*.csv' | sed "s|$ROOT/||" | sort
du -sh "$OUT"
