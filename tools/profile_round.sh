#!/bin/bash
# tools/profile_round.sh <tag> — the rocprofv3 passes a round's profiles/ are made from, run on the GPU box at a COMMITTED
# HEAD (the stamps in the summaries name that commit):
#   per workload (kem768 / ntt / kem1024): kernel-trace + stats of bench.py, two PMC passes (FETCH_SIZE, WRITE_SIZE; separate
#   runs, as MI355X_MICROARCH.md prescribes) whose CSVs tools/pmc_traffic.py turns into profiles/rNN_pmc_traffic_*.json, and
#   an SQ pass (issue utilisation);
#   the default bench line, the two-rank rehearsal (torch.distributed.run, per_gpu) and the in-process 8-member line;
#   the host-pointer call latency, the PCIe-inclusive streaming rate and the batch-size sweep.
# Output: gpurun_out/prof_<tag>/.   The program after `--` is python3 itself (no env / bash -c hop under the profiler).
# Usage: tools/profile_round.sh <tag> [lines|prof|all]   (two gpurun calls of <= 20 minutes: `lines` = bench lines, sweeps, energy,
#        test tier; `prof` = the rocprofv3 passes)
set -u
TAG=${1:-r04}
PART=${2:-all}
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
# the bench lines come FIRST, on the box as the driver finds it (a GPU that has been under load for minutes clocks ~3 % lower)
cd "$ROOT"
rm -f "$OUT/source_id.txt"
python3 -c "import bench; print(bench.source_id())" > "$OUT/source_id.txt"   # refresh_profiles.sh checks it against the tree it stamps
if [ "$PART" != "prof" ]; then
echo "== bench lines" >&2
timeout -k 10 400 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || { echo "FAILED default bench" >&2; exit 1; }
for wl in kem512 kem768_shared; do
    timeout -k 10 300 python3 bench.py --workload $wl > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err" || { echo "FAILED $wl" >&2; exit 1; }
done
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse \
    > "$OUT/rehearsal_gpus2.log" 2> "$OUT/rehearsal_gpus2.err" || { echo "FAILED rehearsal" >&2; exit 1; }
grep -h '^{' "$OUT/rehearsal_gpus2.log" | tail -1 > "$OUT/rehearsal_gpus2.json"
timeout -k 10 400 python3 bench.py --inproc --gpus 8 > "$OUT/inproc8.json" 2> "$OUT/inproc8.err" || { echo "FAILED inproc" >&2; exit 1; }
# host-pointer path: call latency at 1 / 64 / 1024 items, PCIe-inclusive streaming rate, and the device-resident batch sweep
timeout -k 10 200 python3 tools/host_latency.py 2>/dev/null > "$OUT/host_latency.txt" || { echo "FAILED host_latency" >&2; exit 1; }
timeout -k 10 400 python3 tools/stream_bench.py 2>/dev/null | grep '^{' | tail -1 > "$OUT/stream_bench.json" || { echo "FAILED stream_bench" >&2; exit 1; }
timeout -k 10 400 bash tools/batch_sweep.sh > "$OUT/batch_sweep.txt" 2>/dev/null || { echo "FAILED batch_sweep" >&2; exit 1; }
# round 4: small calls (one workgroup per item against the batch path), KeyGen latency, the wave-wide Keccak, energy by kernel family
timeout -k 10 400 bash tools/small_sweep.sh > "$OUT/small_sweep.txt" 2>/dev/null || { echo "FAILED small_sweep" >&2; exit 1; }
timeout -k 10 100 python3 tools/keygen_latency.py 2>/dev/null | grep "per call" > "$OUT/keygen_latency.txt" || { echo "FAILED keygen_latency" >&2; exit 1; }
timeout -k 10 100 ./tools/keccak_wave_ubench.bin > "$OUT/keccak_wave_ubench.txt" 2>&1 || { echo "FAILED keccak_wave_ubench" >&2; exit 1; }
timeout -k 10 100 ./tools/small_stamps.bin > "$OUT/small_stamps.txt" 2>&1 || { echo "FAILED small_stamps" >&2; exit 1; }
timeout -k 10 100 ./tools/host_path_breakdown.bin > "$OUT/host_path_breakdown.txt" 2>&1 || { echo "FAILED host_path_breakdown" >&2; exit 1; }
{ echo "# C++ host threads (tools/host_threads_c.cpp): default (lanes + combining), lanes only, one engine"; timeout -k 10 200 ./tools/host_threads_c.bin && MLKEM_HOST_COMBINE=0 timeout -k 10 200 ./tools/host_threads_c.bin && MLKEM_HOST_COMBINE=0 MLKEM_HOST_LANES=0 timeout -k 10 200 ./tools/host_threads_c.bin && echo "# Python host threads (tools/host_threads.py): default" && timeout -k 10 200 python3 tools/host_threads.py 2>/dev/null | grep threads=; } > "$OUT/host_threads.txt" || { echo "FAILED host_threads" >&2; exit 1; }
timeout -k 10 200 python3 tools/soak_small.py 60 2>/dev/null | grep "small-call soak" > "$OUT/soak_small.txt" || { echo "FAILED soak_small" >&2; exit 1; }
timeout -k 10 300 python3 tools/energy_probe.py 2>/dev/null > "$OUT/energy.txt" || { echo "FAILED energy_probe" >&2; exit 1; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q 2>&1 | tail -3 > "$OUT/gpu_tier.log" || { echo "FAILED gpu tier" >&2; cat "$OUT/gpu_tier.log" >&2; exit 1; }
fi
[ "$PART" = "lines" ] && { ls "$OUT"; exit 0; }
cd /tmp
for wl in kem768 ntt kem1024; do
    run kt_$wl --kernel-trace --stats --output-format csv -d "$OUT/kt_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also || exit 1
    run fetch_$wl --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also --steps 3 --warmup 1 || exit 1
    run write_$wl --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also --steps 3 --warmup 1 || exit 1
    # exact request-size counters (tools/pmc_traffic.py: read = 128 B x RDREQ_128B + ..., write = 64 B x WRREQ_64B + ...)
    run rdreq_$wl --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$OUT/rdreq_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also --steps 3 --warmup 1 || exit 1
    run wrreq_$wl --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d "$OUT/wrreq_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also --steps 3 --warmup 1 || exit 1
    # SQ view of the same command: issue utilisation (VALU instructions per SIMD-cycle), waiting, LDS use
    run sq_$wl --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d "$OUT/sq_$wl" -o $wl -- python3 "$ROOT/bench.py" --workload $wl --no-cpu --no-also --steps 3 --warmup 1 || exit 1
done
bash "$ROOT/tools/small_profile.sh" > /dev/null || { echo "FAILED small_profile" >&2; exit 1; }
cp "$ROOT/gpurun_out/prof_small/summary.txt" "$OUT/small_kernel_stats.txt"
find "$OUT" -name '*_kernel_trace.csv' -delete     # the per-dispatch traces are not used (stats + counter_collection are) and would not fit the 64 MiB that travel back
find "$OUT" -name '*.csv' | sed "s|$ROOT/||" | sort
du -sh "$OUT"
