#!/bin/bash
# usage: icache_probe.sh <lib> <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export MLKEM_LIB_PATH=$ROOT/$1
OUT=$ROOT/gpurun_out/ic_$2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o kem768 -- python3 $ROOT/bench.py --workload kem768 --no-cpu --no-also --steps 3 --warmup 1 > $OUT/log_a.txt 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/a/kem768_counter_collection.csv --filter k_sample_main | cut -c1-600
python tools/pmc_summary.py $OUT/a/kem768_counter_collection.csv --filter "k_hash_decaps" | cut -c1-600
tail -3 $OUT/log_a.txt | cut -c1-300
