#!/bin/bash
# usage: sq_probe.sh <lib> <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export MLKEM_LIB_PATH=$ROOT/$1
OUT=$ROOT/gpurun_out/sq_$2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o kem768 -- python3 $ROOT/bench.py --workload kem768 --no-cpu --no-also --steps 3 --warmup 1 > $OUT/log.txt 2>&1
cd $ROOT; python tools/pmc_summary.py $OUT/kem768_counter_collection.csv --filter k_encrypt | cut -c1-400
