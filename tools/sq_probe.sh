#!/bin/bash
# tools/sq_probe.sh <lib> <tag> [kernel filter] — SQ counters of one build on the ML-KEM-768 workload (two PMC passes)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export MLKEM_LIB_PATH=$ROOT/$1
OUT=$ROOT/gpurun_out/sq_$2
FILT=${3:-k_encrypt}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o kem768 -- python3 $ROOT/bench.py --workload kem768 --no-cpu --no-also --steps 3 --warmup 1 > $OUT/log_a.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/b -o kem768 -- python3 $ROOT/bench.py --workload kem768 --no-cpu --no-also --steps 3 --warmup 1 > $OUT/log_b.txt 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/a/kem768_counter_collection.csv --filter $FILT | cut -c1-420
python tools/pmc_summary.py $OUT/b/kem768_counter_collection.csv --filter $FILT | cut -c1-420
