#!/bin/bash
# A/B/C... of several builds of libmlkem_amd.so on the SAME box (box-to-box spread is +-2-5 %, larger than most kernel
# changes):   tools/ab_bench.sh <workload> <rounds> <lib1.so> <lib2.so> [...]
# round-robin over the builds; prints ms/step and the per-kernel HIP-event averages of every run.
WL=$1; R=$2; shift 2
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    MLKEM_LIB_PATH=$PWD/$lib python bench.py --workload $WL --no-cpu --no-also --steps ${STEPS:-20} > gpurun_out/ab/$tag$r.json 2> gpurun_out/ab/$tag$r.err || true
    python - $tag $r <<'P'
import json,sys
d=json.load(open('gpurun_out/ab/%s%s.json'%(sys.argv[1],sys.argv[2])))
k=d['kernels']
print('%-10s'%sys.argv[1],sys.argv[2],'ms/step %.3f'%d['ms_per_step'],'ok' if d['correct'] else 'WRONG',' '.join('%s=%.3f'%(n.replace('k_',''),v['ms_avg']) for n,v in k.items() if v['ms_avg']>0.05))
P
  done
done
