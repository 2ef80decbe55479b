#!/bin/bash
# engine chunk size sweep on one box: tools/chunk_sweep.sh [workload] [chunks...]
WL=${1:-kem768}; shift
mkdir -p gpurun_out/chunk
for c in ${@:-16384 32768 65536 131072 262144 524288}; do
  for ov in 0; do
    python bench.py --workload $WL --chunk $c --no-cpu --no-also --steps 20 > gpurun_out/chunk/c$c.$ov.json 2> gpurun_out/chunk/c$c.$ov.err || { tail -3 gpurun_out/chunk/c$c.$ov.err; exit 1; }
    python - $c $ov <<'P'
import json,sys
d=json.load(open('gpurun_out/chunk/c%s.%s.json'%(sys.argv[1],sys.argv[2])))
print('chunk',sys.argv[1],'overlap',sys.argv[2],'ms/step %.3f'%d['ms_per_step'],'ok' if d['correct'] else 'WRONG')
P
  done
done
