#!/bin/bash
# sweep MLKEM_ARITH_GRID for the looped k_encrypt
mkdir -p gpurun_out/r02m
for g in ${GRIDS:-0 1280 2560 5120 10240}; do
  MLKEM_ARITH_GRID=$g python bench.py --no-cpu --no-also --steps 20 > gpurun_out/r02m/g$g.json 2> gpurun_out/r02m/g$g.err || exit 1
  python - $g <<'P'
import json,sys
g=sys.argv[1]
d=json.load(open('gpurun_out/r02m/g%s.json'%g))
k=d['kernels']
print('grid',g,'ms/step %.3f'%d['ms_per_step'],'correct',d['correct'],' '.join('%s=%.3f'%(n,k[n]['ms_avg']) for n in ('k_encrypt','k_encrypt_cmp','k_decrypt','k_sample_main')))
P
done
