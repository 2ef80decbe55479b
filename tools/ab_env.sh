#!/bin/bash
# A/B of one build under several settings of an environment variable, alternating on the SAME box:
#   tools/ab_env.sh <workload> <rounds> <VAR> <value A> <value B> [...]
WL=$1; R=$2; VAR=$3; shift 3
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for val in "$@"; do
    env $VAR=$val python bench.py --workload $WL --no-cpu --no-also --steps ${STEPS:-20} > gpurun_out/ab/env_$val.$r.json 2> gpurun_out/ab/env_$val.$r.err || true
    python - $VAR $val $r <<'P'
import json,sys
d=json.load(open('gpurun_out/ab/env_%s.%s.json'%(sys.argv[2],sys.argv[3])))
k=d['kernels']
print('%s=%-4s'%(sys.argv[1],sys.argv[2]),sys.argv[3],'ms/step %.4f'%d['ms_per_step'],'ok' if d['correct'] else 'WRONG',' '.join('%s=%.4f'%(n.replace('k_',''),v['ms_avg']) for n,v in k.items() if v['ms_avg']>0.05))
P
  done
done
