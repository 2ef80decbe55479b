#!/bin/bash
# tools/batch_sweep.sh [lib] — device-resident ML-KEM-768 encaps+decaps rate and time per step against the batch size
# (bench.py --batch N --no-cpu --no-also): how far below 2^20 the engine keeps the GPU full.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -n "$1" ] && export MLKEM_LIB_PATH=$ROOT/$1
cd $ROOT
for lg in 0 6 8 10 12 14 16 18 20; do
  n=$((1 << lg)); steps=$(( lg < 16 ? 200 : 40 ))
  timeout -k 10 200 python3 bench.py --batch $n --steps $steps --warmup 5 --no-cpu --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('n=2^%-2d  %10.4g pairs/s  %9.4f ms/step  ok=%s | ' % ($lg, d['value'], d['ms_per_step'], d['correct']) + ' '.join('%s=%.4f' % (a.replace('k_',''), b['ms_total']) for a,b in k.items()))
"
done
