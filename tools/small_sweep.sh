#!/bin/bash
# tools/small_sweep.sh — where the one-workgroup-per-item kernels (mlkem_small.hpp) stop paying: ML-KEM-768 encaps + decaps,
# device-resident, against the batch size with MLKEM_SMALL_ITEMS = 0 (never) and 100000 (always); the wave-wide hash kernels
# (MLKEM_WIDE_HASH_ITEMS) likewise for the sizes above; the small kernels in both forms (MLKEM_SMALL_LATENCY_ITEMS: eight waves
# per item up to that size, four above).  Output: one line per (size, setting).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
run() {  # label env n
  env $2 timeout -k 10 120 python3 bench.py --batch $3 --steps 200 --warmup 10 --no-cpu --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%-24s n=%5d  %8.4f ms/step  ok=%s | ' % ('$1', $3, d['ms_per_step'], d['correct']) + ' '.join('%s=%.4f' % (a.replace('k_',''), b['ms_total']) for a,b in k.items()))
"
}
for n in 1 16 64 128 256 384 512 768 1024 1536 2048 3072 4096; do
  run "small=always, 8 waves" "MLKEM_SMALL_ITEMS=100000 MLKEM_SMALL_LATENCY_ITEMS=100000" $n
  run "small=always, 4 waves" "MLKEM_SMALL_ITEMS=100000 MLKEM_SMALL_LATENCY_ITEMS=0" $n
  run "small=never,wide=always" "MLKEM_SMALL_ITEMS=0 MLKEM_WIDE_HASH_ITEMS=100000" $n
done
for n in 512 1024 2048 4096 8192; do
  run "small=never,wide=never" "MLKEM_SMALL_ITEMS=0 MLKEM_WIDE_HASH_ITEMS=0" $n
done
