cd $GRAFT_REPO_ROOT
for w in 100000 0; do
for n in 64 256 512 1024 2048 4096 8192; do
  MLKEM_WIDE_HASH_ITEMS=$w python3 bench.py --batch $n --steps 200 --warmup 5 --no-cpu --no-also 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('wide_max=%-6s n=%5d  %.4f ms/step ok=%s  hash_encaps=%.4f hash_decaps=%.4f' % ('$w', $n, d['ms_per_step'], d['correct'], k['k_hash_encaps']['ms_total'], k['k_hash_decaps']['ms_total']))"
done; done
