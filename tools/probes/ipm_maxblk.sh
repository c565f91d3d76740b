cd /root/repo
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print('$1', '| s/batch', round(c.get('seconds_per_batch'), 3), '| ms/Newton', round(c.get('ms_per_newton_iteration_of_the_batch'), 2), '| max', c.get('max_newton_iterations'))"; }
for mb in 1024 768 512 384; do
  for B in 60 256; do DSP_IPM_MAXBLK=$mb timeout 300 python bench.py --workload price_taker --batch $B --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | line "maxblk $mb B=$B"; done
done
