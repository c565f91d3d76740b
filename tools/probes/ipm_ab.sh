#!/bin/bash
# A / B of two builds of the library on ONE box (boxes differ by ~4 % on the bandwidth-bound 256-LP solve): tools/probes/ipm_ab.sh <other .so name> [rounds]
repo="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$repo"
other=$1; rounds=${2:-3}
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print('$1', '| s/batch', round(c.get('seconds_per_batch'), 4), '| ms/Newton', round(c.get('ms_per_newton_iteration_of_the_batch'), 3), '| max', c.get('max_newton_iterations'), '| err', c.get('max_rel_objective_error_vs_oracle_fixture'), '| ipm', c.get('ipm_solved'))"; }
for r in $(seq $rounds); do
  for B in 256 60; do
    timeout 300 python bench.py --workload price_taker --batch $B --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | line "base  B=$B"
    DSP_LIB=$other timeout 300 python bench.py --workload price_taker --batch $B --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | line "other B=$B"
  done
done
