#!/bin/bash
# the year-long solve lines of the three families under a setting of the development knobs: tools/probes/ipm_env.sh "ENV=VAL ENV=VAL" ["..." ...]
repo="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$repo"
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print('$1', '|', round(d['value'], 1), d['unit'], '| s/batch', round(c.get('seconds_per_batch'), 3), '| ms/Newton', round(c.get('ms_per_newton_iteration_of_the_batch'), 2), '| newton mean', round(c.get('newton_iterations_per_scenario'), 1), 'max', c.get('max_newton_iterations'),
      '| err', c.get('max_rel_objective_error_vs_oracle_fixture'), '| solved', c.get('solved_to_optimality'), 'ipm', c.get('ipm_solved'))"; }
run() { name="$1"; shift; env $SET timeout 300 python bench.py "$@" --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | line "$name"; }
for SET in "$@"; do
  for B in 256 60 30; do run "[$SET] price_taker B=$B" --workload price_taker --batch $B; done
  run "[$SET] pem B=64" --workload pem_price_taker --batch 64
  run "[$SET] nuclear B=60" --workload nuclear_price_taker --batch 60 --horizon 8784
done
