#!/bin/bash
# quick look at the year-long solve lines: tools/probes/ipm_quick.sh "<B list>" [ENV=VAL ...]
repo="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$repo"
Bs=${1:-"256 60"}; shift
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print('$1', '|', round(d['value'], 1), d['unit'], '| s/batch', round(c.get('seconds_per_batch'), 3), '| ms/Newton', round(c.get('ms_per_newton_iteration_of_the_batch'), 2), '| newton mean', round(c.get('newton_iterations_per_scenario'), 1), 'max', c.get('max_newton_iterations'),
      '| err', c.get('max_rel_objective_error_vs_oracle_fixture'), 'of', c.get('members_with_oracle_fixture'), '| solved', c.get('solved_to_optimality'), 'ipm', c.get('ipm_solved'))"; }
for B in $Bs; do
  env "$@" timeout 300 python bench.py --workload price_taker --batch $B --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | line "B=$B $*"
done
