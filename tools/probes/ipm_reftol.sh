#!/bin/bash
# refinement tolerances of the interior-point form over the three year-long families: tools/probes/ipm_reftol.sh "<tol>:<tol_end> ..."
repo="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$repo"
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print('$1', '|', round(d['value'], 1), d['unit'], '| s/batch', round(c.get('seconds_per_batch'), 3), '| ms/Newton', round(c.get('ms_per_newton_iteration_of_the_batch'), 2), '| newton mean', round(c.get('newton_iterations_per_scenario'), 1), 'max', c.get('max_newton_iterations'),
      '| err', c.get('max_rel_objective_error_vs_oracle_fixture'), '| solved', c.get('solved_to_optimality'), 'ipm', c.get('ipm_solved'))"; }
for p in $1; do
  export DSP_IPM_REFTOL=${p%%:*} DSP_IPM_REFTOL_END=${p##*:}
  for B in 256 60; do timeout 300 python bench.py --workload price_taker --batch $B --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | line "reftol $p price_taker B=$B"; done
  timeout 300 python bench.py --workload pem_price_taker --batch 64 --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | line "reftol $p pem B=64"
  timeout 300 python bench.py --workload nuclear_price_taker --batch 60 --horizon 8784 --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | line "reftol $p nuclear B=60"
done
