#!/bin/bash
# as ipm_reftol.sh, plus the steps taken back (DSP_IPM_TRACE=1 prints them): tools/probes/ipm_reftol2.sh "<tol>:<tol_end> ..." ["<B list>"]
repo="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$repo"
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print('$1', '|', round(d['value'], 1), d['unit'], '| s/batch', round(c.get('seconds_per_batch'), 3), '| ms/Newton', round(c.get('ms_per_newton_iteration_of_the_batch'), 2), '| newton mean', round(c.get('newton_iterations_per_scenario'), 1), 'max', c.get('max_newton_iterations'),
      '| err', c.get('max_rel_objective_error_vs_oracle_fixture'), '| solved', c.get('solved_to_optimality'), 'ipm', c.get('ipm_solved'))"; }
run() { name="$1"; shift; timeout 300 python bench.py "$@" --solve --warmup 0 --cpu-sample 0 2>/tmp/ipm_err | tail -1 | line "$name"; grep "taken back:" /tmp/ipm_err | tail -1; }
export DSP_IPM_TRACE=1
for p in $1; do
  export DSP_IPM_REFTOL=${p%%:*} DSP_IPM_REFTOL_END=${p##*:}
  for B in ${2:-256 60 30}; do run "reftol $p price_taker B=$B" --workload price_taker --batch $B; done
  run "reftol $p pem B=64" --workload pem_price_taker --batch 64
  run "reftol $p nuclear B=60" --workload nuclear_price_taker --batch 60 --horizon 8784
done
