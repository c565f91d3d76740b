#!/bin/bash
# development: bench.py throughput under DSP_OPTIONS overrides (one line per setting)
cd "$(dirname "$0")/.."
WL=${WL:-wind_battery_24h}
for o in "$@"; do
  [ "$o" = base ] && export DSP_OPTIONS="" || export DSP_OPTIONS="$o"
  timeout 200 python bench.py --workload $WL --cpu-sample 0 --no-spmv --streams 8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$o', 'value %.0f'%d['value'],'ms/step %.2f'%d['ms_per_step'],'single %.2f'%c['single_batch_latency_ms'],'mean_iters',c.get('mean_iterations'),'max',c.get('max_iterations'),'opt',c.get('optimal'),'err',c.get('max_rel_obj_err_vs_oracle_fixture'))"
done
