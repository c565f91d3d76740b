#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) + bench line of one workload:  bash tools/gpu_traffic.sh <tag> <workload>
tag=${1:-traffic}; wl=${2:-wind_battery_48h}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bench="python $repo/bench.py --workload $wl --no-spmv --cpu-sample 0"
$bench --steps 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('bench', d['value'], d['ms_per_step'], c['mean_iterations'], c['max_iterations'], c['single_batch_latency_ms'], c['optimal'], c['max_rel_obj_err_vs_oracle_fixture'])"
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/traf_$set; timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/traf_$set -- $bench --steps 8 --warmup 1 --streams 8 > /dev/null 2>&1
  python - $set <<'PY'
import csv, glob, sys
tot, n = 0.0, 0
for f in glob.glob(f"/tmp/traf_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pdlp_solve_kernel" in row["Kernel_Name"]:
            tot += float(row["Counter_Value"]); n += 1
print(sys.argv[1], "mean per launch [KB]", round(tot / max(n, 1), 1), "launches", n)
PY
done
