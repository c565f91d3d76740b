#!/bin/bash
# The other BASELINE configurations (GPU legs only): bash tools/gpu_configs.sh <tag> -> gpurun_out/<tag>_configs.jsonl
tag=${1:-cfg}
repo="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$repo/gpurun_out"; out="$repo/gpurun_out/${tag}_configs.jsonl"; : > "$out"
cd "$repo"
run() { timeout 300 python bench.py --cpu-sample 0 --spmv-large-mult 0 "$@" 2>/dev/null | tail -1 >> "$out"; }
run --workload nuclear_24h --batch 256
run --workload nuclear_24h
run --workload wind_pem_48h
run --workload wind_battery_48h --batch 1024
run --workload wind_battery_48h
run --workload nuclear_48h
timeout 300 python bench.py --workload double_loop --total 1024 --steps 5 --warmup 1 2>/dev/null | tail -1 > "$repo/gpurun_out/${tag}_double_loop.json"; cut -c1-400 "$repo/gpurun_out/${tag}_double_loop.json"
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]
    print(c["workload"].split(":")[0], c["batch_per_gpu"], "value %.0f" % d["value"], "lone %.2f ms" % c["single_batch_latency_ms"],
          "iters %.0f / %d" % (c["mean_iterations"], c["max_iterations"]), "optimal", c["optimal"], "streams", c["streams"],
          "err", c.get("max_rel_obj_err_vs_oracle_fixture"),
          "spmv_step %.2f us frac %.2f" % (1e3 * d["spmv_step"]["kernel_ms"], d["spmv_step"]["frac"]) if "spmv_step" in d else "")
PY
