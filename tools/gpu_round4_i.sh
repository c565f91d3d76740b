#!/bin/bash
# Round 4, call I: whole GPU suite at HEAD, streaming bench lines (rate at 64 / 256 scenarios, full solves), counters of the lane kernel.
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > "$out/r40i_tests.log" 2>&1; tail -6 "$out/r40i_tests.log"
{
timeout 200 python bench.py --workload price_taker --batch 64 --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload price_taker --batch 256 --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload price_taker --batch 1024 --steps 20 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload price_taker --batch 256 --throughput chain --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload pem_price_taker --batch 64 --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload pem_price_taker --batch 256 --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload nuclear_price_taker --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload nuclear_price_taker --batch 240 --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 400 python bench.py --workload price_taker --batch 64 --solve --warmup 1 2>/dev/null | tail -1
timeout 400 python bench.py --workload price_taker --batch 16 --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1
} > "$out/r40i_stream_bench.jsonl"
python - "$out/r40i_stream_bench.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]; r = d["roofline"]
    print(d["metric"][:70], "| value %.4g %s | frac %.3f | form %s | us/it %.1f | cpu %s" % (d["value"], d["unit"], r["frac"], c.get("stream_form"), c["us_per_batch_iteration"], d.get("cpu_baseline", {}).get("value")))
PY
bash tools/gpu_lane_pmc.sh r40i 256 640 | tail -3
