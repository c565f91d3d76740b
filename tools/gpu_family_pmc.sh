#!/bin/bash
# counters of the lane kernel for the PEM and nuclear families and for 64 scenarios + bench lines priced with them: bash tools/gpu_family_pmc.sh
cd "$(dirname "$0")/.."; repo=$(pwd); out=gpurun_out; mkdir -p $out
STREAM_FAMILY=pem WORKLOAD=pem_price_taker bash tools/gpu_lane_pmc.sh ${1:-fam} 256 320 | tail -3
STREAM_FAMILY=nuclear WORKLOAD=nuclear_price_taker T=8784 bash tools/gpu_lane_pmc.sh ${1:-fam} 240 320 | tail -3
bash tools/gpu_lane_pmc.sh ${1:-fam} 64 640 | tail -3
cd $repo; cp $out/${1:-fam}_*pmc_summary*.csv profiles/ 2>/dev/null
{
timeout 300 python bench.py --workload price_taker --batch 256 --solve --warmup 1 2>/dev/null | tail -1
timeout 200 python bench.py --workload pem_price_taker --batch 256 --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload pem_price_taker --batch 64 --steps 50 --warmup 2 2>/dev/null | tail -1
} > $out/${1:-fam}_stream_bench.jsonl
python - $out/${1:-fam}_stream_bench.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]; r = d["roofline"]
    print(d["metric"][:64], "| %.4g %s | frac %.3f | %s | us/it %.1f | traffic %s %s" % (d["value"], d["unit"], r["frac"], c.get("stream_form"), c["us_per_batch_iteration"], r.get("traffic"), r.get("traffic_from")))
PY
