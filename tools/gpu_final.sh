#!/bin/bash
# End-of-round check on one MI355X:  bash tools/gpu_final.sh <tag>
#   build check, smoke, the whole -m gpu suite, the default bench line, the streaming lines, counters + trace of the lane kernel
tag=${1:-final}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/${tag}_smoke.log" 2>&1; tail -2 "$out/${tag}_smoke.log"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > "$out/${tag}_tests.log" 2>&1; tail -4 "$out/${tag}_tests.log"
python bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"; tail -c 300 "$out/${tag}_bench.json"; echo
# same-round kernel trace of the default line's kernels (pdlp_solve_kernel, spmv_step_kernel; without the configs array and the CPU leg)
( cd /tmp; rm -rf /tmp/trz; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trz -- python $repo/bench.py --cpu-sample 0 --no-configs > /dev/null 2>&1
  f=$(find /tmp/trz -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv" && head -6 "$f" | cut -c1-200 )
timeout 200 python tools/gpu_bidder_profile.py > "$out/${tag}_bidder_profile.log" 2>&1; head -30 "$out/${tag}_bidder_profile.log" | tail -22
{
timeout 200 python bench.py --workload price_taker --batch 64 --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload price_taker --batch 256 --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload pem_price_taker --batch 256 --steps 50 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload nuclear_price_taker --steps 12 --warmup 2 2>/dev/null | tail -1
timeout 200 python bench.py --workload nuclear_price_taker --batch 240 --steps 12 --warmup 2 2>/dev/null | tail -1
# (full solves: the HiGHS leg of the 256-member batch keeps the host's 256 threads busy for ~ 5 minutes - profiles/r50d_solve256.json -, so the
#  end-of-round check runs the GPU side only and a 64-process leg on the 64-member batch)
timeout 600 python bench.py --workload price_taker --batch 64 --solve --warmup 1 --cpu-sample 64 2>/dev/null | tail -1
timeout 400 python bench.py --workload price_taker --batch 256 --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1
} > "$out/${tag}_stream_bench.jsonl"
python - "$out/${tag}_stream_bench.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]; r = d["roofline"]
    fr = "%.3f" % r["frac"] if r.get("frac") is not None else "-"
    per = "us/it %.1f" % c["us_per_batch_iteration"] if "us_per_batch_iteration" in c else "ms/Newton %.2f (%s partitions)" % (c.get("ms_per_newton_iteration_of_the_batch", float("nan")), c.get("time_partitions"))
    print(d["metric"][:64], "| %.4g %s | frac %s | %s | %s | done early %s | traffic %s" % (d["value"], d["unit"], fr, c.get("stream_form"), per, c.get("finished_before_the_cap"), r.get("traffic_from")))
PY
bash tools/gpu_lane_pmc.sh $tag 256 640 | tail -4
cd /tmp; rm -rf /tmp/tr; DSP_LANE_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python $repo/tools/gpu_stream.py 8736 256 1280 64 > /dev/null 2>&1
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_lane_kernel_stats_B256.csv" && head -5 "$f" | cut -c1-180
