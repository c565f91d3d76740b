#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun):  bash tools/gpu_profile.sh r01k
#   1. bench.py (full default run)                          -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/<tag>_kernel_stats.csv
#      + the same on ONE stream (launches do not overlap: AverageNs is a per-step duration) -> <tag>_kernel_stats_1stream.csv
#   3. rocprofv3 --pmc passes (own runs, no trace domains)  -> gpurun_out/<tag>_pmc_summary.csv (mean per dispatch)
#   optional 2nd argument: another bench workload (e.g. wind_battery_48h), profiled with --no-spmv
tag=${1:-prof}
wl=${2:-}
repo="$(cd "$(dirname "$0")/.." && pwd)"
out="$repo/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bench="python $repo/bench.py"
[ -n "$wl" ] && bench="python $repo/bench.py --workload $wl --no-spmv"
$bench > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
tail -c 600 "$out/${tag}_bench.json"; echo
rm -rf /tmp/prof_trace; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -- $bench --cpu-sample 0 --no-configs > /dev/null 2>&1
f=$(find /tmp/prof_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv" && head -4 "$f" | cut -c1-200
rm -rf /tmp/prof_trace1; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace1 -- $bench --cpu-sample 0 --no-configs --streams 1 --no-spmv --min-time 0.2 > "$out/${tag}_bench_1stream.json" 2>/dev/null
f=$(find /tmp/prof_trace1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats_1stream.csv" && head -3 "$f" | cut -c1-200
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/prof_pmc$i
  extra="--no-spmv"; [ $i -le 2 ] && extra="--spmv-large-mult 0"
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/prof_pmc$i -- $bench --cpu-sample 0 --no-configs --steps 8 --warmup 1 --streams 8 --min-time 0 $extra > /dev/null 2>&1
done
python - "$out/${tag}_pmc_summary.csv" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/prof_pmc*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row["Counter_Name"], row["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
with open(sys.argv[1], "w") as o:
    w = csv.writer(o); w.writerow(["counter", "kernel", "dispatches", "mean_counter_value"])
    for (c, k), (n, s) in acc.items():
        if "dsp::" in k:
            w.writerow([c, k, n, round(s / n, 1)])
print(open(sys.argv[1]).read())
PY
# scenario-iterations of one launch of this command (bench.py scales the counters by live / recorded iterations when the shipped
# options change the iteration count after a profile was taken)
python - "$out/${tag}_bench.json" "$out/${tag}_pmc_iterations.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
grid = c.get("grid")
key = "pdlp_solve_kernel"
import csv, os
summ = sys.argv[2].replace("_pmc_iterations.json", "_pmc_summary.csv")
if os.path.exists(summ):
    names = {r["kernel"] for r in csv.DictReader(open(summ)) if "pdlp_solve_kernel<" in r["kernel"]}
    if len(names) == 1:
        key = "pdlp_solve_kernel<" + ", ".join(next(iter(names)).split("pdlp_solve_kernel<")[1].split(", ")[:2]) + ","
json.dump({key: c["mean_iterations"] * c["batch_per_gpu"]}, open(sys.argv[2], "w"))
print(open(sys.argv[2]).read())
PY
