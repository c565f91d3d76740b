#!/bin/bash
# First-hand counters + bench lines for every BASELINE config (round 4):  bash tools/gpu_profile_configs.sh r40p
#   per (workload, batch): three rocprofv3 --pmc passes of the bench command (FETCH_SIZE | WRITE_SIZE | SQ set; no trace domains)
#   -> gpurun_out/<tag>_<workload>_B<batch>_pmc_summary.csv + _pmc_iterations.json, copied into profiles/ ON THE BOX so that the bench
#   lines that follow price every kernel with this round's counters at its own batch -> gpurun_out/<tag>_configs.jsonl
tag=${1:-r40p}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
configs="wind_battery_24h:4096 wind_pem_48h:4096 nuclear_24h:256 nuclear_24h:4096 wind_battery_48h:4096"
for cfg in $configs; do
  wl=${cfg%%:*}; B=${cfg##*:}
  bench="python $repo/bench.py --workload $wl --batch $B --cpu-sample 0 --no-spmv --no-eps4 --no-configs"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES"; do
    i=$((i+1)); rm -rf /tmp/pc_$i
    timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pc_$i -- $bench --steps 8 --warmup 1 --streams 8 --min-time 0 > /tmp/pc_$i.json 2>/dev/null
  done
  python - "$out/${tag}_${wl}_B${B}_pmc_summary.csv" /tmp/pc_3.json <<'PY'
import csv, glob, sys, collections, json
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/pc_[123]/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row["Counter_Name"], row["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
names = set()
with open(sys.argv[1], "w") as o:
    w = csv.writer(o); w.writerow(["counter", "kernel", "dispatches", "mean_counter_value"])
    for (c, k), (n, s) in acc.items():
        if "pdlp_solve_kernel" in k or "simplex" in k:
            w.writerow([c, k, n, round(s / n, 1)])
            if "pdlp_solve_kernel<" in k: names.add(k)
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); c = d["config"]
    key = "pdlp_solve_kernel"
    if len(names) == 1:
        key = "pdlp_solve_kernel<" + ", ".join(next(iter(names)).split("pdlp_solve_kernel<")[1].split(", ")[:2]) + ","
    json.dump({key: c["mean_iterations"] * c["batch_per_gpu"], "batch": c["batch_per_gpu"]}, open(sys.argv[1].replace("_pmc_summary.csv", "_pmc_iterations.json"), "w"))
except Exception as e:
    print("iterations record failed:", e)
print(sys.argv[1], [ (r["counter"], r["mean_counter_value"]) for r in csv.DictReader(open(sys.argv[1])) ][:12])
PY
  cp "$out/${tag}_${wl}_B${B}_pmc_summary.csv" "$out/${tag}_${wl}_B${B}_pmc_iterations.json" "$repo/profiles/" 2>/dev/null
done
# the lines themselves (same box, same build): headline with its CPU leg, SpMV figures and the eps = 1e-4 entry; the others GPU only
python $repo/bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"; tail -c 400 "$out/${tag}_bench.json"; echo
rm -f "$out/${tag}_configs.jsonl"
for cfg in $configs; do
  wl=${cfg%%:*}; B=${cfg##*:}
  [ "$cfg" = "wind_battery_24h:4096" ] && continue
  timeout 200 python $repo/bench.py --workload $wl --batch $B --cpu-sample 0 --no-spmv --no-configs 2>/dev/null | tail -1 >> "$out/${tag}_configs.jsonl"
done
python - "$out/${tag}_configs.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["roofline"]; c = d["config"]
    print(c["workload"][:34], "| %.4g /s | frac %.3f lds %.3f | traffic/io %s | from %s x%.3f | iters %.0f/%d | lone %.2f ms" % (
        d["value"], r["frac"] or 0, r.get("frac_lds") or 0, r.get("traffic_over_true_io"), r.get("counters_from"), r.get("counters_scaled_by") or 0,
        c["mean_iterations"], c["max_iterations"], c["single_batch_latency_ms"]))
PY
