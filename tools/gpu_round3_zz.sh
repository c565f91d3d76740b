#!/bin/bash
# Round 3, GPU call ZZ: FETCH_SIZE / WRITE_SIZE of the iteration kernel on the PEM and nuclear price-taker families (the bench lines of
# those workloads quote their OWN kernel's traffic), then the three bench lines
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd /tmp
export TMPDIR=/tmp
for wl in pem_price_taker nuclear_price_taker; do
  for set in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/sp_${wl}_$set; rm -rf $d
    timeout 120 rocprofv3 --pmc $set --output-format csv -d $d -- python $repo/bench.py --workload $wl --steps 6 --warmup 1 > /dev/null 2>&1
  done
  python - "$out/r30zz_stream_pmc_summary_$wl.csv" $wl <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob(f"/tmp/sp_{sys.argv[2]}_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row["Counter_Name"], row["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
with open(sys.argv[1], "w") as o:
    w = csv.writer(o); w.writerow(["counter", "kernel", "dispatches", "mean_counter_value"])
    for (c, k), (n, s) in acc.items():
        if "dsp::" in k:
            w.writerow([c, k, n, round(s / n, 1)])
print("\n".join(l[:40] + " ... " + l[-30:] for l in open(sys.argv[1]).read().splitlines() if "fused" in l))
PY
  cp "$out/r30zz_stream_pmc_summary_$wl.csv" "$repo/profiles/"
done
cd "$repo"
for wl in price_taker pem_price_taker nuclear_price_taker; do timeout 150 python bench.py --workload $wl --steps 16 --warmup 2 2>/dev/null | tail -1; done > "$out/r30zz_stream_bench.jsonl"
python - "$out/r30zz_stream_bench.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["roofline"]
    print(d["config"]["workload"][:22], round(d["value"]), round(r["frac"], 4), r["traffic"], r["traffic_from"], r["algorithmic_bytes_per_scenario_iteration"])
PY
