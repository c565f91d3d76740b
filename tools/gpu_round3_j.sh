#!/bin/bash
# Round 3, GPU call J: grid order (scenario groups fastest) of the fused iteration - tests, trace, FETCH/WRITE, SG scan
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_hip_stream.py -m gpu -q --timeout 300 -k "fused or year_long or golden" > "$out/r30j_stream_tests.log" 2>&1; tail -4 "$out/r30j_stream_tests.log"
for cfg in "500 4" "500 2" "250 4" "250 2" "250 1"; do
  set -- $cfg
  echo "== DSP_FUSED_RB=$1 DSP_FUSED_SG=$2, B = 64, 4096 iterations"
  DSP_FUSED_RB=$1 DSP_FUSED_SG=$2 timeout 200 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T="
done > "$out/r30j_fused_scan.log" 2>&1; cat "$out/r30j_fused_scan.log"
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/sp_$set; timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/sp_$set -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
done
rm -rf /tmp/sp_trace; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > /dev/null 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30j_stream_kernel_stats.csv" && head -3 "$f" | cut -c1-220
python - "$out/r30j_stream_pmc_summary.csv" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/sp_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row["Counter_Name"], row["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
with open(sys.argv[1], "w") as o:
    w = csv.writer(o); w.writerow(["counter", "kernel", "dispatches", "mean_counter_value"])
    for (c, k), (n, s) in acc.items():
        if "dsp::" in k:
            w.writerow([c, k, n, round(s / n, 1)])
print("\n".join(l for l in open(sys.argv[1]).read().splitlines() if "fused" in l))
PY
