#!/bin/bash
# Round 3, GPU call B: the fused one-launch iteration of the streaming path (k_fused) - parity tests, rate per tile size /
# scenarios per workgroup at T = 8736, FETCH_SIZE / WRITE_SIZE and kernel trace at B = 64.
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_stream.py -m gpu -q -x > "$out/r30b_stream_tests.log" 2>&1; tail -12 "$out/r30b_stream_tests.log"
for cfg in "768 0" "384 0" "1536 0" "768 1" "768 2" "384 2"; do
  set -- $cfg
  echo "== DSP_FUSED_RB=$1 DSP_FUSED_SG=$2 (0 = automatic), B = 64, 4096 iterations"
  DSP_FUSED_RB=$1 DSP_FUSED_SG=$2 timeout 300 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T="
done > "$out/r30b_fused_scan.log" 2>&1; cat "$out/r30b_fused_scan.log"
echo "== B = 16, 4096 iterations"; timeout 300 python tools/gpu_stream.py 8736 16 4096 64 2>&1 | grep "^T=" | tee "$out/r30b_fused_B16.log"
echo "== B = 256, 1024 iterations"; timeout 300 python tools/gpu_stream.py 8736 256 1024 64 2>&1 | grep "^T=" | tee "$out/r30b_fused_B256.log"
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/sp_$set; timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/sp_$set -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
done
rm -rf /tmp/sp_trace; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > /dev/null 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30b_stream_kernel_stats.csv" && head -8 "$f" | cut -c1-160
python - "$out/r30b_stream_pmc_summary.csv" <<'PY'
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
print(open(sys.argv[1]).read()[:2500])
PY
