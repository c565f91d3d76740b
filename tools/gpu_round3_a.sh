#!/bin/bash
# Round 3, GPU call A (gpurun):  bash tools/gpu_round3_a.sh
#   1. the -m gpu suite                                   -> gpurun_out/r30a_tests.log
#   2. the round profile recipe at HEAD (bench + kernel traces + 5 PMC passes, counters first-hand: no rescale)
#   3. year-long price-taker LP on the streaming path (converged power iteration + physical scaling factors): does it converge?
#   4. FETCH_SIZE / WRITE_SIZE + kernel trace of the streaming kernels at B = 64
#   5. flagged scenarios before / after the re-solves
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > "$out/r30a_tests.log" 2>&1; tail -15 "$out/r30a_tests.log"
timeout 900 bash tools/gpu_profile.sh r30a 2>&1 | tail -40
timeout 600 python tools/gpu_stream.py 8736 16 1600000 64 > "$out/r30a_stream_T8736_B16.log" 2>&1; tail -6 "$out/r30a_stream_T8736_B16.log"
STREAM_THROUGHPUT=scan timeout 600 python tools/gpu_stream.py 8736 16 1600000 64 > "$out/r30a_stream_T8736_B16_scan.log" 2>&1; tail -6 "$out/r30a_stream_T8736_B16_scan.log"
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/sp_$set; timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/sp_$set -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
done
rm -rf /tmp/sp_trace; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > "$out/r30a_stream_T8736_B64.log" 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30a_stream_kernel_stats.csv" && head -8 "$f" | cut -c1-160
python - "$out/r30a_stream_pmc_summary.csv" <<'PY'
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
print(open(sys.argv[1]).read()[:3000])
PY
cd "$repo"
timeout 600 python tools/gpu_recertify.py > "$out/r30a_recertify.log" 2>&1; cat "$out/r30a_recertify.log"
