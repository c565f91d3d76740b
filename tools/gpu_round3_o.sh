#!/bin/bash
# Round 3, GPU call O: k_fused_pre with the matrix entries requested after the barriers (88 VGPRs, 5 waves per SIMD) against the
# all-loads-up-front form (122 VGPRs, 4 waves): streaming tests on both, alternating rates, trace + counters of the deferred form
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
DSP_FUSED_DEFER=1 timeout 500 python -m pytest tests/test_hip_stream.py -m gpu -q --timeout 300 -k "fused or year_long" > "$out/r30o_stream_tests_defer.log" 2>&1; tail -3 "$out/r30o_stream_tests_defer.log"
{
for rep in 1 2 3; do for d in 0 1; do
  echo -n "defer=$d: "; DSP_FUSED_DEFER=$d timeout 200 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T=" | sed 's/.*-> //'
done; done
for B in 16 256; do for d in 0 1; do
  echo -n "B=$B defer=$d: "; DSP_FUSED_DEFER=$d timeout 200 python tools/gpu_stream.py 8736 $B 2048 64 2>&1 | grep "^T=" | sed 's/.*-> //'
done; done
for d in 0 1; do echo -n "pem defer=$d: "; DSP_FUSED_DEFER=$d timeout 200 python bench.py --workload pem_price_taker --steps 16 --warmup 2 2>/dev/null | tail -1 | cut -c1-200; done
for d in 0 1; do echo -n "nuclear defer=$d: "; DSP_FUSED_DEFER=$d timeout 200 python bench.py --workload nuclear_price_taker --steps 16 --warmup 2 2>/dev/null | tail -1 | cut -c1-200; done
} | tee "$out/r30o_fused_defer.log"
cd /tmp
for set in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  d=/tmp/sp_d_$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf $d
  DSP_FUSED_DEFER=1 timeout 200 rocprofv3 --pmc $set --output-format csv -d $d -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
done
python - "$out/r30o_stream_pmc_summary_defer.csv" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/sp_d_*/**/*counter_collection.csv", recursive=True)):
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
rm -rf /tmp/sp_trace; DSP_FUSED_DEFER=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > /dev/null 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30o_stream_kernel_stats_defer.csv" && head -3 "$f" | cut -c1-200
