#!/bin/bash
# Round 3, GPU call F: XCD-aware workgroup order for k_fused_pre (matrix slices served by the XCD's L2)
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
for cfg in "2 250 1 1" "2 250 2 1" "2 250 4 1" "2 500 1 1" "2 500 2 1" "2 752 1 1" "2 250 1 0" "2 250 4 0" "1 752 2 0"; do
  set -- $cfg
  echo "== DSP_FUSED_V=$1 DSP_FUSED_RB=$2 DSP_FUSED_SG=$3 DSP_FUSED_XCD=$4, B = 64, 4096 iterations"
  DSP_FUSED_V=$1 DSP_FUSED_RB=$2 DSP_FUSED_SG=$3 DSP_FUSED_XCD=$4 timeout 300 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T="
done > "$out/r30f_fused_scan.log" 2>&1; cat "$out/r30f_fused_scan.log"
cd /tmp
for sg in 1 2; do
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/sp_$set; DSP_FUSED_V=2 DSP_FUSED_RB=250 DSP_FUSED_SG=$sg timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/sp_$set -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
done
python - "$out/r30f_stream_pmc_summary_sg$sg.csv" <<'PY'
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
done
