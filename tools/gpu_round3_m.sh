#!/bin/bash
# Round 3, GPU call M: fused-iteration variants (staged fallback restored, XCD-major workgroup order, long-column loads hoisted):
# streaming tests, A/B rates (libdsp_hip_a.so = before the hoist), occupancy / traffic counters, warm-start traces of the rolling loop
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_hip_stream.py -m gpu -q --timeout 300 > "$out/r30m_stream_tests.log" 2>&1; tail -4 "$out/r30m_stream_tests.log"
{
for rep in 1 2; do
  for lib in libdsp_hip_a.so libdsp_hip.so; do for xcd in 0 1; do
    echo -n "lib=$lib xcd=$xcd: "; DSP_LIB=$lib DSP_FUSED_XCD=$xcd timeout 200 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T=" | sed 's/.*-> //'
  done; done
done
} | tee "$out/r30m_fused_ab.log"
cd /tmp
for xcd in 0 1; do
  for set in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
    d=/tmp/sp_${xcd}_$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf $d
    DSP_FUSED_XCD=$xcd timeout 200 rocprofv3 --pmc $set --output-format csv -d $d -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
  done
  python - "$out/r30m_stream_pmc_summary_xcd$xcd.csv" $xcd <<'PY'
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
print("\n".join(l[:160] for l in open(sys.argv[1]).read().splitlines() if "fused" in l))
PY
done
rm -rf /tmp/sp_trace; DSP_FUSED_XCD=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > /dev/null 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30m_stream_kernel_stats_xcd1.csv" && head -3 "$f" | cut -c1-200
cd "$repo"
for warm in 0 1 2; do timeout 200 python tools/gpu_rolling_year.py 1024 4 $warm 2>&1 | grep -v amdgpu; done | tee "$out/r30m_warm_start.log"
