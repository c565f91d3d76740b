#!/bin/bash
# Round 3, GPU call W: final state - smoke, the whole -m gpu suite, streaming counters + trace + bench lines, default bench line
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/r30w_smoke.log" 2>&1; tail -2 "$out/r30w_smoke.log"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > "$out/r30w_tests.log" 2>&1; tail -3 "$out/r30w_tests.log"
cd /tmp
for set in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  d=/tmp/sp_w_$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf $d
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $d -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
done
python - "$out/r30w_stream_pmc_summary.csv" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/sp_w_*/**/*counter_collection.csv", recursive=True)):
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
rm -rf /tmp/sp_trace; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > /dev/null 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30w_stream_kernel_stats.csv" && head -9 "$f" | cut -c1-200
cd "$repo"
for wl in price_taker pem_price_taker nuclear_price_taker; do timeout 200 python bench.py --workload $wl --steps 16 --warmup 2 2>/dev/null | tail -1; done > "$out/r30w_stream_bench.jsonl"; cut -c1-330 "$out/r30w_stream_bench.jsonl"
timeout 300 python tools/gpu_stream.py 8736 16 1600000 64 2>&1 | grep -v amdgpu | grep "^T=\|obj err" -A 3 | head -6 | tee "$out/r30w_stream_T8736_B16.log"
timeout 400 python bench.py > "$out/r30w_bench.json" 2> "$out/r30w_bench.err"; tail -c 600 "$out/r30w_bench.json"; echo
