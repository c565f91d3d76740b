#!/bin/bash
# Round 3, GPU call I: streaming tests after the host-side fixes (batch presolve, wrap-around columns long), k_fused_pre with the
# halo loads pinned, bench lines of the three streaming families, and the lab's iteration cuts on the metric workload
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_hip_stream.py -m gpu -q --timeout 300 > "$out/r30i_stream_tests.log" 2>&1; tail -12 "$out/r30i_stream_tests.log"
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/sp_$set; timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/sp_$set -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
done
rm -rf /tmp/sp_trace; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > /dev/null 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30i_stream_kernel_stats.csv" && head -3 "$f" | cut -c1-220
python - "$out/r30i_stream_pmc_summary.csv" <<'PY'
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
cd "$repo"
for wl in price_taker pem_price_taker nuclear_price_taker; do timeout 200 python bench.py --workload $wl --steps 16 --warmup 2 2>/dev/null | tail -1; done > "$out/r30i_stream_bench.jsonl"; cut -c1-330 "$out/r30i_stream_bench.jsonl"
for o in "" "ruiz_iters=3" "check_every=12" "ruiz_iters=3,check_every=12"; do
  echo "== DSP_OPTIONS=$o"; DSP_OPTIONS="$o" timeout 200 python bench.py --no-spmv --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), 'mean it', round(c['mean_iterations'],1), 'max', c['max_iterations'], 'lone ms', round(c['single_batch_latency_ms'],3), 'optimal', c['optimal'], 'err', c.get('max_rel_obj_err_vs_oracle_fixture'))"
done 2>&1 | tee "$out/r30i_iteration_cuts.log"
