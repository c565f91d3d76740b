#!/bin/bash
# Round 3, GPU call H: sanity of the per-scenario-bounds form of the fused iteration, the whole -m gpu suite (per-test timeout), the
# round profile recipe at HEAD, FETCH/WRITE + trace of the streaming kernels at their defaults, bench lines of the streaming workloads
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 150 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -6
import sys, time; sys.path.insert(0, ".")
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
from oracle import dispatch_lp_oracle as orc
s = HipPdlpSolver(device=0, check_every=64, max_iter=200000)
h, m = scenarios.nuclear_price_taker_batch(720, 12, s)
t = time.time(); s.solve(m, tee=True)
cl = np.array([-1e-6 * orc.nuclear_price_taker_closed_form(m.lmp, hp, pc * 400.0) for hp, pc in m.family])
print("nuclear T=720 B=12: status", m.status.tolist(), "iters", m.iterations.tolist(), "max rel err", np.abs(m.objective - cl).max() / np.abs(cl).max(), f"{time.time()-t:.1f}s")
PY
timeout 1300 python -m pytest tests -m gpu -q --timeout 240 > "$out/r30k_tests.log" 2>&1; tail -25 "$out/r30k_tests.log"
timeout 600 bash tools/gpu_profile.sh r30k 2>&1 | tail -32
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/sp_$set; timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/sp_$set -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
done
rm -rf /tmp/sp_trace; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > /dev/null 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30k_stream_kernel_stats.csv" && head -4 "$f" | cut -c1-200
python - "$out/r30k_stream_pmc_summary.csv" <<'PY'
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
for wl in price_taker pem_price_taker nuclear_price_taker; do timeout 200 python bench.py --workload $wl --steps 16 --warmup 2 2>/dev/null | tail -1; done > "$out/r30k_stream_bench.jsonl"; cut -c1-400 "$out/r30k_stream_bench.jsonl"
for wl in wind_battery_48h wind_pem_48h nuclear_24h; do timeout 200 python bench.py --workload $wl --no-spmv --cpu-sample 0 2>/dev/null | tail -1; done > "$out/r30k_configs.jsonl"; cut -c1-300 "$out/r30k_configs.jsonl"
