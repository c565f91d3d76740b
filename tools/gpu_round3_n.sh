#!/bin/bash
# Round 3, GPU call N: rolling warm start of the day-ahead LP with persistent buffers + hipGraphs: tests, 60 simulated days cold / warm,
# config-4 bench line both ways
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_rolling.py -m gpu -q --timeout 300 > "$out/r30n_rolling_tests.log" 2>&1; tail -6 "$out/r30n_rolling_tests.log"
for warm in 0 1; do timeout 300 python tools/gpu_rolling_year.py 1024 60 $warm 2>&1 | grep -v amdgpu; done > "$out/r30n_warm_start_60d.log"
grep -c "day" "$out/r30n_warm_start_60d.log"; grep "plants x" "$out/r30n_warm_start_60d.log"
python - "$out/r30n_warm_start_60d.log" <<'PY'
import re, sys, numpy as np
d = {0: [], 1: []}
for l in open(sys.argv[1]):
    m = re.match(r"day (\d+) \(warm start (\d)\): DA iterations mean (\d+) max (\d+); ([\d.]+) ms", l)
    if m: d[int(m.group(2))].append((int(m.group(3)), int(m.group(4)), float(m.group(5))))
for w, v in d.items():
    a = np.array(v)
    if len(a): print(f"warm {w}: {len(a)} days, DA mean iterations {a[:,0].mean():.0f}, mean of daily max {a[:,1].mean():.0f}, worst max {a[:,1].max():.0f}, median ms/day {np.median(a[:,2]):.2f}")
PY
for warm in 0 1; do timeout 300 python bench.py --workload double_loop --total 1024 --steps 30 --warmup 3 --warm-start $warm 2>/dev/null | tail -1; done > "$out/r30n_double_loop.jsonl"; cut -c1-1200 "$out/r30n_double_loop.jsonl"
