#!/bin/bash
# End-of-round check (round 6) on one MI355X: build check + smoke, the whole -m gpu suite, the default bench line (all BASELINE configs),
# kernel trace of the default line's kernels and of the year loop.      bash tools/gpu_final6.sh <tag>
tag=${1:-final}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/${tag}_smoke.log" 2>&1; tail -2 "$out/${tag}_smoke.log"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > "$out/${tag}_gpu_tests.log" 2>&1; tail -4 "$out/${tag}_gpu_tests.log"
python bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"; tail -c 300 "$out/${tag}_bench.json"; echo
( cd /tmp; D=/tmp/trz_$$; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $repo/bench.py --cpu-sample 0 --no-configs > /dev/null 2>&1
  f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv" && head -6 "$f" | cut -c1-200 )
( cd /tmp; D=/tmp/try_$$; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $repo/bench.py --workload double_loop --total 8192 --steps 20 --warmup 2 > /dev/null 2>&1
  f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_year_loop_kernel_stats.csv" && head -5 "$f" | cut -c1-200 )
