#!/bin/bash
# End-of-round check of round 5's last state on one MI355X (the GPU minutes left did not allow tools/gpu_final.sh's full list):
#   build check + smoke, the whole -m gpu suite, the default bench line (all BASELINE configs), kernel trace of its kernels,
#   the year-long solves by the time-parallel interior-point form (256 and 64 members) with their kernel trace.      bash tools/gpu_final5.sh <tag>
tag=${1:-final}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$out/${tag}_smoke.log" 2>&1; tail -2 "$out/${tag}_smoke.log"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > "$out/${tag}_tests.log" 2>&1; tail -4 "$out/${tag}_tests.log"
python bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"; tail -c 300 "$out/${tag}_bench.json"; echo
( cd /tmp; rm -rf /tmp/trz; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trz -- python $repo/bench.py --cpu-sample 0 --no-configs > /dev/null 2>&1
  f=$(find /tmp/trz -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv" && head -6 "$f" | cut -c1-200 )
bash tools/gpu_ipm_par.sh $tag " " | tail -12 | cut -c1-220
