#!/bin/bash
# Round 4, call D: lane form with the records through the LDS stage (no scalar-cache misses in the walk).
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_stream.py -m gpu -q --timeout 800 > "$out/r40d_stream_tests.log" 2>&1; tail -5 "$out/r40d_stream_tests.log"
rate() { timeout 200 python tools/gpu_stream.py 8736 $1 ${2:-6400} 64 2>&1 | grep "^T=" | sed 's/.*status/status/' | cut -c1-200; }
{
for B in 1 16 64 256 1024; do echo -n "B=$B lane: "; rate $B; done
for B in 64 256; do for wv in 1024 4096 8192; do echo -n "B=$B waves=$wv: "; DSP_LANE_WAVES=$wv rate $B; done; done
for B in 64 256; do echo -n "B=$B ring<=8: "; DSP_LANE_RING_MAX=8 rate $B;  echo -n "B=$B ring<=8 waves=4096: "; DSP_LANE_RING_MAX=8 DSP_LANE_WAVES=4096 rate $B; done
echo -n "chain B=64: "; STREAM_THROUGHPUT=chain rate 64
echo -n "chain B=256: "; STREAM_THROUGHPUT=chain rate 256
} 2>&1 | tee "$out/r40d_lane_rates.log"
cd /tmp
for B in 64 256; do
  rm -rf /tmp/tr_$B; DSP_LANE_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$B -- python $repo/tools/gpu_stream.py 8736 $B 1280 64 > /dev/null 2>&1
  f=$(find /tmp/tr_$B -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/r40d_lane_kernel_stats_B$B.csv" && head -8 "$f" | cut -c1-200
done
