#!/bin/bash
# Round 4, call B: first run of the lane-per-scenario streaming form (dsp_stream_lane.hip) - the streaming GPU tests, then the
# year-long price-taker batches at 16 / 64 / 256 scenarios against the round-3 fused form (DSP_STREAM_NO_LANE=1).
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_stream.py -m gpu -q -x --timeout 500 > "$out/r40b_stream_tests.log" 2>&1; tail -15 "$out/r40b_stream_tests.log"
{
for B in 16 64 256; do for lane in 0 1; do
  echo -n "no_lane=$lane B=$B: "; DSP_STREAM_NO_LANE=$lane timeout 200 python tools/gpu_stream.py 8736 $B 6400 64 2>&1 | grep "^T=" | cut -c1-260
done; done
} | tee "$out/r40b_lane_rates.log"
