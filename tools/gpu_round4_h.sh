#!/bin/bash
# Round 4, call H: idle lanes masked out of the walk: small batches, finished scenarios; full solves of the year-long fixture batches.
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_stream.py -m gpu -q -x --timeout 500 2>&1 | tail -3 | tee "$out/r40h_stream_tests.log"
rate() { timeout 300 python tools/gpu_stream.py 8736 $1 ${2:-3200} 64 2>&1 | grep "^T=\|obj err" | sed 's/.*solve wall/solve wall/' | cut -c1-330; }
{
for B in 1 4 16 32 64 256; do echo -n "B=$B: "; rate $B; done
for B in 1 16; do echo -n "B=$B round-3 form: "; DSP_STREAM_NO_LANE=1 rate $B; done
for B in 16 64; do echo "full solve B=$B:"; rate $B 1600000; done
echo "full solve B=16 round-3 form:"; DSP_STREAM_NO_LANE=1 rate 16 1600000
} 2>&1 | tee "$out/r40h_lane_rates.log"
