#!/bin/bash
# Round 3, GPU call D: k_fused_pre without data-dependent control flow in its load phase; the price-taker goldens through the HIP path
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_stream.py -m gpu -q > "$out/r30d_stream_tests.log" 2>&1; tail -15 "$out/r30d_stream_tests.log"
for cfg in "1 752 2" "2 250 4" "2 250 2" "2 250 1" "2 500 2" "2 500 1" "2 752 2" "2 752 1" "2 500 4"; do
  set -- $cfg
  echo "== DSP_FUSED_V=$1 DSP_FUSED_RB=$2 DSP_FUSED_SG=$3, B = 64, 4096 iterations"
  DSP_FUSED_V=$1 DSP_FUSED_RB=$2 DSP_FUSED_SG=$3 timeout 300 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T="
done > "$out/r30d_fused_scan.log" 2>&1; cat "$out/r30d_fused_scan.log"
for B in 16 256; do echo "== default, B = $B"; timeout 300 python tools/gpu_stream.py 8736 $B 2048 64 2>&1 | grep "^T="; done | tee "$out/r30d_fused_B.log"
