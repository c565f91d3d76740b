#!/bin/bash
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
echo "== XCD=1 diagnosis"; DSP_FUSED_V=2 DSP_FUSED_RB=250 DSP_FUSED_SG=4 DSP_FUSED_XCD=1 timeout 90 python tools/gpu_stream.py 8736 64 256 64 2>&1 | grep -v amdgpu.ids | tail -15
echo "== new tests with XCD=0"; DSP_FUSED_XCD=0 timeout 600 python -m pytest "tests/test_hip_stream.py::test_nuclear_price_taker_enumeration_on_the_gpu" "tests/test_hip_rolling.py::test_rolling_hours_are_optimal_for_the_oracles_lps" -m gpu -q 2>&1 | tail -15
