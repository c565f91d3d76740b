#!/bin/bash
# Round 3, GPU call L: k_control with parallel strands - streaming tests + rate + trace
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_hip_stream.py -m gpu -q --timeout 300 > "$out/r30l_stream_tests.log" 2>&1; tail -4 "$out/r30l_stream_tests.log"
for B in 16 64 256; do timeout 200 python tools/gpu_stream.py 8736 $B 4096 64 2>&1 | grep "^T="; done | tee "$out/r30l_fused_B.log"
cd /tmp; rm -rf /tmp/sp_trace; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > /dev/null 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30l_stream_kernel_stats.csv" && head -8 "$f" | cut -c1-200
cd "$repo"; timeout 300 python tools/gpu_stream.py 8736 16 1600000 64 2>&1 | grep -v amdgpu | tee "$out/r30l_stream_T8736_B16.log" | tail -8
