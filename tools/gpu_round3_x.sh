#!/bin/bash
# Round 3, GPU call X: the streaming tests on the final build (host-side LDS guard added after call W)
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_hip_stream.py tests/test_hip_parity.py -m gpu -q --timeout 300 > "$out/r30x_tests.log" 2>&1; tail -3 "$out/r30x_tests.log"
timeout 200 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T=" | tee "$out/r30x_rate.log"
