#!/bin/bash
# Round 3, GPU call Y: small batches x scenarios per workgroup (rounds of workgroups vs wave slots)
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
for B in 1 4 8 16 32; do for sg in 1 2 4; do
  echo -n "B=$B sg=$sg: "; DSP_FUSED_SG=$sg timeout 100 python tools/gpu_stream.py 8736 $B 2048 64 2>&1 | grep "^T=" | sed 's/.*-> //'
done; done | tee "$out/r30y_small_batches.log"
