#!/bin/bash
# Round 4, call J: lane kernel variants: interleave of 2 (128 VGPRs, 4 waves per SIMD) vs 4, non-temporal iterate traffic.
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
cp dispatches_amd/libdsp_hip.so /tmp/libdsp_default.so
rate() { timeout 200 python tools/gpu_stream.py 8736 $1 ${2:-3200} 64 2>&1 | grep "^T=" | sed 's/.*status/status/' | cut -c1-200; }
{
for v in default il2 il2nt il4nt default; do
  if [ $v = default ]; then cp /tmp/libdsp_default.so dispatches_amd/libdsp_hip.so; else cp dispatches_amd/libdsp_hip_$v.so dispatches_amd/libdsp_hip.so; fi
  for B in 64 256; do echo -n "$v B=$B: "; rate $B; done
  for B in 64 256; do echo -n "$v B=$B waves=4096: "; DSP_LANE_WAVES=4096 rate $B; done
  echo -n "$v B=256 ring>=16: "; DSP_LANE_RING_MIN=16 rate 256
  echo -n "$v B=32: "; rate 32
done
} 2>&1 | tee "$out/r40j_lane_variants.log"
cp /tmp/libdsp_default.so dispatches_amd/libdsp_hip.so
