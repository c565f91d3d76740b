#!/bin/bash
# Build variants of the lane form (development):  bash tools/gpu_lane_variants.sh <tag>
tag=${1:-variants}
cd "$(dirname "$0")/.."; out=gpurun_out; mkdir -p $out
log=$out/${tag}_lane_variants.log; : > $log
run() {  # label, env..., -- args of tools/gpu_stream.py
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo -n "$label: " >> $log
  env "${envs[@]}" DSP_LANE_MIN_B=1 timeout 120 python tools/gpu_stream.py "$@" 2>&1 | grep -A1 "per batch-iteration" | sed 's/.*iters/iters/' | tr '\n' ' ' | cut -c1-260 >> $log; echo >> $log
}
for rep in 1 2; do
for lib in libdsp_hip.so libdsp_pf.so libdsp_ch8.so libdsp_pf8.so; do
  for B in 64 256; do
    run "$lib two-level B=$B waves 1024" DSP_LIB=$lib DSP_LANE_WAVES=1024 -- 8736 $B 3200 64
  done
  run "$lib two-level B=16 waves 1024" DSP_LIB=$lib DSP_LANE_WAVES=1024 -- 8736 16 3200 64
done
done
cat $log
