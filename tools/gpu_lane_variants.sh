#!/bin/bash
# Build variants of the lane form (development):  bash tools/gpu_lane_variants.sh <tag> <lib> [<lib> ...]
#   a variant library = csrc/dsp_stream_lane.hip compiled with the switch under test, linked with the other objects of _build/
tag=${1:-variants}; shift
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(libdsp_hip.so)
cd "$(dirname "$0")/.."; out=gpurun_out; mkdir -p $out
log=$out/${tag}_lane_variants.log; : > $log
run() {  # label, env..., -- args of tools/gpu_stream.py
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo -n "$label: " >> $log
  env "${envs[@]}" DSP_LANE_MIN_B=1 timeout 120 python tools/gpu_stream.py "$@" 2>&1 | grep -A1 "per batch-iteration" | sed 's/.*iters/iters/' | tr '\n' ' ' | sed 's/ Traceback.*//' | cut -c1-260 >> $log; echo >> $log
}
for rep in 1 2; do
for lib in "${libs[@]}"; do
  for B in 1 16 64 256; do run "$lib two-level B=$B" DSP_LIB=$lib -- 8736 $B 3200 64; done
  run "$lib PEM B=64" DSP_LIB=$lib STREAM_FAMILY=pem -- 8736 64 1600 64
done
done
cat $log
