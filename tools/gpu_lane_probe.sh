#!/bin/bash
# Time stamps of the lane kernel's waves:  bash tools/build_lane_probe.sh (here), then gpurun -- bash tools/gpu_lane_probe.sh <tag>
#   hipcc ... -DDSP_LANE_PROBE=1|2 -c csrc/dsp_stream_lane.hip, linked with the other objects of _build/)
tag=${1:-probe}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
for v in ${PROBES:-1 2 3}; do for B in 1 64 256; do
  DSP_LIB=libdsp_probe$v.so DSP_LANE_MIN_B=1 DSP_LANE_PROBE_OUT=$out/${tag}_probe${v}_B$B.bin timeout 120 python tools/gpu_stream.py 8736 $B 192 64 2>&1 | grep "per batch-iteration" | sed "s/.*->/probe$v B=$B:/"
done; done
python tools/lane_probe_report.py $out/${tag}_probe*_B*.bin | tee $out/${tag}_probe_report.txt
