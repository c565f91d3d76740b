#!/bin/bash
# Builds the time-stamp variants of the library (libdsp_probe1.so .. libdsp_probe3.so next to libdsp_hip.so) for tools/gpu_lane_probe.sh:
#   bash tools/build_lane_probe.sh        (run it HERE, before gpurun: the objects of _build/ must be current - __graft_entry__.build())
set -e
repo="$(cd "$(dirname "$0")/.." && pwd)"; B="$repo/dispatches_amd/_build"
objs="$B/dsp_kernels.o $B/dsp_simplex.o $B/dsp_stream.o $B/dsp_qp.o $B/dsp_capi.o"
for v in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDSP_LANE_PROBE=$v -c "$repo/dispatches_amd/csrc/dsp_stream_lane.hip" -o "/tmp/lane_probe$v.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$repo/dispatches_amd/libdsp_probe$v.so" $objs "/tmp/lane_probe$v.o"
  echo "built dispatches_amd/libdsp_probe$v.so"
done
