#!/bin/bash
# Round 4, call F: lane kernel with `compute` written in steps (records, gathers, arithmetic over all slots of a unit), with and
# without the second register set (prefetch a unit ahead).
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
cp dispatches_amd/libdsp_hip.so /tmp/libdsp_default.so
rate() { timeout 200 python tools/gpu_stream.py 8736 $1 ${2:-3200} 64 2>&1 | grep "^T=" | sed 's/.*status/status/' | cut -c1-200; }
{
for v in default pf; do
  if [ $v = default ]; then cp /tmp/libdsp_default.so dispatches_amd/libdsp_hip.so; else cp dispatches_amd/libdsp_hip_$v.so dispatches_amd/libdsp_hip.so; fi
  for B in 1 16 64 256; do echo -n "$v B=$B: "; rate $B; done
  for B in 64 256; do echo -n "$v B=$B waves=3072: "; DSP_LANE_WAVES=3072 rate $B; done
  for B in 64 256; do echo -n "$v B=$B waves=4096: "; DSP_LANE_WAVES=4096 rate $B; done
  echo -n "$v B=256 ring>=16: "; DSP_LANE_RING_MIN=16 rate 256
  echo -n "$v B=256 ring>=16 waves=3072: "; DSP_LANE_WAVES=3072 DSP_LANE_RING_MIN=16 rate 256
  echo -n "$v B=1024: "; rate 1024 1280
done
} 2>&1 | tee "$out/r40f_lane_variants.log"
cp /tmp/libdsp_default.so dispatches_amd/libdsp_hip.so
timeout 600 python -m pytest tests/test_hip_stream.py -m gpu -q -x --timeout 500 2>&1 | tail -3 | tee "$out/r40f_stream_tests.log"
cd /tmp; rm -rf /tmp/tr; DSP_LANE_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python $repo/tools/gpu_stream.py 8736 256 1280 64 > /dev/null 2>&1
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/r40f_lane_kernel_stats_B256.csv" && head -6 "$f" | cut -c1-200
