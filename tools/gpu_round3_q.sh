#!/bin/bash
# Round 3, GPU call Q: deferred form with y0 / row bounds parked in LDS: unconstrained registers (libdsp_hip_a.so: 85 / 99 VGPRs = 5 / 4
# waves per SIMD) against the allocator held to 6 / 5 waves (libdsp_hip.so: 80 / 94 VGPRs, no scratch)
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp DSP_FUSED_DEFER=1
timeout 500 python -m pytest tests/test_hip_stream.py -m gpu -q --timeout 300 > "$out/r30q_stream_tests.log" 2>&1; tail -3 "$out/r30q_stream_tests.log"
{
for rep in 1 2 3; do for lib in libdsp_hip_a.so libdsp_hip.so; do
  echo -n "B=64 $lib: "; DSP_LIB=$lib timeout 200 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T=" | sed 's/.*-> //'
done; done
for B in 16 256; do for lib in libdsp_hip_a.so libdsp_hip.so; do
  echo -n "B=$B $lib: "; DSP_LIB=$lib timeout 200 python tools/gpu_stream.py 8736 $B 2048 64 2>&1 | grep "^T=" | sed 's/.*-> //'
done; done
for wl in pem_price_taker nuclear_price_taker; do for lib in libdsp_hip_a.so libdsp_hip.so; do echo -n "$wl $lib: "; DSP_LIB=$lib timeout 200 python bench.py --workload $wl --steps 16 --warmup 2 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])"; done; done
} | tee "$out/r30q_fused_waves.log"
cd /tmp; rm -rf /tmp/sp_trace; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $repo/tools/gpu_stream.py 8736 64 2048 64 > /dev/null 2>&1
f=$(find /tmp/sp_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r30q_stream_kernel_stats.csv" && head -3 "$f" | cut -c1-200
