#!/bin/bash
# Round 3, GPU call P: deferred matrix loads x scenarios per workgroup (1: 68 VGPRs = 7 waves per SIMD, 2: 88 = 5)
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp DSP_FUSED_DEFER=1
DSP_FUSED_SG=1 timeout 500 python -m pytest tests/test_hip_stream.py -m gpu -q --timeout 300 -k "fused or year_long" > "$out/r30p_stream_tests_sg1.log" 2>&1; tail -3 "$out/r30p_stream_tests_sg1.log"
{
for rep in 1 2; do for sg in 1 2; do
  echo -n "B=64 sg=$sg: "; DSP_FUSED_SG=$sg timeout 200 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T=" | sed 's/.*-> //'
done; done
for B in 16 256; do for sg in 1 2; do
  echo -n "B=$B sg=$sg: "; DSP_FUSED_SG=$sg timeout 200 python tools/gpu_stream.py 8736 $B 2048 64 2>&1 | grep "^T=" | sed 's/.*-> //'
done; done
for wl in pem_price_taker nuclear_price_taker; do for sg in 1 2; do echo -n "$wl sg=$sg: "; DSP_FUSED_SG=$sg timeout 200 python bench.py --workload $wl --steps 16 --warmup 2 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])"; done; done
for rb in 125 188; do echo -n "B=64 sg=1 rb=$rb: "; DSP_FUSED_RB=$rb DSP_FUSED_SG=1 timeout 200 python tools/gpu_stream.py 8736 64 2048 64 2>&1 | grep "^T=" | sed 's/.*-> //'; done
} | tee "$out/r30p_fused_sg.log"
