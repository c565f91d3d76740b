#!/bin/bash
# Round 3, GPU call S: 4 scenarios per workgroup under the deferred form (122 VGPRs = 4 waves x 4 scenarios) against 2 (80 = 6 x 2)
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
{
for rep in 1 2; do for sg in 2 4; do
  echo -n "B=64 sg=$sg: "; DSP_FUSED_SG=$sg timeout 200 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T=" | sed 's/.*-> //'
done; done
for sg in 2 4; do echo -n "B=256 sg=$sg: "; DSP_FUSED_SG=$sg timeout 200 python tools/gpu_stream.py 8736 256 2048 64 2>&1 | grep "^T=" | sed 's/.*-> //'; done
for wl in pem_price_taker nuclear_price_taker; do for sg in 2 4; do echo -n "$wl sg=$sg: "; DSP_FUSED_SG=$sg timeout 200 python bench.py --workload $wl --steps 16 --warmup 2 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])"; done; done
for rb in 240 256; do echo -n "B=64 sg=2 rb=$rb: "; DSP_FUSED_RB=$rb timeout 200 python tools/gpu_stream.py 8736 64 2048 64 2>&1 | grep "^T=" | sed 's/.*-> //'; done
} | tee "$out/r30s_fused_sg4.log"
