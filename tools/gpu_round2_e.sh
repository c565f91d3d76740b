#!/bin/bash
# Round-2 measurement recipe (profiles/r02e_*.log): branch-free queue pull under stress, reproduction of the round-1 hang
# with the legacy pull (build that library first:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDSP_LEGACY_PULL
#   -o dispatches_amd/libdsp_legacy_pull.so dispatches_amd/csrc/dsp_{kernels,simplex,stream,capi}.hip ), streaming PDLP,
# hourly LPs on the simplex, SpMV-step variants.
cd "$(dirname "$0")/.."
o=gpurun_out/r02e.log; : > $o
run() { echo "== $*" >> $o; timeout ${TMO:-120} "$@" >> $o 2>&1; echo "rc=$?" >> $o; }
python -c "import torch; torch.ones(1, device='cuda')" > /dev/null 2>&1
run python tools/gpu_drain_probe.py 4096
run python tools/gpu_stress_queue.py 65536 8 1
run python tools/gpu_stress_queue.py 65536 8 0
DSP_LIB=libdsp_legacy_pull.so TMO=40 run python tools/gpu_drain_probe.py 512
run python tools/gpu_stream.py 168 8 200000 64
run python tools/gpu_stream.py 168 16 200000 64
run python -m pytest tests/test_hip_stream.py -m gpu -q -x
run python tools/gpu_hourly_lps.py
for wl in wind_battery_24h wind_battery_48h; do
  run python tools/gpu_spmv_sweep.py $wl
  DSP_SPMV_WAVES_PER_CU=8 run python tools/gpu_spmv_sweep.py $wl
  DSP_SPMV_WAVES_PER_CU=16 run python tools/gpu_spmv_sweep.py $wl
  DSP_SPMV_LDS=1 run python tools/gpu_spmv_sweep.py $wl
done
grep -v amdgpu.ids $o | cut -c1-400
