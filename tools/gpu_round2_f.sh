#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r02f.log; : > $o
run() { echo "== $*" >> $o; timeout ${TMO:-150} "$@" >> $o 2>&1; echo "rc=$?" >> $o; }
python -c "import torch; torch.ones(1, device='cuda')" > /dev/null 2>&1
run python tools/gpu_hourly_lps.py
run python tools/gpu_stream.py 8736 4 20000 64
run python tools/gpu_stream.py 8736 16 20000 64
run python tools/gpu_stream.py 2184 16 400000 64
DSP_SPMV_WPB=4 run python tools/gpu_spmv_sweep.py wind_battery_48h
DSP_SPMV_WPB=2 run python tools/gpu_spmv_sweep.py wind_battery_48h
run python -m pytest tests/test_hip_parity.py -m gpu -q -k "stress or golden or edge or invalid"
grep -v amdgpu.ids $o | cut -c1-400
