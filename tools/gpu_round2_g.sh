#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r02g.log; : > $o
run() { echo "== $*" >> $o; timeout ${TMO:-150} "$@" >> $o 2>&1; echo "rc=$?" >> $o; }
python -c "import torch; torch.ones(1, device='cuda')" > /dev/null 2>&1
run python -m pytest tests/test_hip_stream.py -m gpu -q -x
run python tools/gpu_stream.py 168 16 200000 64
run python tools/gpu_stream.py 8736 4 6400 64
run python tools/gpu_stream.py 8736 16 6400 64
run python tools/gpu_stream.py 8736 64 3200 64
grep -v amdgpu.ids $o | cut -c1-400
