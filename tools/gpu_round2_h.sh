#!/bin/bash
cd "$(dirname "$0")/.."
tag=${1:-r02h}
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/${tag}_pytest.log; tail -3 gpurun_out/${tag}_pytest.log
bash tools/gpu_profile.sh $tag 2>&1 | tail -30
bash tools/gpu_configs.sh $tag
