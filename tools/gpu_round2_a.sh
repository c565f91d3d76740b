#!/bin/bash
# Round-2 first GPU call: full -m gpu suite (with the batch-scale setpoint parity tests), solution dumps for offline
# analysis, the default bench line and the other configurations.   bash tools/gpu_round2_a.sh <tag>
tag=${1:-r02a}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"
cd "$repo"
nproc > "$out/${tag}_host.txt"; python -c "import ctypes; ctypes.CDLL('libhiprtc.so'); print('hiprtc ok')" >> "$out/${tag}_host.txt" 2>&1
DSP_DUMP_DIR="$out/${tag}_dump" DSP_DUMP_FULL=wind_battery_24h timeout 900 python -m pytest tests -m gpu -q -rA 2>&1 | tail -120 > "$out/${tag}_pytest.log"
grep -E "passed|failed" "$out/${tag}_pytest.log" | tail -2
timeout 600 python bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"; tail -c 1500 "$out/${tag}_bench.json"; echo
bash tools/gpu_configs.sh "$tag"
