#!/bin/bash
# Round 3, GPU call Z: the whole -m gpu suite + smoke on the last commit of the round
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/r30z_smoke.log" 2>&1; tail -2 "$out/r30z_smoke.log"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > "$out/r30z_tests.log" 2>&1; tail -5 "$out/r30z_tests.log"
