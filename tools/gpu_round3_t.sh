#!/bin/bash
# Round 3, GPU call T: what the driver runs at round end - smoke(), the whole -m gpu suite, the default bench line - at HEAD
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/r30t_smoke.log" 2>&1; tail -2 "$out/r30t_smoke.log"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > "$out/r30t_tests.log" 2>&1; tail -5 "$out/r30t_tests.log"
timeout 400 python bench.py > "$out/r30t_bench.json" 2> "$out/r30t_bench.err"; tail -c 900 "$out/r30t_bench.json"; echo
