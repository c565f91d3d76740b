#!/bin/bash
# trace the first lane the interior-point form gives up on: tools/probes/ipm_trace_failing.sh <B> [ENV=VAL ...]
repo="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$repo"
B=$1; shift
env "$@" DSP_IPM_TRACE=1 DSP_IPM_COMPACT=0 timeout 300 python bench.py --workload price_taker --batch $B --solve --warmup 0 --cpu-sample 0 2>/tmp/t1.err >/dev/null
lane=$(grep "lanes (state" /tmp/t1.err | tail -1 | python -c "
import sys
w = sys.stdin.read().split(':', 1)[1].split()[3:] if False else None
" 2>/dev/null)
lane=$(grep "lanes (state" /tmp/t1.err | tail -1 | sed 's/.*iterations)://' | tr ' ' '\n' | grep -n "^[2356]:" | head -1 | cut -d: -f1)
echo "states: $(grep 'lanes (state' /tmp/t1.err | tail -1 | cut -c1-1500)"
echo "first failing lane (1-based among printed): $lane"
[ -z "$lane" ] && exit 0
env "$@" DSP_IPM_TRACE=$lane DSP_IPM_COMPACT=0 timeout 300 python bench.py --workload price_taker --batch $B --solve --warmup 0 --cpu-sample 0 2>&1 >/dev/null | grep "^\[ipm\]" | cut -c1-220
