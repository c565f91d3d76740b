# GPU recipe (round 6, last state): interior-point form - tests, solve lines, and the roofline evidence of its kernels at FULL width
# (DSP_IPM_COMPACT=0: every launch processes all lane groups, so average kernel durations and counters belong to one geometry).
#      bash tools/gpu_ipm_r6b.sh <tag>
tag=${1:-r64a}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ipm.py -m gpu -x -q -s > "$out/${tag}_ipm_tests.log" 2>&1; grep -a "\[ipm\]\|passed\|failed\|Error" "$out/${tag}_ipm_tests.log" | tail -6
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print('$1', '|', round(d['value'], 1), d['unit'], '| s/batch', round(c.get('seconds_per_batch'), 3), '| ms/Newton', round(c.get('ms_per_newton_iteration_of_the_batch'), 2), '| newton mean', round(c.get('newton_iterations_per_scenario'), 1), 'max', c.get('max_newton_iterations'),
      '| err', c.get('max_rel_objective_error_vs_oracle_fixture'), 'of', c.get('members_with_oracle_fixture'), '| solved', c.get('solved_to_optimality'), 'ipm', c.get('ipm_solved'), '| distinct', c.get('distinct_members'))"; }
for B in 256 64 60 30; do
  timeout 300 python bench.py --workload price_taker --batch $B --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > "$out/${tag}_solve${B}.json"
  line "B=$B" < "$out/${tag}_solve${B}.json"
done 2>&1 | tee "$out/${tag}_solve_lines.log"
( cd /tmp; rm -rf /tmp/trp; DSP_IPM_COMPACT=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -- python $repo/bench.py --workload price_taker --batch 256 --solve --warmup 1 --cpu-sample 0 > /dev/null 2>&1
  f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_ipm_kernel_stats_T8736_B256.csv" && head -8 "$f" | cut -c1-150 )
DSP_IPM_COMPACT=0 bash tools/gpu_ipm_pmc.sh $tag 256 | tail -5
