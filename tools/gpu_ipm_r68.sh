# GPU recipe (round 6, after the small-batch work on the interior-point form): tests, solve lines of the three year-long families, kernel
# stats at 60 and 256 LPs, and the roofline evidence at full width (tools/gpu_ipm_roofline.sh).      bash tools/gpu_ipm_r68.sh <tag>
tag=${1:-r68h}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ipm.py -m gpu -x -q -s > "$out/${tag}_ipm_tests.log" 2>&1; grep -a "\[ipm\]\|passed\|failed\|Error" "$out/${tag}_ipm_tests.log" | tail -8
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print('$1', '|', round(d['value'], 1), d['unit'], '| s/batch', round(c.get('seconds_per_batch'), 3), '| ms/Newton', round(c.get('ms_per_newton_iteration_of_the_batch'), 2), '| newton mean', round(c.get('newton_iterations_per_scenario'), 1), 'max', c.get('max_newton_iterations'),
      '| err', c.get('max_rel_objective_error_vs_oracle_fixture'), 'of', c.get('members_with_oracle_fixture'), '| solved', c.get('solved_to_optimality'), 'ipm', c.get('ipm_solved'), '| distinct', c.get('distinct_members'))"; }
{
for B in 256 64 60 30; do
  timeout 300 python bench.py --workload price_taker --batch $B --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > "$out/${tag}_solve${B}.json"
  line "price_taker B=$B" < "$out/${tag}_solve${B}.json"
done
timeout 300 python bench.py --workload pem_price_taker --batch 64 --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > "$out/${tag}_solve_pem64.json"; line "pem B=64" < "$out/${tag}_solve_pem64.json"
timeout 300 python bench.py --workload nuclear_price_taker --batch 60 --horizon 8784 --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > "$out/${tag}_solve_nuclear60.json"; line "nuclear B=60" < "$out/${tag}_solve_nuclear60.json"
} 2>&1 | tee "$out/${tag}_solve_lines.log"
for B in 60 256; do
  ( cd /tmp; D=/tmp/trp_${B}_$$; rm -rf $D; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $repo/bench.py --workload price_taker --batch $B --solve --warmup 1 --cpu-sample 0 > /dev/null 2>&1
    f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_ipm_kernel_stats_distinct_T8736_B$B.csv" && head -6 "$f" | cut -c1-150 )
done
