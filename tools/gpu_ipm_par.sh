# GPU development recipe: the time-parallel form of the interior-point solves (csrc/dsp_ipm_seq.hpp) - tests, the 256-member year-long
# solve under a kernel trace, a sweep of the partition count.      bash tools/gpu_ipm_par.sh <tag> [partition counts to sweep]
tag=${1:-ipmpar}; shift; sweep=${@:-"32 128"}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ipm.py -x -q --timeout 500 > "$out/${tag}_ipm_tests.log" 2>&1; tail -5 "$out/${tag}_ipm_tests.log"
( cd /tmp; rm -rf /tmp/trp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -- python $repo/bench.py --workload price_taker --batch 256 --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > "$out/${tag}_solve256.json"
  f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_ipm_kernel_stats_T8736_B256.csv" && head -24 "$f" | cut -c1-170 )
python - "$out/${tag}_solve256.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); c = d["config"]
print("B=256:", d["value"], d["unit"], "| s/batch", c.get("seconds_per_batch"), "| ms/Newton", c.get("ms_per_newton_iteration_of_the_batch"), "| parts", c.get("time_partitions"),
      "| newton max", c.get("max_newton_iterations"), "| err", c.get("max_rel_objective_error_vs_oracle_fixture"), "| solved", c.get("solved_to_optimality"))
PY
for P in $sweep; do
  DSP_IPM_PARTS=$P timeout 200 python bench.py --workload price_taker --batch 256 --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print('parts', c.get('time_partitions'), '|', d['value'], d['unit'], '| ms/Newton', c.get('ms_per_newton_iteration_of_the_batch'), '| err', c.get('max_rel_objective_error_vs_oracle_fixture'))"
done 2>&1 | tee "$out/${tag}_parts_sweep.log"
DSP_IPM_PARTS=64 timeout 200 python bench.py --workload price_taker --batch 64 --solve --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > "$out/${tag}_solve64.json"; cut -c1-400 "$out/${tag}_solve64.json"
