#!/bin/bash
# Round 6: what the scratch traffic of the 48-h wind + battery kernel costs (pdlp_solve_kernel<7, 4, ...>: 256 VGPRs, 864 B / lane of scratch at two
# waves per SIMD).  Counters of the shipped kernel at the config-4 day-ahead batch: L2 hit rate, vector-memory instructions, where the wave
# cycles go.        bash tools/gpu_scratch_48h.sh <tag>
tag=${1:-r62b}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_HIT_sum\|TCC_MISS_sum\|TCC_EA0_RDREQ_sum\|TCC_EA0_WRREQ_sum\|SQ_INSTS_VMEM_RD\|SQ_INSTS_VMEM_WR\|SQ_INSTS_FLAT\|SQ_INST_CYCLES_VMEM\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_ANY\|SQ_ACTIVE_INST_VMEM\|SQ_ACTIVE_INST_FLAT\|TCP_TCC_READ_REQ_sum\|TCP_TCC_WRITE_REQ_sum\|TCC_REQ_sum\|TCC_READ_sum\|TCC_WRITE_sum\|SQ_INSTS_VALU\|SQ_INSTS_LDS\|SQ_INSTS_SALU" | sort -u | tr '\n' ' ' > "$out/${tag}_available_counters.txt"; cat "$out/${tag}_available_counters.txt"; echo
bench="python $repo/bench.py --workload wind_battery_48h --batch 4096 --cpu-sample 0 --no-spmv --no-eps4 --no-configs --steps 8 --warmup 1 --streams 8 --min-time 0"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/ps_$i
  timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/ps_$i -- $bench > /tmp/ps_$i.json 2>/tmp/ps_$i.err || tail -2 /tmp/ps_$i.err
done
python - "$out/${tag}_wind_battery_48h_scratch_pmc.csv" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/ps_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "pdlp_solve_kernel<7" not in row["Kernel_Name"]: continue
        a = acc.setdefault(row["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
with open(sys.argv[1], "w") as o:
    w = csv.writer(o); w.writerow(["counter", "dispatches", "mean_counter_value"])
    for c, (n, s) in acc.items(): w.writerow([c, n, round(s / n, 1)])
v = {c: s / n for c, (n, s) in acc.items()}
print(open(sys.argv[1]).read())
g = lambda k: v.get(k, float("nan"))
print("L2 hit rate %.4f | EA read requests per launch %.3g (x 64 B = %.3g MB) | vmem rd %.3g wr %.3g valu %.3g lds %.3g wave-instr per launch" % (
    g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_sum") * 64 / 1e6, g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS")))
print("wave cycles %.4g: parked (WAIT_ANY) %.3f, issue-stalled (WAIT_INST_ANY) %.3f, issuing (ACTIVE_INST_ANY) %.3f; of which VMEM %.4f LDS %.3f VALU %.3f" % (
    g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
    g("SQ_ACTIVE_INST_VMEM") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_LDS") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
PY
timeout 200 python $repo/bench.py --workload wind_battery_48h --batch 4096 --cpu-sample 0 --no-spmv --no-configs 2>/dev/null | tail -1 > "$out/${tag}_48h_line.json"
python -c "
import json; d=json.load(open('$out/${tag}_48h_line.json')); print('line:', d['value'], d['ms_per_step'], d['config'].get('single_batch_latency_ms'))"
