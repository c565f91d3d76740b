#!/bin/bash
# counters of the lane kernel (k_lane<.., 0>) on the year-long batch: tools/gpu_lane_pmc.sh <tag> <B> [iterations]
#   STREAM_FAMILY=pem | nuclear (tools/gpu_stream.py) + WORKLOAD=<bench.py workload name>: the summary is then named
#   <tag>_pmc_summary_<workload>_B<B>.csv, the pattern bench.py looks for; T=8784 for the nuclear family
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"
tag=$1; B=$2; it=${3:-640}; T=${T:-8736}
name="${tag}_lane_pmc_summary_B$B.csv"; [ -n "$WORKLOAD" ] && name="${tag}_pmc_summary_${WORKLOAD}_B$B.csv"
export TMPDIR=/tmp; cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  d=/tmp/pmc_$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf $d
  DSP_LANE_GRAPH=0 timeout 200 rocprofv3 --pmc $set --output-format csv -d $d -- python $repo/tools/gpu_stream.py $T $B $it 64 > /dev/null 2>&1
done
python - "$out/$name" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row["Counter_Name"], row["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
with open(sys.argv[1], "w") as o:
    w = csv.writer(o); w.writerow(["counter", "kernel", "dispatches", "mean_counter_value"])
    for (c, k), (n, s) in acc.items():
        if "k_lane" in k:
            w.writerow([c, k.replace("(anonymous namespace)::", "").split("(")[0][-70:], n, round(s / n, 1)])
print("\n".join(l for l in open(sys.argv[1]).read().splitlines() if "false, 0>" in l or "k_lane_long<0>" in l))
PY
