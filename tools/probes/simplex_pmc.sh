#!/bin/bash
# instruction mix of the hourly simplex kernel inside one simulated day of the wind + battery loop (graphs off): tools/probes/simplex_pmc.sh <tag>
repo="$(cd "$(dirname "$0")/../.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; tag=${1:-sx}
export TMPDIR=/tmp; cd /tmp
cat > /tmp/oneday.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
loop = BatchedWindBatteryDoubleLoop(4096, device=0, use_graphs=False)
loop.run_day(); loop.run_day()
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT"; do
  d=/tmp/sxp_$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf $d
  R=$repo timeout 200 rocprofv3 --pmc $set --output-format csv -d $d -- python /tmp/oneday.py > /dev/null 2>&1
done
python - "$out/${tag}_simplex_pmc.csv" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/sxp_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "simplex" not in row["Kernel_Name"]: continue
        a = acc.setdefault(row["Counter_Name"], []); a.append(float(row["Counter_Value"]))
with open(sys.argv[1], "w") as o:
    w = csv.writer(o); w.writerow(["counter", "dispatches", "mean", "min", "max"])
    for c, v in acc.items():
        w.writerow([c, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
print(open(sys.argv[1]).read())
PY
