#!/bin/bash
# development: LDS counters of the solve kernel for the current library (one rocprofv3 --pmc pass)
repo="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_lds; timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d /tmp/pmc_lds -- python $repo/bench.py --cpu-sample 0 --steps 8 --warmup 1 --streams 8 --no-spmv ${WL:+--workload $WL} > /tmp/pmc_lds.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.OrderedDict()
for f in glob.glob("/tmp/pmc_lds/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pdlp_solve" in row["Kernel_Name"]:
            a = acc.setdefault(row["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
print({k: "%.3e" % (v[1] / v[0]) for k, v in acc.items()})
PY
tail -1 /tmp/pmc_lds.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['simulated_lds_gather_conflict_cycles_per_iteration'])"
