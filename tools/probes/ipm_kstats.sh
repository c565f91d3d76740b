#!/bin/bash
# kernel stats of one year-long solve line: tools/probes/ipm_kstats.sh <tag> <B> [rows]
repo="$(cd "$(dirname "$0")/../.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; export TMPDIR=/tmp
tag=$1; B=$2; rows=${3:-14}
cd /tmp; D=/tmp/ks_${B}_$$; rm -rf $D
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $repo/bench.py --workload price_taker --batch $B --solve --warmup 1 --cpu-sample 0 > /dev/null 2>&1
f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_ipm_kernel_stats_distinct_T8736_B$B.csv"
python - "$f" $rows <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1))
for r in rows[:int(sys.argv[2])]:
    print("%-60s %5s %8.1f us %5.1f %%" % (r["Name"].replace("dsp::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
