#!/bin/bash
# HBM counters of the interior-point form's kernels on the year-long batch (separate passes for FETCH_SIZE and WRITE_SIZE, as the guide's
# HBM section prescribes; gfx950: FETCH_SIZE counts half the bytes of a coalesced streaming read - the summary's consumers double it):
#   bash tools/gpu_ipm_pmc.sh <tag> <B>      ->  gpurun_out/<tag>_ipm_pmc_summary_B<B>.csv  (mean counter value per kernel and dispatch)
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"
tag=$1; B=${2:-256}; T=${T:-8736}
export TMPDIR=/tmp; cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/ipmpmc_$set; rm -rf $d
  IPM_CHECK_SKIP_PDHG=1 IPM_CHECK_HIGHS=0 timeout 250 rocprofv3 --pmc $set --output-format csv -d $d -- python $repo/tools/gpu_ipm_check.py $T $B 2>/dev/null | grep "^ipm:"
done
python - "$out/${tag}_ipm_pmc_summary_B$B.csv" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/ipmpmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row["Counter_Name"], row["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
with open(sys.argv[1], "w") as o:
    w = csv.writer(o); w.writerow(["counter", "kernel", "dispatches", "mean_counter_value"])
    for (c, k), (n, s) in acc.items():
        if "ipm" in k or "k_seq" in k:
            w.writerow([c, k.replace("dsp::", "").split("(")[0][-60:], n, round(s / n, 1)])
print(open(sys.argv[1]).read())
PY
