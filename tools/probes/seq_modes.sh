#!/bin/bash
# walk kernels of the interior-point form at one lane group: full / no compute / no data movement (DSP_IPM_SEQ_MODE), 12 Newton iterations
repo="$(cd "$(dirname "$0")/../.." && pwd)"; export TMPDIR=/tmp; cd /tmp
cat > /tmp/sm.py <<PY
import sys; sys.path.insert(0, "$repo")
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
s = HipPdlpSolver(device=0, check_every=64, max_iter=1100 * 64, recertify=0, no_interior_point=0)
m = scenarios.price_taker_batch(8736, ${1:-60}, s, throughput="chain", family="wide")[1]
s.solve(m)
PY
for mode in 0 1 2; do
  D=/tmp/sm_${mode}_$$; rm -rf $D
  DSP_IPM_SEQ_MODE=$mode DSP_IPM_MAXIT=12 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python /tmp/sm.py > /dev/null 2>&1
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  echo "mode $mode"
  python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_seq" in r["Name"] or "red_solve" in r["Name"] or "border" in r["Name"]:
        print("   %-62s calls %5s avg %7.1f us" % (r["Name"].replace("dsp::", "")[5:67], r["Calls"], float(r["AverageNs"]) / 1000))
PY
done
