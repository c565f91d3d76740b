#!/bin/bash
# First GPU call of the next round (prepared at the end of round 3, when the GPU minutes were spent): the whole -m gpu suite (it now
# holds the year-long price-taker LPs in the two-level form of the accumulator), the two-level form at 16 / 64 scenarios against the
# chain, the two-level bidding LPs on the base workloads' fixtures, and the counters of the 8-wide instantiation the two-level LP runs in.
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > "$out/r40a_tests.log" 2>&1; tail -4 "$out/r40a_tests.log"
{
for thr in chain two_level; do for B in 16 64; do
  echo -n "$thr B=$B: "; STREAM_THROUGHPUT=$thr timeout 300 python tools/gpu_stream.py 8736 $B 1600000 64 2>&1 | grep "^T="
done; done
} | tee "$out/r40a_two_level_solves.log"
for T in 24 48; do timeout 300 python tools/gpu_two_level_bidding.py $T 2 2>&1 | grep -v amdgpu; done | tee "$out/r40a_two_level_bidding.log"
for wl in price_taker pem_price_taker; do for thr in chain two_level; do timeout 200 python bench.py --workload $wl --throughput $thr --steps 16 --warmup 2 2>/dev/null | tail -1; done; done > "$out/r40a_stream_bench.jsonl"
cut -c1-300 "$out/r40a_stream_bench.jsonl"
cd /tmp
for set in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU"; do
  d=/tmp/sp_tl_$(echo $set | tr ' ' '_' | cut -c1-30); rm -rf $d
  STREAM_THROUGHPUT=two_level timeout 200 rocprofv3 --pmc $set --output-format csv -d $d -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
done
python - "$out/r40a_two_level_pmc_summary.csv" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/sp_tl_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row["Counter_Name"], row["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
with open(sys.argv[1], "w") as o:
    w = csv.writer(o); w.writerow(["counter", "kernel", "dispatches", "mean_counter_value"])
    for (c, k), (n, s) in acc.items():
        if "dsp::" in k:
            w.writerow([c, k, n, round(s / n, 1)])
print("\n".join(l[:40] + " ... " + l[-30:] for l in open(sys.argv[1]).read().splitlines() if "fused" in l))
PY
