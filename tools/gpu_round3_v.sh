#!/bin/bash
# Round 3, GPU call V: 32-bit offsets + x0 parked in LDS against HEAD (libdsp_hip_a.so)
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_hip_stream.py -m gpu -q --timeout 300 > "$out/r30v_stream_tests.log" 2>&1; tail -3 "$out/r30v_stream_tests.log"
{
for rep in 1 2 3; do for lib in libdsp_hip_a.so libdsp_hip.so; do
  echo -n "B=64 $lib: "; DSP_LIB=$lib timeout 200 python tools/gpu_stream.py 8736 64 4096 64 2>&1 | grep "^T=" | sed 's/.*-> //'
done; done
for lib in libdsp_hip_a.so libdsp_hip.so; do echo -n "B=256 $lib: "; DSP_LIB=$lib timeout 200 python tools/gpu_stream.py 8736 256 2048 64 2>&1 | grep "^T=" | sed 's/.*-> //'; done
for wl in pem_price_taker nuclear_price_taker; do for lib in libdsp_hip_a.so libdsp_hip.so; do echo -n "$wl $lib: "; DSP_LIB=$lib timeout 200 python bench.py --workload $wl --steps 16 --warmup 2 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])"; done; done
} | tee "$out/r30v_fused_narrow.log"
cd /tmp; rm -rf /tmp/sp_u; timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/sp_u -- python $repo/tools/gpu_stream.py 8736 64 1024 64 > /dev/null 2>&1
python - <<'PY' | tee "$out/r30v_stream_pmc_insts.log"
import csv, glob, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/sp_u/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "k_fused" in row["Kernel_Name"]:
            a = acc.setdefault(row["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
for c, (n, s) in acc.items(): print(c, n, round(s / n, 1))
PY
