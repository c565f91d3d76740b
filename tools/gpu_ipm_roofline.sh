# Roofline evidence of the interior-point form's banded solve at FULL width: the 16-member family 16 times over (every lane group holds every
# member: all groups stay full to the end) with lane packing off, so that the average kernel durations and counters belong to ONE geometry.
#      bash tools/gpu_ipm_roofline.sh <tag>
tag=${1:-r65b}
repo="$(cd "$(dirname "$0")/.." && pwd)"; out="$repo/gpurun_out"; mkdir -p "$out"; cd "$repo"; export TMPDIR=/tmp
( cd /tmp; rm -rf /tmp/trp; DSP_IPM_COMPACT=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -- python $repo/bench.py --workload price_taker --family base --batch 256 --solve --warmup 1 --cpu-sample 0 > /tmp/trp.json 2>/dev/null
  tail -1 /tmp/trp.json | cut -c1-300
  f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_ipm_kernel_stats_T8736_B256.csv" && head -7 "$f" | cut -c1-150 )
IPM_CHECK_FAMILY=base DSP_IPM_COMPACT=0 bash tools/gpu_ipm_pmc.sh $tag 256 | tail -3
