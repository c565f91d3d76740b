#!/bin/bash
# development: run bench.py against dispatches_amd/libdsp_hip_<tag>.so variants (copies each over libdsp_hip.so)
#   WL=<workload> STREAMS="1 8" bash tools/gpu_bench_variant.sh tagA tagB ...
cd "$(dirname "$0")/.."
WL=${WL:-wind_battery_24h}
cp dispatches_amd/libdsp_hip.so /tmp/libdsp_hip.orig.so
for tag in "$@"; do
  cp dispatches_amd/libdsp_hip_$tag.so dispatches_amd/libdsp_hip.so; touch dispatches_amd/libdsp_hip.so
  for st in ${STREAMS:-8 12}; do
  timeout 200 python bench.py --workload $WL --cpu-sample 0 --no-spmv --streams $st 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag $WL streams $st', 'value %.0f'%d['value'],'ms/step %.2f'%d['ms_per_step'],'single %.2f'%d['config']['single_batch_latency_ms'],'grid',d['config']['grid'])"
  done
done
cp /tmp/libdsp_hip.orig.so dispatches_amd/libdsp_hip.so
