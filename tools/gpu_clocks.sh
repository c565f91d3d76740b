cd "$(dirname "$0")/.."
cp dispatches_amd/libdsp_hip.so /tmp/orig.so
cp dispatches_amd/libdsp_hip_clocks.so dispatches_amd/libdsp_hip.so; touch dispatches_amd/libdsp_hip.so
echo "== lone launch B=4096"; timeout 120 python tools/gpu_variants.py 4096 2>&1 | grep "clocks\]\|kernel" | sort | uniq -c | sort -rn | head -12
echo "== bench streams 8"; timeout 200 python bench.py --cpu-sample 0 --no-spmv --streams 8 --steps 8 --warmup 2 2>&1 | grep "clocks\]" | awk '{mhz+=$8; cyc+=$(NF-4); n++} END{print n, "samples: mean shader MHz", mhz/n, "mean cycles/iter", cyc/n}'
timeout 200 python bench.py --cpu-sample 0 --no-spmv --streams 8 --steps 8 --warmup 2 2>&1 | grep "clocks\]" | tail -5
cp /tmp/orig.so dispatches_amd/libdsp_hip.so
