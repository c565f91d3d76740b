# GPU development recipe: kernel times of the interior-point form on a small batch (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
T=${1:-672}; B=${2:-16}
rm -rf /tmp/prof_ipm
(cd $GRAFT_REPO_ROOT && IPM_CHECK_SKIP_PDHG=1 IPM_CHECK_HIGHS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ipm -- python tools/gpu_ipm_check.py $T $B) 2>&1 | grep "^ipm:"
f=$(find /tmp/prof_ipm -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/ipm_kernel_stats_T${T}_B${B}.csv
head -16 "$f" | cut -c1-150
