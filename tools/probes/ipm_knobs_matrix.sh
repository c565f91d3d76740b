cd /root/repo
echo "== T=8736"; IPM_T=8736 python tools/gpu_ipm_knobs.py "32 48 64 96 128" "" "DSP_IPM_MAX_UNDO=0"
echo "== T=4368"; IPM_T=4368 python tools/gpu_ipm_knobs.py "1 32 64" "" "DSP_IPM_MAX_UNDO=0"
echo "== T=336"; IPM_T=336 python tools/gpu_ipm_knobs.py "1 7" "" "DSP_IPM_MAX_UNDO=0"
echo "== T=672"; IPM_T=672 python tools/gpu_ipm_knobs.py "1 8" ""
echo "== T=2688"; IPM_T=2688 python tools/gpu_ipm_knobs.py "1 32" ""
