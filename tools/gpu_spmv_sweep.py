"""GPU dev tool: streaming SpMV step, time per launch vs batch size for the variants selected by the environment
(DSP_SPMV_LDS, DSP_SPMV_WAVES_PER_CU).   python tools/gpu_spmv_sweep.py <workload>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import DeviceLP, HipPdlpSolver
wl = sys.argv[1] if len(sys.argv) > 1 else "wind_battery_24h"
_, model = scenarios.make_batch(wl, 2, HipPdlpSolver(device=0))
lp = model.lp
dlp = DeviceLP(lp, 0)
dev = torch.device("cuda", 0)
out = []
for B in (4096, 8192, 16384, 32768, 131072):
    X = torch.randn((B, lp.n), dtype=torch.float64, device=dev); Y = torch.randn((B, lp.m), dtype=torch.float64, device=dev)
    AX = torch.empty((B, lp.m), dtype=torch.float64, device=dev); ATY = torch.empty((B, lp.n), dtype=torch.float64, device=dev)
    for _ in range(5): dlp.spmv_step(X, Y, AX, ATY)
    reps = 200 if B <= 16384 else 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): dlp.spmv_step(X, Y, AX, ATY)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    byt = B * 16 * (lp.n + lp.m)
    out.append(f"B={B}: {us:.2f} us {byt / us / 1e6:.2f} TB/s")
    if B == 4096:
        A = lp.csr(); x = X[:64].cpu().numpy(); y = Y[:64].cpu().numpy()
        err = max(np.abs(AX[:64].cpu().numpy() - x @ A.T).max(), np.abs(ATY[:64].cpu().numpy() - y @ A).max())
        out.append(f"err {err:.1e}")
print(wl, os.environ.get("DSP_SPMV_LDS", "0"), os.environ.get("DSP_SPMV_WAVES_PER_CU", "32"), "|", " | ".join(out))
