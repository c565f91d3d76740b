"""GPU development tool: per-scenario iteration counts / jumps / flags of whole batches -> gpurun_out/<tag>_iters.npz
    python tools/gpu_iteration_dump.py <tag> [workload ...] [k=v solver options]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver

tag = sys.argv[1]
wls = [a for a in sys.argv[2:] if "=" not in a] or ["wind_battery_24h", "wind_battery_48h", "wind_battery_24h_qp01"]
opts = {k: float(v) if "." in v or "e" in v else int(v) for k, v in (a.split("=") for a in sys.argv[2:] if "=" in a)}
out = {}
for wl in wls:
    solver = HipPdlpSolver(device=0, **opts)
    bidder, model = scenarios.make_batch(wl, 4096, solver)
    solver.solve(model)
    solver.solve(model)
    st = solver.last_stats
    it = model.iterations
    print(f"{wl}: kernel {st.kernel_ms:.2f} ms optimal {st.n_optimal} mean {it.mean():.0f} p50 {np.median(it):.0f} p90 {np.quantile(it, .9):.0f} "
          f"p99 {np.quantile(it, .99):.0f} p99.9 {np.quantile(it, .999):.0f} max {it.max()} flags {np.bincount(model.flags, minlength=4).tolist()} "
          f"top {np.argsort(-it)[:8].tolist()} {np.sort(it)[::-1][:8].tolist()}", flush=True)
    out[f"{wl}/iters"], out[f"{wl}/jumps"], out[f"{wl}/flags"], out[f"{wl}/status"] = it, model.jumps, model.flags, model.status
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"{tag}_iters.npz"), **out)
