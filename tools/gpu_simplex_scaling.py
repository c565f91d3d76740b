"""GPU development tool: in-wave simplex kernel time vs batch size (one LP per wave: latency- or LDS-throughput-bound?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "oracle_hourly.npz"))
for case in ("wind_battery_rt4", "wind_battery_track4", "nuclear_rt12"):
    inp_all = {k.split("/", 1)[1]: fx[k] for k in fx.files if k.startswith(case + "/")}
    for B in (256, 1024, 2048, 3840, 4096):
        inp = {k: (v[:B] if getattr(v, "ndim", 0) >= 1 and len(v) >= B else v) for k, v in inp_all.items()}
        solver = HipPdlpSolver(device=0)
        if "rt" in case.split("_")[-1]:
            _, model = scenarios.hourly_bid_batch(case, inp, solver)
        else:
            _, model = scenarios.hourly_tracking_batch(case, inp, solver)
        solver.solve(model); solver.solve(model)
        best = 1e9
        for _ in range(5):
            solver.solve(model); best = min(best, solver.last_stats.kernel_ms)
        print(f"{case} B={B}: kernel {best*1e3:.0f} us, pivots mean {model.iterations.mean():.1f} max {model.iterations.max()}, optimal {(model.status==0).mean():.3f}", flush=True)
