import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
for kw in (dict(polish_patience=8), dict(polish_patience=1), dict(polish_patience=8, eps_obj=5e-9), dict(polish_patience=64, eps_obj=5e-10), dict(stall_rescue=200), dict(stall_rescue=50, polish_patience=0),
           dict(polish_patience=8, eps_obj=5e-9, recertify_passes=3), dict(stall_rescue=200, recertify_passes=3)):
    for wl in ("wind_battery_48h", "wind_battery_24h"):
        solver = HipPdlpSolver(device=0, recertify=0, **kw)
        bidder, model = scenarios.make_batch(wl, 1024, solver)
        solver.solve(model)
        print(wl, kw, "optimal", int((model.status == 0).sum()), "flagged", int(((model.flags & 1) != 0).sum()), "iters mean", float(model.iterations.mean()), "max", int(model.iterations.max()), flush=True)
