import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
B = 4096
runs = {}
for name, kw in (("default", {}), ("v1", dict(pid_kp=0.45, restart_artificial=0.3)), ("v2", dict(pid_kp=0.8, restart_artificial=0.15)), ("v3", dict(pid_kp=0.3, restart_artificial=0.5, check_every=12))):
    solver = HipPdlpSolver(device=0, recertify=0, **kw)
    bidder, model = scenarios.make_batch("wind_battery_48h", B, solver)
    solver.solve(model)
    runs[name] = model.iterations.copy()
    print(name, "mean", runs[name].mean().round(0), "p99", int(np.quantile(runs[name], 0.99)), "max", runs[name].max(), "optimal", int((model.status == 0).sum()))
d = runs["default"]
top = np.argsort(-d)[:32]
print("default's 32 slowest:", d[top][:12], "...")
for v in ("v1", "v2", "v3"):
    print(" under", v, ":", runs[v][top][:12], "... median", int(np.median(runs[v][top])), "max", runs[v][top].max(), "| corr with default", np.corrcoef(d, runs[v])[0, 1].round(3))
