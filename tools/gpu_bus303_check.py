import sys, numpy as np
sys.path.insert(0, "/root/repo")
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
from oracle import dispatch_lp_oracle as orc
solver = HipPdlpSolver(device=0)
bidder, model = scenarios.wind_battery_batch(4096, 24, solver, series="rts_gmlc_303.npz", stride=37)
scenarios.load_prices(bidder, model)
solver.solve(model)
print("status", np.bincount(model.status), "flags", np.bincount(model.flags), "iters mean/max", model.iterations.mean(), model.iterations.max(), "1217:", model.iterations[1217], model.flags[1217])
s = scenarios.load_series("rts_gmlc_303.npz")
N, T = len(s["rt_lmp"]), 24
ids = sorted(set([1217] + list(range(0, 4096, 41)) + np.nonzero(model.flags & 1)[0].tolist()))
worst = 0
for k in ids:
    h0 = (37 * k) % (N - T)
    P, *_ = orc.wind_battery_da(T, s["rt_cf"][h0:h0 + T], np.clip(s["da_lmp"][h0:h0 + T], 0, 500), np.clip(s["rt_lmp"][h0:h0 + T], 0, 500))
    ref = P.solve(tight=True)[1]
    e = abs(model.objective[k] - ref) / max(1.0, abs(ref))
    if e > 2e-7 or k == 1217 or (model.flags[k] & 1):
        print(k, "obj", model.objective[k], "ref", ref, "rel err", e, "flags", model.flags[k], "iters", model.iterations[k], "scale", float(np.abs(model.c[k] * model.x[k]).sum()))
    worst = max(worst, e)
print("worst plain relative error over", len(ids), "scenarios:", worst)
