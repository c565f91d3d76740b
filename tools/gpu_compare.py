"""Development: per-scenario iteration differences between two option settings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios

wl = sys.argv[1]
res = []
for spec in sys.argv[2:4]:
    os.environ["DSP_OPTIONS"] = "" if spec == "base" else spec
    solver = hip_solver.HipPdlpSolver(device=0)
    bidder, model = scenarios.make_batch(wl, 4096, solver)
    solver.solve(model)
    res.append((model.iterations.copy(), model.primal_weight.copy(), model.status.copy(), model.jumps.copy()))
(ia, wa, sa, ja), (ib, wb, sb, jb) = res
d = np.nonzero(ia != ib)[0]
print(wl, sys.argv[2], "vs", sys.argv[3], ": differing", len(d), "mean", ia.mean(), ib.mean(), "max", ia.max(), ib.max())
for s in d[np.argsort(-np.abs(ia[d] - ib[d]))][:25]:
    print(f"  scenario {s}: iters {ia[s]} vs {ib[s]}  w {wa[s]:.3e} vs {wb[s]:.3e}  jumps {ja[s]} vs {jb[s]} status {sa[s]} {sb[s]}")
