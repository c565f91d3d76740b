import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
T, B, bad = 8736, 64, 37
solver = HipPdlpSolver(device=0, check_every=64, max_iter=1_000_000)
handles, model = scenarios.price_taker_batch(T, B, solver, throughput="chain", family="wide")
lb, ub, rlo, rhi = model.scenario_bounds()
row = next(i for i, nm in enumerate(model.lp.row_names) if nm.startswith("splitter.sum_split[5]"))
solver.solve(model)
t0 = time.perf_counter(); solver.solve(model); print("clean", time.perf_counter() - t0, solver.last_stats.newton_iterations if hasattr(solver.last_stats, "newton_iterations") else "")
model.rlo, model.rhi = np.tile(rlo, (B, 1)), np.tile(rhi, (B, 1))
model.rlo[bad, row] = model.rhi[bad, row] = 1e9
t0 = time.perf_counter(); solver.solve(model); print("bad", time.perf_counter() - t0)
st = solver.last_stats
print({k: getattr(st, k) for k, _ in st._fields_ if not k.startswith("reserved")})
print("iters of the bad member", model.iterations[bad], "status", model.status[bad])
