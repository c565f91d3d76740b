"""Development: per-scenario iterations / jumps / final weights of a workload -> gpurun_out/iters_<workload>.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios
for wl in sys.argv[1:]:
    solver = hip_solver.HipPdlpSolver(device=0)
    bidder, model = scenarios.make_batch(wl, 4096, solver)
    solver.solve(model)
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez(f"gpurun_out/iters_{wl}.npz", iters=model.iterations, jumps=model.jumps, w=model.primal_weight, status=model.status, obj=model.objective)
    o = np.argsort(-model.iterations)[:12]
    print(wl, "slowest", [(int(i), int(model.iterations[i]), int(model.jumps[i]), float("%.2e" % model.primal_weight[i])) for i in o])
