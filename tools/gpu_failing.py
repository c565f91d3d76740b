"""Development: list the scenarios that miss the iteration limit under DSP_OPTIONS-style overrides."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios

wl = sys.argv[1]
for spec in sys.argv[2:]:
    os.environ["DSP_OPTIONS"] = "" if spec == "base" else spec
    solver = hip_solver.HipPdlpSolver(device=0)
    bidder, model = scenarios.make_batch(wl, 4096, solver)
    solver.solve(model)
    bad = np.nonzero(model.status != 0)[0]
    print(spec, "failing", bad.tolist()[:8], len(bad), "jumps", model.jumps[bad].tolist()[:8],
          "w", model.primal_weight[bad].tolist()[:8], "mean", model.iterations.mean(), "slowest ok", np.sort(model.iterations[model.status == 0])[-3:].tolist(), flush=True)
