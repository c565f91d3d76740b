"""Development: the scenario sets the 8-GPU weak-scaling bench gives to ranks 0..7 (ids r*4096 .. (r+1)*4096-1): all must
converge; per-rank iteration statistics; objectives saved for an oracle spot check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios
wl = sys.argv[1] if len(sys.argv) > 1 else "wind_battery_24h"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
solver = hip_solver.HipPdlpSolver(device=0)
bidder, model = scenarios.make_batch(wl, 4096 * W, solver)
solver.solve(model)
it, st = model.iterations, model.status
for r in range(W):
    s = slice(r * 4096, (r + 1) * 4096)
    print(f"{wl} rank {r}: optimal {(st[s] == 0).sum()}/4096 mean {it[s].mean():.0f} p99 {np.percentile(it[s], 99):.0f} max {it[s].max()} "
          f"(lone-batch estimate {it[s].max() * 0.36e-3:.1f} ms)", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
np.savez(f"gpurun_out/ranks_{wl}.npz", obj=model.objective, iters=it, status=st)
