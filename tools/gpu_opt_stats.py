"""Development: iteration statistics of workloads under DSP_OPTIONS settings (one solve each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios
wls = sys.argv[1].split(",")
for spec in sys.argv[2:]:
    os.environ["DSP_OPTIONS"] = "" if spec == "base" else spec
    for wl in wls:
        solver = hip_solver.HipPdlpSolver(device=0)
        if wl == "wb303":
            bidder, model = scenarios.wind_battery_batch(4096, 24, solver, series="rts_gmlc_303.npz", stride=37); scenarios.load_prices(bidder, model)
        else:
            bidder, model = scenarios.make_batch(wl, 4096, solver)
        solver.solve(model); solver.solve(model)
        it = model.iterations
        print(f"[{spec}] {wl}: optimal {(model.status == 0).sum()} mean {it.mean():.0f} p99 {np.percentile(it, 99):.0f} top5 {np.sort(it)[-5:].tolist()} kernel {solver.last_stats.kernel_ms:.1f} ms", flush=True)
