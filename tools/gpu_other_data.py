"""Development: robustness on scenario sets the defaults were NOT tuned on (other bus, stride, seed, horizons)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios
cases = [("wind_battery 24h bus303 stride 37", lambda s: scenarios.wind_battery_batch(4096, 24, s, series="rts_gmlc_303.npz", stride=37)),
         ("wind_battery 24h stride 23 cap 200", lambda s: scenarios.wind_battery_batch(4096, 24, s, stride=23, price_cap=200.0)),
         ("wind_battery 36h", lambda s: scenarios.wind_battery_batch(4096, 36, s)),
         ("wind_battery 12h", lambda s: scenarios.wind_battery_batch(4096, 12, s)),
         ("wind_pem 24h bus309", lambda s: scenarios.wind_pem_batch(4096, 24, s, series="rts_gmlc_309.npz", stride=17)),
         ("wind_pem 48h stride 11", lambda s: scenarios.wind_pem_batch(4096, 48, s, stride=11)),
         ("nuclear 24h seed 7", lambda s: scenarios.nuclear_batch(4096, 24, s, seed=7)),
         ("nuclear 36h", lambda s: scenarios.nuclear_batch(4096, 36, s))]
for name, fn in cases:
    solver = hip_solver.HipPdlpSolver(device=0)
    bidder, model = fn(solver)
    scenarios.load_prices(bidder, model)
    solver.solve(model)
    it, st = model.iterations, model.status
    s_ = solver.last_stats
    print(f"{name}: n={model.lp.n} m={model.lp.m} matreg {s_.matreg} optimal {(st == 0).sum()}/4096 mean {it.mean():.0f} p99 {np.percentile(it, 99):.0f} max {it.max()} kernel {s_.kernel_ms:.1f} ms", flush=True)
