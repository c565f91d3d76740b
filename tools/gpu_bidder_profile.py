"""GPU development tool: where the host time of Bidder.compute_day_ahead_bids goes (cProfile over a few calls at the metric batch)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dispatches_amd import hip_solver, scenarios

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
solver = hip_solver.HipPdlpSolver(device=0)
bidder, model = scenarios.wind_battery_batch(B, 24, solver)
days = [f"2020-01-{d:02d}" for d in range(2, 28)]
for d in days[:3]:
    bidder.compute_day_ahead_bids(d, 0)
pr = cProfile.Profile()
pr.enable()
for d in days[3:13]:
    bidder.compute_day_ahead_bids(d, 0)
pr.disable()
print("10 calls of compute_day_ahead_bids, B =", B)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
