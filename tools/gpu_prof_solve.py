import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
from dispatches_amd import hip_solver, scenarios
solver = hip_solver.HipPdlpSolver(device=0)
bidder, model = scenarios.wind_battery_batch(4096, 24, solver)
bidder.compute_day_ahead_bids("2020-01-02", 0); bidder.compute_day_ahead_bids("2020-01-03", 0)
cProfile.run('bidder.compute_day_ahead_bids("2020-01-04", 0)', '/tmp/p.out')
pstats.Stats('/tmp/p.out').sort_stats('cumulative').print_stats(30)
