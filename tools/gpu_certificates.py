"""GPU development tool: at which iteration the infeasibility / unboundedness certificates end a scenario, per path (the numbers quoted
in DESIGN.md section 4 "Certificates"; the tests - tests/test_hip_infeasible.py - only assert <= 5000).   python tools/gpu_certificates.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver


def own(model):
    B = model.n_scenario
    lb, ub, rlo, rhi = model.scenario_bounds()
    full = lambda a: np.broadcast_to(a, (B, a.shape[-1])).copy()
    model.lb, model.ub, model.rlo, model.rhi = full(lb), full(ub), full(rlo), full(rhi)
    model.c = model.c.copy()
    model.x = model.y = None


solver = HipPdlpSolver(device=0)
for wl in ("wind_battery_24h", "wind_battery_48h", "wind_battery_24h_qp01"):
    bidder, model = scenarios.make_batch(wl, 64, solver)
    own(model)
    j = model.lp.col_names.index("battery.initial_state_of_charge")
    model.lb[::2, j] = model.ub[::2, j] = 1.0e6
    solver.solve(model)
    it = model.iterations[::2]
    print(f"{wl}: initial charge 10 x the battery in 32 of 64 scenarios: statuses {np.bincount(model.status, minlength=4).tolist()} certified at iteration "
          f"min {it.min()} median {int(np.median(it))} max {it.max()}; kernel {solver.last_stats.kernel_ms:.2f} ms", flush=True)
for wl, col in (("nuclear_24h", "day_ahead_power[5]"), ("wind_battery_24h", "day_ahead_power[5]")):
    bidder, model = scenarios.make_batch(wl, 64, solver)
    own(model)
    j = model.lp.col_names.index(col)
    model.lb[::2, j] = -np.inf
    model.c[::2, j] = 3.0
    solver.solve(model)
    it = model.iterations[::2]
    print(f"{wl}: {col} free and paid in 32 of 64 scenarios: statuses {np.bincount(model.status, minlength=4).tolist()} ended at iteration "
          f"min {it.min()} median {int(np.median(it))} max {it.max()}", flush=True)
for form, T, B, env in (("lane", 336, 40, {}), ("tile", 336, 6, {"DSP_STREAM_NO_LANE": "1"}), ("two_launch", 336, 6, {"DSP_STREAM_NO_LANE": "1", "DSP_STREAM_NO_FUSED": "1"}),
                        ("block", 96, 5, {}), ("lane, year-long", 8736, 64, {})):
    for k in ("DSP_STREAM_NO_LANE", "DSP_STREAM_NO_FUSED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    s2 = HipPdlpSolver(device=0, check_every=64)
    handles, model = scenarios.price_taker_batch(T, B, s2)
    own(model)
    i = model.lp.row_names.index("splitter.sum_split[5]")
    model.rlo[3, i] = model.rhi[3, i] = 1.0e9
    s2.solve(model)
    print(f"streaming {form} (T = {T}, B = {B}): impossible power balance in member 3: status {model.status[3]} at iteration {model.iterations[3]}; the others "
          f"{np.bincount(np.delete(model.status, 3), minlength=2).tolist()} after {int(np.delete(model.iterations, 3).mean())} iterations on average; "
          f"solve {s2.last_stats.kernel_ms:.0f} ms, phases {s2.last_stats.stream_phases}", flush=True)
