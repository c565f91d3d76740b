"""Development: the solves of the double loop for consecutive simulated days (no Prescient: the market clears every bid at
its maximum) - day-ahead bids for B scenarios, then per hour real-time bids for B scenarios, the tracker's LP and the
rolling-horizon model updates.  Prints wall time per simulated day and its breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios
from dispatches_amd.workflow import Tracker

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
days = int(sys.argv[2]) if len(sys.argv) > 2 else 3
solver = hip_solver.HipPdlpSolver(device=0)
bidder, model = scenarios.wind_battery_batch(B, 24, solver)
tracker = Tracker(tracking_model_object=bidder.bidding_model_object.__class__(
    model_data=bidder.bidding_model_object.model_data, wind_capacity_factors=list(scenarios.load_series("rts_gmlc_309.npz")["rt_cf"]),
    wind_pmax_mw=200, battery_pmax_mw=25, battery_energy_capacity_mwh=100), tracking_horizon=4, n_tracking_hour=1,
    solver=hip_solver.HipPdlpSolver(device=0), warm_start=True)
gen = bidder.generator
for day in range(days):
    date = f"2020-01-{2 + day:02d}"
    t0 = time.perf_counter()
    da = bidder.compute_day_ahead_bids(date, 0)
    t_da = time.perf_counter() - t0
    p_da = [da[t][gen]["p_max"] for t in range(24)]
    prices = [float(p) for p in np.clip(scenarios.load_series("rts_gmlc_309.npz")["da_lmp"][24 * day:24 * day + 24], 0, 500)]
    t_rt = t_tr = t_up = 0.0
    for hour in range(24):
        t = time.perf_counter(); rt = bidder.compute_real_time_bids(date, hour, prices, p_da); t_rt += time.perf_counter() - t
        dispatch = [rt[hour + k][gen]["p_max"] for k in range(4)]
        t = time.perf_counter(); prof = tracker.track_market_dispatch(market_dispatch=dispatch, date=date, hour=hour); t_tr += time.perf_counter() - t
        t = time.perf_counter(); tracker.update_model(**prof); bidder.update_real_time_model(**prof); t_up += time.perf_counter() - t
    t = time.perf_counter(); bidder.update_day_ahead_model(**prof); t_up += time.perf_counter() - t
    tot = time.perf_counter() - t0
    print(f"day {day} ({B} scenarios): {tot*1e3:.0f} ms = DA bids {t_da*1e3:.0f} + 24 x RT bids {t_rt*1e3:.0f} + 24 x tracker {t_tr*1e3:.0f} "
          f"+ model updates {t_up*1e3:.0f}  ->  {365 * tot:.0f} s per simulated year", flush=True)
