"""Development: the hourly part of the double loop on the GPU - real-time bids (B scenarios x 4 h) and the tracker
(1 LP x 4 h) - wall time per call, iterations, kernel geometry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios
from dispatches_amd.workflow import Tracker
from tests.test_hip_parity import _wind_battery_objects

d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dispatches_amd/data/rts_gmlc_309.npz"))
rts = {k: d[k] for k in d.files}
for B in (256, 4096):
    solver = hip_solver.HipPdlpSolver(device=0)
    bidder, model = scenarios.wind_battery_batch(B, 24, solver)
    da_bids = bidder.compute_day_ahead_bids("2020-01-02", 0)
    da_prices = list(np.clip(rts["da_lmp"][:24], 0, 500)); da_disp = [0.0] * 24
    for hour in range(4):
        t = time.perf_counter(); bidder.compute_real_time_bids("2020-01-02", hour, da_prices, da_disp); tot = time.perf_counter() - t
        m = bidder.real_time_model; st = solver.last_stats
        print(f"RT bids B={B} hour {hour}: {tot*1e3:.1f} ms (kernel {st.kernel_ms:.2f} ms, n={m.lp.n} m={m.lp.m}, iters mean {m.iterations.mean():.0f} max {m.iterations.max()}, "
              f"grid {st.grid_blocks}x{st.block_threads}, matreg {st.matreg}, optimal {(m.status == 0).sum()}/{B})", flush=True)
for warm in (False, True):
    solver = hip_solver.HipPdlpSolver(device=0)
    tr = Tracker(tracking_model_object=_wind_battery_objects(rts, False), tracking_horizon=4, n_tracking_hour=1, solver=solver, warm_start=warm)
    for h in range(6):
        cf = rts["rt_cf"][h:h + 4] * 200
        D = [max(0.0, 0.8 * c) for c in cf]
        t = time.perf_counter(); prof = tr.track_market_dispatch(market_dispatch=D, date="2020-01-02", hour=h); tot = time.perf_counter() - t
        tr.update_model(**prof)
        print(f"tracker {'warm' if warm else 'cold'} hour {h}: {tot*1e3:.2f} ms (kernel {solver.last_stats.kernel_ms:.2f} ms, iters {int(tr.model.iterations[0])}, n={tr.model.lp.n} m={tr.model.lp.m})", flush=True)
