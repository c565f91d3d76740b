"""Development: wall time of the reference-level call Bidder.compute_day_ahead_bids at the metric batch (host + GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for wl, T in (("wind_battery", 24), ("wind_battery", 48)):
    solver = hip_solver.HipPdlpSolver(device=0)
    t = time.perf_counter()
    bidder, model = getattr(scenarios, wl + "_batch")(B, T, solver)
    t_build = time.perf_counter() - t
    t = time.perf_counter(); solver._device_lp(model); t_create = time.perf_counter() - t
    print(f"{wl} {T} h: DeviceLP (dsp_create: scaling, layouts, uploads; first call also loads the code object) {t_create*1e3:.0f} ms", flush=True)
    inner = []
    orig = solver.solve
    def timed(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); inner.append(time.perf_counter() - t0); return r
    solver.solve = timed
    for day in ("2020-01-02", "2020-01-03", "2020-01-04", "2020-01-05", "2020-01-06", "2020-01-07"):
        t = time.perf_counter(); bids = bidder.compute_day_ahead_bids(day, 0); tot = time.perf_counter() - t
        print(f"{wl} {T} h B={B}: build {t_build*1e3:.0f} ms; compute_day_ahead_bids {tot*1e3:.1f} ms of which solver.solve "
              f"{inner[-1]*1e3:.1f} ms (kernel {solver.last_stats.kernel_ms:.1f} ms), host rest {(tot-inner[-1])*1e3:.1f} ms; "
              f"optimal {int((model.status == 0).sum())}/{B}", flush=True)
