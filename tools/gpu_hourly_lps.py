"""GPU dev tool: the six hourly LP families at batch 4096 (fixture inputs): status, pivots / iterations, kernel time, how
many scenarios the in-wave simplex left to the PDLP kernel, objective error vs the oracle fixture.
    python tools/gpu_hourly_lps.py [no_simplex]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver

fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "oracle_hourly.npz"))
extra = dict(no_simplex=1) if "no_simplex" in sys.argv[1:] else {}
for case in ("wind_battery_rt4", "wind_pem_rt4", "nuclear_rt12", "wind_battery_track4", "wind_pem_track4", "nuclear_track4"):
    inp = {k.split("/", 1)[1]: fx[k] for k in fx.files if k.startswith(case + "/")}
    res = {}
    for tag, kw in (("default", {}), ("simplex only", dict(max_iter=1))):
        solver = HipPdlpSolver(device=0, **kw, **extra)
        if "rt" in case.split("_")[-1]:
            _, model = scenarios.hourly_bid_batch(case, inp, solver)
            shift = (inp["da"] * inp["dispatch"]).sum(1)
        else:
            _, model = scenarios.hourly_tracking_batch(case, inp, solver)
            shift = 0.0
        solver.solve(model)
        solver.solve(model)
        st = solver.last_stats
        err = np.abs(model.objective + shift - inp["obj"]) / np.maximum(1, np.abs(inp["obj"]))
        ok = model.status == 0
        res[tag] = (f"{tag}: status {np.bincount(model.status, minlength=5).tolist()} iters mean {model.iterations.mean():.1f} max "
                    f"{model.iterations.max()} kernel {st.kernel_ms:.3f} ms simplex={st.simplex} obj err max {err[ok].max() if ok.any() else float('nan'):.2e}")
    print(case, "|", res["default"], "|", res["simplex only"], flush=True)
