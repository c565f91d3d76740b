"""GPU dev tool: how many scenarios the kernel accepts WITHOUT certifying the objective accuracy (DSP_FLAG_OBJ_WAIVED) on every
bench workload, and what HipPdlpSolver's re-solves (RECERTIFY_VARIANTS) do with them.  python tools/gpu_recertify.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
GOLD = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
fx = np.load(os.path.join(GOLD, "oracle_objectives.npz"))
qfx = np.load(os.path.join(GOLD, "oracle_qp.npz"))
cases = [(w, None) for w in list(scenarios.WORKLOADS) + list(scenarios.QP_WORKLOADS)]
cases.append(("wind_battery 24h bus303 stride 37", lambda s: scenarios.wind_battery_batch(B, 24, s, series="rts_gmlc_303.npz", stride=37)))
for name, fn in cases:
    for rc in (0, 3):
        solver = hip_solver.HipPdlpSolver(device=0, recertify=rc)
        if fn is None:
            bidder, model = scenarios.make_batch(name, B, solver)
        else:
            bidder, model = fn(solver); scenarios.load_prices(bidder, model)
        t = time.time(); solver.solve(model); wall = time.time() - t
        st, fl = model.status, model.flags
        err = None
        if name in fx.files and len(fx[name]) >= B:
            ref = fx[name][:B]; e = np.abs(model.objective - ref) / np.maximum(1, np.abs(ref)); err = (e.max(), e[model.uncertified].max() if model.uncertified.any() else 0.0)
        elif f"{name}/upper" in qfx.files:
            u, l = qfx[f"{name}/upper"][:B], qfx[f"{name}/lower"][:B]; o = model.objective[:len(u)]
            e = np.maximum(np.maximum(l - o, o - u), 0) / np.maximum(1, np.abs(u)); err = (e.max(), e[model.uncertified[:len(u)]].max() if model.uncertified[:len(u)].any() else 0.0)
        print(f"{name} recertify={rc}: optimal {(st == 0).sum()}/{B} flagged-left {int(model.uncertified.sum())} recertified {solver.last_recertified} "
              f"iters mean {model.iterations.mean():.0f} max {model.iterations.max()} wall {1e3 * wall:.1f} ms err(max, max over flagged) {err}", flush=True)
