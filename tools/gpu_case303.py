import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import hip_solver, scenarios
for spec in ["", "stall_rescue=0", "kkt_gate=0,kkt_every=1", "ray_jumps=0", "check_every=12", "pid_kp=0.5", "jump_tol=0.001"]:
    os.environ["DSP_OPTIONS"] = spec
    solver = hip_solver.HipPdlpSolver(device=0)
    bidder, model = scenarios.wind_battery_batch(4096, 24, solver, series="rts_gmlc_303.npz", stride=37)
    scenarios.load_prices(bidder, model)
    solver.solve(model)
    it, st = model.iterations, model.status
    bad = np.nonzero(st != 0)[0]
    o = np.argsort(-it)[:6]
    print(f"[{spec}] optimal {(st == 0).sum()} mean {it.mean():.0f} p99 {np.percentile(it, 99):.0f}; failing {bad.tolist()} jumps {model.jumps[bad].tolist()} w {model.primal_weight[bad].tolist()}; slowest {[(int(i), int(it[i]), int(model.jumps[i]), float('%.1e' % model.primal_weight[i])) for i in o]}", flush=True)
