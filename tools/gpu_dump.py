"""Development tool: solve whole workloads on the GPU and dump per-scenario status / iterations / objectives
to gpurun_out/ for offline analysis (convergence stragglers, parity against the oracle on the CPU box)."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(int(os.environ.get('DSP_WATCHDOG', '120')), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

g.build()
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver

out = {}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(scenarios.WORKLOADS)
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v or "e" in v else int(v)
for wl in names:
    solver = HipPdlpSolver(device=0, **kw)
    bidder, model = scenarios.make_batch(wl, B, solver)
    t = time.time()
    res = solver.solve(model, tee=True)
    print(wl, "wall", time.time() - t, res, flush=True)
    it = model.iterations
    print(wl, "iters mean/med/p99/max", it.mean(), np.median(it), np.percentile(it, 99), it.max(),
          "not optimal:", np.nonzero(model.status != 0)[0][:20], flush=True)
    out[wl + "_status"] = model.status
    out[wl + "_iters"] = it
    out[wl + "_obj"] = model.objective
    out[wl + "_cx"] = model.objective - model.c0
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/dump.npz", **out)
