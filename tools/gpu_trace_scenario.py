"""GPU development tool: KKT history of single scenarios (needs a -DDSP_KKT_TRACE build of the library: DSP_LIB=libdsp_trace.so).
    python tools/gpu_trace_scenario.py <workload> <scenario,scenario,...> [k=v options]  -> gpurun_out/trace_<workload>_<s>.npy"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wl = sys.argv[1]
ids = [int(a) for a in sys.argv[2].split(",")]
opts = {k: float(v) if "." in v or "e" in v else int(v) for k, v in (a.split("=") for a in sys.argv[3:])}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
for s in ids:
    path = os.path.join(ROOT, "gpurun_out", f"trace_{wl}_{s}.bin")
    os.environ["DSP_TRACE_SCENARIO"] = str(s)
    os.environ["DSP_TRACE_FILE"] = path
    solver = HipPdlpSolver(device=0, **opts)
    bidder, model = scenarios.make_batch(wl, 4096, solver)
    solver.solve(model)
    t = np.fromfile(path).reshape(4096, 12)
    t = t[t[:, 0] > 0]
    np.save(path.replace(".bin", ".npy"), t)
    os.remove(path)
    print(f"{wl} scenario {s}: iterations {model.iterations[s]} flags {model.flags[s]} status {model.status[s]} kkt tests {len(t)}")
    step = max(1, len(t) // 40)
    for row in t[::step]:
        print("   it %6d rp %.1e rd %.1e rg %.1e gap/lim %.1e yviol/lim %.1e dresx/lim %.1e w %.3g k %5d w_lo %.2g w_hi %.2g obj %.6f" % tuple(row))
