"""Development tool: time library build variants (dispatches_amd/libdsp_hip_*.so) on the same batch."""
import faulthandler, glob, os, sys, time
faulthandler.dump_traceback_later(int(os.environ.get("DSP_WATCHDOG", "150")), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dispatches_amd import hip_solver, scenarios

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wls = sys.argv[2].split(",") if len(sys.argv) > 2 else ["wind_battery_24h"]
ref = np.load("gpurun_out/dump.npz") if os.path.exists("gpurun_out/dump.npz") else None
for path in sorted(glob.glob(os.path.join(os.path.dirname(hip_solver.__file__), "libdsp_hip_*.so"))):
    hip_solver._lib = hip_solver.load_library(path)
    for wl in wls:
        solver = hip_solver.HipPdlpSolver(device=0)
        bidder, model = scenarios.make_batch(wl, B, solver)
        for rep in range(2):
            model.solve_handle = None if rep == 0 else model.solve_handle
            res = solver.solve(model)
        st = solver.last_stats
        it = model.iterations
        msg = (f"{os.path.basename(path)} {wl}: kernel {st.kernel_ms:.3f} ms  {B / st.kernel_ms * 1e3:.0f} scen/s  optimal {st.n_optimal}/{B} "
               f"iters mean {it.mean():.0f} med {np.median(it):.0f} p99 {np.percentile(it, 99):.0f} max {it.max()} jumps {model.jumps.sum()} "
               f"grid {st.grid_blocks}x{st.block_threads} lds {st.lds_bytes}  {it.sum() / st.kernel_ms / 1e3:.0f} M iter/s")
        if ref is not None and wl + "_obj" in ref and B == len(ref[wl + "_obj"]):
            ok = ref[wl + "_status"] == 0
            err = np.abs(model.objective - ref[wl + "_obj"]) / np.maximum(1, np.abs(ref[wl + "_obj"]))
            msg += f"  max rel obj diff vs v1 (converged) {err[ok].max():.2e}"
        print(msg, flush=True)
