"""GPU dev tool: the price-taker family with the throughput chain as linked equalities vs in the hierarchical basis, on the
streaming path as it is (the hierarchical LP is not banded: two-launch form, its wide hats are long vectors).
    python tools/gpu_hier_probe.py T B"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
T, B = int(sys.argv[1]), int(sys.argv[2])
obj = {}
for thr in ("chain", "hier"):
    solver = HipPdlpSolver(device=0, check_every=64, max_iter=600000)
    t = time.time(); handles, model = scenarios.price_taker_batch(T, B, solver, throughput=thr); tb = time.time() - t
    t = time.time(); solver.solve(model); ts = time.time() - t
    st, its = solver.last_stats, model.iterations
    obj[thr] = model.objective.copy()
    print(f"{thr}: T={T} B={B} n={model.lp.n} m={model.lp.m} nnz={model.lp.nnz} build {tb:.1f}s solve {ts:.2f}s kernel {st.kernel_ms:.1f} ms status {np.bincount(model.status, minlength=5).tolist()} "
          f"iterations min/mean/max {its.min()}/{its.mean():.0f}/{its.max()} -> {1e3 * st.kernel_ms / max(1, its.max()):.2f} us per batch-iteration", flush=True)
print("max relative objective difference", float(np.max(np.abs(obj["chain"] - obj["hier"]) / np.maximum(1, np.abs(obj["chain"])))))
