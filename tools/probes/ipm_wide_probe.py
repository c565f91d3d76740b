import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
solver = HipPdlpSolver(device=0, check_every=64, max_iter=max_iter, recertify=0)
handles, model = scenarios.price_taker_batch(8736, B, solver, throughput="chain", family="wide")
for rep in range(2):
    t = time.time(); solver.solve(model); wall = time.time() - t
    st = solver.last_stats
    print(f"rep {rep}: wall {wall:.2f} s kernel {st.kernel_ms:.0f} ms form {st.stream_form} ipm_solved {st.ipm_solved} status {np.bincount(model.status, minlength=3).tolist()}")
bad = np.nonzero(model.iterations > 300)[0]
print("handed to the PDHG form:", [(int(k), model.family[k], int(model.iterations[k]), int(model.status[k])) for k in bad])
it = model.iterations[model.iterations <= 300]
print("newton iterations of the rest: min", it.min(), "mean", round(it.mean(), 1), "max", it.max(), "hist", np.histogram(it, bins=[0, 40, 60, 80, 100, 120, 140, 160, 200, 251])[0].tolist())
fx = np.load("tests/golden/oracle_price_taker.npz")
ks = np.concatenate([np.arange(16), fx["T8736w/k"]]); ref = np.concatenate([fx["T8736/obj"], fx["T8736w/obj"]]); keep = ks < B
err = np.abs(model.objective[ks[keep]] - ref[keep]) / np.maximum(1, np.abs(ref[keep]))
print("max rel objective error vs fixture", err.max(), "at member", int(ks[keep][np.argmax(err)]))
