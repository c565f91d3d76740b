"""GPU dev tool: streaming PDLP on the price-taker family.  python tools/gpu_stream.py T B [max_iter] [check_every]
(STREAM_FAMILY=pem: the wind + battery + PEM family, chain accumulator; nuclear: the nuclear enumeration, B design points)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
T = int(sys.argv[1]); B = int(sys.argv[2]); mi = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
ce = int(sys.argv[4]) if len(sys.argv) > 4 else 64
solver = HipPdlpSolver(device=0, check_every=ce, max_iter=mi)
t = time.time()
if os.environ.get("STREAM_FAMILY") == "pem":
    handles, model = scenarios.pem_price_taker_batch(T, B, solver, inputs="rts303", throughput=os.environ.get('STREAM_THROUGHPUT', 'chain'))
elif os.environ.get("STREAM_FAMILY") == "nuclear":
    handles, model = scenarios.nuclear_price_taker_batch(T, B, solver)
else:
    handles, model = scenarios.price_taker_batch(T, B, solver, throughput=os.environ.get('STREAM_THROUGHPUT', 'two_level'))
tb = time.time() - t
for rep in range(int(os.environ.get("STREAM_REPS", 1))):
    t = time.time(); solver.solve(model); ts = time.time() - t
    st = solver.last_stats
    its = model.iterations
    per_it_us = 1e3 * st.kernel_ms / max(1, its.max())
    gbs = st.stream_bytes_per_iteration * its.sum() / (st.kernel_ms * 1e-3) / 1e9
    print(f"T={T} B={B} n={model.lp.n} m={model.lp.m} build {tb:.1f}s solve wall {ts:.2f}s kernel {st.kernel_ms:.1f} ms status {np.bincount(model.status, minlength=5).tolist()} "
          f"iters min/mean/max {its.min()}/{its.mean():.0f}/{its.max()} -> {per_it_us:.2f} us per batch-iteration, algorithmic {gbs:.0f} GB/s", flush=True)
fx_path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "oracle_price_taker.npz")
fx = np.load(fx_path)
if f"T{T}/obj" in fx.files:
    ref = fx[f"T{T}/obj"][:B]
    print("obj err", np.abs(model.objective - ref[:B]) / np.maximum(1, np.abs(ref[:B])))
if "battery_system_capacity" in handles:
    print("objective", model.objective[:8], "battery MW", model.x[:8, handles["battery_system_capacity"].index] * 1e-3)
