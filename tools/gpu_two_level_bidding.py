"""GPU dev tool (first call of the next round): the day-ahead bidding LPs with the two-level form of the battery's throughput accumulator
(MultiPeriodWindBattery(throughput_nodes=2), scenarios.EXPERIMENTAL_WORKLOADS) against the reference's form, on the oracle fixtures of the
base workloads - the LP is the same in the reference's variables, so objectives and setpoint faces are the fixtures' - with iteration counts
and kernel time of a lone 4096-batch.        python tools/gpu_two_level_bidding.py [24|48] [nodes]
A new sparsity: no ahead-of-time register-resident specialisation - the handle compiles one through hiprtc (dsp_stats::matreg says whether
it got it) or falls back to the LDS-matrix kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver

T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
base = f"wind_battery_{T}h"
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ref = np.load(os.path.join(GOLD, "oracle_objectives.npz"))[base]
sp = np.load(os.path.join(GOLD, "oracle_setpoints.npz"))
B = len(sp[f"{base}/P_T_lo"])
for tag, kw in ((base, {}), (f"{base} two-level K={nodes}", dict(throughput_nodes=nodes))):
    solver = HipPdlpSolver(device=0)
    bidder, model = scenarios.wind_battery_batch(B=B, T=T, solver=solver, **kw)
    scenarios.load_prices(bidder, model)
    solver.solve(model)                                             # warm-up (handle, code objects, run-time compilation)
    t = time.time(); solver.solve(model); wall = time.time() - t
    st = solver.last_stats
    err = np.abs(model.objective - ref[:B]) / np.maximum(1.0, np.abs(ref[:B]))
    P_T = model.expression_values("P_T")
    lo, width = sp[f"{base}/P_T_lo"], sp[f"{base}/P_T_width"]
    p_max = bidder.bidding_model_object.model_data.p_max
    out = np.maximum(lo - P_T, P_T - (lo + width)).max()
    print(f"{tag}: n={model.lp.n} m={model.lp.m} nnz={model.lp.nnz} optimal {(model.status == 0).sum()}/{B} flagged {int(((model.flags & 1) != 0).sum())} "
          f"iterations mean {model.iterations.mean():.0f} max {model.iterations.max()} kernel {st.kernel_ms:.2f} ms (wall {1e3 * wall:.1f} ms) "
          f"matreg={st.matreg} max objective error {err.max():.2e} worst P_T distance from its optimal-face range {out:.2e} MW (p_max {p_max})", flush=True)
