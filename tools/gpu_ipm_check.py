"""GPU development tool: the interior-point form (csrc/dsp_ipm.hip) against the PDHG forms and HiGHS on price-taker batches.
    python tools/gpu_ipm_check.py T B [family]        (DSP_IPM_TRACE=<lane + 1> prints that lane's Newton iterations)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver

T = int(sys.argv[1]) if len(sys.argv) > 1 else 168
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fam = sys.argv[3] if len(sys.argv) > 3 else "wb"
ref_n = int(os.environ.get("IPM_CHECK_HIGHS", "4"))


def build(solver):
    if fam == "pem":
        return scenarios.pem_price_taker_batch(T, B, solver, inputs="rts303")
    if fam == "nuclear":
        return scenarios.nuclear_price_taker_batch(T, B, solver)
    return scenarios.price_taker_batch(T, B, solver, throughput="chain", family=os.environ.get("IPM_CHECK_FAMILY", "wide"))


out = {}
for name, env in (("ipm", "0"), ("pdhg", "1")):
    if name == "pdhg" and os.environ.get("IPM_CHECK_SKIP_PDHG") == "1":
        continue
    os.environ["DSP_NO_IPM"] = env
    solver = HipPdlpSolver(device=0, check_every=64, max_iter=2_000_000)
    handles, model = build(solver)
    t = time.time()
    solver.solve(model)
    wall = time.time() - t
    st = solver.last_stats
    print(f"{name}: T={T} B={B} n={model.lp.n} m={model.lp.m} form={st.stream_form} status={np.bincount(model.status, minlength=4).tolist()} "
          f"iterations mean {model.iterations.mean():.0f} max {model.iterations.max()} kernel {st.kernel_ms:.1f} ms wall {wall:.2f} s", flush=True)
    out[name] = (model.objective.copy(), model)
if "pdhg" in out:
    a, b = out["ipm"][0], out["pdhg"][0]
    print("max rel objective difference ipm vs pdhg", np.max(np.abs(a - b) / np.maximum(1, np.abs(b))))
if ref_n:
    import stream_lab as lab
    model = out["ipm"][1]
    lb, ub, rlo, rhi = model.scenario_bounds()
    errs = []
    for k in range(min(ref_n, B)):
        pick = lambda v: (v[k] if v.ndim == 2 else v).astype(float)
        P = dict(A=model.lp.csr(), c=model.c[k].astype(float), lb=pick(lb), ub=pick(ub), rlo=pick(rlo), rhi=pick(rhi), c0=float(model.c0[k]))
        ref = lab.highs(P)[0]
        errs.append(abs(model.objective[k] - ref) / max(1, abs(ref)))
    print("rel objective error vs HiGHS (first members)", ["%.1e" % e for e in errs])
    x = model.x
    print("x finite", np.isfinite(x).all(), "bound violation", float(np.max(np.maximum(pick(lb) - x[min(ref_n, B) - 1], 0))), float(np.max(np.maximum(x[min(ref_n, B) - 1] - pick(ub), 0))))
