"""Route B of INTEGRATION.md on the GPU: Pyomo-shaped scenario blocks (stand-ins of tests/test_pyomo_adapter.py) through
`HipPyomoSolver` -> `HipPdlpSolver` -> C ABI, against the oracle."""
import numpy as np
import pytest

from dispatches_amd.pyomo_adapter import HipPyomoSolver
from test_pyomo_adapter import CTYPES, ObjData, QuadExpr, _qp_bracket, build_tracking_model, generate_standard_repn

pytestmark = pytest.mark.gpu


def _blocks(rts309, B):
    rng = np.random.default_rng(7)
    cfs = [list(rts309["rt_cf"][(13 * k) % 8000:(13 * k) % 8000 + 4]) for k in range(B)]
    Ds = [list(np.round(rng.random(4) * 30.0, 2)) for _ in range(B)]
    return cfs, Ds, [build_tracking_model(cf, D)[0] for cf, D in zip(cfs, Ds)]


def test_pyomo_scenario_blocks_solve_as_one_gpu_batch(rts309):
    from oracle import dispatch_lp_oracle as orc
    cfs, Ds, blocks = _blocks(rts309, 64)
    solver = HipPyomoSolver(device=0, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    res = solver.solve(blocks)
    assert res.solver.termination_condition == "optimal"
    obj = solver.last_batch.objective
    ref = np.array([orc.wind_battery_track(4, cf, D)[0].solve()[1] for cf, D in zip(cfs, Ds)])
    assert np.abs(obj - ref).max() <= 1e-6 * np.maximum(1.0, np.abs(ref)).max()
    np.testing.assert_allclose(obj, ref, rtol=1e-6, atol=1e-6)
    under = [v.value for v in blocks[5].vars if v.name.startswith("under[")]
    assert all(u is not None and u >= -1e-6 for u in under)                      # solution loaded into the Vars


def test_pyomo_quadratic_objective_on_the_gpu(rts309):
    rho = 40.0
    cfs, Ds, blocks = _blocks(rts309, 8)
    for blk in blocks:
        G = [v for v in blk.vars if v.name.startswith("grid[")]
        O = [v for v in blk.vars if v.name.startswith("batt_out[")]
        quad = QuadExpr(blk.objs[0].expr, [])
        for t in range(1, 4):
            quad = quad + QuadExpr.square(1e-3 * G[t] + 1e-3 * O[t] - 1e-3 * G[t - 1] - 1e-3 * O[t - 1], rho)
        blk.objs[0] = ObjData(quad, sense=1)
    solver = HipPyomoSolver(device=0, ctypes=CTYPES, generate_standard_repn=generate_standard_repn, solver_hints={"geo_iters": 8})
    res = solver.solve(blocks)
    batch = solver.last_batch
    assert (batch.lp.row_compliance > 0).sum() == 3
    assert res.solver.termination_condition == "optimal", batch.status
    for i, view in enumerate(batch.views):
        br = _qp_bracket(view.lp)
        tol = 1e-6 * max(1.0, abs(br["upper"]))
        assert br["lower"] - tol <= batch.objective[i] <= br["upper"] + tol, (i, batch.objective[i], br["lower"], br["upper"])
