"""GPU tests of BASELINE config 5: the stochastic bidder's day-ahead problems with a quadratic ramp cost (convex QP, soft
rows with a dual compliance: include/dsp_hip.h dsp_batch::row_compliance) against the certified brackets of the QP oracle
(tests/golden/oracle_qp.npz, oracle/qp_cutting_plane.py), and the float32 side of the tolerance sweep.

Parity bar of the float64 path, per scenario: status optimal, objective within 1e-6 max(1, |obj|) of the bracket
[lower, upper] for every scenario the solver does not FLAG (DSP_FLAG_OBJ_WAIVED: the iteration stalled twice on its rounding
floor and terminated on the eps_rel tests alone - near-zero objectives that are the difference of terms ~1e6 times larger;
at most 16 scenarios in 4096 here, and those stay within 2e-5 = twice the accepted bound 10 eps_obj (1 + |obj|) at |obj| ~ 1); setpoints: the hour-to-hour ramps of the delivered power, the quantities the quadratic term makes unique,
within the strong-convexity bound  |M x - M x*|^2 <= 2 (f(x) - f*) / rho  of the objective tolerance."""
import os

import numpy as np
import pytest

gpu = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RHO = {"wind_battery_24h_qp001": 0.01, "wind_battery_24h_qp01": 0.1, "wind_battery_24h_qp1": 1.0}


def _solver(**kw):
    from dispatches_amd.hip_solver import HipPdlpSolver
    return HipPdlpSolver(device=0, **kw)


def _bracket_error(obj, lo, up):
    """distance of obj from the bracket [lo, up], relative to max(1, |up|)"""
    return np.maximum(np.maximum(lo - obj, obj - up), 0.0) / np.maximum(1.0, np.abs(up))


@gpu
@pytest.mark.parametrize("workload", sorted(RHO))
def test_qp_batch_parity_vs_oracle_brackets(workload):
    from dispatches_amd import scenarios
    fx = np.load(os.path.join(GOLD, "oracle_qp.npz"))
    B = 4096
    solver = _solver()
    bidder, model = scenarios.make_batch(workload, B, solver)
    res = solver.solve(model)
    st = solver.last_stats
    assert st.quadratic == 1 and st.matreg == 1 and st.simplex == 0, (st.quadratic, st.matreg)
    assert (model.status == 0).all(), (np.bincount(model.status), res)
    up, lo = fx[f"{workload}/upper"][:B], fx[f"{workload}/lower"][:B]
    err = _bracket_error(model.objective, lo, up)
    waived = (model.flags & 1) != 0
    assert (err[~waived] < 1e-6).all(), (float(err[~waived].max()), int(np.nonzero(~waived)[0][err[~waived].argmax()]))
    # whatever is still flagged after the solver's re-solves is reported as UNCERTIFIED (never a bid: workflow/bidder.py); with the
    # variable scaling on none is left on these workloads (profiles/r30a_recertify.log) - round 2 allowed 16 of them up to 2e-5
    assert waived.sum() <= 2 and model.uncertified[waived].all() and not model.uncertified[~waived].any(), int(waived.sum())
    print(f"{workload}: max bracket error {err.max():.2e}, flagged (objective tests waived): {int(waived.sum())} of {B}; "
          f"iterations mean {model.iterations.mean():.0f} max {model.iterations.max()}; kernel {st.kernel_ms:.2f} ms")
    # the reported objective is the objective of the returned point (linear part + soft rows)
    for k in (0, 1, 2, 999):
        assert model.lp.objective(model.x[k], c=model.c[k], c0=model.c0[k]) == pytest.approx(model.objective[k], rel=1e-9, abs=1e-6)
    # ramps of the delivered power
    rho = RHO[workload]
    P_T = model.expression_values("P_T")
    ramp, ramp_ref = np.diff(P_T, axis=1), np.diff(fx[f"{workload}/P_T"][:B].astype(float), axis=1)
    bound = np.sqrt(2.0 * 2.1e-6 * np.maximum(1.0, np.abs(up)) / rho)[:, None] + 1e-3        # + float32 storage of the fixture
    dev = np.sqrt(((ramp - ramp_ref) ** 2).sum(axis=1, keepdims=True))
    assert (dev <= bound).all(), (float((dev / bound).max()), int((dev / bound).argmax()))
    # the duals of the soft rows are the marginal ramp costs: y_i = -rho (a_i.x - b_i)
    soft = np.nonzero(model.lp.row_compliance)[0]
    A = model.lp.csr()
    k = 7
    r = (A @ model.x[k])[soft]
    assert np.allclose(model.y[k][soft], -rho * r, atol=1e-5 * max(1.0, np.abs(rho * r).max()))


@gpu
def test_generic_qp_kernel_agrees_with_the_register_resident_one():
    from dispatches_amd import scenarios
    fx = np.load(os.path.join(GOLD, "oracle_qp.npz"))
    B = 256
    solver = _solver(no_matreg=1)
    bidder, model = scenarios.make_batch("wind_battery_24h_qp01", B, solver)
    solver.solve(model)
    assert solver.last_stats.matreg == 0 and solver.last_stats.quadratic == 1
    assert (model.status == 0).all()
    err = _bracket_error(model.objective, fx["wind_battery_24h_qp01/lower"][:B], fx["wind_battery_24h_qp01/upper"][:B])
    assert err.max() < 1e-6, float(err.max())


@gpu
def test_zero_compliance_is_the_lp():
    """row_compliance = 0 everywhere goes through the QP instantiation and must reproduce the LP fixture."""
    import torch
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import DeviceLP, default_options
    B = 512
    solver = _solver()
    bidder, model = scenarios.make_batch("wind_battery_24h", B, solver)
    lb, ub, rlo, rhi = model.scenario_bounds()
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    dlp = DeviceLP(model.lp, 0, default_options())
    # (obj_offset: the solver scales its objective-accuracy test with the TRUE objective c.x + c0; without the model constant
    # c.x alone is ~500x larger here and the answer would only be good to ~1e-5 of the true objective)
    out = dlp.solve(B, t(model.c), t(lb), t(ub), t(rlo), t(rhi), obj_offset=t(model.c0),
                    row_compliance=torch.zeros(model.lp.m, dtype=torch.float64, device=dev))
    assert out["stats"].quadratic == 1
    ref = np.load(os.path.join(GOLD, "oracle_objectives.npz"))["wind_battery_24h"][:B]
    obj = out["obj"].cpu().numpy() + model.c0
    assert (out["status"].cpu().numpy() == 0).all()
    assert (np.abs(obj - ref) / np.maximum(1.0, np.abs(ref))).max() < 1e-6


@gpu
def test_soft_row_needs_one_finite_target():
    import torch
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import DeviceLP, default_options
    solver = _solver()
    bidder, model = scenarios.make_batch("wind_battery_24h_qp01", 4, solver)
    lb, ub, rlo, rhi = model.scenario_bounds()
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    rhi = np.array(rhi, float)
    soft = np.nonzero(model.lp.row_compliance)[0]
    rhi[soft[0]] = 1.0                     # lo = 0 < hi: not a target
    dlp = DeviceLP(model.lp, 0, default_options())
    out = dlp.solve(4, t(model.c), t(lb), t(ub), t(rlo), t(rhi), row_compliance=t(model.lp.row_compliance))
    assert (out["status"].cpu().numpy() == 2).all()


@gpu
def test_float32_iterates_stop_short_of_the_contract():
    """The float32 side of the tolerance sweep (dsp_options::precision = 1): it terminates at loose tolerances, its returned
    points are worth what float64 says they are worth, and it cannot reach the 1e-6 objective contract - which is the finding."""
    from dispatches_amd import scenarios
    fx = np.load(os.path.join(GOLD, "oracle_qp.npz"))
    B = 1024
    up = fx["wind_battery_24h_qp01/upper"][:B]
    solver = _solver(precision=1, eps_rel=1e-4, eps_obj=0.0, max_iter=20000)
    bidder, model = scenarios.make_batch("wind_battery_24h_qp01", B, solver)
    solver.solve(model)
    st = solver.last_stats
    assert st.precision == 1 and st.quadratic == 1
    assert (model.status == 0).mean() > 0.95, np.bincount(model.status)
    for k in (0, 5, 77):
        assert model.lp.objective(model.x[k], c=model.c[k], c0=model.c0[k]) == pytest.approx(model.objective[k], rel=1e-7, abs=1e-3)
    err32 = np.abs(model.objective - up) / np.maximum(1.0, np.abs(up))
    assert np.median(err32) < 0.2                                  # a sane answer ...
    solver64 = _solver(eps_rel=1e-4, eps_obj=0.0)
    bidder64, model64 = scenarios.make_batch("wind_battery_24h_qp01", B, solver64)
    solver64.solve(model64)
    assert (model64.status == 0).all()
    # ... at a tolerance where float64 is no better (the relative KKT test at 1e-4 is weak on these problems) ...
    err64 = np.abs(model64.objective - up) / np.maximum(1.0, np.abs(up))
    assert np.median(err64) < 0.2
    # ... but at the contract tolerance float32 does not terminate
    solver_t = _solver(precision=1, max_iter=20000)
    bidder_t, model_t = scenarios.make_batch("wind_battery_24h_qp01", 256, solver_t)
    solver_t.solve(model_t)
    assert (model_t.status == 0).mean() < 0.5


@gpu
def test_bidder_with_ramp_cost_end_to_end():
    from dispatches_amd import scenarios
    solver = _solver()
    bidder, model = scenarios.make_batch("wind_battery_24h_qp1", 64, solver)
    bids = bidder.compute_day_ahead_bids(date="2020-01-02", hour=0)
    assert len(bids) == 24 and not bidder.failed_scenarios
    # a ramp cost flattens the delivered-power profile compared with the LP
    bidder0, model0 = scenarios.make_batch("wind_battery_24h", 64, solver)
    bidder0.compute_day_ahead_bids(date="2020-01-02", hour=0)
    r1 = np.abs(np.diff(bidder.day_ahead_model.expression_values("P_T"), axis=1)).sum()
    r0 = np.abs(np.diff(bidder0.day_ahead_model.expression_values("P_T"), axis=1)).sum()
    assert r1 < r0
