"""GPU tests of the infeasibility / unboundedness certificates of the PDLP kernels (dsp_options::eps_infeasible, ABI 9).

Reference behaviour: the solver hands back a termination condition and the callers act on it
(dispatches/case_studies/renewables_case/solar_battery_hydrogen.py:451-458).  A certificate is a proof whatever the iterate, so the
tests need no oracle objective: HiGHS is asked to confirm the verdict on the same data, and the feasible scenarios of the same
batch must come back exactly as without the edits."""
import numpy as np
import pytest

gpu = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pdhg_forms(monkeypatch):
    """These tests pin the PDHG forms of the HBM-resident path (and their certificates); the interior-point form that takes time-banded
    LPs first since round 5 (csrc/dsp_ipm.hip) has its own tests (tests/test_hip_ipm.py).  Read at every dsp_create."""
    monkeypatch.setenv("DSP_NO_IPM", "1")


def _solver(**kw):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from dispatches_amd.hip_solver import HipPdlpSolver
    return HipPdlpSolver(device=0, **kw)


def _own_bounds(model):
    """Give the batch model per-scenario copies of all four bound arrays (so that single scenarios can be edited)."""
    B = model.n_scenario
    lb, ub, rlo, rhi = model.scenario_bounds()
    full = lambda a: np.broadcast_to(a, (B, a.shape[-1])).copy()
    model.lb, model.ub, model.rlo, model.rhi = full(lb), full(ub), full(rlo), full(rhi)
    model.c = model.c.copy()
    model.x = model.y = None


def _highs_status(model, k):
    """scipy/HiGHS verdict on scenario k of the batch as handed to the solver: 0 optimal, 2 infeasible, 3 unbounded."""
    from scipy.optimize import linprog
    import scipy.sparse as sp
    A = model.lp.csr()
    lb, ub, rlo, rhi = (a[k] for a in (model.lb, model.ub, model.rlo, model.rhi))
    eq = np.isfinite(rlo) & (rlo == rhi)
    up, dn = np.isfinite(rhi) & ~eq, np.isfinite(rlo) & ~eq
    Aub = sp.vstack([A[up], -A[dn]]).tocsr()
    res = linprog(model.c[k], A_ub=Aub if Aub.shape[0] else None, b_ub=np.concatenate([rhi[up], -rlo[dn]]) if Aub.shape[0] else None,
                  A_eq=A[eq] if eq.any() else None, b_eq=rhi[eq] if eq.any() else None, bounds=np.stack([lb, ub], 1), method="highs")
    return {0: 0, 2: 2, 3: 3}.get(res.status, -1)


@gpu
def test_infeasible_bidding_lps_are_certified_primal_infeasible():
    """A 24-h wind + battery bidding LP whose initial state of charge is 10 x the battery's energy capacity has no feasible point
    (the state-of-charge bound of hour 0 cannot be met at the discharge limit): status 2 within 5 k iterations instead of the
    iteration limit of 200 k; the feasible scenarios of the same batch are untouched."""
    from dispatches_amd import scenarios
    solver = _solver()
    bidder, model = scenarios.make_batch("wind_battery_24h", 8, solver)
    solver.solve(model)
    ref = model.objective.copy()
    assert (model.status == 0).all()
    _own_bounds(model)
    j = model.lp.col_names.index("battery.initial_state_of_charge")
    bad = [1, 4, 6]
    model.lb[bad, j] = model.ub[bad, j] = 1.0e6                     # kWh; the battery holds 1e5
    res = solver.solve(model)
    want = np.zeros(8, int); want[bad] = 2
    assert model.status.tolist() == want.tolist(), (model.status, model.iterations)
    assert (model.iterations[bad] <= 5000).all(), model.iterations
    assert res.solver.termination_condition == "infeasible"
    keep = [k for k in range(8) if k not in bad]
    np.testing.assert_allclose(model.objective[keep], ref[keep], rtol=1e-9)
    for k in bad[:2]:
        assert _highs_status(model, k) == 2
    # the bidder never turns such a scenario into a bid
    assert not getattr(model, "uncertified", np.zeros(8, bool))[bad].any()


@gpu
def test_unbounded_bidding_lps_are_certified_dual_infeasible():
    """An LP with a free column that pays: the day-ahead offer of hour 5 of the nuclear bidding LP without its lower bound and with
    a positive cost (the under-bid row u_5 >= pda_5 - P_T[5] holds for every pda_5 -> -inf): status 3 within 5 k iterations."""
    from dispatches_amd import scenarios
    solver = _solver()
    bidder, model = scenarios.make_batch("nuclear_24h", 8, solver)
    solver.solve(model)
    ref = model.objective.copy()
    _own_bounds(model)
    j = model.lp.col_names.index("day_ahead_power[5]")
    bad = [0, 3, 5, 7]
    model.lb[bad, j] = -np.inf
    model.c[bad, j] = 3.0
    res = solver.solve(model)
    want = np.zeros(8, int); want[bad] = 3
    assert model.status.tolist() == want.tolist(), (model.status, model.iterations)
    assert (model.iterations[bad] <= 5000).all(), model.iterations
    assert res.solver.termination_condition == "unbounded"
    keep = [k for k in range(8) if k not in bad]
    np.testing.assert_allclose(model.objective[keep], ref[keep], rtol=1e-9)
    assert _highs_status(model, bad[0]) in (2, 3)                   # HiGHS reports "infeasible or unbounded" as either


@gpu
def test_certificates_off_means_iteration_limit():
    """eps_infeasible = 0 restores the old behaviour (the scenario runs into max_iter): the certificates are what ends it."""
    from dispatches_amd import scenarios
    solver = _solver(eps_infeasible=0.0, max_iter=3000)
    bidder, model = scenarios.make_batch("wind_battery_24h", 2, solver)
    _own_bounds(model)
    j = model.lp.col_names.index("battery.initial_state_of_charge")
    model.lb[1, j] = model.ub[1, j] = 1.0e6
    solver.solve(model)
    assert model.status.tolist() == [0, 1] and model.iterations[1] == 3000


# (no feasible scenario is ever called infeasible: tests/test_hip_parity.py::test_full_batch_objective_parity_vs_oracle_fixture and
#  tests/test_hip_batch_parity.py assert status 0 for all 4096 scenarios of every bench workload with the certificates on)


STREAM_ENV = ("DSP_STREAM_NO_LANE", "DSP_STREAM_NO_FUSED", "DSP_STREAM_NO_BLOCK", "DSP_LANE_MIN_B")


@gpu
@pytest.mark.parametrize("form,T,B,env,stream_form", [
    ("lane", 336, 40, {}, 3),                                                        # lane per scenario (batches of 32 and more)
    ("tile", 336, 6, {"DSP_STREAM_NO_LANE": "1"}, 2),                                 # workgroup per tile
    ("two_launch", 336, 6, {"DSP_STREAM_NO_LANE": "1", "DSP_STREAM_NO_FUSED": "1"}, 1),
    ("block", 96, 5, {}, 4)])                                                         # the whole solve in one launch, state in LDS
def test_streaming_forms_certify_an_infeasible_member(monkeypatch, form, T, B, env, stream_form):
    """HBM-resident path (dsp_stream.hip / dsp_stream_lane.hip): one member of a price-taker design batch gets an impossible power
    balance (a period whose splitter row asks for 1e9 kW more than the wind plant can ever deliver) - every form of the streaming
    iteration reports it primal infeasible (status 2) within 5 k iterations through the certificate sequence (stream_certify /
    the block form's in-kernel evaluation) while the other members reach their optima as without the edit."""
    from dispatches_amd import scenarios
    for k in STREAM_ENV:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    solver = _solver(check_every=64)
    handles, model = scenarios.price_taker_batch(T, B, solver)
    solver.solve(model)
    st = solver.last_stats
    assert st.streaming == 1 and st.stream_form == stream_form and (model.status == 0).all(), (form, st.stream_form, model.status)
    ref = model.objective.copy()
    _own_bounds(model)
    i = model.lp.row_names.index("splitter.sum_split[5]")
    bad = 3
    model.rlo[bad, i] = model.rhi[bad, i] = 1.0e9
    res = solver.solve(model)
    want = np.zeros(B, int); want[bad] = 2
    assert solver.last_stats.stream_form == stream_form
    assert model.status.tolist() == want.tolist(), (form, model.status, model.iterations)
    assert model.iterations[bad] <= 5000, model.iterations
    assert res.solver.termination_condition == "infeasible"
    keep = [k for k in range(B) if k != bad]
    np.testing.assert_allclose(model.objective[keep], ref[keep], rtol=1e-6, atol=1e-6)
    assert _highs_status(model, bad) == 2


@gpu
def test_streaming_path_certifies_an_unbounded_member():
    """The same batch with one member's power balance of period 5 switched off (a free row): its grid sale G_5 is paid and nothing
    limits it any more - unbounded.  Status 3 from the recession-direction test of the certificate sequence."""
    from dispatches_amd import scenarios
    solver = _solver(check_every=64, max_iter=60_000)
    handles, model = scenarios.price_taker_batch(336, 40, solver)
    _own_bounds(model)
    i = model.lp.row_names.index("splitter.sum_split[5]")
    j = model.lp.col_names.index("splitter.grid_elec[5]")
    bad = 11
    assert model.c[bad, j] < 0                                       # selling at a positive price: the direction pays
    model.rlo[bad, i], model.rhi[bad, i] = -np.inf, np.inf
    res = solver.solve(model)
    assert model.status[bad] == 3, (model.status, model.iterations)
    assert (np.delete(model.status, bad) == 0).all()
    assert res.solver.termination_condition == "unbounded"
    assert _highs_status(model, bad) in (2, 3)


@gpu
def test_infeasible_qp_scenario_and_the_bidder_around_it():
    """The ramp-cost QP instantiation (soft rows: no multiplier ray on them) certifies the same impossible initial charge, and the
    Bidder built on top never turns such a scenario into a bid: it is listed under its status in `failed_scenarios`, the hourly curves
    come from the other scenarios (exactly the curves of a batch without it), and `strict=True` raises instead."""
    import warnings
    from dispatches_amd import scenarios
    solver = _solver()
    bidder, model = scenarios.make_batch("wind_battery_24h_qp01", 6, solver)
    solver.solve(model)
    assert (model.status == 0).all() and solver.last_stats.quadratic == 1
    ref = model.objective.copy()
    _own_bounds(model)
    j = model.lp.col_names.index("battery.initial_state_of_charge")
    model.lb[2, j] = model.ub[2, j] = 1.0e6
    solver.solve(model)
    assert model.status.tolist() == [0, 0, 2, 0, 0, 0] and model.iterations[2] <= 5000, (model.status, model.iterations)
    np.testing.assert_allclose(np.delete(model.objective, 2), np.delete(ref, 2), rtol=1e-9)
    # the Bidder: one infeasible price scenario among eight
    bidder, model = scenarios.wind_battery_batch(8, 24, solver)
    full = bidder.compute_day_ahead_bids("2020-01-02", 0)
    lb, ub, rlo, rhi = model.scenario_bounds()
    model.lb, model.ub = np.broadcast_to(lb, (8, model.lp.n)).copy(), np.broadcast_to(ub, (8, model.lp.n)).copy()
    j = model.lp.col_names.index("battery.initial_state_of_charge")
    model.lb[5, j] = model.ub[5, j] = 1.0e6
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        bids = bidder.compute_day_ahead_bids("2020-01-02", 0)
    assert any("did not reach optimality" in str(x.message) for x in w)
    assert bidder.failed_scenarios[("2020-01-02", 0, "Day-ahead")] == {5: 2}
    assert model.status[5] == 2 and (np.delete(model.status, 5) == 0).all()
    gen = bidder.generator
    for t in bids:                                               # every point of every curve is offered by a feasible scenario
        pts = {p for p, _ in bids[t][gen]["p_cost"]}
        assert pts <= {p for p, _ in full[t][gen]["p_cost"]} | {0.0}, t
    bidder.strict = True
    with pytest.raises(RuntimeError):
        bidder.compute_day_ahead_bids("2020-01-02", 0)
