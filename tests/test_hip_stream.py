"""GPU tests of the HBM-resident (streaming) PDLP: the long-horizon price-taker design LPs of the reference
(wind_battery_LMP.py:172-269) at a one-week horizon (n = 1011, m = 1010: beyond the register/LDS-resident kernels)
against the independent oracle (un-reduced LP + HiGHS, tests/golden/oracle_price_taker.npz)."""
import os

import numpy as np
import pytest

gpu = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pdhg_forms(monkeypatch):
    """These tests pin the PDHG forms of the HBM-resident path (and their certificates); the interior-point form that takes time-banded
    LPs first since round 5 (csrc/dsp_ipm.hip) has its own tests (tests/test_hip_ipm.py).  Read at every dsp_create."""
    monkeypatch.setenv("DSP_NO_IPM", "1")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@gpu
def test_price_taker_family_matches_oracle_and_is_reproducible():
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    from tests.test_hip_parity import _kkt_certificate
    fx = np.load(os.path.join(GOLD, "oracle_price_taker.npz"))
    T, B = 168, 8
    solver = HipPdlpSolver(device=0, check_every=64)
    handles, model = scenarios.price_taker_batch(T, B, solver)
    assert model.lp.n == 6 * T + 3 and model.lp.m == 6 * T + 2
    solver.solve(model, tee=True)
    assert solver.last_stats.streaming == 1
    assert (model.status == 0).all(), (np.bincount(model.status), model.iterations)
    ref = fx["T168/obj"][:B]
    err = np.abs(model.objective - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() < 1e-6, (err, model.iterations)
    rp, rd, rg = _kkt_certificate(model)
    assert max(rp.max(), rd.max()) < 5e-9 and rg.max() < 1.05
    # the design decision itself: optimal battery size [MW] (unique whenever a battery is built)
    batt = model.x[:, handles["battery_system_capacity"].index] * 1e-3
    np.testing.assert_allclose(batt, fx["T168/batt_mw"][:B], rtol=1e-3, atol=0.5)
    # ordered two-stage reductions: a second solve reproduces the first bit for bit
    obj1, it1 = model.objective.copy(), model.iterations.copy()
    solver.solve(model)
    assert np.array_equal(model.objective, obj1) and np.array_equal(model.iterations, it1)


@gpu
def test_one_launch_forms_reproduce_the_two_launch_form(monkeypatch):
    """Two weeks of the price-taker family (n = 2019: beyond the block-resident form, so the launch-per-step forms run) in every form
    of the streaming iteration: the lane-per-scenario form (round 4, the default: scenario-minor storage, one lane per scenario
    walking a tile of rows and columns alone, rings in LDS, scalar matrix loads; default tiling and 12-row tiles), the round-3
    workgroup-per-tile form (k_fused_pre / k_fused, DSP_STREAM_NO_LANE=1) and the two-launch form all three replace - same
    termination, same objectives to rounding, iteration counts within a check period or two, every form against the oracle; with the
    reference's chain and with the two-level accumulator (4 long columns)."""
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    from oracle import dispatch_lp_oracle as orc
    T, B = 336, 6
    cf, lmp = scenarios.price_taker_inputs(T)
    ref = []
    for bf, lm in scenarios.PRICE_TAKER_FAMILY[:B]:
        P, _ = orc.wind_battery_price_taker(T, cf, lmp * lm, batt_cap_factor=bf)
        ref.append(P.solve(tight=True)[1])
    ref = np.array(ref)
    keys = ("DSP_STREAM_NO_LANE", "DSP_STREAM_NO_FUSED", "DSP_FUSED_V", "DSP_FUSED_XCD", "DSP_FUSED_RB", "DSP_FUSED_DEFER", "DSP_LANE_ROWS", "DSP_LANE_MIN_B")
    for thr in ("chain", "two_level"):
        out = {}
        # (DSP_LANE_MIN_B: batches below 32 scenarios run the round-3 form by default)
        for form, extra in (("lane", {"DSP_LANE_MIN_B": "1"}), ("lane_small_tiles", {"DSP_LANE_MIN_B": "1", "DSP_LANE_ROWS": "12"}),
                            ("lane_large_tiles", {"DSP_LANE_MIN_B": "1", "DSP_LANE_ROWS": "400"}),
                            ("fused", {"DSP_STREAM_NO_LANE": "1"}), ("fused_staged", {"DSP_STREAM_NO_LANE": "1", "DSP_FUSED_V": "1"}),
                            ("fused_small_tiles", {"DSP_STREAM_NO_LANE": "1", "DSP_FUSED_RB": "64"}),
                            ("two_launch", {"DSP_STREAM_NO_LANE": "1", "DSP_STREAM_NO_FUSED": "1"})):
            for k in keys:
                monkeypatch.delenv(k, raising=False)
            for k, v in extra.items():
                monkeypatch.setenv(k, v)
            solver = HipPdlpSolver(device=0, check_every=64)
            handles, model = scenarios.price_taker_batch(T, B, solver, throughput=thr)
            solver.solve(model, tee=True)
            st = solver.last_stats
            assert st.streaming == 1 and (model.status == 0).all(), (thr, form, model.status, model.iterations)
            n, m = model.lp.n, model.lp.m
            # algorithmic bytes per scenario-iteration: 4 n + 3 m doubles in one launch (the family shares its bounds), 8 n + 6 m in two
            assert st.stream_bytes_per_iteration == (8 * (8 * n + 6 * m) if form == "two_launch" else 8 * (4 * n + 3 * m)), (thr, form)
            err = np.abs(model.objective - ref) / np.maximum(1.0, np.abs(ref))
            assert err.max() < 1e-6, (thr, form, err)
            out[form] = (model.objective.copy(), model.iterations.copy())
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for form in out:
            assert np.allclose(out[form][0], out["two_launch"][0], rtol=1e-7, atol=1e-7), (thr, form)
            assert (np.abs(out[form][1] - out["two_launch"][1]) <= 0.05 * out["two_launch"][1] + 128).all(), (thr, form, out[form][1], out["two_launch"][1])
        # the tiling changes the summation order of the long columns' A^T y only
        assert np.allclose(out["lane_small_tiles"][0], out["lane"][0], rtol=1e-9, atol=1e-9)
        assert np.allclose(out["lane_large_tiles"][0], out["lane"][0], rtol=1e-9, atol=1e-9)
        assert np.allclose(out["fused_staged"][0], out["fused"][0], rtol=1e-9, atol=1e-9)


@gpu
def test_lane_form_with_per_scenario_bounds_and_with_soft_rows(monkeypatch):
    """The two instantiation families of the lane form the price-taker families do not reach on their own: (a) bounds that differ
    per scenario on SHORT columns (a grid-connection limit per member: every bound is then read per lane, 6 n + 5 m doubles per
    scenario-iteration; the family's own batches share theirs) and (b) soft rows (convex QP: a compliance on every 5th balance
    row) - each against the two-launch form of the same solve: same termination, same objectives, the lane form forced onto the
    small batch with DSP_LANE_MIN_B."""
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    T, B = 336, 6
    keys = ("DSP_STREAM_NO_LANE", "DSP_STREAM_NO_FUSED", "DSP_LANE_MIN_B")

    def solve(form, mutate):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in ({"DSP_LANE_MIN_B": "1"} if form == "lane" else {"DSP_STREAM_NO_LANE": "1", "DSP_STREAM_NO_FUSED": "1"}).items():
            monkeypatch.setenv(k, v)
        solver = HipPdlpSolver(device=0, check_every=64, max_iter=400_000)
        handles, model = scenarios.price_taker_batch(T, B, solver)
        mutate(model)
        solver.solve(model, tee=True)
        st = solver.last_stats
        assert st.streaming == 1 and (model.status == 0).all(), (form, model.status, model.iterations)
        return model.objective.copy(), model.iterations.copy(), int(st.stream_bytes_per_iteration), int(st.stream_form), int(st.quadratic), model

    def grid_limits(model):
        lb, ub, _, _ = model.block.current_bounds()
        model.lb, model.ub = np.tile(lb, (B, 1)), np.tile(ub, (B, 1))
        cols = [j for j, name in enumerate(model.lp.col_names) if name.startswith("splitter.grid_elec[")]
        for k in range(B):
            model.ub[k, cols] = 847.0e3 * (0.55 + 0.08 * k)             # kW: binding for the smaller ones

    def soft_rows(model):
        lp = model.lp
        kappa = np.zeros(lp.m)
        eq = np.flatnonzero((lp.rlo == lp.rhi) & np.isfinite(lp.rlo))
        kappa[eq[::5]] = 1e-4
        lp.row_compliance = kappa

    try:
        for what, mutate, per_iter, qp in (("bounds", grid_limits, lambda n, m: 8 * (6 * n + 5 * m), 0), ("soft rows", soft_rows, lambda n, m: 8 * (4 * n + 3 * m) + 8 * m, 1)):
            lane = solve("lane", mutate)
            two = solve("two_launch", mutate)
            n, m = lane[5].lp.n, lane[5].lp.m
            assert lane[3] == 3 and two[3] == 1, (what, lane[3], two[3])          # DSP_STREAM_FORM_LANE / _TWO_LAUNCH
            assert lane[2] == per_iter(n, m) and lane[4] == qp == two[4], (what, lane[2], lane[4])
            assert np.allclose(lane[0], two[0], rtol=1e-6, atol=1e-6), (what, lane[0], two[0])
            assert (np.abs(lane[1] - two[1]) <= 0.05 * two[1] + 128).all(), (what, lane[1], two[1])
            if what == "bounds":
                assert len(set(np.round(lane[0], 6))) == B               # the limits bind differently: six different optima
    finally:
        for k in keys:
            monkeypatch.delenv(k, raising=False)


@gpu
@pytest.mark.parametrize("B,bounds", [(200, "shared"), (130, "per_scenario")])
def test_lane_form_packs_the_scenarios_still_iterating_into_fewer_groups(monkeypatch, B, bounds):
    """A lane-form solve of more than 64 scenarios runs in phases (dsp_stream_lane.hip: lane_run): when a quarter of its groups of 64
    lanes could be freed, the iterate goes back to the scenario-major workspace and the scenarios still iterating are packed into
    fewer groups (dsp_stats::stream_phases).  200 two-week design LPs (4 groups -> 3 -> 2 -> 1), and 130 (a last group of 2 lanes)
    with a grid-connection limit per member (bounds read per lane: they move with their scenario), against the same solve with
    every scenario keeping its lane to the end (DSP_LANE_COMPACT=0): same terminations, same optima; a scenario's iteration count
    may differ by a check period or two (the long columns' partial sums are added over another tiling)."""
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    T = 336
    res = {}
    try:
        for packed in (True, False):
            monkeypatch.delenv("DSP_LANE_COMPACT", raising=False)
            if not packed:
                monkeypatch.setenv("DSP_LANE_COMPACT", "0")
            solver = HipPdlpSolver(device=0, check_every=64, max_iter=400_000)
            handles, model = scenarios.price_taker_batch(T, B, solver)
            if bounds == "per_scenario":
                lb, ub, _, _ = model.block.current_bounds()
                model.lb, model.ub = np.tile(lb, (B, 1)), np.tile(ub, (B, 1))
                cols = [j for j, name in enumerate(model.lp.col_names) if name.startswith("splitter.grid_elec[")]
                for k in range(B):
                    model.ub[k, cols] = 847.0e3 * (0.55 + 0.4 * ((37 * k) % B) / B)      # kW: binding for the smaller ones
            solver.solve(model)
            st = solver.last_stats
            assert st.streaming == 1 and st.stream_form == 3 and (model.status == 0).all(), (packed, st.stream_form, np.bincount(model.status))
            n, m = model.lp.n, model.lp.m
            assert st.stream_bytes_per_iteration == (8 * (4 * n + 3 * m) if bounds == "shared" else 8 * (6 * n + 5 * m))
            res[packed] = (model.objective.copy(), model.iterations.copy(), int(st.stream_phases), model.x[:, handles["battery_system_capacity"].index].copy())
    finally:
        monkeypatch.delenv("DSP_LANE_COMPACT", raising=False)
    (obj_p, it_p, ph_p, cap_p), (obj_k, it_k, ph_k, cap_k) = res[True], res[False]
    assert ph_k == 1 and 2 <= ph_p <= 4, (ph_p, ph_k)
    assert it_p.max() > 1.5 * it_p.min()                                  # (the scenarios do finish at different times)
    assert np.allclose(obj_p, obj_k, rtol=1e-6, atol=1e-6), np.abs(obj_p - obj_k).max()
    assert np.allclose(cap_p, cap_k, rtol=1e-4, atol=1.0), np.abs(cap_p - cap_k).max()
    assert (np.abs(it_p - it_k) <= 0.1 * it_k + 256).all(), (it_p, it_k)
    if bounds == "per_scenario":
        assert len(set(np.round(obj_p, 4))) > B // 2                      # the limits bind differently


@gpu
def test_lane_form_wide_records_and_eight_long_columns(monkeypatch):
    """The wide instantiations of the lane form on the device.  (a) WC = WR = 8: the two-week design LP with every row written
    TWICE (rows 2 i and 2 i + 1: the same feasible set and optimum; every column then has up to 8 short entries) through the plain
    C-ABI wrapper, against the original LP's solve.  (b) NLP = 8: the wind + battery + PEM design LP with the throughput accumulator
    on 3 nodes (6 long columns: nodes, battery power, PEM size, the periodic state of charge) against its chain form."""
    import dataclasses
    import torch
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import DeviceLP, HipPdlpSolver, default_options
    monkeypatch.setenv("DSP_LANE_MIN_B", "1")
    try:
        # ---- (a) ---------------------------------------------------------------------------------------------------------------
        T, B = 336, 6
        solver = HipPdlpSolver(device=0, check_every=64, max_iter=400_000)
        handles, model = scenarios.price_taker_batch(T, B, solver)
        solver.solve(model)
        assert (model.status == 0).all() and solver.last_stats.stream_form == 3
        lp = model.lp
        A = lp.csr()
        rep = np.repeat(np.arange(lp.m), 2)
        A2 = A[rep].tocsr()
        A2.sort_indices()
        lp2 = dataclasses.replace(lp, m=2 * lp.m, indptr=A2.indptr.astype(np.int32), indices=A2.indices.astype(np.int32), data=A2.data.astype(np.float64),
                                  rlo=lp.rlo[rep], rhi=lp.rhi[rep], row_names=[lp.row_names[i] for i in rep] if lp.row_names else [])
        dev = torch.device("cuda", 0)
        up = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float64)).to(dev)
        lb, ub, _, _ = model.scenario_bounds()
        dlp = DeviceLP(lp2, 0, default_options(check_every=64, max_iter=400_000))
        out = dlp.solve(B, up(model.c), up(lb), up(ub), up(lp2.rlo), up(lp2.rhi), obj_offset=up(model.c0))
        st = dlp.last_stats
        assert st.streaming == 1 and st.stream_form == 3 and st.n_optimal == B, (st.stream_form, st.n_optimal)
        obj2 = out["obj"].cpu().numpy() + model.c0
        assert np.allclose(obj2, model.objective, rtol=1e-6, atol=1e-6), (obj2, model.objective)
        dlp.close()
        # ---- (b) ---------------------------------------------------------------------------------------------------------------
        T, B = 1000, 4
        res = {}
        for thr, nodes in (("chain", 1), ("two_level", 3)):
            solver = HipPdlpSolver(device=0, check_every=64, max_iter=1_000_000)
            handles, m2 = scenarios.pem_price_taker_batch(T, B, solver, inputs="rts303", throughput=thr, coarse_nodes=nodes)
            solver.solve(m2, tee=True)
            assert (m2.status == 0).all() and solver.last_stats.stream_form == 3, (thr, m2.status, solver.last_stats.stream_form)
            res[thr] = m2.objective.copy()
        assert np.allclose(res["chain"], res["two_level"], rtol=1e-6, atol=1e-6), res
    finally:
        monkeypatch.delenv("DSP_LANE_MIN_B", raising=False)


@gpu
@pytest.mark.parametrize("throughput,B", [("chain", 8), ("two_level", 16), ("two_level", 64)])
def test_year_long_price_taker_lps_converge(throughput, B):
    """The reference's own horizon (wind_battery_LMP.py: 8736 hourly periods, n = m = 52 419) for the first 8 members of the family
    against the oracle fixture (HiGHS on the un-reduced LP, tools/make_price_taker_fixtures.py).  Round 2 never converged here:
    the step size rested on a 500-iteration power-iteration estimate of ||A||, 1 % short for this near-Toeplitz matrix.
    `two_level`: the same LPs with the battery's accumulated throughput as 3 node values + local deviations (an exact change of
    variables, flowsheets/price_taker.py): same optima on the same kernels in a fifth of the iterations (the 16 scenarios of the
    fixture: 404 k -> 75 k on average, slowest 770 k -> 139 k; profiles/r30_two_level_probe_B16.log).  16 members: the whole fixture
    on the workgroup-per-tile form of round 3 (small batches); 64 members (the family four times over): the lane-per-scenario form
    of round 4 (csrc/dsp_stream_lane.hip), whose lanes drop out of the walk as their scenarios finish."""
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    fx = np.load(os.path.join(GOLD, "oracle_price_taker.npz"))
    T = 8736
    solver = HipPdlpSolver(device=0, check_every=64, max_iter=2_000_000)
    handles, model = scenarios.price_taker_batch(T, B, solver, throughput=throughput)
    solver.solve(model, tee=True)
    assert (model.status == 0).all(), (model.status, model.iterations)
    if throughput == "two_level":
        assert solver.last_stats.streaming == 1 and model.iterations.max() < 250_000, model.iterations
    member = np.arange(B) % len(scenarios.PRICE_TAKER_FAMILY)
    ref = fx["T8736/obj"][member]
    err = np.abs(model.objective - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() < 1e-6, (err, model.iterations)
    batt = model.x[:, handles["battery_system_capacity"].index] * 1e-3
    np.testing.assert_allclose(batt, fx["T8736/batt_mw"][member], rtol=2e-3, atol=1.0)
    if B >= 32:
        n, m = model.lp.n, model.lp.m
        assert solver.last_stats.stream_bytes_per_iteration == 8 * (4 * n + 3 * m)


@gpu
@pytest.mark.parametrize("B", [8, 32])
def test_year_long_pem_price_taker_lps_converge(B):
    """LP #5 (reference wind_battery_pem_optimize, wind_battery_PEM_LMP.py:180-298) at the horizon the reference's sweeps run it at
    (run_pricetaker_wind_PEM.py:54-56: every hour of the year, 8736 periods; n = 61 156): the members of the hydrogen-price x PEM-
    capital-cost family against the oracle fixture (HiGHS on the un-reduced LP, 27 s per member: tools/make_price_taker_fixtures.py
    --pem) - objectives to 1e-6, the PEM's size within the reference test's own tolerance (test_RE_flowsheet.py:139-164: +- 1 MW on
    487 MW), no battery.  8 members: the small-batch form; 32 (the family twice over): the lane-per-scenario form."""
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    fx = np.load(os.path.join(GOLD, "oracle_price_taker.npz"))
    T = 8736
    solver = HipPdlpSolver(device=0, check_every=64, max_iter=2_000_000)
    handles, model = scenarios.pem_price_taker_batch(T, B, solver, inputs="rts303")
    solver.solve(model, tee=True)
    assert solver.last_stats.streaming == 1 and (model.status == 0).all(), (model.status, model.iterations)
    member = np.arange(B) % len(scenarios.PEM_PRICE_TAKER_FAMILY)
    ref = fx["pem_T8736/obj"][member]
    err = np.abs(model.objective - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() < 1e-6, (err, model.iterations)
    pem = model.x[:, handles["pem_system_capacity"].index] * 1e-3
    np.testing.assert_allclose(pem, fx["pem_T8736/pem_mw"][member], rtol=2e-3, atol=1.0)
    batt = model.x[:, handles["battery_system_capacity"].index] * 1e-3
    np.testing.assert_allclose(batt, fx["pem_T8736/batt_mw"][member], rtol=2e-3, atol=1.0)


@gpu
def test_reference_price_taker_goldens_on_the_gpu(golden):
    """The reference's own price-taker tests (renewables_case/tests/test_RE_flowsheet.py:123-161) through the HIP path: LP #4 (wind +
    battery, one week) and LP #5 (wind + battery + PEM, six days, hydrogen at 2.5 $/kg) on the reference's inputs - SRW wind speeds
    through the PySAM-free wind resource model, LMPs capped at 200 - reproduce its NPV / revenues / PEM size (the reference asserts
    rel 1e-3 / 1e-2; here 1e-6), and every member of the two scenario families matches the independent oracle to 1e-6."""
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    from oracle import dispatch_lp_oracle as orc
    g8, g10 = golden["G8_price_taker_wind_battery"], golden["G10_price_taker_wind_battery_pem"]
    solver = HipPdlpSolver(device=0, check_every=64, max_iter=1_000_000)
    # ---- LP #4 ------------------------------------------------------------------------------------------------------------
    T, B = g8["n_time_points"], 4
    handles, model = scenarios.price_taker_batch(T, B, solver, inputs="reference")
    solver.solve(model, tee=True)
    assert solver.last_stats.streaming == 1 and (model.status == 0).all(), (model.status, model.iterations)
    cf, lmp = scenarios.price_taker_reference_inputs(T)
    assert np.abs(cf - orc.sam_weibull_capacity_factor(np.load(os.path.join(os.path.dirname(GOLD), "..", "dispatches_amd", "data",
                                                                              "price_taker_inputs.npz"))["wind_speed_m_s"][:T])).max() < 1e-12
    ref = np.array([orc.wind_battery_price_taker(T, cf, lmp * lm, batt_cap_factor=bf)[0].solve(tight=True)[1] for bf, lm in model.family])
    assert (np.abs(model.objective - ref) / np.maximum(1.0, np.abs(ref))).max() < 1e-6, (model.objective, ref)
    x = model.x[0]
    assert model.block.expressions["NPV"][0].value(x) == pytest.approx(g8["NPV"], rel=1e-6)
    assert model.block.expressions["annual_revenue"][0].value(x) == pytest.approx(g8["annual_revenue"], rel=1e-6)
    assert x[handles["nameplate_power"].index] == pytest.approx(0.0, abs=g8["battery_abs"])
    # ---- LP #5 ------------------------------------------------------------------------------------------------------------
    T, B = g10["time_points"], 8
    handles, model = scenarios.pem_price_taker_batch(T, B, solver, design_opt=True)
    solver.solve(model, tee=True)
    assert solver.last_stats.streaming == 1 and (model.status == 0).all(), (model.status, model.iterations)
    cf, lmp = scenarios.price_taker_reference_inputs(T)
    ref = []
    for h2, pf in model.family:
        P, info = orc.wind_battery_pem_price_taker(T, cf, lmp, h2, True)
        c = P.c.copy()
        c[info["Cp"]] += 1e-5 * (pf - 1.0) * orc.PEM_CAP_COST                 # the family's PEM capital-cost factor
        ref.append(P.solve(c=c, tight=True)[1])
    ref = np.array(ref)
    assert (np.abs(model.objective - ref) / np.maximum(1.0, np.abs(ref))).max() < 1e-6, (model.objective, ref, model.iterations)
    x = model.x[1]                                                             # member 1 = the reference test's setting
    assert -model.objective[1] * 1e5 == pytest.approx(g10["NPV"], rel=1e-6)
    assert x[handles["pem_system_capacity"].index] * 1e-3 == pytest.approx(g10["pem_mw"], abs=g10["pem_mw_abs_full_design"])
    assert x[handles["battery_system_capacity"].index] * 1e-3 == pytest.approx(g10["batt_mw"], abs=0.5)
    assert model.block.expressions["annual_rev_E"][0].value(x) == pytest.approx(g10["annual_rev_E"], rel=1e-4)
    assert model.block.expressions["annual_rev_h2"][0].value(x) / 2.0 * g10["h2_price_per_kg"] == pytest.approx(g10["annual_rev_h2"], rel=1e-4)


@gpu
def test_nuclear_price_taker_enumeration_on_the_gpu():
    """The reference's 60-point nuclear + PEM enumeration (price_taker_analysis.py:353-419) at its own horizon, 366 x 24 hourly
    periods (n = 70 275), as ONE streaming batch sharing the constraint matrix - per-scenario objective AND per-scenario bounds
    (the fixed pem_capacity column), three design columns that touch every period - against the closed form (no tank: the hours
    decouple) and, for three points, the oracle's HiGHS solve of a shorter horizon through the same code path."""
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    from oracle import dispatch_lp_oracle as orc
    solver = HipPdlpSolver(device=0, check_every=64, max_iter=400_000)
    T, B = 8784, 60
    handles, model = scenarios.nuclear_price_taker_batch(T, B, solver)
    solver.solve(model, tee=True)
    assert solver.last_stats.streaming == 1 and (model.status == 0).all(), (np.bincount(model.status), model.iterations)
    closed = np.array([-1e-6 * orc.nuclear_price_taker_closed_form(model.lmp, hp, pc * 400.0) for hp, pc in model.family])
    err = np.abs(model.objective - closed) / np.maximum(1.0, np.abs(closed))
    assert err.max() < 1e-6, (err.max(), model.iterations)
    # ... and, at this full horizon, against HiGHS on the oracle's LP for four points of the grid (committed fixture:
    # tools/make_price_taker_fixtures.py --nuclear; HiGHS' presolve decouples the hours, under a second per point) - the closed form lives in the oracle module itself
    fx = np.load(os.path.join(GOLD, "oracle_price_taker.npz"))
    for k, ref in zip(fx["nuclear_T8784/k"], fx["nuclear_T8784/obj"]):
        assert abs(model.objective[k] - ref) <= 1e-6 * max(1.0, abs(ref)), (k, model.objective[k], ref)
    # n (incl. 3 design columns) and the algorithmic bytes: the members differ in the bounds of ONE design column (a long column, whose
    # bounds the lane form keeps per scenario in any case), so the batch shares every other bound: 4 n + 3 m doubles (the round-3
    # form read all bounds per scenario: 6 n + 5 m)
    n, m = model.lp.n, model.lp.m
    assert n == 8 * T + 3 and solver.last_stats.stream_bytes_per_iteration == 8 * (4 * n + 3 * m)
    # the same path at a horizon the oracle solves in seconds
    T2, B2 = 720, 12
    handles, small = scenarios.nuclear_price_taker_batch(T2, B2, solver)
    solver.solve(small)
    assert (small.status == 0).all()
    for k in (0, 5, 11):
        hp, pc = small.family[k]
        ref = orc.nuclear_price_taker(T2, small.lmp, hp, pc * 400.0)[0].solve(tight=True)[1]
        assert abs(small.objective[k] - ref) <= 1e-6 * max(1.0, abs(ref))


@gpu
def test_streaming_edge_cases():
    """B = 1, an invalid scenario (crossed bounds), the iteration limit."""
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    fx = np.load(os.path.join(GOLD, "oracle_price_taker.npz"))
    solver = HipPdlpSolver(device=0, check_every=64)
    handles, model = scenarios.price_taker_batch(168, 1, solver)
    solver.solve(model)
    assert model.status.tolist() == [0]
    assert abs(model.objective[0] - fx["T168/obj"][0]) <= 1e-6 * abs(fx["T168/obj"][0])
    handles, model = scenarios.price_taker_batch(168, 3, solver)
    lb, ub, _, _ = model.block.current_bounds()
    model.ub = np.tile(ub, (3, 1))
    j = model.lp.col_names.index("splitter.grid_elec[0]")
    assert lb[j] == 0.0
    model.ub[1, j] = -1.0                                   # below the column's lower bound
    solver.solve(model)
    assert model.status.tolist() == [0, 2, 0] and np.isnan(model.objective[1])
    limited = HipPdlpSolver(device=0, check_every=64, max_iter=128)
    handles, model = scenarios.price_taker_batch(168, 2, limited)
    limited.solve(model)
    assert model.status.tolist() == [1, 1] and (model.iterations == 128).all()


@gpu
@pytest.mark.parametrize("mode", ["non_anticipative", "monotone"])
def test_coupled_stochastic_bidders_on_the_gpu(rts309, mode):
    """n_scenario = 3 with different scenarios: the coupled day-ahead LP (582 columns, 408 / 432 rows: beyond the fused
    kernels, so the streaming path solves it) against the oracle's independent coupled formulation."""
    from dispatches_amd.hip_solver import HipPdlpSolver
    from dispatches_amd.workflow import Bidder, SelfScheduler
    from oracle import dispatch_lp_oracle as orc
    from tests.test_workflow_cpu import _thermal_bidder
    T, S = 24, 3
    cls, kw = (SelfScheduler, {}) if mode == "non_anticipative" else (Bidder, dict(scenario_coupling="monotone"))
    solver = HipPdlpSolver(device=0, check_every=64)
    bidder = _thermal_bidder(rts309, solver, S, cls=cls, history_days=3, **kw)
    bidder.compute_day_ahead_bids(date="2020-01-02")
    model = bidder.day_ahead_model
    assert solver.last_stats.streaming == 1 and (model.status == 0).all()
    P, _ = orc.wind_battery_da_coupled(T, rts309["rt_cf"][:T], model.da_prices, model.rt_prices, mode)
    ref = P.solve(tight=True)[1]
    assert model.coupled_objective == pytest.approx(ref, rel=1e-6)
    pda = model.x[:, model.pda_cols]
    if mode == "non_anticipative":
        assert np.allclose(pda, pda[0], atol=1e-4)
    else:
        da = model.da_prices
        for j in range(S):
            for k in range(j + 1, S):
                assert np.all((pda[k] - pda[j]) * (da[k] - da[j]) >= -1e-3)


@gpu
def test_coupled_stochastic_bidder_with_a_ramp_cost(rts309):
    """BASELINE config 5 in the upstream sense of "stochastic bidder": n_scenario = 3 DIFFERENT scenarios coupled by the
    monotone-bid rows, each copy with the quadratic ramp cost (soft rows with a compliance).  582 columns x 477 + 69 rows:
    beyond the fused kernels, so the HBM-resident streaming path (block-resident form) solves the QP; checked against the
    certified bracket of the QP oracle on its independent coupled formulation."""
    from dispatches_amd.hip_solver import HipPdlpSolver
    from dispatches_amd.workflow import Bidder
    from oracle import qp_cutting_plane as qp
    from tests.test_workflow_cpu import _thermal_bidder
    T, S, rho = 24, 3, 0.1
    solver = HipPdlpSolver(device=0, check_every=64)
    bidder = _thermal_bidder(rts309, solver, S, cls=Bidder, history_days=3, scenario_coupling="monotone", ramp_cost=rho)
    bidder.compute_day_ahead_bids(date="2020-01-02")
    model = bidder.day_ahead_model
    st = solver.last_stats
    assert st.streaming == 1 and st.quadratic == 1 and (model.status == 0).all(), (st.streaming, st.quadratic, model.status)
    out, P, pdas = qp.wind_battery_da_coupled_qp(T, rts309["rt_cf"][:T], model.da_prices, model.rt_prices, "monotone", rho)
    tol = 1e-6 * max(1.0, abs(out["upper"]))
    assert out["lower"] - tol <= model.coupled_objective <= out["upper"] + tol, (model.coupled_objective, out["lower"], out["upper"])
    assert float(np.sum(model.objective)) == pytest.approx(model.coupled_objective, rel=1e-9, abs=1e-6)
    # the ramp cost really is in there: the same coupled problem without it is cheaper and ramps more
    plain = _thermal_bidder(rts309, HipPdlpSolver(device=0, check_every=64), S, cls=Bidder, history_days=3, scenario_coupling="monotone")
    plain.compute_day_ahead_bids(date="2020-01-02")
    assert plain.day_ahead_model.coupled_objective < model.coupled_objective
    ramps = lambda m: np.abs(np.diff(m.expression_values("P_T"), axis=1)).sum()
    assert ramps(model) < ramps(plain.day_ahead_model)
