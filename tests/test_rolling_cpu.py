"""CPU tier of the device-resident batched double loop (dispatches_amd/rolling.py, BASELINE config 4): the same window
gathers, objective / bound rewrites and state hand-off on CPU tensors with a HiGHS stand-in for the device solver,
against the host-object path (Bidder / Tracker, one scenario each) - and the shard / all-gather logic of
`bench.py --workload double_loop` under gloo with world size 2."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_reference(k, stride, offers, da_prices, hours):
    """First hours of day 0 of plant k through the product's Bidder / Tracker objects (reference call order)."""
    from dispatches_amd import scenarios
    from dispatches_amd.flowsheets import MultiPeriodWindBattery
    from dispatches_amd.workflow import Bidder, Tracker
    from tests._highs_solver import HighsTestSolver
    s = scenarios.load_series("rts_gmlc_309.npz")
    N = len(s["rt_lmp"])
    start = (stride * k) % N
    cf = np.roll(s["rt_cf"], -start)
    fc = scenarios.WindowForecaster(s["da_lmp"], s["rt_lmp"], [start], clip=(0.0, 500.0))
    mk = lambda: MultiPeriodWindBattery(scenarios._thermal_data("309_WIND_1", "Carter", 200.0, 25.0), wind_capacity_factors=list(cf),
                                        wind_pmax_mw=200.0, battery_pmax_mw=25.0, battery_energy_capacity_mwh=100.0)
    bidder = Bidder(mk(), day_ahead_horizon=48, real_time_horizon=4, n_scenario=1, solver=HighsTestSolver(), forecaster=fc)
    tracker = Tracker(tracking_model_object=mk(), tracking_horizon=4, n_tracking_hour=1, solver=HighsTestSolver())
    out = []
    for h in range(hours):
        bidder.compute_real_time_bids("2020-01-02", h, list(da_prices), list(offers))
        dispatch = [float(v) for v in bidder.real_time_model.expression_values("P_T")[0]]
        prof = tracker.track_market_dispatch(market_dispatch=dispatch, date="2020-01-02", hour=h)
        delivered = tracker.get_last_delivered_power()
        tracker.update_model(**prof)
        bidder.update_real_time_model(**prof)
        out.append((delivered, round(prof["realized_soc"][-1], 2), round(prof["realized_energy_throughput"][-1], 2)))
    return out


def test_batched_double_loop_logic_on_cpu():
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from tests._highs_solver import HighsTensorLP
    B, hours, stride = 2, 3, 17
    loop = BatchedWindBatteryDoubleLoop(B, stride=stride, lp_backend=HighsTensorLP)
    offers = loop.day_ahead().numpy()
    da_prices = loop.da_prices.numpy()
    assert offers.shape == (B, 24) and (offers >= -1e-9).all() and (offers <= 225 + 1e-6).all()
    got = []
    for _ in range(hours):
        d = loop.hour_step().numpy()
        got.append((d.copy(), loop.soc.numpy().copy(), loop.thr.numpy().copy()))
    res, ok = loop.results()
    assert ok and loop.hour == hours and loop.solves == B * (1 + 2 * hours)
    for k in range(B):
        ref = _host_reference(k, stride, offers[k], da_prices[k], hours)
        for h in range(hours):
            # both paths solve the same LPs with the same simplex code: the vertices agree
            assert got[h][0][k] == pytest.approx(ref[h][0], abs=1e-6 * 225), (k, h)
            assert got[h][1][k] == pytest.approx(ref[h][1], abs=0.011), (k, h)
            assert got[h][2][k] == pytest.approx(ref[h][2], abs=0.011), (k, h)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from dispatches_amd.distributed import gather_device_results, make_gather_buffers, shard_bounds
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from tests._highs_solver import HighsTensorLP
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(total, world, rank)
        per = max(shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world))
        loop = BatchedWindBatteryDoubleLoop(hi - lo, first_scenario=lo, lp_backend=HighsTensorLP)
        loop.day_ahead()
        loop.hour_step()
        buffers = make_gather_buffers(world, per, torch.device("cpu"), width=2)
        everything = gather_device_results(dict(obj=loop.revenue, status=torch.zeros(hi - lo, dtype=torch.float64)), buffers, per)
        q.put((rank, lo, hi, everything.numpy().copy(), loop.revenue.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_double_loop_shards_world2_gloo():
    """3 plants over 2 ranks (ragged 2 + 1): every rank ends with every plant's revenue, equal to the single-process run."""
    import torch.multiprocessing as mp
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from tests._highs_solver import HighsTensorLP
    total, world = 3, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = BatchedWindBatteryDoubleLoop(total, lp_backend=HighsTensorLP)
    single.day_ahead()
    single.hour_step()
    ref = single.revenue.numpy()
    for rank, lo, hi, everything, mine in got:
        np.testing.assert_allclose(mine, ref[lo:hi], rtol=1e-9, atol=1e-6)
        flat = np.concatenate([everything[r, :b - a, 0] for r, (a, b) in
                               enumerate([(0, 2), (2, 3)])])
        np.testing.assert_allclose(flat, ref, rtol=1e-9, atol=1e-6)


def test_rolling_hours_are_optimal_for_the_oracles_lps_cpu_backend():
    """The same oracle-anchored check as the GPU tier (tests/_rolling_oracle.py) with the HiGHS stand-in as the loop's LP backend:
    pins the loop's window / state / objective hand-off to the ORACLE's formulation of the hourly LPs, not to the product's."""
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from tests._highs_solver import HighsTensorLP
    from tests._rolling_oracle import check_rolling_hours_against_the_oracle
    loop = BatchedWindBatteryDoubleLoop(4, stride=17, lp_backend=HighsTensorLP)
    check_rolling_hours_against_the_oracle(loop, hours=5, stride=17)


def test_day_ahead_warm_start_buffers_hold_yesterdays_solution_shifted_by_a_day():
    """warm_start=True: after a day-ahead solve the persistent start buffers hold period t + 24 of its solution in period t
    (the last 24 periods keep their own values) - columns by their names, whatever order the flattener emitted them in."""
    import re
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from tests._highs_solver import HighsTensorLP
    loop = BatchedWindBatteryDoubleLoop(2, stride=17, lp_backend=HighsTensorLP, warm_start=True)
    assert not loop.da_x0.any() and not loop.da_y0.any()               # zero = the cold start of the first day
    loop.day_ahead()
    x, x0 = loop.da.out["x"].numpy(), loop.da_x0.numpy()
    names = loop.da.lp.col_names
    where = {nm: k for k, nm in enumerate(names)}
    pat = re.compile(r"^(.*)\[(\d+)\]$")
    shifted = kept = 0
    for k, nm in enumerate(names):
        mt = pat.match(nm)
        src = where.get(f"{mt.group(1)}[{int(mt.group(2)) + 24}]") if mt else None
        if src is None:
            assert (x0[:, k] == x[:, k]).all(), nm
            kept += 1
        else:
            assert (x0[:, k] == x[:, src]).all(), nm
            shifted += 1
    assert shifted >= 24 * 6 and kept >= 24 * 6
    # the second day starts from the buffers and is still an optimal day (HiGHS ignores the start point: same offers as cold)
    cold = BatchedWindBatteryDoubleLoop(2, stride=17, lp_backend=HighsTensorLP, warm_start=False)
    cold.day_ahead()
    assert np.allclose(cold.da_offer.numpy(), loop.da_offer.numpy())


def test_recorded_trajectory_check_and_year_fixture_on_cpu():
    """The machinery of the full-year GPU test, on the HiGHS stand-in backend: the loop RECORDS two plants over three days (state before
    every hour, whole solutions), the recorded LPs are checked against the oracle's own LPs (teacher forced, in day blocks as the GPU test
    splits them), and the per-day revenue / state of charge rebuilt from the record equals (a) the loop's accumulators and (b) the
    committed free-run oracle fixture tests/golden/rolling_year.npz (both paths are simplex codes: same vertices)."""
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from tests._highs_solver import HighsTensorLP
    from tests._rolling_oracle import check_recorded_plant, column_maps
    fx = np.load(os.path.join(ROOT, "tests", "golden", "rolling_year.npz"))
    plants, days = [0, 1], 3
    assert list(fx["plants"][:2]) == plants and int(fx["days"]) == 366 and fx["revenue"].shape == (16, 366)
    loop = BatchedWindBatteryDoubleLoop(2, stride=int(fx["stride"]), lp_backend=HighsTensorLP, record=(plants, days))
    for _ in range(days):
        loop.run_day()
    rec = loop.recorded()
    maps = column_maps(loop)
    for j, k in enumerate(plants):
        mine = {key: v[:, j] for key, v in rec.items() if key != "plants"}
        whole = check_recorded_plant((k, 17, maps, mine, range(24 * days)))
        assert max(whole["worst"].values()) < 1e-9
        # the same in blocks of one day, each with the hour before it
        rev = np.zeros(days)
        for d in range(days):
            h0 = 24 * d
            lead = 1 if d else 0
            part = {key: (v[d:d + 1] if key.startswith("da_") else v[h0 - lead:h0 + 24]) for key, v in mine.items()}
            rev[d] = check_recorded_plant((k, 17, maps, part, range(h0, h0 + 24)), base_hour=h0 - lead, base_day=d)["revenue"][0]
        assert np.allclose(rev, whole["revenue"], rtol=1e-12)
        assert np.isclose(whole["revenue"].sum(), loop.revenue[j].item(), rtol=1e-12)
        assert np.allclose(whole["revenue"], fx["revenue"][j, :days], rtol=1e-9)
        assert np.allclose(whole["soc"], fx["soc"][j, :days], atol=0.011)
        assert np.allclose(whole["delivered"], fx["delivered"][j, :days], rtol=1e-9)
    # a tampered record is caught: a state that is not the rounding of what the tracker realised
    bad = {key: v[:, 0].copy() for key, v in rec.items() if key != "plants"}
    bad["state"][30, 0] += 1.0
    with pytest.raises(AssertionError):
        check_recorded_plant((0, 17, maps, bad, range(24 * days)))


def test_year_fixture_wraps_the_data_end():
    """Plant 513 of the fixture starts 15 hours before the end of the 8736-hour data: its first day-ahead window wraps (the modulo window =
    parametrized_bidder.py:52-58's padding from the start of the data).  Two days of the oracle's free run reproduce the fixture."""
    from oracle import double_loop_oracle as dl
    fx = np.load(os.path.join(ROOT, "tests", "golden", "rolling_year.npz"))
    j = list(fx["plants"]).index(513)
    da, rt, cf = dl.load_year()
    assert (17 * 513) % len(rt) == 8721 and len(rt) == 8736
    w = dl.window(rt, 8721, 0, 48)
    assert (w[:15] == rt[8721:]).all() and (w[15:] == rt[:33]).all()
    out = dl.roll(513, 2)
    assert np.allclose(out["revenue"], fx["revenue"][j, :2], rtol=1e-12) and np.allclose(out["soc"], fx["soc"][j, :2])


def test_two_optimal_trajectories_of_the_same_loop_drift_apart():
    """Why the full-year GPU test compares FREE runs in aggregate only: the day-ahead LPs are degenerate (a fifth of the prices are exactly
    0), and a run that takes its day-ahead offers from an interior point of the optimal face (HiGHS interior point without crossover -
    what a first-order method tends to return) instead of a simplex vertex is every bit as optimal, hour by hour, yet leaves the vertex
    trajectory within days - by the same 1e-3 of revenue the GPU loop differs from the fixture by."""
    from oracle import double_loop_oracle as dl
    fx = np.load(os.path.join(ROOT, "tests", "golden", "rolling_year.npz"))
    days = 20
    o = dl.roll(0, days, interior_day_ahead=True)
    ref_rev, ref_mwh = fx["revenue"][0, :days], fx["delivered"][0, :days]
    same = np.abs(o["revenue"] - ref_rev) <= 1e-6 * np.maximum(1.0, np.abs(ref_rev))
    assert 0 < same.sum() < days                                   # parts, but not everywhere
    assert abs(o["revenue"].sum() - ref_rev.sum()) <= 2.5e-3 * ref_rev.sum()
    assert abs(o["delivered"].sum() - ref_mwh.sum()) <= 2e-4 * ref_mwh.sum()


def test_generic_loop_reproduces_the_wind_battery_loop_on_cpu():
    """dispatches_amd/rolling_flowsheets.py::BatchedDoubleLoop("wind_battery") - the loop written over a descriptor of the flowsheet's
    rolling state - gives exactly the day of the specialised BatchedWindBatteryDoubleLoop (same windows, objective vectors and constants,
    bounds, hand-off), on the HiGHS stand-in backend."""
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from dispatches_amd.rolling_flowsheets import BatchedDoubleLoop
    from tests._highs_solver import HighsTensorLP
    B = 3
    a = BatchedWindBatteryDoubleLoop(B, lp_backend=HighsTensorLP)
    g = BatchedDoubleLoop("wind_battery", B, lp_backend=HighsTensorLP)
    a.run_day()
    g.run_day()
    # (power output as a dense row product there, as the sum of its two columns here: the last bits of the sums may differ)
    assert np.allclose(a.revenue.numpy(), g.revenue.numpy(), rtol=1e-12) and np.allclose(a.energy_mwh.numpy(), g.energy_mwh.numpy(), rtol=1e-12)
    assert np.array_equal(a.soc.numpy(), g.state.numpy()[:, 0]) and np.array_equal(a.thr.numpy(), g.state.numpy()[:, 1])
    for x, y in ((a.da, g.da), (a.rt, g.rt), (a.tr, g.tr)):
        assert np.allclose(x.c0.numpy(), y.c0.numpy(), rtol=1e-13) and np.allclose(x.c.numpy(), y.c.numpy(), rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("flowsheet", ["nuclear", "wind_pem"])
def test_generic_loop_hours_are_optimal_for_the_oracles_lps_on_cpu(flowsheet):
    """The rolling double loop of the nuclear (config 2 is a nuclear DOUBLE LOOP: holdup hand-off `round(holdup[-1])`,
    nuclear_flowsheet_multiperiod_class.py:218-239; 12-h real-time horizon) and wind + PEM (capacity-factor shift only,
    wind_PEM_double_loop.py:185-204) flowsheets, on the HiGHS stand-in backend: hourly objectives against the oracle's own LPs."""
    from dispatches_amd.rolling_flowsheets import BatchedDoubleLoop
    from tests._highs_solver import HighsTensorLP
    from tests._rolling_oracle import check_flowsheet_hours_against_the_oracle
    loop = BatchedDoubleLoop(flowsheet, 3, lp_backend=HighsTensorLP)
    assert loop.rt.T == (12 if flowsheet == "nuclear" else 4) and loop.tr.T == 4 and loop.state.shape == (3, 1 if flowsheet == "nuclear" else 0)
    worst = check_flowsheet_hours_against_the_oracle(loop, 4)
    assert worst <= 1e-6
    if flowsheet == "nuclear":                                        # the tank fills: the hand-off is not the trivial zero
        assert float(loop.state.abs().max()) > 0
