"""GPU test of the device-resident batched double loop (dispatches_amd/rolling.py, BASELINE config 4) against the
host-object path: for three plants the first hours of a simulated day are replayed through the product's own
Bidder / Tracker objects (one scenario each, the reference's call order: real-time bid -> tracking -> update_model on
tracker and bidder) and must give the same real-time offers, delivered power and realised state."""
import numpy as np
import pytest

gpu = pytest.mark.gpu


@gpu
def test_batched_double_loop_matches_host_objects():
    import torch
    from dispatches_amd import scenarios
    from dispatches_amd.flowsheets import MultiPeriodWindBattery
    from dispatches_amd.hip_solver import HipPdlpSolver
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from dispatches_amd.workflow import Bidder, Tracker
    B, hours, stride = 3, 5, 17
    loop = BatchedWindBatteryDoubleLoop(B, device=0, stride=stride)
    offers = loop.day_ahead().cpu().numpy()                       # [B, 24] day-ahead offers = cleared dispatch (stub market)
    da_prices = loop.da_prices.cpu().numpy()
    dev = dict(delivered=[], soc=[], thr=[])
    for _ in range(hours):
        dev["delivered"].append(loop.hour_step().cpu().numpy())
        dev["soc"].append(loop.soc.cpu().numpy())
        dev["thr"].append(loop.thr.cpu().numpy())
    res, all_optimal = loop.results()
    assert all_optimal
    s = scenarios.load_series("rts_gmlc_309.npz")
    N = len(s["rt_lmp"])
    for k in range(B):
        start = (stride * k) % N
        cf = np.roll(s["rt_cf"], -start)
        fc = scenarios.WindowForecaster(s["da_lmp"], s["rt_lmp"], [start], clip=(0.0, 500.0))
        mk = lambda: MultiPeriodWindBattery(scenarios._thermal_data("309_WIND_1", "Carter", 200.0, 25.0),
                                            wind_capacity_factors=list(cf), wind_pmax_mw=200.0, battery_pmax_mw=25.0,
                                            battery_energy_capacity_mwh=100.0)
        bidder = Bidder(mk(), day_ahead_horizon=48, real_time_horizon=4, n_scenario=1, solver=HipPdlpSolver(device=0), forecaster=fc)
        tracker = Tracker(tracking_model_object=mk(), tracking_horizon=4, n_tracking_hour=1, solver=HipPdlpSolver(device=0))
        gen = bidder.generator
        for h in range(hours):
            rt_bids = bidder.compute_real_time_bids("2020-01-02", h, list(da_prices[k]), list(offers[k]))
            dispatch = [rt_bids[h + j][gen]["p_max"] for j in range(4)]
            # the device loop hands the un-rounded real-time offer to the tracker; bids carry it rounded to 2 dp
            dispatch = [float(v) for v in bidder.real_time_model.expression_values("P_T")[0]]
            prof = tracker.track_market_dispatch(market_dispatch=dispatch, date="2020-01-02", hour=h)
            delivered = tracker.get_last_delivered_power()
            tracker.update_model(**prof)
            bidder.update_real_time_model(**prof)
            assert delivered == pytest.approx(dev["delivered"][h][k], abs=1e-6 * 225), (k, h)
            assert round(prof["realized_soc"][-1], 2) == pytest.approx(dev["soc"][h][k], abs=0.011), (k, h)
            assert round(prof["realized_energy_throughput"][-1], 2) == pytest.approx(dev["thr"][h][k], abs=0.011), (k, h)


@gpu
def test_graph_replay_reproduces_the_eager_loop():
    """From the second simulated day on the day-ahead step and the 24 hour steps are replayed from hipGraphs captured on day 2
    (the clock, the realised state and the cleared day-ahead dispatch live in persistent device tensors).  Six days with
    graphs must give bit-identical results to six days issued eagerly - at a batch (1024) and a number of replays (5) where a
    hipMemsetAsync captured into the graph did NOT run again on replay (round 6: the day-ahead work queue stayed exhausted and every
    later day returned the captured day's offers; dsp_capi.hip::queue_reset_kernel), with the day-ahead iteration counts changing
    from day to day as the LPs do."""
    import torch
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    B, days = 1024, 6
    res = {}
    for graphs in (False, True):
        loop = BatchedWindBatteryDoubleLoop(B, device=0, use_graphs=graphs)
        iters = []
        for _ in range(days):
            loop.run_day()
            iters.append(int(loop.da.out["iters"].sum().item()))
        assert len(set(iters)) == days, iters
        out, ok = loop.results()
        assert ok and loop.hour == 24 * days and int(loop.hour_t.item()) == 24 * days
        assert (len(loop._graphs) == 25) == graphs
        res[graphs] = {k: v.cpu().numpy().copy() for k, v in out.items()}
        res[graphs]["da_energy"] = loop.da_energy_mwh.cpu().numpy().copy()
    for k in res[False]:
        assert np.array_equal(res[False][k], res[True][k]), k
    assert np.abs(res[True]["obj"]).max() > 0


@gpu
def test_fused_update_kernel_is_bit_identical_to_the_tensor_operations():
    """dsp_wb_rolling_update (three launches per hour: prices / state / bounds before the real-time solve, offer -> dispatch
    rows between the solves, realised state / revenue / clock after the tracking solve) against the ~45 element-wise tensor
    operations it replaces: two simulated days, every result bit for bit."""
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    B, days = 96, 2
    res = {}
    for fused in (False, True):
        loop = BatchedWindBatteryDoubleLoop(B, device=0, use_graphs=False, use_fused=fused)
        assert loop.use_fused == fused
        for _ in range(days):
            loop.run_day()
        out, ok = loop.results()
        assert ok and int(loop.hour_t.item()) == 24 * days
        res[fused] = {k: v.cpu().numpy().copy() for k, v in out.items()}
        res[fused]["rt_c"] = loop.rt.c.cpu().numpy().copy()
        res[fused]["rt_ub"] = loop.rt.ub.cpu().numpy().copy()
        res[fused]["tr_rlo"] = loop.tr.rlo.cpu().numpy().copy()
    for k in res[False]:
        assert np.array_equal(res[False][k], res[True][k]), k


@gpu
def test_rolling_hours_are_optimal_for_the_oracles_lps():
    """Oracle-anchored check of the device-resident loop (tests/_rolling_oracle.py): every hourly solution of the HIP loop, mapped
    into the oracle's variables, is feasible and optimal for the oracle's own real-time bidding / tracking LP of the loop's state."""
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from tests._rolling_oracle import check_rolling_hours_against_the_oracle
    loop = BatchedWindBatteryDoubleLoop(12, device=0, stride=17, use_graphs=False)
    check_rolling_hours_against_the_oracle(loop, hours=8, stride=17)
    assert int(loop.uncertified.item()) == 0


@gpu
@pytest.mark.parametrize("graphs", [False, True])
def test_day_ahead_warm_start_solves_the_same_lps_in_fewer_iterations(graphs):
    """Rolling warm start of the 48-h day-ahead LP (yesterday's solution shifted by a day + its primal weight, persistent device
    buffers, one hipGraph for every day): on day 3 the warm solve reaches the SAME objectives as a cold solve of the same LP data
    (1e-6), every plant optimal and certified, in fewer iterations on average."""
    import torch
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    B = 256
    loop = BatchedWindBatteryDoubleLoop(B, device=0, warm_start=True, use_graphs=graphs)
    loop.run_day()
    loop.run_day()
    loop.day_ahead()                                    # day 3: warm (replayed from the graph when graphs are on)
    torch.cuda.synchronize()
    out = loop.da.out
    obj_w, it_w = out["obj"].cpu().numpy().copy(), out["iters"].cpu().numpy().copy()
    assert (out["status"].cpu().numpy() == 0).all() and loop.da_x0.abs().sum().item() > 0
    cold = loop.da.solve(B)                             # the same vectors, cold start, automatic weight
    torch.cuda.synchronize()
    obj_c, it_c = cold["obj"].cpu().numpy(), cold["iters"].cpu().numpy()
    assert (cold["status"].cpu().numpy() == 0).all()
    err = np.abs(obj_w - obj_c) / np.maximum(1.0, np.abs(obj_c))
    assert err.max() < 1e-6, err.max()
    assert it_w.mean() < 0.85 * it_c.mean(), (it_w.mean(), it_c.mean())
    res, ok = loop.results()
    assert ok and int(loop.uncertified.item()) == 0


@gpu
def test_pipelined_groups_reproduce_the_single_loop():
    """rolling.PipelinedDoubleLoops: the same plants as two independent loops on two HIP streams (their simulated days overlap) end
    with exactly the per-plant revenues, states and energies of one loop - plants do not interact and a scenario's solve does not
    depend on its batch."""
    import torch
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop, PipelinedDoubleLoops
    n, days = 2048, 5
    one = BatchedWindBatteryDoubleLoop(n, device=0, first_scenario=5)
    two = PipelinedDoubleLoops(n, device=0, first_scenario=5, groups=2)
    assert two.groups == 2 and PipelinedDoubleLoops(8, device=0, groups=0).groups == 1
    free = PipelinedDoubleLoops(n, device=0, first_scenario=5, groups=2)
    seen = []
    for _ in range(days):
        one.run_day()
        two.run_day()
    free.run_days(days, per_day=lambda g, l: seen.append(g))           # the groups free-running on their streams, joined once (bench.py's year)
    torch.cuda.synchronize()
    r1, ok1 = one.results()
    r2, ok2 = two.results()
    r3, ok3 = free.results()
    assert ok1 and ok2 and ok3 and seen == [0, 1] * days
    for k in ("obj", "energy_mwh", "soc", "throughput"):
        assert torch.equal(r1[k], r2[k]), k
        assert torch.equal(r1[k], r3[k]), k
    assert int(two.uncertified) == int(one.uncertified) == int(free.uncertified) == 0


@gpu
def test_full_year_double_loop_8192_plants_against_the_oracle():
    """BASELINE config 4 as it is written: 8192 plants, 366 simulated days (8784 hand-offs; the data holds 8736 hours, so EVERY plant's
    windows wrap the data end - parametrized_bidder.py:52-58, wind_battery_double_loop.py:211-228), 147 M LP solves, the two-group
    pipelined loop replayed from hipGraphs as bench.py runs it.

    16 plants of the batch (tools/make_rolling_year_fixture.py::PLANTS: both groups, the shard edges of an 8-way split, plants that wrap
    in their first day) are RECORDED hour by hour on the device and checked two ways:
      1. teacher forced, every hour of the year: each of the 17 934 LPs of a plant is rebuilt by the oracle from the state the loop
         recorded (oracle/double_loop_oracle.py, un-reduced rows, HiGHS); the loop's solution mapped into the oracle's variables is
         feasible and optimal to 1e-6, and every state hand-off is the 2-dp rounding of what the tracker realised the hour before;
      2. against the committed FREE-RUN trajectory of the oracle for the same plants (tests/golden/rolling_year.npz): day-by-day
         revenue, delivered energy and end-of-day state of charge.  The LPs are degenerate (a fifth of the prices are exactly 0), so
         two optimal trajectories part at the first tie and never meet again: the ORACLE ALONE, taking its day-ahead offers from an
         interior point of the optimal face instead of a vertex, moves 60 days of revenue by 9e-4 (tests/test_rolling_cpu.py::
         test_two_optimal_trajectories_of_the_same_loop_drift_apart).  Measured for the GPU loop (profiles/r60c_rolling_tests.log):
         3426 of 5856 plant-days agree to 1e-6, annual revenue within 1.2e-3, delivered energy within 4.5e-5 with every hourly LP solved
         from the slack basis; 2793 plant-days, 4.1e-3 and 1.4e-4 with the simplex started from the previous hour's basis and the day-ahead solve from
         yesterday's shifted solution (other sequences of optimal points; `profiles/r69a_rolling_tests.log`).  The test reports the agreeing days and requires the annual totals within 6e-3 /
         5e-4: a check of the aggregate, the parity claim is check 1."""
    import multiprocessing as mp
    import os
    import time
    import torch
    from dispatches_amd.rolling import PipelinedDoubleLoops
    from tests._rolling_oracle import check_recorded_plant, column_maps
    from tools.make_rolling_year_fixture import PLANTS
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rolling_year.npz"))
    assert list(fx["plants"]) == PLANTS
    B, days = 8192, int(os.environ.get("DSP_YEAR_DAYS", "366"))
    assert days <= int(fx["days"])
    loops = PipelinedDoubleLoops(B, device=0, record=(PLANTS, days))
    t0 = time.perf_counter()
    for _ in range(days):
        loops.run_day()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    res, ok = loops.results()
    assert ok, "a non-optimal solve entered the year"
    assert int(loops.uncertified.item()) == 0
    rec = loops.recorded()
    assert list(rec["plants"]) == PLANTS and rec["state"].shape[0] == 24 * days
    maps = column_maps(loops.loops[0])
    # one task per (plant, block of days): the GPU box has far more cores than plants
    block = 6
    tasks = []
    for j, k in enumerate(PLANTS):
        mine = {key: np.ascontiguousarray(v[:, j]) for key, v in rec.items() if key != "plants"}
        for d0 in range(0, days, block):
            d1 = min(days, d0 + block)
            # every task carries the hour before its block too (the hand-off into the block's first hour is checked against it)
            h0, h1 = 24 * d0, 24 * d1
            part = {key: (v[d0:d1] if key.startswith("da_") else v[max(h0 - 1, 0):h1]) for key, v in mine.items()}
            tasks.append((k, j, d0, d1, part, h0))
    t1 = time.perf_counter()
    with mp.get_context("fork").Pool(min(len(tasks), max(1, (os.cpu_count() or 2) - 2))) as pool:
        outs = pool.map(_year_block, [(t, maps) for t in tasks], chunksize=1)
    check_wall = time.perf_counter() - t1
    revenue, delivered, soc = (np.zeros((len(PLANTS), days)) for _ in range(3))
    worst = dict(da=0.0, rt=0.0, tr=0.0)
    hours = 0
    for (k, j, d0, d1, _, _), o in zip(tasks, outs):
        revenue[j, d0:d1], delivered[j, d0:d1], soc[j, d0:d1] = o["revenue"], o["delivered"], o["soc"]
        hours += o["hours"]
        for key in worst:
            worst[key] = max(worst[key], o["worst"][key])
    assert hours == len(PLANTS) * 24 * days
    # the loop's own annual accumulators are the sums of what was recorded
    total = res["obj"].cpu().numpy()[PLANTS]
    assert np.allclose(total, revenue.sum(1), rtol=1e-9), (total, revenue.sum(1))
    # 2. the oracle's free-run trajectory
    rel = lambda a, b: np.abs(a - b) / np.maximum(1.0, np.abs(b))
    day_err = rel(revenue, fx["revenue"][:, :days])
    same = (day_err <= 1e-6) & (np.abs(soc - fx["soc"][:, :days]) <= 0.011)
    first_split = [int(np.argmin(s)) if not s.all() else days for s in same]
    annual = rel(revenue.sum(1), fx["revenue"][:, :days].sum(1))
    energy = rel(delivered.sum(1), fx["delivered"][:, :days].sum(1))
    print(f"\n[year] {B} plants x {days} days in {wall:.1f} s ({1e3 * wall / days:.1f} ms per simulated day, first day eager); {hours} recorded hours "
          f"checked against the oracle in {check_wall:.0f} s: worst objective gap day-ahead {worst['da']:.2e}, real-time {worst['rt']:.2e}, tracking {worst['tr']:.2e}; "
          f"free-run fixture: {int(same.sum())} of {same.size} plant-days agree to 1e-6 (first differing day per plant {first_split}), annual revenue within "
          f"{annual.max():.2e}, delivered energy within {energy.max():.2e}")
    assert annual.max() <= 6e-3 and energy.max() <= 5e-4, (annual, energy)


def _year_block(arg):
    from tests._rolling_oracle import check_recorded_plant
    (k, j, d0, d1, part, h0), maps = arg
    # re-base the block: the oracle check indexes hours / days from the plant's first recorded hour
    lead = 1 if h0 > 0 else 0
    return check_recorded_plant((k, 17, maps, part, range(h0, 24 * d1)), base_hour=h0 - lead, base_day=d0)


@gpu
@pytest.mark.parametrize("flowsheet", ["nuclear", "wind_pem"])
def test_generic_double_loop_hours_are_optimal_for_the_oracles_lps(flowsheet):
    """BASELINE config 2 is a nuclear DOUBLE LOOP: the device-resident loop over a descriptor of the flowsheet's rolling state
    (dispatches_amd/rolling_flowsheets.py) for the nuclear (tank holdup handed on as round(holdup[-1]), 12-h real-time horizon) and
    wind + PEM (capacity-factor shift only) flowsheets - every hourly objective of the first hours against the oracle's own un-reduced
    LPs of the loop's state (HiGHS), 1e-6."""
    from dispatches_amd.rolling_flowsheets import BatchedDoubleLoop
    from tests._rolling_oracle import check_flowsheet_hours_against_the_oracle
    loop = BatchedDoubleLoop(flowsheet, 6, device=0)
    worst = check_flowsheet_hours_against_the_oracle(loop, 5)
    assert worst <= 1e-6
    print(f"\n[{flowsheet}] worst hourly objective gap against the oracle {worst:.2e}")


@gpu
@pytest.mark.parametrize("flowsheet", ["wind_battery", "nuclear", "wind_pem"])
def test_batched_double_loop_of_every_flowsheet_matches_host_objects(flowsheet):
    """The generic device loop against the product's own host objects (Bidder / Tracker on the flowsheet's model object, one scenario each,
    the reference's call order: real-time bid -> tracking -> update_model on tracker and bidder): delivered power and realised state of
    the first hours of a day; and the graph-replayed days equal the eagerly issued ones."""
    import torch
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    from dispatches_amd.rolling_flowsheets import BatchedDoubleLoop, _templates
    from dispatches_amd.workflow import Bidder, Tracker
    B, hours = 3, 5
    loop = BatchedDoubleLoop(flowsheet, B, device=0)
    offers = loop.day_ahead().cpu().numpy()
    da_prices = loop.da_prices.cpu().numpy()
    dev = dict(delivered=[], state=[])
    for _ in range(hours):
        dev["delivered"].append(loop.hour_step().cpu().numpy())
        dev["state"].append(loop.state.cpu().numpy().copy())
    assert loop.results()[1]
    da_s, rt_s = loop.da_series.cpu().numpy(), loop.rt_series.cpu().numpy()
    cf_s = loop.cf_series.cpu().numpy() if loop.cf_series is not None else None
    for k in range(B):
        start = int(loop.start[k].item())
        fc = scenarios.WindowForecaster(da_s, rt_s, [start])
        tmpl = _templates(flowsheet, 48, 4)[0].bidding_model_object
        if flowsheet == "nuclear":
            mk = lambda: tmpl.__class__(tmpl.model_data)
        elif flowsheet == "wind_pem":
            mk = lambda: tmpl.__class__(tmpl.model_data, wind_capacity_factors=list(np.roll(cf_s, -start)), wind_pmax_mw=tmpl._wind_pmax_mw, pem_pmax_mw=tmpl._pem_pmax_mw)
        else:
            mk = lambda: tmpl.__class__(model_data=tmpl.model_data, wind_capacity_factors=list(np.roll(cf_s, -start)), wind_pmax_mw=200.0,
                                        battery_pmax_mw=25.0, battery_energy_capacity_mwh=100.0)
        bidder = Bidder(mk(), day_ahead_horizon=48, real_time_horizon=loop.rt.T, n_scenario=1, solver=HipPdlpSolver(device=0), forecaster=fc)
        tracker = Tracker(tracking_model_object=mk(), tracking_horizon=4, n_tracking_hour=1, solver=HipPdlpSolver(device=0))
        for h in range(hours):
            bidder.compute_real_time_bids("2020-01-02", h, list(da_prices[k]), list(offers[k]))
            dispatch = [float(v) for v in bidder.real_time_model.expression_values("P_T")[0]][:4]
            prof = tracker.track_market_dispatch(market_dispatch=dispatch, date="2020-01-02", hour=h)
            delivered = tracker.get_last_delivered_power()
            tracker.update_model(**prof)
            bidder.update_real_time_model(**prof)
            assert dev["delivered"][h][k] == pytest.approx(delivered, abs=1e-6 * 1000), (flowsheet, k, h)
            for j, (key, scale) in enumerate(zip(prof, loop.scale)):
                assert dev["state"][h][k, j] == pytest.approx(round(prof[key][-1], int(round(np.log10(scale)))), abs=1.01 / scale), (flowsheet, k, h, key)
    # graphs: three days replayed = three days issued eagerly
    res = {}
    for graphs in (False, True):
        l2 = BatchedDoubleLoop(flowsheet, 64, device=0, use_graphs=graphs)
        for _ in range(3):
            l2.run_day()
        torch.cuda.synchronize()
        res[graphs] = (l2.revenue.cpu().numpy().copy(), l2.state.cpu().numpy().copy(), l2.results()[1])
    assert res[True][2] and res[False][2]
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])


@gpu
@pytest.mark.parametrize("flowsheet", ["wind_battery", "nuclear", "wind_pem"])
def test_generic_fused_update_kernel_matches_the_tensor_operations(flowsheet):
    """dsp_loop_update (include/dsp_hip.h: the hour step's hand-off for a flowsheet given by a descriptor, three launches) against the ~100
    tensor operations it replaces, two simulated days of 128 plants: the same realised states, delivered energy and revenue (sums and
    products are associated differently: 1e-12), every solve optimal."""
    import torch
    from dispatches_amd.rolling_flowsheets import BatchedDoubleLoop
    res = {}
    for fused in (False, True):
        loop = BatchedDoubleLoop(flowsheet, 128, device=0, use_fused=fused)
        assert loop.use_fused == fused
        for _ in range(2):
            loop.run_day()
        torch.cuda.synchronize()
        out, ok = loop.results()
        assert ok and int(loop.hour_t.item()) == 48
        res[fused] = {k: v.cpu().numpy().copy() for k, v in out.items()}
        res[fused]["c0"] = [m.c0.cpu().numpy().copy() for m in (loop.rt, loop.tr)]
    assert np.array_equal(res[True]["state"], res[False]["state"])
    for k in ("obj", "energy_mwh"):
        assert np.allclose(res[True][k], res[False][k], rtol=1e-12, atol=1e-9), k
    for a, b in zip(res[True]["c0"], res[False]["c0"]):
        assert np.allclose(a, b, rtol=1e-12)
    assert np.abs(res[True]["obj"]).max() > 0


@gpu
@pytest.mark.parametrize("which", ["specialised", "generic"])
def test_fused_update_kernels_report_a_solve_that_did_not_finish(which):
    """The fused update kernels fold the outcome of the hourly solves into the loop's device-side flags (status != optimal -> `bad`; flagged
    DSP_FLAG_OBJ_WAIVED -> `uncertified`) instead of six tensor launches per solve: a tracking LP with crossed bounds on one plant (status 2:
    invalid input) must make `results()` report it, on both loops."""
    from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
    from dispatches_amd.rolling_flowsheets import BatchedDoubleLoop
    loop = BatchedWindBatteryDoubleLoop(8, device=0, use_graphs=False) if which == "specialised" else BatchedDoubleLoop("nuclear", 8, device=0, use_graphs=False)
    assert loop.use_fused
    loop.day_ahead()
    loop.hour_step()
    assert loop.results()[1]
    import torch
    col = int(loop.tr.pt_cols[3, 0]) if which == "specialised" else int(torch.nonzero(loop.tr.PT[3])[0, 0])      # a column the update never rewrites
    loop.tr.lb[5, col], loop.tr.ub[5, col] = 1e9, 0.0
    loop.hour_step()
    assert int(loop.tr.out["status"][5].item()) != 0
    assert not loop.results()[1]
