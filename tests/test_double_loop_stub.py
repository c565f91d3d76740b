"""End-to-end double loop against a STUB of Prescient's plugin context (Prescient itself is out of scope and absent):
DoubleLoopCoordinator.register_plugins -> DA bid -> RUC results -> RT bids -> SCED tracking -> next-day DA bid with the
projection tracker -> result files.  Call order and data shapes follow the reference's run_double_loop_battery.py:222-305
and the upstream coordinator callbacks.  The CPU run uses the test-only HiGHS stand-in
solver; the `gpu` run drives the same loop through libdsp_hip.so."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from dispatches_amd.flowsheets import MultiPeriodWindBattery
from dispatches_amd.workflow import (Backcaster, Bidder, DoubleLoopCoordinator, RenewableGeneratorModelData,
                                     SelfScheduler, ThermalGeneratorModelData, Tracker)
from tests.test_workflow_cpu import generator_params, thermal_params


# The registration methods of prescient.plugins.plugin_registration.PluginRegistrationContext (gridx-prescient 2.2:
# every `register_*_callback` the plugin API offers; SURVEY.md App. B, reference coordinator.py:29-40 uses three of
# them by name).  Anything else is a misspelt hook and must fail here as it would inside Prescient.
PRESCIENT_CALLBACK_HOOKS = (
    "options_preview", "initialization", "finalization",
    "after_get_initial_actuals_model_for_sced", "after_get_initial_actuals_model_for_simulation_actuals",
    "after_get_initial_forecast_model_for_ruc", "after_get_initial_forecast_model_for_simulation_actuals",
    "after_ruc_generation", "after_ruc_activation", "before_ruc_solve", "before_operations_solve",
    "after_operations", "update_operations_stats",
)


class _StubContext:
    """Collects what the coordinator registers, like prescient.plugins.PluginRegistrationContext: ONLY the hook names
    Prescient really has exist as methods; an unknown `register_*_callback` raises AttributeError."""

    def __init__(self):
        self.callbacks = {}

    def __getattr__(self, name):
        if name.startswith("register_") and name.endswith("_callback"):
            hook = name[len("register_"):-len("_callback")]
            if hook in PRESCIENT_CALLBACK_HOOKS:
                def reg(fn, _name=hook):
                    if not callable(fn):
                        raise TypeError(f"{name} needs a callable")
                    self.callbacks.setdefault(_name, []).append(fn)
                return reg
        raise AttributeError(f"PluginRegistrationContext has no attribute {name!r}")


def test_stub_context_rejects_unknown_hooks():
    ctx = _StubContext()
    with pytest.raises(AttributeError):
        ctx.register_before_ruc_solv_callback(lambda: None)
    with pytest.raises(AttributeError):
        ctx.register_after_sced_callback(lambda: None)
    ctx.register_before_ruc_solve_callback(lambda: None)
    assert list(ctx.callbacks) == ["before_ruc_solve"]


def _instance(gen):
    return SimpleNamespace(data={"elements": {"generator": {gen: {"generator_type": "thermal"}}}})


def _run_double_loop(solver_factory, tmp_path, rts309, thermal=True, throughput_nodes=0):
    md = ThermalGeneratorModelData(**thermal_params()) if thermal else RenewableGeneratorModelData(**generator_params)
    mk = lambda: MultiPeriodWindBattery(model_data=md, wind_capacity_factors=list(rts309["rt_cf"][:400]),
                                        wind_pmax_mw=200, battery_pmax_mw=25, battery_energy_capacity_mwh=100,
                                        throughput_nodes=throughput_nodes)
    fc = Backcaster({"Carter": rts309["da_lmp"][:48].tolist()}, {"Carter": rts309["rt_lmp"][:48].tolist()})
    cls = Bidder if thermal else SelfScheduler
    bidder = cls(bidding_model_object=mk(), day_ahead_horizon=24, real_time_horizon=4, n_scenario=1,
                 solver=solver_factory(), forecaster=fc)
    tracker = Tracker(tracking_model_object=mk(), tracking_horizon=4, n_tracking_hour=1, solver=solver_factory())
    projection = Tracker(tracking_model_object=mk(), tracking_horizon=4, n_tracking_hour=1, solver=solver_factory())
    coord = DoubleLoopCoordinator(bidder=bidder, tracker=tracker, projection_tracker=projection)

    ctx = _StubContext()
    plugin = coord.prescient_plugin_module
    assert "bidding_generator" in plugin.get_configuration("doubleloop")
    plugin.register_plugins(ctx, options=None, plugin_config=SimpleNamespace(bidding_generator=md.gen_name))
    for name in ("initialization", "before_ruc_solve", "before_operations_solve", "after_operations",
                 "update_operations_stats", "after_ruc_activation", "after_ruc_generation", "finalization",
                 "after_get_initial_actuals_model_for_sced", "after_get_initial_forecast_model_for_ruc",
                 "after_get_initial_actuals_model_for_simulation_actuals"):
        assert name in ctx.callbacks, name
    assert set(ctx.callbacks) <= set(PRESCIENT_CALLBACK_HOOKS)

    gen = md.gen_name
    options = SimpleNamespace(output_directory=str(tmp_path))
    simulator = SimpleNamespace(data_manager=SimpleNamespace(extensions={}),
                                time_manager=SimpleNamespace(current_time=SimpleNamespace(date="2020-01-02", hour=0)))
    ctx.callbacks["initialization"][0](options, simulator)

    # static generator parameters pushed into a RUC instance (reference coordinator.py:46-87)
    ruc = _instance(gen)
    ctx.callbacks["after_get_initial_forecast_model_for_ruc"][0](options, ruc)
    gd = ruc.data["elements"]["generator"][gen]
    assert gd["p_max"] == md.p_max and gd["bus"] == md.bus

    # day 0, day-ahead market
    bids = ctx.callbacks["before_ruc_solve"][0](options, simulator, ruc, "2020-01-02", 0)
    assert sorted(bids) == list(range(24))
    obj_day0 = float(bidder.day_ahead_model.objective[0])
    assert gd["p_max"]["data_type"] == "time_series" and len(gd["p_max"]["values"]) == 24
    p_da = [bids[t][gen]["p_max"] for t in range(24)]
    # RUC clears the bids at their maximum; prices = the backcast DA prices
    market = SimpleNamespace(day_ahead_prices={(md.bus, t): float(rts309["da_lmp"][t]) for t in range(24)},
                             thermal_gen_cleared_DA={(gen, t): p_da[t] for t in range(24)},
                             renewable_gen_cleared_DA={(gen, t): p_da[t] for t in range(24)})
    ctx.callbacks["after_ruc_generation"][0](options, simulator, SimpleNamespace(ruc_market=market), "2020-01-02", 0)
    ctx.callbacks["after_ruc_activation"][0](options, simulator)
    assert coord.current_DA_dispatches == p_da

    # real-time loop for three hours
    delivered = []
    for hour in range(3):
        simulator.time_manager.current_time.hour = hour
        sced = _instance(gen)
        rt_bids = ctx.callbacks["before_operations_solve"][0](options, simulator, sced)
        assert sorted(rt_bids) == [hour + k for k in range(4)]
        dispatch = [rt_bids[hour + k][gen]["p_max"] for k in range(4)]        # SCED takes the whole offer
        sced.data["elements"]["generator"][gen]["pg"] = {"data_type": "time_series", "values": dispatch}
        profiles = ctx.callbacks["after_operations"][0](options, simulator, sced, lmp_sced=None)
        assert set(profiles) == {"realized_soc", "realized_energy_throughput"}
        stats = SimpleNamespace(observed_thermal_dispatch_levels={}, observed_renewables_levels={},
                                observed_bus_LMPs={md.bus: float(rts309["rt_lmp"][hour])})
        ctx.callbacks["update_operations_stats"][0](options, simulator, stats)
        got = (stats.observed_thermal_dispatch_levels if thermal else stats.observed_renewables_levels)[gen]
        assert got == pytest.approx(dispatch[0], abs=2e-3)                      # tracker delivers the dispatch
        delivered.append(got)

    # day 1 day-ahead bid: goes through the projection tracker and update_day_ahead_model
    bids1 = ctx.callbacks["before_ruc_solve"][0](options, simulator, _instance(gen), "2020-01-03", 3)
    assert sorted(bids1) == list(range(24))
    ctx.callbacks["finalization"][0](options, simulator)
    for f in ("bidder_detail.csv", "tracker_detail.csv"):
        assert os.path.getsize(os.path.join(str(tmp_path), f)) > 0
    return np.array(p_da), np.array(delivered), obj_day0


def test_double_loop_stub_cpu(tmp_path, rts309):
    from tests._highs_solver import HighsTestSolver
    p_da, delivered, _ = _run_double_loop(HighsTestSolver, tmp_path, rts309)
    assert (p_da >= -1e-9).all() and (p_da <= 225 + 1e-6).all()


def test_double_loop_stub_with_the_two_level_throughput_accumulator(tmp_path, rts309):
    """The whole plugin sequence (bids, clearing, real-time hours through tracker and projection tracker, next day's bids after
    update_day_ahead_model) with the day-ahead LP in the two-level form of the battery's throughput accumulator
    (MultiPeriodWindBattery(throughput_nodes=2)): the same day-0 objective and offers as the reference's form."""
    from tests._highs_solver import HighsTestSolver
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    p0, d0, obj0 = _run_double_loop(HighsTestSolver, tmp_path / "a", rts309)
    p2, d2, obj2 = _run_double_loop(HighsTestSolver, tmp_path / "b", rts309, throughput_nodes=2)
    assert obj2 == pytest.approx(obj0, rel=1e-9, abs=1e-6)
    assert (p2 >= -1e-9).all() and (p2 <= 225 + 1e-6).all() and len(d2) == len(d0)


@pytest.mark.gpu
def test_double_loop_stub_gpu_matches_cpu(tmp_path, rts309):
    from dispatches_amd.hip_solver import HipPdlpSolver
    from tests._highs_solver import HighsTestSolver
    (tmp_path / "gpu").mkdir()
    (tmp_path / "cpu").mkdir()
    p_gpu, d_gpu, obj_gpu = _run_double_loop(lambda: HipPdlpSolver(device=0), tmp_path / "gpu", rts309)
    p_cpu, d_cpu, obj_cpu = _run_double_loop(HighsTestSolver, tmp_path / "cpu", rts309)
    # the day-0 bidding objective agrees to the parity bar; the hourly offers themselves may differ on the degenerate
    # (zero / equal price) hours, where the optimal face is not a point
    assert abs(obj_gpu - obj_cpu) <= 2e-6 * max(1.0, abs(obj_cpu))
    assert (p_gpu >= -1e-6).all() and (p_gpu <= 225 + 1e-6).all() and (d_gpu >= -1e-6).all()
