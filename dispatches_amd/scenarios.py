"""Synthetic RTS-GMLC scenario batches for the BASELINE.json configurations.

Every scenario k is one independent day-ahead bidding LP of a model object; scenarios share the constraint
matrix and differ in prices (objective) and, for the wind cases, in the capacity-factor window (column bounds +
objective constant).  Price / CF series come from the RTS-GMLC extracts under ``dispatches_amd/data``
(produced by ``tools/extract_reference_data.py``); no dataset is downloaded.
"""
from __future__ import annotations

import os

import numpy as np

from .flowsheets import MultiPeriodNuclear, MultiPeriodWindBattery, MultiPeriodWindPEM
from .workflow import Bidder, ThermalGeneratorModelData
from .workflow.forecaster import AbstractPrescientPriceForecaster

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_series(name):
    d = np.load(os.path.join(_DATA, name))
    return {k: d[k] for k in d.files}


class WindowForecaster(AbstractPrescientPriceForecaster):
    """Scenario i = the horizon-long window of a long DA/RT price series starting at start_hours[i]."""

    def __init__(self, da, rt, start_hours, clip=(0.0, 500.0)):
        self.da = np.clip(np.asarray(da, float), *clip)
        self.rt = np.clip(np.asarray(rt, float), *clip)
        self.start_hours = np.asarray(start_hours, int)

    def windows(self, series, hour, horizon, n_samples):
        # rows of a sliding-window view over the series continued by its own start (the windows wrap around the year): one gather of
        # n_samples rows instead of an [n_samples, horizon] index array, a modulo and an element-wise gather
        # (the cache entry keeps the series it was built from and is used only for that very object: an id() alone is reused by a later
        #  array; a horizon longer than the series wraps more than once: np.resize tiles it)
        cache = self.__dict__.setdefault("_views", {})
        hit = cache.get((id(series), horizon))
        if hit is None or hit[0] is not series:
            ext = np.resize(series, len(series) + horizon - 1) if horizon > 1 else series
            hit = cache[(id(series), horizon)] = (series, np.lib.stride_tricks.sliding_window_view(ext, horizon))
        return hit[1][(self.start_hours[:n_samples] + int(hour)) % len(series)]

    def forecast_day_ahead_and_real_time_prices(self, date, hour, bus, horizon, n_samples):
        return self.windows(self.da, hour, horizon, n_samples), self.windows(self.rt, hour, horizon, n_samples)

    def forecast_day_ahead_prices(self, date, hour, bus, horizon, n_samples):
        return self.windows(self.da, hour, horizon, n_samples)

    def forecast_real_time_prices(self, date, hour, bus, horizon, n_samples):
        return self.windows(self.rt, hour, horizon, n_samples)

    def fetch_hourly_stats_from_prescient(self, prescient_hourly_stats):
        pass

    def fetch_day_ahead_stats_from_prescient(self, uc_date, uc_hour, day_ahead_result):
        pass


def _thermal_data(gen, bus, p_max, extra):
    return ThermalGeneratorModelData(
        gen_name=gen, bus=bus, p_min=0, p_max=p_max, min_down_time=0, min_up_time=0,
        ramp_up_60min=p_max + extra, ramp_down_60min=p_max + extra, shutdown_capacity=p_max + extra,
        startup_capacity=0, initial_status=1, initial_p_output=0,
        production_cost_bid_pairs=[(0, 0), (p_max, 0)], include_default_p_cost=False,
        startup_cost_pairs=[(0, 0)], fixed_commitment=None)


def _apply_cf_windows(model, wind_cols, wind_kw, cf_windows, cf_template, waste_cost_per_kw):
    """Per-scenario wind availability: column upper bounds + the objective constant of the curtailment term."""
    B = model.n_scenario
    lb, ub, _, _ = model.block.current_bounds()
    model.ub = np.tile(ub, (B, 1))
    model.ub[:, wind_cols] = wind_kw * cf_windows
    # tot_cost carries  waste_cost_per_kw * wind_kw * cf_t  as a constant: swap the template's for the window's
    model.c0_shift = waste_cost_per_kw * wind_kw * (cf_windows.sum(1) - np.sum(cf_template))


def wind_battery_batch(B, T, solver, series="rts_gmlc_309.npz", stride=17, wind_mw=200.0, batt_mw=25.0,
                       price_cap=500.0, ramp_cost=0.0, throughput_nodes=0):
    """LP #1 (wind + battery) day-ahead bidding, B scenarios x T hours (BASELINE metric workload: T=24, B=4096;
    config 4 shape: T=48, bus 309, start hours (17 k) mod (N - T))."""
    s = load_series(series)
    N = len(s["rt_lmp"])
    starts = (stride * np.arange(B)) % (N - T)
    fc = WindowForecaster(s["da_lmp"], s["rt_lmp"], starts, clip=(0.0, price_cap))
    mp = MultiPeriodWindBattery(_thermal_data("309_WIND_1", "Carter", wind_mw, batt_mw),
                                wind_capacity_factors=list(s["rt_cf"]), wind_pmax_mw=wind_mw,
                                battery_pmax_mw=batt_mw, battery_energy_capacity_mwh=4 * batt_mw, throughput_nodes=throughput_nodes)
    bidder = Bidder(mp, day_ahead_horizon=T, real_time_horizon=4, n_scenario=B, solver=solver, forecaster=fc,
                    ramp_cost=ramp_cost)
    model = bidder.day_ahead_model
    cfw = fc.windows(s["rt_cf"], 0, T, B)
    wind_cols = np.array([p["wind"].index for p in model.block.windBattery["periods"]])
    _apply_cf_windows(model, wind_cols, wind_mw * 1e3, cfw, s["rt_cf"][:T], mp.wind_waste_penalty * 1e-3)
    return bidder, model


def wind_pem_batch(B, T, solver, series="rts_gmlc_303.npz", stride=37, wind_mw=847.0, pem_mw=211.75,
                   price_cap=500.0):
    """LP #2 (wind + PEM), BASELINE config 3: 4096 x 48 h, bus 303 windows at (37 k) mod 8736."""
    s = load_series(series)
    N = len(s["rt_lmp"])
    starts = (stride * np.arange(B)) % (N - T)
    fc = WindowForecaster(s["da_lmp"], s["rt_lmp"], starts, clip=(0.0, price_cap))
    mp = MultiPeriodWindPEM(_thermal_data("303_WIND_1", "Caesar", wind_mw, pem_mw),
                            wind_capacity_factors=list(s["rt_cf"]), wind_pmax_mw=wind_mw, pem_pmax_mw=pem_mw)
    bidder = Bidder(mp, day_ahead_horizon=T, real_time_horizon=4, n_scenario=B, solver=solver, forecaster=fc)
    model = bidder.day_ahead_model
    cfw = fc.windows(s["rt_cf"], 0, T, B)
    wind_cols = np.array([p["wind"].index for p in model.block.windPEM["periods"]])
    _apply_cf_windows(model, wind_cols, wind_mw * 1e3, cfw, s["rt_cf"][:T], 1.0)
    return bidder, model


def nuclear_prices(B, T, seed=2020):
    """(DA, RT) price matrices [B, T] of the nuclear workloads: first B of the 3100 day-signals as DA prices,
    RT = DA * (1 + 0.1 N(0,1)) clipped at 0 (one seeded draw for the whole batch)."""
    lmp = load_series("nuclear_lmp_signal.npz")["lmp"]
    rng = np.random.default_rng(seed)
    da = lmp[np.arange(B) % len(lmp)]
    da = np.tile(da, (1, (T + 23) // 24))[:, :T]
    rt = np.clip(da * (1 + 0.1 * rng.standard_normal(da.shape)), 0, None)
    return da, rt


def nuclear_batch(B, T, solver, seed=2020):
    """LP #3 (nuclear + PEM + tank), BASELINE config 2."""
    da, rt = nuclear_prices(B, T, seed)

    class _Fixed(AbstractPrescientPriceForecaster):
        def forecast_day_ahead_and_real_time_prices(self, date, hour, bus, horizon, n_samples):
            return da[:n_samples, :horizon], rt[:n_samples, :horizon]

        def fetch_hourly_stats_from_prescient(self, s):
            pass

        def fetch_day_ahead_stats_from_prescient(self, *a):
            pass

    md = ThermalGeneratorModelData(
        gen_name="121_NUCLEAR_1", bus="Attlee", p_min=400, p_max=500, min_down_time=48, min_up_time=24,
        ramp_up_60min=100, ramp_down_60min=100, shutdown_capacity=500, startup_capacity=500, initial_status=-1,
        initial_p_output=0, production_cost_bid_pairs=[(400, 15), (450, 17.5), (500, 20)],
        startup_cost_pairs=[(48, 7355.42)], fixed_commitment=1)
    bidder = Bidder(MultiPeriodNuclear(md), day_ahead_horizon=T, real_time_horizon=min(12, T), n_scenario=B,
                    solver=solver, forecaster=_Fixed())
    return bidder, bidder.day_ahead_model


def load_prices(bidder, model, date="2020-01-02", hour=0):
    """Fill model.c / model.c0 for every scenario from the bidder's forecaster WITHOUT solving (bench / tests)."""
    bus = bidder.bidding_model_object.model_data.bus
    da, rt = bidder.forecaster.forecast_day_ahead_and_real_time_prices(
        date=date, hour=hour, bus=bus, horizon=len(model.HOUR), n_samples=model.n_scenario)
    da = bidder._as_matrix(da, model.n_scenario, len(model.HOUR))
    rt = bidder._as_matrix(rt, model.n_scenario, len(model.HOUR))
    bidder._pass_price_forecasts(model, da, rt)
    return model


WORKLOADS = {
    # name: (builder, kwargs)   -- BASELINE.json metric + configs
    "wind_battery_24h": (wind_battery_batch, dict(T=24)),        # metric workload: RTS-GMLC 24 h, batch 4096
    "wind_battery_48h": (wind_battery_batch, dict(T=48)),        # config 4 per-GPU shape
    "wind_pem_48h": (wind_pem_batch, dict(T=48)),                # config 3
    "nuclear_24h": (nuclear_batch, dict(T=24)),                  # config 2
    "nuclear_48h": (nuclear_batch, dict(T=48)),
}
# BASELINE config 5: the metric LP with a quadratic ramp cost rho/2 sum (P_T[t] - P_T[t-1])^2 (convex QP: T - 1 soft rows)
QP_WORKLOADS = {f"wind_battery_24h_qp{tag}": (wind_battery_batch, dict(T=24, ramp_cost=rho))
                for tag, rho in (("001", 0.01), ("01", 0.1), ("1", 1.0))}


# NOT part of the measured workloads (no oracle fixtures, no ahead-of-time kernel specialisation): the bidding LPs with the two-level
# form of the battery's throughput accumulator (flowsheets/units.py::two_level_accumulator) - same optima, fewer PDHG iterations in
# the lab; for the labs (tools/pdlp_lab.py) and the first GPU measurements of the next round
EXPERIMENTAL_WORKLOADS = {
    "wind_battery_24h_tl2": (wind_battery_batch, dict(T=24, throughput_nodes=2)),
    "wind_battery_48h_tl2": (wind_battery_batch, dict(T=48, throughput_nodes=2)),
}


def make_batch(name, B, solver):
    fn, kw = (WORKLOADS.get(name) or QP_WORKLOADS.get(name) or EXPERIMENTAL_WORKLOADS[name])
    bidder, model = fn(B=B, solver=solver, **kw)
    load_prices(bidder, model)
    return bidder, model


# ---- the HOURLY LPs of the double loop at batch scale (real-time bids and tracking) ----------------
# One simulated day is 1 day-ahead solve + 24 x (real-time bid + tracking) solves per scenario: 24 of the 25 solves are
# these small LPs.  Each scenario carries its own rolling-horizon state (initial SOC / throughput / tank holdup, fixed
# through column bounds exactly as `update_model` does), capacity-factor window and realised dispatch.
class _MatrixForecaster(AbstractPrescientPriceForecaster):
    def __init__(self, da, rt):
        self.da, self.rt = np.asarray(da, float), np.asarray(rt, float)

    def forecast_day_ahead_and_real_time_prices(self, date, hour, bus, horizon, n_samples):
        return self.da[:n_samples, :horizon], self.rt[:n_samples, :horizon]

    def fetch_hourly_stats_from_prescient(self, s):
        pass

    def fetch_day_ahead_stats_from_prescient(self, *a):
        pass


def _per_scenario_bounds(model):
    B = model.n_scenario
    lb, ub, _, _ = model.block.current_bounds()
    return np.tile(lb, (B, 1)), np.tile(ub, (B, 1))


def _nuclear_model_data():
    return ThermalGeneratorModelData(
        gen_name="121_NUCLEAR_1", bus="Attlee", p_min=400, p_max=500, min_down_time=48, min_up_time=24,
        ramp_up_60min=100, ramp_down_60min=100, shutdown_capacity=500, startup_capacity=500, initial_status=-1,
        initial_p_output=0, production_cost_bid_pairs=[(400, 15), (450, 17.5), (500, 20)],
        startup_cost_pairs=[(48, 7355.42)], fixed_commitment=1)


def _hourly_model_object(case, T, cf_template):
    if case.startswith("wind_battery"):
        return MultiPeriodWindBattery(_thermal_data("309_WIND_1", "Carter", 200.0, 25.0),
                                      wind_capacity_factors=list(cf_template), wind_pmax_mw=200.0,
                                      battery_pmax_mw=25.0, battery_energy_capacity_mwh=100.0)
    if case.startswith("wind_pem"):
        return MultiPeriodWindPEM(_thermal_data("309_WIND_1", "Carter", 200.0, 25.0),
                                  wind_capacity_factors=list(cf_template), wind_pmax_mw=200.0, pem_pmax_mw=25.0)
    return MultiPeriodNuclear(_nuclear_model_data())


def _apply_hourly_state(case, model, inp, lb, ub):
    """Per-scenario rolling-horizon state and wind availability as column bounds (what `update_model` writes)."""
    blk = model.block
    if case.startswith("wind"):
        fam = blk.windBattery if case.startswith("wind_battery") else blk.windPEM
        wind_cols = np.array([p["wind"].index for p in fam["periods"]])
        ub[:, wind_cols] = fam["wind_kw"] * inp["cf"]
        if case.startswith("wind_battery"):
            for key, col in (("soc0", fam["soc_init"].index), ("e0", fam["thr_init"].index)):
                lb[:, col] = ub[:, col] = inp[key]
        cf_t = np.array([p["wind"].ub for p in fam["periods"]]) / fam["wind_kw"]
        per_kw = 1.0 if case.startswith("wind_pem") else 1e-3 * 1e3          # curtailment weight per kW
        return per_kw * fam["wind_kw"] * (inp["cf"].sum(1) - cf_t.sum())      # objective-constant shift per scenario
    col = blk.nuclear["holdup_init"].index
    lb[:, col] = ub[:, col] = inp["holdup0"]
    return np.zeros(model.n_scenario)


def hourly_bid_batch(case, inp, solver):
    """Real-time bidding LPs (`case` = wind_battery_rt4 / wind_pem_rt4 / nuclear_rt12) for B scenarios: day_ahead_power
    fixed to each scenario's realised dispatch.  Returns (bidder, model); model.objective follows the product's
    convention (it keeps the constant DA revenue sum_t DA_t * dispatch_t, the oracle's RT objective does not)."""
    B, T = inp["rt"].shape
    mo = _hourly_model_object(case, T, inp["cf"][0] if "cf" in inp else None)
    bidder = Bidder(mo, day_ahead_horizon=T, real_time_horizon=T, n_scenario=B, solver=solver,
                    forecaster=_MatrixForecaster(inp["da"], inp["rt"]))
    model = bidder.real_time_model
    lb, ub = _per_scenario_bounds(model)
    model.c0_shift = _apply_hourly_state(case, model, inp, lb, ub)
    lb[:, model.pda_cols] = ub[:, model.pda_cols] = inp["dispatch"]
    model.lb, model.ub = lb, ub
    bidder._pass_price_forecasts(model, np.asarray(inp["da"], float), np.asarray(inp["rt"], float))
    return bidder, model


def hourly_tracking_batch(case, inp, solver):
    """Tracking LPs (`case` = wind_battery_track4 / wind_pem_track4 / nuclear_track4) for B scenarios, each with its own
    dispatch signal and state.  Returns (tracker, model)."""
    from .workflow import Tracker
    B, T = inp["dispatch"].shape
    mo = _hourly_model_object(case, T, inp["cf"][0] if "cf" in inp else None)
    tracker = Tracker(tracking_model_object=mo, tracking_horizon=T, n_tracking_hour=1, solver=solver)
    model = tracker.model
    tracker._pass_market_dispatch([0.0] * T)                    # rows become equalities; the values are per scenario
    model.n_scenario, model.SCENARIOS = B, range(B)
    lb, ub = _per_scenario_bounds(model)
    shift = _apply_hourly_state(case, model, inp, lb, ub)
    model.lb, model.ub = lb, ub
    _, _, rlo, rhi = model.block.current_bounds()
    rlo, rhi = np.tile(rlo, (B, 1)), np.tile(rhi, (B, 1))
    rows = np.array([model.block.kept_row_index(r) for r in model.tracking_rows])
    rlo[:, rows] = rhi[:, rows] = inp["dispatch"] - model.PT_const[None, :]
    model.rlo, model.rhi = rlo, rhi
    model.c = np.tile(model.c[0], (B, 1))
    model.c0 = np.full(B, float(model.c0[0])) + shift
    return tracker, model


# ---- long-horizon price-taker design LPs: a scenario family sharing one constraint matrix ---------------
PRICE_TAKER_FAMILY = [(bf, lm) for lm in (1.0, 1.5, 2.0, 3.0) for bf in (1.0, 0.5, 0.25, 0.1)]   # (battery capital-cost factor, LMP multiplier)
# 256 DISTINCT members (round 6: a 256-scenario batch of the 16-member family is 16 LPs 16 times over - every 64-lane group then holds
# every member, and the distribution of Newton iterations is that of 16 problems): the 16 above first (their fixtures stay valid), then
# 15 LMP multipliers 1.15 .. 3.25 x 16 battery capital-cost factors 1 .. 0.104, none of which coincides with a member above
PRICE_TAKER_FAMILY_WIDE = PRICE_TAKER_FAMILY + [(round(0.86 ** j, 6), round(1.0 + 0.15 * k, 2)) for k in range(1, 16) for j in range(16)]
assert len(set(PRICE_TAKER_FAMILY_WIDE)) == 256


def price_taker_inputs(T, series="rts_gmlc_303.npz", price_cap=200.0):
    """Capacity factors and day-ahead LMPs of the first T hours of the bus-303 series (LMPs capped at 200 $/MWh as the
    reference's test fixture does, tests/test_RE_flowsheet.py:24-26)."""
    s = load_series(series)
    idx = np.arange(T) % len(s["da_lmp"])
    return s["rt_cf"][idx], np.minimum(s["da_lmp"][idx], price_cap)


def price_taker_reference_inputs(T):
    """Capacity factors and LMPs of the reference's OWN price-taker tests (tests/test_RE_flowsheet.py:22-43): the first T hourly wind
    speeds of its Wind Toolkit SRW file through the wind resource model (flowsheets/wind_resource.py), the first T day-ahead LMPs of
    rts_results_all_prices.npy capped at 200 $/MWh.  Data: dispatches_amd/data/price_taker_inputs.npz (tools/extract_reference_data.py)."""
    from .flowsheets.wind_resource import capacity_factor_from_speed
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "price_taker_inputs.npz"))
    return capacity_factor_from_speed(d["wind_speed_m_s"][:T]), np.minimum(d["da_lmp"][:T], 200.0)


# (hydrogen price $/kg, PEM capital-cost factor): the wind + battery + PEM design family; member 1 = the reference test's setting
PEM_PRICE_TAKER_FAMILY = [(h2, pf) for pf in (1.0, 0.75, 0.5, 1.25) for h2 in (2.0, 2.5, 3.0, 4.0)]


def pem_price_taker_batch(T, B, solver, design_opt=True, inputs="reference", throughput="chain", coarse_nodes=1):
    """Wind + battery + PEM price-taker design LP over T hourly periods (reference wind_battery_pem_optimize) for the first B members
    of PEM_PRICE_TAKER_FAMILY: scenarios differ in the objective only (hydrogen price, PEM capital cost).  Returns (handles, model)."""
    from .flowsheets.price_taker import wind_battery_pem_price_taker
    from .workflow.batch_model import ScenarioBatchModel
    cf, lmp = price_taker_reference_inputs(T) if inputs == "reference" else price_taker_inputs(T)
    block, objective, handles = wind_battery_pem_price_taker(T, cf, lmp, design_opt=design_opt, throughput=throughput, coarse_nodes=coarse_nodes)
    model = ScenarioBatchModel(block, B, T, indexed=True)
    model.finalize(objective)
    fam = [PEM_PRICE_TAKER_FAMILY[i % len(PEM_PRICE_TAKER_FAMILY)] for i in range(B)]
    model.c = np.stack([handles["objective_vector"](model.lp.n, h2_price=h2, pem_cap_factor=pf) for h2, pf in fam])
    model.c0 = np.full(B, model.lp.c0)
    model.lp.col_scale = handles["column_scales"](model.lp.n)
    model.family = fam
    model.solver = solver
    return handles, model


def nuclear_price_taker_batch(T, B, solver, market="RT", pem_capex=1200.0):
    """The nuclear + PEM price-taker enumeration of the reference (nuclear_case/report/price_taker_analysis.py:353-419) as ONE batch:
    scenario k = (hydrogen price, PEM capacity) point k of the 6 x 10 grid (hydrogen price outermost, as the reference's loops),
    all sharing the constraint matrix; they differ in the objective (hydrogen price) and in the bounds of the one fixed design column
    `pem_capacity`.  T hourly periods of the bus-Attlee LMPs (`market`: "RT", "DA" or "Max-DA-RT", :45-113; the reference uses all
    8784).  Returns (handles, model)."""
    from .flowsheets.price_taker import NUCLEAR_H2_PRICES, NUCLEAR_PEM_FRACTIONS, NP_CAPACITY_MW, nuclear_price_taker
    from .workflow.batch_model import ScenarioBatchModel
    d = load_series("nuclear_price_taker_lmps.npz")
    lmp = {"RT": d["rt_lmp"], "DA": d["da_lmp"], "Max-DA-RT": np.maximum(d["rt_lmp"], d["da_lmp"])}[market][:T]
    block, objective, handles = nuclear_price_taker(T, lmp, pem_capex=pem_capex)
    model = ScenarioBatchModel(block, B, T, indexed=True)
    model.finalize(objective)
    grid = [(hp, pc) for hp in NUCLEAR_H2_PRICES for pc in NUCLEAR_PEM_FRACTIONS]
    fam = [grid[i % len(grid)] for i in range(B)]
    model.c = np.stack([handles["objective_vector"](model.lp.n, hp) for hp, _ in fam])
    model.c0 = np.full(B, model.lp.c0)
    lb, ub, _, _ = block.current_bounds()
    model.lb, model.ub = np.tile(lb, (B, 1)), np.tile(ub, (B, 1))
    j = handles["pem_capacity"].index
    model.lb[:, j] = model.ub[:, j] = [pc * NP_CAPACITY_MW for _, pc in fam]
    model.lp.col_scale = handles["column_scales"](model.lp.n)
    model.family, model.lmp = fam, lmp
    model.solver = solver
    return handles, model


def price_taker_batch(T, B, solver, wind_mw=847.0, throughput="two_level", inputs="rts303", coarse_nodes=3, family="base"):
    """Wind + battery price-taker design LP over T hourly periods (reference wind_battery_optimize) for the first B members
    of PRICE_TAKER_FAMILY: scenarios differ in the objective only.  n = 6 T + 3, m = 6 T + 2: beyond the fused kernels for
    T >= 107, i.e. solved by the HBM-resident streaming PDLP.  `throughput`: the statement of the battery's accumulated-throughput
    chain (flowsheets/price_taker.py) - "two_level" (default since round 4: `coarse_nodes` node values + local deviations, an exact
    change of variables with the reference's optima, 5 x fewer PDHG iterations at the year-long horizon) or "chain" (the reference's
    own linked columns).  Returns (handles, model)."""
    from .flowsheets.price_taker import wind_battery_price_taker
    from .workflow.batch_model import ScenarioBatchModel
    cf, lmp = price_taker_reference_inputs(T) if inputs == "reference" else price_taker_inputs(T)
    block, objective, handles = wind_battery_price_taker(T, cf, lmp, wind_mw=wind_mw, throughput=throughput, coarse_nodes=coarse_nodes)
    model = ScenarioBatchModel(block, B, T, indexed=True)
    model.finalize(objective)
    members = {"base": PRICE_TAKER_FAMILY, "wide": PRICE_TAKER_FAMILY_WIDE}[family]
    fam = [members[i % len(members)] for i in range(B)]
    model.c = np.stack([handles["objective_vector"](model.lp.n, lmp_multiplier=lm, batt_cap_factor=bf) for bf, lm in fam])
    model.c0 = np.full(B, model.lp.c0)
    model.lp.col_scale = handles["column_scales"](model.lp.n)       # physical scaling factors of the flowsheet's variables
    model.family = fam
    model.solver = solver
    return handles, model
