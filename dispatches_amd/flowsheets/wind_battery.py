"""Wind + battery multi-period model object for the double loop (LP #1 of SURVEY.md App. A.1).

API mirror of ``dispatches/case_studies/renewables_case/wind_battery_double_loop.py:101-352``
(`MultiPeriodWindBattery`): same constructor arguments, same protocol methods (`populate_model`,
`update_model`, `get_last_delivered_power`, `get_implemented_profile`, `record_results`, `write_results`,
`power_output`, `total_cost`), same units (kW / kWh inside, MW for `P_T`).  Instead of cloning a Pyomo
flowsheet per period, `populate_model` appends the rows of every period to ONE LinearBlock that is flattened
once and then only has bounds / right-hand sides rewritten between solves.
"""
from __future__ import annotations

from collections import deque

import numpy as np
import pandas as pd

from . import parameters as prm
from . import units


def create_multiperiod_wind_battery_model(b, n_time_points, wind_cfs, input_params):
    """Rows of `n_time_points` linked periods (reference: wind_battery_double_loop.py:26-51 ->
    wind_battery_LMP.py:106-169 -> RE_flowsheet.create_model; link pairs wind_battery_LMP.py:22-37)."""
    wind_kw = input_params["wind_mw"] * 1e3
    batt_kw = input_params["batt_mw"] * 1e3
    batt_kwh = input_params.get("batt_mwh", input_params["batt_mw"] * 4) * 1e3
    # initial conditions are columns so that update_model() only ever touches bounds
    soc_init = b.var("battery.initial_state_of_charge", 0.0, 0.0, mutable=True, hull=(0.0, batt_kwh))
    thr_init = b.var("battery.initial_energy_throughput", 0.0, np.inf, mutable=True, hull=(0.0, np.inf))
    periods = []
    soc_prev, thr_prev = soc_init, thr_init
    # OPTIONAL (input_params["throughput_nodes"] = K > 0, horizons beyond 8 periods): the accumulated throughput as K node values +
    # local deviations instead of one linked column per period (units.two_level_accumulator: an exact change of variables that
    # takes the slow mode of the chain out of the LP - numpy PDHG with the GPU's settings: 24 h 1213 -> 896 iterations, 48 h 2386 ->
    # 1360 with K = 2).  Off by default: the reference's form.
    nodes = int(input_params.get("throughput_nodes", 0) or 0)
    acc = units.two_level_accumulator(b, n_time_points, nodes, base=thr_init, prefix="battery.throughput") \
        if nodes > 0 and n_time_points > 8 else None
    for t in range(n_time_points):
        w = units.wind_power(b, t, wind_kw, wind_cfs[t])
        grid, to_batt = units.splitter(b, t, w, ("grid_elec", "battery_elec"))
        thr_t = None if acc is None else acc(t, lambda tt: b.var(f"battery.throughput_fine[{tt}]", -np.inf, np.inf))
        bat = units.battery(b, t, to_batt, soc_prev, thr_prev, batt_kw, batt_kwh,
                            prm.battery_charging_eta, prm.battery_discharging_eta,
                            prm.battery_degradation_rate, ramp_rate=prm.battery_ramp_rate, throughput=thr_t)
        periods.append(dict(wind=w, grid_elec=grid, elec_in=to_batt, thr_prev=thr_prev, **bat))
        soc_prev, thr_prev = bat["state_of_charge"], bat["energy_throughput"]
    return dict(periods=periods, soc_init=soc_init, thr_init=thr_init, wind_kw=wind_kw,
                batt_kw=batt_kw, batt_kwh=batt_kwh)


class MultiPeriodWindBattery(units.ResultRecords):
    # scaling hint for the HIP solver: column ranges implied by the bounds (lp.implied_column_ranges) - this LP mixes kW, MW and
    # (wind + battery) kWh of accumulated throughput; the reference sets IDAES scaling factors on the same variables
    column_scaling = "implied_ranges"

    def __init__(self, model_data, wind_capacity_factors=None, wind_pmax_mw=200.0, battery_pmax_mw=25.0,
                 battery_energy_capacity_mwh=100.0, throughput_nodes=0):
        self.model_data = model_data
        self._throughput_nodes = int(throughput_nodes)      # > 0: two-level form of the throughput accumulator (see above)
        if wind_capacity_factors is None:
            raise ValueError("Please provide wind capacity factors.")
        self._wind_capacity_factors = wind_capacity_factors
        self._wind_pmax_mw = wind_pmax_mw
        self._battery_pmax_mw = battery_pmax_mw
        self._battery_energy_capacity_mwh = battery_energy_capacity_mwh
        self.wind_waste_penalty = 1e3          # $/MWh, wind_battery_double_loop.py:172
        self.result_list = []

    # ------------------------------------------------------------------------------------------------------
    def populate_model(self, b, horizon):
        """Fill block `b` with the `horizon`-period wind+battery LP (reference :136-179)."""
        if not b.is_constructed():
            b.construct()
        cfs = list(self._wind_capacity_factors[0:horizon])
        b.windBattery = create_multiperiod_wind_battery_model(
            b, horizon, cfs,
            dict(wind_mw=self._wind_pmax_mw, batt_mw=self._battery_pmax_mw,
                 batt_mwh=self._battery_energy_capacity_mwh, throughput_nodes=self._throughput_nodes))
        b._time_idx = 0
        b.HOUR = range(horizon)
        self._write_expressions(b, cfs)

    def _write_expressions(self, b, cfs):
        """P_T [MW], wind_waste [MW], tot_cost [$]  (reference :169-177; O&M terms wind_battery_LMP.py:53-71)."""
        mp = b.windBattery
        fixed = mp["wind_kw"] * prm.wind_op_cost / 8760
        for t, p in enumerate(mp["periods"]):
            b.expression("P_T", t, (p["grid_elec"] + p["elec_out"]) * 1e-3)
            waste = (mp["wind_kw"] * cfs[t] - p["wind"]) * 1e-3
            b.expression("wind_waste", t, waste)
            var_cost = prm.battery_degradation_rate * (p["energy_throughput"] - p["thr_prev"]) * prm.batt_rep_cost_kwh
            b.expression("tot_cost", t, fixed + var_cost + self.wind_waste_penalty * waste)

    def update_model(self, b, realized_soc, realized_energy_throughput):
        """Rolling-horizon update (reference :181-209): re-fix the initial SOC / throughput to the last realised
        values ROUNDED TO 2 dp, advance the clock by min(len, 24) and load the new capacity factors."""
        mp = b.windBattery
        mp["soc_init"].fix(round(realized_soc[-1], 2))
        mp["thr_init"].fix(round(realized_energy_throughput[-1], 2))
        b._time_idx = b._time_idx + min(len(realized_soc), 24)
        cfs = self._get_capacity_factors(b)
        for p, cf in zip(mp["periods"], cfs):
            p["wind"].setub(mp["wind_kw"] * cf)
        self._write_expressions(b, cfs)

    def _get_capacity_factors(self, b):
        """Capacity factors of the next `horizon` hours, wrapping at the end of the data (reference :211-228)."""
        horizon_len = len(b.windBattery["periods"])
        ans = list(self._wind_capacity_factors[b._time_idx: b._time_idx + horizon_len])
        if len(ans) < horizon_len:
            ans += list(self._wind_capacity_factors[0:horizon_len - len(ans)])
        return ans

    @staticmethod
    def get_last_delivered_power(b, last_implemented_time_step):
        return b.value(b.P_T[last_implemented_time_step])

    @staticmethod
    def get_implemented_profile(b, last_implemented_time_step):
        """Realised SOC / throughput for t <= last implemented step (reference :245-274)."""
        per = b.windBattery["periods"]
        return {
            "realized_soc": deque(per[t]["state_of_charge"].value for t in range(last_implemented_time_step + 1)),
            "realized_energy_throughput": deque(
                (per[t]["energy_throughput"].value if hasattr(per[t]["energy_throughput"], "index") else b.value(per[t]["energy_throughput"]))
                for t in range(last_implemented_time_step + 1)),
        }

    def _result_columns(self, b):
        """One row per horizon hour, same column names as the reference (:276-335), built column-wise (units.ResultRecords)."""
        per = b.windBattery["periods"]
        T = len(per)
        x = np.asarray(b.solution)
        col = lambda key: x[..., [p[key].index for p in per]]
        return T, {
            "Total Wind Generation [MW]": np.round(col("wind") * 1e-3, 2),
            "Total Power Output [MW]": np.round(b.family_values("P_T")[..., :T], 2),
            "Wind Power Output [MW]": np.round(col("grid_elec") * 1e-3, 2),
            # the reference reports wind_waste[0] in every row (:312); kept for CSV compatibility
            "Wind Curtailment [MW]": self._round_scalar(b.family_values("wind_waste")[..., 0]),
            "Battery Power Output [MW]": np.round(col("elec_out") * 1e-3, 2),
            "Wind Power to Battery [MW]": np.round(col("elec_in") * 1e-3, 2),
            "State of Charge [MWh]": np.round(col("state_of_charge") * 1e-3, 2),
            "Total Cost [$]": np.round(b.family_values("tot_cost")[..., :T], 2),
        }

    @property
    def power_output(self):
        return "P_T"

    @property
    def total_cost(self):
        return ("tot_cost", 1)
