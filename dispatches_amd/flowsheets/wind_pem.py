"""Wind + PEM multi-period model object (LP #2 of SURVEY.md App. A.2).

API mirror of ``dispatches/case_studies/renewables_case/wind_PEM_double_loop.py:103-337`` (`MultiPeriodWindPEM`).
The battery branch of the reference flowsheet is forced to 0 MW (:38-39,153) and therefore contributes no
columns here; ``pem_system_capacity`` stays a FREE non-negative column exactly as in the reference (:76).
"""
from __future__ import annotations

from collections import deque

import numpy as np
import pandas as pd

from . import parameters as prm
from . import units


def create_multiperiod_wind_pem_model(b, n_time_points, wind_cfs, input_params):
    if input_params.get("batt_mw", 0) != 0:
        raise ValueError("This model is not used for hybrids with battery. Battery MW must be 0.")
    wind_kw = input_params["wind_mw"] * 1e3
    cap = b.var("pem_system_capacity", 0.0, np.inf)          # wind_PEM_double_loop.py:76
    periods = []
    for t in range(n_time_points):
        w = units.wind_power(b, t, wind_kw, wind_cfs[t])
        grid, pem = units.splitter(b, t, w, ("grid_elec", "pem_elec"))
        units.pem_electrolyzer(b, t, pem, cap)
        periods.append(dict(wind=w, grid_elec=grid, pem_elec=pem))
    return dict(periods=periods, pem_system_capacity=cap, wind_kw=wind_kw)


class MultiPeriodWindPEM(units.ResultRecords):
    # scaling hint for the HIP solver: column ranges implied by the bounds (lp.implied_column_ranges) - this LP mixes kW, MW and
    # (wind + battery) kWh of accumulated throughput; the reference sets IDAES scaling factors on the same variables
    column_scaling = "implied_ranges"

    # cadence hint for the HIP solver (include/dsp_hip.h check_every): this LP has no storage state, its active set is found
    # after a few hundred iterations and the ray jump (tested at checks) finishes it - checking every 12 iterations instead
    # of 16 takes 30 % off the iteration count (profiles/r04v_knob_scan2.log, r04w_cadence.log: 4.1 -> 5.3 M scenarios/s)
    bidding_solver_hints = {"check_every": 12, "eps_rel": 1e-10}

    def __init__(self, model_data, wind_capacity_factors, wind_pmax_mw=200.0, pem_pmax_mw=25.0):
        self.model_data = model_data
        if wind_capacity_factors is None:
            raise ValueError("Please provide wind capacity factors.")
        self._wind_capacity_factors = wind_capacity_factors
        self._wind_pmax_mw = wind_pmax_mw
        self._pem_pmax_mw = pem_pmax_mw
        self.result_list = []

    def populate_model(self, b, horizon):
        if not b.is_constructed():
            b.construct()
        cfs = list(self._wind_capacity_factors[0:horizon])
        b.windPEM = create_multiperiod_wind_pem_model(
            b, horizon, cfs, dict(wind_mw=self._wind_pmax_mw, pem_mw=self._pem_pmax_mw, batt_mw=0))
        b._time_idx = 0
        b.HOUR = range(horizon)
        self._write_expressions(b, cfs)

    def _write_expressions(self, b, cfs):
        """P_T = grid_elec*1e-3 [MW]; wind_waste in kW with unit weight; tot_cost (reference :172-182)."""
        mp = b.windPEM
        for t, p in enumerate(mp["periods"]):
            b.expression("P_T", t, p["grid_elec"] * 1e-3)
            waste = mp["wind_kw"] * cfs[t] - p["wind"]
            b.expression("wind_waste", t, waste)
            b.expression("tot_cost", t,
                         mp["wind_kw"] * prm.wind_op_cost / 8760
                         + mp["pem_system_capacity"] * (prm.pem_op_cost / 8760)
                         + p["pem_elec"] * prm.pem_var_cost
                         + waste)

    def update_model(self, b, realized_h2_sales):
        """Only the capacity-factor window advances (reference :185-204)."""
        mp = b.windPEM
        b._time_idx = b._time_idx + min(len(realized_h2_sales), 24)
        cfs = self._get_capacity_factors(b)
        for p, cf in zip(mp["periods"], cfs):
            p["wind"].setub(mp["wind_kw"] * cf)
        self._write_expressions(b, cfs)

    def _get_capacity_factors(self, b):
        horizon_len = len(b.windPEM["periods"])
        ans = list(self._wind_capacity_factors[b._time_idx: b._time_idx + horizon_len])
        if len(ans) < horizon_len:
            ans += list(self._wind_capacity_factors[0:horizon_len - len(ans)])
        return ans

    @staticmethod
    def get_last_delivered_power(b, last_implemented_time_step):
        return b.value(b.P_T[last_implemented_time_step])

    @staticmethod
    def _h2_kg_per_hr(pem_kw):
        return pem_kw * prm.pem_electricity_to_mol / prm.h2_mols_per_kg * 3600

    @staticmethod
    def get_implemented_profile(b, last_implemented_time_step):
        per = b.windPEM["periods"]
        return {"realized_h2_sales": deque(
            MultiPeriodWindPEM._h2_kg_per_hr(per[t]["pem_elec"].value) for t in range(last_implemented_time_step + 1))}

    def _result_columns(self, b):
        per = b.windPEM["periods"]
        T = len(per)
        x = np.asarray(b.solution)
        col = lambda key: x[..., [p[key].index for p in per]]
        return T, {
            "Total Wind Generation [MW]": np.round(col("wind") * 1e-3, 2),
            "Total Power Output [MW]": np.round(b.family_values("P_T")[..., :T], 2),
            "Wind Power Output [MW]": np.round(col("grid_elec") * 1e-3, 2),
            "Wind to PEM [MW]": np.round(col("pem_elec") * 1e-3, 2),
            "Wind Curtailment [MW]": self._round_scalar(b.family_values("wind_waste")[..., 0]),
            "Hydrogen Sales [kg]": np.round(self._h2_kg_per_hr(col("pem_elec")), 2),
            "Total Cost [$]": np.round(b.family_values("tot_cost")[..., :T], 2),
        }

    @property
    def power_output(self):
        return "P_T"

    @property
    def total_cost(self):
        return ("tot_cost", 1)
