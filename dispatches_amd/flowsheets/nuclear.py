"""Nuclear + PEM + hydrogen-tank multi-period model object (LP #3 of SURVEY.md App. A.3).

API mirror of ``dispatches/case_studies/nuclear_case/nuclear_flowsheet_multiperiod_class.py:158-344``
(`MultiPeriodNuclear`).  Flowsheet: ``nuclear_flowsheet.py:74-222`` with the turbine excluded (:99 of the
multiperiod class); the NPP output is fixed so the splitter's split-fraction rows are linear and reduce to
np_to_grid + np_to_pem = capacity.
"""
from __future__ import annotations

from collections import deque

import numpy as np
import pandas as pd

from . import parameters as prm
from . import units


def create_multiperiod_nuclear_model(b, n_time_points=4, h2_demand=0.35, demand_type="variable", h2_price=4):
    np_kw = prm.np_capacity_mw * 1e3
    pem_kw = prm.nuclear_pem_capacity_mw * 1e3
    tank_mol = prm.tank_capacity_kg / prm.mw_h2
    demand_mol_s = h2_demand / prm.mw_h2
    holdup_init = b.var("h2_tank.tank_holdup_previous[0]", 0.0, 0.0, mutable=True, hull=(0.0, tank_mol))
    periods = []
    prev = holdup_init
    for t in range(n_time_points):
        to_grid = b.var(f"np_power_split.np_to_grid[{t}]")
        to_pem = b.var(f"np_power_split.np_to_pem[{t}]", 0.0, pem_kw)           # nuclear_flowsheet.py:137-138
        b.equality(f"np_power_split.sum_split[{t}]", to_grid + to_pem, np_kw)  # electricity fixed (:125)
        # the capacity bound lives on tank_holdup_previous, i.e. on every holdup that feeds a later period
        tank = units.hydrogen_tank(
            b, t, prev, to_pem * prm.nuclear_pem_electricity_to_mol, 3600.0,
            demand_ub_mol_s=demand_mol_s if demand_type == "variable" else np.inf,
            holdup_ub_mol=tank_mol if t < n_time_points - 1 else np.inf)
        if demand_type == "fixed":
            tank["outlet_to_pipeline"].fix(demand_mol_s)
        periods.append(dict(np_to_grid=to_grid, pem_elec=to_pem, holdup_prev=prev, **tank))
        # operating cost (multiperiod_class.py:149-153): H2 revenue as negative cost, small storage cost
        b.expression("operating_cost", t,
                     np_kw * 1e-3 * prm.npp_vom
                     + to_pem * (1e-3 * prm.nuclear_pem_vom)
                     + tank["tank_holdup"] * (prm.mw_h2 * prm.tank_vom)
                     - tank["outlet_to_pipeline"] * (prm.mw_h2 * 3600 * h2_price))
        prev = tank["tank_holdup"]
    return dict(periods=periods, holdup_init=holdup_init)


class MultiPeriodNuclear(units.ResultRecords):
    # preconditioner hint for the HIP solver (include/dsp_hip.h geo_iters): the hydrogen-tank rows mix 3600 s/h,
    # mol/s and kW coefficients; geometric pre-equilibration halves the PDLP iteration count of this flowsheet
    solver_hints = {"geo_iters": 8}

    def __init__(self, model_data):
        self.mp_nuclear = None
        self.result_list = []
        self.model_data = model_data
        self.p_lower = model_data.p_min
        self.p_upper = model_data.p_max
        self.generator = model_data.gen_name

    def populate_model(self, blk, horizon):
        if not blk.is_constructed():
            blk.construct()
        mp = create_multiperiod_nuclear_model(blk, n_time_points=horizon)
        blk.nuclear = mp
        mp["holdup_init"].fix(0)                                         # reference :203
        blk.HOUR = range(horizon)
        for t, p in enumerate(mp["periods"]):
            blk.expression("P_T", t, p["np_to_grid"] * 1e-3)            # :211
            blk.expression("tot_cost", t, blk.operating_cost[t])        # :212
        self.mp_nuclear = mp

    @staticmethod
    def update_model(b, implemented_tank_holdup):
        """Re-fix the initial holdup to round(last implemented holdup) (reference :219-238)."""
        b.nuclear["holdup_init"].fix(round(implemented_tank_holdup[-1]))

    @staticmethod
    def get_last_delivered_power(b, last_implemented_time_step):
        return b.value(b.P_T[last_implemented_time_step])

    @staticmethod
    def get_implemented_profile(b, last_implemented_time_step):
        per = b.nuclear["periods"]
        return {"implemented_tank_holdup": deque(
            [per[t]["tank_holdup"].value for t in range(last_implemented_time_step + 1)])}

    record_generator = False

    def _result_columns(self, blk):
        per = blk.nuclear["periods"]
        T = len(per)
        x = np.asarray(blk.solution)
        col = lambda key: x[..., [p[key].index for p in per]]
        return T, {
            "Power to Grid [MW]": np.round(blk.family_values("P_T")[..., :T], 2),
            "Power to PEM [MW]": np.round(col("pem_elec") * 1e-3, 2),
            "Initial holdup [kg]": np.round(col("holdup_prev") * prm.mw_h2, 2),
            "Final holdup [kg]": np.round(col("tank_holdup") * prm.mw_h2, 2),
            "Hydrogen Market [kg/hr]": np.round(col("outlet_to_pipeline") * prm.mw_h2 * 3600, 2),
            "Total Cost [$]": np.round(blk.family_values("tot_cost")[..., :T], 2),
        }

    @property
    def power_output(self):
        return "P_T"

    @property
    def total_cost(self):
        return ("tot_cost", 1)

    @property
    def pmin(self):
        return self.p_lower
