"""Long-horizon price-taker design LPs: wind + battery sized and operated against a year of LMPs.

Restates ``wind_battery_optimize`` (``dispatches/case_studies/renewables_case/wind_battery_LMP.py:172-269``) on a
LinearBlock: a MultiPeriodModel of `n_time_points` hourly flowsheets (``RE_flowsheet.create_model`` with wind,
splitter and battery: wind_power.py:120-122, elec_splitter.py:115-117, battery.py:145-165, four-hour battery
RE_flowsheet.py:155-156), linked by state of charge / energy throughput / nameplate power
(wind_battery_LMP.py:22-37), periodic in the state of charge (:40-51), with the design columns
`wind_system_capacity` / `battery_system_capacity` (:203-213) and the objective  min -NPV * 1e-5  (:253-264).

Reduction (exact): the per-period `nameplate_power` columns are linked EQUAL across all periods (:34, :50), so they are
one column `P`; `nameplate_energy = 4 P` (RE_flowsheet.py:156) is substituted.  With `extant_wind` (the default input
parameters, load_parameters.py:137-138) the per-period wind capacity stays fixed and only `wind_system_capacity >= it`
remains.  T = 8736 gives n = 6 T + 3, m = 6 T + 2: far beyond the register/LDS-resident kernels - this family runs
on the streaming (HBM-resident) PDLP path.
"""
from __future__ import annotations

import numpy as np

from ..lp import LinearBlock, LinExpr
from . import parameters as prm

# load_parameters.py:40-44,119-121 (wind_battery_cost_parameter.json, moderate / 2023, 4-h battery)
BATT_OP_COST = 31.39              # $/kW-yr   battery.fixed_om.moderate.2023[1]  (4-h battery: arg_duration = 1)
BATT_CAP_COST_KW = 236.365        # $/kW      battery.batt_cap_cost_param.moderate.2023[0]
BATT_CAP_COST_KWH = 254.835       # $/kWh     battery.batt_cap_cost_param.moderate.2023[1]
WIND_CAP_COST = 1308.0            # $/kW      wind.capital.moderate.2023[0]
DISCOUNT_RATE, YEARS = 0.08, 30
PA = ((1 + DISCOUNT_RATE) ** YEARS - 1) / (DISCOUNT_RATE * (1 + DISCOUNT_RATE) ** YEARS)
DURATION = 4.0


def wind_battery_price_taker(n_time_points, capacity_factors, lmps, wind_mw=847.0, wind_mw_ub=10000.0, batt_mw=0.0,
                             extant_wind=True):
    """Build the LP.  `lmps` in $/MWh (the reference multiplies by 1e-3: $/kWh, :249); returns (block, objective,
    handles) where objective is the LinExpr of  -NPV * 1e-5  to MINIMISE."""
    if not extant_wind:
        raise NotImplementedError("only the default extant_wind = True design problem is restated")
    T = int(n_time_points)
    cf = np.asarray(capacity_factors, float)[:T]
    lmp = np.asarray(lmps, float)[:T] * 1e-3
    b = LinearBlock("price_taker")
    wind_kw = wind_mw * 1e3
    Cw = b.var("wind_system_capacity", 0.0, wind_mw_ub * 1e3)
    Pb = b.var("battery_system_capacity", 0.0, np.inf)
    P = b.var("battery.nameplate_power", 0.0, 1e8)                       # battery.py bounds (0, 1e8)
    b.constraint("wind_max_p", LinExpr({Cw.index: 1.0}), wind_kw, np.inf)          # fixed block capacity <= Cw  (:212)
    b.constraint("battery_max_p", P - Pb, -np.inf, 0.0)                            # (:213)
    eta_c, eta_d, d = prm.battery_charging_eta, prm.battery_discharging_eta, prm.battery_degradation_rate
    per = []
    soc_prev = thr_prev = None
    revenue = LinExpr()
    for t in range(T):
        W = b.var(f"windpower.electricity[{t}]", 0.0, wind_kw * cf[t])
        G = b.var(f"splitter.grid_elec[{t}]")
        I = b.var(f"splitter.battery_elec[{t}]")
        O = b.var(f"battery.elec_out[{t}]")
        S = b.var(f"battery.state_of_charge[{t}]", 0.0, 0.0 if t == T - 1 else np.inf)   # periodic: S_{T-1} = S_init = 0 (:40-51, :199)
        E = b.var(f"battery.energy_throughput[{t}]")
        b.equality(f"splitter.sum_split[{t}]", W - G - I, 0.0)
        soc_rhs = S - eta_c * I + O / eta_d
        thr_rhs = E - 0.5 * I - 0.5 * O
        if soc_prev is not None:
            soc_rhs, thr_rhs = soc_rhs - soc_prev, thr_rhs - thr_prev
        b.equality(f"battery.state_evolution[{t}]", soc_rhs, 0.0)                  # initial SOC / throughput fixed 0 (:199-200)
        b.equality(f"battery.accumulate_energy_throughput[{t}]", thr_rhs, 0.0)
        b.constraint(f"battery.state_of_charge_bounds[{t}]", S + d * E - DURATION * P, -np.inf, 0.0)
        b.constraint(f"battery.power_bound_in[{t}]", I - P, -np.inf, 0.0)
        b.constraint(f"battery.power_bound_out[{t}]", O - P, -np.inf, 0.0)
        revenue = revenue + (G + O) * float(lmp[t])
        per.append(dict(wind=W, grid_elec=G, elec_in=I, elec_out=O, state_of_charge=S, energy_throughput=E))
        soc_prev, thr_prev = S, E
    n_weeks = T / (7 * 24)
    op_cost = (Cw * (prm.wind_op_cost / 8760) + Pb * (BATT_OP_COST / 8760)) * T
    annual_revenue = (revenue - op_cost) * (52 / n_weeks)
    npv = annual_revenue * PA - Pb * (BATT_CAP_COST_KW + BATT_CAP_COST_KWH * DURATION)     # extant wind: no wind capital (:250-251)
    objective = npv * -1e-5
    b.expression("NPV", 0, npv)
    b.expression("annual_revenue", 0, annual_revenue)
    handles = dict(periods=per, wind_system_capacity=Cw, battery_system_capacity=Pb, nameplate_power=P)

    def objective_vector(n_cols, lmp_multiplier=1.0, batt_cap_factor=1.0):
        """Dense cost vector of  -NPV * 1e-5  for a scenario of the family: all LMPs scaled by `lmp_multiplier`, battery
        capital cost scaled by `batt_cap_factor` (the constraint matrix and every bound stay the same)."""
        c = np.zeros(n_cols)
        k = PA * (52 / n_weeks) * -1e-5
        sell = np.array([[p["grid_elec"].index, p["elec_out"].index] for p in per])
        c[sell[:, 0]] = c[sell[:, 1]] = k * lmp * lmp_multiplier
        c[Cw.index] = -k * T * prm.wind_op_cost / 8760
        c[Pb.index] = -k * T * BATT_OP_COST / 8760 + 1e-5 * batt_cap_factor * (BATT_CAP_COST_KW + BATT_CAP_COST_KWH * DURATION)
        return c
    handles["objective_vector"] = objective_vector
    return b, objective, handles
