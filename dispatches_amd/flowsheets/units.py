"""Linear unit-model rows (the equations of the reference's IDAES unit models, restated on a LinearBlock).

Units follow the reference: power kW, energy kWh, dt = 1 h inside the flowsheet; P_T and bids in MW.
Every function appends columns/rows for ONE time period `t` and returns the handles.
"""
from __future__ import annotations

import numpy as np

from ..lp import INF, LinearBlock


def wind_power(b: LinearBlock, t: int, system_capacity_kw: float, capacity_factor: float):
    """Wind plant: 0 <= electricity <= system_capacity * capacity_factor (curtailment allowed).
    Reference: dispatches/unit_models/wind_power.py:120-122.  The availability is a mutable column bound
    (`update_wind_capacity_factor`, wind_battery_double_loop.py:86-98)."""
    return b.var(f"windpower.electricity[{t}]", 0.0, system_capacity_kw * capacity_factor,
                 mutable=True, hull=(0.0, system_capacity_kw))


def splitter(b: LinearBlock, t: int, inlet, outlet_names):
    """Electrical splitter: inlet = sum(outlets).  Reference: dispatches/unit_models/elec_splitter.py:115-117."""
    outs = [b.var(f"splitter.{nm}[{t}]") for nm in outlet_names]
    body = inlet
    for o in outs:
        body = body - o
    b.equality(f"splitter.sum_split[{t}]", body, 0.0)
    return outs


def two_level_accumulator(b: LinearBlock, T: int, n_nodes: int, base=None, prefix="throughput"):
    """A running sum E_0 .. E_{T-1} (the battery's accumulated energy throughput) as `n_nodes` node values + local deviations:
    E_t = base + the piecewise-linear interpolant of the node values v_k (nodes at the ends of `n_nodes` equal stretches of the
    horizon; 0 before the first period) + e_t, with e_t a column of its own for every period that is not a node.  `base`: the
    column (or None = 0) the sum starts from.  Still one column per period, an exact change of variables; a period touches at
    most two node columns, and the node columns are the only ones that are not local.

    Why: E_t - E_{t-1} = (in + out) / 2 is a first difference over the whole horizon - one slow mode per stretch of it, a smallest
    singular value ~ 1 / T, O(T) PDHG iterations.  The node columns move whole stretches at once and leave the deviations a chain of
    length T / n_nodes whose ends are pinned (year-long price-taker LPs 404 k -> 75 k iterations on the GPU;
    the 24-h / 48-h bidding LPs 1213 -> 896 / 2386 -> 1360 in the lab with 2 nodes).

    Returns expression(t, fine_var): the LinExpr of E_t; `fine_var(t)` must create period t's deviation column - called where the
    flowsheet creates the period's other columns, so that a multi-period matrix stays banded in the order it is handed over."""
    from ..lp import LinExpr
    K = max(1, int(n_nodes))
    nodes = sorted({min(T - 1, max(0, int(round((k + 1) * T / K)) - 1)) for k in range(K)} | {T - 1})
    v = [b.var(f"{prefix}_node[{t}]", -INF, INF) for t in nodes]
    where = {}
    k = 0                                               # the stretch (nodes[k-1], nodes[k]] that holds t
    for t in range(T):
        while t > nodes[k]:
            k += 1
        lo = nodes[k - 1] if k > 0 else -1
        w = (t - lo) / (nodes[k] - lo)
        terms = {v[k].index: w}
        if k > 0 and w < 1.0:
            terms[v[k - 1].index] = 1.0 - w
        if base is not None:
            terms[base.index] = 1.0
        where[t] = (terms, t != nodes[k])

    def expression(t, fine_var):
        terms, has_fine = where[t]
        terms = dict(terms)
        if has_fine:
            terms[fine_var(t).index] = 1.0
        return LinExpr(terms)
    return expression


def battery(b: LinearBlock, t: int, elec_in, soc_prev, thr_prev, nameplate_power_kw, nameplate_energy_kwh,
            charging_eta=0.95, discharging_eta=0.95, degradation_rate=1e-4, ramp_rate=None, throughput=None):
    """Battery storage rows for one period (dt = 1 h).

    Reference: dispatches/unit_models/battery.py
      :145-149 state_evolution         soc = soc_prev + eta_c*elec_in - elec_out/eta_d
      :151-153 accumulate_throughput   thr = thr_prev + (elec_in + elec_out)/2
      :155-157 state_of_charge_bounds  soc <= nameplate_energy - degradation_rate*thr
      :159-165 power bounds            elec_in, elec_out <= nameplate_power   (column bounds here)
    and the (never-binding, 1e8) energy ramp rows of wind_battery_LMP.py:139-142 when `ramp_rate` is given.
    `elec_in` is an existing column (the splitter outlet; arcs RE_flowsheet.py:389,396 equate them).
    `throughput`: the period's accumulated throughput as an EXPRESSION (two_level_accumulator) instead of a column of its own.
    """
    if elec_in.ub > nameplate_power_kw:
        elec_in.setub(nameplate_power_kw)
    elec_out = b.var(f"battery.elec_out[{t}]", 0.0, nameplate_power_kw)
    soc = b.var(f"battery.state_of_charge[{t}]")
    thr = b.var(f"battery.energy_throughput[{t}]") if throughput is None else throughput
    b.equality(f"battery.state_evolution[{t}]",
               soc - soc_prev - charging_eta * elec_in + elec_out / discharging_eta, 0.0)
    b.equality(f"battery.accumulate_energy_throughput[{t}]",
               thr - thr_prev - 0.5 * elec_in - 0.5 * elec_out, 0.0)
    b.constraint(f"battery.state_of_charge_bounds[{t}]", soc + degradation_rate * thr, -INF, nameplate_energy_kwh)
    if ramp_rate is not None:
        b.constraint(f"battery.energy_ramp[{t}]", soc - soc_prev, -ramp_rate, ramp_rate)
    return dict(elec_out=elec_out, state_of_charge=soc, energy_throughput=thr)


def pem_electrolyzer(b: LinearBlock, t: int, electricity, system_capacity):
    """PEM: electricity <= pem_system_capacity (a free NonNegative column shared by all periods,
    wind_PEM_double_loop.py:76-80); H2 flow = electricity_to_mol * electricity is an output-only expression
    (dispatches/unit_models/pem_electrolyzer.py:111-114)."""
    b.constraint(f"pem.max_p[{t}]", electricity - system_capacity, -INF, 0.0)


def hydrogen_tank(b: LinearBlock, t: int, holdup_prev, inlet_mol_per_s, dt_s=3600.0, demand_ub_mol_s=INF,
                  holdup_ub_mol=INF):
    """Simplified tank: holdup - holdup_prev = dt * (inlet - outlet_to_pipeline - outlet_to_turbine[=0]).
    Reference: dispatches/unit_models/hydrogen_tank_simplified.py:177-183."""
    holdup = b.var(f"h2_tank.tank_holdup[{t}]", 0.0, holdup_ub_mol)
    out = b.var(f"h2_tank.outlet_to_pipeline.flow_mol[{t}]", 0.0, demand_ub_mol_s)
    b.equality(f"h2_tank.tank_material_balance[{t}]", holdup - holdup_prev - dt_s * inlet_mol_per_s + dt_s * out, 0.0)
    return dict(tank_holdup=holdup, outlet_to_pipeline=out)


class ResultRecords:
    """record_results / write_results of the three multi-period model objects (reference: `record_results` of
    wind_battery_double_loop.py:276-335, wind_PEM_double_loop.py, nuclear multiperiod model: one row per horizon hour, the
    reference's column names).  A model object states its columns ONCE, shape-agnostically (`_result_columns(b)`: from a block
    whose solution is one scenario [n] or several [S, n]); `record_results` is the reference's per-block call, and
    `record_results_many` records S scenarios of a batch with the same dozen numpy operations instead of S times a dozen -
    the Bidder's per-scenario detail rows were 0.9 ms of a 6-ms compute_day_ahead_bids."""

    record_generator = True           # "Generator" is the first column (the nuclear model object has none)

    def _record_head(self, date, hour, T):
        head = {"Generator": self.model_data.gen_name} if self.record_generator else {}
        head.update({"Date": date, "Hour": hour, "Horizon [hr]": np.arange(T, dtype=int)})
        return head

    @staticmethod
    def _round_scalar(v):
        """round(value, 2) as Python does it, for one scenario (a float) or several (a list of floats)"""
        return round(float(v), 2) if np.ndim(v) == 0 else [round(float(w), 2) for w in v]

    def record_results(self, b, date=None, hour=None, **kwargs):
        T, cols = self._result_columns(b)
        # kept as a plain dict; the frames are built once in write_results (one pandas constructor per recorded
        # scenario and call was most of the host time of an hourly real-time bid)
        self.result_list.append({**self._record_head(date, hour, T), **cols, **kwargs})     # kwargs: e.g. Scenario=, Market= (last columns)

    def record_results_many(self, b, scenarios, date=None, hour=None, **kwargs):
        """b.solution: [len(scenarios), n].  The records record_results would append scenario by scenario with Scenario=i."""
        T, cols = self._result_columns(b)
        head = self._record_head(date, hour, T)
        for r, i in enumerate(scenarios):
            self.result_list.append({**head, **{k: v[r] for k, v in cols.items()}, "Scenario": i, **kwargs})

    def write_results(self, path):
        import pandas as pd
        pd.concat([pd.DataFrame(r) for r in self.result_list]).to_csv(path, index=False)
