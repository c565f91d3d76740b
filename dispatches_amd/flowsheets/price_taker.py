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


def _prefix_network(b, leaves, fan_out=6):
    """Exclusive prefix sums of the expressions `leaves` as the variables of a work-efficient parallel-prefix network
    (up-sweep block sums + down-sweep prefixes) instead of a T-long integrator chain.  Returns the list of LinExpr
    `before[t] = sum_{s < t} leaves[s]`.

    Why: the reference accumulates the battery's energy throughput through T linked equalities E_t = E_{t-1} + (I_t + O_t) / 2
    (battery.py:155-159 + the linking pairs wind_battery_LMP.py:22-37) and E_t enters every period's state-of-charge bound
    with the degradation rate (battery.py:161-165).  For a first-order method that chain is a T-step integrator: information
    moves one period per iteration and the iteration count grows LINEARLY with T (lab, tools/stream_lab.py: 7.9 k / 19 k /
    39 k iterations at T = 168 / 336 / 672, against 5.1 k / 6.3 k / 7.8 k with the chain cut).  The same prefix sums written as a
    network of depth 2 log2 T - every variable the sum of two others, every variable used a bounded number of times - are
    the same LP (an exact linear change of variables: E_t = before[t] + leaf_t) with a dependency depth of ~27 instead of
    8736.  `fan_out`: a down-sweep prefix handed down a left spine is copied into a fresh variable after this many uses,
    which bounds the column length (the streaming kernels keep columns of <= 8 entries in their ELL part)."""
    T = len(leaves)
    block_sum = {}

    def up(lo, hi):                                     # variable (or leaf expression) of sum_{lo <= s < hi} leaves[s]
        if hi - lo == 1:
            return leaves[lo]
        key = (lo, hi)
        if key not in block_sum:
            mid = (lo + hi) // 2
            v = b.var(f"throughput_sum[{lo}:{hi}]")
            b.equality(f"throughput_sum_def[{lo}:{hi}]", v - up(lo, mid) - up(mid, hi), 0.0)
            block_sum[key] = v
        return block_sum[key]

    before = [None] * T

    def down(lo, hi, prefix, uses):                     # prefix: LinExpr / Var of sum_{s < lo} leaves[s], or None for zero
        if hi - lo == 1:
            before[lo] = LinExpr() if prefix is None else LinExpr._as(prefix)
            return
        mid = (lo + hi) // 2
        if prefix is not None and uses >= fan_out:      # bound the column length: continue with a copy
            cp = b.var(f"throughput_before_copy[{lo}:{hi}]")
            b.equality(f"throughput_before_copy_def[{lo}:{hi}]", cp - prefix, 0.0)
            prefix, uses = cp, 0
        left_sum = up(lo, mid)
        if prefix is None:
            right_prefix = left_sum                     # alias: the prefix before `mid` IS the left block's sum
            down(lo, mid, None, 0)
            down(mid, hi, right_prefix, 1 if hi - lo > 2 else 0)
            return
        rp = b.var(f"throughput_before[{mid}]")
        b.equality(f"throughput_before_def[{mid}]", rp - prefix - left_sum, 0.0)
        down(lo, mid, prefix, uses + 1)
        down(mid, hi, rp, 0)

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    down(0, T, None, 0)
    return before


def _hierarchical_basis(b, T):
    """The accumulated throughput E_0 .. E_{T-1} as a combination of hierarchical hat functions (`throughput="hier"`): returns the T
    expressions E_t = z_const + z_lin t / (T - 1) + sum_k z_k hat_k(t), hat_k the piecewise-linear function of the bisection tree's
    node k (1 at its midpoint, 0 at and beyond the ends of its interval); T free columns `throughput_hier[...]` replace the T columns
    `battery.energy_throughput[t]`.

    Why: E_t - E_{t-1} = (I_t + O_t) / 2 is a first difference over the WHOLE horizon (the state of charge resets every day, the
    throughput never does) - the smallest singular value of that block falls like 1 / T and PDHG pays O(T) iterations for it.  In
    this basis the difference operator has ORTHOGONAL columns ((D H)^T (D H) is diagonal: a hat's first differences are +1 / left
    width on its left half, -1 / right width on its right half, and hats of one level do not overlap, those of different levels are
    orthogonal in the derivative), so the block is perfectly conditioned after column scaling, and E >= 0 needs no rows (it is
    implied by the accumulation).  An exact change of variables: same optimum (tests/test_price_taker_cpu.py); numpy PDHG needs
    4-5 x fewer iterations (tools/stream_hier_lab.py, profiles/r30_lab_hierarchical.log).  Cost: 2 T log2 T nonzeros instead of 3 T,
    in columns as long as their hats - not banded."""
    zc = b.var("throughput_hier[const]", -np.inf, np.inf)
    zl = b.var("throughput_hier[linear]", -np.inf, np.inf)
    terms = [{zc.index: 1.0, zl.index: t / max(T - 1, 1)} for t in range(T)]
    terms[0].pop(zl.index)                               # (coefficient 0)
    stack = [(0, T - 1)]
    while stack:
        lo, hi = stack.pop()
        if hi - lo < 2:
            continue
        mid = (lo + hi) // 2
        z = b.var(f"throughput_hier[{lo}:{mid}:{hi}]", -np.inf, np.inf)
        for t in range(lo + 1, mid + 1):
            terms[t][z.index] = (t - lo) / (mid - lo)
        for t in range(mid + 1, hi):
            terms[t][z.index] = (hi - t) / (hi - mid)
        stack += [(lo, mid), (mid, hi)]
    return [LinExpr(dict(tm)) for tm in terms]


def _two_level_basis(b, T, n_nodes=3):
    """The accumulated throughput as `n_nodes` coarse values + local deviations (`throughput="two_level"`; units.two_level_accumulator,
    starting from the fixed initial throughput 0).  With 3 nodes the LP has 4 long columns (the nodes + the battery's power) - what the
    fused streaming iteration carries through per-tile partial sums today (csrc/dsp_stream.hpp, kFusedMaxLong) - and stays banded
    otherwise.  Lab (tools/stream_hier_lab.py, profiles/r30_lab_hierarchical.log): T = 2688: 171 k iterations -> 44 k with 4 nodes (40 k
    with 8 or 16; the full hierarchical basis: 39 k); GPU, T = 8736: 404 k -> 75 k on average (profiles/r30_two_level_probe_B16.log)."""
    from .units import two_level_accumulator
    return two_level_accumulator(b, T, n_nodes, base=None, prefix="throughput")


def wind_battery_price_taker(n_time_points, capacity_factors, lmps, wind_mw=847.0, wind_mw_ub=10000.0, batt_mw=0.0,
                             extant_wind=True, throughput="chain", coarse_nodes=3):
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
    if throughput not in ("chain", "scan", "hier", "two_level"):
        raise ValueError("throughput: 'chain' (the reference's linked equalities), 'scan' (parallel-prefix network), 'hier' "
                         "(the chain in a hierarchical basis) or 'two_level' (coarse node values + local deviations)")
    cols = []
    two_level = _two_level_basis(b, T, coarse_nodes) if throughput == "two_level" else None
    for t in range(T):
        W = b.var(f"windpower.electricity[{t}]", 0.0, wind_kw * cf[t])
        G = b.var(f"splitter.grid_elec[{t}]")
        I = b.var(f"splitter.battery_elec[{t}]")
        O = b.var(f"battery.elec_out[{t}]")
        S = b.var(f"battery.state_of_charge[{t}]", 0.0, 0.0 if t == T - 1 else np.inf)   # periodic: S_{T-1} = S_init = 0 (:40-51, :199)
        E = b.var(f"battery.energy_throughput[{t}]") if throughput == "chain" else None
        if throughput == "two_level":
            E = two_level(t, lambda tt: b.var(f"throughput_fine[{tt}]", -np.inf, np.inf))
        cols.append((W, G, I, O, S, E))
    if throughput == "scan":
        before = _prefix_network(b, [0.5 * I + 0.5 * O for (_W, _G, I, O, _S, _E) in cols])
    hier = _hierarchical_basis(b, T) if throughput == "hier" else [c[5] for c in cols] if throughput == "two_level" else None
    # (rows written as coefficient dictionaries: the operator form `S - eta_c * I + O / eta_d - soc_prev` copies a dictionary per
    #  operator - half of the time a year-long LP took to build; same coefficients to the bit)
    c_in, c_out = -(1.0 * eta_c), 1.0 * (1.0 / eta_d)
    for t, (W, G, I, O, S, E) in enumerate(cols):
        b.equality(f"splitter.sum_split[{t}]", LinExpr({W.index: 1.0, G.index: -1.0, I.index: -1.0}), 0.0)
        soc_row = {S.index: 1.0, I.index: c_in, O.index: c_out}
        if soc_prev is not None:
            soc_row[soc_prev.index] = -1.0
        b.equality(f"battery.state_evolution[{t}]", LinExpr(soc_row), 0.0)         # initial SOC / throughput fixed 0 (:199-200)
        if throughput == "chain":
            thr_row = {E.index: 1.0, I.index: -0.5, O.index: -0.5}
            if thr_prev is not None:
                thr_row[thr_prev.index] = -1.0
            b.equality(f"battery.accumulate_energy_throughput[{t}]", LinExpr(thr_row), 0.0)
            Et = LinExpr._as(E)
        elif throughput in ("hier", "two_level"):
            Et = hier[t]
            thr_rhs = Et - 0.5 * I - 0.5 * O
            if t > 0:
                thr_rhs = thr_rhs - hier[t - 1]
            b.equality(f"battery.accumulate_energy_throughput[{t}]", thr_rhs, 0.0)
        else:
            Et = before[t] + 0.5 * I + 0.5 * O                                      # the same E_t, as an expression
        b.constraint(f"battery.state_of_charge_bounds[{t}]", S + d * Et - DURATION * P, -np.inf, 0.0)
        b.constraint(f"battery.power_bound_in[{t}]", LinExpr({I.index: 1.0, P.index: -1.0}), -np.inf, 0.0)
        b.constraint(f"battery.power_bound_out[{t}]", LinExpr({O.index: 1.0, P.index: -1.0}), -np.inf, 0.0)
        revenue.accumulate(G, float(lmp[t])).accumulate(O, float(lmp[t]))
        per.append(dict(wind=W, grid_elec=G, elec_in=I, elec_out=O, state_of_charge=S, energy_throughput=Et))
        soc_prev, thr_prev = S, (E if throughput == "chain" else None)
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

    def column_scales(n_cols=None):
        """Variable scaling factors (dsp_lp_desc::col_scale: typical magnitude of every column), the physical ones a flowsheet
        knows - the reference sets them with iscale.set_scaling_factor on exactly these variables (RE_flowsheet.py:229-296,
        there as reciprocals).  Every kW column at the wind plant's nameplate, the state of charge at DURATION hours of it, the
        accumulated throughput at the energy a quarter-duty battery of that size moves up to its place in the horizon.  The
        ranges the BOUNDS imply (lp.implied_column_ranges) are meaningless here: the battery is a free design variable."""
        s = np.full(len(b.col_names) if n_cols is None else n_cols, wind_kw)
        duty = 0.25
        for j, name in enumerate(b.col_names):
            if name.startswith("battery.state_of_charge["):
                s[j] = DURATION * wind_kw
            elif name.startswith(("battery.energy_throughput[", "throughput_hier[", "throughput_node[", "throughput_fine[")):
                s[j] = wind_kw * max(T / 2, 1)
            elif name.startswith("throughput_sum[") or name.startswith("throughput_before_copy["):
                lo, hi = map(int, name[name.index("[") + 1:-1].split(":"))
                s[j] = wind_kw * duty * ((hi - lo) if name.startswith("throughput_sum[") else max(lo, 1))
            elif name.startswith("throughput_before["):
                s[j] = wind_kw * duty * max(int(name[name.index("[") + 1:-1]), 1)
        return s
    handles["column_scales"] = column_scales
    return b, objective, handles


PEM_CAP_COST = prm.pem_cap_cost          # $/kW  load_parameters.py:49


def wind_battery_pem_price_taker(time_points, capacity_factors, lmps, h2_price_per_kg=2.0, design_opt=True, wind_mw=847.0,
                                 throughput="chain", coarse_nodes=1):
    """Wind + battery + PEM price-taker design LP: the reference's ``wind_battery_pem_optimize``
    (``dispatches/case_studies/renewables_case/wind_battery_PEM_LMP.py:180-298``) on a LinearBlock.

    The LP #4 flowsheet plus an electrolyzer per period (splitter outlet `pem_elec` = `pem.electricity`, RE_flowsheet.py:391-398;
    hydrogen flow 0.00275984 mol/s per kW, :131) and the design column `pem_system_capacity` (:226, :243).  What differs from
    ``wind_battery_optimize``: only block 0's initial THROUGHPUT is fixed (:222) - its initial state of charge is a free column tied
    to the last period's by the periodic pair (:36-47); hydrogen is sold at `h2_price_per_kg` (:281: price * flow_mol / 500 *
    3600 per hour); the PEM pays 0.03 * 1200 $/kW-yr on its capacity (:275-277) and 1200 $/kW in the NPV (:291-294);
    `design_opt="PEM"` fixes the battery's nameplate power to 0 (:237-238); the wind farm is extant (capacity fixed, no capital
    cost: :234, :254-255 - the reference's default input parameters).  Same reductions as LP #4 (one nameplate-power column, 4-h
    energy substituted).  `throughput="hier"` / `"two_level"`: the accumulated-throughput chain in the hierarchical basis
    (`_hierarchical_basis`) or as `coarse_nodes` node values + local deviations (`_two_level_basis`; this LP already has three
    columns that span the horizon - battery power, PEM capacity, the periodic initial state of charge - so ONE node keeps it within
    the four the fused streaming iteration carries): exact changes of variables.  Returns (block, objective LinExpr of
    -NPV * 1e-5, handles)."""
    if throughput not in ("chain", "hier", "two_level"):
        raise ValueError("throughput: 'chain' (the reference's linked equalities), 'hier' (the chain in a hierarchical basis) or "
                         "'two_level' (coarse node values + local deviations)")
    T = int(time_points)
    cf = np.asarray(capacity_factors, float)[:T]
    lmp = np.asarray(lmps, float)[:T] * 1e-3                                          # $/kWh (:280)
    if design_opt not in (True, "PEM"):
        raise NotImplementedError("design_opt: True (battery and PEM sized) or 'PEM' (battery fixed at 0)")
    b = LinearBlock("pem_price_taker")
    wind_kw = wind_mw * 1e3
    Pb = b.var("battery_system_capacity", 0.0, np.inf)
    Cp = b.var("pem_system_capacity", 0.0, np.inf)
    P = b.var("battery.nameplate_power", 0.0, 0.0 if design_opt == "PEM" else 1e8)
    S_init = b.var("battery.initial_state_of_charge[0]")
    b.constraint("battery_max_p", P - Pb, -np.inf, 0.0)                               # :242
    eta_c, eta_d, d = prm.battery_charging_eta, prm.battery_discharging_eta, prm.battery_degradation_rate
    h2_per_kwh = prm.pem_electricity_to_mol / prm.h2_mols_per_kg * 3600.0            # kg of hydrogen per kWh into the PEM
    per = []
    soc_prev, thr_prev = S_init, None
    rev_e, h2_kg = LinExpr(), LinExpr()
    hier = None
    if throughput == "hier":                                                          # (columns after the design variables)
        hier = _hierarchical_basis(b, T)
    two_level = _two_level_basis(b, T, coarse_nodes) if throughput == "two_level" else None
    for t in range(T):
        W = b.var(f"windpower.electricity[{t}]", 0.0, wind_kw * cf[t])
        G = b.var(f"splitter.grid_elec[{t}]")
        I = b.var(f"splitter.battery_elec[{t}]")
        X = b.var(f"splitter.pem_elec[{t}]")
        O = b.var(f"battery.elec_out[{t}]")
        S = b.var(f"battery.state_of_charge[{t}]")
        if two_level is not None:
            E = two_level(t, lambda tt: b.var(f"throughput_fine[{tt}]", -np.inf, np.inf))
        else:
            E = b.var(f"battery.energy_throughput[{t}]") if hier is None else hier[t]
        b.equality(f"splitter.sum_split[{t}]", W - G - I - X, 0.0)
        b.equality(f"battery.state_evolution[{t}]", S - soc_prev - eta_c * I + O / eta_d, 0.0)
        thr = E - 0.5 * I - 0.5 * O
        b.equality(f"battery.accumulate_energy_throughput[{t}]", thr if thr_prev is None else thr - thr_prev, 0.0)
        b.constraint(f"battery.state_of_charge_bounds[{t}]", S + d * E - DURATION * P, -np.inf, 0.0)
        b.constraint(f"battery.power_bound_in[{t}]", I - P, -np.inf, 0.0)
        b.constraint(f"battery.power_bound_out[{t}]", O - P, -np.inf, 0.0)
        b.constraint(f"pem_max_p[{t}]", X - Cp, -np.inf, 0.0)                         # :243
        rev_e.accumulate(G, float(lmp[t])).accumulate(O, float(lmp[t]))
        h2_kg.accumulate(X, h2_per_kwh)
        per.append(dict(wind=W, grid_elec=G, elec_in=I, pem_elec=X, elec_out=O, state_of_charge=S, energy_throughput=E))
        soc_prev, thr_prev = S, E
    b.equality("battery.periodic_state_of_charge", soc_prev - S_init, 0.0)            # :36-47
    k = 52 / (T / (7 * 24))
    fixed = (wind_kw * (prm.wind_op_cost / 8760) + Pb * (BATT_OP_COST / 8760) + Cp * (prm.pem_op_cost / 8760)) * T
    annual = (rev_e + h2_kg * float(h2_price_per_kg) - fixed) * k
    npv = annual * PA - Pb * (BATT_CAP_COST_KW + BATT_CAP_COST_KWH * DURATION) - Cp * PEM_CAP_COST
    objective = npv * -1e-5
    b.expression("NPV", 0, npv)
    b.expression("annual_rev_E", 0, rev_e * k)
    b.expression("annual_rev_h2", 0, h2_kg * (float(h2_price_per_kg) * k))
    handles = dict(periods=per, battery_system_capacity=Pb, pem_system_capacity=Cp, nameplate_power=P, initial_state_of_charge=S_init)

    def objective_vector(n_cols, h2_price=h2_price_per_kg, pem_cap_factor=1.0, lmp_multiplier=1.0):
        """Dense cost vector of -NPV * 1e-5 for a member of the family (hydrogen price, PEM capital-cost factor, LMP multiplier):
        matrix and bounds stay the same."""
        c = np.zeros(n_cols)
        kk = PA * k * -1e-5
        sell = np.array([[p["grid_elec"].index, p["elec_out"].index] for p in per])
        c[sell[:, 0]] = c[sell[:, 1]] = kk * lmp * lmp_multiplier
        c[[p["pem_elec"].index for p in per]] = kk * h2_per_kwh * h2_price
        c[Pb.index] = -kk * T * BATT_OP_COST / 8760 + 1e-5 * (BATT_CAP_COST_KW + BATT_CAP_COST_KWH * DURATION)
        c[Cp.index] = -kk * T * prm.pem_op_cost / 8760 + 1e-5 * pem_cap_factor * PEM_CAP_COST
        return c
    handles["objective_vector"] = objective_vector
    handles["objective_constant"] = lambda: -PA * k * T * wind_kw * (prm.wind_op_cost / 8760) * -1e-5

    def column_scales(n_cols=None):
        """Physical scaling factors (see wind_battery_price_taker)."""
        s = np.full(len(b.col_names) if n_cols is None else n_cols, wind_kw)
        for j, name in enumerate(b.col_names):
            if "state_of_charge" in name:
                s[j] = DURATION * wind_kw
            elif name.startswith(("battery.energy_throughput[", "throughput_hier[", "throughput_node[", "throughput_fine[")):
                s[j] = wind_kw * max(T / 2, 1)
        return s
    handles["column_scales"] = column_scales
    return b, objective, handles


# ---- nuclear + PEM price-taker enumeration -----------------------------------------------------------------------
# Reference: dispatches/case_studies/nuclear_case/report/price_taker_analysis.py - build_ne_flowsheet (:116-172), the
# MultiPeriodModel of 366 x 24 hourly flowsheets linked by the tank holdup (:175-222), hourly cash flow (:225-254), annualised NPV
# (:257-322) and `run_exhaustive_enumeration` (:353-419): 6 hydrogen prices x 10 PEM capacities = 60 LPs that share ONE constraint
# matrix and differ in the objective (hydrogen price) and in one fixed design column (pem_capacity) - the natural batch.
NP_CAPACITY_MW = 400.0            # :140
H2_PROD_RATE = 20.0               # kg of hydrogen per MWh into the electrolyzer (:42)
TURBINE_MWH_PER_KG = 0.0125       # :160-163
NUCLEAR_H2_PRICES = (0.75, 1.0, 1.25, 1.5, 1.75, 2.0)                       # $/kg  (:358)
NUCLEAR_PEM_FRACTIONS = tuple(i / 100 for i in range(5, 51, 5))              # of the plant's 400 MW (:359)


def nuclear_price_taker(n_time_points, lmps, h2_demand=NP_CAPACITY_MW * H2_PROD_RATE, pem_capex=1200.0, vom_npp=2.3, vom_pem=0.0,
                        vom_turbine=4.25, plant_life=30, tax_rate=0.2, discount_rate=0.08, capex_tank=29.0, capex_turbine=947.0,
                        fom_turbine=7.0):
    """The LP of one enumeration point as the reference builds it (tank and turbine present in the flowsheet, their capacities fixed
    to 0 by the study, :374-377; variable hydrogen demand: h2_to_pipeline <= 400 * 20 kg/h, :360-361, :218-219).  `lmps` in $/MWh.
    Objective: MAXIMISE the annualised NPV  net_profit - capex / constant_cf_factor  (:318-322), returned as the LinExpr to
    MINIMISE (its negative, scaled by 1e-6: the reference reports million $).  pem_capacity is a column whose bounds the caller
    fixes per scenario (`m.pem_capacity.fix(pc * 400)`, :399).  Returns (block, objective, handles)."""
    T = int(n_time_points)
    lmp = np.asarray(lmps, float)[:T]
    b = LinearBlock("nuclear_price_taker")
    pem_cap = b.var("pem_capacity", 0.0, np.inf, mutable=True, hull=(0.0, NP_CAPACITY_MW))
    tank_cap = b.var("tank_capacity", 0.0, 0.0)                                  # fixed to 0 by the study (:376)
    turb_cap = b.var("h2_turbine_capacity", 0.0, 0.0)                            # (:377)
    per = []
    hold_prev = None
    cash = LinExpr()
    for t in range(T):
        g = b.var(f"fs.np_to_grid[{t}]")
        e = b.var(f"fs.np_to_electrolyzer[{t}]")
        h = b.var(f"fs.h2_production[{t}]")
        hold = b.var(f"fs.tank_holdup[{t}]")
        pipe = b.var(f"fs.h2_to_pipeline[{t}]", 0.0, h2_demand)
        turb_in = b.var(f"fs.h2_to_turbine[{t}]")
        turb_p = b.var(f"fs.h2_turbine_power[{t}]")
        net = b.var(f"fs.net_power[{t}]")
        b.equality(f"fs.np_power_balance[{t}]", g + e, NP_CAPACITY_MW)                         # np_power fixed at 400 (:140-146)
        b.equality(f"fs.calc_h2_production_rate[{t}]", h - H2_PROD_RATE * e, 0.0)              # :149-152
        bal = hold - h + pipe + turb_in                                                       # :154-157; holdup_previous[0] = 0 (:375)
        b.equality(f"fs.tank_mass_balance[{t}]", bal if hold_prev is None else bal - hold_prev, 0.0)
        b.equality(f"fs.calc_turbine_power[{t}]", turb_p - TURBINE_MWH_PER_KG * turb_in, 0.0)  # :164-167
        b.equality(f"fs.grid_power_balance[{t}]", net - g - turb_p, 0.0)                       # :169-171
        b.constraint(f"pem_capacity_constraint[{t}]", e - pem_cap, -np.inf, 0.0)               # :200-202
        b.constraint(f"tank_capacity_constraint[{t}]", hold - tank_cap, -np.inf, 0.0)          # :204-206
        b.constraint(f"turbine_capacity_constraint[{t}]", turb_p - turb_cap, -np.inf, 0.0)     # :208-210
        cash.accumulate(net, float(lmp[t])).accumulate(e, -vom_pem).accumulate(turb_p, -vom_turbine)   # :239-254 (hydrogen revenue below)
        per.append(dict(np_to_grid=g, np_to_electrolyzer=e, h2_production=h, tank_holdup=hold, h2_to_pipeline=pipe,
                        h2_to_turbine=turb_in, h2_turbine_power=turb_p, net_power=net))
        hold_prev = hold
    cf_factor = (1 - (1 + discount_rate) ** (-plant_life)) / discount_rate                    # :308
    fom_pem = 0.03 * pem_capex                                                                 # :395
    capex = pem_cap * (pem_capex * 1000) + tank_cap * (capex_tank * 33.3) + turb_cap * (capex_turbine * 1000)        # :274-283
    fixed_om = pem_cap * (1000 * fom_pem) + turb_cap * (1000 * fom_turbine) + 120 * 1000 * NP_CAPACITY_MW            # :277-290
    npp_vom = vom_npp * NP_CAPACITY_MW * T

    def annualised_npv(h2_price):
        inflow = cash - npp_vom
        for p in per:
            inflow.accumulate(p["h2_to_pipeline"], float(h2_price))
        depreciation = capex / plant_life                                                      # :298-301
        net_profit = depreciation + (inflow - fixed_om - depreciation) * (1 - tax_rate)       # :303-305
        return net_profit - capex * (1 / cf_factor), net_profit                               # :318-322
    npv, net_profit = annualised_npv(NUCLEAR_H2_PRICES[0])
    objective = npv * -1e-6
    b.expression("annualised_npv", 0, npv)
    b.expression("net_profit", 0, net_profit)
    handles = dict(periods=per, pem_capacity=pem_cap, tank_capacity=tank_cap, h2_turbine_capacity=turb_cap)

    def objective_vector(n_cols, h2_price):
        """Dense cost vector of -annualised NPV * 1e-6 at hydrogen price `h2_price` (matrix and all other data unchanged)."""
        c = np.zeros(n_cols)
        k = -1e-6 * (1 - tax_rate)
        for t, p in enumerate(per):
            c[p["net_power"].index] = k * lmp[t]
            c[p["np_to_electrolyzer"].index] = -k * vom_pem
            c[p["h2_turbine_power"].index] = -k * vom_turbine
            c[p["h2_to_pipeline"].index] = k * h2_price
        # design columns: depreciation tax shield, fixed O&M, annualised capital
        for col, cap, fom in ((pem_cap, pem_capex * 1000, 1000 * fom_pem), (tank_cap, capex_tank * 33.3, 0.0), (turb_cap, capex_turbine * 1000, 1000 * fom_turbine)):
            c[col.index] = -1e-6 * (cap / plant_life * tax_rate - (1 - tax_rate) * fom - cap / cf_factor)
        return c
    handles["objective_vector"] = objective_vector
    handles["objective_constant"] = -1e-6 * (1 - tax_rate) * (-npp_vom - 120 * 1000 * NP_CAPACITY_MW)

    def column_scales(n_cols=None):
        s = np.full(len(b.col_names) if n_cols is None else n_cols, NP_CAPACITY_MW)
        for j, name in enumerate(b.col_names):
            if any(k in name for k in ("h2_production", "tank_holdup", "h2_to_pipeline", "h2_to_turbine", "tank_capacity")):
                s[j] = NP_CAPACITY_MW * H2_PROD_RATE
        return s
    handles["column_scales"] = column_scales
    return b, objective, handles
