"""Structural pin of the stochastic bidding problems on the reference's own solver logs.

No reference vector exercises the cross-scenario coupling rows with DIFFERENT scenarios, but the notebook
`dispatches/case_studies/renewables_case/DoubleLoopOptimization.ipynb` keeps the Xpress log of every bidding LP it solved
(3 price scenarios; day-ahead horizon 28, real-time horizon 4): rows x columns x elements of the problem Pyomo's direct
interface handed over (golden G11).  That interface adds EVERY Var of the block as a column (fixed ones with lb = ub) and
every active Constraint as a row (fixed variables folded into the constant; rows whose body is constant stay, with no
elements).  The table below lists the reference's components per scenario and period with where they are declared; the three
logged shapes follow from it only with

  * S * S coupling rows per period (`Bidder`: one per ORDERED scenario pair, the diagonal included) - not S - 1
    (non-anticipativity only) and not 2 (S - 1) (neighbouring scenarios only): the oracle's / the product's monotone form takes
    every unordered pair once, which spans the same feasible set;
  * no elements in them when the scenarios are identical (price difference 0: the notebook's backcaster has one historical day);
  * exactly ONE free variable of (day_ahead_power, real_time_underbid_power) in the underbid row of either problem;
  * `initial_energy_throughput` of period 0 a free column until the first `update_model` fixes it (351 -> 348 elements from
    the second real-time solve on: wind_battery_double_loop.py:197-200) - what oracle and product restate as `e0=None`.

The second half maps the oracle's (reduced) restatement onto the same table: every row / column the oracle does not carry is
one of the listed eliminations.
"""
import json
import os

import numpy as np
import pytest

from oracle import dispatch_lp_oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["G11_bidding_lp_shapes_xpress"]

# (rows, elements, elements at period 0) per scenario and period; reference file:line of the declaration
FLOWSHEET_ROWS = [
    ("windpower.elec_from_capacity_factor", 1, 1, 1),        # unit_models/wind_power.py:120-122 (system_capacity fixed)
    ("wind_to_splitter_expanded", 1, 2, 2),                  # RE_flowsheet.py:389 + expand_arcs :420
    ("splitter.sum_split", 1, 3, 3),                         # unit_models/elec_splitter.py:115-117
    ("splitter_to_battery_expanded", 1, 2, 2),               # RE_flowsheet.py:396
    ("battery.state_evolution", 1, 4, 3),                    # unit_models/battery.py:145-149; initial_state_of_charge fixed at t = 0
    ("battery.accumulate_energy_throughput", 1, 4, 4),       # battery.py:151-153
    ("battery.state_of_charge_bounds", 1, 2, 2),             # battery.py:155-157 (nameplate_energy fixed)
    ("battery.power_bound_in", 1, 1, 1),                     # battery.py:159-161 (nameplate_power fixed)
    ("battery.power_bound_out", 1, 1, 1),                    # battery.py:163-165
    ("battery.energy_down_ramp", 1, 2, 1),                   # wind_battery_LMP.py:139-140
    ("battery.energy_up_ramp", 1, 2, 1),                     # wind_battery_LMP.py:141-142
]
# free columns per scenario and period + columns that exist but are fixed
FLOWSHEET_COLS = dict(
    free=["windpower.electricity", "splitter.electricity", "splitter.grid_elec", "splitter.battery_elec",
          "battery.initial_state_of_charge", "battery.initial_energy_throughput", "battery.elec_in", "battery.elec_out",
          "battery.state_of_charge", "battery.energy_throughput"],
    fixed=["windpower.system_capacity", "battery.nameplate_power", "battery.nameplate_energy"],   # wind_battery_double_loop.py:71-74
)
# wind_battery_LMP.py:33-37: three linking pairs per period boundary (state of charge, throughput, nameplate_power: the last
# joins two fixed variables = a row without elements); :47-49 periodic pairs: [0] deactivated (wind_battery_double_loop.py:80-81),
# [1] (nameplate_power) stays as an empty row
LINK_ROWS, LINK_ELEMENTS, PERIODIC_ROWS = 3, 4, 1
BID_COLS = 2                                                  # day_ahead_power, real_time_underbid_power (idaes bidder)
UNDERBID_ELEMENTS = 4 - 1                                     # pda / underbid power + grid_elec + elec_out, ONE of the first two fixed


def pyomo_shape(S, T, throughput_init_fixed=False, coupling_rows_per_period=None, coupling_elements=0):
    rows_t = sum(r for _n, r, _e, _e0 in FLOWSHEET_ROWS) + 1
    el_t = sum(e for _n, _r, e, _e0 in FLOWSHEET_ROWS) + UNDERBID_ELEMENTS
    el_0 = sum(e0 for _n, _r, _e, e0 in FLOWSHEET_ROWS) + UNDERBID_ELEMENTS - (1 if throughput_init_fixed else 0)
    rows = S * (rows_t * T + LINK_ROWS * (T - 1) + PERIODIC_ROWS)
    cols = S * T * (len(FLOWSHEET_COLS["free"]) + len(FLOWSHEET_COLS["fixed"]) + BID_COLS)
    elements = S * (el_t * (T - 1) + el_0 + LINK_ELEMENTS * (T - 1))
    c = S * S if coupling_rows_per_period is None else coupling_rows_per_period
    return rows + c * T, cols, elements + coupling_elements * T


def test_component_table_reproduces_the_logged_shapes():
    S = GOLD["n_scenario"]
    da, rt0, rt1 = GOLD["day_ahead"], GOLD["real_time_first"], GOLD["real_time_later"]
    assert pyomo_shape(S, da["horizon"]) == (da["rows"], da["cols"], da["elements"])
    assert pyomo_shape(S, rt0["horizon"]) == (rt0["rows"], rt0["cols"], rt0["elements"])
    assert pyomo_shape(S, rt1["horizon"], throughput_init_fixed=True) == (rt1["rows"], rt1["cols"], rt1["elements"])


@pytest.mark.parametrize("name,per_period", [("non-anticipativity only", 2), ("one per scenario", 3),
                                             ("neighbouring scenarios, both orders", 4), ("unordered pairs", 3),
                                             ("ordered pairs without the diagonal", 6)])
def test_other_coupling_forms_do_not_fit_the_log(name, per_period):
    S = GOLD["n_scenario"]
    for key in ("day_ahead", "real_time_first"):
        g = GOLD[key]
        assert pyomo_shape(S, g["horizon"], coupling_rows_per_period=per_period)[0] != g["rows"], name


def test_both_logged_problems_determine_the_coupling_count():
    """rows = S (a T + b) + c T with unknown a, b, c: the two horizons give c T-proportional part; with the per-scenario part of
    the table (a = 15, b = -2) c = 9 = S * S is the only solution."""
    S = GOLD["n_scenario"]
    da, rt = GOLD["day_ahead"], GOLD["real_time_first"]
    a = sum(r for _n, r, _e, _e0 in FLOWSHEET_ROWS) + 1 + LINK_ROWS
    b = PERIODIC_ROWS - LINK_ROWS
    c_da = (da["rows"] - S * (a * da["horizon"] + b)) / da["horizon"]
    c_rt = (rt["rows"] - S * (a * rt["horizon"] + b)) / rt["horizon"]
    assert c_da == c_rt == S * S


def _oracle_shape(S, T, mode="monotone"):
    rng = np.random.default_rng(3)
    cf = rng.uniform(0.1, 0.9, T)
    da = rng.uniform(10, 40, (S, T))
    rt = rng.uniform(10, 40, (S, T))
    P, _ = O.wind_battery_da_coupled(T, cf, da, rt, mode, wind_kw=147.6e3)
    return P.A.shape


@pytest.mark.parametrize("T", [4, 28])
def test_oracle_restatement_is_the_reference_problem_minus_listed_eliminations(T):
    S = GOLD["n_scenario"]
    rows, cols, _ = pyomo_shape(S, T)
    m, n = _oracle_shape(S, T)
    # columns the oracle does not carry: fixed ones, the two arc duplicates (splitter.electricity = windpower.electricity,
    # battery.elec_in = splitter.battery_elec), the link duplicates (initial_* of period t = state of period t - 1; period 0:
    # initial_state_of_charge is a constant, initial_energy_throughput stays as ONE column)
    eliminated_cols = S * T * (len(FLOWSHEET_COLS["fixed"]) + 2 + 2) - S
    assert n == cols - eliminated_cols
    # rows: arcs (2), capacity-factor row and the two power bounds become column bounds (3), the two ramp rows one ranged row (1);
    # links and the periodic remainder vanish with their columns; coupling: S (S - 1) / 2 unordered pairs instead of S * S ordered
    eliminated_rows = S * T * (2 + 3 + 1) + S * (LINK_ROWS * (T - 1) + PERIODIC_ROWS) + (S * S - S * (S - 1) // 2) * T
    assert m == rows - eliminated_rows


def test_product_coupling_rows_match_the_oracle_count():
    """The product's coupled LP (workflow/coupling.py) carries one row per unordered pair and period, as the oracle."""
    from dispatches_amd.workflow.coupling import CoupledScenarioModel

    class _LP:
        n, m = 5, 2
        indptr = np.array([0, 1, 2], np.int32)
        indices = np.array([0, 1], np.int32)
        data = np.ones(2)
        col_names = [f"c{i}" for i in range(5)]
        row_names = ["r0", "r1"]
        row_compliance = None

    class _M:
        lp, n_scenario, HOUR, pda_cols, block = _LP(), 3, range(4), [0, 1, 2, 3], None

    S, T = 3, 4
    assert CoupledScenarioModel(_M(), "monotone").lp.m == S * _LP.m + S * (S - 1) // 2 * T
    assert CoupledScenarioModel(_M(), "non_anticipative").lp.m == S * _LP.m + (S - 1) * T
