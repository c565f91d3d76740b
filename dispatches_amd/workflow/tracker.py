"""Tracker: follow a market dispatch signal at minimum cost.

Restates the upstream ``idaes.apps.grid_integration.tracker.Tracker`` that DISPATCHES instantiates at
``renewables_case/run_double_loop_battery.py:255-285`` and tests at
``tests/test_multiperiod_wind_battery_doubleloop.py:66-85``: same constructor, ``track_market_dispatch``,
``model.fs``, ``power_output[t]``, ``update_model``, ``get_last_delivered_power``, ``write_results``.

    min  sum_t  w*tot_cost[t] + 1e4*(under[t] + over[t])
    s.t. flowsheet rows,   P_T[t] + under[t] - over[t] = dispatch[t]   (t < len(dispatch))
"""
from __future__ import annotations

import os

import numpy as np
import pandas as pd

from ..lp import LinearBlock, LinExpr
from .batch_model import ScenarioBatchModel


class _PowerOutput:
    """`tracker.power_output[t]` -> the P_T[t] expression of the tracked block (value via `pyo_value`)."""

    def __init__(self, tracker):
        self._tracker = tracker

    def __getitem__(self, t):
        return self._tracker.model.fs.expressions[self._tracker.tracking_model_object.power_output][t]


class Tracker:
    deviation_penalty = 10000.0

    def __init__(self, tracking_model_object, tracking_horizon, n_tracking_hour, solver, warm_start=True):
        self.tracking_model_object = tracking_model_object
        self.tracking_horizon = tracking_horizon
        self.n_tracking_hour = n_tracking_hour
        self.solver = solver
        self.warm_start = bool(warm_start)      # rolling-horizon warm start when the solver supports it (HIP solver)
        self._check_inputs()
        self.projection = None
        self.result_list = []
        self.daily_stats = None
        self.formulate_tracking_problem()

    def _check_inputs(self):
        for m in ("populate_model", "get_implemented_profile", "update_model", "get_last_delivered_power",
                  "record_results", "write_results"):
            if not callable(getattr(self.tracking_model_object, m, None)):
                raise AttributeError(f"Tracking model object does not have the required method {m}().")
        for a in ("power_output", "total_cost"):
            if not hasattr(self.tracking_model_object, a):
                raise AttributeError(f"Tracking model object does not have the required attribute '{a}'.")
        for name in ("tracking_horizon", "n_tracking_hour"):
            v = getattr(self, name)
            if not isinstance(v, int):
                raise TypeError(f"{name} should be an integer, but a {type(v).__name__} was given.")
            if v <= 0:
                raise ValueError(f"{name} should be greater than zero, but {v} was given.")
        if self.n_tracking_hour > self.tracking_horizon:
            raise ValueError("n_tracking_hour cannot exceed tracking_horizon")
        if not callable(getattr(self.solver, "solve", None)):
            raise TypeError("The provided solver must expose solve(model, tee=...).")

    def formulate_tracking_problem(self):
        block = LinearBlock("fs")
        self.tracking_model_object.populate_model(block, self.tracking_horizon)
        model = ScenarioBatchModel(block, 1, self.tracking_horizon, indexed=False)
        P_T = block.expressions[self.tracking_model_object.power_output]
        model.power_underdelivered, model.power_overdelivered, model.tracking_rows = [], [], []
        for t in model.HOUR:
            under = block.var(f"power_underdelivered[{t}]")
            over = block.var(f"power_overdelivered[{t}]")
            # until a dispatch is passed the row is free (lo=-inf, hi=+inf) -- it is never presolved away
            row = block.constraint(f"tracking_dispatch_constraints[{t}]", P_T[t] + under - over, -np.inf, np.inf,
                                   mutable=True)
            model.power_underdelivered.append(under)
            model.power_overdelivered.append(over)
            model.tracking_rows.append(row)
        # the dispatch rows P_T[MW] + under - over = D mix 1e-3 (kW -> MW) and 1.0 coefficients: ask the solver for
        # geometric pre-equilibration (include/dsp_hip.h: geo_iters) and the conservative restart / ray-jump settings
        # that these tiny, badly scaled LPs need; model objects may override via `solver_hints`
        model.solver_hints = {"geo_iters": 8, "check_every": 32, "jump_tol": 1e-3, "pid_kp": 0.5,
                              **(getattr(self.tracking_model_object, "solver_hints", None) or {})}
        self.model = model
        cost_name, weight = self.tracking_model_object.total_cost
        model._tot_cost_family, model.cost_weight = cost_name, weight
        self._refresh_objective()
        self.power_output = _PowerOutput(self)

    def _refresh_objective(self):
        model = self.model
        block = model.block
        obj = LinExpr()
        for t in model.HOUR:
            obj = obj + block.expressions[model._tot_cost_family][t] * model.cost_weight
            obj = obj + (model.power_underdelivered[t] + model.power_overdelivered[t]) * self.deviation_penalty
        if model.lp is None:
            model.finalize(obj)
            P_T = block.expressions[self.tracking_model_object.power_output]
            model.PT_const = np.array([P_T[t].const for t in model.HOUR])
        model.c = obj.dense(model.lp.n)[None, :]
        model.c0 = np.array([obj.const])

    def _pass_market_dispatch(self, market_dispatch):
        model = self.model
        for t in model.HOUR:
            row = model.tracking_rows[t]
            if t < len(market_dispatch) and market_dispatch[t] is not None:
                rhs = float(market_dispatch[t]) - model.PT_const[t]
                model.block.set_row_bounds(row, rhs, rhs)
            else:
                model.block.set_row_bounds(row, -np.inf, np.inf)

    def track_market_dispatch(self, market_dispatch, date, hour):
        """Solve the tracking LP for the given dispatch [MW per hour], record, and roll the model forward by
        `n_tracking_hour` implemented steps."""
        self._pass_market_dispatch(market_dispatch)
        if getattr(self.solver, "supports_warm_start", False) and self.warm_start and self.model.x is not None:
            # rolling horizon: this hour's LP is the previous one shifted by the implemented steps
            self.solver.solve(self.model, tee=False, warm_start=True, shift=self.n_tracking_hour)
        else:
            self.solver.solve(self.model, tee=False)
        status = np.asarray(self.model.status).copy()
        from ..hip_solver import STATUS_UNCERTIFIED, uncertified
        status[uncertified(status, getattr(self.model, "flags", None))] = STATUS_UNCERTIFIED
        if not (status == 0).all():
            # an unconverged (or NaN, or uncertified) solution must never become the implemented profile / the next fixed SOC
            raise RuntimeError(f"tracking problem ({date}, hour {hour}) did not reach optimality: solver status "
                               f"{status.tolist()} (0 optimal, 1 iteration limit, 2 infeasible input, 4 numerical, "
                               f"5 objective accuracy not certified)")
        self.record_results(date=date, hour=hour)
        profiles = self.tracking_model_object.get_implemented_profile(
            b=self.model.fs, last_implemented_time_step=self.n_tracking_hour - 1)
        self._record_daily_stats(profiles)
        return profiles

    def update_model(self, **profiles):
        self.tracking_model_object.update_model(self.model.block, **profiles)
        self._refresh_objective()

    def _record_daily_stats(self, profiles):
        if self.daily_stats is None:
            self.daily_stats = {}
        for k, v in profiles.items():
            self.daily_stats.setdefault(k, []).extend(v)
        # keep one day
        for k in self.daily_stats:
            self.daily_stats[k] = self.daily_stats[k][-24:]

    def get_last_delivered_power(self):
        return self.tracking_model_object.get_last_delivered_power(
            b=self.model.fs, last_implemented_time_step=self.n_tracking_hour - 1)

    def record_results(self, **kwargs):
        self.tracking_model_object.record_results(self.model.fs, **kwargs)
        date, hour = kwargs.get("date"), kwargs.get("hour")
        fs = self.model.fs
        P_T = fs.expressions[self.tracking_model_object.power_output]
        rows = []
        for t in self.model.HOUR:
            lo = fs.row_lo[self.model.tracking_rows[t]]
            rows.append({
                "Date": date, "Hour": hour, "Horizon [hr]": int(t),
                "Power Dispatch [MW]": None if not np.isfinite(lo) else round(lo + self.model.PT_const[t], 2),
                "Power Output [MW]": round(fs.value(P_T[t]), 2),
                "Power Underdelivered [MW]": round(self.model.power_underdelivered[t].value, 2),
                "Power Overdelivered [MW]": round(self.model.power_overdelivered[t].value, 2),
            })
        self.result_list.append(rows)              # row dicts; the frame is built once in write_results

    def write_results(self, path):
        print("")
        print("Saving tracking results to disk...")
        pd.DataFrame([row for rows in self.result_list for row in rows]).to_csv(
            os.path.join(path, "tracker_detail.csv"), index=False)
        self.tracking_model_object.write_results(path=os.path.join(path, "tracking_model_detail.csv"))
