"""Parametrized (closed-form) bidders.

API mirror of ``dispatches/workflow/parametrized_bidder.py:73-213`` (`ParametrizedBidder`) and of its two users
``renewables_case/PEM_parametrized_bidder.py:18-122`` and ``renewables_case/battery_parametrized_bidder.py:19-123``.
These contain no optimisation; they are users of the bidder boundary and must keep working against it.
"""
from __future__ import annotations

import os

import pandas as pd

from .bidder import AbstractBidder
from .forecaster import PerfectForecaster  # noqa: F401  (re-exported like the reference module)
from .utils import convert_marginal_costs_to_actual_costs


class ParametrizedBidder(AbstractBidder):
    """Template for bidders whose DA / RT curves are closed-form functions of parameters."""

    def __init__(self, bidding_model_object, day_ahead_horizon, real_time_horizon, solver, forecaster):
        self.bidding_model_object = bidding_model_object
        self.day_ahead_horizon = day_ahead_horizon
        self.real_time_horizon = real_time_horizon
        self.n_scenario = 1
        self.solver = solver
        self.forecaster = forecaster
        self.real_time_underbid_penalty = 500
        self._check_inputs()
        self.generator = self.bidding_model_object.model_data.gen_name
        self.bids_result_list = []
        self.battery_marginal_cost = 25
        self.battery_capacity_ratio = 0.4

    def _check_solver(self):
        # closed-form bids: the solver is carried for API compatibility but never called
        pass

    def formulate_DA_bidding_problem(self):
        pass

    def formulate_RT_bidding_problem(self):
        pass

    def compute_day_ahead_bids(self, date, hour=0):
        raise NotImplementedError

    def compute_real_time_bids(self, date, hour, realized_day_ahead_prices, realized_day_ahead_dispatches,
                               tracker_profile=None):
        raise NotImplementedError

    def update_real_time_model(self, **kwargs):
        pass

    def update_day_ahead_model(self, **kwargs):
        pass

    def record_bids(self, bids, model, date, hour, market):
        self._record_bids(bids, date, hour, Market=market)
        for i in model.SCENARIOS:
            self.bidding_model_object.record_results(model.fs[i], date=date, hour=hour, Scenario=i, Market=market)

    def _record_bids(self, bids, date, hour, **kwargs):
        rows = []
        for t in bids:
            for gen in bids[t]:
                row = {"Generator": gen, "Date": date, "Hour": t}
                row.update(kwargs)
                pairs = bids[t][gen]["p_cost"]
                for idx, (power, cost) in enumerate(pairs):
                    row[f"Power {idx} [MW]"] = power
                    row[f"Cost {idx} [$]"] = cost
                for idx in range(len(pairs), self.n_scenario):
                    row[f"Power {idx} [MW]"] = None
                    row[f"Cost {idx} [$]"] = None
                rows.append(row)
        self.bids_result_list.append(pd.DataFrame(rows))

    def write_results(self, path):
        print("")
        print("Saving bidding results to disk...")
        pd.concat(self.bids_result_list).to_csv(os.path.join(path, "bidder_detail.csv"), index=False)

    @property
    def generator(self):
        return self._generator

    @generator.setter
    def generator(self, name):
        self._generator = name


def _validate_cost_curve(curve, p_min, p_max, gen, t):
    """Light-weight stand-in for egret's validate_and_clean_cost_curve: monotone power, convex cost."""
    pts = curve
    for (p0, c0), (p1, c1) in zip(pts, pts[1:]):
        if p1 < p0 - 1e-12:
            raise ValueError(f"cost curve of {gen} at t={t} is not sorted by power")
    slopes = [(c1 - c0) / (p1 - p0) for (p0, c0), (p1, c1) in zip(pts, pts[1:]) if p1 > p0]
    for s0, s1 in zip(slopes, slopes[1:]):
        if s1 < s0 - 1e-9:
            raise ValueError(f"cost curve of {gen} at t={t} is not convex")


class _TwoTierBidder(ParametrizedBidder):
    """Shared body of the PEM / battery parametrized bidders: per hour a two-tier marginal-cost curve."""

    def _tiers(self, wind_mw_available):
        raise NotImplementedError

    def _bids(self, forecast, horizon, hour, validate):
        gen = self.generator
        full_bids = {}
        for t_idx in range(horizon):
            tiers, p_max = self._tiers(forecast[t_idx] * self.wind_mw)
            cost_curve = convert_marginal_costs_to_actual_costs(tiers)
            if validate:
                _validate_cost_curve(cost_curve, 0, max(p for p, _ in cost_curve), gen, t_idx)
            full_bids[t_idx + hour] = {gen: {"p_cost": cost_curve, "p_min": 0, "p_max": p_max,
                                             "startup_capacity": p_max, "shutdown_capacity": p_max}}
        return full_bids

    def compute_day_ahead_bids(self, date, hour=0):
        forecast = self.forecaster.forecast_day_ahead_capacity_factor(date, hour, self.generator, self.day_ahead_horizon)
        bids = self._bids(forecast, self.day_ahead_horizon, hour, validate=True)
        self._record_bids(bids, date, hour, Market="Day-ahead")
        return bids

    def compute_real_time_bids(self, date, hour, realized_day_ahead_prices, realized_day_ahead_dispatches,
                               tracker_profile=None):
        forecast = self.forecaster.forecast_real_time_capacity_factor(date, hour, self.generator, self.day_ahead_horizon)
        bids = self._bids(forecast, self.real_time_horizon, hour, validate=False)
        self._record_bids(bids, date, hour, Market="Real-time")
        return bids


class PEMParametrizedBidder(_TwoTierBidder):
    """Wind + PEM: 0 $/MWh up to (wind - PEM capacity) MW, `pem_marginal_cost` above, max bid = available wind
    (reference PEM_parametrized_bidder.py:50-122)."""

    def __init__(self, bidding_model_object, day_ahead_horizon, real_time_horizon, solver, forecaster,
                 pem_marginal_cost, pem_mw):
        super().__init__(bidding_model_object, day_ahead_horizon, real_time_horizon, solver, forecaster)
        self.wind_marginal_cost = 0
        self.wind_mw = self.bidding_model_object._wind_pmax_mw
        self.pem_marginal_cost = pem_marginal_cost
        self.pem_mw = pem_mw

    def _tiers(self, wind):
        grid_wind = max(0, wind - self.pem_mw)
        return [(0, 0), (grid_wind, 0), (wind, self.pem_marginal_cost)], wind


class FixedParametrizedBidder(_TwoTierBidder):
    """Wind + battery: the storable part of the wind is bid at `storage_marginal_cost`
    (reference battery_parametrized_bidder.py:47-123)."""

    def __init__(self, bidding_model_object, day_ahead_horizon, real_time_horizon, solver, forecaster,
                 storage_marginal_cost, storage_mw):
        super().__init__(bidding_model_object, day_ahead_horizon, real_time_horizon, solver, forecaster)
        self.wind_marginal_cost = 0
        self.wind_mw = self.bidding_model_object._wind_pmax_mw
        self.storage_marginal_cost = storage_marginal_cost
        self.storage_mw = storage_mw

    def _tiers(self, wind):
        p_max = max(wind, self.storage_mw)
        return [(0, 0), (max(0, wind - self.storage_mw), 0), (p_max, self.storage_marginal_cost)], p_max
