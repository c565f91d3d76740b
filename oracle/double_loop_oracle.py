"""CPU ORACLE (test infrastructure - NOT part of the product path): the rolling double loop of ONE wind + battery plant over
simulated days, restated on the oracle's own un-reduced LPs (dispatch_lp_oracle.py) and HiGHS.  Only tests/, tools/make_*_fixtures.py
and bench.py's cpu_baseline leg may import this.

What is restated (one plant; the product runs B of them at once in dispatches_amd/rolling.py, written separately):

    every day  : day-ahead bids      Bidder.compute_day_ahead_bids          48-h LP #1 + A.4 DA objective  
    every hour : real-time bids      Bidder.compute_real_time_bids          4-h LP, day_ahead_power fixed to the cleared dispatch for the
                                                                            hours of the cleared day, FREE for the hours past it
                 tracking            Tracker.track_market_dispatch          4-h LP #1 + A.5, dispatch = the real-time offer
                 hand-off            get_implemented_profile -> update_model: initial SOC / energy throughput re-fixed to the realised values
                                     ROUNDED to 2 dp (wind_battery_double_loop.py:194-200), clock + 1 h, capacity factors of the new window
                                     (:203-228, padded from the START of the data when the window runs past its end: the modulo below)
    forecasts  : PerfectForecaster.get_column_from_data (parametrized_bidder.py:52-58): the next `horizon` values of the series, padded
                 from the start of the data past its end - (start + hour + t) mod N here
    market     : a stub that clears every offer at its maximum (day-ahead dispatch = day-ahead offer, real-time dispatch = real-time
                 offer), as tests/test_double_loop_stub.py and rolling.py; day-ahead bids of day d are computed at hour 0 of day d

Plant k sees the year that starts at hour (stride * k) mod N of the series (BASELINE config-4 windows).

Two ways to run it:
  * free run (`roll`): the oracle's own trajectory from its own solutions.  The hourly LPs are DEGENERATE (a fifth of the prices are
    exactly zero; curtailment now or an hour later costs the same), so a second solver's trajectory may leave this one at a tie and
    never come back: free runs are compared in aggregate only.
    (`roll(..., interior_day_ahead=True)` shows it on the oracle alone: day-ahead offers from an interior point of the optimal face
    instead of a vertex move 60 days of revenue by 9e-4 and leave 44 of 60 days identical.)
  * teacher forced (tests/_rolling_oracle.py::check_recorded_plant): a RECORDED trajectory (the GPU loop's state before every hour and
    its whole solutions) is checked hour by hour: each of its LPs is rebuilt from the recorded state by the functions below and solved
    by HiGHS; the recorded solution, mapped into these un-reduced variables, must be feasible and reach the optimum to 1e-6, and every
    state hand-off must be the 2-dp rounding of what the tracker realised.  That is degeneracy-proof, and nothing of the product's
    formulation enters it.
"""
from __future__ import annotations

import numpy as np

from . import dispatch_lp_oracle as orc

WIND_KW, BATT_KW, BATT_KWH = 200e3, 25e3, 100e3
PRICE_CAP = 500.0


def load_year(series="rts_gmlc_309.npz"):
    """(da, rt, cf) of the plant's bus: prices clipped to [0, cap] as the product's WindowForecaster does."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dispatches_amd", "data", series)
    with np.load(path) as z:
        return np.clip(z["da_lmp"], 0.0, PRICE_CAP), np.clip(z["rt_lmp"], 0.0, PRICE_CAP), np.asarray(z["rt_cf"], float)


def window(series, start, hour, T):
    """parametrized_bidder.py:52-58 / wind_battery_double_loop.py:211-228: T values from `hour`, padded from the start of the data"""
    return series[(start + hour + np.arange(T)) % len(series)]


def day_ahead_lp(cf, da, rt, soc, thr):
    return orc.wind_battery_da(len(cf), cf, da, rt, WIND_KW, BATT_KW, BATT_KWH, soc0=soc, e0=thr)


def real_time_lp(cf, rt, da, cleared, known, soc, thr):
    """4-h real-time bidding LP at an hour of which `known` hours lie inside the cleared day: those carry the cleared day-ahead dispatch
    as a parameter (A.4 RT problem), the others keep day_ahead_power as a variable with the forecast day-ahead price (A.4 DA terms) -
    upstream `_pass_realized_day_ahead_dispatches` fixes only the hours it has a dispatch for."""
    T = len(cf)
    lp = orc._LP()
    fs = orc.wind_battery_rows(lp, T, cf, WIND_KW, BATT_KW, BATT_KWH, soc, thr)
    u, pda = [], []
    for t in range(T):
        b = lp.var(f"u{t}")
        u.append(b)
        d, k = fs["P_T"][t]
        if t < known:
            r = {b: 1.0}
            for j, vv in d.items():
                r[j] = r.get(j, 0.0) + vv
            lp.row(r, cleared[t] - k, np.inf)
            lp.add_cost(orc._lin(const=(rt[t] - da[t]) * cleared[t]))     # revenue of the fixed day-ahead position (a constant)
            pda.append(None)
        else:
            a = lp.var(f"pda{t}")
            pda.append(a)
            r = {a: 1.0, b: -1.0}
            for j, vv in d.items():
                r[j] = r.get(j, 0.0) - vv
            lp.row(r, -np.inf, k)
            lp.add_cost(orc._lin((a, -(da[t] - rt[t]))))
        lp.add_cost(fs["P_T"][t], -rt[t])
        lp.add_cost(fs["cost"][t], 1.0)
        lp.add_cost(orc._lin((b, orc.UNDERBID_PENALTY)))
    return orc.PreparedLP(lp), fs, u, pda


def tracking_lp(cf, dispatch, soc, thr):
    return orc.wind_battery_track(len(cf), cf, dispatch, WIND_KW, BATT_KW, BATT_KWH, soc0=soc, e0=thr)


def _pt(P, fs, x):
    return np.array([P.value(e, x) for e in fs["P_T"]])


def _solve_interior(P):
    """An optimal point from the RELATIVE INTERIOR of the optimal face (HiGHS interior point, no crossover) instead of a vertex: what a
    first-order method tends to return.  Used only to show how far two optimal trajectories of the SAME loop drift apart."""
    from . import highs_direct as hd
    M = hd.from_prepared(P, tol=1e-9)
    M.h.setOptionValue("solver", "ipm")
    M.h.setOptionValue("run_crossover", "off")
    M.h.setOptionValue("ipm_optimality_tolerance", 1e-10)
    x, f, _ = M.solve()
    return x, f


def roll(k, days, stride=17, series="rts_gmlc_309.npz", da_horizon=48, rt_horizon=4, year=None, interior_day_ahead=False):
    """Free run of plant k for `days` simulated days on the oracle's own solutions.
    interior_day_ahead: take the day-ahead offers from an interior point of the optimal face instead of a simplex vertex (see
    _solve_interior; the hourly LPs stay on the simplex).
    -> dict of per-day arrays: revenue [$], delivered [MWh], da_energy [MWh], soc / thr at the end of the day (2 dp),
       and per-hour soc / thr before the hour, delivered power."""
    da_s, rt_s, cf_s = load_year(series) if year is None else year
    N = len(rt_s)
    start = (stride * k) % N
    soc = thr = 0.0
    out = dict(revenue=np.zeros(days), delivered=np.zeros(days), da_energy=np.zeros(days), soc=np.zeros(days), thr=np.zeros(days),
               h_soc=np.zeros(24 * days), h_thr=np.zeros(24 * days), h_delivered=np.zeros(24 * days))
    for d in range(days):
        hour = 24 * d
        da, rt, cf = (window(s, start, hour, da_horizon) for s in (da_s, rt_s, cf_s))
        P, fs, pda, _ = day_ahead_lp(cf, da, rt, soc, thr)
        x, _ = _solve_interior(P) if interior_day_ahead else P.solve(tight=True)
        offer, prices = np.maximum(x[pda][:24], 0.0), da[:24].copy()
        out["da_energy"][d] = offer.sum()
        for h in range(24):
            hour = 24 * d + h
            rt, cf, daw = (window(s, start, hour, rt_horizon) for s in (rt_s, cf_s, da_s))
            known = min(rt_horizon, 24 - h)
            daw = daw.copy()
            daw[:known] = prices[h:h + known]
            cleared = np.zeros(rt_horizon)
            cleared[:known] = offer[h:h + known]
            out["h_soc"][hour], out["h_thr"][hour] = soc, thr
            P, fs, _, _ = real_time_lp(cf, rt, daw, cleared, known, soc, thr)
            x, _ = P.solve(tight=True)
            dispatch = _pt(P, fs, x)                                        # real-time offer = SCED dispatch in the stub market
            P, fs, _, _ = tracking_lp(cf, dispatch, soc, thr)
            x, _ = P.solve(tight=True)
            delivered = _pt(P, fs, x)[0]
            v = fs["vars"][0]
            soc, thr = round(float(x[v["S"]]), 2), round(float(x[v["E"]]), 2)       # wind_battery_double_loop.py:194-200
            out["h_delivered"][hour] = delivered
            out["revenue"][d] += delivered * rt[0] + offer[h] * (prices[h] - rt[0])
            out["delivered"][d] += delivered
        out["soc"][d], out["thr"][d] = soc, thr
    return out
