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
  * teacher forced (`check_trajectory`): a RECORDED trajectory (the GPU loop's state before every hour, its offers and what it delivered)
    is checked hour by hour: each of its LPs is rebuilt here from the recorded state, solved by HiGHS, and the recorded objective must
    reach the oracle's optimum to 1e-6 while the recorded hand-off (delivered power, next state) must lie on the oracle's optimal face
    (re-solve with those quantities fixed: same optimum to 1e-6).  That is degeneracy-proof, and nothing of the product's formulation
    enters it.
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


def roll(k, days, stride=17, series="rts_gmlc_309.npz", da_horizon=48, rt_horizon=4, year=None):
    """Free run of plant k for `days` simulated days on the oracle's own solutions.
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
        x, _ = P.solve(tight=True)
        offer, prices = x[pda][:24].copy(), da[:24].copy()
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


def _fixed(P, rows):
    """P with extra equality rows (dict col -> coef, rhs): -> optimum of the restricted LP, inf when infeasible"""
    import scipy.sparse as sp
    from scipy.optimize import linprog
    n = len(P.c)
    A = sp.lil_matrix((len(rows), n))
    b = np.zeros(len(rows))
    for i, (d, rhs) in enumerate(rows):
        for j, v in d.items():
            A[i, j] = v
        b[i] = rhs
    Aeq = sp.vstack([P.A_eq, A.tocsr()]).tocsr() if P.A_eq is not None else A.tocsr()
    beq = np.concatenate([P.b_eq, b]) if P.A_eq is not None else b
    res = linprog(P.c, A_ub=P.A_ub, b_ub=P.b_ub, A_eq=Aeq, b_eq=beq, bounds=P.bounds, method="highs")
    return float(res.fun + P.c0) if res.status == 0 else np.inf


def check_trajectory(k, rec, hours=None, stride=17, series="rts_gmlc_309.npz", da_horizon=48, rt_horizon=4, tol=1e-6, year=None):
    """Teacher-forced check of a recorded trajectory of plant k (see the module docstring).  `rec` holds, per simulated hour i:
        soc[i], thr[i]                    state BEFORE the hour (what update_model fixed: 2-dp values)
        rt_offer[i, 4]                    real-time offer = the dispatch handed to the tracker [MW]
        delivered[i]                      P_T[0] of the tracking solution [MW]
        soc_next[i], thr_next[i]          UN-rounded state of charge / throughput after the hour, from the tracking solution
        rt_obj[i], tr_obj[i]              objectives of the two hourly LPs as the solver reported them
      and per simulated day j: da_offer[j, 24], da_obj[j], da_soc[j], da_thr[j] (state the day-ahead LP was built from).
    `hours`: iterable of hour indices to check (default: all recorded).  -> dict(max_err=..., worst=..., checked=...); raises
    AssertionError on the first violation."""
    da_s, rt_s, cf_s = load_year(series) if year is None else year
    N = len(rt_s)
    start = (stride * k) % N
    n_hours = len(rec["soc"])
    hours = range(n_hours) if hours is None else hours
    worst = dict(da=0.0, rt=0.0, tr=0.0, face=0.0, state=0.0)
    rel = lambda a, b: abs(a - b) / max(1.0, abs(b))
    days_done = set()
    checked = 0
    for i in hours:
        d, h = divmod(int(i), 24)
        offer = rec["da_offer"][d]
        if d not in days_done:
            days_done.add(d)
            da, rt, cf = (window(s, start, 24 * d, da_horizon) for s in (da_s, rt_s, cf_s))
            P, fs, pda, _ = day_ahead_lp(cf, da, rt, float(rec["da_soc"][d]), float(rec["da_thr"][d]))
            f = P.solve(tight=True)[1]
            e = rel(rec["da_obj"][d], f)
            worst["da"] = max(worst["da"], e)
            assert e <= tol, ("day-ahead objective", k, d, rec["da_obj"][d], f)
            # the offer the loop took from its solution is on the optimal face of the oracle's LP
            g = _fixed(P, [({pda[t]: 1.0}, offer[t]) for t in range(24)])
            e = rel(g, f)
            worst["face"] = max(worst["face"], e)
            assert e <= tol, ("day-ahead offer off the optimal face", k, d, g, f)
        prices = window(da_s, start, 24 * d, 24)
        rt, cf, daw = (window(s, start, i, rt_horizon) for s in (rt_s, cf_s, da_s))
        known = min(rt_horizon, 24 - h)
        daw = daw.copy()
        daw[:known] = prices[h:h + known]
        cleared = np.zeros(rt_horizon)
        cleared[:known] = offer[h:h + known]
        soc, thr = float(rec["soc"][i]), float(rec["thr"][i])
        if i > 0 and (i - 1) in set(hours) if not isinstance(hours, range) else i > hours.start:
            # the state the loop fixed = the previous hour's realised state, rounded as update_model does
            e = max(abs(soc - round(float(rec["soc_next"][i - 1]), 2)), abs(thr - round(float(rec["thr_next"][i - 1]), 2)))
            worst["state"] = max(worst["state"], e)
            assert e <= 1e-9, ("state hand-off", k, i, soc, rec["soc_next"][i - 1], thr, rec["thr_next"][i - 1])
        P, fs, _, _ = real_time_lp(cf, rt, daw, cleared, known, soc, thr)
        f = P.solve(tight=True)[1]
        e = rel(rec["rt_obj"][i], f)
        worst["rt"] = max(worst["rt"], e)
        assert e <= tol, ("real-time objective", k, i, rec["rt_obj"][i], f)
        dispatch = np.asarray(rec["rt_offer"][i], float)
        g = _fixed(P, [(fs["P_T"][t][0], dispatch[t] - fs["P_T"][t][1]) for t in range(rt_horizon)])
        e = rel(g, f)
        worst["face"] = max(worst["face"], e)
        assert e <= tol, ("real-time offer off the optimal face", k, i, g, f)
        P, fs, _, _ = tracking_lp(cf, dispatch, soc, thr)
        f = P.solve(tight=True)[1]
        e = rel(rec["tr_obj"][i], f)
        worst["tr"] = max(worst["tr"], e)
        assert e <= tol, ("tracking objective", k, i, rec["tr_obj"][i], f)
        v = fs["vars"][0]
        g = _fixed(P, [(fs["P_T"][0][0], rec["delivered"][i] - fs["P_T"][0][1]), ({v["S"]: 1.0}, rec["soc_next"][i]),
                       ({v["E"]: 1.0}, rec["thr_next"][i])])
        e = rel(g, f)
        worst["face"] = max(worst["face"], e)
        assert e <= tol, ("tracking hand-off off the optimal face", k, i, g, f)
        checked += 1
    return dict(worst=worst, checked=checked)
