"""Shared by tests/test_hip_rolling.py (GPU) and tests/test_rolling_cpu.py (HiGHS stand-in backend): the oracle-anchored check of
the device-resident double loop."""
import numpy as np


def check_rolling_hours_against_the_oracle(loop, hours, stride):
    """Oracle-anchored check of the device-resident loop (round-2 review: the tests above compare the HIP loop with HIP host
    objects).  For the first hours of a simulated day and every plant of a small batch, the oracle's OWN real-time bidding LP and
    tracking LP are built from the loop's state at that hour - realised state of charge / throughput before the hour, capacity-factor
    and price windows, the cleared day-ahead dispatch, the real-time offer the loop handed to its tracker - and the solution the
    device loop computed is mapped into the oracle's variables: it must be feasible for the oracle's rows and reach the oracle's
    optimal objective (HiGHS) to 1e-6.  Nothing of the product's LP formulation enters the comparison."""

    from dispatches_amd import scenarios
    from oracle import dispatch_lp_oracle as orc
    T = 4
    B = loop.B
    offers = loop.day_ahead().cpu().numpy()
    s = scenarios.load_series("rts_gmlc_309.npz")
    N = len(s["rt_lmp"])
    rt_series, cf_series = np.clip(s["rt_lmp"], 0.0, 500.0), s["rt_cf"]

    def mapped(P, fs, per, x, extra):
        """the product's solution x in the oracle's variable order (+ the bidding / tracking slacks the oracle carries)"""
        z = np.zeros(len(P.c))
        for t in range(T):
            v, p = fs["vars"][t], per[t]
            for key, col in (("W", "wind"), ("G", "grid_elec"), ("I", "elec_in"), ("O", "elec_out"), ("S", "state_of_charge"), ("E", "energy_throughput")):
                z[v[key]] = x[p[col].index]
        for j, val in extra:
            z[j] = val
        return z

    def check(P, z, what):
        f_ref = P.solve(tight=True)[1]
        Az = P.A @ z
        scale = 1.0 + np.abs(z).max()
        assert (Az >= P.lo - 1e-7 * scale).all() and (Az <= P.hi + 1e-7 * scale).all(), what
        assert (z >= P.lb - 1e-7 * scale).all() and (z <= P.ub + 1e-7 * scale).all(), what
        f = float(P.c @ z + P.c0)
        assert abs(f - f_ref) <= 1e-6 * max(1.0, abs(f_ref)), (what, f, f_ref)

    for h in range(hours):
        soc0, thr0 = loop.soc.cpu().numpy().copy(), loop.thr.cpu().numpy().copy()
        loop.hour_step()
        x_rt, x_tr = loop.rt.out["x"].cpu().numpy(), loop.tr.out["x"].cpu().numpy()
        assert int(loop.rt.out["status"].abs().sum().item()) == 0 and int(loop.tr.out["status"].abs().sum().item()) == 0
        for k in range(B):
            idx = ((stride * k) % N + h + np.arange(T)) % N
            cf, rt = cf_series[idx], rt_series[idx]
            da_disp = offers[k, h:h + T]
            # real-time bidding LP of the oracle for this state
            P, fs, u = orc.wind_battery_rt(T, cf, rt, da_disp, soc0=soc0[k], e0=thr0[k])
            xk = x_rt[k]
            pt = np.array([1e-3 * (xk[p["grid_elec"].index] + xk[p["elec_out"].index]) for p in loop.rt_periods])
            z = mapped(P, fs, loop.rt_periods, xk, [(u[t], max(0.0, da_disp[t] - pt[t])) for t in range(T)])
            check(P, z, ("rt", h, k))
            # tracking LP of the oracle: dispatch = the un-rounded real-time offer the loop passed on
            P, fs, under, over = orc.wind_battery_track(T, cf, pt, soc0=soc0[k], e0=thr0[k])
            xk = x_tr[k]
            ptt = np.array([1e-3 * (xk[p["grid_elec"].index] + xk[p["elec_out"].index]) for p in loop.tr_periods])
            extra = [(under[t], max(0.0, pt[t] - ptt[t])) for t in range(T)] + [(over[t], max(0.0, ptt[t] - pt[t])) for t in range(T)]
            check(P, mapped(P, fs, loop.tr_periods, xk, extra), ("track", h, k))
    res, ok = loop.results()
    assert ok


# ---- the whole year: teacher-forced check of a RECORDED trajectory (BASELINE config 4) ----------------------------------------------
_PHYS = (("W", "wind"), ("G", "grid_elec"), ("I", "elec_in"), ("O", "elec_out"), ("S", "state_of_charge"), ("E", "energy_throughput"))


def column_maps(loop):
    """Plain-integer description of where the three LPs of a BatchedWindBatteryDoubleLoop keep their physical columns (picklable:
    the checks below run in worker processes)."""
    per = lambda periods: np.array([[p[col].index for _, col in _PHYS] for p in periods], dtype=np.int64)
    return dict(da=per(loop.da_periods), rt=per(loop.rt_periods), tr=per(loop.tr_periods),
                da_pda=loop.da.pda_cols.cpu().numpy().astype(np.int64), rt_pda=loop.rt.pda_cols.cpu().numpy().astype(np.int64),
                da_u=loop.da.u_cols.cpu().numpy().astype(np.int64), rt_u=loop.rt.u_cols.cpu().numpy().astype(np.int64))


def _mapped(P, fs, cols, x, extra):
    z = np.zeros(len(P.c))
    for t in range(len(cols)):
        v = fs["vars"][t]
        for j, (key, _) in enumerate(_PHYS):
            z[v[key]] = x[cols[t, j]]
    for j, val in extra:
        z[j] = val
    return z


def _feasible_and_optimal(P, z, what, tol=1e-6):
    """z is feasible for the oracle's rows and bounds and reaches the oracle's optimum (HiGHS) to tol; -> relative objective gap"""
    f_ref = P.solve(tight=True)[1]
    Az = P.A @ z
    scale = 1.0 + np.abs(z).max()
    assert (Az >= P.lo - 1e-7 * scale).all() and (Az <= P.hi + 1e-7 * scale).all(), (what, "rows")
    assert (z >= P.lb - 1e-7 * scale).all() and (z <= P.ub + 1e-7 * scale).all(), (what, "bounds")
    f = float(P.c @ z + P.c0)
    gap = abs(f - f_ref) / max(1.0, abs(f_ref))
    assert gap <= tol, (what, f, f_ref)
    return gap


def check_recorded_plant(args, base_hour=0, base_day=0):
    """Every recorded LP of ONE plant over the hours `hours`, against the oracle's own LPs of the recorded state (oracle/double_loop_oracle.py):
    the day-ahead LP of every day touched, the real-time bidding LP and the tracking LP of every hour - each solution, mapped into the
    oracle's variables, must be feasible for the oracle's rows and optimal to 1e-6 - and the state hand-off itself: the state an hour
    starts from is the previous hour's tracking solution rounded to 2 dp (wind_battery_double_loop.py:194-200), the tracker's dispatch is
    the real-time offer, the day-ahead position of an hour is that day's offer.
    args = (plant id k, stride, maps, rec arrays of this plant, hours) -> dict(worst gaps, per-day revenue / delivered / soc).
    base_hour / base_day: the simulated hour / day the first row of the hourly / daily arrays belongs to (a block of a longer run)."""
    from oracle import double_loop_oracle as dl
    k, stride, maps, rec, hours = args
    year = dl.load_year()
    da_s, rt_s, cf_s = year
    N = len(rt_s)
    start = (stride * k) % N
    T = maps["rt"].shape[0]
    Tda = maps["da"].shape[0]
    pt = lambda cols, x: 1e-3 * (x[cols[:, 1]] + x[cols[:, 3]])
    worst = dict(da=0.0, rt=0.0, tr=0.0)
    hours = sorted(int(i) for i in hours)
    n_days = len(rec["da_obj"])
    revenue, delivered_mwh, soc_end = np.zeros(n_days), np.zeros(n_days), np.full(n_days, np.nan)
    checked_days = set()
    H = lambda i: i - base_hour                    # row of an hourly array
    for i in hours:
        d, h = divmod(i, 24)
        d -= base_day                               # row of a daily array; 24 * (d + base_day) + h = i
        x_da = rec["da_x"][d]
        offer = x_da[maps["da_pda"]][:24]
        prices = dl.window(da_s, start, 24 * (d + base_day), 24)
        if d not in checked_days:
            checked_days.add(d)
            da, rt, cf = (dl.window(s, start, 24 * (d + base_day), Tda) for s in year)
            soc, thr = rec["da_state"][d]
            P, fs, pda, u = dl.day_ahead_lp(cf, da, rt, float(soc), float(thr))
            ptd = pt(maps["da"], x_da)
            xp = x_da[maps["da_pda"]]
            # (the underbid slack is the solver's own column, not max(0, pda - P_T) recomputed: a row residual of 1e-6 MW inside the
            #  solver's feasibility tolerance would enter the objective 1e4-fold through the penalty; the row test below still holds it
            #  to 1e-7 of the solution's scale)
            xu = x_da[maps["da_u"]]
            extra = [(pda[t], xp[t]) for t in range(Tda)] + [(u[t], xu[t]) for t in range(Tda)]
            worst["da"] = max(worst["da"], _feasible_and_optimal(P, _mapped(P, fs, maps["da"], x_da, extra), ("da", k, d + base_day)))
            # the day starts from the state the loop carried there
            assert (rec["da_state"][d] == rec["state"][H(24 * (d + base_day))]).all(), ("day-ahead state", k, d + base_day)
        soc, thr = (float(v) for v in rec["state"][H(i)])
        if H(i) > 0:
            prev = rec["tr_x"][H(i) - 1]
            # the state this hour starts from is a 2-dp value within half a cent of what the tracker realised the hour before
            # (the device rounds half away from zero on the binary value, Python's round half to even on the decimal one)
            for got, col in ((soc, 4), (thr, 5)):
                real = float(prev[maps["tr"][0, col]])
                assert abs(got - real) <= 0.005 + 1e-9 * max(1.0, abs(real)) and abs(got * 100 - round(got * 100)) <= 1e-6 * max(1.0, abs(got)), ("state hand-off", k, i, got, real)
        rt, cf, daw = (dl.window(s, start, i, T) for s in (rt_s, cf_s, da_s))
        known = min(T, 24 - h)
        daw = daw.copy()
        daw[:known] = prices[h:h + known]
        cleared = np.zeros(T)
        cleared[:known] = offer[h:h + known]
        x = rec["rt_x"][H(i)]
        P, fs, u, pda = dl.real_time_lp(cf, rt, daw, cleared, known, soc, thr)
        ptr = pt(maps["rt"], x)
        xp = x[maps["rt_pda"]]
        assert np.abs(xp[:known] - cleared[:known]).max() <= 1e-9 * 225, ("cleared day-ahead position", k, i)
        xu = x[maps["rt_u"]]                       # (the solver's own slack columns, as for the day-ahead LP above)
        extra = [(u[t], xu[t]) for t in range(T)] + [(pda[t], xp[t]) for t in range(known, T)]
        worst["rt"] = max(worst["rt"], _feasible_and_optimal(P, _mapped(P, fs, maps["rt"], x, extra), ("rt", k, i)))
        x = rec["tr_x"][H(i)]
        P, fs, under, over = dl.tracking_lp(cf, ptr, soc, thr)
        ptt = pt(maps["tr"], x)
        extra = [(under[t], max(0.0, ptr[t] - ptt[t])) for t in range(T)] + [(over[t], max(0.0, ptt[t] - ptr[t])) for t in range(T)]
        worst["tr"] = max(worst["tr"], _feasible_and_optimal(P, _mapped(P, fs, maps["tr"], x, extra), ("tr", k, i)))
        revenue[d] += ptt[0] * rt[0] + offer[h] * (prices[h] - rt[0])
        delivered_mwh[d] += ptt[0]
        if h == 23:
            soc_end[d] = round(float(x[maps["tr"][0, 4]]), 2)
    return dict(plant=k, worst=worst, hours=len(hours), revenue=revenue, delivered=delivered_mwh, soc=soc_end)


# ---- the generic loop (dispatches_amd/rolling_flowsheets.py): nuclear and wind + PEM against the oracle's own hourly LPs ----------------
def check_flowsheet_hours_against_the_oracle(loop, hours):
    """First `hours` hours of a simulated day of a BatchedDoubleLoop for "nuclear" or "wind_pem": the oracle's real-time bidding LP and
    tracking LP (oracle/dispatch_lp_oracle.py: nuclear_rt / nuclear_track, wind_pem_rt / wind_pem_track - un-reduced rows) are built from
    the loop's state at that hour (realised holdup, windows, cleared day-ahead dispatch, the offer handed to the tracker) and their
    optimal objectives (HiGHS) must equal the loop's (solver objective + the objective constant it maintains) to 1e-6; the state the
    next hour starts from must be the tracker's realised value rounded as update_model does."""
    from oracle import dispatch_lp_oracle as orc
    B = loop.B
    offers = loop.day_ahead().cpu().numpy()
    da_prices = loop.da_prices.cpu().numpy()
    rt_s = loop.rt_series.cpu().numpy()
    cf_s = loop.cf_series.cpu().numpy() if loop.cf_series is not None else None
    start = loop.start.cpu().numpy()
    N = loop.N
    Trt, Ttr = loop.rt.T, loop.tr.T
    worst = 0.0
    for h in range(hours):
        assert h + Trt <= 24, "the oracle's real-time LP fixes every hour of its horizon: stay inside the cleared day"
        state0 = loop.state.cpu().numpy().copy()
        loop.hour_step()
        x_rt = loop.rt.out["x"].cpu().numpy()
        obj_rt = (loop.rt.out["obj"] + loop.rt.c0).cpu().numpy()
        obj_tr = (loop.tr.out["obj"] + loop.tr.c0).cpu().numpy()
        x_tr = loop.tr.out["x"].cpu().numpy()
        assert int(loop.rt.out["status"].abs().sum().item()) == 0 and int(loop.tr.out["status"].abs().sum().item()) == 0
        PT, PTc = loop.rt.PT.cpu().numpy(), loop.rt.PT_const.cpu().numpy()
        for k in range(B):
            idx = (start[k] + h + np.arange(Trt)) % N
            rt = rt_s[idx]
            cleared = offers[k, h:h + Trt]
            offer = x_rt[k] @ PT.T + PTc                                         # what the loop handed to its tracker
            if loop.flowsheet == "nuclear":
                P = orc.nuclear_rt(Trt, rt, cleared, holdup0=float(state0[k, 0]))[0]
                Q = orc.nuclear_track(Ttr, offer[:Ttr], holdup0=float(state0[k, 0]))[0]
            else:
                kw = loop.rt.wind[1]
                P = orc.wind_pem_rt(Trt, cf_s[idx], rt, cleared, wind_kw=kw)[0]
                Q = orc.wind_pem_track(Ttr, cf_s[idx][:Ttr], offer[:Ttr], wind_kw=kw)[0]
            # (the product keeps day_ahead_power as a fixed column: its objective carries - DA x cleared, the oracle's A.4 RT form does not)
            ref_rt = P.solve(tight=True)[1] - float(da_prices[k, h:h + Trt] @ cleared)
            ref_tr = Q.solve(tight=True)[1]
            for got, ref, what in ((obj_rt[k], ref_rt, "rt"), (obj_tr[k], ref_tr, "track")):
                gap = abs(got - ref) / max(1.0, abs(ref))
                worst = max(worst, gap)
                assert gap <= 1e-6, (loop.flowsheet, what, h, k, got, ref)
        state1 = loop.state.cpu().numpy()
        for j, col in enumerate(loop.tr.state_real):
            real = x_tr[:, col]
            assert np.abs(state1[:, j] - real).max() <= 0.5 / loop.scale[j] + 1e-9 * max(1.0, np.abs(real).max()), (loop.flowsheet, "state hand-off", h)
            assert np.abs(state1[:, j] * loop.scale[j] - np.round(state1[:, j] * loop.scale[j])).max() <= 1e-6 * max(1.0, np.abs(state1[:, j]).max() * loop.scale[j])
    res, ok = loop.results()
    assert ok
    return worst
