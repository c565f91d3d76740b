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
