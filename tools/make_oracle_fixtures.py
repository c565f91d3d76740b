"""Generate the oracle fixtures the GPU parity tests compare against (no oracle solve on the GPU box needed):

  tests/golden/oracle_objectives.npz     objectives of the first B scenarios of every day-ahead bench workload
  tests/golden/oracle_setpoints.npz      per (scenario, hour): the range [lo, lo + width] of P_T[t] and of
                                         day_ahead_power[t] over the OPTIMAL FACE of the scenario's LP, more precisely
                                         over all feasible points whose objective is within 1e-7 relative of the
                                         optimum (width ~0 = the setpoint is unique).  RTS-GMLC prices repeat and are exactly 0 for hours, so
                                         many hourly setpoints are not unique; a simplex code returns one vertex of
                                         the face, a first-order method another point of it.
  tests/golden/oracle_hourly.npz         the hourly LPs of the double loop at batch scale (4-h real-time bids, 12-h
                                         nuclear real-time bids, 4-h tracking for the three flowsheets): seeded inputs
                                         (capacity factors, prices, realised dispatch, initial state), oracle objective
                                         and P_T[t] face range per scenario.

All from the INDEPENDENT CPU oracle (oracle/dispatch_lp_oracle.py: un-reduced Appendix-A LPs) solved by HiGHS (dual
simplex, feasibility tolerances 1e-9) through oracle/highs_direct.py.     python tools/make_oracle_fixtures.py [B]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


FACE_EPS = 1e-7


def _solve_with_ranges(P, fs, extra=()):
    """objective + face range of P_T[t] (and of the `extra` single columns, e.g. day_ahead_power[t])."""
    from oracle import highs_direct as hd
    M = hd.from_prepared(P)
    _, f, _ = M.solve()
    T = len(fs["P_T"])
    ex = [fs["P_T"][t] for t in range(T)] + [({j: 1.0}, 0.0) for j in extra]
    # the face of the solutions whose objective is within FACE_EPS (relative) of the optimum: a tenth of the 1e-6
    # objective contract and exactly the objective accuracy the solver guarantees (dsp_options::eps_obj): a solver that
    # promises the objective to eps can only promise membership of the eps-optimal face.  (With an exact face the GPU
    # batches of round 2 left it by <= 2e-4 MW on 0.03 % of the entries and by 12 MW in ONE nuclear hour whose DA and RT
    # prices differ by 1e-4 $/MWh; with eps = 1e-8 by <= 0.01 MW on 8 of 8e5 entries.)  An hour whose DA and RT prices differ by 1e-4 $/MWh is, for every purpose of the contract,
    # as indifferent as one where they are equal; with an exact face its setpoint would count as "unique".
    R = M.face_ranges(ex, slack=FACE_EPS * max(1.0, abs(f)))
    return f, R


# ---- day-ahead workloads (scenarios.WORKLOADS) ---------------------------------------------------------------
def _da_work(args):
    wl, ids = args
    from dispatches_amd import scenarios
    from oracle import dispatch_lp_oracle as orc
    T = 48 if wl.endswith("48h") else 24
    obj, lo, wd = [], [], []
    if wl.startswith("wind"):
        s = scenarios.load_series("rts_gmlc_309.npz" if "battery" in wl else "rts_gmlc_303.npz")
        N = len(s["rt_lmp"])
        stride = 17 if "battery" in wl else 37
    else:
        da_all, rt_all = scenarios.nuclear_prices(max(4096, max(ids) + 1), T)
    for k in ids:
        if wl.startswith("wind"):
            h0 = (stride * k) % (N - T)
            da, rt = np.clip(s["da_lmp"][h0:h0 + T], 0, 500), np.clip(s["rt_lmp"][h0:h0 + T], 0, 500)
            cf = s["rt_cf"][h0:h0 + T]
            P, fs, pda, u = (orc.wind_battery_da(T, cf, da, rt) if "battery" in wl
                             else orc.wind_pem_da(T, cf, da, rt, wind_kw=847e3))
        else:
            P, fs, pda, u = orc.nuclear_da(T, da_all[k], rt_all[k])
        f, R = _solve_with_ranges(P, fs, pda)
        obj.append(f)
        lo.append(R[:, 0])
        wd.append(R[:, 1] - R[:, 0])
    return np.array(obj), np.array(lo), np.array(wd)


# ---- hourly LPs at batch scale: seeded inputs -------------------------------------------------------------------
def hourly_inputs(case, B, seed=20200102):
    """Inputs of the hourly-LP fixtures (stored in the fixture file; the GPU tests read them from there)."""
    from dispatches_amd import scenarios
    rng = np.random.default_rng([seed, sum(map(ord, case))])
    k = np.arange(B)
    if case.startswith("nuclear"):
        T = 12 if case == "nuclear_rt12" else 4
        da, rt = scenarios.nuclear_prices(B, 24)
        hmax = 5000.0 / 2.016e-3
        inp = dict(rt=rt[:, :T], da=da[:, :T],
                   holdup0=np.round(rng.uniform(0.0, hmax, B)),                 # update_model fixes round(holdup)
                   dispatch=np.round(rng.uniform(395.0, 500.0, (B, T)), 4))
    else:
        T = 4
        s = scenarios.load_series("rts_gmlc_309.npz")
        N = len(s["rt_lmp"])
        h0 = (17 * k) % (N - T)
        idx = h0[:, None] + np.arange(T)[None, :]
        cf = s["rt_cf"][idx]
        avail = 200.0 * cf                                                       # MW of wind available
        inp = dict(cf=cf, rt=np.clip(s["rt_lmp"][idx], 0, 500), da=np.clip(s["da_lmp"][idx], 0, 500),
                   dispatch=np.round(rng.uniform(0.0, 1.15, (B, T)) * (avail + 25.0 * ("battery" in case)), 4))
        if "battery" in case:
            inp["soc0"] = np.round(rng.uniform(0.0, 100e3, B), 2)                 # update_model rounds to 2 dp
            inp["e0"] = np.round(rng.uniform(0.0, 5e5, B), 2)
    return T, inp


HOURLY_CASES = ("wind_battery_rt4", "wind_pem_rt4", "nuclear_rt12", "wind_battery_track4", "wind_pem_track4",
                "nuclear_track4")


def _hourly_work(args):
    case, ids, T, inp = args
    from oracle import dispatch_lp_oracle as orc
    obj, lo, wd = [], [], []
    for i, k in enumerate(ids):
        g = {key: v[i] for key, v in inp.items()}
        if case == "wind_battery_rt4":
            P, fs, _ = orc.wind_battery_rt(T, g["cf"], g["rt"], g["dispatch"], soc0=g["soc0"], e0=g["e0"])
        elif case == "wind_pem_rt4":
            P, fs, _ = orc.wind_pem_rt(T, g["cf"], g["rt"], g["dispatch"])
        elif case == "nuclear_rt12":
            P, fs, _ = orc.nuclear_rt(T, g["rt"], g["dispatch"], holdup0=g["holdup0"])
        elif case == "wind_battery_track4":
            P, fs, *_ = orc.wind_battery_track(T, g["cf"], g["dispatch"], soc0=g["soc0"], e0=g["e0"])
        elif case == "wind_pem_track4":
            P, fs, *_ = orc.wind_pem_track(T, g["cf"], g["dispatch"])
        elif case == "nuclear_track4":
            P, fs, *_ = orc.nuclear_track(T, g["dispatch"], holdup0=g["holdup0"])
        else:
            raise KeyError(case)
        f, R = _solve_with_ranges(P, fs)
        obj.append(f)
        lo.append(R[:, 0])
        wd.append(R[:, 1] - R[:, 0])
    return np.array(obj), np.array(lo), np.array(wd)


def _width32(w):
    """widths as float32, rounded UP (a stored range never excludes a point of the true one)"""
    w32 = np.maximum(w, 0.0).astype(np.float32)
    return np.where(w32 < w, np.nextafter(w32, np.float32(np.inf)), w32).astype(np.float32)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    only = sys.argv[2:]                       # optional: restrict to named workloads / cases
    from dispatches_amd import scenarios
    procs = os.cpu_count() or 1
    split = lambda n: [c for c in np.array_split(np.arange(n), procs * 8) if len(c)]
    with mp.get_context("spawn").Pool(procs) as pool:
        objs, setp = {}, {}
        for wl in scenarios.WORKLOADS:
            if only and wl not in only:
                continue
            T = 48 if wl.endswith("48h") else 24
            out = pool.map(_da_work, [(wl, c.tolist()) for c in split(B)])
            objs[wl] = np.concatenate([o[0] for o in out])
            lo = np.concatenate([o[1] for o in out])
            wd = np.concatenate([o[2] for o in out])
            setp[f"{wl}/P_T_lo"], setp[f"{wl}/P_T_width"] = lo[:, :T], _width32(wd[:, :T])
            setp[f"{wl}/pda_lo"], setp[f"{wl}/pda_width"] = lo[:, T:], _width32(wd[:, T:])
            print(wl, objs[wl][:2], "unique P_T", float((wd[:, :T] < 1e-9).mean()), "unique pda",
                  float((wd[:, T:] < 1e-9).mean()), flush=True)
        hourly = {}
        for case in HOURLY_CASES:
            if only and case not in only:
                continue
            T, inp = hourly_inputs(case, B)
            chunks = [(case, c.tolist(), T, {k: v[c] for k, v in inp.items()}) for c in split(B)]
            out = pool.map(_hourly_work, chunks)
            for k, v in inp.items():
                hourly[f"{case}/{k}"] = v
            hourly[f"{case}/obj"] = np.concatenate([o[0] for o in out])
            lo, wd = np.concatenate([o[1] for o in out]), np.concatenate([o[2] for o in out])
            hourly[f"{case}/P_T_lo"], hourly[f"{case}/P_T_width"] = lo, _width32(wd)
            print(case, hourly[f"{case}/obj"][:2], "unique P_T", float((wd < 1e-9).mean()), flush=True)

    def merge(path, new):
        old = dict(np.load(path)) if (only and os.path.exists(path)) else {}
        old.update(new)
        if old:
            np.savez_compressed(path, **old)
    merge(os.path.join(GOLD, "oracle_objectives.npz"), objs)
    merge(os.path.join(GOLD, "oracle_setpoints.npz"), setp)
    merge(os.path.join(GOLD, "oracle_hourly.npz"), hourly)
