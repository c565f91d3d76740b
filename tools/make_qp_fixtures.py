"""Oracle fixtures of BASELINE config 5 (day-ahead bidding LP of the metric workload + quadratic ramp cost):

  tests/golden/oracle_qp.npz     per QP workload (scenarios.QP_WORKLOADS) and scenario: the CERTIFIED bracket
                                 [lower, upper] of the optimal value and the delivered power P_T[t] of the bracket's
                                 feasible point (the ramp term makes the optimal P_T profile essentially unique)

from oracle/qp_cutting_plane.py: Kelley's cutting planes on the un-reduced LP oracle (HiGHS dual simplex), no QP solver
involved; upper - lower <= 1e-9 (1 + |upper|).        python tools/make_qp_fixtures.py [B]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def qp_scenario(k, T=24):
    """(cf, da, rt) of scenario k of the wind+battery workloads (same windows as scenarios.load_prices / make_oracle_fixtures)."""
    from dispatches_amd import scenarios
    s = scenarios.load_series("rts_gmlc_309.npz")
    N = len(s["rt_lmp"])
    h0 = (17 * k) % (N - T)
    return s["rt_cf"][h0:h0 + T], np.clip(s["da_lmp"][h0:h0 + T], 0, 500), np.clip(s["rt_lmp"][h0:h0 + T], 0, 500)


def _work(args):
    rho, ids = args
    from oracle import qp_cutting_plane as qp
    up, lo, pt = [], [], []
    for k in ids:
        cf, da, rt = qp_scenario(k)
        out, *_ = qp.wind_battery_da_qp(24, cf, da, rt, rho)
        up.append(out["upper"]); lo.append(out["lower"]); pt.append(out["P_T"])
    return np.array(up), np.array(lo), np.array(pt)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    from dispatches_amd import scenarios
    procs = os.cpu_count() or 1
    chunks = [c.tolist() for c in np.array_split(np.arange(B), procs * 8) if len(c)]
    out = {}
    with mp.get_context("spawn").Pool(procs) as pool:
        for wl, (_fn, kw) in scenarios.QP_WORKLOADS.items():
            res = pool.map(_work, [(kw["ramp_cost"], c) for c in chunks])
            out[f"{wl}/upper"] = np.concatenate([r[0] for r in res])
            out[f"{wl}/lower"] = np.concatenate([r[1] for r in res])
            out[f"{wl}/P_T"] = np.concatenate([r[2] for r in res]).astype(np.float32)
            gap = out[f"{wl}/upper"] - out[f"{wl}/lower"]
            print(wl, out[f"{wl}/upper"][:3], "max bracket width (relative)", float(np.max(gap / (1 + np.abs(out[f"{wl}/upper"])))), flush=True)
    np.savez_compressed(os.path.join(GOLD, "oracle_qp.npz"), **out)
