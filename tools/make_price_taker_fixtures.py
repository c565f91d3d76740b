"""Oracle fixtures of the long-horizon price-taker design family (LP #4: reference wind_battery_optimize,
wind_battery_LMP.py:172-269) -> tests/golden/oracle_price_taker.npz.

    python tools/make_price_taker_fixtures.py [T ...]        (default: 168 8736)

For every horizon T and every member of scenarios.PRICE_TAKER_FAMILY (battery capital-cost factor x LMP multiplier, 16
members): the oracle's UN-REDUCED restatement (oracle/dispatch_lp_oracle.py::wind_battery_price_taker) solved by HiGHS with
feasibility tolerances tightened to 1e-9 -> objective (-NPV * 1e-5), NPV, optimal battery size [MW].  Inputs: the first T hours
of the in-tree bus-303 series (scenarios.price_taker_inputs).  The year-long horizon takes HiGHS minutes per member: the
members run on a process pool.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def one(args):
    T, k = args
    from dispatches_amd import scenarios
    from oracle import dispatch_lp_oracle as orc
    cf, lmp = scenarios.price_taker_inputs(T)
    bf, lm = scenarios.PRICE_TAKER_FAMILY[k]
    t = time.time()
    P, info = orc.wind_battery_price_taker(T, cf, lmp * lm, batt_cap_factor=bf)
    x, obj = P.solve(tight=True)
    return k, obj, P.value(info["npv"], x), x[info["Pb"]] * 1e-3, time.time() - t


if __name__ == "__main__":
    import multiprocessing as mp
    from dispatches_amd import scenarios
    horizons = [int(a) for a in sys.argv[1:]] or [168, 8736]
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "oracle_price_taker.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    nfam = len(scenarios.PRICE_TAKER_FAMILY)
    with mp.get_context("spawn").Pool(min(os.cpu_count() or 1, nfam)) as pool:
        for T in horizons:
            res = sorted(pool.map(one, [(T, k) for k in range(nfam)]))
            out[f"T{T}/obj"] = np.array([r[1] for r in res])
            out[f"T{T}/npv"] = np.array([r[2] for r in res])
            out[f"T{T}/batt_mw"] = np.array([r[3] for r in res])
            print(f"T={T}: HiGHS {np.mean([r[4] for r in res]):.1f} s per member; obj {out[f'T{T}/obj'][:4]} batt MW {out[f'T{T}/batt_mw'][:8]}", flush=True)
    np.savez(path, **out)
