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
    bf, lm = scenarios.PRICE_TAKER_FAMILY_WIDE[k]                 # (members 0 .. 15 = PRICE_TAKER_FAMILY)
    t = time.time()
    P, info = orc.wind_battery_price_taker(T, cf, lmp * lm, batt_cap_factor=bf)
    x, obj = P.solve(tight=True)
    return k, obj, P.value(info["npv"], x), x[info["Pb"]] * 1e-3, time.time() - t


def one_pem(args):
    """LP #5 (reference wind_battery_pem_optimize, wind_battery_PEM_LMP.py:180-298) for member k of scenarios.PEM_PRICE_TAKER_FAMILY on
    the bus-303 series (the inputs of scenarios.pem_price_taker_batch(..., inputs="rts303"))."""
    T, k = args
    from dispatches_amd import scenarios
    from oracle import dispatch_lp_oracle as orc
    cf, lmp = scenarios.price_taker_inputs(T)
    h2, pf = scenarios.PEM_PRICE_TAKER_FAMILY[k]
    t = time.time()
    P, info = orc.wind_battery_pem_price_taker(T, cf, lmp, h2, True)
    c = P.c.copy()
    c[info["Cp"]] += 1e-5 * (pf - 1.0) * orc.PEM_CAP_COST                  # the family's PEM capital-cost factor
    x, obj = P.solve(c=c, tight=True)
    return k, obj, x[info["Cp"]] * 1e-3, x[info["Pb"]] * 1e-3, time.time() - t


def pem_fixtures(T, members):
    """python tools/make_price_taker_fixtures.py --pem [T] [members]: adds pem_T<T>/{obj, pem_mw, batt_mw} for the first `members`
    members of the wind + battery + PEM family (round 4: the year-long horizon the reference's sweeps run LP #5 at)."""
    import multiprocessing as mp
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "oracle_price_taker.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    with mp.get_context("spawn").Pool(min(os.cpu_count() or 1, members)) as pool:
        res = sorted(pool.map(one_pem, [(T, k) for k in range(members)]))
    out[f"pem_T{T}/obj"] = np.array([r[1] for r in res])
    out[f"pem_T{T}/pem_mw"] = np.array([r[2] for r in res])
    out[f"pem_T{T}/batt_mw"] = np.array([r[3] for r in res])
    print(f"PEM T={T}: HiGHS {np.mean([r[4] for r in res]):.1f} s per member; obj {out[f'pem_T{T}/obj']} PEM MW {out[f'pem_T{T}/pem_mw']} batt MW {out[f'pem_T{T}/batt_mw']}", flush=True)
    np.savez(path, **out)


def one_nuclear(args):
    """LP #6 (nuclear + PEM + tank price-taker, price_taker_analysis.py:116-222) for point k of the 6 x 10 (hydrogen price x PEM
    capacity) grid of scenarios.nuclear_price_taker_batch, at the reference's own horizon."""
    T, k = args
    from dispatches_amd import scenarios
    from dispatches_amd.flowsheets.price_taker import NUCLEAR_H2_PRICES, NUCLEAR_PEM_FRACTIONS, NP_CAPACITY_MW
    from oracle import dispatch_lp_oracle as orc
    lmp = scenarios.load_series("nuclear_price_taker_lmps.npz")["rt_lmp"][:T]
    grid = [(hp, pc) for hp in NUCLEAR_H2_PRICES for pc in NUCLEAR_PEM_FRACTIONS]
    hp, pc = grid[k]
    t = time.time()
    obj = orc.nuclear_price_taker(T, lmp, hp, pc * NP_CAPACITY_MW)[0].solve(tight=True)[1]
    return k, obj, time.time() - t


def nuclear_fixtures(T, points):
    """python tools/make_price_taker_fixtures.py --nuclear [T] [k ...]: adds nuclear_T<T>/{k, obj}: HiGHS objectives of the given grid
    points (round 4: the full-horizon enumeration was checked against the oracle's closed form only)."""
    import multiprocessing as mp
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "oracle_price_taker.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    with mp.get_context("spawn").Pool(min(os.cpu_count() or 1, len(points))) as pool:
        res = sorted(pool.map(one_nuclear, [(T, k) for k in points]))
    out[f"nuclear_T{T}/k"] = np.array([r[0] for r in res])
    out[f"nuclear_T{T}/obj"] = np.array([r[1] for r in res])
    print(f"nuclear T={T}: HiGHS {np.mean([r[2] for r in res]):.1f} s per point; k {out[f'nuclear_T{T}/k']} obj {out[f'nuclear_T{T}/obj']}", flush=True)
    np.savez(path, **out)


def wide_fixtures(T, members, procs):
    """python tools/make_price_taker_fixtures.py --wide [T] [stride] [processes]: adds T<T>w/{k, obj, npv, batt_mw} for every `stride`-th
    member from 16 on of the 256-member scenarios.PRICE_TAKER_FAMILY_WIDE (round 6: the year-long batch as 256 DISTINCT LPs; members
    0 .. 15 are the T<T>/ entries)."""
    import multiprocessing as mp
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "oracle_price_taker.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    with mp.get_context("spawn").Pool(procs) as pool:
        res = []
        for r in pool.imap_unordered(one, [(T, k) for k in members]):
            res.append(r)
            print(f"member {r[0]}: obj {r[1]:.6f} batt {r[3]:.2f} MW, HiGHS {r[4]:.0f} s", flush=True)
    res.sort()
    out[f"T{T}w/k"] = np.array([r[0] for r in res])
    out[f"T{T}w/obj"] = np.array([r[1] for r in res])
    out[f"T{T}w/npv"] = np.array([r[2] for r in res])
    out[f"T{T}w/batt_mw"] = np.array([r[3] for r in res])
    print(f"wide T={T}: {len(res)} members, HiGHS {np.mean([r[4] for r in res]):.1f} s per member", flush=True)
    np.savez(path, **out)


if __name__ == "__main__":
    import multiprocessing as mp
    from dispatches_amd import scenarios
    if len(sys.argv) > 1 and sys.argv[1] == "--wide":
        T = int(sys.argv[2]) if len(sys.argv) > 2 else 8736
        stride = int(sys.argv[3]) if len(sys.argv) > 3 else 4
        wide_fixtures(T, list(range(16, 256, stride)), int(sys.argv[4]) if len(sys.argv) > 4 else max(1, (os.cpu_count() or 2) - 2))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--nuclear":
        nuclear_fixtures(int(sys.argv[2]) if len(sys.argv) > 2 else 8784, [int(a) for a in sys.argv[3:]] or [0, 17, 34, 59])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--pem":
        pem_fixtures(int(sys.argv[2]) if len(sys.argv) > 2 else 8736, int(sys.argv[3]) if len(sys.argv) > 3 else 8)
        sys.exit(0)
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
