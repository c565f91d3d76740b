"""Lab (development tool; NOT product, NOT oracle): does a year-long price-taker LP start faster from the stitched solutions of its
weeks?  The full-horizon LP #4 of member `member` is solved by the numpy restatement of the streaming PDLP (tools/stream_lab.py)
  (a) cold,
  (b) from the exact primal-dual optimum (HiGHS) - the floor of any warm start,
  (c) from the optimum with relative noise,
  (d) from the optima of its W-hour pieces solved on their own (HiGHS here; on the GPU they would be ONE batch of T / W small LPs)
      stitched by variable / row name: period t of piece w -> period t + w W, accumulated throughput offset by the earlier pieces',
      design variables by their maximum over the pieces.

    python tools/stream_warm_lab.py T=1344 W=168 member=5
"""
import os
import re
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.sparse as sp

import stream_lab as lab

fin = lab.fin
PAT = re.compile(r"^(.*)\[(\d+)\]$")


def build_piece(T, member, cf, lmp):
    """the LP of stream_lab.build on the given capacity-factor / price slices"""
    from dispatches_amd import scenarios
    from dispatches_amd.flowsheets.price_taker import wind_battery_price_taker
    from dispatches_amd.workflow.batch_model import ScenarioBatchModel
    block, objective, handles = wind_battery_price_taker(T, cf, lmp, wind_mw=847.0)
    model = ScenarioBatchModel(block, 1, T, indexed=True)
    model.finalize(objective)
    bf, lm = scenarios.PRICE_TAKER_FAMILY[member % len(scenarios.PRICE_TAKER_FAMILY)]
    c = handles["objective_vector"](model.lp.n, lmp_multiplier=lm, batt_cap_factor=bf)
    lb, ub, rlo, rhi = model.scenario_bounds()
    pick = lambda a: (a[0] if a.ndim == 2 else a).astype(float)
    return dict(A=model.lp.csr(), c=c.astype(float), lb=pick(lb), ub=pick(ub), rlo=pick(rlo), rhi=pick(rhi), c0=float(model.lp.c0), lp=model.lp)


def highs_pd(P):
    """optimal x AND row multipliers y in the PDHG sign convention (reduced cost c - A^T y; y > 0: lower side active)"""
    from scipy.optimize import linprog
    A, lo, hi = P["A"], P["rlo"], P["rhi"]
    eq = np.isfinite(lo) & (lo == hi); up = np.isfinite(hi) & ~eq; dn = np.isfinite(lo) & ~eq
    Aub = sp.vstack([A[up], -A[dn]]).tocsr(); bub = np.concatenate([hi[up], -lo[dn]])
    res = linprog(P["c"], A_ub=Aub if Aub.shape[0] else None, b_ub=bub if Aub.shape[0] else None,
                  A_eq=A[eq] if eq.any() else None, b_eq=hi[eq] if eq.any() else None,
                  bounds=np.stack([P["lb"], P["ub"]], 1), method="highs")
    assert res.status == 0, res.message
    y = np.zeros(A.shape[0])
    if eq.any():
        y[eq] = res.eqlin.marginals
    if Aub.shape[0]:
        mu = res.ineqlin.marginals
        nu = int(up.sum())
        y[up] = mu[:nu]
        y[dn] = -mu[nu:]
    return res.x, y, res.fun + P["c0"]


def stitch(P, pieces, W):
    lp = P["lp"]
    col = {nm: j for j, nm in enumerate(lp.col_names)}
    row = {nm: i for i, nm in enumerate(lp.row_names)}
    x, y = np.zeros(lp.n), np.zeros(lp.m)
    design = {}
    thr_off = 0.0
    for w, (Q, xw, yw) in enumerate(pieces):
        thr_end = 0.0
        for j, nm in enumerate(Q["lp"].col_names):
            mt = PAT.match(nm)
            if mt is None:
                design.setdefault(nm, []).append(xw[j])
                continue
            full = f"{mt.group(1)}[{int(mt.group(2)) + w * W}]"
            if full in col:
                v = xw[j]
                if "energy_throughput" in nm:
                    thr_end = max(thr_end, v)
                    v += thr_off
                x[col[full]] = v
        thr_off += thr_end
        for i, nm in enumerate(Q["lp"].row_names):
            mt = PAT.match(nm)
            if mt is None:
                continue
            full = f"{mt.group(1)}[{int(mt.group(2)) + w * W}]"
            if full in row:
                y[row[full]] = yw[i]
    for nm, vals in design.items():
        if nm in col:
            x[col[nm]] = max(vals)
    return x, y


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    T, W, member = int(kw.get("T", 672)), int(kw.get("W", 168)), int(kw.get("member", 5))
    from dispatches_amd import scenarios
    cf, lmp = scenarios.price_taker_inputs(T)
    P = build_piece(T, member, cf, lmp)
    cs = lab.physical_scales(P, T)
    t = time.time()
    xs, ys, ref = highs_pd(P)
    print(f"T={T} member={member} n={P['lp'].n} m={P['lp'].m}: HiGHS {ref:.9e} in {time.time() - t:.1f}s", flush=True)

    def run(tag, **o):
        t = time.time()
        X, Y, it, nrs, done, _ = lab.solve(P, colscale=cs, **o)
        obj = P["c"] @ X + P["c0"]
        print(f"  {tag:34s} done={done} iterations={it:7d} restarts={nrs:3d} relerr={abs(obj - ref) / max(1, abs(ref)):.1e} ({time.time() - t:.0f}s)", flush=True)
        return it
    cold = run("cold")
    run("exact optimum", x0=xs, y0=ys)
    rng = np.random.default_rng(0)
    for d in (1e-3, 1e-2):
        run(f"optimum x (1 + {d:g} N(0,1))", x0=xs * (1 + d * rng.standard_normal(xs.size)), y0=ys * (1 + d * rng.standard_normal(ys.size)))
    pieces = []
    t = time.time()
    for w in range(T // W):
        Q = build_piece(W, member, cf[w * W:(w + 1) * W], lmp[w * W:(w + 1) * W])
        xw, yw, fw = highs_pd(Q)
        pieces.append((Q, xw, yw))
    x0, y0 = stitch(P, pieces, W)
    AX = P["A"] @ x0
    viol = np.maximum(P["rlo"] - AX, 0) + np.maximum(AX - P["rhi"], 0)
    print(f"  {T // W} pieces of {W} h solved + stitched in {time.time() - t:.1f}s: objective of the stitched point {P['c'] @ x0 + P['c0']:.6e}, "
          f"row violation {np.linalg.norm(viol):.2e} (|q| {np.linalg.norm(fin(P['rhi'])):.2e})", flush=True)
    run("stitched pieces (x and y)", x0=x0, y0=y0)
    run("stitched pieces (x only)", x0=x0)
    run("stitched pieces (y only)", y0=y0)
