"""Algorithm lab for the HBM-resident streaming PDLP (development tool; NOT product, NOT oracle): ONE scenario, scipy
sparse products, the algorithm of csrc/dsp_stream.hip (restarted reflected Halpern PDHG, PDLP restarts, proportional
primal-weight controller, KKT + objective-bound termination) with switches for the ideas under study.

    python tools/stream_lab.py T=1344 member=5 colscale=phys jump=1 ...
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.sparse as sp

import pdlp_proto as pp

fin = lambda a: np.where(np.isfinite(a), a, 0.0)


def build(T, member=0, degr=None, throughput="chain", family="base"):
    """price-taker LP #4 at horizon T for member `member` of the scenario family; returns dict(A, c, lb, ub, rlo, rhi, c0, names)"""
    from dispatches_amd import scenarios
    from dispatches_amd.flowsheets import parameters as prm
    if degr is not None:
        prm.battery_degradation_rate = degr

    class Dummy:
        def solve(self, *a, **k):
            raise RuntimeError
    handles, model = scenarios.price_taker_batch(T, max(member + 1, 1), Dummy(), throughput=throughput, family=family)
    lp = model.lp
    lb, ub, rlo, rhi = model.scenario_bounds()
    pick = lambda a: (a[member] if a.ndim == 2 else a).astype(float)
    return dict(A=lp.csr(), c=model.c[member].astype(float), lb=pick(lb), ub=pick(ub), rlo=pick(rlo), rhi=pick(rhi),
                c0=float(model.c0[member]), lp=lp, handles=handles, model=model)


def highs(P):
    from scipy.optimize import linprog
    A, lo, hi = P["A"], P["rlo"], P["rhi"]
    eq = np.isfinite(lo) & (lo == hi); up = np.isfinite(hi) & ~eq; dn = np.isfinite(lo) & ~eq
    Aub = sp.vstack([A[up], -A[dn]]).tocsr(); bub = np.concatenate([hi[up], -lo[dn]])
    t = time.time()
    res = linprog(P["c"], A_ub=Aub if Aub.shape[0] else None, b_ub=bub if Aub.shape[0] else None,
                  A_eq=A[eq] if eq.any() else None, b_eq=hi[eq] if eq.any() else None,
                  bounds=np.stack([P["lb"], P["ub"]], 1), method="highs")
    return res.fun + P["c0"], res.x, time.time() - t


def physical_scales(P, T, thr_duty=0.25, e_mult=1.0):
    """column scales from the model object: kW columns at the wind plant's range, SOC at 4 h of it, throughput at T/2 of it"""
    lp = P["lp"]
    wind_kw = 847e3
    s = np.full(lp.n, wind_kw)
    for j, name in enumerate(lp.col_names):
        if "state_of_charge" in name:
            s[j] = 4 * wind_kw
        elif "energy_throughput" in name:
            s[j] = wind_kw * max(T / 2, 1) * e_mult
        elif name.startswith("throughput_sum[") or name.startswith("throughput_before_copy["):
            lo, hi = map(int, name[name.index("[") + 1:-1].split(":"))
            s[j] = wind_kw * ((hi - lo) if "sum" in name else max(lo, 1)) * thr_duty
        elif name.startswith("throughput_before["):
            s[j] = wind_kw * int(name[name.index("[") + 1:-1]) * thr_duty
    return s


def solve(P, eps=1e-9, eps_obj=5e-7, max_iter=2_000_000, check=64, kp=0.7, beta=(0.2, 0.8, 0.36), colscale=None, n_ruiz=10,
          verbose=0, maxdl=np.log(30.0), x0=None, y0=None, w0=None, log_every=0, norm_iters=4000, beta3=None, beta1=None, post_boost=None,
          aitken=None, aitken_gain=1.0):
    if beta3 is not None or beta1 is not None:
        beta = (beta[0] if beta1 is None else beta1, beta[1], beta[2] if beta3 is None else beta3)
    A0 = sp.csr_matrix(P["A"])
    n, m = A0.shape[1], A0.shape[0]
    if colscale is not None:
        A0 = sp.csr_matrix(A0 @ sp.diags(colscale))
    As, dr, dc = pp.ruiz_pc_scaling(A0, n_ruiz=int(n_ruiz))
    if colscale is not None:
        dc = dc * colscale
    if post_boost:                      # {column: factor}: extra column scale AFTER Ruiz / Pock-Chambolle (round 5: design columns)
        f = np.ones(n)
        for j, v in post_boost.items():
            f[j] = v
        As = sp.csr_matrix(As @ sp.diags(f)); dc = dc * f
    AsT = sp.csr_matrix(As.T)
    c = P["c"] * dc; lb, ub = P["lb"] / dc, P["ub"] / dc; rlo, rhi = P["rlo"] * dr, P["rhi"] * dr
    nrm = pp.spectral_norm(As, iters=int(norm_iters))
    if verbose:
        print("spectral norm estimates", pp.spectral_norm(As, 200), pp.spectral_norm(As, 1000), nrm)
    eta = 0.998 / nrm
    qs = np.sqrt(np.sum(np.maximum(np.abs(fin(rlo)), np.abs(fin(rhi))) ** 2)); cs = np.linalg.norm(c)
    w = cs / qs if (cs > 1e-10 and qs > 1e-10) else 1.0
    if w0:
        w = w0
    x = np.clip(np.zeros(n) if x0 is None else x0 / dc, lb, ub)
    y = np.zeros(m) if y0 is None else y0 / dr
    xa, ya = x.copy(), y.copy()
    k = 0; r0 = np.inf; rprev = np.inf; nrs = 0
    qn = np.sqrt(np.sum(np.maximum(np.abs(fin(P["rlo"])), np.abs(fin(P["rhi"]))) ** 2) + np.sum(fin(P["lb"]) ** 2 + fin(P["ub"]) ** 2))
    cn = np.linalg.norm(P["c"])
    A = P["A"]; AT = sp.csr_matrix(A.T)
    hist = []
    hist_p, naitken = [], [0]
    solve.naitken = naitken
    t0 = time.time()
    for it in range(int(max_iter)):
        tau, sig = eta / w, eta * w
        xp = np.clip(x - tau * (c - AsT @ y), lb, ub)
        axb = As @ (2 * xp - x)
        wv = y - sig * axb
        yp = wv + np.clip(-wv, sig * rlo, sig * rhi)
        k += 1
        if (it + 1) % check == 0:
            dx, dy = xp - x, yp - y
            r = np.sqrt(max(w * dx @ dx - 2 * eta * dy @ (As @ dx) + dy @ dy / w, 0.0))
            Xu, Yu = xp * dc, yp * dr
            AX = A @ Xu
            viol = np.maximum(P["rlo"] - AX, 0) + np.maximum(AX - P["rhi"], 0)
            rc = P["c"] - AT @ Yu
            lp_ = np.where(np.isfinite(P["lb"]), np.maximum(rc, 0), 0.0); lm_ = np.where(np.isfinite(P["ub"]), np.maximum(-rc, 0), 0.0)
            dres = rc - lp_ + lm_
            po = P["c"] @ Xu
            do = np.sum(np.maximum(Yu, 0) * fin(P["rlo"]) - np.maximum(-Yu, 0) * fin(P["rhi"])) + np.sum(lp_ * fin(P["lb"]) - lm_ * fin(P["ub"]))
            rp, rd = np.linalg.norm(viol) / (1 + qn), np.linalg.norm(dres) / (1 + cn)
            gap = abs(po - do)
            bound = gap + np.sum(np.abs(Yu) * viol) + np.sum(np.abs(dres) * np.abs(Xu))
            lim = max(eps_obj * (1 + abs(po + P["c0"])), 1e-12 * np.sum(np.abs(P["c"] * Xu)))
            if log_every and ((it + 1) // check) % log_every == 0:
                print(f"  it {it+1} r {r:.3e} rp {rp:.2e} rd {rd:.2e} bound/lim {bound/lim:.2e} obj {po + P['c0']:.9e} w {w:.3e} k {k} t {time.time()-t0:.0f}s", flush=True)
            hist.append((it + 1, rp, rd, bound / lim, po + P["c0"]))
            if rp <= eps and rd <= eps and bound <= lim:
                return Xu, Yu, it + 1, nrs, True, hist
            first = not np.isfinite(r0)
            if first:
                r0 = r
            decayed = (r <= beta[0] * r0) or (r <= beta[1] * r0 and r > rprev)
            artificial = k >= beta[2] * (it + 1)
            rprev = r
            if not first and (decayed or artificial):
                ddx, ddy = np.linalg.norm(xp - xa), np.linalg.norm(yp - ya)
                if ddx > 1e-14 and ddy > 1e-14:
                    e = np.log(w) + np.log(ddx) - np.log(ddy)
                    w = w * np.exp(np.clip(-kp * e, -maxdl, maxdl))
                x, y, xa, ya = xp.copy(), yp.copy(), xp.copy(), yp.copy()
                if aitken is not None:
                    # round 5 experiment: Aitken extrapolation of the DESIGN columns (indices `aitken`) over their values at the last
                    # three restarts - their slow 1-D dynamics (step ~ 1 / T) is what the free-design solve waits for
                    hist_p.append(x[aitken].copy())
                    if len(hist_p) >= 3:
                        p0, p1, p2 = hist_p[-3], hist_p[-2], hist_p[-1]
                        d1, d2 = p1 - p0, p2 - p1
                        den = d2 - d1
                        ok = (np.abs(d2) < np.abs(d1)) & (d1 * d2 > 0) & (np.abs(den) > 1e-300)
                        ext = np.where(ok, p2 - aitken_gain * d2 * d2 / np.where(ok, den, 1.0), p2)
                        ext = np.clip(ext, lb[aitken], ub[aitken])
                        if ok.any():
                            x[aitken] = ext; xa[aitken] = ext
                            hist_p.clear()
                            naitken[0] += 1
                k = 0; r0 = np.inf; rprev = np.inf; nrs += 1
                continue
        lam = (k + 1) / (k + 2)
        x = lam * (2 * xp - x) + (1 - lam) * xa
        y = lam * (2 * yp - y) + (1 - lam) * ya
    return xp * dc, yp * dr, int(max_iter), nrs, False, hist


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    T = int(kw.pop("T", 168)); member = int(kw.pop("member", 5))
    degr = kw.pop("degr", None)
    thr = kw.pop("throughput", "chain")
    P = build(T, member, None if degr is None else float(degr), thr)
    print(f"T={T} member={member} n={P['lp'].n} m={P['lp'].m} nnz={P['lp'].nnz}", flush=True)
    ref, xr, th = highs(P)
    print(f"HiGHS obj {ref:.10e} in {th:.1f}s", flush=True)
    fixp = float(kw.pop("fixp", -1))
    if fixp >= 0:                      # fix the design columns at fixp x their optimal values (1 = the optimum)
        for j, name in enumerate(P["lp"].col_names):
            if name in ("battery_system_capacity", "battery.nameplate_power"):
                P["lb"][j] = P["ub"][j] = xr[j] * fixp
                print("fixed", name, xr[j] * fixp)
        ref, xr, th = highs(P)
        print(f"HiGHS (fixed design) obj {ref:.10e}")
    cs = kw.pop("colscale", "none")
    opts = {k: float(v) for k, v in kw.items()}
    e_mult = float(kw.pop("e_mult", 1.0)); thr_duty = float(kw.pop("thr_duty", 0.25))
    opts = {k: float(v) for k, v in kw.items()}
    if cs == "phys":
        opts["colscale"] = physical_scales(P, T, thr_duty, e_mult)
    t = time.time()
    X, Y, iters, nrs, done, hist = solve(P, **opts)
    obj = P["c"] @ X + P["c0"]
    print(f"done={done} iters={iters} restarts={nrs} obj={obj:.10e} relerr={abs(obj-ref)/max(1,abs(ref)):.2e} t={time.time()-t:.0f}s")
