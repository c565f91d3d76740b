"""numpy executable specification of the QP / reduced-precision kernel (csrc/dsp_qp.hip) — development tool, NOT product,
NOT oracle.  Restarted reflected Halpern PDHG on

    min c.x + sum_{i soft} (a_i.x)^2 / (2 kappa_i)   s.t.  rlo <= A x <= rhi (hard rows),  lb <= x <= ub

The quadratic term is FACTORED, Q = sum_i a_i a_i^T / kappa_i, and lives in the DUAL: a soft row is an equality row
a_i.x = 0 whose multiplier pays kappa_i y_i^2 / 2 (compliance), so the dual step of such a row is the proximal step
y+ = (y - sig a_i.xbar) / (1 + sig kappa_i) and nothing else changes.  (The first attempt - lifting to a diagonal Q on
extra columns with the primal proximal step x+ = clip((x - tau (c - A^T y)) / (1 + tau q)) - needed 10-20x the
iterations of the LP and did not reach 1e-9 in 40 k; the compliance form needs about as many as the LP.)
Iterates are held in `dtype` (float64 or float32), every reduction / KKT quantity is accumulated in float64 — the split
the kernel makes.

    python tools/pdqp_proto.py [workload] [B] [float32|float64] [eps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from tools.pdlp_proto import Problem, build, ruiz_pc_scaling, spectral_norm


def solve(P, kappa=None, dtype=np.float64, eps=1e-9, eps_obj=1e-7, max_iter=200000, check_every=16, kp=0.7,
          max_dlog=np.log(30.0), beta=(0.2, 0.8, 0.36), eta_scale=0.998, guard=4.0):
    R = dtype
    lp = P.lp
    As, dr, dc = ruiz_pc_scaling(P.A)
    B, n, m = P.c.shape[0], lp.n, lp.m
    kappa = np.zeros(m) if kappa is None else np.asarray(kappa, float)
    hard = kappa == 0
    rho = np.where(hard, 0.0, 1.0 / np.where(hard, 1.0, kappa))
    eta = eta_scale / spectral_norm(As)
    c64, lb64, ub64 = P.c * dc, P.lb / dc, P.ub / dc
    rlo64, rhi64 = P.rlo * dr, P.rhi * dr
    ks64 = kappa * dr * dr
    fin = lambda a: np.where(np.isfinite(a), a, 0.0)
    qs = np.sqrt(np.sum(np.maximum(np.abs(fin(rlo64)), np.abs(fin(rhi64))) ** 2, 1))
    cs = np.linalg.norm(c64, axis=1)
    qn = np.sqrt(np.sum(np.maximum(np.abs(fin(P.rlo)), np.abs(fin(P.rhi))) ** 2, 1) + np.sum(fin(P.lb) ** 2 + fin(P.ub) ** 2, 1))
    cn = np.linalg.norm(P.c, axis=1)
    w = np.where((cs > 1e-10) & (qs > 1e-10), cs / np.maximum(qs, 1e-300), 1.0)
    u = float(np.finfo(R).eps) / 2
    qall = np.sqrt(qs ** 2 + np.sum(fin(lb64) ** 2 + fin(ub64) ** 2, 1))
    w_lo = guard * eta * u * np.abs(c64).max(1) / (eps * (1 + qall))
    qmax = np.maximum(np.abs(fin(rlo64)), np.abs(fin(rhi64))).max(1)
    w_hi = np.where(qmax > 0, eps * (1 + cs) / (guard * eta * u * np.maximum(qmax, 1e-300)), np.inf)
    Am = As.astype(R)
    AmT = sp.csr_matrix(Am.T)
    c, lb, ub, rlo, rhi = (a.astype(R) for a in (c64, lb64, ub64, rlo64, rhi64))
    x = np.clip(np.zeros((B, n), R), lb, ub)
    y = np.zeros((B, m), R)
    x0, y0 = x.copy(), y.copy()
    k = np.zeros(B)
    total = np.zeros(B)
    r0 = np.full(B, np.inf)
    rprev = np.full(B, np.inf)
    done = np.zeros(B, bool)
    iters = np.zeros(B, int)
    Xout, Yout, obj = np.zeros((B, n)), np.zeros((B, m)), np.zeros(B)
    for it in range(max_iter):
        tau = (eta / w).astype(R)[:, None]
        sig = (eta * w).astype(R)[:, None]
        srow = (1.0 / (1.0 + sig.astype(float) * ks64)).astype(R)
        xp = np.clip(x - tau * c + tau * (y @ Am), lb, ub)
        wv = y - sig * ((2 * xp - x) @ AmT)
        yp = (wv - np.clip(wv, -sig * rhi, -sig * rlo)) * srow
        total += 1
        k += 1
        keep = np.ones(B, bool)
        if (it + 1) % check_every == 0:
            dx, dy = (xp - x).astype(float), (yp - y).astype(float)
            adx = ((xp - x) @ AmT).astype(float)
            r = np.maximum(w * np.sum(dx * dx, 1) - 2 * eta * np.sum(dy * adx, 1) + np.sum(dy * dy, 1) / w, 0)   # squared
            # KKT in the original space, float64 accumulation from the R-precision iterates
            X, Y = xp.astype(float) * dc, yp.astype(float) * dr
            AX = X @ P.A.T
            pres = (np.maximum(P.rlo - AX, 0) + np.maximum(AX - P.rhi, 0)) * hard
            rc = P.c - Y @ P.A
            lp_ = np.where(np.isfinite(P.lb), np.maximum(rc, 0), 0.0)
            lm_ = np.where(np.isfinite(P.ub), np.maximum(-rc, 0), 0.0)
            dres = rc - lp_ + lm_
            po = np.sum(P.c * X, 1) + 0.5 * np.sum(rho * AX * AX, 1)
            do = (-0.5 * np.sum(kappa * Y * Y, 1) + np.sum(np.maximum(Y, 0) * fin(P.rlo) - np.maximum(-Y, 0) * fin(P.rhi), 1)
                  + np.sum(lp_ * fin(P.lb) - lm_ * fin(P.ub), 1))
            rp = np.linalg.norm(pres, axis=1) / (1 + qn)
            rd = np.linalg.norm(dres, axis=1) / (1 + cn)
            gap = np.abs(po - do)
            rg = gap / (1 + np.abs(po) + np.abs(do))
            conv = (rp <= eps) & (rd <= eps) & (rg <= eps)
            if eps_obj > 0:
                lim = np.maximum(eps_obj * (1 + np.abs(po + P.c0)), 1e-12 * np.sum(np.abs(P.c * X), 1))
                conv &= (gap <= lim) & (np.sum(np.abs(Y) * pres, 1) <= lim) & (np.sum(np.abs(dres) * np.abs(X), 1) <= lim)
            conv &= ~done
            if conv.any():
                Xout[conv], Yout[conv], obj[conv] = X[conv], Y[conv], (po + P.c0)[conv]
                iters[conv] = it + 1
                done |= conv
            if done.all():
                break
            first = ~np.isfinite(r0)
            b1, b2, b3 = beta
            rs = ~first & ((r <= b1 * b1 * r0) | ((r <= b2 * b2 * r0) & (r > rprev)) | (k >= b3 * total))
            r0 = np.where(first, r, r0)
            rprev = r
            if rs.any():
                d0 = np.sum((xp - x0).astype(float) ** 2, 1)
                d1 = np.sum((yp - y0).astype(float) ** 2, 1)
                ok = rs & (d0 > 1e-28) & (d1 > 1e-28)
                e = np.log(w) + 0.5 * (np.log(np.maximum(d0, 1e-300)) - np.log(np.maximum(d1, 1e-300)))
                w = np.where(ok, w * np.exp(np.clip(-kp * e, -max_dlog, max_dlog)), w)
                w = np.where(rs, np.minimum(np.maximum(w, w_lo), np.maximum(w_hi, w_lo)), w)
                x = np.where(rs[:, None], xp, x)
                y = np.where(rs[:, None], yp, y)
                x0 = np.where(rs[:, None], xp, x0)
                y0 = np.where(rs[:, None], yp, y0)
                k = np.where(rs, 0, k)
                r0 = np.where(rs, np.inf, r0)
                rprev = np.where(rs, np.inf, rprev)
                keep = ~rs
        lam = (1.0 / (k + 2)).astype(R)[:, None]
        tx, ty = 2 * xp - x, 2 * yp - y
        xn = tx + lam * (x0 - tx)
        yn = ty + lam * (y0 - ty)
        x = np.where(keep[:, None], xn, x)
        y = np.where(keep[:, None], yn, y)
    nd = ~done
    iters[nd] = max_iter
    X = xp.astype(float) * dc
    Xout[nd], Yout[nd] = X[nd], (yp.astype(float) * dr)[nd]
    obj[nd] = (np.sum(P.c * X, 1) + 0.5 * np.sum(rho * (X @ P.A.T) ** 2, 1) + P.c0)[nd]
    return Xout, Yout, obj, iters, done


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "wind_battery_24h_qp01"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dt = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "float32") else np.float64
    eps = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-9
    model, P = build(wl, B)
    q = model.lp.row_compliance
    print(wl, "n,m,nnz =", P.lp.n, P.lp.m, P.lp.nnz, "soft rows:", 0 if q is None else int(np.count_nonzero(q)))
    t = time.time()
    X, Y, obj, iters, done = solve(P, q, dt, eps=eps, eps_obj=min(1e-7, 100 * eps) if eps < 1e-6 else 0.0,
                                   max_iter=int(os.environ.get("MAXIT", 60000)))
    print("time %.1fs" % (time.time() - t), "iters mean/max", iters.mean(), iters.max(), "done", done.mean())
    print("objectives", obj[:4])
