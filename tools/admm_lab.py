"""Lab (development tool; NOT product, NOT oracle): ADMM (OSQP's splitting) with an EXACT solve of the time-banded normal equations on
the year-long price-taker LP - the question is how many iterations a method needs whose linear algebra sees the whole horizon at once
(the design column and the state-of-charge / throughput chains included), against the 75 k of diagonally preconditioned PDHG.

    min c x   s.t.  rlo <= A x <= rhi,  lb <= x <= ub        (rows of A and the identity rows of the bounds: Atilde = [A; I])
    x~  = (sigma I + Atilde' R Atilde)^-1 (sigma x - c + Atilde' (R z - y))
    z~  = Atilde x~ ;  x+ = a x~ + (1 - a) x ;  z+ = clip(a z~ + (1 - a) z + y / R) ;  y+ = y + R (a z~ + (1 - a) z - z+)

    python tools/admm_lab.py T=672 member=5
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

import pdlp_proto as pp
import stream_lab as lab

fin = lab.fin


def solve(P, rho0=0.1, sigma=1e-6, alpha=1.6, eps=1e-9, eps_obj=5e-7, max_iter=20000, check=25, adapt=True, colscale=None, eq_boost=1e3,
          verbose=0, n_ruiz=10):
    A0 = sp.csr_matrix(P["A"])
    m, n = A0.shape
    if colscale is not None:
        A0 = sp.csr_matrix(A0 @ sp.diags(colscale))
    As, dr, dc = pp.ruiz_pc_scaling(A0, n_ruiz=int(n_ruiz))
    if colscale is not None:
        dc = dc * colscale
    c = P["c"] * dc
    cs = max(np.abs(c).max(), 1e-12)
    c = c / cs                                                   # cost scaling (OSQP)
    lb, ub, rlo, rhi = P["lb"] / dc, P["ub"] / dc, P["rlo"] * dr, P["rhi"] * dr
    At = sp.vstack([As, sp.identity(n, format="csr")]).tocsc()
    lo, hi = np.concatenate([rlo, lb]), np.concatenate([rhi, ub])
    eq = lo == hi
    free = ~np.isfinite(lo) & ~np.isfinite(hi)
    base = np.where(eq, eq_boost, 1.0)
    base[free] = 1e-6
    rho = rho0
    x = np.clip(np.zeros(n), lb, ub); z = np.clip(At @ x, lo, hi); y = np.zeros(m + n)
    A, AT = P["A"], sp.csr_matrix(P["A"].T)
    qn = np.sqrt(np.sum(np.maximum(np.abs(fin(P["rlo"])), np.abs(fin(P["rhi"]))) ** 2) + np.sum(fin(P["lb"]) ** 2 + fin(P["ub"]) ** 2))
    cn = np.linalg.norm(P["c"])
    nfac = 0

    def factor(rho):
        R = rho * base
        K = (sigma * sp.identity(n) + At.T @ sp.diags(R) @ At).tocsc()
        return R, spl.factorized(K)
    R, ksolve = factor(rho); nfac += 1
    t0 = time.time()
    for it in range(1, int(max_iter) + 1):
        xt = ksolve(sigma * x - c + At.T @ (R * z - y))
        zt = At @ xt
        xn = alpha * xt + (1 - alpha) * x
        zr = alpha * zt + (1 - alpha) * z
        zn = np.clip(zr + y / R, lo, hi)
        y = y + R * (zr - zn)
        x, z = xn, zn
        if it % check == 0:
            # unscaled point: x (columns), duals of the rows of A: y[:m]; the bound duals are implied (reduced costs)
            Xu = np.clip(x, lb, ub) * dc
            Yu = -y[:m] * dr * cs                                  # sign: our convention y >= 0 on lower-bounded rows
            AX = A @ Xu
            viol = np.maximum(P["rlo"] - AX, 0) + np.maximum(AX - P["rhi"], 0)
            rc = P["c"] - AT @ Yu
            lp_ = np.where(np.isfinite(P["lb"]), np.maximum(rc, 0), 0.0); lm_ = np.where(np.isfinite(P["ub"]), np.maximum(-rc, 0), 0.0)
            dres = rc - lp_ + lm_
            po = P["c"] @ Xu
            do = np.sum(np.maximum(Yu, 0) * fin(P["rlo"]) - np.maximum(-Yu, 0) * fin(P["rhi"])) + np.sum(lp_ * fin(P["lb"]) - lm_ * fin(P["ub"]))
            rp, rd = np.linalg.norm(viol) / (1 + qn), np.linalg.norm(dres) / (1 + cn)
            bound = abs(po - do) + np.sum(np.abs(Yu) * viol) + np.sum(np.abs(dres) * np.abs(Xu))
            lim = max(eps_obj * (1 + abs(po + P["c0"])), 1e-12 * np.sum(np.abs(P["c"] * Xu)))
            # ADMM residuals (scaled space) for the rho controller
            r_p = np.linalg.norm(At @ x - z, np.inf); r_d = np.linalg.norm(c + At.T @ y, np.inf)
            n_p = max(np.linalg.norm(At @ x, np.inf), np.linalg.norm(z, np.inf), 1e-12)
            n_d = max(np.linalg.norm(At.T @ y, np.inf), np.linalg.norm(c, np.inf), 1e-12)
            if verbose and (it // check) % verbose == 0:
                print(f"  it {it} rp {rp:.2e} rd {rd:.2e} bound/lim {bound/lim:.2e} obj {po + P['c0']:.9e} rho {rho:.2e} admm {r_p/n_p:.1e}/{r_d/n_d:.1e} t {time.time()-t0:.0f}s", flush=True)
            if rp <= eps and rd <= eps and bound <= lim:
                return Xu, Yu, it, nfac, True
            if adapt:
                ratio = np.sqrt((r_p / n_p) / max(r_d / n_d, 1e-300))
                if ratio > 5 or ratio < 0.2:
                    rho = float(np.clip(rho * ratio, 1e-6, 1e6))
                    R, ksolve = factor(rho); nfac += 1
    return np.clip(x, lb, ub) * dc, -y[:m] * dr * cs, int(max_iter), nfac, False


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    T = int(kw.pop("T", 672)); member = int(kw.pop("member", 5))
    P = lab.build(T, member, None, kw.pop("throughput", "chain"))
    ref, xr, th = lab.highs(P)
    print(f"T={T} member={member} n={P['lp'].n} m={P['lp'].m} nnz={P['lp'].nnz} HiGHS {ref:.10e} ({th:.1f}s)", flush=True)
    cs = kw.pop("colscale", "phys")
    opts = {k: float(v) for k, v in kw.items()}
    if cs == "phys":
        opts["colscale"] = lab.physical_scales(P, T)
    t = time.time()
    X, Y, it, nfac, done = solve(P, **opts)
    obj = P["c"] @ X + P["c0"]
    print(f"done={done} iters={it} factorizations={nfac} obj={obj:.10e} relerr={abs(obj-ref)/max(1,abs(ref)):.2e} t={time.time()-t:.0f}s")
