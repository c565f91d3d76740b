"""Lab (development tool; NOT product, NOT oracle): a primal-dual interior-point method (Mehrotra predictor-corrector) whose Newton
systems are solved EXACTLY through the time-banded structure of the year-long price-taker LPs: the normal matrix A Theta A' without the
design column(s) has half-bandwidth 6 in the natural row order (m = 6 T rows), the dense columns come back through Sherman-Morrison-
Woodbury.  Question: Newton iterations to the parity tolerance, against the 75 k PDHG iterations of the streaming path.

    python tools/ipm_lab.py T=672 member=5
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

import pdlp_proto as pp
import stream_lab as lab

fin = lab.fin


def banded_from(N, w):
    """upper banded storage [w + 1, M] of a symmetric sparse matrix (scipy.linalg.cholesky_banded layout)"""
    N = sp.coo_matrix(N)
    M = N.shape[0]
    ab = np.zeros((w + 1, M))
    k = N.col - N.row
    keep = (k >= 0) & (k <= w)
    np.add.at(ab, (w - k[keep], N.col[keep]), N.data[keep])
    return ab


def solve(P, eps=1e-9, eps_obj=5e-7, max_iter=200, colscale=None, n_ruiz=10, dense_thr=16, verbose=0, reg=1e-12, prox=0.0, frac=0.99, refine=3, theta_cap=1e30, sigmin=0.05, mu0=0.0, gondzio=0, pcg=0, pcg_tol=1e-11, prefine=0, absref=0, rpk=0.0, nbhd=0.0, nb_back=0.8, samestep=0):
    """frac / sigmin: the product's settings since round 5's last commits (csrc/dsp_ipm.hip: 0.99 to the boundary, sigma >= 0.05); the first
    version ran 0.9995 / 0 (`frac=0.9995 sigmin=0` on the command line)"""
    A0 = sp.csr_matrix(P["A"])
    m, n = A0.shape
    if colscale is not None:
        A0 = sp.csr_matrix(A0 @ sp.diags(colscale))
    As, dr, dc = pp.ruiz_pc_scaling(A0, n_ruiz=int(n_ruiz))
    if colscale is not None:
        dc = dc * colscale
    c = P["c"] * dc
    cs = max(np.abs(c).max(), 1e-300); c = c / cs
    lb, ub, rlo, rhi = P["lb"] / dc, P["ub"] / dc, P["rlo"] * dr, P["rhi"] * dr
    # fixed columns out, slack columns in
    fixed = lb == ub
    xfix = np.where(fixed, lb, 0.0)
    eq = rlo == rhi
    ine = np.nonzero(~eq)[0]
    b = np.where(eq, rhi, 0.0) - As @ xfix
    cols = np.nonzero(~fixed)[0]
    E = sp.csr_matrix((np.ones(len(ine)), (ine, np.arange(len(ine)))), shape=(m, len(ine)))
    Abar = sp.hstack([As[:, cols], -E]).tocsc()
    N = Abar.shape[1]
    cbar = np.concatenate([c[cols], np.zeros(len(ine))])
    l = np.concatenate([lb[cols], rlo[ine]]); u = np.concatenate([ub[cols], rhi[ine]])
    hl, hu = np.isfinite(l), np.isfinite(u)
    assert (hl | hu).all(), "free columns: not handled here"
    # wide columns: rows further apart than the band allows (design columns, periodic conditions) - Woodbury
    span = np.array([(Abar.indices[Abar.indptr[j]:Abar.indptr[j + 1]].max() - Abar.indices[Abar.indptr[j]:Abar.indptr[j + 1]].min())
                     if Abar.indptr[j + 1] > Abar.indptr[j] else 0 for j in range(N)])
    dense = np.nonzero(span > dense_thr)[0]
    sparse_cols = np.setdiff1d(np.arange(N), dense)
    Asp, Ad = Abar[:, sparse_cols].tocsr(), Abar[:, dense].toarray()
    pat = (abs(Asp) @ abs(Asp).T).tocoo()
    w = int(np.abs(pat.row - pat.col).max())
    if verbose:
        print(f"  N={N} M={m} dense columns {len(dense)} half-bandwidth {w}")
    AspT = Asp.T.tocsr()
    # starting point: inside the box, duals 1
    v = np.where(hl & hu, 0.5 * (fin(l) + fin(u)), np.where(hl, fin(l) + 1.0, fin(u) - 1.0))
    y = np.zeros(m)
    z = np.where(hl, 1.0, 0.0); f = np.where(hu, 1.0, 0.0)
    if mu0 > 0:                                               # centred start: every complementarity product = mu0 (experiment, round 6)
        z = np.where(hl, mu0 / np.where(hl, v - fin(l), 1.0), 0.0); f = np.where(hu, mu0 / np.where(hu, fin(u) - v, 1.0), 0.0)
    A, AT = P["A"], sp.csr_matrix(P["A"].T)
    qn = np.sqrt(np.sum(np.maximum(np.abs(fin(P["rlo"])), np.abs(fin(P["rhi"]))) ** 2) + np.sum(fin(P["lb"]) ** 2 + fin(P["ub"]) ** 2))
    cn = np.linalg.norm(P["c"])
    nb = hl.sum() + hu.sum()
    t0 = time.time()
    for it in range(1, int(max_iter) + 1):
        wl = np.where(hl, v - fin(l), 1.0); tu = np.where(hu, fin(u) - v, 1.0)
        rp = b - Abar @ v
        rd = cbar - Abar.T @ y - z + f
        mu = (z @ (wl * hl) + f @ (tu * hu)) / nb
        theta = 1.0 / np.maximum(np.where(hl, z / wl, 0.0) + np.where(hu, f / tu, 0.0), 1e-300)
        theta = np.minimum(theta, theta_cap)
        if not (np.isfinite(theta).all() and np.isfinite(rp).all() and mu > 0):
            return None, None, it, False
        # factor B = Asp Theta Asp' (banded) ; dense columns through Woodbury
        th_s, th_d = theta[sparse_cols], theta[dense]
        Nm = (Asp @ sp.diags(th_s) @ AspT)
        ab = banded_from(Nm, w)
        delta = prox * np.median(ab[w])                      # dual proximal term (rank-deficient rows): K = A Theta A' + delta I
        ab[w] = ab[w] * (1.0 + reg) + delta                  # + relative diagonal regularisation (roundoff of the factorisation)
        cb = sla.cholesky_banded(ab, lower=False)
        bsolve = lambda r: sla.cho_solve_banded((cb, False), r)
        if len(dense):
            BiAd = bsolve(Ad)
            S = np.diag(1.0 / th_d) + Ad.T @ BiAd
            nsolve = lambda r: (lambda q: q - BiAd @ np.linalg.solve(S, Ad.T @ q))(bsolve(r))
        else:
            nsolve = bsolve

        def direction(sig_mu, corr_l, corr_u):
            # complementarity targets: z w + w dz + z dv = sig_mu - corr ; f t + t df - f dv = sig_mu - corr
            cz = np.where(hl, (sig_mu - corr_l) / wl - z, 0.0)
            cf = np.where(hu, (sig_mu - corr_u) / tu - f, 0.0)
            rt = rd - cz + cf
            rhs = rp + Abar @ (theta * rt)
            dy = nsolve(rhs)
            if pcg:
                # preconditioned conjugate gradients on the full normal equations, the factorisation (band + Woodbury) as preconditioner
                # (experiment, round 6): plain refinement is a Richardson iteration and stagnates where I - M^-1 N is not a contraction
                Nmul = lambda d: Abar @ (theta * (Abar.T @ d)) + delta * d
                r = rhs - Nmul(dy)
                nr0 = np.abs(rhs).max()
                zv = nsolve(r); pv = zv.copy(); rz = r @ zv
                k = 0
                while k < int(pcg) and np.abs(r).max() > pcg_tol * nr0:
                    Np = Nmul(pv)
                    al = rz / (pv @ Np)
                    dy = dy + al * pv
                    r = r - al * Np
                    zv = nsolve(r)
                    rz_new = r @ zv
                    pv = zv + (rz_new / rz) * pv
                    rz = rz_new
                    k += 1
                solve.pcg_steps = getattr(solve, "pcg_steps", 0) + k
                solve.pcg_res = max(getattr(solve, "pcg_res", 0.0), np.abs(r).max() / nr0)
            elif absref:
                # refinement until every row's residual, UNSCALED, is a fraction of what the termination test allows that row (experiment,
                # round 6): the relative test max |res| <= tol max |rhs| lets a row whose own right-hand side is 1e-10 keep an error of
                # 1e-11 x the largest entry (Theta up to 1e17 makes those 1e6) - a primal residual of 4e-5 that comes back after every step
                atol = absref * eps * (1.0 + qn) / np.sqrt(m)
                k = 0
                while k < 8:
                    res = rhs - Abar @ (theta * (Abar.T @ dy)) - delta * dy
                    if np.abs(res / dr).max() <= atol:
                        break
                    dy = dy + nsolve(res); k += 1
                solve.ref_steps = getattr(solve, "ref_steps", 0) + k
                solve.ref_max = max(getattr(solve, "ref_max", 0), k)
            else:
              for _ in range(int(refine)):                     # iterative refinement on the full normal equations
                res = rhs - Abar @ (theta * (Abar.T @ dy)) - delta * dy
                dy = dy + nsolve(res)
            dv = theta * (Abar.T @ dy - rt)
            # refinement of the PRIMAL Newton equation Abar dv = rp on the direction as it was actually formed (experiment, round 6): for
            # a basic column Theta ~ 1e13 multiplies a difference of O(1) numbers that is ~ dv / Theta - the rounding of that difference
            # is an error of 1e-6 |dv| that the normal equations' residual never sees
            if verbose > 1 and it >= 64:
                resn = rhs - Abar @ (theta * (Abar.T @ dy)) - delta * dy
                r1 = rp - Abar @ dv
                i1 = int(np.argmax(np.abs(r1)))
                print(f"      direction: normal-eq residual max {np.abs(resn).max():.2e} (|rhs| max {np.abs(rhs).max():.2e}); primal Newton residual |rp - Abar dv| max {np.abs(r1).max():.2e} at row {i1} (rp there {rp[i1]:.2e}, |rp| max {np.abs(rp).max():.2e}); |dv| max {np.abs(dv).max():.2e} |dy| max {np.abs(dy).max():.2e}")
            for _ in range(int(prefine)):
                r1 = rp - Abar @ dv
                dy1 = nsolve(r1)
                dv = dv + theta * (Abar.T @ dy1)
                dy = dy + dy1
            dz = np.where(hl, cz - z / wl * dv, 0.0)
            df = np.where(hu, cf + f / tu * dv, 0.0)
            return dv, dy, dz, df

        def steps(dv, dz, df):
            ap = 1.0
            neg = hl & (dv < 0)
            if neg.any():
                ap = min(ap, np.min(-wl[neg] / dv[neg]))
            pos = hu & (dv > 0)
            if pos.any():
                ap = min(ap, np.min(tu[pos] / dv[pos]))
            ad = 1.0
            nz = hl & (dz < 0)
            if nz.any():
                ad = min(ad, np.min(-z[nz] / dz[nz]))
            nf = hu & (df < 0)
            if nf.any():
                ad = min(ad, np.min(-f[nf] / df[nf]))
            return ap, ad
        dva, dya, dza, dfa = direction(0.0, 0.0, 0.0)
        apa, ada = steps(dva, dza, dfa)
        mu_aff = ((z + ada * dza) @ ((wl + apa * dva) * hl) + (f + ada * dfa) @ ((tu - apa * dva) * hu)) / nb
        sigma = max((mu_aff / mu) ** 3, sigmin)
        if rpk > 0:                                           # complementarity may not run ahead of primal feasibility (experiment, round 6)
            sigma = min(1.0, max(sigma, rpk * np.abs(rp).max() / mu))
        dv, dy, dz, df = direction(sigma * mu, dva * dza, -dva * dfa)
        ap, ad = steps(dv, dz, df)
        # Gondzio's multiple centrality correctors (experiment, round 6): aim at a longer step, push the complementarity products of the
        # trial point back into [beta_min, beta_max] x sigma mu, keep the corrected direction if the step grows
        ncorr = 0
        for _ in range(int(gondzio)):
            apt, adt = min(1.0, 1.3 * ap + 0.1), min(1.0, 1.3 * ad + 0.1)
            tgt = sigma * mu
            pl = (wl + apt * dv) * (z + adt * dz); pu = (tu - apt * dv) * (f + adt * df)
            tl = np.clip(pl, 0.1 * tgt, 10.0 * tgt) - pl; tu_ = np.clip(pu, 0.1 * tgt, 10.0 * tgt) - pu
            tl = np.maximum(tl, -10.0 * tgt); tu_ = np.maximum(tu_, -10.0 * tgt)
            cz = np.where(hl, tl / wl, 0.0); cf = np.where(hu, tu_ / tu, 0.0)
            rt = -cz + cf
            dyc = nsolve(Abar @ (theta * rt))
            dvc = theta * (Abar.T @ dyc - rt)
            dzc = np.where(hl, cz - z / wl * dvc, 0.0); dfc = np.where(hu, cf + f / tu * dvc, 0.0)
            ap2, ad2 = steps(dv + dvc, dz + dzc, df + dfc)
            if min(ap2, ad2) >= min(ap, ad) * 1.01 or (ap2 + ad2) >= 1.05 * (ap + ad):
                dv, dy, dz, df = dv + dvc, dy + dyc, dz + dzc, df + dfc
                ap, ad = ap2, ad2
                ncorr += 1
            else:
                break
        solve.ncorr = getattr(solve, "ncorr", 0) + ncorr
        ap, ad = min(1.0, frac * ap), min(1.0, frac * ad)
        if samestep:
            ap = ad = min(ap, ad)
        if nbhd > 0:
            # wide neighbourhood N_-inf(gamma): back off until every complementarity product of the trial point is at least gamma x their mean
            # (experiment, round 6: the slow members take steps of 0.01 - 0.3 for a hundred iterations - a few products run ahead to zero)
            for _ in range(40):
                pl = np.where(hl, (wl + ap * dv) * (z + ad * dz), np.inf); pu = np.where(hu, (tu - ap * dv) * (f + ad * df), np.inf)
                mu_t = (pl[hl].sum() + pu[hu].sum()) / nb
                if min(pl.min(), pu.min()) >= nbhd * mu_t:
                    break
                ap *= nb_back; ad *= nb_back
            solve.nb_backs = getattr(solve, "nb_backs", 0) + _
        v = v + ap * dv; y = y + ad * dy; z = z + ad * dz; f = f + ad * df
        # termination on the unscaled problem, the streaming path's test
        xs = xfix.copy(); xs[cols] = v[:len(cols)]
        Xu = xs * dc
        Yu = y * dr * cs                    # row multipliers: c - A' y = reduced costs
        AX = A @ Xu
        viol = np.maximum(P["rlo"] - AX, 0) + np.maximum(AX - P["rhi"], 0)
        rc = P["c"] - AT @ Yu
        lp_ = np.where(np.isfinite(P["lb"]), np.maximum(rc, 0), 0.0); lm_ = np.where(np.isfinite(P["ub"]), np.maximum(-rc, 0), 0.0)
        dres = rc - lp_ + lm_
        po = P["c"] @ Xu
        do = np.sum(np.maximum(Yu, 0) * fin(P["rlo"]) - np.maximum(-Yu, 0) * fin(P["rhi"])) + np.sum(lp_ * fin(P["lb"]) - lm_ * fin(P["ub"]))
        rpn, rdn = np.linalg.norm(viol) / (1 + qn), np.linalg.norm(dres) / (1 + cn)
        bound = abs(po - do) + np.sum(np.abs(Yu) * viol) + np.sum(np.abs(dres) * np.abs(Xu))
        lim = max(eps_obj * (1 + abs(po + P["c0"])), 1e-12 * np.sum(np.abs(P["c"] * Xu)))
        if verbose > 1:
            rpi = b - Abar @ v
            iv = int(np.argmax(viol)); ir = int(np.argmax(np.abs(rpi)))
            print(f"    internal |b - Abar v| max {np.abs(rpi).max():.2e} at row {ir} ({P['lp'].row_names[ir]}), dr {dr[ir]:.2e}; viol max {viol.max():.2e} at row {iv} ({P['lp'].row_names[iv]}) rlo {P['rlo'][iv]:.6g} rhi {P['rhi'][iv]:.6g} AX {AX[iv]:.9g} dr {dr[iv]:.2e} qn {qn:.3e}")
        if verbose:
            print(f"  it {it} mu {mu:.2e} sigma {sigma:.1e} ap {ap:.3f} ad {ad:.3f} rp {rpn:.2e} rd {rdn:.2e} bound/lim {bound/lim:.2e} obj {po + P['c0']:.10e} "
                  f"theta {theta.min():.1e}..{theta.max():.1e} t {time.time()-t0:.1f}s", flush=True)
        if rpn <= eps and rdn <= eps and bound <= lim:
            if verbose or absref:
                print(f"  [refinement steps total {getattr(solve, 'ref_steps', 0)} max {getattr(solve, 'ref_max', 0)} over {2 * it} systems]")
            solve.ref_steps = 0; solve.ref_max = 0
            return Xu, Yu, it, True
    return Xu, Yu, int(max_iter), False


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    T = int(kw.pop("T", 672)); members = [int(k) for k in kw.pop("member", "5").split(",")]
    cs = kw.pop("colscale", "phys"); fam = kw.pop("family", "base")
    opts = {k: float(v) for k, v in kw.items()}
    for member in members:
        P = lab.build(T, member, None, "chain", family=fam)
        ref, xr, th = lab.highs(P)
        print(f"T={T} member={member} n={P['lp'].n} m={P['lp'].m} nnz={P['lp'].nnz} HiGHS {ref:.10e} ({th:.1f}s)", flush=True)
        if cs == "phys":
            opts["colscale"] = lab.physical_scales(P, T)
        t = time.time()
        X, Y, it, done = solve(P, **opts)
        obj = P["c"] @ X + P["c0"]
        print(f"done={done} newton_iterations={it} obj={obj:.10e} relerr={abs(obj-ref)/max(1,abs(ref)):.2e} t={time.time()-t:.1f}s", flush=True)
