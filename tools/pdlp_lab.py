"""Algorithm lab (development tool; NOT product, NOT oracle): batched numpy r2HPDHG with switchable primal-weight /
restart rules, used to pick the rules the HIP kernel implements.  python tools/pdlp_lab.py <workload> <B> [k=v ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.sparse as sp
import pdlp_proto as pp

fin = lambda a: np.where(np.isfinite(a), a, 0.0)


def geo_scaling(A, iters=8):
    """geometric-mean equilibration: r_i = 1/sqrt(max_j|a_ij| * min_j|a_ij|) over the nonzeros, same for columns"""
    A = sp.csr_matrix(A, copy=True).astype(float)
    m, n = A.shape
    dr, dc = np.ones(m), np.ones(n)
    for _ in range(iters):
        coo = A.tocoo(); a = np.abs(coo.data)
        rmax = np.zeros(m); rmin = np.full(m, np.inf); np.maximum.at(rmax, coo.row, a); np.minimum.at(rmin, coo.row, a)
        r = np.where(rmax > 0, 1 / np.sqrt(rmax * np.where(np.isfinite(rmin), rmin, 1)), 1.0)
        A = sp.diags(r) @ A; dr *= r
        coo = A.tocoo(); a = np.abs(coo.data)
        cmax = np.zeros(n); cmin = np.full(n, np.inf); np.maximum.at(cmax, coo.col, a); np.minimum.at(cmin, coo.col, a)
        c = np.where(cmax > 0, 1 / np.sqrt(cmax * np.where(np.isfinite(cmin), cmin, 1)), 1.0)
        A = A @ sp.diags(c); dc *= c
    return sp.csr_matrix(A), dr, dc


def solve(P, eps=1e-9, max_iter=60000, check=32, wrule="pid", kp=0.7, maxdl=np.log(30), r0_mode="k1",
          beta=(0.2, 0.8, 0.36), eta_scale=0.998, n_ruiz=10, wclamp=None, bal_gain=0.25, bal_thresh=10.0,
          c0_gap=True, verbose=False, colscale=None, stall=0, stall_frac=0.5, kp_decay=0.5, xinit=0, term=0, eps_obj=1e-7, jump=0, jtol=1e-2, jsteady=0.05, jmin=4.0, winit=0.0, geo=0, jumpw=0, wfloor=0.0, jk=2.0, jacc=0.0, jchain=0, jland=-1.0, wfreeze=0, wfreeze_err=0.0, retry=0.0, retry_kps=(0.5, 0.35, 0.7, 0.25), rescue_k=0, rescue_mode=0, rescue_zone=0.0, jrel=0.0, polish=0, polish_mult=4.0, polish_max=3, refl=1.0, lam_shift=0.0):
    lp = P.lp
    A0 = P.A
    if colscale is not None:
        A0 = sp.csr_matrix(A0 @ sp.diags(colscale))
    if geo:
        A0, gr, gc = geo_scaling(A0, int(geo))
    As, dr, dc = pp.ruiz_pc_scaling(A0, n_ruiz=int(n_ruiz))
    if geo:
        dr, dc = dr * gr, dc * gc
    if colscale is not None:
        dc = dc * colscale
    AsT = sp.csr_matrix(As.T)
    B, n, m = P.c.shape[0], lp.n, lp.m
    c = P.c * dc; lb, ub = P.lb / dc, P.ub / dc; rlo, rhi = P.rlo * dr, P.rhi * dr
    eta = eta_scale / pp.spectral_norm(As)
    qs = np.sqrt(np.sum(np.maximum(np.abs(fin(rlo)), np.abs(fin(rhi))) ** 2, 1)); cs = np.linalg.norm(c, axis=1)
    w = np.where((cs > 1e-10) & (qs > 1e-10), cs / np.maximum(qs, 1e-300), 1.0)
    if winit:
        w = np.full(B, winit)
    w0 = w.copy()
    x = np.clip(np.zeros((B, n)), lb, ub)
    if xinit:
        x = np.where((c < 0) & np.isfinite(ub), ub, x)
    y = np.zeros((B, m)); x0, y0 = x.copy(), y.copy()
    k = np.zeros(B); r0 = np.full(B, np.inf); rprev = np.full(B, np.inf)
    done = np.zeros(B, bool); iters = np.full(B, max_iter); Xo = np.zeros((B, n)); Yo = np.zeros((B, m)); nrs = np.zeros(B, int)
    c0 = P.c0 if c0_gap else 0.0
    best = np.full(B, np.inf); wbest = w.copy(); itbest = np.zeros(B); kpv = np.full(B, kp); nrev = np.zeros(B, int)
    max_iter = int(max_iter)
    njump = np.zeros(B, int); jtot = np.zeros(B); lastjump = np.zeros(B, bool); wfrozen = np.zeros(B, bool)
    attempt = np.zeros(B, int); t_attempt = np.zeros(B); budget = np.full(B, retry * (n + m) if retry else np.inf)
    pbest = np.full(B, np.inf); pit = np.zeros(B); wmult = np.ones(B); wdiv = np.ones(B); nboost = np.zeros(B, int); boosted = np.zeros(B, bool)
    xstart0 = x.copy(); xstart1 = np.where((c < 0) & np.isfinite(ub), ub, x)
    for it in range(max_iter):
        tau = (eta / w)[:, None]; sig = (eta * w)[:, None]
        xp = np.clip(x - tau * (c - y @ As), lb, ub)
        Axb = (2 * xp - x) @ AsT
        wv = y - sig * Axb
        yp = wv + np.clip(-wv, sig * rlo, sig * rhi)
        k += 1
        chk = (it + 1) % check == 0
        need = (k == 1) if r0_mode == "k1" else np.zeros(B, bool)
        if chk or need.any():
            dx, dy = xp - x, yp - y
            r = np.sqrt(np.maximum(w * np.sum(dx * dx, 1) - 2 * eta * np.sum(dy * (dx @ AsT), 1) + np.sum(dy * dy, 1) / w, 0))
            r0 = np.where(need, r, r0); rprev = np.where(need, r, rprev)
        rs = np.zeros(B, bool); jumped = np.zeros(B, bool)
        if chk:
            rp, rd, rg, po, do = pp.kkt_unscaled(P, xp * dc, yp * dr)
            if c0_gap:
                rg = np.abs(po - do) / (1 + np.abs(po) + np.abs(do))      # po/do already include c0
            else:
                rg = np.abs(po - do) / (1 + np.abs(po - P.c0) + np.abs(do - P.c0))
            rg_raw = rg.copy()
            if term:
                Xu, Yu = xp * dc, yp * dr
                AX = Xu @ P.A.T
                viol = np.maximum(P.rlo - AX, 0) + np.maximum(AX - P.rhi, 0)
                ierr = np.sum(np.abs(Yu) * viol, 1)
                scale = np.sum(np.abs(P.c * Xu), 1)
                gap = np.abs(po - do)
                okg = gap <= np.maximum(eps_obj * (1 + np.abs(po)), 1e-12 * scale)
                oki = ierr <= np.maximum(eps_obj * (1 + np.abs(po)), 1e-12 * scale)
                rc_ = P.c - Yu @ P.A
                lp_ = np.where(np.isfinite(P.lb), np.maximum(rc_, 0), 0.0); lm_ = np.where(np.isfinite(P.ub), np.maximum(-rc_, 0), 0.0)
                derr = np.sum(np.abs(rc_ - lp_ + lm_) * np.abs(Xu), 1)
                oki &= derr <= np.maximum(eps_obj * (1 + np.abs(po)), 1e-12 * scale)
                rg = np.where(okg & oki, rg, np.maximum(rg, 2 * eps))
                lim_ = np.maximum(eps_obj * (1 + np.abs(po)), 1e-12 * scale)
                solve.rho_obj = np.maximum(np.maximum(gap, ierr), derr) / lim_
                if polish:
                    # polish phase: the eps_rel tests hold, only the objective-accuracy tests are missing.  No 2x improvement of
                    # their worst ratio for `polish` iterations = the iterate sits on its rounding floor: the step that
                    # amplifies the noise is shortened (primal noise: w up; dual noise: w down) through the weight guard
                    inpol = (rp <= eps) & (rd <= eps) & ~done       # (the kernel's relative gap is taken without c0 and passes long before)
                    imp_ = inpol & (solve.rho_obj < 0.5 * pbest)
                    pbest = np.where(imp_, solve.rho_obj, pbest); pit = np.where(imp_ | ~inpol, it + 1, pit)
                    boosted = inpol & ~imp_ & (it + 1 - pit >= polish) & (nboost < polish_max) & (solve.rho_obj > 1)
                    prim = np.maximum(gap, ierr) >= derr
                    wmult = np.where(boosted & prim, wmult * polish_mult, wmult); wdiv = np.where(boosted & ~prim, wdiv * polish_mult, wdiv)
                    nboost += boosted; pbest = np.where(boosted, np.inf, pbest); pit = np.where(boosted, it + 1, pit)
                    solve.boosts = getattr(solve, "boosts", 0) + int(boosted.sum())
            if getattr(solve, "hook", None) is not None:       # development hook (tools/polish_lab.py)
                solve.hook(it + 1, dict(xp=xp, yp=yp, gx=x - tau * (c - y @ As), gy=wv, lb=lb, ub=ub, rlo=rlo, rhi=rhi,
                                        sig=sig, tau=tau, As=As, c=c, done=done, dc=dc, dr=dr, w=w, x=x, y=y, k=k, x0=x0, y0=y0))
            conv = (rp <= eps) & (rd <= eps) & (rg <= eps) & ~done
            if getattr(solve, "trace", None) is not None:      # (iteration, r, rho = worst criterion / its limit, done)
                rho = np.maximum(np.maximum(rp, rd), rg) / eps
                if term:
                    rho = np.maximum(rho, solve.rho_obj)
                solve.trace.append((it + 1, r.copy(), rho, done.copy()))
            Xo[conv], Yo[conv] = (xp * dc)[conv], (yp * dr)[conv]; iters[conv] = it + 1; done |= conv
            if done.all():
                break
            err = np.maximum(np.maximum(rp, rd), rg)
            imp = err < 0.5 * best
            best = np.where(imp, err, best); wbest = np.where(imp, w, wbest); itbest = np.where(imp, it + 1, itbest)
            first = ~np.isfinite(r0)
            r0 = np.where(first, r, r0)
            rs = ~first & ((r <= beta[0] * r0) | ((r <= beta[1] * r0) & (r > rprev)) | (k >= beta[2] * (it + 1 - t_attempt)))
            if polish:
                rs = rs | boosted
            stalled = ~first & ~((r <= beta[0] * r0) | ((r <= beta[1] * r0) & (r > rprev))) & (k >= beta[2] * (it + 1 - t_attempt)) & (k >= rescue_k) if rescue_k else np.zeros(B, bool)
            steady = (np.abs(r - rprev) <= jsteady * r) & (k >= jk * check) & ~done & ~rs if jump else np.zeros(B, bool)
            if jump and jchain:
                steady = steady | (lastjump & (k >= check) & ~done & ~rs)
            rprev = r
            if jump and steady.any():
                gx0 = x - tau * (c - y @ As); gy0 = wv
                gx1 = xp - tau * (c - yp @ As); x2 = np.clip(gx1, lb, ub)
                gy1 = yp - sig * ((2 * x2 - xp) @ AsT); y2 = gy1 + np.clip(-gy1, sig * rlo, sig * rhi)
                v1x, v1y, v2x, v2y = xp - x, yp - y, x2 - xp, y2 - yp
                nv = lambda a, b: np.sqrt(w * np.sum(a * a, 1) + np.sum(b * b, 1) / w)
                trans = steady & (nv(v2x - v1x, v2y - v1y) <= jtol * nv(v2x, v2y)) & (nv(v2x, v2y) > 0)
                if trans.any():
                    dgx = gx1 - gx0; dgy = -(gy1 - gy0); g = -gy1; lo_, hi_ = sig * rlo, sig * rhi
                    big = 1e300
                    def ratio(gv, dg, lo, hi):
                        with np.errstate(divide='ignore', invalid='ignore'):
                            a_up = np.where(dg > 0, np.where(gv < lo, (lo - gv) / dg, np.where(gv <= hi, (hi - gv) / dg, big)), big)
                            a_dn = np.where(dg < 0, np.where(gv > hi, (hi - gv) / dg, np.where(gv >= lo, (lo - gv) / dg, big)), big)
                        a = np.minimum(a_up, a_dn)
                        return np.where(np.isfinite(a), a, big)
                    ax_ = ratio(gx1, dgx, lb, ub).min(1); ay_ = ratio(g, dgy, lo_, hi_).min(1)
                    alpha = np.minimum(ax_, ay_)
                    dojump = trans & (alpha >= jmin) & (alpha >= jrel * k) & (alpha < 1e200)
                    if getattr(solve, "jump_hook", None) is not None:      # development hook (tools/infeas_lab.py)
                        stop = solve.jump_hook(it + 1, dict(alpha=alpha, trans=trans, x=x, y=y, xp=xp, yp=yp, x2=x2, y2=y2, lb=lb, ub=ub, rlo=rlo, rhi=rhi,
                                                            As=As, c=c, k=k, w=w, done=done))
                        if stop is not None:
                            newly = stop & ~done
                            iters[newly] = it + 1; done |= newly
                            if done.all():
                                break
                    if dojump.any():
                        a = np.where(dojump, np.maximum(np.floor(alpha) + jland, 0.0), 0.0)[:, None]
                        xn = np.clip(x2 + a * v2x, lb, ub); yn = y2 + a * v2y
                        if jacc:
                            xl = np.clip(xn - tau * (c - yn @ As), lb, ub); gl = yn - sig * ((2 * xl - xn) @ AsT); yl = gl + np.clip(-gl, sig * rlo, sig * rhi)
                            dxl, dyl = xl - xn, yl - yn
                            rl = np.sqrt(np.maximum(w * np.sum(dxl * dxl, 1) - 2 * eta * np.sum(dyl * (dxl @ AsT), 1) + np.sum(dyl * dyl, 1) / w, 0))
                            rej = dojump & ~(rl <= (1 + jacc) * r)
                            solve.rejected = getattr(solve, 'rejected', 0) + rej.sum()
                            dojump = dojump & ~rej
                        if jumpw:
                            ddx = np.linalg.norm(xn - x0, axis=1); ddy = np.linalg.norm(yn - y0, axis=1)
                            okj = dojump & (ddx > 1e-14) & (ddy > 1e-14)
                            ej = np.where(okj, np.log(w) + np.log(np.maximum(ddx, 1e-300)) - np.log(np.maximum(ddy, 1e-300)), 0.0)
                            w = w * np.exp(np.clip(-kpv * ej, -maxdl, maxdl))
                        m_ = dojump[:, None]
                        x = np.where(m_, xn, x); y = np.where(m_, yn, y); x0 = np.where(m_, xn, x0); y0 = np.where(m_, yn, y0)
                        xp = np.where(m_, xn, xp); yp = np.where(m_, yn, yp)
                        k = np.where(dojump, 0, k); r0 = np.where(dojump, np.inf, r0); rprev = np.where(dojump, np.inf, rprev)
                        njump += dojump; jtot += np.where(dojump, alpha, 0)
                        jumped = dojump
                        lastjump = lastjump | dojump
                        if verbose: print(it + 1, "jump", np.nonzero(dojump)[0], alpha[dojump])
            if rs.any():
                ddx = np.linalg.norm(xp - x0, axis=1); ddy = np.linalg.norm(yp - y0, axis=1)
                ok = rs & (ddx > 1e-14) & (ddy > 1e-14)
                e = np.where(ok, np.log(w) + np.log(np.maximum(ddx, 1e-300)) - np.log(np.maximum(ddy, 1e-300)), 0.0)
                dl = np.clip(-kpv * e, -maxdl, maxdl)
                if wrule == "balance":
                    ratio = np.log(np.maximum(rp, 1e-300) / np.maximum(rd, 1e-300))
                    dl = np.clip(bal_gain * ratio, -maxdl, maxdl)
                elif wrule == "hybrid":
                    ratio = np.maximum(rp, 1e-300) / np.maximum(rd, 1e-300)
                    dl = np.where((ratio > bal_thresh) & (dl < 0), 0.0, dl)
                    dl = np.where((ratio < 1 / bal_thresh) & (dl > 0), 0.0, dl)
                elif wrule == "hybrid2":
                    ratio = np.log(np.maximum(rp, 1e-300) / np.maximum(rd, 1e-300))
                    bad = ((ratio > np.log(bal_thresh)) & (dl < 0)) | ((ratio < -np.log(bal_thresh)) & (dl > 0))
                    dl = np.where(bad, np.clip(bal_gain * ratio, -maxdl, maxdl), dl)
                if wfreeze:
                    dl = np.where(nrs >= wfreeze, 0.0, dl)
                if wfreeze_err:
                    frozen_ = np.maximum(np.maximum(rp, rd), rg) <= wfreeze_err
                    wfrozen[:] = wfrozen | frozen_
                    dl = np.where(wfrozen, 0.0, dl)
                logw = np.log(w) + np.where(rs, dl, 0.0)
                if rescue_k and rescue_zone:
                    cmax_ = np.max(np.abs(c), 1); qall_ = np.sqrt(qs ** 2 + np.sum(fin(lb) ** 2 + fin(ub) ** 2, 1))
                    wlo_ = wfloor * eta * 1.1e-16 * cmax_ / (eps * (1 + qall_))
                    if verbose and stalled.any(): print(it + 1, "stalled", w[stalled], "w_lo", wlo_[stalled])
                    stalled = stalled & (w < rescue_zone * wlo_)
                if rescue_k and stalled.any():
                    # stalled at the noise floor: pull the weight back toward its initial value
                    tgt = 0.5 * (np.log(w) + np.log(w0)) if rescue_mode == 0 else np.log(w) + np.sign(np.log(w0) - np.log(w)) * np.log(30.0)
                    logw = np.where(stalled, tgt, logw)
                    solve.rescues = getattr(solve, "rescues", 0) + int(stalled.sum())
                    if verbose: print(it + 1, "RESCUE", np.nonzero(stalled)[0], w[stalled], "->", np.exp(logw[stalled]))
                if verbose: print(it + 1, 'restart', np.nonzero(rs)[0][:4], 'k', k[rs][:4], 'r', r[rs][:4], 'w', w[rs][:4], '->', np.exp(logw[rs][:4]), 'kkt', rp[rs][:4], rd[rs][:4], rg[rs][:4])
                if stall:
                    st = rs & ((it + 1 - itbest) > stall + stall_frac * itbest)
                    logw = np.where(st, np.log(wbest), logw); kpv = np.where(st, kpv * kp_decay, kpv); itbest = np.where(st, it + 1, itbest); nrev += st
                if wclamp is not None:
                    logw = np.clip(logw, np.log(w0) - np.log(wclamp), np.log(w0) + np.log(wclamp))
                w = np.exp(logw)
                if wfloor:
                    cmax = np.max(np.abs(c), 1)
                    qall = np.sqrt(qs ** 2 + np.sum(fin(lb) ** 2 + fin(ub) ** 2, 1))
                    w = np.maximum(w, wmult * wfloor * eta * 1.1e-16 * cmax / (eps * (1 + qall)))
                    w = np.minimum(w, 1.0 / (wfloor * eta * 1.1e-16 * np.max(np.abs(np.where(np.isfinite(rlo), rlo, 0)) + np.abs(np.where(np.isfinite(rhi), rhi, 0)), 1) / (eps * (1 + cs)) + 1e-300) / wdiv)
                m_ = rs[:, None]
                x = np.where(m_, xp, x); y = np.where(m_, yp, y); x0 = np.where(m_, xp, x0); y0 = np.where(m_, yp, y0)
                k = np.where(rs, 0, k); r0 = np.where(rs, np.inf, r0); rprev = np.where(rs, np.inf, rprev); nrs += rs
                lastjump = lastjump & ~rs
        if retry and chk:
            rt = ~done & ((it + 1 - t_attempt) >= budget)
            if rt.any():
                attempt = attempt + rt
                kpv = np.where(rt, np.array(retry_kps)[(attempt - 1) % len(retry_kps)], kpv)
                xs = np.where((attempt % 2 == 1)[:, None], xstart1, xstart0)
                m_ = rt[:, None]
                x = np.where(m_, xs, x); y = np.where(m_, 0.0, y); x0 = np.where(m_, xs, x0); y0 = np.where(m_, 0.0, y0)
                xp = np.where(m_, xs, xp); yp = np.where(m_, 0.0, yp)
                w = np.where(rt, w0, w); k = np.where(rt, 0, k); r0 = np.where(rt, np.inf, r0); rprev = np.where(rt, np.inf, rprev)
                t_attempt = np.where(rt, it + 1, t_attempt); budget = np.where(rt, budget * 2, budget)
                jumped = jumped | rt
                if verbose: print(it + 1, 'RETRY', np.nonzero(rt)[0], attempt[rt])
        keep = ~(rs | jumped)
        # refl: reflection strength rho of the relaxed operator (1 + rho) T - rho I (the kernel: 1); lam_shift: anchor weight
        # (k + 1 + s) / (k + 2 + s) - exploration only
        lam = ((k + 1 + lam_shift) / (k + 2 + lam_shift))[:, None]
        x = np.where(keep[:, None], lam * ((1 + refl) * xp - refl * x) + (1 - lam) * x0, x)
        y = np.where(keep[:, None], lam * ((1 + refl) * yp - refl * y) + (1 - lam) * y0, y)
    Xo[~done], Yo[~done] = (xp * dc)[~done], (yp * dr)[~done]
    solve.last_jumps = (njump, jtot); solve.last_w = w; solve.last_attempts = attempt
    return Xo, Yo, iters, nrs, done


# The settings that mirror the kernel's defaults (include/dsp_hip.h): python tools/pdlp_lab.py wind_battery_24h gpu
# Not mirrored here: the residual-gated KKT schedule (moves only the stopping time), the stall logic of
# dsp_options::stall_rescue (use rescue_k / rescue_zone for its first stage) and the re-test delay after a too-short ray.
# (medians of this lab match the GPU's: wind+battery 24 h p50 1776 under beta_3 = 0.36 / kp = 0.7 and 1600 vs 1568 under the
#  shipped 0.2 / 0.6 - profiles/r04j_restart_scan.log; the tails differ, the near-miss logic is not mirrored)
GPU_DEFAULTS = dict(check=16, jump=1, jtol=3e-3, jsteady=0.05, jmin=4.0, jrel=3.0, term=1, eps_obj=5e-7, wfloor=4, kp=0.6,
                    beta=(0.2, 0.8, 0.2))


HARD = {"wind_battery_24h": [746, 1449, 2233, 2445, 2768], "wind_battery_48h": [527, 1216, 1847, 2636, 3147, 3562]}


def subset(wl, nrand=int(os.environ.get("NRAND", 27)), seed=0, B=4096):
    model, P = pp.build(wl, B)
    rng = np.random.default_rng(seed)
    ids = HARD.get(wl, []) + sorted(rng.choice(B, nrand, replace=False).tolist())
    return ids, pp.Problem(P.lp, P.c[ids], P.lb[ids], P.ub[ids], P.rlo[ids], P.rhi[ids], P.c0[ids])


if __name__ == "__main__":
    wl = sys.argv[1]
    ids, sub = subset(wl)
    ref = np.array([pp.highs_obj(sub, i)[0] for i in range(len(ids))])
    variants = [{**(GPU_DEFAULTS if "gpu" in v.split(",") else {}), **dict(a.split("=") for a in v.split(",") if a and a != "gpu")}
                for v in sys.argv[2:]] or [{}]
    # the product's variable scaling factors for this model family (lp.implied_column_ranges through the Bidder), if it asks for
    # them: every variant runs with them unless it says colscale=0
    product_scale = getattr(pp.build(wl, 4)[0].lp, "col_scale", None)
    for kw in variants:
        kw = {k: (v if k in ("wrule", "r0_mode") or not isinstance(v, str) else float(v)) for k, v in kw.items()}
        if kw.pop("colscale", 1.0) and product_scale is not None:
            kw["colscale"] = product_scale
        t = time.time()
        X, Y, iters, nrs, done = solve(sub, **kw)
        obj = np.sum(sub.c * X, 1) + sub.c0
        err = np.abs(obj - ref) / np.maximum(1, np.abs(ref))
        print({k: v for k, v in kw.items() if k != "colscale"}, "scaled" if "colscale" in kw else "unscaled", f"done {done.sum()}/{len(ids)} mean {iters.mean():.0f} med {np.median(iters):.0f} max {iters.max()} "
              f"hard {iters[:len(HARD.get(wl, []))]} jumps {solve.last_jumps[0].sum()} jsum {solve.last_jumps[1].sum():.0f} maxerr {err[done].max():.2e} t {time.time()-t:.1f}s", flush=True)
