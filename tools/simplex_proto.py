"""numpy prototype of the in-wave dense simplex for the TINY hourly LPs (development tool; NOT product, NOT oracle).

Executable specification of csrc/dsp_simplex.hip: bounded-variable primal simplex on the dense tableau, one LP per
wave, all LPs of the batch advanced in lock step here (numpy batch axis = the GPU's waves).

    min c.x   s.t.  rlo <= A x <= rhi,  lb <= x <= ub        (scaled by the handle's D_r, D_c)
    z = (x, s),  s = A x  in [rlo, rhi];  start: all slacks basic, structurals at the finite bound nearest to 0
    phase 1: minimise the sum of bound violations of the basic variables (costs -1 / +1 on violated basics),
             an infeasible basic blocks when it reaches the bound it violates
    phase 2: Dantzig pricing on reduced costs recomputed from the tableau every pivot (no drift), two-pass ratio test
             (min ratio with tolerance, then the largest pivot among the ties), bound flips
Run:  python tools/simplex_proto.py            (all six hourly fixtures, 4096 scenarios each, against the oracle)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

BIG = 1e300


def simplex_batch(A, c, lb, ub, rlo, rhi, max_pivots=None, tol_p=1e-9, tol_d=1e-9, tol_piv=1e-9):
    """A [m, n] dense (shared), c/lb/ub [B, n], rlo/rhi [B, m].  Returns x [B, n], y [B, m], status [B], pivots [B]."""
    m, n = A.shape
    B = c.shape[0]
    N = n + m
    max_pivots = max_pivots or 20 * N
    T = np.tile(np.hstack([-A, np.eye(m)])[None], (B, 1, 1))           # row i: s_i - sum_j a_ij x_j = 0, basis = slacks
    lo = np.hstack([lb, rlo]); hi = np.hstack([ub, rhi])                # [B, N]
    cost = np.hstack([c, np.zeros((B, m))])
    fixed = lo == hi
    # nonbasic structurals at the finite bound nearest to zero (all dispatch columns have lb = 0)
    val = np.where(np.isfinite(lo) & (np.abs(lo) <= np.abs(np.where(np.isfinite(hi), hi, BIG))), lo,
                   np.where(np.isfinite(hi), hi, np.where(np.isfinite(lo), lo, 0.0)))
    at_upper = np.isfinite(hi) & (val == hi) & ~(val == lo)
    basis = np.tile(np.arange(n, N)[None], (B, 1))                       # [B, m] variable index of each row's basic
    is_basic = np.zeros((B, N), bool); is_basic[:, n:] = True
    beta = val[:, :n] @ A.T                                              # s = A x_N
    val[:, n:] = beta
    status = np.full(B, -1)             # -1 running, 0 optimal, 1 pivot limit, 2 infeasible, 3 unbounded
    pivots = np.zeros(B, int)
    rows = np.arange(B)
    ctol = tol_d * (1.0 + np.abs(c).max(1))
    for it in range(max_pivots + 1):
        run = status < 0
        if not run.any():
            break
        blo = np.take_along_axis(lo, basis, 1); bhi = np.take_along_axis(hi, basis, 1)
        ptol = tol_p * (1.0 + np.maximum(np.abs(np.where(np.isfinite(blo), blo, 0)), np.abs(np.where(np.isfinite(bhi), bhi, 0))))
        below = beta < blo - ptol; above = beta > bhi + ptol
        phase1 = (below | above).any(1)
        # basic costs: phase 1 -> -1 below / +1 above, phase 2 -> the LP costs
        cB = np.where(phase1[:, None], np.where(below, -1.0, np.where(above, 1.0, 0.0)), np.take_along_axis(cost, basis, 1))
        cN = np.where(phase1[:, None], 0.0, cost)
        d = cN - np.einsum("bi,bij->bj", cB, T)                           # reduced costs of every column
        dtol = np.where(phase1, 1e-9, ctol)[:, None]
        elig = ~is_basic & ~fixed & (((~at_upper) & (d < -dtol)) | (at_upper & (d > dtol)))
        score = np.where(elig, np.abs(d), -1.0)
        j = score.argmax(1)
        none = score[rows, j] < 0
        status = np.where(run & none, np.where(phase1, 2, 0), status)
        run = status < 0
        if not run.any():
            break
        if it == max_pivots:
            status = np.where(run, 1, status)
            break
        sgn = np.where(at_upper[rows, j], -1.0, 1.0)                      # entering moves up from its lower / down from its upper bound
        alpha = T[rows, :, j]                                             # [B, m]
        delta = -sgn[:, None] * alpha                                     # d beta / d t
        amax = np.abs(alpha).max(1, keepdims=True)
        ptv = tol_piv * np.maximum(1.0, amax)
        dec = delta < -ptv; inc = delta > ptv
        # distance to the blocking bound of every basic variable
        t_i = np.full((B, m), BIG)
        feas = ~below & ~above
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            t_i = np.where(feas & dec & np.isfinite(blo), (beta - blo) / -delta, t_i)
            t_i = np.where(feas & inc & np.isfinite(bhi), (bhi - beta) / delta, t_i)
            t_i = np.where(below & inc, (blo - beta) / delta, t_i)       # infeasible basic reaches the bound it violates
            t_i = np.where(above & dec, (beta - bhi) / -delta, t_i)
        t_i = np.maximum(t_i, 0.0)
        t_flip = (hi - lo)[rows, j]
        t_flip = np.where(np.isfinite(t_flip), t_flip, BIG)
        tmin = np.minimum(t_i.min(1), t_flip)
        unb = run & (tmin >= BIG)
        status = np.where(unb, np.where(phase1, 2, 3), status)
        run = status < 0
        # second pass: among the rows within a hair of the minimum take the largest pivot
        tie = t_i <= (tmin * (1 + 1e-9) + 1e-12)[:, None]
        r = np.where(tie, np.abs(alpha), -1.0).argmax(1)
        flip = run & (t_flip <= tmin) & ~(tie.any(1) & (t_flip >= tmin) & False)
        flip = run & (t_flip <= t_i.min(1))
        piv = run & ~flip
        t = np.where(run, tmin, 0.0)
        # move
        beta = beta + delta * t[:, None]
        newval = val[rows, j] + sgn * t
        # bound flip: entering stays nonbasic at its other bound
        val[rows[flip], j[flip]] = newval[flip]
        at_upper[rows[flip], j[flip]] = ~at_upper[rows[flip], j[flip]]
        if piv.any():
            pb = rows[piv]; pr = r[piv]; pj = j[piv]
            leave = basis[pb, pr]
            # leaving variable goes to the bound it reached
            lv = beta[pb, pr]
            l_lo, l_hi = lo[pb, leave], hi[pb, leave]
            to_upper = np.abs(lv - l_hi) < np.abs(lv - l_lo)
            to_upper = np.where(np.isfinite(l_hi) & ~np.isfinite(l_lo), True, np.where(~np.isfinite(l_hi), False, to_upper))
            val[pb, leave] = np.where(to_upper, l_hi, l_lo)
            at_upper[pb, leave] = to_upper
            is_basic[pb, leave] = False
            is_basic[pb, pj] = True
            basis[pb, pr] = pj
            beta[pb, pr] = newval[piv]
            # tableau update
            prow = T[pb, pr, :] / alpha[pb, pr][:, None]                   # [P, N]
            colj = alpha[pb].copy()                                       # [P, m]
            T[pb] -= colj[:, :, None] * prow[:, None, :]
            T[pb, pr, :] = prow
            pivots[pb] += 1
    # solution
    x = val.copy()
    np.put_along_axis(x, basis, beta, 1)
    cB = np.take_along_axis(cost, basis, 1)
    d = cost - np.einsum("bi,bij->bj", cB, T)
    y = d[:, n:]                                                         # multiplier of row i = reduced cost of its slack
    return x[:, :n], y, status, pivots


def solve_model(model, max_pivots=None):
    """Product-side ScenarioBatchModel -> scaled dense data -> simplex -> unscaled (x, y, obj)."""
    import scipy.sparse as sp
    import pdlp_proto as pp
    lp = model.lp
    A = lp.csr()
    As, dr, dc = pp.ruiz_pc_scaling(A)
    lb, ub, rlo, rhi = [np.broadcast_to(a, (model.n_scenario, a.shape[-1])) for a in model.scenario_bounds()]
    x, y, st, piv = simplex_batch(As.toarray(), model.c * dc, lb / dc, ub / dc, rlo * dr, rhi * dr, max_pivots)
    X = x * dc
    return X, y * dr, np.sum(model.c * X, 1) + model.c0, st, piv


if __name__ == "__main__":
    import time
    from dispatches_amd import scenarios
    fx = np.load(os.path.join(ROOT, "tests", "golden", "oracle_hourly.npz"))
    Bmax = int(sys.argv[1]) if len(sys.argv) > 1 else 4096

    class _S:
        def solve(self, *a, **k):
            raise RuntimeError
    for case in ("wind_battery_rt4", "wind_pem_rt4", "nuclear_rt12", "wind_battery_track4", "wind_pem_track4", "nuclear_track4"):
        inp = {k.split("/", 1)[1]: fx[k][:Bmax] for k in fx.files if k.startswith(case + "/")}
        if "rt" in case.split("_")[-1]:
            _, model = scenarios.hourly_bid_batch(case, inp, _S())
            shift = (inp["da"] * inp["dispatch"]).sum(1)
        else:
            _, model = scenarios.hourly_tracking_batch(case, inp, _S())
            shift = 0.0
        t = time.time()
        X, Y, obj, st, piv = solve_model(model)
        err = np.abs(obj + shift - inp["obj"]) / np.maximum(1, np.abs(inp["obj"]))
        model.x = X
        pt = model.expression_values("P_T")
        lo, hi = inp["P_T_lo"], inp["P_T_lo"] + inp["P_T_width"].astype(float)
        viol = np.maximum(lo - pt, pt - hi) / (1e-6 * np.maximum(1, np.abs(pt)))
        print(f"{case}: n={model.lp.n} m={model.lp.m}  status {np.bincount(st + 1, minlength=5)[1:]}  pivots mean {piv.mean():.1f} max {piv.max()}"
              f"  obj err max {err[st == 0].max():.2e}  setpoint viol max {viol[st == 0].max():.2e} x tol  ({time.time() - t:.1f} s)")
