"""Infeasibility / unboundedness certificates from the fixed-point displacement of the restarted Halpern PDHG (development tool;
NOT product, NOT oracle).  python tools/infeas_lab.py

On an infeasible or unbounded LP the operator T of PDHG has no fixed point and T(z) - z converges to the infimal displacement
vector v = (dx, dy): dy is a Farkas ray of the primal problem (primal infeasible), dx a recession direction that improves the
objective (dual infeasible = unbounded).  The lab runs tools/pdlp_lab.py's numpy iteration (the GPU's settings) on three LPs -
a 24-h bidding LP whose initial state of charge is 10 x the battery, the same LP with one free column of negative cost, and a
feasible control - and prints, per check, the two certificates' quality so that the kernel's thresholds can be chosen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.sparse as sp
import pdlp_proto as pp
import pdlp_lab as lab

fin = lambda a: np.where(np.isfinite(a), a, 0.0)


def certificates(d):
    """(primal-infeasibility quality, dual-infeasibility quality) of dz = T(z) - z in the SCALED space, both as
    max violation / certificate value (<= tol means certified); inf when the certificate value has the wrong sign."""
    As, lb, ub, rlo, rhi, c = d["As"], d["lb"], d["ub"], d["rlo"], d["rhi"], d["c"]
    dx, dy = d["xp"] - d["x"], d["yp"] - d["y"]
    AsT = sp.csr_matrix(As.T)
    # --- Farkas ray dy of {rlo <= A x <= rhi, lb <= x <= ub}: rc = -A^T dy split over the finite column bounds
    rc = -(dy @ As)
    lp = np.where(np.isfinite(lb), np.maximum(rc, 0), 0.0); lm = np.where(np.isfinite(ub), np.maximum(-rc, 0), 0.0)
    res = rc - lp + lm
    ypos, yneg = np.maximum(dy, 0), np.maximum(-dy, 0)
    ybad = np.where(np.isfinite(rlo), 0, ypos) + np.where(np.isfinite(rhi), 0, yneg)
    dobj = np.sum(ypos * fin(rlo) - yneg * fin(rhi), 1) + np.sum(lp * fin(lb) - lm * fin(ub), 1)
    dviol = np.sqrt(np.sum(res ** 2, 1) + np.sum(ybad ** 2, 1))
    # scale-free: the ray's objective against (norm of the residual) x (norm of the data the residual could multiply)
    bnorm = np.sqrt(np.sum(fin(lb) ** 2 + fin(ub) ** 2, 1) + np.sum(np.maximum(np.abs(fin(rlo)), np.abs(fin(rhi))) ** 2, 1))
    qp = np.where(dobj > 0, dviol * (1 + bnorm) / np.maximum(dobj, 1e-300), np.inf)
    # --- recession direction dx: c.dx < 0, A dx inside the recession cone of the rows, dx inside that of the bounds
    adx = dx @ AsT
    rv = np.where(np.isfinite(rlo), np.maximum(-adx, 0), 0.0) + np.where(np.isfinite(rhi), np.maximum(adx, 0), 0.0)
    cv = np.where(np.isfinite(lb), np.maximum(-dx, 0), 0.0) + np.where(np.isfinite(ub), np.maximum(dx, 0), 0.0)
    cdx = np.sum(c * dx, 1)
    pviol = np.sqrt(np.sum(rv ** 2, 1) + np.sum(cv ** 2, 1))
    cnorm = np.sqrt(np.sum(c ** 2, 1))
    qd = np.where(cdx < 0, pviol * (1 + cnorm) / np.maximum(-cdx, 1e-300), np.inf)
    return qp, qd, dobj, cdx


def run(name, P, colscale, max_iter=20000):
    log = []

    def hook(it, d):
        qp, qd, dobj, cdx = certificates(d)
        log.append((it, qp.copy(), qd.copy(), d["k"].copy(), d["done"].copy()))
    lab.solve.hook = hook
    kw = dict(lab.GPU_DEFAULTS)
    X, Y, iters, nrs, done = lab.solve(P, max_iter=max_iter, colscale=colscale, **kw)
    lab.solve.hook = None
    B = P.c.shape[0]
    for tol in (1e-6, 1e-8, 1e-10):
        firstp = np.full(B, -1); firstd = np.full(B, -1)
        for it, qp, qd, k, dn in log:
            firstp = np.where((firstp < 0) & (qp <= tol) & ~dn, it, firstp)
            firstd = np.where((firstd < 0) & (qd <= tol) & ~dn, it, firstd)
        print(f"{name}: tol {tol:g}: primal-infeasible certified at {firstp.tolist()}  dual-infeasible at {firstd.tolist()}  (done {done.tolist()} iters {iters.tolist()})")
    best = np.min([np.minimum(q[1], q[2]) for q in log], 0)
    print(f"{name}: best certificate quality over the run {best}")


if __name__ == "__main__":
    B = 6
    model, P = pp.build("wind_battery_24h", B)
    cs = getattr(model.lp, "col_scale", None)
    names = model.lp.col_names
    j_soc = names.index("battery.initial_state_of_charge")
    run("feasible", P, cs)
    Pi = pp.Problem(P.lp, P.c.copy(), P.lb.copy(), P.ub.copy(), P.rlo.copy(), P.rhi.copy(), P.c0.copy())
    Pi.lb[:, j_soc] = Pi.ub[:, j_soc] = 1e6            # 10 x the 100 MWh battery: no discharge path empties it within the SOC bound
    run("soc_init_10x", Pi, cs)
    # an unbounded LP: the under-bid column of hour 5 pays instead of costing (u_5 >= pda_5 - P_T[5] holds for every large u_5)
    Pu = pp.Problem(P.lp, P.c.copy(), P.lb.copy(), P.ub.copy(), P.rlo.copy(), P.rhi.copy(), P.c0.copy())
    ju = names.index("real_time_underbid_power[5]")
    Pu.c[:, ju] = -1.0
    run("negative_cost_ray", Pu, cs)
    # the same with the day-ahead offer of that hour free in both directions and paid: pda_5 -> -inf along the row's recession cone
    Pv = pp.Problem(P.lp, P.c.copy(), P.lb.copy(), P.ub.copy(), P.rlo.copy(), P.rhi.copy(), P.c0.copy())
    jp = names.index("day_ahead_power[5]")
    Pv.lb[:, jp] = -np.inf; Pv.c[:, jp] = 3.0
    run("free_column_ray", Pv, cs)


def trajectory(P, colscale, s=0, max_iter=20000, every=40):
    def hook(it, d):
        if (it // 16) % every == 0:
            qp, qd, dobj, cdx = certificates(d)
            dx = d["xp"] - d["x"]
            print(it, "k", int(d["k"][s]), "w %.3g" % d["w"][s], "qd %.3e" % qd[s], "c.dx %.3e" % cdx[s], "|dx| %.3e" % np.linalg.norm(dx[s]), "|x| %.3e" % np.linalg.norm(d["xp"][s]))
    lab.solve.hook = hook
    lab.solve(P, max_iter=max_iter, colscale=colscale, **lab.GPU_DEFAULTS)
    lab.solve.hook = None
