"""Development lab (NOT product, NOT oracle): at which PDHG iteration would an exact active-set finish succeed?

At every check of the numpy r2HPDHG (tools/pdlp_lab.py, GPU-default settings) the active set is read off the last PDHG
application (columns whose unprojected step left [lb, ub] are fixed at that bound, rows whose unprojected dual step
left the clamp window are active on that side), the two linear systems

    A[act, free] x_free = b_act - A[act, fixed] x_fixed          (primal vertex / face point)
    A[act, free]^T y_act = c_free                                 (dual)

are solved in the least-squares sense, and the result is accepted when it satisfies every sign / bound / feasibility
condition of the KKT system.  Prints, per scenario, the first iteration at which that works next to the iteration at
which PDHG itself terminates.     python tools/polish_lab.py wind_battery_24h [nrand]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import pdlp_lab as lab
import pdlp_proto as pp

fin = lambda a: np.where(np.isfinite(a), a, 0.0)


def polish_one(d, s, tol=float(os.environ.get("POLISH_TOL", 1e-9))):
    As = d["As"]
    A = As.toarray() if not hasattr(polish_one, "_A") or polish_one._A[0] is not As else polish_one._A[1]
    polish_one._A = (As, A)
    gx, gy, lb, ub, rlo, rhi = d["gx"][s], d["gy"][s], d["lb"][s], d["ub"][s], d["rlo"][s], d["rhi"][s]
    sig, c = d["sig"][s, 0], d["c"][s]
    atL, atU = gx <= lb, gx >= ub
    F = ~(atL | atU)
    xN = np.where(atL, lb, np.where(atU, ub, 0.0))
    eq = np.isfinite(rlo) & (rlo == rhi)
    actL = (-gy < sig * rlo) & ~eq          # y > 0: lower side active
    actU = (-gy > sig * rhi) & ~eq
    act = eq | actL | actU
    b = np.where(actU, rhi, rlo)            # eq rows: rlo == rhi
    AaF = A[np.ix_(act, F)]
    rhs = b[act] - A[np.ix_(act, ~F)] @ xN[~F]
    xF0 = d["xp"][s][F]
    dxF = np.linalg.lstsq(AaF, rhs - AaF @ xF0, rcond=1e-12)[0] if AaF.size else np.zeros(F.sum())
    x = xN.copy(); x[F] = xF0 + dxF
    ya0 = d["yp"][s][act]
    dy = np.linalg.lstsq(AaF.T, c[F] - AaF.T @ ya0, rcond=1e-12)[0] if AaF.size else np.zeros(act.sum())
    y = np.zeros(len(gy)); y[act] = ya0 + dy
    # checks (scaled space)
    xs = 1.0 + np.abs(x).max(); ys = 1.0 + np.abs(y).max(); cs = 1.0 + np.abs(c).max()
    ax = A @ x
    bs = 1.0 + np.abs(fin(rlo)).max() + np.abs(fin(rhi)).max()
    ok_p = (np.all(x >= lb - tol * xs) and np.all(x <= ub + tol * xs) and np.all(ax >= rlo - tol * bs) and np.all(ax <= rhi + tol * bs))
    rc = c - A.T @ y
    ok_d = (np.all(np.abs(rc[F]) <= tol * cs) and np.all(rc[atL & ~(lb == ub)] >= -tol * cs) and np.all(rc[atU & ~(lb == ub)] <= tol * cs)
            and np.all(y[actL] >= -tol * ys) and np.all(y[actU] <= tol * ys))
    return ok_p and ok_d, x, y, (int(F.sum()), int(act.sum()))


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "wind_battery_24h"
    if len(sys.argv) > 2:
        os.environ["NRAND"] = sys.argv[2]
    ids, sub = lab.subset(wl, nrand=int(os.environ.get("NRAND", 27)))
    B = len(ids)
    ref = np.array([pp.highs_obj(sub, i)[0] for i in range(B)])
    first = np.zeros(B, int); err = np.full(B, np.nan); shape = [None] * B
    every = int(os.environ.get("POLISH_EVERY", 1)); cnt = [0]

    stable_need = int(os.environ.get("POLISH_STABLE", 0))
    sig_prev = [None] * B; stable = np.zeros(B, int); attempts = np.zeros(B, int)

    def signature(d, s):
        gx, gy, sg = d["gx"][s], d["gy"][s], d["sig"][s, 0]
        return (gx <= d["lb"][s]).tobytes() + (gx >= d["ub"][s]).tobytes() + (-gy < sg * d["rlo"][s]).tobytes() + (-gy > sg * d["rhi"][s]).tobytes()

    def hook(it, d):
        cnt[0] += 1
        if cnt[0] % every:
            return
        for s in range(B):
            if first[s] or d["done"][s]:
                continue
            sg = signature(d, s)
            stable[s] = stable[s] + 1 if sg == sig_prev[s] else 0
            sig_prev[s] = sg
            if stable[s] < stable_need or (stable_need and stable[s] % stable_need):
                continue
            attempts[s] += 1
            ok, x, y, sh = polish_one(d, s)
            if ok:
                first[s] = it
                obj = sub.c[s] @ (x * d["dc"]) + sub.c0[s]
                err[s] = abs(obj - ref[s]) / max(1.0, abs(ref[s]))
                shape[s] = sh
    lab.solve.hook = hook
    t = time.time()
    X, Y, iters, nrs, done = lab.solve(sub, **lab.GPU_DEFAULTS)
    first_or_end = np.where(first > 0, first, iters)
    print("attempts mean", attempts.mean(), "max", attempts.max())
    print(f"{wl}: {B} scenarios  PDHG done {done.sum()}  mean {iters.mean():.0f} max {iters.max()}  |  polish success {int((first>0).sum())}"
          f"  mean {first_or_end.mean():.0f} max {first_or_end.max()}  max obj err {np.nanmax(err):.2e}   ({time.time()-t:.0f} s)")
    for s in np.argsort(-iters)[:12]:
        print(f"  scenario {ids[s]:5d}: PDHG {iters[s]:6d}  polish at {first[s]:6d}  err {err[s]:.1e}  (free, active) {shape[s]}")
