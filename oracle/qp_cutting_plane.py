"""CPU ORACLE (test infrastructure — NOT part of the product path) for the convex QPs of BASELINE config 5:

        min  c.x + c0 + (rho / 2) |M x|^2      s.t.  lo <= A x <= hi,  lb <= x <= ub

(`M x` = the T - 1 hour-to-hour differences of the delivered power: dispatch_lp_oracle.wind_battery_da_qp; OUR
extension - the reference has no quadratic term, so nothing here can be pinned against a reference vector).

Round 2 found no QP SOLVER in this container that reaches the 1e-6 parity bar on these degenerate problems (HiGHS'
active-set QP cycles, a textbook interior-point method stalls at 1e-5, trust-constr never converged on the un-reduced
formulation).  This oracle therefore does not use a QP solver at all.  It is Kelley's cutting-plane method on top of
the LP oracle that G1-G7 pin (HiGHS dual simplex on the UN-reduced LP):

    epigraph columns s_t >= (rho / 2) r_t^2,  r = M x;  a convex parabola lies above every tangent, so the rows
        s_t >= rho r_k (M x)_t - (rho / 2) r_k^2          (tangent at r_k)
    are valid for the QP and the LP with any finite set of them is a RELAXATION:  its optimum is a LOWER bound;
    the objective of the QP at the relaxation's x (which is feasible for the QP) is an UPPER bound.

Every round adds the tangents at the current r and re-solves (hot start); it stops when upper - lower <= gap_rel
(1 + |upper|).  What is returned is therefore a CERTIFIED bracket of the optimal value, not the output of a solver one
has to trust: the only trusted component is the LP solve.  Separable one-dimensional parabolas make Kelley fast here
(the model error at the kink between two tangents a, b is (rho / 2) ((b - a) / 2)^2: it falls 4x per round); 15-30
rounds reach 1e-9.

Only tests/, tools/make_oracle_fixtures.py, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import highs_direct as hd

_hc = hd._hc


def solve_qp_bracket(P, M, rho, gap_rel=1e-9, max_rounds=200, tol=1e-9):
    """P: dispatch_lp_oracle.PreparedLP (objective c.x + c0), M: sparse [R, n], rho >= 0.
    -> dict(x, upper, lower, rounds, cuts): lower <= optimal value <= upper = objective at the feasible point x."""
    M = sp.csr_matrix(M)
    R, n = M.shape
    A = sp.csr_matrix(P.A)
    m = A.shape[0]
    # columns [x, s]; the first cut (tangent at r = 0) is the column bound s >= 0
    A_ext = sp.hstack([A, sp.csr_matrix((m, R))]).tocsc()
    c_ext = np.concatenate([P.c, np.ones(R)])
    lb = np.concatenate([P.lb, np.zeros(R)])
    ub = np.concatenate([P.ub, np.full(R, np.inf)])
    H = hd.HighsModel(c_ext, A_ext, P.lo, P.hi, lb, ub, c0=P.c0, tol=tol)
    best_up, best_x = np.inf, None
    lower = -np.inf
    cuts = 0
    for rnd in range(1, max_rounds + 1):
        z, f, _ = H.solve()
        x, s = z[:n], z[n:]
        r = M @ x
        lower = max(lower, f)                                   # optimum of a relaxation (f includes c0)
        upper = float(P.c @ x + P.c0 + 0.5 * rho * (r @ r))      # x satisfies every row and bound of the QP
        if upper < best_up:
            best_up, best_x = upper, x.copy()
        if best_up - lower <= gap_rel * (1.0 + abs(best_up)):
            break
        # tangents at the current r for every parabola the model under-estimates
        # (a parabola whose under-estimate is below 1 / (10 R) of the allowed gap cannot matter for the bracket; and
        # HiGHS drops matrix values below 1e-9, which tangents at r ~ 0 would have)
        under = 0.5 * rho * r * r - s
        add = np.nonzero(under > max(1e-13, 0.1 * gap_rel * (1.0 + abs(best_up)) / R))[0]
        if len(add) == 0:
            break
        for t in add:
            row = M.getrow(t)
            idx = np.concatenate([row.indices, [n + t]]).astype(np.int32)
            val = np.concatenate([rho * r[t] * row.data, [-1.0]])
            #  rho r_k (M x)_t - s_t <= (rho / 2) r_k^2
            assert H.h.addRow(-hd._INF, 0.5 * rho * r[t] * r[t], len(idx), idx, val) != _hc.HighsStatus.kError
            cuts += 1
    return dict(x=best_x, upper=best_up, lower=lower, rounds=rnd, cuts=cuts)


def ramp_matrix(lp, fs):
    """M with (M x)_t = P_T[t+1] - P_T[t] in the oracle's ORIGINAL columns (P_T[t] is a linear expression)."""
    n = len(lp.names)
    T = len(fs["P_T"])
    E = sp.lil_matrix((T, n))
    for t, (d, _k) in enumerate(fs["P_T"]):
        for j, v in d.items():
            E[t, j] = v
    D = sp.diags([-np.ones(T - 1), np.ones(T - 1)], [0, 1], shape=(T - 1, T))
    return sp.csr_matrix(D @ E.tocsr())


def wind_battery_da_qp(T, cf, da, rt, rho, gap_rel=1e-9, **kw):
    """Day-ahead bidding problem of LP #1 + A.4 with the ramp cost (rho / 2) sum_t (P_T[t] - P_T[t-1])^2.
    -> (bracket dict, PreparedLP, flowsheet dict, day-ahead columns)."""
    from . import dispatch_lp_oracle as orc
    P, _Q, fs, pda = orc.wind_battery_da_qp(T, cf, da, rt, rho, **kw)
    out = solve_qp_bracket(P, ramp_matrix(P.lp, fs), rho, gap_rel=gap_rel)
    out["P_T"] = np.array([P.value(fs["P_T"][t], out["x"]) for t in range(T)])
    return out, P, fs, pda


def wind_battery_da_coupled_qp(T, cf, da, rt, mode, rho, gap_rel=1e-9, **kw):
    """The COUPLED day-ahead problem of the stochastic bidders (dispatch_lp_oracle.wind_battery_da_coupled: S scenario copies +
    non-anticipativity / monotone-bid rows) with the ramp cost (rho / 2) sum_t (P_T[s, t] - P_T[s, t-1])^2 on every copy.
    -> (bracket dict, PreparedLP, [day-ahead columns per scenario])."""
    from . import dispatch_lp_oracle as orc
    P, pdas, flowsheets = orc.wind_battery_da_coupled(T, cf, da, rt, mode, return_flowsheets=True, **kw)
    M = sp.vstack([ramp_matrix(P.lp, fs) for fs in flowsheets]).tocsr()
    out = solve_qp_bracket(P, M, rho, gap_rel=gap_rel)
    out["P_T"] = np.array([[P.value(fs["P_T"][t], out["x"]) for t in range(T)] for fs in flowsheets])
    return out, P, pdas
