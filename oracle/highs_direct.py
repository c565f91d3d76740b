"""CPU ORACLE helper (test infrastructure — NOT part of the product path): direct access to the HiGHS library that
scipy vendors (``scipy.optimize._highspy``), for what ``scipy.optimize.linprog`` cannot do:

  * hot-started re-solves after an objective change -> the per-hour [min, max] range of a setpoint over the OPTIMAL
    FACE of a degenerate dispatch LP (96-192 LPs per scenario; milliseconds instead of a cold solve each);
  * convex QPs (``passHessian``) -> oracle of the quadratic ramp-cost variant (BASELINE config 5).

Same solver as ``dispatch_lp_oracle.PreparedLP.solve`` (HiGHS dual simplex); only the calling convention differs.
Only tests/, tools/make_oracle_fixtures.py, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.optimize._highspy import _core as _hc

_INF = _hc.kHighsInf


def _f(a):
    a = np.array(a, dtype=np.float64, copy=True)
    a[a == np.inf] = _INF
    a[a == -np.inf] = -_INF
    return a


class HighsModel:
    """min c.x + c0 (+ 1/2 x'Qx)  s.t.  lo <= A x <= hi,  lb <= x <= ub   on a persistent HiGHS instance."""

    def __init__(self, c, A, lo, hi, lb, ub, c0=0.0, Q=None, tol=1e-9):
        A = sp.csc_matrix(A)
        self.n, self.m = A.shape[1], A.shape[0]
        self.c = np.asarray(c, float).copy()
        self.c0 = float(c0)
        self.tol = float(tol)
        lp = _hc.HighsLp()
        lp.num_col_, lp.num_row_ = self.n, self.m
        lp.col_cost_ = self.c
        lp.col_lower_, lp.col_upper_ = _f(lb), _f(ub)
        lp.row_lower_, lp.row_upper_ = _f(lo), _f(hi)
        lp.a_matrix_.format_ = _hc.MatrixFormat.kColwise
        lp.a_matrix_.start_ = A.indptr.astype(np.int32)
        lp.a_matrix_.index_ = A.indices.astype(np.int32)
        lp.a_matrix_.value_ = A.data.astype(np.float64)
        h = _hc._Highs()
        h.setOptionValue("output_flag", False)
        h.setOptionValue("primal_feasibility_tolerance", tol)
        h.setOptionValue("dual_feasibility_tolerance", tol)
        assert h.passModel(lp) == _hc.HighsStatus.kOk
        if Q is not None:
            Ql = sp.csc_matrix(sp.tril(sp.csc_matrix(Q)))        # HiGHS wants the lower triangle, column-wise
            hs = _hc.HighsHessian()
            hs.dim_ = self.n
            hs.format_ = _hc.HessianFormat.kTriangular
            hs.start_ = Ql.indptr.astype(np.int32)
            hs.index_ = Ql.indices.astype(np.int32)
            hs.value_ = Ql.data.astype(np.float64)
            assert h.passHessian(hs) == _hc.HighsStatus.kOk
        self.h = h
        self._cap_row = None

    def _run(self):
        """Run HiGHS; a hot start from the previous basis occasionally ends "Unknown" (or falsely "Infeasible") at 1e-9
        feasibility tolerances, so fall back step by step: from scratch without presolve, primal simplex, a decade
        looser tolerances, and finally the interior-point solver with crossover (which has not failed yet: the dense
        objective-cap row of face_ranges is what upsets the simplex codes)."""
        h = self.h
        ok = _hc.HighsModelStatus.kOptimal
        h.run()
        st = h.getModelStatus()
        if st != ok:
            attempts = [dict(presolve="off"), dict(presolve="off", simplex_strategy=4), dict(simplex_strategy=4),
                        dict(presolve="off", tol=10 * self.tol), dict(solver="ipm"), dict(solver="ipm", tol=10 * self.tol),
                        dict(solver="ipm", presolve="off", tol=100 * self.tol)]
            for a in attempts:
                h.clearSolver()
                h.setOptionValue("solver", a.get("solver", "choose"))
                h.setOptionValue("presolve", a.get("presolve", "choose"))
                h.setOptionValue("simplex_strategy", a.get("simplex_strategy", 1))
                h.setOptionValue("primal_feasibility_tolerance", a.get("tol", self.tol))
                h.setOptionValue("dual_feasibility_tolerance", a.get("tol", self.tol))
                h.run()
                st = h.getModelStatus()
                if st == ok:
                    break
            h.setOptionValue("solver", "choose")
            h.setOptionValue("presolve", "choose")
            h.setOptionValue("simplex_strategy", 1)
            h.setOptionValue("primal_feasibility_tolerance", self.tol)
            h.setOptionValue("dual_feasibility_tolerance", self.tol)
        if st != ok:
            raise RuntimeError(f"HiGHS: {h.modelStatusToString(st)}")
        sol = h.getSolution()
        return np.array(sol.col_value), h.getInfo().objective_function_value

    def solve(self):
        """-> (x, objective incl. c0, row duals)"""
        x, f = self._run()
        self._x, self._f = x, f
        return x, f + self.c0, np.array(self.h.getSolution().row_dual)

    def face_ranges(self, exprs, slack=None):
        """[min, max] of each linear expression (dict col -> coef, const) over the optimal face {x feasible,
        c.x <= f* + slack}.  Call after solve().  Hot-started from the optimal basis: a few pivots per LP.
        Default slack = the rounding level of the cap row itself, 1e-12 sum_j |c_j x_j| + 1e-9 (c.x is a sum of terms up
        to 1e5 times larger than the objective; an exact cap is reported infeasible by HiGHS).  The dense cap row makes
        a few of these LPs numerically nasty; an expression whose LP fails under every fallback of _run is retried with
        the cap loosened 10x at a time (its range then errs on the wide side, never on the narrow one)."""
        h = self.h
        if slack is None:
            slack = 1e-12 * float(np.abs(self.c * self._x).sum()) + 1e-9
        nz = np.nonzero(self.c)[0].astype(np.int32)
        zero = np.zeros(self.n)
        idx = np.arange(self.n, dtype=np.int32)
        cap_row = np.array([self.m], dtype=np.int32)
        out = np.empty((len(exprs), 2))

        def set_cap(sl):
            assert h.changeRowBounds(self.m, -_INF, self._f + sl) == _hc.HighsStatus.kOk

        def extreme(cost):
            for grow in (1.0, 10.0, 100.0, 1e3, 1e4):
                try:
                    if grow != 1.0:
                        set_cap(slack * grow)
                        h.clearSolver()
                    h.changeColsCost(self.n, idx, cost)
                    return self._run()[1]
                except RuntimeError:
                    if grow == 1e4:
                        raise
                finally:
                    if grow != 1.0:
                        set_cap(slack)
            raise AssertionError

        assert h.addRow(-_INF, self._f + slack, len(nz), nz, self.c[nz]) == _hc.HighsStatus.kOk
        try:
            for k, (d, const) in enumerate(exprs):
                cost = zero.copy()
                for j, v in d.items():
                    cost[j] = v
                out[k] = (extreme(cost) + const, -extreme(-cost) + const)
        finally:
            h.changeColsCost(self.n, idx, self.c)
            h.deleteRows(1, cap_row)
        return out


def from_prepared(P, c=None, Q=None, tol=1e-9):
    """HighsModel of a dispatch_lp_oracle.PreparedLP (optionally with another cost vector / a Hessian)."""
    return HighsModel(P.c if c is None else c, P.A, P.lo, P.hi, P.lb, P.ub, c0=P.c0, Q=Q, tol=tol)
