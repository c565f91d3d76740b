"""TEST-ONLY solver: drives the product's host logic (flattener, bidder, tracker) on CPU with scipy's HiGHS so
that `-m "not gpu"` tests can pin the host side against the reference's goldens.  It lives under tests/ and is
never importable from the `dispatches_amd` package: the product has no CPU solve path."""
import numpy as np
from scipy.optimize import linprog

from dispatches_amd.workflow.batch_model import SolveResults


class HighsTestSolver:
    def solve(self, model, tee=False):
        lp = model.lp
        A = lp.csr()
        lb, ub, rlo, rhi = model.scenario_bounds()
        B = model.n_scenario
        X = np.zeros((B, lp.n))
        Y = np.zeros((B, lp.m))
        obj = np.zeros(B)
        pick = lambda a, i: a[i] if a.ndim == 2 else a
        for i in range(B):
            l, u, lo, hi = pick(lb, i), pick(ub, i), pick(rlo, i), pick(rhi, i)
            eq = np.isfinite(lo) & (lo == hi)
            up = np.isfinite(hi) & ~eq
            dn = np.isfinite(lo) & ~eq
            Aub = [A[up], -A[dn]]
            bub = [hi[up], -lo[dn]]
            import scipy.sparse as sp
            res = linprog(model.c[i], A_ub=sp.vstack(Aub).tocsr() if (up.any() or dn.any()) else None,
                          b_ub=np.concatenate(bub) if (up.any() or dn.any()) else None,
                          A_eq=A[eq] if eq.any() else None, b_eq=hi[eq] if eq.any() else None,
                          bounds=np.stack([l, u], 1), method="highs")
            assert res.status == 0, res.message
            X[i] = res.x
            obj[i] = res.fun + model.c0[i]
        model.store_solution(X, Y, obj, np.zeros(B, np.int32))
        return SolveResults("ok", "optimal")
