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


class HighsTensorLP:
    """TEST-ONLY stand-in for dispatches_amd.hip_solver.DeviceLP: same `solve` signature, CPU torch tensors in and out,
    HiGHS per scenario.  Lets the device-resident batched double loop (dispatches_amd/rolling.py) run its window / objective /
    state-hand-off logic in the CPU test tier."""

    def __init__(self, lp):
        self.lp = lp

    def solve(self, B, c, lb=None, ub=None, rlo=None, rhi=None, x0=None, y0=None, options=None, out=None, sync_stats=True,
              obj_offset=None, primal_weight=None):
        import torch
        from oracle.highs_direct import HighsModel
        lp = self.lp
        A = lp.csr()
        pick = lambda t, i: (t[i] if t.dim() == 2 else t).numpy()
        X, Y = np.zeros((B, lp.n)), np.zeros((B, lp.m))
        obj, st = np.zeros(B), np.zeros(B, np.int32)
        for i in range(B):
            M = HighsModel(pick(c, i), A, pick(rlo, i), pick(rhi, i), pick(lb, i), pick(ub, i))
            x, f, y = M.solve()
            X[i], Y[i], obj[i] = x, y, f
        return dict(x=torch.as_tensor(X), y=torch.as_tensor(Y), obj=torch.as_tensor(obj), status=torch.as_tensor(st),
                    iters=torch.zeros(B, dtype=torch.int32), jumps=torch.zeros(B, dtype=torch.int32))
