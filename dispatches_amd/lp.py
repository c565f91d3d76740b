"""Flatten-once linear modelling layer.

The reference builds every scenario's multi-period flowsheet as a tree of Pyomo blocks and re-writes an LP/NL
file on every ``solver.solve``.  Here a model object populates ONE :class:`LinearBlock`
(named columns, rows, and named linear expressions such as ``P_T[t]`` / ``tot_cost[t]``); the block is
flattened ONCE to a :class:`StandardFormLP` (shared CSR constraint matrix + template vectors) and the only
things that ever change afterwards are dense per-scenario vectors (objective, column bounds, row bounds) that
are handed to the HIP solver as ``[B, n]`` / ``[B, m]`` arrays.

Standard form:   min c.x + c0   s.t.  rlo <= A x <= rhi,   lb <= x <= ub.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import math

import numpy as np

INF = float("inf")


class LinExpr:
    """Sparse affine expression  sum_j coef_j x_j + const  over the columns of one LinearBlock."""

    __slots__ = ("coef", "const")

    def __init__(self, coef: Optional[Dict[int, float]] = None, const: float = 0.0):
        self.coef = coef if coef is not None else {}
        self.const = float(const)

    @staticmethod
    def _as(x) -> "LinExpr":
        if isinstance(x, LinExpr):
            return x
        if isinstance(x, Var):
            return LinExpr({x.index: 1.0})
        return LinExpr(None, float(x))

    def copy(self):
        return LinExpr(dict(self.coef), self.const)

    def __add__(self, other):
        o = LinExpr._as(other)
        r = self.copy()
        for j, v in o.coef.items():
            r.coef[j] = r.coef.get(j, 0.0) + v
        r.const += o.const
        return r

    __radd__ = __add__

    def __neg__(self):
        return self * -1.0

    def __sub__(self, other):
        return self + (LinExpr._as(other) * -1.0)

    def __rsub__(self, other):
        return (self * -1.0) + other

    def __mul__(self, s):
        s = float(s)
        return LinExpr({j: v * s for j, v in self.coef.items()}, self.const * s)

    __rmul__ = __mul__

    def __truediv__(self, s):
        return self * (1.0 / float(s))

    def accumulate(self, other, scale: float = 1.0) -> "LinExpr":
        """self += scale * other IN PLACE (returns self).  For sums over thousands of periods: `total = total + term` copies the
        growing dictionary every time - quadratic in the horizon (the year-long price-taker objectives took minutes)."""
        o = LinExpr._as(other)
        s = float(scale)
        for j, v in o.coef.items():
            self.coef[j] = self.coef.get(j, 0.0) + s * v
        self.const += s * o.const
        return self

    def value(self, x: np.ndarray) -> float:
        return self.const + sum(v * x[j] for j, v in self.coef.items())

    def dense(self, n: int) -> np.ndarray:
        out = np.zeros(n)
        for j, v in self.coef.items():
            out[j] = v
        return out


class Var:
    """Handle of one column of a LinearBlock."""

    __slots__ = ("block", "index", "name")

    def __init__(self, block, index, name):
        self.block, self.index, self.name = block, index, name

    # arithmetic promotes to LinExpr
    def __add__(self, o):
        return LinExpr._as(self) + o

    __radd__ = __add__

    def __sub__(self, o):
        return LinExpr._as(self) - o

    def __rsub__(self, o):
        return LinExpr._as(o) - self

    def __mul__(self, s):
        return LinExpr._as(self) * s

    __rmul__ = __mul__

    def __truediv__(self, s):
        return LinExpr._as(self) / s

    def __neg__(self):
        return LinExpr._as(self) * -1.0

    # pyomo-flavoured conveniences used by the model objects
    @property
    def lb(self):
        return self.block.col_lb[self.index]

    @property
    def ub(self):
        return self.block.col_ub[self.index]

    def setlb(self, v):
        self.block.set_bounds(self, lb=v)

    def setub(self, v):
        self.block.set_bounds(self, ub=v)

    def fix(self, value):
        self.block.set_bounds(self, lb=value, ub=value)

    @property
    def value(self):
        x = self.block.solution
        return None if x is None else float(x[self.index])


@dataclass
class StandardFormLP:
    """One scenario's LP in standard form; `indptr/indices/data` is the CSR of A shared by the whole batch."""

    n: int
    m: int
    indptr: np.ndarray      # int32 [m+1]
    indices: np.ndarray     # int32 [nnz]
    data: np.ndarray        # float64 [nnz]
    c: np.ndarray           # [n]
    c0: float
    lb: np.ndarray          # [n]  (-inf allowed)
    ub: np.ndarray          # [n]  (+inf allowed)
    rlo: np.ndarray         # [m]  (-inf allowed)
    rhi: np.ndarray         # [m]  (+inf allowed)
    col_names: List[str] = field(default_factory=list)
    row_names: List[str] = field(default_factory=list)
    # [m] compliance kappa_i >= 0 of every row (None = LP).  A row with kappa_i > 0 is SOFT: not a constraint but the
    # objective term (a_i.x - b_i)^2 / (2 kappa_i), b_i = its (equal) row bounds - the factored form Q = sum_i a_i a_i^T /
    # kappa_i of a convex quadratic objective, which first-order solvers take in the DUAL (LinearBlock.quadratic)
    row_compliance: Optional[np.ndarray] = None
    # variable scaling factors: the typical magnitude of each column (x_j = s_j x~_j, x~ of order one), handed to the solver as
    # dsp_lp_desc::col_scale - what an IDAES model carries as iscale.set_scaling_factor (there 1 / s).  None = none.
    col_scale: Optional[np.ndarray] = None

    @property
    def nnz(self) -> int:
        return int(self.indptr[-1])

    def csr(self):
        import scipy.sparse as sp

        return sp.csr_matrix((self.data, self.indices, self.indptr), shape=(self.m, self.n))

    def objective(self, x: np.ndarray, c: Optional[np.ndarray] = None, c0: Optional[float] = None):
        c = self.c if c is None else c
        c0 = self.c0 if c0 is None else c0
        return float(c @ x + c0) + self.quadratic_value(x)

    def quadratic_value(self, x: np.ndarray, rhs: Optional[np.ndarray] = None) -> float:
        """sum over the soft rows of (a_i.x - b_i)^2 / (2 kappa_i)"""
        if self.row_compliance is None:
            return 0.0
        soft = self.row_compliance > 0
        r = (self.csr() @ x - (self.rhi if rhs is None else rhs))[soft]
        return float(0.5 * np.sum(r * r / self.row_compliance[soft]))

    def max_violation(self, x, lb=None, ub=None, rlo=None, rhi=None):
        lb = self.lb if lb is None else lb
        ub = self.ub if ub is None else ub
        rlo = self.rlo if rlo is None else rlo
        rhi = self.rhi if rhi is None else rhi
        ax = self.csr() @ x
        v = max(np.max(np.maximum(lb - x, 0)), np.max(np.maximum(x - ub, 0)))
        if self.m:
            hard = 1.0 if self.row_compliance is None else (self.row_compliance == 0)       # soft rows constrain nothing
            v = max(v, np.max(np.maximum(rlo - ax, 0) * hard), np.max(np.maximum(ax - rhi, 0) * hard))
        return float(v)


def implied_column_ranges(lp: "StandardFormLP", lb=None, ub=None, passes: int = 4) -> np.ndarray:
    """Typical magnitude of every column from what the model itself says: ub - lb where both are finite (per-scenario bound
    arrays: the widest over the scenarios), and for an unbounded column j the largest |a_ik| range_k / |a_ij| over the rows i it
    shares with a ranged column k - the most it can be asked to balance (`passes` Gauss-Seidel sweeps in column order so that
    chains such as splitter outlet -> day-ahead power resolve).  Columns nothing can be said about, and fixed ones, get 1.

    The flowsheets of the reference mix kW (1e5), MW (1e2) and kWh of accumulated throughput (1e8) in one LP; Ruiz / Pock-Chambolle
    equilibration balances the MATRIX and is blind to these ranges.  Started from them the wind + battery bidding LPs need 20-30 %
    fewer PDHG iterations, the wind + PEM ones 60 % fewer (tools/pdlp_lab.py `colscale`)."""
    lb = lp.lb if lb is None else np.asarray(lb, float)
    ub = lp.ub if ub is None else np.asarray(ub, float)
    lo = lb.min(axis=0) if lb.ndim == 2 else lb
    hi = ub.max(axis=0) if ub.ndim == 2 else ub
    with np.errstate(invalid="ignore"):
        rng = hi - lo
    fill = np.where(np.isfinite(rng), rng, np.nan)
    # a long chain of derived columns (a storage recursion whose power bound is itself a free design variable: the price-taker
    # LP) would multiply its way up; nothing derived may exceed 1e6 x the widest range the model states
    stated = np.where(np.isfinite(rng) & (rng > 0), rng, 0.0)
    cap = 1e6 * stated.max() if stated.any() else np.inf
    A = lp.csr()
    Ac = A.tocsc()
    absA = abs(A).tocsr()
    for _ in range(int(passes)):
        for j in np.nonzero(~np.isfinite(fill))[0]:
            best = 0.0
            for p in range(Ac.indptr[j], Ac.indptr[j + 1]):
                i, aij = Ac.indices[p], abs(Ac.data[p])
                if aij == 0.0:
                    continue
                ks = absA.indices[absA.indptr[i]:absA.indptr[i + 1]]
                vs = absA.data[absA.indptr[i]:absA.indptr[i + 1]]
                ok = (ks != j) & np.isfinite(fill[ks]) & (fill[ks] > 0)
                if ok.any():
                    best = max(best, float((vs[ok] * fill[ks[ok]]).max() / aij))
            if best > 0.0:
                fill[j] = min(best, cap)
    return np.where(np.isfinite(fill) & (fill > 0), fill, 1.0)


class LinearBlock:
    """The block a model object's ``populate_model(b, horizon)`` fills (stand-in for a Pyomo Block).

    Columns, rows and named expression families are appended in call order.  ``flatten()`` produces the
    StandardFormLP after a light presolve (drops rows that bound propagation proves can never bind, e.g. the
    1e8 battery ramp rows of wind_battery_LMP.py:139-142).  Bounds / row bounds stay mutable afterwards and
    are re-read through ``current_bounds()`` at every solve.
    """

    presolve_sequential_limit = 256      # candidates confirmed one at a time up to this many, in one batch beyond (flatten)

    def __init__(self, name: str = "fs"):
        self.name = name
        self.col_names: List[str] = []
        self.col_lb: List[float] = []
        self.col_ub: List[float] = []
        self.col_mutable: List[bool] = []
        self.col_hull: List[tuple] = []       # widest bounds a mutable column may ever take
        self.row_names: List[str] = []
        self.row_expr: List[Dict[int, float]] = []
        self.row_lo: List[float] = []
        self.row_hi: List[float] = []
        self.row_mutable: List[bool] = []
        self.expressions: Dict[str, Dict[int, LinExpr]] = {}
        self.row_soft: Dict[int, float] = {}        # row -> compliance kappa = 1 / weight of an objective term (w / 2) body^2
        self.solution: Optional[np.ndarray] = None
        self._constructed = False
        self._kept_rows: Optional[np.ndarray] = None
        self._version = 0            # bumped on every bound change (solver re-uploads the template)

    # -- pyomo-protocol shims ------------------------------------------------------------------------------
    def is_constructed(self):
        return self._constructed

    def construct(self):
        self._constructed = True

    # -- building ------------------------------------------------------------------------------------------
    def var(self, name: str, lb: float = 0.0, ub: float = INF, mutable: bool = False, hull=None) -> Var:
        """New column.  `mutable` columns may have their bounds changed after flatten(); `hull` is the widest
        interval they will ever be given (presolve only trusts the hull, never the current value)."""
        self.col_names.append(name)
        self.col_lb.append(float(lb))
        self.col_ub.append(float(ub))
        self.col_mutable.append(bool(mutable))
        self.col_hull.append((float(hull[0]), float(hull[1])) if hull is not None else
                             ((-INF, INF) if mutable else (float(lb), float(ub))))
        return Var(self, len(self.col_names) - 1, name)

    def constraint(self, name: str, body, lo: float = -INF, hi: float = INF, mutable: bool = False):
        """lo <= body <= hi ; the constant of `body` is moved to the bounds.  Rows whose bounds will be rewritten
        after flatten() (tracker dispatch rows) must be declared `mutable` so presolve never removes them."""
        e = LinExpr._as(body)
        self.row_names.append(name)
        self.row_mutable.append(bool(mutable))
        self.row_expr.append({j: v for j, v in e.coef.items() if v != 0.0})
        self.row_lo.append(lo - e.const if np.isfinite(lo) else -INF)
        self.row_hi.append(hi - e.const if np.isfinite(hi) else INF)
        return len(self.row_names) - 1

    def equality(self, name, body, rhs=0.0):
        return self.constraint(name, body, rhs, rhs)

    def quadratic(self, name: str, expr, weight: float) -> int:
        """Objective term (weight / 2) * expr^2 as a SOFT ROW: the row expr = 0 with compliance kappa = 1 / weight.
        By convex duality (weight / 2) r^2 = max_y [-y r - y^2 / (2 weight)], so the term is an equality row whose
        multiplier pays kappa y^2 / 2; a primal-dual first-order method then only changes the dual step of that row to
        y+ = (y - sigma a.xbar) / (1 + sigma kappa) - no extra column, nothing else changes (include/dsp_hip.h,
        dsp_batch::row_compliance).  Any convex quadratic objective x'Qx / 2 can be handed over this way through a
        factorisation Q = sum_i w_i a_i a_i'.  (Round 2 first LIFTED the term to a diagonal Q on extra free columns with
        the primal proximal step; that form needed 10-20x the iterations of the LP - tools/pdqp_proto.py.)
        Returns the row index."""
        if not weight > 0:
            raise ValueError("quadratic terms must be strictly convex in their expression (weight > 0)")
        row = self.constraint(name, expr, 0.0, 0.0, mutable=True)      # mutable: never presolved away, never propagated
        self.row_soft[row] = 1.0 / float(weight)
        return row

    def set_row_bounds(self, row: int, lo: float, hi: float):
        if self._kept_rows is not None and not self.row_mutable[row]:
            raise ValueError(f"row {self.row_names[row]} was not declared mutable before flatten()")
        self.row_lo[row], self.row_hi[row] = float(lo), float(hi)
        self._version += 1

    def set_bounds(self, v: Var, lb=None, ub=None):
        j = v.index
        if self._kept_rows is not None and not self.col_mutable[j]:
            raise ValueError(f"column {v.name} was not declared mutable before flatten()")
        if lb is not None:
            self.col_lb[j] = float(lb)
        if ub is not None:
            self.col_ub[j] = float(ub)
        if self._kept_rows is None and not self.col_mutable[j]:
            self.col_hull[j] = (self.col_lb[j], self.col_ub[j])
        h = self.col_hull[j]
        if self.col_lb[j] < h[0] - 1e-9 * max(1, abs(h[0])) or self.col_ub[j] > h[1] + 1e-9 * max(1, abs(h[1])):
            raise ValueError(f"bounds of {v.name} leave the declared hull {h}")
        self._version += 1

    def expression(self, family: str, index: int, expr):
        self.expressions.setdefault(family, {})[index] = LinExpr._as(expr)
        self.__dict__.setdefault("_family_cache", {}).pop(family, None)

    def family_values(self, family: str) -> np.ndarray:
        """Values of every member of an expression family (`P_T`, `tot_cost`, ...) at the current solution, index order: the same
        products and sums as `value(b.P_T[t])` member by member (constant + the terms in their stored order), as ONE gather over
        cached index / coefficient arrays - record_results of a 24-h block made 49 Python-level evaluations per scenario."""
        cache = self.__dict__.setdefault("_family_cache", {})
        hit = cache.get(family)
        if hit is None:
            fam = self.expressions[family]
            keys = sorted(fam)
            K = max(1, max(len(fam[t].coef) for t in keys))
            cols = np.zeros((len(keys), K), dtype=np.intp)
            vals = np.zeros((len(keys), K))
            for r, t in enumerate(keys):
                for e, (j, v) in enumerate(fam[t].coef.items()):
                    cols[r, e], vals[r, e] = j, v
            hit = cache[family] = (cols, vals, np.array([fam[t].const for t in keys]), K)
        cols, vals, const, K = hit
        terms = vals * np.asarray(self.solution)[..., cols]     # (solution [n], or [S, n]: S scenarios at once -> [S, members])
        acc = terms[..., 0].copy()
        for e in range(1, K):                      # left to right, like the Python sum of LinExpr.value
            acc += terms[..., e]
        return const + acc

    def __getattr__(self, item):
        # `b.P_T[t]`, `b.tot_cost[t]` like the Pyomo Expression families of the reference
        ex = self.__dict__.get("expressions", {})
        if item in ex:
            return ex[item]
        raise AttributeError(item)

    def value(self, expr) -> float:
        return LinExpr._as(expr).value(self.solution)

    # -- flattening ----------------------------------------------------------------------------------------
    def _propagate_bounds(self, active):
        """Implied column bounds from the rows flagged in `active` (bound propagation), starting from immutable bounds /
        declared hulls only.  Mutable columns are never tightened."""
        # (plain Python floats and lists: the same arithmetic in the same order as the first version on numpy scalars, several times
        #  faster - this pass was 80 % of the 6 s a year-long price-taker LP took to flatten)
        isfinite = math.isfinite
        lb = [float(h[0]) for h in self.col_hull]
        ub = [float(h[1]) for h in self.col_hull]
        mutable = self.col_mutable
        rows = [([(j, float(a)) for j, a in self.row_expr[i].items()], float(self.row_lo[i]), float(self.row_hi[i]))
                for i in range(len(self.row_expr)) if active[i] and not self.row_mutable[i]]
        # A row whose columns' bounds have not changed since it was last looked at would derive the same bounds again: skipped (the
        # later sweeps of a multi-period LP skip the rows no design column reaches; same result, bit for bit - year-long LP: 2.35 -> 2.0 s)
        clock = 1
        changed_at = [1] * len(lb)                   # value of `clock` when the column's bounds last changed
        seen_at = [0] * len(rows)                    # ... when the row was last evaluated
        for _ in range(3):
            for r, (items, lo, hi) in enumerate(rows):
                seen = seen_at[r]
                for j, _a in items:
                    if changed_at[j] > seen:
                        break
                else:
                    continue
                seen_at[r] = clock
                mins = [(a * lb[j] if a > 0 else a * ub[j]) for j, a in items]
                maxs = [(a * ub[j] if a > 0 else a * lb[j]) for j, a in items]
                smin, smax = sum(mins), sum(maxs)
                hi_ok, lo_ok = isfinite(hi), isfinite(lo)
                for k, (j, a) in enumerate(items):
                    if mutable[j]:
                        continue
                    if hi_ok:
                        rest = smin - mins[k] if isfinite(mins[k]) else sum(mins[:k]) + sum(mins[k + 1:])
                        if isfinite(rest):
                            b = (hi - rest) / a
                            if a > 0:
                                if b < ub[j]:
                                    ub[j] = b
                                    clock += 1; changed_at[j] = clock
                            elif b > lb[j]:
                                lb[j] = b
                                clock += 1; changed_at[j] = clock
                    if lo_ok:
                        rest = smax - maxs[k] if isfinite(maxs[k]) else sum(maxs[:k]) + sum(maxs[k + 1:])
                        if isfinite(rest):
                            b = (lo - rest) / a
                            if a > 0:
                                if b > lb[j]:
                                    lb[j] = b
                                    clock += 1; changed_at[j] = clock
                            elif b < ub[j]:
                                ub[j] = b
                                clock += 1; changed_at[j] = clock
        lb, ub = np.array(lb), np.array(ub)
        return lb, ub

    def _row_range(self, i, lb, ub):
        r = self.row_expr[i]
        amin = sum((a * lb[j] if a > 0 else a * ub[j]) for j, a in r.items())
        amax = sum((a * ub[j] if a > 0 else a * lb[j]) for j, a in r.items())
        return amin, amax

    def _never_binds(self, i, lb, ub):
        lo, hi = self.row_lo[i], self.row_hi[i]
        if self.row_mutable[i] or lo == hi:
            return False
        amin, amax = self._row_range(i, lb, ub)
        tol = 1e-9 * max(1.0, abs(lo) if np.isfinite(lo) else 0.0, abs(hi) if np.isfinite(hi) else 0.0)
        return amin >= lo - tol and amax <= hi + tol

    def flatten(self, objective: Optional[LinExpr] = None, presolve: bool = True) -> StandardFormLP:
        """Presolve drops a row only when the column bounds implied by the declared hulls and by the OTHER rows that
        stay in the LP prove it can never bind.  (A row must not certify its own redundancy: bounds propagated from row
        R are never written into the LP's column bounds, so testing R against them would drop e.g. the singleton row
        x <= 5 and leave x unbounded.)  Candidates are found with one propagation over all rows and then confirmed
        one at a time against the rows still kept, so two rows can never vouch for each other either."""
        n, m_all = len(self.col_names), len(self.row_names)
        keep = np.ones(m_all, bool)
        if presolve and m_all:
            lb, ub = self._propagate_bounds(keep)
            candidates = [i for i in range(m_all) if self._never_binds(i, lb, ub)]
            if len(candidates) <= self.presolve_sequential_limit:
                for i in candidates:
                    keep[i] = False
                    lb, ub = self._propagate_bounds(keep)
                    if not self._never_binds(i, lb, ub):
                        keep[i] = True
            else:
                # Many candidates (a year-long horizon has thousands of never-binding capacity / ramp rows): one propagation per
                # candidate is quadratic in the horizon (the 8784-period nuclear LP: hours).  All candidates are set aside AT
                # ONCE and each is confirmed against the bounds that the NON-candidate rows and the declared hulls imply - no row
                # vouches for itself or for another candidate, so the rule of the docstring holds a fortiori; a candidate whose
                # proof needed another candidate stays in the LP (harmless: it never binds).
                keep[candidates] = False
                lb, ub = self._propagate_bounds(keep)
                for i in candidates:
                    if not self._never_binds(i, lb, ub):
                        keep[i] = True
        self._kept_rows = np.nonzero(keep)[0]
        indptr, indices, data = [0], [], []
        for i in self._kept_rows:
            for j in sorted(self.row_expr[i]):
                indices.append(j)
                data.append(self.row_expr[i][j])
            indptr.append(len(indices))
        obj = LinExpr._as(objective) if objective is not None else LinExpr()
        lbv, ubv, rlo, rhi = self.current_bounds()
        return StandardFormLP(
            n=n, m=len(self._kept_rows),
            indptr=np.asarray(indptr, np.int32), indices=np.asarray(indices, np.int32),
            data=np.asarray(data, np.float64), c=obj.dense(n), c0=obj.const,
            lb=lbv, ub=ubv, rlo=rlo, rhi=rhi,
            col_names=list(self.col_names), row_names=[self.row_names[i] for i in self._kept_rows],
            row_compliance=(None if not self.row_soft else
                            np.array([self.row_soft.get(int(i), 0.0) for i in self._kept_rows], np.float64)),
        )

    def current_bounds(self):
        """Current (lb, ub, rlo, rhi) in flattened row order."""
        kr = self._kept_rows if self._kept_rows is not None else np.arange(len(self.row_names))
        return (np.asarray(self.col_lb, np.float64), np.asarray(self.col_ub, np.float64),
                np.asarray(self.row_lo, np.float64)[kr], np.asarray(self.row_hi, np.float64)[kr])

    def kept_row_index(self, row: int) -> int:
        """Position of original row `row` in the flattened LP (-1 if presolved away)."""
        pos = np.nonzero(self._kept_rows == row)[0]
        return int(pos[0]) if len(pos) else -1
