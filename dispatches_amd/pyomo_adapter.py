"""Live-Pyomo flattener: turn a POPULATED Pyomo block (what the reference's model objects build in
``populate_model``: wind_battery_double_loop.py:136-179, nuclear_flowsheet_multiperiod_class.py:158-212, and what the
upstream Bidder / Tracker wrap around it) into the ``StandardFormLP`` + per-solve vectors the HIP solver takes - so that
ANY linear DISPATCHES flowsheet can use the backend without being restated on a ``LinearBlock``.

Pyomo is NOT installed in the build container, so this module never imports it at module level and is written against
the small part of Pyomo's public API it needs (duck-typed, exercised in tests/test_pyomo_adapter.py with stand-in objects
that mimic that API; it has NOT been run against a real Pyomo model):

    block.component_data_objects(ctype, active=True, descend_into=True)     Var / Constraint / Objective data objects
    var.lb, var.ub, var.fixed, var.value, var.name
    con.body, con.lower, con.upper, con.name          (lower / upper may be None; equality: lower == upper)
    obj.expr, obj.sense                               (sense: +1 minimise, -1 maximise)
    generate_standard_repn(expr, compute_values=True) -> .linear_vars, .linear_coefs, .constant, .is_linear()
    (objective only)  .quadratic_vars [(v1, v2), ...], .quadratic_coefs, .is_quadratic(), .nonlinear_expr

What it does (the reference's solvers do the same inside their LP writers + presolve):
  * fixed variables (``var.fixed``) are folded into row constants / the objective constant; they get no column;
  * a row whose body is a constant after that is checked for consistency and dropped;
  * ``refresh()`` re-reads everything that mutable Params / ``fix()`` calls can change between solves (objective
    coefficients, variable bounds, row bounds and the constants that fixed variables contribute) and verifies that the
    constraint MATRIX did not change - the flatten-once contract of the batched solver.  A changed matrix (e.g. a
    mutable Param that multiplies a variable inside a constraint) raises, so nothing is silently solved with stale data;
  * a convex QUADRATIC objective (BASELINE config 5: e.g. a ramp cost written as a sum of squares and expanded by Pyomo into
    x_i x_j products) is handed to the solver the way `LinearBlock.quadratic` does it: the form x'Hx / 2 over the free
    variables is factored as sum_k (w_k / 2) (a_k . x)^2 by an LDL' elimination in column order (sparse a_k for banded H:
    the tridiagonal ramp cost gives two entries per row), and each term becomes a SOFT ROW (row_lb = row_ub = 0,
    row_compliance = 1 / w_k: include/dsp_hip.h).  A form that is not positive semidefinite raises.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np

from .lp import StandardFormLP

INF = float("inf")


def factor_quadratic_form(quad: Dict, n: int, tol: float = 1e-12):
    """sum_{(i,j)} q_ij x_i x_j  (free columns i, j; any order, repeated pairs add up)  ->  [(cols, vals, w_k)] with
    sum_ij q_ij x_i x_j = sum_k (w_k / 2) (sum_t vals[t] x_cols[t])^2, w_k > 0.

    LDL' without pivoting on the support of the form, in column order: x'Hx / 2 with H_ii = 2 q_ii, H_ij = H_ji = q_ij;
    H = L D L' (L unit lower triangular) gives x'Hx = sum_k D_k (L[:, k] . x)^2.  A negative pivot, or a zero pivot with a
    nonzero column below it, means H is not positive semidefinite (a zero pivot with a zero column is a direction the form
    does not see: skipped)."""
    support = sorted({i for ij in quad for i in ij})
    pos = {j: k for k, j in enumerate(support)}
    sN = len(support)
    H = np.zeros((sN, sN))
    for (i, j), q in quad.items():
        if i == j:
            H[pos[i], pos[i]] += 2.0 * q
        else:
            H[pos[i], pos[j]] += q
            H[pos[j], pos[i]] += q
    scale = max(1.0, float(np.abs(H).max())) if sN else 1.0
    rows = []
    for k in range(sN):
        d = H[k, k]
        below = H[k + 1:, k]
        if d < -tol * scale or (d <= tol * scale and np.abs(below).max(initial=0.0) > 1e-9 * scale):
            raise ValueError("the quadratic objective is not convex (its matrix is not positive semidefinite): the HIP backend "
                             "takes convex quadratic objectives only")
        if d <= tol * scale:
            continue
        l = np.concatenate([[1.0], below / d])
        H[k + 1:, k + 1:] -= d * np.outer(l[1:], l[1:])
        nz = np.nonzero(np.abs(l) > 1e-14)[0]
        rows.append(([support[k + t] for t in nz], [float(l[t]) for t in nz], float(d)))
    return rows


class MatrixChanged(ValueError):
    """refresh() found a different constraint matrix than flatten() saw (a variable was fixed / unfixed, a mutable Param that
    multiplies a variable changed, the quadratic objective changed): the flattened LP must be rebuilt."""


def _pyomo():
    """Import the pieces of Pyomo this adapter uses (only when no stand-ins are injected)."""
    from pyomo.core.base.constraint import Constraint
    from pyomo.core.base.objective import Objective
    from pyomo.core.base.var import Var
    from pyomo.repn import generate_standard_repn
    return Var, Constraint, Objective, generate_standard_repn


def _num(x):
    """float of a Pyomo numeric thing: plain numbers, and NumericValue objects (mutable Params, expressions of them - e.g.
    `con.upper` of a constraint whose bound is a Param), which are evaluated by calling them."""
    try:
        return float(x)
    except TypeError:
        return float(x())


class PyomoLP:
    """StandardFormLP view of a populated Pyomo block, refreshable after Param / bound / fix changes."""

    def __init__(self, block, objective=None, ctypes=None, generate_standard_repn: Optional[Callable] = None):
        if ctypes is None or generate_standard_repn is None:
            Var, Constraint, Objective, gsr = _pyomo()
            ctypes = ctypes or (Var, Constraint, Objective)
            generate_standard_repn = generate_standard_repn or gsr
        self._Var, self._Constraint, self._Objective = ctypes
        self._repn = generate_standard_repn
        self.block = block
        self._objective = objective
        self.lp: Optional[StandardFormLP] = None
        self._cols: Dict[int, int] = {}          # id(var data) -> column
        self._vars: List = []
        self._rows: List = []                    # constraint data objects kept as rows
        self._pattern = None
        self._quad: Dict = {}
        self._soft: List = []
        self.flatten()

    # ---- walking ------------------------------------------------------------------------------------------------------
    def scaling_factors(self):
        """The model's own scaling factors of the flattened columns (Pyomo `scaling_factor` Suffix on the variable's parent
        block, where IDAES' `iscale.set_scaling_factor` puts them; the indexed parent component's entry as a fallback): array
        [n] with 1 where a variable has none, or None when no variable has one."""
        out, found = np.ones(len(self._vars)), False
        for j, v in enumerate(self._vars):
            try:
                suffix = getattr(v.parent_block(), "scaling_factor", None)
            except AttributeError:
                suffix = None
            if suffix is None:
                continue
            val = suffix.get(v, None)
            if val is None and hasattr(v, "parent_component"):
                val = suffix.get(v.parent_component(), None)
            if val is not None and float(val) > 0.0:
                out[j], found = float(val), True
        return out if found else None

    def _active_objective(self):
        if self._objective is not None:
            return self._objective
        objs = list(self.block.component_data_objects(self._Objective, active=True, descend_into=True))
        if len(objs) != 1:
            raise ValueError(f"expected exactly one active objective on the block, found {len(objs)}")
        return objs[0]

    def _linear(self, expr, what):
        r = self._repn(expr, compute_values=True)
        if not r.is_linear():
            raise ValueError(f"{what} is not linear: the HIP backend solves LPs only")
        return r

    def _row_data(self, con):
        """(dict column -> coefficient, constant incl. fixed variables, lower, upper) of one constraint."""
        r = self._linear(con.body, f"constraint {con.name}")
        coefs: Dict[int, float] = {}
        const = float(r.constant)
        for v, a in zip(r.linear_vars, r.linear_coefs):
            a = float(a)
            if v.fixed:
                const += a * float(v.value)
            else:
                j = self._cols.get(id(v))
                if j is None:
                    raise ValueError(f"variable {v.name} of constraint {con.name} is not a variable of the flattened block")
                coefs[j] = coefs.get(j, 0.0) + a
        lo = -INF if con.lower is None else _num(con.lower)
        hi = INF if con.upper is None else _num(con.upper)
        return coefs, const, lo, hi

    # ---- flatten once ---------------------------------------------------------------------------------------------------
    def flatten(self) -> StandardFormLP:
        self._vars = [v for v in self.block.component_data_objects(self._Var, active=True, descend_into=True) if not v.fixed]
        self._cols = {id(v): j for j, v in enumerate(self._vars)}
        n = len(self._vars)
        indptr, indices, data, rlo, rhi, names = [0], [], [], [], [], []
        self._rows = []
        for con in self.block.component_data_objects(self._Constraint, active=True, descend_into=True):
            coefs, const, lo, hi = self._row_data(con)
            coefs = {j: a for j, a in coefs.items() if a != 0.0}
            if not coefs:
                tol = 1e-8 * max(1.0, abs(const))
                if const < lo - tol or const > hi + tol:
                    raise ValueError(f"constraint {con.name} is infeasible once the fixed variables are substituted")
                continue
            if lo > hi:
                raise ValueError(f"constraint {con.name} has crossed bounds ({lo} > {hi})")
            for j in sorted(coefs):
                indices.append(j)
                data.append(coefs[j])
            indptr.append(len(indices))
            rlo.append(lo - const if np.isfinite(lo) else -INF)
            rhi.append(hi - const if np.isfinite(hi) else INF)
            names.append(con.name)
            self._rows.append(con)
        c, c0 = self._objective_vector(n)
        # convex quadratic objective -> soft rows behind the constraint rows (module docstring)
        self._soft = factor_quadratic_form(self._quad, n) if self._quad else []
        compliance = [0.0] * len(self._rows)
        for k, (cols, vals, w) in enumerate(self._soft):
            order = np.argsort(cols)
            indices.extend(int(cols[t]) for t in order)
            data.extend(float(vals[t]) for t in order)
            indptr.append(len(indices))
            rlo.append(0.0)
            rhi.append(0.0)
            names.append(f"objective_quadratic[{k}]")
            compliance.append(1.0 / w)
        lb, ub = self._bounds()
        self._pattern = (np.asarray(indptr, np.int32), np.asarray(indices, np.int32), np.asarray(data, np.float64))
        self.lp = StandardFormLP(n=n, m=len(self._rows) + len(self._soft), indptr=self._pattern[0], indices=self._pattern[1],
                                 data=self._pattern[2], c=c, c0=c0, lb=lb, ub=ub,
                                 rlo=np.asarray(rlo, np.float64), rhi=np.asarray(rhi, np.float64),
                                 col_names=[v.name for v in self._vars], row_names=names,
                                 row_compliance=(np.asarray(compliance, np.float64) if self._soft else None))
        return self.lp

    def _objective_vector(self, n):
        """(c, c0) of the minimised objective; a quadratic part is left in self._quad as {(col_i, col_j): coefficient}."""
        obj = self._active_objective()
        r = self._repn(obj.expr, compute_values=True)
        quadratic = not r.is_linear()
        if quadratic and not (getattr(r, "is_quadratic", lambda: False)() and getattr(r, "nonlinear_expr", None) is None):
            raise ValueError("objective is neither linear nor quadratic: the HIP backend solves LPs and convex QPs only")
        sign = -1.0 if getattr(obj, "sense", 1) in (-1, "maximize") else 1.0        # the solver minimises
        c = np.zeros(n)
        c0 = float(r.constant)
        for v, a in zip(r.linear_vars, r.linear_coefs):
            if v.fixed:
                c0 += float(a) * float(v.value)
            else:
                c[self._cols[id(v)]] += float(a)
        quad: Dict = {}
        if quadratic:
            for (v1, v2), q in zip(r.quadratic_vars, r.quadratic_coefs):
                q = float(q)
                if v1.fixed and v2.fixed:
                    c0 += q * float(v1.value) * float(v2.value)
                elif v1.fixed or v2.fixed:
                    fx, fr = (v1, v2) if v1.fixed else (v2, v1)
                    c[self._cols[id(fr)]] += q * float(fx.value)
                else:
                    i, j = sorted((self._cols[id(v1)], self._cols[id(v2)]))
                    quad[(i, j)] = quad.get((i, j), 0.0) + sign * q
        self._quad = {ij: q for ij, q in quad.items() if q != 0.0}
        self.objective_sign = sign
        return sign * c, sign * c0

    def _bounds(self):
        lb = np.array([-INF if v.lb is None else _num(v.lb) for v in self._vars], np.float64)
        ub = np.array([INF if v.ub is None else _num(v.ub) for v in self._vars], np.float64)
        return lb, ub

    # ---- refresh between solves ---------------------------------------------------------------------------------------
    def refresh(self):
        """Re-read (c, c0, lb, ub, rlo, rhi) after mutable Params / bounds / values of FIXED variables changed.  The set of
        fixed variables and the matrix must be what flatten() saw."""
        now_free = [v for v in self.block.component_data_objects(self._Var, active=True, descend_into=True) if not v.fixed]
        if len(now_free) != len(self._vars) or any(a is not b for a, b in zip(now_free, self._vars)):
            raise MatrixChanged("the set of fixed variables changed since flatten(): flatten again (new constraint matrix)")
        lp = self.lp
        indptr, indices, data = self._pattern
        rlo, rhi = np.empty(lp.m), np.empty(lp.m)
        for i, con in enumerate(self._rows):
            coefs, const, lo, hi = self._row_data(con)
            cols = sorted(j for j, a in coefs.items() if a != 0.0)
            a, b = indptr[i], indptr[i + 1]
            if len(cols) != b - a or any(cj != ij for cj, ij in zip(cols, indices[a:b])) or \
                    any(abs(coefs[cj] - dv) > 1e-12 * max(1.0, abs(dv)) for cj, dv in zip(cols, data[a:b])):
                raise MatrixChanged(f"constraint {con.name} changed its coefficients since flatten(): a mutable Param multiplies a "
                                    "variable there; the batched solver shares ONE matrix across solves - flatten again")
            rlo[i] = lo - const if np.isfinite(lo) else -INF
            rhi[i] = hi - const if np.isfinite(hi) else INF
        rlo[len(self._rows):] = 0.0
        rhi[len(self._rows):] = 0.0
        quad_was = dict(self._quad)
        lp.c, lp.c0 = self._objective_vector(lp.n)
        if set(quad_was) != set(self._quad) or any(abs(quad_was[k] - q) > 1e-12 * max(1.0, abs(q)) for k, q in self._quad.items()):
            raise MatrixChanged("the quadratic part of the objective changed since flatten(): its factors are rows of the shared "
                                "matrix - flatten again")
        lp.lb, lp.ub = self._bounds()
        lp.rlo, lp.rhi = rlo, rhi
        return lp

    # ---- solution back into the Pyomo variables ---------------------------------------------------------------------------
    def load_solution(self, x):
        for v, xv in zip(self._vars, np.asarray(x, float)):
            v.value = float(xv)

    def objective_value(self, x):
        """Objective in the model's own sense (a maximisation problem reports the maximum)."""
        return self.objective_sign * self.lp.objective(np.asarray(x, float))


class PyomoScenarioBatch:
    """B structurally identical Pyomo blocks (the reference's `model.fs[i]`, one per price scenario) as ONE batch for
    `HipPdlpSolver.solve`: the batch-model protocol of dispatches_amd/workflow/batch_model.py (lp, n_scenario, c, c0,
    scenario_bounds(), store_solution(), solver_hints) over a list of `PyomoLP` views.  The blocks must flatten to the same
    matrix (same variables, rows and coefficients in the same order; soft rows of a quadratic objective included) - they
    differ in what a batch may differ in: objective vector and constant, variable bounds, row bounds."""

    def __init__(self, blocks, objectives=None, ctypes=None, generate_standard_repn: Optional[Callable] = None,
                 solver_hints: Optional[dict] = None, column_scaling: Optional[str] = "auto"):
        blocks = list(blocks)
        if not blocks:
            raise ValueError("no scenario blocks")
        objectives = list(objectives) if objectives is not None else [None] * len(blocks)
        self.views = [PyomoLP(b, objective=o, ctypes=ctypes, generate_standard_repn=generate_standard_repn)
                      for b, o in zip(blocks, objectives)]
        ref = self.views[0]
        for i, v in enumerate(self.views[1:], 1):
            same = (v.lp.n == ref.lp.n and v.lp.m == ref.lp.m and all(np.array_equal(a, b) for a, b in zip(v._pattern, ref._pattern))
                    and np.array_equal(v.lp.row_compliance if v.lp.row_compliance is not None else 0.0,
                                       ref.lp.row_compliance if ref.lp.row_compliance is not None else 0.0))
            if not same:
                raise ValueError(f"scenario block {i} does not flatten to the matrix of block 0: a batch shares ONE constraint matrix")
        self.lp = ref.lp
        self.n_scenario = len(self.views)
        self.solver_hints = dict(solver_hints or {})
        self.solve_handle = None
        self.x = self.y = self.objective = self.status = self.iterations = None
        self._stack()
        self.column_scaling = column_scaling
        self.lp.col_scale = self._column_scales(column_scaling)

    def _column_scales(self, mode):
        """Variable scaling factors for the solver (dsp_lp_desc::col_scale = typical magnitude of every column), so that a
        flowsheet handed over as a Pyomo model gets what the product's own model objects give their LPs (Bidder, column_scaling =
        "implied_ranges") without per-model hints:
          "suffix"          the reciprocals of the scaling factors the model carries (IDAES: `iscale.set_scaling_factor(var, s)` =
                            `var.parent_block().scaling_factor[var] = s`; 1 where a variable has none)
          "implied_ranges"  dispatches_amd.lp.implied_column_ranges over the batch's current bounds
          "auto" (default)  the suffix if any variable carries a factor other than 1, else the implied ranges for LPs beyond the
                            in-wave simplex (n + m > 128), else none
          None              none."""
        if mode is None:
            return None
        if mode not in ("auto", "suffix", "implied_ranges"):
            raise ValueError(f"column_scaling = {mode!r}")
        if mode in ("auto", "suffix"):
            sf = self.views[0].scaling_factors()
            if sf is not None and (mode == "suffix" or np.any(sf != 1.0)):
                return 1.0 / sf
            if mode == "suffix":
                return None
        if mode == "implied_ranges" or self.lp.n + self.lp.m > 128:
            from .lp import implied_column_ranges
            return implied_column_ranges(self.lp, self.lb, self.ub)
        return None

    def _stack(self):
        vs = self.views
        self.c = np.stack([v.lp.c for v in vs])
        self.c0 = np.array([v.lp.c0 for v in vs], np.float64)
        self.lb, self.ub = np.stack([v.lp.lb for v in vs]), np.stack([v.lp.ub for v in vs])
        self.rlo, self.rhi = np.stack([v.lp.rlo for v in vs]), np.stack([v.lp.rhi for v in vs])

    def refresh(self):
        for v in self.views:
            v.refresh()
        self._stack()

    def scenario_bounds(self):
        return self.lb, self.ub, self.rlo, self.rhi

    def store_solution(self, x, y, objective, status, iterations=None):
        self.x, self.y = np.asarray(x), np.asarray(y)
        self.status = np.asarray(status)
        self.iterations = None if iterations is None else np.asarray(iterations)
        # objectives in the models' own sense; solutions back into the Vars of every scenario that solved
        self.objective = np.array([v.objective_sign * float(f) for v, f in zip(self.views, np.asarray(objective))])
        for v, xi, st in zip(self.views, self.x, self.status):
            if st == 0:
                v.load_solution(xi)


class HipPyomoSolver:
    """Drop-in for the reference's `pyo.SolverFactory(name)` object (run_double_loop_battery.py:123) on populated LINEAR /
    convex-quadratic Pyomo models: `solve(model)` flattens once, refreshes the mutable data on every later call, solves
    all scenario blocks in one batch on the GPU and loads the solutions back into the Vars.

        solver = HipPyomoSolver(device=0)
        results = solver.solve(m)                      # one block with its own objective
        results = solver.solve([m.fs[i] for i in m.fs.index_set()])     # one batch of identical scenario blocks

    `backend` is the batched solver object (default: `HipPdlpSolver`, which fails loudly without the HIP library or a GPU;
    the tests inject a CPU stand-in).  Written against Pyomo's public API, never run against a real Pyomo model (module
    docstring)."""

    def __init__(self, device: int = 0, backend=None, ctypes=None, generate_standard_repn: Optional[Callable] = None,
                 solver_hints: Optional[dict] = None, column_scaling: Optional[str] = "auto", **solver_options):
        self._backend = backend
        self._column_scaling = column_scaling
        self._device, self._solver_options = device, solver_options
        self._ctypes, self._repn, self._hints = ctypes, generate_standard_repn, solver_hints
        self._batches: Dict[tuple, PyomoScenarioBatch] = {}
        self.reflattened = 0             # how many times a structural change between two solves forced a new flatten

    def available(self, exception_flag=False):
        b = self._get_backend()
        return b.available(exception_flag) if hasattr(b, "available") else True

    def _get_backend(self):
        if self._backend is None:
            from .hip_solver import HipPdlpSolver
            self._backend = HipPdlpSolver(device=self._device, **self._solver_options)
        return self._backend

    def solve(self, model, tee=False, objectives=None, **kwargs):
        blocks = list(model) if isinstance(model, (list, tuple)) else [model]
        key = tuple(id(b) for b in blocks)
        batch = self._batches.get(key)
        if batch is None:
            batch = self._batches[key] = PyomoScenarioBatch(blocks, objectives, self._ctypes, self._repn, self._hints, self._column_scaling)
        else:
            try:
                batch.refresh()
            except MatrixChanged:
                # the model changed structurally between two solves (the reference's model objects do this legitimately: e.g.
                # transform_design_model_to_operation_model fixes design variables after a first solve, a Var is unfixed for a
                # sweep): what a Pyomo solver object does on every call - write the model again - happens here only now.  The
                # old device handle goes with the old batch object.
                self.reflattened += 1
                batch = self._batches[key] = PyomoScenarioBatch(blocks, objectives, self._ctypes, self._repn, self._hints, self._column_scaling)
        results = self._get_backend().solve(batch, tee=tee)
        self.last_batch = batch
        return results
