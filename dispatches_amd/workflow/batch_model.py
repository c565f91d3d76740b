"""The object a Bidder / Tracker hands to ``solver.solve(model)``.

The reference hands a Pyomo ConcreteModel with one block per scenario (``model.fs[i]``, ``model.SCENARIOS``;
``parametrized_bidder.py:146-149``).  Here the scenarios share ONE flattened LP; what differs per scenario are
dense vectors stored as ``[B, n]`` / ``[B, m]`` arrays.  ``model.fs[i]`` still works: it returns the shared
LinearBlock positioned on scenario i's solution, so model objects' ``record_results(model.fs[i], ...)`` and
``get_implemented_profile(model.fs[i], ...)`` run unchanged.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..lp import LinearBlock, LinExpr, StandardFormLP

# per-scenario termination codes written by the solver (include/dsp_hip.h)
STATUS_OPTIMAL, STATUS_ITERATION_LIMIT, STATUS_PRIMAL_INFEASIBLE, STATUS_DUAL_INFEASIBLE, STATUS_NUMERICAL = range(5)


class _ScenarioIndexer:
    def __init__(self, model):
        self._model = model

    def __getitem__(self, i):
        m = self._model
        if not 0 <= i < m.n_scenario:
            raise KeyError(i)
        m.block.solution = m.row(i)
        return m.block

    def index_set(self):
        return self._model.SCENARIOS

    def __len__(self):
        return self._model.n_scenario

    def __iter__(self):
        return iter(self._model.SCENARIOS)


class SolveResults:
    """Minimal stand-in for a Pyomo results object (status / termination_condition strings)."""

    def __init__(self, status, termination_condition, **info):
        self.solver = type("SolverInfo", (), dict(status=status, termination_condition=termination_condition))()
        self.info = info

    def __repr__(self):
        return f"SolveResults(status={self.solver.status}, termination={self.solver.termination_condition}, {self.info})"


class ScenarioBatchModel:
    def __init__(self, block: LinearBlock, n_scenario: int, horizon: int, indexed: bool = True):
        self.block = block
        self.n_scenario = int(n_scenario)
        self.SCENARIOS = range(self.n_scenario)
        self.HOUR = range(horizon)
        self.lp: Optional[StandardFormLP] = None
        self._indexed = indexed
        # per-scenario data (None -> template)
        self.c = None
        self.c0 = None
        self.lb = self.ub = self.rlo = self.rhi = None
        # results (x / y may be held by the solver on the device and fetched on first access: store_solution(lazy=...))
        self._x = self._y = self._lazy = None
        self._row_cache = {}
        self.objective = self.status = self.iterations = None
        self.solve_handle = None      # solver-owned cache (device copy of A, scaling, workspace)
        # the objective vectors may be a RECIPE instead of a [B, n] array (Bidder._pass_price_forecasts): base vector + price windows,
        # which a solver with device-side pricing turns into c on the device; `c` materialises it on the host on first access
        self._c = None
        self.c_recipe = None

    supports_lazy_solution = True     # HipPdlpSolver leaves x / y on the device until somebody reads them

    # -- lazily materialised arrays ------------------------------------------------------------------------------------
    @property
    def x(self):
        if self._x is None and self._lazy is not None:
            self._x, self._y = self._lazy.fetch()
        return self._x

    @x.setter
    def x(self, v):
        self._x = v
        if v is None:
            self._lazy = None

    @property
    def y(self):
        if self._y is None and self._lazy is not None:
            self._x, self._y = self._lazy.fetch()
        return self._y

    @y.setter
    def y(self, v):
        self._y = v

    @property
    def has_solution(self):
        return self._x is not None or self._lazy is not None

    def row(self, i):
        """x[i] (None before the first solve); with the solution still on the device, rows are fetched sixteen at a time - record_bids
        walks the first `detail_scenarios` scenarios one by one."""
        if self._x is None and self._lazy is not None:
            blk = i // 16
            hit = self._row_cache.get(blk)
            if hit is None:
                hit = self._row_cache[blk] = self._lazy.rows(16 * blk, min(16 * blk + 16, self.n_scenario))
            return hit[i - 16 * blk]
        return None if self._x is None else self._x[i]

    def block_of(self, scenarios):
        """The block with the solutions of SEVERAL scenarios at once (block.solution = x[scenarios], [S, n]): what a model object's
        record_results_many reads - the per-scenario `model.fs[i]` walk evaluates the same expressions S times."""
        self.block.solution = np.stack([self.row(i) for i in scenarios])
        return self.block

    def columns(self, cols):
        """x[:, cols] without fetching the whole solution when it still sits on the device."""
        cols = np.asarray(cols)
        if self._x is None and self._lazy is not None:
            flat = self._lazy.columns(cols.reshape(-1))
            return flat.reshape((flat.shape[0],) + cols.shape)
        return self.x[:, cols]

    @property
    def c(self):
        if self._c is None and self.c_recipe is not None:
            self._c = self.c_recipe.dense()
        return self._c

    @c.setter
    def c(self, v):
        self._c = v
        self.c_recipe = None

    # model.fs[i] (bidder) or model.fs (tracker: a single block)
    @property
    def fs(self):
        if self._indexed:
            return _ScenarioIndexer(self)
        self.block.solution = self.row(0)
        return self.block

    def finalize(self, objective: LinExpr):
        """Flatten once; afterwards only vectors change."""
        self.lp = self.block.flatten(objective)
        self.c = np.tile(self.lp.c, (self.n_scenario, 1))
        self.c0 = np.full(self.n_scenario, self.lp.c0)
        return self.lp

    def scenario_bounds(self):
        """Current (lb, ub, rlo, rhi): per-scenario arrays where set, otherwise the block's current template."""
        tlb, tub, trlo, trhi = self.block.current_bounds()
        pick = lambda a, t: t if a is None else a
        return pick(self.lb, tlb), pick(self.ub, tub), pick(self.rlo, trlo), pick(self.rhi, trhi)

    def store_solution(self, x, y, objective, status, iterations=None, lazy=None):
        """lazy: an object with fetch() -> (x, y) and columns(cols) -> x[:, cols] (hip_solver.DeviceSolution): the solution stays
        where the solver left it until `x` / `y` are read; `fs[i]` positions the block on scenario i as before."""
        self._row_cache = {}
        if lazy is not None:
            self._x = self._y = None
            self._lazy = lazy
        else:
            self._lazy = None
            self._x, self._y = np.asarray(x), np.asarray(y)
        self.objective, self.status = np.asarray(objective), np.asarray(status)
        self.iterations = None if iterations is None else np.asarray(iterations)
        self.block.solution = None if lazy is not None else self._x[0]

    def expression_values(self, family: str) -> np.ndarray:
        """[B, T] values of an expression family (e.g. 'P_T') for every scenario at once.  The expressions have two or
        three terms each, so this is a gather over the few columns involved, not a dense [B, n] x [n, T] product."""
        fam = self.block.expressions[family]
        T = len(fam)
        K = max(1, max(len(fam[t].coef) for t in range(T)))
        cols = np.zeros((T, K), dtype=np.intp)
        vals = np.zeros((T, K))
        k = np.zeros(T)
        for t in range(T):
            for e, (j, v) in enumerate(fam[t].coef.items()):
                cols[t, e], vals[t, e] = j, v
            k[t] = fam[t].const
        return (self.columns(cols) * vals).sum(axis=2) + k
