"""Cross-scenario coupling of the stochastic bidders.

Upstream idaes couples the n_scenario copies of the day-ahead problem:

    SelfScheduler   day_ahead_power[s, t] == day_ahead_power[0, t]                      (non-anticipativity: ONE schedule)
    Bidder          (day_ahead_power[k, t] - day_ahead_power[j, t]) (price[k, t] - price[j, t]) >= 0   for all pairs
                                                                                         (monotone bid curve)

and minimises the SUM of the scenario objectives (weight 1 per scenario: pinned by the reference's notebook log G4).
No reference vector exercises these rows (every golden has identical scenarios, where they are vacuous), so this
restatement is UNPINNED against the reference; it is checked against an independent oracle formulation
(oracle/dispatch_lp_oracle.py: coupled_da) in tests.

The coupled problem is ONE LP: block-diagonal copies of the scenario LP plus the coupling rows on the
day_ahead_power columns.  Coefficients of the coupling rows never change (+1 / -1); for the monotone form the PRICE
ORDER only decides which side of the row is active, i.e. it lives in the (mutable) row bounds - the constraint matrix
is flattened once, as everywhere else.  Sizes: S (8 T + 2) columns - beyond the register/LDS-resident kernels from
S = 3 at 48 h, so the solver's HBM-resident streaming path takes it; a few dozen scenarios at most (the monotone form
has S (S - 1) / 2 * T rows), not the 4096-scenario independent batches of the throughput benchmark.
"""
from __future__ import annotations

import numpy as np

from ..lp import StandardFormLP


class CoupledScenarioModel:
    """The coupled LP presented to `solver.solve` as a one-scenario batch model."""

    def __init__(self, model, mode):
        if mode not in ("non_anticipative", "monotone"):
            raise ValueError(mode)
        self.mode = mode
        self.base = model
        lp, S = model.lp, model.n_scenario
        self.S, self.n1, self.m1 = S, lp.n, lp.m
        T = len(model.HOUR)
        pda = np.asarray(model.pda_cols)
        pairs = [(0, s) for s in range(1, S)] if mode == "non_anticipative" else \
            [(j, k) for j in range(S) for k in range(j + 1, S)]
        self.pairs = pairs
        # block diagonal + coupling rows  pda[k, t] - pda[j, t]
        indptr, indices, data = [0], [], []
        for s in range(S):
            for i in range(lp.m):
                a, b = lp.indptr[i], lp.indptr[i + 1]
                indices.extend((lp.indices[a:b] + s * lp.n).tolist())
                data.extend(lp.data[a:b].tolist())
                indptr.append(len(indices))
        for j, k in pairs:
            for t in range(T):
                cj, ck = j * lp.n + pda[t], k * lp.n + pda[t]
                indices.extend([cj, ck])
                data.extend([-1.0, 1.0])
                indptr.append(len(indices))
        n, m = S * lp.n, S * lp.m + len(pairs) * T
        names_c = [f"s{s}.{nm}" for s in range(S) for nm in lp.col_names]
        names_r = [f"s{s}.{nm}" for s in range(S) for nm in lp.row_names] + \
                  [f"coupling[{j},{k},{t}]" for j, k in pairs for t in range(T)]
        self.lp = StandardFormLP(n=n, m=m, indptr=np.asarray(indptr, np.int32), indices=np.asarray(indices, np.int32),
                                 data=np.asarray(data, np.float64), c=np.zeros(n), c0=0.0, lb=np.zeros(n), ub=np.zeros(n),
                                 rlo=np.zeros(m), rhi=np.zeros(m), col_names=names_c, row_names=names_r,
                                 # soft rows (quadratic ramp cost) of every scenario copy; the coupling rows are hard
                                 row_compliance=(None if lp.row_compliance is None else
                                                 np.concatenate([np.tile(lp.row_compliance, S), np.zeros(len(pairs) * T)])))
        self.n_scenario, self.SCENARIOS, self.HOUR = 1, range(1), model.HOUR
        self.block = model.block
        self.solve_handle = None
        self.solver_hints = dict(getattr(model, "solver_hints", None) or {})
        self.x = self.y = self.objective = self.status = self.iterations = None
        self.c = self.c0 = None
        self._bounds = None

    def load(self, prices):
        """Per-call data from the scenario model: objective, bounds, and the coupling rows' bounds from `prices` [S, T]
        (the energy prices the bid curve is monotone in: day-ahead prices for the day-ahead problem)."""
        m0 = self.base
        S, T = self.S, len(self.HOUR)
        lb, ub, rlo, rhi = [np.broadcast_to(a, (S, a.shape[-1])) for a in m0.scenario_bounds()]
        self.c = np.asarray(m0.c, float).reshape(1, -1).copy()
        self.c0 = np.array([float(np.sum(np.broadcast_to(m0.c0, (S,))))])
        clo, chi = [], []
        for j, k in self.pairs:
            if self.mode == "non_anticipative":
                clo.append(np.zeros(T)); chi.append(np.zeros(T))
            else:
                d = np.asarray(prices[k], float)[:T] - np.asarray(prices[j], float)[:T]
                clo.append(np.where(d > 0, 0.0, -np.inf))       # price_k > price_j  =>  pda_k - pda_j >= 0
                chi.append(np.where(d < 0, 0.0, np.inf))        # price_k < price_j  =>  pda_k - pda_j <= 0
        self._bounds = (lb.reshape(-1), ub.reshape(-1),
                        np.concatenate([rlo.reshape(-1)] + clo), np.concatenate([rhi.reshape(-1)] + chi))

    def scenario_bounds(self):
        return self._bounds

    def store_solution(self, x, y, objective, status, iterations=None):
        self.x, self.y = np.asarray(x), np.asarray(y)
        self.objective, self.status = np.asarray(objective), np.asarray(status)
        self.iterations = None if iterations is None else np.asarray(iterations)
        S, n1, m1 = self.S, self.n1, self.m1
        xs = self.x[0].reshape(S, n1)
        ys = self.y[0][:S * m1].reshape(S, m1)
        m0 = self.base
        obj = np.sum(np.asarray(m0.c, float) * xs, axis=1) + np.broadcast_to(m0.c0, (S,))
        if m0.lp.row_compliance is not None:
            obj = obj + np.array([m0.lp.quadratic_value(xs[s]) for s in range(S)])
        m0.store_solution(xs, ys, obj, np.full(S, int(self.status[0]), np.int32),
                          None if self.iterations is None else np.full(S, int(self.iterations[0])))
        m0.coupled_objective = float(self.objective[0])
