"""Scenario sharding across the GPUs of one node.

Scenarios are independent LPs, so the batch partitions with NO data-path collective: rank r of G (one process per
GPU, `torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) owns the contiguous
slice [r*ceil(B/G), ...) of the scenario batch, solves it with its own device handle, and a single all-gather of
the converged objectives (8 B per scenario; optionally status and the P_T setpoints) makes the results of the whole
batch available on every rank.  Message sizes are 32 KiB-3 MiB: latency-bound, one collective per solve.

The reference has no counterpart (its only parallelism is `multiprocessing.Pool` over sweep points,
run_pricetaker_wind_PEM.py:106-107); this is the MI355X-native replacement of that process pool for the
price-scenario axis of Bidder.compute_day_ahead_bids.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_scenario: int, world: int, rank: int):
    """Contiguous, balanced partition: the first (n % world) ranks get one extra scenario."""
    base, extra = divmod(int(n_scenario), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class _ShardView:
    """The slice [lo, hi) of a ScenarioBatchModel, presented to a solver as a model of its own."""

    def __init__(self, model, lo, hi):
        self._m, self.lo, self.hi = model, lo, hi
        self.lp, self.block = model.lp, model.block
        self.n_scenario = hi - lo
        self.HOUR = model.HOUR
        sl = slice(lo, hi)
        self.c, self.c0 = model.c[sl], np.broadcast_to(model.c0, (model.n_scenario,))[sl]
        self.x = self.y = self.objective = self.status = self.iterations = None
        self.solve_handle = getattr(model, "solve_handle", None)
        self.solver_hints = getattr(model, "solver_hints", None)

    def scenario_bounds(self):
        sl = slice(self.lo, self.hi)
        return tuple(a[sl] if np.ndim(a) == 2 else a for a in self._m.scenario_bounds())

    def store_solution(self, x, y, objective, status, iterations=None):
        self.x, self.y = np.asarray(x), np.asarray(y)
        self.objective, self.status = np.asarray(objective), np.asarray(status)
        self.iterations = None if iterations is None else np.asarray(iterations)


def solve_sharded(model, solver, group=None, gather_solution: bool = False):
    """Solve `model`'s scenarios sharded over the ranks of `group` and all-gather the results.

    Every rank must call this with an identical `model` (same LP, same per-scenario data).  After the call
    ``model.objective`` / ``model.status`` / ``model.iterations`` hold ALL scenarios on every rank;
    ``model.x`` / ``model.y`` hold all scenarios if `gather_solution`, otherwise only the local shard is valid
    (rows outside the shard are NaN).  Returns (lo, hi), the shard this rank solved."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = model.n_scenario
    lo, hi = shard_bounds(B, world, rank)
    view = _ShardView(model, lo, hi)
    if hi > lo:
        solver.solve(view)
        model.solve_handle = view.solve_handle
    n, m = model.lp.n, model.lp.m
    per = max(shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world))
    width = 3 + ((n + m) if gather_solution else 0)          # objective, status, iterations [, x, y]
    dev = torch.device("cuda", torch.cuda.current_device()) if (dist.is_initialized() and
                                                                 dist.get_backend(group) == "nccl") else "cpu"
    local = torch.full((per, width), float("nan"), dtype=torch.float64, device=dev)
    if hi > lo:
        cols = [view.objective, view.status.astype(np.float64),
                (view.iterations if view.iterations is not None else np.zeros(hi - lo)).astype(np.float64)]
        packed = np.stack(cols, 1)
        if gather_solution:
            packed = np.concatenate([packed, view.x, view.y.reshape(hi - lo, m)], 1)
        local[:hi - lo] = torch.as_tensor(packed, dtype=torch.float64).to(dev)
    if world > 1:
        everything = torch.empty((world * per, width), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(everything, local, group=group)       # the ONE collective of the solve
        everything = everything.cpu().numpy().reshape(world, per, width)
    else:
        everything = local.cpu().numpy()[None]
    obj = np.empty(B)
    status = np.empty(B, np.int32)
    iters = np.empty(B, np.int64)
    x = np.full((B, n), np.nan)
    y = np.full((B, m), np.nan)
    for r in range(world):
        a, b = shard_bounds(B, world, r)
        blk = everything[r, :b - a]
        obj[a:b], status[a:b], iters[a:b] = blk[:, 0], blk[:, 1].astype(np.int32), blk[:, 2].astype(np.int64)
        if gather_solution:
            x[a:b], y[a:b] = blk[:, 3:3 + n], blk[:, 3 + n:3 + n + m]
    if not gather_solution and hi > lo:
        x[lo:hi], y[lo:hi] = view.x, view.y.reshape(hi - lo, m)
    model.store_solution(x, y, obj, status, iters)
    return lo, hi
