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

import os

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


def make_gather_buffers(world: int, per_rank: int, device, width: int = 2):
    """Send / receive buffers of the per-solve collective: `width` float64 columns per scenario (objective, status
    [, iterations, x, y]), shards padded to `per_rank` rows (the largest shard) with NaN."""
    import torch
    return dict(local=torch.full((per_rank, width), float("nan"), dtype=torch.float64, device=device),
                all=torch.empty((world * per_rank, width), dtype=torch.float64, device=device))


def gather_device_results(out, buffers, per_rank=None, group=None, columns=("obj", "status")):
    """The ONE collective of a sharded solve, straight from the solver's DEVICE outputs: the columns of `out` (device
    tensors of this rank's shard: `obj` [b], `status` [b], `iters` [b], `x` [b, n], `y` [b, m]) are packed on the device
    into buffers["local"] and all-gathered into buffers["all"] (RCCL over xGMI under backend "nccl"; no host hop).
    Stream-ordered on the current stream.  Returns buffers["all"] viewed as [world, per_rank, width]."""
    import torch.distributed as dist
    local, everything = buffers["local"], buffers["all"]
    per = local.shape[0]
    col = 0
    for key in columns:
        t = out[key]
        b = t.shape[0]
        w = 1 if t.dim() == 1 else t.shape[1]
        local[:b, col:col + w] = t.reshape(b, w)            # dtype conversion (int32 status -> float64) happens here
        col += w
    if local.is_cuda and dist.get_backend(group) != "nccl":
        # REHEARSAL of the multi-GPU path on a box with one GPU (several ranks sharing cuda:0 cannot form an RCCL communicator):
        # the same device-side packing, the collective itself on host copies through gloo, the result back on the device.
        host_all = everything.cpu()
        dist.all_gather_into_tensor(host_all, local.cpu(), group=group)
        everything.copy_(host_all)
    else:
        dist.all_gather_into_tensor(everything, local, group=group)
    return everything.view(-1, per, local.shape[1])


def solve_sharded(model, solver, group=None, gather_solution: bool = False):
    """Solve `model`'s scenarios sharded over the ranks of `group` and all-gather the results.

    Every rank must call this with an identical `model` (same LP, same per-scenario data).  After the call
    ``model.objective`` / ``model.status`` / ``model.iterations`` / ``model.flags`` hold ALL scenarios on every rank;
    ``model.x`` / ``model.y`` hold all scenarios if `gather_solution`, otherwise only the local shard is valid
    (rows outside the shard are NaN).  Returns (lo, hi), the shard this rank solved.

    Under RCCL (backend "nccl") the collective reads the solver's device outputs directly
    (`solver.last_device_out`, set by HipPdlpSolver.solve) and the gathered batch is copied to the host once; under
    gloo (CPU tests) the same packing runs on host tensors."""
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
    width = 4 + ((n + m) if gather_solution else 0)          # objective, status, iterations, flags [, x, y]
    # DSP_REHEARSE_ON_DEVICE=1: take the device path under gloo too (ranks sharing one GPU: gather_device_results stages the
    # collective through the host) - what the single-GPU rehearsal of the multi-GPU bench uses
    on_gpu = dist.is_initialized() and (dist.get_backend(group) == "nccl" or
                                        (os.environ.get("DSP_REHEARSE_ON_DEVICE") == "1" and torch.cuda.is_available()))
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    columns = ("obj", "status", "iters", "flags") + (("x", "y") if gather_solution else ())
    dev_out = getattr(solver, "last_device_out", None) if on_gpu else None
    if hi > lo and dev_out is not None and dev_out["obj"].shape[0] == hi - lo:
        # device path: objective is c.x on the device, the model's constant is added after the gather
        out = dict(dev_out)
        out["y"] = out["y"][:, :m]
        c0_local = torch.as_tensor(np.ascontiguousarray(view.c0, np.float64)).to(dev)
        out["obj"] = out["obj"] + c0_local
    elif hi > lo:
        it = view.iterations if view.iterations is not None else np.zeros(hi - lo)
        fl = getattr(view, "flags", None)
        host = dict(obj=view.objective, status=view.status, iters=it, flags=(np.zeros(hi - lo) if fl is None else fl), x=view.x,
                    y=np.reshape(view.y, (hi - lo, m)))
        out = {k: torch.as_tensor(np.ascontiguousarray(host[k], np.float64)).to(dev) for k in columns}
    else:
        out = None
    buffers = make_gather_buffers(world, per, dev, width)
    if world > 1:
        if out is None:                                     # empty shard: still takes part in the collective, through the SAME
            everything = gather_device_results({}, buffers, per, group, ())     # path as the others (host staging under the rehearsal)
        else:
            everything = gather_device_results(out, buffers, per, group, columns)   # the ONE collective of the solve
        everything = everything.cpu().numpy()
    else:
        col = 0
        for key in columns:
            t = out[key]
            w = 1 if t.dim() == 1 else t.shape[1]
            buffers["local"][:hi - lo, col:col + w] = t.reshape(hi - lo, w)
            col += w
        everything = buffers["local"].cpu().numpy()[None]
    obj = np.empty(B)
    status = np.empty(B, np.int32)
    iters = np.empty(B, np.int64)
    flags = np.zeros(B, np.int32)
    x = np.full((B, n), np.nan)
    y = np.full((B, m), np.nan)
    for r in range(world):
        a, b = shard_bounds(B, world, r)
        blk = everything[r, :b - a]
        obj[a:b], status[a:b], iters[a:b] = blk[:, 0], blk[:, 1].astype(np.int32), blk[:, 2].astype(np.int64)
        flags[a:b] = blk[:, 3].astype(np.int32)
        if gather_solution:
            x[a:b], y[a:b] = blk[:, 4:4 + n], blk[:, 4 + n:4 + n + m]
    if not gather_solution and hi > lo:
        x[lo:hi], y[lo:hi] = view.x, np.reshape(view.y, (hi - lo, m))
    model.store_solution(x, y, obj, status, iters)
    # DSP_FLAG_* bits of every scenario on every rank: a scenario accepted without a certified objective accuracy on ANOTHER
    # rank must not look certified here (Bidder / Tracker read model.flags)
    model.flags = flags
    return lo, hi
