"""world_size-2 `gloo` test of the scenario sharding + all-gather path on CPU.

The local solves use the TEST-ONLY HiGHS stand-in solver (tests/_highs_solver.py); what is exercised here is the
product's partitioning, packing and collective logic, which is identical under RCCL on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, gather, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from dispatches_amd import scenarios
    from dispatches_amd.distributed import shard_bounds, solve_sharded
    from tests._highs_solver import HighsTestSolver
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        solver = HighsTestSolver()
        bidder, model = scenarios.make_batch("nuclear_24h", B, solver)
        lo, hi = solve_sharded(model, solver, gather_solution=gather)
        assert (lo, hi) == shard_bounds(B, world, rank)
        q.put((rank, lo, hi, model.objective.copy(), model.status.copy(), np.isnan(model.x).any(axis=1)))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_partition():
    from dispatches_amd.distributed import shard_bounds
    for B in (0, 1, 7, 8, 4096, 8191):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(B, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _run_world(world, B, gather, timeout=600):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("gather", [False, True])
def test_sharded_solve_world2_gloo(gather):
    from dispatches_amd import scenarios
    from tests._highs_solver import HighsTestSolver
    B, world = 7, 2                       # ragged: shards of 4 and 3
    got = _run_world(world, B, gather)
    # single-process reference
    solver = HighsTestSolver()
    bidder, model = scenarios.make_batch("nuclear_24h", B, solver)
    solver.solve(model)
    for rank, lo, hi, obj, status, xnan in got:
        np.testing.assert_allclose(obj, model.objective, rtol=1e-12)      # every rank holds ALL objectives
        assert (status == 0).all()
        if gather:
            assert not xnan.any()
        else:
            assert not xnan[lo:hi].any() and xnan[:lo].all() and xnan[hi:].all()


def test_sharded_solve_world8_gloo_ragged():
    """The node the scaling bench runs on has 8 ranks: BASELINE config 4's 8192 scenarios minus one (8191 = 7 shards of 1024 + one of
    1023) through the same partition / pack / all-gather code under gloo; every rank ends up with all 8191 objectives, and they are the
    single-process ones (spot-checked: a full single-process solve would double the test's time)."""
    from dispatches_amd import scenarios
    from dispatches_amd.distributed import shard_bounds
    from tests._highs_solver import HighsTestSolver
    B, world = 8191, 8
    got = _run_world(world, B, False, timeout=900)
    assert sorted(g[0] for g in got) == list(range(world))
    sizes = sorted(hi - lo for _, lo, hi, *_ in got)
    assert sizes == [1023] + [1024] * 7
    ref = got[0][3]
    for rank, lo, hi, obj, status, xnan in got:
        assert (lo, hi) == shard_bounds(B, world, rank)
        assert obj.shape == (B,) and (status == 0).all() and np.isfinite(obj).all()
        np.testing.assert_array_equal(obj, ref)                             # every rank holds the same 8191 objectives
        assert not xnan[lo:hi].any() and xnan[:lo].all() and xnan[hi:].all()
    solver = HighsTestSolver()
    bidder, model = scenarios.make_batch("nuclear_24h", B, solver)
    pick = np.r_[0:8, 1020:1028, 4090:4100, 7160:7170, 8183:8191]            # around the shard edges
    sub = _subset(model, pick)
    solver.solve(sub)
    np.testing.assert_allclose(ref[pick], sub.objective, rtol=1e-12)


def _subset(model, ids):
    """The scenarios `ids` of a batch model as a batch model of their own (shares the LP and the block)."""
    return _ShardView(model, ids)


class _ShardView:
    def __init__(self, model, ids):
        self.lp, self.block, self.n_scenario = model.lp, model.block, len(ids)
        self.c, self.c0 = model.c[ids], np.broadcast_to(model.c0, (model.n_scenario,))[ids]
        lb, ub, rlo, rhi = model.scenario_bounds()
        pick = lambda a: a[ids] if a.ndim == 2 else a
        self._bounds = (pick(lb), pick(ub), pick(rlo), pick(rhi))
        self.solve_handle = None

    def scenario_bounds(self):
        return self._bounds

    def store_solution(self, x, y, objective, status, iterations=None):
        self.x, self.y, self.objective, self.status = x, y, np.asarray(objective), np.asarray(status)
