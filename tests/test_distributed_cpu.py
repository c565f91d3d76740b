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


@pytest.mark.parametrize("gather", [False, True])
def test_sharded_solve_world2_gloo(gather):
    import torch.multiprocessing as mp
    from dispatches_amd import scenarios
    from tests._highs_solver import HighsTestSolver
    B, world = 7, 2                       # ragged: shards of 4 and 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
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
