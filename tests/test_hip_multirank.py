"""Rehearsal of the multi-GPU paths on ONE GPU: two ranks share cuda:0, the collectives run over gloo on host
copies (RCCL cannot put two ranks on one device).  Everything else is what the 8-GPU launch runs: `solve_sharded` through the
DEVICE gather path (HIP solves, device-side packing, one all-gather, flags included) and `bench.py --gpus 2` launched by
torch.distributed.run exactly as the driver does (sharding, barriers, MAX-over-ranks timing, one JSON line from rank 0)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

gpu = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DSP_REHEARSE_ON_DEVICE="1")
    import torch
    import torch.distributed as dist
    from dispatches_amd import scenarios
    from dispatches_amd.distributed import shard_bounds, solve_sharded
    from dispatches_amd.hip_solver import HipPdlpSolver
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        solver = HipPdlpSolver(device=0)
        bidder, model = scenarios.make_batch("nuclear_24h", B, solver)
        lo, hi = solve_sharded(model, solver, gather_solution=(rank >= 0))
        assert (lo, hi) == shard_bounds(B, world, rank)
        assert solver.last_device_out["obj"].is_cuda and solver.last_device_out["obj"].shape[0] == hi - lo
        q.put((rank, model.objective.copy(), model.status.copy(), model.flags.copy(), np.isnan(model.x).any()))
    finally:
        dist.destroy_process_group()


@gpu
def test_sharded_solve_two_ranks_on_one_gpu():
    import torch.multiprocessing as mp
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    B, world = 7, 2                                       # ragged: shards of 4 and 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    solver = HipPdlpSolver(device=0)
    bidder, model = scenarios.make_batch("nuclear_24h", B, solver)
    solver.solve(model)
    for rank, obj, status, flags, xnan in got:
        np.testing.assert_allclose(obj, model.objective, rtol=1e-9)           # every rank holds ALL objectives
        assert (status == 0).all() and (flags & 1 == 0).all() and not xnan


def _run_bench(extra, timeout=600, nproc=2):
    env = dict(os.environ, DSP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                               # ONE JSON line, from rank 0
    return json.loads(lines[0])


@gpu
def test_bench_launch_contract_with_two_ranks():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` as the driver launches it: metric workload (weak
    scaling: every rank its own 4096 scenarios), the strong-scaling split of BASELINE config 4's day-ahead shape, and the rolling
    double loop of config 4."""
    d = _run_bench(["--steps", "4", "--warmup", "1", "--no-spmv", "--cpu-sample", "0", "--min-time", "0.05", "--streams", "8"])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["rehearsal"] is True and d["scaling"] == "weak"
    assert d["config"]["scenarios"] == 8192 and d["config"]["optimal"] == 8192 and d["value"] > 0 and d["steps"] == 4
    d = _run_bench(["--steps", "2", "--warmup", "1", "--no-spmv", "--cpu-sample", "0", "--min-time", "0", "--streams", "8",
                    "--workload", "wind_battery_48h", "--total", "1025"])
    assert d["scaling"] == "strong" and d["config"]["scenarios"] == 1025 and d["config"]["batch_per_gpu"] == 513
    assert d["config"]["optimal"] == 1025
    d = _run_bench(["--workload", "double_loop", "--total", "65", "--steps", "1", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["config"]["all_optimal"] is True and d["value"] > 0 and d["rehearsal"] is True


@gpu
def test_bench_launch_contract_with_eight_ranks():
    """The launch the scaling bench makes on an 8-GPU node, rehearsed with eight ranks on this box's one GPU (gloo on host copies):
    the metric workload with the stream depth chosen in the warm-up (eight ranks x up to 24 streams, every step an all-gather on
    the one communicator), BASELINE config 4's 8192 scenarios minus one as a ragged strong-scaling split, and config 4 itself -
    the rolling double loop of 8192 plants, 1024 per rank.  Not scaling numbers (`rehearsal`); what is pinned is that the
    first real 8-rank launch cannot fail on plumbing, and that the line proves how many ranks and distinct GPUs took part."""
    import time
    t0 = time.time()
    d = _run_bench(["--steps", "4", "--warmup", "1", "--no-spmv", "--cpu-sample", "0", "--min-time", "0.05", "--batch", "512"], timeout=900, nproc=8)
    assert d["n_gpus"] == 8 and d["world_size"] == 8 and d["dist_world_size"] == 8 and d["rehearsal"] is True and d["scaling"] == "weak"
    assert len(d["rank_devices"]) == 8 and sorted(r["rank"] for r in d["rank_devices"]) == list(range(8)) and d["distinct_gpus"] == 1
    assert d["config"]["scenarios"] == 4096 and d["config"]["optimal"] == 4096 and d["config"]["flagged"] == 0 and d["value"] > 0
    d = _run_bench(["--steps", "2", "--warmup", "1", "--no-spmv", "--cpu-sample", "0", "--min-time", "0", "--streams", "8",
                    "--workload", "wind_battery_48h", "--total", "8191"], timeout=900, nproc=8)
    assert d["scaling"] == "strong" and d["config"]["scenarios"] == 8191 and d["config"]["batch_per_gpu"] == 1024
    assert d["config"]["optimal"] == 8191 and d["dist_world_size"] == 8
    d = _run_bench(["--workload", "double_loop", "--total", "8192", "--steps", "1", "--warmup", "1"], timeout=900, nproc=8)
    assert d["n_gpus"] == 8 and d["config"]["all_optimal"] is True and d["config"]["uncertified_solves"] == 0 and d["dist_world_size"] == 8
    assert "8192 wind+battery plants (1024 per GPU)" in d["config"]["workload"]
    assert time.time() - t0 < 1800
