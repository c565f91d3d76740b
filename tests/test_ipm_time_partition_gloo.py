"""Rehearsal (CPU, gloo, world size 2) of the multi-GPU split DESIGN.md section 6 proposes for SMALL batches of year-long LPs: the interior-point
form's time partitions of the horizon go to different ranks instead of the scenarios (60 LPs over 8 GPUs leave 8 of 64 lanes per wave busy).

What is rehearsed is the data that crosses ranks and that the result does not depend on the split - in numpy, with the elimination order of
csrc/dsp_ipm_seq.hpp ("all interiors at once, then the separators"; per-lane arithmetic of the product: tests/ipm_par_harness.cpp):

    per factorisation   every partition eliminates its interior and contributes to the separators at its two ends: a W x W Schur update to
                        each and the W x W coupling block between them        -> ONE all-gather of 3 W^2 doubles per partition and lane
    per solve           the interiors' forward solves contribute W numbers to each of the two separators  -> ONE all-gather of 2 W doubles
                        per partition and lane; the reduced block-tridiagonal system is solved redundantly on every rank; the
                        back-substitution of an interior needs only its own two separators' solution
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _banded_spd(m, W, lanes, seed):
    """[lanes, m, m] symmetric positive definite matrices of half-bandwidth W whose entries span four decades, and right-hand sides"""
    rng = np.random.default_rng(seed)
    A = np.zeros((lanes, m, m))
    for k in range(1, W + 1):
        v = rng.standard_normal((lanes, m - k)) * 10.0 ** rng.uniform(-2, 2, (lanes, m - k))
        i = np.arange(m - k)
        A[:, i + k, i] = v
        A[:, i, i + k] = v
    A[:, np.arange(m), np.arange(m)] = np.abs(A).sum(2) + 10.0 ** rng.uniform(-2, 2, (lanes, m))
    return A, rng.standard_normal((lanes, m))


def _geometry(m, W, P):
    Lp = -(-m // P)
    parts = []
    for p in range(P):
        r0, r1 = p * Lp, min(m, (p + 1) * Lp)
        last = p == P - 1
        parts.append(dict(I=np.arange(r0, r1 if last else r1 - W), S=None if last else np.arange(r1 - W, r1)))
    return parts


def _factor_partition(A, parts, p):
    """eliminate the interior of partition p: -> (Cholesky factor of B_II, C_left, C_right, Schur updates to the left / own separator, coupling)"""
    I = parts[p]["I"]
    L = np.linalg.cholesky(A[:, I[:, None], I[None, :]])
    sl = parts[p - 1]["S"] if p > 0 else None
    sr = parts[p]["S"]
    out = dict(L=L)
    solve = lambda M: np.linalg.solve(A[:, I[:, None], I[None, :]], M)
    W = len(sr) if sr is not None else len(sl)
    lanes = A.shape[0]
    Cl = A[:, I[:, None], sl[None, :]] if sl is not None else np.zeros((lanes, len(I), W))
    Cr = A[:, I[:, None], sr[None, :]] if sr is not None else np.zeros((lanes, len(I), W))
    Gl, Gr = solve(Cl), solve(Cr)
    out.update(Cl=Cl, Cr=Cr, Gl=Gl, Gr=Gr)
    # what leaves the rank: 3 W x W blocks per partition and lane
    out["msg"] = np.stack([np.einsum("lia,lib->lab", Cl, Gl), np.einsum("lia,lib->lab", Cr, Gr), np.einsum("lia,lib->lab", Cr, Gl)], 1)   # [lanes, 3, W, W]
    return out


def _solve_split(A, r, W, P, rank, world):
    """the solve as rank `rank` of `world` performs it - a generator: it YIELDS the array it contributes to an all-gather and is SENT the
    list of every rank's array; returns (x with its own rows filled, its partitions, the geometry, bytes per factor / solve message)"""
    lanes, m = r.shape
    parts = _geometry(m, W, P)
    mine = [p for p in range(P) if p * world // P == rank]
    F = {p: _factor_partition(A, parts, p) for p in mine}
    # ---- factorisation: one all-gather of the partitions' separator blocks -----------------------------------------------------------
    msgs = np.concatenate((yield np.stack([F[p]["msg"] for p in mine], 0)), 0)          # [P, lanes, 3, W, W]
    bytes_factor = msgs[0].nbytes
    nb = P - 1
    D = np.stack([A[:, parts[s]["S"][:, None], parts[s]["S"][None, :]] for s in range(nb)], 0)            # [nb, lanes, W, W]
    O = np.zeros((nb, lanes, W, W))                                                                       # coupling S_s - S_(s-1)
    for p in range(P):
        if p < nb:
            D[p] -= msgs[p][:, 1]
        if p > 0:
            D[p - 1] -= msgs[p][:, 0]
            if p < nb:
                O[p] = -msgs[p][:, 2]
    # ---- solve: interiors forward, one all-gather of their 2 W numbers per partition, reduced system redundantly -------------------
    y = {p: np.linalg.solve(A[:, parts[p]["I"][:, None], parts[p]["I"][None, :]], r[:, parts[p]["I"], None])[..., 0] for p in mine}
    contrib = np.stack([np.stack([np.einsum("lia,li->la", F[p]["Cl"], y[p]), np.einsum("lia,li->la", F[p]["Cr"], y[p])], 1) for p in mine], 0)
    allc = np.concatenate((yield contrib), 0)                                            # [P, lanes, 2, W]
    bytes_solve = allc[0].nbytes
    rs = np.stack([r[:, parts[s]["S"]] for s in range(nb)], 0)
    for p in range(P):
        if p < nb:
            rs[p] -= allc[p][:, 1]
        if p > 0:
            rs[p - 1] -= allc[p][:, 0]
    # block-tridiagonal system of the separators, dense per lane (63 x 6 in the product: a latency chain every rank repeats)
    xs = np.zeros((nb, lanes, W))
    for l in range(lanes):
        R = np.zeros((nb * W, nb * W))
        for s in range(nb):
            R[s * W:(s + 1) * W, s * W:(s + 1) * W] = D[s, l]
            if s > 0:
                R[s * W:(s + 1) * W, (s - 1) * W:s * W] = O[s, l]
                R[(s - 1) * W:s * W, s * W:(s + 1) * W] = O[s, l].T
        xs[:, l] = np.linalg.solve(R, rs[:, l].reshape(-1)).reshape(nb, W)
    # ---- back-substitution of the own interiors ----------------------------------------------------------------------------------------
    x = np.zeros((lanes, m))
    for p in mine:
        xi = y[p].copy()
        if p > 0:
            xi -= np.einsum("lia,la->li", F[p]["Gl"], xs[p - 1])
        if p < nb:
            xi -= np.einsum("lia,la->li", F[p]["Gr"], xs[p])
            x[:, parts[p]["S"]] = xs[p]
        x[:, parts[p]["I"]] = xi
    return x, mine, parts, bytes_factor, bytes_solve


def _drive(gens):
    """all ranks in one process: advance every generator to its next all-gather, hand each the list of all contributions"""
    msgs = [next(g) for g in gens]
    results = [None] * len(gens)
    while any(r is None for r in results):
        nxt = []
        for k, g in enumerate(gens):
            try:
                nxt.append(g.send(list(msgs)))
            except StopIteration as stop:
                results[k] = stop.value
        msgs = nxt
    return results


def test_time_partition_split_single_process():
    """the arithmetic itself, `world` ranks simulated in one process: the split solve equals the dense solve whatever the split"""
    m, W, P, lanes = 230, 6, 8, 3
    A, r = _banded_spd(m, W, lanes, 5)
    ref = np.linalg.solve(A, r[..., None])[..., 0]
    for world in (1, 2, 4):
        x = np.zeros_like(ref)
        for xr, mine, parts, bf, bs in _drive([_solve_split(A, r, W, P, rank, world) for rank in range(world)]):
            for p in mine:
                rows = np.concatenate([parts[p]["I"], parts[p]["S"]]) if parts[p]["S"] is not None else parts[p]["I"]
                x[:, rows] = xr[:, rows]
            assert bf == lanes * 3 * W * W * 8 and bs == lanes * 2 * W * 8
        assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max(), world


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, W, P, lanes = 410, 6, 8, 4
        A, r = _banded_spd(m, W, lanes, 11)
        sent = []

        def gather(local):
            t = torch.as_tensor(np.ascontiguousarray(local))
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            sent.append(t.numel() * 8)
            return [o.numpy() for o in out]
        gen = _solve_split(A, r, W, P, rank, world)
        msg = next(gen)
        while True:
            try:
                msg = gen.send(gather(msg))
            except StopIteration as stop:
                x, mine, parts, bf, bs = stop.value
                break
        rows = np.concatenate([np.concatenate([parts[p]["I"], parts[p]["S"]]) if parts[p]["S"] is not None else parts[p]["I"] for p in mine])
        q.put((rank, rows, x[:, rows], sent))
    finally:
        dist.destroy_process_group()


def test_time_partition_split_world2_gloo():
    """two processes, the partitions of the horizon split between them, two all-gathers (one per factorisation, one per solve): together they
    hold the dense solution; the bytes a rank sends are 3 W^2 resp. 2 W doubles per partition and lane"""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m, W, P, lanes = 410, 6, 8, 4
    A, r = _banded_spd(m, W, lanes, 11)
    ref = np.linalg.solve(A, r[..., None])[..., 0]
    x = np.full_like(ref, np.nan)
    for rank, rows, xr, sent in got:
        x[:, rows] = xr
        assert sent == [(P // world) * lanes * 3 * W * W * 8, (P // world) * lanes * 2 * W * 8], sent
    assert np.isfinite(x).all() and np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()
