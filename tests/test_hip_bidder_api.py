"""GPU: the reference-level call `Bidder.compute_day_ahead_bids` / `compute_real_time_bids` through the device-side path (objective
vectors formed on the device, solution left there, roundings + sorts of the bid assembly as tensor operations) against the host
path (dense objective upload, full download, numpy assembly): the same bid dictionaries, bit for bit, at the metric batch.
Goldens G1 / G2 through this path: tests/test_hip_parity.py::test_golden_self_schedule_and_bid_curves (HipPdlpSolver's default)."""
import numpy as np
import pytest

gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("B", [4096, 257])
def test_device_bid_assembly_reproduces_the_host_path(B):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    out = []
    for lazy in (True, False):
        solver = HipPdlpSolver(device=0, lazy_solution=lazy)
        bidder, model = scenarios.wind_battery_batch(B, 24, solver)
        da = bidder.compute_day_ahead_bids("2020-01-03", 0)
        assert (model.status == 0).all()
        assert (model._x is None) == lazy                          # the device path never downloaded the whole solution
        obj = model.objective.copy()
        rt = bidder.compute_real_time_bids("2020-01-03", 2, realized_day_ahead_prices=None, realized_day_ahead_dispatches=None)
        assert (bidder.real_time_model.status == 0).all()
        out.append((da, rt, obj, model))
    (da0, rt0, obj0, m0), (da1, rt1, obj1, m1) = out
    np.testing.assert_array_equal(obj0, obj1)                     # the device-formed objective vectors are the host's, bit for bit
    assert da0 == da1 and rt0 == rt1
    np.testing.assert_array_equal(m0.x, m1.x)                     # and the lazily fetched solution is the one the eager path stored
    assert max(len(v["309_WIND_1"]["p_cost"]) for v in da0.values()) > 2


@gpu
@pytest.mark.parametrize("B,T,terms", [(4096, 24, 0), (257, 48, 2), (1, 4, 1), (1500, 24, 2), (9000, 3, 0)])
def test_bid_points_kernel_is_the_tensor_path(B, T, terms):
    """dsp_bid_points (csrc/dsp_bids.hip: exact cents, LDS bitonic sort, ordered compaction, one launch) against the tensor statement
    of the same arithmetic (workflow/bid_curves.py, itself pinned to Python's round() and to the numpy path on the CPU): exact ties,
    duplicates, negative and non-finite numbers, failed scenarios, one- and two-term power expressions, batches that are not a power
    of two and beyond 64 KB of LDS.  Integers compared exactly."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from dispatches_amd.hip_solver import DeviceSolution
    from dispatches_amd.workflow import bid_curves as bc
    rng = np.random.default_rng(B + T)
    n = 50
    x = rng.uniform(-5, 220, (B, n))
    dec = 10.0 ** rng.integers(0, 4, (B, n))
    x = np.round(x * dec) / dec                                          # many duplicates and near-ties
    x[rng.random((B, n)) < 0.2] = 0.0
    x[::7, 3] = 12.345
    x[::5, 4] = 0.125                                                    # an exact tie
    x[::11, 5] = 2.675
    if B > 20:
        x[13, :] = np.nan
        x[17, 3] = np.inf
    price = np.round(rng.uniform(-30, 300, (B, T + 2)), rng.integers(0, 5))
    price[::3, 1] = 20.005
    if B > 20:
        price[19, 0] = np.nan
    ok = rng.random(B) > 0.05
    cols = rng.integers(0, n, (T, 2)).astype(np.int32)
    cols[:6, 0] = [3, 4, 5, 3, 4, 5][:min(6, T)] if T >= 6 else cols[:6, 0]
    vals = np.where(rng.random((T, 2)) < 0.5, 1e-3, rng.uniform(0.5, 2.0, (T, 2)))
    k0 = np.round(rng.uniform(-1, 1, T), 3)
    dev = torch.device("cuda", 0)
    xd, pd_ = torch.as_tensor(x, device=dev), torch.as_tensor(price, device=dev)
    for p_min, okm in ((0.0, None), (10.126, ok)):
        sol = DeviceSolution({"x": xd, "y": xd}, 0, dev)
        counts, pc, cc = sol.bid_points(cols, vals, k0, terms, pd_, p_min, okm)
        if terms == 0:
            power = xd[:, torch.as_tensor(cols[:, 0].astype(np.int64), device=dev)]
        else:
            terms_ = xd[:, torch.as_tensor(cols.reshape(-1).astype(np.int64), device=dev)].reshape(B, T, 2) * torch.as_tensor(vals, device=dev)
            power = (terms_[:, :, 0] if terms == 1 else terms_[:, :, 0] + terms_[:, :, 1]) + torch.as_tensor(k0, device=dev)
        okd = None if okm is None else torch.as_tensor(okm, device=dev)
        packed, want_counts = bc.compact(torch, *bc.sorted_pairs(torch, power, pd_[:, :T], p_min, okd))
        np.testing.assert_array_equal(counts, want_counts)
        ends = np.cumsum(want_counts)
        for t in range(T):
            k = int(want_counts[t])
            np.testing.assert_array_equal(pc[t, :k], packed[ends[t] - k:ends[t], 0])
            np.testing.assert_array_equal(cc[t, :k], packed[ends[t] - k:ends[t], 1])
    assert counts.max() > 0 or B == 1
