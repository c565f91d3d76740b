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
