"""CPU tests of the banded solves of the interior-point form (csrc/dsp_ipm_seq.hpp: the per-lane arithmetic the kernels of csrc/dsp_ipm.hip
run): the sequential walks and the TIME-PARALLEL form (partitions of the horizon, spikes, block-tridiagonal system of the separators,
border sums and corrections) against a banded Cholesky in long double on random symmetric positive definite band matrices whose entries
span four decades.  tests/ipm_par_harness.cpp runs the arithmetic in the kernels' order (one state per lane and partition, the border
sums as partial sums of four waves).  GPU side: tests/test_hip_ipm.py (both forms on the same LPs)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ipm") / "ipm_par_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "ipm_par_harness.cpp")], check=True)
    return exe


@pytest.mark.parametrize("m,W,Lp,parts", [
    (400, 6, 50, 8), (400, 8, 50, 8),            # even partitions
    (1010, 6, 337, 3), (1010, 8, 127, 8),        # the one-week price-taker LP's rows: the automatic geometry (3) and a finer one
    (333, 6, 40, 9),                             # ragged tail (13 rows: one more than 2 W - a partition of its own)
    (330, 6, 40, 8),                             # tail of 10 rows: joins the partition before it
    (2000, 8, 17, 117), (500, 6, 13, 38),        # partitions barely longer than two separators
    (5000, 8, 79, 64),                           # 64 partitions (the year-long LPs' count)
])
def test_time_parallel_banded_solve_equals_the_sequential_one(harness, m, W, Lp, parts):
    res = json.loads(subprocess.run([harness, str(m), str(W), str(Lp), "11"], check=True, capture_output=True, text=True).stdout)
    assert res["P"] == parts, res
    assert 0 <= res["err_seq"] < 1e-11, res
    assert 0 <= res["err_par"] < 1e-11, res


def test_bench_roofline_object_of_the_interior_point_line():
    """bench.py's `roofline` object for a --solve line of the interior-point form: the banded solve's algorithmic bytes 8 Bp m (4 W + 9), its
    duration from the newest committed kernel-trace summary of this round at that batch, the HBM bytes of the same five kernels from the
    committed counter summary (FETCH_SIZE doubled as the guide prescribes for gfx950).  Without a committed summary for the batch, or with
    one partition (the sequential walks), the bandwidth fields stay null instead of being guessed."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    m, W = 52418, 6
    r = bench._ipm_roofline(256, m, 64)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBS
    assert r["algorithmic_bytes_per_solve"] == 8 * 256 * m * (4 * W + 9)
    assert set(r["kernel_us"]) == {"ForwardBody", "BackwardBody", "k_ipm_border_dot", "k_ipm_red_solve", "k_ipm_border_apply"}
    assert abs(r["achieved"] - r["algorithmic_bytes_per_solve"] / (r["solve_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert 0.3 < r["frac"] < 0.8 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["archived_from"].endswith("_ipm_kernel_stats_T8736_B256.csv") and os.path.exists(os.path.join(ROOT, "profiles", r["archived_from"]))
    assert 0.9 < r["traffic_over_algorithmic"] < 1.1 and os.path.exists(os.path.join(ROOT, "profiles", r["traffic_from"]))
    none = bench._ipm_roofline(192, m, 64)                       # no committed summary at this batch
    assert none["achieved"] is None and none["frac"] is None and none["traffic"] is None
    seq = bench._ipm_roofline(256, m, 1)
    assert seq["bound"] == "latency" and seq["frac"] is None


@pytest.mark.parametrize("T", [168, 336])
def test_numpy_statement_of_the_interior_point_form_against_highs(T):
    """tools/ipm_lab.py is the numpy statement of the method csrc/dsp_ipm.hip implements (Mehrotra predictor-corrector on [A | -I], banded
    normal matrix in the natural row order + Sherman-Morrison-Woodbury for the wide columns, refinement on the full normal equations, the
    streaming path's KKT test on the unscaled problem) with the product's settings (0.99 to the boundary, sigma >= 0.05): all 16 members of
    the price-taker family (reference wind_battery_LMP.py:172-269 at one and two weeks) to 1e-6 of HiGHS on the un-reduced LP, in at most
    100 Newton iterations, inside the bounds."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import ipm_lab
    import stream_lab as lab
    worst = 0
    for member in range(16):
        P = lab.build(T, member, None, "chain")
        ref = lab.highs(P)[0]
        X, Y, it, done = ipm_lab.solve(P, colscale=lab.physical_scales(P, T))
        obj = float(P["c"] @ X + P["c0"])
        assert done and it <= 100, (member, it)
        assert abs(obj - ref) <= 1e-6 * max(1.0, abs(ref)), (member, obj, ref)
        assert (X >= P["lb"] - 1e-7 * np.maximum(1.0, np.abs(P["lb"]))).all() and (X <= P["ub"] + 1e-7 * np.maximum(1.0, np.abs(P["ub"]))).all()
        worst = max(worst, it)
    assert worst >= 10                                            # (an interior-point method ran: not a presolve accident)
