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
