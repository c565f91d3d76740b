"""Host-side preparation of the solver (dispatches_amd/csrc/dsp_prepare.hpp) checked on the CPU: the C++ header is
compiled into a small harness (tests/prepare_harness.cpp, g++) and fed the real dispatch LPs.  Covers what dsp_create does
before anything reaches the GPU: preconditioner, step size, register-resident layouts, LDS slot permutation."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("prep") / "prepare_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "prepare_harness.cpp")], check=True)
    return exe


class _NoSolver:
    def solve(self, *a, **k):
        raise AssertionError("not solved here")


def _lp(workload):
    from dispatches_amd import scenarios
    fn, kw = scenarios.WORKLOADS[workload]
    bidder, model = fn(B=2, solver=_NoSolver(), **kw)
    return model.lp


def _run(harness, lp, tmp_path):
    A = sp.csr_matrix((lp.data, lp.indices, lp.indptr), shape=(lp.m, lp.n))
    A.sort_indices()
    path = str(tmp_path / "a.bin")
    with open(path, "wb") as f:
        np.array([lp.m, lp.n, A.nnz], np.int32).tofile(f)
        A.indptr.astype(np.int32).tofile(f)
        A.indices.astype(np.int32).tofile(f)
        A.data.astype(np.float64).tofile(f)
    cpl, rpl = (lp.n + 63) // 64, (lp.m + 63) // 64
    out = subprocess.run([harness, path, str(cpl), str(rpl)], check=True, capture_output=True, text=True).stdout.splitlines()
    res = json.loads(out[0])
    dr = np.array(out[1].split(), float)
    dc = np.array(out[2].split(), float)
    return A, res, dr, dc


@pytest.mark.parametrize("workload", ["wind_battery_24h", "wind_pem_48h", "nuclear_24h", "wind_battery_48h"])
def test_preparation_of_the_dispatch_lps(harness, workload, tmp_path):
    lp = _lp(workload)
    A, res, dr, dc = _run(harness, lp, tmp_path)
    assert res["scale_err"] < 1e-12 and (dr > 0).all() and (dc > 0).all()
    # equilibrated: every non-empty row and column of D_r A D_c has inf-norm within a small factor of 1
    S = sp.diags(dr) @ A @ sp.diags(dc)
    rown = np.abs(S).max(axis=1).toarray().ravel()
    coln = np.abs(S).max(axis=0).toarray().ravel()
    assert 0.1 < rown[rown > 0].min() and rown.max() < 10 and 0.1 < coln[coln > 0].min() and coln.max() < 10
    # the step size rests on this norm: power iteration vs the largest singular value
    smax = np.linalg.svd(S.toarray(), compute_uv=False)[0]
    assert abs(res["norm2"] - smax) <= 2e-3 * smax and res["norm2"] <= smax * (1 + 1e-9)
    # register-resident layout + slot map compute the same A x and A^T y as the CSR
    assert res["product_err"] < 1e-12
    # slot maps: permutations within 32-slot blocks whose 16-lane store groups stay conflict-free
    assert res["bad_perm"] == 0 and res["bad_store"] == 0
    # 11 owned elements per lane (48 h): the kernel takes ONE store address per buffer, so every 64-block has the same map
    assert res["shared"] == (workload == "wind_battery_48h") and res["bad_shared"] == 0
    for key in ("conflicts_x", "conflicts_y"):
        ident, rot, final = res[key]
        assert final <= rot <= ident
    assert res["search_ms"] < 2000


def _registered_lps():
    """(name, lp) of every LP the reference workflows solve at the benchmark horizons: day-ahead bidding models of the
    BASELINE workloads, their real-time bidding models, the 4 h tracking model."""
    from dispatches_amd import scenarios
    from dispatches_amd.workflow import Tracker
    out = []
    rt_done = set()
    for wl, (fn, kw) in scenarios.WORKLOADS.items():
        bidder, model = fn(B=2, solver=_NoSolver(), **kw)
        out.append((wl, model.lp))
        fam = wl.rsplit("_", 1)[0]
        if fam not in rt_done:
            rt_done.add(fam)
            out.append((fam + " real-time", bidder.real_time_model.lp))
    bidder, _ = scenarios.wind_battery_batch(2, 24, _NoSolver())
    tr = Tracker(tracking_model_object=bidder.bidding_model_object, tracking_horizon=4, n_tracking_hour=1, solver=_NoSolver())
    out.append(("wind_battery tracker", tr.model.lp))
    return out


def test_reference_lps_hit_a_register_resident_specialisation(harness, tmp_path):
    """Every LP of the reference workflows must keep hitting a register-resident specialisation compiled for it
    (csrc/dsp_kernels.hip: DSP_MATREG_SHAPES); a change of the flattening that alters the per-slot widths would silently
    fall back to the 3-4x slower LDS-matrix kernel."""
    import re
    src = open(os.path.join(ROOT, "dispatches_amd", "csrc", "dsp_kernels.hip")).read()
    table = src[src.index("#define DSP_MATREG_SHAPES"):]
    table = table[:table.index("\n\n")]
    shapes = {(int(a), int(b), int(c, 16), int(d, 16), e == "true")
              for a, b, c, d, e in re.findall(r"X\((\d+), (\d+), (0x[0-9a-f]+)u, (0x[0-9a-f]+)u, (true|false)\)", table)}
    assert len(shapes) >= 8
    for name, lp in _registered_lps():
        _, res, _, _ = _run(harness, lp, tmp_path)
        key = ((lp.n + 63) // 64, (lp.m + 63) // 64, res["pack_c"], res["pack_r"], res["long_c"] + res["long_r"] > 0)
        assert key in shapes, (name, key[:2], hex(key[2]), hex(key[3]), key[4])


@pytest.fixture(scope="module")
def fused_harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fused") / "fused_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "fused_harness.cpp")], check=True)
    return exe


def _write_csr(A, path):
    with open(path, "wb") as f:
        np.array([A.shape[0], A.shape[1], A.nnz], np.int32).tofile(f)
        A.indptr.astype(np.int32).tofile(f)
        A.indices.astype(np.int32).tofile(f)
        A.data.astype(np.float64).tofile(f)


def test_fused_iteration_plan_carries_the_two_level_throughput_basis(fused_harness, tmp_path):
    """The price-taker LP with the accumulated throughput as 3 node values + local deviations (flowsheets/price_taker.py,
    throughput="two_level"): 4 long columns (the nodes + the battery's power), rows of up to 6 entries, otherwise banded - the plan
    of the one-launch iteration exists and one iteration run tile by tile as the kernel's stages do reproduces the plain step."""
    from dispatches_amd import scenarios
    _, model = scenarios.price_taker_batch(1344, 1, _NoSolver(), throughput="two_level")
    lp = model.lp
    A = sp.csr_matrix((lp.data, lp.indices, lp.indptr), shape=(lp.m, lp.n))
    A.sort_indices()
    path = str(tmp_path / "a.bin")
    _write_csr(A, path)
    res = json.loads(subprocess.run([fused_harness, path, "250", "8"], check=True, capture_output=True, text=True).stdout)
    assert res["ntile"] == -(-lp.m // 250) and res["long_cols"] == 4 and res["wr"] == 6 and res["wc"] == 4
    assert res["ny"] <= 250 + 16 and res["nxb"] <= 250 + 16
    assert res["out_of_range"] == 0 and res["uninitialised"] == 0 and res["missing"] == 0
    assert res["err_x"] < 1e-12 and res["err_y"] < 1e-12 and res["err_lp"] < 1e-8


@pytest.mark.parametrize("T,rows_per_tile", [(168, 768), (336, 128), (1000, 768), (8736, 768)])
def test_fused_iteration_plan_on_the_price_taker_lps(fused_harness, tmp_path, T, rows_per_tile):
    """The one-launch iteration of the streaming PDLP (dsp_stream.hip: k_fused) cuts the banded multi-period LP into tiles with
    halo columns / rows and carries the design column through per-tile partial sums.  The harness builds ELLs and plan with the
    library's own host code and runs one iteration tile by tile as the kernel's stages do: same iterates as the plain step on the
    CSR, every staged slot inside its buffer and written before it is read, every row and column written exactly once."""
    from dispatches_amd import scenarios
    _, model = scenarios.price_taker_batch(T, 1, _NoSolver(), throughput="chain")
    lp = model.lp
    A = sp.csr_matrix((lp.data, lp.indices, lp.indptr), shape=(lp.m, lp.n))
    A.sort_indices()
    path = str(tmp_path / "a.bin")
    _write_csr(A, path)
    res = json.loads(subprocess.run([fused_harness, path, str(rows_per_tile), "8"], check=True, capture_output=True, text=True).stdout)
    assert res["ntile"] == -(-lp.m // rows_per_tile) and res["long_cols"] == 1          # the nameplate-power design column
    assert res["wc"] == 4 and res["wr"] == 4                    # the long column must not pad every other vector to the cap
    assert res["ny"] <= rows_per_tile + 16 and res["nxb"] <= rows_per_tile + 16          # halo = one period each side
    assert res["out_of_range"] == 0 and res["uninitialised"] == 0 and res["missing"] == 0
    assert res["err_x"] < 1e-12 and res["err_y"] < 1e-12 and res["err_lp"] < 1e-8


def test_fused_iteration_plan_with_long_columns_at_index_zero(fused_harness, tmp_path):
    """The nuclear price-taker LP's design variables are columns 0 .. 2 - long columns INSIDE the hull of the first tiles.  The row
    ELL's padding entries used to keep index 0, i.e. the hull slot of a column no phase of the kernel writes: 0 x stale LDS bits
    (round-3 advisor finding).  The harness poisons every staged slot with NaN and multiplies padding entries like the kernel does."""
    from dispatches_amd import scenarios
    _, model = scenarios.nuclear_price_taker_batch(720, 1, _NoSolver())
    lp = model.lp
    A = sp.csr_matrix((lp.data, lp.indices, lp.indptr), shape=(lp.m, lp.n))
    A.sort_indices()
    path = str(tmp_path / "a.bin")
    _write_csr(A, path)
    res = json.loads(subprocess.run([fused_harness, path, "250", "8"], check=True, capture_output=True, text=True).stdout)
    assert res["ntile"] > 0 and res["long_cols"] == 3
    assert res["out_of_range"] == 0 and res["uninitialised"] == 0 and res["missing"] == 0
    assert res["err_x"] < 1e-12 and res["err_y"] < 1e-12 and res["err_lp"] < 1e-8


@pytest.fixture(scope="module")
def lane_harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lane") / "lane_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "lane_harness.cpp")], check=True)
    return exe


def test_lane_walk_with_one_register_set(tmp_path):
    """The measurement variant of the walk (DSP_LANE_NO_PREFETCH: a unit's rows requested right before its arithmetic; the probe build
    of tools/gpu_lane_probe.sh uses it) computes the same iteration."""
    exe = str(tmp_path / "lane_harness_np")
    subprocess.run(["g++", "-O2", "-std=c++17", "-DDSP_LANE_NO_PREFETCH", "-o", exe, os.path.join(ROOT, "tests", "lane_harness.cpp")], check=True)
    lp = _lane_case("wind_battery_two_level", 336).lp
    A = sp.csr_matrix((lp.data, lp.indices, lp.indptr), shape=(lp.m, lp.n))
    A.sort_indices()
    path = str(tmp_path / "a.bin")
    _write_csr(A, path)
    res = json.loads(subprocess.run([exe, path, "24", "16"], check=True, capture_output=True, text=True).stdout)
    assert res["ok"] and res["missing"] == 0 and res["nan_partials"] == 0
    assert max(res["err_x"], res["err_y"], res["err_xp"], res["err_yp"]) < 1e-11 and res["err_sums"] < 1e-11


def _lane_case(name, T):
    from dispatches_amd import scenarios
    if name == "wind_battery_chain":
        return scenarios.price_taker_batch(T, 1, _NoSolver(), throughput="chain")[1]
    if name == "wind_battery_two_level":
        return scenarios.price_taker_batch(T, 1, _NoSolver())[1]
    if name == "pem_chain":
        return scenarios.pem_price_taker_batch(T, 1, _NoSolver(), inputs="rts303")[1]
    if name == "pem_two_level":
        return scenarios.pem_price_taker_batch(T, 1, _NoSolver(), inputs="rts303", throughput="two_level", coarse_nodes=3)[1]
    return scenarios.nuclear_price_taker_batch(T, 1, _NoSolver())[1]


@pytest.mark.parametrize("family,T,rows,long_cols,widths", [
    ("wind_battery_chain", 336, 12, 1, (4, 4, 4)), ("wind_battery_chain", 336, 100, 1, (4, 4, 4)),
    ("wind_battery_two_level", 1344, 24, 4, (4, 4, 4)), ("wind_battery_two_level", 1344, 96, 4, (4, 4, 4)),
    ("pem_chain", 1000, 24, 3, (4, 4, 4)), ("pem_two_level", 1000, 48, 6, (4, 4, 8)),
    ("nuclear", 720, 16, 3, (4, 8, 4)), ("nuclear", 720, 96, 3, (4, 8, 4))])
def test_lane_form_of_the_streaming_iteration(lane_harness, tmp_path, family, T, rows, long_cols, widths):
    """The lane-per-scenario form of the streaming iteration (round 4; csrc/dsp_lane_plan.hpp, dsp_lane_tile.hpp, dsp_stream_lane.hip):
    the harness builds records, tiles and walks with the library's host code and runs the SAME per-lane routine the kernel runs
    (LaneTile::run: rings poisoned with NaN, three scenarios with their own step sizes and anchor weights) for the plain iteration,
    the check iteration and the reduced costs: iterates, x+ / y+, the thirteen check sums and the long columns' partial sums of
    A^T y against their definitions on the CSR; every row and column is written."""
    lp = _lane_case(family, T).lp
    A = sp.csr_matrix((lp.data, lp.indices, lp.indptr), shape=(lp.m, lp.n))
    A.sort_indices()
    path = str(tmp_path / "a.bin")
    _write_csr(A, path)
    res = json.loads(subprocess.run([lane_harness, path, str(rows)], check=True, capture_output=True, text=True).stdout)
    assert res["ok"], res
    assert res["long_cols"] == long_cols and (res["wc"], res["wr"], res["nlp"]) == widths
    assert res["ntile"] == -(-lp.m // rows) and res["ring"] <= 16
    assert res["missing"] == 0 and res["nan_partials"] == 0
    assert max(res["err_x"], res["err_y"], res["err_xp"], res["err_yp"]) < 1e-11 and res["err_sums"] < 1e-11
    assert res["err_lp"] < 1e-7                                    # (absolute, on sums of thousands of unscaled terms)
    assert res["nunit"] <= 2.5 * -(-lp.m // 4)                     # the walks keep their units reasonably full (12-row tiles, ring of 8: 2.3 x)


@pytest.mark.parametrize("seed", range(12))
def test_lane_form_on_random_banded_matrices(lane_harness, tmp_path, seed):
    """Property test of plan + walk: random banded matrices - n != m, 1 .. 4 (or .. 7) short entries per row inside a band around
    the diagonal, empty rows and empty columns, 0 .. 6 long columns of random density, tiles of 8 .. 96 rows - through the harness:
    whenever the planner accepts the matrix, one iteration / check / reduced-cost pass of the per-lane routine reproduces the plain
    formulas on the CSR and writes every row and column; a matrix it cannot schedule is refused, never mis-computed."""
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.integers(120, 420))
    n = int(m * rng.uniform(0.7, 1.5))
    band = int(rng.integers(2, 10))
    wmax = 7 if seed % 4 == 3 else 4
    rows, cols, vals = [], [], []
    for i in range(m):
        if rng.uniform() < 0.05:
            continue                                            # an empty row
        centre = i * n / m
        k = int(rng.integers(1, wmax + 1))
        cand = np.unique(np.clip(np.round(centre + rng.integers(-band, band + 1, size=k)), 0, n - 1).astype(int))
        for j in cand:
            rows.append(i); cols.append(int(j)); vals.append(float(rng.uniform(0.2, 2.0) * rng.choice([-1.0, 1.0])))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n)).tolil()
    dead = rng.choice(n, size=max(1, n // 25), replace=False)     # a few empty columns
    A[:, dead] = 0.0
    nlong = int(rng.integers(0, 7))
    for j in rng.choice(np.setdiff1d(np.arange(n), dead), size=nlong, replace=False):
        hit = rng.uniform(size=m) < rng.uniform(0.3, 1.0)
        hit[:12] = True                                           # more than 8 entries: a long column whatever the draw
        A[np.flatnonzero(hit), j] = rng.uniform(0.1, 1.0, size=int(hit.sum()))
    A = sp.csr_matrix(A)
    A.eliminate_zeros()
    A.sort_indices()
    # columns other than the planted ones may have collected more than 8 entries (dense bands): they become long columns too
    path = str(tmp_path / "a.bin")
    _write_csr(A, path)
    rows_per_tile = int(rng.choice([8, 12, 20, 40, 96]))
    ring_min = (8, 16)[seed % 2]                                # (the device starts from 16; the records are packed for the tiling's ring)
    res = json.loads(subprocess.run([lane_harness, path, str(rows_per_tile), str(ring_min)], check=True, capture_output=True, text=True).stdout)
    if not res["ok"]:
        assert res["why"] in ("plan", "tiles")
        collen = np.diff(sp.csc_matrix(A).indptr)
        assert res["why"] == "tiles" or (collen > 8).sum() > 8 or wmax > 4, (res, int((collen > 8).sum()))
        return
    assert res["missing"] == 0 and res["nan_partials"] == 0, res
    assert max(res["err_x"], res["err_y"], res["err_xp"], res["err_yp"]) < 1e-11 and res["err_sums"] < 1e-11, res
    assert res["err_lp"] < 1e-9, res


def test_lane_plan_refuses_matrices_that_are_not_banded(lane_harness, tmp_path):
    """The parallel-prefix form of the accumulator reaches across the whole horizon: no walk fits the ring budget, the handle keeps
    the two-launch form."""
    from dispatches_amd import scenarios
    _, model = scenarios.price_taker_batch(1000, 1, _NoSolver(), throughput="scan")
    lp = model.lp
    A = sp.csr_matrix((lp.data, lp.indices, lp.indptr), shape=(lp.m, lp.n))
    A.sort_indices()
    path = str(tmp_path / "a.bin")
    _write_csr(A, path)
    res = json.loads(subprocess.run([lane_harness, path, "64"], check=True, capture_output=True, text=True).stdout)
    assert not res["ok"]


def test_fused_plan_refuses_matrices_that_are_not_banded(fused_harness, tmp_path):
    """A matrix whose products reach across the whole index range (here: the parallel-prefix form of the throughput accumulator,
    and a random sparse matrix) gets no plan: the two-launch form stays."""
    from dispatches_amd import scenarios
    _, model = scenarios.price_taker_batch(1000, 1, _NoSolver(), throughput="scan")
    lp = model.lp
    rng = np.random.default_rng(0)
    for A in (sp.csr_matrix((lp.data, lp.indices, lp.indptr), shape=(lp.m, lp.n)),
              sp.random(9000, 9000, density=3.0 / 9000, random_state=rng, format="csr") + sp.eye(9000, format="csr")):
        A = sp.csr_matrix(A)
        A.sort_indices()
        path = str(tmp_path / "b.bin")
        _write_csr(A, path)
        res = json.loads(subprocess.run([fused_harness, path, "768", "8"], check=True, capture_output=True, text=True).stdout)
        assert res["ntile"] == 0


def test_xcd_major_workgroup_order_is_a_bijection(fused_harness):
    """The renumbering of the fused launch's workgroups (dsp_prepare.hpp::fused_workgroup, the function the kernel calls) for every
    shape up to 40 scenario groups x 70 tiles: each (tile, group) exactly once, all groups of a renumbered tile on one XCD,
    nothing beyond the plan (the first version ran on into tile ids that do not exist: a memory fault on the GPU)."""
    res = json.loads(subprocess.run([fused_harness, "--workgroup-order"], check=True, capture_output=True, text=True).stdout)
    assert res == {"ok": True}
