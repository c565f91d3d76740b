"""Host-side flatten-once layer (dispatches_amd/lp.py): the flattened StandardFormLP equals the block's algebra, the
light presolve only removes rows that can never bind, and mutable bounds stay live after flatten()."""
import numpy as np
import pytest
from scipy.optimize import linprog

from dispatches_amd.lp import LinearBlock, LinExpr


def _solve(lp, lb, ub, rlo, rhi):
    A = lp.csr().toarray()
    eq = np.isfinite(rlo) & (rlo == rhi)
    up = np.isfinite(rhi) & ~eq
    dn = np.isfinite(rlo) & ~eq
    Aub = np.vstack([A[up], -A[dn]]) if (up.any() or dn.any()) else None
    bub = np.concatenate([rhi[up], -rlo[dn]]) if Aub is not None else None
    r = linprog(lp.c, A_ub=Aub, b_ub=bub, A_eq=A[eq] if eq.any() else None, b_eq=rhi[eq] if eq.any() else None,
                bounds=np.stack([lb, ub], 1), method="highs")
    assert r.status == 0, r.message
    return r.fun + lp.c0, r.x


def _random_block(rng, n=9, m=7):
    b = LinearBlock("t")
    xs = [b.var(f"x[{j}]", lb=0.0, ub=float(rng.integers(5, 50)), mutable=(j % 3 == 0), hull=(0.0, 100.0) if j % 3 == 0 else None)
          for j in range(n)]
    rows = []
    for i in range(m):
        cols = rng.choice(n, size=3, replace=False)
        body = LinExpr()
        for j in cols:
            body = body + xs[j] * float(rng.integers(1, 5))
        body = body + float(rng.integers(0, 3))                       # constants move to the bounds
        if i % 3 == 0:
            rows.append(b.equality(f"e[{i}]", body, rhs=float(rng.integers(10, 40))))
        elif i % 3 == 1:
            rows.append(b.constraint(f"le[{i}]", body, hi=float(rng.integers(60, 120))))
        else:
            rows.append(b.constraint(f"never[{i}]", body, hi=1e8))   # can never bind: presolve must drop it
    obj = LinExpr()
    for j in range(n):
        obj = obj + xs[j] * float(rng.normal())
    return b, xs, obj + 3.5


@pytest.mark.parametrize("seed", range(5))
def test_flatten_matches_algebra_and_presolve_is_safe(seed):
    rng = np.random.default_rng(seed)
    b, xs, obj = _random_block(rng)
    full = b.flatten(obj, presolve=False)
    b2, xs2, obj2 = _random_block(np.random.default_rng(seed))
    red = b2.flatten(obj2, presolve=True)
    assert red.m < full.m and all("never" not in r for r in red.row_names)
    assert full.c0 == pytest.approx(3.5) and red.n == full.n
    # the row bodies: A x + moved constants reproduce the original expressions
    x = rng.random(full.n) * 10
    A = full.csr().toarray()
    for i, expr in enumerate(b.row_expr):
        assert A[i] @ x == pytest.approx(sum(v * x[j] for j, v in expr.items()))
    try:
        f_full, _ = _solve(full, *b.current_bounds())
    except AssertionError:
        pytest.skip("random instance infeasible")
    f_red, _ = _solve(red, *b2.current_bounds())
    assert f_red == pytest.approx(f_full, rel=1e-9, abs=1e-9)


def test_mutable_bounds_stay_live_and_immutable_ones_are_protected():
    b = LinearBlock("t")
    x = b.var("x[0]", 0.0, 10.0, mutable=True, hull=(0.0, 20.0))
    y = b.var("y[0]", 0.0, 5.0)
    r = b.constraint("cap[0]", x + y, hi=12.0, mutable=True)
    lp = b.flatten(x * -1.0 + y * -2.0)
    f0, _ = _solve(lp, *b.current_bounds())
    x.setub(20.0)                                    # inside the declared hull: allowed after flatten()
    b.set_row_bounds(r, -np.inf, 18.0)
    f1, sol = _solve(lp, *b.current_bounds())
    assert f1 < f0 and sol[0] + sol[1] <= 18.0 + 1e-9
    with pytest.raises(ValueError):
        y.setub(7.0)                                 # immutable column
    with pytest.raises(ValueError):
        x.setub(25.0)                                # leaves the hull


def test_period_shift_maps_for_rolling_warm_start():
    """Warm start of hour h+1 from hour h: column / row "name[t]" starts from the previous "name[t + shift]"; entries
    without a later period (the last `shift` periods, unindexed names) keep their own value.  The maps must be
    permutation-free index gathers over the flattened LP's own names."""
    from dispatches_amd.hip_solver import period_shift_maps
    from dispatches_amd import scenarios

    class _No:
        def solve(self, *a, **k):
            raise AssertionError

    bidder, model = scenarios.wind_battery_batch(2, 24, _No())
    lp = bidder.real_time_model.lp
    import re
    pat = re.compile(r"^(.*)\[(\d+)\]$")
    for shift in (1, 2):
        cmap, rmap = period_shift_maps(lp, shift)
        assert cmap.shape == (lp.n,) and rmap.shape == (lp.m,)
        for names, mp in ((lp.col_names, cmap), (lp.row_names, rmap)):
            index = {nm: k for k, nm in enumerate(names)}
            moved = 0
            for k, nm in enumerate(names):
                mt = pat.match(nm)
                target = f"{mt.group(1)}[{int(mt.group(2)) + shift}]" if mt else None
                if target in index:
                    assert mp[k] == index[target]
                    moved += 1
                else:
                    assert mp[k] == k
            assert moved > 0
    assert period_shift_maps(lp, 1) is period_shift_maps(lp, 1)            # cached on the LP


def test_presolve_never_lets_a_row_certify_itself():
    """Round-1 advisor repros: bounds propagated from a row are not written into the LP, so that row (or a row that
    depends on it) must survive presolve."""
    import numpy as np
    from dispatches_amd.lp import LinearBlock
    # 1: singleton row x <= 5 is the only upper bound of x
    b = LinearBlock()
    x = b.var("x", 0.0, np.inf)
    b.constraint("cap", x, -np.inf, 5.0)
    lp = b.flatten(x * -1.0)
    assert lp.m == 1 and lp.rhi[0] == 5.0
    # 2: x + y <= 10 with y fixed at 3
    b = LinearBlock()
    x, y = b.var("x", 0.0, np.inf), b.var("y", 3.0, 3.0)
    b.constraint("sum", x + y, -np.inf, 10.0)
    assert b.flatten(x * -1.0).m == 1
    # 3: x - y <= 0 together with y <= 5: neither row may vouch for the other
    b = LinearBlock()
    x, y = b.var("x", 0.0, np.inf), b.var("y", 0.0, np.inf)
    b.constraint("link", x - y, -np.inf, 0.0)
    b.constraint("cap", y, -np.inf, 5.0)
    assert b.flatten(x * -1.0).m == 2
    # a row that other rows + hulls really make redundant is still dropped (the 1e8 battery ramp rows)
    b = LinearBlock()
    s0, s1 = b.var("s0", 0.0, 100.0), b.var("s1", 0.0, np.inf)
    b.constraint("cap", s1, -np.inf, 100.0)
    b.constraint("ramp", s1 - s0, -1e8, 1e8)
    lp = b.flatten(s1 * 1.0)
    assert lp.row_names == ["cap"]


def test_implied_column_ranges_are_capped_along_chains():
    """Ranges stated by bounds are kept; an unbounded column takes the largest |a_ik| range_k / |a_ij| it has to balance; a
    recursion whose columns are all derived cannot multiply its way past 1e6 x the widest stated range."""
    from dispatches_amd.lp import implied_column_ranges
    b = LinearBlock()
    x = b.var("x", 0.0, 50.0)
    y = b.var("y")                                   # unbounded: y = 1e-3 x  ->  range 0.05
    s0 = b.var("s[0]")                               # chain s[t] = 10 s[t-1], s[0] = x  ->  50, 500, ... capped at 5e7
    b.equality("def_y", y - 1e-3 * x, 0.0)
    b.equality("s_start", s0 - x, 0.0)
    prev = s0
    for t in range(1, 12):
        st = b.var(f"s[{t}]")
        b.equality(f"s_rec[{t}]", st - 10.0 * prev, 0.0)
        prev = st
    z = b.var("z", 3.0, 3.0)                         # fixed: nothing to scale
    lp = b.flatten(x + y + prev + z, presolve=False)
    r = dict(zip(lp.col_names, implied_column_ranges(lp)))
    assert r["x"] == 50.0 and r["y"] == pytest.approx(0.05) and r["z"] == 1.0
    assert r["s[0]"] == 50.0 and r["s[3]"] == pytest.approx(5e4)
    assert r["s[11]"] == pytest.approx(5e7) and max(r.values()) <= 5e7 + 1
