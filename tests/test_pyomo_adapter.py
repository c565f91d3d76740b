"""The live-Pyomo flattener (dispatches_amd/pyomo_adapter.py) exercised with stand-in objects that
mimic the part of Pyomo's API it uses (Pyomo itself is absent from the build container): a block with Var / Constraint /
Objective data objects, mutable Params and ``generate_standard_repn``.  The stand-in model is the 4-period wind + battery
tracking LP written the way the reference writes it in Pyomo (per-period blocks, fixed design variables, mutable
capacity-factor Params); the flattened LP must have the same optimum as the product's LinearBlock formulation and the
oracle, must refresh when Params / fixed values change, and must refuse changes that alter the matrix."""
import numpy as np
import pytest

from dispatches_amd.pyomo_adapter import HipPyomoSolver, PyomoLP


# ---- stand-ins for pyomo.core (duck-typed: only what the adapter touches) -------------------------------------------------
class Param:
    def __init__(self, value):
        self.value = float(value)


class Expr:
    """Linear expression sum coef * Var + const; coefficients may be Params (mutable) or products of them."""

    def __init__(self, terms=None, const=0.0):
        self.terms = list(terms or [])          # (coef or Param or callable, var)
        self.const = const

    @staticmethod
    def _val(c):
        return c.value if isinstance(c, Param) else (c() if callable(c) else float(c))

    def __add__(self, o):
        o = o if isinstance(o, Expr) else (Expr([(1.0, o)]) if isinstance(o, VarData) else Expr(const=o))
        return Expr(self.terms + o.terms, _sum(self.const, o.const))

    __radd__ = __add__

    def __sub__(self, o):
        return self + (-1.0) * (o if isinstance(o, Expr) else (Expr([(1.0, o)]) if isinstance(o, VarData) else Expr(const=o)))

    def __rmul__(self, s):
        return Expr([(_mul(s, c), v) for c, v in self.terms], _mul(s, self.const))

    __mul__ = __rmul__


def _mul(a, b):
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return a * b
    return lambda: Expr._val(a) * Expr._val(b)


def _sum(a, b):
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return a + b
    return lambda: Expr._val(a) + Expr._val(b)


class VarData:
    def __init__(self, name, lb=0.0, ub=None):
        self.name, self.lb, self.ub, self.fixed, self.value = name, lb, ub, False, None

    def fix(self, v):
        self.fixed, self.value = True, float(v)

    def __rmul__(self, s):
        return Expr([(s, self)])

    __mul__ = __rmul__

    def __add__(self, o):
        return Expr([(1.0, self)]) + o

    __radd__ = __add__

    def __sub__(self, o):
        return Expr([(1.0, self)]) - o


class ConData:
    def __init__(self, name, body, lower=None, upper=None):
        self.name, self.body, self.lower, self.upper = name, body, lower, upper


class ObjData:
    def __init__(self, expr, sense=1):
        self.expr, self.sense = expr, sense


class Block:
    def __init__(self):
        self.vars, self.cons, self.objs = [], [], []

    def component_data_objects(self, ctype, active=True, descend_into=True):
        return iter({VarData: self.vars, ConData: self.cons, ObjData: self.objs}[ctype])


class QuadExpr:
    """Linear expression + products of two variables (what Pyomo holds for an expanded sum of squares)."""

    def __init__(self, lin, quad):
        self.lin, self.quad = lin, list(quad)          # quad: (coef, var1, var2)

    @staticmethod
    def square(expr, weight):
        """(weight / 2) * expr^2 for a constant-free linear Expr, expanded into products."""
        assert Expr._val(expr.const) == 0.0
        q = [(0.5 * weight * Expr._val(a) * Expr._val(b), u, v) for a, u in expr.terms for b, v in expr.terms]
        return QuadExpr(Expr(), q)

    def __add__(self, o):
        if isinstance(o, QuadExpr):
            return QuadExpr(self.lin + o.lin, self.quad + o.quad)
        return QuadExpr(self.lin + o, self.quad)

    __radd__ = __add__


class Repn:
    def __init__(self, expr):
        self.quadratic_vars, self.quadratic_coefs, self.nonlinear_expr = [], [], None
        if isinstance(expr, QuadExpr):
            self.quadratic_vars = [(u, v) for _, u, v in expr.quad]
            self.quadratic_coefs = [q for q, _, _ in expr.quad]
            expr = expr.lin
        acc = {}
        for c, v in expr.terms:
            acc.setdefault(id(v), [v, 0.0])[1] += Expr._val(c)
        self.linear_vars = [v for v, _ in acc.values()]
        self.linear_coefs = [a for _, a in acc.values()]
        self.constant = Expr._val(expr.const)

    def is_linear(self):
        return not self.quadratic_vars

    def is_quadratic(self):
        return bool(self.quadratic_vars)


def generate_standard_repn(expr, compute_values=True):
    return Repn(expr if isinstance(expr, (Expr, QuadExpr)) else Expr([(1.0, expr)]))


CTYPES = (VarData, ConData, ObjData)


# ---- the tracking LP, "in Pyomo" ---------------------------------------------------------------------------------------------
def build_tracking_model(cf, dispatch, soc0=0.0, wind_kw=200e3, batt_kw=25e3):
    """Per period: wind <= capacity * cf (capacity a FIXED var, cf a mutable Param), splitter, battery rows with fixed
    nameplate power / energy, P_T + under - over = dispatch; min sum cost + 1e4 (under + over)   (SURVEY A.1 + A.5)."""
    b = Block()
    T = len(cf)
    cfp = [Param(v) for v in cf]
    disp = [Param(v) for v in dispatch]
    cap = VarData("windpower.system_capacity"); cap.fix(wind_kw)
    pw = VarData("battery.nameplate_power"); pw.fix(batt_kw)
    en = VarData("battery.nameplate_energy"); en.fix(4 * batt_kw)
    soc_init = VarData("battery.initial_state_of_charge"); soc_init.fix(soc0)
    thr_init = VarData("battery.initial_energy_throughput"); thr_init.fix(0.0)
    b.vars += [cap, pw, en, soc_init, thr_init]
    cost = Expr()
    sp, tp = soc_init, thr_init
    for t in range(T):
        W, G, I, O = (VarData(f"{nm}[{t}]") for nm in ("wind", "grid", "batt_in", "batt_out"))
        S, E, un, ov = (VarData(f"{nm}[{t}]") for nm in ("soc", "thr", "under", "over"))
        b.vars += [W, G, I, O, S, E, un, ov]
        b.cons += [ConData(f"wind_cf[{t}]", W - cfp[t] * cap, upper=0.0),             # mutable Param x FIXED var: rhs only
                   ConData(f"split[{t}]", W - G - I, 0.0, 0.0),
                   ConData(f"soc[{t}]", S - sp - 0.95 * I + (1 / 0.95) * O, 0.0, 0.0),
                   ConData(f"thr[{t}]", E - tp - 0.5 * I - 0.5 * O, 0.0, 0.0),
                   ConData(f"soc_cap[{t}]", S + 1e-4 * E - en, upper=0.0),
                   ConData(f"pin[{t}]", I - pw, upper=0.0), ConData(f"pout[{t}]", O - pw, upper=0.0),
                   ConData(f"track[{t}]", 1e-3 * G + 1e-3 * O + un - ov - Expr(const=lambda d=disp[t]: d.value), 0.0, 0.0)]
        waste = 1e-3 * (cfp[t] * cap) - 1e-3 * W
        cost = cost + (41.78 / 8760) * cap + (1e-4 * 29.545625) * (E - tp) + 1e3 * waste + 1e4 * (un + ov)
        sp, tp = S, E
    b.objs.append(ObjData(cost, sense=1))
    return b, cfp, disp, soc_init


def _solve(lp):
    from oracle.highs_direct import HighsModel
    M = HighsModel(lp.c, lp.csr(), lp.rlo, lp.rhi, lp.lb, lp.ub, c0=lp.c0)
    x, f, _ = M.solve()
    return x, f


def test_flattened_pyomo_model_matches_the_oracle_and_refreshes(golden, rts309):
    from oracle import dispatch_lp_oracle as orc
    g = golden["G3_tracker_wind_battery"]
    D = g["market_dispatch_mw"]
    cf = list(rts309["rt_cf"][:4])
    blk, cfp, disp, soc_init = build_tracking_model(cf, D)
    P = PyomoLP(blk, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    lp = P.lp
    assert lp.n == 8 * 4 and lp.m == 8 * 4            # the five fixed design / initial-state variables get no column
    x, f = _solve(lp)
    ref = orc.wind_battery_track(4, cf, D)[0].solve()[1]
    assert f == pytest.approx(ref, rel=1e-9)
    P.load_solution(x)
    wind = [v.value for v in blk.vars if v.name.startswith("wind[")]
    assert wind == pytest.approx(g["expected_wind_power_kw"], rel=1e-3)            # the reference's golden (G3)
    # rolling-horizon update: new capacity factors, dispatch signal and initial SOC are Params / fixed values
    cf2 = list(rts309["rt_cf"][1:5])
    D2 = [1.0, 12.0, 20.0, 3.0]
    for p, v in zip(cfp, cf2):
        p.value = v
    for p, v in zip(disp, D2):
        p.value = v
    soc_init.fix(1234.57)
    P.refresh()
    x2, f2 = _solve(P.lp)
    ref2 = orc.wind_battery_track(4, cf2, D2, soc0=1234.57)[0].solve()[1]
    assert f2 == pytest.approx(ref2, rel=1e-9)
    assert P.objective_value(x2) == pytest.approx(ref2, rel=1e-9)


def test_refresh_refuses_a_changed_matrix_or_fixed_set(rts309):
    blk, cfp, disp, soc_init = build_tracking_model(list(rts309["rt_cf"][:4]), [0, 1.5, 15, 24.5])
    P = PyomoLP(blk, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    eta = Param(0.95)
    blk.cons[2].body = blk.cons[2].body + Expr([(lambda: eta.value - 0.95, blk.vars[7])])   # a mutable Param on a free variable
    P.refresh()                                                                             # unchanged value: fine
    eta.value = 0.9
    with pytest.raises(ValueError, match="changed its coefficients"):
        P.refresh()
    eta.value = 0.95
    blk.vars[6].fix(10.0)                                                                   # a column disappears
    with pytest.raises(ValueError, match="set of fixed variables changed"):
        P.refresh()


def test_maximisation_and_nonlinear_rejection():
    b = Block()
    x, y = VarData("x", 0.0, 4.0), VarData("y", 0.0, None)
    b.vars += [x, y]

    class Callable6:                      # a bound held as a NumericValue (mutable Param / expression): evaluated by calling it
        def __call__(self):
            return 6.0
    b.cons += [ConData("c", x + y, upper=Callable6())]
    b.objs.append(ObjData(3.0 * x + 2.0 * y, sense=-1))
    P = PyomoLP(b, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    xs, f = _solve(P.lp)
    assert P.objective_value(xs) == pytest.approx(16.0)        # max 3x + 2y: x = 4, y = 2

    class NL(Repn):
        def is_linear(self):
            return False
    with pytest.raises(ValueError, match="not linear"):
        PyomoLP(b, ctypes=CTYPES, generate_standard_repn=lambda e, compute_values=True: NL(e if isinstance(e, Expr) else Expr([(1.0, e)])))


# ---- convex quadratic objectives (BASELINE config 5) ------------------------------------------------------------------------
def _qp_bracket(lp):
    """Certified bracket of the optimal value of an LP with soft rows (oracle/qp_cutting_plane.py: Kelley on the LP oracle)."""
    from types import SimpleNamespace

    import scipy.sparse as sp
    from oracle.qp_cutting_plane import solve_qp_bracket
    soft = lp.row_compliance > 0
    A = lp.csr()
    P = SimpleNamespace(A=A[~soft], c=lp.c, c0=lp.c0, lo=lp.rlo[~soft], hi=lp.rhi[~soft], lb=lp.lb, ub=lp.ub)
    M = sp.diags(1.0 / np.sqrt(lp.row_compliance[soft])) @ A[soft]            # |M x|^2 / 2 = sum (a.x)^2 / (2 kappa)
    return solve_qp_bracket(P, M, 1.0, gap_rel=1e-10)


def test_quadratic_objective_becomes_sparse_soft_rows(rts309):
    """A ramp cost (rho / 2) sum_t (P_t - P_{t-1})^2, P_t = 1e-3 (grid_t + batt_out_t), handed over the way Pyomo holds it -
    expanded into products of variables - comes out as soft rows: same objective for any x, as many rows as the form has rank,
    the same optimum as the formulation that lists the squares directly, and refresh() keeps it."""
    rho = 40.0
    cf, D = list(rts309["rt_cf"][:4]), [0.0, 1.5, 15.0, 24.5]
    blk, cfp, disp, soc_init = build_tracking_model(cf, D)
    G = [v for v in blk.vars if v.name.startswith("grid[")]
    O = [v for v in blk.vars if v.name.startswith("batt_out[")]
    obj = blk.objs[0]
    ramps = [1e-3 * G[t] + 1e-3 * O[t] - 1e-3 * G[t - 1] - 1e-3 * O[t - 1] for t in range(1, 4)]
    quad = QuadExpr(obj.expr, [])
    for r in ramps:
        quad = quad + QuadExpr.square(r, rho)
    blk.objs[0] = ObjData(quad, sense=1)
    P = PyomoLP(blk, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    lp = P.lp
    soft = np.nonzero(lp.row_compliance > 0)[0]
    assert len(soft) == 3 and lp.m == 8 * 4 + 3                        # rank of the form = number of ramps
    assert (np.diff(lp.indptr)[soft] <= 4).all()                         # LDL' in column order keeps the rows short
    assert (lp.rlo[soft] == 0).all() and (lp.rhi[soft] == 0).all()
    # the same objective for any x
    rng = np.random.default_rng(3)
    cols = {v.name: j for j, v in enumerate(P._vars)}
    for _ in range(5):
        x = rng.random(lp.n) * 1e4
        direct = float(lp.c @ x + lp.c0)
        for t in range(1, 4):
            p1 = 1e-3 * (x[cols[f"grid[{t}]"]] + x[cols[f"batt_out[{t}]"]])
            p0 = 1e-3 * (x[cols[f"grid[{t - 1}]"]] + x[cols[f"batt_out[{t - 1}]"]])
            direct += 0.5 * rho * (p1 - p0) ** 2
        assert lp.objective(x) == pytest.approx(direct, rel=1e-11)
        assert P.objective_value(x) == pytest.approx(direct, rel=1e-11)
    # the same optimum as the squares listed directly (another factorisation of the same form)
    got = _qp_bracket(lp)
    import scipy.sparse as sp
    from dispatches_amd.lp import StandardFormLP
    hard = lp.row_compliance == 0
    A = lp.csr()
    rows = []
    for t in range(1, 4):
        r = np.zeros(lp.n)
        for nm, sg in ((f"grid[{t}]", 1e-3), (f"batt_out[{t}]", 1e-3), (f"grid[{t - 1}]", -1e-3), (f"batt_out[{t - 1}]", -1e-3)):
            r[cols[nm]] += sg
        rows.append(r)
    A2 = sp.vstack([A[hard], sp.csr_matrix(np.array(rows))]).tocsr()
    lp2 = StandardFormLP(n=lp.n, m=A2.shape[0], indptr=A2.indptr.astype(np.int32), indices=A2.indices.astype(np.int32), data=A2.data,
                         c=lp.c, c0=lp.c0, lb=lp.lb, ub=lp.ub, rlo=np.concatenate([lp.rlo[hard], np.zeros(3)]),
                         rhi=np.concatenate([lp.rhi[hard], np.zeros(3)]), col_names=lp.col_names, row_names=None,
                         row_compliance=np.concatenate([np.zeros(int(hard.sum())), np.full(3, 1.0 / rho)]))
    ref = _qp_bracket(lp2)
    tol = 1e-8 * (1 + abs(ref["upper"]))
    assert got["lower"] <= ref["upper"] + tol and ref["lower"] <= got["upper"] + tol
    assert got["upper"] - got["lower"] <= 1e-8 * (1 + abs(got["upper"]))
    # the ramp cost must matter here (otherwise the test pins nothing): the LP optimum is lower
    lp_only = StandardFormLP(**{**lp2.__dict__, "row_compliance": None, "m": int(hard.sum()), "indptr": A[hard].indptr.astype(np.int32),
                                "indices": A[hard].indices.astype(np.int32), "data": A[hard].data, "rlo": lp.rlo[hard], "rhi": lp.rhi[hard]})
    assert _solve(lp_only)[1] < ref["lower"] - 1e-3
    # refresh: Params move the linear part, the factors stay; a changed quadratic coefficient is refused
    disp[1].value = 3.0
    P.refresh()
    assert (P.lp.row_compliance[soft] > 0).all() and P.lp.rlo[8 * 1 + 7] == pytest.approx(3.0)
    blk.objs[0] = ObjData(quad + QuadExpr.square(ramps[0], 1.0), sense=1)
    with pytest.raises(ValueError, match="quadratic part of the objective changed"):
        P.refresh()


def test_nonconvex_and_concave_quadratics():
    b = Block()
    x, y = VarData("x", -5.0, 5.0), VarData("y", -5.0, 5.0)
    b.vars += [x, y]
    b.cons += [ConData("c", x + y, upper=6.0)]
    b.objs.append(ObjData(QuadExpr(Expr([(1.0, x)]), [(1.0, x, x), (-1.0, y, y)]), sense=1))          # x^2 - y^2: indefinite
    with pytest.raises(ValueError, match="not convex"):
        PyomoLP(b, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    b.objs[0] = ObjData(QuadExpr(Expr([(1.0, x)]), [(1.0, x, y)]), sense=1)                            # x y: zero pivot, nonzero column
    with pytest.raises(ValueError, match="not convex"):
        PyomoLP(b, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    # maximising a concave form is a convex problem: max x - (x - y)^2 - y^2  ->  (x, y) = (1, 1/2), value 1/2
    b.objs[0] = ObjData(QuadExpr(Expr([(1.0, x)]), [(-1.0, x, x), (2.0, x, y), (-2.0, y, y)]), sense=-1)
    P = PyomoLP(b, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    assert (P.lp.row_compliance > 0).sum() == 2
    assert P.objective_value(np.array([1.0, 0.5])) == pytest.approx(0.5)
    br = _qp_bracket(P.lp)
    assert -br["upper"] == pytest.approx(0.5, abs=1e-7) and br["x"] == pytest.approx([1.0, 0.5], abs=1e-3)
    # a fixed variable inside a product moves into the linear part
    y.fix(0.5)
    P2 = PyomoLP(b, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    assert P2.lp.n == 1 and (P2.lp.row_compliance > 0).sum() == 1
    assert P2.objective_value(np.array([1.0])) == pytest.approx(0.5)


def test_solver_object_solves_scenario_blocks_as_one_batch(rts309):
    """Route B of INTEGRATION.md end to end with stand-ins: `HipPyomoSolver.solve([blocks])` = one batch over identical
    scenario blocks (different capacity factors / dispatch signals), solutions back in the Vars, refresh on the second call;
    a block with another matrix is refused."""
    from _highs_solver import HighsTestSolver
    from oracle import dispatch_lp_oracle as orc
    cfs = [list(rts309["rt_cf"][k:k + 4]) for k in (0, 5, 11)]
    Ds = [[0.0, 1.5, 15.0, 24.5], [2.0, 2.0, 30.0, 1.0], [10.0, 0.0, 0.0, 5.0]]
    built = [build_tracking_model(cf, D) for cf, D in zip(cfs, Ds)]
    blocks = [b[0] for b in built]
    solver = HipPyomoSolver(backend=HighsTestSolver(), ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    res = solver.solve(blocks)
    assert res.solver.termination_condition == "optimal"
    batch = solver.last_batch
    assert batch.n_scenario == 3 and batch.c.shape == (3, batch.lp.n)
    for i, (cf, D) in enumerate(zip(cfs, Ds)):
        ref = orc.wind_battery_track(4, cf, D)[0].solve()[1]
        assert batch.objective[i] == pytest.approx(ref, rel=1e-9)
        track = [v.value for v in blocks[i].vars if v.name.startswith("grid[") or v.name.startswith("batt_out[")]
        assert all(t is not None for t in track)                                  # solution loaded into the Vars
    # second call: Params changed in ONE scenario -> refreshed, same handle / matrix
    for p, v in zip(built[1][2], [1.0, 12.0, 20.0, 3.0]):
        p.value = v
    lp_before = batch.lp
    solver.solve(blocks)
    assert solver.last_batch is batch and batch.lp is lp_before
    assert batch.objective[1] == pytest.approx(orc.wind_battery_track(4, cfs[1], [1.0, 12.0, 20.0, 3.0])[0].solve()[1], rel=1e-9)
    assert batch.objective[0] == pytest.approx(orc.wind_battery_track(4, cfs[0], Ds[0])[0].solve()[1], rel=1e-9)
    # a block with a different matrix cannot join the batch
    odd, *_ = build_tracking_model(cfs[0], Ds[0])
    odd.cons[2].body = odd.cons[2].body + Expr([(0.5, odd.vars[7])])
    with pytest.raises(ValueError, match="does not flatten to the matrix of block 0"):
        HipPyomoSolver(backend=HighsTestSolver(), ctypes=CTYPES, generate_standard_repn=generate_standard_repn).solve([blocks[0], odd])
