"""The live-Pyomo flattener (dispatches_amd/pyomo_adapter.py) exercised with stand-in objects that
mimic the part of Pyomo's API it uses (Pyomo itself is absent from the build container): a block with Var / Constraint /
Objective data objects, mutable Params and ``generate_standard_repn``.  The stand-in model is the 4-period wind + battery
tracking LP written the way the reference writes it in Pyomo (per-period blocks, fixed design variables, mutable
capacity-factor Params); the flattened LP must have the same optimum as the product's LinearBlock formulation and the
oracle, must refresh when Params / fixed values change, and must refuse changes that alter the matrix."""
import numpy as np
import pytest

from dispatches_amd.pyomo_adapter import PyomoLP


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


class Repn:
    def __init__(self, expr):
        acc = {}
        for c, v in expr.terms:
            acc.setdefault(id(v), [v, 0.0])[1] += Expr._val(c)
        self.linear_vars = [v for v, _ in acc.values()]
        self.linear_coefs = [a for _, a in acc.values()]
        self.constant = Expr._val(expr.const)

    def is_linear(self):
        return True


def generate_standard_repn(expr, compute_values=True):
    return Repn(expr if isinstance(expr, Expr) else Expr([(1.0, expr)]))


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
    b.cons += [ConData("c", x + y, upper=6.0)]
    b.objs.append(ObjData(3.0 * x + 2.0 * y, sense=-1))
    P = PyomoLP(b, ctypes=CTYPES, generate_standard_repn=generate_standard_repn)
    xs, f = _solve(P.lp)
    assert P.objective_value(xs) == pytest.approx(16.0)        # max 3x + 2y: x = 4, y = 2

    class NL(Repn):
        def is_linear(self):
            return False
    with pytest.raises(ValueError, match="not linear"):
        PyomoLP(b, ctypes=CTYPES, generate_standard_repn=lambda e, compute_values=True: NL(e if isinstance(e, Expr) else Expr([(1.0, e)])))
