"""Device-resident batched double loop for ANY of the three flowsheets of the reference (BASELINE config 2 is a "Nuclear case
double-loop"; config 4 the wind + battery one, whose specialised form - fused update kernel, recording - is dispatches_amd/rolling.py).

The same simulated day as rolling.py (1 day-ahead bidding solve, then per hour: real-time bid, stub clearing, tracking, state hand-off)
written once over a small DESCRIPTOR of the flowsheet's rolling-horizon state:

    flowsheet        realised state handed from the tracker to the next hour's LPs                          window data
    wind_battery     state of charge, energy throughput, rounded to 2 dp  (wind_battery_double_loop.py:181-209)    capacity factors
    wind_pem         none: only the capacity-factor window advances        (wind_PEM_double_loop.py:185-204)        capacity factors
    nuclear          tank holdup, rounded to an integer `round(holdup[-1])` (nuclear_flowsheet_multiperiod_class.py:218-239)   -

Everything per plant lives in HBM and is updated by device index operations; the solves go through the C ABI on device pointers
(day-ahead: PDLP kernel; 4-h / 12-h LPs: in-wave simplex first); the steps of a day are captured into hipGraphs on the second day and
replayed.  Power output and objective come from the flowsheet's own expressions as dense rows (P_T = PT x + PT_const), so nothing
below knows a flowsheet's columns except through the descriptor.  The market is the stub of rolling.py (every offer clears at its
maximum; day-ahead bids of day d at hour 0 of day d)."""
from __future__ import annotations

import numpy as np

from . import scenarios
from .hip_solver import DeviceLP, default_options
from .rolling import _NoSolver


def _dense_rows(block, family, n, hours):
    ex = block.expressions[family]
    return np.stack([ex[t].dense(n) for t in hours]), np.array([ex[t].const for t in hours])


class _Model:
    """One of the three LPs on the device, described without reference to a flowsheet."""

    def __init__(self, model, block_family, B, dev, device_index, power_output, state_init, wind, lp_backend=None):
        import torch
        self.lp = model.lp
        self.T = len(model.HOUR)
        n = self.lp.n
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float64), device=dev)
        idx = lambda cols: torch.as_tensor(np.asarray(cols, np.int64), device=dev)
        lb, ub, rlo, rhi = model.block.current_bounds()
        base_c = model.base_c if hasattr(model, "base_c") else model.c[0]
        self.base_c = t(base_c)
        self.base_c0 = float(model.base_c0) if hasattr(model, "base_c0") else float(np.asarray(model.c0).ravel()[0])
        self.c = self.base_c.repeat(B, 1)
        self.c0 = torch.full((B,), self.base_c0, dtype=torch.float64, device=dev)
        self.lb, self.ub = t(lb).repeat(B, 1), t(ub).repeat(B, 1)
        self.rlo, self.rhi = t(rlo).repeat(B, 1), t(rhi).repeat(B, 1)
        PT, PT_const = _dense_rows(model.block, power_output, n, model.HOUR)
        self.PT, self.PT_const = t(PT), t(PT_const)                       # [T, n], [T]
        self.state_init = [int(c) for c in state_init]                    # columns fixed to the realised state
        self.wind = None
        if wind is not None:                                              # (columns, kW, curtailment cost per kW, template availability sum)
            cols, kw, per_kw, template_sum = wind
            self.wind = (idx(cols), float(kw), float(per_kw))
            self.base_c0 -= float(per_kw) * float(template_sum)           # the template's curtailment constant leaves; the window's enters
        if lp_backend is None:
            extra = {"recertify_passes": 3} if self.T > 16 else {"recertify_passes": 0, "eps_infeasible": 0.0}     # (as rolling.py)
            self.opts = default_options(**{**extra, **(getattr(model, "solver_hints", None) or {})})
            from .hip_solver import DspOptions
            self.opts_warm, self.opts_first = DspOptions.from_buffer_copy(self.opts), DspOptions.from_buffer_copy(self.opts)
            if self.T <= 16:                       # hourly LPs: the simplex starts from the previous hour's basis (dsp_options::simplex_warm)
                self.opts_warm.simplex_warm, self.opts_first.simplex_warm = 1, 2
            self.dlp = DeviceLP(self.lp, device_index, self.opts)
            m = max(self.lp.m, 1)
            self.out = dict(x=torch.zeros((B, n), dtype=torch.float64, device=dev), y=torch.zeros((B, m), dtype=torch.float64, device=dev),
                            obj=torch.zeros(B, dtype=torch.float64, device=dev), status=torch.zeros(B, dtype=torch.int32, device=dev),
                            iters=torch.zeros(B, dtype=torch.int32, device=dev), jumps=torch.zeros(B, dtype=torch.int32, device=dev),
                            flags=torch.zeros(B, dtype=torch.int32, device=dev))
        else:
            self.opts, self.dlp, self.out = None, lp_backend(self.lp), None

    def power_output(self, x):
        return x @ self.PT.T + self.PT_const                              # [B, T] MW

    def solve(self, B, hour=None):
        opts = self.opts if (hour is None or self.opts is None) else (self.opts_first if hour == 0 else self.opts_warm)
        self.out = self.dlp.solve(B, self.c, self.lb, self.ub, self.rlo if self.lp.m else None, self.rhi if self.lp.m else None,
                                  options=opts, out=self.out, sync_stats=False, obj_offset=self.c0)
        return self.out


def _templates(flowsheet, day_ahead_horizon, tracking_horizon):
    """(bidder, day-ahead model, real-time model, tracker, descriptor) built ONCE through the product's own model objects (B = 1)."""
    from .workflow import Tracker
    if flowsheet == "wind_battery":
        series, stride, cap = "rts_gmlc_309.npz", 17, 500.0
        bidder, da = scenarios.wind_battery_batch(1, day_ahead_horizon, _NoSolver(), series=series, stride=stride)
        s = scenarios.load_series(series)
        mo = bidder.bidding_model_object
        tr_obj = mo.__class__(model_data=mo.model_data, wind_capacity_factors=list(s["rt_cf"][:tracking_horizon]), wind_pmax_mw=200.0,
                              battery_pmax_mw=25.0, battery_energy_capacity_mwh=100.0)
        fam = "windBattery"
        desc = dict(family=fam, per_kw=mo.wind_waste_penalty * 1e-3, decimals=[2, 2],
                    init=lambda blk: [getattr(blk, fam)["soc_init"].index, getattr(blk, fam)["thr_init"].index],
                    real=lambda blk: [getattr(blk, fam)["periods"][0]["state_of_charge"].index, getattr(blk, fam)["periods"][0]["energy_throughput"].index])
        prices = (np.clip(s["da_lmp"], 0.0, cap), np.clip(s["rt_lmp"], 0.0, cap), s["rt_cf"])
    elif flowsheet == "wind_pem":
        series, stride, cap = "rts_gmlc_303.npz", 37, 500.0
        bidder, da = scenarios.wind_pem_batch(1, day_ahead_horizon, _NoSolver(), series=series, stride=stride)
        s = scenarios.load_series(series)
        mo = bidder.bidding_model_object
        tr_obj = mo.__class__(mo.model_data, wind_capacity_factors=list(s["rt_cf"][:tracking_horizon]), wind_pmax_mw=mo._wind_pmax_mw,
                              pem_pmax_mw=mo._pem_pmax_mw)
        fam = "windPEM"
        desc = dict(family=fam, per_kw=1.0, decimals=[], init=lambda blk: [], real=lambda blk: [])
        prices = (np.clip(s["da_lmp"], 0.0, cap), np.clip(s["rt_lmp"], 0.0, cap), s["rt_cf"])
    elif flowsheet == "nuclear":
        stride = 29
        bidder, da = scenarios.nuclear_batch(1, day_ahead_horizon, _NoSolver())
        s = scenarios.load_series("nuclear_price_taker_lmps.npz")        # bus Attlee, generator 121_NUCLEAR_1 (rts_gmlc_15_500.csv)
        tr_obj = bidder.bidding_model_object.__class__(bidder.bidding_model_object.model_data)
        fam = "nuclear"
        desc = dict(family=fam, per_kw=None, decimals=[0],
                    init=lambda blk: [blk.nuclear["holdup_init"].index], real=lambda blk: [blk.nuclear["periods"][0]["tank_holdup"].index])
        prices = (np.clip(s["da_lmp"], 0.0, None), np.clip(s["rt_lmp"], 0.0, None), None)
    else:
        raise ValueError(f"unknown flowsheet {flowsheet!r}: wind_battery, wind_pem or nuclear")
    tracker = Tracker(tracking_model_object=tr_obj, tracking_horizon=tracking_horizon, n_tracking_hour=1, solver=_NoSolver())
    tracker._pass_market_dispatch([0.0] * tracking_horizon)              # dispatch rows become equalities
    desc.update(stride=stride, prices=prices)
    return bidder, da, bidder.real_time_model, tracker, desc


class BatchedDoubleLoop:
    def __init__(self, flowsheet, n_scenarios, device=0, first_scenario=0, day_ahead_horizon=48, tracking_horizon=4, lp_backend=None,
                 use_graphs=True, use_fused=True, simplex_warm=True):
        """flowsheet: "wind_battery", "wind_pem" or "nuclear".  Plant k sees the year that starts at hour (stride * k) mod N of its bus's
        series (strides 17 / 37 / 29).  lp_backend: tests pass tests/_highs_solver.py::HighsTensorLP to run the same logic on CPU tensors.
        use_fused: on the GPU the ~100 element-wise tensor operations of an hour step are THREE launches of one HIP kernel driven by the
        descriptor (dsp_loop_update, include/dsp_hip.h) - the nuclear loop of 256 plants is launch-bound otherwise (21 ms per simulated day)."""
        import torch
        self.flowsheet = flowsheet
        self.B = B = int(n_scenarios)
        self.dev = dev = torch.device("cuda", device) if lp_backend is None else torch.device("cpu")
        bidder, da_model, rt_model, tracker, d = _templates(flowsheet, day_ahead_horizon, tracking_horizon)
        tr_model = tracker.model
        self.bidder, self.tracker_template = bidder, tracker
        da_s, rt_s, cf_s = d["prices"]
        self.N = N = len(rt_s)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float64), device=dev)
        idx = lambda cols: torch.as_tensor(np.asarray(cols, np.int64), device=dev)
        self.da_series, self.rt_series = t(da_s), t(rt_s)
        self.cf_series = t(cf_s) if cf_s is not None else None
        self.stride = d["stride"]
        self.start = idx((d["stride"] * (first_scenario + np.arange(B))) % N)
        power = bidder.bidding_model_object.power_output
        fam = d["family"]

        def wind_of(model):
            if cf_s is None:
                return None
            f = getattr(model.block, fam)
            cols = [p["wind"].index for p in f["periods"]]
            return cols, f["wind_kw"], d["per_kw"], f["wind_kw"] * float(np.sum(cf_s[:len(cols)]))
        mk = lambda model: _Model(model, fam, B, dev, device, power, d["init"](model.block), wind_of(model), lp_backend)
        self.da, self.rt, self.tr = mk(da_model), mk(rt_model), mk(tr_model)
        self.da.pda_cols, self.rt.pda_cols = idx(da_model.pda_cols), idx(rt_model.pda_cols)
        self.tr.track_rows = idx([tr_model.block.kept_row_index(r) for r in tr_model.tracking_rows])
        self.tr.state_real = d["real"](tr_model.block)
        self.scale = [10.0 ** k for k in d["decimals"]]
        self.penalty = float(bidder.real_time_underbid_penalty)
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)
        self.state = z(B, len(self.scale))                                # realised state of every plant
        self.revenue, self.energy_mwh, self.delivered = z(B), z(B), z(B)
        self.da_offer, self.da_prices = z(B, 24), z(B, 24)
        self.bad = torch.zeros((), dtype=torch.bool, device=dev)
        self.uncertified = torch.zeros((), dtype=torch.int64, device=dev)
        self.hour_t = torch.zeros((), dtype=torch.int64, device=dev)     # the clock on the device (graphs replay across days)
        self._scale_t = [torch.full((), s, dtype=torch.float64, device=dev) for s in self.scale]
        self.hour = self.solves = 0
        self.use_graphs = bool(use_graphs) and lp_backend is None
        self.simplex_warm = bool(simplex_warm) and lp_backend is None
        self._graphs, self._warm = {}, False
        self.use_fused = bool(use_fused) and lp_backend is None and self.rt.T <= 16 and self.tr.T <= 16 and len(self.scale) <= 2
        if self.use_fused:
            self._fused_setup()

    def _fused_setup(self):
        from .hip_solver import DspLoopModel, DspLoopState, load_library
        self._lib = load_library()

        def struct(m, pda=None, track=None, real=None):
            w = DspLoopModel()
            w.c, w.lb, w.ub, w.rlo, w.rhi = (t.data_ptr() for t in (m.c, m.lb, m.ub, m.rlo, m.rhi))
            w.base_c, w.x, w.c0 = m.base_c.data_ptr(), m.out["x"].data_ptr(), m.c0.data_ptr()
            w.n, w.m, w.T, w.n_state = m.lp.n, m.lp.m, m.T, len(self.scale)
            PT, PTc = m.PT.cpu().numpy(), m.PT_const.cpu().numpy()
            wind = m.wind[0].cpu().tolist() if m.wind is not None else []
            for t in range(16):
                nz = np.nonzero(PT[t])[0] if t < m.T else []
                if len(nz) > 2:
                    raise ValueError("the fused update kernel takes power outputs of at most two columns per period")
                for e in range(2):
                    w.pt_cols[t][e] = int(nz[e]) if e < len(nz) else -1
                    w.pt_coef[t][e] = float(PT[t, nz[e]]) if e < len(nz) else 0.0
                w.pt_const[t] = float(PTc[t]) if t < m.T else 0.0
                w.pda_cols[t] = int(pda[t]) if pda is not None and t < len(pda) else -1
                w.track_rows[t] = int(track[t]) if track is not None and t < len(track) else -1
                w.wind_cols[t] = int(wind[t]) if t < len(wind) else -1
            for j in range(2):
                w.state_init[j] = m.state_init[j] if j < len(m.state_init) else 0
                w.state_real[j] = real[j] if real is not None and j < len(real) else 0
            w.wind_kw = m.wind[1] if m.wind is not None else 0.0
            w.waste_per_kw = m.wind[2] if m.wind is not None else 0.0
            w.c0_base = m.base_c0
            w.status, w.flags = m.out["status"].data_ptr(), m.out["flags"].data_ptr()
            return w
        st = DspLoopState()
        st.B, st.N = self.B, self.N
        st.start, st.hour = self.start.data_ptr(), self.hour_t.data_ptr()
        st.da_series, st.rt_series = self.da_series.data_ptr(), self.rt_series.data_ptr()
        st.cf_series = self.cf_series.data_ptr() if self.cf_series is not None else None
        st.state = self.state.data_ptr()
        for j in range(2):
            st.state_scale[j] = self.scale[j] if j < len(self.scale) else 1.0
        st.da_offer, st.da_prices = self.da_offer.data_ptr(), self.da_prices.data_ptr()
        st.delivered, st.revenue, st.energy_mwh = self.delivered.data_ptr(), self.revenue.data_ptr(), self.energy_mwh.data_ptr()
        st.bad, st.uncertified = self.bad.data_ptr(), self.uncertified.data_ptr()
        self._loop_state = st
        self._loop_rt = struct(self.rt, pda=self.rt.pda_cols.cpu().tolist())
        self._loop_tr = struct(self.tr, track=self.tr.track_rows.cpu().tolist(), real=self.tr.state_real)

    def _fused(self, phase, k):
        import ctypes as C
        import torch
        rc = self._lib.dsp_loop_update(C.byref(self._loop_state), C.byref(self._loop_rt), C.byref(self._loop_tr), phase, k,
                                       C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"dsp_loop_update failed ({rc})")

    # -- pieces of a step (all capturable: persistent tensors, the clock read on the device) ---------------------------------------
    def _window(self, series, T):
        import torch
        return series[(self.start[:, None] + self.hour_t + torch.arange(T, device=self.dev)[None, :]) % self.N]

    def _set_prices(self, m, da, rt):
        """c = base - RT . dP_T/dx - (DA - RT) on day_ahead_power ; c0 = base - RT . PT_const (Bidder._pass_price_forecasts)"""
        m.c.copy_(m.base_c - rt @ m.PT)
        m.c[:, m.pda_cols] -= da - rt
        return -(rt @ m.PT_const)                                         # the prices' share of the objective constant [B]

    def _set_state(self, m, price_c0=None):
        """what update_model writes: the state columns fixed to the realised values, wind availability of the window; and the objective
        constant of every plant (dsp_batch::obj_offset: the scale of the solver's objective-accuracy test, cf. rolling.py)"""
        for k, col in enumerate(m.state_init):
            m.lb[:, col] = self.state[:, k]
            m.ub[:, col] = self.state[:, k]
        m.c0.fill_(m.base_c0)
        if price_c0 is not None:
            m.c0 += price_c0
        if m.wind is not None:
            cols, kw, per_kw = m.wind
            avail = kw * self._window(self.cf_series, m.T)
            m.ub[:, cols] = avail
            m.c0 += per_kw * avail.sum(1)

    def _check(self, out):
        self.bad |= (out["status"] != 0).any()
        if out.get("flags") is not None:
            self.uncertified += ((out["flags"] & 1) != 0).sum()

    def _day_ahead_step(self):
        m = self.da
        da, rt = self._window(self.da_series, m.T), self._window(self.rt_series, m.T)
        self._set_state(m, self._set_prices(m, da, rt))
        m.lb.index_fill_(1, m.pda_cols, 0.0)
        m.ub.index_fill_(1, m.pda_cols, float("inf"))
        out = m.solve(self.B)
        self._check(out)
        self.da_offer.copy_(out["x"][:, m.pda_cols][:, :24])
        self.da_prices.copy_(da[:, :24])

    def _hour_step(self, k):
        import torch
        if self.use_fused:
            hour = k if self.simplex_warm else None
            self._fused(0, k)
            self.rt.solve(self.B, hour=hour)            # (status / flags: checked by the kernel's next phase)
            self._fused(1, k)
            self.tr.solve(self.B, hour=hour)
            self._fused(2, k)
            return
        m = self.rt
        rt = self._window(self.rt_series, m.T)
        da = self._window(self.da_series, m.T).clone()
        known = min(m.T, 24 - k)                                          # hours of the horizon inside the cleared day
        da[:, :known] = self.da_prices[:, k:k + known]
        self._set_state(m, self._set_prices(m, da, rt))
        m.lb.index_fill_(1, m.pda_cols, 0.0)
        m.ub.index_fill_(1, m.pda_cols, float("inf"))
        m.lb[:, m.pda_cols[:known]] = self.da_offer[:, k:k + known]
        m.ub[:, m.pda_cols[:known]] = self.da_offer[:, k:k + known]
        hour = k if self.simplex_warm else None
        out = m.solve(self.B, hour=hour) if m.opts is not None else m.solve(self.B)
        self._check(out)
        offer = m.power_output(out["x"])                                  # real-time offer = SCED dispatch in the stub market
        tr = self.tr
        self._set_state(tr)
        rhs = offer[:, :tr.T] - tr.PT_const
        tr.rlo[:, tr.track_rows] = rhs
        tr.rhi[:, tr.track_rows] = rhs
        out = tr.solve(self.B, hour=hour) if tr.opts is not None else tr.solve(self.B)
        self._check(out)
        x = out["x"]
        self.delivered.copy_(tr.power_output(x)[:, 0])
        for j, col in enumerate(tr.state_real):                           # implemented profile -> next hour's state, rounded as update_model does
            self.state[:, j] = torch.round(x[:, col] * self.scale[j]) / self._scale_t[j]
        self.revenue += self.delivered * rt[:, 0] + self.da_offer[:, k] * (self.da_prices[:, k] - rt[:, 0])
        self.energy_mwh += self.delivered
        self.hour_t += 1

    def _run(self, key, fn):
        import torch
        if not self.use_graphs or not self._warm:
            fn()
            return
        g = self._graphs.get(key)
        if g is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            self._graphs[key] = g
        g.replay()

    # -- the loop ----------------------------------------------------------------------------------------------------------------------
    def day_ahead(self):
        self.day_start = self.hour
        self._run("da", self._day_ahead_step)
        self.solves += self.B
        return self.da_offer.clone()

    def hour_step(self):
        k = self.hour - self.day_start
        self._run(k, lambda: self._hour_step(k))
        self.solves += 2 * self.B
        self.hour += 1
        return self.delivered.clone()

    def run_day(self):
        self.day_ahead()
        for _ in range(24):
            self.hour_step()
        self._warm = True

    def reset(self):
        for t in (self.state, self.revenue, self.energy_mwh, self.delivered, self.da_offer, self.da_prices, self.hour_t, self.uncertified):
            t.zero_()
        self.bad.zero_()
        self.hour = self.solves = 0

    def results(self):
        return dict(obj=self.revenue, energy_mwh=self.energy_mwh, state=self.state), not bool(self.bad.item())
