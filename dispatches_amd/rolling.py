"""Device-resident batched double loop: the rolling-horizon solves of B independent wind + battery plants (BASELINE config 4:
"RTS-GMLC full-year double loop, 8192 scenarios sharded across 8 x MI355X").

What the reference does for ONE plant through Prescient's callbacks (run_double_loop_battery.py:222-305 ->
DoubleLoopCoordinator.bid_into_DAM / bid_into_RTM / track_sced_signal):

    every day   : day-ahead bids            Bidder.compute_day_ahead_bids          (48-h LP)
    every hour  : real-time bids            Bidder.compute_real_time_bids          (4-h LP, day_ahead_power fixed to the
                                                                                    cleared DA dispatch where it is known)
                  tracking of the dispatch  Tracker.track_market_dispatch          (4-h LP)
                  state hand-off            get_implemented_profile -> update_model on tracker and bidder
                                            (initial SOC / energy throughput re-fixed to the realised values ROUNDED to
                                            2 dp, clock advanced one hour, capacity-factor window shifted;
                                            wind_battery_double_loop.py:181-274)

is done here for B plants at once with every per-plant vector resident in HBM: price / capacity-factor windows are
gathered on the device, the objective vectors and the mutable bounds of the three LPs are rewritten by device index
operations, the solves go straight through the C ABI on device pointers (day-ahead: PDLP kernel; the hourly LPs: the
in-wave simplex), and the realised state (SOC, throughput) never leaves the device between hours - the rolling
hand-off of SURVEY.md 8(f)-3.  No Prescient: the market is a stub that clears every offer at its maximum (day-ahead
dispatch = day-ahead offer, real-time dispatch = real-time offer), as in tests/test_double_loop_stub.py.  Day-ahead bids
for day d are computed at hour 0 of day d from the realised state (the reference computes them during day d - 1 from a
projected state; without a unit-commitment run in between the two coincide in the stub market).

Scenario k is the plant seen through the price / capacity-factor year that starts at hour (stride * k) mod N of the
RTS-GMLC series (the BASELINE config-4 windows of SURVEY.md 8(d)).  Shards are contiguous scenario ranges: one rank
per GPU runs its shard with no communication and the per-scenario annual results are all-gathered once at the end
(dispatches_amd.distributed.gather_device_results).
"""
from __future__ import annotations

import numpy as np

from . import scenarios
from .hip_solver import DeviceLP, default_options


class _NoSolver:
    def solve(self, *a, **k):
        raise RuntimeError("template model: never solved on the host")


class _DeviceModel:
    """One of the three LPs on the device: handle + per-scenario tensors + the index sets the rolling updates touch."""

    def __init__(self, model, B, dev, device_index, hints=None, lp_backend=None, waste_cost_per_kw=None, cf_template_sum=0.0):
        import torch
        self.lp = model.lp
        # objective constant of every plant (dsp_batch::obj_offset: the scale of the solver's objective-accuracy test - the reference's
        # objective INCLUDES the constants of tot_cost, wind_battery_double_loop.py:175-177, above all waste penalty x available wind,
        # which is ~100 x the objective itself on a windy day): the template's constant with the template window's curtailment term
        # swapped for the plant's own (scenarios._apply_cf_windows).  None = not maintained (LPs the simplex solves to a vertex).
        self.c0 = None
        if waste_cost_per_kw is not None:
            self.c0 = torch.zeros(B, dtype=torch.float64, device=dev)
            base = float(model.base_c0) if hasattr(model, "base_c0") else float(np.asarray(model.c0).ravel()[0])
            self._c0_base = base - float(waste_cost_per_kw) * float(model.block.windBattery["wind_kw"]) * float(cf_template_sum)
            self._waste_per_kw = float(waste_cost_per_kw)
        self.T = len(model.HOUR)
        t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        lb, ub, rlo, rhi = model.block.current_bounds()
        self.c = t(model.base_c if hasattr(model, "base_c") else model.c[0]).repeat(B, 1)
        self.base_c = self.c[0].clone()
        self.lb, self.ub = t(lb).repeat(B, 1), t(ub).repeat(B, 1)
        self.rlo, self.rhi = t(rlo).repeat(B, 1), t(rhi).repeat(B, 1)
        fam = model.block.windBattery
        per = fam["periods"]
        idx = lambda cols: torch.as_tensor(np.asarray(cols, np.int64), device=dev)
        self.wind_cols = idx([p["wind"].index for p in per])
        self.soc_init, self.thr_init = fam["soc_init"].index, fam["thr_init"].index
        # (period 0's throughput is a column in the reference's form of the accumulator; in the two-level form it is an expression and
        #  this model cannot hand the realised state over - the loop takes it from the tracking model, whose horizon keeps the chain)
        # -> thr0 = None, never -1: an index of -1 reads the LAST column on the host and runs off the array in the fused update kernel
        self.soc0, self.thr0 = per[0]["state_of_charge"].index, getattr(per[0]["energy_throughput"], "index", None)
        self.wind_kw = float(fam["wind_kw"])
        # P_T[t] = 1e-3 (grid_elec[t] + elec_out[t]): the two columns of every hour
        self.pt_cols = idx([[p["grid_elec"].index, p["elec_out"].index] for p in per])       # [T, 2]
        if lp_backend is None:
            # recertify: the loop never reads a flag back between solves (its days are hipGraph replays), so a solve accepted without a
            # certified objective accuracy is re-solved on the device under other settings (dsp_options::recertify_passes)
            # (three passes for the day-ahead LP, which the PDLP kernel solves; one for the hourly LPs, which the in-wave simplex solves to a
            #  vertex and which pay an empty launch per pass and solve, 49 solves per plant-day)
            # The hourly LPs get neither: they carry slack columns (always feasible - and the simplex reports an infeasible input itself), and a
            # first-order fallback has not been needed once (zero hand-overs); should one ever come back flagged, the loop's `uncertified`
            # count says so.  Two empty launches per solve less, 96 per plant-day.
            extra = {"recertify_passes": 3} if self.T > 16 else {"recertify_passes": 0, "eps_infeasible": 0.0}
            self.opts = default_options(**{**extra, **(hints or {})})
            # The hourly LPs of a plant share one matrix from hour to hour: the simplex starts hour k from hour k - 1's final basis
            # (dsp_options::simplex_warm: 2 - 4 pivots instead of ~26) and from the slack basis in the first hour of every day.
            from .hip_solver import DspOptions
            self.opts_warm, self.opts_first = DspOptions.from_buffer_copy(self.opts), DspOptions.from_buffer_copy(self.opts)
            if self.T <= 16:
                self.opts_warm.simplex_warm, self.opts_first.simplex_warm = 1, 2
            self.dlp = DeviceLP(self.lp, device_index, self.opts)
            # output buffers with fixed addresses from the start (the fused update kernel and the hipGraphs hold pointers)
            n, m = self.lp.n, max(self.lp.m, 1)
            self.out = dict(x=torch.zeros((B, n), dtype=torch.float64, device=dev), y=torch.zeros((B, m), dtype=torch.float64, device=dev),
                            obj=torch.zeros(B, dtype=torch.float64, device=dev), status=torch.zeros(B, dtype=torch.int32, device=dev),
                            iters=torch.zeros(B, dtype=torch.int32, device=dev), jumps=torch.zeros(B, dtype=torch.int32, device=dev),
                            flags=torch.zeros(B, dtype=torch.int32, device=dev))
        else:                                   # tests: a stand-in with DeviceLP.solve's signature (CPU tensors + HiGHS)
            self.opts = None
            self.dlp = lp_backend(self.lp)
        if lp_backend is not None:
            self.out = None

    def wb_struct(self, needs_state=True):
        """dsp_wb_model of this LP (include/dsp_hip.h) for the fused rolling-update kernel.  needs_state: the kernel reads the
        realised state (soc0 / thr0) from THIS model's solution - true for the tracking model only; the real-time bidding model
        hands over thr_init alone and may hold period 0's throughput as an expression."""
        from .hip_solver import DspWbModel
        w = DspWbModel()
        w.c, w.lb, w.ub, w.rlo, w.rhi = (t.data_ptr() for t in (self.c, self.lb, self.ub, self.rlo, self.rhi))
        w.base_c, w.x = self.base_c.data_ptr(), self.out["x"].data_ptr()
        w.n, w.m, w.T = self.lp.n, self.lp.m, self.T
        if self.thr0 is None and needs_state:
            raise ValueError("this model holds period 0's throughput as an expression (two-level accumulator): it cannot hand the "
                             "realised state to the fused rolling-update kernel")
        # thr0 of a model the kernel never reads the state from: its own thr_init column (a valid index; never -1, which would
        # run off the array if the field were ever read)
        w.soc_init, w.thr_init, w.soc0, w.thr0 = self.soc_init, self.thr_init, self.soc0, (self.thr_init if self.thr0 is None else self.thr0)
        wc, pt = self.wind_cols.cpu().tolist(), self.pt_cols.cpu().tolist()
        pda = self.pda_cols.cpu().tolist() if hasattr(self, "pda_cols") else []
        trk = self.track_rows.cpu().tolist() if hasattr(self, "track_rows") else []
        for t in range(8):
            w.wind_cols[t] = wc[t] if t < len(wc) else -1
            w.pt_cols[t][0], w.pt_cols[t][1] = (pt[t] if t < len(pt) else (-1, -1))
            w.pda_cols[t] = pda[t] if t < len(pda) else -1
            w.track_rows[t] = trk[t] if t < len(trk) else -1
        w.wind_kw = self.wind_kw
        if self.c0 is not None:
            w.c0, w.c0_base, w.waste_per_kw = self.c0.data_ptr(), self._c0_base, self._waste_per_kw
        w.status, w.flags = self.out["status"].data_ptr(), self.out["flags"].data_ptr()       # (folded into the loop's flags by the next phase)
        return w

    def power_output(self, x):
        return 1e-3 * x[:, self.pt_cols].sum(dim=2)                                            # [B, T] MW

    def solve(self, B, x0=None, y0=None, primal_weight=None, hour=None):
        """hour: hour of the day of an hourly LP (None: no warm start of the simplex)"""
        opts = self.opts if (hour is None or self.opts is None) else (self.opts_first if hour == 0 else self.opts_warm)
        self.out = self.dlp.solve(B, self.c, self.lb, self.ub, self.rlo if self.lp.m else None,
                                  self.rhi if self.lp.m else None, x0=x0, y0=y0, primal_weight=primal_weight,
                                  options=opts, out=self.out, sync_stats=False, obj_offset=self.c0)
        return self.out


class BatchedWindBatteryDoubleLoop:
    def __init__(self, n_scenarios, device=0, first_scenario=0, series="rts_gmlc_309.npz", stride=17,
                 day_ahead_horizon=48, real_time_horizon=4, tracking_horizon=4, wind_mw=200.0, batt_mw=25.0,
                 price_cap=500.0, warm_start=True, lp_backend=None, use_graphs=True, use_fused=True, record=None, simplex_warm=True, warm_patience=10000):
        """lp_backend: None = the HIP solver on GPU `device`; tests pass a factory lp -> object with DeviceLP.solve's
        signature working on CPU tensors (tests/_highs_solver.py::HighsTensorLP), which runs the SAME window / objective /
        state-hand-off logic without a GPU.
        use_graphs: on the GPU, the day-ahead step and the 24 hour steps of a day (about 45 small device operations + two
        solver launches per hour) are captured into hipGraphs the first time they run and REPLAYED for every later day: the
        loop is launch-bound otherwise (one simulated day of 1024 plants: 23 ms issued from Python).  Everything a step
        reads or writes therefore lives in persistent device tensors that are updated in place, including the clock.
        use_fused: on the GPU, the ~45 element-wise tensor operations of an hour step (price / capacity-factor windows,
        objective and bound rewrites, real-time offer -> dispatch rows, realised state, revenue) are THREE launches of one HIP
        kernel (dsp_wb_rolling_update, include/dsp_hip.h), bit-identical to the tensor operations they replace.
        record: (plants, days) - keep, for these plants of the batch (local indices) and up to `days` simulated days, the state every
        LP was built from and its whole solution, hour by hour, in device buffers (what the reference's record_results keeps per hour,
        wind_battery_double_loop.py:276-340): `recorded()` returns them.  The writes are indexed by the device clock, so they are part
        of the captured day like everything else."""
        import torch
        from .workflow import Tracker
        self.B = B = int(n_scenarios)
        self.dev = dev = torch.device("cuda", device) if lp_backend is None else torch.device("cpu")
        s = scenarios.load_series(series)
        self.N = N = len(s["rt_lmp"])
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float64), device=dev)
        self.da_series = t(np.clip(s["da_lmp"], 0.0, price_cap))
        self.rt_series = t(np.clip(s["rt_lmp"], 0.0, price_cap))
        self.cf_series = t(s["rt_cf"])
        ids = first_scenario + np.arange(B)
        self.start = torch.as_tensor((stride * ids) % N, dtype=torch.int64, device=dev)
        self.T_da, self.T_rt, self.T_tr = day_ahead_horizon, real_time_horizon, tracking_horizon
        # ---- the three LPs, built ONCE through the product's own model objects (B = 1 templates) --------------------
        bidder, da_model = scenarios.wind_battery_batch(1, day_ahead_horizon, _NoSolver(), series=series, stride=stride,
                                                        wind_mw=wind_mw, batt_mw=batt_mw, price_cap=price_cap)
        rt_model = bidder.real_time_model
        if real_time_horizon != 4:
            raise NotImplementedError("real_time_horizon is 4 in every reference driver")
        tracker = Tracker(tracking_model_object=bidder.bidding_model_object.__class__(
            model_data=bidder.bidding_model_object.model_data, wind_capacity_factors=list(s["rt_cf"][:tracking_horizon]),
            wind_pmax_mw=wind_mw, battery_pmax_mw=batt_mw, battery_energy_capacity_mwh=4 * batt_mw),
            tracking_horizon=tracking_horizon, n_tracking_hour=1, solver=_NoSolver())
        tracker._pass_market_dispatch([0.0] * tracking_horizon)           # dispatch rows become equalities
        tr_model = tracker.model
        self.penalty = float(bidder.real_time_underbid_penalty)
        self.da = _DeviceModel(da_model, B, dev, device, hints=getattr(da_model, "solver_hints", None), lp_backend=lp_backend,
                               waste_cost_per_kw=bidder.bidding_model_object.wind_waste_penalty * 1e-3,
                               cf_template_sum=float(np.sum(s["rt_cf"][:day_ahead_horizon])))
        waste = bidder.bidding_model_object.wind_waste_penalty * 1e-3
        self.rt = _DeviceModel(rt_model, B, dev, device, hints=getattr(rt_model, "solver_hints", None), lp_backend=lp_backend,
                               waste_cost_per_kw=waste, cf_template_sum=float(np.sum(s["rt_cf"][:real_time_horizon])))
        self.tr = _DeviceModel(tr_model, B, dev, device, hints=getattr(tr_model, "solver_hints", None), lp_backend=lp_backend,
                               waste_cost_per_kw=waste, cf_template_sum=float(np.sum(s["rt_cf"][:tracking_horizon])))
        if self.tr.thr0 is None:
            raise ValueError("the tracking model must carry period 0's throughput as a column: the loop reads the realised state from it")
        idx = lambda cols: torch.as_tensor(np.asarray(cols, np.int64), device=dev)
        self.da.pda_cols, self.rt.pda_cols = idx(da_model.pda_cols), idx(rt_model.pda_cols)
        self.da.u_cols, self.rt.u_cols = (idx([v.index for v in mm.real_time_underbid_power]) for mm in (da_model, rt_model))
        # column handles of the hourly models' periods (tests map device solutions into the oracle's variables through these)
        self.rt_periods, self.tr_periods = rt_model.block.windBattery["periods"], tr_model.block.windBattery["periods"]
        self.da_periods = da_model.block.windBattery["periods"]
        # Rolling warm start of the day-ahead LP (warm_start=True): day d + 1's 48-h problem is day d's shifted by 24 h, so period t
        # starts from yesterday's period t + 24 (the last 24 periods keep their own old values); x, y and the primal weight
        # stay on the device (dsp_batch::x0 / y0 / primal_weight) in PERSISTENT buffers that start at zero - which is the cold
        # start (x = clamp(0), y = 0, weight 0 = automatic) - so the first day needs no special case and the day-ahead step is ONE
        # hipGraph for every day.  ON by default since the end of round 6, ON PATIENCE (dsp_options::warm_patience = 10 000 iterations,
        # then the cold point inside the same launch): 8192 plants 22.2 -> 19.1 ms per simulated day, mean 3650 -> 2210 iterations,
        # slowest plant-day of 60 days 42 880 -> 27 072, all optimal (profiles/r69a_warm_patience.log).  Why it was OFF before: round 2 measured it far worse (15 k iterations against 4 k: before the variable
        # scaling and the objective-error termination).  Round 3 (profiles/r30n_warm_start_60d.log, r30n_double_loop.jsonl: 1024
        # plants, 60 days): the MEAN drops to 2248 iterations from 3564 (63 %), but a day of the loop is ONE batch whose time is its
        # slowest plant, and the tail gets heavier - mean daily maximum 10.9 k against 8.5 k iterations, and one plant-day of 61 440
        # ran into the iteration limit: 14.1 ms per simulated day against 11.8 ms cold.  A caller that pipelines many batches
        # (throughput, not latency) gets the mean.  warm_start="weight" carries the primal weight only (3.0-3.2 k, r30m).
        from .hip_solver import period_shift_maps
        self.weight_only = warm_start == "weight"
        self.warm_start = bool(warm_start) and not self.weight_only and day_ahead_horizon > 24
        cmap, rmap = period_shift_maps(da_model.lp, 24)
        self.da_cmap, self.da_rmap = idx(cmap), idx(rmap)
        if self.warm_start:
            # ... on patience (dsp_options::warm_patience, ABI 12): a plant whose day-ahead solve has not finished after `warm_patience`
            # iterations from yesterday's point starts again from the cold point inside the same launch
            if lp_backend is None and self.da.opts is not None:
                import os
                self.da.opts.warm_patience = int(os.environ.get("DSP_WARM_PATIENCE", warm_patience))
            self.da_x0 = torch.zeros((B, da_model.lp.n), dtype=torch.float64, device=dev)
            self.da_y0 = torch.zeros((B, max(da_model.lp.m, 1)), dtype=torch.float64, device=dev)
        self.tr.track_rows = idx([tr_model.block.kept_row_index(r) for r in tr_model.tracking_rows])
        self.tr.c[:] = t(tr_model.c[0])
        # ---- realised state + annual accumulators (device) -----------------------------------------------------------
        z = lambda: torch.zeros(B, dtype=torch.float64, device=dev)
        self.soc, self.thr = z(), z()
        self.revenue, self.energy_mwh, self.da_energy_mwh = z(), z(), z()
        self.bad = torch.zeros((), dtype=torch.bool, device=dev)          # any non-optimal status so far
        self.uncertified = torch.zeros((), dtype=torch.int64, device=dev)  # solves accepted with DSP_FLAG_OBJ_WAIVED so far
        self.hour = 0
        self.hour_t = torch.zeros((), dtype=torch.int64, device=dev)      # the clock ON THE DEVICE (graphs replay across days)
        self.da_offer = torch.zeros((B, 24), dtype=torch.float64, device=dev)
        self.da_prices = torch.zeros((B, 24), dtype=torch.float64, device=dev)
        self.da_pw = torch.zeros(B, dtype=torch.float64, device=dev)
        self.delivered = z()
        self._hundred = torch.full((), 100.0, dtype=torch.float64, device=dev)
        self.solves = 0
        self.use_graphs = bool(use_graphs) and lp_backend is None
        self.simplex_warm = bool(simplex_warm) and lp_backend is None      # hourly LPs start from the previous hour's basis (_DeviceModel.solve)
        self.use_fused = bool(use_fused) and lp_backend is None and real_time_horizon <= 8 and tracking_horizon <= 8
        if self.use_fused:
            from .hip_solver import DspWbState, load_library
            self._lib = load_library()
            st = DspWbState()
            st.B, st.N = B, N
            st.start, st.hour = self.start.data_ptr(), self.hour_t.data_ptr()
            st.da_series, st.rt_series, st.cf_series = self.da_series.data_ptr(), self.rt_series.data_ptr(), self.cf_series.data_ptr()
            st.soc, st.thr = self.soc.data_ptr(), self.thr.data_ptr()
            st.da_offer, st.da_prices = self.da_offer.data_ptr(), self.da_prices.data_ptr()
            st.delivered, st.revenue, st.energy_mwh = self.delivered.data_ptr(), self.revenue.data_ptr(), self.energy_mwh.data_ptr()
            st.bad, st.uncertified = self.bad.data_ptr(), self.uncertified.data_ptr()
            self._wb_state, self._wb_rt, self._wb_tr = st, self.rt.wb_struct(needs_state=False), self.tr.wb_struct()
        self._graphs = {}                                                  # "da" / hour of day -> captured hipGraph
        self._rec = None
        if record is not None:
            plants, days = record
            R, H = len(plants), 24 * int(days)
            buf = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)
            self._rec = dict(plants=idx(plants), days=int(days),
                             state=buf(H + 1, R, 2), rt_x=buf(H + 1, R, self.rt.lp.n), tr_x=buf(H + 1, R, self.tr.lp.n),
                             rt_obj=buf(H + 1, R), tr_obj=buf(H + 1, R),
                             da_state=buf(int(days) + 1, R, 2), da_x=buf(int(days) + 1, R, self.da.lp.n), da_obj=buf(int(days) + 1, R))
            self._day_t = torch.zeros((), dtype=torch.int64, device=dev)
            self._limit_h = torch.full((), H, dtype=torch.int64, device=dev)      # (rows past the recorded span land in the spare last row)
            self._limit_d = torch.full((), int(days), dtype=torch.int64, device=dev)

    def _record(self, what, value, daily=False):
        """rec[what][clock] = value[plants]   (clock read on the device: capturable)"""
        import torch
        r = self._rec
        at = torch.minimum(torch.div(self.hour_t, 24, rounding_mode="floor"), self._limit_d) if daily else torch.minimum(self.hour_t, self._limit_h)
        r[what].index_copy_(0, at.view(1), value.index_select(0, r["plants"]).unsqueeze(0))

    def recorded(self):
        """-> dict of host arrays [hours or days, plants, ...] of the recorded span (record=... at construction)"""
        r = self._rec
        H, D = min(self.hour, 24 * r["days"]), min(self.hour // 24, r["days"])
        out = {k: r[k][:H].cpu().numpy() for k in ("state", "rt_x", "tr_x", "rt_obj", "tr_obj")}
        out.update({k: r[k][:D].cpu().numpy() for k in ("da_state", "da_x", "da_obj")})
        out["plants"] = r["plants"].cpu().numpy()
        return out

    def reset(self):
        """Back to hour 0 with an empty battery and zeroed accumulators; handles, buffers and captured graphs are kept (they refer to
        persistent tensors only), so a warmed-up loop can be timed from the first hour of the year."""
        for t in (self.soc, self.thr, self.revenue, self.energy_mwh, self.da_energy_mwh, self.da_pw, self.delivered, self.hour_t, self.uncertified,
                  self.da_offer, self.da_prices):
            t.zero_()
        self.bad.zero_()
        if self.warm_start:
            self.da_x0.zero_(), self.da_y0.zero_()
        self.hour = self.solves = 0

    # -- windows -----------------------------------------------------------------------------------------------------------
    def _window(self, series, T):
        """[B, T] window of `series` that starts at the current hour of every plant (clock read on the device)"""
        import torch
        idx = (self.start[:, None] + self.hour_t + torch.arange(T, device=self.dev)[None, :]) % self.N
        return series[idx]

    def _set_prices(self, m, da, rt):
        """c = base - RT x dP_T/dx - (DA - RT) on day_ahead_power   (Bidder._pass_price_forecasts, on the device)"""
        m.c[:] = m.base_c
        m.c[:, m.pt_cols[:, 0]] -= 1e-3 * rt
        m.c[:, m.pt_cols[:, 1]] -= 1e-3 * rt
        m.c[:, m.pda_cols] -= da - rt

    def _set_state(self, m):
        """What update_model writes: initial SOC / throughput fixed to the realised values, wind availability of the window"""
        m.lb[:, m.soc_init] = self.soc
        m.ub[:, m.soc_init] = self.soc
        m.lb[:, m.thr_init] = self.thr
        m.ub[:, m.thr_init] = self.thr
        avail = m.wind_kw * self._window(self.cf_series, m.T)
        m.ub[:, m.wind_cols] = avail
        if m.c0 is not None:
            total = avail[:, 0]
            for t in range(1, m.T):                # (in the order of the fused kernel's loop: bit-identical constants)
                total = total + avail[:, t]
            m.c0.copy_(m._c0_base + m._waste_per_kw * total)

    def _check(self, out):
        self.bad |= (out["status"] != 0).any()
        # accepted without a certified objective accuracy (DSP_FLAG_OBJ_WAIVED): counted, on the device (the loop is replayed
        # from hipGraphs: no host round trip to re-solve them here); results() reports the count next to `ok`
        if out.get("flags") is not None:
            self.uncertified += ((out["flags"] & 1) != 0).sum()

    # -- one simulated day -------------------------------------------------------------------------------------------------
    def _day_ahead_step(self):
        """Device work of the day-ahead bids (capturable: reads / writes persistent tensors only)."""
        import torch
        m = self.da
        da, rt = self._window(self.da_series, m.T), self._window(self.rt_series, m.T)
        self._set_prices(m, da, rt)
        self._set_state(m)
        m.lb.index_fill_(1, m.pda_cols, 0.0)          # (index_fill_, not lb[:, cols] = 0.0: a Python scalar on the right-hand
        m.ub.index_fill_(1, m.pda_cols, float("inf"))  #  side becomes a host-to-device copy, which a graph capture refuses)
        if self.warm_start:
            out = m.solve(self.B, x0=self.da_x0, y0=self.da_y0, primal_weight=self.da_pw)
            torch.index_select(out["x"], 1, self.da_cmap, out=self.da_x0)
            torch.index_select(out["y"], 1, self.da_rmap, out=self.da_y0)
        else:
            if not self.weight_only:
                self.da_pw.zero_()
            out = m.solve(self.B, primal_weight=self.da_pw)
        self._check(out)
        if self._rec is not None:
            self._record("da_state", torch.stack([self.soc, self.thr], 1), daily=True)
            self._record("da_x", out["x"], daily=True)
            self._record("da_obj", out["obj"], daily=True)
        self.da_offer.copy_(out["x"][:, m.pda_cols][:, :24])
        self.da_prices.copy_(da[:, :24])
        self.da_energy_mwh += self.da_offer.sum(1)

    def _fused(self, phase, k):
        import ctypes as C
        import torch
        rc = self._lib.dsp_wb_rolling_update(C.byref(self._wb_state), C.byref(self._wb_rt), C.byref(self._wb_tr), phase, k,
                                             C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"dsp_wb_rolling_update failed ({rc})")

    def _hour_step(self, k):
        """Device work of hour k of the day: real-time bid, stub clearing, tracking, state hand-off, clock (capturable)."""
        import torch
        if self.use_fused:
            if self._rec is not None:
                self._record("state", torch.stack([self.soc, self.thr], 1))
            self._fused(0, k)
            self.rt.solve(self.B, hour=k if self.simplex_warm else None)   # (status / flags of the two solves: checked by the kernel's next phase)
            self._fused(1, k)
            self.tr.solve(self.B, hour=k if self.simplex_warm else None)
            if self._rec is not None:
                for key, m in (("rt", self.rt), ("tr", self.tr)):
                    self._record(key + "_x", m.out["x"])
                    self._record(key + "_obj", m.out["obj"])
            self._fused(2, k)
            return
        m = self.rt
        if self._rec is not None:
            self._record("state", torch.stack([self.soc, self.thr], 1))
        rt = self._window(self.rt_series, m.T)
        da = self._window(self.da_series, m.T).clone()
        known = min(m.T, 24 - k)                                         # hours of the horizon inside the cleared day
        da[:, :known] = self.da_prices[:, k:k + known]
        self._set_prices(m, da, rt)
        self._set_state(m)
        m.lb.index_fill_(1, m.pda_cols, 0.0)          # (index_fill_, not lb[:, cols] = 0.0: a Python scalar on the right-hand
        m.ub.index_fill_(1, m.pda_cols, float("inf"))  #  side becomes a host-to-device copy, which a graph capture refuses)
        m.lb[:, m.pda_cols[:known]] = self.da_offer[:, k:k + known]
        m.ub[:, m.pda_cols[:known]] = self.da_offer[:, k:k + known]
        hour = k if self.simplex_warm else None
        out = m.solve(self.B, hour=hour) if self.rt.opts is not None else m.solve(self.B)
        self._check(out)
        offer = m.power_output(out["x"])                                 # real-time offer = SCED dispatch in the stub market
        # tracking
        tr = self.tr
        self._set_state(tr)
        tr.rlo[:, tr.track_rows] = offer[:, :tr.T]
        tr.rhi[:, tr.track_rows] = offer[:, :tr.T]
        out = tr.solve(self.B, hour=hour) if self.tr.opts is not None else tr.solve(self.B)
        self._check(out)
        if self._rec is not None:
            for key, mm in (("rt", self.rt), ("tr", self.tr)):
                self._record(key + "_x", mm.out["x"])
                self._record(key + "_obj", mm.out["obj"])
        x = out["x"]
        self.delivered.copy_(tr.power_output(x)[:, 0])
        # implemented profile -> next hour's initial state, rounded to 2 dp as update_model does
        # (divided by a TENSOR: a Python-scalar divisor makes torch multiply by the rounded reciprocal on the GPU, 1 ulp off
        # the correctly rounded quotient that Python's round(x, 2) and the fused kernel produce)
        self.soc.copy_(torch.round(x[:, tr.soc0] * 100.0) / self._hundred)
        self.thr.copy_(torch.round(x[:, tr.thr0] * 100.0) / self._hundred)
        self.revenue += self.delivered * rt[:, 0] + self.da_offer[:, k] * (self.da_prices[:, k] - rt[:, 0])
        self.energy_mwh += self.delivered
        self.hour_t += 1

    def _run(self, key, fn):
        """Run one step: eagerly, or - with use_graphs - captured once into a hipGraph and replayed from then on."""
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

    _warm = False          # the first day runs eagerly (handles, output buffers and kernels get created), then graphs

    def day_ahead(self):
        """Day-ahead bids of every plant for the day that starts at self.hour: returns the offers [B, 24] (= cleared dispatch)."""
        self.day_start = self.hour
        self._run("da", self._day_ahead_step)
        self.solves += self.B
        return self.da_offer.clone()

    def hour_step(self):
        """Real-time bid, stub clearing, tracking and state hand-off of ONE hour for every plant (all on the device)."""
        k = self.hour - self.day_start                                  # hour of the day
        self._run(k, lambda: self._hour_step(k))
        self.solves += 2 * self.B
        self.hour += 1
        return self.delivered.clone()

    def run_day(self):
        self.day_ahead()
        for _ in range(24):
            self.hour_step()
        self._warm = True

    def results(self):
        """Per-scenario totals so far (device tensors) + whether every solve was optimal (one device->host sync)."""
        return dict(obj=self.revenue, energy_mwh=self.energy_mwh, soc=self.soc, throughput=self.thr), not bool(self.bad.item())


class PipelinedDoubleLoops:
    """`n_scenarios` plants as `groups` independent BatchedWindBatteryDoubleLoop objects on as many HIP streams.

    A simulated day of ONE loop is a chain of lone batches: its day-ahead solve ends with its slowest plant (up to 12 x the mean
    iteration count) while the chip idles, and nothing of the hourly steps can start before it has.  Plants do not interact, so
    the groups' days overlap - one group's day-ahead tail runs beside the others' hourly steps.  One MI355X, 8192 plants
    (profiles/r50f_double_loop_groups.log): 44.2 ms per simulated day as one loop, 35.4 ms as two, 53 - 57 ms as four or eight (smaller
    batches per launch, 25 graph replays per group and day); two groups against one at 1024 / 2048 / 4096 plants: 12.9 / 17.8 / 24.0 ms
    against 13.5 / 19.9 / 28.1 (r50g).  `groups=0` picks two from 1024 plants on, else one."""

    def __init__(self, n_scenarios, device=0, first_scenario=0, groups=0, record=None, **kw):
        import torch
        from .distributed import shard_bounds
        n = int(n_scenarios)
        G = int(groups) if groups and groups > 0 else (2 if n >= 1024 else 1)
        self.groups = G = max(1, min(G, max(n, 1)))
        self.dev = torch.device("cuda", device)
        cuts = [shard_bounds(n, G, g) for g in range(G)]
        # record = (plants of THIS object's batch, days): every group records its own (BatchedWindBatteryDoubleLoop.recorded)
        rec = lambda b0, b1: None if record is None else ([p - b0 for p in record[0] if b0 <= p < b1], record[1])
        self.record_order = None if record is None else [p for b0, b1 in cuts for p in record[0] if b0 <= p < b1]
        self.loops = [BatchedWindBatteryDoubleLoop(b1 - b0, device=device, first_scenario=first_scenario + b0, record=rec(b0, b1), **kw) for b0, b1 in cuts]
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(G)] if G > 1 else [None]

    def run_day(self):
        import torch
        if self.groups == 1:
            return self.loops[0].run_day()
        cur = torch.cuda.current_stream(self.dev)
        for loop, s in zip(self.loops, self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                loop.run_day()
        for s in self.streams:
            cur.wait_stream(s)

    def run_days(self, n, per_day=None):
        """`n` simulated days with the groups FREE-RUNNING: every group enqueues its days on its own stream and the groups are joined
        once, at the end - `run_day` joins them after every day, so each day costs what its slowest group costs (the day-ahead solve
        ends with its slowest plant: 9 - 27 k iterations at a mean of 2.2 k).  Plants do not interact; the results are those of
        `run_day` bit for bit.  `per_day(g, loop)`, if given, is called after each day of group g inside that group's stream.
        8192 plants, one MI355X, the whole year (profiles/r70o_year_free.log, r70p_year_groups_366.log): 2 groups 7.02 -> 6.86 s; 4 groups
        are faster over 60 days (18.4 ms per day) and slower over the year (25 ms: ONE host thread feeds all streams, and once a
        group's queue is full it waits on that group while the others run dry)."""
        import torch
        if self.groups == 1:
            for _ in range(n):
                self.loops[0].run_day()
                if per_day is not None:
                    per_day(0, self.loops[0])
            return
        cur = torch.cuda.current_stream(self.dev)
        for s in self.streams:
            s.wait_stream(cur)
        for _ in range(n):
            for g, (loop, s) in enumerate(zip(self.loops, self.streams)):
                with torch.cuda.stream(s):
                    loop.run_day()
                    if per_day is not None:
                        per_day(g, loop)
        for s in self.streams:
            cur.wait_stream(s)

    def reset(self):
        for l in self.loops:
            l.reset()

    def recorded(self):
        """recorded arrays of all groups, plants in the order of `record_order` (batch indices of this object)"""
        parts = [l.recorded() for l in self.loops if l._rec is not None and len(l._rec["plants"])]
        out = {k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0] if k != "plants"}
        out["plants"] = np.asarray(self.record_order)
        return out

    @property
    def hour(self):
        return self.loops[0].hour

    @property
    def revenue(self):
        import torch
        return self.loops[0].revenue if self.groups == 1 else torch.cat([l.revenue for l in self.loops])

    @property
    def uncertified(self):
        return sum(l.uncertified for l in self.loops)

    @property
    def warm_start(self):
        return self.loops[0].warm_start

    def day_ahead_iterations(self):
        import torch
        return torch.cat([l.da.out["iters"] for l in self.loops])

    def results(self):
        import torch
        parts = [l.results() for l in self.loops]
        keys = parts[0][0].keys()
        return {k: torch.cat([p[0][k] for p in parts]) for k in keys}, all(p[1] for p in parts)
