"""DoubleLoopCoordinator: glue between Prescient's plugin callbacks and the bidder / trackers.

API mirror of ``dispatches/workflow/coordinator.py:21-93`` (``PrescientPluginModule``, ``DoubleLoopCoordinator``
with ``register_plugins``, ``prescient_plugin_module``, ``update_static_params``) PLUS the upstream idaes
``DoubleLoopCoordinator`` base behaviour the reference inherits: ``bid_into_DAM``,
``bid_into_RTM``, ``track_sced_signal``, ``update_observed_dispatch``, ``write_plugin_results``.  Prescient itself
is not part of this package; the callbacks operate on Prescient's plain ``instance.data`` dictionaries, so they can
be driven by the real simulator or by a stub context.
"""
from __future__ import annotations

from types import ModuleType

from .utils import convert_marginal_costs_to_actual_costs


class PrescientPluginModule(ModuleType):
    def __init__(self, get_configuration, register_plugins):
        self.get_configuration = get_configuration
        self.register_plugins = register_plugins


class DoubleLoopCoordinator:
    def __init__(self, bidder, tracker, projection_tracker):
        self.bidder = bidder
        self.tracker = tracker
        self.projection_tracker = projection_tracker
        self.current_DA_bids = self.next_DA_bids = self.current_avail_DA_bids = None
        self.current_DA_prices = self.next_DA_prices = None
        self.current_DA_dispatches = self.next_DA_dispatches = None
        self.current_RT_bids = None

    # ---- plugin wiring -----------------------------------------------------------------------------------
    def get_configuration(self, key):
        """Prescient plugin configuration: one option, the generator this coordinator bids for."""
        return {"bidding_generator": {"domain": str, "default": None,
                                      "description": "Specifies the generator we derive bidding strategis for."}}

    def register_plugins(self, context, options, plugin_config):
        self.plugin_config = plugin_config
        self._register_initialization_callbacks(context, options, plugin_config)
        self._register_before_ruc_solve_callbacks(context, options, plugin_config)
        self._register_before_operations_solve_callbacks(context, options, plugin_config)
        self._register_after_operations_callbacks(context, options, plugin_config)
        self._register_update_operations_stats_callbacks(context, options, plugin_config)
        self._register_after_ruc_activation_callbacks(context, options, plugin_config)
        self._register_finalization_callbacks(context, options, plugin_config)
        # the three callbacks DISPATCHES adds (reference coordinator.py:32-40)
        context.register_after_get_initial_actuals_model_for_sced_callback(self.update_static_params)
        context.register_after_get_initial_actuals_model_for_simulation_actuals_callback(self.update_static_params)
        context.register_after_get_initial_forecast_model_for_ruc_callback(self.update_static_params)

    def _register_initialization_callbacks(self, context, options, plugin_config):
        context.register_initialization_callback(self.initialize_customized_results)

    def _register_before_ruc_solve_callbacks(self, context, options, plugin_config):
        context.register_before_ruc_solve_callback(self.bid_into_DAM)

    def _register_before_operations_solve_callbacks(self, context, options, plugin_config):
        context.register_before_operations_solve_callback(self.bid_into_RTM)

    def _register_after_operations_callbacks(self, context, options, plugin_config):
        context.register_after_operations_callback(self.track_sced_signal)

    def _register_update_operations_stats_callbacks(self, context, options, plugin_config):
        context.register_update_operations_stats_callback(self.update_observed_dispatch)

    def _register_after_ruc_activation_callbacks(self, context, options, plugin_config):
        context.register_after_ruc_activation_callback(self.activate_pending_DA_bids)
        context.register_after_ruc_generation_callback(self.fetch_DA_results)

    def _register_finalization_callbacks(self, context, options, plugin_config):
        context.register_finalization_callback(self.write_plugin_results)

    @property
    def prescient_plugin_module(self):
        return PrescientPluginModule(self.get_configuration, self.register_plugins)

    # ---- static parameters (reference coordinator.py:46-93) -----------------------------------------------
    def _update_static_params(self, gen_dict):
        md = self.bidder.bidding_model_object.model_data
        is_thermal = md.generator_type == "thermal"
        is_renewable = md.generator_type == "renewable"
        for param, value in md:
            if param == "gen_name" or value is None:
                continue
            elif (param in gen_dict and isinstance(gen_dict[param], dict)
                  and gen_dict[param]["data_type"] == "time_series"):
                continue          # time-varying entries are written by the bids
            elif param == "p_cost":
                if is_thermal:
                    gen_dict[param] = {"data_type": "cost_curve", "cost_curve_type": "piecewise",
                                       "values": convert_marginal_costs_to_actual_costs(value)}
                elif is_renewable:
                    gen_dict[param] = value
                else:
                    raise NotImplementedError("generator_type must be either 'thermal' or 'renewable'")
            else:
                gen_dict[param] = value

    def update_static_params(self, options, instance):
        gen_name = self.bidder.bidding_model_object.model_data.gen_name
        self._update_static_params(instance.data["elements"]["generator"][gen_name])

    def pass_static_params_to_DA(self, *args, **kwargs):
        pass

    def pass_static_params_to_RT(self, *args, **kwargs):
        pass

    # ---- day-ahead loop ----------------------------------------------------------------------------------
    def initialize_customized_results(self, options, simulator):
        simulator.data_manager.extensions["customized_results"] = {}

    def _pass_DA_bid_to_prescient(self, options, ruc_instance, bids):
        """Write the time-varying bid entries into the RUC instance's generator dict."""
        gen_name = self.bidder.bidding_model_object.model_data.gen_name
        gen_dict = ruc_instance.data["elements"]["generator"][gen_name]
        hours = sorted(bids)
        for param in next(iter(bids[hours[0]].values())):
            values = [bids[t][gen_name][param] for t in hours]
            if param == "p_cost":
                gen_dict[param] = {"data_type": "time_series",
                                   "values": [{"data_type": "cost_curve", "cost_curve_type": "piecewise",
                                               "values": v} for v in values]}
            else:
                gen_dict[param] = {"data_type": "time_series", "values": values}

    def bid_into_DAM(self, options, simulator, ruc_instance, ruc_date, ruc_hour):
        """Before each RUC solve: project the tracker to the end of the day (day > 0), update the DA model
        with the projected state and submit the DA bids."""
        is_first_day = self.current_DA_bids is None and self.next_DA_bids is None
        if not is_first_day:
            profiles = self._project_tracking_trajectory(options, simulator, ruc_hour)
            self.bidder.update_day_ahead_model(**profiles)
        bids = self.bidder.compute_day_ahead_bids(date=ruc_date, hour=0)
        if is_first_day:
            self.current_DA_bids = self.current_avail_DA_bids = bids
        self.next_DA_bids = bids
        self._pass_DA_bid_to_prescient(options, ruc_instance, bids)
        return bids

    def _project_tracking_trajectory(self, options, simulator, ruc_hour):
        """Clone the tracker state into the projection tracker and run it over the remaining DA dispatches."""
        src, dst = self.tracker, self.projection_tracker
        lb, ub, rlo, rhi = src.model.block.current_bounds()
        for j, v in enumerate(zip(lb, ub)):
            if dst.model.block.col_mutable[j]:
                dst.model.block.col_lb[j], dst.model.block.col_ub[j] = float(v[0]), float(v[1])
        if hasattr(src.model.block, "_time_idx"):
            dst.model.block._time_idx = src.model.block._time_idx
        remaining = list((self.current_DA_dispatches or [])[ruc_hour:])
        profiles_all = {}
        for k in range(0, len(remaining), dst.n_tracking_hour):
            window = remaining[k:k + dst.tracking_horizon]
            profiles = dst.track_market_dispatch(window, date=None, hour=ruc_hour + k)
            dst.update_model(**profiles)
            for key, val in profiles.items():
                profiles_all.setdefault(key, []).extend(val)
        if not profiles_all:
            profiles_all = self.tracker.tracking_model_object.get_implemented_profile(
                b=self.tracker.model.fs, last_implemented_time_step=self.tracker.n_tracking_hour - 1)
        return profiles_all

    def fetch_DA_results(self, options, simulator, ruc_plan, ruc_date, ruc_hour):
        """After RUC generation: keep the cleared DA prices / dispatches and feed the forecaster."""
        md = self.bidder.bidding_model_object.model_data
        market = ruc_plan.ruc_market
        prices = [market.day_ahead_prices.get((md.bus, t)) for t in range(24)]
        dispatches = [market.thermal_gen_cleared_DA.get((md.gen_name, t),
                      getattr(market, "renewable_gen_cleared_DA", {}).get((md.gen_name, t), 0.0)) for t in range(24)]
        if self.current_DA_prices is None:
            self.current_DA_prices, self.current_DA_dispatches = prices, dispatches
        self.next_DA_prices, self.next_DA_dispatches = prices, dispatches
        self.bidder.forecaster.fetch_day_ahead_stats_from_prescient(ruc_date, ruc_hour, market)

    def activate_pending_DA_bids(self, options, simulator):
        self.current_DA_bids = self.next_DA_bids
        self.current_DA_prices, self.current_DA_dispatches = self.next_DA_prices, self.next_DA_dispatches

    # ---- real-time loop ----------------------------------------------------------------------------------
    def bid_into_RTM(self, options, simulator, sced_instance):
        date, hour = simulator.time_manager.current_time.date, simulator.time_manager.current_time.hour
        bids = self.bidder.compute_real_time_bids(
            date=date, hour=hour, realized_day_ahead_prices=self.current_DA_prices,
            realized_day_ahead_dispatches=self.current_DA_dispatches)
        self.current_RT_bids = bids
        self._pass_DA_bid_to_prescient(options, sced_instance, bids)
        return bids

    def track_sced_signal(self, options, simulator, sced_instance, lmp_sced):
        """After SCED: track the cleared dispatch, then push the implemented profile into tracker and bidder."""
        md = self.bidder.bidding_model_object.model_data
        date, hour = simulator.time_manager.current_time.date, simulator.time_manager.current_time.hour
        pg = sced_instance.data["elements"]["generator"][md.gen_name]["pg"]["values"]
        dispatch = list(pg)[: self.tracker.tracking_horizon]
        profiles = self.tracker.track_market_dispatch(market_dispatch=dispatch, date=date, hour=hour)
        self.tracker.update_model(**profiles)
        self.bidder.update_real_time_model(**profiles)
        return profiles

    def update_observed_dispatch(self, options, simulator, ops_stats):
        md = self.bidder.bidding_model_object.model_data
        delivered = self.tracker.get_last_delivered_power()
        target = ops_stats.observed_thermal_dispatch_levels if md.generator_type == "thermal" \
            else ops_stats.observed_renewables_levels
        target[md.gen_name] = delivered
        self.bidder.forecaster.fetch_hourly_stats_from_prescient(ops_stats)

    def write_plugin_results(self, options, simulator):
        self.bidder.write_results(path=options.output_directory)
        self.tracker.write_results(path=options.output_directory)
