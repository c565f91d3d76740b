"""Generator model data containers (stand-ins for idaes.apps.grid_integration.model_data.*).

Field names follow the dictionaries the reference passes in
(``run_double_loop_battery.py:126-158``, ``test_multiperiod_wind_battery_doubleloop.py:52-60,195-213``).
Iterating a model-data object yields ``(param, value)`` pairs, which is what
``DoubleLoopCoordinator._update_static_params`` consumes (``dispatches/workflow/coordinator.py:54``).
"""
from __future__ import annotations


class GeneratorModelData:
    generator_type = None
    _fields = ("gen_name", "bus", "p_min", "p_max", "p_cost", "fixed_commitment")

    def __init__(self, gen_name, bus, p_min, p_max, p_cost=0, fixed_commitment=None, **kw):
        if p_min > p_max:
            raise ValueError(f"p_min ({p_min}) must not exceed p_max ({p_max})")
        self.gen_name, self.bus = gen_name, bus
        self.p_min, self.p_max = float(p_min), float(p_max)
        self.p_cost = p_cost
        self.fixed_commitment = fixed_commitment
        if kw:
            raise TypeError(f"unexpected generator parameters: {sorted(kw)}")

    def __iter__(self):
        for name in self._fields:
            yield name, getattr(self, name)


class RenewableGeneratorModelData(GeneratorModelData):
    generator_type = "renewable"


class ThermalGeneratorModelData(GeneratorModelData):
    generator_type = "thermal"
    _fields = GeneratorModelData._fields + (
        "min_down_time", "min_up_time", "ramp_up_60min", "ramp_down_60min", "shutdown_capacity",
        "startup_capacity", "initial_status", "initial_p_output", "startup_cost", "startup_fuel")

    def __init__(self, gen_name, bus, p_min, p_max, min_down_time, min_up_time, ramp_up_60min, ramp_down_60min,
                 shutdown_capacity, startup_capacity, initial_status=1, initial_p_output=0.0,
                 production_cost_bid_pairs=None, startup_cost_pairs=None, fixed_commitment=None,
                 include_default_p_cost=True):
        self.production_cost_bid_pairs = production_cost_bid_pairs
        self.include_default_p_cost = include_default_p_cost
        p_cost = self._default_p_cost(p_min, p_max, production_cost_bid_pairs)
        super().__init__(gen_name, bus, p_min, p_max, p_cost, fixed_commitment)
        self.min_down_time, self.min_up_time = min_down_time, min_up_time
        self.ramp_up_60min, self.ramp_down_60min = ramp_up_60min, ramp_down_60min
        self.shutdown_capacity, self.startup_capacity = shutdown_capacity, startup_capacity
        self.initial_status, self.initial_p_output = initial_status, initial_p_output
        self.startup_cost = list(startup_cost_pairs) if startup_cost_pairs is not None else [(min_down_time, 0.0)]
        self.startup_fuel = None

    @staticmethod
    def _default_p_cost(p_min, p_max, pairs):
        """Default marginal-cost bid curve [(MW, $/MWh)...] sorted by power, must start at p_min."""
        if pairs is None:
            return [(p_min, 0.0), (p_max, 0.0)]
        pairs = sorted((float(p), float(c)) for p, c in pairs)
        if pairs[0][0] != p_min or pairs[-1][0] != p_max:
            raise ValueError("production_cost_bid_pairs must span [p_min, p_max]")
        return pairs

    @property
    def default_bids(self):
        return {p: c for p, c in self.p_cost}
