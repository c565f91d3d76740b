"""Wind resource of the reference's wind plant: hourly capacity factor from an hourly wind speed.

The reference's price-taker studies configure ``Wind_Power`` with ``resource_speed`` (one wind speed per period,
``dispatches/case_studies/renewables_case/wind_battery_LMP.py:160-168``, ``tests/test_RE_flowsheet.py:33-40``) and
``Wind_Power.setup_resource`` (``dispatches/unit_models/wind_power.py:163-177``) turns every speed into a capacity factor by
running NREL-PySAM's Windpower module: ATB 2018 market-average 5 MW turbine (power curve on integer wind speeds 0 .. 27 m/s,
``wind_power.py:128-143``), one turbine, Weibull resource model with shape factor 100 and the reference height at the hub - a
distribution so narrow that it is "this speed" -, SAM's default loss categories.  PySAM is a third-party dependency that this
framework does not carry; what that configuration computes is a closed form:

    lambda = v / Gamma(1 + 1/k)                                   (Weibull scale for mean speed v, k = 100)
    p_i    = F(ws_i + 0.125) - F(ws_{i-1} + 0.125),  F(w) = 1 - exp(-(w / lambda)^k)       (SSC bins the curve's speeds +- half of
                                                                                              SAM's 0.25 m/s default step)
    cf     = (1 - losses) * sum_i p_i * P(ws_i) / P_rated

i.e. a speed inside (ws_{i-1} + 0.12, ws_i + 0.12) produces (almost exactly) the curve's value at ws_i - a staircase - with a
smooth hand-over a few hundredths of a m/s wide at the bin edges.  `losses` is the product of SAM's default wind-farm loss
percentages (16.56 %).  Checked against the reference's own numbers in tests/test_lp_flatten.py (unit-model known answers
30083.39 kW and 0.5755) and through the price-taker goldens (tests/test_hip_stream.py).
"""
from __future__ import annotations

import math

import numpy as np

# wind_power.py:136-138 (kW at 0, 1, ..., 27 m/s)
POWER_CURVE_KW = np.array([0, 0, 0, 40.5, 177.7, 403.9, 737.6, 1187.2, 1771.1, 2518.6, 3448.4, 4562.5] + [5000.0] * 14 + [0, 0])
RATED_KW = 5000.0
WEIBULL_K = 100.0                      # wind_power.py:171
BIN_HALF_WIDTH = 0.125                 # m/s
# SAM "WindpowerSingleowner" defaults, percent: availability (BOP, grid, turbine), electrical (efficiency, parasitic),
# environmental (degradation, environmental, exposure, icing), operations (environmental, grid, load, strategies), turbine
# (generic, hysteresis, performance, site-specific), wake (external, future, internal)
DEFAULT_LOSSES_PERCENT = {"avail_bop": 0.5, "avail_grid": 1.5, "avail_turb": 3.58, "elec_eff": 1.91, "elec_parasitic": 0.1,
                          "env_degrad": 1.8, "env_env": 0.4, "env_exposure": 0.0, "env_icing": 0.21, "ops_env": 1.0, "ops_grid": 0.84,
                          "ops_load": 0.99, "ops_strategies": 0.0, "turb_generic": 1.7, "turb_hysteresis": 0.4, "turb_perf": 1.1,
                          "turb_specific": 0.81, "wake_ext": 1.1, "wake_future": 0.0, "wake_int": 0.0}


def loss_factor(losses_percent=None) -> float:
    keep = 1.0
    for pct in (DEFAULT_LOSSES_PERCENT if losses_percent is None else losses_percent).values():
        keep *= 1.0 - pct / 100.0
    return keep


def capacity_factor_from_speed(speed_m_s, losses_percent=None):
    """Capacity factor(s) for hourly mean wind speed(s) at hub height [m/s] (`resource_speed` of the reference)."""
    v = np.atleast_1d(np.asarray(speed_m_s, dtype=float))
    scale = v / math.gamma(1.0 + 1.0 / WEIBULL_K)                          # [N]
    upper = np.arange(len(POWER_CURVE_KW)) + BIN_HALF_WIDTH                # upper edge of bin i
    with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
        cdf = -np.expm1(-np.power(upper[None, :] / scale[:, None], WEIBULL_K))     # [N, bins]
    cdf = np.where(scale[:, None] > 0.0, cdf, 1.0)                          # no wind: all mass in the first bin (0 kW)
    mass = np.diff(cdf, axis=1, prepend=0.0)
    cf = loss_factor(losses_percent) * (mass @ POWER_CURVE_KW) / RATED_KW
    return cf if np.ndim(speed_m_s) else float(cf[0])


def capacity_factor_from_distribution_point(speed_m_s, losses_percent=None) -> float:
    """The single-point `resource_probability_density` configuration (wind_power.py:147-162): the curve interpolated at the
    speed, same losses."""
    return loss_factor(losses_percent) * float(np.interp(speed_m_s, np.arange(len(POWER_CURVE_KW)), POWER_CURVE_KW)) / RATED_KW
