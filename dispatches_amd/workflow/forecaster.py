"""Price / capacity-factor forecasters.

``PerfectForecaster`` mirrors ``dispatches/workflow/parametrized_bidder.py:19-70``.
``Backcaster`` restates the upstream idaes ``Backcaster`` used at
``renewables_case/run_double_loop_battery.py:230`` and in the reference tests
(``test_multiperiod_wind_battery_doubleloop.py:128-129``): scenario i is the (i+1)-th most recent stored day,
read forward from `hour` and wrapping over the stored history.  The one-sample behaviour is pinned by the
reference's golden bids; n_samples > 1 is not pinned by any reference vector.
"""
from __future__ import annotations

import numpy as np
import pandas as pd


class AbstractPrescientPriceForecaster:
    def forecast_day_ahead_and_real_time_prices(self, date, hour, bus, horizon, n_samples):
        raise NotImplementedError

    def forecast_day_ahead_prices(self, date, hour, bus, horizon, n_samples):
        raise NotImplementedError

    def forecast_real_time_prices(self, date, hour, bus, horizon, n_samples):
        raise NotImplementedError

    def fetch_hourly_stats_from_prescient(self, prescient_hourly_stats):
        raise NotImplementedError

    def fetch_day_ahead_stats_from_prescient(self, uc_date, uc_hour, day_ahead_result):
        raise NotImplementedError


class Backcaster(AbstractPrescientPriceForecaster):
    def __init__(self, historical_da_prices, historical_rt_prices, max_historical_days=10):
        self.max_historical_days = int(max_historical_days)
        self._historical_da_prices = self._validate(historical_da_prices)
        self._historical_rt_prices = self._validate(historical_rt_prices)
        # RT LMPs of the day in progress, per bus: they join the history only as a WHOLE day (the forecast indexes the
        # history as 24 * day + hour, so a partial day would shift every stored day by the hours appended so far)
        self._current_day_rt_prices = {bus: [] for bus in self._historical_rt_prices}

    def _validate(self, prices):
        if not isinstance(prices, dict):
            raise TypeError("historical prices must be a dict {bus: [prices]}")
        out = {}
        for bus, p in prices.items():
            p = [float(v) for v in p]
            if len(p) < 24:
                raise ValueError(f"at least one day (24 h) of history is required for bus {bus}")
            if len(p) % 24:
                raise ValueError(f"history of bus {bus} must hold whole days (multiples of 24 h)")
            out[bus] = p[-24 * self.max_historical_days:]
        return out

    @property
    def historical_da_prices(self):
        return self._historical_da_prices

    @property
    def historical_rt_prices(self):
        return self._historical_rt_prices

    @staticmethod
    def _forecast(hist, hour, horizon, n_samples):
        hist = np.asarray(hist, float)
        n_days = len(hist) // 24
        t = np.arange(horizon)
        out = {}
        for i in range(n_samples):
            start = 24 * ((n_days - 1 - i) % n_days) + int(hour)
            out[i] = hist[(start + t) % len(hist)].tolist()
        return out

    def forecast_day_ahead_prices(self, date, hour, bus, horizon, n_samples):
        return self._forecast(self._historical_da_prices[bus], hour, horizon, n_samples)

    def forecast_real_time_prices(self, date, hour, bus, horizon, n_samples):
        return self._forecast(self._historical_rt_prices[bus], hour, horizon, n_samples)

    def forecast_day_ahead_and_real_time_prices(self, date, hour, bus, horizon, n_samples):
        return (self.forecast_day_ahead_prices(date, hour, bus, horizon, n_samples),
                self.forecast_real_time_prices(date, hour, bus, horizon, n_samples))

    # -- history roll-over from Prescient -------------------------------------------------------------------
    def _append(self, store, bus, values):
        store.setdefault(bus, [])
        store[bus] = (store[bus] + [float(v) for v in values])[-24 * self.max_historical_days:]

    def fetch_hourly_stats_from_prescient(self, prescient_hourly_stats):
        """Collect the hour's RT LMPs (Prescient hourly stats expose `observed_bus_LMPs`) in the current-day buffer; the
        buffer is rolled into the history (dropping the oldest stored day at the cap) once it holds 24 values, as the
        upstream Backcaster does."""
        for bus in self._historical_rt_prices:
            lmp = prescient_hourly_stats.observed_bus_LMPs[bus]
            day = self._current_day_rt_prices.setdefault(bus, [])
            day.append(float(lmp))
            if len(day) >= 24:
                self._append(self._historical_rt_prices, bus, day[:24])
                self._current_day_rt_prices[bus] = day[24:]

    def fetch_day_ahead_stats_from_prescient(self, uc_date, uc_hour, day_ahead_result):
        """Append the cleared day's 24 DA LMPs (Prescient RUC market exposes `day_ahead_prices[(bus, t)]`)."""
        for bus in self._historical_da_prices:
            day = [day_ahead_result.day_ahead_prices.get((bus, t)) for t in range(24)]
            self._append(self._historical_da_prices, bus, day)


class PerfectForecaster(AbstractPrescientPriceForecaster):
    """Reads {bus}-DALMP / {bus}-RTLMP / {gen}-DACF / {gen}-RTCF columns of a time-indexed DataFrame
    (reference: dispatches/workflow/parametrized_bidder.py:19-70; wraps to the start of the data, :52-58)."""

    def __init__(self, data_path_or_df):
        if isinstance(data_path_or_df, str):
            self.data = pd.read_csv(data_path_or_df, index_col="Datetime", parse_dates=True)
        elif isinstance(data_path_or_df, pd.DataFrame):
            self.data = data_path_or_df
        else:
            raise ValueError

    def __getitem__(self, index):
        return self.data[index]

    def fetch_hourly_stats_from_prescient(self, prescient_hourly_stats):
        pass

    def fetch_day_ahead_stats_from_prescient(self, uc_date, uc_hour, day_ahead_result):
        pass

    def get_column_from_data(self, date, hour, horizon, col):
        start = pd.to_datetime(date) + pd.Timedelta(hours=hour)
        pos = int(np.searchsorted(self.data.index.values, np.datetime64(start), side="left"))
        values = self.data[col].values[pos:pos + horizon]
        if len(values) < horizon:
            values = np.append(values, self.data[col].values[:horizon - len(values)])
        return values

    def forecast_day_ahead_and_real_time_prices(self, date, hour, bus, horizon, _):
        rt = self.forecast_real_time_prices(date, hour, bus, horizon, _)
        da = self.forecast_day_ahead_prices(date, hour, bus, horizon, _)
        return da, rt

    def forecast_day_ahead_prices(self, date, hour, bus, horizon, _):
        return self.get_column_from_data(date, hour, horizon, f"{bus}-DALMP")

    def forecast_real_time_prices(self, date, hour, bus, horizon, _):
        return self.get_column_from_data(date, hour, horizon, f"{bus}-RTLMP")

    def forecast_day_ahead_capacity_factor(self, date, hour, gen, horizon):
        return self.get_column_from_data(date, hour, horizon, f"{gen}-DACF")

    def forecast_real_time_capacity_factor(self, date, hour, gen, horizon):
        return self.get_column_from_data(date, hour, horizon, f"{gen}-RTCF")
