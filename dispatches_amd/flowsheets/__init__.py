from .wind_battery import MultiPeriodWindBattery
from .wind_pem import MultiPeriodWindPEM
from .nuclear import MultiPeriodNuclear

__all__ = ["MultiPeriodWindBattery", "MultiPeriodWindPEM", "MultiPeriodNuclear"]
