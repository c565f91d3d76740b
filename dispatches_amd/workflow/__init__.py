from .bidder import AbstractBidder, Bidder, SelfScheduler, StochasticProgramBidder
from .coordinator import DoubleLoopCoordinator, PrescientPluginModule
from .forecaster import AbstractPrescientPriceForecaster, Backcaster, PerfectForecaster
from .model_data import GeneratorModelData, RenewableGeneratorModelData, ThermalGeneratorModelData
from .parametrized_bidder import FixedParametrizedBidder, ParametrizedBidder, PEMParametrizedBidder
from .tracker import Tracker
from .utils import convert_marginal_costs_to_actual_costs

__all__ = [
    "AbstractBidder", "Bidder", "SelfScheduler", "StochasticProgramBidder", "DoubleLoopCoordinator",
    "PrescientPluginModule", "AbstractPrescientPriceForecaster", "Backcaster", "PerfectForecaster",
    "GeneratorModelData", "RenewableGeneratorModelData", "ThermalGeneratorModelData",
    "FixedParametrizedBidder", "ParametrizedBidder", "PEMParametrizedBidder", "Tracker",
    "convert_marginal_costs_to_actual_costs",
]
