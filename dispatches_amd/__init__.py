"""dispatches_amd — MI355X-native batched LP dispatch solver behind the DISPATCHES double-loop plugin API.

Layout (only what the hot path needs):
  lp.py            flatten-once linear modelling layer (LinearBlock -> StandardFormLP)
  flowsheets/      linear unit-model rows + the three multi-period model objects (wind+battery, wind+PEM, nuclear)
  workflow/        Bidder / SelfScheduler / Tracker / DoubleLoopCoordinator / ParametrizedBidder / forecasters
  hip_solver.py    ctypes binding of the C ABI in include/dsp_hip.h  (csrc/*.hip, built by __graft_entry__.build)
  scenarios.py     synthetic RTS-GMLC scenario batches (BASELINE.json configs)
"""
__version__ = "0.1.0"
