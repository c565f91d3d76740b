"""Development: tracker over consecutive hours, cold vs rolling-horizon warm start (iterations per solve)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dispatches_amd.hip_solver import HipPdlpSolver
from dispatches_amd.workflow import Tracker
from tests.test_hip_parity import _wind_battery_objects
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dispatches_amd/data/rts_gmlc_309.npz"))
rts = {k: d[k] for k in d.files}
for warm in (False, True):
    tr = Tracker(tracking_model_object=_wind_battery_objects(rts, False), tracking_horizon=4, n_tracking_hour=1,
                 solver=HipPdlpSolver(device=0), warm_start=warm)
    its, objs = [], []
    for h in range(12):
        cf = rts["rt_cf"][h:h + 4] * 200
        D = [max(0.0, 0.8 * c) for c in cf]
        prof = tr.track_market_dispatch(market_dispatch=D, date="2020-01-02", hour=h)
        tr.update_model(**prof)
        its.append(int(tr.model.iterations[0])); objs.append(float(tr.model.objective[0]))
        assert tr.model.status[0] == 0
    print("warm" if warm else "cold", its, np.round(objs, 3))
