import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
for warm in (False, True):
    loop = BatchedWindBatteryDoubleLoop(4096, device=0, use_graphs=False, simplex_warm=warm)
    loop.run_day()
    loop.day_ahead()
    rows = []
    for h in range(24):
        loop.hour_step()
        r, t = loop.rt.out["iters"].float(), loop.tr.out["iters"].float()
        rows.append((h, r.mean().item(), r.max().item(), (r > 20).float().mean().item(), t.mean().item(), t.max().item(), (t > 20).float().mean().item()))
    print("simplex_warm", warm)
    for row in rows[:6] + rows[-3:]:
        print("  hour %2d  rt pivots mean %5.1f max %4.0f share>20 %.3f | tr mean %5.1f max %4.0f share>20 %.3f" % row)
    a = np.array(rows)
    print("  day: rt mean %.1f tr mean %.1f" % (a[:, 1].mean(), a[:, 4].mean()))
