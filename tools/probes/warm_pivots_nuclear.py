import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np, time
from dispatches_amd.rolling_flowsheets import BatchedDoubleLoop
for fs in ("nuclear", "wind_pem"):
  for warm in (False, True):
    loop = BatchedDoubleLoop(fs, 256, device=0, use_graphs=False, simplex_warm=warm)
    loop.run_day()
    loop.day_ahead()
    rows = []
    for h in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loop.hour_step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        r, t = loop.rt.out["iters"].float(), loop.tr.out["iters"].float()
        rows.append((h, r.mean().item(), r.max().item(), t.mean().item(), t.max().item(), dt * 1e3))
    a = np.array(rows)
    print(fs, "simplex_warm", warm, "T rt", loop.rt.T, "tr", loop.tr.T)
    for row in rows[:3] + rows[-2:]:
        print("  hour %2d  rt pivots mean %5.1f max %4.0f | tr mean %5.1f max %4.0f | %.3f ms" % row)
    print("  day: rt mean %.1f tr mean %.1f, hour %.3f ms" % (a[1:, 1].mean(), a[1:, 3].mean(), a[1:, 5].mean()))
