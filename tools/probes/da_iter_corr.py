import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
loop = BatchedWindBatteryDoubleLoop(4096, device=0)
its = []
for d in range(12):
    loop.run_day()
    its.append(loop.da.out["iters"].cpu().numpy().copy())
its = np.array(its, float)
for d in range(3, 12):
    a, b = its[d - 1], its[d]
    top = np.argsort(-b)[:64]
    rank_prev = np.argsort(np.argsort(-a))[top]
    print("day", d, "mean", b.mean().round(0), "max", b.max(), "corr", np.corrcoef(a, b)[0, 1].round(3), "| of today's 64 slowest,", int((rank_prev < 512).sum()), "were among yesterday's 512 slowest; slowest's rank yesterday", int(rank_prev[0]))
v = its[2:].ravel()
print("quantiles 50/90/99/99.9/max", [int(np.quantile(v, q)) for q in (0.5, 0.9, 0.99, 0.999)], int(v.max()))
for p in (4000, 6000, 8000, 10000):
    t = v[v > p]
    print("beyond", p, ":", len(t), "of", len(v), "| their totals: median", int(np.median(t)) if len(t) else None, "p90", int(np.quantile(t, 0.9)) if len(t) else None, "max", int(t.max()) if len(t) else None)
print("daily max", [int(x) for x in its[2:].max(1)])
