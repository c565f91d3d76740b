"""GPU development tool: would the day-ahead batch of the rolling double loop end sooner if its scenarios were started in the order of
YESTERDAY's iteration counts (longest first)?  Prints, for consecutive simulated days, the rank correlation of the per-plant iteration
counts and the makespan of a greedy 2048-slot schedule in natural order, in yesterday's order and in the true order (the bound).
    python tools/gpu_tail_order.py [plants] [days]"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
days = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def makespan(it, order, slots=2048):
    h = [0.0] * slots
    heapq.heapify(h)
    for s in order:
        heapq.heappush(h, heapq.heappop(h) + it[s] + 40.0)
    return max(h)


loop = BatchedWindBatteryDoubleLoop(B, device=0)
prev = None
for d in range(days):
    loop.run_day()
    it = loop.da.out["iters"].cpu().numpy().astype(float)
    line = f"day {d}: day-ahead iterations mean {it.mean():.0f} p99 {np.quantile(it, .99):.0f} max {it.max():.0f} | makespan natural {makespan(it, np.arange(B)):.0f} true LPT {makespan(it, np.argsort(-it)):.0f}"
    if prev is not None:
        rk = lambda a: np.argsort(np.argsort(a))
        line += f" yesterday's order {makespan(it, np.argsort(-prev)):.0f} | rank correlation with yesterday {np.corrcoef(rk(it), rk(prev))[0, 1]:.2f}"
        top = np.argsort(-it)[:20]
        line += f" | of today's 20 slowest, in yesterday's slowest 10 %: {int((rk(-prev)[top] < B // 10).sum())}"
    print(line, flush=True)
    prev = it
