"""GPU dev tool / config-4 driver: the device-resident batched double loop for B plants over `days` simulated days.
    python tools/gpu_rolling_year.py [B] [days] [warm: 0 cold | 1 shifted point + weight | 2 weight only]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
days = int(sys.argv[2]) if len(sys.argv) > 2 else 3
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 1            # 0 cold, 1 shifted point + weight, 2 weight only
loop = BatchedWindBatteryDoubleLoop(B, device=0, warm_start={0: False, 1: True, 2: "weight"}[warm])
loop.run_day(); torch.cuda.synchronize()                          # warm-up day (handles, code objects)
t0 = time.perf_counter()
for d in range(days):
    t = time.perf_counter(); loop.run_day(); torch.cuda.synchronize()
    print(f"day {d + 1} (warm start {warm}): DA iterations mean {loop.da.out['iters'].float().mean().item():.0f} max {loop.da.out['iters'].max().item()}; {1e3 * (time.perf_counter() - t):.1f} ms for {B} plants (1 day-ahead + 24 x (real-time + tracking) solves each)", flush=True)
el = time.perf_counter() - t0
res, ok = loop.results()
print(f"{B} plants x {days} days: {el:.2f} s = {B * days / el:.0f} plant-days/s = {B * days * 49 / el:.0f} LP solves/s; all optimal: {ok}; "
      f"-> {366 * el / days:.1f} s per simulated year; mean revenue/day {res['obj'].mean().item() / (days + 1):.0f} $, "
      f"mean delivered {res['energy_mwh'].mean().item() / (days + 1):.0f} MWh/day")
