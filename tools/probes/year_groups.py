"""GPU probe: the wind + battery year loop (config 4) as G groups of plants on G streams, with the per-day join of the groups
(PipelinedDoubleLoops.run_day) and FREE-RUNNING (every group enqueues its days on its own stream, one join at the end).
    python tools/probes/year_groups.py [plants] [days] [G list]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dispatches_amd.rolling import PipelinedDoubleLoops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
days = int(sys.argv[2]) if len(sys.argv) > 2 else 60
Gs = [int(g) for g in (sys.argv[3] if len(sys.argv) > 3 else "2 4 8").split()]
dev = torch.device("cuda", 0)
for G in Gs:
    loop = PipelinedDoubleLoops(B, device=0, groups=G)
    for _ in range(3):
        loop.run_day()
    torch.cuda.synchronize()
    for mode in ("join per day", "free-running"):
        loop.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "join per day" or G == 1:
            for d in range(days):
                loop.run_day()
        else:
            cur = torch.cuda.current_stream(dev)
            for s in loop.streams:
                s.wait_stream(cur)
            for d in range(days):
                for l, s in zip(loop.loops, loop.streams):
                    with torch.cuda.stream(s):
                        l.run_day()
            for s in loop.streams:
                cur.wait_stream(s)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        res, ok = loop.results()
        print(f"plants {B} groups {G} {mode:13s}: {1e3 * el / days:7.2f} ms per simulated day ({days} days; host enqueue {1e3 * t_enq / days:.2f} ms per day) "
              f"all optimal {ok} uncertified {int(loop.uncertified)} revenue checksum {float(res['obj'].sum()):.6f}", flush=True)
    del loop
    gc.collect(); torch.cuda.empty_cache()
