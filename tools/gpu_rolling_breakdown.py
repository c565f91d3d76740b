"""GPU development tool: where one simulated day of the batched double loop goes (day-ahead step vs hour steps, eager vs graphs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for graphs in (False, True):
    loop = BatchedWindBatteryDoubleLoop(B, device=0, use_graphs=graphs)
    loop.run_day(); loop.run_day()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    e[0].record(); loop.day_ahead(); e[1].record()
    t1 = time.perf_counter()
    for _ in range(24): loop.hour_step()
    e[2].record(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    st = loop.da.dlp.last_stats
    print(f"B={B} graphs={graphs}: day-ahead {e[0].elapsed_time(e[1]):.2f} ms (host issue {1e3*(t1-t0):.2f}), 24 hour steps {e[1].elapsed_time(e[2]):.2f} ms "
          f"(host issue {1e3*(t2-t1):.2f}), total wall {1e3*(t3-t0):.2f} ms; DA kernel geometry grid {st.grid_blocks} x {st.block_threads}", flush=True)
    it = loop.da.out["iters"].cpu().numpy()
    print(f"   day-ahead iterations mean {it.mean():.0f} p99 {sorted(it)[int(0.99*len(it))]} max {it.max()}")
