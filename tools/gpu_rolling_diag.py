"""GPU dev tool: per-solve status / iteration diagnostics of the batched double loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
days = int(sys.argv[2]) if len(sys.argv) > 2 else 3
loop = BatchedWindBatteryDoubleLoop(B, device=0)
def rep(tag, out, t):
    st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
    print(f"  {tag}: {1e3*t:.1f} ms status {np.bincount(st, minlength=5).tolist()} iters mean {it.mean():.0f} max {it.max()}", flush=True)
for d in range(days):
    torch.cuda.synchronize(); t = time.perf_counter(); loop.day_ahead(); torch.cuda.synchronize()
    print(f"day {d}: soc mean {loop.soc.mean().item():.0f} thr mean {loop.thr.mean().item():.0f}")
    rep("DA", loop.da.out, time.perf_counter() - t)
    trt = ttr = 0.0
    worst = None
    for h in range(24):
        torch.cuda.synchronize(); t = time.perf_counter(); loop.hour_step(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        for tag, m in (("RT", loop.rt), ("TR", loop.tr)):
            st = m.out["status"].cpu().numpy()
            if (st != 0).any() and worst is None:
                worst = (h, tag, np.bincount(st, minlength=5).tolist(), int(m.out["iters"].max().item()))
                k = np.nonzero(st != 0)[0]
                np.savez(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", f"r02k_fail_d{d}.npz"), ids=k, tag=tag,
                         c=m.c[k].cpu().numpy(), lb=m.lb[k].cpu().numpy(), ub=m.ub[k].cpu().numpy(),
                         rlo=m.rlo[k].cpu().numpy(), rhi=m.rhi[k].cpu().numpy(), x=m.out["x"][k].cpu().numpy(),
                         status=st[k], iters=m.out["iters"][k].cpu().numpy())
        trt += dt
    print(f"  24 hours: {1e3*trt:.1f} ms; first non-optimal hourly solve: {worst}", flush=True)
