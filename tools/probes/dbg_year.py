import sys, os
sys.path.insert(0, os.getcwd())
import torch
import dispatches_amd.rolling as R
from dispatches_amd.rolling import PipelinedDoubleLoops
def show(tag, loop, d):
    torch.cuda.synchronize()
    its = [l.da.out["iters"].float().mean().item() for l in loop.loops]
    print(tag, d, [int(l.hour_t.item()) for l in loop.loops], its, [float(l.revenue.mean()) for l in loop.loops], [l.da.dlp.last_stats.rtc for l in loop.loops], flush=True)
if len(sys.argv) > 1:
    orig = R.default_options
    def patched(**kw):
        o = orig(**kw); o.no_rtc = 1; return o
    R.default_options = patched
for rep in range(2):
    loop = PipelinedDoubleLoops(1024, device=0, groups=1)
    for d in range(4):
        loop.run_day(); show("B1024 g1", loop, d)
loop = PipelinedDoubleLoops(2048, device=0, groups=2)
for d in range(4):
    loop.run_day(); show("B2048 g2", loop, d)
