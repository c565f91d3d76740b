"""GPU development tool: what a plain device-to-device copy of the SpMV step's bytes costs (hipGraph replay of 50 copies):
the size-dependent copy ceiling the streaming SpMV step should be read against."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
for label, nbytes_read in (("T=24 B=4096 (10.3 MB in, 10.3 MB out)", 4096 * 8 * (194 + 120)), ("T=48 B=4096 (20.5 MB in, 20.5 MB out)", 4096 * 8 * (386 + 240)),
                           ("T=24 B=131072 (329 MB in, 329 MB out)", 131072 * 8 * (194 + 120))):
    n = nbytes_read // 8
    x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y.copy_(x); s.synchronize()
        g.capture_begin()
        for _ in range(50): y.copy_(x)
        g.capture_end()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200)
    print(f"copy {label}: {best*1e3:.2f} us = {2*nbytes_read/best/1e6:.0f} GB/s ({2*nbytes_read/best/1e6/8000:.3f} of 8 TB/s)", flush=True)
