import sys, os, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver, DeviceLP
for wl, T in (("wind_battery_24h", 24), ("wind_battery_48h", 48)):
    bidder, model = scenarios.make_batch(wl, 4, HipPdlpSolver(device=0))
    lp = model.lp
    dlp = DeviceLP(lp, 0)
    for B in (4096, 131072):
        X = torch.randn(B, lp.n, dtype=torch.float64, device="cuda"); Y = torch.randn(B, lp.m, dtype=torch.float64, device="cuda")
        AX = torch.empty(B, lp.m, dtype=torch.float64, device="cuda"); ATY = torch.empty(B, lp.n, dtype=torch.float64, device="cuda")
        for _ in range(20): dlp.spmv_step(X, Y, AX, ATY)
        torch.cuda.synchronize()
        reps = 200 if B == 4096 else 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(5):
            e0.record()
            for _ in range(reps): dlp.spmv_step(X, Y, AX, ATY)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        byt = B * 16 * (lp.n + lp.m) + 2 * (lp.nnz * 12 + 4 * (lp.m + 1))
        # the same launches replayed from a hipGraph
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): dlp.spmv_step(X, Y, AX, ATY)
            torch.cuda.synchronize()
            g.capture_begin()
            for _ in range(reps // 4): dlp.spmv_step(X, Y, AX, ATY)
            g.capture_end()
        torch.cuda.synchronize()
        gb = 1e9
        for rep in range(5):
            e0.record()
            for _ in range(4): g.replay()
            e1.record(); torch.cuda.synchronize()
            gb = min(gb, e0.elapsed_time(e1) / reps)
        print(f"{wl} B={B} WPB={os.environ.get('DSP_SPMV_WPB','8')}: stream {best*1e3:.2f} us = {byt/best/1e6:.0f} GB/s ({byt/best/1e6/8000:.3f}); graph {gb*1e3:.2f} us = {byt/gb/1e6:.0f} GB/s ({byt/gb/1e6/8000:.3f})", flush=True)
