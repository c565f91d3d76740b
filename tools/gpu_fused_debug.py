import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from dispatches_amd.rolling import BatchedWindBatteryDoubleLoop
B = 96
A = BatchedWindBatteryDoubleLoop(B, device=0, use_graphs=False, use_fused=False)
F = BatchedWindBatteryDoubleLoop(B, device=0, use_graphs=False, use_fused=True)
def cmp(tag):
    bad = []
    for name, ta, tf in (("rt.c", A.rt.c, F.rt.c), ("rt.lb", A.rt.lb, F.rt.lb), ("rt.ub", A.rt.ub, F.rt.ub), ("tr.lb", A.tr.lb, F.tr.lb), ("tr.ub", A.tr.ub, F.tr.ub),
                         ("tr.rlo", A.tr.rlo, F.tr.rlo), ("tr.rhi", A.tr.rhi, F.tr.rhi), ("soc", A.soc, F.soc), ("thr", A.thr, F.thr), ("delivered", A.delivered, F.delivered),
                         ("revenue", A.revenue, F.revenue), ("energy", A.energy_mwh, F.energy_mwh), ("rt.x", A.rt.out["x"], F.rt.out["x"]), ("tr.x", A.tr.out["x"], F.tr.out["x"]),
                         ("da_offer", A.da_offer, F.da_offer), ("da_prices", A.da_prices, F.da_prices)):
        a, f = ta.cpu().numpy(), tf.cpu().numpy()
        if not np.array_equal(a, f):
            d = np.abs(a - f); i = np.unravel_index(np.nanargmax(np.where(np.isfinite(d), d, 0)), d.shape)
            bad.append((name, float(np.nanmax(np.where(np.isfinite(d), d, 0))), i, a[i], f[i], int((a != f).sum())))
    if bad:
        print(tag, bad[:6]); return True
    return False
A.day_ahead(); F.day_ahead()
cmp("after DA")
for k in range(24):
    A.hour_step(); F.hour_step()
    if cmp(f"after hour {k}"):
        break
