"""Summary of a DSP_LANE_PROBE dump (dsp_stream_lane.hip): where a wave of the lane kernel spends its time.

    python tools/lane_probe_report.py <dump> [<dump> ...]
"""
import sys
import numpy as np


def report(path):
    raw = np.fromfile(path, dtype=np.uint8)
    ntile, G, waves_per_wg, slots = np.frombuffer(raw[:16].tobytes(), dtype=np.int32)
    d = np.frombuffer(raw[16:].tobytes(), dtype=np.uint64).reshape(-1, slots).astype(np.int64)
    d = d[d[:, 2] > 0]                                  # waves that walked
    wall0, c0, nu, hw, cwalk, cbar, cend, wall1 = (d[:, k] for k in range(8))
    # clock ticks per microsecond from the waves' own (clock, wall clock = 100 MHz) pairs
    long_ = (wall1 - wall0) > 500
    tick = np.median((cend - c0)[long_] / ((wall1 - wall0)[long_] / 100.0)) if long_.any() else 100.0
    us = lambda x: x / tick
    t0 = wall0.min()
    start = (wall0 - t0) / 100.0
    end = (wall1 - t0) / 100.0
    print(f"{path}: {len(d)} waves walked (tiles {ntile} x groups {G}), clock = {tick:.1f} ticks / us")
    print(f"  launch: first wave starts at 0, median start {np.median(start):.2f} us, last start {start.max():.2f} us; last end {end.max():.2f} us")
    print(f"  per wave: prologue+walk {np.median(us(cwalk - c0)):.2f} us (p10 {np.percentile(us(cwalk - c0), 10):.2f}, p90 {np.percentile(us(cwalk - c0), 90):.2f}),"
          f" barrier wait {np.median(us(cbar - cwalk)):.2f} us, sums + stores {np.median(us(cend - cbar)):.2f} us; units per wave {np.median(nu):.0f}")
    # per unit
    first, tot, wait, comp, gap = [], [], [], [], []
    for r in d:
        k = int(r[2])
        if 8 + 3 * k > slots:
            k = (slots - 8) // 3
        u = r[8:8 + 3 * k].reshape(k, 3)
        first.append(us(u[0, 0] - r[1]))
        tot.append(us(u[:, 2] - u[:, 0]))
        if (u[:, 1] > 0).all():
            wait.append(us(u[:, 1] - u[:, 0])); comp.append(us(u[:, 2] - u[:, 1]))
    tot = np.concatenate(tot)
    print(f"  before the first unit (tile record, scalars, ring zeroing): {np.median(first):.2f} us")
    print(f"  a unit: median {np.median(tot):.2f} us, mean {tot.mean():.2f}, p10 {np.percentile(tot, 10):.2f}, p90 {np.percentile(tot, 90):.2f}")
    if wait:
        wait = np.concatenate(wait); comp = np.concatenate(comp)
        print(f"    requests -> data in registers: median {np.median(wait):.2f} us (mean {wait.mean():.2f}); arithmetic + LDS: median {np.median(comp):.2f} us (mean {comp.mean():.2f})")
    # how the waves spread over time: rounds
    xcc = (hw >> 32) & 0xf
    cu = (hw >> 8) & 0xf
    se = (hw >> 13) & 0x7
    print(f"  XCDs seen {sorted(set(xcc.tolist()))}; waves per XCD {np.bincount(xcc.astype(int)).tolist()}")
    late = start > 0.25 * end.max()
    print(f"  waves that start after a quarter of the launch (a second round): {late.sum()}")


for p in sys.argv[1:]:
    report(p)
