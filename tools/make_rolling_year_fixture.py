"""Generate tests/golden/rolling_year.npz: the FULL-YEAR rolling double loop of a few wind + battery plants on the INDEPENDENT CPU
oracle (oracle/double_loop_oracle.py: un-reduced Appendix-A LPs, HiGHS dual simplex at 1e-9, the stub market, the 2-dp state
hand-off of wind_battery_double_loop.py:194-200, windows that wrap the data end as parametrized_bidder.py:52-58).

BASELINE config 4 pin: plant k of the 8192-plant batch sees the year that starts at hour (17 k) mod 8736; the fixture holds, per
plant and simulated day, revenue [$], delivered energy [MWh], day-ahead energy [MWh] and the state of charge / throughput [kWh] the
day ended with.  366 days x (1 day-ahead + 24 x (real-time + tracking)) = 17 934 HiGHS solves per plant.

    python tools/make_rolling_year_fixture.py [days] [processes]
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

# both halves of the two-group pipelined loop, shard edges of an 8-way split (1024 k), the last plant, and plants whose year wraps the
# data end early (start hour 17 k mod 8736: k = 513 starts at hour 8721, 15 h before the end)
PLANTS = [0, 1, 513, 1023, 1024, 2047, 3000, 4095, 4096, 5000, 6143, 7000, 7168, 8000, 8190, 8191]


def _work(args):
    k, days = args
    from oracle import double_loop_oracle as dl
    t = time.time()
    out = dl.roll(k, days)
    return k, out, time.time() - t


def main():
    days = int(sys.argv[1]) if len(sys.argv) > 1 else 366
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, (os.cpu_count() or 2) - 2)
    t0 = time.time()
    with mp.Pool(procs) as pool:
        res = {}
        for k, out, el in pool.imap_unordered(_work, [(k, days) for k in PLANTS]):
            res[k] = out
            print(f"plant {k}: {days} days in {el:.0f} s, annual revenue {out['revenue'].sum():.2f} $, delivered {out['delivered'].sum():.1f} MWh", flush=True)
    keys = ("revenue", "delivered", "da_energy", "soc", "thr")
    arrays = {key: np.stack([res[k][key] for k in PLANTS]) for key in keys}
    # hourly state before every hour (2-dp values: exact in float64) and delivered power, for the teacher-forced spot checks
    arrays["h_soc"] = np.stack([res[k]["h_soc"] for k in PLANTS])
    arrays["h_delivered"] = np.stack([res[k]["h_delivered"] for k in PLANTS]).astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "rolling_year.npz"), plants=np.array(PLANTS), days=days, stride=17, **arrays)
    print(f"{len(PLANTS)} plants x {days} days = {len(PLANTS) * days * 49} HiGHS solves in {time.time() - t0:.0f} s -> tests/golden/rolling_year.npz")


if __name__ == "__main__":
    main()
