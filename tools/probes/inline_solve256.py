"""GPU probe: why do the 256 year-long LPs take 0.82 s inside the default bench line and 0.71 s as a process of their own?  The same
solve before and after other legs of the line, in one process.      python tools/probes/inline_solve256.py"""
import argparse
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
base = dict(gpus=1, steps=1, warmup=1, workload="price_taker", batch=256, eps=None, horizon=8736, cpu_sample=0, streams=0, solve=True, throughput=None,
            pdhg=False, family="wide", warm_start=-1, groups=0, total=0, join_days=False, flowsheet=None)


def solve256(tag):
    a = argparse.Namespace(**base)
    line = bench.bench_price_taker(a, 0, 0, 1, dev)
    c = line["config"]
    print(f"{tag}: {c['seconds_per_batch']:.3f} s, {c['ms_per_newton_iteration_of_the_batch']:.2f} ms per Newton iteration, max {c['max_newton_iterations']}", flush=True)


solve256("first thing in the process")
solve256("again")
a = argparse.Namespace(**dict(base, workload="double_loop", flowsheet="wind_battery", total=8192, steps=int(os.environ.get("PROBE_DAYS", "30")), warmup=2))
bench.bench_double_loop(a, 0, 0, 1, dev)
solve256("after the year loop (objects alive until collected)")
gc.collect(); torch.cuda.empty_cache()
solve256("after gc + empty_cache")
