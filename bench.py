#!/usr/bin/env python
"""bench.py — LP scenarios solved / second on the BASELINE.json metric workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one cold-start solve of the rank's whole scenario batch (default 4096 scenarios x 24 h wind+battery
day-ahead bidding LPs drawn from the RTS-GMLC bus-309 series) by the fused HIP PDLP kernel, inputs already
resident in HBM, followed (N > 1) by the RCCL all-gather of the converged objectives.  Weak scaling: every rank
solves its own `--batch` scenarios (scenario ids offset by rank); value = all scenarios of all ranks / max-over-
ranks time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# more hardware queues than the runtime's default 4, so that the launches of the pipelined steps (one HIP stream each)
# really run concurrently; must be set before the HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); 6.29 TB/s measured copy ceiling


SIMDS = 256 * 4           # MI355X: 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9     # max shader clock (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 1/2 of the 157.3 TF FP32 vector peak


def _profiled_counters(kernel, batch=None):
    """Mean per-launch PMC counters of `kernel` from the NEWEST committed rocprofv3 summary that holds that kernel
    (profiles/rNN*_pmc_summary.csv: separate --pmc passes of this same bench command, tools/gpu_profile.sh).  `kernel` is a
    substring of the demangled name - the solve kernel is looked up WITH its template arguments ("pdlp_solve_kernel<4, 2,"), so
    a 48-h run is never priced with the 24-h kernel's counters.  ({}, None) if no committed profile has the kernel."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")), key=os.path.getmtime)
    # (mtime is the checkout time in a fresh clone, so ties are broken by name: later round tags sort later)
    files = sorted(files, key=lambda f: os.path.basename(f))
    # a summary collected at THIS batch size (tools/gpu_profile_configs.sh: r4xp_<workload>_B<batch>_pmc_summary.csv) wins over one
    # of the same kernel at another batch (FETCH_SIZE / WRITE_SIZE are per launch: they do not scale like the SQ counters)
    if batch is not None:
        files = [f for f in files if f"_B{batch}_" not in os.path.basename(f)] + [f for f in files if f"_B{batch}_" in os.path.basename(f)]
    for f in reversed(files):
        out, generic = {}, {}
        for row in csv.DictReader(open(f)):
            if kernel in row["kernel"]:
                # the generic (LDS-matrix) instantiation of the same cols / rows per lane is launched after every register-resident
                # solve as the certificate pass and returns at once on a feasible batch: not the kernel that is priced here
                (generic if ", 0u, 0u," in row["kernel"] else out)[row["counter"]] = float(row["mean_counter_value"])
        out = out or generic
        if out:
            return out, os.path.basename(f)
    return {}, None


def _profiled_traffic(kernel, batch=None):
    """HBM bytes per launch of `kernel` from that summary: FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md (both counters are in KB).  None if no profile is committed."""
    c, _ = _profiled_counters(kernel, batch)
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0


_NUCLEAR_PRICES = {}


def _oracle_worker(args):
    """CPU baseline leg: solve a chunk of scenarios with the HiGHS oracle (build once per scenario + solve)."""
    workload, T, ids = args
    sys.path.insert(0, ROOT)
    from dispatches_amd import scenarios
    from oracle import dispatch_lp_oracle as orc
    t_solve = 0.0
    objs = []
    if workload.startswith("wind"):
        battery = "battery" in workload
        s = scenarios.load_series("rts_gmlc_309.npz" if battery else "rts_gmlc_303.npz")
        N = len(s["rt_lmp"])
        stride = 17 if battery else 37
        for k in ids:
            h0 = (stride * k) % (N - T)
            da, rt = np.clip(s["da_lmp"][h0:h0 + T], 0, 500), np.clip(s["rt_lmp"][h0:h0 + T], 0, 500)
            cf = s["rt_cf"][h0:h0 + T]
            P = orc.wind_battery_da(T, cf, da, rt)[0] if battery else orc.wind_pem_da(T, cf, da, rt, wind_kw=847e3)[0]
            t0 = time.perf_counter()
            objs.append(P.solve()[1])
            t_solve += time.perf_counter() - t0
    else:
        nb, da_all, rt_all = _NUCLEAR_PRICES.get(T, (0, None, None))
        if nb <= max(ids):                         # RT prices are one seeded draw over the whole [B, T] batch
            nb = max(4096, max(ids) + 1)
            da_all, rt_all = scenarios.nuclear_prices(nb, T)
            _NUCLEAR_PRICES[T] = (nb, da_all, rt_all)
        for k in ids:
            P = orc.nuclear_da(T, da_all[k], rt_all[k])[0]
            t0 = time.perf_counter()
            objs.append(P.solve()[1])
            t_solve += time.perf_counter() - t0
    return t_solve, objs


def cpu_baseline(workload, T, sample, procs, min_wall=8.0, max_passes=64):
    """Oracle ('port' of the reference's Pyomo+solver path: HiGHS via scipy) on `procs` host processes.
    The first `sample` scenarios of the batch are solved repeatedly (whole passes) until >= min_wall seconds of
    wall time have been spent, so the rate is not dominated by pool latency on a many-core host."""
    import multiprocessing as mp
    ids = np.arange(sample)
    chunks = [(workload, T, c.tolist()) for c in np.array_split(ids, procs) if len(c)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        pool.map(_oracle_worker, [(workload, T, [0])] * procs)          # warm the workers (imports)
        wall, passes, t_solve = 0.0, 0, 0.0
        while wall < min_wall and passes < max_passes:
            t0 = time.perf_counter()
            out = pool.map(_oracle_worker, chunks)
            wall += time.perf_counter() - t0
            passes += 1
            t_solve += sum(t for t, _ in out)
    objs = np.concatenate([o for _, o in out])
    n = sample * passes
    return dict(value=n / wall, unit="scenarios/s", cores=procs, kind="port",
                sample=f"{passes} pass(es) over the first {sample} scenarios of the same batch = {n} LP solves, "
                       f"scipy.optimize.linprog(method='highs') one LP per call, {procs} processes, wall {wall:.2f} s "
                       f"(LP build time included: the reference rebuilds and re-writes its model every solve)",
                solve_only_value=n / (t_solve / procs)), objs


def bench_double_loop(args, rank, local_rank, world, dev):
    """BASELINE config 4: the rolling double loop (device-resident, dispatches_amd/rolling.py) for --total plants (default
    8192) sharded contiguously over the ranks.  One step = ONE SIMULATED DAY of every plant of the shard: 1 day-ahead
    (48-h LP) + 24 x (real-time 4-h LP + tracking 4-h LP) solves with the state hand-off on the device, followed (N > 1) by
    the all-gather of the day's per-plant revenue.  value = plant-days / s over all ranks (strong scaling)."""
    import torch
    import torch.distributed as dist
    from dispatches_amd.distributed import gather_device_results, make_gather_buffers, shard_bounds
    from dispatches_amd.rolling import PipelinedDoubleLoops
    total = args.total if args.total > 0 else 8192
    lo, hi = shard_bounds(total, world, rank)
    B = hi - lo
    per = max(shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world))
    kw = {} if args.warm_start < 0 else {"warm_start": bool(args.warm_start)}
    # --groups G: the rank's plants as G independent loops on G HIP streams whose simulated days overlap (rolling.PipelinedDoubleLoops:
    # 8192 plants on one MI355X 44.2 -> 35.4 ms per simulated day with two groups; 0 = automatic: two from 1024 plants per rank on)
    flowsheet = getattr(args, "flowsheet", None) or "wind_battery"
    if flowsheet == "wind_battery":
        loop = PipelinedDoubleLoops(B, device=local_rank, first_scenario=lo, groups=int(getattr(args, "groups", 0) or 0), **kw)
        G = loop.groups
    else:
        # the nuclear (BASELINE config 2 is a nuclear DOUBLE LOOP) and wind + PEM flowsheets: the loop written over a descriptor of the
        # flowsheet's rolling state (dispatches_amd/rolling_flowsheets.py), one group
        from dispatches_amd.rolling_flowsheets import BatchedDoubleLoop
        loop = BatchedDoubleLoop(flowsheet, B, device=local_rank, first_scenario=lo)
        loop.day_ahead_iterations = lambda: loop.da.out["iters"]
        loop.warm_start = False
        G = 1
    buffers = make_gather_buffers(world, per, dev, width=2) if world > 1 else None
    status_ok = torch.zeros(B, dtype=torch.float64, device=dev)

    def step():
        loop.run_day()
        if world > 1:
            gather_device_results(dict(obj=loop.revenue, status=status_ok), buffers, per)
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    # the warm-up days created handles, code objects and the hipGraphs of a day; the clock goes back to hour 0 with an empty battery so
    # that the timed days are days 0 .. steps - 1 of the year (steps = 366: the whole year as BASELINE config 4 states it, every window
    # wrapping the 8736-hour data once)
    loop.reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    days = args.steps
    marks = sorted({0, days} | ({round(days * q / 4) for q in (1, 2, 3)} if days >= 8 else set()))
    events = {d: torch.cuda.Event(enable_timing=True) for d in marks}
    da_stats = []                                   # per day, on the device: mean / max day-ahead iterations (no host sync inside the year)
    t0 = time.perf_counter()
    free = world == 1 and G > 1 and not getattr(args, "join_days", False)
    if free:
        # one GPU: nothing is exchanged between the days, so the groups run FREE (PipelinedDoubleLoops.run_days: joined at the quarter
        # marks only); the per-day statistics are taken per group, on its stream
        gstats = [[] for _ in range(G)]

        def per_day(g, l):
            it = l.da.out["iters"].float()
            gstats[g].append(torch.stack([it.sum(), it.max()]))
        for a_, b_ in zip(marks[:-1], marks[1:]):
            events[a_].record()
            loop.run_days(b_ - a_, per_day)
        da_stats = [torch.stack([sum(gs[d][0] for gs in gstats) / B, torch.stack([gs[d][1] for gs in gstats]).max()]) for d in range(days)]
    else:
        for d in range(days):
            if d in events:
                events[d].record()
            step()
            it = loop.day_ahead_iterations().float()
            da_stats.append(torch.stack([it.mean(), it.max()]))
    events[days].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    res, ok = loop.results()
    okt = torch.tensor([1 if ok else 0], device=dev)
    # solves the loop accepted with DSP_FLAG_OBJ_WAIVED (the objective bound of the returned point was waived after a stall): the loop
    # counts them on the device and carries their solutions on - the line is only valid while there are none
    unc = loop.uncertified.to(torch.int64).reshape(1).clone()
    da_iters = loop.day_ahead_iterations().float()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        dist.all_reduce(unc, op=dist.ReduceOp.SUM)
    assert int(unc.item()) == 0, f"{int(unc.item())} uncertified solves entered the realised state of the rolling loop"
    rank_devices, dist_world = _rank_devices(world, dev)
    elapsed = float(t.item())
    line = None
    if rank == 0:
        da_stats = torch.stack(da_stats).cpu().numpy()
        cuts = list(zip(marks[:-1], marks[1:]))
        spans = [{"days": [a, b], "ms_per_day": events[a].elapsed_time(events[b]) / (b - a),
                  "day_ahead_iterations_mean": float(da_stats[a:b, 0].mean()), "day_ahead_iterations_max": int(da_stats[a:b, 1].max())} for a, b in cuts]
        line = ({
            "metric": f"plant-days simulated/sec, RTS-GMLC rolling double loop ({'config 4' if flowsheet == 'wind_battery' else flowsheet}), {total} plants", "value": total * days / elapsed,
            "unit": "plant-days/s", "n_gpus": world, "steps": days, "warmup": max(1, args.warmup), "ms_per_step": 1e3 * elapsed / days,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "world_size": world, "collective_backend": (("gloo on host copies (REHEARSAL: ranks share cuda:0)" if os.environ.get("DSP_BENCH_SHARE_GPU") == "1" else
                                                         f"nccl(RCCL) {'.'.join(map(str, torch.cuda.nccl.version()))}") if world > 1 else None),
            "rehearsal": os.environ.get("DSP_BENCH_SHARE_GPU") == "1", "dist_world_size": dist_world, "rank_devices": rank_devices,
            "distinct_gpus": len({d["uuid"] for d in rank_devices}),
            "config": {"workload": f"double_loop: {total} {flowsheet.replace('_', '+')} plants ({per} per GPU), per simulated day 1 x 48-h day-ahead LP (PDLP "
                                   f"kernel) + 24 x ({12 if flowsheet == 'nuclear' else 4}-h real-time LP + 4-h tracking LP) (in-wave simplex), stub market, state hand-off on device",
                       "flowsheet": flowsheet,
                       "lp_solves_per_s": total * days * 49 / elapsed, "all_optimal": bool(okt.item()), "uncertified_solves": int(unc.item()),
                       "day_ahead_warm_start": bool(loop.warm_start),
                       "groups_per_gpu": G, "groups_joined": "at the quarter marks" if free else "every day",
                       "day_ahead_iterations_last_day": {"mean": float(da_iters.mean().item()), "max": int(da_iters.max().item())},
                       "days": days, "year_measured": days >= 366, "lp_solves": total * days * 49,
                       "seconds_per_simulated_year": elapsed if days == 366 else 366 * elapsed / days,
                       "spans": spans,
                       "day_ahead_iterations_per_day": {"mean_first_week": float(da_stats[:7, 0].mean()), "mean_last_week": float(da_stats[-7:, 0].mean()),
                                                        "max_over_run": int(da_stats[:, 1].max())},
                       "mean_revenue_per_plant_day": float(res["obj"].mean().item()) / days}})
    return line


def _rank_devices(world, dev):
    """[{rank, device, uuid, name}] of every rank, gathered through the process group itself (one all_gather of 64 bytes per rank), and
    the world size the collective backend reports: a SCALE record then shows N distinct GPUs, not N ranks on one."""
    import torch
    import torch.distributed as dist
    p = torch.cuda.get_device_properties(dev)
    uuid = str(getattr(p, "uuid", "")) or "unknown"
    mine = f"{dev.index}|{uuid}|{p.name}"[:63].encode()
    buf = torch.zeros(64, dtype=torch.uint8)
    buf[:len(mine)] = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
    if world > 1:
        share = os.environ.get("DSP_BENCH_SHARE_GPU") == "1"
        src = buf if share else buf.to(dev)
        outb = [torch.zeros_like(src) for _ in range(world)]
        dist.all_gather(outb, src)
        rows = [bytes(b.cpu().numpy().tobytes()).rstrip(b"\0").decode() for b in outb]
        ws = dist.get_world_size()
    else:
        rows, ws = [mine.decode()], 1
    devs = []
    for r, row in enumerate(rows):
        d, u, nm = (row.split("|") + ["", "", ""])[:3]
        devs.append({"rank": r, "device": d, "uuid": u, "name": nm})
    return devs, ws


STREAM_FORMS = {0: "none", 1: "two_launch", 2: "tile", 3: "lane", 4: "block", 5: "interior_point"}
STREAM_KERNELS = {1: "k_primal + k_dual_halpern", 2: "k_fused_pre / k_fused", 3: "k_lane<.., 0> (+ k_lane_long<0>)", 4: "k_block_solve",
                  5: "k_seq<FactorBody / ForwardBody / BackwardBody> (banded LDL' and substitutions, csrc/dsp_ipm.hip)"}


def _ipm_roofline(B, m, parts):
    """Roofline object of an interior-point --solve line.  With the banded factorisations and solves time-parallel (csrc/dsp_ipm_seq.hpp) a
    Newton iteration is ~30 launches that stream scenario-minor arrays; its largest single item is the banded solve (5.8 per Newton
    iteration at 256 scenarios: forward walk, border sums, reduced system, border correction, backward walk).  ALGORITHMIC bytes of one
    solve: 8 Bp m (4 W + 9) - the factor's W streams once per walk, the spikes' W streams for the sums and once more for the correction, the
    vector read and written by each of the four passes, 1 / d twice.  Its duration is ARCHIVED: the sum of the five kernels' average
    durations in the newest committed `rocprofv3 --kernel-trace --stats` summary of this form at this batch (profiles/*_ipm_kernel_stats_T8736_B<B>.csv);
    the bench itself times the whole solve (HIP events), not single kernels."""
    import csv
    import glob
    import re
    Bp = (B + 63) // 64 * 64
    out = {"bound": "hbm", "kernel": "one banded solve of the time-parallel form: k_seq<ForwardBody> + k_ipm_border_dot + k_ipm_red_solve + k_ipm_border_apply + "
                                     "k_seq<BackwardBody> (csrc/dsp_ipm.hip)", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
           "time_partitions": parts,
           "note": "ARCHIVED kernel durations (not measured in this run: the line's time is the whole solve); the sequential form (time_partitions = 1) "
                   "is a latency chain of m dependent rows per walk, 5 - 9 ms each whatever the batch"}
    if parts <= 1:
        out.update(bound="latency", peak=None, unit=None)
        return out
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[56]*_ipm_kernel_stats_T8736_B{B}.csv")), reverse=True):
        dur, W = {}, None
        for row in csv.DictReader(open(f)):
            nm = row["Name"]
            for key in ("ForwardBody", "BackwardBody", "k_ipm_border_dot", "k_ipm_red_solve", "k_ipm_border_apply"):
                if key in nm:
                    dur[key] = float(row["AverageNs"])
                    mm = re.search(r"ForwardBody<(\d+)>", nm)
                    if mm:
                        W = int(mm.group(1))
        if len(dur) == 5 and W:
            byt = 8.0 * Bp * m * (4 * W + 9)
            t = sum(dur.values()) * 1e-9
            out.update(achieved=byt / t / 1e9, frac=byt / t / 1e9 / HBM_PEAK_GBS, algorithmic_bytes_per_solve=int(byt), solve_us=t * 1e6,
                       kernel_us={k: v * 1e-3 for k, v in dur.items()}, archived_from=os.path.basename(f))
            break
    # HBM bytes of the same five kernels per solve: FETCH_SIZE x 2 (gfx950: half the bytes of a coalesced streaming read are counted) + WRITE_SIZE,
    # KB, from the newest committed counter summary of this form at this batch (tools/gpu_ipm_pmc.sh: one --pmc pass per counter)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[56]*_ipm_pmc_summary_B{B}.csv")), reverse=True):
        tr = {}
        for row in csv.DictReader(open(f)):
            for key in ("ForwardBody", "BackwardBody", "k_ipm_border_dot", "k_ipm_red_solve", "k_ipm_border_apply"):
                if key in row["kernel"] and row["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    tr[key] = tr.get(key, 0.0) + (2.0 if row["counter"] == "FETCH_SIZE" else 1.0) * float(row["mean_counter_value"]) * 1024.0
        if len(tr) == 5:
            out.update(traffic=sum(tr.values()), traffic_from=os.path.basename(f))
            if out.get("algorithmic_bytes_per_solve"):
                out["traffic_over_algorithmic"] = out["traffic"] / out["algorithmic_bytes_per_solve"]
            break
    return out


def _price_taker_cpu_worker(args):
    """CPU baseline leg of --workload price_taker --solve: one member of the family by the oracle (HiGHS on the un-reduced LP)."""
    T, k = args
    sys.path.insert(0, ROOT)
    from dispatches_amd import scenarios
    from oracle import dispatch_lp_oracle as orc
    cf, lmp = scenarios.price_taker_inputs(T)
    bf, lm = scenarios.PRICE_TAKER_FAMILY_WIDE[k % len(scenarios.PRICE_TAKER_FAMILY_WIDE)]     # (members 0 .. 15 = PRICE_TAKER_FAMILY)
    t = time.perf_counter()
    P, _ = orc.wind_battery_price_taker(T, cf, lmp * lm, batt_cap_factor=bf)
    x, obj = P.solve(tight=True)
    return k, float(obj), time.perf_counter() - t


def bench_price_taker(args, rank, local_rank, world, dev):
    """The HBM-bound workloads of the path: the reference's long-horizon price-taker design LPs on the HBM-resident streaming PDLP,
    --batch scenarios per GPU sharing ONE constraint matrix:
        price_taker          wind + battery, wind_battery_optimize (T = --horizon hourly periods, default 8736; n = m = 6 T)
        pem_price_taker      wind + battery + PEM, wind_battery_pem_optimize (n = 7 T)
        nuclear_price_taker  the 60-point (hydrogen price x PEM capacity) enumeration of the nuclear case (n = 8 T; the members differ
                             in the bounds of one design column)
    Default: one step = one check period (64 PDHG iterations) of the whole batch; the solve is capped at steps x 64 iterations (the
    iteration RATE; choose --steps so that no scenario finishes before the cap - `finished_before_the_cap` must be 0, else the rate
    is diluted by lanes that sat out part of the run: the nuclear family converges in 1.3 - 3.1 k iterations, --steps 12).  --solve: the batch is solved to optimality instead (value = year-long LPs solved / s; `seconds_per_batch`;
    objectives against the committed oracle fixture where it holds them) and `cpu_baseline` times the oracle (HiGHS, un-reduced LP)
    on a bounded sample of the same members on the host's cores.  roofline.bound = "hbm": algorithmic bytes per scenario-iteration
    of the form that ran (dsp_stats::stream_form / stream_bytes_per_iteration: one-launch forms 4 n + 3 m doubles with shared bounds,
    two-launch form 8 n + 6 m) / HIP-event time of the solve on its stream (check sequences included); `traffic` = FETCH_SIZE x 2 +
    WRITE_SIZE per launch of the iteration kernel OF THAT FORM from the newest committed rocprofv3 PMC summary of this round
    (profiles/r4*_pmc_summary*.csv, named in `traffic_from`; null if there is none: it is an archived measurement, not one of this run)."""
    import torch
    import torch.distributed as dist
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    default_T = {"price_taker": 8736, "pem_price_taker": 8736, "nuclear_price_taker": 8784}[args.workload]
    T = args.horizon if args.horizon != 8736 or args.workload != "nuclear_price_taker" else default_T
    B, ce = (args.batch if args.batch != 4096 else (60 if args.workload == "nuclear_price_taker" else 64)), 64
    # form of the throughput accumulator (flowsheets/price_taker.py): "two_level" is the wind + battery family's default since round 4
    # (scenarios.price_taker_batch), the PEM family keeps the reference's chain (the electrolyzer takes the energy a battery would cycle)
    # interior-point form (round 5, csrc/dsp_ipm.hip): the --solve lines use it unless --pdhg; it wants the reference's own chain form of the
    # accumulator (the two_level change of variables has free columns: a first-order device).  The capped iteration-throughput lines measure
    # the PDHG kernels and switch it off.
    ipm = bool(args.solve) and not getattr(args, "pdhg", False)
    thr = args.throughput or ("two_level" if args.workload == "price_taker" and not ipm else "chain")
    # members of the wind + battery family: the 256 DISTINCT ones of scenarios.PRICE_TAKER_FAMILY_WIDE (round 6; the first 16 are the
    # round-4 family, which a batch of 256 used to repeat 16 times)
    family = getattr(args, "family", None) or "wide"
    build = {"price_taker": lambda s: scenarios.price_taker_batch(T, B, s, throughput=thr, family=family)[1],
             "pem_price_taker": lambda s: scenarios.pem_price_taker_batch(T, B, s, inputs="rts303", throughput=thr)[1],
             "nuclear_price_taker": lambda s: scenarios.nuclear_price_taker_batch(T, B, s)[1]}[args.workload]

    def run(periods):
        solver = HipPdlpSolver(device=local_rank, check_every=ce, max_iter=periods * ce, recertify=0, no_interior_point=0 if (ipm and periods > 1000) else 1)
        if not hasattr(run, "model"):
            run.model = build(solver)
        run.model.solve_handle = None          # the handle carries its options (max_iter): a fresh one per run
        solver.solve(run.model)
        return solver.last_stats, run.model
    run(max(1, args.warmup))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    st, model = run(40000 if args.solve else args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, st.kernel_ms * 1e-3], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rank_devices, dist_world = _rank_devices(world, dev)
    if rank == 0:
        import csv
        import glob
        k_s = float(t[1].item())
        its = int(model.iterations.sum())
        byt = float(st.stream_bytes_per_iteration) * its
        n, m = model.lp.n, model.lp.m
        form = int(getattr(st, "stream_form", 0))
        # archived counters of the iteration kernel of the form that ran, this round's summaries only
        traffic = src = None
        kname = {3: "k_lane<", 2: "k_fused"}.get(form)
        if kname:
            pats = [f"r[45]*_pmc_summary_{args.workload}_B*.csv"] + (["r[45]*_lane_pmc_summary_B*.csv", "r[45]*_stream_pmc_summary*.csv"] if args.workload == "price_taker" else [])
            files = [f for pat in pats for f in glob.glob(os.path.join(ROOT, "profiles", pat))]
            # the newest summary collected at this line's batch, else the newest of the round
            for f in sorted(set(files), key=lambda f: (f"_B{B}." in os.path.basename(f), os.path.basename(f)), reverse=True):
                c = {}
                for row in csv.DictReader(open(f)):
                    if kname in row["kernel"] and ("0>" in row["kernel"] or form == 2):
                        c.setdefault(row["counter"], float(row["mean_counter_value"]))
                if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    traffic, src = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, os.path.basename(f)
                    break
        line = {
            "metric": f"PDHG scenario-iterations/sec, {args.workload} design LP, T={T} (n={n}, m={m}), batch={B}",
            "value": world * its / k_s, "unit": "scenario-iterations/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": 1e3 * k_s / max(1, args.steps), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "world_size": world, "dist_world_size": dist_world, "rank_devices": rank_devices,
            "distinct_gpus": len({d["uuid"] for d in rank_devices}),
            "config": {"workload": f"{args.workload}: {B} scenarios/GPU sharing one constraint matrix, T = {T} h, streaming PDLP "
                                   + ("solved to optimality" if args.solve else f"capped at {args.steps} check periods of {ce} iterations")
                                   + ("" if thr == "chain" else f", throughput accumulator in its {thr} form"),
                       "throughput_form": thr, "stream_form": STREAM_FORMS.get(form, str(form)), "stream_phases": int(getattr(st, "stream_phases", 0)),
                       "solved_to_optimality": int((model.status == 0).sum()),
                       "iterations_per_scenario": float(model.iterations.mean()), "max_iterations": int(model.iterations.max()),
                       "status_counts": np.bincount(model.status, minlength=5).tolist(),
                       "finished_before_the_cap": int((model.status == 0).sum()) if not args.solve else None,
                       "us_per_batch_iteration": 1e6 * k_s / max(1, int(model.iterations.max())), "host_wall_s": float(t[0].item())},
            "roofline": {"bound": "hbm", "kernel": STREAM_KERNELS.get(form, "?") + f" (+ check sequence every {ce} iterations)", "achieved": byt / k_s / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byt / k_s / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_from": src,
                         "traffic_note": "ARCHIVED: HBM bytes per launch of the iteration kernel of this form (FETCH_SIZE x 2 + WRITE_SIZE) from the newest committed "
                                         "counter summary of this round, at the batch it was collected on (its file name); null if none - not measured in this run",
                         "algorithmic_bytes_per_scenario_iteration": int(st.stream_bytes_per_iteration),
                         "two_launch_form_bytes_per_scenario_iteration": 8 * (8 * n + 6 * m),
                         "note": "HBM-resident PDLP: x, x0, c, y, y0 read + x, y written per scenario-iteration (bounds are shared by the batch or kept per "
                                 "scenario for the long columns only); xbar and both products' gathers stay in LDS; time = HIP events around the whole "
                                 "solve on its stream (includes the long-column launches and the check sequences; with --solve also the iterations "
                                 "finished scenarios no longer take part in: frac then understates the kernel)"}}
        if args.solve and form == 5:
            # interior point: `iterations` are Newton iterations; the time is the sequential chain of the banded factorisation and
            # substitutions (m dependent steps per pass, ~100 ns each), not a bandwidth
            line["config"]["workload"] = (f"{args.workload}: {B} scenarios/GPU sharing one constraint matrix, T = {T} h, interior point with banded LDL' "
                                          f"factorisations, one lane per scenario (csrc/dsp_ipm.hip), solved to optimality")
            line["config"]["newton_iterations_per_scenario"] = line["config"].pop("iterations_per_scenario")
            line["config"]["max_newton_iterations"] = line["config"].pop("max_iterations")
            line["config"]["ms_per_newton_iteration_of_the_batch"] = 1e3 * k_s / max(1, int(model.iterations.max()))
            line["config"]["time_partitions"] = line["config"].pop("stream_phases")      # of the banded factorisations / solves (1: sequential walks)
            line["config"].pop("us_per_batch_iteration", None)
            line["roofline"] = _ipm_roofline(B, m, int(line["config"]["time_partitions"]))
        if args.solve:
            line["metric"] = f"year-long design LPs solved/sec, {args.workload}, T={T} (n={n}, m={m}), batch={B}"
            line["value"] = world * int((model.status == 0).sum()) / k_s
            line["unit"] = "LPs/s"
            line["steps"] = 1
            line["ms_per_step"] = 1e3 * k_s
            line["config"]["seconds_per_batch"] = k_s
            fx_path = os.path.join(ROOT, "tests", "golden", "oracle_price_taker.npz")
            if args.workload == "price_taker" and os.path.exists(fx_path):
                fx = np.load(fx_path)
                if f"T{T}/obj" in fx.files and family == "base":
                    ref = fx[f"T{T}/obj"][np.arange(B) % len(scenarios.PRICE_TAKER_FAMILY)]
                    line["config"]["max_rel_objective_error_vs_oracle_fixture"] = float((np.abs(model.objective - ref) / np.maximum(1.0, np.abs(ref))).max())
                elif f"T{T}/obj" in fx.files:
                    # wide family: the fixture holds members 0 .. 15 and every fourth one from 16 on (tools/make_price_taker_fixtures.py --wide)
                    ks, ref = np.arange(16), fx[f"T{T}/obj"]
                    if f"T{T}w/obj" in fx.files:
                        ks, ref = np.concatenate([ks, fx[f"T{T}w/k"]]), np.concatenate([ref, fx[f"T{T}w/obj"]])
                    keep = ks < B
                    ks, ref = ks[keep], ref[keep]
                    line["config"]["max_rel_objective_error_vs_oracle_fixture"] = float((np.abs(model.objective[ks] - ref) / np.maximum(1.0, np.abs(ref))).max())
                    line["config"]["members_with_oracle_fixture"] = int(len(ks))
            if args.workload == "price_taker":
                line["config"]["distinct_members"] = int(len(set(model.family)))
                line["config"]["ipm_solved"] = int(getattr(st, "ipm_solved", 0))
            if args.workload == "price_taker" and args.cpu_sample != 0:
                # The WHOLE host beside the GPU: one HiGHS process per hardware thread (the reference's own sweep runs its members as
                # a process pool: renewables_case/run_pricetaker_wind_PEM.py:106-107), each solving one member of the family - as many
                # members as the GPU batch holds, at most one per thread, and no more processes than the free memory holds at ~0.5 GB
                # each (a member's HiGHS solve peaks at 0.37 GB).  --cpu-sample N caps the processes.
                import multiprocessing as mp
                import psutil
                mem_cap = max(1, int(psutil.virtual_memory().available / (0.6 * 2 ** 30)))
                nsample = max(1, min(B, (os.cpu_count() or 1), mem_cap, args.cpu_sample if args.cpu_sample > 0 else 1 << 30))
                t1 = time.perf_counter()
                with mp.get_context("spawn").Pool(nsample) as pool:
                    res = pool.map(_price_taker_cpu_worker, [(T, k) for k in range(nsample)], chunksize=1)
                wall = time.perf_counter() - t1
                err = max(abs(model.objective[k] - obj) / max(1.0, abs(obj)) for k, obj, _ in res)
                line["cpu_baseline"] = {"value": nsample / wall, "unit": "LPs/s", "cores": nsample, "kind": "port", "host_threads": os.cpu_count(),
                                        "sample": f"{nsample} members of the same batch, one HiGHS process "
                                                  f"each on {os.cpu_count()} hardware threads (oracle/dispatch_lp_oracle.py: the un-reduced LP, feasibility tolerances "
                                                  f"1e-9), wall {wall:.1f} s incl. process start, {np.mean([r[2] for r in res]):.1f} s per member (max {max(r[2] for r in res):.1f} s)",
                                        "max_rel_objective_difference_gpu_vs_these": float(err)}
                line["config"]["gpu_over_full_host"] = line["value"] / line["cpu_baseline"]["value"]
        return line
    return None


def bench_bidder_api(args, rank, local_rank, world, dev):
    """The number a DISPATCHES user sees: wall time of the reference-level call `Bidder.compute_day_ahead_bids` (forecast -> objective
    vectors -> solve -> bid curves -> records) for --batch price scenarios x 24 h of the wind + battery plant on one GPU, host arrays in,
    the reference's bid dictionaries out (upstream idaes Bidder.compute_day_ahead_bids; driver call sites
    dispatches/case_studies/renewables_case/run_double_loop_battery.py:222-285).  One step = one call; value = scenarios / s through
    the plugin API.  The same day is also assembled by the numpy path (solution downloaded in full) and the two bid dictionaries
    are compared bit for bit."""
    import torch
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    B, T = args.batch, 24
    solver = HipPdlpSolver(device=local_rank)
    bidder, model = scenarios.wind_battery_batch(B, T, solver)
    inner = []
    orig = solver.solve

    def timed(*a, **k):
        t0 = time.perf_counter()
        r = orig(*a, **k)
        inner.append((time.perf_counter() - t0, float(solver.last_stats.kernel_ms)))
        return r
    solver.solve = timed
    days = [f"2020-01-{d:02d}" for d in range(2, 30)]
    for d in days[:max(2, args.warmup)]:
        bidder.compute_day_ahead_bids(d, 0)
    inner.clear()
    calls = []
    for k in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bids = bidder.compute_day_ahead_bids(days[k % len(days)], 0)
        calls.append(time.perf_counter() - t0)
    n_opt = int((model.status == 0).sum())
    # the same day through the numpy path (eager download): identical dictionaries
    ref_solver = HipPdlpSolver(device=local_rank, lazy_solution=False)
    ref_bidder, ref_model = scenarios.wind_battery_batch(B, T, ref_solver)
    day = days[(args.steps - 1) % len(days)]
    ref_bids = ref_bidder.compute_day_ahead_bids(day, 0)
    t0 = time.perf_counter()
    ref_bids = ref_bidder.compute_day_ahead_bids(day, 0)
    ref_ms = 1e3 * (time.perf_counter() - t0)
    same = bids == ref_bids
    med = float(np.median(calls))
    k_med = int(np.argsort(calls)[len(calls) // 2])
    points = [len(v[bidder.generator]["p_cost"]) for v in bids.values()]
    return {"metric": f"LP scenarios/sec through Bidder.compute_day_ahead_bids, wind + battery 24 h, batch={B}", "value": B / med, "unit": "scenarios/s",
            "n_gpus": 1, "steps": args.steps, "warmup": max(2, args.warmup), "ms_per_step": 1e3 * med, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"bidder_api: {B} price scenarios x {T} h, Bidder (thermal generator data) + MultiPeriodWindBattery + HipPdlpSolver, host arrays in, bid dictionaries out",
                       "call_ms": {"min": 1e3 * min(calls), "median": 1e3 * med, "max": 1e3 * max(calls)},
                       "solver_solve_ms": 1e3 * inner[k_med][0], "kernel_ms": inner[k_med][1], "host_rest_ms": 1e3 * (calls[k_med] - inner[k_med][0]),
                       "optimal": n_opt, "curve_points_per_hour": {"mean": float(np.mean(points)), "max": int(max(points))},
                       "bids_identical_to_numpy_path": bool(same), "numpy_path_call_ms": ref_ms,
                       "path": "objective vectors formed on the device from the uploaded price windows (PriceObjective), x / y left on the device "
                               "(DeviceSolution), exact roundings + per-hour sorts + distinct points of the bid assembly in ONE kernel launch "
                               "(dsp_bid_points, csrc/dsp_bids.hip), p_min point / running maximum / cost integration for all hours at once on the "
                               "host (workflow/bid_curves.py::curves), p_cost tuple lists (31 k tuples: ~0.8 ms of CPython), records and the "
                               "first 16 scenarios' detail rows (one set of numpy operations) on the host; floor of this call: the lone-batch "
                               "kernel latency (its slowest scenario) + the tuple lists"}}


def _qp_oracle_worker(args):
    """CPU baseline leg of qp_sweep: certified brackets of a chunk of scenarios (oracle/qp_cutting_plane.py)."""
    rho, ids = args
    sys.path.insert(0, ROOT)
    from oracle import qp_cutting_plane as qp
    from tools.make_qp_fixtures import qp_scenario
    out = []
    for k in ids:
        cf, da, rt = qp_scenario(k)
        out.append(qp.wind_battery_da_qp(24, cf, da, rt, rho)[0]["upper"])
    return out


def bench_qp_sweep(args, rank, local_rank, world, dev):
    """BASELINE config 5: the stochastic bidder's day-ahead problems with a quadratic ramp cost (convex QP: soft rows with
    a dual compliance) on --batch scenarios per GPU, float64 vs float32 iterates over the tolerance ladder 1e-3 ... 1e-9.
    value = QP scenarios solved / s at the CONTRACT setting (float64, eps_rel = 1e-9, eps_obj = 1e-7: parity with the
    oracle's certified brackets to 1e-6), steps pipelined over 8 HIP streams like the metric workload.  `sweep` lists, per
    precision and eps_rel (eps_obj tests off: the sweep is about eps_rel alone), one lone batch: kernel time, scenarios
    that terminated, iterations, and the error of the objective of the RETURNED point against the oracle bracket."""
    import torch
    import torch.distributed as dist
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import DeviceLP, HipPdlpSolver, default_options
    wl = args.qp_workload
    B = args.batch
    fx = np.load(os.path.join(ROOT, "tests", "golden", "oracle_qp.npz"))
    solver = HipPdlpSolver(device=local_rank)
    fn, kw = scenarios.QP_WORKLOADS[wl]
    bidder, model = fn(B=B * world, solver=solver, **kw)
    scenarios.load_prices(bidder, model)
    sl = slice(rank * B, (rank + 1) * B)
    lp = model.lp
    lb, ub, rlo, rhi = model.scenario_bounds()
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float64)).to(dev)
    c_d, lb_d, ub_d, rlo_d, rhi_d = up(model.c[sl]), up(lb), up(ub), up(rlo), up(rhi)
    kap_d, c0_d = up(lp.row_compliance), up(np.ascontiguousarray(model.c0[sl]))
    hints = dict(getattr(model, "solver_hints", None) or {})        # what Bidder.solve passes for this model (soft rows: weight_guard)
    dlp = DeviceLP(lp, local_rank, default_options(**hints))
    have = min(B * world, len(fx[f"{wl}/upper"]))
    ref_up, ref_lo = fx[f"{wl}/upper"], fx[f"{wl}/lower"]

    def errors(out):
        ids = np.arange(rank * B, (rank + 1) * B)
        ok = ids < have
        obj = (out["obj"].cpu().numpy() + model.c0[sl])[ok]
        u, l = ref_up[ids[ok]], ref_lo[ids[ok]]
        return np.maximum(np.maximum(l - obj, obj - u), 0.0) / np.maximum(1.0, np.abs(u))

    def lone(opts):
        out = dlp.solve(B, c_d, lb_d, ub_d, rlo_d, rhi_d, options=opts, sync_stats=True, obj_offset=c0_d, row_compliance=kap_d)
        st = out["stats"]
        e = errors(out)
        stat = out["status"].cpu().numpy()
        return dict(kernel_ms=float(st.kernel_ms), terminated=int(st.n_optimal), mean_iterations=float(st.total_iterations) / B,
                    max_iterations=int(st.max_iterations), obj_err_median=float(np.median(e)), obj_err_p99=float(np.quantile(e, 0.99)),
                    obj_err_max=float(e.max()), obj_err_median_of_terminated=(float(np.median(e[stat[:len(e)] == 0])) if (stat[:len(e)] == 0).any() else None),
                    scenarios_per_s_lone_batch=B / (1e-3 * float(st.kernel_ms)))
    sweep = []
    if rank == 0 and not getattr(args, "no_sweep", False):
        for prec in (0, 1):
            for eps in (getattr(args, "sweep_rungs", None) or (1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9)):
                o = default_options(**{**hints, "precision": prec, "eps_rel": eps, "eps_obj": 0.0, "max_iter": args.sweep_max_iter})
                lone(o)
                sweep.append(dict(precision="f32" if prec else "f64", eps_rel=eps, **lone(o)))
    # ---- the contract setting, pipelined ---------------------------------------------------------------------------
    if args.eps is None:
        args.eps = 1e-9
    opts = default_options(**{**hints, "eps_rel": args.eps})
    depth = args.streams if args.streams > 0 else 8
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    outs = [None] * depth
    first = lone(opts)
    first = lone(opts)

    def step(i):
        k = i % depth
        with torch.cuda.stream(streams[k]):
            outs[k] = dlp.solve(B, c_d, lb_d, ub_d, rlo_d, rhi_d, options=opts, out=outs[k], sync_stats=False, obj_offset=c0_d,
                                row_compliance=kap_d)
    for i in range(max(args.warmup, depth)):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        e = errors(outs[0])
        cpu = None
        if args.cpu_sample != 0:
            import multiprocessing as mp
            procs = os.cpu_count() or 1
            sample = min(B, 4096)
            rho = kw["ramp_cost"]
            chunks = [(rho, c.tolist()) for c in np.array_split(np.arange(sample), procs * 4) if len(c)]
            with mp.get_context("spawn").Pool(procs) as pool:
                pool.map(_qp_oracle_worker, [(rho, [0])] * procs)
                tc = time.perf_counter()
                pool.map(_qp_oracle_worker, chunks)
                wall = time.perf_counter() - tc
            cpu = dict(value=sample / wall, unit="scenarios/s", cores=procs, kind="port",
                       sample=f"the first {sample} scenarios of the same batch, Kelley cutting planes on HiGHS LPs to a certified 1e-9 "
                              f"bracket (oracle/qp_cutting_plane.py; no QP solver in the image reaches the bar), {procs} processes, wall {wall:.2f} s")
        return ({
            "metric": f"QP scenarios solved/sec, RTS-GMLC 24h day-ahead bidding + quadratic ramp cost (config 5), batch={B}",
            "value": world * B * args.steps / elapsed, "unit": "scenarios/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, depth),
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "world_size": world,
            "config": {"workload": f"{wl}: {B} scenarios/GPU x 24 h day-ahead bidding QP (n={lp.n}, m={lp.m}, nnz={lp.nnz}, "
                                   f"{int(np.count_nonzero(lp.row_compliance))} soft rows, rho={kw['ramp_cost']})",
                       "eps_rel": args.eps, "streams": depth, "lone_batch": first, "optimal": first["terminated"],
                       "max_rel_obj_err_vs_oracle_bracket": float(e.max()), "scenarios_beyond_1e-6": int((e >= 1e-6).sum()),
                       "flagged": int(((outs[0]["flags"] & 1) != 0).sum().item()) if "flags" in outs[0] else None},
            "sweep": sweep, "cpu_baseline": cpu})
    return None


def _source_hash():
    """dsp_source_hash() of the library this process loaded (include/dsp_hip.h): names the code that produced the numbers."""
    from dispatches_amd import hip_solver
    return hip_solver.load_library().dsp_source_hash().decode()


def _valu_lds_fractions(solve_kernel, B, sum_iters, step_s):
    """VALU / LDS busy fractions of a fused solve kernel from the newest committed SQ counters of that kernel at this batch
    (profiles/r*_<workload>_B<batch>_pmc_summary.csv; instruction counts per launch are deterministic for a build + batch and are
    scaled by the iteration ratio when the recorded launch ran a different total) and the sustained step time measured HERE.
    Returns (valu_frac, lds_frac, valu_rate, counters_file, scaled_by, traffic)."""
    pmc, pmc_file = _profiled_counters(solve_kernel, B)
    traffic = _profiled_traffic(solve_kernel, B)
    scaled = 1.0
    if pmc_file:
        rec_path = os.path.join(ROOT, "profiles", pmc_file.replace("_pmc_summary.csv", "_pmc_iterations.json"))
        if os.path.exists(rec_path):
            rec = json.load(open(rec_path))
            per_launch = next((v for k, v in rec.items() if k in solve_kernel or solve_kernel in k), None)
            if per_launch:
                scaled = float(sum_iters) / float(per_launch)
                pmc = {k: (v * scaled if k.startswith("SQ_") else v) for k, v in pmc.items()}
    valu_frac = lds_frac = valu_rate = None
    if "SQ_INSTS_VALU" in pmc:
        valu_rate = pmc["SQ_INSTS_VALU"] / step_s
        valu_frac = pmc["SQ_INSTS_VALU"] * 4.0 / (SIMDS * step_s * PEAK_CLOCK_HZ)
    if "SQ_LDS_IDX_ACTIVE" in pmc:
        lds_frac = pmc["SQ_LDS_IDX_ACTIVE"] / (256 * step_s * PEAK_CLOCK_HZ)
    return valu_frac, lds_frac, valu_rate, pmc_file, scaled, traffic


def _fused_config_entry(tag, workload, B, local_rank, dev, steps, depth=8):
    """One BASELINE config of the fused one-wave kernels as an entry of the default line's `configs` array: the same measurement as
    the headline (lone batch + `steps` steps pipelined over `depth` HIP streams between synchronisations, inputs resident in HBM),
    parity of the timed batch against the committed oracle fixture, flags, and the VALU / LDS fractions from the committed
    counters of this kernel at this batch."""
    import torch
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import DeviceLP, HipPdlpSolver, default_options
    solver = HipPdlpSolver(device=local_rank)
    fn, kw = scenarios.WORKLOADS[workload]
    bidder, model = fn(B=B, solver=solver, **kw)
    scenarios.load_prices(bidder, model)
    lp = model.lp
    lb, ub, rlo, rhi = model.scenario_bounds()
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float64)).to(dev)
    c_d, lb_d, ub_d, rlo_d, rhi_d, c0_d = up(model.c), up(lb), up(ub), up(rlo), up(rhi), up(np.ascontiguousarray(model.c0))
    opts = default_options(**(getattr(model, "solver_hints", None) or {}))
    dlp = DeviceLP(lp, local_rank, opts)
    solve = lambda out, sync: dlp.solve(B, c_d, lb_d, ub_d, rlo_d, rhi_d, options=opts, out=out, sync_stats=sync, obj_offset=c0_d)
    out = solve(None, True)
    out = solve(out, True)
    st = dlp.last_stats
    lone_ms, sum_iters, max_iters = float(st.kernel_ms), int(st.total_iterations), int(st.max_iterations)
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    outs = [None] * depth

    def burst(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % depth]):
                outs[i % depth] = solve(outs[i % depth], False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    burst(depth)                                           # every stream launches once (hardware queues, output buffers)
    times = [burst(steps) for _ in range(3)]
    elapsed = float(np.median(times))
    step_s = elapsed / steps
    status = out["status"].cpu().numpy()
    flags = out["flags"].cpu().numpy()
    entry = {"config": tag, "workload": workload, "batch": B, "horizon_h": len(model.HOUR), "n": lp.n, "m": lp.m,
             "value": B * steps / elapsed, "unit": "scenarios/s", "steps": steps, "streams": depth, "ms_per_step": 1e3 * step_s,
             "lone_batch_ms": lone_ms, "lone_batch_scenarios_per_s": B / (1e-3 * lone_ms),
             "mean_iterations": sum_iters / B, "max_iterations": max_iters,
             "optimal": int((status == 0).sum()), "flagged": int(((flags & 1) != 0).sum())}
    fx_path = os.path.join(ROOT, "tests", "golden", "oracle_objectives.npz")
    if os.path.exists(fx_path):
        fx = np.load(fx_path)
        if workload in fx.files and len(fx[workload]) >= B:
            ref = fx[workload][:B]
            mine = out["obj"].cpu().numpy() + model.c0
            entry["max_rel_obj_err_vs_oracle_fixture"] = float(np.max(np.abs(mine - ref) / np.maximum(1.0, np.abs(ref))))
    kernel = f"pdlp_solve_kernel<{int(st.cols_per_lane)}, {int(st.rows_per_lane)}," if int(st.matreg) else "pdlp_solve_kernel"
    vf, lf, vr, cfile, scaled, traffic = _valu_lds_fractions(kernel, B, sum_iters, step_s)
    w = 8
    true_io = (sum(t.numel() for t in (c_d, lb_d, ub_d, rlo_d, rhi_d) if t.dim() == 2) * w + B * w * (lp.n + lp.m + 1) + B * 12)
    entry["roofline"] = {"bound": "valu+lds", "kernel": kernel, "frac": vf, "frac_lds": lf, "achieved": (vr / 1e9) if vr else None,
                         "peak": SIMDS * PEAK_CLOCK_HZ / 4.0 / 1e9, "unit": "G FP64-wave-instr/s", "counters_from": cfile,
                         "counters_scaled_by": scaled, "traffic": traffic, "true_io_bytes_per_launch": true_io,
                         "traffic_over_true_io": (traffic / true_io) if traffic else None}
    dlp.close()
    return entry


def _condense(tag, line, keys=()):
    """A full bench line of another workload as an entry of `configs`: headline fields, its roofline, the parity / certification fields."""
    cfg = line.get("config", {})
    e = {"config": tag, "metric": line["metric"], "value": line["value"], "unit": line["unit"], "steps": line["steps"],
         "ms_per_step": line["ms_per_step"], "workload": cfg.get("workload")}
    for k in keys:
        if k in cfg:
            e[k] = cfg[k]
    if "roofline" in line:
        r = line["roofline"]
        e["roofline"] = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_from",
                                                "algorithmic_bytes_per_scenario_iteration", "algorithmic_bytes_per_solve", "archived_from", "traffic_over_algorithmic") if k in r}
    return e


def baseline_configs(args, local_rank, dev, budget_s=360.0, depth=8):
    """Every BASELINE.json config beside the metric one, measured in THIS run (1 GPU), as the `configs` array of the default line:
        1  the reference's own CPU-runnable case as plumbing: boundary classes on LP #1 with the 24 _get_lmp prices, one scenario
        2  nuclear 24 h, 256 scenarios                        3  wind + PEM 48 h, 4096 scenarios
        4  wind + battery 48 h, 4096 (the day-ahead shape) AND the rolling double loop itself at its stated size, 8192 plants
        5  24-h bidding QP (quadratic ramp cost), 4096 scenarios, contract setting (the fp64 / fp32 ladder: --workload qp_sweep)
        +  the plugin API itself (Bidder.compute_day_ahead_bids, 4096 x 24 h, host arrays in, bid dictionaries out)
        +  the HBM-bound streaming path: year-long price-taker design LPs, 256 scenarios, iteration rate over 50 check periods
    Entries are skipped (and say so) once `budget_s` seconds have gone, so that the whole command stays within a few minutes."""
    import copy
    import gc
    import torch
    t_start = time.perf_counter()
    out = []

    def leg(tag, fn):
        if time.perf_counter() - t_start > budget_s:
            out.append({"config": tag, "skipped": f"time budget of {budget_s:.0f} s for the configs array spent"})
            return
        gc.collect()                                       # (the previous leg's handles, streams and hipGraphs go before this one is timed)
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        try:
            e = fn()
        except Exception as exc:                           # noqa: BLE001 - one broken leg must not lose the headline line
            e = {"config": tag, "error": f"{type(exc).__name__}: {exc}"[:300]}
        e["wall_s"] = time.perf_counter() - t0
        out.append(e)

    def config1():
        from dispatches_amd import scenarios
        from dispatches_amd.flowsheets import MultiPeriodWindBattery
        from dispatches_amd.hip_solver import HipPdlpSolver
        from dispatches_amd.workflow import Backcaster, RenewableGeneratorModelData, SelfScheduler
        g13 = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))["G13_usc_pricetaker_lmp_24h"]
        lmp = g13["lmp"]
        s = scenarios.load_series("rts_gmlc_309.npz")
        md = RenewableGeneratorModelData(gen_name="309_WIND_1", bus="Carter", p_min=0, p_max=200, p_cost=0, fixed_commitment=None)
        mp = MultiPeriodWindBattery(model_data=md, wind_capacity_factors=s["rt_cf"], wind_pmax_mw=200, battery_pmax_mw=25,
                                    battery_energy_capacity_mwh=100)
        solver = HipPdlpSolver(device=local_rank)
        bidder = SelfScheduler(bidding_model_object=mp, day_ahead_horizon=24, real_time_horizon=4, n_scenario=1, solver=solver,
                               forecaster=Backcaster({"Carter": list(lmp)}, {"Carter": list(lmp)}))
        bidder.compute_day_ahead_bids(date="2020-01-02")            # first call: handle creation, code-object load
        t0 = time.perf_counter()
        bidder.compute_day_ahead_bids(date="2020-01-02")
        call_ms = 1e3 * (time.perf_counter() - t0)
        m = bidder.day_ahead_model
        ref = g13["oracle_objective_lp1_24h"]["value"]            # committed fixture of the oracle's objective for this LP (tests/golden)
        return {"config": "1", "workload": "plumbing: SelfScheduler + MultiPeriodWindBattery, 24 h, 1 price scenario = the 24 LMPs of the fossil case "
                                           "study's _get_lmp (tests/golden/reference_vectors.json G13), through HipPdlpSolver",
                "value": 1e3 / call_ms, "unit": "compute_day_ahead_bids calls/s", "call_ms": call_ms, "kernel_ms": float(solver.last_stats.kernel_ms),
                "iterations": int(m.iterations[0]), "optimal": int((m.status == 0).sum()), "flagged": int(m.uncertified.sum()),
                "rel_obj_err_vs_oracle_fixture": float(abs(m.objective[0] - ref) / max(1.0, abs(ref)))}
    leg("1", config1)

    def sub(**kw):
        a = copy.copy(args)
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    # (steps = a multiple of the stream depth the headline chose in its warm-up: every stream gets the same number of launches)
    leg("2", lambda: _fused_config_entry("2", "nuclear_24h", 256, local_rank, dev, steps=4 * depth, depth=depth))
    dl_keys = ("flowsheet", "days", "year_measured", "lp_solves", "lp_solves_per_s", "all_optimal", "uncertified_solves", "groups_per_gpu", "spans",
               "day_ahead_iterations_per_day", "seconds_per_simulated_year", "mean_revenue_per_plant_day")
    # config 2 is a DOUBLE LOOP: 256 nuclear plants through four simulated weeks (holdup hand-off on the device, 12-h real-time horizon)
    leg("2-double-loop", lambda: _condense("2-double-loop", bench_double_loop(sub(workload="double_loop", flowsheet="nuclear", total=256, steps=28, warmup=2, groups=0),
                                                                             0, local_rank, 1, dev), dl_keys))
    leg("3", lambda: _fused_config_entry("3", "wind_pem_48h", 4096, local_rank, dev, steps=2 * depth, depth=depth))
    leg("4-day-ahead", lambda: _fused_config_entry("4-day-ahead", "wind_battery_48h", 4096, local_rank, dev, steps=depth, depth=depth))

    leg("4", lambda: _condense("4", bench_double_loop(sub(workload="double_loop", flowsheet="wind_battery", total=8192, steps=366, warmup=2, groups=0), 0, local_rank, 1, dev),
                               ("days", "year_measured", "lp_solves", "lp_solves_per_s", "all_optimal", "uncertified_solves", "groups_per_gpu", "spans",
                                "day_ahead_iterations_per_day", "seconds_per_simulated_year", "mean_revenue_per_plant_day")))
    def config5():
        # the contract setting pipelined + the fp64 / fp32 tolerance ladder in three rungs (config 5's stated content; all seven: --workload qp_sweep)
        line = bench_qp_sweep(sub(workload="qp_sweep", batch=4096, steps=12, warmup=3, no_sweep=False, sweep_rungs=(1e-4, 1e-6, 1e-9), cpu_sample=0, eps=None,
                                  streams=0), 0, local_rank, 1, dev)
        e = _condense("5", line, ("eps_rel", "optimal", "flagged", "max_rel_obj_err_vs_oracle_bracket", "scenarios_beyond_1e-6", "lone_batch"))
        e["ladder"] = [{k: r.get(k) for k in ("precision", "eps_rel", "kernel_ms", "terminated", "mean_iterations", "obj_err_median", "obj_err_max")} for r in line.get("sweep", [])]
        return e
    leg("5", config5)
    leg("bidder_api", lambda: _condense("bidder_api", bench_bidder_api(sub(workload="bidder_api", batch=4096, steps=12, warmup=2), 0, local_rank, 1, dev),
                                        ("call_ms", "solver_solve_ms", "kernel_ms", "host_rest_ms", "optimal", "curve_points_per_hour",
                                         "bids_identical_to_numpy_path", "numpy_path_call_ms")))
    pt_keys = ("stream_form", "time_partitions", "solved_to_optimality", "distinct_members", "ipm_solved", "newton_iterations_per_scenario", "max_newton_iterations",
               "seconds_per_batch", "ms_per_newton_iteration_of_the_batch", "max_rel_objective_error_vs_oracle_fixture", "members_with_oracle_fixture", "throughput_form",
               "gpu_over_full_host")

    def price_taker_solve(tag, batch, cpu_sample):
        # (cpu_sample processes of HiGHS on as many members beside the GPU line: a BOUNDED sample - a member takes HiGHS 9 - 13 s alone and
        #  longer in company; the whole-host figures - 16 / 64 / 256 processes: 0.37 / 0.92 / 0.88 LPs/s on 256 threads - are
        #  profiles/r50d_solve256.json)
        line = bench_price_taker(sub(workload="price_taker", batch=batch, steps=1, warmup=1, solve=True, horizon=8736, throughput=None, cpu_sample=cpu_sample,
                                     pdhg=False, family="wide"), 0, local_rank, 1, dev)
        e = _condense(tag, line, pt_keys)
        if "cpu_baseline" in line:
            e["cpu_baseline"] = line["cpu_baseline"]
        return e
    # the year-long design LPs as 256 DISTINCT members, and at the sizes of the reference's own sweeps (30 points: run_pricetaker_wind_PEM.py:99-107;
    # 60 points: price_taker_analysis.py:358-359)
    leg("price_taker_solve", lambda: price_taker_solve("price_taker_solve", 256, 24))
    leg("price_taker_solve_60", lambda: price_taker_solve("price_taker_solve_60", 60, 0))
    leg("price_taker_solve_30", lambda: price_taker_solve("price_taker_solve_30", 30, 0))
    leg("streaming", lambda: _condense("streaming", bench_price_taker(sub(workload="price_taker", batch=256, steps=50, warmup=1, solve=False, horizon=8736,
                                                                           throughput=None, cpu_sample=0), 0, local_rank, 1, dev),
                                       ("stream_form", "stream_phases", "iterations_per_scenario", "finished_before_the_cap", "us_per_batch_iteration")))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="wind_battery_24h")
    ap.add_argument("--batch", type=int, default=4096, help="scenarios per GPU")
    ap.add_argument("--eps", type=float, default=None,
                    help="eps_rel; default: the model family's solver hint if it has one, else 1e-9 (the contract setting)")
    ap.add_argument("--horizon", type=int, default=8736, help="--workload price_taker: hourly periods of the design LP")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="scenarios for the CPU baseline (0 = skip)")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the steps are pipelined over (0 = choose among 8/12/16/24 during the warm-up)")
    ap.add_argument("--total", type=int, default=0,
                    help="STRONG scaling: this many scenarios in total, sharded over the ranks through "
                         "dispatches_amd.distributed.solve_sharded_device (e.g. --total 8192 --workload wind_battery_48h = "
                         "BASELINE config 4: 1024 per GPU on 8 GPUs); 0 = weak scaling with --batch scenarios per GPU")
    ap.add_argument("--qp-workload", default="wind_battery_24h_qp01", help="--workload qp_sweep: which of scenarios.QP_WORKLOADS")
    ap.add_argument("--sweep-max-iter", type=int, default=40000, help="--workload qp_sweep: iteration cap of the sweep entries")
    ap.add_argument("--min-time", type=float, default=0.5,
                    help="the burst of --steps steps is repeated until the timed bursts cover this many seconds; the median burst is reported")
    ap.add_argument("--max-bursts", type=int, default=500)
    ap.add_argument("--no-eps4", action="store_true", help="skip the extra eps_rel = 1e-4 (PDLP default tolerance) leg of the LP metric line")
    ap.add_argument("--flowsheet", default=None, choices=["wind_battery", "nuclear", "wind_pem"],
                    help="--workload double_loop: the flowsheet of the plants (default wind_battery = BASELINE config 4; nuclear = config 2's double loop)")
    ap.add_argument("--family", default=None, choices=["base", "wide"],
                    help="--workload price_taker: members of the batch - wide (default) = the 256 distinct ones of scenarios.PRICE_TAKER_FAMILY_WIDE, base = the 16-member family cycled")
    ap.add_argument("--pdhg", action="store_true", help="--solve lines: the PDHG forms instead of the interior-point form (dsp_options::no_interior_point)")
    ap.add_argument("--solve", action="store_true", help="--workload price_taker / pem_price_taker / nuclear_price_taker: solve the batch to optimality (full-solve line)")
    ap.add_argument("--throughput", default=None, choices=["chain", "two_level", "hier"],
                    help="--workload price_taker / pem_price_taker: form of the battery's throughput accumulator (flowsheets/price_taker.py); "
                         "with --steps large enough the batch runs to optimality and config.solved_to_optimality / iterations_per_scenario tell")
    ap.add_argument("--warm-start", type=int, default=-1,
                    help="--workload double_loop: 1 / 0 = rolling warm start of the day-ahead LP on / off (-1 = the loop's default)")
    ap.add_argument("--groups", type=int, default=0,
                    help="--workload double_loop: the rank's plants as this many independent loops on as many HIP streams (their days overlap); "
                         "0 = automatic (2 from 1024 plants per rank on)")
    ap.add_argument("--join-days", action="store_true", help="--workload double_loop on one GPU: join the groups after every simulated day instead of letting them run free")
    ap.add_argument("--no-configs", action="store_true", help="default line only: skip the `configs` array (the other BASELINE configs measured in the same run)")
    ap.add_argument("--no-sweep", action="store_true", help="--workload qp_sweep: the contract entry only, without the fp64 / fp32 tolerance ladder")
    ap.add_argument("--no-spmv", action="store_true", help="skip the streaming SpMV-step roofline measurement")
    ap.add_argument("--spmv-large-mult", type=int, default=32,
                    help="also time spmv_step on a batch this many times larger (0 = skip; the PMC passes skip it so "
                         "that the per-dispatch counter means belong to one batch size)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    # DSP_BENCH_SHARE_GPU=1: REHEARSAL of the N-rank launch on a box with fewer GPUs than ranks - every rank uses cuda:0 and the
    # collectives run over gloo on host copies (RCCL cannot put two ranks on one device).  Same sharding, packing, barriers,
    # MAX-over-ranks timing and JSON line as the real N-GPU run; the numbers it prints are NOT scaling numbers (the ranks share one
    # GPU) and carry "rehearsal": true.  tests/test_hip_multirank.py runs it so that the first real 8-GPU launch cannot fail on plumbing.
    share_gpu = os.environ.get("DSP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import DeviceLP, HipPdlpSolver, default_options

    from dispatches_amd.distributed import shard_bounds
    other = {"double_loop": bench_double_loop, "price_taker": bench_price_taker, "pem_price_taker": bench_price_taker,
             "nuclear_price_taker": bench_price_taker, "qp_sweep": bench_qp_sweep, "bidder_api": bench_bidder_api}.get(args.workload)
    if other is not None:
        line = other(args, rank, local_rank, world, dev)
        if rank == 0:
            line["source_hash"] = _source_hash()
            print(json.dumps(line))
        if world > 1:
            dist.destroy_process_group()
        return
    solver = HipPdlpSolver(device=local_rank, **({} if args.eps is None else {"eps_rel": args.eps}))
    fn, kw = scenarios.WORKLOADS[args.workload]
    if args.total > 0:
        # strong scaling: the SAME total batch at every N, rank r owns the contiguous shard shard_bounds(total, N, r)
        n_total = args.total
        lo_s, hi_s = shard_bounds(n_total, world, rank)
    else:
        # weak scaling: rank r owns scenarios [r*B, (r+1)*B) of a batch that grows with N
        n_total = args.batch * world
        lo_s, hi_s = rank * args.batch, (rank + 1) * args.batch
    B = hi_s - lo_s
    B_max = max(shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)) \
        if args.total > 0 else B
    bidder, model = fn(B=n_total, solver=solver, **kw)
    scenarios.load_prices(bidder, model)
    sl = slice(lo_s, hi_s)
    lp = model.lp
    lb, ub, rlo, rhi = model.scenario_bounds()
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float64)).to(dev)
    pick = lambda a: a[sl] if a.ndim == 2 else a
    c_d, lb_d, ub_d = up(model.c[sl]), up(pick(lb)), up(pick(ub))
    rlo_d, rhi_d = up(pick(rlo)), up(pick(rhi))
    c0 = model.c0[sl]
    # the model family's preconditioner hints (e.g. geo_iters of the nuclear flowsheet), as HipPdlpSolver applies them
    opts = default_options(**{**(getattr(model, "solver_hints", None) or {}), **({} if args.eps is None else {"eps_rel": args.eps})})
    args.eps = float(opts.eps_rel)
    dlp = DeviceLP(lp, local_rank, opts)
    def new_out():
        return dict(x=torch.empty((B, lp.n), dtype=torch.float64, device=dev),
                    y=torch.empty((B, lp.m), dtype=torch.float64, device=dev),
                    obj=torch.empty(B, dtype=torch.float64, device=dev),
                    status=torch.empty(B, dtype=torch.int32, device=dev),
                    iters=torch.empty(B, dtype=torch.int32, device=dev),
                    jumps=torch.empty(B, dtype=torch.int32, device=dev),
                    flags=torch.zeros(B, dtype=torch.int32, device=dev))

    c0_d = up(np.ascontiguousarray(c0))
    # Pipeline: consecutive steps (independent batches) are issued round-robin on `--streams` HIP streams, each with
    # its own output buffers, so that one batch's straggler scenarios do not idle the GPU: iteration counts differ
    # ~10x between price scenarios and a single launch ends with a long, nearly empty tail.
    # --streams 0 (default): the depth is picked during the untimed warm-up from a short trial of each candidate (how
    # many kernels really run concurrently differs between boxes); rank 0's choice is broadcast so all ranks agree.
    candidates = [args.streams] if args.streams > 0 else [8, 12, 16, 24]
    max_depth = max(candidates)
    streams = [torch.cuda.Stream(device=dev) for _ in range(max_depth)]
    outs = [new_out() for _ in range(max_depth)]
    # the ONE collective of a step: all-gather of the converged objectives and statuses straight from the solver's
    # device outputs (dispatches_amd.distributed.gather_device_results: RCCL over xGMI, no host hop); ragged shards
    # are padded to the largest shard
    from dispatches_amd.distributed import gather_device_results, make_gather_buffers
    gathered = [make_gather_buffers(world, B_max, dev, width=2) if world > 1 else None for _ in range(max_depth)]
    out = outs[0]
    depth = candidates[0]

    # single-batch latency (one synchronous solve; also the untimed warm-up of the library / geometry cache)
    dlp.solve(B, c_d, lb_d, ub_d, rlo_d, rhi_d, options=opts, out=out, sync_stats=True, obj_offset=c0_d)
    dlp.solve(B, c_d, lb_d, ub_d, rlo_d, rhi_d, options=opts, out=out, sync_stats=True, obj_offset=c0_d)
    st = dlp.last_stats
    single_batch_ms = float(st.kernel_ms)
    sum_iters_one = int(st.total_iterations)
    max_iters_one = int(st.max_iterations)
    geometry = [int(st.grid_blocks), int(st.block_threads), int(st.lds_bytes), int(st.matreg)]
    lds_conflicts = [int(st.lds_conflicts_identity), int(st.lds_conflicts_chosen)]

    events = []

    def step(i, record):
        k = i % depth
        with torch.cuda.stream(streams[k]):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            dlp.solve(B, c_d, lb_d, ub_d, rlo_d, rhi_d, options=opts, out=outs[k], sync_stats=False, obj_offset=c0_d)
            e1.record()
            if world > 1:
                gather_device_results(outs[k], gathered[k], B_max)
        if record:
            events.append((e0, e1))

    torch.cuda.synchronize()
    trials = {}
    if len(candidates) > 1:
        depth = max_depth                             # one launch per stream first: a stream's first launch creates its
        for i in range(max_depth):                    # hardware queue (milliseconds) and would bias the trials below
            step(i, False)
        torch.cuda.synchronize()
        for cand in candidates:                       # untimed (warm-up phase): 2 * cand steps at each depth
            depth = cand
            torch.cuda.synchronize()
            t_try = time.perf_counter()
            for i in range(2 * cand):
                step(i, False)
            torch.cuda.synchronize()
            trials[cand] = (time.perf_counter() - t_try) / (2 * cand)
        best = torch.tensor([min(trials, key=trials.get)], device=dev)
        if world > 1:
            dist.broadcast(best, src=0)
        depth = int(best.item())
    for i in range(max(args.warmup, depth)):      # every stream launches at least once before the timed region
        step(i, False)                            # (a stream's first launch creates its hardware queue: milliseconds)
    # The timed region: bursts of EXACTLY --steps steps, each bracketed by barrier + synchronize and reduced with MAX over the
    # ranks.  One burst of 20 steps is ~20 ms - a single sample of a pipeline that is still filling and draining - so the burst
    # is repeated until the bursts together cover >= --min-time seconds (default 0.5 s; the repeat count is agreed on by all
    # ranks from the first burst) and the MEDIAN burst is what `value` / `ms_per_step` report; min / max / count and the total
    # are in the JSON line (`timed_region_s`, `bursts`).
    def burst(record):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, record)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    bursts = [burst(True)]
    n_bursts = int(min(args.max_bursts, max(1, np.ceil(args.min_time / max(bursts[0], 1e-6)))))
    nb = torch.tensor([n_bursts], device=dev)
    if world > 1:
        dist.broadcast(nb, src=0)
    for _ in range(int(nb.item()) - 1):
        bursts.append(burst(False))
    elapsed = float(np.median(bursts))
    kernel_ms = [a.elapsed_time(b) for a, b in events]
    sum_iters = [sum_iters_one]

    n_opt = torch.tensor([min(int((o["status"] == 0).sum().item()) for o in outs)], device=dev)
    # scenarios the kernel accepted without certifying the objective accuracy (DSP_FLAG_OBJ_WAIVED): status 0, but HipPdlpSolver
    # would re-solve them and the Bidder would not bid them - counted here because the timed region calls DeviceLP.solve directly
    n_flag = torch.tensor([max(int(((o["flags"] & 1) != 0).sum().item()) for o in outs)], device=dev)
    if world > 1:
        dist.all_reduce(n_flag)
    if world > 1:
        dist.all_reduce(n_opt)
    rank_devices, dist_world = _rank_devices(world, dev)

    if rank == 0:
        total = n_total * args.steps
        value = total / elapsed
        # ---- roofline of the dominant kernel (the fused solve) ---------------------------------------------------
        # The fused kernel keeps x / y / A in registers and LDS, so HBM is NOT what bounds it (SURVEY.md 8(d) asks for
        # "effective bandwidth + the true limiter" in that case).  The limiters are FP64 VALU issue and the LDS pipe:
        #   VALU: SQ_INSTS_VALU wave-instructions per launch x 4 clk (an FP64 wave64 op occupies its SIMD for 4 cycles)
        #         / (1024 SIMDs x step time x clock)
        #   LDS : SQ_LDS_IDX_ACTIVE (LDS-array cycles per launch, summed over CUs) / (256 CUs x step time x clock)
        # Instruction / cycle counts per launch are deterministic for a given build + batch and come from the newest
        # committed rocprofv3 PMC summary of this same command (tools/gpu_profile.sh); the step time is measured live
        # here (whole timed region / steps, launches of different streams overlap, so this is the sustained rate).
        # Fractions are quoted at the 2.4 GHz peak clock, i.e. they are LOWER bounds of the busy fraction at the
        # clock the kernel really sustains.
        w = 8
        bytes_iter = 2 * w * (lp.n + lp.m)                                   # per scenario-iteration
        bytes_shared = 2 * (lp.nnz * (w + 4) + 4 * (lp.m + 1))
        step_s = elapsed / args.steps
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = float(np.mean(sum_iters)) * bytes_iter + bytes_shared + B * w * (3 * lp.n + 2 * lp.m + 1)
        # true per-launch I/O of the fused kernel: inputs that vary per scenario + outputs (x, y, obj, status, iters, jumps)
        true_io = (sum(t.numel() for t in (c_d, lb_d, ub_d, rlo_d, rhi_d) if t.dim() == 2) * w
                   + B * w * (lp.n + lp.m + 1) + B * 12)
        solve_kernel = f"pdlp_solve_kernel<{int(st.cols_per_lane)}, {int(st.rows_per_lane)}," if geometry[3] else "pdlp_solve_kernel"
        # The SQ counters of a launch are proportional to the scenario-iterations it runs (hot loop + checks; the per-scenario
        # prologue is < 1 %).  A profile records the iterations of its launch (profiles/<tag>_pmc_iterations.json); when the
        # shipped options / model hints have changed the iteration count since, the counters are scaled by the ratio so that
        # instructions and time belong to the same work.  `counters_scaled_by` = 1 when nothing changed or no record exists.
        valu_frac, lds_frac, valu_rate, pmc_file, counters_scaled_by, traffic = _valu_lds_fractions(solve_kernel, B, float(np.mean(sum_iters)), step_s)
        flops = 4.0 * lp.nnz * float(np.mean(sum_iters)) / step_s / 1e12      # SURVEY 8(d) flop unit: 4 nnz per scenario-iteration
        roofline = dict(
            bound="valu+lds", kernel="pdlp_solve_kernel",
            achieved=(valu_rate / 1e9) if valu_rate else None, peak=SIMDS * PEAK_CLOCK_HZ / 4.0 / 1e9,
            unit="G FP64-wave-instr/s", frac=valu_frac, frac_lds=lds_frac,
            traffic=traffic, true_io_bytes_per_launch=true_io,
            traffic_over_true_io=(traffic / true_io) if traffic else None,
            counters_from=pmc_file, counters_scaled_by=counters_scaled_by, step_ms=1e3 * step_s, kernel_latency_ms=k_ms,
            spmv_flops=dict(achieved=flops, peak=FP64_VECTOR_PEAK_TFLOPS, unit="TFLOP/s", frac=flops / FP64_VECTOR_PEAK_TFLOPS,
                            note="4 nnz flop per scenario-iteration (SURVEY 8(d)); the SpMV FMAs are 16 of the ~52 FP64 "
                                 "ops of an iteration, the rest is the PDHG update / Halpern step on the same registers"),
            effective_hbm=dict(achieved=alg_bytes / step_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=alg_bytes / step_s / 1e9 / HBM_PEAK_GBS, algorithmic_bytes_per_launch=alg_bytes,
                               note="SURVEY 8(d) algorithmic SpMV bytes / sustained step time: an EFFECTIVE bandwidth (may "
                                    "exceed the HBM peak) - these bytes are served from VGPRs / LDS and never reach HBM; the "
                                    "HBM-streaming form of the same step is reported under spmv_step"),
            note="fused register/LDS-resident solve: limiter = FP64 VALU issue + LDS pipe (frac / frac_lds, from the "
                 "committed SQ counters of this command and the step time measured here, at the 2.4 GHz peak clock); "
                 "traffic = FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE per launch from the same profile; kernel_latency_ms "
                 "= mean HIP-event duration of one launch on its stream (launches overlap, so it exceeds step_ms)")
        result = {
            "metric": "LP scenarios solved/sec, RTS-GMLC 24h multi-period dispatch, batch=4096" if (
                args.workload == "wind_battery_24h" and B == 4096) else
                f"LP scenarios solved/sec, {args.workload}, batch={B}",
            "value": value, "unit": "scenarios/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "timed_region_s": float(np.sum(bursts)), "bursts": len(bursts),
            "burst_ms": {"min": 1e3 * float(np.min(bursts)), "median": 1e3 * elapsed, "max": 1e3 * float(np.max(bursts))},
            "value_mean_over_bursts": total * len(bursts) / float(np.sum(bursts)),
            "lone_batch_scenarios_per_s": world * B / (1e-3 * single_batch_ms),
            "scaling": "strong" if args.total > 0 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "world_size": world, "collective_backend": (("gloo on host copies (REHEARSAL: ranks share cuda:0)" if share_gpu else
                                                         f"nccl(RCCL) {'.'.join(map(str, torch.cuda.nccl.version()))}")
                                                        if world > 1 else None),
            "rehearsal": bool(share_gpu), "dist_world_size": dist_world, "rank_devices": rank_devices,
            "distinct_gpus": len({d["uuid"] for d in rank_devices}),
            "config": {"workload": f"{args.workload}: {B_max} scenarios/GPU x {len(model.HOUR)} h day-ahead bidding LP "
                                   f"(n={lp.n}, m={lp.m}, nnz={lp.nnz}), synthetic scenarios from the in-tree RTS-GMLC / nuclear "
                                   f"LMP series (dispatches_amd/scenarios.py)",
                       "batch_per_gpu": B_max, "eps_rel": args.eps, "eps_obj": float(opts.eps_obj), "parallelism": f"scenario-sharded x{world}",
                       "parity_contract": "objective within 1e-6 relative of the oracle (HiGHS, tolerances 1e-9) for every scenario; every hourly setpoint inside "
                                          "the oracle's 1e-7-optimal face range +- 1e-6 * max(|setpoint|, generator p_max) - a NAMEPLATE tolerance: the optima are "
                                          "degenerate (repeated / zero prices), a setpoint is a range, not a number (tests/test_hip_batch_parity.py, DESIGN 2)",
                       "mean_iterations": float(np.mean(sum_iters)) / B, "max_iterations": max_iters_one,
                       "optimal": int(n_opt.item()), "flagged": int(n_flag.item()), "scenarios": n_total,
                       "grid": geometry[:2], "lds_bytes": geometry[2], "register_resident_matrix": bool(geometry[3]),
                       "simulated_lds_gather_conflict_cycles_per_iteration": {"identity_layout": lds_conflicts[0],
                                                                              "slot_permutation": lds_conflicts[1]},
                       "streams": depth, "stream_trials_ms_per_step": {str(k): 1e3 * v for k, v in trials.items()},
                       "single_batch_latency_ms": single_batch_ms,
                       "pipeline": f"steps issued round-robin on {depth} HIP streams (independent batches overlap; "
                                   "a lone batch takes single_batch_latency_ms, dominated by its slowest scenario: "
                                   "lone_batch_scenarios_per_s); value = median over `bursts` bursts of --steps steps each"},
            "roofline": roofline,
            "source_hash": _source_hash(),
        }
        # ---- streaming SpMV step (vectors in HBM): the kernel SURVEY 8(d) quotes the HBM roofline on -------
        if not args.no_spmv:
            def time_spmv(dlp, lp, Bs, reps):
                X = torch.randn((Bs, lp.n), dtype=torch.float64, device=dev)
                Y = torch.randn((Bs, lp.m), dtype=torch.float64, device=dev)
                AX = torch.empty((Bs, lp.m), dtype=torch.float64, device=dev)
                ATY = torch.empty((Bs, lp.n), dtype=torch.float64, device=dev)
                for _ in range(5):
                    dlp.spmv_step(X, Y, AX, ATY)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                # The launches are replayed from a hipGraph (50 per graph) and timed with events on the replay stream: a
                # 6-10 us kernel issued from Python through ctypes is HOST-bound (~5-8 us per call), which is not what this
                # figure is about - the streaming solver enqueues its sweeps the same way.  The plain loop is
                # the fallback if the capture fails.
                ms, how = None, "hipGraph replay of 50 back-to-back launches, events on the replay stream"
                try:
                    per_graph = min(50, reps)
                    side = torch.cuda.Stream(device=dev)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.stream(side):
                        dlp.spmv_step(X, Y, AX, ATY)
                        side.synchronize()
                        graph.capture_begin()
                        for _ in range(per_graph):
                            dlp.spmv_step(X, Y, AX, ATY)
                        graph.capture_end()
                    torch.cuda.synchronize()
                    graph.replay()
                    torch.cuda.synchronize()
                    n_rep = max(1, reps // per_graph)
                    best = None
                    for _ in range(3):
                        e0.record()
                        for _ in range(n_rep):
                            graph.replay()
                        e1.record()
                        torch.cuda.synchronize()
                        t = e0.elapsed_time(e1) / (n_rep * per_graph)
                        best = t if best is None else min(best, t)
                    ms = best
                except Exception as exc:                      # noqa: BLE001
                    how = f"plain launch loop (graph capture failed: {type(exc).__name__})"
                if ms is None:
                    e0.record()
                    for _ in range(reps):
                        dlp.spmv_step(X, Y, AX, ATY)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / reps
                bsp = Bs * 2 * w * (lp.n + lp.m) + 2 * (lp.nnz * (w + 4) + 4 * (lp.m + 1))
                return dict(bound="hbm", kernel="spmv_step_kernel", batch=Bs, achieved=bsp / (ms * 1e-3) / 1e9,
                            peak=HBM_PEAK_GBS, unit="GB/s", frac=bsp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            traffic=None, kernel_ms=ms, algorithmic_bytes_per_launch=bsp, timing=how)

            result["spmv_step"] = time_spmv(dlp, lp, B, 200)
            result["spmv_step"]["traffic"] = _profiled_traffic("spmv_step_kernel")
            result["spmv_step"]["note"] = ("one A x + one A^T y for every scenario of the batch with vectors streamed "
                                           "from/to HBM, results through non-temporal stores (matrix staged in LDS per 8-wave "
                                           "block, one scenario per wave, everything in flight at once); a 20 MB launch "
                                           "is bound by one memory round trip + the dispatch ramp - the BASELINE.md "
                                           "section 3 configuration (T = 48, 41 MB) is spmv_step_T48_B4096")
            # the same kernel on a batch large enough to be bandwidth bound (0.67 GB per launch, beyond L2 + MALL), and on twice the
            # metric batch - where the 20 MB launch's round trip + dispatch ramp stop dominating
            if args.spmv_large_mult > 0:
                result["spmv_step_large_batch"] = time_spmv(dlp, lp, args.spmv_large_mult * B, 20)
                result["spmv_step_2x_batch"] = time_spmv(dlp, lp, 2 * B, 100)
            # BASELINE.md section 3 quotes the contract figure at T = 48, B = 4096 (40.9 MB, <= 10.2 us <=> >= 50 %)
            if args.workload == "wind_battery_24h":
                _, m48 = scenarios.WORKLOADS["wind_battery_48h"][0](B=2, solver=solver, T=48)
                d48 = DeviceLP(m48.lp, local_rank, opts)
                result["spmv_step_T48_B4096"] = time_spmv(d48, m48.lp, 4096, 200)
                result["spmv_step_T48_B4096"]["note"] = "BASELINE.md section 3 configuration: wind+battery 48 h, 4096 scenarios"
                d48.close()
            # north_star: ">= 50 % of the HBM roofline on the PDLP SpMV step" - said plainly, per measured batch
            fr = {k: result[k]["frac"] for k in ("spmv_step", "spmv_step_2x_batch", "spmv_step_T48_B4096", "spmv_step_large_batch") if k in result}
            result["spmv_step_target"] = {"target_frac": 0.5, "measured": fr, "met_at": [k for k, v in fr.items() if v >= 0.5],
                                          "not_met_at": [k for k, v in fr.items() if v < 0.5],
                                          "statement": "the 0.5 target is met at BASELINE.md section 3's configuration (T = 48, 4096 scenarios, 41 MB) and from "
                                                       "about twice the metric batch on; NOT at the 20 MB metric batch itself, whose launch is one memory round "
                                                       "trip + the dispatch ramp (latency bound)"}
        # ---- the same batch at PDLP's default tolerance ("also report eps = 1e-4") --------------------------------
        # relative primal / dual residual and relative gap <= 1e-4, the objective-error bound of the contract setting off
        if world == 1 and not args.no_eps4:
            import copy
            o4 = copy.copy(opts)
            o4.eps_rel, o4.eps_obj = 1e-4, 0.0
            out4 = new_out()
            dlp.solve(B, c_d, lb_d, ub_d, rlo_d, rhi_d, options=o4, out=out4, sync_stats=True, obj_offset=c0_d)
            st4 = dlp.last_stats
            outs4 = [new_out() for _ in range(depth)]
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for i in range(args.steps):
                with torch.cuda.stream(streams[i % depth]):
                    dlp.solve(B, c_d, lb_d, ub_d, rlo_d, rhi_d, options=o4, out=outs4[i % depth], sync_stats=False, obj_offset=c0_d)
            torch.cuda.synchronize()
            t4 = time.perf_counter() - t4
            e4 = {"eps_rel": 1e-4, "eps_obj": 0.0, "value": B * args.steps / t4, "unit": "scenarios/s", "lone_batch_ms": float(st4.kernel_ms),
                  "mean_iterations": float(st4.total_iterations) / B, "max_iterations": int(st4.max_iterations), "optimal": int(st4.n_optimal),
                  "note": "PDLP's default termination (relative KKT error 1e-4) on the same batch, one burst of --steps steps on the same streams; "
                          "NOT the parity setting: see max_rel_obj_err_vs_oracle_fixture of this entry"}
            fx4 = os.path.join(ROOT, "tests", "golden", "oracle_objectives.npz")
            if os.path.exists(fx4):
                fx = np.load(fx4)
                if args.workload in fx.files and len(fx[args.workload]) >= B:
                    ref = fx[args.workload][:B]
                    mine4 = out4["obj"].cpu().numpy() + c0
                    e4["max_rel_obj_err_vs_oracle_fixture"] = float(np.max(np.abs(mine4 - ref) / np.maximum(1.0, np.abs(ref))))
            result["eps_1e-4"] = e4
        # ---- CPU baseline on this box's host cores (bounded sample) ------------------------------------------
        if world == 1 and args.cpu_sample != 0:
            procs = os.cpu_count() or 1
            sample = args.cpu_sample if args.cpu_sample > 0 else B
            base, ref_obj = cpu_baseline(args.workload, len(model.HOUR), sample, procs)
            result["cpu_baseline"] = base
            mine = out["obj"].cpu().numpy()[:sample] + c0[:sample]
            result["config"]["max_rel_obj_err_vs_cpu_baseline_sample"] = float(
                np.max(np.abs(mine - ref_obj) / np.maximum(1.0, np.abs(ref_obj))))
        # parity of the timed batch against the committed oracle fixture (HiGHS at tightened tolerances)
        fx_path = os.path.join(ROOT, "tests", "golden", "oracle_objectives.npz")
        if world == 1 and os.path.exists(fx_path):
            fx = np.load(fx_path)
            if args.workload in fx.files and len(fx[args.workload]) >= B:
                ref = fx[args.workload][:B]
                mine = out["obj"].cpu().numpy() + c0
                result["config"]["max_rel_obj_err_vs_oracle_fixture"] = float(
                    np.max(np.abs(mine - ref) / np.maximum(1.0, np.abs(ref))))
        # ---- every other BASELINE config, same run (default command only) ----------------------------------------------
        if world == 1 and args.workload == "wind_battery_24h" and args.total == 0 and not args.no_configs:
            result["configs"] = baseline_configs(args, local_rank, dev, depth=depth)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
