"""GPU development tool: where the cycles of the fused solve kernel go (hot loop / check base / KKT test / restart / ray-jump
test / iteration tail).  Needs a library built with -DDSP_PROF (prints one [prof] line per sampled block):
    DSP_LIB=libdsp_hip_prof.so python tools/gpu_check_profile.py [workload ...] 2>&1 | python tools/gpu_check_profile.py --sum"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--sum" in sys.argv:
    names = ["hot", "check base", "kkt", "restart", "jump test", "tail"]
    wl = None
    acc = {}
    for line in sys.stdin:
        if line.startswith("== "):
            wl = line[3:].strip(); acc[wl] = [[0, 0] for _ in names]
        m = re.findall(r"(-?\d+) (?:cyc )?/ (\d+)", line) if line.startswith("[prof]") else None
        if m and wl:
            for i, (c, n) in enumerate(m[:6]):
                acc[wl][i][0] += int(c); acc[wl][i][1] += int(n)
    for wl, a in acc.items():
        tot = sum(c for c, _ in a)
        nchk = max(1, a[1][1])
        print(wl, f"total {tot:.3g} cycles, {nchk} checks sampled, hot segment {a[0][0] / max(1, a[0][1]):.0f} cycles")
        for nm, (c, n) in zip(names, a):
            print(f"   {nm:10s} {100.0 * c / tot:5.1f} % of cycles   {n / nchk:6.3f} per check   {c / max(1, n):8.0f} cycles each")
    sys.exit(0)

if "--one" in sys.argv:                      # one workload per process: device printf output only arrives when the process ends
    from dispatches_amd import scenarios
    from dispatches_amd.hip_solver import HipPdlpSolver
    wl = sys.argv[sys.argv.index("--one") + 1]
    solver = HipPdlpSolver(device=0)
    bidder, model = scenarios.make_batch(wl, 4096, solver)
    solver.solve(model)
    print("iterations mean", model.iterations.mean(), flush=True)
    sys.exit(0)

import subprocess
for wl in (sys.argv[1:] or ["wind_battery_24h", "wind_battery_48h"]):
    print("==", wl, flush=True)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", wl], capture_output=True, text=True)
    print(out.stdout + out.stderr, flush=True)
