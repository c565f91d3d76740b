"""GPU development tool: which scenarios of a year-long batch the interior-point form gives up on for a given count of time partitions
(DSP_IPM_PARTS), and the Newton iterations of the first one of them next to the same scenario under another count.
    python tools/gpu_ipm_diag.py <parts to examine> <parts to compare with> [T] [B]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from dispatches_amd import scenarios
from dispatches_amd.hip_solver import HipPdlpSolver
T, B = int(sys.argv[1]), int(sys.argv[2])
solver = HipPdlpSolver(device=0, check_every=64, max_iter=256)          # (the PDHG forms stop at once: only the interior-point form is looked at)
handles, model = scenarios.price_taker_batch(T, B, solver, throughput="chain")
t = time.time(); solver.solve(model); w = time.time() - t
st = solver.last_stats
print("RESULT parts", os.environ.get("DSP_IPM_PARTS"), "form", st.stream_form, "phases", st.stream_phases, "status", np.bincount(model.status, minlength=5).tolist(), "wall %%.2f" %% w, flush=True)
''' % ROOT


def run(parts, trace, T, B):
    env = dict(os.environ, DSP_IPM_PARTS=str(parts), DSP_IPM_TRACE=str(trace))
    p = subprocess.run([sys.executable, "-c", CHILD, str(T), str(B)], env=env, capture_output=True, text=True, timeout=400)
    return p.stdout, p.stderr


def main():
    bad, good = sys.argv[1], sys.argv[2]
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 8736
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    out, err = run(bad, 1, T, B)
    print(out.strip())
    lanes = [l for l in err.splitlines() if l.startswith("[ipm] lanes")]
    print(lanes[-1][:1200] if lanes else err[-2000:])
    failing = []
    if lanes:
        for tok in lanes[-1].split(":", 1)[1].split():
            pass
        pairs = re.findall(r"(\d+):(\d+)", lanes[-1].split("iterations):")[1])
        failing = [k for k, (s, it) in enumerate(pairs) if s == "2"]
    print("given up:", failing)
    lane = failing[0] if failing else 0
    for parts in (bad, good):
        out, err = run(parts, lane + 1, T, B)
        print(out.strip())
        its = [l for l in err.splitlines() if l.startswith("[ipm] it ")]
        print(f"--- parts {parts}, lane {lane}: {len(its)} Newton iterations; the last 40:")
        for l in its[-40:]:
            print(l[:230])


if __name__ == "__main__":
    main()
