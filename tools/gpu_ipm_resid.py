"""GPU development tool: how well the Newton systems of one scenario are solved (|rhs - N dy| / |rhs| after the refinement steps, predictor
and corrector) under several counts of time partitions.    python tools/gpu_ipm_resid.py <lane> <parts> [<parts> ...]"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_ipm_diag import CHILD        # noqa: E402  (importing runs nothing: main() is guarded below)

lane = int(sys.argv[1])
for parts in sys.argv[2:]:
    env = dict(os.environ, DSP_IPM_PARTS=parts, DSP_IPM_TRACE=str(lane + 1), DSP_IPM_DEBUG="1")
    p = subprocess.run([sys.executable, "-c", CHILD, "8736", "64"], env=env, capture_output=True, text=True, timeout=600)
    print(p.stdout.strip())
    rows, cur = [], []
    for l in p.stderr.splitlines():
        m = re.search(r"mode (\d): \|rhs\| (\S+) \|dy\| (\S+) .* (\S+)$", l)
        if m:
            cur.append(float(m.group(4)))
        m = re.match(r"\[ipm\] it (\d+) lane \d+: mu (\S+) .* ap (\S+) ad (\S+) rp (\S+) rd (\S+) rg (\S+) .* refine (\d+)", l)
        if m:
            rows.append((int(m.group(1)), m.group(2), m.group(3), m.group(4), m.group(5), m.group(7), m.group(8), cur))
            cur = []
    print(f"--- parts {parts}, lane {lane}: it  mu  ap  ad  rp  rg  refine | relative residual of the Newton system (predictor, corrector)")
    for r in rows[:140]:
        if r[0] % 4 == 0 or r[0] > 70:
            print(r[0], r[1], r[2], r[3], r[4], r[5], r[6], "|", " ".join("%.1e" % v for v in r[7]))
