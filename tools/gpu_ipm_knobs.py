"""GPU development tool: robustness of the interior-point form against the rounding of its linear solves - the year-long family (64 lanes =
4 x its 16 members) under several counts of time partitions (each count is another elimination order: the same Newton iterations up to
rounding) and settings of the development knobs (DSP_IPM_REG / STEP / SIGMIN / REFTOL).
    python tools/gpu_ipm_knobs.py "<parts> <parts> ..." "KEY=VAL,KEY=VAL" ["KEY=VAL,..." ...]        (IPM_T / IPM_B: horizon and batch, default 8736 / 64)"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_ipm_diag import CHILD        # noqa: E402

parts_list = sys.argv[1].split()
for setting in sys.argv[2:]:
    knobs = dict(kv.split("=") for kv in setting.split(",") if "=" in kv)
    for parts in parts_list:
        env = dict(os.environ, DSP_IPM_PARTS=parts, DSP_IPM_TRACE="1", **knobs)
        p = subprocess.run([sys.executable, "-c", CHILD, os.environ.get("IPM_T", "8736"), os.environ.get("IPM_B", "64")], env=env, capture_output=True, text=True, timeout=600)
        lanes = [l for l in p.stderr.splitlines() if l.startswith("[ipm] lanes")]
        pairs = re.findall(r"(\d+):(\d+)", lanes[-1].split("iterations):")[1])[:16] if lanes else []
        res = re.search(r"form (\d+) .* wall (\S+)", p.stdout)
        print(setting or "default", "| parts", parts, "| form", res.group(1) if res else "?", "wall", res.group(2) if res else "?",
              "| given up:", [k for k, (s, it) in enumerate(pairs) if s != "1"], "| Newton iterations:", " ".join(it for s, it in pairs), flush=True)
