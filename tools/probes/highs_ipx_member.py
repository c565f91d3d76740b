import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, scipy.sparse as sp
import stream_lab as lab
from scipy.optimize import linprog
m=int(sys.argv[1])
P = lab.build(8736, m, None, "chain", family="wide")
A, lo, hi = P["A"], P["rlo"], P["rhi"]
eq = np.isfinite(lo) & (lo == hi); up = np.isfinite(hi) & ~eq; dn = np.isfinite(lo) & ~eq
Aub = sp.vstack([A[up], -A[dn]]).tocsr(); bub = np.concatenate([hi[up], -lo[dn]])
t=time.time()
res = linprog(P["c"], A_ub=Aub, b_ub=bub, A_eq=A[eq], b_eq=hi[eq], bounds=np.stack([P["lb"], P["ub"]], 1), method="highs-ipm", options=dict(disp=True))
print(res.status, res.fun+P["c0"], res.nit, time.time()-t)
