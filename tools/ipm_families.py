"""Lab driver (development tool; NOT product, NOT oracle): tools/ipm_lab.py on the three year-long price-taker families.
    python tools/ipm_families.py wb|pem|nuclear T member,member,..."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp
import stream_lab as lab, ipm_lab as ipm
from dispatches_amd import scenarios
class Dummy:
    def solve(self,*a,**k): raise RuntimeError
fam=sys.argv[1]; T=int(sys.argv[2]); members=[int(k) for k in sys.argv[3].split(",")]
B=max(members)+1
if fam=="pem": handles, model = scenarios.pem_price_taker_batch(T,B,Dummy())
elif fam=="nuclear": handles, model = scenarios.nuclear_price_taker_batch(T,B,Dummy())
else: handles, model = scenarios.price_taker_batch(T,B,Dummy(),throughput="chain")
lp=model.lp
lb,ub,rlo,rhi=model.scenario_bounds()
for member in members:
    pick=lambda a:(a[member] if a.ndim==2 else a).astype(float)
    P=dict(A=lp.csr(), c=model.c[member].astype(float), lb=pick(lb), ub=pick(ub), rlo=pick(rlo), rhi=pick(rhi), c0=float(model.c0[member]), lp=lp)
    ref,xr,th=lab.highs(P)
    free=(~np.isfinite(P["lb"])&~np.isfinite(P["ub"])).sum()
    t=time.time()
    try:
        X,Y,it,done=ipm.solve(P, verbose=0)
        obj=(P["c"]@X+P["c0"]) if X is not None else np.nan
        print(f"{fam} T={T} member={member} n={lp.n} m={lp.m} free={free} HiGHS {ref:.9e} ({th:.1f}s) ipm done={done} it={it} relerr={abs(obj-ref)/max(1,abs(ref)):.2e} t={time.time()-t:.1f}s",flush=True)
    except Exception as e:
        print(f"{fam} T={T} member={member} free={free} FAILED {type(e).__name__}: {e}",flush=True)
