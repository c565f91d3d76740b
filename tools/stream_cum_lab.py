"""Lab (development tool; NOT product, NOT oracle): the accumulated energy throughput of the price-taker LP ELIMINATED.

E_t = sum_{s <= t} (I_s + O_s) / 2 appears in ONE place, the state-of-charge bound S_t + d E_t <= 4 P.  Substituting the sum removes
the T free columns E_t and the T equality rows E_t - E_{t-1} = (I_t + O_t) / 2 - the integrator whose multipliers diffuse one period
per iteration - and leaves a lower-triangular block d / 2 (I_s + O_s), s <= t, in the bound rows: T^2 entries as a matrix, but as an
OPERATOR a running sum forward in time (A x) and backward in time (A^T y): exactly what a lane that walks its tile's periods in
order can carry in a register, with one exclusive prefix over the tiles per product.
    python tools/stream_cum_lab.py T=672 member=5
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.sparse as sp

import stream_lab as lab


def eliminate_throughput(P, T):
    lp = P["lp"]
    names = lp.col_names
    idx = lambda fam: np.array([names.index(f"{fam}[{t}]") for t in range(T)])
    E, I, O = idx("battery.energy_throughput"), idx("splitter.battery_elec"), idx("battery.elec_out")
    keep_cols = np.setdiff1d(np.arange(lp.n), E)
    pos = -np.ones(lp.n, int); pos[keep_cols] = np.arange(len(keep_cols))
    # x_old = G x_new
    G = sp.lil_matrix((lp.n, len(keep_cols)))
    for j in keep_cols:
        G[j, pos[j]] = 1.0
    L = sp.tril(np.ones((T, T)), format="coo")
    rows = np.concatenate([E[L.row], E[L.row]]); cols = np.concatenate([pos[I[L.col]], pos[O[L.col]]])
    G = sp.csr_matrix(G) + sp.csr_matrix((np.full(len(rows), 0.5), (rows, cols)), shape=G.shape)
    A = sp.csr_matrix(P["A"] @ G)
    acc = np.array([i for i, nm in enumerate(lp.row_names) if nm.startswith("battery.accumulate_energy_throughput[")])
    A.eliminate_zeros()
    chk = abs(A[acc]).max() if len(acc) else 0.0
    assert chk < 1e-9, chk                                # the accumulation rows are identities now
    keep_rows = np.setdiff1d(np.arange(lp.m), acc)
    Q = dict(P)
    Q["A"] = sp.csr_matrix(A[keep_rows])
    Q["c"] = np.asarray(G.T @ P["c"]).ravel()
    Q["lb"], Q["ub"] = P["lb"][keep_cols], P["ub"][keep_cols]
    Q["rlo"], Q["rhi"] = P["rlo"][keep_rows], P["rhi"][keep_rows]
    assert np.all(P["lb"][E] <= 0) and np.all(np.isinf(P["ub"][E]))          # E >= 0 is implied (I, O >= 0)

    class LP:
        pass
    q = LP(); q.n, q.m = len(keep_cols), len(keep_rows); q.col_names = [names[j] for j in keep_cols]; q.row_names = [lp.row_names[i] for i in keep_rows]
    q.nnz = Q["A"].nnz
    Q["lp"] = q
    return Q


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    T = int(kw.pop("T", 672)); member = int(kw.pop("member", 5))
    which = kw.pop("which", "both")
    P = lab.build(T, member, None, "chain")
    ref, xr, th = lab.highs(P)
    print(f"T={T} member={member} chain: n={P['lp'].n} m={P['lp'].m} nnz={P['lp'].nnz} HiGHS {ref:.10e} ({th:.1f}s)", flush=True)
    opts = {k: float(v) for k, v in kw.items()}
    if which in ("both", "chain"):
        t = time.time()
        X, Y, it, nrs, done, _ = lab.solve(P, colscale=lab.physical_scales(P, T), **opts)
        obj = P["c"] @ X + P["c0"]
        print(f"  chain       done={done} iters={it} restarts={nrs} relerr={abs(obj-ref)/max(1,abs(ref)):.2e} t={time.time()-t:.0f}s", flush=True)
    Q = eliminate_throughput(P, T)
    print(f"  eliminated: n={Q['lp'].n} m={Q['lp'].m} nnz={Q['lp'].nnz}", flush=True)
    t = time.time()
    X, Y, it, nrs, done, _ = lab.solve(Q, colscale=lab.physical_scales(Q, T), **opts)
    obj = Q["c"] @ X + Q["c0"]
    print(f"  eliminated  done={done} iters={it} restarts={nrs} relerr={abs(obj-ref)/max(1,abs(ref)):.2e} t={time.time()-t:.0f}s", flush=True)
