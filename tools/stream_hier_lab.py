"""Lab (development tool; NOT product, NOT oracle): the storage chains of the price-taker LP in a HIERARCHICAL basis.

PDHG needs O(T) iterations on LP #4 because the state recursions S_t - S_{t-1} = ... are first differences: sigma_min of the
difference operator falls like 1 / T.  In the hierarchical hat basis S = H z the difference operator has orthogonal columns
(D H)^T (D H) = diagonal, so the recursion rows are perfectly conditioned in z - at the price of O(T log T) nonzeros, a matrix that
is no longer banded, and bound rows (S >= 0, S <= cap - d E) that now carry H.  This script substitutes S (and optionally E) and
counts iterations of the numpy restatement of the streaming PDLP.       python tools/stream_hier_lab.py T=672 member=5 which=S
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.sparse as sp

import stream_lab as lab


def hat_basis(T):
    """T x T hierarchical hat basis on the points 0 .. T-1 (plus the left end value as its own column): column 0 = constant 1,
    column 1 = the linear function t / (T - 1), then the midpoint hats of the bisection tree."""
    cols = [np.ones(T), np.arange(T) / max(T - 1, 1)]
    stack = [(0, T - 1)]
    while stack:
        lo, hi = stack.pop()
        if hi - lo < 2:
            continue
        mid = (lo + hi) // 2
        v = np.zeros(T)
        v[lo:mid + 1] = (np.arange(lo, mid + 1) - lo) / (mid - lo)
        v[mid:hi + 1] = (hi - np.arange(mid, hi + 1)) / (hi - mid)
        cols.append(v)
        stack += [(lo, mid), (mid, hi)]
    H = sp.csr_matrix(np.stack(cols, 1))
    assert H.shape == (T, T), H.shape
    return H


def transform(P, T, which):
    lp = P["lp"]
    names = lp.col_names
    fam = {"S": "battery.state_of_charge", "E": "battery.energy_throughput"}
    G = sp.identity(lp.n, format="lil")
    lb, ub = P["lb"].copy(), P["ub"].copy()
    extra_rows, extra_lo, extra_hi = [], [], []
    for key in which:
        idx = np.array([names.index(f"{fam[key]}[{t}]") for t in range(T)])
        H = hat_basis(T).tocoo()
        G[np.ix_(idx, idx)] = 0.0
        for i, j, v in zip(H.row, H.col, H.data):
            G[idx[i], idx[j]] = v
        # the chain's column bounds become rows on H z
        rows = sp.lil_matrix((T, lp.n))
        for i, j, v in zip(H.row, H.col, H.data):
            rows[i, idx[j]] = v
        if key == "S":                                 # S >= 0 (E >= 0 is implied by the accumulation)
            extra_rows.append(rows.tocsr()); extra_lo.append(lb[idx].copy()); extra_hi.append(ub[idx].copy())
        lb[idx], ub[idx] = -np.inf, np.inf
    G = sp.csr_matrix(G)
    A = sp.csr_matrix(P["A"] @ G)
    A.eliminate_zeros()
    rlo, rhi = P["rlo"], P["rhi"]
    if extra_rows:
        A = sp.vstack([A] + extra_rows).tocsr()
        rlo, rhi = np.concatenate([rlo] + extra_lo), np.concatenate([rhi] + extra_hi)
    Q = dict(P)
    Q.update(A=A, c=G.T @ P["c"], lb=lb, ub=ub, rlo=rlo, rhi=rhi)
    return Q, G


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    T, member, which = int(kw.get("T", 672)), int(kw.get("member", 5)), kw.get("which", "S")
    P = lab.build(T, member)
    ref, xr, th = lab.highs(P)
    cs = lab.physical_scales(P, T)
    print(f"T={T} member={member} n={P['lp'].n} m={P['lp'].m} nnz={P['A'].nnz}: HiGHS {ref:.9e}", flush=True)
    t = time.time()
    X, Y, it, nrs, done, _ = lab.solve(P, colscale=cs)
    print(f"  original          iterations={it:7d} restarts={nrs} done={done} relerr={abs(P['c'] @ X + P['c0'] - ref) / max(1, abs(ref)):.1e} ({time.time() - t:.0f}s)", flush=True)
    for w in which.split(","):
        Q, G = transform(P, T, w)
        t = time.time()
        X, Y, it, nrs, done, _ = lab.solve(Q, colscale=cs)
        x = G @ X
        print(f"  hierarchical {w:4s} iterations={it:7d} restarts={nrs} done={done} relerr={abs(P['c'] @ x + P['c0'] - ref) / max(1, abs(ref)):.1e} "
              f"nnz={Q['A'].nnz} rows={Q['A'].shape[0]} ({time.time() - t:.0f}s)", flush=True)
