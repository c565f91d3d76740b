"""Generate tests/golden/oracle_objectives.npz: objectives of the first B scenarios of every bench workload from the
INDEPENDENT CPU oracle (oracle/dispatch_lp_oracle.py: un-reduced Appendix-A LPs solved by HiGHS with tightened
tolerances).  The GPU parity tests compare the HIP path against these at the full BASELINE batch size without
running the oracle on the GPU box.   python tools/make_oracle_fixtures.py [B]"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _work(args):
    wl, ids = args
    sys.path.insert(0, ROOT)
    from dispatches_amd import scenarios
    from oracle import dispatch_lp_oracle as orc
    T = 48 if wl.endswith("48h") else 24
    out = []
    if wl.startswith("wind"):
        s = scenarios.load_series("rts_gmlc_309.npz" if "battery" in wl else "rts_gmlc_303.npz")
        N = len(s["rt_lmp"])
        stride = 17 if "battery" in wl else 37
        for k in ids:
            h0 = (stride * k) % (N - T)
            da, rt = np.clip(s["da_lmp"][h0:h0 + T], 0, 500), np.clip(s["rt_lmp"][h0:h0 + T], 0, 500)
            cf = s["rt_cf"][h0:h0 + T]
            P = orc.wind_battery_da(T, cf, da, rt)[0] if "battery" in wl else orc.wind_pem_da(T, cf, da, rt, wind_kw=847e3)[0]
            out.append(P.solve(tight=True)[1])
    else:
        class _NoSolver:
            def solve(self, *a, **k):
                raise RuntimeError
        _, model = scenarios.make_batch(wl, max(ids) + 1, _NoSolver())
        for k in ids:
            out.append(orc.nuclear_da(T, model.da_prices[k], model.rt_prices[k])[0].solve(tight=True)[1])
    return out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    from dispatches_amd import scenarios
    res = {}
    procs = os.cpu_count() or 1
    with mp.get_context("spawn").Pool(procs) as pool:
        for wl in scenarios.WORKLOADS:
            chunks = [(wl, c.tolist()) for c in np.array_split(np.arange(B), procs * 4)]
            res[wl] = np.concatenate(pool.map(_work, chunks))
            print(wl, res[wl][:3], flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_objectives.npz"), **res)
