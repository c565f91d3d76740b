"""The expensive half of the Bidder's bid assembly as tensor operations (torch: the device the solution lives on, or the CPU).

Reference: upstream idaes `Bidder._assemble_bids` - per hour the (power, marginal price) pairs of all scenarios, each rounded with
Python's round(x, 2), grouped by power, the highest price per power kept (golden G2:
dispatches/case_studies/renewables_case/tests/test_multiperiod_wind_battery_doubleloop.py:245-250).  For 4096 scenarios x 24 h that
is 98 k roundings and 24 sorts - most of the host time of `compute_day_ahead_bids` once the solve takes 2 ms.  Here:

    cents(a)            round(a, 2) * 100 as int64, EXACTLY Python's round (correctly rounded decimal of the double, ties to even):
                        a * 200 is formed exactly as a double-double (Dekker's product, no FMA needed) and compared with the odd
                        integer 2 r + 1 - plain IEEE double operations, identical on CPU and GPU
    sorted_pairs(...)   per hour: pairs below p_min or of failed scenarios dropped, ONE batched sort of a composite integer key
                        (power ascending, price descending), so that the first pair of every run of equal powers carries the
                        run's highest price

What is left for the host (workflow/bidder.py) works on the distinct powers of an hour - a few dozen to a few thousand numbers:
the p_min point, the running maximum, the cost integration (kept sequential: bit-identical to the numpy path).
`tests/test_bid_curves_cpu.py` pins both functions against the numpy path on adversarial inputs (exact ties, duplicates)."""
from __future__ import annotations

import numpy as np

_SPLIT = 134217729.0          # 2^27 + 1 (Veltkamp split of a double into two 26-bit halves)
_KEY_OFF = 1 << 31            # price cents are reflected into [0, 2^32) in the low half of the composite key
_DROP = (1 << 63) - 1         # key of a pair that takes no part (sorted to the end)


def cents(torch, a):
    """round(a, 2) * 100 as an int64 tensor, for every finite |a| < 2^31 / 100 (NaN / inf / beyond that: 0 - the caller masks those)."""
    a = a.to(torch.float64)
    p = a * 200.0                                   # rounded product
    c = a * _SPLIT
    hi = c - (c - a)
    lo = a - hi
    err = (hi * 200.0 - p) + lo * 200.0             # exact: a * 200 = p + err  (hi * 200 and lo * 200 are exact, 26 + 8 bits)
    r = torch.floor(a * 100.0)                      # candidate: the exact quotient floor is r or r +- 1 (a * 100 is off by <= 1 ulp)
    # exact comparison of a * 200 with the odd integers around it: move r until 2 r <= a * 200 < 2 r + 2
    for _ in range(2):
        below = ((p - 2.0 * r) + err) < 0.0         # a * 100 < r
        r = torch.where(below, r - 1.0, r)
        above = ((p - (2.0 * r + 2.0)) + err) >= 0.0   # a * 100 >= r + 1
        r = torch.where(above, r + 1.0, r)
    d = (p - (2.0 * r + 1.0)) + err                 # sign of (a * 100 - (r + 1/2)), exact
    odd = torch.remainder(r, 2.0) != 0.0
    up = (d > 0.0) | ((d == 0.0) & odd)             # above the midpoint, or exactly on it with an odd floor: round to even
    r = torch.where(up, r + 1.0, r)
    ok = torch.isfinite(a) & (a.abs() < 2.0e7)
    return torch.where(ok, r, torch.zeros_like(r)).to(torch.int64)


def sorted_pairs(torch, power, price, p_min, ok=None):
    """power, price: [B, T] float64 tensors (same device).  Returns (p_cents, c_cents, first) as [B, T] tensors sorted per hour by power
    (ascending): `first[i, t]` marks the first pair of a run of equal powers, whose c_cents is the highest price offered at that power;
    dropped pairs (power < p_min after rounding, scenarios with ok[s] False) sit at the end with first = False."""
    pc, cc = cents(torch, power), cents(torch, price)
    keep = (pc.to(torch.float64) / 100.0) >= float(p_min)
    if ok is not None:
        keep = keep & ok.reshape(-1, 1)
    keep = keep & torch.isfinite(power) & torch.isfinite(price)
    # signed power cents in the high half, the reflected price in [0, 2^32) below it: power ascending, then price DESCENDING
    # (the same key as csrc/dsp_bids.hip)
    key = pc * (1 << 32) + ((_KEY_OFF - 1) - cc)
    key = torch.where(keep, key, torch.full_like(key, _DROP))
    key, _ = torch.sort(key, dim=0)
    live = key != _DROP
    ps = key >> 32                                                  # arithmetic shift = floor: the low half is non-negative
    cs = (_KEY_OFF - 1) - (key & 0xFFFFFFFF)
    first = live.clone()
    first[1:] &= ps[1:] != ps[:-1]
    return ps, cs, first


def compact(torch, ps, cs, first):
    """One download for the whole day: (points [n_total, 2] int32 as a numpy array - power cents, price cents, hour after hour -, counts [T])."""
    pt, ct, ft = ps.t().contiguous(), cs.t().contiguous(), first.t().contiguous()       # hour-major: an hour's points are contiguous
    counts = ft.sum(dim=1)
    packed = torch.stack([pt[ft], ct[ft]], dim=1).to(torch.int32)
    return packed.cpu().numpy(), counts.cpu().numpy()


def hour_points(packed, counts):
    """Host side: per hour (distinct powers, highest price at each) as float64 numpy arrays, from compact()'s download."""
    vals = packed / 100.0
    ends = np.cumsum(counts)
    return [(vals[e - n:e, 0], vals[e - n:e, 1]) for n, e in zip(counts.tolist(), ends.tolist())]


def padded(points, width=None):
    """hour_points() / numpy-path lists -> (counts [T], powers [T, W], prices [T, W]) with one spare column (curves() may insert a point)."""
    counts = np.array([len(p) for p, _ in points], np.int64)
    W = max(int(counts.max()) if len(counts) else 0, width or 0) + 1
    U, M = np.zeros((len(points), W)), np.zeros((len(points), W))
    for t, (p, c) in enumerate(points):
        U[t, :len(p)], M[t, :len(p)] = p, c
    return counts, U, M


def curves(counts, U, M, p_min, pmin2):
    """The host half of the bid assembly for ALL hours at once (numpy).  counts [T]; U, M [T, W]: per hour the distinct offered powers
    (ascending) and the highest marginal price at each, W > max(counts).  Per hour, exactly what the reference does to its sorted
    pairs: an hour without a point AT p_min gets (round(p_min, 2), lowest price seen or 0) inserted in order; marginal prices made
    non-decreasing; integrated to costs, cost[0] = power[0] * price[0], cost[k] = cost[0] + (d_1 + ... + d_k) with
    d_i = (power[i] - power[i-1]) * price[i] summed left to right (np.cumsum along the hour: the order of the per-hour loop this
    replaces, so the floats are the same - tests/test_bid_curves_cpu.py compares them bit for bit).
    Returns (counts, powers, costs)."""
    T, W = U.shape
    n = np.asarray(counts, np.int64)
    idx = np.arange(W)[None, :]
    valid = idx < n[:, None]
    ins = ~((U == p_min) & valid).any(axis=1)
    if ins.any():
        lowest = np.where(valid, M, np.inf).min(axis=1)
        lowest[n == 0] = 0.0
        k = (valid & (U < pmin2)).sum(axis=1)                       # np.searchsorted(up, pmin2) of the sorted powers
        shift = ins[:, None] & (idx > k[:, None])
        U, M = np.take_along_axis(U, idx - shift, 1), np.take_along_axis(M, idx - shift, 1)
        at = ins[:, None] & (idx == k[:, None])
        U, M = np.where(at, pmin2, U), np.where(at, lowest[:, None], M)
        n = n + ins
        valid = idx < n[:, None]
    M = np.maximum.accumulate(np.where(valid, M, -np.inf), axis=1)   # (every hour has a point now: finite from column 0 on)
    U = np.where(valid, U, U[np.arange(T), n - 1][:, None])          # padding repeats the last power: zero increments behind it
    cost = np.empty_like(U)
    cost[:, 0] = U[:, 0] * M[:, 0]
    if W > 1:
        cost[:, 1:] = cost[:, :1] + np.cumsum(np.diff(U, axis=1) * M[:, 1:], axis=1)
    return n, U, cost



class PairList:
    """The (power, cost) pairs of ONE bid curve as the reference's consumers read them - a sequence of 2-tuples - held as the two
    arrays the curve was computed in.  The tuples are made when somebody looks (iteration, indexing, comparison, repr): a 4096-scenario
    day is 24 curves of ~1 300 points, and building 31 k tuples eagerly was a fifth of `compute_day_ahead_bids` although a market
    reads one generator-hour at a time (and the records never read them: `_record_bids` takes the arrays)."""
    __slots__ = ("power", "cost", "_pairs")

    def __init__(self, power, cost):
        self.power, self.cost, self._pairs = power, cost, None

    def _list(self):
        if self._pairs is None:
            self._pairs = list(zip(self.power.tolist(), self.cost.tolist()))
        return self._pairs

    def __len__(self):
        return len(self.power)

    def __getitem__(self, i):
        return self._list()[i]

    def __iter__(self):
        return iter(self._list())

    def __eq__(self, other):
        if isinstance(other, PairList):
            other = other._list()
        if isinstance(other, (list, tuple)):
            return self._list() == list(other)
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __repr__(self):
        return repr(self._list())

    def __reduce__(self):                               # pickles / deep copies as the plain list
        return (list, (self._list(),))


import collections.abc as _abc
_abc.Sequence.register(PairList)
