// dsp_prepare.hpp — host-side, once-per-(flowsheet, horizon) preparation of the shared problem data:
//   * diagonal preconditioner  (Ruiz inf-norm equilibration, then Pock-Chambolle alpha = 1)
//   * step size eta = step_scale / ||D_r A D_c||_2   (power iteration)
//   * lane-major ELL storage of the scaled A (rows) and A^T (columns) for a 64-lane wave that owns
//     columns {lane, lane+64, ...} and rows {lane, lane+64, ...}; vectors longer than the ELL width go to a
//     "long vector" list that the whole wave reduces cooperatively.
// Everything the reference delegates to CBC / IPOPT presolve+scaling lives here; there is no reference source
// for it (K5), the algorithm is PDLP's published preconditioning.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "dsp_shapes.hpp"

namespace dsp {

struct HostCSR {
  int m = 0, n = 0;
  std::vector<int32_t> ptr, idx;
  std::vector<double> val;
  int64_t nnz() const { return (int64_t)idx.size(); }
};

inline HostCSR transpose(const HostCSR &A) {
  HostCSR T;
  T.m = A.n; T.n = A.m;
  T.ptr.assign(A.n + 1, 0);
  for (int32_t j : A.idx) T.ptr[j + 1]++;
  for (int j = 0; j < A.n; ++j) T.ptr[j + 1] += T.ptr[j];
  T.idx.resize(A.idx.size()); T.val.resize(A.val.size());
  std::vector<int32_t> pos(T.ptr.begin(), T.ptr.end() - 1);
  for (int i = 0; i < A.m; ++i)
    for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
      int q = pos[A.idx[p]]++;
      T.idx[q] = i; T.val[q] = A.val[p];
    }
  return T;
}

// Ruiz (inf-norm, `iters` passes) followed by Pock-Chambolle (alpha = 1).  A is scaled in place:
// A <- diag(dr) A diag(dc).  `col0` (optional, [n], > 0): the caller's variable scaling factors - typical magnitudes of the
// columns, x_j = col0_j x~_j - applied BEFORE the equilibration and folded into dc (dsp_lp_desc::col_scale): Ruiz and
// Pock-Chambolle balance the matrix, they know nothing about the ranges the variables live in (kW next to MW next to kWh of
// throughput: 8 decades in the wind + battery LP), and the fixed point they reach depends on where they start.
inline void equilibrate(HostCSR &A, int ruiz_iters, std::vector<double> &dr, std::vector<double> &dc, int geo_iters = 0,
                        const double *col0 = nullptr) {
  dr.assign(A.m, 1.0); dc.assign(A.n, 1.0);
  std::vector<double> rs(A.m), cs(A.n);
  auto apply = [&]() {
    for (int i = 0; i < A.m; ++i)
      for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) A.val[p] *= rs[i] * cs[A.idx[p]];
    for (int i = 0; i < A.m; ++i) dr[i] *= rs[i];
    for (int j = 0; j < A.n; ++j) dc[j] *= cs[j];
  };
  if (col0) {
    std::fill(rs.begin(), rs.end(), 1.0);
    for (int j = 0; j < A.n; ++j) cs[j] = col0[j];
    apply();
  }
  // optional geometric-mean passes: r_i = 1/sqrt(max_j|a_ij| min_j|a_ij|), then the same for columns
  for (int it = 0; it < geo_iters; ++it) {
    std::vector<double> mx(A.m, 0.0), mn(A.m, INFINITY);
    for (int i = 0; i < A.m; ++i)
      for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
        double a = std::fabs(A.val[p]);
        if (a > 0) { mx[i] = std::max(mx[i], a); mn[i] = std::min(mn[i], a); }
      }
    for (int i = 0; i < A.m; ++i) rs[i] = mx[i] > 0 ? 1.0 / std::sqrt(mx[i] * mn[i]) : 1.0;
    std::fill(cs.begin(), cs.end(), 1.0);
    apply();
    std::vector<double> cmx(A.n, 0.0), cmn(A.n, INFINITY);
    for (int i = 0; i < A.m; ++i)
      for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
        double a = std::fabs(A.val[p]);
        if (a > 0) { cmx[A.idx[p]] = std::max(cmx[A.idx[p]], a); cmn[A.idx[p]] = std::min(cmn[A.idx[p]], a); }
      }
    for (int j = 0; j < A.n; ++j) cs[j] = cmx[j] > 0 ? 1.0 / std::sqrt(cmx[j] * cmn[j]) : 1.0;
    std::fill(rs.begin(), rs.end(), 1.0);
    apply();
  }
  for (int it = 0; it < ruiz_iters; ++it) {
    std::fill(rs.begin(), rs.end(), 0.0); std::fill(cs.begin(), cs.end(), 0.0);
    for (int i = 0; i < A.m; ++i)
      for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
        double a = std::fabs(A.val[p]);
        rs[i] = std::max(rs[i], a); cs[A.idx[p]] = std::max(cs[A.idx[p]], a);
      }
    for (auto &v : rs) v = v > 0 ? 1.0 / std::sqrt(v) : 1.0;
    for (auto &v : cs) v = v > 0 ? 1.0 / std::sqrt(v) : 1.0;
    apply();
  }
  // Pock-Chambolle, alpha = 1: row scale 1/sqrt(sum_j |a_ij|), column scale 1/sqrt(sum_i |a_ij|)
  std::fill(rs.begin(), rs.end(), 0.0); std::fill(cs.begin(), cs.end(), 0.0);
  for (int i = 0; i < A.m; ++i)
    for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
      double a = std::fabs(A.val[p]);
      rs[i] += a; cs[A.idx[p]] += a;
    }
  for (auto &v : rs) v = v > 0 ? 1.0 / std::sqrt(v) : 1.0;
  for (auto &v : cs) v = v > 0 ? 1.0 / std::sqrt(v) : 1.0;
  apply();
}

// ||A||_2 by power iteration on A^T A.  The iterates' estimates increase monotonically towards sigma_max, i.e. every stop
// is an UNDER-estimate, and eta = step_scale / estimate with an estimate short by more than 1 - step_scale (0.2 %) makes the PDHG
// operator expansive: the iteration drifts off at constant speed instead of converging.  Small LPs converge to machine precision
// within `iters`; the long-horizon price-taker LPs (block-bidiagonal in time: a near-continuous top of the spectrum) do not -
// 200 iterations are 0.6 % short at T = 1344 (tools/stream_lab.py), which is what kept the T = 8736 solves of round 2 from ever
// converging.  So: at least `iters` iterations (unchanged results where those suffice), then on until the estimate has stopped
// growing (relative growth < 1e-10 over 50 iterations) or `max_iters`; if it is still growing, the bound of the
// Pock-Chambolle scaling that `equilibrate` always ends with (alpha = 1: ||D_r A D_c||_2 <= 1) is returned instead.
inline double spectral_norm(const HostCSR &A, const HostCSR &AT, int iters = 300, int max_iters = 4000, double pc_bound = 1.0) {
  if (A.nnz() == 0) return 1.0;
  std::vector<double> v(A.n), u(A.m);
  uint64_t s = 0x9E3779B97F4A7C15ull;                       // fixed-seed xorshift start vector
  for (auto &x : v) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
  double nrm = 0, ref = 0;
  auto normalise = [&](std::vector<double> &w) {
    double t = 0; for (double x : w) t += x * x; t = std::sqrt(t);
    if (t > 0) for (double &x : w) x /= t;
    return t;
  };
  normalise(v);
  bool converged = false;
  for (int it = 0; it < max_iters; ++it) {
    for (int i = 0; i < A.m; ++i) { double t = 0; for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) t += A.val[p] * v[A.idx[p]]; u[i] = t; }
    for (int j = 0; j < AT.m; ++j) { double t = 0; for (int p = AT.ptr[j]; p < AT.ptr[j + 1]; ++p) t += AT.val[p] * u[AT.idx[p]]; v[j] = t; }
    nrm = std::sqrt(normalise(v));                           // ||A^T A v|| -> sigma_max^2
    if ((it + 1) % 50 == 0) {
      converged = it + 1 > 50 && nrm - ref <= 1e-10 * nrm;
      ref = nrm;
      if (it + 1 >= iters && converged) break;
    }
  }
  if (!(nrm > 0)) return 1.0;
  return converged ? nrm : std::max(nrm, pc_bound);
}

// ---- layouts of the HBM-resident streaming PDLP (dsp_stream.hip), host side ----------------------------------------------------
// entry-major ELL ([W][nvec]: thread v reads entry e at e * nvec + v, coalesced) + CSR segments of the "long" vectors (more than
// W entries), cut into chunks of at most `chunk` entries
struct HostStreamELL {
  int nvec = 0, W = 0;
  std::vector<double> val;               // [W][nvec]
  std::vector<int32_t> idx;              // [W][nvec]
  std::vector<uint8_t> is_long;          // [nvec]
  std::vector<int32_t> long_id, long_ptr, long_idx;
  std::vector<double> long_val;
  std::vector<int32_t> chunk_begin, chunk_end, long_chunk_ptr;
};

// `span_limit` > 0: a vector whose entries span more than that many indices is "long" whatever its length (the wrap-around
// column of a periodic boundary condition: two entries, first and last period - left in the ELL part it would stretch the
// hull of its tile over the whole matrix and cost the LP its banded plan; as a long vector it goes through the partial sums)
inline HostStreamELL build_stream_ell(const HostCSR &M, int max_w, int chunk, int span_limit = 0) {
  HostStreamELL E;
  const int nv = M.m;
  std::vector<int> len(nv);
  for (int v = 0; v < nv; ++v) {
    len[v] = M.ptr[v + 1] - M.ptr[v];
    if (span_limit > 0 && len[v] > 1) {
      int lo = M.idx[M.ptr[v]], hi = lo;
      for (int p = M.ptr[v]; p < M.ptr[v + 1]; ++p) { lo = std::min(lo, (int)M.idx[p]); hi = std::max(hi, (int)M.idx[p]); }
      if (hi - lo > span_limit) len[v] = max_w + 1 + len[v];            // classified long below (its true length is re-read there)
    }
  }
  // the smallest width that leaves long only what is long at max_w anyway (a design column that touches every period stays
  // long at any width: it must not pad every other vector to max_w entries)
  int nl_max = 0;
  for (int v = 0; v < nv; ++v) nl_max += len[v] > max_w;
  int W = 1;
  for (W = 1; W < max_w; ++W) {
    int nl = 0;
    for (int v = 0; v < nv; ++v) nl += len[v] > W;
    if (nl == nl_max) break;
  }
  E.nvec = nv; E.W = W;
  E.val.assign((size_t)W * nv, 0.0);
  E.idx.assign((size_t)W * nv, 0);
  E.is_long.assign(std::max(nv, 1), 0);
  E.long_ptr.push_back(0);
  for (int v = 0; v < nv; ++v) {
    if (len[v] <= W) {
      for (int e = 0; e < len[v]; ++e) { E.val[(size_t)e * nv + v] = M.val[M.ptr[v] + e]; E.idx[(size_t)e * nv + v] = M.idx[M.ptr[v] + e]; }
    } else {
      E.is_long[v] = 1;
      E.long_id.push_back(v);
      for (int p = M.ptr[v]; p < M.ptr[v + 1]; ++p) { E.long_idx.push_back(M.idx[p]); E.long_val.push_back(M.val[p]); }
      E.long_ptr.push_back((int32_t)E.long_idx.size());
    }
  }
  E.long_chunk_ptr.push_back(0);
  for (size_t l = 0; l < E.long_id.size(); ++l) {
    for (int p = E.long_ptr[l]; p < E.long_ptr[l + 1]; p += chunk) { E.chunk_begin.push_back(p); E.chunk_end.push_back(std::min(p + chunk, E.long_ptr[l + 1])); }
    E.long_chunk_ptr.push_back((int32_t)E.chunk_begin.size());
  }
  return E;
}

// Tiles of the fused one-launch iteration (dsp_stream.hpp: FusedPlan): contiguous row / column ranges + the hulls of what
// each tile's two products touch.  tile[8 t ..]: i0, i1 (own rows), j0, j1 (own columns), c_lo, c_hi (columns whose primal step
// the tile computes: its own + those its rows touch), r_lo, r_hi (rows whose y it stages: its own + those these columns touch).
// ridx_enc: the row ELL's column indices with long column l encoded as -1 - l (padding entries: value 0, index of the row's first entry).
// ntile = 0 = not applicable: long rows, more than `max_long` long columns, or hulls beyond `lds_budget` bytes per scenario
// (a matrix that is not banded in the order it was handed over).
struct HostFusedPlan {
  int ntile = 0, rows_per_tile = 0, ny_max = 0, nxb_max = 0;
  int own_max = 0;                     // most rows or columns any tile owns
  int halo_max = 0;                    // most halo rows or halo columns (hull minus own range) of any tile
  std::vector<int32_t> tile, ridx_enc;
};

// Workgroup -> (tile, scenario group) of the fused iteration's launch (grid: G groups x ntile tiles, linear id lin = group + G tile).
// Workgroups go to the 8 XCDs round-robin in launch order; the XCD-major renumbering gives workgroup 8 q + c (XCD c) group q mod G
// of tile 8 (q / G) + c, so that ALL groups of a tile run on ONE XCD, back to back, and the tile's slice of the matrix is fetched
// into one L2 once.  Only whole rounds of 8 tiles are renumbered (xcd_full = 8 G (ntile / 8) workgroups); the last ntile mod 8
// tiles keep the grid order - a renumbering that runs on into tile ids beyond the plan reads outside every array (the memory
// fault of the first attempt).  Shared by the kernel and the CPU test (tests/fused_harness.cpp: a bijection for every G, ntile).
#if defined(__HIPCC__)
#define DSP_HOST_DEVICE __host__ __device__
#else
#define DSP_HOST_DEVICE
#endif
DSP_HOST_DEVICE inline void fused_workgroup(int lin, int G, int xcd_full, int &tile, int &grp) {
  if (lin < xcd_full) {
    const int q = lin >> 3, c = lin & 7;
    tile = 8 * (q / G) + c;
    grp = q % G;
  } else {
    tile = lin / G;
    grp = lin % G;
  }
}
inline int fused_xcd_full(int G, int ntile) { return 8 * G * (ntile / 8); }

inline HostFusedPlan build_fused_plan(const HostCSR &A, const HostCSR &AT, const HostStreamELL &Er, const HostStreamELL &Ec, int max_long,
                                      int rows_per_tile, size_t lds_budget) {
  HostFusedPlan F;
  const int n = A.n, m = A.m;
  if (!Er.long_id.empty() || (int)Ec.long_id.size() > max_long || m < 1 || n < 1) return F;
  const int nlong = (int)Ec.long_id.size();
  const int RB = std::max(64, rows_per_tile);
  const int ntile = (m + RB - 1) / RB;
  std::vector<int32_t> tile((size_t)ntile * 8);
  int ny = 0, nxb = 0;
  for (int t = 0; t < ntile; ++t) {
    // own columns follow the own rows proportionally (a banded matrix keeps column j near row j m / n)
    const int i0 = t * RB, i1 = std::min(m, i0 + RB);
    const int j0 = (int)((int64_t)i0 * n / m), j1 = i1 == m ? n : (int)((int64_t)i1 * n / m);
    int c_lo = j0 < j1 ? j0 : n, c_hi = j0 < j1 ? j1 : 0;
    for (int i = i0; i < i1; ++i)
      for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
        const int j = A.idx[p];
        if (Ec.is_long[j]) continue;
        c_lo = std::min(c_lo, j); c_hi = std::max(c_hi, j + 1);
      }
    if (c_lo >= c_hi) { c_lo = 0; c_hi = 1; }
    int r_lo = i0, r_hi = i1;
    for (int j = c_lo; j < c_hi; ++j) {
      if (Ec.is_long[j]) continue;
      for (int p = AT.ptr[j]; p < AT.ptr[j + 1]; ++p) { r_lo = std::min(r_lo, (int)AT.idx[p]); r_hi = std::max(r_hi, (int)AT.idx[p] + 1); }
    }
    int32_t *tp = &tile[(size_t)t * 8];
    tp[0] = i0; tp[1] = i1; tp[2] = j0; tp[3] = j1; tp[4] = c_lo; tp[5] = c_hi; tp[6] = r_lo; tp[7] = r_hi;
    ny = std::max(ny, r_hi - r_lo); nxb = std::max(nxb, nlong + c_hi - c_lo);
    F.own_max = std::max(F.own_max, std::max(i1 - i0, j1 - j0));
    F.halo_max = std::max(F.halo_max, std::max((r_hi - r_lo) - (i1 - i0), (c_hi - c_lo) - (j1 - j0)));
  }
  if ((size_t)(ny + nxb) * sizeof(double) > lds_budget) return F;
  std::vector<int32_t> rank(n, -1);
  for (int l = 0; l < nlong; ++l) rank[Ec.long_id[l]] = l;
  // Padding entries (value 0) repeat the row's FIRST real entry: a slot the tile has written whatever the hull looks like.  (They
  // used to keep index 0; when column 0 is a long column inside the hull of the first tiles - the nuclear price-taker LP: columns
  // 0 .. 2 are its design variables - that is the hull slot of a column no phase writes, and 0 x stale-LDS-bits can be NaN.)
  F.ridx_enc.assign((size_t)Er.W * m, nlong > 0 ? -1 : 0);
  for (int i = 0; i < m; ++i) {
    const int len = A.ptr[i + 1] - A.ptr[i];
    for (int e = 0; e < Er.W; ++e) {
      if (len == 0) break;                                          // an empty row keeps long slot 0 / column 0 (clamped by the kernel)
      const int j = A.idx[A.ptr[i] + (e < len ? e : 0)];
      F.ridx_enc[(size_t)e * m + i] = rank[j] >= 0 ? -1 - rank[j] : j;
    }
  }
  F.tile = tile; F.ntile = ntile; F.rows_per_tile = RB; F.ny_max = ny; F.nxb_max = nxb;
  return F;
}

// Lane-major ELL of a CSR whose "rows" are the vectors owned by lanes: vector v is owned by lane v % 64,
// slot v / 64.  Entry (e, slot, lane) lives at ((e*slots + slot)*64 + lane): for a fixed e the `slots` entries of a
// lane are independent multiply-add chains (instruction-level parallelism inside the wave).  Vectors with more than
// W entries are moved WHOLE to the long list (their ELL entries stay zero).  idx = element index of the multiplied
// vector; the C ABI turns it into a byte offset when it packs the device entries.
struct LaneELL {
  int slots = 0, W = 0;
  std::vector<double> val;       // [W*slots*64]
  std::vector<uint16_t> idx;     // [W*slots*64]
  // long vectors
  std::vector<int32_t> long_owner;   // vector id
  std::vector<int32_t> long_start;   // offset into tail arrays (multiple of 64)
  std::vector<int32_t> long_len;     // padded length (multiple of 64)
  std::vector<double> tail_val;
  std::vector<uint16_t> tail_idx;
};

inline LaneELL build_lane_ell(const HostCSR &M, int slots) {
  LaneELL E;
  E.slots = slots;
  std::vector<int> len(M.m);
  int maxlen = 0;
  for (int v = 0; v < M.m; ++v) { len[v] = M.ptr[v + 1] - M.ptr[v]; maxlen = std::max(maxlen, len[v]); }
  // cost model: per-iteration lane work = W*slots (ELL) + sum over long vectors (ceil(len/64) + reduction ~8)
  int bestW = std::max(maxlen, 1); double bestCost = 1e300;
  for (int W = 1; W <= std::max(maxlen, 1); ++W) {
    double cost = (double)W * slots;
    int nlong = 0;
    for (int v = 0; v < M.m; ++v) if (len[v] > W) { cost += (len[v] + 63) / 64 + 8; ++nlong; }
    if (nlong > 32) continue;
    if (cost < bestCost) { bestCost = cost; bestW = W; }
  }
  E.W = bestW;
  E.val.assign((size_t)slots * E.W * 64, 0.0);
  E.idx.assign((size_t)slots * E.W * 64, 0);
  for (int v = 0; v < M.m; ++v) {
    int lane = v & 63, slot = v >> 6;
    if (len[v] <= E.W) {
      for (int e = 0; e < len[v]; ++e) {
        size_t at = ((size_t)(e * slots + slot)) * 64 + lane;
        E.val[at] = M.val[M.ptr[v] + e];
        E.idx[at] = (uint16_t)M.idx[M.ptr[v] + e];
      }
    } else {
      int padded = ((len[v] + 63) / 64) * 64;
      E.long_owner.push_back(v);
      E.long_start.push_back((int32_t)E.tail_val.size());
      E.long_len.push_back(padded);
      for (int e = 0; e < padded; ++e) {
        E.tail_val.push_back(e < len[v] ? M.val[M.ptr[v] + e] : 0.0);
        E.tail_idx.push_back(e < len[v] ? (uint16_t)M.idx[M.ptr[v] + e] : (uint16_t)0);
      }
    }
  }
  return E;
}

// ---- register-resident-matrix layout: ownership sorted by vector length, per-slot ELL widths -------------------------
// A lane-slot (q, lane) = POSITION p = q*64 + lane owns the vector at[p].  Vectors are placed by decreasing length
// (long vectors last), so a slot's width is the longest vector IN THAT SLOT: wind+battery 24 h needs 3+3+1+1 = 8
// column entries per lane instead of 4 x 3 = 12.  Everything the kernel exchanges through LDS lives in position space.
struct SortedLayout {
  int slots = 0;
  std::vector<int32_t> at;     // [slots*64] position -> vector id, -1 = padding
  std::vector<int32_t> pos;    // vector id -> position
};

inline SortedLayout sorted_layout(const HostCSR &M, int slots, const std::vector<int32_t> &long_ids) {
  SortedLayout L;
  L.slots = slots;
  std::vector<char> is_long(M.m, 0);
  for (int32_t v : long_ids) is_long[v] = 1;
  std::vector<int32_t> order(M.m);
  std::iota(order.begin(), order.end(), 0);
  auto key = [&](int32_t v) { return is_long[v] ? -1 : (int)(M.ptr[v + 1] - M.ptr[v]); };
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key(a) > key(b); });
  L.at.assign((size_t)slots * 64, -1);
  L.pos.assign(M.m, -1);
  for (int p = 0; p < M.m; ++p) { L.at[p] = order[p]; L.pos[order[p]] = p; }
  return L;
}

struct SlotELL {
  int slots = 0;
  int width[16] = {0};
  uint32_t pack = 0;             // 4 bits per slot
  int total = 0;                 // sum of widths
  std::vector<double> val;       // [(base(q)+e)*64 + lane]
  std::vector<uint32_t> off;     // byte offset (position * 8) of the gathered element
  std::vector<int32_t> long_owner_pos, long_start, long_len;
  std::vector<double> tail_val;
  std::vector<uint32_t> tail_off;
};

// M: CSR whose rows are the vectors owned through `own`; the gathered vector's elements sit at `other.pos`.
// force_width > 0: every slot gets that width (zero-padded entries) - the layout of the PADDED register-resident
// specialisations that serve LP shapes without a tight one; requires every non-long vector to have <= force_width entries
inline SlotELL build_slot_ell(const HostCSR &M, const SortedLayout &own, const SortedLayout &other,
                              const std::vector<int32_t> &long_ids, int force_width = 0) {
  SlotELL E;
  E.slots = own.slots;
  std::vector<char> is_long(M.m, 0);
  for (int32_t v : long_ids) is_long[v] = 1;
  for (int q = 0; q < own.slots; ++q) {
    int w = 0;
    for (int l = 0; l < 64; ++l) {
      int32_t v = own.at[q * 64 + l];
      if (v >= 0 && !is_long[v]) w = std::max(w, (int)(M.ptr[v + 1] - M.ptr[v]));
    }
    if (force_width > 0) w = force_width;
    E.width[q] = w;
    E.pack |= (uint32_t)(w & 15) << (4 * q);
    E.total += w;
  }
  E.val.assign((size_t)std::max(E.total, 1) * 64, 0.0);
  E.off.assign((size_t)std::max(E.total, 1) * 64, 0u);
  int base = 0;
  for (int q = 0; q < own.slots; ++q) {
    for (int l = 0; l < 64; ++l) {
      int32_t v = own.at[q * 64 + l];
      if (v < 0 || is_long[v]) continue;
      for (int e = 0; e < M.ptr[v + 1] - M.ptr[v]; ++e) {
        size_t at = (size_t)(base + e) * 64 + l;
        E.val[at] = M.val[M.ptr[v] + e];
        E.off[at] = (uint32_t)other.pos[M.idx[M.ptr[v] + e]] * 8u;
      }
    }
    base += E.width[q];
  }
  for (int32_t v : long_ids) {
    int len = M.ptr[v + 1] - M.ptr[v], padded = ((len + 63) / 64) * 64;
    E.long_owner_pos.push_back(own.pos[v]);
    E.long_start.push_back((int32_t)E.tail_val.size());
    E.long_len.push_back(padded);
    for (int e = 0; e < padded; ++e) {
      E.tail_val.push_back(e < len ? M.val[M.ptr[v] + e] : 0.0);
      E.tail_off.push_back(e < len ? (uint32_t)other.pos[M.idx[M.ptr[v] + e]] * 8u : 0u);
    }
  }
  return E;
}

// ---- LDS bank conflicts of the gathers: slot permutation of the exchange buffers -----------------------------------
// Position p of an exchange buffer (the element owned by lane p % 64, slot p / 64) is stored at LDS slot slot[p].
// Slots are permuted only WITHIN each 32-slot block (one 256-byte LDS bank row of 8-byte elements) and so that the 16
// consecutive owner lanes a ds_write_b64 services per LDS cycle keep 16 slots that are distinct mod 16 (32 four-byte
// banks = 16 eight-byte pairs on the store side): the owner stores stay conflict-free whatever the permutation (an
// unrestricted within-block search lowered the simulated gather conflicts further but doubled the MEASURED
// SQ_LDS_BANK_CONFLICT through 2-way store conflicts).  The gathers see bank pair slot[p] % 32 (64 banks on the
// ds_read_b64 side).  Start: every block rotated by r * block (best of 32 rotations); then a deterministic local
// search applies store-safe swaps (two positions of one 16-lane group, or two slots congruent mod 16) whenever that
// does not increase the simulated gather conflicts.
inline uint32_t rotation_slot(uint32_t p, uint32_t r) {
  const uint32_t blk = p >> 5;
  return (blk << 5) | ((p + r * blk) & 31u);
}

// Extra LDS cycles of all gathers of E: a ds_read_b64 is serviced in two 32-lane groups; within a group, distinct
// addresses falling into the same (slot mod 32) bank pair cost one extra cycle each (equal addresses broadcast).
// E.off holds POSITION * 8 (un-swizzled).  Padding entries (value 0) are skipped: apply_slots points them at an
// address their group reads anyway.
inline int gather_conflicts(const SlotELL &E, const std::vector<int32_t> &slot) {
  int extra = 0;
  for (int t = 0; t < E.total; ++t)
    for (int g = 0; g < 2; ++g) {
      int count[32] = {0};
      int32_t seen[32];
      int ns = 0;
      for (int l = 32 * g; l < 32 * g + 32; ++l) {
        size_t at = (size_t)t * 64 + l;
        if (E.val[at] == 0.0) continue;
        int32_t sl = slot[E.off[at] / 8u];
        bool dup = false;
        for (int k = 0; k < ns; ++k) dup |= seen[k] == sl;
        if (dup) continue;
        seen[ns++] = sl;
        count[sl & 31]++;
      }
      int worst = 1;
      for (int b = 0; b < 32; ++b) worst = std::max(worst, count[b]);
      extra += worst - 1;
    }
  return extra;
}

struct SlotMap {
  std::vector<int32_t> slot;      // [npad] position -> LDS slot
  int cost_identity = 0, cost_rotation = 0, cost_final = 0;
};

// shared = true: every 64-position block gets the SAME map, slot(p + 64 q) = slot(p) + 64 q.  A lane's store addresses in an
// exchange buffer are then ONE register + compile-time offsets 512 q (and pair up into ds_write2_b64) instead of one address
// register per owned element - for the shapes with 11 owned elements per lane (48-h wind+battery) those registers were
// the difference between a spill-free hot loop and three scratch reloads per iteration.
inline SlotMap optimise_slots(const SlotELL &E, int npad, int sweeps = 4000, bool shared = false) {
  SlotMap M;
  M.slot.resize((size_t)npad);
  for (int p = 0; p < npad; ++p) M.slot[p] = p;
  M.cost_identity = gather_conflicts(E, M.slot);
  std::vector<int32_t> trial((size_t)npad);
  int best = M.cost_identity;
  for (uint32_t r = 1; r < 32; ++r) {
    for (int p = 0; p < npad; ++p)
      trial[p] = shared ? (int32_t)(((uint32_t)p & ~31u) | (((uint32_t)p + r * (((uint32_t)p >> 5) & 1u)) & 31u))
                        : (int32_t)rotation_slot((uint32_t)p, r);
    int c = gather_conflicts(E, trial);
    if (c < best) { best = c; M.slot = trial; }
  }
  M.cost_rotation = best;
  uint64_t rng = 0x9E3779B97F4A7C15ull;                       // fixed seed: the layout is a function of the matrix only
  auto next = [&rng]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); };
  const int blocks = npad / 32;
  for (int it = 0; it < sweeps && best > 0; ++it) {
    const int b = (int)(next() % (uint32_t)(shared ? 2 : blocks)), i = (int)(next() & 31u), j = (int)(next() & 31u);
    if (i == j) continue;
    // store-safe moves only: same 16-lane group, or slots congruent mod 16
    if ((i >> 4) != (j >> 4) && ((M.slot[b * 32 + i] ^ M.slot[b * 32 + j]) & 15) != 0) continue;
    auto swap_all = [&]() {
      if (!shared) { std::swap(M.slot[b * 32 + i], M.slot[b * 32 + j]); return; }
      for (int bb = b; bb < blocks; bb += 2) {                  // the same swap in every 64-block: only the low 5 bits move
        const int32_t si = M.slot[bb * 32 + i], sj = M.slot[bb * 32 + j];
        M.slot[bb * 32 + i] = (si & ~31) | (sj & 31);
        M.slot[bb * 32 + j] = (sj & ~31) | (si & 31);
      }
    };
    swap_all();
    int c = gather_conflicts(E, M.slot);
    if (c <= best) best = c;
    else swap_all();
  }
  M.cost_final = best;
  return M;
}

// rewrite the gather offsets of E (position * 8 -> slot * 8); padding entries read an address of their own 32-lane
// group (a broadcast), or slot 0 if the group has no real entry
inline void apply_slots(SlotELL &E, const std::vector<int32_t> &slot) {
  for (int t = 0; t < E.total; ++t)
    for (int g = 0; g < 2; ++g) {
      uint32_t any = 0;
      for (int l = 32 * g; l < 32 * g + 32; ++l) {
        size_t at = (size_t)t * 64 + l;
        if (E.val[at] != 0.0) { E.off[at] = (uint32_t)slot[E.off[at] / 8u] * 8u; any = E.off[at]; }
      }
      for (int l = 32 * g; l < 32 * g + 32; ++l) {
        size_t at = (size_t)t * 64 + l;
        if (E.val[at] == 0.0) E.off[at] = any;
      }
    }
  for (size_t k = 0; k < E.tail_off.size(); ++k) E.tail_off[k] = (uint32_t)slot[E.tail_off[k] / 8u] * 8u;
}

}  // namespace dsp
