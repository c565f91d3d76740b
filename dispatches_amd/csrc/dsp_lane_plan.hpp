// dsp_lane_plan.hpp — host-side plan of the LANE-PER-SCENARIO streaming iteration (dsp_stream_lane.hip, dsp_lane_tile.hpp).
//
// Round 4.  The one-launch fused iteration of round 3 (k_fused_pre) gave a workgroup a tile of 250 rows x 2 scenarios: lanes ran
// along the ROWS, both products gathered through index arithmetic, two workgroup barriers and a block reduction per launch - 476
// VALU instructions per wave against ~75 of FP64 arithmetic, a wave lifetime of 10 us for 7 KB of traffic, 0.37 of the HBM peak
// beyond the Infinity Cache.  Here lanes run along the SCENARIOS: the batch is stored scenario-minor ([group][element][64]), lane
// s of a wave owns scenario s of its group and walks through a tile of consecutive rows and columns ALONE.  Consequences:
//   * the matrix entries, the shared bounds and every index are WAVE-UNIFORM (scalar loads, scalar address arithmetic);
//   * both products gather from a lane-private ring buffer in LDS at uniform slots (conflict-free, no index VALU beyond one add);
//   * nothing ever crosses lanes: no barrier inside the tile, no wave reduction - a long column's partial sum of A^T y is one
//     register per lane; the waves of a workgroup meet once, at the end, to add their partial sums;
//   * every global access is a 512-byte row of one vector: x, x0, c, y, y0 read once, x, y written once (4 n + 3 m doubles per
//     scenario-iteration with shared bounds, as before), the next unit's rows requested while the current one is computed.
// The plan: long columns (design variables that touch every period; <= 8) are taken out of the row records and kept as dense
// per-row coefficients; what is left must be BANDED in the order it was handed over - a tile's walk keeps only a window of y and
// of xbar in its rings, and the planner below schedules the walk (units of <= CH staged rows, <= CH primal columns, <= CH dual
// rows) so that everything a product reads is in the window and nothing live is overwritten.  No schedule within the ring
// budget = plan not applicable = the caller keeps the two-launch form (dsp_stream.hip).
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <vector>

#include "dsp_prepare.hpp"

namespace dsp {

constexpr int kLaneMaxLong = 8;          // long columns the lane form carries
constexpr int kLaneTileInts = 16;        // ints per entry of the tile table (dsp_lane_tile.hpp reads an entry as ONE 64-byte scalar load)
// largest ring (slots per window) the planner tries: the check kernel keeps THREE windows per wave, four waves per workgroup -
// 3 x 16 x 512 B x 4 + the record stages = 104 KB of a CU's 160 KB of LDS; 32 slots (200 KB) would not launch.  A band that needs
// more is not scheduled: the handle keeps the other forms.
constexpr int kLaneMaxRing = 16;

struct HostLanePlan {
  bool ok = false;
  int n = 0, m = 0;
  int WC = 4, WR = 4;                    // record widths (4 or 8): most short entries of a column / of a row
  int nl = 0, NLP = 4;                   // long columns, padded count (4 or 8)
  std::vector<int32_t> long_id;          // [nl] column index
  std::vector<int32_t> long_rank;        // [n]  -1 or position in long_id
  std::vector<double> cval, rval, ral;   // [n][WC], [m][WR], [m][NLP]
  std::vector<int32_t> cidx, ridx;       // [n][WC] row indices (bit 30 of cidx[j][0]: long column), [m][WR] column indices
  std::vector<int32_t> rmin, rmax;       // [n] hull of the rows a short column touches (INT_MAX / -1: none)
  std::vector<int32_t> cmin, cmax;       // [m] hull of the short columns a row touches
};

// A: scaled constraint matrix by rows, AT: its transpose.  span_limit: a column whose entries span more rows is long whatever its
// length (the wrap-around column of a periodic boundary condition).
inline HostLanePlan build_lane_plan(const HostCSR &A, const HostCSR &AT, int span_limit) {
  HostLanePlan P;
  const int n = A.n, m = A.m;
  P.n = n; P.m = m;
  if (n < 1 || m < 1 || std::max(n, m) >= (1 << 22)) return P;         // (record indices are stored << 9, sign bit = flag)
  P.long_rank.assign(n, -1);
  for (int j = 0; j < n; ++j) {
    const int len = AT.ptr[j + 1] - AT.ptr[j];
    bool lng = len > 8;
    if (!lng && len > 1 && span_limit > 0) {
      int lo = INT_MAX, hi = -1;
      for (int p = AT.ptr[j]; p < AT.ptr[j + 1]; ++p) { lo = std::min(lo, (int)AT.idx[p]); hi = std::max(hi, (int)AT.idx[p]); }
      lng = hi - lo > span_limit;
    }
    if (lng) { P.long_rank[j] = (int)P.long_id.size(); P.long_id.push_back(j); }
  }
  P.nl = (int)P.long_id.size();
  if (P.nl > kLaneMaxLong) return P;
  P.NLP = P.nl <= 4 ? 4 : 8;
  int wc = 1, wr = 1;
  for (int j = 0; j < n; ++j) if (P.long_rank[j] < 0) wc = std::max(wc, (int)(AT.ptr[j + 1] - AT.ptr[j]));
  for (int i = 0; i < m; ++i) {
    int s = 0;
    for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) s += P.long_rank[A.idx[p]] < 0;
    wr = std::max(wr, s);
  }
  if (wc > 8 || wr > 8) return P;                       // a long ROW (or a dense short part): not this form
  P.WC = wc <= 4 ? 4 : 8; P.WR = wr <= 4 ? 4 : 8;
  if (P.WC == 8) P.WR = 8;                              // the kernels exist for (4, 4), (4, 8), (8, 8): wide columns take the wide row record too
  P.cval.assign((size_t)n * P.WC, 0.0); P.cidx.assign((size_t)n * P.WC, 0);
  P.rval.assign((size_t)m * P.WR, 0.0); P.ridx.assign((size_t)m * P.WR, 0);
  P.ral.assign((size_t)m * P.NLP, 0.0);
  P.rmin.assign(n, INT_MAX); P.rmax.assign(n, -1); P.cmin.assign(m, INT_MAX); P.cmax.assign(m, -1);
  for (int j = 0; j < n; ++j) {
    int32_t *ix = &P.cidx[(size_t)j * P.WC];
    double *vv = &P.cval[(size_t)j * P.WC];
    if (P.long_rank[j] >= 0) {                           // long column: bit 30 of the first index (ring slots mask it away), all zeros
      for (int e = 0; e < P.WC; ++e) ix[e] = std::min(m - 1, (int)((int64_t)j * m / n));
      ix[0] |= 1 << 30;
      continue;
    }
    const int len = AT.ptr[j + 1] - AT.ptr[j];
    for (int e = 0; e < P.WC; ++e) {
      // padding entries: value 0 at the column's FIRST row - a slot of the window that is live whenever the column is computed
      const int p = AT.ptr[j] + (e < len ? e : 0);
      ix[e] = len ? AT.idx[p] : std::min(m - 1, (int)((int64_t)j * m / n));
      vv[e] = e < len ? AT.val[p] : 0.0;
      if (e < len) { P.rmin[j] = std::min(P.rmin[j], ix[e]); P.rmax[j] = std::max(P.rmax[j], ix[e]); }
    }
  }
  for (int i = 0; i < m; ++i) {
    int32_t *ix = &P.ridx[(size_t)i * P.WR];
    double *vv = &P.rval[(size_t)i * P.WR];
    int e = 0, first = -1;
    for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
      const int j = A.idx[p];
      if (P.long_rank[j] >= 0) { P.ral[(size_t)i * P.NLP + P.long_rank[j]] += A.val[p]; continue; }
      if (first < 0) first = j;
      ix[e] = j; vv[e] = A.val[p]; ++e;
      P.cmin[i] = std::min(P.cmin[i], j); P.cmax[i] = std::max(P.cmax[i], j);
    }
    for (; e < P.WR; ++e) { ix[e] = first >= 0 ? first : std::min(n - 1, (int)((int64_t)i * n / m)); vv[e] = 0.0; }
  }
  P.ok = true;
  return P;
}

// The records as the kernels read them (dsp_lane_tile.hpp: layout, lane_crec / lane_rrec) for a ring of `ring` slots - an index is
// stored as the byte offset of its ring slot, (index & (ring - 1)) << 9: one add per gather on the device -, bounds infinite and
// scale factors 1 until someone writes them (device: k_lane_fill_records per solve; tests: set_lane_record_bounds), 1 KiB of slack behind each
// array (a unit's records are fetched as one 16-byte-per-lane block that may run past the last record).
inline void pack_lane_records(const HostLanePlan &P, int ring, std::vector<char> &crec, std::vector<char> &rrec) {
  const uint32_t M = (uint32_t)ring - 1u;
  const int CREC = P.WC * 12 + 32, RREC = P.WR * 12 + 32 + P.NLP * 8;
  crec.assign((size_t)P.n * CREC + 1024, 0); rrec.assign((size_t)P.m * RREC + 1024, 0);
  const double inf = INFINITY, one = 1.0;
  for (int j = 0; j < P.n; ++j) {
    char *r = &crec[(size_t)j * CREC];
    std::memcpy(r, &P.cval[(size_t)j * P.WC], (size_t)P.WC * 8);
    for (int e = 0; e < P.WC; ++e) {                       // ring slot << 9, long flag (bit 30 of the plan's index) -> sign bit
      const int32_t v = P.cidx[(size_t)j * P.WC + e];
      const uint32_t enc = (((uint32_t)(v & 0x3fffffff) & M) << 9) | ((v >> 30) & 1 ? 0x80000000u : 0u);
      std::memcpy(r + P.WC * 8 + e * 4, &enc, 4);
    }
    const double lo = -inf;
    std::memcpy(r + P.WC * 12, &lo, 8); std::memcpy(r + P.WC * 12 + 8, &inf, 8); std::memcpy(r + P.WC * 12 + 16, &one, 8);
  }
  for (int i = 0; i < P.m; ++i) {
    char *r = &rrec[(size_t)i * RREC];
    std::memcpy(r, &P.rval[(size_t)i * P.WR], (size_t)P.WR * 8);
    for (int e = 0; e < P.WR; ++e) { const uint32_t enc = ((uint32_t)P.ridx[(size_t)i * P.WR + e] & M) << 9; std::memcpy(r + P.WR * 8 + e * 4, &enc, 4); }
    const double lo = -inf;
    std::memcpy(r + P.WR * 12, &lo, 8); std::memcpy(r + P.WR * 12 + 8, &inf, 8);
    std::memcpy(r + P.WR * 12 + 16, &P.ral[(size_t)i * P.NLP], (size_t)P.NLP * 8);
    std::memcpy(r + P.WR * 12 + 16 + P.NLP * 8, &one, 8);
  }
}
inline void set_lane_record_bounds(const HostLanePlan &P, std::vector<char> &crec, std::vector<char> &rrec, const double *lb, const double *ub,
                                   const double *rlo, const double *rhi, const double *col_scale, const double *row_scale) {
  const int CREC = P.WC * 12 + 32, RREC = P.WR * 12 + 32 + P.NLP * 8;
  for (int j = 0; j < P.n; ++j) {
    char *r = &crec[(size_t)j * CREC + P.WC * 12];
    std::memcpy(r, &lb[j], 8); std::memcpy(r + 8, &ub[j], 8); std::memcpy(r + 16, &col_scale[j], 8);
  }
  for (int i = 0; i < P.m; ++i) {
    char *r = &rrec[(size_t)i * RREC + P.WR * 12];
    std::memcpy(r, &rlo[i], 8); std::memcpy(r + 8, &rhi[i], 8); std::memcpy(r + 16 + P.NLP * 8, &row_scale[i], 8);
  }
}

// Tiles + their walks.  tiles[16 t ..]: i0, i1 (own rows), j0, j1 (own columns), first unit, one past the last unit, c_lo, r_lo, then
// the first unit's six numbers (a copy) and two spare ints.
// units[8 u ..]: ys0, nys (rows whose y is staged into the ring), cx0, ncx (columns whose primal step is computed: own ones are
// written, halo ones only feed the window), rd0, nrd (own rows whose dual step is computed); counts <= CH; a start with count 0
// is still a valid index (the kernel's loads are unconditional, to clamped addresses).
struct HostLaneTiles {
  bool ok = false;
  int ntile = 0, rows_per_tile = 0, CH = 0, ring = 0, nunit = 0;
  int max_units = 0;                     // most units of any tile
  double halo_rows = 0.0;                // staged + computed halo elements / own elements (traffic overhead estimate)
  std::vector<int32_t> tiles, units;
};

inline HostLaneTiles build_lane_tiles_ring(const HostLanePlan &P, int rows_per_tile, int CH, int R) {
  HostLaneTiles T;
  const int n = P.n, m = P.m;
  const int RB = std::max(1, rows_per_tile);
  const int ntile = (m + RB - 1) / RB;
  std::vector<int32_t> tiles((size_t)ntile * kLaneTileInts, 0), units;
  int max_units = 0;
  int64_t halo = 0;
  for (int t = 0; t < ntile; ++t) {
    const int i0 = t * RB, i1 = std::min(m, i0 + RB);
    const int j0 = (int)((int64_t)i0 * n / m), j1 = i1 == m ? n : (int)((int64_t)i1 * n / m);
    int c_lo = j0, c_hi = j1;
    for (int i = i0; i < i1; ++i) if (P.cmax[i] >= 0) { c_lo = std::min(c_lo, (int)P.cmin[i]); c_hi = std::max(c_hi, (int)P.cmax[i] + 1); }
    int r_lo = i0, r_hi = i1;
    for (int j = c_lo; j < c_hi; ++j) if (P.rmax[j] >= 0) { r_lo = std::min(r_lo, (int)P.rmin[j]); r_hi = std::max(r_hi, (int)P.rmax[j] + 1); }
    // suffix minima: lowest row a not-yet-computed column reads, lowest column a not-yet-done own row reads
    std::vector<int> sufr(c_hi - c_lo + 1, INT_MAX), sufc(i1 - i0 + 1, INT_MAX);
    for (int j = c_hi - 1; j >= c_lo; --j) sufr[j - c_lo] = std::min(sufr[j - c_lo + 1], (int)P.rmin[j]);
    for (int i = i1 - 1; i >= i0; --i) sufc[i - i0] = std::min(sufc[i - i0 + 1], (int)P.cmin[i]);
    int ry = r_lo, cx = c_lo, rd = i0;
    const int ubeg = (int)(units.size() / 8);
    while (rd < i1 || cx < c_hi) {
      // the window of y holds [lowY, ry): rows the columns still to come read + the own rows whose dual step is still to come;
      // the window of xbar holds [lowX, cx).  Within a unit the kernel stages, then computes columns, then rows: the bounds are
      // taken BEFORE the unit's columns / rows are counted off, i.e. on the safe side.
      const int lowY = std::min(rd < i1 ? rd : INT_MAX, sufr[cx - c_lo]);
      const int lowX = sufc[rd - i0];
      int32_t u[8] = {std::min(ry, m - 1), 0, std::min(cx, n - 1), 0, std::min(rd, m - 1), 0, 0, 0};
      while (ry < r_hi && u[1] < CH && ry - std::min(lowY, ry) < R) { ++ry; ++u[1]; }
      while (cx < c_hi && u[3] < CH && (P.long_rank[cx] >= 0 || P.rmax[cx] < ry) && cx - std::min(lowX, cx) < R) { ++cx; ++u[3]; }
      while (rd < i1 && u[5] < CH && P.cmax[rd] < cx && rd < ry) { ++rd; ++u[5]; }
      if (u[1] + u[3] + u[5] == 0) return T;                 // no progress: this ring cannot hold the band
      units.insert(units.end(), u, u + 8);
    }
    const int uend = (int)(units.size() / 8);
    max_units = std::max(max_units, uend - ubeg);
    halo += (r_hi - r_lo) - (i1 - i0) + 2 * ((c_hi - c_lo) - (j1 - j0));
    int32_t *tp = &tiles[(size_t)t * kLaneTileInts];
    tp[0] = i0; tp[1] = i1; tp[2] = j0; tp[3] = j1; tp[4] = ubeg; tp[5] = uend; tp[6] = c_lo; tp[7] = r_lo;
    // the walk's first unit rides along: the wave has it with the tile's entry instead of after a second, dependent scalar load
    for (int q = 0; q < 6; ++q) tp[8 + q] = uend > ubeg ? units[(size_t)ubeg * 8 + q] : 0;
  }
  T.ok = true; T.ntile = ntile; T.rows_per_tile = RB; T.CH = CH; T.ring = R; T.nunit = (int)(units.size() / 8);
  T.max_units = max_units; T.halo_rows = (double)halo / (7.0 * std::max(1, m));
  T.tiles.swap(tiles); T.units.swap(units);
  return T;
}

// The smallest ring (ring_min .. kLaneMaxRing slots per window; LDS per wave = 2 or 3 windows x ring x 512 bytes) that holds the
// band.  A ring that only just holds it leaves the three streams of a walk (stage, primal, dual) waiting for each other: 17 - 40 %
// more, emptier units on the year-long LPs at 8 slots than at 16 (profiles/r40x_ .. r41a_lane_variants.log; the device asks for 16).
// `slack` > 1 prefers the smallest ring within that factor of the fewest units any ring needs instead.
inline HostLaneTiles build_lane_tiles(const HostLanePlan &P, int rows_per_tile, int CH, int ring_min = 8, int ring_max = kLaneMaxRing, double slack = 0.0) {
  HostLaneTiles best;
  if (!P.ok) return best;
  std::vector<HostLaneTiles> cand;
  for (int R = std::max(8, ring_min); R <= std::min(ring_max, kLaneMaxRing); R *= 2) {
    HostLaneTiles T = build_lane_tiles_ring(P, rows_per_tile, CH, R);
    if (T.ok) cand.push_back(std::move(T));
  }
  if (cand.empty()) return best;
  int fewest = INT_MAX;
  for (const HostLaneTiles &T : cand) fewest = std::min(fewest, T.nunit);
  if (slack <= 1.0) return std::move(cand.front());
  for (HostLaneTiles &T : cand) if ((double)T.nunit <= slack * fewest) return std::move(T);
  return std::move(cand.back());
}

}  // namespace dsp
