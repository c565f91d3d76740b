// CPU run of the lane-per-scenario streaming iteration (dispatches_amd/csrc/dsp_lane_plan.hpp, dsp_lane_tile.hpp, dsp_stream_lane.hip),
// built by tests/test_prepare_cpu.py with g++.  Reads a CSR, builds the plan and a tiling with the library's own host code, and runs
// the SAME per-lane tile routine the kernel runs (LaneTile<...>::run, lane after lane, rings on plain arrays poisoned with NaN
// outside what the routine itself zeroes) for the three modes:
//   0  plain iteration           against the plain PDHG + Halpern step on the CSR, per scenario (each lane its own tau, sig, k)
//   1  check iteration           x+, y+ and the residual / row sums against their definitions (k_check_rows of dsp_stream.hip)
//   2  reduced costs at (x+, y+) the column sums (k_kkt_cols)
// plus the long columns' partial sums of A^T y and what the planner promises (every row and column written exactly once, sink rows
// aside).  Prints one JSON object.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../dispatches_amd/csrc/dsp_lane_plan.hpp"
#include "../dispatches_amd/csrc/dsp_lane_tile.hpp"

using namespace dsp;

static double clampd(double v, double lo, double hi) { return std::fmin(std::fmax(v, lo), hi); }
static double fin0(double v) { return std::fabs(v) < INFINITY ? v : 0.0; }

struct Problem {
  HostCSR A, AT;
  HostLanePlan H;
  HostLaneTiles T;
  int S;                                   // scenarios (lanes in use)
  std::vector<double> x, x0, c, y, y0, lb, ub, rlo, rhi, col_scale, row_scale;      // lane layout [len + 1][64] / per element
  std::vector<double> tau, sig, oml;
  std::vector<int> done;
};

template <int WC, int WR, int NLP>
static int run_all(Problem &Q) {
  const HostLanePlan &H = Q.H;
  const HostLaneTiles &T = Q.T;
  const int n = H.n, m = H.m, nl = H.nl, S = Q.S, R = T.ring;
  LaneProblem P{};
  P.n = n; P.m = m; P.nl = nl;
  std::vector<char> crec, rrec;
  pack_lane_records(H, R, crec, rrec);
  set_lane_record_bounds(H, crec, rrec, Q.lb.data(), Q.ub.data(), Q.rlo.data(), Q.rhi.data(), Q.col_scale.data(), Q.row_scale.data());
  P.crec = crec.data(); P.rrec = rrec.data();
  P.tiles = T.tiles.data(); P.units = T.units.data(); P.ntile = T.ntile; P.ring_mask = R - 1;
  const size_t NC = (size_t)(n + 1) * 64, NR = (size_t)(m + 1) * 64;
  auto X = [&](const std::vector<double> &v, int e, int s) { return v[(size_t)e * 64 + s]; };

  // ---- reference, per scenario ----------------------------------------------------------------------------------------------------
  std::vector<double> xp_ref(NC, 0.0), xbar(NC, 0.0), xn_ref(NC, 0.0), yp_ref(NR, 0.0), yn_ref(NR, 0.0);
  std::vector<double> vref((size_t)13 * 64, 0.0);
  for (int s = 0; s < S; ++s) {
    for (int j = 0; j < n; ++j) {
      double aty = 0.0;
      for (int p = Q.AT.ptr[j]; p < Q.AT.ptr[j + 1]; ++p) aty += Q.AT.val[p] * X(Q.y, Q.AT.idx[p], s);
      const double x = X(Q.x, j, s);
      const double xp = clampd(std::fma(-Q.tau[s], X(Q.c, j, s) - aty, x), Q.lb[j], Q.ub[j]);
      xp_ref[(size_t)j * 64 + s] = xp;
      xbar[(size_t)j * 64 + s] = 2.0 * xp - x;
      xn_ref[(size_t)j * 64 + s] = std::fma(Q.oml[s], X(Q.x0, j, s) - xbar[(size_t)j * 64 + s], xbar[(size_t)j * 64 + s]);
      const double dx = xp - x, d0 = xp - X(Q.x0, j, s);
      vref[0 * 64 + s] += dx * dx; vref[6 * 64 + s] += d0 * d0;
    }
    for (int i = 0; i < m; ++i) {
      double ax = 0.0, axp = 0.0;
      for (int p = Q.A.ptr[i]; p < Q.A.ptr[i + 1]; ++p) { ax += Q.A.val[p] * xbar[(size_t)Q.A.idx[p] * 64 + s]; axp += Q.A.val[p] * xp_ref[(size_t)Q.A.idx[p] * 64 + s]; }
      const double y = X(Q.y, i, s);
      const double gy = std::fma(-Q.sig[s], ax, y);
      const double yp = gy - clampd(gy, -Q.sig[s] * Q.rhi[i], -Q.sig[s] * Q.rlo[i]);
      yp_ref[(size_t)i * 64 + s] = yp;
      const double tt = 2.0 * yp - y;
      yn_ref[(size_t)i * 64 + s] = std::fma(Q.oml[s], X(Q.y0, i, s) - tt, tt);
      const double dy = yp - y, nsadx = -Q.sig[s] * (ax - axp);
      vref[1 * 64 + s] += dy * std::fma(2.0, nsadx, dy);
      const double viol_s = std::fmax(Q.rlo[i] - axp, 0.0) + std::fmax(axp - Q.rhi[i], 0.0);
      vref[4 * 64 + s] += std::fmax(yp, 0.0) * fin0(Q.rlo[i]) - std::fmax(-yp, 0.0) * fin0(Q.rhi[i]);
      const double viol = viol_s / Q.row_scale[i];
      vref[2 * 64 + s] += viol * viol; vref[3 * 64 + s] += std::fabs(yp) * viol_s;
      const double d0 = yp - X(Q.y0, i, s);
      vref[5 * 64 + s] += d0 * d0;
    }
    for (int j = 0; j < n; ++j) {
      double aty = 0.0;
      for (int p = Q.AT.ptr[j]; p < Q.AT.ptr[j + 1]; ++p) aty += Q.AT.val[p] * yp_ref[(size_t)Q.AT.idx[p] * 64 + s];
      const double cj = X(Q.c, j, s), xp = xp_ref[(size_t)j * 64 + s], rc = cj - aty;
      const double lp = std::fabs(Q.lb[j]) < INFINITY ? std::fmax(rc, 0.0) : 0.0, lm = std::fabs(Q.ub[j]) < INFINITY ? std::fmax(-rc, 0.0) : 0.0;
      const double dr = (rc - lp + lm) / Q.col_scale[j];
      vref[8 * 64 + s] += dr * dr; vref[9 * 64 + s] += cj * xp; vref[10 * 64 + s] += lp * fin0(Q.lb[j]) - lm * fin0(Q.ub[j]);
      vref[11 * 64 + s] += std::fabs(cj * xp); vref[12 * 64 + s] += std::fabs(rc - lp + lm) * std::fabs(xp);
    }
  }

  // ---- the long columns' step (k_lane_long): A^T y from the tiles' partial sums of the CURRENT y (k_lane_apply computes those) -------
  std::vector<double> xbl((size_t)NLP * 64, 0.0), xpl((size_t)NLP * 64, 0.0);
  std::vector<double> x_out(NC, NAN), y_out(NR, NAN), xp_out(NC, NAN), yp_out(NR, NAN);
  std::vector<double> vsum((size_t)13 * 64, 0.0);
  for (int l = 0; l < nl; ++l) {
    const int j = H.long_id[l];
    for (int s = 0; s < S; ++s) {
      double aty = 0.0;                                    // sum over the tiles of their rows' terms, tile by tile
      for (int t = 0; t < T.ntile; ++t) {
        double part = 0.0;
        for (int i = T.tiles[kLaneTileInts * t]; i < T.tiles[kLaneTileInts * t + 1]; ++i) part = std::fma(H.ral[(size_t)i * NLP + l], X(Q.y, i, s), part);
        aty += part;
      }
      const double x = X(Q.x, j, s);
      const double xp = clampd(std::fma(-Q.tau[s], X(Q.c, j, s) - aty, x), Q.lb[j], Q.ub[j]);
      const double tt = 2.0 * xp - x;
      xbl[(size_t)l * 64 + s] = tt; xpl[(size_t)l * 64 + s] = xp;
      x_out[(size_t)j * 64 + s] = std::fma(Q.oml[s], X(Q.x0, j, s) - tt, tt);
      xp_out[(size_t)j * 64 + s] = xp;
      const double dx = xp - x, d0 = xp - X(Q.x0, j, s);
      vsum[0 * 64 + s] += dx * dx; vsum[6 * 64 + s] += d0 * d0;
    }
  }

  // ---- the tiles, lane by lane, mode by mode ------------------------------------------------------------------------------------------
  int bad_ring = 0;
  std::vector<double> lp0((size_t)NLP * 64, 0.0), lp1((size_t)NLP * 64, 0.0);
  for (int mode = 0; mode < 3; ++mode) {
    for (int t = 0; t < T.ntile; ++t)
      for (int s = 0; s < S; ++s) {
        std::vector<double> ring((size_t)3 * R * 64, NAN);
        LaneGroup G{};
        G.x_in = Q.x.data(); G.y_in = Q.y.data(); G.x0 = Q.x0.data(); G.c = Q.c.data(); G.y0 = Q.y0.data();
        G.x_out = x_out.data(); G.y_out = y_out.data(); G.xbl = xbl.data(); G.xpl = xpl.data(); G.xp = xp_out.data(); G.yp = yp_out.data();
        LaneScalars sc{Q.tau[s], Q.sig[s], Q.oml[s], false, true};
        LaneOut<NLP> out;
        if (mode == 0) LaneTile<WC, WR, NLP, 4, true, false, 0>::run(P, G, t, s, sc, ring.data(), nullptr, out);
        else if (mode == 1) LaneTile<WC, WR, NLP, 4, true, false, 1>::run(P, G, t, s, sc, ring.data(), nullptr, out);
        else LaneTile<WC, WR, NLP, 4, true, false, 2>::run(P, G, t, s, sc, ring.data(), nullptr, out);
        for (int l = 0; l < NLP; ++l) { if (mode == 0) lp0[(size_t)l * 64 + s] += out.lp[l]; if (mode == 1) lp1[(size_t)l * 64 + s] += out.lp[l]; }
        if (mode == 1) for (int q = 0; q < 8; ++q) vsum[(size_t)q * 64 + s] += out.v[q];
        if (mode == 2) for (int q = 8; q < 13; ++q) vsum[(size_t)q * 64 + s] += out.v[q];
        for (double v : out.lp) if (std::isnan(v)) bad_ring++;
      }
  }
  // long columns' reduced costs (k_lane_long<2>) from the partial sums of A^T y+
  for (int l = 0; l < nl; ++l) {
    const int j = H.long_id[l];
    for (int s = 0; s < S; ++s) {
      const double aty = lp1[(size_t)l * 64 + s], cj = X(Q.c, j, s), xp = xp_out[(size_t)j * 64 + s], rc = cj - aty;
      const double lp = std::fabs(Q.lb[j]) < INFINITY ? std::fmax(rc, 0.0) : 0.0, lm = std::fabs(Q.ub[j]) < INFINITY ? std::fmax(-rc, 0.0) : 0.0;
      const double dr = (rc - lp + lm) / Q.col_scale[j];
      vsum[8 * 64 + s] += dr * dr; vsum[9 * 64 + s] += cj * xp; vsum[10 * 64 + s] += lp * fin0(Q.lb[j]) - lm * fin0(Q.ub[j]);
      vsum[11 * 64 + s] += std::fabs(cj * xp); vsum[12 * 64 + s] += std::fabs(rc - lp + lm) * std::fabs(xp);
    }
  }
  // ---- compare ----------------------------------------------------------------------------------------------------------------------------
  double ex = 0, ey = 0, exp_ = 0, eyp = 0, elp = 0, ev = 0;
  int missing = 0, first_missing_col = -1, first_missing_row = -1;
  for (int s = 0; s < S; ++s) {
    for (int j = 0; j < n; ++j) {
      const size_t at = (size_t)j * 64 + s;
      if (std::isnan(x_out[at]) || std::isnan(xp_out[at])) { missing++; if (first_missing_col < 0) first_missing_col = j; continue; }
      ex = std::fmax(ex, std::fabs(x_out[at] - xn_ref[at])); exp_ = std::fmax(exp_, std::fabs(xp_out[at] - xp_ref[at]));
    }
    for (int i = 0; i < m; ++i) {
      const size_t at = (size_t)i * 64 + s;
      if (std::isnan(y_out[at]) || std::isnan(yp_out[at])) { missing++; if (first_missing_row < 0) first_missing_row = i; continue; }
      ey = std::fmax(ey, std::fabs(y_out[at] - yn_ref[at])); eyp = std::fmax(eyp, std::fabs(yp_out[at] - yp_ref[at]));
    }
    for (int l = 0; l < nl; ++l) {
      const int j = H.long_id[l];
      double a0 = 0.0, a1 = 0.0;
      for (int p = Q.AT.ptr[j]; p < Q.AT.ptr[j + 1]; ++p) { a0 += Q.AT.val[p] * yn_ref[(size_t)Q.AT.idx[p] * 64 + s]; a1 += Q.AT.val[p] * yp_ref[(size_t)Q.AT.idx[p] * 64 + s]; }
      elp = std::fmax(elp, std::fmax(std::fabs(lp0[(size_t)l * 64 + s] - a0), std::fabs(lp1[(size_t)l * 64 + s] - a1)));
    }
    for (int q = 0; q < 13; ++q) {
      if (q == 7) continue;
      const double r = vref[(size_t)q * 64 + s], v = vsum[(size_t)q * 64 + s];
      ev = std::fmax(ev, std::fabs(v - r) / std::fmax(1.0, std::fabs(r)));
    }
  }
  // lanes that are not in use and the sink rows aside, nothing else was touched: count the finite entries per array
  printf("{\"ok\": true, \"ntile\": %d, \"nunit\": %d, \"max_units\": %d, \"ring\": %d, \"wc\": %d, \"wr\": %d, \"nlp\": %d, \"long_cols\": %d, "
         "\"halo\": %.4f, \"err_x\": %.3e, \"err_y\": %.3e, \"err_xp\": %.3e, \"err_yp\": %.3e, \"err_lp\": %.3e, \"err_sums\": %.3e, "
         "\"missing\": %d, \"first_missing_col\": %d, \"first_missing_row\": %d, \"nan_partials\": %d}\n",
         T.ntile, T.nunit, T.max_units, T.ring, WC, WR, NLP, nl, T.halo_rows, ex, ey, exp_, eyp, elp, ev, missing, first_missing_col, first_missing_row, bad_ring);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  const int rows_per_tile = atoi(argv[2]);
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 4;
  Problem Q;
  HostCSR &A = Q.A;
  A.m = hdr[0]; A.n = hdr[1];
  A.ptr.resize(A.m + 1); A.idx.resize(hdr[2]); A.val.resize(hdr[2]);
  if (fread(A.ptr.data(), 4, A.ptr.size(), f) != A.ptr.size() || fread(A.idx.data(), 4, A.idx.size(), f) != A.idx.size() ||
      fread(A.val.data(), 8, A.val.size(), f) != A.val.size()) return 5;
  fclose(f);
  const int n = A.n, m = A.m;
  Q.AT = transpose(A);
  Q.H = build_lane_plan(A, Q.AT, std::max(4096, A.m / 4));                       // as lane_create
  if (!Q.H.ok) { printf("{\"ok\": false, \"why\": \"plan\", \"long_cols\": %d}\n", Q.H.nl); return 0; }
  Q.T = build_lane_tiles(Q.H, rows_per_tile, 4, argc > 3 ? atoi(argv[3]) : 8);
  if (!Q.T.ok) { printf("{\"ok\": false, \"why\": \"tiles\", \"long_cols\": %d}\n", Q.H.nl); return 0; }
  Q.S = 3;
  const size_t NC = (size_t)(n + 1) * 64, NR = (size_t)(m + 1) * 64;
  Q.x.assign(NC, NAN); Q.x0.assign(NC, NAN); Q.c.assign(NC, NAN); Q.y.assign(NR, NAN); Q.y0.assign(NR, NAN);
  Q.lb.resize(n); Q.ub.resize(n); Q.rlo.resize(m); Q.rhi.resize(m); Q.col_scale.resize(n); Q.row_scale.resize(m);
  for (int j = 0; j < n; ++j) {
    Q.lb[j] = (j % 3 == 0) ? -INFINITY : -0.8; Q.ub[j] = (j % 5 == 0) ? INFINITY : 0.9; Q.col_scale[j] = 1.0 + 0.25 * (j % 4);
    for (int s = 0; s < Q.S; ++s) {
      Q.x[(size_t)j * 64 + s] = std::sin(0.37 * j + 1.0 + s); Q.x0[(size_t)j * 64 + s] = std::cos(0.11 * j + 0.3 * s);
      Q.c[(size_t)j * 64 + s] = 0.3 * std::sin(0.7 * j + 2.0 * s);
    }
  }
  for (int i = 0; i < m; ++i) {
    Q.rlo[i] = (i % 2) ? -0.1 : -INFINITY; Q.rhi[i] = (i % 3) ? 0.2 : ((i % 2) ? -0.1 : INFINITY);
    if (Q.rhi[i] < Q.rlo[i]) Q.rhi[i] = Q.rlo[i];
    Q.row_scale[i] = 1.0 + 0.5 * (i % 3);
    for (int s = 0; s < Q.S; ++s) { Q.y[(size_t)i * 64 + s] = std::cos(0.53 * i + 2.0 + s); Q.y0[(size_t)i * 64 + s] = std::sin(0.29 * i + 0.7 * s); }
  }
  Q.tau = {0.41, 0.23, 0.57}; Q.sig = {0.37, 0.61, 0.19}; Q.oml = {1.0 / 7.0, 1.0 / 3.0, 1.0 / 40.0};
  const int wc = Q.H.WC, wr = Q.H.WR, nlp = Q.H.NLP;
  if (nlp == 4) {
    if (wc == 4 && wr == 4) return run_all<4, 4, 4>(Q);
    if (wc == 4 && wr == 8) return run_all<4, 8, 4>(Q);
    return run_all<8, 8, 4>(Q);
  }
  if (wc == 4 && wr == 4) return run_all<4, 4, 8>(Q);
  if (wc == 4 && wr == 8) return run_all<4, 8, 8>(Q);
  return run_all<8, 8, 8>(Q);
}
