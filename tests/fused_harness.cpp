// CPU emulation of the fused one-launch iteration of the HBM-resident streaming PDLP (dispatches_amd/csrc/dsp_stream.hip:
// k_fused / k_long_partials), built by tests/test_prepare_cpu.py with g++.  Reads a CSR, builds the streaming ELLs and the tile plan
// with the SAME host code the library uses (dsp_prepare.hpp), runs one iteration tile by tile exactly as the kernel's stages do
// (LDS buffers, halo columns, long-column slots, padding entries, per-tile partial sums) and compares with the plain PDHG +
// Halpern step on the CSR.  Prints one JSON object.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../dispatches_amd/csrc/dsp_prepare.hpp"

using namespace dsp;

static double clampd(double v, double lo, double hi) { return std::fmin(std::fmax(v, lo), hi); }

// every (groups, tiles) shape: the XCD-major renumbering of the fused launch's workgroups is a bijection onto groups x tiles, the
// renumbered part puts all groups of a tile on ONE XCD (workgroup id mod 8), and nothing maps beyond the plan
static int check_workgroup_order() {
  for (int G = 1; G <= 40; ++G)
    for (int ntile = 1; ntile <= 70; ++ntile) {
      const int full = fused_xcd_full(G, ntile), total = G * ntile;
      std::vector<int> seen(total, 0), xcd_of_tile(ntile, -1);
      for (int lin = 0; lin < total; ++lin) {
        int tile = -1, grp = -1;
        fused_workgroup(lin, G, full, tile, grp);
        if (tile < 0 || tile >= ntile || grp < 0 || grp >= G) { printf("{\"ok\": false, \"G\": %d, \"ntile\": %d, \"lin\": %d}\n", G, ntile, lin); return 1; }
        if (seen[tile * G + grp]++) { printf("{\"ok\": false, \"dup\": true, \"G\": %d, \"ntile\": %d}\n", G, ntile); return 1; }
        if (lin < full) {
          if (xcd_of_tile[tile] < 0) xcd_of_tile[tile] = lin & 7;
          if (xcd_of_tile[tile] != (lin & 7)) { printf("{\"ok\": false, \"split\": true, \"G\": %d, \"ntile\": %d}\n", G, ntile); return 1; }
        }
        // grid order off: identity
        int t0, g0;
        fused_workgroup(lin, G, 0, t0, g0);
        if (t0 != lin / G || g0 != lin % G) return 1;
      }
    }
  printf("{\"ok\": true}\n");
  return 0;
}

int main(int argc, char **argv) {
  if (argc == 2 && std::string(argv[1]) == "--workgroup-order") return check_workgroup_order();
  if (argc < 4) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  const int rows_per_tile = atoi(argv[2]), max_w = atoi(argv[3]);
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 4;
  HostCSR A;
  A.m = hdr[0]; A.n = hdr[1];
  A.ptr.resize(A.m + 1); A.idx.resize(hdr[2]); A.val.resize(hdr[2]);
  if (fread(A.ptr.data(), 4, A.ptr.size(), f) != A.ptr.size() || fread(A.idx.data(), 4, A.idx.size(), f) != A.idx.size() ||
      fread(A.val.data(), 8, A.val.size(), f) != A.val.size()) return 5;
  fclose(f);
  const int n = A.n, m = A.m;
  HostCSR AT = transpose(A);
  HostStreamELL Er = build_stream_ell(A, max_w, 2048), Ec = build_stream_ell(AT, max_w, 2048, std::max(4096, A.m / 4));   // as stream_create
  HostFusedPlan F = build_fused_plan(A, AT, Er, Ec, 4, rows_per_tile, 40 * 1024);
  const int nlong = (int)Ec.long_id.size();
  if (F.ntile == 0) { printf("{\"ntile\": 0, \"long_rows\": %zu, \"long_cols\": %d}\n", Er.long_id.size(), nlong); return 0; }
  // deterministic state
  std::vector<double> x(n), x0(n), c(n), lb(n), ub(n), y(m), y0(m), rlo(m), rhi(m);
  for (int j = 0; j < n; ++j) {
    x[j] = std::sin(0.37 * j + 1.0); x0[j] = std::cos(0.11 * j); c[j] = 0.3 * std::sin(0.7 * j);
    lb[j] = (j % 3 == 0) ? -INFINITY : -0.8; ub[j] = (j % 5 == 0) ? INFINITY : 0.9;
  }
  for (int i = 0; i < m; ++i) {
    y[i] = std::cos(0.53 * i + 2.0); y0[i] = std::sin(0.29 * i);
    rlo[i] = (i % 2) ? -0.1 : -INFINITY; rhi[i] = (i % 3) ? 0.2 : ((i % 2) ? -0.1 : INFINITY);
    if (rhi[i] < rlo[i]) rhi[i] = rlo[i];
  }
  const double tau = 0.41, sig = 0.37, oml = 1.0 / 7.0;
  // ---- reference: plain step on the CSR -------------------------------------------------------------------------------------
  std::vector<double> xbar(n), xn_ref(n), yn_ref(m);
  for (int j = 0; j < n; ++j) {
    double aty = 0.0;
    for (int p = AT.ptr[j]; p < AT.ptr[j + 1]; ++p) aty += AT.val[p] * y[AT.idx[p]];
    const double xp = clampd(x[j] - tau * (c[j] - aty), lb[j], ub[j]);
    xbar[j] = 2.0 * xp - x[j];
    xn_ref[j] = std::fma(oml, x0[j] - xbar[j], xbar[j]);
  }
  for (int i = 0; i < m; ++i) {
    double ax = 0.0;
    for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) ax += A.val[p] * xbar[A.idx[p]];
    const double gy = std::fma(-sig, ax, y[i]);
    const double yp = gy - clampd(gy, -sig * rhi[i], -sig * rlo[i]);
    const double tt = 2.0 * yp - y[i];
    yn_ref[i] = std::fma(oml, y0[i] - tt, tt);
  }
  // ---- k_long_partials on the current y -------------------------------------------------------------------------------------
  const int NY = F.ny_max, NXB = F.nxb_max, ntile = F.ntile;
  std::vector<double> lp_in((size_t)std::max(1, nlong) * ntile, 0.0), lp_out(lp_in.size(), 0.0);
  for (int t = 0; t < ntile; ++t) {
    const int32_t *tp = &F.tile[(size_t)t * 8];
    for (int i = tp[0]; i < tp[1]; ++i)
      for (int e = 0; e < Er.W; ++e) {
        const int gi = F.ridx_enc[(size_t)e * m + i];
        const double v = Er.val[(size_t)e * m + i];
        if (gi < 0 && v != 0.0) lp_in[(size_t)(-1 - gi) * ntile + t] += v * y[i];
      }
  }
  // ---- k_fused, tile by tile ------------------------------------------------------------------------------------------------
  std::vector<double> x_out(n, NAN), y_out(m, NAN);
  int oob = 0, uninit = 0;
  for (int t = 0; t < ntile; ++t) {
    const int32_t *tp = &F.tile[(size_t)t * 8];
    const int i0 = tp[0], i1 = tp[1], j0 = tp[2], j1 = tp[3], c_lo = tp[4], c_hi = tp[5], r_lo = tp[6], r_hi = tp[7];
    std::vector<double> ys(NY, NAN), xb(NXB, NAN);
    if (r_hi - r_lo > NY || nlong + c_hi - c_lo > NXB) oob++;
    for (int r = 0; r < r_hi - r_lo; ++r) ys[r] = y[r_lo + r];                                    // stage 0
    for (int l = 0; l < nlong; ++l) {
      double aty = 0.0;
      for (int q = 0; q < ntile; ++q) aty += lp_in[(size_t)l * ntile + q];
      const int j = Ec.long_id[l];
      const double xp = clampd(std::fma(-tau, c[j] - aty, x[j]), lb[j], ub[j]);
      xb[l] = 2.0 * xp - x[j];
    }
    for (int j = c_lo; j < c_hi; ++j) {                                                           // stage 1
      if (Ec.is_long[j]) continue;
      double aty = 0.0;
      for (int e = 0; e < Ec.W; ++e) {
        const double v = Ec.val[(size_t)e * n + j];
        const int gi = Ec.idx[(size_t)e * n + j];
        const int id = (gi < r_lo || gi >= r_hi) ? 0 : gi - r_lo;
        if (v != 0.0 && (gi < r_lo || gi >= r_hi)) oob++;
        if (std::isnan(ys[id])) uninit++;
        aty = std::fma(v, ys[id], aty);
      }
      const double xp = clampd(std::fma(-tau, c[j] - aty, x[j]), lb[j], ub[j]);
      xb[nlong + (j - c_lo)] = 2.0 * xp - x[j];
    }
    std::vector<double> lp(std::max(1, nlong), 0.0);
    for (int i = i0; i < i1; ++i) {                                                               // stage 2
      double ax = 0.0;
      std::vector<int> slot(Er.W);
      for (int e = 0; e < Er.W; ++e) {
        const double v = Er.val[(size_t)e * m + i];
        const int gi = F.ridx_enc[(size_t)e * m + i];
        slot[e] = gi < 0 ? -1 - gi : nlong + gi - c_lo;
        if (gi >= 0 && (gi < c_lo || gi >= c_hi)) { if (v != 0.0) oob++; slot[e] = 0; }
        if (slot[e] < 0 || slot[e] >= NXB) { oob++; slot[e] = 0; }
        if (std::isnan(xb[slot[e]])) uninit++;            // the kernel multiplies whatever the slot holds, padding entries (v = 0) included
        ax = std::fma(v, xb[slot[e]], ax);
      }
      const double yv = ys[i - r_lo];
      const double gy = std::fma(-sig, ax, yv);
      const double yp = gy - clampd(gy, -sig * rhi[i], -sig * rlo[i]);
      const double tt = 2.0 * yp - yv;
      const double yn = std::fma(oml, y0[i] - tt, tt);
      if (!std::isnan(y_out[i])) oob++;                                   // every row written exactly once
      y_out[i] = yn;
      for (int e = 0; e < Er.W; ++e) {
        const double v = Er.val[(size_t)e * m + i];
        if (slot[e] < nlong && v != 0.0 && F.ridx_enc[(size_t)e * m + i] < 0) lp[slot[e]] += v * yn;
      }
    }
    for (int j = j0; j < j1; ++j) {                                                               // stage 3
      if (Ec.is_long[j]) continue;
      const double tt = xb[nlong + (j - c_lo)];
      if (!std::isnan(x_out[j])) oob++;
      x_out[j] = std::fma(oml, x0[j] - tt, tt);
    }
    for (int l = 0; l < nlong; ++l) {
      const int j = Ec.long_id[l];
      if (j >= j0 && j < j1) { if (!std::isnan(x_out[j])) oob++; x_out[j] = std::fma(oml, x0[j] - xb[l], xb[l]); }
      lp_out[(size_t)l * ntile + t] = lp[l];
    }
  }
  double ex = 0.0, ey = 0.0, elp = 0.0;
  int missing = 0;
  for (int j = 0; j < n; ++j) { if (std::isnan(x_out[j])) missing++; else ex = std::fmax(ex, std::fabs(x_out[j] - xn_ref[j])); }
  for (int i = 0; i < m; ++i) { if (std::isnan(y_out[i])) missing++; else ey = std::fmax(ey, std::fabs(y_out[i] - yn_ref[i])); }
  for (int l = 0; l < nlong; ++l) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < ntile; ++q) a += lp_out[(size_t)l * ntile + q];
    const int j = Ec.long_id[l];
    for (int p = AT.ptr[j]; p < AT.ptr[j + 1]; ++p) b += AT.val[p] * yn_ref[AT.idx[p]];
    elp = std::fmax(elp, std::fabs(a - b));
  }
  printf("{\"ntile\": %d, \"ny\": %d, \"nxb\": %d, \"long_cols\": %d, \"wr\": %d, \"wc\": %d, \"err_x\": %.3e, \"err_y\": %.3e, \"err_lp\": %.3e, "
         "\"out_of_range\": %d, \"uninitialised\": %d, \"missing\": %d}\n", ntile, NY, NXB, nlong, Er.W, Ec.W, ex, ey, elp, oob, uninit, missing);
  return 0;
}
