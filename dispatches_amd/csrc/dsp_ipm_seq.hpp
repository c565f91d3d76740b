// dsp_ipm_seq.hpp — the per-lane arithmetic of the banded solves of dsp_ipm.hip, gfx950; also compiled into the CPU harness
// tests/ipm_par_harness.cpp (every function here is one scenario's arithmetic on scenario-minor arrays: `row[q * 64]` is stream q of the
// current row for this lane, `p[index * Bp + s]` entry `index` of scenario s; nothing crosses lanes).
//
// Sequential form: the normal matrix  B = A_s Theta A_s' + Theta_r  (half-bandwidth W, m rows) is factorised  L D L'  by one walk over
// its m rows and solved by one walk forward and one backward: m dependent steps each, ~60 - 100 ns per step - 5 - 9 ms per walk at
// T = 8736, whatever the batch (round 5, first form: 56 of the 73 ms of a Newton iteration).
//
// TIME-PARALLEL form (this file's second half): the rows are cut into P partitions of Lp rows; the last W rows of every partition but the
// last are a SEPARATOR - with half-bandwidth W the interiors of two partitions do not see each other - and the elimination order is
// "all interiors (independently), then the separators":
//   factor   per partition p: banded LDL' of its interior; the walk simply continues over its own separator rows, so that its final
//            window is the separator's diagonal block with the interior eliminated from the right-hand side of the band (FactorBody::finish);
//            the W columns that couple the interior to the separator on its LEFT (entries B(t, t - k) with t - k before the partition:
//            k_ipm_assemble diverts them into the spike streams) are carried through the interior's forward substitution (SpikeBody):
//            spike j of column c is the factor's entry L(left separator row j, c); its walk also leaves the coupling block between the
//            two separators (its values in the partition's own separator rows) and the left separator's Schur update (finish);
//   reduced  the separators form a block-tridiagonal system of P - 1 blocks of W x W: block LDL', one lane per scenario, sequential
//            (ipm_red_factor_lane; 63 blocks instead of 52 k rows);
//   solve    forward walk per partition (ForwardBody, as before, fresh per partition) -> border sums (spikes . z, per partition) ->
//            reduced solve (ipm_red_solve_lane) -> border correction of the interiors' right-hand sides -> backward walk per partition
//            (BackwardBody as before: it passes through the partition's separator rows, whose factor slots hold identity columns).
// A symmetric positive definite matrix may be eliminated in any order: the result is the sequential form's up to rounding
// (tests/test_ipm_par_cpu.py: both against a dense solve; GPU: tests/test_hip_ipm.py runs both forms on the same batch).
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define IPM_HD __host__ __device__ __forceinline__
#else
#define IPM_HD inline
#endif

namespace dsp {

#if defined(__HIPCC__)
#define IPM_FMA(a, b, c) fma((a), (b), (c))
#else
}  // namespace dsp
#include <cmath>
namespace dsp {
#define IPM_FMA(a, b, c) std::fma((a), (b), (c))
#endif

// geometry of a partitioned walk over m rows (P = 1: the sequential form)
struct IpmParts {
  int P, Lp, m, W;
  IPM_HD int start(int p) const { return p * Lp; }
  IPM_HD int cols(int p) const { return p == P - 1 ? m - p * Lp : Lp; }             // rows of the partition in a solve
  IPM_HD int interior(int p) const { return p == P - 1 ? m - p * Lp : Lp - W; }     // columns eliminated inside the partition
  IPM_HD int part_of(int t) const { const int p = t / Lp; return p < P ? p : P - 1; }
};
IPM_HD constexpr int ipm_tri(int a, int b) { return a * (a + 1) / 2 + b; }          // lower triangle, a >= b
// the band entry B(t, t - k), k >= 1, of a row in a partition's first W rows whose column lies before the partition belongs to the
// spikes: spike j = t - k - (start - W) of row t (k_ipm_assemble; tests/ipm_par_harness.cpp)
IPM_HD bool ipm_diverted(const IpmParts &g, int t, int k) {
  if (t >= g.m) return false;
  const int p = g.part_of(t);
  return p >= 1 && t - k < g.start(p);
}

// banded LDL': streams q = 0 .. W hold B(t, t - q) on the way in; on the way out row t holds column t - W of the factor:
// stream 0 = 1 / d, stream q = L(t - W + q, t - W).  A walk that starts at a partition's first row starts from the identity window:
// the first W rows it writes are identity columns (1 / d = 1, L = 0), the slots of the separator columns the band does not eliminate.
template <int W>
struct FactorBody {
  struct State { double S[W + 1][W + 1]; };
  double *sfin;                              // [P - 1][W (W + 1) / 2][Bp]: the final window of every partition but the last (null: P = 1)
  size_t Bp;
  int P;
  IPM_HD void init(State &st, int) const {
#pragma unroll
    for (int a = 0; a <= W; ++a)
#pragma unroll
      for (int b = 0; b <= W; ++b) st.S[a][b] = a == b ? 1.0 : 0.0;
  }
  IPM_HD void step(State &st, double *row, int) const {
#pragma unroll
    for (int b = 0; b <= W; ++b) st.S[W][b] = row[(size_t)(W - b) * 64];
    double d = st.S[0][0];
    if (!(d > 1e-200)) d = 1e64;                      // a dependent row: dropped
    const double inv = 1.0 / d;
    double l[W + 1], c0[W + 1];                        // c0: column 0 of the window (the update below overwrites it in place)
#pragma unroll
    for (int a = 1; a <= W; ++a) { c0[a] = st.S[a][0]; l[a] = c0[a] * inv; }
    row[0] = inv;
#pragma unroll
    for (int a = 1; a <= W; ++a) row[(size_t)a * 64] = l[a];
#pragma unroll
    for (int a = 1; a <= W; ++a)
#pragma unroll
      for (int b = 1; b <= a; ++b) st.S[a - 1][b - 1] = IPM_FMA(-l[a], c0[b], st.S[a][b]);
  }
  IPM_HD void finish(const State &st, int p, size_t s) const {
    if (!sfin || p >= P - 1) return;
#pragma unroll
    for (int a = 0; a < W; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) sfin[((size_t)p * (W * (W + 1) / 2) + ipm_tri(a, b)) * Bp + s] = st.S[a][b];
  }
};

// forward substitution L z = r: streams 0 .. W-1 = L(i + q + 1, i) (factor row i + W), stream W = r / z
template <int W>
struct ForwardBody {
  struct State { double acc[W]; };
  IPM_HD void init(State &st, int) const {
#pragma unroll
    for (int k = 0; k < W; ++k) st.acc[k] = 0.0;
  }
  IPM_HD void step(State &st, double *row, int) const {
    const double x = row[(size_t)W * 64] + st.acc[0];
    row[(size_t)W * 64] = x;
#pragma unroll
    for (int k = 1; k <= W; ++k) st.acc[k - 1] = IPM_FMA(-row[(size_t)(k - 1) * 64], x, k < W ? st.acc[k] : 0.0);
  }
  IPM_HD void finish(const State &, int, size_t) const {}
};

// D^-1 and backward substitution L' x = z (rows in reverse): stream 0 = 1 / d, streams 1 .. W = L(i + q, i), stream W + 1 = z / x
template <int W>
struct BackwardBody {
  struct State { double xw[W]; };
  IPM_HD void init(State &st, int) const {
#pragma unroll
    for (int k = 0; k < W; ++k) st.xw[k] = 0.0;
  }
  IPM_HD void step(State &st, double *row, int) const {
    double x = row[(size_t)(W + 1) * 64] * row[0];
#pragma unroll
    for (int k = 0; k < W; ++k) x = IPM_FMA(-row[(size_t)(k + 1) * 64], st.xw[k], x);
    row[(size_t)(W + 1) * 64] = x;
#pragma unroll
    for (int k = W - 1; k > 0; --k) st.xw[k] = st.xw[k - 1];
    st.xw[0] = x;
  }
  IPM_HD void finish(const State &, int, size_t) const {}
};

// The spikes of a partition (p >= 1): W simultaneous forward substitutions on the columns that couple its interior to the separator on
// its left.  Stream 0 = 1 / d_c, streams 1 .. W = L(c + q, c) (factor row c + W: BackwardBody's view), streams W+1 .. 2W = spike j at row c: on the
// way in the diverted band entries B(c, left separator row j) in the partition's first W rows (anything later is ignored), on the way
// out z_j(c) / d_c = L(left separator row j, c) for an interior column, and for the partition's own separator rows z_j(c) itself:
// the coupling block between the two separators.  finish: C = sum over the interior of z_j z_j' / d, the left separator's Schur update.
template <int W>
struct SpikeBody {
  struct State { double acc[W][W]; double C[W * (W + 1) / 2]; int nint; };
  double *cfin;                              // [P][W (W + 1) / 2][Bp] (entry 0 unused)
  size_t Bp;
  IpmParts g;
  IPM_HD void init(State &st, int p) const {
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int k = 0; k < W; ++k) st.acc[j][k] = 0.0;
#pragma unroll
    for (int e = 0; e < W * (W + 1) / 2; ++e) st.C[e] = 0.0;
    st.nint = g.interior(p);
  }
  IPM_HD void step(State &st, double *row, int lr) const {
    const bool inner = lr < st.nint, head = lr < W;
    const double inv = row[0];
    double z[W], zs[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const double in = row[(size_t)(W + 1 + j) * 64];
      z[j] = (head ? in : 0.0) + st.acc[j][0];
    }
#pragma unroll
    for (int k = 1; k <= W; ++k) {
      const double lk = row[(size_t)k * 64];
#pragma unroll
      for (int j = 0; j < W; ++j) st.acc[j][k - 1] = IPM_FMA(-lk, z[j], k < W ? st.acc[j][k] : 0.0);
    }
#pragma unroll
    for (int j = 0; j < W; ++j) { zs[j] = inner ? z[j] * inv : z[j]; row[(size_t)(W + 1 + j) * 64] = zs[j]; }
    if (inner) {
#pragma unroll
      for (int j = 0; j < W; ++j)
#pragma unroll
        for (int j2 = 0; j2 <= j; ++j2) st.C[ipm_tri(j, j2)] = IPM_FMA(z[j], zs[j2], st.C[ipm_tri(j, j2)]);
    }
  }
  IPM_HD void finish(const State &st, int p, size_t s) const {
#pragma unroll
    for (int e = 0; e < W * (W + 1) / 2; ++e) cfin[((size_t)p * (W * (W + 1) / 2) + e) * Bp + s] = st.C[e];
  }
};

// ---- the reduced system: separators sigma = 0 .. P-2, block sigma = rows [start(sigma) + interior(sigma), start(sigma + 1)) ---------------
// per block in `redf` ([P - 1][W W + W (W + 1) / 2][Bp]): K_sigma = O_sigma Dt_(sigma-1)^-1 (row-major W x W), then the LDL' of
// Dt_sigma = D_sigma - K_sigma O_sigma': strictly lower L row by row, then 1 / d
template <int W>
struct IpmRed {
  static constexpr int NT = W * (W + 1) / 2, NK = W * W, NR = NK + NT;
};

// x := Dt^-1 x given the block's LDL' (Lt strictly lower, dinv)
template <int W>
IPM_HD void ipm_red_ldl_solve(const double (&Lt)[W][W], const double (&dinv)[W], double (&x)[W]) {
#pragma unroll
  for (int i = 1; i < W; ++i)
#pragma unroll
    for (int k = 0; k < i; ++k) x[i] = IPM_FMA(-Lt[i][k], x[k], x[i]);
#pragma unroll
  for (int i = 0; i < W; ++i) x[i] *= dinv[i];
#pragma unroll
  for (int i = W - 2; i >= 0; --i)
#pragma unroll
    for (int k = i + 1; k < W; ++k) x[i] = IPM_FMA(-Lt[k][i], x[k], x[i]);
}

// gs: spike stream j = gs + j * gs_stride, row r at r * Bp
template <int W>
IPM_HD void ipm_red_factor_lane(const IpmParts &g, const double *__restrict__ sfin, const double *__restrict__ cfin, const double *__restrict__ gs,
                                size_t gs_stride, double *__restrict__ redf, size_t Bp, size_t s) {
  constexpr int NT = IpmRed<W>::NT, NK = IpmRed<W>::NK, NR = IpmRed<W>::NR;
  double Lt[W][W], dinv[W];
#pragma unroll
  for (int a = 0; a < W; ++a) {
    dinv[a] = 1.0;
#pragma unroll
    for (int b = 0; b < W; ++b) Lt[a][b] = 0.0;
  }
  for (int sg = 0; sg < g.P - 1; ++sg) {
    double D[W][W], K[W][W];
#pragma unroll
    for (int a = 0; a < W; ++a)
#pragma unroll
      for (int b = 0; b < W; ++b) {
        K[a][b] = 0.0;
        D[a][b] = b <= a ? sfin[((size_t)sg * NT + ipm_tri(a, b)) * Bp + s] - cfin[((size_t)(sg + 1) * NT + ipm_tri(a, b)) * Bp + s] : 0.0;
      }
    if (sg >= 1) {
      const size_t e0 = (size_t)(g.start(sg) + g.interior(sg));
      double O[W][W];
#pragma unroll
      for (int a = 0; a < W; ++a)
#pragma unroll
        for (int j = 0; j < W; ++j) O[a][j] = gs[(size_t)j * gs_stride + (e0 + a) * Bp + s];
#pragma unroll
      for (int a = 0; a < W; ++a) {
        double y[W];
#pragma unroll
        for (int j = 0; j < W; ++j) y[j] = O[a][j];
        ipm_red_ldl_solve<W>(Lt, dinv, y);
#pragma unroll
        for (int j = 0; j < W; ++j) K[a][j] = y[j];
      }
#pragma unroll
      for (int a = 0; a < W; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          double t = D[a][b];
#pragma unroll
          for (int j = 0; j < W; ++j) t = IPM_FMA(-K[a][j], O[b][j], t);
          D[a][b] = t;
        }
    }
#pragma unroll
    for (int c = 0; c < W; ++c) {
      double d = D[c][c];
      if (!(d > 1e-200)) d = 1e64;
      dinv[c] = 1.0 / d;
#pragma unroll
      for (int a = c + 1; a < W; ++a) Lt[a][c] = D[a][c] * dinv[c];
#pragma unroll
      for (int a = c + 1; a < W; ++a)
#pragma unroll
        for (int b = c + 1; b <= a; ++b) D[a][b] = IPM_FMA(-Lt[a][c], D[b][c], D[a][b]);
    }
    double *out = redf + (size_t)sg * NR * Bp + s;
#pragma unroll
    for (int a = 0; a < W; ++a)
#pragma unroll
      for (int j = 0; j < W; ++j) out[(size_t)(a * W + j) * Bp] = K[a][j];
#pragma unroll
    for (int a = 0; a < W; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) out[(size_t)(NK + ipm_tri(a, b)) * Bp] = a == b ? dinv[a] : Lt[a][b];
  }
}

// x: the [m][Bp] vector after the partitions' forward walks; bd: the border sums [P][W][Bp] (entry p = what partition p's interior takes
// from the separator on its left).  On return the separator rows of x hold the solution there.
template <int W>
IPM_HD void ipm_red_solve_lane(const IpmParts &g, const double *__restrict__ redf, const double *__restrict__ bd, double *x, size_t Bp, size_t s) {
  constexpr int NK = IpmRed<W>::NK, NR = IpmRed<W>::NR;
  const int nb = g.P - 1;
  if (nb <= 0) return;
  double Kc[W][W], Kn[W][W], wp[W];
  auto load_k = [&](int sg, double (&K)[W][W]) {
    const double *in = redf + (size_t)sg * NR * Bp + s;
#pragma unroll
    for (int a = 0; a < W; ++a)
#pragma unroll
      for (int j = 0; j < W; ++j) K[a][j] = in[(size_t)(a * W + j) * Bp];
  };
#pragma unroll
  for (int j = 0; j < W; ++j) wp[j] = 0.0;
  load_k(0, Kc);
  for (int sg = 0; sg < nb; ++sg) {
    if (sg + 1 < nb) load_k(sg + 1, Kn);                     // (requested before this block's arithmetic: the chain does not wait for it)
    const size_t e0 = (size_t)(g.start(sg) + g.interior(sg));
    double w[W];
#pragma unroll
    for (int a = 0; a < W; ++a) {
      double t = x[(e0 + a) * Bp + s] - bd[((size_t)(sg + 1) * W + a) * Bp + s];
#pragma unroll
      for (int j = 0; j < W; ++j) t = IPM_FMA(-Kc[a][j], wp[j], t);
      w[a] = t;
    }
#pragma unroll
    for (int a = 0; a < W; ++a) { x[(e0 + a) * Bp + s] = w[a]; wp[a] = w[a]; }
#pragma unroll
    for (int a = 0; a < W; ++a)
#pragma unroll
      for (int j = 0; j < W; ++j) Kc[a][j] = Kn[a][j];
  }
  // backward: x_sigma = Dt_sigma^-1 w_sigma - K_(sigma+1)' x_(sigma+1)
  double xn[W];
#pragma unroll
  for (int j = 0; j < W; ++j) xn[j] = 0.0;
#pragma unroll
  for (int a = 0; a < W; ++a)
#pragma unroll
    for (int j = 0; j < W; ++j) Kc[a][j] = 0.0;                // K of block sigma + 1 (none above the last block)
  for (int sg = nb - 1; sg >= 0; --sg) {
    const double *in = redf + (size_t)sg * NR * Bp + s;
    double Lt[W][W], dinv[W], u[W];
#pragma unroll
    for (int a = 0; a < W; ++a)
#pragma unroll
      for (int b = 0; b < W; ++b) {
        if (b < a) Lt[a][b] = in[(size_t)(NK + ipm_tri(a, b)) * Bp];
        else { Lt[a][b] = 0.0; if (b == a) dinv[a] = in[(size_t)(NK + ipm_tri(a, a)) * Bp]; }
      }
    load_k(sg, Kn);                                            // this block's K: used by the block below
    const size_t e0 = (size_t)(g.start(sg) + g.interior(sg));
#pragma unroll
    for (int a = 0; a < W; ++a) u[a] = x[(e0 + a) * Bp + s];
    ipm_red_ldl_solve<W>(Lt, dinv, u);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      double t = u[j];
#pragma unroll
      for (int a = 0; a < W; ++a) t = IPM_FMA(-Kc[a][j], xn[a], t);
      u[j] = t;
    }
#pragma unroll
    for (int j = 0; j < W; ++j) { x[(e0 + j) * Bp + s] = u[j]; xn[j] = u[j]; }
#pragma unroll
    for (int a = 0; a < W; ++a)
#pragma unroll
      for (int j = 0; j < W; ++j) Kc[a][j] = Kn[a][j];
  }
}

// border sums of partition p (>= 1): acc[j] += sum over the interior columns c = first + wv, first + wv + nwv, ... of spike_j(c) z(c)
template <int W>
IPM_HD void ipm_border_dot_lane(const IpmParts &g, int p, int wv, int nwv, const double *__restrict__ gs, size_t gs_stride,
                                const double *__restrict__ z, size_t Bp, size_t s, double (&acc)[W]) {
  const int c0 = g.start(p), c1 = c0 + g.interior(p);
#pragma unroll
  for (int j = 0; j < W; ++j) acc[j] = 0.0;
#pragma unroll 4
  for (int c = c0 + wv; c < c1; c += nwv) {
    const double zc = z[(size_t)c * Bp + s];
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] = IPM_FMA(gs[(size_t)j * gs_stride + (size_t)c * Bp + s], zc, acc[j]);
  }
}

// border correction of partition p (>= 1) before its backward walk: z(c) -= d_c sum_j spike_j(c) x(left separator row j)
// (inv: stream 0 of the factor, whose row c + W holds 1 / d_c)
template <int W>
IPM_HD void ipm_border_apply_lane(const IpmParts &g, int p, int wv, int nwv, const double *__restrict__ gs, size_t gs_stride,
                                  const double *__restrict__ inv, double *x, size_t Bp, size_t s) {
  const int c0 = g.start(p), c1 = c0 + g.interior(p);
  double xl[W];
#pragma unroll
  for (int j = 0; j < W; ++j) xl[j] = x[(size_t)(c0 - W + j) * Bp + s];
#pragma unroll 4
  for (int c = c0 + wv; c < c1; c += nwv) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < W; ++j) t = IPM_FMA(gs[(size_t)j * gs_stride + (size_t)c * Bp + s], xl[j], t);
    x[(size_t)c * Bp + s] -= t / inv[(size_t)(c + W) * Bp + s];
  }
}

}  // namespace dsp
