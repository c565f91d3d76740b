// CPU harness of the banded solves of csrc/dsp_ipm.hip (test infrastructure): the per-lane arithmetic of csrc/dsp_ipm_seq.hpp - the
// sequential walks and the time-parallel form (partitions, spikes, reduced system, border sums / corrections) - is run here exactly as
// the kernels run it (k_seq: one State per lane and partition, rows in order, streams gathered into `row[q * 64]`; the elementwise
// kernels: one call per lane) on random symmetric positive definite band matrices, and compared with a plain banded Cholesky in long
// double.  Usage: ipm_par_harness m W Lp seed  ->  one JSON line.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../dispatches_amd/csrc/dsp_ipm_seq.hpp"

using namespace dsp;

static const size_t Bp = 64;
static const int NL = 3;                      // lanes with data (each its own matrix)

template <int NS, class Body>
static void walk(const Body &body, double *const (&streams)[NS], unsigned outmask, int row0, int rows, bool reverse, int p) {
  for (int l = 0; l < NL; ++l) {
    typename Body::State st;
    body.init(st, p);
    for (int lr = 0; lr < rows; ++lr) {
      const size_t phys = (size_t)row0 + (reverse ? rows - 1 - lr : lr);
      double buf[NS * 64];
      for (int q = 0; q < NS; ++q) buf[q * 64] = streams[q][phys * Bp + l];
      body.step(st, buf, lr);
      for (int q = 0; q < NS; ++q)
        if ((outmask >> q) & 1u) streams[q][phys * Bp + l] = buf[q * 64];
    }
    body.finish(st, p, (size_t)l);
  }
}

template <int W>
static int run(int m, int Lp, unsigned seed) {
  const int Mp = m + W, W1 = W + 1;
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  // B = F F' + diag with F lower banded of half-bandwidth W (so is B), entries spanning decades
  std::vector<double> band0((size_t)W1 * Mp * Bp, 0.0);
  auto BAND = [&](std::vector<double> &b, int k, int t, int l) -> double & { return b[((size_t)k * Mp + t) * Bp + l]; };
  for (int l = 0; l < NL; ++l) {
    const int hw = W;
    std::vector<double> F((size_t)m * (hw + 1));
    for (int t = 0; t < m; ++t)
      for (int k = 0; k <= hw; ++k) F[(size_t)t * (hw + 1) + k] = (t - k >= 0) ? U(rng) * std::pow(10.0, 2.0 * U(rng)) : 0.0;
    // (F F')(t, t - k) = sum_c F(t, c) F(t - k, c), c in [t - hw, t - k]
    for (int t = 0; t < m; ++t)
      for (int k = 0; k <= W && k <= t; ++k) {
        double v = 0.0;
        for (int c = std::max(0, t - hw); c <= t - k; ++c) {
          const int k1 = t - c, k2 = t - k - c;
          if (k1 <= hw && k2 <= hw && k2 >= 0) v += F[(size_t)t * (hw + 1) + k1] * F[(size_t)(t - k) * (hw + 1) + k2];
        }
        if (k == 0) v += 1e-3 + std::fabs(U(rng));
        BAND(band0, k, t, l) = v;
      }
    for (int t = m; t < Mp; ++t) BAND(band0, 0, t, l) = 1.0;
  }
  std::vector<double> rhs((size_t)m * Bp, 0.0);
  for (int l = 0; l < NL; ++l)
    for (int t = 0; t < m; ++t) rhs[(size_t)t * Bp + l] = U(rng);
  // reference: banded Cholesky in long double
  std::vector<long double> ref((size_t)m * NL);
  for (int l = 0; l < NL; ++l) {
    std::vector<long double> Lb((size_t)m * W1, 0.0L), d(m);
    for (int t = 0; t < m; ++t) {
      for (int k = std::min(W, t); k >= 0; --k) {
        long double v = BAND(band0, k, t, l);
        const int c = t - k;
        for (int c2 = std::max(0, t - W); c2 < c; ++c2) {
          const int ka = t - c2, kb = c - c2;
          if (kb <= W) v -= Lb[(size_t)t * W1 + ka] * Lb[(size_t)c * W1 + kb] * d[c2];
        }
        if (k == 0) d[t] = v; else Lb[(size_t)t * W1 + k] = v / d[c];
      }
    }
    std::vector<long double> z(m);
    for (int t = 0; t < m; ++t) {
      long double v = rhs[(size_t)t * Bp + l];
      for (int k = 1; k <= std::min(W, t); ++k) v -= Lb[(size_t)t * W1 + k] * z[t - k];
      z[t] = v;
    }
    for (int t = m - 1; t >= 0; --t) {
      long double v = z[t] / d[t];
      for (int k = 1; k <= W && t + k < m; ++k) v -= Lb[(size_t)(t + k) * W1 + k] * ref[(size_t)(t + k) * NL + l];
      ref[(size_t)t * NL + l] = v;
    }
  }
  auto err_of = [&](const std::vector<double> &x) {
    long double e = 0.0L, s = 0.0L;
    for (int l = 0; l < NL; ++l)
      for (int t = 0; t < m; ++t) {
        e = std::max(e, fabsl((long double)x[(size_t)t * Bp + l] - ref[(size_t)t * NL + l]));
        s = std::max(s, fabsl(ref[(size_t)t * NL + l]));
      }
    return (double)(e / s);
  };

  // ---- sequential form (the kernels' launch geometry with one partition) -----------------------------------------------------------------
  double err_seq;
  {
    std::vector<double> band = band0, x = rhs;
    const size_t stride = (size_t)Mp * Bp;
    {
      double *st[W + 1];
      for (int q = 0; q <= W; ++q) st[q] = band.data() + q * stride;
      walk<W + 1>(FactorBody<W>{nullptr, Bp, 1}, st, (1u << (W + 1)) - 1u, 0, Mp, false, 0);
    }
    {
      double *st[W + 1];
      for (int q = 0; q < W; ++q) st[q] = band.data() + ((size_t)Mp + W) * Bp + q * stride;
      st[W] = x.data();
      walk<W + 1>(ForwardBody<W>{}, st, 1u << W, 0, m, false, 0);
    }
    {
      double *st[W + 2];
      for (int q = 0; q <= W; ++q) st[q] = band.data() + (size_t)W * Bp + q * stride;
      st[W + 1] = x.data();
      walk<W + 2>(BackwardBody<W>{}, st, 1u << (W + 1), 0, m, true, 0);
    }
    err_seq = err_of(x);
  }

  // ---- time-parallel form ------------------------------------------------------------------------------------------------------------------
  IpmParts g{};
  g.m = m; g.W = W; g.Lp = Lp; g.P = (m + Lp - 1) / Lp;
  if (g.P > 1 && m - (g.P - 1) * Lp < 2 * W + 1) g.P -= 1;
  double err_par = -1.0;
  if (g.P >= 2 && Lp >= 2 * W + 1) {
    constexpr int NT = IpmRed<W>::NT, NR = IpmRed<W>::NR;
    std::vector<double> band = band0, x = rhs, gs((size_t)W * Mp * Bp, NAN);          // (spike streams: never-written rows must not matter)
    std::vector<double> sfin((size_t)g.P * NT * Bp, NAN), cfin((size_t)g.P * NT * Bp, 0.0), redf((size_t)g.P * NR * Bp, NAN), bd((size_t)g.P * W * Bp, 0.0);
    const size_t stride = (size_t)Mp * Bp;
    // assembly's diversion (k_ipm_assemble)
    for (int l = 0; l < NL; ++l)
      for (int t = 0; t < m; ++t) {
        const int p = g.part_of(t), i = t - g.start(p);
        if (p < 1 || i >= W) continue;
        for (int j = 0; j < W; ++j) {
          const int k = i - j + W;
          gs[((size_t)j * Mp + t) * Bp + l] = k <= W ? BAND(band, k, t, l) : 0.0;
        }
        for (int k = 1; k <= W; ++k)
          if (ipm_diverted(g, t, k)) BAND(band, k, t, l) = 0.0;
      }
    for (int p = 0; p < g.P; ++p) {
      double *st[W + 1];
      for (int q = 0; q <= W; ++q) st[q] = band.data() + q * stride;
      const int rows = p == g.P - 1 ? Mp - g.start(p) : Lp;
      walk<W + 1>(FactorBody<W>{sfin.data(), Bp, g.P}, st, (1u << (W + 1)) - 1u, g.start(p), rows, false, p);
    }
    for (int p = 1; p < g.P; ++p) {
      double *st[2 * W + 1];
      for (int q = 0; q <= W; ++q) st[q] = band.data() + (size_t)W * Bp + q * stride;
      for (int j = 0; j < W; ++j) st[W + 1 + j] = gs.data() + j * stride;
      walk<2 * W + 1>(SpikeBody<W>{cfin.data(), Bp, g}, st, ((1u << W) - 1u) << (W + 1), g.start(p), g.cols(p), false, p);
    }
    for (int l = 0; l < NL; ++l) ipm_red_factor_lane<W>(g, sfin.data(), cfin.data(), gs.data(), stride, redf.data(), Bp, (size_t)l);
    // one solve
    for (int p = 0; p < g.P; ++p) {
      double *st[W + 1];
      for (int q = 0; q < W; ++q) st[q] = band.data() + ((size_t)Mp + W) * Bp + q * stride;
      st[W] = x.data();
      walk<W + 1>(ForwardBody<W>{}, st, 1u << W, g.start(p), g.cols(p), false, p);
    }
    for (int p = 1; p < g.P; ++p)
      for (int l = 0; l < NL; ++l) {
        double tot[W];
        for (int j = 0; j < W; ++j) tot[j] = 0.0;
        for (int wv = 0; wv < 4; ++wv) {                     // four waves' partial sums, as the kernel forms them
          double acc[W];
          ipm_border_dot_lane<W>(g, p, wv, 4, gs.data(), stride, x.data(), Bp, (size_t)l, acc);
          for (int j = 0; j < W; ++j) tot[j] += acc[j];
        }
        for (int j = 0; j < W; ++j) bd[((size_t)p * W + j) * Bp + l] = tot[j];
      }
    for (int l = 0; l < NL; ++l) ipm_red_solve_lane<W>(g, redf.data(), bd.data(), x.data(), Bp, (size_t)l);
    for (int p = 1; p < g.P; ++p)
      for (int l = 0; l < NL; ++l)
        for (int wv = 0; wv < 4; ++wv) ipm_border_apply_lane<W>(g, p, wv, 4, gs.data(), stride, band.data(), x.data(), Bp, (size_t)l);
    for (int p = 0; p < g.P; ++p) {
      double *st[W + 2];
      for (int q = 0; q <= W; ++q) st[q] = band.data() + (size_t)W * Bp + q * stride;
      st[W + 1] = x.data();
      walk<W + 2>(BackwardBody<W>{}, st, 1u << (W + 1), g.start(p), g.cols(p), true, p);
    }
    err_par = err_of(x);
  }
  printf("{\"m\": %d, \"W\": %d, \"Lp\": %d, \"P\": %d, \"err_seq\": %.3e, \"err_par\": %.3e}\n", m, W, Lp, g.P, err_seq, err_par);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s m W Lp seed\n", argv[0]); return 2; }
  const int m = atoi(argv[1]), W = atoi(argv[2]), Lp = atoi(argv[3]);
  const unsigned seed = (unsigned)atoi(argv[4]);
  if (W == 6) return run<6>(m, Lp, seed);
  if (W == 8) return run<8>(m, Lp, seed);
  return 2;
}
