// dsp_stream.hip — HBM-resident ("streaming") batched PDLP for LPs too large for the register/LDS-resident kernels
// (gfx950).  Workload: the long-horizon price-taker design LPs of the reference (SURVEY.md 8(f)-4:
// wind_battery_LMP.py:172-269, 8736 hourly periods -> n, m ~ 5e4) solved for a family of B scenarios that share the
// constraint matrix and differ in (c, bounds) - e.g. capital-cost x price sweeps, the 60-point (h2_price x
// pem_capacity) enumeration of price_taker_analysis.py:353-419.
//
// Here one scenario's state (x, anchor, y, anchor, c, bounds: ~12 vectors of n or m doubles, MBs) cannot live on a
// CU, so - unlike the fused kernels - the PDHG iteration really streams from HBM, and the 8 TB/s roofline is what
// bounds it.  Same algorithm as dsp_kernels.hip (restarted, reflected Halpern PDHG on the Ruiz + Pock-Chambolle scaled
// problem, PDLP restart criteria, proportional primal-weight controller with rounding guard, relative KKT termination in
// the original space + the eps_obj objective-error bounds); no ray jumps / stall rescue on this path.
//
// Structure: an iteration is two grid-wide dependent steps (A^T y for the primal step, A (2x+ - x) for the dual step),
// i.e. two kernel launches (a kernel boundary costs ~1.5 us on MI355X, an in-kernel grid barrier ~4 us, so launches
// it is).  The Halpern averaging is fused into the dual-step launch (rows do the dual step, columns the averaging of
// x: the reflected point 2x+ - x IS xbar, so x_new = x0/(k+2) + (1 - 1/(k+2)) xbar needs neither x nor x+), and x+ / y+
// themselves are only materialised at check iterations.  A plain iteration moves, per scenario, 8 n + 6 m doubles:
//     k_primal        reads x, c, lb, ub (+ gathers y)                  writes xbar = 2x+ - x
//     k_dual_halpern  reads y, y0, rlo, rhi (+ gathers xbar)            writes y ;   reads xbar, x0, writes x
// Every `check_every` iterations the dual step is replaced by a check sequence (k_check_rows, k_kkt_cols, k_control,
// k_apply) that produces the fixed-point residual, the KKT quantities and the restart / termination decision of every
// scenario ON THE DEVICE; per-scenario control state lives in HBM, so the host only enqueues launches and polls a
// "scenarios finished" counter every few check periods.  Reductions are two-stage and ordered (block partials in a
// fixed layout, summed by one block per scenario), so a solve is bit-reproducible.
//
// Matrix layout: ELL, entry-major ([W][nvec]: thread v reads entry e at e * nvec + v, coalesced) for A (rows) and A^T
// (columns); the matrix is read once per thread and reused across the SG scenarios a thread processes.  Vectors longer
// than the ELL width (design columns that touch every period) are "long": CSR segments reduced by one block each.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "dsp_device.hpp"
#include "dsp_stream.hpp"
#include "dsp_wave.hpp"

namespace dsp {

namespace {

constexpr int kTB = 256;                // threads per block
constexpr int kNQ = 16;                 // partial-sum slots per (scenario, block)
constexpr int kLongChunk = 2048;        // entries of a long vector per block in the plain-iteration kernels

__device__ __forceinline__ double fin0(double v) { return (fabs(v) < INFINITY) ? v : 0.0; }
__device__ __forceinline__ bool finite_d(double v) { return fabs(v) < INFINITY; }
__device__ __forceinline__ double clampd2(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// block-wide ordered sum of NQ quantities -> partial[(b * nblk + blk) * kNQ + q]   (wave shuffles, then LDS across waves)
template <int NQ>
__device__ __forceinline__ void block_partials(double (&v)[NQ], double *partial_out) {
  __shared__ double red[kTB / 64][kNQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double t = v[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if (lane == 0) red[wave][q] = t;
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kTB / 64; ++w) t += red[w][threadIdx.x];
    partial_out[threadIdx.x] = t;
  }
  __syncthreads();
}

// sum_e val[e][v] * vec[idx[e][v]]  for one vector v of an entry-major ELL
__device__ __forceinline__ double ell_dot(const StreamMatrix &M, int v, const double *__restrict__ vec) {
  double s = 0.0;
  for (int e = 0; e < M.W; ++e) s = fma(M.val[(size_t)e * M.nvec + v], vec[M.idx[(size_t)e * M.nvec + v]], s);
  return s;
}

// one block reduces one CHUNK (kLongChunk entries) of a long vector (CSR segment) against `vec`; the chunk partials are
// summed in chunk order by the *_long_finish kernels (a design column that touches every period has 3 T entries: one
// block per vector was the critical path of the whole iteration)
__device__ __forceinline__ double long_dot(const StreamMatrix &M, int chunk, const double *__restrict__ vec) {
  __shared__ double red[kTB / 64];
  double s = 0.0;
  const int p0 = M.chunk_begin[chunk], p1 = M.chunk_end[chunk];
  for (int p = p0 + threadIdx.x; p < p1; p += kTB) s = fma(M.long_val[p], vec[M.long_idx[p]], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < kTB / 64; ++w) t += red[w];
  __syncthreads();
  return t;                                     // every thread holds the total
}

// whole long vector by one block, chunk after chunk (check kernels only: once per check period)
__device__ __forceinline__ double long_dot_all(const StreamMatrix &M, int l, const double *__restrict__ vec) {
  double t = 0.0;
  for (int ch = M.long_chunk_ptr[l]; ch < M.long_chunk_ptr[l + 1]; ++ch) t += long_dot(M, ch, vec);
  return t;
}

// ---- init: scale the scenario's data, starting point, norms -----------------------------------------------------------
// partial slots: 0 |q|^2 unscaled (row + col bounds), 1 |c|^2 unscaled, 2 |q_rows|^2 scaled, 3 |c|^2 scaled,
//                4 sum of squared finite scaled column bounds; max slots (combined with fmax): 5 cmax, 6 qmax, 7 bad
__global__ void k_init(StreamArgs a) {
  const StreamProblem &P = a.P;
  const dsp_batch &b = a.b;
  const int t = blockIdx.x * kTB + threadIdx.x;
  const int s = blockIdx.y;
  const size_t on = (size_t)s * P.n, om = (size_t)s * P.m;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (t < P.n) {
    const double d = P.col_scale[t];
    const double cu = b.c[(size_t)s * b.c_stride + t];
    const double lu = b.var_lb ? b.var_lb[(size_t)s * b.var_lb_stride + t] : -INFINITY;
    const double uu = b.var_ub ? b.var_ub[(size_t)s * b.var_ub_stride + t] : INFINITY;
    const double cs = cu * d, ls = lu / d, us = uu / d;
    a.W.c[on + t] = cs; a.W.lb[on + t] = ls; a.W.ub[on + t] = us;
    const double x0 = clampd2((b.x0 ? b.x0[on + t] / d : 0.0), ls, us);
    a.W.x[on + t] = x0; a.W.x0[on + t] = x0; a.W.xp[on + t] = x0; a.W.xbar[on + t] = x0;
    v[0] += fin0(lu) * fin0(lu) + fin0(uu) * fin0(uu);
    v[1] += cu * cu;
    v[3] += cs * cs;
    v[4] += fin0(ls) * fin0(ls) + fin0(us) * fin0(us);
    v[5] = fabs(cs);
    if (!(lu <= uu) || !(cu == cu)) v[7] = 1.0;
  }
  if (t < P.m) {
    const double d = P.row_scale[t];
    const double lo = b.row_lb ? b.row_lb[(size_t)s * b.row_lb_stride + t] : -INFINITY;
    const double hi = b.row_ub ? b.row_ub[(size_t)s * b.row_ub_stride + t] : INFINITY;
    const double ls = lo * d, hs = hi * d;
    a.W.rlo[om + t] = ls; a.W.rhi[om + t] = hs;
    if (b.row_compliance) {
      // soft row (convex QP, include/dsp_hip.h): the term (a.x - b)^2 / (2 kappa), one finite target b = row_lb = row_ub
      const double kp = b.row_compliance[(size_t)s * b.row_compliance_stride + t];
      a.W.kap[om + t] = kp * d * d;
      if (!(kp >= 0.0) || (kp > 0.0 && !(lo == hi && finite_d(lo)))) v[7] = 1.0;
    }
    double ys = b.y0 ? b.y0[om + t] / d : 0.0;
    if (!finite_d(ls)) ys = fmin(ys, 0.0);
    if (!finite_d(hs)) ys = fmax(ys, 0.0);
    a.W.y[om + t] = ys; a.W.y0[om + t] = ys; a.W.yp[om + t] = ys;
    const double big = fmax(fabs(fin0(lo)), fabs(fin0(hi))), bigs = fmax(fabs(fin0(ls)), fabs(fin0(hs)));
    v[0] += big * big;
    v[2] += bigs * bigs;
    v[6] = bigs;
    if (!(lo <= hi)) v[7] = 1.0;
  }
  // sums in slots 0-4, maxima in 5-7: reduce the maxima separately
  __shared__ double mx[kTB / 64][3];
  double m5 = v[5], m6 = v[6], m7 = v[7];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m5 = fmax(m5, __shfl_down(m5, off, 64)); m6 = fmax(m6, __shfl_down(m6, off, 64)); m7 = fmax(m7, __shfl_down(m7, off, 64));
  }
  if ((threadIdx.x & 63) == 0) { mx[threadIdx.x >> 6][0] = m5; mx[threadIdx.x >> 6][1] = m6; mx[threadIdx.x >> 6][2] = m7; }
  double sums[5] = {v[0], v[1], v[2], v[3], v[4]};
  double *out = a.W.partial + ((size_t)s * a.nblk + blockIdx.x) * kNQ;
  block_partials<5>(sums, out);
  if (threadIdx.x == 0) {
    double q5 = 0, q6 = 0, q7 = 0;
    for (int w = 0; w < kTB / 64; ++w) { q5 = fmax(q5, mx[w][0]); q6 = fmax(q6, mx[w][1]); q7 = fmax(q7, mx[w][2]); }
    out[5] = q5; out[6] = q6; out[7] = q7;
  }
}

// one block per scenario: finish the init reductions, set up the control block
__global__ void k_init_control(StreamArgs a) {
  const int s = blockIdx.x;
  __shared__ double acc[8];
  if (threadIdx.x < 8) {
    double t = 0.0;
    for (int blk = 0; blk < a.nblk; ++blk) {
      const double p = a.W.partial[((size_t)s * a.nblk + blk) * kNQ + threadIdx.x];
      t = threadIdx.x < 5 ? t + p : fmax(t, p);
    }
    acc[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  StreamCtrl &c = a.W.ctrl[s];
  const dsp_options &o = a.opt;
  c = StreamCtrl{};
  c.qn = sqrt(acc[0]); c.cn = sqrt(acc[1]);
  const double qs = sqrt(acc[2]), cs = sqrt(acc[3]);
  const double qall = sqrt(acc[2] + acc[4]);
  c.c0 = a.b.obj_offset ? a.b.obj_offset[(size_t)s * a.b.obj_offset_stride] : 0.0;
  double w = (cs > 1e-10 && qs > 1e-10) ? cs / qs : 1.0;
  c.w_lo = o.weight_guard > 0.0 ? o.weight_guard * a.eta * 1.1e-16 * acc[5] / (o.eps_rel * (1.0 + qall)) : 0.0;
  c.w_hi = (o.weight_guard > 0.0 && acc[6] > 0.0) ? o.eps_rel * (1.0 + cs) / (o.weight_guard * a.eta * 1.1e-16 * acc[6]) : INFINITY;
  if (a.b.primal_weight) { const double wi = a.b.primal_weight[s]; if (wi > 0.0 && finite_d(wi)) w = wi; }
  c.w = w; c.tau = a.eta / w; c.sig = a.eta * w;
  c.r0 = INFINITY; c.rprev = INFINITY;
  c.k = 0; c.it = 0; c.status = DSP_STATUS_ITERATION_LIMIT; c.done = 0; c.mode = 0;
  if (acc[7] > 0.0) {                       // crossed bounds / NaN input: flagged, never iterated
    c.done = 1;
    c.status = (acc[0] == acc[0] && acc[1] == acc[1]) ? DSP_STATUS_PRIMAL_INFEASIBLE : DSP_STATUS_NUMERICAL;
    c.pobj = NAN;
    atomicAdd(a.W.ndone, 1);
  }
}

// ---- primal step -----------------------------------------------------------------------------------------------------
// grid: (column blocks + one block per long column, scenario groups)
template <int SG, bool WRITE_XP>
__global__ void k_primal(StreamArgs a) {
  const StreamProblem &P = a.P;
  const int b0 = blockIdx.y * SG;
  if ((int)blockIdx.x >= a.nblk_n) {
    // chunk of a long column: partial A^T y, finished by k_primal_long_finish
    const int ch = blockIdx.x - a.nblk_n;
    for (int u = 0; u < SG; ++u) {
      const int s = b0 + u;
      if (s >= a.b.B) break;
      if (a.W.ctrl[s].done) continue;
      const double part = long_dot(P.C, ch, a.W.y + (size_t)s * P.m);
      if (threadIdx.x == 0) a.W.long_partial[(size_t)s * a.nchunk_max + ch] = part;
    }
    return;
  }
  const int j = blockIdx.x * kTB + threadIdx.x;
  if (j >= P.n || P.C.is_long[j]) return;
  double val[kStreamMaxW];
  int idx[kStreamMaxW];
  for (int e = 0; e < P.C.W; ++e) { val[e] = P.C.val[(size_t)e * P.n + j]; idx[e] = P.C.idx[(size_t)e * P.n + j]; }
#pragma unroll
  for (int u = 0; u < SG; ++u) {
    const int s = b0 + u;
    if (s >= a.b.B) break;
    const StreamCtrl &c = a.W.ctrl[s];
    if (c.done) continue;
    const double *__restrict__ y = a.W.y + (size_t)s * P.m;
    double aty = 0.0;
    for (int e = 0; e < P.C.W; ++e) aty = fma(val[e], y[idx[e]], aty);
    const size_t at = (size_t)s * P.n + j;
    const double x = a.W.x[at];
    const double gx = fma(-c.tau, a.W.c[at] - aty, x);
    const double xp = clampd2(gx, a.W.lb[at], a.W.ub[at]);
    if (WRITE_XP) a.W.xp[at] = xp;
    a.W.xbar[at] = 2.0 * xp - x;
  }
}

// one thread per (long column, scenario): ordered sum of its chunk partials, then the primal update of that column
__global__ void k_primal_long_finish(StreamArgs a, int write_xp) {
  const StreamProblem &P = a.P;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.C.nlong * a.b.B) return;
  const int l = t % P.C.nlong, s = t / P.C.nlong;
  const StreamCtrl &c = a.W.ctrl[s];
  if (c.done) return;
  double aty = 0.0;
  for (int ch = P.C.long_chunk_ptr[l]; ch < P.C.long_chunk_ptr[l + 1]; ++ch) aty += a.W.long_partial[(size_t)s * a.nchunk_max + ch];
  const int j = P.C.long_id[l];
  const size_t at = (size_t)s * P.n + j;
  const double x = a.W.x[at];
  const double gx = fma(-c.tau, a.W.c[at] - aty, x);
  const double xp = clampd2(gx, a.W.lb[at], a.W.ub[at]);
  if (write_xp) a.W.xp[at] = xp;
  a.W.xbar[at] = 2.0 * xp - x;
}

// ---- dual step + Halpern averaging (plain iterations) -----------------------------------------------------------------
// thread t: row t (dual step, y averaging) and column t (x averaging); kofs = iterations since the last check
template <int SG>
__global__ void k_dual_halpern(StreamArgs a, int kofs) {
  const StreamProblem &P = a.P;
  const int b0 = blockIdx.y * SG;
  if ((int)blockIdx.x >= a.nblk) {
    const int ch = blockIdx.x - a.nblk;         // chunk of a long row: partial A xbar, finished by k_dual_long_finish
    for (int u = 0; u < SG; ++u) {
      const int s = b0 + u;
      if (s >= a.b.B) break;
      if (a.W.ctrl[s].done) continue;
      const double part = long_dot(P.R, ch, a.W.xbar + (size_t)s * P.n);
      if (threadIdx.x == 0) a.W.long_partial[(size_t)s * a.nchunk_max + ch] = part;
    }
    return;
  }
  const int t = blockIdx.x * kTB + threadIdx.x;
  const bool row = t < P.m && !P.R.is_long[t < P.m ? t : 0];
  double val[kStreamMaxW];
  int idx[kStreamMaxW];
  if (row) for (int e = 0; e < P.R.W; ++e) { val[e] = P.R.val[(size_t)e * P.m + t]; idx[e] = P.R.idx[(size_t)e * P.m + t]; }
#pragma unroll
  for (int u = 0; u < SG; ++u) {
    const int s = b0 + u;
    if (s >= a.b.B) break;
    const StreamCtrl &c = a.W.ctrl[s];
    if (c.done) continue;
    const double oml = 1.0 / (double)(c.k + kofs + 3);   // k counts this iteration: anchor weight 1 / (k + 2)
    if (row) {
      const double *__restrict__ xb = a.W.xbar + (size_t)s * P.n;
      double ax = 0.0;
      for (int e = 0; e < P.R.W; ++e) ax = fma(val[e], xb[idx[e]], ax);
      const size_t at = (size_t)s * P.m + t;
      const double y = a.W.y[at];
      const double gy = fma(-c.sig, ax, y);
      double yp = gy - clampd2(gy, -c.sig * a.W.rhi[at], -c.sig * a.W.rlo[at]);
      if (a.b.row_compliance) yp /= fma(c.sig, a.W.kap[at], 1.0);          // soft rows: proximal shrink (1 for hard rows)
      const double tt = 2.0 * yp - y;
      a.W.y[at] = fma(oml, a.W.y0[at] - tt, tt);
    }
    if (t < P.n) {
      const size_t at = (size_t)s * P.n + t;
      const double tt = a.W.xbar[at];                      // 2 x+ - x
      a.W.x[at] = fma(oml, a.W.x0[at] - tt, tt);
    }
  }
}

__global__ void k_dual_long_finish(StreamArgs a, int kofs) {
  const StreamProblem &P = a.P;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.R.nlong * a.b.B) return;
  const int l = t % P.R.nlong, s = t / P.R.nlong;
  const StreamCtrl &c = a.W.ctrl[s];
  if (c.done) return;
  double ax = 0.0;
  for (int ch = P.R.long_chunk_ptr[l]; ch < P.R.long_chunk_ptr[l + 1]; ++ch) ax += a.W.long_partial[(size_t)s * a.nchunk_max + ch];
  const size_t at = (size_t)s * P.m + P.R.long_id[l];
  const double y = a.W.y[at];
  const double gy = fma(-c.sig, ax, y);
  double yp = gy - clampd2(gy, -c.sig * a.W.rhi[at], -c.sig * a.W.rlo[at]);
  if (a.b.row_compliance) yp /= fma(c.sig, a.W.kap[at], 1.0);
  const double oml = 1.0 / (double)(c.k + kofs + 3);
  const double tt = 2.0 * yp - y;
  a.W.y[at] = fma(oml, a.W.y0[at] - tt, tt);
}

// ---- check iteration, rows: dual step WITHOUT averaging + residual and KKT row quantities -------------------------------
// partial slots (per scenario, block): 0 px = |dx|^2, 1 py = sum dy (2 (-sig A dx) + dy), 2 pres^2, 3 sum |y+| viol,
// 4 dual objective (row part, incl. -kappa y^2 / 2 of the soft rows), 5 |y+ - y0|^2, 6 |x+ - x0|^2,
// 7 primal objective of the soft rows sum (a.x - b)^2 / (2 kappa)
template <int SG>
__global__ void k_check_rows(StreamArgs a) {
  const StreamProblem &P = a.P;
  const int b0 = blockIdx.y * SG;
  const int t = blockIdx.x * kTB + threadIdx.x;
  const bool is_long_block = (int)blockIdx.x >= a.nblk;
  const bool row = !is_long_block && t < P.m && !P.R.is_long[t < P.m ? t : 0];
  double val[kStreamMaxW];
  int idx[kStreamMaxW];
  if (row) for (int e = 0; e < P.R.W; ++e) { val[e] = P.R.val[(size_t)e * P.m + t]; idx[e] = P.R.idx[(size_t)e * P.m + t]; }
  for (int u = 0; u < SG; ++u) {
    const int s = b0 + u;
    if (s >= a.b.B) break;
    const StreamCtrl &c = a.W.ctrl[s];
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!c.done) {
      const double *__restrict__ xb = a.W.xbar + (size_t)s * P.n;
      const double *__restrict__ xpv = a.W.xp + (size_t)s * P.n;
      int i = -1;
      double axb = 0.0, axp = 0.0;
      if (is_long_block) {
        const int l = blockIdx.x - a.nblk;
        const double t1 = long_dot_all(P.R, l, xb), t2 = long_dot_all(P.R, l, xpv);
        if (threadIdx.x == 0) { i = P.R.long_id[l]; axb = t1; axp = t2; }
      } else if (row) {
        i = t;
        for (int e = 0; e < P.R.W; ++e) { axb = fma(val[e], xb[idx[e]], axb); axp = fma(val[e], xpv[idx[e]], axp); }
      }
      if (i >= 0) {
        const size_t at = (size_t)s * P.m + i;
        const double y = a.W.y[at], rlo = a.W.rlo[at], rhi = a.W.rhi[at];
        const double gy = fma(-c.sig, axb, y);
        const double kp = a.b.row_compliance ? a.W.kap[at] : 0.0;
        double yp = gy - clampd2(gy, -c.sig * rhi, -c.sig * rlo);
        if (kp > 0.0) yp /= fma(c.sig, kp, 1.0);
        a.W.yp[at] = yp;
        const double dy = yp - y;
        const double nsadx = -c.sig * (axb - axp);                 // -sig A (x+ - x)   (xbar - x+ = x+ - x)
        v[1] = dy * fma(2.0, nsadx, dy);
        double viol_s = fmax(rlo - axp, 0.0) + fmax(axp - rhi, 0.0);
        v[4] = fmax(yp, 0.0) * fin0(rlo) - fmax(-yp, 0.0) * fin0(rhi);
        if (kp > 0.0) {                                              // soft row: no violation, quadratic terms of both objectives
          const double dev = axp - rlo;
          v[7] = 0.5 * dev * dev / kp;
          v[4] -= 0.5 * kp * yp * yp;
          viol_s = 0.0;
        }
        const double viol = viol_s / P.row_scale[i];
        v[2] = viol * viol;
        v[3] = fabs(yp) * viol_s;
        const double d0 = yp - a.W.y0[at];
        v[5] = d0 * d0;
      }
      if (!is_long_block && t < P.n) {
        const size_t at = (size_t)s * P.n + t;
        const double xp = a.W.xp[at];
        const double dx = xp - a.W.x[at], d0 = xp - a.W.x0[at];
        v[0] = dx * dx;
        v[6] = d0 * d0;
      }
    }
    block_partials<8>(v, a.W.partial + ((size_t)s * a.nblk_tot + blockIdx.x) * kNQ);
  }
}

// ---- check iteration, columns: reduced costs at y+ -----------------------------------------------------------------------
// partial slots: 8 dres^2, 9 pobj = c.x+, 10 dual objective (column part), 11 sum |c x|, 12 sum |dual residual| |x|
// the five column quantities of one column (slots 8 .. 12) from its reduced cost
__device__ __forceinline__ void kkt_col_terms(const StreamArgs &a, int s, int j, double aty, double (&v)[5]) {
  const StreamProblem &P = a.P;
  const size_t at = (size_t)s * P.n + j;
  const double cj = a.W.c[at], lb = a.W.lb[at], ub = a.W.ub[at], xp = a.W.xp[at];
  const double rc = cj - aty;
  const double lp = finite_d(lb) ? fmax(rc, 0.0) : 0.0;
  const double lm = finite_d(ub) ? fmax(-rc, 0.0) : 0.0;
  const double dr = (rc - lp + lm) / P.col_scale[j];
  v[0] = dr * dr;
  v[1] = cj * xp;
  v[2] = lp * fin0(lb) - lm * fin0(ub);
  v[3] = fabs(cj * xp);
  v[4] = fabs(rc - lp + lm) * fabs(xp);
}

// grid: (column blocks + one block per CHUNK of a long column, scenario groups).  A long column (the design variable of the
// price-taker LPs: 26 k entries at T = 8736) used to be walked chunk after chunk by ONE workgroup: 142 us per check whatever the
// batch - a fifth of a lone scenario's time per iteration.  Its chunks are separate blocks now, as in k_primal; k_kkt_long_finish adds
// their partial sums in chunk order (the same order: bit-identical) and writes the column's five quantities.
template <int SG>
__global__ void k_kkt_cols(StreamArgs a) {
  const StreamProblem &P = a.P;
  const int b0 = blockIdx.y * SG;
  if ((int)blockIdx.x >= a.nblk_n) {
    const int ch = blockIdx.x - a.nblk_n;
    for (int u = 0; u < SG; ++u) {
      const int s = b0 + u;
      if (s >= a.b.B) break;
      if (a.W.ctrl[s].done) continue;
      const double part = long_dot(P.C, ch, a.W.yp + (size_t)s * P.m);
      if (threadIdx.x == 0) a.W.long_partial[(size_t)s * a.nchunk_max + ch] = part;
    }
    return;
  }
  const int t = blockIdx.x * kTB + threadIdx.x;
  const bool col = t < P.n && !P.C.is_long[t < P.n ? t : 0];
  double val[kStreamMaxW];
  int idx[kStreamMaxW];
  if (col) for (int e = 0; e < P.C.W; ++e) { val[e] = P.C.val[(size_t)e * P.n + t]; idx[e] = P.C.idx[(size_t)e * P.n + t]; }
  for (int u = 0; u < SG; ++u) {
    const int s = b0 + u;
    if (s >= a.b.B) break;
    const StreamCtrl &c = a.W.ctrl[s];
    double v[5] = {0, 0, 0, 0, 0};
    if (!c.done && col) {
      const double *__restrict__ ypv = a.W.yp + (size_t)s * P.m;
      double aty = 0.0;
      for (int e = 0; e < P.C.W; ++e) aty = fma(val[e], ypv[idx[e]], aty);
      kkt_col_terms(a, s, t, aty, v);
    }
    block_partials<5>(v, a.W.partial + ((size_t)s * a.nblk_tot + blockIdx.x) * kNQ + 8);
  }
}

// one thread per (long column, scenario): ordered sum of the chunk partials of A^T y+, then the column's quantities into the partial
// slot of block nblk_n + l (the slot the one-block-per-long-column form wrote; k_control adds the blocks' slots in order)
__global__ void k_kkt_long_finish(StreamArgs a) {
  const StreamProblem &P = a.P;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.C.nlong * a.b.B) return;
  const int l = t % P.C.nlong, s = t / P.C.nlong;
  if (a.W.ctrl[s].done) return;
  double aty = 0.0;
  for (int ch = P.C.long_chunk_ptr[l]; ch < P.C.long_chunk_ptr[l + 1]; ++ch) aty += a.W.long_partial[(size_t)s * a.nchunk_max + ch];
  double v[5];
  kkt_col_terms(a, s, P.C.long_id[l], aty, v);
  double *out = a.W.partial + ((size_t)s * a.nblk_tot + a.nblk_n + l) * kNQ + 8;
#pragma unroll
  for (int q = 0; q < 5; ++q) out[q] = v[q];
}

// one block per scenario sums the check partials in a fixed order and decides
__global__ void k_control(StreamArgs a, int iters_this_period) {
  const int s = blockIdx.x;
  StreamCtrl &c = a.W.ctrl[s];
  if (c.done) return;
  // 256 threads: quantity q = t % 16, the blocks' partials in 16 interleaved strands (strand t / 16), strands added in order.  (16
  // threads walking through all the blocks one load after the other took 80 us per check at T = 8736: profiles/r30a_stream_kernel_stats.csv)
  __shared__ double strand[16][kNQ];
  __shared__ double acc[kNQ];
  {
    const int q = threadIdx.x & (kNQ - 1), p = threadIdx.x >> 4;
    double t = 0.0;
    for (int blk = p; blk < a.nblk_tot; blk += 16) t += a.W.partial[((size_t)s * a.nblk_tot + blk) * kNQ + q];
    strand[p][q] = t;
  }
  __syncthreads();
  if (threadIdx.x < kNQ) {
    double t = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += strand[p][threadIdx.x];
    acc[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  control_decide(acc, c, a.opt, a.eta, iters_this_period);
  if (c.done) atomicAdd(a.W.ndone, 1);
  else if (c.suspect) a.W.ndone[1] = 1;
}

// ---- apply: Halpern step or restart after a check ---------------------------------------------------------------------------
template <int SG>
__global__ void k_apply(StreamArgs a) {
  const StreamProblem &P = a.P;
  const int b0 = blockIdx.y * SG;
  const int t = blockIdx.x * kTB + threadIdx.x;
#pragma unroll
  for (int u = 0; u < SG; ++u) {
    const int s = b0 + u;
    if (s >= a.b.B) break;
    const StreamCtrl &c = a.W.ctrl[s];
    if (c.done) continue;
    const double oml = 1.0 / (double)(c.k + 2);
    if (t < P.n) {
      const size_t at = (size_t)s * P.n + t;
      const double xp = a.W.xp[at];
      if (c.mode == 1) { a.W.x[at] = xp; a.W.x0[at] = xp; }
      else { const double x = a.W.x[at]; const double tt = 2.0 * xp - x; a.W.x[at] = fma(oml, a.W.x0[at] - tt, tt); }
    }
    if (t < P.m) {
      const size_t at = (size_t)s * P.m + t;
      const double yp = a.W.yp[at];
      if (c.mode == 1) { a.W.y[at] = yp; a.W.y0[at] = yp; }
      else { const double y = a.W.y[at]; const double tt = 2.0 * yp - y; a.W.y[at] = fma(oml, a.W.y0[at] - tt, tt); }
    }
  }
}

// ---- infeasibility / unboundedness certificates (dsp_options::eps_infeasible) on the scenario-major workspace ------------------------
// Same certificates as the fused kernels (dsp_kernels.hip): the dual part dy of the displacement T(z) - z, with the signs its rows
// cannot take removed, as a FARKAS RAY; the primal part dx, clipped to the recession cone of the column bounds, as a direction of
// unbounded descent.  Run by the host for the scenarios control_decide marked suspect (StreamCtrl::suspect), after k_primal<.., true>
// and k_check_rows have put T(z) into xp / yp:
//   k_ray_prep   d -> xbar, cleaned dy -> ray                    (element-wise)
//   k_ray_rows   A d (ELL rows + long rows), the rows' share of the ray's bound value        -> partial slots 1 .. 3
//   k_ray_cols   A^T dy (ELL columns; long columns: chunk partials + k_ray_long_finish)       -> partial slots 4 .. 8
//   k_ray_decide sums, the two tests, status 2 / 3
template <int SG>
__global__ void k_ray_prep(StreamArgs a) {
  const StreamProblem &P = a.P;
  const int b0 = blockIdx.y * SG;
  const int t = blockIdx.x * kTB + threadIdx.x;
  for (int u = 0; u < SG; ++u) {
    const int s = b0 + u;
    if (s >= a.b.B) break;
    const StreamCtrl &c = a.W.ctrl[s];
    if (c.done || !c.suspect) continue;
    if (t < P.n) {
      const size_t at = (size_t)s * P.n + t;
      const double dx = a.W.xp[at] - a.W.x[at];
      const bool fl = finite_d(a.W.lb[at]), fu = finite_d(a.W.ub[at]);
      a.W.xbar[at] = (fl && fu) ? 0.0 : fl ? fmax(dx, 0.0) : fu ? fmin(dx, 0.0) : dx;
    }
    if (t < P.m) {
      const size_t at = (size_t)s * P.m + t;
      const double dy = a.W.yp[at] - a.W.y[at];
      double yc = (finite_d(a.W.rlo[at]) ? fmax(dy, 0.0) : 0.0) - (finite_d(a.W.rhi[at]) ? fmax(-dy, 0.0) : 0.0);
      if (a.b.row_compliance && a.W.kap[at] > 0.0) yc = 0.0;           // soft rows admit no multiplier ray
      a.W.ray[at] = yc;
    }
  }
}

// grid: (row blocks + one block per long row, scenario groups)
template <int SG>
__global__ void k_ray_rows(StreamArgs a) {
  const StreamProblem &P = a.P;
  const int b0 = blockIdx.y * SG;
  const int t = blockIdx.x * kTB + threadIdx.x;
  const bool is_long_block = (int)blockIdx.x >= a.nblk;
  const bool row = !is_long_block && t < P.m && !P.R.is_long[t < P.m ? t : 0];
  for (int u = 0; u < SG; ++u) {
    const int s = b0 + u;
    if (s >= a.b.B) break;
    const StreamCtrl &c = a.W.ctrl[s];
    double v[3] = {0, 0, 0};                       // 1 bound value of the ray (rows), 2 |bounds|^2 (rows), 3 |recession violation of A d|^2
    if (!c.done && c.suspect) {
      const double *__restrict__ d = a.W.xbar + (size_t)s * P.n;
      int i = -1;
      double ad = 0.0;
      if (is_long_block) {
        const int l = blockIdx.x - a.nblk;
        const double t1 = long_dot_all(P.R, l, d);
        if (threadIdx.x == 0) { i = P.R.long_id[l]; ad = t1; }
      } else if (row) {
        i = t;
        ad = ell_dot(P.R, t, d);
      }
      if (i >= 0) {
        const size_t at = (size_t)s * P.m + i;
        const double yc = a.W.ray[at], rlo = a.W.rlo[at], rhi = a.W.rhi[at];
        v[0] = fmax(yc, 0.0) * fin0(rlo) - fmax(-yc, 0.0) * fin0(rhi);
        const double big = fmax(fabs(fin0(rlo)), fabs(fin0(rhi)));
        v[1] = big * big;
        const double rv = (finite_d(rlo) ? fmax(-ad, 0.0) : 0.0) + (finite_d(rhi) ? fmax(ad, 0.0) : 0.0);
        v[2] = rv * rv;
      }
    }
    block_partials<3>(v, a.W.partial + ((size_t)s * a.nblk_tot + blockIdx.x) * kNQ + 1);
  }
}

// the five column quantities (slots 4 .. 8): c.d, |c|^2, bound value of the ray (columns), |bounds|^2 (columns), |dual residual of the ray|^2
__device__ __forceinline__ void ray_col_terms(const StreamArgs &a, int s, int j, double aty, double (&v)[5]) {
  const size_t at = (size_t)s * a.P.n + j;
  const double cj = a.W.c[at], lb = a.W.lb[at], ub = a.W.ub[at], d = a.W.xbar[at];
  const double rc = -aty;
  const double lp = finite_d(lb) ? fmax(rc, 0.0) : 0.0;
  const double lm = finite_d(ub) ? fmax(-rc, 0.0) : 0.0;
  const double res = rc - lp + lm;
  v[0] = cj * d;
  v[1] = cj * cj;
  v[2] = lp * fin0(lb) - lm * fin0(ub);
  v[3] = fin0(lb) * fin0(lb) + fin0(ub) * fin0(ub);
  v[4] = res * res;
}

// grid: (column blocks + one block per CHUNK of a long column, scenario groups), as k_kkt_cols
template <int SG>
__global__ void k_ray_cols(StreamArgs a) {
  const StreamProblem &P = a.P;
  const int b0 = blockIdx.y * SG;
  if ((int)blockIdx.x >= a.nblk_n) {
    const int ch = blockIdx.x - a.nblk_n;
    for (int u = 0; u < SG; ++u) {
      const int s = b0 + u;
      if (s >= a.b.B) break;
      if (a.W.ctrl[s].done || !a.W.ctrl[s].suspect) continue;
      const double part = long_dot(P.C, ch, a.W.ray + (size_t)s * P.m);
      if (threadIdx.x == 0) a.W.long_partial[(size_t)s * a.nchunk_max + ch] = part;
    }
    return;
  }
  const int t = blockIdx.x * kTB + threadIdx.x;
  const bool col = t < P.n && !P.C.is_long[t < P.n ? t : 0];
  for (int u = 0; u < SG; ++u) {
    const int s = b0 + u;
    if (s >= a.b.B) break;
    const StreamCtrl &c = a.W.ctrl[s];
    double v[5] = {0, 0, 0, 0, 0};
    if (!c.done && c.suspect && col) ray_col_terms(a, s, t, ell_dot(P.C, t, a.W.ray + (size_t)s * P.m), v);
    block_partials<5>(v, a.W.partial + ((size_t)s * a.nblk_tot + blockIdx.x) * kNQ + 4);
  }
}

__global__ void k_ray_long_finish(StreamArgs a) {
  const StreamProblem &P = a.P;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.C.nlong * a.b.B) return;
  const int l = t % P.C.nlong, s = t / P.C.nlong;
  double v[5] = {0, 0, 0, 0, 0};
  if (!a.W.ctrl[s].done && a.W.ctrl[s].suspect) {
    double aty = 0.0;
    for (int ch = P.C.long_chunk_ptr[l]; ch < P.C.long_chunk_ptr[l + 1]; ++ch) aty += a.W.long_partial[(size_t)s * a.nchunk_max + ch];
    ray_col_terms(a, s, P.C.long_id[l], aty, v);
  }
  double *out = a.W.partial + ((size_t)s * a.nblk_tot + a.nblk_n + l) * kNQ + 4;
#pragma unroll
  for (int q = 0; q < 5; ++q) out[q] = v[q];
}

// the two tests on the sums (slot q of `acc`: see the kernels above); returns the status (0 = no certificate)
__device__ inline int ray_verdict(const double *acc, double eps_infeasible) {
  const double bound_value = acc[1] + acc[6], bounds2 = acc[2] + acc[7];
  if (bound_value > 0.0 && sqrt(acc[8]) * (1.0 + sqrt(bounds2)) <= eps_infeasible * bound_value) return DSP_STATUS_PRIMAL_INFEASIBLE;
  if (acc[4] < 0.0 && sqrt(acc[3]) * (1.0 + sqrt(acc[5])) <= eps_infeasible * -acc[4]) return DSP_STATUS_DUAL_INFEASIBLE;
  return 0;
}

__global__ void k_ray_decide(StreamArgs a) {
  const int s = blockIdx.x;
  StreamCtrl &c = a.W.ctrl[s];
  if (c.done || !c.suspect) return;
  __shared__ double strand[16][kNQ];
  __shared__ double acc[kNQ];
  {
    const int q = threadIdx.x & (kNQ - 1), p = threadIdx.x >> 4;
    double t = 0.0;
    if (q >= 1 && q <= 8) for (int blk = p; blk < a.nblk_tot; blk += 16) t += a.W.partial[((size_t)s * a.nblk_tot + blk) * kNQ + q];
    strand[p][q] = t;
  }
  __syncthreads();
  if (threadIdx.x < kNQ) {
    double t = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += strand[p][threadIdx.x];
    acc[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const int verdict = ray_verdict(acc, a.opt.eps_infeasible);
  if (verdict) { c.status = verdict; c.done = 1; atomicAdd(a.W.ndone, 1); }
}

// ---- block-resident solve for MID-SIZE LPs: one workgroup per scenario, the whole state in LDS --------------------------------
// LPs just beyond the fused kernels (coupled stochastic bidders: 582 x 408; a one-week price-taker design LP: 1011 x 1010)
// are far too small for the launch-per-half-step form above (12-16 us per iteration, all launch latency).  When
// 8 (7 n + 6 m) bytes fit the CU's LDS, ONE launch runs the whole solve: workgroup s owns scenario s, every per-element
// vector (x, anchor, c, bounds, x+, xbar; y, anchor, row bounds, y+) lives in LDS, the matrix is read from L2 each half
// step (shared by all scenarios, coalesced), a half step ends in one __syncthreads, the check reductions are block
// reductions, and thread 0 takes the same restart / termination decision (control_decide).  Same iterates as the
// launch-per-step form up to the summation order of the check sums.
__device__ __forceinline__ void block_sum(double *v, int nq, double *red /* [kTB/64 .. waves][kNQ] */, double *acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int q = 0; q < nq; ++q) {
    double t = v[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if (lane == 0) red[wave * kNQ + q] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < nq) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[w * kNQ + threadIdx.x];
    acc[threadIdx.x] = t;
  }
  __syncthreads();
}

// sum over a long vector's entries against an LDS-resident vector, by the whole block; every thread gets the total
__device__ __forceinline__ double block_long_dot(const StreamMatrix &M, int l, const double *vec, double *red, double *acc) {
  double part[1] = {0.0};
  for (int p = M.long_ptr[l] + threadIdx.x; p < M.long_ptr[l + 1]; p += blockDim.x) part[0] = fma(M.long_val[p], vec[M.long_idx[p]], part[0]);
  block_sum(part, 1, red, acc);
  const double t = acc[0];
  __syncthreads();
  return t;
}

__global__ void __launch_bounds__(1024) k_block_solve(StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const StreamProblem &P = a.P;
  const int s = blockIdx.x, n = P.n, m = P.m, NT = blockDim.x, t0 = threadIdx.x;
  double *x = lds, *x0 = x + n, *cc = x0 + n, *lb = cc + n, *ub = lb + n, *xp = ub + n, *xb = xp + n;
  double *y = xb + n, *y0 = y + m, *rlo = y0 + m, *rhi = rlo + m, *yp = rhi + m, *axp = yp + m;
  double *red = axp + m, *acc = red + (1024 / 64) * kNQ;
  __shared__ StreamCtrl c;
  __shared__ int mode_s;
  const double *kapg = a.b.row_compliance ? a.W.kap + (size_t)s * m : nullptr;     // soft rows (convex QP), else LP
  if (t0 == 0) c = a.W.ctrl[s];
  for (int j = t0; j < n; j += NT) {
    const size_t at = (size_t)s * n + j;
    x[j] = a.W.x[at]; x0[j] = a.W.x0[at]; cc[j] = a.W.c[at]; lb[j] = a.W.lb[at]; ub[j] = a.W.ub[at]; xp[j] = x[j]; xb[j] = x[j];
  }
  for (int i = t0; i < m; i += NT) {
    const size_t at = (size_t)s * m + i;
    y[i] = a.W.y[at]; y0[i] = a.W.y0[at]; rlo[i] = a.W.rlo[at]; rhi[i] = a.W.rhi[at]; yp[i] = y[i];
  }
  __syncthreads();
  const int C = a.opt.check_every > 0 ? a.opt.check_every : 64;
  while (!c.done) {
    for (int u = 0; u < C; ++u) {
      const bool check = u == C - 1;
      const double tau = c.tau, sig = c.sig;
      // ---- primal step ------------------------------------------------------------------------------------------
      for (int j = t0; j < n; j += NT) {
        if (P.C.is_long[j]) continue;
        double aty = 0.0;
        for (int e = 0; e < P.C.W; ++e) aty = fma(P.C.val[(size_t)e * n + j], y[P.C.idx[(size_t)e * n + j]], aty);
        const double gx = fma(-tau, cc[j] - aty, x[j]);
        const double v = clampd2(gx, lb[j], ub[j]);
        xp[j] = v; xb[j] = 2.0 * v - x[j];
      }
      for (int l = 0; l < P.C.nlong; ++l) {
        const double aty = block_long_dot(P.C, l, y, red, acc);
        if (t0 == 0) {
          const int j = P.C.long_id[l];
          const double gx = fma(-tau, cc[j] - aty, x[j]);
          const double v = clampd2(gx, lb[j], ub[j]);
          xp[j] = v; xb[j] = 2.0 * v - x[j];
        }
      }
      __syncthreads();
      // ---- dual step (+ Halpern averaging on plain iterations) ----------------------------------------------------
      const double oml = 1.0 / (double)(c.k + u + 3);
      for (int i = t0; i < m; i += NT) {
        if (P.R.is_long[i]) continue;
        double ax = 0.0, ax2 = 0.0;
        for (int e = 0; e < P.R.W; ++e) {
          const double v = P.R.val[(size_t)e * m + i];
          const int id = P.R.idx[(size_t)e * m + i];
          ax = fma(v, xb[id], ax);
          if (check) ax2 = fma(v, xp[id], ax2);
        }
        const double gy = fma(-sig, ax, y[i]);
        double v = gy - clampd2(gy, -sig * rhi[i], -sig * rlo[i]);
        if (kapg) v /= fma(sig, kapg[i], 1.0);      // soft rows (compliance read from the workspace: L2-resident)
        yp[i] = v;
        if (check) axp[i] = ax2;                   // A x+  (A xbar - A x+ = A (x+ - x))
        else { const double tt = 2.0 * v - y[i]; y[i] = fma(oml, y0[i] - tt, tt); }
      }
      for (int l = 0; l < P.R.nlong; ++l) {
        const double ax = block_long_dot(P.R, l, xb, red, acc);
        const double ax2 = check ? block_long_dot(P.R, l, xp, red, acc) : 0.0;
        if (t0 == 0) {
          const int i = P.R.long_id[l];
          const double gy = fma(-sig, ax, y[i]);
          double v = gy - clampd2(gy, -sig * rhi[i], -sig * rlo[i]);
          if (kapg) v /= fma(sig, kapg[i], 1.0);
          yp[i] = v;
          if (check) axp[i] = ax2;
          else { const double tt = 2.0 * v - y[i]; y[i] = fma(oml, y0[i] - tt, tt); }
        }
      }
      if (!check) {
        for (int j = t0; j < n; j += NT) { const double tt = xb[j]; x[j] = fma(oml, x0[j] - tt, tt); }
        __syncthreads();
        continue;
      }
      __syncthreads();
      // ---- check: residual + KKT sums (slot layout of k_check_rows / k_kkt_cols), decision, apply ------------------
      double v[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = t0; i < m; i += NT) {
        double axb = 0.0;                          // A xbar again (not kept by the dual step): A xbar - A x+ = A (x+ - x)
        for (int e = 0; e < P.R.W; ++e) axb = fma(P.R.val[(size_t)e * m + i], xb[P.R.idx[(size_t)e * m + i]], axb);
        const double dy = yp[i] - y[i];
        const double nsadx = -sig * (axb - axp[i]);
        v[1] += dy * fma(2.0, nsadx, dy);
        double viol_s = fmax(rlo[i] - axp[i], 0.0) + fmax(axp[i] - rhi[i], 0.0);
        v[4] += fmax(yp[i], 0.0) * fin0(rlo[i]) - fmax(-yp[i], 0.0) * fin0(rhi[i]);
        if (kapg && kapg[i] > 0.0) {
          const double dev = axp[i] - rlo[i];
          v[7] += 0.5 * dev * dev / kapg[i];
          v[4] -= 0.5 * kapg[i] * yp[i] * yp[i];
          viol_s = 0.0;
        }
        const double viol = viol_s / P.row_scale[i];
        v[2] += viol * viol;
        v[3] += fabs(yp[i]) * viol_s;
        const double d0 = yp[i] - y0[i];
        v[5] += d0 * d0;
      }
      for (int j = t0; j < n; j += NT) {
        const double dx = xp[j] - x[j], d0 = xp[j] - x0[j];
        v[0] += dx * dx;
        v[6] += d0 * d0;
        if (P.C.is_long[j]) continue;
        double aty = 0.0;
        for (int e = 0; e < P.C.W; ++e) aty = fma(P.C.val[(size_t)e * n + j], yp[P.C.idx[(size_t)e * n + j]], aty);
        const double rc = cc[j] - aty;
        const double lp = finite_d(lb[j]) ? fmax(rc, 0.0) : 0.0, lm = finite_d(ub[j]) ? fmax(-rc, 0.0) : 0.0;
        const double dr = (rc - lp + lm) / P.col_scale[j];
        v[8] += dr * dr;
        v[9] += cc[j] * xp[j];
        v[10] += lp * fin0(lb[j]) - lm * fin0(ub[j]);
        v[11] += fabs(cc[j] * xp[j]);
        v[12] += fabs(rc - lp + lm) * fabs(xp[j]);
      }
      for (int l = 0; l < P.C.nlong; ++l) {
        const double aty = block_long_dot(P.C, l, yp, red, acc);
        if (t0 == 0) {
          const int j = P.C.long_id[l];
          const double rc = cc[j] - aty;
          const double lp = finite_d(lb[j]) ? fmax(rc, 0.0) : 0.0, lm = finite_d(ub[j]) ? fmax(-rc, 0.0) : 0.0;
          const double dr = (rc - lp + lm) / P.col_scale[j];
          v[8] += dr * dr; v[9] += cc[j] * xp[j]; v[10] += lp * fin0(lb[j]) - lm * fin0(ub[j]);
          v[11] += fabs(cc[j] * xp[j]); v[12] += fabs(rc - lp + lm) * fabs(xp[j]);
        }
      }
      block_sum(v, 13, red, acc);
      if (t0 == 0) mode_s = control_decide(acc, c, a.opt, a.eta, C);
      __syncthreads();
      if (c.done) break;
      if (c.suspect) {
        // certificates on the displacement (x+ - x, y+ - y) of this check: the same sums as k_ray_rows / k_ray_cols, state in LDS
        // (xb <- clipped dx, axp <- cleaned dy: both are scratch between two iterations)
        for (int j = t0; j < n; j += NT) {
          const double dx = xp[j] - x[j];
          const bool fl = finite_d(lb[j]), fu = finite_d(ub[j]);
          xb[j] = (fl && fu) ? 0.0 : fl ? fmax(dx, 0.0) : fu ? fmin(dx, 0.0) : dx;
        }
        for (int i = t0; i < m; i += NT) {
          const double dy = yp[i] - y[i];
          double yc = (finite_d(rlo[i]) ? fmax(dy, 0.0) : 0.0) - (finite_d(rhi[i]) ? fmax(-dy, 0.0) : 0.0);
          if (kapg && kapg[i] > 0.0) yc = 0.0;
          axp[i] = yc;
        }
        __syncthreads();
        double rv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = t0; i < m; i += NT) {             // (no long rows in this form)
          double ad = 0.0;
          for (int e = 0; e < P.R.W; ++e) ad = fma(P.R.val[(size_t)e * m + i], xb[P.R.idx[(size_t)e * m + i]], ad);
          rv[1] += fmax(axp[i], 0.0) * fin0(rlo[i]) - fmax(-axp[i], 0.0) * fin0(rhi[i]);
          const double big = fmax(fabs(fin0(rlo[i])), fabs(fin0(rhi[i])));
          rv[2] += big * big;
          const double r_ = (finite_d(rlo[i]) ? fmax(-ad, 0.0) : 0.0) + (finite_d(rhi[i]) ? fmax(ad, 0.0) : 0.0);
          rv[3] += r_ * r_;
        }
        auto col_terms = [&](int j, double aty) {
          const double rc = -aty;
          const double lp = finite_d(lb[j]) ? fmax(rc, 0.0) : 0.0, lm = finite_d(ub[j]) ? fmax(-rc, 0.0) : 0.0;
          const double res = rc - lp + lm;
          rv[4] += cc[j] * xb[j]; rv[5] += cc[j] * cc[j]; rv[6] += lp * fin0(lb[j]) - lm * fin0(ub[j]);
          rv[7] += fin0(lb[j]) * fin0(lb[j]) + fin0(ub[j]) * fin0(ub[j]); rv[8] += res * res;
        };
        for (int j = t0; j < n; j += NT) {
          if (P.C.is_long[j]) continue;
          double aty = 0.0;
          for (int e = 0; e < P.C.W; ++e) aty = fma(P.C.val[(size_t)e * n + j], axp[P.C.idx[(size_t)e * n + j]], aty);
          col_terms(j, aty);
        }
        for (int l = 0; l < P.C.nlong; ++l) {
          const double aty = block_long_dot(P.C, l, axp, red, acc);
          if (t0 == 0) col_terms(P.C.long_id[l], aty);
        }
        block_sum(rv, 9, red, acc);
        if (t0 == 0) {
          const int verdict = ray_verdict(acc, a.opt.eps_infeasible);
          if (verdict) { c.status = verdict; c.done = 1; atomicAdd(a.W.ndone, 1); }
        }
        __syncthreads();
        if (c.done) break;
      }
      const double om2 = 1.0 / (double)(c.k + 2);
      if (mode_s == 1) {
        for (int j = t0; j < n; j += NT) { x[j] = xp[j]; x0[j] = xp[j]; }
        for (int i = t0; i < m; i += NT) { y[i] = yp[i]; y0[i] = yp[i]; }
      } else {
        for (int j = t0; j < n; j += NT) { const double tt = 2.0 * xp[j] - x[j]; x[j] = fma(om2, x0[j] - tt, tt); }
        for (int i = t0; i < m; i += NT) { const double tt = 2.0 * yp[i] - y[i]; y[i] = fma(om2, y0[i] - tt, tt); }
      }
      __syncthreads();
    }
  }
  // ---- results back to the workspace (k_finalize unscales them) -------------------------------------------------------
  for (int j = t0; j < n; j += NT) a.W.xp[(size_t)s * n + j] = xp[j];
  for (int i = t0; i < m; i += NT) a.W.yp[(size_t)s * m + i] = yp[i];
  if (t0 == 0) a.W.ctrl[s] = c;
}


// ---- fused iteration for banded LPs: ONE launch per plain iteration (FusedPlan, dsp_stream.hpp) --------------------------------
// grid (scenario groups, tiles); workgroup = one tile x SG scenarios.  Per scenario the launch moves
//     reads  x, x0, c (+ lb, ub unless the batch shares them), y, y0 (+ rlo, rhi unless shared)      writes  x, y
// = 4 n + 3 m doubles with shared bounds (6 n + 5 m otherwise) against the 8 n + 6 m of the two-launch form - and the gathers
// of both products are LDS reads instead of L2 traffic (the two-launch kernels fetched 1.3-1.6x their algorithmic bytes:
// profiles/r30a_stream_pmc_summary.csv).  x / y are double buffered (a neighbouring tile reads this tile's halo while it
// writes), and so are the long columns' partial sums.  Every tile computes the halo columns' primal step with the same
// arithmetic in the same order as their owner, so the iterates are those of the two-launch form (bit for bit when there is no
// long column; a long column's A^T y is summed per tile here and per 2048-entry chunk there).
struct FusedIO {
  const double *x_in, *y_in;
  double *x_out, *y_out;
  const double *lp_in;         // [B][nlong][ntile]
  double *lp_out;
  int xcd_full;                // k_fused_pre: workgroups [0, xcd_full) are renumbered XCD-major (0 = grid order)
};

__device__ __forceinline__ void fused_reduce_long(double (&v)[kFusedMaxLong], int nlong, double *red) {
  // block sum (4 waves) of up to kFusedMaxLong quantities, fixed order; result in red[0 .. nlong)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < kFusedMaxLong; ++q) {
    double t = v[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if (lane == 0) red[kFusedMaxLong + wave * kFusedMaxLong + q] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < nlong) {
    double t = 0.0;
    for (int w = 0; w < kTB / 64; ++w) t += red[kFusedMaxLong + w * kFusedMaxLong + threadIdx.x];
    red[threadIdx.x] = t;
  }
  __syncthreads();
}

template <int SG, bool SHARED, bool QP>
__global__ void __launch_bounds__(kTB) k_fused(StreamArgs a, FusedIO io, int kofs) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const StreamProblem &P = a.P;
  const FusedPlan &F = P.F;
  const int n = P.n, m = P.m, nlong = P.C.nlong, NY = F.ny_max, NXB = F.nxb_max;
  const int tile = blockIdx.y, b0 = blockIdx.x * SG;           // grid (scenario groups, tiles): see k_fused_pre
  const int32_t *tp = F.tile + 8 * tile;
  const int i0 = tp[0], i1 = tp[1], j0 = tp[2], j1 = tp[3], c_lo = tp[4], c_hi = tp[5], r_lo = tp[6], r_hi = tp[7];
  double *ys = lds;                          // [SG][NY]   y of the rows r_lo .. r_hi
  double *xb = lds + (size_t)SG * NY;        // [SG][NXB]  xbar: long columns first, then the columns c_lo .. c_hi
  double *red = xb + (size_t)SG * NXB;       // [(1 + waves) * kFusedMaxLong]
  bool act[SG];
  double tau[SG], sig[SG], oml[SG];
  bool any = false;
#pragma unroll
  for (int u = 0; u < SG; ++u) {
    const int s = b0 + u;
    act[u] = s < a.b.B && !a.W.ctrl[s < a.b.B ? s : 0].done;
    const StreamCtrl &c = a.W.ctrl[s < a.b.B ? s : 0];
    tau[u] = c.tau; sig[u] = c.sig; oml[u] = 1.0 / (double)(c.k + kofs + 3);
    any |= act[u];
  }
  if (!any) return;
  // ---- stage 0: y of the staged rows; the long columns' primal step from the partial sums the previous launch left ----------
  for (int r = threadIdx.x; r < r_hi - r_lo; r += kTB) {
#pragma unroll
    for (int u = 0; u < SG; ++u) if (act[u]) ys[u * NY + r] = io.y_in[(size_t)(b0 + u) * m + r_lo + r];
  }
  // (one wave per (scenario, long column): the ntile partial sums are loaded by the lanes side by side and reduced in a fixed
  // order - a single thread adding them one after the other was a chain of ntile L2 round trips at the head of EVERY workgroup:
  // 115 us per launch at 69 tiles, 155 us at 137, profiles/r30b_fused_scan.log)
  for (int q = threadIdx.x >> 6; q < nlong * SG; q += kTB / 64) {
    const int l = q % nlong, u = q / nlong, s = b0 + u, lane = threadIdx.x & 63;
    if (!act[u]) continue;
    const double *pp = io.lp_in + ((size_t)s * nlong + l) * F.ntile;
    double aty = 0.0;
    for (int t = lane; t < F.ntile; t += 64) aty += pp[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) aty += __shfl_down(aty, off, 64);
    if (lane == 0) {
      const int j = P.C.long_id[l];
      const size_t at = (size_t)s * n + j, ab = SHARED ? (size_t)j : at;
      const double x = io.x_in[at];
      const double xp = clampd2(fma(-tau[u], a.W.c[at] - aty, x), a.W.lb[ab], a.W.ub[ab]);
      xb[u * NXB + l] = 2.0 * xp - x;
    }
  }
  __syncthreads();
  // ---- stage 1: primal step of the columns c_lo .. c_hi (own + halo) into LDS ---------------------------------------------
  for (int j = c_lo + threadIdx.x; j < c_hi; j += kTB) {
    if (P.C.is_long[j]) continue;
    double val[kStreamMaxW];
    int idx[kStreamMaxW];
#pragma unroll
    for (int e = 0; e < kStreamMaxW; ++e)
      if (e < P.C.W) {
        val[e] = P.C.val[(size_t)e * n + j];
        const int gi = P.C.idx[(size_t)e * n + j];
        idx[e] = (gi < r_lo || gi >= r_hi) ? 0 : gi - r_lo;       // padding entries (val 0, index 0): any staged slot will do
      } else { val[e] = 0.0; idx[e] = 0; }
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      if (!act[u]) continue;
      const double *yv = ys + u * NY;
      double aty = 0.0;
#pragma unroll
      for (int e = 0; e < kStreamMaxW; ++e) if (e < P.C.W) aty = fma(val[e], yv[idx[e]], aty);
      const size_t at = (size_t)(b0 + u) * n + j, ab = SHARED ? (size_t)j : at;
      const double x = io.x_in[at];
      const double xp = clampd2(fma(-tau[u], a.W.c[at] - aty, x), a.W.lb[ab], a.W.ub[ab]);
      xb[u * NXB + nlong + (j - c_lo)] = 2.0 * xp - x;
    }
  }
  __syncthreads();
  // ---- stage 2: dual step + Halpern averaging of the own rows; their contribution to A^T y of the long columns ---------------
  double lp[SG][kFusedMaxLong];
#pragma unroll
  for (int u = 0; u < SG; ++u)
#pragma unroll
    for (int q = 0; q < kFusedMaxLong; ++q) lp[u][q] = 0.0;
  for (int i = i0 + threadIdx.x; i < i1; i += kTB) {
    double val[kStreamMaxW];
    int slot[kStreamMaxW];
    bool touches_long = false;
#pragma unroll
    for (int e = 0; e < kStreamMaxW; ++e) {
      if (e < P.R.W) {
        val[e] = P.R.val[(size_t)e * m + i];
        const int gi = F.ridx_enc[(size_t)e * m + i];
        slot[e] = gi < 0 ? -1 - gi : nlong + gi - c_lo;          // a padding entry (val 0, index 0) may point below c_lo:
        if (gi >= 0 && (gi < c_lo || gi >= c_hi)) slot[e] = 0;  // it multiplies by 0, any valid slot will do
        touches_long |= gi < 0;
      } else { val[e] = 0.0; slot[e] = 0; }
    }
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      if (!act[u]) continue;
      const double *xv = xb + u * NXB;
      double ax = 0.0;
#pragma unroll
      for (int e = 0; e < kStreamMaxW; ++e) if (e < P.R.W) ax = fma(val[e], xv[slot[e]], ax);
      const size_t at = (size_t)(b0 + u) * m + i, ab = SHARED ? (size_t)i : at;
      const double y = ys[u * NY + (i - r_lo)];
      const double gy = fma(-sig[u], ax, y);
      double yp = gy - clampd2(gy, -sig[u] * a.W.rhi[ab], -sig[u] * a.W.rlo[ab]);
      if (QP) yp /= fma(sig[u], a.W.kap[at], 1.0);
      const double tt = 2.0 * yp - y;
      const double yn = fma(oml[u], a.W.y0[at] - tt, tt);
      io.y_out[at] = yn;
      if (touches_long) {
#pragma unroll
        for (int e = 0; e < kStreamMaxW; ++e) {
          if (e < P.R.W && slot[e] < nlong && val[e] != 0.0) {
#pragma unroll
            for (int q = 0; q < kFusedMaxLong; ++q) lp[u][q] += (slot[e] == q) ? val[e] * yn : 0.0;
          }
        }
      }
    }
  }
  // ---- stage 3: Halpern averaging of the own columns -----------------------------------------------------------------------
  for (int j = j0 + threadIdx.x; j < j1; j += kTB) {
    if (P.C.is_long[j]) continue;
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      if (!act[u]) continue;
      const size_t at = (size_t)(b0 + u) * n + j;
      const double tt = xb[u * NXB + nlong + (j - c_lo)];
      io.x_out[at] = fma(oml[u], a.W.x0[at] - tt, tt);
    }
  }
  if ((int)threadIdx.x < nlong * SG) {
    const int l = threadIdx.x % nlong, u = threadIdx.x / nlong, s = b0 + u;
    const int j = P.C.long_id[l];
    if (j >= j0 && j < j1 && s < a.b.B && !a.W.ctrl[s].done) {
      const size_t at = (size_t)s * n + j;
      const double tt = xb[u * NXB + l];
      io.x_out[at] = fma(1.0 / (double)(a.W.ctrl[s].k + kofs + 3), a.W.x0[at] - tt, tt);
    }
  }
  if (nlong > 0) {
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      fused_reduce_long(lp[u], nlong, red);                    // (block-uniform: every thread takes part for every u)
      if ((int)threadIdx.x < nlong && act[u]) io.lp_out[((size_t)(b0 + u) * nlong + threadIdx.x) * F.ntile + tile] = red[threadIdx.x];
      __syncthreads();
    }
  }
}

// ---- the same iteration with ONE memory phase ---------------------------------------------------------------------------------
// k_fused walks through four dependent global-memory phases per workgroup (y -> LDS | x, c | y0 | x0) with a barrier between
// them: at three workgroups per CU the launch is latency-bound (2.5 TB/s of real traffic, profiles/r30b_*).  Here every thread
// owns K rows and K columns of the tile (row i0 + tid + k NT, column j0 + tid + k NT) and issues ALL its global loads up front
// - y, y0 of its rows, x, c, x0 of its columns for SG scenarios, the ELL entries of both, the few halo elements - before the
// first barrier; after that the workgroup only talks to LDS (y staged, xbar staged) and writes x, y.  MW = the ELL width the
// register arrays are sized for (4 or 8: at 8 the entries of K = 2 rows + columns alone are 96 VGPRs).
// DEFER: the ELL entries of the own / halo columns are requested after the first barrier and those of the own rows after the
// column products (they come from L2, the vectors from HBM): 24 + 12 fewer registers while the HBM loads are in flight.
// Resident waves per SIMD the register allocator is held to (second __launch_bounds__ argument) for the instantiations where it
// gets there WITHOUT scratch (-Rpass-analysis=kernel-resource-usage): the year-long families run <2, 1, 4, shared, LP> (80 VGPRs = 6
// waves; 85 unconstrained) and <2, 1, 4, per-scenario bounds, LP> (94 = 5 waves; 99 unconstrained).  Everything else: unconstrained.
#ifndef DSP_FUSED_WAVES
#define DSP_FUSED_WAVES 1
#endif
// element `off` of a global array through a byte offset computed in the offset's OWN width: with 32-bit offsets (deferred form; the
// host checks that every array of the solve stays below 4 GiB) a load is `global_load v, v_off, s[base]` - one 32-bit register and no
// 64-bit address arithmetic per access (33 v_lshl_add_u64 + 18 other 64-bit VALU instructions per phase before)
template <class T, class O>
__device__ __forceinline__ T &gat(T *base, O off) {
  using C = typename std::conditional<std::is_const<T>::value, const char, char>::type;
  return *reinterpret_cast<T *>(reinterpret_cast<C *>(base) + (size_t)(O)(off * (O)sizeof(T)));
}
constexpr int fused_pre_waves(int sg, int k, int mw, bool shared, bool qp, int defer) {
  if (!DSP_FUSED_WAVES || !defer || sg != 2 || k != 1 || mw != 4) return 1;
  return shared ? (qp ? 5 : 6) : (qp ? 1 : 5);
}
template <int SG, int K, int MW, bool SHARED, bool QP, int DEFER = 0>
__global__ void __launch_bounds__(kTB, fused_pre_waves(SG, K, MW, SHARED, QP, DEFER)) k_fused_pre(StreamArgs a, FusedIO io, int kofs) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using off_t = typename std::conditional<(DEFER != 0), uint32_t, size_t>::type;
  const StreamProblem &P = a.P;
  const FusedPlan &F = P.F;
  const int n = P.n, m = P.m, nlong = P.C.nlong, NY = F.ny_max, NXB = F.nxb_max, WC = P.C.W, WR = P.R.W;
  // grid (scenario groups, tiles): the groups of ONE tile have consecutive workgroup ids, i.e. they are dispatched at nearly the same
  // time, two or more of them to each XCD - whose L2 then serves the tile's slice of the matrix to all but the first
  // XCD-major renumbering of the workgroups (io.xcd_full > 0; dsp_prepare.hpp::fused_workgroup): all scenario groups of a tile on
  // ONE XCD, whose L2 then fetches the tile's slice of the matrix once instead of once per XCD
  int tile, grp;
  fused_workgroup((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)gridDim.x, io.xcd_full, tile, grp);
  const int b0 = grp * SG, tid = threadIdx.x;
  const int32_t *tp = F.tile + 8 * tile;
  const int i0 = tp[0], i1 = tp[1], j0 = tp[2], j1 = tp[3], c_lo = tp[4], c_hi = tp[5], r_lo = tp[6], r_hi = tp[7];
  double *ys = lds, *xb = lds + (size_t)SG * NY, *red = xb + (size_t)SG * NXB;
  // DEFER: what only the dual step needs is not carried in registers through the column products - y is read back from its staged
  // copy, y0, x0 (and per-scenario row bounds) wait in thread-private LDS slots, shared row bounds are requested with the row entries
  double *park = red + (1 + kTB / 64) * kFusedMaxLong * SG + tid;  // slot q of this thread at park[q * kTB]
  // Phase A is written WITHOUT data-dependent control flow: every load goes to a clamped (always valid) address and is issued
  // before anything that was loaded is looked at; validity (row / column inside the tile, long column, scenario alive) only
  // masks the stores and LDS writes at the end.  The first version predicated each load and turned each loaded index into
  // a local slot on the spot: the compiler emitted s_waitcnt vmcnt(0) after nearly every load and the launch ran 20 % slower
  // than the staged kernel (profiles/r30c_fused_scan.log).
  int su[SG];
  bool act[SG];
  double tau[SG], sig[SG], oml[SG];
#pragma unroll
  for (int u = 0; u < SG; ++u) {
    su[u] = min(b0 + u, a.b.B - 1);
    const StreamCtrl &c = a.W.ctrl[su[u]];
    act[u] = b0 + u < a.b.B && !c.done;
    tau[u] = c.tau; sig[u] = c.sig; oml[u] = 1.0 / (double)(c.k + kofs + 3);
  }
  // every scenario of the workgroup has finished: nothing to load, compute or store (uniform over the workgroup, ahead of its first
  // barrier).  A full solve then stops paying for a scenario pair the moment both have terminated (16 year-long LPs: r41g).
  {
    bool any = false;
#pragma unroll
    for (int u = 0; u < SG; ++u) any = any || act[u];
    if (!any) return;
  }
  // ---- phase A: every global load of the workgroup -----------------------------------------------------------------------------
  int ik[K], jk[K];
  double yr[K][SG], y0r[K][SG], xr[K][SG], cr[K][SG], x0r[K][SG], lbr[K][SHARED ? 1 : SG], ubr[K][SHARED ? 1 : SG],
      rlor[K][SHARED ? 1 : SG], rhir[K][SHARED ? 1 : SG], kapr[K][QP ? SG : 1];
  double cval[K][MW], rval[K][MW];
  int cgi[K][MW], rgi[K][MW];
  unsigned char clong[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    ik[k] = min(i0 + tid + k * kTB, m - 1);
    jk[k] = min(j0 + tid + k * kTB, n - 1);
    clong[k] = gat(P.C.is_long, (off_t)jk[k]);
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      const off_t ar = (off_t)su[u] * (off_t)m + (off_t)ik[k], ac = (off_t)su[u] * (off_t)n + (off_t)jk[k];
      yr[k][u] = gat(io.y_in, ar); y0r[k][u] = gat(a.W.y0, ar);
      xr[k][u] = gat(io.x_in, ac); cr[k][u] = gat(a.W.c, ac); x0r[k][u] = gat(a.W.x0, ac);
      if (!SHARED) { lbr[k][u] = gat(a.W.lb, ac); ubr[k][u] = gat(a.W.ub, ac); rlor[k][u] = gat(a.W.rlo, ar); rhir[k][u] = gat(a.W.rhi, ar); }
      if (QP) kapr[k][u] = gat(a.W.kap, ar);
    }
    if (SHARED) {
      lbr[k][0] = gat(a.W.lb, (off_t)jk[k]); ubr[k][0] = gat(a.W.ub, (off_t)jk[k]);
      if (!DEFER) { rlor[k][0] = gat(a.W.rlo, (off_t)ik[k]); rhir[k][0] = gat(a.W.rhi, (off_t)ik[k]); }
    }
    if (!DEFER) {
#pragma unroll
      for (int e = 0; e < MW; ++e) {
        const off_t oc = (off_t)min(e, WC - 1) * (off_t)n + (off_t)jk[k], orow = (off_t)min(e, WR - 1) * (off_t)m + (off_t)ik[k];
        cval[k][e] = gat(P.C.val, oc); cgi[k][e] = gat(P.C.idx, oc);
        rval[k][e] = gat(P.R.val, orow); rgi[k][e] = gat(F.ridx_enc, orow);
      }
    }
  }
  // one halo column and one halo row (staged y only) per thread: at most a few periods' worth (host: halo_max <= 256)
  const int nhr = (i0 - r_lo) + (r_hi - i1), nhc = (j0 - c_lo) + (c_hi - j1);
  // (threads beyond the halo all read ONE valid address: letting them run on past the tile fetched 256 extra rows and
  //  columns per workgroup - half as much again as the tile's own y, x and c: profiles/r30h_stream_pmc_summary.csv)
  const int hi_ = tid < nhr ? min(tid < i0 - r_lo ? r_lo + tid : i1 + (tid - (i0 - r_lo)), m - 1) : r_lo;
  const int hj = tid < nhc ? min(tid < j0 - c_lo ? c_lo + tid : j1 + (tid - (j0 - c_lo)), n - 1) : c_lo;
  double hy[SG], hx[SG], hc[SG], hlb[SHARED ? 1 : SG], hub[SHARED ? 1 : SG], hval[MW];
  int hgi[MW];
  const unsigned char hlong = gat(P.C.is_long, (off_t)hj);
#pragma unroll
  for (int u = 0; u < SG; ++u) {
    hy[u] = gat(io.y_in, (off_t)su[u] * (off_t)m + (off_t)hi_);
    const off_t ac = (off_t)su[u] * (off_t)n + (off_t)hj;
    hx[u] = gat(io.x_in, ac); hc[u] = gat(a.W.c, ac);
    if (!SHARED) { hlb[u] = gat(a.W.lb, ac); hub[u] = gat(a.W.ub, ac); }
  }
  if (SHARED) { hlb[0] = gat(a.W.lb, (off_t)hj); hub[0] = gat(a.W.ub, (off_t)hj); }
  if (!DEFER) {
#pragma unroll
    for (int e = 0; e < MW; ++e) { const off_t oc = (off_t)min(e, WC - 1) * (off_t)n + (off_t)hj; hval[e] = gat(P.C.val, oc); hgi[e] = gat(P.C.idx, oc); }
  }
  // long columns' A^T y from the per-tile partial sums (one wave per (scenario, long column), fixed order)
  for (int q = tid >> 6; q < nlong * SG; q += kTB / 64) {
    const int l = q % nlong, u = q / nlong, lane = tid & 63;
    const double *pp = io.lp_in + ((size_t)su[u] * nlong + l) * F.ntile;
    // (the column's own x, c and bounds are loaded by lane 0 AFTER the reduction: requested up front - one memory round trip
    //  less for two of the four waves - they changed nothing in the time and cost 8 registers at the kernel's peak, which is
    //  here: profiles/r30m_fused_ab.log)
    double aty = 0.0;
    for (int t = lane; t < F.ntile; t += 64) aty += pp[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) aty += __shfl_down(aty, off, 64);
    if (lane == 0) {
      const int j = P.C.long_id[l];
      const size_t at = (size_t)su[u] * n + j, ab = SHARED ? (size_t)j : at;
      const double x = io.x_in[at];
      const double xp = clampd2(fma(-tau[u], a.W.c[at] - aty, x), a.W.lb[ab], a.W.ub[ab]);
      xb[u * NXB + l] = 2.0 * xp - x;
    }
  }
  // ---- phase B: y into LDS (own rows + halo rows) ----------------------------------------------------------------------------------
  bool rowok[K], colok[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    rowok[k] = i0 + tid + k * kTB < i1;
    colok[k] = j0 + tid + k * kTB < j1 && !clong[k];
    if (rowok[k]) {
#pragma unroll
      for (int u = 0; u < SG; ++u) ys[u * NY + (ik[k] - r_lo)] = yr[k][u];
    }
    if (DEFER) {
#pragma unroll
      for (int u = 0; u < SG; ++u) {
        park[(size_t)((k * SG + u) * (SHARED ? 2 : 4)) * kTB] = y0r[k][u];
        park[(size_t)((k * SG + u) * (SHARED ? 2 : 4) + 1) * kTB] = x0r[k][u];
        if (!SHARED) {
          park[(size_t)((k * SG + u) * 4 + 2) * kTB] = rlor[k][u];
          park[(size_t)((k * SG + u) * 4 + 3) * kTB] = rhir[k][u];
        }
      }
    }
  }
  if (tid < nhr) {
#pragma unroll
    for (int u = 0; u < SG; ++u) ys[u * NY + (hi_ - r_lo)] = hy[u];
  }
  __syncthreads();
  // ---- phase C: primal step of the own columns (registers) and of the halo columns -> xbar in LDS; x written ------------------------
  if (DEFER) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int e = 0; e < MW; ++e) {
        const off_t oc = (off_t)min(e, WC - 1) * (off_t)n + (off_t)jk[k];
        cval[k][e] = gat(P.C.val, oc); cgi[k][e] = gat(P.C.idx, oc);
      }
#pragma unroll
    for (int e = 0; e < MW; ++e) { const off_t oc = (off_t)min(e, WC - 1) * (off_t)n + (off_t)hj; hval[e] = gat(P.C.val, oc); hgi[e] = gat(P.C.idx, oc); }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    int cl[MW];
#pragma unroll
    for (int e = 0; e < MW; ++e) cl[e] = (e >= WC || cgi[k][e] < r_lo || cgi[k][e] >= r_hi) ? 0 : cgi[k][e] - r_lo;
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      const double *yv = ys + u * NY;
      double aty = 0.0;
#pragma unroll
      for (int e = 0; e < MW; ++e) aty = fma(e < WC ? cval[k][e] : 0.0, yv[cl[e]], aty);
      const double xp = clampd2(fma(-tau[u], cr[k][u] - aty, xr[k][u]), lbr[k][SHARED ? 0 : u], ubr[k][SHARED ? 0 : u]);
      const double tt = 2.0 * xp - xr[k][u];
      if (colok[k]) {
        xb[u * NXB + nlong + (jk[k] - c_lo)] = tt;
        if (act[u]) {
          const double x0v = DEFER ? park[(size_t)((k * SG + u) * (SHARED ? 2 : 4) + 1) * kTB] : x0r[k][u];
          gat(io.x_out, (off_t)su[u] * (off_t)n + (off_t)jk[k]) = fma(oml[u], x0v - tt, tt);
        }
      }
    }
  }
  {
    int cl[MW];
#pragma unroll
    for (int e = 0; e < MW; ++e) cl[e] = (e >= WC || hgi[e] < r_lo || hgi[e] >= r_hi) ? 0 : hgi[e] - r_lo;
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      const double *yv = ys + u * NY;
      double aty = 0.0;
#pragma unroll
      for (int e = 0; e < MW; ++e) aty = fma(e < WC ? hval[e] : 0.0, yv[cl[e]], aty);
      const double xp = clampd2(fma(-tau[u], hc[u] - aty, hx[u]), hlb[SHARED ? 0 : u], hub[SHARED ? 0 : u]);
      if (tid < nhc && !hlong) xb[u * NXB + nlong + (hj - c_lo)] = 2.0 * xp - hx[u];
    }
  }
  if (DEFER) {                                  // (requested ahead of the barrier: the L2 round trip overlaps the wait)
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int e = 0; e < MW; ++e) {
        const off_t orow = (off_t)min(e, WR - 1) * (off_t)m + (off_t)ik[k];
        rval[k][e] = gat(P.R.val, orow); rgi[k][e] = gat(F.ridx_enc, orow);
      }
    if (SHARED) {
#pragma unroll
      for (int k = 0; k < K; ++k) { rlor[k][0] = gat(a.W.rlo, (off_t)ik[k]); rhir[k][0] = gat(a.W.rhi, (off_t)ik[k]); }
    }
  }
  if (tid < nlong * SG) {                       // the long columns this tile owns: their Halpern step (xbar written in phase A)
    const int l = tid % nlong, u = tid / nlong;
    const int j = P.C.long_id[l];
    if (j >= j0 && j < j1 && act[u]) {
      const size_t at = (size_t)su[u] * n + j;
      const double tt = xb[u * NXB + l];
      io.x_out[at] = fma(oml[u], a.W.x0[at] - tt, tt);
    }
  }
  __syncthreads();
  // ---- phase D: dual step + Halpern averaging of the own rows, partial sums for the long columns ------------------------------------
  double lp[SG][kFusedMaxLong];
#pragma unroll
  for (int u = 0; u < SG; ++u)
#pragma unroll
    for (int q = 0; q < kFusedMaxLong; ++q) lp[u][q] = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    int sl[MW];
    double rv[MW];
#pragma unroll
    for (int e = 0; e < MW; ++e) {
      const int gi = rgi[k][e];
      int t = gi < 0 ? -1 - gi : nlong + gi - c_lo;
      if (e >= WR || (gi >= 0 && (gi < c_lo || gi >= c_hi))) t = 0;
      sl[e] = t;
      rv[e] = e < WR ? rval[k][e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      const double *xv = xb + u * NXB;
      double ax = 0.0;
#pragma unroll
      for (int e = 0; e < MW; ++e) ax = fma(rv[e], xv[sl[e]], ax);
      double y, y0v, rlo_, rhi_;
      if (DEFER) {
        y = (ys + u * NY)[rowok[k] ? ik[k] - r_lo : 0];
        y0v = park[(size_t)((k * SG + u) * (SHARED ? 2 : 4)) * kTB];
        rlo_ = SHARED ? rlor[k][0] : park[(size_t)((k * SG + u) * 4 + 2) * kTB];
        rhi_ = SHARED ? rhir[k][0] : park[(size_t)((k * SG + u) * 4 + 3) * kTB];
      } else {
        y = yr[k][u]; y0v = y0r[k][u]; rlo_ = rlor[k][SHARED ? 0 : u]; rhi_ = rhir[k][SHARED ? 0 : u];
      }
      const double gy = fma(-sig[u], ax, y);
      double yp = gy - clampd2(gy, -sig[u] * rhi_, -sig[u] * rlo_);
      if (QP) yp /= fma(sig[u], kapr[k][u], 1.0);
      const double tt = 2.0 * yp - y;
      const double yn = fma(oml[u], y0v - tt, tt);
      if (rowok[k] && act[u]) {
        gat(io.y_out, (off_t)su[u] * (off_t)m + (off_t)ik[k]) = yn;
        // this row's terms of A^T y for the long columns.  ONE long column (the design variable of the price-taker families) is the
        // common case: whatever entry is long belongs to it - the select over kFusedMaxLong accumulators per entry was a sixth of
        // the kernel's VALU instructions (ISA count: 83 v_cndmask + 33 v_cmp + 32 of the v_add_f64 in this phase)
        if (nlong == 1) {
#pragma unroll
          for (int e = 0; e < MW; ++e) lp[u][0] += (rgi[k][e] < 0 && e < WR) ? rv[e] * yn : 0.0;
        } else if (nlong > 1) {
#pragma unroll
          for (int e = 0; e < MW; ++e) {
            const double w = (rgi[k][e] < 0 && e < WR) ? rv[e] * yn : 0.0;
#pragma unroll
            for (int q = 0; q < kFusedMaxLong; ++q) lp[u][q] += (sl[e] == q) ? w : 0.0;
          }
        }
      }
    }
  }
  if (nlong > 0) {
    // block sums of the SG x nlong partial sums: wave totals on the VALU (DPP), ONE barrier, thread (u, q) adds the four wave totals
    // in wave order and writes the tile's partial sum (fixed order: bit-reproducible)
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int u = 0; u < SG; ++u)
#pragma unroll
      for (int q = 0; q < kFusedMaxLong; ++q)
        if (q < nlong) {
          const double t = wave_sum(lp[u][q]);
          if (lane == 0) red[(wave * SG + u) * kFusedMaxLong + q] = t;
        }
    __syncthreads();
    if (tid < SG * kFusedMaxLong) {
      const int u = tid / kFusedMaxLong, q = tid % kFusedMaxLong;
      if (q < nlong && act[u]) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kTB / 64; ++w) t += red[(w * SG + u) * kFusedMaxLong + q];
        io.lp_out[((size_t)su[u] * nlong + q) * F.ntile + tile] = t;
      }
    }
  }
}

// partial sums of A^T y for the long columns from the CURRENT y (after the initialisation and after every check's k_apply)
__global__ void __launch_bounds__(kTB) k_long_partials(StreamArgs a, const double *y, double *lp_out) {
  __shared__ double red[(1 + kTB / 64) * kFusedMaxLong];
  const StreamProblem &P = a.P;
  const FusedPlan &F = P.F;
  const int tile = blockIdx.x, s = blockIdx.y, m = P.m, nlong = P.C.nlong;
  if (a.W.ctrl[s].done) return;
  const int32_t *tp = F.tile + 8 * tile;
  double lp[kFusedMaxLong] = {0.0, 0.0, 0.0, 0.0};
  for (int i = tp[0] + threadIdx.x; i < tp[1]; i += kTB) {
    const double yi = y[(size_t)s * m + i];
    for (int e = 0; e < P.R.W; ++e) {
      const int gi = F.ridx_enc[(size_t)e * m + i];
      const double v = P.R.val[(size_t)e * m + i];
      if (gi < 0 && v != 0.0) {
#pragma unroll
        for (int q = 0; q < kFusedMaxLong; ++q) lp[q] += (-1 - gi == q) ? v * yi : 0.0;
      }
    }
  }
  fused_reduce_long(lp, nlong, red);
  if ((int)threadIdx.x < nlong) lp_out[((size_t)s * nlong + threadIdx.x) * F.ntile + tile] = red[threadIdx.x];
}

// ---- results -------------------------------------------------------------------------------------------------------------------
__global__ void k_finalize(StreamArgs a) {
  const StreamProblem &P = a.P;
  const dsp_batch &b = a.b;
  const int t = blockIdx.x * kTB + threadIdx.x;
  const int s = blockIdx.y;
  const StreamCtrl &c = a.W.ctrl[s];
  const bool invalid = c.status == DSP_STATUS_PRIMAL_INFEASIBLE || c.status == DSP_STATUS_NUMERICAL;
  if (t < P.n) b.x[(size_t)s * P.n + t] = invalid && c.it == 0 ? NAN : a.W.xp[(size_t)s * P.n + t] * P.col_scale[t];
  if (t < P.m) b.y[(size_t)s * P.m + t] = invalid && c.it == 0 ? NAN : a.W.yp[(size_t)s * P.m + t] * P.row_scale[t];
  if (t == 0) {
    b.obj[s] = c.pobj;
    b.status[s] = c.status;
    if (b.iters) b.iters[s] = c.it;
    if (b.jumps) b.jumps[s] = 0;
    if (b.flags) b.flags[s] = 0;
    if (b.primal_weight) b.primal_weight[s] = c.w;
  }
}

template <class T>
hipError_t up(std::vector<void *> &allocs, const std::vector<T> &v, const T **out) {
  void *d = nullptr;
  hipError_t e = hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T));
  if (e != hipSuccess) return e;
  allocs.push_back(d);
  if (!v.empty()) { e = hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); if (e != hipSuccess) return e; }
  *out = reinterpret_cast<const T *>(d);
  return hipSuccess;
}

// entry-major ELL + long list of a host CSR (vectors = rows of M): built on the host (dsp_prepare.hpp), uploaded here
hipError_t build_matrix(const HostCSR &M, std::vector<void *> &allocs, StreamMatrix *out, HostStreamELL *host, int span_limit = 0) {
  *host = build_stream_ell(M, kStreamMaxW, kLongChunk, span_limit);
  const HostStreamELL &E = *host;
  if ((int)E.long_id.size() > kStreamMaxLong) return hipErrorInvalidValue;
  out->nvec = E.nvec; out->W = E.W; out->nlong = (int)E.long_id.size(); out->nchunk = (int)E.chunk_begin.size();
  hipError_t e;
  if ((e = up(allocs, E.val, &out->val)) != hipSuccess) return e;
  if ((e = up(allocs, E.idx, &out->idx)) != hipSuccess) return e;
  if ((e = up(allocs, E.is_long, &out->is_long)) != hipSuccess) return e;
  if ((e = up(allocs, E.long_id, &out->long_id)) != hipSuccess) return e;
  if ((e = up(allocs, E.long_ptr, &out->long_ptr)) != hipSuccess) return e;
  if ((e = up(allocs, E.long_idx, &out->long_idx)) != hipSuccess) return e;
  if ((e = up(allocs, E.long_val, &out->long_val)) != hipSuccess) return e;
  if ((e = up(allocs, E.chunk_begin, &out->chunk_begin)) != hipSuccess) return e;
  if ((e = up(allocs, E.chunk_end, &out->chunk_end)) != hipSuccess) return e;
  if ((e = up(allocs, E.long_chunk_ptr, &out->long_chunk_ptr)) != hipSuccess) return e;
  return hipSuccess;
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------
hipError_t stream_create(const HostCSR &A_scaled, const HostCSR &AT_scaled, const double *col_scale_dev,
                         const double *row_scale_dev, StreamSolver *S) {
  S->P.n = A_scaled.n; S->P.m = A_scaled.m;
  S->P.col_scale = col_scale_dev; S->P.row_scale = row_scale_dev;
  hipError_t e;
  HostStreamELL Er, Ec;
  if ((e = build_matrix(A_scaled, S->allocs, &S->P.R, &Er)) != hipSuccess) return e;
  // columns that wrap around the horizon (periodic boundary conditions) are long: see build_stream_ell
  if ((e = build_matrix(AT_scaled, S->allocs, &S->P.C, &Ec, std::max(4096, A_scaled.m / 4))) != hipSuccess) return e;
  // fused one-launch iteration where the matrix is banded (development switches: DSP_STREAM_NO_FUSED=1, DSP_FUSED_RB=<rows per tile>)
  // (read at every create, not once per process: a test builds one handle of each form)
  const int no_fused = getenv("DSP_STREAM_NO_FUSED") ? atoi(getenv("DSP_STREAM_NO_FUSED")) : 0;
  const int rb_env = getenv("DSP_FUSED_RB") ? atoi(getenv("DSP_FUSED_RB")) : 0;
  S->P.F = FusedPlan{};
  if (!no_fused) {
    const HostFusedPlan H = build_fused_plan(A_scaled, AT_scaled, Er, Ec, kFusedMaxLong, rb_env > 0 ? rb_env : 250, 40 * 1024);     // one row + one column per thread (K = 1): measured best, profiles/r30j_fused_scan.log
    if (H.ntile > 0) {
      if ((e = up(S->allocs, H.tile, &S->P.F.tile)) != hipSuccess) return e;
      if ((e = up(S->allocs, H.ridx_enc, &S->P.F.ridx_enc)) != hipSuccess) return e;
      S->P.F.ntile = H.ntile; S->P.F.rows_per_tile = H.rows_per_tile; S->P.F.ny_max = H.ny_max; S->P.F.nxb_max = H.nxb_max;
      S->P.F.own_max = H.own_max; S->P.F.halo_max = H.halo_max;
    }
  }
  // interior-point form (round 5): time-banded matrices with at most 4 wide columns
  if ((e = ipm_create(A_scaled, AT_scaled, S)) != hipSuccess) return e;
  // lane-per-scenario form (round 4): banded matrices with at most 8 long columns
  return lane_create(A_scaled, AT_scaled, S);
}

void stream_destroy(StreamSolver *S) {
  lane_destroy(S);
  ipm_destroy(S);
  for (void *p : S->allocs) (void)hipFree(p);
  for (void *p : S->work_allocs) (void)hipFree(p);
  S->allocs.clear(); S->work_allocs.clear();
  if (S->ndone_host) { (void)hipHostFree(S->ndone_host); S->ndone_host = nullptr; }
}

static hipError_t ensure_workspace(StreamSolver *S, int B) {
  if (B <= S->work_B) return hipSuccess;
  for (void *p : S->work_allocs) (void)hipFree(p);
  S->work_allocs.clear();
  S->work_B = 0;
  const size_t n = S->P.n, m = S->P.m;
  const int nblk = (int)((std::max(n, m) + kTB - 1) / kTB);
  const int nblk_tot = nblk + std::max(S->P.R.nlong, S->P.C.nlong);
  auto alloc = [&](size_t bytes, void **out) {
    hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 8));
    if (e == hipSuccess) S->work_allocs.push_back(*out);
    return e;
  };
  StreamWork &W = S->W;
  hipError_t e;
  double **colv[] = {&W.x, &W.x0, &W.xp, &W.xbar, &W.c, &W.lb, &W.ub};
  double **rowv[] = {&W.y, &W.y0, &W.yp, &W.rlo, &W.rhi, &W.kap};
  for (double **p : colv) if ((e = alloc((size_t)B * n * sizeof(double), (void **)p)) != hipSuccess) return e;
  for (double **p : rowv) if ((e = alloc((size_t)B * m * sizeof(double), (void **)p)) != hipSuccess) return e;
  if ((e = alloc((size_t)B * sizeof(StreamCtrl), (void **)&W.ctrl)) != hipSuccess) return e;
  if ((e = alloc((size_t)B * nblk_tot * kNQ * sizeof(double), (void **)&W.partial)) != hipSuccess) return e;
  if ((e = hipMemset(W.partial, 0, (size_t)B * nblk_tot * kNQ * sizeof(double))) != hipSuccess) return e;
  if ((e = alloc((size_t)B * std::max(1, std::max(S->P.R.nchunk, S->P.C.nchunk)) * sizeof(double), (void **)&W.long_partial)) != hipSuccess) return e;
  if (S->P.F.ntile > 0) {
    if ((e = alloc((size_t)B * n * sizeof(double), (void **)&W.x2)) != hipSuccess) return e;
    if ((e = alloc((size_t)B * m * sizeof(double), (void **)&W.y2)) != hipSuccess) return e;
    const size_t lp = (size_t)B * std::max(1, S->P.C.nlong) * S->P.F.ntile * sizeof(double);
    for (int q = 0; q < 2; ++q) if ((e = alloc(lp, (void **)&W.lpart[q])) != hipSuccess) return e;
  }
  if ((e = alloc((size_t)B * m * sizeof(double), (void **)&W.ray)) != hipSuccess) return e;
  if ((e = alloc(2 * sizeof(int), (void **)&W.ndone)) != hipSuccess) return e;
  if (!S->ndone_host && (e = hipHostMalloc((void **)&S->ndone_host, 2 * sizeof(int))) != hipSuccess) return e;
  S->work_B = B;
  return hipSuccess;
}

size_t stream_bytes_per_iteration(const StreamSolver *S) {
  // per scenario and plain iteration of the form the last solve ran: two launches 8 n + 6 m doubles (see the file header); fused
  // banded form 4 n + 3 m with shared bounds, 6 n + 5 m with per-scenario bounds (k_fused); before any solve: the two-launch figure
  return S->last_bytes_per_iteration ? S->last_bytes_per_iteration : (size_t)8 * (8 * (size_t)S->P.n + 6 * (size_t)S->P.m);
}

// The certificate sequence for the scenarios control_decide marked suspect, on the scenario-major workspace: a.W.x / a.W.y must be
// the current iterate (every form's check sequence leaves it there; the lane form brings it back first).  xp / yp are overwritten
// with T(z) of that iterate - what the next check would write anyway -, xbar and ray are scratch; certified scenarios get
// status 2 / 3, done = 1 and count as finished.  The partial-sum buffer is left zeroed as after k_init.
hipError_t stream_certify(StreamSolver *S, StreamArgs &a, hipStream_t st) {
  const StreamProblem &P = a.P;
  const int B = a.b.B;
  const dim3 blk(kTB);
  const dim3 g_primal(a.nblk_n + P.C.nchunk, B), g_rows(a.nblk + P.R.nlong, B), g_cols(a.nblk_n + P.C.nchunk, B), g_elem(a.nblk, B);
  const int fin_c = (P.C.nlong * B + 63) / 64;
  const size_t pbytes = (size_t)B * a.nblk_tot * kNQ * sizeof(double);
  hipError_t e;
  hipLaunchKernelGGL((k_primal<1, true>), g_primal, blk, 0, st, a);
  if (P.C.nlong) hipLaunchKernelGGL(k_primal_long_finish, dim3(fin_c), dim3(64), 0, st, a, 1);
  hipLaunchKernelGGL((k_check_rows<1>), g_rows, blk, 0, st, a);
  if ((e = hipMemsetAsync(a.W.partial, 0, pbytes, st)) != hipSuccess) return e;
  hipLaunchKernelGGL((k_ray_prep<1>), g_elem, blk, 0, st, a);
  hipLaunchKernelGGL((k_ray_rows<1>), g_rows, blk, 0, st, a);
  hipLaunchKernelGGL((k_ray_cols<1>), g_cols, blk, 0, st, a);
  if (P.C.nlong) hipLaunchKernelGGL(k_ray_long_finish, dim3(fin_c), dim3(64), 0, st, a);
  hipLaunchKernelGGL(k_ray_decide, dim3(B), dim3(256), 0, st, a);
  if ((e = hipMemsetAsync(a.W.partial, 0, pbytes, st)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(a.W.ndone + 1, 0, sizeof(int), st)) != hipSuccess) return e;
  (void)S;
  return hipGetLastError();
}

template <int SG>
static hipError_t run(StreamSolver *S, StreamArgs &a, hipStream_t st, int *periods_run) {
  const StreamProblem &P = a.P;
  const int B = a.b.B;
  const int groups = (B + SG - 1) / SG;
  const dim3 blk(kTB);
  const dim3 g_primal(a.nblk_n + P.C.nchunk, groups), g_dual(a.nblk + P.R.nchunk, groups);
  const dim3 g_rows_chk(a.nblk + P.R.nlong, groups), g_cols(a.nblk_n + P.C.nchunk, groups);
  const dim3 g_elem(a.nblk, groups);
  const int fin_c = (P.C.nlong * B + 63) / 64, fin_r = (P.R.nlong * B + 63) / 64;
  auto primal = [&](bool write_xp) {
    if (write_xp) hipLaunchKernelGGL((k_primal<SG, true>), g_primal, blk, 0, st, a);
    else hipLaunchKernelGGL((k_primal<SG, false>), g_primal, blk, 0, st, a);
    if (P.C.nlong) hipLaunchKernelGGL(k_primal_long_finish, dim3(fin_c), dim3(64), 0, st, a, write_xp ? 1 : 0);
  };
  const int C = a.opt.check_every > 0 ? a.opt.check_every : 64;       // 0 = automatic (include/dsp_hip.h)
  const int max_periods = (a.opt.max_iter + C - 1) / C;
  const int poll = 4;                                        // check periods between two looks at the finished counter
  int period = 0;
  hipError_t e = hipSuccess;
  for (; period < max_periods; ++period) {
    for (int u = 0; u < C - 1; ++u) {
      primal(false);
      hipLaunchKernelGGL((k_dual_halpern<SG>), g_dual, blk, 0, st, a, u);
      if (P.R.nlong) hipLaunchKernelGGL(k_dual_long_finish, dim3(fin_r), dim3(64), 0, st, a, u);
    }
    primal(true);
    hipLaunchKernelGGL((k_check_rows<SG>), g_rows_chk, blk, 0, st, a);
    hipLaunchKernelGGL((k_kkt_cols<SG>), g_cols, blk, 0, st, a);
    if (P.C.nlong) hipLaunchKernelGGL(k_kkt_long_finish, dim3((P.C.nlong * B + 63) / 64), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_control, dim3(B), dim3(256), 0, st, a, C);
    hipLaunchKernelGGL((k_apply<SG>), g_elem, blk, 0, st, a);
    if ((period + 1) % poll == 0 || period + 1 == max_periods) {
      if ((e = hipMemcpyAsync(S->ndone_host, a.W.ndone, 2 * sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
      if (S->ndone_host[0] >= B) { ++period; break; }
      // suspects (relative gap >= 1/2 after 2048 iterations): the certificate sequence, at most every 16 check periods
      if (S->ndone_host[1] && (period + 1) % 16 == 0) {
        if ((e = stream_certify(S, a, st)) != hipSuccess) return e;
      }
    }
  }
  *periods_run = period;
  return hipGetLastError();
}

// BLOCKING and one call at a time per handle (include/dsp_hip.h, dsp_solve): the per-scenario state lives in ONE workspace per
// handle (reallocated when a larger batch arrives) and the host polls the device's finished-counter while it enqueues the
// iteration launches, so - unlike the fused path with its ring of work queues - two solves of one handle cannot overlap.  The
// mutex serialises callers on different threads / streams; the call returns after its stream work has completed.
static hipError_t stream_solve_locked(StreamSolver *S, const dsp_batch &batch, const dsp_options &opt, double eta, hipStream_t st,
                                      int *periods_run);

// The fused form of `run`: C - 1 plain iterations = C - 1 launches of k_fused (x / y ping-pong), then the SAME check sequence
// as the two-launch form on whichever buffers are current, then the long columns' partial sums of the y the check left.
template <int SG>
static hipError_t run_fused(StreamSolver *S, StreamArgs &a, hipStream_t st, int *periods_run, bool shared, bool qp) {
  const StreamProblem &P = a.P;
  const FusedPlan &F = P.F;
  const int B = a.b.B;
  const int groups = (B + SG - 1) / SG;
  const dim3 blk(kTB);
  const dim3 g_fused(groups, F.ntile), g_primal(a.nblk_n + P.C.nchunk, groups);
  const dim3 g_rows_chk(a.nblk + P.R.nlong, groups), g_cols(a.nblk_n + P.C.nchunk, groups), g_elem(a.nblk, groups);
  const int fin_c = (P.C.nlong * B + 63) / 64;
  const int K_own = (F.own_max + kTB - 1) / kTB;
  // (+ the thread-private slots of k_fused_pre's deferred form: y0, x0, and the row bounds when they differ per scenario)
  const size_t lds = ((size_t)SG * (F.ny_max + F.nxb_max) + (1 + kTB / 64) * kFusedMaxLong * SG + (size_t)K_own * SG * (shared ? 2 : 4) * kTB) * sizeof(double);
  // k_fused_pre (every HBM load of the workgroup ahead of the first barrier, matrix entries requested after the barriers) where a
  // thread owns ONE row and ONE column of the tile (250-row tiles: always) and the halo columns fit one pass; k_fused (staged
  // phases) otherwise.  DSP_FUSED_V=1 forces the staged form (development).  Since round 4 this form serves batches below 32
  // scenarios only (larger ones run the lane form, dsp_stream_lane.hip): the instantiations of round 3 for 2 - 3 rows per thread,
  // 4 scenarios per workgroup and loads-up-front are gone (144 -> 16 + 8).
  const void *fn = nullptr;
  const int v_env = getenv("DSP_FUSED_V") ? atoi(getenv("DSP_FUSED_V")) : 0;      // (read per solve: tests switch forms)
  const int K = (F.own_max + kTB - 1) / kTB;
  const int mw = std::max(P.C.W, P.R.W);
  // the deferred form addresses with 32-bit byte offsets: every array it touches must stay below 4 GiB
  const bool narrow_ok = (uint64_t)std::max(B, mw) * (uint64_t)std::max(P.n, P.m) * 8ull < (1ull << 32);
  const bool pre = v_env != 1 && K == 1 && mw <= 8 && F.halo_max <= kTB && narrow_ok;
#define DSP_PICK(MM)                                                                                                                 \
  (shared ? (qp ? reinterpret_cast<const void *>(&k_fused_pre<SG, 1, MM, true, true, 1>) : reinterpret_cast<const void *>(&k_fused_pre<SG, 1, MM, true, false, 1>)) \
          : (qp ? reinterpret_cast<const void *>(&k_fused_pre<SG, 1, MM, false, true, 1>) : reinterpret_cast<const void *>(&k_fused_pre<SG, 1, MM, false, false, 1>)))
  if (pre) fn = mw <= 4 ? DSP_PICK(4) : DSP_PICK(8);
  else fn = shared ? (qp ? reinterpret_cast<const void *>(&k_fused<SG, true, true>) : reinterpret_cast<const void *>(&k_fused<SG, true, false>))
                   : (qp ? reinterpret_cast<const void *>(&k_fused<SG, false, true>) : reinterpret_cast<const void *>(&k_fused<SG, false, false>));
#undef DSP_PICK
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int xcd_env = getenv("DSP_FUSED_XCD") ? atoi(getenv("DSP_FUSED_XCD")) : 1;   // default on (DSP_FUSED_XCD=0: grid order)
  const int xcd_full = (pre && xcd_env) ? fused_xcd_full(groups, F.ntile) : 0;
  double *xcur = a.W.x, *ycur = a.W.y, *xalt = a.W.x2, *yalt = a.W.y2;
  int lp_cur = 0;
  auto partials = [&]() {
    if (P.C.nlong) hipLaunchKernelGGL(k_long_partials, dim3(F.ntile, B), blk, 0, st, a, (const double *)ycur, a.W.lpart[lp_cur]);
  };
  partials();
  const int C = a.opt.check_every > 0 ? a.opt.check_every : 64;
  const int max_periods = (a.opt.max_iter + C - 1) / C;
  const int poll = 4;
  int period = 0;
  for (; period < max_periods; ++period) {
    for (int u = 0; u < C - 1; ++u) {
      FusedIO io{xcur, ycur, xalt, yalt, a.W.lpart[lp_cur], a.W.lpart[lp_cur ^ 1], xcd_full};
      int kofs = u;
      void *params[] = {&a, &io, &kofs};
      if ((e = hipLaunchKernel(fn, g_fused, blk, params, lds, st)) != hipSuccess) return e;
      std::swap(xcur, xalt); std::swap(ycur, yalt); lp_cur ^= 1;
    }
    // the check sequence works in place on the current buffers
    StreamArgs ac = a;
    ac.W.x = xcur; ac.W.y = ycur;
    hipLaunchKernelGGL((k_primal<SG, true>), g_primal, blk, 0, st, ac);
    if (P.C.nlong) hipLaunchKernelGGL(k_primal_long_finish, dim3(fin_c), dim3(64), 0, st, ac, 1);
    hipLaunchKernelGGL((k_check_rows<SG>), g_rows_chk, blk, 0, st, ac);
    hipLaunchKernelGGL((k_kkt_cols<SG>), g_cols, blk, 0, st, ac);
    if (P.C.nlong) hipLaunchKernelGGL(k_kkt_long_finish, dim3(fin_c), dim3(64), 0, st, ac);
    hipLaunchKernelGGL(k_control, dim3(B), dim3(256), 0, st, ac, C);
    hipLaunchKernelGGL((k_apply<SG>), g_elem, blk, 0, st, ac);
    partials();
    if ((period + 1) % poll == 0 || period + 1 == max_periods) {
      if ((e = hipMemcpyAsync(S->ndone_host, a.W.ndone, 2 * sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
      if (S->ndone_host[0] >= B) { ++period; break; }
      if (S->ndone_host[1] && (period + 1) % 16 == 0) {          // certificate sequence on the buffers that are current
        StreamArgs ar = a;
        ar.W.x = xcur; ar.W.y = ycur;
        if ((e = stream_certify(S, ar, st)) != hipSuccess) return e;
      }
    }
  }
  *periods_run = period;
  return hipGetLastError();
}

hipError_t stream_solve(StreamSolver *S, const dsp_batch &batch, const dsp_options &opt, double eta, hipStream_t st,
                        int *periods_run) {
  std::lock_guard<std::mutex> lock(S->mu);
  hipError_t e = stream_solve_locked(S, batch, opt, eta, st, periods_run);
  // (the interior-point form solved part of the batch and a PDHG form the rest: reported as the former, dsp_stats::ipm_solved says how many)
  if (e == hipSuccess && S->last_ipm_solved > 0) { S->last_form = DSP_STREAM_FORM_IPM; S->last_phases = ipm_partitions(S); }
  const hipError_t es = hipStreamSynchronize(st);
  return e != hipSuccess ? e : es;
}

static hipError_t stream_solve_locked(StreamSolver *S, const dsp_batch &batch, const dsp_options &opt, double eta, hipStream_t st,
                                      int *periods_run) {
  const int B = batch.B;
  hipError_t e = ensure_workspace(S, B);
  if (e != hipSuccess) return e;
  StreamArgs a{};
  a.P = S->P; a.W = S->W; a.b = batch; a.opt = opt; a.eta = eta;
  a.nblk_n = (S->P.n + kTB - 1) / kTB;
  a.nblk = (std::max(S->P.n, S->P.m) + kTB - 1) / kTB;
  a.nblk_tot = a.nblk + std::max(S->P.R.nlong, S->P.C.nlong);
  a.nchunk_max = std::max(1, std::max(S->P.R.nchunk, S->P.C.nchunk));
  if ((e = hipMemsetAsync(a.W.ndone, 0, 2 * sizeof(int), st)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(a.W.partial, 0, (size_t)B * a.nblk_tot * kNQ * sizeof(double), st)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_init, dim3(a.nblk, B), dim3(kTB), 0, st, a);
  hipLaunchKernelGGL(k_init_control, dim3(B), dim3(64), 0, st, a);
  // k_init wrote its partials with stride nblk; the check kernels use nblk_tot: clear again before the first check
  if ((e = hipMemsetAsync(a.W.partial, 0, (size_t)B * a.nblk_tot * kNQ * sizeof(double), st)) != hipSuccess) return e;
  // time-banded LPs: interior point with exact banded solves first (dsp_ipm.hip); what it does not finish continues below as before
  S->last_newton = 0;
  S->last_ipm_solved = 0;
  std::vector<int> left;                               // scenarios the interior-point form left (empty: it did not run, or solved nothing)
  if (S->ipm && !opt.no_interior_point && !batch.row_compliance) {
    bool all = false;
    int n_solved = 0;
    if ((e = ipm_run(S, a, st, &all, &S->last_newton, &n_solved)) != hipSuccess) return e;
    S->last_ipm_solved = n_solved;
    if (all) {
      S->last_form = DSP_STREAM_FORM_IPM;
      S->last_phases = ipm_partitions(S);
      S->last_bytes_per_iteration = 0;
      hipLaunchKernelGGL(k_finalize, dim3(a.nblk, B), dim3(kTB), 0, st, a);
      *periods_run = -1;
      return hipGetLastError();
    }
    if (n_solved > 0) {
      // Per-scenario fallback: the scenarios it solved are done (ctrl.done, x+ / y+ exported); only the others go on - packed into as
      // few lane groups as they need when the lane form applies (its phases' slot -> scenario map), skipped by every other form
      // (their kernels leave scenarios with ctrl.done alone).  One infeasible member of 256 then costs one lane group of PDHG
      // iterations until its certificate, not the whole batch over again.
      std::vector<StreamCtrl> hc((size_t)B);
      if ((e = hipMemcpyAsync(hc.data(), a.W.ctrl, (size_t)B * sizeof(StreamCtrl), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
      for (int s = 0; s < B; ++s) if (!hc[s].done) left.push_back(s);
      if (left.empty()) {                              // (cannot happen: all == false; defensive)
        hipLaunchKernelGGL(k_finalize, dim3(a.nblk, B), dim3(kTB), 0, st, a);
        *periods_run = -1;
        return hipGetLastError();
      }
    }
  }
  const bool after_ipm = !left.empty();
  // mid-size LPs: the whole solve in ONE launch, one workgroup per scenario with its state in LDS
  {
    const size_t lds = ((size_t)7 * S->P.n + 7 * S->P.m + (1024 / 64) * kNQ + kNQ) * sizeof(double);
    static const int no_block = getenv("DSP_STREAM_NO_BLOCK") ? atoi(getenv("DSP_STREAM_NO_BLOCK")) : 0;
    const bool long_rows_ok = S->P.R.nlong == 0;      // long ROWS are not handled by the check of the block form
    if (!no_block && long_rows_ok && lds <= S->lds_limit) {
      const int nt = std::max(S->P.n, S->P.m) > 2048 ? 1024 : (std::max(S->P.n, S->P.m) > 512 ? 512 : 256);
      hipError_t be = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_block_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (be != hipSuccess) return be;
      S->last_form = DSP_STREAM_FORM_BLOCK;
      hipLaunchKernelGGL(k_block_solve, dim3(B), dim3(nt), lds, st, a);
      hipLaunchKernelGGL(k_finalize, dim3(a.nblk, B), dim3(kTB), 0, st, a);
      *periods_run = -1;
      return hipGetLastError();
    }
  }
  // banded matrix with a handful of long columns: the lane-per-scenario form (dsp_stream_lane.hip)
  {
    bool used = false;
    if ((e = lane_run(S, a, st, periods_run, &used, after_ipm ? &left : nullptr)) != hipSuccess) return e;
    if (used) {
      S->last_form = DSP_STREAM_FORM_LANE;
      hipLaunchKernelGGL(k_finalize, dim3(a.nblk, B), dim3(kTB), 0, st, a);
      return hipGetLastError();
    }
  }
  // scenarios per thread: enough (element blocks x scenario groups) to fill the chip, matrix reuse otherwise
  const long blocks1 = a.nblk;
  int sg = 4;
  if (blocks1 * ((B + 3) / 4) < 1024) sg = 2;
  if (blocks1 * ((B + 1) / 2) < 1024) sg = 1;
  a.nblk_tot = a.nblk + std::max(S->P.R.nlong, S->P.C.nlong);
  static const int sg_env = getenv("DSP_STREAM_SG") ? atoi(getenv("DSP_STREAM_SG")) : 0;     // development override
  if (sg_env == 1 || sg_env == 2 || sg_env == 4 || sg_env == 8) sg = sg_env;
  S->last_bytes_per_iteration = (size_t)8 * (8 * (size_t)S->P.n + 6 * (size_t)S->P.m);
  S->last_form = S->P.F.ntile > 0 ? DSP_STREAM_FORM_TILE : DSP_STREAM_FORM_TWO_LAUNCH;
  if (S->P.F.ntile > 0) {
    // banded matrix: one launch per plain iteration.  Bounds that are the same template for every scenario of the batch
    // (stride 0: the price-taker families differ in the objective only) are read once per workgroup from scenario 0's copy.
    const bool shared = (!batch.var_lb || batch.var_lb_stride == 0) && (!batch.var_ub || batch.var_ub_stride == 0) &&
                        (!batch.row_lb || batch.row_lb_stride == 0) && (!batch.row_ub || batch.row_ub_stride == 0);
    const bool qp = batch.row_compliance != nullptr;
    int fsg = 2;                       // scenarios per workgroup (matrix entries are loaded once per workgroup): 2 measured best with
    if ((long)S->P.F.ntile * ((B + 1) / 2) < 1024) fsg = 1;      // 250-row tiles (81 vs 89-91 us at 4, 102 at 1: profiles/r30j_fused_scan.log)
    static const int fsg_env = getenv("DSP_FUSED_SG") ? atoi(getenv("DSP_FUSED_SG")) : 0;
    if (fsg_env == 1 || fsg_env == 2) fsg = fsg_env;
    // the workgroup's LDS (staged y + xbar per scenario, the deferred form's thread-private slots) must fit a CU's 160 KB: plans
    // with wide hulls (the host admits up to 40 KB per scenario) run with fewer scenarios per workgroup
    {
      const FusedPlan &F = S->P.F;
      const size_t k_own = (size_t)(F.own_max + kTB - 1) / kTB;
      auto lds_need = [&](int g) {
        return ((size_t)g * (F.ny_max + F.nxb_max) + (size_t)(1 + kTB / 64) * kFusedMaxLong * g + k_own * g * (shared ? 2 : 4) * kTB) * sizeof(double);
      };
      while (fsg > 1 && lds_need(fsg) > (size_t)160 * 1024) fsg /= 2;
    }
    S->last_bytes_per_iteration = (size_t)8 * (shared ? 4 * (size_t)S->P.n + 3 * (size_t)S->P.m : 6 * (size_t)S->P.n + 5 * (size_t)S->P.m)
                                  + (qp ? 8 * (size_t)S->P.m : 0);
    if (fsg == 2) e = run_fused<2>(S, a, st, periods_run, shared, qp);
    else e = run_fused<1>(S, a, st, periods_run, shared, qp);
  } else
  if (sg == 8) e = run<8>(S, a, st, periods_run);
  else if (sg == 4) e = run<4>(S, a, st, periods_run);
  else if (sg == 2) e = run<2>(S, a, st, periods_run);
  else e = run<1>(S, a, st, periods_run);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_finalize, dim3(a.nblk, B), dim3(kTB), 0, st, a);
  return hipGetLastError();
}

}  // namespace dsp
