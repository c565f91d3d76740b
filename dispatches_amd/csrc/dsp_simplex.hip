// dsp_simplex.hip — in-wave dense simplex for the TINY LPs of the double loop (gfx950, wave64).
//
// 24 of the 25 solves of a simulated day are hourly LPs with 20-110 columns + rows (4-h real-time bids, 12-h nuclear
// real-time bids, 4-h tracking: SURVEY.md 3.3).  A first-order method needs thousands of iterations on them (1e4
// penalty columns against 1e-2 costs) and fails outright when the dispatch signal cannot be met; a vertex method needs
// ~(n + m) / 2 pivots and returns the exact vertex - the same kind of answer the reference's CBC / Xpress return.
//
// One LP per wave.  Bounded-variable primal simplex on the dense tableau (executable specification:
// tools/simplex_proto.py, which this file follows step by step):
//     min c.x  s.t.  rlo <= A x <= rhi,  lb <= x <= ub                 (scaled by the handle's D_r, D_c)
//     z = (x, s),  s = A x in [rlo, rhi];   start: all slacks basic, structurals at their finite bound nearest 0
//     phase 1: minimise the sum of bound violations of the basic variables; an infeasible basic blocks when it
//              reaches the bound it violates
//     phase 2: Dantzig pricing on reduced costs RECOMPUTED from the tableau every pivot (no drift), two-pass ratio test
//              (minimum ratio, then the largest pivot among the ties), bound flips
// Layout: the tableau T [m][N] (N = n + m <= 128) lives in the wave's LDS with an odd row stride, so that both access
// patterns are bank-conflict free: lane j sweeping its column(s) j, j + 64 (pricing, pivot update) and lane i reading
// its row's entry of the entering column (ratio test).  Lane i < m also owns row i's basic variable (value, bounds,
// index), lane j owns column j's status.  All control flow is wave-uniform; reductions run on the VALU (DPP).
// A solved LP is certified against the ORIGINAL rows before it is reported optimal; anything else (pivot limit,
// failed certificate) is left to the PDLP kernel that runs afterwards on the scenarios still marked unsolved.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "dsp_device.hpp"
#include "dsp_wave.hpp"

namespace dsp {

namespace {

constexpr double kBig = 1e300;

__device__ __forceinline__ double lds_d(const double *base, int idx) { return base[idx]; }

// lowest set lane of a wave-uniform 64-bit ballot
__device__ __forceinline__ int first_lane(unsigned long long mask) { return __ffsll((long long)mask) - 1; }

template <int CQ>
__global__ void __launch_bounds__(64, CQ == 1 ? 4 : 3) simplex_kernel(SimplexArgs a) {
  // CQ == 1 (the 4-h wind + battery LPs, 20 x 54): 128 VGPRs and 9.5 KB of LDS -> 4 waves per SIMD, 16 per CU: a 4096-plant batch is ONE
  // round of waves over the 256 CUs (at 137 VGPRs / 14.6 KB it was 12 per CU, a full round and a quarter-full one)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const dsp_batch &b = a.b;
  const int lane = threadIdx.x;
  const int n = a.n, m = a.m, N = n + m;
  const int RS = a.row_stride;                       // odd number of doubles per tableau row
  double *T = reinterpret_cast<double *>(smem);      // [m][RS]
  const int mp = (m + 1) & ~1, Np = (N + 1) & ~1;
  double *alpha_s = T + (size_t)m * RS;              // [mp]  entering column
  double *cB_s = alpha_s + mp;                       // [mp]  cost of each row's basic variable (current phase)
  double *xval = cB_s + mp;                          // [Np]  value of every variable (final assembly / start)
  double *scal = xval + Np;                          // [8]   broadcast scalars
  int *iscal = reinterpret_cast<int *>(scal + 8);    // [8]

  for (int s = blockIdx.x; s < b.B; s += gridDim.x) {
    // Warm start (a.warm == 1 and a saved state for this scenario): the hourly LPs of a plant's rolling loop share ONE matrix and
    // differ in costs, bounds and right-hand sides only; the final basis of the previous hour - its tableau B^-1 [A | I] kept in HBM -
    // is 2 - 4 pivots from the next hour's optimum where the slack basis is ~26 (HiGHS primal simplex on a recorded plant: cold
    // 26.2 / 27.4, hot 3.9 / 1.9 iterations per real-time / tracking LP).  The same pivot rules run from there (phase 1 if the new
    // bounds leave a basic value outside); a warm attempt that does not end in a CERTIFIED optimum is repeated from the slack basis.
    bool use_warm = a.warm == 1 && a.warm_valid[s] != 0;
    int pivots_before = 0;
  restart:
    // ---- this scenario's data: columns j = lane + 64 q (structural j < n, slack n <= j < N) ---------------------
    double lo[CQ], hi[CQ], cost[CQ], val[CQ];
    bool basic[CQ], upper[CQ], fixedv[CQ], live[CQ];
    double cabs = 0.0;
    bool bad = false;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const int j = lane + 64 * q;
      live[q] = j < N;
      lo[q] = 0.0; hi[q] = 0.0; cost[q] = 0.0;
      if (j < n) {
        const double d = a.col_scale[j];
        const double cu = b.c[(size_t)s * b.c_stride + j];
        const double lu = b.var_lb ? b.var_lb[(size_t)s * b.var_lb_stride + j] : -INFINITY;
        const double uu = b.var_ub ? b.var_ub[(size_t)s * b.var_ub_stride + j] : INFINITY;
        cost[q] = cu * d; lo[q] = lu / d; hi[q] = uu / d;
        if (!(lu <= uu) || !(cu == cu)) bad = true;
      } else if (j < N) {
        const int i = j - n;
        const double d = a.row_scale[i];
        const double l = b.row_lb ? b.row_lb[(size_t)s * b.row_lb_stride + i] : -INFINITY;
        const double u = b.row_ub ? b.row_ub[(size_t)s * b.row_ub_stride + i] : INFINITY;
        lo[q] = l * d; hi[q] = u * d;
        if (!(l <= u)) bad = true;
      }
      cabs = fmax(cabs, fabs(cost[q]));
      fixedv[q] = lo[q] == hi[q];
      basic[q] = live[q] && j >= n;
      // nonbasic start: the finite bound nearest to zero (free columns start at 0)
      const bool lo_f = is_finite(lo[q]), hi_f = is_finite(hi[q]);
      double v = 0.0;
      if (lo_f && (!hi_f || fabs(lo[q]) <= fabs(hi[q]))) v = lo[q];
      else if (hi_f) v = hi[q];
      val[q] = v;
      upper[q] = hi_f && v == hi[q] && !(v == lo[q]);
    }
    if (__ballot(bad) != 0ull) {
      // crossed bounds / NaN input: flagged exactly as the PDLP kernel does, nothing is iterated
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < n) b.x[(size_t)s * n + j] = NAN; }
      if (lane < m) b.y[(size_t)s * m + lane] = NAN;
      bool nanc = false;
#pragma unroll
      for (int q = 0; q < CQ; ++q) nanc |= !(cost[q] == cost[q]);
      const bool any_nan = __ballot(nanc) != 0ull;
      if (lane == 0) {
        b.obj[s] = NAN;
        b.status[s] = any_nan ? DSP_STATUS_NUMERICAL : DSP_STATUS_PRIMAL_INFEASIBLE;
        if (b.iters) b.iters[s] = 0;
        if (b.jumps) b.jumps[s] = 0;
        if (b.flags) b.flags[s] = 0;
      }
      continue;
    }
    const double ctol = a.tol_d * (1.0 + wave_max(cabs));
    int bvar = (lane < m) ? n + lane : -1;
    double beta = 0.0, blo = -INFINITY, bhi = INFINITY;
    if (use_warm) {
      // ---- tableau, basis and nonbasic sides of the saved vertex; values from the NEW bounds -----------------------------
      const double *Ts = a.warm_T + (size_t)s * m * N;
      if (lane < m) bvar = a.warm_basis[(size_t)s * m + lane];
      for (int i0 = 0; i0 < m; i0 += 4) {                              // four rows' loads in flight together (HBM latency, not bytes)
        double tv[4][CQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = min(i0 + u, m - 1);
#pragma unroll
          for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; tv[u][q] = (j < N) ? Ts[(size_t)i * N + j] : 0.0; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u < m) {
#pragma unroll
            for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) T[(i0 + u) * RS + j] = tv[u][q]; }
          }
        }
      }
      // bounds of each row's basic variable (may be a structural): looked up through the value buffer, lower then upper
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = lo[q]; }
      wave_lds_fence();
      if (lane < m) blo = xval[bvar];
      wave_lds_fence();
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = hi[q]; }
      wave_lds_fence();
      if (lane < m) bhi = xval[bvar];
      wave_lds_fence();
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = 0.0; }
      wave_lds_fence();
      if (lane < m) xval[bvar] = 1.0;                                  // marks the basic columns
      wave_lds_fence();
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        const int j = lane + 64 * q;
        basic[q] = live[q] && xval[j] != 0.0;
        const bool lo_f = is_finite(lo[q]), hi_f = is_finite(hi[q]);
        bool up = live[q] && a.warm_upper[(size_t)s * N + j] != 0 && hi_f && !fixedv[q];
        if (!lo_f && hi_f) up = true;                                  // (a side that is no longer finite: the other one)
        val[q] = up ? hi[q] : (lo_f ? lo[q] : 0.0);
        upper[q] = up;
      }
      wave_lds_fence();
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = basic[q] ? 0.0 : val[q]; }
      wave_lds_fence();
      if (lane < m) {
#pragma unroll 8
        for (int j = 0; j < N; ++j) beta = fma(-T[lane * RS + j], xval[j], beta);       // basic_i + sum over nonbasic T_ij z_j = 0
      }
    } else {
      // ---- tableau: row i = s_i - sum_j a_ij x_j = 0, basis = slacks --------------------------------------------------------
      for (int i0 = 0; i0 < m; i0 += 4) {
        double tv[4][CQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = min(i0 + u, m - 1);
#pragma unroll
          for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; tv[u][q] = (j < n) ? -a.A_dense[(size_t)i * n + j] : ((j - n == i) ? 1.0 : 0.0); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u < m) {
#pragma unroll
            for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) T[(i0 + u) * RS + j] = tv[u][q]; }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = val[q]; }
      wave_lds_fence();
      // row lanes: basic variable of row i (slack n + i), its value s_i = sum_j a_ij x_j and bounds
      if (lane < m) {
#pragma unroll 8
        for (int j = 0; j < n; ++j) beta = fma(-T[lane * RS + j], xval[j], beta);
        const double d = a.row_scale[lane];
        const double l = b.row_lb ? b.row_lb[(size_t)s * b.row_lb_stride + lane] : -INFINITY;
        const double u = b.row_ub ? b.row_ub[(size_t)s * b.row_ub_stride + lane] : INFINITY;
        blo = l * d; bhi = u * d;
      }
    }
    const int it_limit = use_warm ? min(a.max_pivots, 6 * N) : a.max_pivots;      // (a warm attempt gives up early: the slack basis needs ~N / 2)
    int status = -1, pivots = pivots_before;
    bool p1_priced_out = false;        // phase 1 ended because no column prices in (the only phase-1 stop that proves infeasibility)
    bool phase1 = false;

    int refines = 0;
    double res_c = 0.0, mag_c = 0.0;   // the refinement's row residual (and the magnitude of its terms): the certificate's, unless pivots followed
    bool res_valid = false;
    bool tight = false;                // after a refinement of the basic values: feasibility to 1e-3 of the working tolerance (below)
  pivot:
    for (int it = 0;; ++it) {
      // ---- phase and basic costs --------------------------------------------------------------------------------
      const double ptol = (tight ? 1e-3 : 1.0) * a.tol_p * (1.0 + fmax(fabs(finite_or_zero(blo)), fabs(finite_or_zero(bhi))));
      const bool below = lane < m && beta < blo - ptol;
      const bool above = lane < m && beta > bhi + ptol;
      phase1 = __ballot(below || above) != 0ull;
      if (lane < m) {
        double cb;
        if (phase1) cb = below ? -1.0 : (above ? 1.0 : 0.0);
        else cb = 0.0;                                  // filled below from the owning column lane (phase 2)
        cB_s[lane] = cb;
      }
      wave_lds_fence();
      if (!phase1) {
        // phase 2: cost of a basic variable lives with its column owner: scatter it to the row that holds it
#pragma unroll
        for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = cost[q]; }
        wave_lds_fence();
        if (lane < m) cB_s[lane] = xval[bvar];
        wave_lds_fence();
      }
      // ---- pricing: d_j = c_j - sum_i cB_i T[i][j] ---------------------------------------------------------------
      double dj[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) dj[q] = phase1 ? 0.0 : cost[q];
#pragma unroll 4
      for (int i = 0; i < m; ++i) {               // loads only: unrolled so that four rows' LDS reads are in flight together
        const double cb = cB_s[i];
#pragma unroll
        for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) dj[q] = fma(-cb, T[i * RS + j], dj[q]); }
      }
      const double dtol = phase1 ? 1e-9 : ctol;
      double score[CQ];
      double best = -1.0;
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        const bool free_col = !is_finite(lo[q]) && !is_finite(hi[q]);
        const bool elig = live[q] && !basic[q] && !fixedv[q] &&
                          (free_col ? fabs(dj[q]) > dtol : ((!upper[q] && dj[q] < -dtol) || (upper[q] && dj[q] > dtol)));
        score[q] = elig ? fabs(dj[q]) : -1.0;
        best = fmax(best, score[q]);
      }
      best = wave_max(best);
      if (best < 0.0) { status = phase1 ? DSP_STATUS_PRIMAL_INFEASIBLE : DSP_STATUS_OPTIMAL; p1_priced_out = phase1; break; }
      if (it >= it_limit) { status = -1; break; }
      // entering column: the smallest index among the maxima (as numpy's argmax in the prototype)
      int jin = -1;
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        if (jin < 0) {
          const unsigned long long mk = __ballot(score[q] == best);
          if (mk) jin = first_lane(mk) + 64 * q;
        }
      }
      const int jl = jin & 63, jq = jin >> 6;
      // its owner broadcasts direction, bounds and value
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        if (q == jq && lane == jl) {
          const bool free_col = !is_finite(lo[q]) && !is_finite(hi[q]);
          const bool down = free_col ? (dj[q] > 0.0) : upper[q];
          scal[0] = down ? -1.0 : 1.0;
          scal[1] = lo[q]; scal[2] = hi[q]; scal[3] = val[q];
        }
      }
      wave_lds_fence();
      const double sgn = scal[0], jlo = scal[1], jhi = scal[2], jval = scal[3];
      // ---- ratio test on the row lanes ----------------------------------------------------------------------------
      const double al = (lane < m) ? T[lane * RS + jin] : 0.0;
      if (lane < m) alpha_s[lane] = al;
      const double amax = wave_max(fabs(al));
      const double ptv = a.tol_piv * fmax(1.0, amax);
      const double delta = -sgn * al;
      const bool dec = delta < -ptv, inc = delta > ptv;
      const bool feas = !below && !above;
      double ti = kBig;
      if (lane < m) {
        if (feas && dec && is_finite(blo)) ti = (beta - blo) / -delta;
        if (feas && inc && is_finite(bhi)) ti = (bhi - beta) / delta;
        if (below && inc) ti = (blo - beta) / delta;
        if (above && dec) ti = (beta - bhi) / -delta;
        ti = fmax(ti, 0.0);
      }
      double tflip = jhi - jlo;
      if (!is_finite(tflip)) tflip = kBig;
      const double trow = wave_min(ti);
      const double tmin = fmin(trow, tflip);
      if (tmin >= kBig) { status = phase1 ? DSP_STATUS_PRIMAL_INFEASIBLE : DSP_STATUS_DUAL_INFEASIBLE; break; }
      const bool flip = tflip <= trow;
      // second pass: the largest pivot among the rows within a hair of the minimum
      const bool tie = lane < m && ti <= tmin * (1.0 + 1e-9) + 1e-12;
      const double pv = wave_max(tie ? fabs(al) : -1.0);
      const int r = first_lane(__ballot(tie && fabs(al) == pv));
      // ---- move ---------------------------------------------------------------------------------------------------
      if (lane < m) beta = fma(delta, tmin, beta);
      const double newval = fma(sgn, tmin, jval);
      if (flip) {
#pragma unroll
        for (int q = 0; q < CQ; ++q)
          if (q == jq && lane == jl) { val[q] = newval; upper[q] = !upper[q]; }
        continue;
      }
      // leaving variable: the basic of row r, to the bound it reached
      if (lane == r) {
        const bool hi_f = is_finite(bhi), lo_f = is_finite(blo);
        bool to_up = fabs(beta - bhi) < fabs(beta - blo);
        if (hi_f && !lo_f) to_up = true;
        if (!hi_f) to_up = false;
        iscal[0] = bvar;
        iscal[1] = to_up ? 1 : 0;
        scal[4] = to_up ? bhi : blo;
        bvar = jin; beta = newval; blo = jlo; bhi = jhi;
      }
      wave_lds_fence();
      const int jout = iscal[0];
      const bool out_up = iscal[1] != 0;
      const double out_val = scal[4];
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        const int j = lane + 64 * q;
        if (j == jout) { basic[q] = false; val[q] = out_val; upper[q] = out_up; }
        if (j == jin) basic[q] = true;
      }
      // ---- tableau update: row r /= alpha_r, rows i != r -= alpha_i * row r -------------------------------------
      const double inv = 1.0 / alpha_s[r];
      double prow[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; prow[q] = (j < N) ? T[r * RS + j] * inv : 0.0; }
      // four rows per step: all their LDS reads are issued before the first dependent store (the compiler cannot
      // prove that the rows do not alias and would otherwise serialise load -> fma -> store per row)
      for (int i0 = 0; i0 < m; i0 += 4) {
        double ai[4], tv[4][CQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = min(i0 + u, m - 1);
          ai[u] = alpha_s[i];
#pragma unroll
          for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; tv[u][q] = (j < N) ? T[i * RS + j] : 0.0; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u;
          if (i < m) {
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
              const int j = lane + 64 * q;
              if (j < N) T[i * RS + j] = (i == r) ? prow[q] : fma(-ai[u], prow[q], tv[u][q]);
            }
          }
        }
      }
      wave_lds_fence();
      ++pivots;
    }

    // ---- refinement of an optimal vertex' basic values ------------------------------------------------------------
    // The basic values are carried through the pivots (and, warm, through the hours): on 1e5-kW variables they end ~1e-7 off the
    // original rows - a thousand times inside every tolerance here, but a row that is short by 3e-7 kW can be one whose exact solution
    // pays the 1e4 $/MWh under-delivery penalty on those 3e-7 kW: 3e-6 $ on an hourly objective of 0.5 $ (seen in the year-long oracle
    // check, plant 4095 hour 317).  One step of iterative refinement with the tableau's own B^-1 (its slack columns): residual of the
    // ORIGINAL rows at the vertex, basic values corrected by B^-1 x residual; a corrected value that now lies outside its bounds by more
    // than 1e-3 of the working tolerance sends the vertex back into the pivot loop (phase 1 on that row) under the tight tolerance.
    if (status == DSP_STATUS_OPTIMAL && refines < 2) {
      ++refines;
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = val[q]; }
      wave_lds_fence();
      if (lane < m) xval[bvar] = beta;
      wave_lds_fence();
      if (lane < m) {
        double res = -xval[n + lane], mag = fabs(xval[n + lane]);
#pragma unroll 8
        for (int j = 0; j < n; ++j) {
          const double t = a.A_dense[(size_t)lane * n + j] * xval[j];
          res += t; mag += fabs(t);
        }
        cB_s[lane] = res;
        res_c = res; mag_c = mag;
      }
      res_valid = true;
      wave_lds_fence();
      bool out = false;
      if (lane < m) {
        double d = 0.0;
#pragma unroll 4
        for (int i = 0; i < m; ++i) d = fma(T[lane * RS + n + i], cB_s[i], d);
        beta += d;
        const double tt = 1e-3 * a.tol_p * (1.0 + fmax(fabs(finite_or_zero(blo)), fabs(finite_or_zero(bhi))));
        out = beta < blo - tt || beta > bhi + tt;
      }
      wave_lds_fence();
      if (__ballot(out) != 0ull) { tight = true; status = -1; res_valid = false; goto pivot; }
    }

    // ---- assemble the vertex, certify it against the original rows, store ----------------------------------------
#pragma unroll
    for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = val[q]; }
    wave_lds_fence();
    if (lane < m) xval[bvar] = beta;
    wave_lds_fence();
    double xs[CQ];
    double po = 0.0;
    bool okv = true, okb = true, okr = true;
    double viol = 0.0;                                        // largest bound violation of the vertex, relative to its variable's scale
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const int j = lane + 64 * q;
      xs[q] = (j < N) ? xval[j] : 0.0;
      if (j < N) {
        // same scale as the simplex' own feasibility tolerance (tol_p (1 + the variable's larger finite bound)), 10x looser
        const double scl = 1.0 + fmax(fabs(xs[q]), fmax(fabs(finite_or_zero(lo[q])), fabs(finite_or_zero(hi[q]))));
        const double tol = 1e-9 * scl;
        if (xs[q] < lo[q] - tol || xs[q] > hi[q] + tol || !(xs[q] == xs[q])) { okv = false; okb = false; }
        viol = fmax(viol, fmax(lo[q] - xs[q], xs[q] - hi[q]) / scl);
      }
      if (j < n) po = fma(cost[q], xs[q], po);
    }
    po = wave_sum(po);
    {
      // row residuals with the ORIGINAL scaled matrix, sum_j a_ij x_j - s_i, against the scale of the whole vertex: the
      // basic values carry an ABSOLUTE drift of ~1e-11 x the largest number in the tableau (1e5 kWh states), which lands
      // on rows whose own terms may all be ~0 (seen: 4.9e-6 on such a row, 5e-11 of the state-of-charge scale).  The
      // certificate is there to catch a broken tableau, not rounding.
      // (a vertex that was refined and not pivoted on since: the residual BEFORE the correction is the statement about the tableau)
      double res = res_c, mag = mag_c;
      if (lane < m && !res_valid) {
        res = -xval[n + lane]; mag = fabs(xval[n + lane]);
#pragma unroll 8
        for (int j = 0; j < n; ++j) {
          const double t = a.A_dense[(size_t)lane * n + j] * xval[j];
          res += t; mag += fabs(t);
        }
      }
      const double gmag = wave_max(mag);
      if (lane < m && !(fabs(res) <= 1e-9 * (1.0 + gmag))) { okv = false; okr = false; }
    }
    const bool certified = __ballot(!okv) == 0ull;
    const unsigned long long bad_bounds = __ballot(!okb), bad_any = __ballot(!okv);
    if (use_warm && !(status == DSP_STATUS_OPTIMAL && certified)) {
      // a warm attempt that did not end in a certified optimum (a drifted tableau, a stop that proves nothing): once more from the slack basis
      use_warm = false;
      pivots_before = pivots;
      wave_lds_fence();
      goto restart;
    }
    if (a.warm) {
      // the final basis for the next solve of this scenario (only a certified optimum is worth starting from)
      const bool keep = status == DSP_STATUS_OPTIMAL && certified;
      if (keep) {
        double *Ts = a.warm_T + (size_t)s * m * N;
        for (int i = 0; i < m; ++i) {
#pragma unroll
          for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) Ts[(size_t)i * N + j] = T[i * RS + j]; }
        }
        if (lane < m) a.warm_basis[(size_t)s * m + lane] = bvar;
#pragma unroll
        for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) a.warm_upper[(size_t)s * N + j] = (!basic[q] && upper[q]) ? 1 : 0; }
      }
      if (lane == 0) a.warm_valid[s] = keep ? 1 : 0;
    }
    int reason = status == -1 ? 1 : 0;                       // 1 = pivot limit
    if (status == DSP_STATUS_OPTIMAL && !certified) { status = -1; reason = 2; }   // 2 = vertex failed its certificate
    // 3 = a phase-1 stop / an empty ratio test: only OPTIMAL vertices carry a certificate against the original rows, and on a
    // degenerate hourly LP either stop can be a tolerance artefact (no |d_j| above 1e-9, every ratio below the pivot tolerance).
    // Not reported as infeasible / unbounded: the PDLP pass takes the scenario (a genuinely infeasible LP ends there at the
    // iteration limit - dispatch LPs carry slack columns, so that is an input error, not a regular outcome).
    // ... EXCEPT a phase-1 optimum whose vertex is clearly outside its bounds (> 1e-6 of the variable's scale, a thousand times the
    // feasibility tolerance) on an intact tableau (row residuals certified): that is a proof of infeasibility, reported as such -
    // a caller that hands over an infeasible hourly LP gets status 2 like from the reference's CBC, not an iteration limit.
    // Only a phase 1 that ended because NO column prices in (all reduced costs of the infeasibility sum within tolerance: its
    // dual feasibility) counts; an empty ratio test in phase 1 is an artefact and goes to the PDLP pass, which has certificates of its own.
    const bool clearly_infeasible = status == DSP_STATUS_PRIMAL_INFEASIBLE && p1_priced_out && __ballot(!okr) == 0ull && wave_max(viol) > 1e-6;
    if (clearly_infeasible) {
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < n) b.x[(size_t)s * n + j] = NAN; }
      if (lane < m) b.y[(size_t)s * m + lane] = NAN;
      if (lane == 0) {
        b.obj[s] = NAN;
        b.status[s] = DSP_STATUS_PRIMAL_INFEASIBLE;
        if (b.iters) b.iters[s] = pivots;
        if (b.jumps) b.jumps[s] = 0;
        if (b.flags) b.flags[s] = 0;
      }
      wave_lds_fence();
      continue;
    }
    if (status == DSP_STATUS_PRIMAL_INFEASIBLE || status == DSP_STATUS_DUAL_INFEASIBLE) { status = -1; reason = 3; }
    if (status == -1) {
      if (lane == 0) {
        if (a.debug_keep) {                                  // development: report why instead of handing over to PDLP
          b.status[s] = 50 + reason;
          if (b.iters) b.iters[s] = pivots;
          if (b.jumps) b.jumps[s] = bad_bounds ? 1000 + first_lane(bad_bounds) : (bad_any ? 2000 + first_lane(bad_any) : 0);
          b.obj[s] = po;
          if (b.flags) b.flags[s] = 0;
        } else {
          b.status[s] = DSP_STATUS_UNSOLVED;                 // the PDLP kernel takes it from here
          atomicAdd(a.unsolved, 1);
        }
      }
      continue;
    }
    // duals: y_i = reduced cost of slack i with the phase-2 costs (recomputed so that a phase-1 stop reports something sane)
#pragma unroll
    for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = cost[q]; }
    wave_lds_fence();
    if (lane < m) cB_s[lane] = xval[bvar];
    wave_lds_fence();
    double dsl[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) dsl[q] = cost[q];
#pragma unroll 4
    for (int i = 0; i < m; ++i) {
      const double cb = cB_s[i];
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) dsl[q] = fma(-cb, T[i * RS + j], dsl[q]); }
    }
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const int j = lane + 64 * q;
      if (j < n) b.x[(size_t)s * n + j] = xs[q] * a.col_scale[j];
      else if (j < N) b.y[(size_t)s * m + (j - n)] = dsl[q] * a.row_scale[j - n];
    }
    if (lane == 0) {
      b.obj[s] = po;
      b.status[s] = status;
      if (b.iters) b.iters[s] = pivots;
      if (b.jumps) b.jumps[s] = 0;
      if (b.flags) b.flags[s] = 0;
    }
    wave_lds_fence();
  }
}


// ---- the same simplex with the tableau in REGISTERS ------------------------------------------------------------------------
// The LDS-tableau kernel above is latency-bound: a LONE wave needs 4.2 us per pivot of a 20 x 54 tableau (126 us for 30
// pivots at one LP per CU, profiles/r04c_simplex_scaling.log) - every pivot is a chain of LDS round trips (basic costs,
// pricing sweep, entering column, broadcast scalars, row-by-row update).  Here lane j (+ 64 q) holds COLUMN j of the tableau
// in VGPRs (MR rows x CQ slots), so
//   * pricing is MR FMAs with the basic costs read lane by lane (v_readlane -> SGPR), no memory at all;
//   * the entering column is one lane's registers: its MR entries are read with v_readlane once and serve both as the
//     wave-uniform multipliers of the update and (selected by lane id) as the row lanes' ratio-test entries;
//   * the pivot row T[r][.] is picked out of the register file with MR uniform selects, the update is MR x CQ FMAs;
//   * every broadcast (entering bounds / value / cost, leaving variable) is a v_readlane.
// LDS is only the 1 KB variable-value buffer used to assemble and certify the vertex.  Same pivot rules, same certificate,
// same pivot counts.  Measured (profiles/r04d_simplex_reg.log): 4-h wind+battery LPs (20 x 54), 1024 of them 141 -> 102 us
// (real-time bids) / 161 -> 115 us (tracking), a lone LP 126 -> 93 us; but a pivot is still ~1000 VALU instructions of ONE
// wave (2 readlanes + 2 selects per row and use), the kernel needs 256 VGPRs (2 waves per SIMD: a 4096-batch takes two
// passes, 286 vs 264 us) and the 36-row nuclear LP spills (841 vs 515 us).  So it is used where it wins - at most 24 rows,
// at most 64 columns + rows, batches up to 2048 (the per-GPU shards of the double loop, the Tracker's single LPs) - and the
// LDS-tableau kernel everywhere else.
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <int CQ, int MR>
__global__ void __launch_bounds__(64, 2) simplex_reg_kernel(SimplexArgs a) {
  __shared__ double xval[128];                       // value (or cost) of every variable: vertex assembly / certificate
  const dsp_batch &b = a.b;
  const int lane = threadIdx.x;
  const int n = a.n, m = a.m, N = n + m;

  for (int s = blockIdx.x; s < b.B; s += gridDim.x) {
    double lo[CQ], hi[CQ], cost[CQ], val[CQ];
    bool basic[CQ], upper[CQ], fixedv[CQ], live[CQ];
    double cabs = 0.0;
    bool bad = false;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const int j = lane + 64 * q;
      live[q] = j < N;
      lo[q] = 0.0; hi[q] = 0.0; cost[q] = 0.0;
      if (j < n) {
        const double d = a.col_scale[j];
        const double cu = b.c[(size_t)s * b.c_stride + j];
        const double lu = b.var_lb ? b.var_lb[(size_t)s * b.var_lb_stride + j] : -INFINITY;
        const double uu = b.var_ub ? b.var_ub[(size_t)s * b.var_ub_stride + j] : INFINITY;
        cost[q] = cu * d; lo[q] = lu / d; hi[q] = uu / d;
        if (!(lu <= uu) || !(cu == cu)) bad = true;
      } else if (j < N) {
        const int i = j - n;
        const double d = a.row_scale[i];
        const double l = b.row_lb ? b.row_lb[(size_t)s * b.row_lb_stride + i] : -INFINITY;
        const double u = b.row_ub ? b.row_ub[(size_t)s * b.row_ub_stride + i] : INFINITY;
        lo[q] = l * d; hi[q] = u * d;
        if (!(l <= u)) bad = true;
      }
      cabs = fmax(cabs, fabs(cost[q]));
      fixedv[q] = lo[q] == hi[q];
      basic[q] = live[q] && j >= n;
      const bool lo_f = is_finite(lo[q]), hi_f = is_finite(hi[q]);
      double v = 0.0;
      if (lo_f && (!hi_f || fabs(lo[q]) <= fabs(hi[q]))) v = lo[q];
      else if (hi_f) v = hi[q];
      val[q] = v;
      upper[q] = hi_f && v == hi[q] && !(v == lo[q]);
    }
    if (__ballot(bad) != 0ull) {
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < n) b.x[(size_t)s * n + j] = NAN; }
      if (lane < m) b.y[(size_t)s * m + lane] = NAN;
      bool nanc = false;
#pragma unroll
      for (int q = 0; q < CQ; ++q) nanc |= !(cost[q] == cost[q]);
      const bool any_nan = __ballot(nanc) != 0ull;
      if (lane == 0) {
        b.obj[s] = NAN;
        b.status[s] = any_nan ? DSP_STATUS_NUMERICAL : DSP_STATUS_PRIMAL_INFEASIBLE;
        if (b.iters) b.iters[s] = 0;
        if (b.jumps) b.jumps[s] = 0;
        if (b.flags) b.flags[s] = 0;
      }
      continue;
    }
    const double ctol = a.tol_d * (1.0 + wave_max(cabs));
    // ---- tableau columns in registers: row i = s_i - sum_j a_ij x_j = 0, basis = slacks (coalesced loads per row) --------
    double T[MR][CQ];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        const int j = lane + 64 * q;
        double t = 0.0;
        if (i < m && j < n) t = -a.A_dense[(size_t)i * n + j];
        else if (i < m && j < N && j - n == i) t = 1.0;
        T[i][q] = t;
      }
    }
    // row lanes: basic variable of row i (slack n + i), its value s_i = sum_j a_ij x_j, bounds, phase-2 cost of the basic
    int bvar = (lane < m) ? n + lane : -1;
    double beta = 0.0, blo = -INFINITY, bhi = INFINITY, cbas = 0.0;
#pragma unroll
    for (int i0 = 0; i0 < MR; i0 += 4) {
      double part[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < n) t = fma(T[i0 + u][q], val[q], t); }
        part[u] = t;
      }
      wave_sums<4>(part);
#pragma unroll
      for (int u = 0; u < 4; ++u) if (lane == i0 + u) beta = -part[u];
    }
    if (lane < m) {
      const double d = a.row_scale[lane];
      const double l = b.row_lb ? b.row_lb[(size_t)s * b.row_lb_stride + lane] : -INFINITY;
      const double u = b.row_ub ? b.row_ub[(size_t)s * b.row_ub_stride + lane] : INFINITY;
      blo = l * d; bhi = u * d;
    } else {
      beta = 0.0;
    }
    int status = -1, pivots = 0;
    bool p1_priced_out = false;        // phase 1 ended because no column prices in (the only phase-1 stop that proves infeasibility)
    bool phase1 = false;

    int refines = 0;
    bool tight = false;                // (see the LDS-tableau kernel: refinement of an optimal vertex' basic values)
  pivot:
    for (int it = 0;; ++it) {
      const double ptol = (tight ? 1e-3 : 1.0) * a.tol_p * (1.0 + fmax(fabs(finite_or_zero(blo)), fabs(finite_or_zero(bhi))));
      const bool below = lane < m && beta < blo - ptol;
      const bool above = lane < m && beta > bhi + ptol;
      phase1 = __ballot(below || above) != 0ull;
      const double cb_lane = (lane < m) ? (phase1 ? (below ? -1.0 : (above ? 1.0 : 0.0)) : cbas) : 0.0;
      // ---- pricing: d_j = c_j - sum_i cB_i T[i][j], basic costs read lane by lane ---------------------------------------
      double dj[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) dj[q] = phase1 ? 0.0 : cost[q];
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const double cb = readlane_f64(cb_lane, i);
#pragma unroll
        for (int q = 0; q < CQ; ++q) dj[q] = fma(-cb, T[i][q], dj[q]);
      }
      const double dtol = phase1 ? 1e-9 : ctol;
      double score[CQ];
      double best = -1.0;
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        const bool free_col = !is_finite(lo[q]) && !is_finite(hi[q]);
        const bool elig = live[q] && !basic[q] && !fixedv[q] &&
                          (free_col ? fabs(dj[q]) > dtol : ((!upper[q] && dj[q] < -dtol) || (upper[q] && dj[q] > dtol)));
        score[q] = elig ? fabs(dj[q]) : -1.0;
        best = fmax(best, score[q]);
      }
      best = wave_max(best);
      if (best < 0.0) { status = phase1 ? DSP_STATUS_PRIMAL_INFEASIBLE : DSP_STATUS_OPTIMAL; p1_priced_out = phase1; break; }
      if (it >= a.max_pivots) { status = -1; break; }
      int jin = -1;
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        if (jin < 0) {
          const unsigned long long mk = __ballot(score[q] == best);
          if (mk) jin = first_lane(mk) + 64 * q;
        }
      }
      const int jl = jin & 63, jq = jin >> 6;
      // the entering column's owner: direction, bounds, value, cost (v_readlane from lane jl, slot jq)
      double e_sgn = 0.0, e_lo = 0.0, e_hi = 0.0, e_val = 0.0, e_cost = 0.0;
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        if (q == jq) {
          const bool free_col = !is_finite(lo[q]) && !is_finite(hi[q]);
          const bool down = free_col ? (dj[q] > 0.0) : upper[q];
          e_sgn = down ? -1.0 : 1.0; e_lo = lo[q]; e_hi = hi[q]; e_val = val[q]; e_cost = cost[q];
        }
      }
      const double sgn = readlane_f64(e_sgn, jl), jlo = readlane_f64(e_lo, jl), jhi = readlane_f64(e_hi, jl);
      const double jval = readlane_f64(e_val, jl), jcost = readlane_f64(e_cost, jl);
      // ---- entering column: row lane i picks up entry i of lane jl's column (read again, wave-uniform, in the update:
      // keeping all MR of them between the two places costs 2 MR registers the allocator does not have) --------------------
      double al = 0.0;
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        double t = T[i][0];
#pragma unroll
        for (int q = 1; q < CQ; ++q) t = (jq == q) ? T[i][q] : t;
        const double ai = readlane_f64(t, jl);
        al = (lane == i) ? ai : al;
      }
      const double amax = wave_max(fabs(al));
      const double ptv = a.tol_piv * fmax(1.0, amax);
      const double delta = -sgn * al;
      const bool dec = delta < -ptv, inc = delta > ptv;
      const bool feas = !below && !above;
      double ti = kBig;
      if (lane < m) {
        if (feas && dec && is_finite(blo)) ti = (beta - blo) / -delta;
        if (feas && inc && is_finite(bhi)) ti = (bhi - beta) / delta;
        if (below && inc) ti = (blo - beta) / delta;
        if (above && dec) ti = (beta - bhi) / -delta;
        ti = fmax(ti, 0.0);
      }
      double tflip = jhi - jlo;
      if (!is_finite(tflip)) tflip = kBig;
      const double trow = wave_min(ti);
      const double tmin = fmin(trow, tflip);
      if (tmin >= kBig) { status = phase1 ? DSP_STATUS_PRIMAL_INFEASIBLE : DSP_STATUS_DUAL_INFEASIBLE; break; }
      const bool flip = tflip <= trow;
      const bool tie = lane < m && ti <= tmin * (1.0 + 1e-9) + 1e-12;
      const double pv = wave_max(tie ? fabs(al) : -1.0);
      const int r = first_lane(__ballot(tie && fabs(al) == pv));
      // ---- move -------------------------------------------------------------------------------------------------------------
      if (lane < m) beta = fma(delta, tmin, beta);
      const double newval = fma(sgn, tmin, jval);
      if (flip) {
#pragma unroll
        for (int q = 0; q < CQ; ++q)
          if (q == jq && lane == jl) { val[q] = newval; upper[q] = !upper[q]; }
        continue;
      }
      // leaving variable: the basic of row r, to the bound it reached (decided in lane r, read by everybody)
      bool to_up = fabs(beta - bhi) < fabs(beta - blo);
      if (is_finite(bhi) && !is_finite(blo)) to_up = true;
      if (!is_finite(bhi)) to_up = false;
      const int jout = __builtin_amdgcn_readlane(bvar, r);
      const bool out_up = __builtin_amdgcn_readlane(to_up ? 1 : 0, r) != 0;
      const double out_val = readlane_f64(to_up ? bhi : blo, r);
      const double alpha_r = readlane_f64(al, r);
      if (lane == r) { bvar = jin; beta = newval; blo = jlo; bhi = jhi; cbas = jcost; }
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        const int j = lane + 64 * q;
        if (j == jout) { basic[q] = false; val[q] = out_val; upper[q] = out_up; }
        if (j == jin) basic[q] = true;
      }
      // ---- tableau update: row r /= alpha_r, rows i != r -= alpha_i * row r (all in registers) -----------------------------
      const double inv = 1.0 / alpha_r;
      double prow[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) prow[q] = 0.0;
#pragma unroll
      for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) prow[q] = (i == r) ? T[i][q] : prow[q];
      }
#pragma unroll
      for (int q = 0; q < CQ; ++q) prow[q] *= inv;
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        double t = T[i][0];
#pragma unroll
        for (int q = 1; q < CQ; ++q) t = (jq == q) ? T[i][q] : t;
        const double ai = readlane_f64(t, jl);                  // entry i of the entering column, before row i changes
#pragma unroll
        for (int q = 0; q < CQ; ++q) T[i][q] = (i == r) ? prow[q] : fma(-ai, prow[q], T[i][q]);
      }
      ++pivots;
    }

    // ---- refinement of an optimal vertex' basic values (as in the LDS-tableau kernel): residual of the original rows, correction by
    // B^-1 = the slack COLUMNS of the tableau, which here are the registers of lanes n .. N - 1 -----------------------------------------
    if (status == DSP_STATUS_OPTIMAL && refines < 2) {
      ++refines;
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = val[q]; }
      wave_lds_fence();
      if (lane < m) xval[bvar] = beta;
      wave_lds_fence();
      double xr[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; xr[q] = (j < N) ? xval[j] : 0.0; }
      double res = 0.0;
      for (int i0 = 0; i0 < m; i0 += 4) {
        double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u;
          if (i < m) {
#pragma unroll
            for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < n) part[u] = fma(a.A_dense[(size_t)i * n + j], xr[q], part[u]); }
          }
        }
        wave_sums<4>(part);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (lane == i0 + u) res = part[u] - xval[n + lane];
      }
      wave_lds_fence();
      if (lane < m) xval[n + lane] = res;                       // residual of row i where column n + i's lane finds it
      wave_lds_fence();
      double rj[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; rj[q] = (j >= n && j < N) ? xval[j] : 0.0; }
#pragma unroll
      for (int i0 = 0; i0 < MR; i0 += 4) {
        double part[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          double t = 0.0;
#pragma unroll
          for (int q = 0; q < CQ; ++q) t = fma(T[i0 + u][q], rj[q], t);
          part[u] = t;
        }
        wave_sums<4>(part);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (lane == i0 + u && lane < m) beta += part[u];
      }
      wave_lds_fence();
      bool out = false;
      if (lane < m) {
        const double tt = 1e-3 * a.tol_p * (1.0 + fmax(fabs(finite_or_zero(blo)), fabs(finite_or_zero(bhi))));
        out = beta < blo - tt || beta > bhi + tt;
      }
      if (__ballot(out) != 0ull) { tight = true; status = -1; goto pivot; }
    }

    // ---- assemble the vertex, certify it against the original rows, store -----------------------------------------------
#pragma unroll
    for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < N) xval[j] = val[q]; }
    wave_lds_fence();
    if (lane < m) xval[bvar] = beta;
    wave_lds_fence();
    double xs[CQ];
    double po = 0.0;
    bool okv = true, okb = true, okr = true;
    double viol = 0.0;                                        // largest bound violation of the vertex, relative to its variable's scale
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const int j = lane + 64 * q;
      xs[q] = (j < N) ? xval[j] : 0.0;
      if (j < N) {
        const double scl = 1.0 + fmax(fabs(xs[q]), fmax(fabs(finite_or_zero(lo[q])), fabs(finite_or_zero(hi[q]))));
        const double tol = 1e-9 * scl;
        if (xs[q] < lo[q] - tol || xs[q] > hi[q] + tol || !(xs[q] == xs[q])) { okv = false; okb = false; }
        viol = fmax(viol, fmax(lo[q] - xs[q], xs[q] - hi[q]) / scl);
      }
      if (j < n) po = fma(cost[q], xs[q], po);
    }
    po = wave_sum(po);
    {
      // row residuals with the ORIGINAL scaled matrix, sum_j a_ij x_j - s_i, against the scale of the whole vertex (see the
      // LDS-tableau kernel): coalesced row loads + wave reductions, four rows at a time
      double res = 0.0, mag = 0.0;
      for (int i0 = 0; i0 < m; i0 += 2) {
        double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = i0 + u;
          if (i < m) {
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
              const int j = lane + 64 * q;
              if (j < n) { const double t = a.A_dense[(size_t)i * n + j] * xs[q]; part[2 * u] += t; part[2 * u + 1] += fabs(t); }
            }
          }
        }
        wave_sums<4>(part);
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (lane == i0 + u) { res = part[2 * u] - xval[n + lane]; mag = part[2 * u + 1] + fabs(xval[n + lane]); }
      }
      const double gmag = wave_max(mag);
      if (lane < m && !(fabs(res) <= 1e-9 * (1.0 + gmag))) { okv = false; okr = false; }
    }
    const bool certified = __ballot(!okv) == 0ull;
    const unsigned long long bad_bounds = __ballot(!okb), bad_any = __ballot(!okv);
    int reason = status == -1 ? 1 : 0;
    if (status == DSP_STATUS_OPTIMAL && !certified) { status = -1; reason = 2; }
    // (a phase-1 optimum clearly outside its bounds on an intact tableau is reported infeasible: see the LDS-tableau kernel)
    const bool clearly_infeasible = status == DSP_STATUS_PRIMAL_INFEASIBLE && p1_priced_out && __ballot(!okr) == 0ull && wave_max(viol) > 1e-6;
    if (clearly_infeasible) {
#pragma unroll
      for (int q = 0; q < CQ; ++q) { const int j = lane + 64 * q; if (j < n) b.x[(size_t)s * n + j] = NAN; }
      if (lane < m) b.y[(size_t)s * m + lane] = NAN;
      if (lane == 0) {
        b.obj[s] = NAN;
        b.status[s] = DSP_STATUS_PRIMAL_INFEASIBLE;
        if (b.iters) b.iters[s] = pivots;
        if (b.jumps) b.jumps[s] = 0;
        if (b.flags) b.flags[s] = 0;
      }
      wave_lds_fence();
      continue;
    }
    if (status == DSP_STATUS_PRIMAL_INFEASIBLE || status == DSP_STATUS_DUAL_INFEASIBLE) { status = -1; reason = 3; }
    if (status == -1) {
      if (lane == 0) {
        if (a.debug_keep) {
          b.status[s] = 50 + reason;
          if (b.iters) b.iters[s] = pivots;
          if (b.jumps) b.jumps[s] = bad_bounds ? 1000 + first_lane(bad_bounds) : (bad_any ? 2000 + first_lane(bad_any) : 0);
          b.obj[s] = po;
          if (b.flags) b.flags[s] = 0;
        } else {
          b.status[s] = DSP_STATUS_UNSOLVED;
          atomicAdd(a.unsolved, 1);
        }
      }
      wave_lds_fence();
      continue;
    }
    // duals: y_i = reduced cost of slack i with the phase-2 costs
    double dsl[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) dsl[q] = cost[q];
    const double cb2 = (lane < m) ? cbas : 0.0;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const double cb = readlane_f64(cb2, i);
#pragma unroll
      for (int q = 0; q < CQ; ++q) dsl[q] = fma(-cb, T[i][q], dsl[q]);
    }
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const int j = lane + 64 * q;
      if (j < n) b.x[(size_t)s * n + j] = xs[q] * a.col_scale[j];
      else if (j < N) b.y[(size_t)s * m + (j - n)] = dsl[q] * a.row_scale[j - n];
    }
    if (lane == 0) {
      b.obj[s] = po;
      b.status[s] = status;
      if (b.iters) b.iters[s] = pivots;
      if (b.jumps) b.jumps[s] = 0;
      if (b.flags) b.flags[s] = 0;
    }
    wave_lds_fence();
  }
}

}  // namespace

size_t simplex_lds_bytes(int n, int m, int *row_stride) {
  const int N = n + m;
  int rs = N | 1;                                    // odd number of doubles: conflict-free column AND row sweeps
  if (row_stride) *row_stride = rs;
  const int mp = (m + 1) & ~1, Np = (N + 1) & ~1;
  return ((size_t)m * rs + mp + mp + Np + 8 + 8) * sizeof(double);
}

hipError_t launch_simplex(const SimplexArgs &a, int grid, size_t lds, hipStream_t st) {
  const int cq = (a.n + a.m + 63) / 64;
  if (cq > 2) return hipErrorInvalidValue;
  SimplexArgs args = a;
  void *params[] = {&args};
  // register-resident tableau where the rows fit (DSP_SX_LDS=1 forces the LDS-tableau kernel: development / comparison)
  static const int force_lds = getenv("DSP_SX_LDS") ? atoi(getenv("DSP_SX_LDS")) : 0;
  if (!force_lds && !a.warm && a.m <= 24 && cq <= 1 && a.b.B <= 2048) {       // (the warm start lives in the LDS-tableau kernel)
    const void *fr = reinterpret_cast<const void *>(&simplex_reg_kernel<1, 24>);
    return hipLaunchKernel(fr, dim3(a.b.B < grid ? a.b.B : grid), dim3(64), params, 0, st);
  }
  const void *fn = cq <= 1 ? reinterpret_cast<const void *>(&simplex_kernel<1>)
                           : reinterpret_cast<const void *>(&simplex_kernel<2>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  return hipLaunchKernel(fn, dim3(grid), dim3(64), params, lds, st);
}

}  // namespace dsp
