// dsp_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the batched dispatch-LP solver.
//
// Hot path: one price scenario per 64-lane wave, several waves (scenarios) per workgroup.
//   * lane l owns columns {l, l+64, ...} (CPL of them) and rows {l, l+64, ...} (RPL of them); every per-scenario
//     vector (x, anchor, c, bounds, y, row bounds, A x) lives in that lane's registers for the whole solve;
//   * the scaled constraint matrix, shared by all scenarios, is staged ONCE per workgroup into LDS in a
//     lane-major ELL layout for A (row products) and A^T (column products): conflict-free ds_read_b64 streams,
//     plus a per-wave LDS exchange buffer through which x-bar / y are gathered;
//   * vectors longer than the ELL width (e.g. the shared PEM-capacity column) are reduced cooperatively by the
//     wave with cross-lane shuffles ("LDS-staged partials + wavefront reductions");
//   * the iteration is the restarted, reflected Halpern PDHG (r2HPDHG): two SpMVs per iteration, NO reduction on
//     the per-iteration path; restart / KKT tests every `check_every` iterations use 7 wave reductions;
//   * scenarios are pulled from a device-side work queue (one atomicAdd per scenario) so waves retire
//     independently — iteration counts differ several-fold between price scenarios.
// No MFMA: the work is sparse BLAS-2 with ~3 nonzeros per row.  HBM is touched only to load a scenario's
// (c, bounds) and to store (x, y, obj); the limiter is LDS issue + FP64 VALU.
//
// The streaming SpMV step kernel at the bottom keeps X/Y in HBM and is the kernel whose HBM roofline
// SURVEY.md 8(d) defines.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>

#include "dsp_device.hpp"

namespace dsp {

// ---- wave-level helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {
  // Single-wave producer/consumer through LDS: the LDS unit executes one wave's DS ops in order, so only the
  // COMPILER must be kept from reordering the exchange-buffer stores and gathers.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// three / four sums at once (independent shuffle chains interleave)
__device__ __forceinline__ void wave_sum3(double &a, double &b, double &c) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    double ta = __shfl_xor(a, off, 64), tb = __shfl_xor(b, off, 64), tc = __shfl_xor(c, off, 64);
    a += ta; b += tb; c += tc;
  }
}
__device__ __forceinline__ void wave_sum4(double &a, double &b, double &c, double &d) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    double ta = __shfl_xor(a, off, 64), tb = __shfl_xor(b, off, 64), tc = __shfl_xor(c, off, 64),
           td = __shfl_xor(d, off, 64);
    a += ta; b += tb; c += tc; d += td;
  }
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }
__device__ __forceinline__ double finite_or_zero(double v) { return (fabs(v) < INFINITY) ? v : 0.0; }

// ---- lane-major ELL products --------------------------------------------------------------------------------
// out[q] = sum_e val[(q*W+e)*64+lane] * vec[idx[...]]   for the S slots this lane owns, + long vectors.
template <int S>
__device__ __forceinline__ void ell_product(double (&out)[S], const double *__restrict__ ell_val,
                                            const uint16_t *__restrict__ ell_idx, int W,
                                            const double *vec /* per-wave LDS exchange buffer */, int lane,
                                            const LongList &ll, const double *tail_val, const uint16_t *tail_idx) {
#pragma unroll
  for (int q = 0; q < S; ++q) {
    double acc = 0.0;
    const int base = q * W * 64 + lane;
    for (int e = 0; e < W; ++e) {
      const double a = ell_val[base + e * 64];
      const int j = ell_idx[base + e * 64];
      acc = fma(a, vec[j], acc);
    }
    out[q] = acc;
  }
  for (int l = 0; l < ll.count; ++l) {
    const int owner = ll.owner[l], start = ll.start[l], len = ll.len[l];
    double part = 0.0;
    for (int t = lane; t < len; t += 64) part = fma(tail_val[start + t], vec[tail_idx[start + t]], part);
    part = wave_sum(part);
#pragma unroll
    for (int q = 0; q < S; ++q)
      if (owner == lane + 64 * q) out[q] += part;
  }
}

// ---- the fused, LDS-resident PDLP solve ---------------------------------------------------------------------
template <int CPL, int RPL>
__global__ void __launch_bounds__(512) pdlp_solve_kernel(SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const DeviceProblem &P = a.P;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;

  // ---- carve LDS: shared matrix region, then one exchange buffer pair per wave ----------------------------
  double *ellc_val = reinterpret_cast<double *>(smem);                 // A^T (columns)  [CPL*Wc*64]
  double *ellr_val = ellc_val + P.ellc_entries;                        // A   (rows)     [RPL*Wr*64]
  double *tailc_val = ellr_val + P.ellr_entries;
  double *tailr_val = tailc_val + P.tailc_entries;
  double *wave_buf = tailr_val + P.tailr_entries;                      // [waves][n_pad + m_pad]
  uint16_t *ellc_idx = reinterpret_cast<uint16_t *>(wave_buf + (size_t)a.waves_per_block * (P.n_pad + P.m_pad));
  uint16_t *ellr_idx = ellc_idx + P.ellc_entries;
  uint16_t *tailc_idx = ellr_idx + P.ellr_entries;
  uint16_t *tailr_idx = tailc_idx + P.tailc_entries;

  for (int t = threadIdx.x; t < P.ellc_entries; t += blockDim.x) { ellc_val[t] = P.ellc_val[t]; ellc_idx[t] = P.ellc_idx[t]; }
  for (int t = threadIdx.x; t < P.ellr_entries; t += blockDim.x) { ellr_val[t] = P.ellr_val[t]; ellr_idx[t] = P.ellr_idx[t]; }
  for (int t = threadIdx.x; t < P.tailc_entries; t += blockDim.x) { tailc_val[t] = P.tailc_val[t]; tailc_idx[t] = P.tailc_idx[t]; }
  for (int t = threadIdx.x; t < P.tailr_entries; t += blockDim.x) { tailr_val[t] = P.tailr_val[t]; tailr_idx[t] = P.tailr_idx[t]; }
  __syncthreads();

  double *xb = wave_buf + (size_t)wave * (P.n_pad + P.m_pad);          // gathered by row products
  double *yb = xb + P.n_pad;                                           // gathered by column products

  const int n = P.n, m = P.m;
  const double eta = a.eta;
  const double eps = a.opt.eps_rel;

  for (;;) {
    // ---- pull the next scenario off the work queue ---------------------------------------------------------
    int s = 0;
    if (lane == 0) s = atomicAdd(a.queue, 1);
    s = __builtin_amdgcn_readfirstlane(s);
    if (s >= a.B) break;

    // ---- load + scale this scenario's vectors (coalesced: lane-consecutive addresses) -----------------------
    double x[CPL], x0[CPL], c[CPL], lb[CPL], ub[CPL];
    double y[RPL], y0[RPL], rlo[RPL], rhi[RPL], ax[RPL], ax0[RPL];
    double qn2 = 0.0, cn2 = 0.0, qs2 = 0.0, cs2 = 0.0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int j = lane + 64 * q;
      const bool ok = j < n;
      const double d = ok ? P.col_scale[j] : 1.0;
      const double cu = ok ? a.c[(size_t)s * a.c_stride + j] : 0.0;
      const double lu = ok ? (a.var_lb ? a.var_lb[(size_t)s * a.var_lb_stride + j] : -INFINITY) : 0.0;
      const double uu = ok ? (a.var_ub ? a.var_ub[(size_t)s * a.var_ub_stride + j] : INFINITY) : 0.0;
      c[q] = cu * d;
      lb[q] = lu / d;
      ub[q] = uu / d;
      cn2 += cu * cu;
      cs2 += c[q] * c[q];
      const double lf = finite_or_zero(lu), uf = finite_or_zero(uu);
      qn2 += lf * lf + uf * uf;
      double xs = (a.x0 && ok) ? a.x0[(size_t)s * n + j] / d : 0.0;
      x[q] = clampd(xs, lb[q], ub[q]);
      x0[q] = x[q];
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int i = lane + 64 * q;
      const bool ok = i < m;
      const double d = ok ? P.row_scale[i] : 1.0;
      const double lo = (ok && a.row_lb) ? a.row_lb[(size_t)s * a.row_lb_stride + i] : -INFINITY;
      const double hi = (ok && a.row_ub) ? a.row_ub[(size_t)s * a.row_ub_stride + i] : INFINITY;
      rlo[q] = lo * d;
      rhi[q] = hi * d;
      const double big = fmax(fabs(finite_or_zero(lo)), fabs(finite_or_zero(hi)));
      qn2 += big * big;
      const double bigs = fmax(fabs(finite_or_zero(rlo[q])), fabs(finite_or_zero(rhi[q])));
      qs2 += bigs * bigs;
      double ys = (a.y0 && ok) ? a.y0[(size_t)s * m + i] / d : 0.0;
      // keep the warm start dual-feasible in sign
      if (!(fabs(rlo[q]) < INFINITY)) ys = fmin(ys, 0.0);
      if (!(fabs(rhi[q]) < INFINITY)) ys = fmax(ys, 0.0);
      y[q] = ys;
      y0[q] = ys;
    }
    wave_sum4(qn2, cn2, qs2, cs2);
    const double qn = sqrt(qn2), cn = sqrt(cn2);
    const double qs = sqrt(qs2), cs = sqrt(cs2);
    double w = (cs > 1e-10 && qs > 1e-10) ? cs / qs : 1.0;       // primal weight

    // A x for the starting point
#pragma unroll
    for (int q = 0; q < CPL; ++q) xb[lane + 64 * q] = x[q];
    wave_lds_fence();
    ell_product<RPL>(ax, ellr_val, ellr_idx, P.Wr, xb, lane, P.long_r, tailr_val, tailr_idx);
#pragma unroll
    for (int q = 0; q < RPL; ++q) ax0[q] = ax[q];

    int k = 0;                       // iterations since the last restart
    int it = 0;
    double r0 = INFINITY, rprev = INFINITY;
    int status = DSP_STATUS_ITERATION_LIMIT;
    double xp[CPL], yp[RPL], axb[RPL];
    double pobj = 0.0;

    for (it = 0; it < a.opt.max_iter; ++it) {
      const double tau = eta / w, sig = eta * w;
      // ---- column products: A^T y -------------------------------------------------------------------------
#pragma unroll
      for (int q = 0; q < RPL; ++q) yb[lane + 64 * q] = y[q];
      wave_lds_fence();
      double aty[CPL];
      ell_product<CPL>(aty, ellc_val, ellc_idx, P.Wc, yb, lane, P.long_c, tailc_val, tailc_idx);
      // ---- primal step, reflection point, row products: A (2 x+ - x) ---------------------------------------
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        xp[q] = clampd(x[q] - tau * (c[q] - aty[q]), lb[q], ub[q]);
        xb[lane + 64 * q] = 2.0 * xp[q] - x[q];
      }
      wave_lds_fence();
      ell_product<RPL>(axb, ellr_val, ellr_idx, P.Wr, xb, lane, P.long_r, tailr_val, tailr_idx);
      // ---- dual step ---------------------------------------------------------------------------------------
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const double wv = y[q] - sig * axb[q];
        yp[q] = wv + clampd(-wv, sig * rlo[q], sig * rhi[q]);
      }
      ++k;
      const bool check = ((it + 1) % a.opt.check_every) == 0;
      const bool need_r0 = (k == 1);
      bool restarted = false;
      if (check || need_r0) {
        // fixed-point residual in the PDHG metric: w|dx|^2 - 2 eta dy.A dx + |dy|^2 / w
        double sxx = 0.0, syy = 0.0, sxy = 0.0;
#pragma unroll
        for (int q = 0; q < CPL; ++q) { const double dx = xp[q] - x[q]; sxx = fma(dx, dx, sxx); }
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          const double dy = yp[q] - y[q];
          syy = fma(dy, dy, syy);
          sxy = fma(dy, 0.5 * (axb[q] - ax[q]), sxy);
        }
        wave_sum3(sxx, syy, sxy);
        const double r = sqrt(fmax(w * sxx - 2.0 * eta * sxy + syy / w, 0.0));
        if (!(r == r)) { status = DSP_STATUS_NUMERICAL; break; }
        if (need_r0) { r0 = r; rprev = r; }
        if (check) {
          // ---- KKT test at (x+, y+) in the ORIGINAL (unscaled) space ----------------------------------------
#pragma unroll
          for (int q = 0; q < RPL; ++q) yb[lane + 64 * q] = yp[q];
          wave_lds_fence();
          double atyp[CPL];
          ell_product<CPL>(atyp, ellc_val, ellc_idx, P.Wc, yb, lane, P.long_c, tailc_val, tailc_idx);
          double pres2 = 0.0, dres2 = 0.0, po = 0.0, dobj = 0.0;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const int j = lane + 64 * q;
            const double d = (j < n) ? P.col_scale[j] : 1.0;
            const double rc = c[q] - atyp[q];
            const double lp = (fabs(lb[q]) < INFINITY) ? fmax(rc, 0.0) : 0.0;
            const double lm = (fabs(ub[q]) < INFINITY) ? fmax(-rc, 0.0) : 0.0;
            const double dr_ = (rc - lp + lm) / d;
            dres2 = fma(dr_, dr_, dres2);
            po = fma(c[q], xp[q], po);
            dobj += lp * finite_or_zero(lb[q]) - lm * finite_or_zero(ub[q]);
          }
#pragma unroll
          for (int q = 0; q < RPL; ++q) {
            const int i = lane + 64 * q;
            const double d = (i < m) ? P.row_scale[i] : 1.0;
            const double axp = 0.5 * (axb[q] + ax[q]);
            const double viol = (fmax(rlo[q] - axp, 0.0) + fmax(axp - rhi[q], 0.0)) / d;
            pres2 = fma(viol, viol, pres2);
            dobj += fmax(yp[q], 0.0) * finite_or_zero(rlo[q]) - fmax(-yp[q], 0.0) * finite_or_zero(rhi[q]);
          }
          wave_sum4(pres2, dres2, po, dobj);
          pobj = po;
          const double rp = sqrt(pres2) / (1.0 + qn);
          const double rd = sqrt(dres2) / (1.0 + cn);
          const double rg = fabs(po - dobj) / (1.0 + fabs(po) + fabs(dobj));
          if (rp <= eps && rd <= eps && rg <= eps) { status = DSP_STATUS_OPTIMAL; ++it; break; }
          // ---- restart test -----------------------------------------------------------------------------
          const bool do_restart = (r <= a.opt.restart_sufficient * r0) ||
                                  (r <= a.opt.restart_necessary * r0 && r > rprev) ||
                                  ((double)k >= a.opt.restart_artificial * (double)(it + 1));
          rprev = r;
          if (do_restart) {
            double ddx = 0.0, ddy = 0.0, dummy = 0.0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) { const double t = xp[q] - x0[q]; ddx = fma(t, t, ddx); }
#pragma unroll
            for (int q = 0; q < RPL; ++q) { const double t = yp[q] - y0[q]; ddy = fma(t, t, ddy); }
            wave_sum3(ddx, ddy, dummy);
            ddx = sqrt(ddx); ddy = sqrt(ddy);
            if (ddx > 1e-14 && ddy > 1e-14) {
              const double e = log(w) + log(ddx) - log(ddy);
              const double dl = clampd(-a.opt.pid_kp * e, -a.opt.max_dlog_weight, a.opt.max_dlog_weight);
              w *= exp(dl);
            }
#pragma unroll
            for (int q = 0; q < CPL; ++q) { x[q] = xp[q]; x0[q] = xp[q]; }
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
              const double axp = 0.5 * (axb[q] + ax[q]);
              y[q] = yp[q]; y0[q] = yp[q]; ax[q] = axp; ax0[q] = axp;
            }
            k = 0; r0 = INFINITY; rprev = INFINITY;
            restarted = true;
          }
        }
      }
      if (!restarted) {
        // ---- reflected Halpern step toward the anchor ------------------------------------------------------
        const double lam = (double)(k + 1) / (double)(k + 2), oml = 1.0 - lam;
#pragma unroll
        for (int q = 0; q < CPL; ++q) x[q] = lam * (2.0 * xp[q] - x[q]) + oml * x0[q];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          y[q] = lam * (2.0 * yp[q] - y[q]) + oml * y0[q];
          ax[q] = lam * axb[q] + oml * ax0[q];
        }
      }
    }

    // ---- store the scenario's result (unscaled) ----------------------------------------------------------
    if (status != DSP_STATUS_OPTIMAL) {
      double po = 0.0;
#pragma unroll
      for (int q = 0; q < CPL; ++q) po = fma(c[q], xp[q], po);
      pobj = wave_sum(po);
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int j = lane + 64 * q;
      if (j < n) a.x[(size_t)s * n + j] = xp[q] * P.col_scale[j];
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int i = lane + 64 * q;
      if (i < m) a.y[(size_t)s * m + i] = yp[q] * P.row_scale[i];
    }
    if (lane == 0) {
      a.obj[s] = pobj;
      a.status[s] = status;
      if (a.iters) a.iters[s] = it;
    }
  }
}

// ---- streaming SpMV step: AX = A X, ATY = A^T Y with vectors in HBM --------------------------------------
// One scenario per wave; the (unscaled) ELL matrix is staged into LDS once per workgroup; each wave streams its
// x (coalesced) into its LDS exchange buffer, forms the row products from LDS and writes them coalesced.
// Algorithmic HBM bytes per scenario: 2*8*(n+m)  (read x,y; write Ax, A^T y).
template <int CPL, int RPL>
__global__ void __launch_bounds__(512) spmv_step_kernel(SpmvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const DeviceProblem &P = a.P;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double *ellc_val = reinterpret_cast<double *>(smem);
  double *ellr_val = ellc_val + P.ellc_entries;
  double *tailc_val = ellr_val + P.ellr_entries;
  double *tailr_val = tailc_val + P.tailc_entries;
  double *wave_buf = tailr_val + P.tailr_entries;
  uint16_t *ellc_idx = reinterpret_cast<uint16_t *>(wave_buf + (size_t)a.waves_per_block * (P.n_pad + P.m_pad));
  uint16_t *ellr_idx = ellc_idx + P.ellc_entries;
  uint16_t *tailc_idx = ellr_idx + P.ellr_entries;
  uint16_t *tailr_idx = tailc_idx + P.tailc_entries;
  for (int t = threadIdx.x; t < P.ellc_entries; t += blockDim.x) { ellc_val[t] = P.ellc_val_unscaled[t]; ellc_idx[t] = P.ellc_idx[t]; }
  for (int t = threadIdx.x; t < P.ellr_entries; t += blockDim.x) { ellr_val[t] = P.ellr_val_unscaled[t]; ellr_idx[t] = P.ellr_idx[t]; }
  for (int t = threadIdx.x; t < P.tailc_entries; t += blockDim.x) { tailc_val[t] = P.tailc_val_unscaled[t]; tailc_idx[t] = P.tailc_idx[t]; }
  for (int t = threadIdx.x; t < P.tailr_entries; t += blockDim.x) { tailr_val[t] = P.tailr_val_unscaled[t]; tailr_idx[t] = P.tailr_idx[t]; }
  __syncthreads();
  double *xb = wave_buf + (size_t)wave * (P.n_pad + P.m_pad);
  double *yb = xb + P.n_pad;
  const int n = P.n, m = P.m;
  const int waves_total = gridDim.x * a.waves_per_block;
  for (int s = blockIdx.x * a.waves_per_block + wave; s < a.B; s += waves_total) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) { const int j = lane + 64 * q; xb[j] = (j < n) ? a.X[(size_t)s * n + j] : 0.0; }
#pragma unroll
    for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; yb[i] = (i < m) ? a.Y[(size_t)s * m + i] : 0.0; }
    wave_lds_fence();
    double axv[RPL], atyv[CPL];
    ell_product<RPL>(axv, ellr_val, ellr_idx, P.Wr, xb, lane, P.long_r, tailr_val, tailr_idx);
    ell_product<CPL>(atyv, ellc_val, ellc_idx, P.Wc, yb, lane, P.long_c, tailc_val, tailc_idx);
#pragma unroll
    for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; if (i < m) a.AX[(size_t)s * m + i] = axv[q]; }
#pragma unroll
    for (int q = 0; q < CPL; ++q) { const int j = lane + 64 * q; if (j < n) a.ATY[(size_t)s * n + j] = atyv[q]; }
    wave_lds_fence();
  }
}

// ---- launch tables ------------------------------------------------------------------------------------------
template <int CPL, int RPL>
static hipError_t launch_solve_t(const SolveArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pdlp_solve_kernel<CPL, RPL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((pdlp_solve_kernel<CPL, RPL>), grid, block, lds, st, a);
  return hipGetLastError();
}
template <int CPL, int RPL>
static hipError_t launch_spmv_t(const SpmvArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_step_kernel<CPL, RPL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((spmv_step_kernel<CPL, RPL>), grid, block, lds, st, a);
  return hipGetLastError();
}

#ifdef DSP_SINGLE_VARIANT   /* development: instantiate only <4,2> for -save-temps inspection */
#define DSP_FOR_CPL(F) if (cpl == 4 && rpl == 2) return F<4, 2>(a, grid, block, lds, st); return hipErrorInvalidValue;
#else
#define DSP_FOR_RPL(F, C)                                                                                   \
  switch (rpl) {                                                                                            \
    case 1: return F<C, 1>(a, grid, block, lds, st);                                                        \
    case 2: return F<C, 2>(a, grid, block, lds, st);                                                        \
    case 3: return F<C, 3>(a, grid, block, lds, st);                                                        \
    case 4: return F<C, 4>(a, grid, block, lds, st);                                                        \
    case 6: return F<C, 6>(a, grid, block, lds, st);                                                        \
    default: return hipErrorInvalidValue;                                                                   \
  }
#define DSP_FOR_CPL(F)                                                                                      \
  switch (cpl) {                                                                                            \
    case 1: DSP_FOR_RPL(F, 1)                                                                               \
    case 2: DSP_FOR_RPL(F, 2)                                                                               \
    case 3: DSP_FOR_RPL(F, 3)                                                                               \
    case 4: DSP_FOR_RPL(F, 4)                                                                               \
    case 5: DSP_FOR_RPL(F, 5)                                                                               \
    case 7: DSP_FOR_RPL(F, 7)                                                                               \
    case 10: DSP_FOR_RPL(F, 10)                                                                             \
    default: return hipErrorInvalidValue;                                                                   \
  }
#endif

hipError_t launch_solve(int cpl, int rpl, const SolveArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  DSP_FOR_CPL(launch_solve_t)
}
hipError_t launch_spmv(int cpl, int rpl, const SpmvArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  DSP_FOR_CPL(launch_spmv_t)
}

}  // namespace dsp
