// dsp_qp.hip — the float32 form of the fused PDLP solve (dsp_options::precision = 1), gfx950 only.
//
// BASELINE config 5 asks for an fp64-vs-fp32 tolerance sweep of the stochastic bidder's problems (LP, and QP with the
// quadratic ramp cost as soft rows: dsp_batch::row_compliance).  The float64 side of the sweep is the production kernel
// (dsp_kernels.hip, QP instantiation).  This kernel is the float32 side: the same restarted, reflected Halpern PDHG
// (same restart tests, same primal-weight controller and rounding guard - with the float32 unit roundoff -, same
// termination tests), one scenario per wave, with
//   * iterates, anchors, cost, bounds, the scaled matrix and both SpMVs in float32 (half the registers, half the LDS
//     traffic of the generic float64 kernel: 8-byte {value, offset} matrix entries, one ds_read_b64 each, 4-byte
//     exchange slots);
//   * every reduction, the fixed-point residual and all KKT quantities accumulated in float64 from the float32 data;
//   * the objective of the RETURNED point evaluated in float64 from the caller's float64 cost vector (what the point is
//     worth, not what float32 arithmetic thinks it is worth).
// No ray jumps, no stall rescue, no register-resident matrix: this is a measuring instrument for the sweep, not the
// parity path.  Float32 cannot reach the 1e-6 objective contract on these LPs (1e4 $/MWh penalty columns against 1e-2
// costs, 1e5 kWh states): bench.py --workload qp_sweep reports where it stops.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "dsp_device.hpp"
#include "dsp_wave.hpp"

namespace dsp {

struct __attribute__((aligned(8))) Entry32 {
  float v;
  uint32_t off;     // BYTE offset of the multiplied element in the wave's float exchange buffer
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ bool finitef(float v) { return fabsf(v) < INFINITY; }
__device__ __forceinline__ double fin0f(float v) { return finitef(v) ? (double)v : 0.0; }

// out[q] = sum_e ell[(e*S+q)*64+lane].v * vec[off]  (lane-major ELL in LDS, conflict-free ds_read_b64 per entry)
template <int S>
__device__ __forceinline__ void ell_product_f32(float (&out)[S], const Entry32 *ell, int W, const char *vec, int lane) {
#pragma unroll
  for (int q = 0; q < S; ++q) out[q] = 0.0f;
  const Entry32 *p = ell + lane;
#pragma unroll 1
  for (int e = 0; e < W; ++e, p += S * 64) {
    Entry32 en[S];
#pragma unroll
    for (int q = 0; q < S; ++q) en[q] = p[q * 64];
    float xv[S];
#pragma unroll
    for (int q = 0; q < S; ++q) xv[q] = *reinterpret_cast<const float *>(vec + en[q].off);
#pragma unroll
    for (int q = 0; q < S; ++q) out[q] = fmaf(en[q].v, xv[q], out[q]);
  }
}

constexpr int kF32Waves = 4;      // waves (= scenarios in flight) per workgroup

template <int CPL, int RPL>
__global__ void __launch_bounds__(64 * kF32Waves) pdlp_solve_f32_kernel(SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const DeviceProblem &P = a.P;
  const dsp_batch &b = a.b;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = P.n, m = P.m;
  Entry32 *ellc = reinterpret_cast<Entry32 *>(smem);                   // A^T  [Wc*CPL*64]
  Entry32 *ellr = ellc + P.ellc_entries;                               // A    [Wr*RPL*64]
  char *wave_buf = reinterpret_cast<char *>(ellr + P.ellr_entries);
  for (int t = threadIdx.x; t < P.ellc_entries; t += blockDim.x) { ellc[t].v = (float)P.ellc[t].v; ellc[t].off = P.ellc[t].off >> 1; }
  for (int t = threadIdx.x; t < P.ellr_entries; t += blockDim.x) { ellr[t].v = (float)P.ellr[t].v; ellr[t].off = P.ellr[t].off >> 1; }
  __syncthreads();
  char *xb = wave_buf + (size_t)wave * (P.n_pad + P.m_pad) * 4;        // gathered by the row products
  char *yb = xb + (size_t)P.n_pad * 4;                                 // gathered by the column products
  float *xbl = reinterpret_cast<float *>(xb) + lane, *ybl = reinterpret_cast<float *>(yb) + lane;

  const double eta = a.eta, eps = a.opt.eps_rel, eps_obj = a.opt.eps_obj;
  const int check_every = a.opt.check_every > 0 ? a.opt.check_every : 16;
  const double beta_s2 = a.opt.restart_sufficient * a.opt.restart_sufficient;
  const double beta_n2 = a.opt.restart_necessary * a.opt.restart_necessary;
  constexpr double kU32 = 5.96e-8;                                      // float32 unit roundoff

  for (;;) {
    __builtin_amdgcn_wave_barrier();
    const int ticket = (int)(unsigned)atomicAdd(a.queue, lane == 0 ? 1 : 0);     // see dsp_kernels.hip: branch-free pull
    const int s = __builtin_amdgcn_readlane(ticket, 0);
    if ((unsigned)s >= (unsigned)b.B) break;

    float x[CPL], x0[CPL], c[CPL], lb[CPL], ub[CPL], xp[CPL];
    float y[RPL], y0[RPL], rlo[RPL], rhi[RPL], yp[RPL], kap[RPL], srow[RPL];
    double nrm[5] = {0, 0, 0, 0, 0};       // |q|^2 unscaled, |c|^2 unscaled, |q|^2 scaled, |c|^2 scaled, column bounds^2 scaled
    double cmax = 0.0, qmax = 0.0, bad = 0.0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int j = lane + 64 * q;
      const bool ok = j < n;
      const double d = ok ? P.col_scale[j] : 1.0;
      const double cu = ok ? b.c[(size_t)s * b.c_stride + j] : 0.0;
      const double lu = ok ? (b.var_lb ? b.var_lb[(size_t)s * b.var_lb_stride + j] : -INFINITY) : 0.0;
      const double uu = ok ? (b.var_ub ? b.var_ub[(size_t)s * b.var_ub_stride + j] : INFINITY) : 0.0;
      const double cs_ = cu * d, ls = lu / d, us = uu / d;
      c[q] = (float)cs_; lb[q] = (float)ls; ub[q] = (float)us;
      if (!(lu <= uu) || !(cu == cu)) bad = 1.0;
      nrm[1] += cu * cu;
      nrm[3] += cs_ * cs_;
      cmax = fmax(cmax, fabs(cs_));
      nrm[4] += finite_or_zero(ls) * finite_or_zero(ls) + finite_or_zero(us) * finite_or_zero(us);
      nrm[0] += finite_or_zero(lu) * finite_or_zero(lu) + finite_or_zero(uu) * finite_or_zero(uu);
      x[q] = clampf(0.0f, lb[q], ub[q]);
      x0[q] = x[q]; xp[q] = x[q];
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int i = lane + 64 * q;
      const bool ok = i < m;
      const double d = ok ? P.row_scale[i] : 1.0;
      const double lo = (ok && b.row_lb) ? b.row_lb[(size_t)s * b.row_lb_stride + i] : -INFINITY;
      const double hi = (ok && b.row_ub) ? b.row_ub[(size_t)s * b.row_ub_stride + i] : INFINITY;
      const double kp_ = (ok && b.row_compliance) ? b.row_compliance[(size_t)s * b.row_compliance_stride + i] : 0.0;
      if (!(lo <= hi) || !(kp_ >= 0.0) || (kp_ > 0.0 && !(lo == hi && is_finite(lo)))) bad = 1.0;
      rlo[q] = (float)(lo * d); rhi[q] = (float)(hi * d);
      kap[q] = (float)(kp_ * d * d);
      const double big = fmax(fabs(finite_or_zero(lo)), fabs(finite_or_zero(hi)));
      nrm[0] += big * big;
      const double bigs = fmax(fabs(finite_or_zero(lo * d)), fabs(finite_or_zero(hi * d)));
      nrm[2] += bigs * bigs;
      qmax = fmax(qmax, bigs);
      y[q] = 0.0f; y0[q] = 0.0f; yp[q] = 0.0f;
    }
    wave_sums<5>(nrm);
    if (wave_max(bad) > 0.0) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) { const int j = lane + 64 * q; if (j < n) b.x[(size_t)s * n + j] = NAN; }
#pragma unroll
      for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; if (i < m) b.y[(size_t)s * m + i] = NAN; }
      if (lane == 0) {
        b.obj[s] = NAN;
        b.status[s] = (nrm[0] == nrm[0] && nrm[1] == nrm[1]) ? DSP_STATUS_PRIMAL_INFEASIBLE : DSP_STATUS_NUMERICAL;
        if (b.iters) b.iters[s] = 0;
        if (b.jumps) b.jumps[s] = 0;
        if (b.flags) b.flags[s] = 0;
      }
      continue;
    }
    const double qn = sqrt(nrm[0]), cn = sqrt(nrm[1]), qs = sqrt(nrm[2]), cs = sqrt(nrm[3]);
    const double c0 = b.obj_offset ? b.obj_offset[(size_t)s * b.obj_offset_stride] : 0.0;
    cmax = wave_max(cmax);
    qmax = wave_max(qmax);
    double w = (cs > 1e-10 && qs > 1e-10) ? cs / qs : 1.0;
    const double qall = sqrt(nrm[2] + nrm[4]);
    // rounding guard of the primal weight as in the float64 kernel, with the float32 roundoff; below eps = 1e-5 (which
    // float32 cannot deliver anyway) the two bounds would cross and pin the weight to nonsense, so the guard stops there
    const double eps_g = fmax(eps, 1e-5);
    const double w_lo = a.opt.weight_guard > 0.0 ? a.opt.weight_guard * eta * kU32 * cmax / (eps_g * (1.0 + qall)) : 0.0;
    const double w_hi = (a.opt.weight_guard > 0.0 && qmax > 0.0) ? eps_g * (1.0 + cs) / (a.opt.weight_guard * eta * kU32 * qmax) : INFINITY;
    w = fmin(fmax(w, w_lo), fmax(w_hi, w_lo));

    int k = 0, it = 0, ncheck = 0;
    double r0 = INFINITY, rprev = INFINITY, pobj = 0.0;
    int status = DSP_STATUS_ITERATION_LIMIT;
    float tau, sig;
    float ylo[RPL], yhi[RPL];
    auto set_steps = [&]() __attribute__((always_inline)) {
      tau = (float)(eta / w);
      sig = (float)(eta * w);
#pragma unroll
      for (int q = 0; q < RPL; ++q) { ylo[q] = -(sig * rhi[q]); yhi[q] = -(sig * rlo[q]); srow[q] = 1.0f / fmaf(sig, kap[q], 1.0f); }
    };
    set_steps();
    // T(x, y) -> (xp, yp)
    auto pdhg = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < RPL; ++q) ybl[64 * q] = y[q];
      wave_lds_fence();
      float aty[CPL], ax[RPL];
      ell_product_f32<CPL>(aty, ellc, P.Wc, yb, lane);
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        xp[q] = clampf(fmaf(tau, aty[q] - c[q], x[q]), lb[q], ub[q]);
        xbl[64 * q] = 2.0f * xp[q] - x[q];
      }
      wave_lds_fence();
      ell_product_f32<RPL>(ax, ellr, P.Wr, xb, lane);
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const float gy = fmaf(-sig, ax[q], y[q]);
        yp[q] = (gy - clampf(gy, ylo[q], yhi[q])) * srow[q];
      }
    };
    auto halpern = [&]() __attribute__((always_inline)) {
      const float oml = 1.0f / (float)(k + 2);
#pragma unroll
      for (int q = 0; q < CPL; ++q) { const float t = 2.0f * xp[q] - x[q]; x[q] = fmaf(oml, x0[q] - t, t); }
#pragma unroll
      for (int q = 0; q < RPL; ++q) { const float t = 2.0f * yp[q] - y[q]; y[q] = fmaf(oml, y0[q] - t, t); }
    };
    for (;;) {
      const int plain = min(check_every - 1, a.opt.max_iter - it);
      for (int u = 0; u < plain; ++u) { pdhg(); ++k; halpern(); }
      it += plain;
      if (it >= a.opt.max_iter) break;
      pdhg();
      ++k;
      bool moved = false;
      // ---- fixed-point residual in the PDHG metric (float64 accumulation) ----
      double px = 0.0, py = 0.0;
#pragma unroll
      for (int q = 0; q < CPL; ++q) { const float dx = xp[q] - x[q]; px = fma((double)dx, (double)dx, px); xbl[64 * q] = dx; }
      wave_lds_fence();
      float adx[RPL];
      ell_product_f32<RPL>(adx, ellr, P.Wr, xb, lane);
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const double dy = (double)(yp[q] - y[q]);
        py = fma(dy, fma(-2.0 * (double)sig, (double)adx[q], dy), py);       // |dy|^2 - 2 sig dy.A dx  (times 1/w below)
      }
      const double r = fmax(wave_sum(fma(w, px, py / w)), 0.0);
      if (!(r == r)) { status = DSP_STATUS_NUMERICAL; break; }
      ++ncheck;
      if ((ncheck & 1) == 0 || it + 1 >= a.opt.max_iter) {
        // ---- KKT test at (x+, y+) in the original space ----
#pragma unroll
        for (int q = 0; q < RPL; ++q) ybl[64 * q] = yp[q];
#pragma unroll
        for (int q = 0; q < CPL; ++q) xbl[64 * q] = xp[q];
        wave_lds_fence();
        float atyp[CPL], axp[RPL];
        ell_product_f32<CPL>(atyp, ellc, P.Wc, yb, lane);
        ell_product_f32<RPL>(axp, ellr, P.Wr, xb, lane);
        double red[7] = {0, 0, 0, 0, 0, 0, 0};      // pres^2, dres^2, pobj, dobj, sum|y| viol, sum|c x|, sum|dres||x|
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          const int j = lane + 64 * q;
          const double rc = (double)c[q] - (double)atyp[q];
          const double lp = finitef(lb[q]) ? fmax(rc, 0.0) : 0.0;
          const double lm = finitef(ub[q]) ? fmax(-rc, 0.0) : 0.0;
          const double dr_ = (rc - lp + lm) / ((j < n) ? P.col_scale[j] : 1.0);
          red[1] = fma(dr_, dr_, red[1]);
          red[6] = fma(fabs(rc - lp + lm), fabs((double)xp[q]), red[6]);
          const double cx = (double)c[q] * (double)xp[q];
          red[2] += cx;
          red[5] += fabs(cx);
          red[3] += lp * fin0f(lb[q]) - lm * fin0f(ub[q]);
        }
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
          const int i = lane + 64 * q;
          const double ax_ = (double)axp[q], yq = (double)yp[q];
          double viol_s = fmax((double)rlo[q] - ax_, 0.0) + fmax(ax_ - (double)rhi[q], 0.0);
          if (kap[q] > 0.0f) {
            const double dev = ax_ - (double)rlo[q];
            red[2] = fma(0.5 * dev, dev / (double)kap[q], red[2]);
            red[3] = fma(-0.5 * (double)kap[q] * yq, yq, red[3]);
            viol_s = 0.0;
          }
          const double viol = viol_s / ((i < m) ? P.row_scale[i] : 1.0);
          red[0] = fma(viol, viol, red[0]);
          red[4] = fma(fabs(yq), viol_s, red[4]);
          red[3] += fmax(yq, 0.0) * fin0f(rlo[q]) - fmax(-yq, 0.0) * fin0f(rhi[q]);
        }
        wave_sums<7>(red);
        const double po = red[2], dobj = red[3];
        pobj = po;
        if (!(po == po)) { status = DSP_STATUS_NUMERICAL; break; }
        const double rp = sqrt(red[0]) / (1.0 + qn), rd = sqrt(red[1]) / (1.0 + cn);
        const double gap = fabs(po - dobj), rg = gap / (1.0 + fabs(po) + fabs(dobj));
        bool done;                                               // same tests as the float64 kernel
        if (eps_obj > 0.0) {
          const double lim = fmax(eps_obj * (1.0 + fabs(po + c0)), 1e-12 * red[5]);
          done = rp <= eps && rd <= eps && gap + red[4] + red[6] <= lim;
        } else {
          done = rp <= eps && rd <= eps && rg <= eps;
        }
        if (done) { status = DSP_STATUS_OPTIMAL; ++it; break; }
      }
      // ---- restart test ----
      const bool first = !(r0 < INFINITY);
      const bool decayed = (r <= beta_s2 * r0) || (r <= beta_n2 * r0 && r > rprev);
      const bool artificial = (double)k >= a.opt.restart_artificial * (double)(it + 1);
      if (first) r0 = r;
      rprev = r;
      if (!first && (decayed || artificial)) {
        double dd[2] = {0.0, 0.0};
#pragma unroll
        for (int q = 0; q < CPL; ++q) { const double t = (double)(xp[q] - x0[q]); dd[0] = fma(t, t, dd[0]); }
#pragma unroll
        for (int q = 0; q < RPL; ++q) { const double t = (double)(yp[q] - y0[q]); dd[1] = fma(t, t, dd[1]); }
        wave_sums<2>(dd);
        if (dd[0] > 1e-28 && dd[1] > 1e-28) {
          const float e = __logf((float)w) + 0.5f * (__logf((float)dd[0]) - __logf((float)dd[1]));
          const float dl = fminf(fmaxf(-(float)a.opt.pid_kp * e, -(float)a.opt.max_dlog_weight), (float)a.opt.max_dlog_weight);
          w *= (double)__expf(dl);
        }
        w = fmin(fmax(w, w_lo), fmax(w_hi, w_lo));
        set_steps();
#pragma unroll
        for (int q = 0; q < CPL; ++q) { x[q] = xp[q]; x0[q] = xp[q]; }
#pragma unroll
        for (int q = 0; q < RPL; ++q) { y[q] = yp[q]; y0[q] = yp[q]; }
        k = 0; r0 = INFINITY; rprev = INFINITY;
        moved = true;
      }
      ++it;
      if (!moved) halpern();
    }

    // ---- store the scenario's result (unscaled); the objective of the returned point in float64 from the caller's c ----
    double po = 0.0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int j = lane + 64 * q;
      if (j < n) {
        const double xu = (double)xp[q] * P.col_scale[j];
        b.x[(size_t)s * n + j] = xu;
        po = fma(b.c[(size_t)s * b.c_stride + j], xu, po);
      }
      xbl[64 * q] = xp[q];
    }
    wave_lds_fence();
    float axq[RPL];
    ell_product_f32<RPL>(axq, ellr, P.Wr, xb, lane);
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int i = lane + 64 * q;
      if (i < m) b.y[(size_t)s * m + i] = (double)yp[q] * P.row_scale[i];
      if (kap[q] > 0.0f) { const double dev = (double)axq[q] - (double)rlo[q]; po = fma(0.5 * dev, dev / (double)kap[q], po); }
    }
    wave_lds_fence();
    pobj = wave_sum(po);
    if (lane == 0) {
      b.obj[s] = pobj;
      b.status[s] = status;
      if (b.iters) b.iters[s] = it;
      if (b.jumps) b.jumps[s] = 0;
      if (b.flags) b.flags[s] = 0;
      if (b.primal_weight) b.primal_weight[s] = w;
    }
  }
}

template <int CPL, int RPL>
static hipError_t launch_f32_t(const SolveArgs &a, int num_cus, size_t lds_limit, hipStream_t st, int *grid_out, int *threads_out,
                               size_t *lds_out) {
  const size_t lds = ((size_t)a.P.ellc_entries + a.P.ellr_entries) * sizeof(Entry32) + (size_t)kF32Waves * (a.P.n_pad + a.P.m_pad) * 4;
  if (lds > lds_limit) return hipErrorInvalidValue;
  const void *fn = reinterpret_cast<const void *>(&pdlp_solve_f32_kernel<CPL, RPL>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  int nb = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * kF32Waves, lds);
  if (e != hipSuccess) return e;
  nb = nb < 1 ? 1 : nb;
  int grid = (a.b.B + kF32Waves - 1) / kF32Waves;
  if (grid > num_cus * nb) grid = num_cus * nb;
  SolveArgs args = a;
  void *params[] = {&args};
  *grid_out = grid; *threads_out = 64 * kF32Waves; *lds_out = lds;
  return hipLaunchKernel(fn, dim3(grid), dim3(64 * kF32Waves), params, lds, st);
}

#define DSP_F32_RPL(C)                                                                                       \
  switch (rpl) {                                                                                             \
    case 1: return launch_f32_t<C, 1>(a, num_cus, lds_limit, st, grid, threads, lds);                        \
    case 2: return launch_f32_t<C, 2>(a, num_cus, lds_limit, st, grid, threads, lds);                        \
    case 3: return launch_f32_t<C, 3>(a, num_cus, lds_limit, st, grid, threads, lds);                        \
    case 4: return launch_f32_t<C, 4>(a, num_cus, lds_limit, st, grid, threads, lds);                        \
    case 6: return launch_f32_t<C, 6>(a, num_cus, lds_limit, st, grid, threads, lds);                        \
    default: return hipErrorInvalidValue;                                                                    \
  }

hipError_t launch_solve_f32(int cpl, int rpl, const SolveArgs &a, int num_cus, size_t lds_limit, hipStream_t st, int *grid,
                            int *threads, size_t *lds) {
  switch (cpl) {
    case 1: DSP_F32_RPL(1)
    case 2: DSP_F32_RPL(2)
    case 3: DSP_F32_RPL(3)
    case 4: DSP_F32_RPL(4)
    case 5: DSP_F32_RPL(5)
    case 7: DSP_F32_RPL(7)
    case 10: DSP_F32_RPL(10)
    default: return hipErrorInvalidValue;
  }
}

}  // namespace dsp
