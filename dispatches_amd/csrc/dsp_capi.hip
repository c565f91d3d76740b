// dsp_capi.hip — the C ABI of include/dsp_hip.h: handle management, host-side preparation, launch geometry.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "dsp_device.hpp"
#include "dsp_prepare.hpp"
#include "dsp_rtc.hpp"
#include "dsp_stream.hpp"

using namespace dsp;

static thread_local int g_last_hip_error = 0;

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) {                            \
      g_last_hip_error = (int)_e;                      \
      return DSP_ERR_HIP;                              \
    }                                                  \
  } while (0)

struct Geometry { int wpb = 1, blocks_per_cu = 1; size_t lds = 0; };

struct dsp_handle {
  int device = 0;
  int n = 0, m = 0;
  int64_t nnz = 0;
  int cpl = 0, rpl = 0;
  double eta_unit = 1.0;          // 1 / ||A_scaled||_2
  dsp_options opt;
  DeviceProblem P{};
  std::vector<void *> allocs;     // device allocations owned by the handle
  std::vector<double> dr, dc;
  int *queue = nullptr;           // ring of kQueueRing work-queue slots (64 B apart): launches on different streams may
  std::atomic<unsigned> queue_next{0};  // be in flight together, each needs its own.  Slot = {queue head, count of
                                  // scenarios the simplex pass left unsolved}; both are zeroed on the launch's stream
                                  // right before it (one 8-byte memset), so no host-side bookkeeping of the heads exists
  int lds_limit = 160 * 1024;
  int num_cus = 256;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool geo_valid[2] = {false, false};   // [0] LP kernels, [1] QP instantiations (soft rows)
  Geometry geo[2];
  int matreg = 0;                 // register-resident-matrix kernel available for this shape
  int matreg_qp = 0;              // ... and its QP instantiation
  // run-time compiled specialisation (dsp_rtc.hpp) for shapes without an ahead-of-time one: [0] LP, [1] QP instantiation
  RtcKernel rtc[2];
  int rtc_state[2] = {0, 0};      // 0 = not applicable, 1 = loaded, -1 = to be compiled at the first solve that needs it
  bool rtc_long = false;
  std::string rtc_why;            // why run-time compilation was not possible (dsp_rtc_last_message)
  int simplex = 0;                // tiny LP: the in-wave dense simplex runs first (dsp_simplex.hip)
  const double *A_dense = nullptr;   // [m][n] scaled matrix, row-major (simplex only)
  // warm start of the simplex (dsp_options::simplex_warm): the final basis of every scenario's previous solve on this handle
  double *sx_warm_T = nullptr;
  int *sx_warm_basis = nullptr, *sx_warm_valid = nullptr;
  unsigned char *sx_warm_upper = nullptr;
  int sx_warm_cap = 0;
  std::vector<void *> sx_warm_allocs;   // (earlier, smaller buffers stay alive: captured graphs may hold their addresses)
  int sx_row_stride = 0;
  size_t sx_lds = 0;
  int streaming = 0;              // LP too large for the fused kernels: HBM-resident PDLP (dsp_stream.hip)
  StreamSolver stream;
  int lds_conflicts[4] = {0, 0, 0, 0};   // simulated extra LDS cycles per iteration: y buffer identity/best, x identity/best
};

static int pick(const int *set, int count, int need) {
  for (int i = 0; i < count; ++i)
    if (set[i] >= need) return set[i];
  return -1;
}

template <class T>
static int upload(dsp_handle *h, const std::vector<T> &v, const T **out) {
  void *d = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIP_TRY(hipMalloc(&d, bytes));
  h->allocs.push_back(d);
  if (!v.empty()) HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = reinterpret_cast<const T *>(d);
  return DSP_OK;
}

// pack a lane-major ELL (values + element indices) into the 16-byte device entries (byte offsets)
static void pack_entries(const std::vector<double> &val, const std::vector<uint16_t> &idx, std::vector<Entry> &out) {
  out.resize(val.size());
  for (size_t i = 0; i < val.size(); ++i) { out[i].v = val[i]; out[i].off = (uint32_t)idx[i] * 8u; out[i].pad = 0; }
}

static void pack_entries(const std::vector<double> &val, const std::vector<uint32_t> &off, std::vector<Entry> &out) {
  out.resize(val.size());
  for (size_t i = 0; i < val.size(); ++i) { out[i].v = val[i]; out[i].off = off[i]; out[i].pad = 0; }
}

static int fill_long(LongList &L, const SlotELL &E) {
  if ((int)E.long_owner_pos.size() > kMaxLong) return DSP_ERR_TOO_LARGE;
  L.count = (int)E.long_owner_pos.size();
  for (int i = 0; i < L.count; ++i) { L.owner[i] = E.long_owner_pos[i]; L.start[i] = E.long_start[i]; L.len[i] = E.long_len[i]; }
  return DSP_OK;
}

static int fill_long(LongList &L, const LaneELL &E) {
  if ((int)E.long_owner.size() > kMaxLong) return DSP_ERR_TOO_LARGE;
  L.count = (int)E.long_owner.size();
  for (int i = 0; i < L.count; ++i) { L.owner[i] = E.long_owner[i]; L.start[i] = E.long_start[i]; L.len[i] = E.long_len[i]; }
  return DSP_OK;
}

constexpr int kQueueRing = 256;
constexpr int kQueueStride = 16;       // ints (64 bytes) between ring slots
constexpr int kMaxWavesPerBlock = 8;   // kernels are compiled with __launch_bounds__(512)

// LDS bytes of a block with `wpb` waves
// (solve kernels: + the [cpl + rpl][64] scale factors of the owned columns / rows behind the wave buffers; cpl = 0: spmv_step)
static size_t lds_bytes(const DeviceProblem &P, int wpb, int matreg = 0, int cpl = 0, int rpl = 0) {
  size_t ent = matreg ? (size_t)P.mr_tailc_entries + P.mr_tailr_entries
                      : (size_t)P.tailc_entries + P.tailr_entries + P.ellc_entries + P.ellr_entries;
  size_t per_wave = (size_t)(P.n_pad + P.m_pad) * 8 + (matreg ? (size_t)rare_lds_bytes(rpl) : 0);   // struct Rare, dsp_kernels.hip
  return ent * sizeof(Entry) + (size_t)wpb * per_wave + (size_t)scale_lds_bytes(cpl, rpl);
}

// Launch geometry of the solve kernel: waves (= scenarios in flight) per block and blocks per CU, from the runtime's
// occupancy answer for the compiled kernel (VGPR- and LDS-limited).  Maximise resident waves per CU; on ties prefer
// larger blocks (the shared matrix is staged once per block).  Cached per handle.
static int solve_geometry(dsp_handle *h, int requested, int B, Geometry *g, int qp) {
  const int matreg = qp ? h->matreg_qp : h->matreg;
  if (h->geo_valid[qp] && requested <= 0) { *g = h->geo[qp]; }
  else {
    SolveArgs probe{};
    probe.P = h->P;
    probe.matreg = matreg;
    probe.qp = qp;
    int best_waves = -1;
    Geometry best;
    // generic kernel: larger blocks win ties (the LDS matrix is staged once per block); register-resident matrix:
    // one wave per block wins ties, so every wave's registers are released the moment ITS scenarios are done and a
    // following launch (another stream) can back-fill the CU while this launch's stragglers finish
    for (int step = 0; step < kMaxWavesPerBlock; ++step) {
      const int wpb = matreg ? 1 + step : kMaxWavesPerBlock - step;
      if (requested > 0 && wpb != std::min(requested, kMaxWavesPerBlock)) continue;
      size_t l = lds_bytes(h->P, wpb, matreg, h->cpl, h->rpl);
      if (l > (size_t)h->lds_limit) continue;
      int nb = 0;
      hipError_t e = h->rtc_state[qp] == 1
                         ? hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, h->rtc[qp].fn, 64 * wpb, l)
                         : occupancy_solve(h->cpl, h->rpl, probe, 64 * wpb, l, &nb);
      if (e != hipSuccess) { g_last_hip_error = (int)e; return DSP_ERR_HIP; }
      if (nb * wpb > best_waves) { best_waves = nb * wpb; best.wpb = wpb; best.blocks_per_cu = nb; best.lds = l; }
    }
    if (best_waves <= 0) return DSP_ERR_TOO_LARGE;
    *g = best;
    if (requested <= 0) { h->geo[qp] = best; h->geo_valid[qp] = true; }
  }
  // small batches: spread the scenarios over the CUs instead of packing blocks
  if (requested <= 0) {
    int per_cu = (B + h->num_cus - 1) / h->num_cus;
    int wpb = g->wpb;
    while (wpb > 1 && wpb > per_cu) wpb--;
    if (wpb != g->wpb) { g->wpb = wpb; g->lds = lds_bytes(h->P, wpb, matreg, h->cpl, h->rpl); g->blocks_per_cu = std::max(1, per_cu / wpb); }
  }
  return DSP_OK;
}

// ---- rolling-horizon hand-off of the wind + battery double loop: one thread per plant ------------------------------------------
// Every product is made opaque to the optimiser before it is added (no FMA contraction: the _rn intrinsics are plain
// operators to the compiler and `#pragma clang fp contract(off)` did not keep it from fusing them): the results are
// bit-identical to the element-wise tensor operations this kernel replaces (dispatches_amd/rolling.py, use_fused=False),
// which is how it is tested.
// outcome of the solve that has just finished, folded into the loop's device-side flags (what six tensor launches per solve did)
template <class M, class S>
__device__ __forceinline__ void loop_check(const S &s, const M &m, int b) {
  if (m.status && s.bad && m.status[b] != 0) *s.bad = 1;
  if (m.flags && s.uncertified && (m.flags[b] & DSP_FLAG_OBJ_WAIVED)) atomicAdd(reinterpret_cast<unsigned long long *>(s.uncertified), 1ull);
}

__device__ __forceinline__ double wb_opaque(double v) { asm volatile("" : "+v"(v)); return v; }
__global__ void __launch_bounds__(256) wb_rolling_kernel(dsp_wb_state s, dsp_wb_model rt, dsp_wb_model tr, int phase, int k) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= s.B) return;
  const long long h = *s.hour, st0 = s.start[b];
  auto win = [&](const double *series, int t) { return series[(st0 + h + t) % s.N]; };
  if (phase == 0) {
    const dsp_wb_model &m = rt;
    const int known = min(m.T, 24 - k);
    double *c = m.c + (size_t)b * m.n, *lb = m.lb + (size_t)b * m.n, *ub = m.ub + (size_t)b * m.n;
    double avail_sum = 0.0;
    for (int t = 0; t < m.T; ++t) {
      const double rtp = win(s.rt_series, t);
      const double dap = t < known ? s.da_prices[(size_t)b * 24 + k + t] : win(s.da_series, t);
      const double r3 = wb_opaque(__dmul_rn(1e-3, rtp));
      c[m.pt_cols[t][0]] = __dsub_rn(m.base_c[m.pt_cols[t][0]], r3);
      c[m.pt_cols[t][1]] = __dsub_rn(m.base_c[m.pt_cols[t][1]], r3);
      c[m.pda_cols[t]] = __dsub_rn(m.base_c[m.pda_cols[t]], wb_opaque(__dsub_rn(dap, rtp)));
      const double avail = __dmul_rn(m.wind_kw, win(s.cf_series, t));
      ub[m.wind_cols[t]] = avail;
      avail_sum = t ? __dadd_rn(avail_sum, avail) : avail;
      const double fix = t < known ? s.da_offer[(size_t)b * 24 + k + t] : 0.0;
      lb[m.pda_cols[t]] = fix;
      ub[m.pda_cols[t]] = t < known ? fix : INFINITY;
    }
    if (m.c0) m.c0[b] = __dadd_rn(m.c0_base, wb_opaque(__dmul_rn(m.waste_per_kw, avail_sum)));
    lb[m.soc_init] = s.soc[b]; ub[m.soc_init] = s.soc[b];
    lb[m.thr_init] = s.thr[b]; ub[m.thr_init] = s.thr[b];
  } else if (phase == 1) {
    loop_check(s, rt, b);
    const double *xr = rt.x + (size_t)b * rt.n;
    double *lb = tr.lb + (size_t)b * tr.n, *ub = tr.ub + (size_t)b * tr.n;
    double *rlo = tr.rlo + (size_t)b * tr.m, *rhi = tr.rhi + (size_t)b * tr.m;
    double avail_sum = 0.0;
    for (int t = 0; t < tr.T; ++t) {
      const double offer = __dmul_rn(1e-3, wb_opaque(__dadd_rn(xr[rt.pt_cols[t][0]], xr[rt.pt_cols[t][1]])));
      rlo[tr.track_rows[t]] = offer;
      rhi[tr.track_rows[t]] = offer;
      const double avail = __dmul_rn(tr.wind_kw, win(s.cf_series, t));
      ub[tr.wind_cols[t]] = avail;
      avail_sum = t ? __dadd_rn(avail_sum, avail) : avail;
    }
    if (tr.c0) tr.c0[b] = __dadd_rn(tr.c0_base, wb_opaque(__dmul_rn(tr.waste_per_kw, avail_sum)));
    lb[tr.soc_init] = s.soc[b]; ub[tr.soc_init] = s.soc[b];
    lb[tr.thr_init] = s.thr[b]; ub[tr.thr_init] = s.thr[b];
  } else {
    loop_check(s, tr, b);
    const double *x = tr.x + (size_t)b * tr.n;
    const double delivered = wb_opaque(__dmul_rn(1e-3, wb_opaque(__dadd_rn(x[tr.pt_cols[0][0]], x[tr.pt_cols[0][1]]))));
    const double rt0 = win(s.rt_series, 0);
    s.delivered[b] = delivered;
    s.soc[b] = __ddiv_rn(rint(wb_opaque(__dmul_rn(x[tr.soc0], 100.0))), 100.0);
    s.thr[b] = __ddiv_rn(rint(wb_opaque(__dmul_rn(x[tr.thr0], 100.0))), 100.0);
    const double dao = s.da_offer[(size_t)b * 24 + k], dap = s.da_prices[(size_t)b * 24 + k];
    const double t1 = wb_opaque(__dmul_rn(delivered, rt0)), t2 = wb_opaque(__dmul_rn(dao, wb_opaque(__dsub_rn(dap, rt0))));
    s.revenue[b] = __dadd_rn(s.revenue[b], wb_opaque(__dadd_rn(t1, t2)));
    s.energy_mwh[b] = __dadd_rn(s.energy_mwh[b], delivered);
  }
}

// Zeroes the few ints of a work-queue slot before a solve.  A KERNEL, not hipMemsetAsync: captured into a hipGraph (the rolling loop
// replays its days from graphs) the memset node did not reliably run again on replay (ROCm 7.2, round 6: profiles/r60b_graph_memset.log) -
// the queue head then still held B + waves from the previous replay, every wave left at its first pull, and the day-ahead solve of
// every later day silently returned the captured day's outputs.
__global__ void queue_reset_kernel(int *q, int n) {
  if ((int)threadIdx.x < n) q[threadIdx.x] = 0;
}

// the clock advances AFTER every plant has read it (own launch: stream order is the barrier)
__global__ void wb_clock_kernel(long long *hour) { *hour += 1; }

// ---- the same hand-off for a flowsheet given by a descriptor (include/dsp_hip.h: dsp_loop_model / dsp_loop_state) ------------------------
__device__ __forceinline__ double loop_power(const dsp_loop_model &m, const double *x, int t) {
  double p = m.pt_const[t];
  if (m.pt_cols[t][0] >= 0) p = fma(m.pt_coef[t][0], x[m.pt_cols[t][0]], p);
  if (m.pt_cols[t][1] >= 0) p = fma(m.pt_coef[t][1], x[m.pt_cols[t][1]], p);
  return p;
}
__device__ __forceinline__ void loop_state_and_wind(const dsp_loop_state &s, const dsp_loop_model &m, int b, long long h, long long st0, double c0) {
  double *lb = m.lb + (size_t)b * m.n, *ub = m.ub + (size_t)b * m.n;
  for (int j = 0; j < m.n_state; ++j) {
    const double v = s.state[(size_t)b * m.n_state + j];
    lb[m.state_init[j]] = v; ub[m.state_init[j]] = v;
  }
  if (m.wind_cols[0] >= 0) {
    double sum = 0.0;
    for (int t = 0; t < m.T; ++t) {
      const double avail = m.wind_kw * s.cf_series[(st0 + h + t) % s.N];
      ub[m.wind_cols[t]] = avail;
      sum += avail;
    }
    c0 = fma(m.waste_per_kw, sum, c0);
  }
  m.c0[b] = c0;
}
__global__ void __launch_bounds__(256) loop_update_kernel(dsp_loop_state s, dsp_loop_model rt, dsp_loop_model tr, int phase, int k) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= s.B) return;
  const long long h = *s.hour, st0 = s.start[b];
  auto win = [&](const double *series, int t) { return series[(st0 + h + t) % s.N]; };
  if (phase == 0) {
    const dsp_loop_model &m = rt;
    const int known = min(m.T, 24 - k);
    double *c = m.c + (size_t)b * m.n, *lb = m.lb + (size_t)b * m.n, *ub = m.ub + (size_t)b * m.n;
    double c0 = m.c0_base;
    for (int t = 0; t < m.T; ++t) {
      const double rtp = win(s.rt_series, t);
      const double dap = t < known ? s.da_prices[(size_t)b * 24 + k + t] : win(s.da_series, t);
      for (int e = 0; e < 2; ++e)
        if (m.pt_cols[t][e] >= 0) c[m.pt_cols[t][e]] = fma(-rtp, m.pt_coef[t][e], m.base_c[m.pt_cols[t][e]]);
      c[m.pda_cols[t]] = m.base_c[m.pda_cols[t]] - (dap - rtp);
      c0 = fma(-rtp, m.pt_const[t], c0);
      const double fix = t < known ? s.da_offer[(size_t)b * 24 + k + t] : 0.0;
      lb[m.pda_cols[t]] = fix;
      ub[m.pda_cols[t]] = t < known ? fix : INFINITY;
    }
    loop_state_and_wind(s, m, b, h, st0, c0);
  } else if (phase == 1) {
    loop_check(s, rt, b);
    const double *xr = rt.x + (size_t)b * rt.n;
    double *rlo = tr.rlo + (size_t)b * tr.m, *rhi = tr.rhi + (size_t)b * tr.m;
    for (int t = 0; t < tr.T; ++t) {
      const double rhs = loop_power(rt, xr, t) - tr.pt_const[t];       // real-time offer = SCED dispatch in the stub market
      rlo[tr.track_rows[t]] = rhs;
      rhi[tr.track_rows[t]] = rhs;
    }
    loop_state_and_wind(s, tr, b, h, st0, tr.c0_base);
  } else {
    loop_check(s, tr, b);
    const double *x = tr.x + (size_t)b * tr.n;
    const double delivered = loop_power(tr, x, 0);
    const double rt0 = win(s.rt_series, 0);
    s.delivered[b] = delivered;
    for (int j = 0; j < tr.n_state; ++j)                              // implemented profile -> next hour's state, rounded as update_model does
      s.state[(size_t)b * tr.n_state + j] = rint(x[tr.state_real[j]] * s.state_scale[j]) / s.state_scale[j];
    const double dao = s.da_offer[(size_t)b * 24 + k], dap = s.da_prices[(size_t)b * 24 + k];
    s.revenue[b] += delivered * rt0 + dao * (dap - rt0);
    s.energy_mwh[b] += delivered;
  }
}

extern "C" {

int dsp_wb_rolling_update(const dsp_wb_state *st, const dsp_wb_model *rt, const dsp_wb_model *tr, int32_t phase, int32_t k,
                          void *hipStream) {
  if (!st || !rt || !tr || st->B < 0 || phase < 0 || phase > 2 || k < 0 || k > 23 || rt->T < 1 || rt->T > 8 || tr->T < 1 || tr->T > 8)
    return DSP_ERR_INVALID;
  if (st->B == 0) return DSP_OK;
  hipStream_t s = (hipStream_t)hipStream;
  hipLaunchKernelGGL(wb_rolling_kernel, dim3((st->B + 255) / 256), dim3(256), 0, s, *st, *rt, *tr, (int)phase, (int)k);
  if (phase == 2) hipLaunchKernelGGL(wb_clock_kernel, dim3(1), dim3(1), 0, s, (long long *)st->hour);
  HIP_TRY(hipGetLastError());
  return DSP_OK;
}

int dsp_loop_update(const dsp_loop_state *st, const dsp_loop_model *rt, const dsp_loop_model *tr, int32_t phase, int32_t k, void *hipStream) {
  if (!st || !rt || !tr || st->B < 0 || phase < 0 || phase > 2 || k < 0 || k > 23 || rt->T < 1 || rt->T > DSP_LOOP_MAX_T || tr->T < 1 ||
      tr->T > DSP_LOOP_MAX_T || rt->n_state < 0 || rt->n_state > 2 || tr->n_state != rt->n_state || !rt->c0 || !tr->c0 ||
      ((rt->wind_cols[0] >= 0 || tr->wind_cols[0] >= 0) && !st->cf_series))
    return DSP_ERR_INVALID;
  if (st->B == 0) return DSP_OK;
  hipStream_t s = (hipStream_t)hipStream;
  hipLaunchKernelGGL(loop_update_kernel, dim3((st->B + 255) / 256), dim3(256), 0, s, *st, *rt, *tr, (int)phase, (int)k);
  if (phase == 2) hipLaunchKernelGGL(wb_clock_kernel, dim3(1), dim3(1), 0, s, (long long *)st->hour);
  HIP_TRY(hipGetLastError());
  return DSP_OK;
}

int dsp_bid_points(const dsp_bid_request *rq, void *hipStream) {
  if (!rq || rq->B < 0 || rq->T < 0 || rq->terms < 0 || rq->terms > 2 || rq->ldx < 1 || rq->ldp < rq->T) return DSP_ERR_INVALID;
  if (rq->B > DSP_BID_MAX_SCENARIOS || rq->T > DSP_BID_MAX_HOURS) return DSP_ERR_TOO_LARGE;
  if (rq->T == 0) return DSP_OK;
  if (!rq->out || (rq->B > 0 && (!rq->x || !rq->price))) return DSP_ERR_INVALID;
  for (int t = 0; t < rq->T; ++t)
    for (int e = 0; e < (rq->terms == 2 ? 2 : 1); ++e)
      if (rq->col[t][e] < 0 || rq->col[t][e] >= rq->ldx) return DSP_ERR_INVALID;
  HIP_TRY(launch_bid_points(*rq, (hipStream_t)hipStream));
  return DSP_OK;
}

void dsp_default_options(dsp_options *o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->eps_rel = 1e-9;
  o->eps_obj = 5e-7;
  o->max_iter = 200000;
  o->check_every = 0;
  o->kkt_every = 32;
  o->kkt_gate = 16.0;
  o->stall_rescue = 4000;
  o->no_simplex = 0;
  o->jump_rel = 3.0;
  o->restart_sufficient = 0.2;
  o->restart_necessary = 0.8;
  o->restart_artificial = 0.0;   // automatic (dsp_solve)
  o->pid_kp = 0.0;               // automatic
  o->max_dlog_weight = std::log(30.0);
  o->step_scale = 0.998;
  o->weight_guard = 4.0;
  o->jump_steady = 0.05;
  o->jump_tol = 3e-3;
  o->jump_min = 4.0;
  o->ray_jumps = 1;
  o->ruiz_iters = 10;
  o->waves_per_block = 0;
  o->precision = 0;
  o->polish_patience = 1024;
  o->no_rtc = 0;
  o->no_interior_point = 0;
  o->eps_infeasible = 1e-6;
  o->recertify_passes = 0;
  o->simplex_warm = 0;
  o->warm_patience = 0;
}

int dsp_version(void) { return DSP_VERSION; }
#ifndef DSP_SOURCE_HASH
#define DSP_SOURCE_HASH "unknown"
#endif
// (tagged, so that a build script can read it from the file without loading the library: __graft_entry__._stale)
static const char kSourceHashTag[] = "dsp_source_hash=" DSP_SOURCE_HASH;
const char *dsp_source_hash(void) { return kSourceHashTag + 16; }
int dsp_last_hip_error(void) { return g_last_hip_error; }

const char *dsp_strerror(int code) {
  switch (code) {
    case DSP_OK: return "ok";
    case DSP_ERR_INVALID: return "invalid argument";
    case DSP_ERR_TOO_LARGE: return "LP too large (more than 4096 vectors longer than the ELL width, or dsp_spmv_step on a streaming handle)";
    case DSP_ERR_HIP: return "HIP runtime error (see dsp_last_hip_error)";
    case DSP_ERR_NO_DEVICE: return "no HIP device";
    case DSP_ERR_ALLOC: return "allocation failed";
    default: return "unknown error";
  }
}

int dsp_create(const dsp_lp_desc *d, int device, const dsp_options *opt, dsp_handle **out) {
  if (!d || !out || d->n <= 0 || d->m < 0 || d->nnz < 0 || !d->A_rowptr || (d->nnz && (!d->A_colidx || !d->A_val)))
    return DSP_ERR_INVALID;
  if (d->A_rowptr[0] != 0 || d->A_rowptr[d->m] != d->nnz) return DSP_ERR_INVALID;
  for (int i = 0; i < d->m; ++i) {
    if (d->A_rowptr[i + 1] < d->A_rowptr[i]) return DSP_ERR_INVALID;
    for (int p = d->A_rowptr[i]; p < d->A_rowptr[i + 1]; ++p) {
      if (d->A_colidx[p] < 0 || d->A_colidx[p] >= d->n) return DSP_ERR_INVALID;
      if (p > d->A_rowptr[i] && d->A_colidx[p] <= d->A_colidx[p - 1]) return DSP_ERR_INVALID;
      if (!std::isfinite(d->A_val[p])) return DSP_ERR_INVALID;
    }
  }
  if (d->col_scale)
    for (int j = 0; j < d->n; ++j)
      if (!(d->col_scale[j] > 0.0) || !std::isfinite(d->col_scale[j])) return DSP_ERR_INVALID;
  static const int kCpl[] = {1, 2, 3, 4, 5, 7, 10};
  static const int kRpl[] = {1, 2, 3, 4, 6};
  int cpl = pick(kCpl, 7, (d->n + 63) / 64);
  int rpl = pick(kRpl, 5, std::max(1, (d->m + 63) / 64));
  const bool streaming = cpl < 0 || rpl < 0 || d->n > 65535 || d->m > 65535;
  if (streaming && d->m < 1) return DSP_ERR_INVALID;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DSP_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return DSP_ERR_INVALID;
  HIP_TRY(hipSetDevice(device));

  dsp_handle *h = new (std::nothrow) dsp_handle();
  if (!h) return DSP_ERR_ALLOC;
  h->device = device; h->n = d->n; h->m = d->m; h->nnz = d->nnz; h->cpl = cpl; h->rpl = rpl;
  if (opt) h->opt = *opt; else dsp_default_options(&h->opt);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    h->num_cus = prop.multiProcessorCount;
    h->lds_limit = (int)prop.maxSharedMemoryPerMultiProcessor > 0 ? (int)prop.maxSharedMemoryPerMultiProcessor : 64 * 1024;
  }

  HostCSR A;
  A.m = d->m; A.n = d->n;
  A.ptr.assign(d->A_rowptr, d->A_rowptr + d->m + 1);
  A.idx.assign(d->A_colidx, d->A_colidx + d->nnz);
  A.val.assign(d->A_val, d->A_val + d->nnz);
  HostCSR Au = A;                                   // unscaled copy (streaming SpMV step)
  equilibrate(A, h->opt.ruiz_iters, h->dr, h->dc, std::max(0, h->opt.geo_iters), d->col_scale);
  HostCSR AT = transpose(A), ATu = transpose(Au);
  h->eta_unit = 1.0 / spectral_norm(A, AT, 500);
  if (streaming) {
    // LP beyond the register/LDS-resident kernels (n > 640 or m > 384): HBM-resident PDLP
    int rc2;
    h->streaming = 1; h->cpl = h->rpl = 0;
    if ((rc2 = upload(h, h->dc, &h->P.col_scale)) != DSP_OK || (rc2 = upload(h, h->dr, &h->P.row_scale)) != DSP_OK) { dsp_destroy(h); return rc2; }
    h->P.n = d->n; h->P.m = d->m;
    hipError_t se = stream_create(A, AT, h->P.col_scale, h->P.row_scale, &h->stream);
    if (se != hipSuccess) { g_last_hip_error = (int)se; dsp_destroy(h); return se == hipErrorInvalidValue ? DSP_ERR_TOO_LARGE : DSP_ERR_HIP; }
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { dsp_destroy(h); return DSP_ERR_HIP; }
    *out = h;
    return DSP_OK;
  }

  LaneELL Er = build_lane_ell(A, rpl), Ec = build_lane_ell(AT, cpl);
  // unscaled values in the SAME layout (same sparsity => same W / long split)
  LaneELL Eru = build_lane_ell(Au, rpl), Ecu = build_lane_ell(ATu, cpl);
  DeviceProblem &P = h->P;
  P.n = d->n; P.m = d->m; P.n_pad = cpl * 64; P.m_pad = rpl * 64;
  P.Wc = Ec.W; P.Wr = Er.W;
  P.ellc_entries = (int)Ec.val.size(); P.ellr_entries = (int)Er.val.size();
  P.tailc_entries = (int)Ec.tail_val.size(); P.tailr_entries = (int)Er.tail_val.size();
  int rc;
  if ((rc = fill_long(P.long_c, Ec)) || (rc = fill_long(P.long_r, Er))) { delete h; return rc; }
  // register-resident-matrix layout: ownership sorted by length, per-slot widths, position space
  SortedLayout Lc = sorted_layout(AT, cpl, Ec.long_owner), Lr = sorted_layout(A, rpl, Er.long_owner);
  SlotELL Sc = build_slot_ell(AT, Lc, Lr, Ec.long_owner), Sr = build_slot_ell(A, Lr, Lc, Er.long_owner);
  const bool has_long = P.long_c.count > 0 || P.long_r.count > 0;
  int pad_w = 0;
  // no ahead-of-time specialisation for this shape: compile the tight one now (hiprtc, disk-cached).  Limits: RegEll packs
  // at most 8 slots of width <= 15, and beyond ~32 register-resident entries per lane the kernel would live in scratch.
  const bool aot_lp = matreg_available(cpl, rpl, Sc.pack, Sr.pack, has_long) != 0;
  const bool aot_qp = matreg_available(cpl, rpl, Sc.pack, Sr.pack, has_long, true) != 0;
  bool rtc_ok = false;
  if (!h->opt.no_matreg && !h->opt.no_rtc && !aot_lp && cpl <= 8 && rpl <= 8 && Sc.total + Sr.total <= 32) {
    bool wide = false;
    for (int q = 0; q < Sc.slots; ++q) wide |= Sc.width[q] > 15;
    for (int q = 0; q < Sr.slots; ++q) wide |= Sr.width[q] > 15;
    if (!wide) {
      rtc_ok = rtc_get_kernel(device, cpl, rpl, has_long, Sc.pack, Sr.pack, false, &h->rtc[0], &h->rtc_why);
      if (rtc_ok) { h->rtc_state[0] = 1; h->rtc_long = has_long; if (!aot_qp && !has_long) h->rtc_state[1] = -1; }
    }
  }
  if (!h->opt.no_matreg && !aot_lp && !aot_qp && !rtc_ok) {
    // no tight specialisation for this shape: the PADDED one (every slot kPadWidth wide) if the LP fits it
    int wmax = 0;
    for (int q = 0; q < Sc.slots; ++q) wmax = std::max(wmax, Sc.width[q]);
    for (int q = 0; q < Sr.slots; ++q) wmax = std::max(wmax, Sr.width[q]);
    if (!has_long && wmax <= kPadWidth && matreg_available(cpl, rpl, uniform_pack(cpl, kPadWidth), uniform_pack(rpl, kPadWidth), false)) {
      pad_w = kPadWidth;
      Sc = build_slot_ell(AT, Lc, Lr, Ec.long_owner, pad_w);
      Sr = build_slot_ell(A, Lr, Lc, Er.long_owner, pad_w);
    }
  }
  // the unscaled matrix in the same position-space layout (same sparsity): streaming SpMV step
  SlotELL Scu = build_slot_ell(ATu, Lc, Lr, Ec.long_owner, pad_w), Sru = build_slot_ell(Au, Lr, Lc, Er.long_owner, pad_w);
  // bank-conflict-minimising slot permutation of each exchange buffer: the y buffer is gathered by the A^T entries
  // (Sc), the x buffer by the A entries (Sr)
  const bool shared_slots = shared_slot_maps(cpl, rpl);
  SlotMap My = optimise_slots(Sc, P.m_pad, 4000, shared_slots), Mx = optimise_slots(Sr, P.n_pad, 4000, shared_slots);
  h->lds_conflicts[0] = My.cost_identity; h->lds_conflicts[1] = My.cost_final;
  h->lds_conflicts[2] = Mx.cost_identity; h->lds_conflicts[3] = Mx.cost_final;
  apply_slots(Sc, My.slot);
  apply_slots(Sr, Mx.slot);
  apply_slots(Scu, My.slot);
  apply_slots(Sru, Mx.slot);
  // LDS byte offset of every natural element: column j sits at position Lc.pos[j], whose exchange slot is Mx.slot[...]
  std::vector<int32_t> nat_x((size_t)P.n_pad, 0), nat_y((size_t)P.m_pad, 0);
  for (int j = 0; j < d->n; ++j) nat_x[j] = 8 * Mx.slot[Lc.pos[j]];
  for (int i = 0; i < d->m; ++i) nat_y[i] = 8 * My.slot[Lr.pos[i]];
  P.mr_wc_pack = Sc.pack; P.mr_wr_pack = Sr.pack;
  P.mr_tailc_entries = (int)Sc.tail_val.size(); P.mr_tailr_entries = (int)Sr.tail_val.size();
  if ((rc = fill_long(P.mr_long_c, Sc)) || (rc = fill_long(P.mr_long_r, Sr))) { delete h; return rc; }
  h->matreg = h->opt.no_matreg ? 0 : (rtc_ok ? 1 : matreg_available(cpl, rpl, Sc.pack, Sr.pack, P.long_c.count > 0 || P.long_r.count > 0));
  h->matreg_qp = h->opt.no_matreg ? 0 : matreg_available(cpl, rpl, Sc.pack, Sr.pack, P.long_c.count > 0 || P.long_r.count > 0, true);
  std::vector<Entry> pk;
#define UPE(val, idx, field) pack_entries(val, idx, pk); if ((rc = upload(h, pk, &P.field)) != DSP_OK) { dsp_destroy(h); return rc; }
#define UP(vec, field) if ((rc = upload(h, vec, &P.field)) != DSP_OK) { dsp_destroy(h); return rc; }
  UPE(Ec.val, Ec.idx, ellc) UPE(Er.val, Er.idx, ellr) UPE(Ec.tail_val, Ec.tail_idx, tailc) UPE(Er.tail_val, Er.tail_idx, tailr)
  UPE(Ecu.val, Ecu.idx, ellc_unscaled) UPE(Eru.val, Eru.idx, ellr_unscaled)
  UPE(Ecu.tail_val, Ecu.tail_idx, tailc_unscaled) UPE(Eru.tail_val, Eru.tail_idx, tailr_unscaled)
  UP(h->dc, col_scale) UP(h->dr, row_scale)
  UPE(Sc.val, Sc.off, mr_ellc) UPE(Sr.val, Sr.off, mr_ellr) UPE(Sc.tail_val, Sc.tail_off, mr_tailc) UPE(Sr.tail_val, Sr.tail_off, mr_tailr)
  UP(Lc.at, mr_colat) UP(Lr.at, mr_rowat) UP(Mx.slot, mr_slot_x) UP(My.slot, mr_slot_y)
  UPE(Scu.val, Scu.off, mr_ellc_unscaled) UPE(Sru.val, Sru.off, mr_ellr_unscaled)
  UPE(Scu.tail_val, Scu.tail_off, mr_tailc_unscaled) UPE(Sru.tail_val, Sru.tail_off, mr_tailr_unscaled)
  UP(nat_x, mr_nat_slot_x) UP(nat_y, mr_nat_slot_y)
#undef UP
#undef UPE
  // tiny LPs: dense scaled matrix for the in-wave simplex
  if (!h->opt.no_simplex && d->n + d->m <= 128 && d->m <= 64 && d->m >= 1) {
    h->sx_lds = simplex_lds_bytes(d->n, d->m, &h->sx_row_stride);
    if (h->sx_lds <= (size_t)h->lds_limit) {
      std::vector<double> dense((size_t)d->m * d->n, 0.0);
      for (int i = 0; i < A.m; ++i)
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) dense[(size_t)i * d->n + A.idx[p]] = A.val[p];
      if ((rc = upload(h, dense, &h->A_dense)) != DSP_OK) { dsp_destroy(h); return rc; }
      h->simplex = 1;
    }
  }
  void *q = nullptr;
  if (hipMalloc(&q, sizeof(int) * kQueueRing * kQueueStride) != hipSuccess) { dsp_destroy(h); return DSP_ERR_HIP; }
  h->queue = (int *)q;
  if (hipMemset(q, 0, sizeof(int) * kQueueRing * kQueueStride) != hipSuccess) { dsp_destroy(h); return DSP_ERR_HIP; }
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { dsp_destroy(h); return DSP_ERR_HIP; }
  *out = h;
  return DSP_OK;
}

int dsp_solve(dsp_handle *h, const dsp_batch *batch, const dsp_options *opt, dsp_stats *stats, int sync_stats,
              void *hipStream) {
  if (!h || !batch || batch->B < 0) return DSP_ERR_INVALID;
  const int B = batch->B;
  if (B == 0) { if (stats) std::memset(stats, 0, sizeof(*stats)); return DSP_OK; }
  if (!batch->c || !batch->x || !batch->y || !batch->obj || !batch->status) return DSP_ERR_INVALID;
  hipStream_t st = (hipStream_t)hipStream;
  HIP_TRY(hipSetDevice(h->device));
  SolveArgs a{};
  a.P = h->P; a.b = *batch;
  a.opt = opt ? *opt : h->opt;
  if (a.opt.max_iter < 1 || a.opt.check_every < 0 || a.opt.kkt_every < 0 || !(a.opt.kkt_gate >= 0) || a.opt.stall_rescue < 0 || a.opt.polish_patience < 0 || !(a.opt.jump_rel >= 0) || !(a.opt.eps_rel > 0) || !(a.opt.eps_obj >= 0) || !(a.opt.eps_infeasible >= 0) ||
      !(a.opt.pid_kp >= 0) || !(a.opt.restart_artificial >= 0) || !(a.opt.step_scale > 0))
    return DSP_ERR_INVALID;
  a.eta = a.opt.step_scale * h->eta_unit;
  const int qp = batch->row_compliance != nullptr;
  // 0 = automatic (include/dsp_hip.h): the fused float64 kernels restart their Halpern epochs earlier and steer the primal
  // weight more gently than the HBM-resident and float32 paths, which keep the values they were tuned with
  {
    const bool fused64 = !h->streaming && a.opt.precision == 0;
    if (a.opt.restart_artificial == 0.0) a.opt.restart_artificial = fused64 ? (qp ? 0.3 : 0.2) : 0.36;
    if (a.opt.pid_kp == 0.0) a.opt.pid_kp = fused64 ? 0.6 : 0.7;
  }
  if (a.opt.precision != 0 && a.opt.precision != 1) return DSP_ERR_INVALID;
  // float32 iterates exist in the fused kernels only; soft rows in the fused kernels without long vectors and in the
  // HBM-resident streaming form
  if (a.opt.precision && (h->streaming || h->P.long_c.count > 0 || h->P.long_r.count > 0)) return DSP_ERR_INVALID;
  if (qp && !h->streaming && (h->P.long_c.count > 0 || h->P.long_r.count > 0)) return DSP_ERR_INVALID;
  if (h->streaming) {
    const bool timed_s = stats && sync_stats;
    if (timed_s) HIP_TRY(hipEventRecord(h->ev0, st));
    int periods = 0;
    HIP_TRY(stream_solve(&h->stream, *batch, a.opt, a.eta, st, &periods));
    if (timed_s) HIP_TRY(hipEventRecord(h->ev1, st));
    if (stats) {
      std::memset(stats, 0, sizeof(*stats));
      stats->streaming = 1;
      stats->quadratic = qp;
      stats->grid_blocks = (std::max(h->n, h->m) + 255) / 256; stats->block_threads = 256;
      stats->stream_bytes_per_iteration = (int64_t)stream_bytes_per_iteration(&h->stream);
      stats->stream_form = h->stream.last_form;
      stats->stream_phases = (h->stream.last_form == DSP_STREAM_FORM_LANE || h->stream.last_form == DSP_STREAM_FORM_IPM) ? h->stream.last_phases : 0;
      stats->ipm_solved = h->stream.last_ipm_solved;
      if (sync_stats) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipEventElapsedTime(&stats->kernel_ms, h->ev0, h->ev1));
        std::vector<int32_t> hs(B), hi(B);
        HIP_TRY(hipMemcpy(hs.data(), batch->status, B * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (batch->iters) HIP_TRY(hipMemcpy(hi.data(), batch->iters, B * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < B; ++i) {
          stats->n_optimal += hs[i] == DSP_STATUS_OPTIMAL;
          if (batch->iters) { stats->total_iterations += hi[i]; stats->max_iterations = std::max(stats->max_iterations, hi[i]); }
        }
      }
    }
    return DSP_OK;
  }
  if (a.opt.precision == 1) {
    // float32 iterates (tolerance sweep of BASELINE config 5): its own kernel, one scenario per wave, LDS-resident matrix
    const unsigned slot32 = h->queue_next.fetch_add(1u) % kQueueRing;
    a.queue = h->queue + (size_t)slot32 * kQueueStride;
    a.qp = qp;
    hipLaunchKernelGGL(queue_reset_kernel, dim3(1), dim3(64), 0, st, a.queue, 2);
    const bool timed32 = stats && sync_stats;
    if (timed32) HIP_TRY(hipEventRecord(h->ev0, st));
    int grid32 = 0, threads32 = 0;
    size_t lds32 = 0;
    HIP_TRY(launch_solve_f32(h->cpl, h->rpl, a, h->num_cus, (size_t)h->lds_limit, st, &grid32, &threads32, &lds32));
    if (timed32) HIP_TRY(hipEventRecord(h->ev1, st));
    if (stats) {
      std::memset(stats, 0, sizeof(*stats));
      stats->grid_blocks = grid32; stats->block_threads = threads32; stats->lds_bytes = (int)lds32;
      stats->cols_per_lane = h->cpl; stats->rows_per_lane = h->rpl; stats->quadratic = qp; stats->precision = 1;
      if (sync_stats) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipEventElapsedTime(&stats->kernel_ms, h->ev0, h->ev1));
        std::vector<int32_t> hs(B), hi(B);
        HIP_TRY(hipMemcpy(hs.data(), batch->status, B * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (batch->iters) HIP_TRY(hipMemcpy(hi.data(), batch->iters, B * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < B; ++i) {
          stats->n_optimal += hs[i] == DSP_STATUS_OPTIMAL;
          if (batch->iters) { stats->total_iterations += hi[i]; stats->max_iterations = std::max(stats->max_iterations, hi[i]); }
        }
      }
    }
    return DSP_OK;
  }
  if (qp && h->rtc_state[1] == -1) {
    // first QP solve on a run-time specialised shape: compile its QP instantiation (the layout is the tight one already)
    h->rtc_state[1] = rtc_get_kernel(h->device, h->cpl, h->rpl, false, h->P.mr_wc_pack, h->P.mr_wr_pack, true, &h->rtc[1], &h->rtc_why) ? 1 : 0;
    if (h->rtc_state[1] == 1) h->matreg_qp = 1;
  }
  Geometry geo;
  int grc = solve_geometry(h, a.opt.waves_per_block, B, &geo, qp);
  if (grc != DSP_OK) return grc;
  a.waves_per_block = geo.wpb;
  const size_t lds = geo.lds;
  const unsigned slot = h->queue_next.fetch_add(1u) % kQueueRing;
  a.queue = h->queue + (size_t)slot * kQueueStride;
  a.queue_base = 0u;
  a.unsolved = a.queue + 1;
  a.suspects = a.queue + 2;                          // (+ 3: the work-queue head of the certificate pass; + 4: count of scenarios flagged
                                                     //  DSP_FLAG_OBJ_WAIVED; + 5 .. 7: heads of the re-certification passes)
  hipLaunchKernelGGL(queue_reset_kernel, dim3(1), dim3(64), 0, st, a.queue, 8);
  a.matreg = qp ? h->matreg_qp : h->matreg;
  a.qp = qp;
#ifdef DSP_KKT_TRACE
  const char *trace_env = getenv("DSP_TRACE_SCENARIO");
  double *trace_dev = nullptr;
  if (trace_env) {
    HIP_TRY(hipMalloc((void **)&trace_dev, 4096 * 12 * sizeof(double)));
    HIP_TRY(hipMemsetAsync(trace_dev, 0, 4096 * 12 * sizeof(double), st));
    a.trace = trace_dev; a.trace_scenario = atoi(trace_env);
  }
#endif
  int grid = std::min((B + geo.wpb - 1) / geo.wpb, h->num_cus * geo.blocks_per_cu);
  // geometry of the certificate pass (below), fixed BEFORE the first pass runs: a register-resident kernel may only leave scenarios
  // DSP_STATUS_SUSPECT if the generic kernel that takes them over can be launched (it always can for an LP of the fused path - it is
  // the fallback of every shape -; if not, the first pass keeps every scenario and the certificates are off for this call)
  int cert_wpb = 0;
  size_t cert_lds = 0;
  if (a.matreg && a.opt.eps_infeasible > 0.0) {
    for (int wpb = 1; wpb <= 4; ++wpb) {              // a few waves per block share the LDS matrix; the pass is rare, not tuned
      const size_t l = lds_bytes(h->P, wpb, 0, h->cpl, h->rpl);
      if (l <= (size_t)h->lds_limit) { cert_wpb = wpb; cert_lds = l; }
    }
    if (!cert_wpb) a.opt.eps_infeasible = 0.0;
  }
  const bool timed = stats && sync_stats;
  if (timed) HIP_TRY(hipEventRecord(h->ev0, st));
  if (h->simplex && !qp) {
    // simplex pass: certified vertices get their final status; DSP_STATUS_UNSOLVED marks what the PDLP kernel still has to do
    SimplexArgs sa{};
    sa.n = h->n; sa.m = h->m; sa.row_stride = h->sx_row_stride; sa.max_pivots = 20 * (h->n + h->m);
    sa.A_dense = h->A_dense; sa.col_scale = h->P.col_scale; sa.row_scale = h->P.row_scale; sa.b = *batch;
    sa.tol_p = 1e-10; sa.tol_d = 1e-12; sa.tol_piv = 1e-9;
    { static const char *tp = getenv("DSP_SX_TOLP"); if (tp) sa.tol_p = atof(tp); }      // (development)
    sa.unsolved = a.queue + 1;
    if (a.opt.simplex_warm == 1 || a.opt.simplex_warm == 2) {
      if (B > h->sx_warm_cap) {
        // (allocated outside any stream capture: the rolling loops run their first day eagerly)
        const size_t N = (size_t)h->n + h->m;
        double *t = nullptr; int *bs = nullptr, *vl = nullptr; unsigned char *up = nullptr;
        HIP_TRY(hipMalloc((void **)&t, (size_t)B * h->m * N * sizeof(double)));
        h->sx_warm_allocs.push_back(t);
        HIP_TRY(hipMalloc((void **)&bs, (size_t)B * h->m * sizeof(int)));
        h->sx_warm_allocs.push_back(bs);
        HIP_TRY(hipMalloc((void **)&up, (size_t)B * N));
        h->sx_warm_allocs.push_back(up);
        HIP_TRY(hipMalloc((void **)&vl, (size_t)B * sizeof(int)));
        h->sx_warm_allocs.push_back(vl);
        HIP_TRY(hipMemset(vl, 0, (size_t)B * sizeof(int)));
        h->sx_warm_T = t; h->sx_warm_basis = bs; h->sx_warm_upper = up; h->sx_warm_valid = vl; h->sx_warm_cap = B;
      }
      sa.warm = a.opt.simplex_warm;
      sa.warm_T = h->sx_warm_T; sa.warm_basis = h->sx_warm_basis; sa.warm_upper = h->sx_warm_upper; sa.warm_valid = h->sx_warm_valid;
    }
    static const int sx_debug = getenv("DSP_SX_DEBUG") ? atoi(getenv("DSP_SX_DEBUG")) : 0;
    sa.debug_keep = sx_debug;
    const int per_cu = std::max<int>(1, std::min<int>(32, (int)((size_t)h->lds_limit / h->sx_lds)));
    const int sgrid = std::min(B, h->num_cus * per_cu);
    HIP_TRY(launch_simplex(sa, sgrid, h->sx_lds, st));
    a.skip_solved = 1;
  }
  if (h->rtc_state[qp] == 1 && a.matreg) {
    size_t arg_size = sizeof(a);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_size, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(h->rtc[qp].fn, grid, 1, 1, 64 * a.waves_per_block, 1, 1, (unsigned)lds, st, nullptr, config));
  } else {
    HIP_TRY(launch_solve(h->cpl, h->rpl, a, dim3(grid), dim3(64 * a.waves_per_block), lds, st));
  }
  if (a.matreg && a.opt.eps_infeasible > 0.0) {
    // Certificate pass: the register-resident kernels only WATCH for scenarios whose objectives keep drifting apart and leave them
    // DSP_STATUS_SUSPECT; the generic kernel - which evaluates the infeasibility / unboundedness certificates at its restarts -
    // continues those from their iterate (x0 / y0 = the outputs of the first pass).  One more launch per solve that returns at
    // its first line when the device-side count of suspects is 0: every feasible batch.
    SolveArgs c = a;
    c.matreg = 0;
    c.skip_solved = 2;
    c.queue = a.queue + 3;
    c.b.x0 = a.b.x; c.b.y0 = a.b.y;
    c.waves_per_block = cert_wpb;
    // How many blocks: a caller that waits for the statistics anyway (sync_stats) lets the host read the count of suspects after the
    // first pass - no launch at all on a feasible batch, one block per suspect otherwise.  Asynchronous callers (pipelined batches,
    // the rolling loop's graphs) get a launch of 8 blocks that returns at its first line when there is no suspect: with one block
    // per CU it cost the pipelined metric batch 1.2 % (every block needs a free 256-VGPR wave slot before it can even return), with
    // 8 blocks 0.5 %, without the pass 0 (profiles/r51d_cert_pass_cost.log); 32 waves are plenty for the rare suspect.
    static const int cert_grid_env = getenv("DSP_CERT_GRID") ? atoi(getenv("DSP_CERT_GRID")) : 0;      // development: blocks of the pass
    int nsus = -1;
    if (stats && sync_stats) {
      HIP_TRY(hipMemcpyAsync(&nsus, a.suspects, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    if (nsus != 0) {
      const int want = nsus > 0 ? (nsus + cert_wpb - 1) / cert_wpb : (cert_grid_env > 0 ? cert_grid_env : 8);
      const int grid2 = std::max(1, std::min(std::min((B + cert_wpb - 1) / cert_wpb, want), 2 * h->num_cus));
      HIP_TRY(launch_solve(h->cpl, h->rpl, c, dim3(grid2), dim3(64 * cert_wpb), cert_lds, st));
    }
  }
  if (a.opt.recertify_passes > 0 && batch->flags) {
    // Re-certification passes (dsp_options::recertify_passes): scenarios the passes above accepted with DSP_FLAG_OBJ_WAIVED - the objective-error
    // bound stuck within 10 eps_obj while the objective itself had stopped - are solved again from a cold start under another restart
    // cadence / weight controller, on the device, by the generic kernel; a certified optimum replaces the flagged point, anything else
    // leaves it (and its flag) in place.  Each pass returns at its first line when the device-side count of flagged scenarios is 0:
    // almost every batch (one flagged solve in the 147 M of a simulated year of the rolling loop).  What HipPdlpSolver._recertify does
    // from the host for synchronous callers, for callers that never look (hipGraph replays, pipelined batches).
    static const struct { double pid_kp, restart_artificial; int check_every; } kRecertify[3] = {{0.45, 0.3, 0}, {0.8, 0.15, 0}, {0.3, 0.5, 12}};
    int rwpb = 0;
    size_t rlds = 0;
    for (int wpb = 1; wpb <= 4; ++wpb) {
      const size_t l = lds_bytes(h->P, wpb, 0, h->cpl, h->rpl);
      if (l <= (size_t)h->lds_limit) { rwpb = wpb; rlds = l; }
    }
    for (int v = 0; rwpb && v < std::min(a.opt.recertify_passes, 3); ++v) {
      SolveArgs c = a;
      c.matreg = 0;
      c.skip_solved = 3;
      c.queue = a.queue + 5 + v;
      c.b.x0 = nullptr; c.b.y0 = nullptr;
      c.waves_per_block = rwpb;
      c.opt.pid_kp = kRecertify[v].pid_kp;
      c.opt.restart_artificial = kRecertify[v].restart_artificial;
      if (kRecertify[v].check_every) c.opt.check_every = kRecertify[v].check_every;
      c.opt.polish_patience = std::max(c.opt.polish_patience, 1024);       // (a pass that waives as readily as the first certifies nothing)
      const int rgrid = std::max(1, std::min((B + rwpb - 1) / rwpb, 64));
      HIP_TRY(launch_solve(h->cpl, h->rpl, c, dim3(rgrid), dim3(64 * rwpb), rlds, st));
    }
  }
  if (timed) HIP_TRY(hipEventRecord(h->ev1, st));
#ifdef DSP_KKT_TRACE
  if (trace_dev) {
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<double> tr(4096 * 12);
    HIP_TRY(hipMemcpy(tr.data(), trace_dev, tr.size() * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(trace_dev);
    const char *path = getenv("DSP_TRACE_FILE");
    if (FILE *f = fopen(path ? path : "dsp_trace.bin", "wb")) { fwrite(tr.data(), sizeof(double), tr.size(), f); fclose(f); }
  }
#endif
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->grid_blocks = grid; stats->block_threads = 64 * a.waves_per_block; stats->lds_bytes = (int)lds;
    stats->cols_per_lane = h->cpl; stats->rows_per_lane = h->rpl; stats->matreg = a.matreg; stats->simplex = h->simplex && !qp;
    stats->quadratic = qp;
    stats->rtc = (h->rtc_state[qp] == 1 && a.matreg) ? 1 : 0;
    stats->lds_conflicts_identity = h->lds_conflicts[0] + h->lds_conflicts[2];
    stats->lds_conflicts_chosen = a.matreg ? h->lds_conflicts[1] + h->lds_conflicts[3] : stats->lds_conflicts_identity;
    if (sync_stats) {
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipEventElapsedTime(&stats->kernel_ms, h->ev0, h->ev1));
      std::vector<int32_t> hs(B), hi(B);
      HIP_TRY(hipMemcpy(hs.data(), batch->status, B * sizeof(int32_t), hipMemcpyDeviceToHost));
      if (batch->iters) HIP_TRY(hipMemcpy(hi.data(), batch->iters, B * sizeof(int32_t), hipMemcpyDeviceToHost));
      for (int i = 0; i < B; ++i) {
        stats->n_optimal += hs[i] == DSP_STATUS_OPTIMAL;
        if (batch->iters) { stats->total_iterations += hi[i]; stats->max_iterations = std::max(stats->max_iterations, hi[i]); }
      }
    }
  }
  return DSP_OK;
}

int dsp_spmv_step(dsp_handle *h, int32_t B, const double *X, const double *Y, double *AX, double *ATY, void *hipStream) {
  if (!h || B < 0 || !X || !Y || !AX || !ATY) return DSP_ERR_INVALID;
  if (B == 0) return DSP_OK;
  if (h->streaming) return DSP_ERR_TOO_LARGE;        // the streaming solver has its own fused sweeps
  HIP_TRY(hipSetDevice(h->device));
  SpmvArgs a{};
  a.P = h->P; a.B = B; a.X = X; a.Y = Y; a.AX = AX; a.ATY = ATY;
  // The matrix is staged once per 8-wave block in LDS (shared by its waves).  (A register-resident-matrix form - one scenario per
  // wave, no block barrier - was measured slower at every batch size in round 2 and is gone.)  DSP_SPMV_WPB: waves per block.
  static const int wpb_env = getenv("DSP_SPMV_WPB") ? atoi(getenv("DSP_SPMV_WPB")) : 0;
  // generic form (matrix staged in LDS per block): 8-wave blocks, as many as LDS admits per CU
  constexpr int kSpmvMaxWaves = 16;                 // spmv_step_kernel is compiled with __launch_bounds__(1024)
  a.waves_per_block = wpb_env > 0 ? std::min(wpb_env, kSpmvMaxWaves) : kMaxWavesPerBlock;
  size_t lds = lds_bytes(h->P, a.waves_per_block);
  while (lds > (size_t)h->lds_limit && a.waves_per_block > 1) lds = lds_bytes(h->P, --a.waves_per_block);
  if (lds > (size_t)h->lds_limit) return DSP_ERR_TOO_LARGE;
  int blocks_per_cu = std::max<int>(1, std::min<int>((int)(h->lds_limit / lds), 32 / a.waves_per_block));
  int grid = std::min((B + a.waves_per_block - 1) / a.waves_per_block, h->num_cus * blocks_per_cu);
  const hipError_t se = launch_spmv(h->cpl, h->rpl, a, dim3(grid), dim3(64 * a.waves_per_block), lds, (hipStream_t)hipStream);
  if (se == hipErrorInvalidValue) return DSP_ERR_INVALID;          // a measurement kernel: benchmark shapes only (dsp_kernels.hip)
  HIP_TRY(se);
  return DSP_OK;
}

/* Development / test hook: compile (or fetch from the cache) the run-time specialisation of one shape WITHOUT a GPU.
 * Returns the code-object size in bytes, or 0 with the reason in msg. */
int dsp_rtc_compile_check(int cpl, int rpl, int has_long, unsigned wc_pack, unsigned wr_pack, int qp, char *msg, int msg_len) {
  std::vector<char> code;
  std::string name, why;
  const bool ok = rtc_build_code(cpl, rpl, has_long != 0, wc_pack, wr_pack, qp != 0, &code, &name, &why);
  if (msg && msg_len > 0) { std::snprintf(msg, (size_t)msg_len, "%s", ok ? name.c_str() : why.c_str()); }
  return ok ? (int)code.size() : 0;
}

/* Why the last dsp_create on this handle could not (or did not need to) compile at run time; "" if it did. */
const char *dsp_rtc_message(const dsp_handle *h) { return h ? h->rtc_why.c_str() : ""; }

int dsp_get_dims(const dsp_handle *h, int32_t *n, int32_t *m, int64_t *nnz) {
  if (!h) return DSP_ERR_INVALID;
  if (n) *n = h->n;
  if (m) *m = h->m;
  if (nnz) *nnz = h->nnz;
  return DSP_OK;
}

int dsp_get_scaling(const dsp_handle *h, double *row_scale, double *col_scale, double *step_eta) {
  if (!h) return DSP_ERR_INVALID;
  if (row_scale) std::memcpy(row_scale, h->dr.data(), h->dr.size() * sizeof(double));
  if (col_scale) std::memcpy(col_scale, h->dc.data(), h->dc.size() * sizeof(double));
  if (step_eta) *step_eta = h->opt.step_scale * h->eta_unit;
  return DSP_OK;
}

int dsp_destroy(dsp_handle *h) {
  if (!h) return DSP_OK;
  (void)hipSetDevice(h->device);
  for (void *p : h->allocs) (void)hipFree(p);
  stream_destroy(&h->stream);
  if (h->queue) (void)hipFree(h->queue);
  for (void *p : h->sx_warm_allocs) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return DSP_OK;
}

}  // extern "C"
