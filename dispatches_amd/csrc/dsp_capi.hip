// dsp_capi.hip — the C ABI of include/dsp_hip.h: handle management, host-side preparation, launch geometry.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "dsp_device.hpp"
#include "dsp_prepare.hpp"

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
  int *queue = nullptr;
  int lds_limit = 160 * 1024;
  int num_cus = 256;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
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

static int fill_long(LongList &L, const LaneELL &E) {
  if ((int)E.long_owner.size() > kMaxLong) return DSP_ERR_TOO_LARGE;
  L.count = (int)E.long_owner.size();
  for (int i = 0; i < L.count; ++i) { L.owner[i] = E.long_owner[i]; L.start[i] = E.long_start[i]; L.len[i] = E.long_len[i]; }
  return DSP_OK;
}

constexpr int kMaxWavesPerBlock = 8;   // kernels are compiled with __launch_bounds__(512)

// LDS bytes of a block with `wpb` waves
static size_t lds_bytes(const DeviceProblem &P, int wpb) {
  size_t dbl = (size_t)P.ellc_entries + P.ellr_entries + P.tailc_entries + P.tailr_entries + (size_t)wpb * (P.n_pad + P.m_pad);
  size_t u16 = (size_t)P.ellc_entries + P.ellr_entries + P.tailc_entries + P.tailr_entries;
  return ((dbl * 8 + u16 * 2 + 15) / 16) * 16;
}

// choose waves per block: maximise resident waves per CU (<= 32), prefer fewer, larger blocks on ties
static int choose_wpb(const dsp_handle *h, int requested, int B) {
  if (requested > 0) return std::min(requested, kMaxWavesPerBlock);
  int best = 1, best_waves = 0;
  for (int wpb = 1; wpb <= kMaxWavesPerBlock; ++wpb) {
    size_t l = lds_bytes(h->P, wpb);
    if (l > (size_t)h->lds_limit) break;
    int blocks = std::min<int>((int)(h->lds_limit / l), 32 / wpb);
    int waves = blocks * wpb;
    if (waves > best_waves) { best_waves = waves; best = wpb; }
  }
  // small batches: do not pack more waves into a block than needed to cover B across the CUs
  int per_cu = (B + h->num_cus - 1) / h->num_cus;
  while (best > 1 && best > per_cu) best--;
  return std::max(best, 1);
}

extern "C" {

void dsp_default_options(dsp_options *o) {
  if (!o) return;
  o->eps_rel = 1e-9;
  o->max_iter = 200000;
  o->check_every = 32;
  o->restart_sufficient = 0.2;
  o->restart_necessary = 0.8;
  o->restart_artificial = 0.36;
  o->pid_kp = 0.7;
  o->max_dlog_weight = std::log(30.0);
  o->step_scale = 0.998;
  o->ruiz_iters = 10;
  o->waves_per_block = 0;
}

int dsp_version(void) { return DSP_VERSION; }
int dsp_last_hip_error(void) { return g_last_hip_error; }

const char *dsp_strerror(int code) {
  switch (code) {
    case DSP_OK: return "ok";
    case DSP_ERR_INVALID: return "invalid argument";
    case DSP_ERR_TOO_LARGE: return "LP too large for the LDS-resident solver (n <= 640, m <= 384 and the matrix must fit LDS)";
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
  static const int kCpl[] = {1, 2, 3, 4, 5, 7, 10};
  static const int kRpl[] = {1, 2, 3, 4, 6};
  const int cpl = pick(kCpl, 7, (d->n + 63) / 64);
  const int rpl = pick(kRpl, 5, std::max(1, (d->m + 63) / 64));
  if (cpl < 0 || rpl < 0 || d->n > 65535 || d->m > 65535) return DSP_ERR_TOO_LARGE;

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
  equilibrate(A, h->opt.ruiz_iters, h->dr, h->dc);
  HostCSR AT = transpose(A), ATu = transpose(Au);
  h->eta_unit = 1.0 / spectral_norm(A, AT, 500);

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
#define UP(vec, field) if ((rc = upload(h, vec, &P.field)) != DSP_OK) { dsp_destroy(h); return rc; }
  UP(Ec.val, ellc_val) UP(Er.val, ellr_val) UP(Ec.tail_val, tailc_val) UP(Er.tail_val, tailr_val)
  UP(Ecu.val, ellc_val_unscaled) UP(Eru.val, ellr_val_unscaled)
  UP(Ecu.tail_val, tailc_val_unscaled) UP(Eru.tail_val, tailr_val_unscaled)
  UP(Ec.idx, ellc_idx) UP(Er.idx, ellr_idx) UP(Ec.tail_idx, tailc_idx) UP(Er.tail_idx, tailr_idx)
  UP(h->dc, col_scale) UP(h->dr, row_scale)
#undef UP
  void *q = nullptr;
  if (hipMalloc(&q, sizeof(int)) != hipSuccess) { dsp_destroy(h); return DSP_ERR_HIP; }
  h->queue = (int *)q;
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { dsp_destroy(h); return DSP_ERR_HIP; }
  *out = h;
  return DSP_OK;
}

int dsp_solve(dsp_handle *h, int32_t B, const double *c, int64_t c_stride, const double *var_lb, int64_t var_lb_stride,
              const double *var_ub, int64_t var_ub_stride, const double *row_lb, int64_t row_lb_stride,
              const double *row_ub, int64_t row_ub_stride, const double *x0, const double *y0,
              const dsp_options *opt, double *x, double *y, double *obj, int32_t *status, int32_t *iters,
              dsp_stats *stats, int sync_stats, void *hipStream) {
  if (!h || B < 0 || !c || !x || !y || !obj || !status) return DSP_ERR_INVALID;
  if (B == 0) { if (stats) std::memset(stats, 0, sizeof(*stats)); return DSP_OK; }
  hipStream_t st = (hipStream_t)hipStream;
  HIP_TRY(hipSetDevice(h->device));
  SolveArgs a{};
  a.P = h->P; a.B = B;
  a.opt = opt ? *opt : h->opt;
  if (a.opt.max_iter < 1 || a.opt.check_every < 1 || !(a.opt.eps_rel > 0) || !(a.opt.pid_kp >= 0) || !(a.opt.step_scale > 0))
    return DSP_ERR_INVALID;
  a.eta = a.opt.step_scale * h->eta_unit;
  a.waves_per_block = choose_wpb(h, a.opt.waves_per_block, B);
  size_t lds = lds_bytes(h->P, a.waves_per_block);
  if (lds > (size_t)h->lds_limit) return DSP_ERR_TOO_LARGE;
  a.queue = h->queue;
  a.c = c; a.var_lb = var_lb; a.var_ub = var_ub; a.row_lb = row_lb; a.row_ub = row_ub; a.x0 = x0; a.y0 = y0;
  a.c_stride = c_stride; a.var_lb_stride = var_lb_stride; a.var_ub_stride = var_ub_stride;
  a.row_lb_stride = row_lb_stride; a.row_ub_stride = row_ub_stride;
  a.x = x; a.y = y; a.obj = obj; a.status = status; a.iters = iters;
  int blocks_per_cu = std::max<int>(1, std::min<int>((int)(h->lds_limit / lds), 32 / a.waves_per_block));
  int grid = std::min((B + a.waves_per_block - 1) / a.waves_per_block, h->num_cus * blocks_per_cu);
  HIP_TRY(hipMemsetAsync(h->queue, 0, sizeof(int), st));
  const bool timed = stats && sync_stats;
  if (timed) HIP_TRY(hipEventRecord(h->ev0, st));
  HIP_TRY(launch_solve(h->cpl, h->rpl, a, dim3(grid), dim3(64 * a.waves_per_block), lds, st));
  if (timed) HIP_TRY(hipEventRecord(h->ev1, st));
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->grid_blocks = grid; stats->block_threads = 64 * a.waves_per_block; stats->lds_bytes = (int)lds;
    stats->cols_per_lane = h->cpl; stats->rows_per_lane = h->rpl;
    if (sync_stats) {
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipEventElapsedTime(&stats->kernel_ms, h->ev0, h->ev1));
      std::vector<int32_t> hs(B), hi(B);
      HIP_TRY(hipMemcpy(hs.data(), status, B * sizeof(int32_t), hipMemcpyDeviceToHost));
      if (iters) HIP_TRY(hipMemcpy(hi.data(), iters, B * sizeof(int32_t), hipMemcpyDeviceToHost));
      for (int i = 0; i < B; ++i) {
        stats->n_optimal += hs[i] == DSP_STATUS_OPTIMAL;
        if (iters) { stats->total_iterations += hi[i]; stats->max_iterations = std::max(stats->max_iterations, hi[i]); }
      }
    }
  }
  return DSP_OK;
}

int dsp_spmv_step(dsp_handle *h, int32_t B, const double *X, const double *Y, double *AX, double *ATY, void *hipStream) {
  if (!h || B < 0 || !X || !Y || !AX || !ATY) return DSP_ERR_INVALID;
  if (B == 0) return DSP_OK;
  HIP_TRY(hipSetDevice(h->device));
  SpmvArgs a{};
  a.P = h->P; a.B = B; a.X = X; a.Y = Y; a.AX = AX; a.ATY = ATY;
  a.waves_per_block = choose_wpb(h, 0, B);
  size_t lds = lds_bytes(h->P, a.waves_per_block);
  if (lds > (size_t)h->lds_limit) return DSP_ERR_TOO_LARGE;
  int blocks_per_cu = std::max<int>(1, std::min<int>((int)(h->lds_limit / lds), 32 / a.waves_per_block));
  int grid = std::min((B + a.waves_per_block - 1) / a.waves_per_block, h->num_cus * blocks_per_cu);
  HIP_TRY(launch_spmv(h->cpl, h->rpl, a, dim3(grid), dim3(64 * a.waves_per_block), lds, (hipStream_t)hipStream));
  return DSP_OK;
}

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
  if (h->queue) (void)hipFree(h->queue);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return DSP_OK;
}

}  // extern "C"
