// dsp_device.hpp — plain structs shared by the kernels (dsp_kernels.hip) and the C ABI (dsp_capi.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dsp_hip.h"

namespace dsp {

constexpr int kMaxLong = 32;   // long vectors (longer than the ELL width) per orientation

struct LongList {
  int count;
  int owner[kMaxLong];   // vector (column / row) index the partial sums belong to
  int start[kMaxLong];   // offset into the tail arrays (multiple of 64)
  int len[kMaxLong];     // padded length (multiple of 64)
};

// Device-resident, scenario-independent data of one (flowsheet, horizon).
struct DeviceProblem {
  int n, m;
  int n_pad, m_pad;            // CPL*64, RPL*64
  int Wc, Wr;                  // ELL widths of A^T (columns) and A (rows)
  int ellc_entries, ellr_entries, tailc_entries, tailr_entries;
  const double *ellc_val, *ellr_val, *tailc_val, *tailr_val;                       // scaled matrix
  const double *ellc_val_unscaled, *ellr_val_unscaled, *tailc_val_unscaled, *tailr_val_unscaled;
  const uint16_t *ellc_idx, *ellr_idx, *tailc_idx, *tailr_idx;
  const double *col_scale, *row_scale;                                             // D_c [n], D_r [m]
  LongList long_c, long_r;
};

struct SolveArgs {
  DeviceProblem P;
  int B;
  int waves_per_block;
  double eta;
  dsp_options opt;
  int *queue;                  // device work-queue head (zeroed before the launch)
  const double *c, *var_lb, *var_ub, *row_lb, *row_ub, *x0, *y0;
  long long c_stride, var_lb_stride, var_ub_stride, row_lb_stride, row_ub_stride;
  double *x, *y, *obj;
  int *status, *iters;
};

struct SpmvArgs {
  DeviceProblem P;
  int B;
  int waves_per_block;
  const double *X, *Y;
  double *AX, *ATY;
};

hipError_t launch_solve(int cpl, int rpl, const SolveArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t launch_spmv(int cpl, int rpl, const SpmvArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);

}  // namespace dsp
