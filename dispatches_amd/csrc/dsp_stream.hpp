// dsp_stream.hpp — data structures of the HBM-resident ("streaming") batched PDLP (dsp_stream.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <vector>

#include "../../include/dsp_hip.h"
#include "dsp_prepare.hpp"

namespace dsp {

constexpr int kStreamMaxW = 8;         // ELL width cap: longer vectors go to the long list
constexpr int kStreamMaxLong = 4096;   // long vectors per orientation

// one orientation of the scaled matrix: entry-major ELL ([W][nvec]) + CSR segments of the long vectors
struct StreamMatrix {
  int nvec, W, nlong;
  const double *val;          // [W][nvec]
  const int32_t *idx;         // [W][nvec]
  const uint8_t *is_long;     // [nvec]
  const int32_t *long_id;     // [nlong]   vector id
  const int32_t *long_ptr;    // [nlong+1]
  const int32_t *long_idx;
  const double *long_val;
  int nchunk;                 // chunks of <= 2048 entries over all long vectors
  const int32_t *chunk_begin, *chunk_end;   // [nchunk]   entry range of each chunk
  const int32_t *long_chunk_ptr;            // [nlong+1]  chunks of each long vector
};

// Fused iteration (k_fused): ONE launch per plain iteration for BANDED matrices (the multi-period LPs of the reference: period-
// major columns and rows, every non-long vector touches its own and the neighbouring period).  Rows and columns are cut into
// `ntile` contiguous tiles; a tile's workgroup stages y of its rows (+ the halo rows its columns touch) in LDS, computes the
// primal step of its columns (+ the halo columns its rows touch) into LDS, then the dual step and both Halpern averagings of
// what it owns: xbar never travels through HBM, and x, y are read and written once.  Long columns (design variables that
// touch every period) enter through per-tile partial sums of A^T y that every tile leaves behind for the next launch.
constexpr int kFusedMaxLong = 4;       // long columns the fused form carries (design variables)
struct FusedPlan {
  int ntile = 0;                       // 0 = not applicable (not banded enough, long rows, too many long columns)
  int rows_per_tile = 0;
  int ny_max = 0, nxb_max = 0;         // LDS extents per scenario: staged y rows, xbar slots (long columns first)
  int own_max = 0;                     // most rows or columns any tile owns (k_fused_pre: K = ceil(own_max / 256) per thread)
  int halo_max = 0;                    // most halo rows / columns of any tile (k_fused_pre handles halo columns in one pass: <= 256)
  const int32_t *tile = nullptr;       // [ntile][8]  i0, i1 (own rows), j0, j1 (own columns), c_lo, c_hi, r_lo, r_hi
  const int32_t *ridx_enc = nullptr;   // [R.W][m] row ELL column index; long column l encoded as -1 - l
};

struct StreamProblem {
  int n, m;
  StreamMatrix R, C;          // A (rows) and A^T (columns)
  const double *col_scale, *row_scale;
  FusedPlan F;
};

// per-scenario control block (device memory; written by k_init_control / k_control only)
struct StreamCtrl {
  double w, tau, sig, w_lo, w_hi;
  double r0, rprev;
  double qn, cn, c0, pobj;
  double last_rp, last_rd, last_rg;
  int k, it, status, done, mode, nrestart;
  int suspect;                // the objectives are drifting apart (relative gap >= 1/2 after 2048 iterations): certificate sequence wanted
};

#if defined(__HIPCC__)
// ---- control: the restart / termination decision of one scenario from its check sums (slot layout: k_check_rows / k_kkt_cols of dsp_stream.hip;
// shared with the lane-per-scenario form, dsp_stream_lane.hip)
// returns the mode for the apply step (0 = Halpern step, 1 = restart at (x+, y+)); sets c.done / c.status on termination
__device__ inline int control_decide(const double *acc, StreamCtrl &c, const dsp_options &o, double eta, int iters_this_period) {
  c.it += iters_this_period;
  c.k += iters_this_period;
  const double w = c.w, iw = 1.0 / w;
  const double r = fmax(w * acc[0] + iw * acc[1], 0.0);        // squared fixed-point residual in the PDHG metric
  int mode = 0;
  if (!(r == r)) { c.status = DSP_STATUS_NUMERICAL; c.done = 1; }
  else {
    const double po = acc[9] + acc[7], dobj = acc[4] + acc[10];      // acc[7]: quadratic terms of the soft rows (0 for an LP)
    c.pobj = po;
    const double rp = sqrt(acc[2]) / (1.0 + c.qn), rd = sqrt(acc[8]) / (1.0 + c.cn);
    const double gap = fabs(po - dobj);
    const double rg = gap / (1.0 + fabs(po) + fabs(dobj));
    bool fin;                                                    // same tests as the fused kernel (dsp_kernels.hip)
    if (o.eps_obj > 0.0) {
      const double lim = fmax(o.eps_obj * (1.0 + fabs(po + c.c0)), 1e-12 * acc[11]);
      fin = rp <= o.eps_rel && rd <= o.eps_rel && gap + acc[3] + acc[12] <= lim;
    } else {
      fin = rp <= o.eps_rel && rd <= o.eps_rel && rg <= o.eps_rel;
    }
    c.last_rp = rp; c.last_rd = rd; c.last_rg = rg;
    // An LP without a solution: no fixed point, one of the objectives runs away, the relative gap tends to 1.  A feasible year-long
    // LP passes 1/2 within its first 1.3 - 1.5 k iterations (lab, T = 672 / 1344); a scenario still above it after 2048 is SUSPECT:
    // the host runs the certificate sequence (stream_certify, dsp_stream.hip) on the displacement T(z) - z, at most every 16 checks.
    c.suspect = (o.eps_infeasible > 0.0 && rg >= 0.5 && c.it >= 2048) ? 1 : 0;
    if (fin) { c.status = DSP_STATUS_OPTIMAL; c.done = 1; }
    else if (c.it >= o.max_iter) { c.status = DSP_STATUS_ITERATION_LIMIT; c.done = 1; }
    else {
      const double bs2 = o.restart_sufficient * o.restart_sufficient, bn2 = o.restart_necessary * o.restart_necessary;
      const bool first = !(c.r0 < INFINITY);
      const bool decayed = (r <= bs2 * c.r0) || (r <= bn2 * c.r0 && r > c.rprev);
      const bool artificial = (double)c.k >= o.restart_artificial * (double)c.it;
      if (first) c.r0 = r;
      c.rprev = r;
      if (!first && (decayed || artificial)) {
        double wn = w;
        if (acc[6] > 1e-28 && acc[5] > 1e-28) {
          const double e = log(w) + 0.5 * (log(acc[6]) - log(acc[5]));
          const double dl = fmin(fmax(-o.pid_kp * e, -o.max_dlog_weight), o.max_dlog_weight);
          wn = w * exp(dl);
        }
        wn = fmin(fmax(wn, c.w_lo), fmax(c.w_hi, c.w_lo));
        c.w = wn; c.tau = eta / wn; c.sig = eta * wn;
        c.k = 0; c.r0 = INFINITY; c.rprev = INFINITY;
        c.nrestart += 1;
        mode = 1;
      }
    }
  }
  c.mode = mode;
  return mode;
}

#endif

struct StreamWork {
  double *x, *x0, *xp, *xbar, *c, *lb, *ub;     // [B][n]  scaled space
  double *y, *y0, *yp, *rlo, *rhi, *kap;        // [B][m]   (kap: scaled compliance of the soft rows, QP only)
  double *x2, *y2;                              // [B][n] / [B][m] second buffers of the fused iteration (x / y ping-pong)
  double *lpart[2];                             // [B][nlong][ntile] partial sums of A^T y for the long columns (ping-pong)
  StreamCtrl *ctrl;                             // [B]
  double *partial;                              // [B][nblk_tot][16] ordered block partial sums
  double *long_partial;                         // [B][nchunk_max] chunk partials of the long vectors
  int *ndone;                                   // [2]: scenarios finished; flag "some scenario is suspect" (set by the decision kernels)
  double *ray;                                  // [B][m] scratch of the certificate sequence (the cleaned dual displacement)
};

struct StreamArgs {
  StreamProblem P;
  StreamWork W;
  dsp_batch b;
  dsp_options opt;
  double eta;
  int nblk_n;      // blocks over the columns
  int nblk;        // blocks over max(n, m) elements
  int nblk_tot;    // + blocks of the long vectors (partial-sum stride)
  int nchunk_max;  // stride of long_partial
};

struct LaneState;                      // lane-per-scenario form (dsp_stream_lane.hip); nullptr = not applicable to this matrix
struct IpmState;                       // interior-point form for time-banded LPs (dsp_ipm.hip); nullptr = not applicable

struct StreamSolver {
  StreamProblem P{};
  StreamWork W{};
  int work_B = 0;
  std::vector<void *> allocs, work_allocs;
  int *ndone_host = nullptr;
  LaneState *lane = nullptr;
  IpmState *ipm = nullptr;
  int last_newton = 0;                   // interior-point form: Newton iterations of the last solve
  int last_ipm_solved = 0;               // ... and the scenarios it solved (the others went on to the PDHG forms)
  int last_form = 0;                     // DSP_STREAM_FORM_* of the last solve
  int last_phases = 0;                   // lane form: phases of the last solve (dsp_stats::stream_phases)
  size_t last_bytes_per_iteration = 0;   // algorithmic HBM bytes per scenario and plain iteration of the form the last solve ran
  std::mutex mu;          // one solve at a time per handle: the workspace above is per handle, not per call (stream_solve)
  size_t lds_limit = 160 * 1024 - 2048;   // dynamic LDS available to the block-resident form (static __shared__ on top)
};

hipError_t stream_create(const HostCSR &A_scaled, const HostCSR &AT_scaled, const double *col_scale_dev,
                         const double *row_scale_dev, StreamSolver *S);
void stream_destroy(StreamSolver *S);
hipError_t stream_solve(StreamSolver *S, const dsp_batch &batch, const dsp_options &opt, double eta, hipStream_t st,
                        int *periods_run);
size_t stream_bytes_per_iteration(const StreamSolver *S);

// dsp_stream_lane.hip
hipError_t lane_create(const HostCSR &A_scaled, const HostCSR &AT_scaled, StreamSolver *S);
void lane_destroy(StreamSolver *S);
hipError_t lane_run(StreamSolver *S, StreamArgs &a, hipStream_t st, int *periods_run, bool *used, const std::vector<int> *only = nullptr);
// dsp_ipm.hip: scenarios it solves get ctrl.done = 1 / status 0 and their x+ / y+ in the scenario-major workspace; the others are untouched
hipError_t ipm_create(const HostCSR &A_scaled, const HostCSR &AT_scaled, StreamSolver *S);
void ipm_destroy(StreamSolver *S);
hipError_t ipm_run(StreamSolver *S, StreamArgs &a, hipStream_t st, bool *all_solved, int *newton, int *n_solved);
int ipm_partitions(const StreamSolver *S);     // time partitions of its banded solves (1: sequential walks; 0: no interior-point plan)
// certificate sequence on the scenario-major workspace (x, y = the current iterate): dsp_stream.hip
hipError_t stream_certify(StreamSolver *S, StreamArgs &a, hipStream_t st);

}  // namespace dsp
