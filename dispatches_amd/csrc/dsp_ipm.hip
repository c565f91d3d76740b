// dsp_ipm.hip — interior-point form of the HBM-resident path for TIME-BANDED LPs, gfx950 only.
//
// The year-long price-taker LPs (reference: dispatches/case_studies/renewables_case/wind_battery_LMP.py:172-269 and its sweeps,
// run_pricetaker_wind_PEM.py:106-107) cost the restarted PDHG of dsp_stream*.hip 75 k iterations on average (139 k at worst): the
// state-of-charge / throughput chains are integrators and the design column couples all T periods, and a diagonally preconditioned
// first-order method sees the horizon one period per iteration.  Their structure is what a DIRECT method wants: in the
// natural (period-major) row order the normal matrix  A Theta A'  of all columns but the design column(s) has half-bandwidth 6 - 8 with
// m = 6 T rows.  This file is a primal-dual interior-point method (Mehrotra predictor-corrector; lab: tools/ipm_lab.py - 40 - 144
// Newton iterations on the 16-member year-long fixture, objectives to 2e-7) whose Newton systems are solved exactly:
//     [A | -I] (x, r) = 0,  lb <= x <= ub,  rlo <= r <= rhi        (one slack per row: the structure does not depend on the batch's bounds;
//                                                                     fixed columns / equality rows simply carry Theta = 0)
//     (A_s Theta_s A_s' + Theta_r) dy = rhs    banded LDL' (no pivoting; relative diagonal regularisation 1e-12), half-bandwidth W <= 8
//     wide columns (span > 16 rows: design variables, periodic conditions; K <= 4) through Sherman-Morrison-Woodbury, K x K per scenario
//     iterative refinement on the full normal equations where the solve leaves more than 1e-6 of the right-hand side (max norms; it rarely
//     does), three steps at most.  A step after which the primal residual JUMPS (1000 x its best so far and past the tolerance: the solve
//     error of a near-unit step while mu collapses) is TAKEN BACK (k_ipm_undo: the directions are still in place) and the scenario goes on
//     with its systems refined to 1e-8 / 1e-11 (far out / in its end game: round 5's tolerances for everybody, 2.4 extra solves per Newton
//     iteration; profiles/r70*_reftol.log)
//     steps 0.99 to the boundary (0.9 where the boundary is closer than half the Newton step: round 6, the slow members' hundred blocked
//     steps get fewer), sigma = max((mu_aff / mu)^3, 0.05): the settings under which every horizon, family
//     and elimination order tried finishes (profiles/HISTORY.md "Robustness, measured": mu must not collapse under the solve error)
// ONE LANE PER SCENARIO: every array is scenario-minor ([index][scenario], 64 scenarios = one 512-byte line per index), all lanes walk the
// same rows, and the structure (ELLs of the scaled matrix, the band's product lists) is read through uniform addresses.
// The banded factorisation and solves are chains of dependent rows.  SEQUENTIAL form (below 512 rows; DSP_IPM_PARTS=1): one wave per 64
// scenarios walks all m rows out of LDS while the workgroup's other waves stream chunks of rows HBM -> registers -> LDS -> HBM around it
// (k_seq), so that the chain never waits for memory: 58 / 69 / 158 ns per row forward / backward / factor - 5 - 9 ms per walk at T = 8736
// whatever the batch.  TIME-PARALLEL form (default; dsp_ipm_seq.hpp): the rows are cut into up to 64 partitions whose interiors are
// eliminated at once (k_seq with grid.y = partitions), the W columns coupling an interior to the separator on its left are carried as
// spikes, the separators form a block-tridiagonal system solved per lane (k_ipm_red_factor / k_ipm_red_solve), border sums and corrections
// (k_ipm_border_dot / k_ipm_border_apply) connect the two: a solve is then five bandwidth-bound kernels (3.5 GB at 256 scenarios, 0.59 of
// the HBM peak) instead of two latency chains.  Everything else (residuals, Theta, assembly of the band, directions, step lengths, KKT
// test) is elementwise over (chunk of indices) x (64 scenarios).  256 year-long scenarios: 13 ms per Newton iteration and ~50 GB of HBM traffic
// in round 5; 4.6 ms and ~18 GB at the end of round 6 (lane packing, refinement on demand, fused elementwise passes: DESIGN.md 4f).
//
// Termination is the HBM-resident path's own test (control_decide, dsp_stream.hpp) evaluated on the unscaled problem; a scenario the
// method does not finish (breakdown, 250 Newton iterations, free columns) is left to the PDHG forms - that scenario alone: the ones this
// file solved are exported and stay solved (round 6; round 5 re-ran the whole batch).  dsp_options::no_interior_point = 1 switches it off; dsp_stats::stream_form = DSP_STREAM_FORM_IPM when it solved the batch,
// dsp_stats::stream_phases = the time partitions.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "dsp_ipm_seq.hpp"
#include "dsp_stream.hpp"

namespace dsp {

constexpr int kIpmMaxW = 8, kIpmMaxK = 4, kIpmSpan = 16, kIpmMaxNewton = 250;
constexpr int kIpmNQ = 16;                 // partial-sum slots (the Woodbury matrix needs K * K)
constexpr int kIpmMaxBlocks = 1024;        // blocks (of 4 waves) per lane group in the elementwise kernels, at most

// Blocks per lane group of the elementwise kernels for G groups in use: enough waves to hide the gathers' latency (a wave owns a
// chunk of indices and walks it with a few loads in flight) - 2048 blocks over the groups, at most kIpmMaxBlocks per group.  Round 5
// stopped at 256: 1024 waves for a single group of 64 scenarios = one wave per SIMD, 7.6 ms per Newton iteration at 64 scenarios
// against 13.1 ms at 256.  The partial sums are per block (ipm_put), so the finish kernels walk as many as before.
static inline int ipm_blocks(int G) {
  static const int cap = getenv("DSP_IPM_MAXBLK") ? std::min(std::max(atoi(getenv("DSP_IPM_MAXBLK")), 16), kIpmMaxBlocks) : kIpmMaxBlocks;      // (development)
  static const int tot = getenv("DSP_IPM_TOTBLK") ? atoi(getenv("DSP_IPM_TOTBLK")) : 2048;
  return std::max(16, std::min(cap, tot / std::max(G, 1)));
}

struct IpmPlan {                            // shared by all scenarios (device)
  int n, m, W, K, Mp;                       // Mp = m + W rows of the band arrays
  // the scaled matrix as two ELLs (fixed widths: the address of an entry follows from the index alone - with CSR every product waits
  // for its pointer, then its index, then its operand: three memory round trips in a row); padding: index 0, value 0
  int rw, cw, bw;                           // entries per row (all columns), per NARROW column, products per band entry
  const int32_t *rcol; const double *rval;  // [m][rw]
  const int32_t *crow; const double *cval;  // [n][cw]   (wide columns: empty here)
  const uint8_t *wide;                      // [n] 0 = narrow, k + 1 = wide column k
  int32_t wide_col[kIpmMaxK];
  const int32_t *wptr, *wrow; const double *wval;      // the wide columns' entries: CSR over k
  const int32_t *bcol; const double *bval;  // [m][W + 1][bw]: B(t, t - k) = sum bval * theta[bcol] (narrow columns)
  const double *col_scale, *row_scale;
  IpmParts parts;                           // the time-parallel form's geometry (P = 1: sequential walks; dsp_ipm_seq.hpp)
};

// per-lane scalars
enum { SC_MU = 0, SC_SIGMU, SC_AP, SC_AD, SC_CS, SC_NB, SC_APA, SC_ADA, SC_POBJ, SC_RPMIN, SC_STRICT, SC_UNDO, SC_COUNT };

struct IpmWork {
  size_t Bp;                                // lanes (batch rounded up to 64): the stride of every scenario-minor array
  int B, nch;                               // scenarios, chunks of the elementwise kernels (one per wave: 4 x the blocks of their grids)
  int G;                                    // groups of 64 lanes in use (Bp / 64 until lanes are packed: ipm_repack)
  int *sid;                                 // [Bp] lane -> scenario (identity until lanes are packed)
  int *perm;                                // [Bp] ipm_repack: new lane -> old lane
  double *v, *z, *f, *l, *u, *cb, *th, *rd, *dv, *dz, *df, *rt, *corl, *coru, *tn;      // [n + m][Bp]
  double *y, *dy, *rp, *rhs, *q, *res;      // [m][Bp]
  double *biad;                             // [K][m][Bp]
  double *band;                             // [W + 1][Mp][Bp]
  double *gs;                               // [W][Mp][Bp]  spikes (time-parallel form)
  double *sfin, *cfin;                      // [P][W (W + 1) / 2][Bp]  final windows / Schur updates of the partitions
  double *redf;                             // [P][W W + W (W + 1) / 2][Bp]  the reduced system's block factor
  double *bd;                               // [P][W][Bp]  border sums of a solve
  double *sc;                               // [SC_COUNT][Bp]
  double *part;                             // [kIpmNQ][nch / 4][Bp]: one partial per BLOCK of the elementwise kernels (ipm_put)
  double *sinv, *tk;                        // [K * K][Bp], [K][Bp]
  double *wat;                              // [K][Bp] (Abar' vec) of the wide columns (k_ipm_wide_aty)
  int *state;                               // [Bp] 0 = iterating, 1 = solved, 2 = given up
  int *iters;                               // [Bp]
  int *stall;                               // [Bp] consecutive Newton iterations with a step below 1e-4
  int *endg;                                // [Bp] 1 = the scenario is in its end game (k_ipm_decide): its Newton systems are refined further
  int *counts;                              // [8]: finished (solved or given up), solved, lanes in the end game, (refinement flag), steps to take back
  int ls;                                   // the walks and the reduced solve split a lane group over this many workgroups (1, 2, 4): a workgroup pulls
                                            //  ~35 - 50 GB/s from HBM whatever it asks for, and ONE lane group's walk is 64 of them on 256 CUs; the lanes of
                                            //  a wave then work in duplicate (64 / ls distinct scenarios: same addresses, same values - the loads coalesce)
  int bz;                                   // the border kernels split a partition's rows over this many workgroups (1 .. kIpmBorderSplit): few lane
                                            //  groups leave (P - 1) G workgroups on 256 CUs, each bound by what ONE CU pulls from HBM; bd holds bz partial sets
};

constexpr int kIpmBorderSplit = 4;
struct IpmArgs {
  IpmPlan P;
  IpmWork w;
  StreamWork sw;                            // scenario-major scaled inputs (k_init) and outputs (k_finalize reads xp / yp / ctrl)
  dsp_options opt;
  int it;                                   // Newton iteration (1-based)
  int max_it;                               // give up after this many (kIpmMaxNewton; development: DSP_IPM_MAXIT)
  double reftol, reftol_end;                // refinement of the Newton systems: |rhs - N dy| <= tol |rhs| (max norms), far out / in a scenario's end game
  double step_blocked, step_thr;            // (k_ipm_steps)
  double reg, step, sigmin;                 // primal regularisation of Theta (0: off), step to the boundary (0.99), floor of sigma (0.05); ipm_run
  double thcap;                             // cap of Theta (k_ipm_resid)
  double reftol_strict, reftol_strict_end;  // the tolerances of a scenario whose step was taken back (k_ipm_decide: SC_STRICT)
  int max_undo;                             // steps a scenario may take back before it is given up warm (state 5)
};

struct IpmState {
  IpmPlan P{};
  IpmWork w{};
  std::vector<void *> allocs, work_allocs;
  int work_B = 0;
  int *counts_host = nullptr;
  int *perm_host = nullptr;                 // pinned, work_B rounded up to 64 ints (ipm_repack)
  int last_repacks = 0;
};

__device__ __forceinline__ bool ipm_fin(double v) { return fabs(v) < INFINITY; }
__device__ __forceinline__ double ipm_fin0(double v) { return fabs(v) < INFINITY ? v : 0.0; }

#define IPM_LANE()                                                        \
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;             \
  const size_t Bp = a.w.Bp;                                               \
  const size_t s = (size_t)blockIdx.y * 64 + lane;                        \
  const int cid = blockIdx.x * 4 + wv;                                    \
  (void)cid;                                                              \
  if (a.w.state[s] != 0) return

// chunk [i0, i1) of `count` indices for this wave
__device__ __forceinline__ void ipm_chunk(int count, int nch, int cid, int &i0, int &i1) {
  const int per = (count + nch - 1) / nch;
  i0 = min(cid * per, count); i1 = min(i0 + per, count);
}

// partial q of this block: the four waves' values combined in LDS, one value per (block, lane) in part[q][block][lane] - a quarter of the
// partials per wave that the finish kernels would otherwise walk (they are launched one workgroup per 64 scenarios).  Called by every
// thread of the block that is still running (lanes of finished scenarios have left: on this hardware a barrier waits for the waves that
// exist, and a lane is finished in all four waves or in none).
template <int OP>                          // OP 0: sum, 1: min, 2: max
__device__ __forceinline__ void ipm_put(const IpmArgs &a, int q, double val, size_t s) {
  __shared__ double sh[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();                         // (the previous partial's readers are done with sh)
  sh[wv][lane] = val;
  __syncthreads();
  if (wv == 0) {
    auto op = [](double x, double y) { return OP == 0 ? x + y : OP == 1 ? fmin(x, y) : fmax(x, y); };
    a.w.part[((size_t)q * (a.w.nch / 4) + blockIdx.x) * a.w.Bp + s] = op(op(sh[0][lane], sh[1][lane]), op(sh[2][lane], sh[3][lane]));
  }
}

// sum (or minimum / maximum) over the blocks' partials part[q][block][lane], q < NQ, by a workgroup of 256 threads = 4 waves x the 64
// scenarios of the group: every wave takes a quarter of the chunks with four independent accumulators (the loads of a plain loop queue
// up one memory latency each: a 512-chunk sum took 0.95 ms), LDS combines.  Every thread returns the result.
template <int NQ, int OP, int NW = 4>      // OP 0: sum, 1: min, 2: max; NW waves per workgroup (8 where NQ <= 8: with up to 1024 partials per lane
__device__ __forceinline__ void ipm_finish(const double *part, int nch, size_t Bp, size_t s, double (&out)[NQ], int nq = NQ) {   // the walk is the kernel)
  __shared__ double red[NW][NQ][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto op = [](double x, double y) { return OP == 0 ? x + y : OP == 1 ? fmin(x, y) : fmax(x, y); };
  const double e0 = OP == 0 ? 0.0 : OP == 1 ? 1e300 : -1e300;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    constexpr int U = 16;                                  // loads in flight per thread (4: 128 partials per wave = 32 memory round trips per quantity)
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = e0;
    if (q >= nq) { red[wv][q][lane] = e0; continue; }      // (slots nobody filled: K = 1 uses 1 of the Woodbury matrix's 16)
    int c = wv;
    for (; c + (U - 1) * NW < nch; c += U * NW) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = op(acc[u], part[((size_t)q * nch + c + NW * u) * Bp + s]);
    }
    for (; c < nch; c += NW) acc[0] = op(acc[0], part[((size_t)q * nch + c) * Bp + s]);
#pragma unroll
    for (int w = U / 2; w >= 1; w /= 2)
#pragma unroll
      for (int u = 0; u < w; ++u) acc[u] = op(acc[u], acc[u + w]);
    red[wv][q][lane] = acc[0];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double t = red[0][q][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = op(t, red[w][q][lane]);
    out[q] = t;
  }
}
constexpr int kFinW = 8;                   // waves of the finish kernels (all but the Woodbury matrix's, whose 16 slots x 8 waves would not fit)

// ---- setup -------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ipm_cmax(IpmArgs a) {
  IPM_LANE();
  int j0, j1; ipm_chunk(a.P.n, a.w.nch, cid, j0, j1);
  double mx = 0.0;
  const double *c = a.sw.c + s * a.P.n;
  for (int j = j0; j < j1; ++j) mx = fmax(mx, fabs(c[j]));
  ipm_put<2>(a, 0, mx, s);
}

__global__ __launch_bounds__(64 * kFinW) void k_ipm_cmax_finish(IpmArgs a) {
  const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63), Bp = a.w.Bp;
  double mx[1];
  ipm_finish<1, 2, kFinW>(a.w.part, a.w.nch / 4, Bp, s, mx);
  if (threadIdx.x < 64 && a.w.state[s] == 0) { a.w.sc[SC_CS * Bp + s] = mx[0] > 1e-300 ? mx[0] : 1.0; a.w.sc[SC_RPMIN * Bp + s] = 1e300; a.w.sc[SC_STRICT * Bp + s] = 0.0; a.w.sc[SC_UNDO * Bp + s] = 0.0; a.w.sc[SC_AP * Bp + s] = 1.0; a.w.sc[SC_AD * Bp + s] = 1.0; }
}

__global__ __launch_bounds__(256) void k_ipm_setup(IpmArgs a) {
  IPM_LANE();
  const int n = a.P.n, m = a.P.m, N = n + m;
  int j0, j1; ipm_chunk(N, a.w.nch, cid, j0, j1);
  const double ics = 1.0 / a.w.sc[SC_CS * Bp + s];
  bool bad = false;
  for (int j = j0; j < j1; ++j) {
    double l, u, c;
    if (j < n) { l = a.sw.lb[s * n + j]; u = a.sw.ub[s * n + j]; c = a.sw.c[s * n + j] * ics; }
    else { l = a.sw.rlo[s * m + (j - n)]; u = a.sw.rhi[s * m + (j - n)]; c = 0.0; }
    const bool hl = ipm_fin(l), hu = ipm_fin(u);
    double v, z = 0.0, f = 0.0;
    if (hl && hu && l == u) v = l;                                     // fixed: takes no part (Theta = 0)
    else if (hl && hu) { v = 0.5 * (l + u); z = 1.0; f = 1.0; }
    else if (hl) { v = l + 1.0; z = 1.0; }
    else if (hu) { v = u - 1.0; f = 1.0; }
    else { v = 0.0; if (j < n) bad = true; }                           // free column: not this method's LP (a free ROW: slack with a huge Theta)
    const size_t at = (size_t)j * Bp + s;
    a.w.l[at] = l; a.w.u[at] = u; a.w.cb[at] = c; a.w.v[at] = v; a.w.z[at] = z; a.w.f[at] = f;
  }
  int i0, i1; ipm_chunk(m, a.w.nch, cid, i0, i1);
  for (int i = i0; i < i1; ++i) a.w.y[(size_t)i * Bp + s] = 0.0;
  if (bad) a.w.state[s] = 3;                                           // (3: marked here, counted by k_ipm_begin)
}

// ---- residuals, Theta, mu ----------------------------------------------------------------------------------------------------------
// (Abar' y)_j : narrow structural column j -> sum_i a_ij y_i ; wide column k -> wat[k] (k_ipm_wide_aty ran on y) ; slack of row i -> -y_i
__device__ __forceinline__ double ipm_aty(const IpmPlan &P, const double *y, const double *wat, int j, size_t Bp, size_t s) {
  if (j >= P.n) return -y[(size_t)(j - P.n) * Bp + s];
  const int wk = P.wide[j];
  if (wk) return wat[(size_t)(wk - 1) * Bp + s];
  double t = 0.0;
  const int32_t *ci = P.crow + (size_t)j * P.cw;
  const double *cv = P.cval + (size_t)j * P.cw;
  for (int e = 0; e < P.cw; e += 4) {                 // (widths are multiples of 4: four gathers in flight - one at a time, every entry
    const double y0 = y[(size_t)ci[e] * Bp + s], y1 = y[(size_t)ci[e + 1] * Bp + s];      //  of the ELL cost a round trip to L2)
    const double y2 = y[(size_t)ci[e + 2] * Bp + s], y3 = y[(size_t)ci[e + 3] * Bp + s];
    t = fma(cv[e], y0, t); t = fma(cv[e + 1], y1, t); t = fma(cv[e + 2], y2, t); t = fma(cv[e + 3], y3, t);
  }
  return t;
}
// (Abar u)_i = sum_j a_ij u_j - u_{n+i}
__device__ __forceinline__ double ipm_au(const IpmPlan &P, const double *u, int i, size_t Bp, size_t s) {
  double t = -u[(size_t)(P.n + i) * Bp + s];
  const int32_t *ri = P.rcol + (size_t)i * P.rw;
  const double *rv = P.rval + (size_t)i * P.rw;
  for (int e = 0; e < P.rw; e += 4) {
    const double u0 = u[(size_t)ri[e] * Bp + s], u1 = u[(size_t)ri[e + 1] * Bp + s];
    const double u2 = u[(size_t)ri[e + 2] * Bp + s], u3 = u[(size_t)ri[e + 3] * Bp + s];
    t = fma(rv[e], u0, t); t = fma(rv[e + 1], u1, t); t = fma(rv[e + 2], u2, t); t = fma(rv[e + 3], u3, t);
  }
  return t;
}

// (Abar' vec)_j of the wide columns: partial sums over slices of their entries, finished per scenario into wat[k]
__global__ __launch_bounds__(256) void k_ipm_wide_aty(IpmArgs a, const double *vec) {
  IPM_LANE();
  for (int k = 0; k < a.P.K; ++k) {
    int p0, p1; ipm_chunk(a.P.wptr[k + 1] - a.P.wptr[k], a.w.nch, cid, p0, p1);
    p0 += a.P.wptr[k]; p1 += a.P.wptr[k];
    double t = 0.0;
#pragma unroll 4
    for (int p = p0; p < p1; ++p) t = fma(a.P.wval[p], vec[(size_t)a.P.wrow[p] * Bp + s], t);
    ipm_put<0>(a, k, t, s);
  }
}
__global__ __launch_bounds__(64 * kFinW) void k_ipm_wide_aty_finish(IpmArgs a) {
  const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63), Bp = a.w.Bp;
  double t[kIpmMaxK];
  ipm_finish<kIpmMaxK, 0, kFinW>(a.w.part, a.w.nch / 4, Bp, s, t, a.P.K);
  if (threadIdx.x >= 64 || a.w.state[s] != 0) return;
#pragma unroll
  for (int k = 0; k < kIpmMaxK; ++k) if (k < a.P.K) a.w.wat[(size_t)k * Bp + s] = t[k];
}

__global__ __launch_bounds__(256) void k_ipm_resid(IpmArgs a) {
  IPM_LANE();
  const int n = a.P.n, m = a.P.m, N = n + m;
  int j0, j1; ipm_chunk(N, a.w.nch, cid, j0, j1);
  double comp = 0.0, cnt = 0.0;
  // q[0 .. 7]: the sums of the HBM-resident path's KKT test on the unscaled problem (control_decide, dsp_stream.hpp) for THIS iterate - the
  // one the previous iteration's update left; k_ipm_decide finishes them.  They ride on this kernel's two gathers (A' y, A v): the separate
  // pass that took them right after the update (k_ipm_check, until round 6) read the same arrays once more.
  double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const double cs = a.w.sc[SC_CS * Bp + s];
#pragma unroll 2
  for (int j = j0; j < j1; ++j) {
    const size_t at = (size_t)j * Bp + s;
    const double l = a.w.l[at], u = a.w.u[at], v = a.w.v[at], z = a.w.z[at], f = a.w.f[at];
    const bool hl = ipm_fin(l), hu = ipm_fin(u), fixed = hl && hu && l == u;
    const double cb = a.w.cb[at], aty = ipm_aty(a.P, a.w.y, a.w.wat, j, Bp, s);
    const double rd = cb - aty - z + f;
    a.w.rd[at] = rd;
    if (j < n) {
      const double rc = cb - aty;
      const double lp = hl ? fmax(rc, 0.0) : 0.0, lm = hu ? fmax(-rc, 0.0) : 0.0;
      const double dres = rc - lp + lm;
      const double du = dres * cs / a.P.col_scale[j];
      q[3] += du * du;                                                              // ||dual residual||^2, unscaled
      q[4] += fabs(dres) * fabs(v);                                                 // (x cs)
      q[5] += cb * v;                                                               // primal objective      (x cs)
      q[6] += fabs(cb * v);
      q[7] += lp * ipm_fin0(l) - lm * ipm_fin0(u);                                  // dual objective, bounds (x cs)
    }
    double th = 0.0;
    if (!fixed) {
      double den = 0.0;
      if (hl) { const double wl = v - l; den += z / wl; comp += z * wl; cnt += 1.0; }
      if (hu) { const double tu = u - v; den += f / tu; comp += f * tu; cnt += 1.0; }
      th = (hl || hu) ? 1.0 / fmax(den + a.reg, 1e-300) : 1e20;               // (free slack = free row: dropped by a huge Theta)
      // the cap (a.thcap, 1e11 in the scaled space): a BASIC column's Theta = distance / dual reaches 1e17 .. 1e19 at the mu the objective
      // tolerance asks of LPs with a small objective (mu ~ lim / #bounds = 2e-9), the pivot of a row that shares such a column with its
      // neighbour becomes the difference of two numbers of that size, and the step leaves a primal residual no later step removes
      // (k_ipm_decide: `polluted`).  Capped, those columns carry a primal regularisation of 1 / cap: all 10 of the 256 distinct year-long
      // members that ran into the iteration limit finish (62 - 115 Newton iterations), the others take the iterations they took
      // (lab: tools/ipm_lab.py theta_cap=1e11, profiles/r61c_theta_cap_lab.log; 1e12 and above: failures remain, 1e10: 2 x the iterations)
      if (hl || hu) th = fmin(th, a.thcap);
      th = fmin(th, 1e30);
    }
    a.w.th[at] = th;
    // the predictor's rt / tn (k_ipm_rt's mode 0: targets cz = -z, cf = -f) while everything is in registers: seven array reads less
    const double cz = (hl && th > 0.0) ? 0.0 / (v - l) - z : 0.0, cf = (hu && th > 0.0) ? 0.0 / (u - v) - f : 0.0;
    const double rt = rd - cz + cf;
    a.w.rt[at] = rt;
    a.w.tn[at] = th * rt;
  }
  int i0, i1; ipm_chunk(m, a.w.nch, cid, i0, i1);
  for (int i = i0; i < i1; ++i) {
    const size_t ar = (size_t)(n + i) * Bp + s;
    const double au = ipm_au(a.P, a.w.v, i, Bp, s);
    a.w.rp[(size_t)i * Bp + s] = -au;
    const double rlo = a.w.l[ar], rhi = a.w.u[ar], y = a.w.y[(size_t)i * Bp + s];
    const double ax = au + a.w.v[ar];                                               // (A_s x)_i
    const double viol = fmax(rlo - ax, 0.0) + fmax(ax - rhi, 0.0);
    const double vu = viol / a.P.row_scale[i];
    q[0] += vu * vu;                                                                // ||row violation||^2, unscaled
    q[1] += fabs(y) * viol;                                                         // |y| . violation       (x cs)
    q[2] += fmax(y, 0.0) * ipm_fin0(rlo) - fmax(-y, 0.0) * ipm_fin0(rhi);            // dual objective, rows  (x cs)
  }
  ipm_put<0>(a, 0, comp, s);
  ipm_put<0>(a, 1, cnt, s);
#pragma unroll
  for (int k = 0; k < 8; ++k) ipm_put<0>(a, 2 + k, q[k], s);
}

__global__ __launch_bounds__(64 * kFinW) void k_ipm_mu(IpmArgs a) {
  const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63), Bp = a.w.Bp;
  double t[2];
  ipm_finish<2, 0, kFinW>(a.w.part, a.w.nch / 4, Bp, s, t);
  if (threadIdx.x < 64 && a.w.state[s] == 0) {
    a.w.sc[SC_MU * Bp + s] = t[0] / fmax(t[1], 1.0);
    a.w.sc[SC_NB * Bp + s] = t[1];
  }
}

// ---- band assembly -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ipm_assemble(IpmArgs a) {
  IPM_LANE();
  const int W1 = a.P.W + 1, m = a.P.m, Mp = a.P.Mp;
  const IpmParts g = a.P.parts;
  const bool par = g.P > 1;
  int t0, t1; ipm_chunk(Mp, a.w.nch, cid, t0, t1);
  for (int t = t0; t < t1; ++t)
    for (int k = 0; k < W1; ++k) {
      double val = 0.0;
      if (t < m) {
        const size_t e0 = ((size_t)t * W1 + k) * a.P.bw;
        for (int p = 0; p < a.P.bw; p += 4) {
          const double h0 = a.w.th[(size_t)a.P.bcol[e0 + p] * Bp + s], h1 = a.w.th[(size_t)a.P.bcol[e0 + p + 1] * Bp + s];
          const double h2 = a.w.th[(size_t)a.P.bcol[e0 + p + 2] * Bp + s], h3 = a.w.th[(size_t)a.P.bcol[e0 + p + 3] * Bp + s];
          val = fma(a.P.bval[e0 + p], h0, val); val = fma(a.P.bval[e0 + p + 1], h1, val);
          val = fma(a.P.bval[e0 + p + 2], h2, val); val = fma(a.P.bval[e0 + p + 3], h3, val);
        }
        if (k == 0) { val += a.w.th[(size_t)(a.P.n + t) * Bp + s]; val *= 1.0 + 1e-12; if (!(val > 0.0)) val = 1.0; }
        else if (par && ipm_diverted(g, t, k)) {      // the column lies before this row's partition: a spike's entry (dsp_ipm_seq.hpp)
          const int j = t - k - (g.start(g.part_of(t)) - g.W);
          a.w.gs[((size_t)j * Mp + t) * Bp + s] = val;
          val = 0.0;
        }
      } else if (k == 0) val = 1.0;
      a.w.band[((size_t)k * Mp + t) * Bp + s] = val;
    }
  if (par)                                             // the spikes a head row has no band entry for
    for (int t = t0; t < min(t1, m); ++t) {
      const int p = g.part_of(t), i = t - g.start(p);
      if (p < 1 || i >= g.W) continue;
      for (int j = 0; j < i; ++j) a.w.gs[((size_t)j * Mp + t) * Bp + s] = 0.0;
    }
}

// ---- the sequential kernels ----------------------------------------------------------------------------------------------------------
// NS streams of `rows` rows each ([row][lane], `Bp` doubles between rows) are walked in order (or in reverse) by wave 0 of the workgroup,
// R rows at a time out of LDS; all four waves move the next chunk HBM -> registers while it computes, then registers -> LDS and the
// finished chunk's output streams LDS -> HBM.  Body::step(state, row, lr): row[q * 64] is stream q of the current row for this lane.
template <int NS>
struct SeqArgs {
  double *band;                            // streams 0 .. nband-1: band + q * stride  (no pointer table: a table indexed per thread is a load
  size_t stride;                           //  from the kernel-argument segment in front of every data load, and serialises them)
  int nband;
  double *x;                               // streams nband .. NS-1: x + (q - nband) * xstride (the one vector of a solve; the spikes)
  size_t xstride;
  int rows, last_rows;                     // rows of a partition's walk: every partition but the last / the last one
  int nparts, part0;                       // partitions of the walk (1: the sequential form); this workgroup's = blockIdx.y + part0,
  int reverse;                             //  its first row = partition * rows
  size_t Bp;
  unsigned outmask;
  const int *state;                        // groups whose 64 lanes have all finished are skipped
  int lsplit;                              // (IpmWork::ls) grid.x = groups x lsplit
  int dev_mode;                            // development (DSP_IPM_SEQ_MODE): 1 = no compute, 2 = no data movement after the first chunk
};

// wave 0 computes; the other WAVES - 1 waves move, as ONE set (every mover serves every chunk) or as TWO (the lower half serves the even
// chunks, the upper half the odd ones).  While chunk c is computed in one LDS buffer, the set whose turn it is (1) reads chunk c - 1's
// results out of the other buffer into registers, (2) refills that buffer with chunk c + 1, which it requested one turn earlier - the
// ONLY vector-memory operations those waves have outstanding at that point, so the wait the compiler places there is exact (the
// counters are per wave) -, (3) stores chunk c - 1 and (4) requests the chunk of its next turn.  One barrier per chunk.  With one
// set a load has one chunk's compute time to arrive, with two sets two: HBM answers in ~2.7 us, a chunk of a triangular solve is
// computed in 1.3 us (measured, m = 4034: compute alone 58 / 69 / 158 ns per row forward / backward / factor; one set 137 / 137 /
// 161).  Two sets need 15 movers to keep a thread's share of a chunk in registers - 16 waves, 128 VGPRs per thread: enough for
// the solves, not for the factorisation's 7 x 7 window, which keeps 8 waves and one set (its chunk takes 3 us to compute anyway).
// Time-parallel form: grid.y = the partitions of the walk, each a walk of its own over its rows (fresh Body state, Body::finish at its end).
template <int NS, int R, class Body, int WAVES, int SETS>
__global__ __launch_bounds__(64 * WAVES) void k_seq(SeqArgs<NS> a, Body body) {
  extern __shared__ double lds[];          // [2][R][NS][64]
  const int t = threadIdx.x, lane = t & 63, w4 = t >> 6;
  // lane split (SeqArgs::lsplit = ls > 1): this workgroup walks lw = 64 / ls scenarios of its group; a mover's lanes are ls ROW slots x lw
  // scenarios - one load instruction carries ls rows of a stream (a wave-wide load costs the CU's address path ~20 cycles whatever its
  // lanes ask for, and at one lane group the walks were bound by exactly that: 98 / 113 us with the arithmetic switched off, 59 / 69 us
  // with the data movement switched off, profiles/r68f) -, the computing wave's lanes work in duplicate (lane -> scenario lane % lw)
  const int ls = a.lsplit, lw = 64 / ls, sc = lane & (lw - 1), el = lane / lw;
  const size_t g0 = (size_t)((int)blockIdx.x / ls) * 64 + ((int)blockIdx.x % ls) * lw + sc;
  {
    __shared__ int any;
    if (t == 0) any = 0;
    __syncthreads();
    if (t < 64 && a.state[g0] == 0) any = 1;
    __syncthreads();
    if (!any) return;
  }
  const int part = (int)blockIdx.y + a.part0;
  const int rows = part == a.nparts - 1 ? a.last_rows : a.rows;
  const size_t row0 = (size_t)part * a.rows;
  constexpr int MOV = WAVES - 1, N0 = SETS == 1 ? MOV : MOV / 2, N1 = MOV - N0;
  constexpr int CH = R * NS, RP = (R + N0 - 1) / N0, E = RP * NS;        // a mover serves the rows r = mi, mi + nset, ... of a chunk, all streams
  const int nch = (rows + R - 1) / R;
  const int mv = w4 - 1;                   // mover index (wave 0: -1)
  const int sx = (SETS == 2 && mv >= N0) ? 1 : 0, nset = (SETS == 1 ? MOV : (sx ? N1 : N0)) * ls, mi = (sx ? mv - N0 : mv) * ls + el;
  const bool packed = ls > 1;              // (rows beyond the chunk are then SKIPPED - whole instructions for most movers -, not clipped duplicates)
  auto phys = [&](int lr) { return row0 + (size_t)(a.reverse ? rows - 1 - lr : lr); };
#define SEQ_PTR(q) ((q) < a.nband ? a.band + (size_t)(q) * a.stride : a.x + (size_t)((q) - a.nband) * a.xstride)
#define SEQ_LOAD(c)                                                                                                    \
  _Pragma("unroll") for (int rr = 0; rr < RP; ++rr) {                                                                  \
    const int r = min(mi + nset * rr, R - 1), lr = min((c) * R + r, rows - 1);                                         \
    const size_t off = phys(lr) * a.Bp + g0;                                                                           \
    if (!packed || mi + nset * rr < R) {                                                                               \
      _Pragma("unroll") for (int q = 0; q < NS; ++q) reg[rr * NS + q] = SEQ_PTR(q)[off];                               \
    }                                                                                                                  \
  }
#define SEQ_PUT(buf)                                                                                                   \
  _Pragma("unroll") for (int rr = 0; rr < RP; ++rr) {                                                                  \
    const int r = mi + nset * rr;            /* (no clipped duplicates here: another wave may still be reading that row's results) */ \
    if (r < R) {                                                                                                       \
      _Pragma("unroll") for (int q = 0; q < NS; ++q) lds[((size_t)(buf) * CH + r * NS + q) * 64 + sc] = reg[rr * NS + q]; \
    }                                                                                                                  \
  }
#define SEQ_STORE(c, buf, SRC)                                                                                         \
  _Pragma("unroll") for (int rr = 0; rr < RP; ++rr) {                                                                  \
    const int r = mi + nset * rr, lr = (c) * R + r;                                                                    \
    if (r < R && lr < rows) {                                                                                          \
      const size_t off = phys(lr) * a.Bp + g0;                                                                         \
      _Pragma("unroll") for (int q = 0; q < NS; ++q)                                                                   \
        if ((a.outmask >> q) & 1u) SEQ_PTR(q)[off] = SRC;                                                              \
    }                                                                                                                  \
  }
  if (mv >= 0) {
    double reg[E];
    if (sx == 0) {
      SEQ_LOAD(0)
      SEQ_PUT(0)
      if (SETS < nch) { SEQ_LOAD(SETS) }                          // one set: chunk 1; two sets: chunk 2 (the odd set requests chunk 1)
    } else if (nch > 1) { SEQ_LOAD(1) }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      const int buf = c & 1;
      if (a.dev_mode != 2 && (SETS == 1 || ((c + 1) & 1) == sx)) {
        if (SETS == 1) {                     // one set: results through registers (the stores must not sit in front of the wait for the loads)
          double fl[E];
          if (c >= 1) {
#pragma unroll
            for (int rr = 0; rr < RP; ++rr)
#pragma unroll
              for (int q = 0; q < NS; ++q) fl[rr * NS + q] = lds[((size_t)(buf ^ 1) * CH + min(mi + nset * rr, R - 1) * NS + q) * 64 + sc];
          }
          if (c + 1 < nch) { SEQ_PUT(buf ^ 1) }
          if (c >= 1) { SEQ_STORE(c - 1, buf ^ 1, fl[rr * NS + q]) }
        } else {                             // two sets: a set has two chunks' time per turn - the stores may complete first (no second register array)
          if (c >= 1) { SEQ_STORE(c - 1, buf ^ 1, (lds[((size_t)(buf ^ 1) * CH + r * NS + q) * 64 + sc])) }
          if (c + 1 < nch) { SEQ_PUT(buf ^ 1) }
        }
        if (c + 1 + SETS < nch) { SEQ_LOAD(c + 1 + SETS) }
      }
      __syncthreads();
    }
    if (a.dev_mode != 2 && sx == 0) {
      const int c = nch - 1, buf = c & 1;
      SEQ_STORE(c, buf, (lds[((size_t)buf * CH + r * NS + q) * 64 + sc]))
    }
  } else {
    __syncthreads();
    typename Body::State st;
    body.init(st, part);
    for (int c = 0; c < nch; ++c) {
      if (a.dev_mode != 1) {
        double *base = lds + (size_t)(c & 1) * CH * 64 + sc;
        const int rmax = min(R, rows - c * R);
        int r = 0;
        for (; r + 2 <= rmax; r += 2) {
          body.step(st, base + (size_t)r * NS * 64, c * R + r);
          body.step(st, base + (size_t)(r + 1) * NS * 64, c * R + r + 1);
        }
        for (; r < rmax; ++r) body.step(st, base + (size_t)r * NS * 64, c * R + r);
      }
      __syncthreads();
    }
    body.finish(st, part, g0);
  }
#undef SEQ_PTR
#undef SEQ_LOAD
#undef SEQ_PUT
#undef SEQ_STORE
}

// (the bodies - FactorBody, ForwardBody, BackwardBody, SpikeBody - and the reduced system's per-lane arithmetic: dsp_ipm_seq.hpp)

// ---- the time-parallel form's kernels between the walks (dsp_ipm_seq.hpp) -------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(64) void k_ipm_red_factor(IpmArgs a) {
  const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (a.w.state[s] != 0) return;
  ipm_red_factor_lane<W>(a.P.parts, a.w.sfin, a.w.cfin, a.w.gs, (size_t)a.P.Mp * a.w.Bp, a.w.redf, a.w.Bp, s);
}
template <int W>
__global__ __launch_bounds__(64) void k_ipm_red_solve(IpmArgs a, double *x) {
  const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (a.w.state[s] != 0) return;
  ipm_red_solve_lane<W>(a.P.parts, a.w.redf, a.w.bd, x, a.w.Bp, s);
}
// The same solve with the chain FED: ipm_red_solve_lane is one wave walking 2 (P - 1) dependent block steps, each waiting for its own
// loads (36 + 12 doubles per lane forward, 57 + 6 backward): 136 us for 63 blocks whatever the batch - 1.1 us per step, a memory round
// trip - where the arithmetic of a step is ~100 FMAs.  Here wave 0 computes and kRedMovers waves feed it, as in k_seq: mover k owns the
// steps u = k, k + M, ...; it requests ALL of step u + M's data right after it has put step u's into the LDS slot of its parity, which
// is one barrier before wave 0 reads it - the requests have M - 1 steps to arrive, and the only vector-memory operations a mover has
// outstanding when it waits are the ones it needs.  A lane group is split over kRedLaneSplit workgroups of LW = 16 scenarios, and a
// mover's load instruction carries 64 / LW = 4 ELEMENTS of those 16 scenarios (lane = element slot x scenario): 17 - 19 instructions per
// step instead of 63 - a wave-wide load costs ~16 cycles of address processing whatever its lanes ask for, and with one element per
// instruction the mover on turn held every barrier for 0.9 us (first version of this kernel: 119 us, no better than the lone wave).
// Same operations in the same order as the lane function (the CPU harness' reference): identical results.  The backward steps read the
// separator rows the forward steps of THIS kernel wrote: the two phases are separated by a barrier after a device-scope fence, and those
// rows are loaded past the CU's vector cache.
constexpr int kRedMovers = 15;
constexpr int kRedLaneSplit = 4;
template <int W>
__global__ __launch_bounds__(64 * (kRedMovers + 1)) void k_ipm_red_solve_fed(IpmArgs a, double *x) {
  constexpr int NK = IpmRed<W>::NK, NR = IpmRed<W>::NR, M = kRedMovers;
  constexpr int LS = kRedLaneSplit, LW = 64 / LS, EPL = LS;      // scenarios per workgroup; elements per load instruction
  constexpr int ES = NR + W;                                     // slot: forward K (NK), w rows (W), border sums (W); backward K | LDL' (NR), rows (W)
  constexpr int NLK = (NK + EPL - 1) / EPL, NLW = (W + EPL - 1) / EPL, NLR = (NR + EPL - 1) / EPL;
  __shared__ double slot[3 * ES * LW];                           // [3][ES][LW]: the step being computed, the one before (its K block is
                                                                 //  the backward step's coupling: no copy of it in wave 0's registers), the next
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int sc = lane & (LW - 1), el = lane / LW;                // scenario of the workgroup's LW; element slot of a load
  const int ln = ((int)blockIdx.x % LS) * LW + sc;
  const size_t Bp = a.w.Bp, s0 = (size_t)((int)blockIdx.x / LS) * 64, s = s0 + ln;
  {
    __shared__ int any;
    if (t == 0) any = 0;
    __syncthreads();
    if (t < 64 && a.w.state[s] == 0) any = 1;
    __syncthreads();
    if (!any) return;
  }
  const IpmParts g = a.P.parts;
  const int nb = g.P - 1;
  if (nb <= 0) return;
  const int Z = a.w.bz;
  const double *redf = a.w.redf, *bd = a.w.bd;
  const size_t bdset = (size_t)g.P * W * Bp;
#define RED_SLOT(buf, e) slot[((buf) * ES + (e)) * LW + sc]
  if (wv > 0) {
    const int k = wv - 1;
    // this lane's part of load j of an array of `cnt` elements (rows of Bp doubles from `base`): element min(j EPL + el, cnt - 1)
    auto elem = [&](int j, int cnt) { return min(j * EPL + el, cnt - 1); };
    double rk[NLR], rx[NLW], rb[kIpmBorderSplit][NLW];
    // ---- forward steps ----
    auto request_f = [&](int u) {
      const double *in = redf + (size_t)u * NR * Bp + s0;
#pragma unroll
      for (int j = 0; j < NLK; ++j) rk[j] = (in + (size_t)elem(j, NK) * Bp)[ln];
      const size_t e0 = (size_t)(g.start(u) + g.interior(u));
#pragma unroll
      for (int j = 0; j < NLW; ++j) rx[j] = (x + (e0 + elem(j, W)) * Bp + s0)[ln];
#pragma unroll
      for (int z = 0; z < kIpmBorderSplit; ++z)
#pragma unroll
        for (int j = 0; j < NLW; ++j)      // (sets beyond Z: set 0 again - no select on the VALUE, which would wait for it here)
          rb[z][j] = (bd + (size_t)(z < Z ? z : 0) * bdset + ((size_t)(u + 1) * W + elem(j, W)) * Bp + s0)[ln];
    };
    int mine = k;
    if (mine < nb) request_f(mine);
    for (int it = -1; it < nb; ++it) {
      if (it + 1 == mine) {
        const int buf = mine % 3;
#pragma unroll
        for (int j = 0; j < NLK; ++j) { const int e = j * EPL + el; if (e < NK) slot[(buf * ES + e) * LW + sc] = rk[j]; }
#pragma unroll
        for (int j = 0; j < NLW; ++j) {
          const int i = j * EPL + el;
          double b = rb[0][j];
#pragma unroll
          for (int z = 1; z < kIpmBorderSplit; ++z) if (z < Z) b += rb[z][j];         // (partial border sums in a fixed order)
          if (i < W) { slot[(buf * ES + NK + i) * LW + sc] = rx[j]; slot[(buf * ES + NK + W + i) * LW + sc] = b; }
        }
        mine += M;
        if (mine < nb) request_f(mine);
      }
      __syncthreads();
    }
    __syncthreads();                                               // (wave 0's fence: the forward results are in memory)
    // ---- backward steps: step u is block nb - 1 - u ----
    auto request_b = [&](int u) {
      const int sg = nb - 1 - u;
      const double *in = redf + (size_t)sg * NR * Bp + s0;
#pragma unroll
      for (int j = 0; j < NLR; ++j) rk[j] = (in + (size_t)elem(j, NR) * Bp)[ln];
      const size_t e0 = (size_t)(g.start(sg) + g.interior(sg));
#pragma unroll
      for (int j = 0; j < NLW; ++j) rx[j] = __hip_atomic_load(x + (e0 + elem(j, W)) * Bp + s0 + ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    mine = k;
    if (mine < nb) request_b(mine);
    for (int it = -1; it < nb; ++it) {
      if (it + 1 == mine) {
        const int buf = mine % 3;
#pragma unroll
        for (int j = 0; j < NLR; ++j) { const int e = j * EPL + el; if (e < NR) slot[(buf * ES + e) * LW + sc] = rk[j]; }
#pragma unroll
        for (int j = 0; j < NLW; ++j) { const int i = j * EPL + el; if (i < W) slot[(buf * ES + NR + i) * LW + sc] = rx[j]; }
        mine += M;
        if (mine < nb) request_b(mine);
      }
      __syncthreads();
    }
  } else {
    const bool live = a.w.state[s] == 0;
    double wp[W];
#pragma unroll
    for (int j = 0; j < W; ++j) wp[j] = 0.0;
    for (int it = -1; it < nb; ++it) {
      if (it >= 0) {
        const int buf = it % 3;
        const size_t e0 = (size_t)(g.start(it) + g.interior(it));
        double w[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
          double v = RED_SLOT(buf, NK + i) - RED_SLOT(buf, NK + W + i);
#pragma unroll
          for (int j = 0; j < W; ++j) v = fma(-RED_SLOT(buf, i * W + j), wp[j], v);
          w[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W; ++i) { if (live) x[(e0 + i) * Bp + s] = w[i]; wp[i] = w[i]; }
      }
      __syncthreads();
    }
    __threadfence();
    __syncthreads();
    double xn[W];
#pragma unroll
    for (int j = 0; j < W; ++j) xn[j] = 0.0;
    for (int it = -1; it < nb; ++it) {
      if (it >= 0) {
        const int buf = it % 3, prev = (it + 2) % 3, sg = nb - 1 - it;
        const size_t e0 = (size_t)(g.start(sg) + g.interior(sg));
        double Lt[W][W], dinv[W], u[W];
#pragma unroll
        for (int i = 0; i < W; ++i)
#pragma unroll
          for (int b = 0; b < W; ++b) {
            if (b < i) Lt[i][b] = RED_SLOT(buf, NK + ipm_tri(i, b));
            else { Lt[i][b] = 0.0; if (b == i) dinv[i] = RED_SLOT(buf, NK + ipm_tri(i, i)); }
          }
#pragma unroll
        for (int i = 0; i < W; ++i) u[i] = RED_SLOT(buf, NR + i);
        ipm_red_ldl_solve<W>(Lt, dinv, u);
#pragma unroll
        for (int j = 0; j < W; ++j) {
          double v = u[j];
#pragma unroll
          for (int i = 0; i < W; ++i) v = fma(-(it > 0 ? RED_SLOT(prev, i * W + j) : 0.0), xn[i], v);      // K of block sigma + 1 (none above the last block)
          u[j] = v;
        }
#pragma unroll
        for (int j = 0; j < W; ++j) { if (live) x[(e0 + j) * Bp + s] = u[j]; xn[j] = u[j]; }
      }
      __syncthreads();
    }
  }
#undef RED_SLOT
}
constexpr int kIpmBorderWaves = 8;
// border sums of partition blockIdx.x / bz + 1: NWV waves (x bz workgroups) take every (NWV bz)-th interior row, LDS combines (W = 6: 16
// waves - with 8 the kernel ran at 3.1 TB/s next to the correction's 5.4 on the same streams; W = 8: 8, the combine buffer must stay
// under 64 KB); workgroup z of a partition leaves its sums in partial set z of bd (summed in a fixed order by the reduced solve)
template <int W, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_ipm_border_dot(IpmArgs a, const double *z) {
  __shared__ double red[NWV][W][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, p = (int)blockIdx.x / a.w.bz + 1, zi = (int)blockIdx.x % a.w.bz;
  const size_t s = (size_t)blockIdx.y * 64 + lane;
  double acc[W];
  ipm_border_dot_lane<W>(a.P.parts, p, zi * NWV + wv, NWV * a.w.bz, a.w.gs, (size_t)a.P.Mp * a.w.Bp, z, a.w.Bp, s, acc);
#pragma unroll
  for (int j = 0; j < W; ++j) red[wv][j][lane] = acc[j];
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int j = 0; j < W; ++j) {
      double t = 0.0;
#pragma unroll
      for (int v = 0; v < NWV; ++v) t += red[v][j][lane];
      a.w.bd[(size_t)zi * a.P.parts.P * W * a.w.Bp + ((size_t)p * W + j) * a.w.Bp + s] = t;
    }
  }
}
template <int W>
__global__ __launch_bounds__(64 * kIpmBorderWaves) void k_ipm_border_apply(IpmArgs a, double *x) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, p = (int)blockIdx.x / a.w.bz + 1, zi = (int)blockIdx.x % a.w.bz;
  const size_t s = (size_t)blockIdx.y * 64 + lane;
  if (a.w.state[s] != 0) return;
  ipm_border_apply_lane<W>(a.P.parts, p, zi * kIpmBorderWaves + wv, kIpmBorderWaves * a.w.bz, a.w.gs, (size_t)a.P.Mp * a.w.Bp, a.w.band, x, a.w.Bp, s);
}

// ---- Woodbury ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ipm_wide_rhs(IpmArgs a, int k) {        // q = column `wide_col[k]` of the scaled matrix
  IPM_LANE();
  int i0, i1; ipm_chunk(a.P.m, a.w.nch, cid, i0, i1);
  for (int i = i0; i < i1; ++i) a.w.q[(size_t)i * Bp + s] = 0.0;
  int lo = a.P.wptr[k], hi = a.P.wptr[k + 1];
  const int end = hi;
  while (lo < hi) {                                                     // the entries are sorted by row: the first one at or after i0
    const int mid = (lo + hi) >> 1;                                   // (every wave walking all 8736 entries of a design column: 1.9 ms)
    if (a.P.wrow[mid] < i0) lo = mid + 1; else hi = mid;
  }
  for (int p = lo; p < end; ++p) {
    const int i = a.P.wrow[p];
    if (i >= i1) break;
    a.w.q[(size_t)i * Bp + s] = a.P.wval[p];
  }
}

__global__ __launch_bounds__(256) void k_ipm_copy_m(IpmArgs a, const double *src, double *dst) {
  IPM_LANE();
  int i0, i1; ipm_chunk(a.P.m, a.w.nch, cid, i0, i1);
  for (int i = i0; i < i1; ++i) dst[(size_t)i * Bp + s] = src[(size_t)i * Bp + s];
}

// Woodbury: S = diag(1 / theta_d) + Ad' B^-1 Ad (K <= 4) and t = Ad' q are sums over the wide columns' entries: partials per wave
// over slices of the entries (one wave walking a year-long design column alone took 0.5 ms at T = 672), finished per scenario
__global__ __launch_bounds__(256) void k_ipm_wood_part(IpmArgs a, int what) {       // what 0: S (K * K slots), 1: t (K slots)
  IPM_LANE();
  const int K = a.P.K, m = a.P.m;
  for (int k1 = 0; k1 < K; ++k1) {
    int p0, p1; ipm_chunk(a.P.wptr[k1 + 1] - a.P.wptr[k1], a.w.nch, cid, p0, p1);
    p0 += a.P.wptr[k1]; p1 += a.P.wptr[k1];
    if (what == 0) {
      for (int k2 = 0; k2 < K; ++k2) {
        double t = 0.0;
#pragma unroll 4
        for (int p = p0; p < p1; ++p) t = fma(a.P.wval[p], a.w.biad[((size_t)k2 * m + a.P.wrow[p]) * Bp + s], t);
        ipm_put<0>(a, k1 * K + k2, t, s);
      }
    } else {
      double t = 0.0;
#pragma unroll 4
      for (int p = p0; p < p1; ++p) t = fma(a.P.wval[p], a.w.q[(size_t)a.P.wrow[p] * Bp + s], t);
      ipm_put<0>(a, k1, t, s);
    }
  }
}

// S inverted; a wide column that is fixed in this scenario (theta = 0) drops out
__global__ __launch_bounds__(256) void k_ipm_wood_s(IpmArgs a) {
  const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63), Bp = a.w.Bp;
  const int K = a.P.K;
  double sums[kIpmMaxK * kIpmMaxK];
  ipm_finish<kIpmMaxK * kIpmMaxK, 0>(a.w.part, a.w.nch / 4, Bp, s, sums, K * K);  // (slots beyond K * K: not summed, not looked at)
  if (threadIdx.x >= 64 || a.w.state[s] != 0) return;
  double S[kIpmMaxK][kIpmMaxK], I[kIpmMaxK][kIpmMaxK];
  bool dead[kIpmMaxK];
#pragma unroll
  for (int k1 = 0; k1 < kIpmMaxK; ++k1) {
    dead[k1] = true;
#pragma unroll
    for (int k2 = 0; k2 < kIpmMaxK; ++k2) { S[k1][k2] = k1 == k2 ? 1.0 : 0.0; I[k1][k2] = k1 == k2 ? 1.0 : 0.0; }
  }
#pragma unroll
  for (int k1 = 0; k1 < kIpmMaxK; ++k1) {
    if (k1 >= K) continue;
    const double th = a.w.th[(size_t)a.P.wide_col[k1] * Bp + s];
    dead[k1] = !(th > 0.0);
    if (dead[k1]) continue;
#pragma unroll
    for (int k2 = 0; k2 < kIpmMaxK; ++k2) {
      if (k2 >= K) continue;
      double t = 0.0;
#pragma unroll
      for (int e = 0; e < kIpmMaxK * kIpmMaxK; ++e) if (e == k1 * K + k2) t = sums[e];
      S[k1][k2] = t + (k1 == k2 ? 1.0 / th : 0.0);
    }
  }
#pragma unroll
  for (int k1 = 0; k1 < kIpmMaxK; ++k1)
    if (dead[k1]) {
#pragma unroll
      for (int k2 = 0; k2 < kIpmMaxK; ++k2) { S[k1][k2] = k1 == k2 ? 1.0 : 0.0; S[k2][k1] = k1 == k2 ? 1.0 : 0.0; }
    }
  // Gauss-Jordan (S is symmetric positive definite)
#pragma unroll
  for (int c = 0; c < kIpmMaxK; ++c) {
    const double inv = 1.0 / S[c][c];
#pragma unroll
    for (int k = 0; k < kIpmMaxK; ++k) { S[c][k] *= inv; I[c][k] *= inv; }
#pragma unroll
    for (int r = 0; r < kIpmMaxK; ++r) {
      if (r == c) continue;
      const double g = S[r][c];
#pragma unroll
      for (int k = 0; k < kIpmMaxK; ++k) { S[r][k] = fma(-g, S[c][k], S[r][k]); I[r][k] = fma(-g, I[c][k], I[r][k]); }
    }
  }
#pragma unroll
  for (int k1 = 0; k1 < kIpmMaxK; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < kIpmMaxK; ++k2)
      if (k1 < K && k2 < K) a.w.sinv[((size_t)k1 * K + k2) * Bp + s] = (dead[k1] || dead[k2]) ? 0.0 : I[k1][k2];
}

// g = Sinv (Ad' q)
__global__ __launch_bounds__(64 * kFinW) void k_ipm_wood_g(IpmArgs a) {
  const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63), Bp = a.w.Bp;
  const int K = a.P.K;
  double t[kIpmMaxK];
  ipm_finish<kIpmMaxK, 0, kFinW>(a.w.part, a.w.nch / 4, Bp, s, t, K);
  if (threadIdx.x >= 64 || a.w.state[s] != 0) return;
#pragma unroll
  for (int k1 = 0; k1 < kIpmMaxK; ++k1) {
    if (k1 >= K) continue;
    double g = 0.0;
#pragma unroll
    for (int k2 = 0; k2 < kIpmMaxK; ++k2) if (k2 < K) g = fma(a.w.sinv[((size_t)k1 * K + k2) * Bp + s], t[k2], g);
    a.w.tk[(size_t)k1 * Bp + s] = g;
  }
}

// dy (+)= q - BiAd g
__global__ __launch_bounds__(256) void k_ipm_wood_axpy(IpmArgs a, int accumulate) {
  IPM_LANE();
  const int K = a.P.K, m = a.P.m;
  double g[kIpmMaxK];
#pragma unroll
  for (int k = 0; k < kIpmMaxK; ++k) g[k] = k < K ? a.w.tk[(size_t)k * Bp + s] : 0.0;
  int i0, i1; ipm_chunk(m, a.w.nch, cid, i0, i1);
  for (int i = i0; i < i1; ++i) {
    double x = a.w.q[(size_t)i * Bp + s];
#pragma unroll
    for (int k = 0; k < kIpmMaxK; ++k) if (k < K) x = fma(-g[k], a.w.biad[((size_t)k * m + i) * Bp + s], x);
    const size_t at = (size_t)i * Bp + s;
    a.w.dy[at] = accumulate ? a.w.dy[at] + x : x;
  }
}

// ---- directions ----------------------------------------------------------------------------------------------------------------------
// complementarity targets of one column: cz, cf (mode 0: predictor; 1: corrector with sigma mu and the second-order terms)
__device__ __forceinline__ void ipm_targets(const IpmArgs &a, size_t at, size_t s, int mode, bool hl, bool hu, double wl, double tu, double z, double f,
                                            double &cz, double &cf) {
  const double sm = mode ? a.w.sc[SC_SIGMU * a.w.Bp + s] : 0.0;
  cz = hl ? (sm - (mode ? a.w.corl[at] : 0.0)) / wl - z : 0.0;
  cf = hu ? (sm - (mode ? a.w.coru[at] : 0.0)) / tu - f : 0.0;
}

// rhs = rp + Abar tn   (mode 0: tn of the predictor, k_ipm_resid; mode 1: tn = th ta + sigma mu th tb, k_ipm_dir's mode 0)
__global__ __launch_bounds__(256) void k_ipm_rhs(IpmArgs a, int mode) {
  IPM_LANE();
  int i0, i1; ipm_chunk(a.P.m, a.w.nch, cid, i0, i1);
  const double sm = a.w.sc[SC_SIGMU * Bp + s];
#pragma unroll 2
  for (int i = i0; i < i1; ++i) {
    const size_t at = (size_t)i * Bp + s;
    const double r = mode == 0 ? a.w.rp[at] + ipm_au(a.P, a.w.tn, i, Bp, s)
                               : a.w.rp[at] + (ipm_au(a.P, a.w.rt, i, Bp, s) + sm * ipm_au(a.P, a.w.dz, i, Bp, s));
    a.w.rhs[at] = r;
    a.w.q[at] = r;
  }
}

// refinement: tn = theta (Abar' dy) ; q = res = rhs - Abar tn
__global__ __launch_bounds__(256) void k_ipm_ref_cols(IpmArgs a) {
  IPM_LANE();
  const int N = a.P.n + a.P.m;
  int j0, j1; ipm_chunk(N, a.w.nch, cid, j0, j1);
#pragma unroll 4
  for (int j = j0; j < j1; ++j) {
    const size_t at = (size_t)j * Bp + s;
    a.w.tn[at] = a.w.th[at] * ipm_aty(a.P, a.w.dy, a.w.wat, j, Bp, s);
  }
}
__global__ __launch_bounds__(256) void k_ipm_ref_rows(IpmArgs a) {
  IPM_LANE();
  int i0, i1; ipm_chunk(a.P.m, a.w.nch, cid, i0, i1);
  double mq = 0.0, mr = 0.0;                       // max |rhs - N dy|, max |rhs| of the chunk: what k_ipm_resflag decides on
#pragma unroll 2
  for (int i = i0; i < i1; ++i) {
    const size_t at = (size_t)i * Bp + s;
    const double r = a.w.rhs[at], q = r - ipm_au(a.P, a.w.tn, i, Bp, s);
    a.w.q[at] = q;
    mq = fmax(mq, fabs(q)); mr = fmax(mr, fabs(r));
  }
  ipm_put<2>(a, 0, mq, s);
  ipm_put<2>(a, 1, mr, s);
}

// how well the Newton system is solved: max |rhs - N dy| against max |rhs| per scenario (partial maxima: k_ipm_ref_rows); a scenario beyond
// the tolerance asks for a (further) step of iterative refinement - of the whole batch: the solves cost the same for one lane as for 64
__global__ __launch_bounds__(64 * kFinW) void k_ipm_resflag(IpmArgs a, double tol) {
  const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63), Bp = a.w.Bp;
  double t[2];
  ipm_finish<2, 2, kFinW>(a.w.part, a.w.nch / 4, Bp, s, t);
  if (threadIdx.x >= 64 || a.w.state[s] != 0) return;
  const bool strict = a.w.sc[SC_STRICT * Bp + s] > 0.0;
  const double lim = a.w.endg[s] ? (strict ? a.reftol_strict_end : a.reftol_end) : (strict ? a.reftol_strict : tol);
  if (!(t[0] <= lim * t[1])) atomicAdd(a.w.counts + 3, 1);
}

// dv, dz, df from dy; partial minima of the step lengths
__global__ __launch_bounds__(256) void k_ipm_dir(IpmArgs a, int mode) {
  IPM_LANE();
  const int N = a.P.n + a.P.m;
  int j0, j1; ipm_chunk(N, a.w.nch, cid, j0, j1);
  double ap = 1e300, ad = 1e300, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll 2
  for (int j = j0; j < j1; ++j) {
    const size_t at = (size_t)j * Bp + s;
    const double l = a.w.l[at], u = a.w.u[at], v = a.w.v[at], th = a.w.th[at], z = a.w.z[at], f = a.w.f[at];
    const bool hl = ipm_fin(l) && th > 0.0, hu = ipm_fin(u) && th > 0.0;
    const double wl = v - l, tu = u - v;
    double cz, cf;
    ipm_targets(a, at, s, mode, hl, hu, wl, tu, z, f, cz, cf);
    // mode 0: rt holds the predictor's rt; mode 1: the corrector's rt = ta + sigma mu tb was never formed - rt holds th ta, dz holds th tb
    // (written below by mode 0), and dv = th (A' dy) - th rt
    const double aty = ipm_aty(a.P, a.w.dy, a.w.wat, j, Bp, s);
    const double dv = mode == 0 ? th * (aty - a.w.rt[at]) : th * aty - (a.w.rt[at] + a.w.sc[SC_SIGMU * Bp + s] * a.w.dz[at]);
    const double dz = hl ? cz - z / wl * dv : 0.0;
    const double df = hu ? cf + f / tu * dv : 0.0;
    if (mode == 0) {
      // the predictor's direction is used for two things only: mu after the affine step - a bilinear form in the two step lengths,
      //   sum (z + ad dz)(w + ap dv) + sum (f + ad df)(t - ap dv) = mu nb + ap q1 + ad q2 + ap ad q3,
      // so its three sums are taken HERE, before the step lengths exist (k_ipm_steps finishes them into sigma) - and the second-order
      // terms of the corrector.  Nothing reads the predictor's dv / dz / df afterwards: they are not stored (round 6: the separate
      // k_ipm_muaff pass read nine arrays to do this)
      if (hl) { q1 += z * dv; q2 += wl * dz; q3 += dz * dv; }
      if (hu) { q1 -= f * dv; q2 += tu * df; q3 -= df * dv; }
      const double corl = dv * dz, coru = -dv * df;
      a.w.corl[at] = corl;
      a.w.coru[at] = coru;
      // The corrector's rt = rd - cz + cf is LINEAR in sigma mu, which does not exist before this kernel's sums are finished:
      //   rt = ta + sigma mu tb,   ta = rd + (corl / w + z) - (coru / t + f),   tb = -1 / w + 1 / t        (terms of the bounds that exist)
      // th ta goes where the predictor's rt was, th tb into dz's slot (free until mode 1 writes dz at this very index): k_ipm_rhs forms
      // rp + Abar (th ta) + sigma mu Abar (th tb), mode 1 reads the two - the pass that formed rt / tn (nine arrays read) is gone
      double ta = a.w.rd[at], tb = 0.0;
      if (hl) { ta += corl / wl + z; tb -= 1.0 / wl; }
      if (hu) { ta -= coru / tu + f; tb += 1.0 / tu; }
      a.w.rt[at] = th * ta;
      a.w.dz[at] = th * tb;
    } else {
      a.w.dv[at] = dv; a.w.dz[at] = dz; a.w.df[at] = df;
    }
    if (hl && dv < 0.0) ap = fmin(ap, -wl / dv);
    if (hu && dv > 0.0) ap = fmin(ap, tu / dv);
    if (hl && dz < 0.0) ad = fmin(ad, -z / dz);
    if (hu && df < 0.0) ad = fmin(ad, -f / df);
  }
  ipm_put<1>(a, 0, ap, s);
  ipm_put<1>(a, 1, ad, s);
  if (mode == 0) {
    ipm_put<0>(a, 2, q1, s);
    ipm_put<0>(a, 3, q2, s);
    ipm_put<0>(a, 4, q3, s);
  }
}

__global__ __launch_bounds__(64 * kFinW) void k_ipm_steps(IpmArgs a, int mode) {
  const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63), Bp = a.w.Bp;
  double t[2];
  ipm_finish<2, 1, kFinW>(a.w.part, a.w.nch / 4, Bp, s, t);
  double q[3] = {0.0, 0.0, 0.0};
  if (mode == 0) ipm_finish<3, 0, kFinW>(a.w.part + (size_t)2 * (a.w.nch / 4) * Bp, a.w.nch / 4, Bp, s, q);      // (uniform branch: every thread of the block)
  if (threadIdx.x >= 64 || a.w.state[s] != 0) return;
  if (mode == 0) {
    const double apa = fmin(t[0], 1.0), ada = fmin(t[1], 1.0);
    a.w.sc[SC_APA * Bp + s] = apa; a.w.sc[SC_ADA * Bp + s] = ada;
    // sigma from mu after the affine step (k_ipm_dir's three sums; the cancellation as both steps approach 1 is 1e-16 of mu: sigma has a floor)
    const double mu = a.w.sc[SC_MU * Bp + s], nb = fmax(a.w.sc[SC_NB * Bp + s], 1.0);
    const double mu_aff = fmax((mu * a.w.sc[SC_NB * Bp + s] + apa * q[0] + ada * q[1] + apa * ada * q[2]) / nb, 0.0);
    const double r = mu > 0.0 ? mu_aff / mu : 1.0;
    a.w.sc[SC_SIGMU * Bp + s] = fmin(fmax(r * r * r, a.sigmin), 1.0) * mu;
  } else {
    // fraction of the way to the boundary: a.step (0.99) for a long step, a.step_blocked where the boundary is closer than a.step_thr of the
    // Newton step - the slow members of the wind + battery family take a hundred such steps, and staying further inside shortens that phase
    a.w.sc[SC_AP * Bp + s] = fmin(1.0, (t[0] < a.step_thr ? a.step_blocked : a.step) * t[0]);
    a.w.sc[SC_AD * Bp + s] = fmin(1.0, (t[1] < a.step_thr ? a.step_blocked : a.step) * t[1]);
  }
}

__global__ __launch_bounds__(256) void k_ipm_update(IpmArgs a) {
  IPM_LANE();
  const int N = a.P.n + a.P.m;
  const double ap = a.w.sc[SC_AP * Bp + s], ad = a.w.sc[SC_AD * Bp + s];
  int j0, j1; ipm_chunk(N, a.w.nch, cid, j0, j1);
#pragma unroll 2
  for (int j = j0; j < j1; ++j) {
    const size_t at = (size_t)j * Bp + s;
    if (!(a.w.th[at] > 0.0)) continue;
    a.w.v[at] += ap * a.w.dv[at];
    a.w.z[at] += ad * a.w.dz[at];
    a.w.f[at] += ad * a.w.df[at];
  }
  int i0, i1; ipm_chunk(a.P.m, a.w.nch, cid, i0, i1);
  for (int i = i0; i < i1; ++i) a.w.y[(size_t)i * Bp + s] += ad * a.w.dy[(size_t)i * Bp + s];
}

// A step that polluted the primal residual (k_ipm_decide) is taken back: the directions and step lengths of the iteration are still in
// place, and the scenario's next Newton systems are refined to the strict tolerances.  Launched only when some lane asked for it.
__global__ __launch_bounds__(256) void k_ipm_undo(IpmArgs a) {
  IPM_LANE();
  if (!(a.w.sc[SC_UNDO * Bp + s] > 0.0)) return;
  const int N = a.P.n + a.P.m;
  const double ap = a.w.sc[SC_AP * Bp + s], ad = a.w.sc[SC_AD * Bp + s];
  int j0, j1; ipm_chunk(N, a.w.nch, cid, j0, j1);
#pragma unroll 2
  for (int j = j0; j < j1; ++j) {
    const size_t at = (size_t)j * Bp + s;
    if (!(a.w.th[at] > 0.0)) continue;
    a.w.v[at] -= ap * a.w.dv[at];
    a.w.z[at] -= ad * a.w.dz[at];
    a.w.f[at] -= ad * a.w.df[at];
  }
  int i0, i1; ipm_chunk(a.P.m, a.w.nch, cid, i0, i1);
  for (int i = i0; i < i1; ++i) a.w.y[(size_t)i * Bp + s] -= ad * a.w.dy[(size_t)i * Bp + s];
}
__global__ void k_ipm_undone(IpmArgs a) {
  const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
  a.w.sc[SC_UNDO * a.w.Bp + s] = 0.0;
}

// ---- termination: the HBM-resident path's KKT test on the unscaled problem (control_decide, dsp_stream.hpp) ------------------------------
__global__ __launch_bounds__(64 * kFinW) void k_ipm_decide(IpmArgs a) {
  const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63), Bp = a.w.Bp;
  double q[8];
  ipm_finish<8, 0, kFinW>(a.w.part + (size_t)2 * (a.w.nch / 4) * Bp, a.w.nch / 4, Bp, s, q);      // (k_ipm_resid's partials 2 .. 9)
  if (threadIdx.x >= 64 || a.w.state[s] != 0) return;
  StreamCtrl &c = a.sw.ctrl[a.w.sid[s]];
  const dsp_options &o = a.opt;
  const double cs = a.w.sc[SC_CS * Bp + s];
  const double po = cs * q[5], dobj = cs * (q[2] + q[7]);
  const double rp = sqrt(q[0]) / (1.0 + c.qn), rd = sqrt(q[3]) / (1.0 + c.cn);
  const double gap = fabs(po - dobj);
  bool fin;
  if (o.eps_obj > 0.0) {
    const double lim = fmax(o.eps_obj * (1.0 + fabs(po + c.c0)), 1e-12 * cs * q[6]);
    fin = rp <= o.eps_rel && rd <= o.eps_rel && gap + cs * (q[1] + q[4]) <= lim;
  } else {
    fin = rp <= o.eps_rel && rd <= o.eps_rel && gap / (1.0 + fabs(po) + fabs(dobj)) <= o.eps_rel;
  }
  const double mu = a.w.sc[SC_MU * Bp + s];
  a.w.iters[s] = a.it - 1;                                                        // (the test of iteration `it` is on the iterate `it - 1` steps left)
  c.last_rp = rp; c.last_rd = rd; c.last_rg = gap / (1.0 + fabs(po) + fabs(dobj));
  // The end game's failure mode (round 6, the 256 distinct members: 10 of them; lab trace tools/ipm_lab.py member 58): Theta reaches 1e17,
  // the pivot of a row that shares a basic column with its neighbour is the difference of two numbers of that size - rounding noise -, the
  // step leaves a primal residual in that row (1e-10 -> 5e-8 in one iteration) that no later step removes (refinement does not contract in
  // that direction), while mu collapses at 20 x per iteration: the objective test passes, the feasibility test never does.  Such a lane stops
  // HERE, with its iterate (objective within a few limits of the optimum, residual 1e-7): state 5 = given up WARM - the PDHG form it goes to
  // starts from this point instead of from zero (k_ipm_export).
  const double rpm = fmin(a.w.sc[SC_RPMIN * Bp + s], rp);
  a.w.sc[SC_RPMIN * Bp + s] = rpm;
  const bool finite = po == po && mu == mu && fabs(po) < 1e300;
  const bool polluted = a.w.endg[s] && rp > 30.0 * o.eps_rel && rp > 1e3 * rpm;
  const bool broken = !(po == po) || !(mu == mu) || !(mu > 0.0);
  const bool limit = a.it - 1 >= a.max_it;
  // (steps that stay below 1e-4: an LP without a solution, or a breakdown - not this method's scenario)
  const int stalled = (!fin && !broken && fmin(a.w.sc[SC_AP * Bp + s], a.w.sc[SC_AD * Bp + s]) < 1e-4) ? a.w.stall[s] + 1 : 0;
  a.w.stall[s] = stalled;
  const double bound = gap + cs * (q[1] + q[4]), lim1 = fmax(o.eps_obj * (1.0 + fabs(po + c.c0)), 1e-300);
  int state = 0;
  // a polluted step is taken back (k_ipm_undo) and the scenario goes on under the strict refinement tolerances - a.max_undo times
  // (seen at once, not only in the end game: rp falls monotonically on this method's path; a jump of 1000 x past the tolerance is the pollution)
  const bool jumped = rp > 30.0 * o.eps_rel && rp > 1e3 * rpm;
  const bool undo = !fin && finite && !broken && jumped && !limit && a.w.sc[SC_STRICT * Bp + s] < (double)a.max_undo;
  if (undo) {
    a.w.sc[SC_STRICT * Bp + s] += 1.0;
    a.w.sc[SC_UNDO * Bp + s] = 1.0;
    atomicAdd(a.w.counts + 4, 1);
  }
  if (fin) state = 1;
  else if (undo) state = 0;
  else if (finite && !broken && (polluted || limit)) state = 5;                   // given up warm
  else if (broken || limit || stalled >= 12) state = 2;                          // given up: the PDHG forms take the scenario from their own start
  if (state == 1) a.w.sc[SC_POBJ * Bp + s] = po;
  if (state != 0) {
    a.w.state[s] = state;
    atomicAdd(a.w.counts + 0, 1);
    if (state == 1) atomicAdd(a.w.counts + 1, 1);
  } else if (bound <= 1e4 * lim1) {
    a.w.endg[s] = 1;
    atomicAdd(a.w.counts + 2, 1);                                                 // end game: three refinement steps from here on
  } else if (bound <= 1e7 * lim1) {
    atomicAdd(a.w.counts + 3, 1);                                                 // one refinement step (none while the iterate is far out)
  }
}

// lanes beyond the batch, invalid scenarios (k_init_control), scenarios with free columns (k_ipm_setup)
__global__ void k_ipm_begin(IpmArgs a, int phase) {
  const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (phase == 0) {
    int st = 0;
    if (s >= (size_t)a.w.B) st = 2;
    else if (a.sw.ctrl[s].done) st = 2;
    a.w.state[s] = st;
    a.w.sid[s] = (int)min(s, (size_t)max(a.w.B - 1, 0));
    a.w.iters[s] = 0;
    a.w.stall[s] = 0;
    a.w.endg[s] = 0;
    if (st) atomicAdd(a.w.counts + 0, 1);
  } else if (a.w.state[s] == 3) {
    a.w.state[s] = 2;
    atomicAdd(a.w.counts + 0, 1);
  }
}

// solved scenarios: x+ / y+ of the scenario-major workspace (k_finalize unscales them).  Lanes in state 1 (solved, not exported yet); the
// launch is followed by k_ipm_exported, which moves them to state 4 (every wave of a lane's column of blocks tests the state here).
__global__ __launch_bounds__(256) void k_ipm_export(IpmArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t Bp = a.w.Bp, s = (size_t)blockIdx.y * 64 + lane;
  const int cid = blockIdx.x * 4 + wv;
  const int state = a.w.state[s];
  if (state != 1 && state != 5) return;
  const size_t k = (size_t)a.w.sid[s];                                            // the scenario this lane carries
  const int n = a.P.n, m = a.P.m;
  const double cs = a.w.sc[SC_CS * Bp + s];
  if (state == 5) {
    // given up warm: the iterate becomes the PDHG forms' starting point and restart anchor (scaled space, as k_init would have left a
    // caller's x0 / y0); the scenario's control block stays as k_init_control set it up
    int j0, j1; ipm_chunk(n, a.w.nch, cid, j0, j1);
    for (int j = j0; j < j1; ++j) {
      const double v = fmin(fmax(a.w.v[(size_t)j * Bp + s], a.sw.lb[k * n + j]), a.sw.ub[k * n + j]);
      a.sw.x[k * n + j] = v; a.sw.x0[k * n + j] = v; a.sw.xp[k * n + j] = v; a.sw.xbar[k * n + j] = v;
    }
    int i0, i1; ipm_chunk(m, a.w.nch, cid, i0, i1);
    for (int i = i0; i < i1; ++i) {
      double y = a.w.y[(size_t)i * Bp + s] * cs;
      if (!ipm_fin(a.sw.rlo[k * m + i])) y = fmin(y, 0.0);                        // (dual-feasible in sign, as k_init leaves a caller's y0)
      if (!ipm_fin(a.sw.rhi[k * m + i])) y = fmax(y, 0.0);
      a.sw.y[k * m + i] = y; a.sw.y0[k * m + i] = y; a.sw.yp[k * m + i] = y;
    }
    return;
  }
  if (cid == 0) {
    StreamCtrl &c = a.sw.ctrl[k];
    c.pobj = a.w.sc[SC_POBJ * Bp + s]; c.status = DSP_STATUS_OPTIMAL; c.done = 1; c.it = a.w.iters[s];
    atomicAdd(a.sw.ndone, 1);
  }
  int j0, j1; ipm_chunk(n, a.w.nch, cid, j0, j1);
  for (int j = j0; j < j1; ++j) { const double v = a.w.v[(size_t)j * Bp + s]; a.sw.xp[k * n + j] = v; a.sw.x[k * n + j] = v; }
  int i0, i1; ipm_chunk(m, a.w.nch, cid, i0, i1);
  for (int i = i0; i < i1; ++i) { const double y = a.w.y[(size_t)i * Bp + s] * cs; a.sw.yp[k * m + i] = y; a.sw.y[k * m + i] = y; }
}
__global__ void k_ipm_exported(IpmArgs a) {
  const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (a.w.state[s] == 1) a.w.state[s] = 4;
  else if (a.w.state[s] == 5) a.w.state[s] = 6;
}

// ---- packing the lanes still iterating into fewer groups (ipm_repack) ----------------------------------------------------------------
// one block per row of a scenario-minor array: new lane t takes old lane perm[t]; in place (every source is read before any target is written)
template <class T>
__global__ void k_ipm_pack(T *arr, size_t Bp, const int *perm, int n_new) {
  const size_t row = blockIdx.x;
  const int t = threadIdx.x;
  T val = T(0);
  if (t < n_new) { const int src = perm[t]; if (src >= 0) val = arr[row * Bp + src]; }
  __syncthreads();
  if (t < n_new) arr[row * Bp + t] = val;
}
__global__ void k_ipm_after_pack(IpmArgs a, int active, int n_new) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t < n_new) a.w.state[t] = t < active ? 0 : 2;
  if (t == 0) a.w.counts[0] = n_new - active;                                      // lanes of the groups in use that take no part
}

// ---- host --------------------------------------------------------------------------------------------------------------------------------
template <class T>
static hipError_t ipm_up(std::vector<void *> &allocs, const std::vector<T> &v, const T **out) {
  void *d = nullptr;
  hipError_t e = hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T));
  if (e != hipSuccess) return e;
  allocs.push_back(d);
  if (!v.empty() && (e = hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)) != hipSuccess) return e;
  *out = reinterpret_cast<const T *>(d);
  return hipSuccess;
}

// The plan: wide columns by their row span, the band of the rest, the product lists of its entries.  No plan (S->ipm stays null) when
// the matrix is not banded enough - the PDHG forms are the general path.
hipError_t ipm_create(const HostCSR &A, const HostCSR &AT, StreamSolver *S) {
  S->ipm = nullptr;
  const int off = getenv("DSP_NO_IPM") ? atoi(getenv("DSP_NO_IPM")) : 0;          // (read at every create: tests build handles of every form)
  if (off) return hipSuccess;
  const int n = A.n, m = A.m;
  if (m < 64) return hipSuccess;
  std::vector<uint8_t> wide(n, 0);
  std::vector<int32_t> wide_col;
  for (int j = 0; j < n; ++j) {
    if (AT.ptr[j + 1] == AT.ptr[j]) continue;
    int lo = m, hi = -1;
    for (int p = AT.ptr[j]; p < AT.ptr[j + 1]; ++p) { lo = std::min(lo, (int)AT.idx[p]); hi = std::max(hi, (int)AT.idx[p]); }
    if (hi - lo > kIpmSpan) { wide[j] = 1; wide_col.push_back(j); }
  }
  if ((int)wide_col.size() > kIpmMaxK) return hipSuccess;
  int W = 0;
  for (int j = 0; j < n; ++j)
    if (!wide[j] && AT.ptr[j + 1] > AT.ptr[j]) {
      int lo = m, hi = -1;
      for (int p = AT.ptr[j]; p < AT.ptr[j + 1]; ++p) { lo = std::min(lo, (int)AT.idx[p]); hi = std::max(hi, (int)AT.idx[p]); }
      W = std::max(W, hi - lo);
    }
  if (W > kIpmMaxW) return hipSuccess;
  W = W <= 6 ? 6 : 8;
  const int W1 = W + 1;
  // B(t, t - k) = sum over the narrow columns j of rows t and t - k: a_tj a_(t-k)j theta_j
  std::vector<std::vector<std::pair<int32_t, double>>> prod((size_t)m * W1);
  int bw = 1;
  for (int t = 0; t < m; ++t)
    for (int k = 0; k < W1; ++k) {
      const int r2 = t - k;
      if (r2 < 0) continue;
      auto &pl = prod[(size_t)t * W1 + k];
      for (int p = A.ptr[t]; p < A.ptr[t + 1]; ++p) {
        const int j = A.idx[p];
        if (wide[j]) continue;
        for (int p2 = A.ptr[r2]; p2 < A.ptr[r2 + 1]; ++p2)
          if (A.idx[p2] == j) pl.push_back({j, A.val[p] * A.val[p2]});
      }
      bw = std::max(bw, (int)pl.size());
    }
  bw = (bw + 3) / 4 * 4;                                              // (the kernels take the entries four at a time; padding: index 0, value 0)
  std::vector<int32_t> bcol((size_t)m * W1 * bw, 0);
  std::vector<double> bval((size_t)m * W1 * bw, 0.0);
  for (size_t e = 0; e < prod.size(); ++e)
    for (size_t q = 0; q < prod[e].size(); ++q) { bcol[e * bw + q] = prod[e][q].first; bval[e * bw + q] = prod[e][q].second; }
  // ELL of the rows (all columns) and of the narrow columns; CSR of the wide columns
  int rw = 1, cw = 1;
  for (int i = 0; i < m; ++i) rw = std::max(rw, A.ptr[i + 1] - A.ptr[i]);
  for (int j = 0; j < n; ++j) if (!wide[j]) cw = std::max(cw, AT.ptr[j + 1] - AT.ptr[j]);
  if (rw > 16 || cw > 16 || bw > 16) return hipSuccess;              // not the sparse time-banded kind after all
  rw = (rw + 3) / 4 * 4; cw = (cw + 3) / 4 * 4;
  std::vector<int32_t> rcol((size_t)m * rw, 0), crow((size_t)n * cw, 0), wptr(1, 0), wrow;
  std::vector<double> rval((size_t)m * rw, 0.0), cval((size_t)n * cw, 0.0), wval;
  for (int i = 0; i < m; ++i)
    for (int p = A.ptr[i], q = 0; p < A.ptr[i + 1]; ++p, ++q) { rcol[(size_t)i * rw + q] = A.idx[p]; rval[(size_t)i * rw + q] = A.val[p]; }
  for (int j = 0; j < n; ++j)
    if (!wide[j])
      for (int p = AT.ptr[j], q = 0; p < AT.ptr[j + 1]; ++p, ++q) { crow[(size_t)j * cw + q] = AT.idx[p]; cval[(size_t)j * cw + q] = AT.val[p]; }
  for (size_t k = 0; k < wide_col.size(); ++k) {
    const int j = wide_col[k];
    wide[j] = (uint8_t)(k + 1);
    std::vector<std::pair<int32_t, double>> ent;
    for (int p = AT.ptr[j]; p < AT.ptr[j + 1]; ++p) ent.push_back({(int32_t)AT.idx[p], AT.val[p]});
    std::sort(ent.begin(), ent.end());                                   // k_ipm_wide_rhs searches them by row
    for (const auto &en : ent) { wrow.push_back(en.first); wval.push_back(en.second); }
    wptr.push_back((int32_t)wrow.size());
  }
  IpmState *I = new IpmState();
  IpmPlan &P = I->P;
  P.n = n; P.m = m; P.W = W; P.K = (int)wide_col.size(); P.Mp = m + W;
  P.rw = rw; P.cw = cw; P.bw = bw;
  for (int k = 0; k < kIpmMaxK; ++k) P.wide_col[k] = k < P.K ? wide_col[k] : 0;
  P.col_scale = S->P.col_scale; P.row_scale = S->P.row_scale;
  {
    // time-parallel form: partitions of >= 256 rows, 64 at most (the reduced system's 63 sequential blocks then weigh about as much as
    // a partition's 820 rows at T = 8736); DSP_IPM_PARTS = 1: the sequential walks (development, tests: any count the rows allow)
    int want = std::min(64, m / 256);
    if (getenv("DSP_IPM_PARTS")) want = std::max(1, atoi(getenv("DSP_IPM_PARTS")));
    IpmParts g{};
    g.m = m; g.W = W; g.P = 1; g.Lp = m;
    if (want >= 2) {
      const int Lp = std::max((m + want - 1) / want, 4 * W);
      int parts = (m + Lp - 1) / Lp;
      if (parts > 1 && m - (parts - 1) * Lp < 2 * W + 1) parts -= 1;       // a short tail joins the partition before it
      if (parts >= 2) { g.P = parts; g.Lp = Lp; }
    }
    P.parts = g;
  }
  hipError_t e;
  if ((e = ipm_up(I->allocs, rcol, &P.rcol)) != hipSuccess || (e = ipm_up(I->allocs, rval, &P.rval)) != hipSuccess ||
      (e = ipm_up(I->allocs, crow, &P.crow)) != hipSuccess || (e = ipm_up(I->allocs, cval, &P.cval)) != hipSuccess ||
      (e = ipm_up(I->allocs, wide, &P.wide)) != hipSuccess || (e = ipm_up(I->allocs, wptr, &P.wptr)) != hipSuccess ||
      (e = ipm_up(I->allocs, wrow, &P.wrow)) != hipSuccess || (e = ipm_up(I->allocs, wval, &P.wval)) != hipSuccess ||
      (e = ipm_up(I->allocs, bcol, &P.bcol)) != hipSuccess || (e = ipm_up(I->allocs, bval, &P.bval)) != hipSuccess) {
    for (void *p : I->allocs) (void)hipFree(p);
    delete I;
    return e;
  }
  if ((e = hipHostMalloc((void **)&I->counts_host, 8 * sizeof(int))) != hipSuccess) { for (void *p : I->allocs) (void)hipFree(p); delete I; return e; }
  S->ipm = I;
  return hipSuccess;
}

int ipm_partitions(const StreamSolver *S) { return S->ipm ? S->ipm->P.parts.P : 0; }

void ipm_destroy(StreamSolver *S) {
  IpmState *I = S->ipm;
  if (!I) return;
  for (void *p : I->allocs) (void)hipFree(p);
  for (void *p : I->work_allocs) (void)hipFree(p);
  if (I->counts_host) (void)hipHostFree(I->counts_host);
  if (I->perm_host) (void)hipHostFree(I->perm_host);
  delete I;
  S->ipm = nullptr;
}

static hipError_t ipm_workspace(IpmState *I, int B) {
  if (B <= I->work_B) { I->w.B = B; return hipSuccess; }
  for (void *p : I->work_allocs) (void)hipFree(p);
  I->work_allocs.clear();
  I->work_B = 0;
  IpmWork &w = I->w;
  const IpmPlan &P = I->P;
  w.Bp = (size_t)((B + 63) / 64) * 64;
  w.B = B;
  w.G = (int)(w.Bp / 64);
  w.nch = 4 * kIpmMaxBlocks;               // (capacity of `part`; the solve sets the chunks per group count: ipm_chunks)
  const size_t N = (size_t)P.n + P.m, M = P.m;
  hipError_t e;
  auto alloc = [&](size_t doubles, double **out) {
    e = hipMalloc((void **)out, std::max<size_t>(doubles, 1) * sizeof(double));
    if (e == hipSuccess) I->work_allocs.push_back(*out);
    return e;
  };
  double **nv[] = {&w.v, &w.z, &w.f, &w.l, &w.u, &w.cb, &w.th, &w.rd, &w.dv, &w.dz, &w.df, &w.rt, &w.corl, &w.coru, &w.tn};
  double **mv[] = {&w.y, &w.dy, &w.rp, &w.rhs, &w.q, &w.res};
  for (double **p : nv) if (alloc(N * w.Bp, p) != hipSuccess) return e;
  for (double **p : mv) if (alloc(M * w.Bp, p) != hipSuccess) return e;
  if (alloc((size_t)std::max(P.K, 1) * M * w.Bp, &w.biad) != hipSuccess) return e;
  if (alloc((size_t)(P.W + 1) * P.Mp * w.Bp, &w.band) != hipSuccess) return e;
  w.gs = w.sfin = w.cfin = w.redf = w.bd = nullptr;
  if (P.parts.P > 1) {
    const size_t NT = (size_t)P.W * (P.W + 1) / 2, NR = (size_t)P.W * P.W + NT;
    if (alloc((size_t)P.W * P.Mp * w.Bp, &w.gs) != hipSuccess) return e;
    if (alloc((size_t)P.parts.P * NT * w.Bp, &w.sfin) != hipSuccess) return e;
    if (alloc((size_t)P.parts.P * NT * w.Bp, &w.cfin) != hipSuccess) return e;
    if (alloc((size_t)P.parts.P * NR * w.Bp, &w.redf) != hipSuccess) return e;
    if (alloc((size_t)kIpmBorderSplit * P.parts.P * P.W * w.Bp, &w.bd) != hipSuccess) return e;
  }
  if (alloc((size_t)SC_COUNT * w.Bp, &w.sc) != hipSuccess) return e;
  if (alloc((size_t)kIpmNQ * kIpmMaxBlocks * w.Bp, &w.part) != hipSuccess) return e;
  if (alloc((size_t)kIpmMaxK * kIpmMaxK * w.Bp, &w.sinv) != hipSuccess) return e;
  if (alloc((size_t)kIpmMaxK * w.Bp, &w.tk) != hipSuccess) return e;
  if (alloc((size_t)kIpmMaxK * w.Bp, &w.wat) != hipSuccess) return e;
  if ((e = hipMalloc((void **)&w.state, w.Bp * sizeof(int))) != hipSuccess) return e;
  I->work_allocs.push_back(w.state);
  if ((e = hipMalloc((void **)&w.iters, w.Bp * sizeof(int))) != hipSuccess) return e;
  I->work_allocs.push_back(w.iters);
  if ((e = hipMalloc((void **)&w.stall, w.Bp * sizeof(int))) != hipSuccess) return e;
  I->work_allocs.push_back(w.stall);
  if ((e = hipMalloc((void **)&w.endg, w.Bp * sizeof(int))) != hipSuccess) return e;
  I->work_allocs.push_back(w.endg);
  if ((e = hipMalloc((void **)&w.counts, 8 * sizeof(int))) != hipSuccess) return e;
  I->work_allocs.push_back(w.counts);
  if ((e = hipMalloc((void **)&w.sid, w.Bp * sizeof(int))) != hipSuccess) return e;
  I->work_allocs.push_back(w.sid);
  if ((e = hipMalloc((void **)&w.perm, w.Bp * sizeof(int))) != hipSuccess) return e;
  I->work_allocs.push_back(w.perm);
  if (I->perm_host) { (void)hipHostFree(I->perm_host); I->perm_host = nullptr; }
  if ((e = hipHostMalloc((void **)&I->perm_host, w.Bp * sizeof(int))) != hipSuccess) return e;
  I->work_B = B;
  return hipSuccess;
}

static int ipm_seq_mode() { static const int v = getenv("DSP_IPM_SEQ_MODE") ? atoi(getenv("DSP_IPM_SEQ_MODE")) : 0; return v; }

// one walk: `parts` partitions starting with partition `part0` (grid.y), `rows` rows each (`last_rows` for the last one of the `nparts`)
template <int NS, int R, class Body, int WV, int SETS>
static hipError_t ipm_walk(const IpmArgs &a, SeqArgs<NS> q, const Body &body, int parts, hipStream_t st) {
  q.Bp = a.w.Bp; q.state = a.w.state; q.lsplit = a.w.ls; q.dev_mode = ipm_seq_mode();
  const size_t lds = (size_t)2 * R * NS * 64 * sizeof(double);
  auto fn = k_seq<NS, R, Body, WV, SETS>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3((unsigned)(a.w.G * a.w.ls), (unsigned)parts), dim3(64 * WV), lds, st, q, body);
  return hipGetLastError();
}

// the factorisation of the band in a.w.band (in place).  Time-parallel form (P > 1; dsp_ipm_seq.hpp): the interiors of the P partitions at
// once, their spikes, then the separators' block-tridiagonal system.
template <int W>
static hipError_t ipm_factor(const IpmArgs &a, hipStream_t st) {
  const IpmParts &g = a.P.parts;
  const size_t stride = (size_t)a.P.Mp * a.w.Bp;
  hipError_t e;
  {
    constexpr int NS = W + 1, R = (72 * 1024) / (NS * 512);
    SeqArgs<NS> q{};
    q.band = a.w.band; q.stride = stride; q.nband = NS; q.x = nullptr; q.xstride = 0;
    q.rows = g.P > 1 ? g.Lp : a.P.Mp; q.last_rows = a.P.Mp - (g.P - 1) * g.Lp; q.nparts = g.P; q.part0 = 0; q.reverse = 0;
    q.outmask = (1u << NS) - 1u;
    if ((e = ipm_walk<NS, R, FactorBody<W>, 8, 1>(a, q, FactorBody<W>{g.P > 1 ? a.w.sfin : nullptr, a.w.Bp, g.P}, g.P, st)) != hipSuccess) return e;
  }
  if (g.P <= 1) return hipSuccess;
  {
    constexpr int NS = 2 * W + 1, R = (72 * 1024) / (NS * 512), WV = W <= 6 ? 8 : 4;      // (W = 8: 100 doubles of state - one wave per SIMD)
    SeqArgs<NS> q{};
    q.band = a.w.band + (size_t)W * a.w.Bp; q.stride = stride; q.nband = W + 1; q.x = a.w.gs; q.xstride = stride;      // 1 / d_c, L(c + k, c): factor row c + W
    q.rows = g.Lp; q.last_rows = g.cols(g.P - 1); q.nparts = g.P; q.part0 = 1; q.reverse = 0;
    q.outmask = ((1u << W) - 1u) << (W + 1);
    if ((e = ipm_walk<NS, R, SpikeBody<W>, WV, 1>(a, q, SpikeBody<W>{a.w.cfin, a.w.Bp, g}, g.P - 1, st)) != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_ipm_red_factor<W>, dim3((unsigned)a.w.G), dim3(64), 0, st, a);
  return hipGetLastError();
}

static bool ipm_red_fed(int W) {
  // (0: the one-wave reduced solve - comparison; half-bandwidth 8: 108 doubles per step do not fit a mover's registers)
  return W <= 6 && !(getenv("DSP_IPM_RED_FED") && atoi(getenv("DSP_IPM_RED_FED")) == 0);
}

// x := B^-1 x for the [m][Bp] vector `x` (in place)
template <int W>
static hipError_t ipm_bsolve(const IpmArgs &a, double *x, hipStream_t st) {
  const IpmParts &g = a.P.parts;
  const size_t stride = (size_t)a.P.Mp * a.w.Bp;
  const dim3 border((unsigned)(std::max(g.P - 1, 1) * a.w.bz), (unsigned)a.w.G);
  const bool fed = ipm_red_fed(W);
  static const int one_set = getenv("DSP_IPM_SEQ_SETS") ? atoi(getenv("DSP_IPM_SEQ_SETS")) == 1 : 0;      // development
  hipError_t e;
  {
    constexpr int NS = W + 1, R = (72 * 1024) / (NS * 512);
    SeqArgs<NS> q{};
    q.band = a.w.band + ((size_t)a.P.Mp + W) * a.w.Bp; q.stride = stride; q.nband = W; q.x = x; q.xstride = 0;     // L(i + k + 1, i): factor row i + W
    q.rows = g.P > 1 ? g.Lp : a.P.m; q.last_rows = g.cols(g.P - 1); q.nparts = g.P; q.part0 = 0; q.reverse = 0;
    q.outmask = 1u << W;
    e = one_set ? ipm_walk<NS, R, ForwardBody<W>, 8, 1>(a, q, ForwardBody<W>{}, g.P, st) : ipm_walk<NS, R, ForwardBody<W>, 16, 2>(a, q, ForwardBody<W>{}, g.P, st);
    if (e != hipSuccess) return e;
  }
  if (g.P > 1) {
    constexpr int DW = W <= 6 ? 16 : 8;
    hipLaunchKernelGGL((k_ipm_border_dot<W, DW>), border, dim3(64 * DW), 0, st, a, (const double *)x);
    if constexpr (W <= 6) if (fed) hipLaunchKernelGGL(k_ipm_red_solve_fed<W>, dim3((unsigned)(a.w.G * kRedLaneSplit)), dim3(64 * (kRedMovers + 1)), 0, st, a, x);
    if (!fed) hipLaunchKernelGGL(k_ipm_red_solve<W>, dim3((unsigned)a.w.G), dim3(64), 0, st, a, x);
    hipLaunchKernelGGL(k_ipm_border_apply<W>, border, dim3(64 * kIpmBorderWaves), 0, st, a, x);
  }
  {
    constexpr int NS = W + 2, R = (72 * 1024) / (NS * 512);
    SeqArgs<NS> q{};
    q.band = a.w.band + (size_t)W * a.w.Bp; q.stride = stride; q.nband = W + 1; q.x = x; q.xstride = 0;               // 1 / d_i, then L(i + k, i): factor row i + W
    q.rows = g.P > 1 ? g.Lp : a.P.m; q.last_rows = g.cols(g.P - 1); q.nparts = g.P; q.part0 = 0; q.reverse = 1;
    q.outmask = 1u << (W + 1);
    e = one_set ? ipm_walk<NS, R, BackwardBody<W>, 8, 1>(a, q, BackwardBody<W>{}, g.P, st) : ipm_walk<NS, R, BackwardBody<W>, 16, 2>(a, q, BackwardBody<W>{}, g.P, st);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

// dy = (A Theta A')^-1 rhs (a.w.q holds the right-hand side on entry and is overwritten); iterative refinement on the full normal
// equations while some scenario's residual is above ITS tolerance (k_ipm_resflag; max norms) (at most 3 steps); returns the steps taken
template <int W>
static hipError_t ipm_nsolve(StreamSolver *S, const IpmArgs &a, hipStream_t st, int *steps, bool check = true) {
  const dim3 blk(256), grid((unsigned)(a.w.nch / 4), (unsigned)a.w.G), lanes((unsigned)a.w.G);
  hipError_t e;
  *steps = 0;
  for (int r = 0; r <= (check ? 3 : 0); ++r) {
    if (r > 0) {
      if (a.P.K > 0) { hipLaunchKernelGGL(k_ipm_wide_aty, grid, blk, 0, st, a, (const double *)a.w.dy); hipLaunchKernelGGL(k_ipm_wide_aty_finish, lanes, dim3(64 * kFinW), 0, st, a); }
      hipLaunchKernelGGL(k_ipm_ref_cols, grid, blk, 0, st, a);
      hipLaunchKernelGGL(k_ipm_ref_rows, grid, blk, 0, st, a);
      if ((e = hipMemsetAsync(a.w.counts + 3, 0, sizeof(int), st)) != hipSuccess) return e;
      hipLaunchKernelGGL(k_ipm_resflag, lanes, dim3(64 * kFinW), 0, st, a, a.reftol);
      if ((e = hipMemcpyAsync(S->ipm->counts_host + 3, a.w.counts + 3, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
      if (S->ipm->counts_host[3] == 0) break;
      *steps = r;
    }
    if ((e = ipm_bsolve<W>(a, a.w.q, st)) != hipSuccess) return e;
    if (a.P.K > 0) {
      hipLaunchKernelGGL(k_ipm_wood_part, grid, blk, 0, st, a, 1);
      hipLaunchKernelGGL(k_ipm_wood_g, lanes, dim3(64 * kFinW), 0, st, a);
    }
    else if ((e = hipMemsetAsync(a.w.tk, 0, kIpmMaxK * a.w.Bp * sizeof(double), st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_ipm_wood_axpy, grid, blk, 0, st, a, r > 0 ? 1 : 0);
  }
  return hipGetLastError();
}

static int g_ipm_debug = 0;
// development: one lane of a scenario-minor array as raw doubles under $DSP_IPM_DUMP/
static void ipm_dump(const char *name, const double *dev, size_t rows, size_t Bp, int lane, hipStream_t st) {
  const char *dir = getenv("DSP_IPM_DUMP");
  if (!dir) return;
  (void)hipStreamSynchronize(st);
  std::vector<double> h(rows);
  (void)hipMemcpy2D(h.data(), 8, dev + lane, Bp * 8, 8, rows, hipMemcpyDeviceToHost);
  char path[512];
  snprintf(path, sizeof(path), "%s/ipm_%s.bin", dir, name);
  FILE *f = fopen(path, "wb");
  if (f) { fwrite(h.data(), 8, rows, f); fclose(f); }
}
#define IPM_DBG(what)                                                                                                   \
  do {                                                                                                                  \
    if (g_ipm_debug) {                                                                                                  \
      hipError_t de = hipStreamSynchronize(st);                                                                         \
      if (de != hipSuccess) { fprintf(stderr, "[ipm] %s: %s\n", what, hipGetErrorString(de)); return de; }             \
      if (g_ipm_debug > 1) fprintf(stderr, "[ipm] %s ok\n", what);                                                     \
    }                                                                                                                   \
  } while (0)

// workgroups per partition of the border kernels: enough of them for the chip while the lane groups are few (IpmWork::bz)
static int ipm_lane_split(int G, int P) {
  int ls = 1;
  if (getenv("DSP_IPM_LS")) ls = atoi(getenv("DSP_IPM_LS"));
  else while (ls < 4 && 2 * ls * G * std::max(P, 1) <= 256) ls *= 2;
  return ls == 2 || ls == 4 ? ls : 1;
}
static int ipm_border_split(int G, int P, int W) {
  if (!ipm_red_fed(W)) return 1;                                                          // (the one-wave reduced solve reads partial set 0 only)
  if (getenv("DSP_IPM_BZ")) return std::min(std::max(atoi(getenv("DSP_IPM_BZ")), 1), kIpmBorderSplit);
  return std::min(std::max(256 / (std::max(P - 1, 1) * std::max(G, 1)), 1), kIpmBorderSplit);
}

// The scenarios still iterating (`active` of them) are packed into the first lanes: a Newton iteration costs what its lane groups cost, and
// the members of a batch finish over a 3 x range of iteration counts (40 .. 144 on the year-long wind + battery family).  What persists
// from one iteration to the next is packed - the iterate (v, z, f, y), the problem data (l, u, cb), the objective scale, the per-lane
// counters and the lane -> scenario map; everything else (Theta, residuals, the band and its factor, the directions) is recomputed.
// Solved lanes are exported first; the stride of the arrays (Bp) stays.
static hipError_t ipm_repack(IpmState *I, IpmArgs &a, hipStream_t st, int active) {
  const int n_old = a.w.G * 64, G_new = (active + 63) / 64, n_new = G_new * 64;
  const dim3 blk(256), grid((unsigned)(a.w.nch / 4), (unsigned)a.w.G);
  hipError_t e;
  hipLaunchKernelGGL(k_ipm_export, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_ipm_exported, dim3((unsigned)a.w.G), dim3(64), 0, st, a);
  if ((e = hipMemcpyAsync(I->perm_host, a.w.state, (size_t)n_old * sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  std::vector<int> perm((size_t)n_new, -1);
  int k = 0;
  for (int t = 0; t < n_old; ++t) if (I->perm_host[t] == 0 && k < n_new) perm[(size_t)k++] = t;
  if (k != active) return hipErrorUnknown;                                         // (the device's count and its states disagree)
  for (int t = 0; t < n_new; ++t) I->perm_host[t] = perm[(size_t)t];
  if ((e = hipMemcpyAsync(a.w.perm, I->perm_host, (size_t)n_new * sizeof(int), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  const size_t N = (size_t)a.P.n + a.P.m, M = a.P.m;
  const dim3 tb((unsigned)n_old);
  for (double *arr : {a.w.v, a.w.z, a.w.f, a.w.l, a.w.u, a.w.cb}) hipLaunchKernelGGL(k_ipm_pack<double>, dim3((unsigned)N), tb, 0, st, arr, a.w.Bp, (const int *)a.w.perm, n_new);
  hipLaunchKernelGGL(k_ipm_pack<double>, dim3((unsigned)M), tb, 0, st, a.w.y, a.w.Bp, (const int *)a.w.perm, n_new);
  hipLaunchKernelGGL(k_ipm_pack<double>, dim3((unsigned)SC_COUNT), tb, 0, st, a.w.sc, a.w.Bp, (const int *)a.w.perm, n_new);
  for (int *arr : {a.w.iters, a.w.stall, a.w.endg, a.w.sid}) hipLaunchKernelGGL(k_ipm_pack<int>, dim3(1), tb, 0, st, arr, a.w.Bp, (const int *)a.w.perm, n_new);
  hipLaunchKernelGGL(k_ipm_after_pack, dim3((unsigned)G_new), dim3(64), 0, st, a, active, n_new);
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;                      // (perm_host is reused by the next repack)
  a.w.G = G_new;
  a.w.nch = 4 * ipm_blocks(G_new);
  a.w.bz = ipm_border_split(G_new, a.P.parts.P, a.P.W);
  a.w.ls = ipm_lane_split(G_new, a.P.parts.P);
  return hipGetLastError();
}

template <int W>
static hipError_t ipm_loop(StreamSolver *S, IpmArgs &a, hipStream_t st, bool *all_solved, int *newton, int *n_solved) {
  IpmState *I = S->ipm;
  a.w.G = (int)(a.w.Bp / 64);
  a.w.nch = 4 * ipm_blocks(a.w.G);
  a.w.bz = ipm_border_split(a.w.G, a.P.parts.P, a.P.W);
  a.w.ls = ipm_lane_split(a.w.G, a.P.parts.P);
  const dim3 blk(256);
  dim3 grid((unsigned)(a.w.nch / 4), (unsigned)a.w.G), lanes((unsigned)a.w.G);
  // DSP_IPM_COMPACT=0: every scenario keeps its lane to the end of the solve (measurement); groups of more than 1024 lanes: not packed
  const bool compact = !(getenv("DSP_IPM_COMPACT") && atoi(getenv("DSP_IPM_COMPACT")) == 0) && a.w.Bp <= 1024;
  int repacks = 0;
  const int trace = getenv("DSP_IPM_TRACE") ? atoi(getenv("DSP_IPM_TRACE")) : 0;      // (development; read at every solve: tests switch it on)
  g_ipm_debug = getenv("DSP_IPM_DEBUG") ? atoi(getenv("DSP_IPM_DEBUG")) : 0;
  hipError_t e;
  if ((e = hipMemsetAsync(a.w.counts, 0, 8 * sizeof(int), st)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_ipm_begin, lanes, dim3(64), 0, st, a, 0);
  hipLaunchKernelGGL(k_ipm_cmax, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_ipm_cmax_finish, lanes, dim3(64 * kFinW), 0, st, a);
  hipLaunchKernelGGL(k_ipm_setup, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_ipm_begin, lanes, dim3(64), 0, st, a, 1);
  IPM_DBG("setup");
  int refine = 0, it = 0, undone = 0;
  // the predictor's system (it only sets sigma and the second-order terms) is checked / refined once some scenario runs under the strict
  // tolerances, not before: its check is 3 - 4 % of the solve and has never asked for a step (profiles/r70i_check_pred.log)
  const int check_pred_env = getenv("DSP_IPM_CHECK_PRED") ? atoi(getenv("DSP_IPM_CHECK_PRED")) : -1;
  int lanes_total = (int)a.w.Bp;
  // Pass `it` opens with the residuals of the iterate that `it - 1` steps have left AND its KKT sums (k_ipm_resid), decides on them - a
  // scenario that passes leaves here -, and only then takes step `it`: the test rides on the residual kernel's gathers instead of a pass
  // of its own after the update.  A step taken back or a packing of the lanes changes what the residual kernel saw: it runs again.
  auto residual = [&]() {
    if (a.P.K > 0) { hipLaunchKernelGGL(k_ipm_wide_aty, grid, blk, 0, st, a, (const double *)a.w.y); hipLaunchKernelGGL(k_ipm_wide_aty_finish, lanes, dim3(64 * kFinW), 0, st, a); }
    hipLaunchKernelGGL(k_ipm_resid, grid, blk, 0, st, a);
    hipLaunchKernelGGL(k_ipm_mu, lanes, dim3(64 * kFinW), 0, st, a);
  };
  for (it = 1; it <= a.max_it + 1; ++it) {
    a.it = it;
    residual();
    IPM_DBG("resid");
    if ((e = hipMemsetAsync(a.w.counts + 2, 0, 3 * sizeof(int), st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_ipm_decide, lanes, dim3(64 * kFinW), 0, st, a);
    IPM_DBG("decide");
    if ((e = hipMemcpyAsync(I->counts_host, a.w.counts, 5 * sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    bool again = false;
    if (I->counts_host[4] > 0) {                        // steps taken back (k_ipm_decide; the directions of the previous pass are still in place)
      hipLaunchKernelGGL(k_ipm_undo, grid, blk, 0, st, a);
      hipLaunchKernelGGL(k_ipm_undone, lanes, dim3(64), 0, st, a);
      undone += I->counts_host[4];
      again = true;
      if (trace) fprintf(stderr, "[ipm] it %d: %d step(s) taken back\n", it - 1, I->counts_host[4]);
    }
    if (trace) {
      double sc[SC_COUNT];
      for (int q = 0; q < SC_COUNT; ++q) (void)hipMemcpy(&sc[q], a.w.sc + (size_t)q * a.w.Bp + (trace - 1), sizeof(double), hipMemcpyDeviceToHost);
      StreamCtrl c0{};
      (void)hipMemcpy(&c0, a.sw.ctrl + (trace - 1), sizeof(StreamCtrl), hipMemcpyDeviceToHost);
      fprintf(stderr, "[ipm] it %d lane %d: mu %.3e sigma*mu %.3e ap %.3f ad %.3f rp %.2e rd %.2e rg %.2e | finished %d solved %d endgame %d refine %d\n", it - 1, trace - 1,
              sc[SC_MU], sc[SC_SIGMU], sc[SC_AP], sc[SC_AD], c0.last_rp, c0.last_rd, c0.last_rg, I->counts_host[0], I->counts_host[1], I->counts_host[2], refine);
    }
    if (I->counts_host[0] >= lanes_total || it > a.max_it) break;
    refine = 0;
    {
      const int active = lanes_total - I->counts_host[0];
      if (compact && a.w.G > 1 && active <= 64 * (a.w.G - 1)) {
        if ((e = ipm_repack(I, a, st, active)) != hipSuccess) return e;
        lanes_total = a.w.G * 64;
        grid = dim3((unsigned)(a.w.nch / 4), (unsigned)a.w.G); lanes = dim3((unsigned)a.w.G);
        ++repacks;
        again = true;
        if (trace) fprintf(stderr, "[ipm] it %d: %d scenarios still iterating packed into %d group(s)\n", it - 1, active, a.w.G);
      }
    }
    if (again) residual();
    hipLaunchKernelGGL(k_ipm_assemble, grid, blk, 0, st, a);
    IPM_DBG("assemble");
    if (it == 1 && trace) {
      const size_t N = (size_t)a.P.n + a.P.m;
      ipm_dump("th", a.w.th, N, a.w.Bp, trace - 1, st); ipm_dump("band_in", a.w.band, (size_t)(W + 1) * a.P.Mp, a.w.Bp, trace - 1, st);
      ipm_dump("rp", a.w.rp, a.P.m, a.w.Bp, trace - 1, st); ipm_dump("rd", a.w.rd, N, a.w.Bp, trace - 1, st);
      ipm_dump("v", a.w.v, N, a.w.Bp, trace - 1, st); ipm_dump("l", a.w.l, N, a.w.Bp, trace - 1, st); ipm_dump("u", a.w.u, N, a.w.Bp, trace - 1, st);
      ipm_dump("cb", a.w.cb, N, a.w.Bp, trace - 1, st);
    }
    if ((e = ipm_factor<W>(a, st)) != hipSuccess) return e;
    IPM_DBG("factor");
    for (int k = 0; k < a.P.K; ++k) {
      hipLaunchKernelGGL(k_ipm_wide_rhs, grid, blk, 0, st, a, k);
      if ((e = ipm_bsolve<W>(a, a.w.q, st)) != hipSuccess) return e;
      hipLaunchKernelGGL(k_ipm_copy_m, grid, blk, 0, st, a, (const double *)a.w.q, a.w.biad + (size_t)k * a.P.m * a.w.Bp);
    }
    if (a.P.K > 0) {
      hipLaunchKernelGGL(k_ipm_wood_part, grid, blk, 0, st, a, 0);
      hipLaunchKernelGGL(k_ipm_wood_s, lanes, blk, 0, st, a);
    }
    IPM_DBG("woodbury");
    if (it == 1 && trace) {
      ipm_dump("band_out", a.w.band, (size_t)(W + 1) * a.P.Mp, a.w.Bp, trace - 1, st);
      ipm_dump("biad", a.w.biad, (size_t)std::max(a.P.K, 1) * a.P.m, a.w.Bp, trace - 1, st);
      ipm_dump("sinv", a.w.sinv, (size_t)kIpmMaxK * kIpmMaxK, a.w.Bp, trace - 1, st);
    }
    for (int mode = 0; mode < 2; ++mode) {
      hipLaunchKernelGGL(k_ipm_rhs, grid, blk, 0, st, a, mode);                     // (rt / tn: mode 0 by k_ipm_resid, mode 1 by k_ipm_dir's mode 0)
      IPM_DBG("rhs");
      { int steps = 0; if ((e = ipm_nsolve<W>(S, a, st, &steps, mode == 1 || (check_pred_env >= 0 ? check_pred_env != 0 : undone > 0))) != hipSuccess) return e; refine = std::max(refine, steps); }
      IPM_DBG("nsolve");
      if (it == 1 && trace && mode == 0) {
        ipm_dump("rhs", a.w.rhs, a.P.m, a.w.Bp, trace - 1, st); ipm_dump("dy", a.w.dy, a.P.m, a.w.Bp, trace - 1, st);
        ipm_dump("rt", a.w.rt, (size_t)a.P.n + a.P.m, a.w.Bp, trace - 1, st);
      }
      if (trace && g_ipm_debug) {                       // development: relative residual of the Newton system for the traced lane
        if (a.P.K > 0) { hipLaunchKernelGGL(k_ipm_wide_aty, grid, blk, 0, st, a, (const double *)a.w.dy); hipLaunchKernelGGL(k_ipm_wide_aty_finish, lanes, dim3(64 * kFinW), 0, st, a); }
        hipLaunchKernelGGL(k_ipm_ref_cols, grid, blk, 0, st, a);
        hipLaunchKernelGGL(k_ipm_ref_rows, grid, blk, 0, st, a);
        std::vector<double> hq(a.P.m), hr(a.P.m), hd(a.P.m);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy2D(hq.data(), 8, a.w.q + (trace - 1), a.w.Bp * 8, 8, a.P.m, hipMemcpyDeviceToHost);
        (void)hipMemcpy2D(hr.data(), 8, a.w.rhs + (trace - 1), a.w.Bp * 8, 8, a.P.m, hipMemcpyDeviceToHost);
        (void)hipMemcpy2D(hd.data(), 8, a.w.dy + (trace - 1), a.w.Bp * 8, 8, a.P.m, hipMemcpyDeviceToHost);
        double nq = 0, nr = 0, nd = 0;
        for (int i = 0; i < a.P.m; ++i) { nq += hq[i] * hq[i]; nr += hr[i] * hr[i]; nd += hd[i] * hd[i]; }
        fprintf(stderr, "[ipm]   mode %d: |rhs| %.3e |dy| %.3e |rhs - N dy| / |rhs| %.3e\n", mode, sqrt(nr), sqrt(nd), sqrt(nq) / fmax(sqrt(nr), 1e-300));
      }
      if (a.P.K > 0) { hipLaunchKernelGGL(k_ipm_wide_aty, grid, blk, 0, st, a, (const double *)a.w.dy); hipLaunchKernelGGL(k_ipm_wide_aty_finish, lanes, dim3(64 * kFinW), 0, st, a); }
      hipLaunchKernelGGL(k_ipm_dir, grid, blk, 0, st, a, mode);
      hipLaunchKernelGGL(k_ipm_steps, lanes, dim3(64 * kFinW), 0, st, a, mode);
    }
    IPM_DBG("direction");
    hipLaunchKernelGGL(k_ipm_update, grid, blk, 0, st, a);
    IPM_DBG("update");
  }
  I->last_repacks = repacks;
  *newton = std::min(it - 1, a.max_it);
  IPM_DBG("loop");
  if (trace) {
    std::vector<int> hs(a.w.Bp), hi(a.w.Bp);
    (void)hipMemcpy(hs.data(), a.w.state, a.w.Bp * sizeof(int), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hi.data(), a.w.iters, a.w.Bp * sizeof(int), hipMemcpyDeviceToHost);
    fprintf(stderr, "[ipm] steps taken back: %d\n", undone);
    fprintf(stderr, "[ipm] lanes (state:newton iterations):");
    for (int q = 0; q < a.w.B; ++q) fprintf(stderr, " %d:%d", hs[q], hi[q]);
    fprintf(stderr, "\n");
  }
  // Per scenario: what this method solved is exported (ctrl.done = 1, x+ / y+ in the scenario-major workspace) and stays solved; the
  // scenarios it gave up on (state 2: breakdown, steps that stay tiny - an LP without a solution -, the iteration limit, free columns)
  // are untouched, and the caller hands exactly those to the PDHG forms, which carry the certificates (stream_solve_locked).
  *all_solved = I->counts_host[1] >= a.w.B;
  *n_solved = I->counts_host[1];
  hipLaunchKernelGGL(k_ipm_export, grid, blk, 0, st, a);        // (lanes exported before a packing are in states 4 / 6; nothing to do: returns)
  IPM_DBG("export");
  return hipGetLastError();
}

hipError_t ipm_run(StreamSolver *S, StreamArgs &sa, hipStream_t st, bool *all_solved, int *newton, int *n_solved) {
  *all_solved = false; *newton = 0; *n_solved = 0;
  IpmState *I = S->ipm;
  if (!I) return hipSuccess;
  hipError_t e = ipm_workspace(I, sa.b.B);
  if (e != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }      // no memory for this form: the PDHG forms run
  IpmArgs a{};
  a.P = I->P; a.w = I->w; a.sw = sa.W; a.opt = sa.opt; a.it = 0;
  a.max_it = getenv("DSP_IPM_MAXIT") ? std::max(1, std::min(atoi(getenv("DSP_IPM_MAXIT")), kIpmMaxNewton)) : kIpmMaxNewton;
  a.max_it = std::max(1, std::min(a.max_it, sa.opt.max_iter));            // (dsp_options::max_iter caps the Newton iterations too)
  a.reg = getenv("DSP_IPM_REG") ? atof(getenv("DSP_IPM_REG")) : 0.0;
  a.step = getenv("DSP_IPM_STEP") ? atof(getenv("DSP_IPM_STEP")) : 0.99;
  a.step_blocked = getenv("DSP_IPM_STEP_BLOCKED") ? atof(getenv("DSP_IPM_STEP_BLOCKED")) : 0.9;
  a.step_thr = getenv("DSP_IPM_STEP_THR") ? atof(getenv("DSP_IPM_STEP_THR")) : 0.5;      // (sweep: profiles/r69a_ipm_step_sweep.log)
  a.sigmin = getenv("DSP_IPM_SIGMIN") ? atof(getenv("DSP_IPM_SIGMIN")) : 0.05;
  a.thcap = getenv("DSP_IPM_THCAP") ? atof(getenv("DSP_IPM_THCAP")) : 1e11;
  a.reftol = getenv("DSP_IPM_REFTOL") ? atof(getenv("DSP_IPM_REFTOL")) : 1e-6;
  a.reftol_end = getenv("DSP_IPM_REFTOL_END") ? atof(getenv("DSP_IPM_REFTOL_END")) : 1e-6;
  a.reftol_strict = 1e-8; a.reftol_strict_end = 1e-11;
  a.max_undo = getenv("DSP_IPM_MAX_UNDO") ? atoi(getenv("DSP_IPM_MAX_UNDO")) : 3;
  return I->P.W == 6 ? ipm_loop<6>(S, a, st, all_solved, newton, n_solved) : ipm_loop<8>(S, a, st, all_solved, newton, n_solved);
}

}  // namespace dsp
