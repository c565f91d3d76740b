// dsp_stream_lane.hip — the LANE-PER-SCENARIO form of the HBM-resident streaming PDLP (gfx950), round 4.
//
// Same algorithm, same check sums, same restart / termination decision (control_decide, dsp_stream.hpp) as dsp_stream.hip; what
// changes is who does what.  The batch lives scenario-minor - group g, element e, lane s at ((g (len + 1) + e) 64 + s) - and lane
// s of a wave owns scenario s of its group: it walks through a tile of consecutive rows and columns alone (dsp_lane_tile.hpp), the
// matrix, the shared bounds and every index come as per-element RECORDS (one 16-byte-per-lane load per unit, parked in LDS and read
// back as broadcasts), both products gather from lane-private ring buffers in LDS, and no lane ever needs another lane's value.  A workgroup is 4 waves = 4 neighbouring tiles that meet once, at the end, to
// add their partial sums (of A^T y for the long columns; of the check sums at a check).  Plan and applicability: dsp_lane_plan.hpp.
//
// Per plain iteration, two launches:
//     k_lane_long<0>   one workgroup per (long column, group): its A^T y from the workgroups' partial sums, its primal step
//     k_lane<.., 0>    everything else: reads x, x0, c, y, y0, writes x, y (other buffer) = 4 n + 3 m doubles per scenario
// Per check (every `check_every` iterations): k_lane_long<1>, k_lane<.., 1> (x+, y+, residual and row sums), k_lane<.., 2>
// (reduced costs), k_lane_long<2>, k_lane_sum, k_lane_decide, k_lane_apply - all on the device; the host enqueues and polls the
// finished-counter every few periods, exactly as before.  Initialisation and results go through the scenario-major workspace of
// dsp_stream.hip (k_init, k_init_control, k_finalize) and two transposing kernels.  A solve of more than 64 scenarios runs in
// PHASES (lane_run): when enough scenarios have finished, the rest goes through the scenario-major workspace into fewer groups.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <vector>

#include "dsp_lane_plan.hpp"
#include "dsp_lane_tile.hpp"
#include "dsp_stream.hpp"

namespace dsp {

namespace {

constexpr int kLaneWaves = 4;           // waves (= tiles) per workgroup (8 measured no better: profiles/r40j_lane_variants.log)
// rows / columns per unit of a tile's walk (8 - where a unit's records would still fit one 16-byte-per-lane load - measured no better
// with one or two register sets: profiles/r40e_, r41a_lane_variants.log)
constexpr int lane_ch(int, int, int) { return 4; }
constexpr int kLaneNQ = 16;             // check-sum slots per (group, workgroup)
static_assert(kLaneTileInts == 16, "LaneTile::run reads a tile's entry as LaneVecI<16>");

struct LaneTiling {                     // device copy of one tiling (dsp_lane_plan.hpp: HostLaneTiles)
  int ntile = 0, nwg = 0, ring = 0, rows_per_tile = 0;
  const int32_t *tiles = nullptr, *units = nullptr;
};

struct LaneWork {                       // lane-layout workspace of one solve
  int B = 0, G = 0, nwg_cap = 0;
  bool per_scenario_bounds = false, qp = false;
  double *xA = nullptr, *xB = nullptr, *x0 = nullptr, *c = nullptr, *xp = nullptr, *lb = nullptr, *ub = nullptr;          // [G][n + 1][64]
  double *yA = nullptr, *yB = nullptr, *y0 = nullptr, *yp = nullptr, *rlo = nullptr, *rhi = nullptr, *kap = nullptr;      // [G][m + 1][64]
  double *xbl = nullptr, *xpl = nullptr, *lbl = nullptr, *ubl = nullptr;                                                   // [G][NLP][64]
  double *lpart = nullptr;              // [G][nwg][NLP][64]
  double *partial = nullptr;            // [G][nwg + NLP][16][64]
  double *acc = nullptr;                // [G][16][64]
  double *tau = nullptr, *sig = nullptr;
  int *k = nullptr, *done = nullptr, *mode = nullptr;                                                                       // [G][64]
  int *flag = nullptr;
  std::vector<void *> allocs;
};

}  // namespace

struct LaneState {
  HostLanePlan plan;
  LaneProblem P{};                      // device pointers of the records (tiles / units filled per solve)
  char *crec = nullptr, *rrec = nullptr;   // the records (mutable: shared bounds are written into them per solve)
  int rec_ring = 0;                     // the ring the records' indices are packed for (dsp_lane_plan.hpp: pack_lane_records)
  const int32_t *long_id = nullptr;
  const uint8_t *is_long = nullptr;     // [n]
  std::vector<void *> allocs;
  std::map<long long, LaneTiling> tilings;   // by rows per tile (+ the ring limits of the planner)
  LaneWork W;
  int *flag_host = nullptr;
  int *sid = nullptr;                   // [sid_cap] slot -> scenario map of a compacted phase
  int sid_cap = 0;
  hipStream_t stream = nullptr;         // the iteration loop's own stream (graph replays)
};

namespace {

struct LaneArgs {
  LaneProblem P;
  int NLP, nwg, nslot, B, kofs, iters, rrec_stride, ral_off;   // B: lane slots in use (scenarios of this phase of the solve)
  int sums_only;                        // k_lane_apply: leave the iterate alone, only the long columns' partial sums of A^T y
  const int *sid;                       // slot -> scenario of the batch (nullptr: identity): after finished scenarios were dropped
  const int32_t *long_id;
  const double *x_in, *y_in, *x0, *c, *y0;
  double *x_out, *y_out;
  const double *lb, *ub, *rlo, *rhi, *kap;
  double *xbl, *xpl;
  const double *lbl, *ubl;
  double *xp, *yp;
  double *lpart, *partial, *acc;
  double *tau, *sig;
  int *k, *done, *mode;
  double *x0w, *y0w;                    // anchors (written at a restart)
  double *xa, *ya;                      // buffer A: where k_lane_apply leaves the iterate
  const double *col_scale;
  StreamCtrl *ctrl;
  int *ndone;
  dsp_options opt;
  double eta;
};

template <class T>
hipError_t lane_up(std::vector<void *> &allocs, const std::vector<T> &v, const T **out) {
  void *d = nullptr;
  hipError_t e = hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T));
  if (e != hipSuccess) return e;
  allocs.push_back(d);
  if (!v.empty()) { e = hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); if (e != hipSuccess) return e; }
  *out = reinterpret_cast<const T *>(d);
  return hipSuccess;
}

// ---- layout changes ---------------------------------------------------------------------------------------------------------------
// scenario-major [B][len] (stride `stride`, 0 = one template for every scenario) -> lane layout [G][len + 1][64]; lanes beyond the
// batch get `fill`.  64 x 64 tiles through LDS: both sides coalesced.
// `sid`: slot -> scenario (nullptr: identity), B = slots in use.
__global__ void __launch_bounds__(256) k_lane_in(const double *__restrict__ src, size_t stride, int len, int B, const int *__restrict__ sid, double fill,
                                                 double *__restrict__ dst) {
  __shared__ double tile[64][65];
  const int g = blockIdx.y, e0 = blockIdx.x * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int slot = g * 64 + r, e = e0 + tx;
    const int s = (slot < B && sid) ? sid[slot] : slot;
    tile[r][tx] = (slot < B && e < len) ? src[(size_t)s * stride + e] : fill;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int e = e0 + r;
    if (e < len) dst[((size_t)g * (len + 1) + e) * 64 + tx] = tile[tx][r];
  }
}

__global__ void __launch_bounds__(256) k_lane_out(const double *__restrict__ src, int len, int B, const int *__restrict__ sid, double *__restrict__ dst) {
  __shared__ double tile[64][65];
  const int g = blockIdx.y, e0 = blockIdx.x * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int e = e0 + r;
    tile[r][tx] = e < len ? src[((size_t)g * (len + 1) + e) * 64 + tx] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int slot = g * 64 + r, e = e0 + tx;
    const int s = (slot < B && sid) ? sid[slot] : slot;
    if (slot < B && e < len) dst[(size_t)s * len + e] = tile[tx][r];
  }
}

// shared (scaled) bounds of the batch - scenario 0's copy of the scenario-major workspace - and the scale factors into the records
// (dsp_lane_tile.hpp: record layout); `off` = byte offset of the (lo, hi) pair, `soff` of the scale factor
__global__ void k_lane_fill_records(char *rec, int stride, int off, int soff, const double *__restrict__ lo, const double *__restrict__ hi,
                                    const double *__restrict__ scale, int len) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= len) return;
  double *b = reinterpret_cast<double *>(rec + (size_t)t * stride + off);
  b[0] = lo[t]; b[1] = hi[t];
  *reinterpret_cast<double *>(rec + (size_t)t * stride + soff) = scale[t];
}

// do the bounds of any scenario differ from scenario 0's - on a short column or on a row?  (The long columns carry per-scenario
// bounds in any case: a family that fixes a design variable per member - the nuclear enumeration - still shares the rest.)
__global__ void k_lane_bounds_differ(const double *__restrict__ lo, const double *__restrict__ hi, int len, int B, const uint8_t *__restrict__ skip, int *flag) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y + 1;
  if (t >= len || s >= B || (skip && skip[t])) return;
  if (lo[(size_t)s * len + t] != lo[t] || hi[(size_t)s * len + t] != hi[t]) atomicOr(flag, 1);
}

// per-scenario bounds of the long columns; control blocks -> per-lane arrays
__global__ void k_lane_setup(LaneArgs a, const double *__restrict__ lbW, const double *__restrict__ ubW) {
  const int g = blockIdx.x, lane = threadIdx.x, slot = g * 64 + lane;
  const bool in = slot < a.B;
  const int s = (in && a.sid) ? a.sid[slot] : slot;
  for (int l = 0; l < a.NLP; ++l) {
    double lo = 0.0, hi = 0.0;
    if (in && l < a.P.nl) { const int j = a.long_id[l]; lo = lbW[(size_t)s * a.P.n + j]; hi = ubW[(size_t)s * a.P.n + j]; }
    const_cast<double *>(a.lbl)[((size_t)g * a.NLP + l) * 64 + lane] = lo;
    const_cast<double *>(a.ubl)[((size_t)g * a.NLP + l) * 64 + lane] = hi;
    a.xbl[((size_t)g * a.NLP + l) * 64 + lane] = 0.0;
    a.xpl[((size_t)g * a.NLP + l) * 64 + lane] = 0.0;
  }
  const size_t at = (size_t)g * 64 + lane;
  if (in) { const StreamCtrl &c = a.ctrl[s]; a.tau[at] = c.tau; a.sig[at] = c.sig; a.k[at] = c.k; a.done[at] = c.done; a.mode[at] = 0; }
  else { a.tau[at] = 0.0; a.sig[at] = 0.0; a.k[at] = 0; a.done[at] = 1; a.mode[at] = 0; }
}

// ---- the tile kernel ----------------------------------------------------------------------------------------------------------------
template <int WC, int WR, int NLP, bool SHARED, bool QP, int MODE>
__global__ void __launch_bounds__(kLaneWaves * 64) k_lane(LaneArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = blockIdx.y, wg = blockIdx.x;
  const int tile = wg * kLaneWaves + wv;
  const int R = a.P.ring_mask + 1;
  constexpr int NRING = MODE == 1 ? 3 : 2;
  double *ring = lds + (size_t)wv * NRING * R * 64;
  char *stage = reinterpret_cast<char *>(lds + (size_t)kLaneWaves * NRING * R * 64) + (size_t)wv * kLaneStageBytes;
  const size_t gn = (size_t)g * (a.P.n + 1) * 64, gm = (size_t)g * (a.P.m + 1) * 64, gl = (size_t)g * NLP * 64;
  LaneGroup G;
  G.x_in = a.x_in + gn; G.y_in = a.y_in + gm; G.x0 = a.x0 + gn; G.c = a.c + gn; G.y0 = a.y0 + gm;
  G.x_out = a.x_out + gn; G.y_out = a.y_out + gm;
  G.lb = SHARED ? nullptr : a.lb + gn; G.ub = SHARED ? nullptr : a.ub + gn;
  G.rlo = SHARED ? nullptr : a.rlo + gm; G.rhi = SHARED ? nullptr : a.rhi + gm;
  G.kap = QP ? a.kap + gm : nullptr;
  G.xbl = a.xbl + gl; G.xpl = a.xpl + gl; G.xp = a.xp + gn; G.yp = a.yp + gm;
  LaneScalars sc;
  const size_t at = (size_t)g * 64 + lane;
  sc.tau = a.tau[at]; sc.sig = a.sig[at];
  const int kk = a.k[at];
  sc.done = a.done[at] != 0;
  sc.oml = 1.0 / (double)(kk + a.kofs + 3);
  asm volatile("" ::: "memory");               // the per-lane scalars are requested before the tile's first rows
  LaneOut<NLP> out;
#pragma unroll
  for (int l = 0; l < NLP; ++l) out.lp[l] = 0.0;
#pragma unroll
  for (int q = 0; q < 13; ++q) out.v[q] = 0.0;
  // Lanes without work - beyond the batch, or whose scenario has finished - sit the walk out: the EXEC mask keeps their loads and
  // stores off the memory system - they only help to fetch and park the unit's records (first version: every lane of every wave walked, a batch of 1 moved the bytes of a batch of 64 -
  // 48 us per iteration for 1, 16 or 64 scenarios alike, profiles/r40g_lane_pmc_summary_B16.csv - and a solve paid for its finished
  // scenarios until the last one was through).  Their sums stay zero; a group with no live lane skips the walk altogether.
  sc.active = !sc.done;
  const bool any = __builtin_amdgcn_readfirstlane(__any(sc.active ? 1 : 0)) != 0;
#ifdef DSP_LANE_PROBE
  unsigned long long *pw = g_lane_probe + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * kLaneWaves + wv) * kProbeSlots;
  const bool pon = MODE == 0 && lane == 0 && (size_t)(pw - g_lane_probe) < (size_t)(kProbeWaves - 1) * kProbeSlots;
  if (pon) { pw[0] = wall_clock64(); pw[1] = clock64(); pw[2] = 0; unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); pw[3] = ((unsigned long long)xcc << 32) | hw; }
#endif
  if (tile < a.P.ntile && any) LaneTile<WC, WR, NLP, lane_ch(WC, WR, NLP), SHARED, QP, MODE>::run(a.P, G, tile, lane, sc, ring, stage, out);
#ifdef DSP_LANE_PROBE
  if (pon) pw[4] = clock64();
#endif
  // the workgroup's waves add their partial sums in wave order (the one barrier of the launch; the rings are free by then)
  __syncthreads();
#ifdef DSP_LANE_PROBE
  if (pon) pw[5] = clock64();
#endif
  double *red = lds;
  if (MODE != 2) {
#pragma unroll
    for (int l = 0; l < NLP; ++l) red[((size_t)wv * NLP + l) * 64 + lane] = out.lp[l];
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int l = 0; l < NLP; ++l) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kLaneWaves; ++w) t += red[((size_t)w * NLP + l) * 64 + lane];
        a.lpart[(((size_t)g * a.nwg + wg) * NLP + l) * 64 + lane] = t;
      }
    }
  }
  if (MODE != 0) {
    constexpr int q0 = MODE == 1 ? 0 : 8, nq = MODE == 1 ? 8 : 5;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < nq; ++q) red[((size_t)wv * 8 + q) * 64 + lane] = out.v[q0 + q];
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int q = 0; q < nq; ++q) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kLaneWaves; ++w) t += red[((size_t)w * 8 + q) * 64 + lane];
        a.partial[(((size_t)g * a.nslot + wg) * kLaneNQ + q0 + q) * 64 + lane] = t;
      }
    }
  }
#ifdef DSP_LANE_PROBE
  if (pon) { pw[6] = clock64(); pw[7] = wall_clock64(); }
#endif
}

// ---- the long columns: A^T y from the workgroups' partial sums (fixed order), then the column's own step ------------------------------
// LM 0: plain iteration (xbar for the rows, x averaged into the other buffer); LM 1: check iteration (x+, xbar, the column's terms of
// the residual sums); LM 2: the column's reduced cost at y+ (the partial sums k_lane<.., 1> left are those of A^T y+)
constexpr int kLongWaves = 16;         // waves of a k_lane_long / k_lane_sum workgroup: each adds every 16th partial sum
// sum over the workgroups' partial sums p[w * stride] (w = 0 .. count), fixed order: wave v takes w = v, v + 16, ..., sixteen loads in
// flight at a time (one dependent load per addition was a memory round trip each - the partial sums were written by other XCDs a
// moment ago: 60 us per launch at 470 workgroups), then the waves' sums in wave order.  Every thread of the workgroup must call it;
// the result is valid in wave 0.
__device__ __forceinline__ double lane_sum_partials(const double *p, size_t stride, int count, double (*red)[64]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int U = 16;
  double t = 0.0;
  for (int w = wv; w < count; w += U * kLongWaves) {
    double v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const int ww = w + k * kLongWaves; v[k] = p[(size_t)min(ww, count - 1) * stride + lane]; }
#pragma unroll
    for (int k = 0; k < U; ++k) t += (w + k * kLongWaves < count) ? v[k] : 0.0;
  }
  red[wv][lane] = t;
  __syncthreads();
  double r = 0.0;
  if (wv == 0) {
#pragma unroll
    for (int k = 0; k < kLongWaves; ++k) r += red[k][lane];
  }
  return r;
}

template <int LM>
__global__ void __launch_bounds__(kLongWaves * 64) k_lane_long(LaneArgs a) {
  __shared__ double red[kLongWaves][64];
  const int l = blockIdx.x, g = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int NLP = a.NLP;
  const int j = a.long_id[l];
  const size_t at = ((size_t)g * (a.P.n + 1) + j) * 64 + lane, al = ((size_t)g * NLP + l) * 64 + lane, as = (size_t)g * 64 + lane;
  // the column's own data is requested ahead of the partial sums (wave 0 uses it)
  double lb = 0.0, ub = 0.0, cj = 0.0, x = 0.0, x0 = 0.0, tau = 0.0, xpv = 0.0;
  int kk = 0;
  bool done = true;
  if (wv == 0) {
    lb = a.lbl[al]; ub = a.ubl[al]; cj = a.c[at]; done = a.done[as] != 0;
    if (LM == 2) xpv = a.xp[at];
    else { x = a.x_in[at]; x0 = a.x0[at]; tau = a.tau[as]; kk = a.k[as]; }
  }
  const double aty = lane_sum_partials(a.lpart + ((size_t)g * a.nwg * NLP + l) * 64, (size_t)NLP * 64, a.nwg, red);
  if (wv != 0) return;
  double *slot = a.partial + (((size_t)g * a.nslot + a.nwg + l) * kLaneNQ) * 64 + lane;
  if (LM == 2) {
    const double xp = xpv;
    const double rc = cj - aty;
    const double lp = lane_finite(lb) ? fmax(rc, 0.0) : 0.0;
    const double lm = lane_finite(ub) ? fmax(-rc, 0.0) : 0.0;
    const double dr = (rc - lp + lm) / a.col_scale[j];
    slot[8 * 64] = dr * dr;
    slot[9 * 64] = cj * xp;
    slot[10 * 64] = lp * lane_fin0(lb) - lm * lane_fin0(ub);
    slot[11 * 64] = fabs(cj * xp);
    slot[12 * 64] = fabs(rc - lp + lm) * fabs(xp);
    return;
  }
  const double xp = lane_clamp(fma(-tau, cj - aty, x), lb, ub);
  const double tt = 2.0 * xp - x;
  a.xbl[al] = tt;
  if (LM == 0) {
    const double oml = 1.0 / (double)(kk + a.kofs + 3);
    a.x_out[at] = fma(oml, x0 - tt, tt);
  } else {
    a.xpl[al] = xp;
    if (!done) a.xp[at] = xp;
    const double dx = xp - x, d0 = xp - x0;
#pragma unroll
    for (int q = 0; q < 8; ++q) slot[q * 64] = q == 0 ? dx * dx : (q == 6 ? d0 * d0 : 0.0);
  }
}

// ---- check: sums over the workgroups, decision, apply ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(kLongWaves * 64) k_lane_sum(LaneArgs a) {
  __shared__ double red[kLongWaves][64];
  const int q = blockIdx.x, g = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double t = lane_sum_partials(a.partial + ((size_t)g * a.nslot * kLaneNQ + q) * 64, (size_t)kLaneNQ * 64, a.nslot, red);
  if (wv == 0) a.acc[((size_t)g * kLaneNQ + q) * 64 + lane] = t;
}

__global__ void __launch_bounds__(64) k_lane_decide(LaneArgs a) {
  const int g = blockIdx.x, lane = threadIdx.x, slot = g * 64 + lane;
  const size_t at = (size_t)g * 64 + lane;
  if (slot >= a.B || a.done[at]) return;
  const int s = a.sid ? a.sid[slot] : slot;
  double acc[kLaneNQ];
#pragma unroll
  for (int q = 0; q < 13; ++q) acc[q] = a.acc[((size_t)g * kLaneNQ + q) * 64 + lane];
  StreamCtrl c = a.ctrl[s];
  const int mode = control_decide(acc, c, a.opt, a.eta, a.iters);
  a.ctrl[s] = c;
  a.tau[at] = c.tau; a.sig[at] = c.sig; a.k[at] = c.k; a.done[at] = c.done; a.mode[at] = mode;
  if (c.done) atomicAdd(a.ndone, 1);
  else if (c.suspect) a.ndone[1] = 1;               // the host runs the certificate sequence (stream_certify) at its next look
}

// Halpern step or restart after a check, tile by tile like k_lane; leaves the iterate in buffer A (xa, ya; may be the buffer it reads:
// every element is read before it is written, by the same lane) and the long columns' partial sums of A^T y_new.
// Scenarios that finished (now or earlier) are left alone.
template <int NLP>
__global__ void __launch_bounds__(kLaneWaves * 64) k_lane_apply(LaneArgs a) {
  __shared__ double red[kLaneWaves][NLP][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = blockIdx.y, wg = blockIdx.x, tile = wg * kLaneWaves + wv;
  const size_t gn = (size_t)g * (a.P.n + 1) * 64, gm = (size_t)g * (a.P.m + 1) * 64, as = (size_t)g * 64 + lane;
  const uint32_t l8 = (uint32_t)lane * 8u;
  const bool done = a.done[as] != 0;
  const bool restart = a.mode[as] == 1;
  const bool keep = a.sums_only != 0;                 // (entering a solve or a compacted phase: the iterate stays, only the sums are wanted)
  const double oml = 1.0 / (double)(a.k[as] + 2);
  double lp[NLP];
#pragma unroll
  for (int l = 0; l < NLP; ++l) lp[l] = 0.0;
  if (tile < a.P.ntile && !done) {                   // (lanes whose scenario has finished: no loads, no stores, zero sums)
    const LaneVecI<8> tp = ldu(reinterpret_cast<const LaneVecI<8> *>(a.P.tiles + (size_t)tile * kLaneTileInts));
    const int i0 = tp.v[0], i1 = tp.v[1], j0 = tp.v[2], j1 = tp.v[3];
    constexpr int U = 8;
    for (int b = j0; b < j1 && !keep; b += U) {
      double xp[U], x[U], x0[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = min(b + k, j1 - 1);
        xp[k] = lane_ld(a.xp + gn, j, l8); x[k] = lane_ld(a.x_in + gn, j, l8); x0[k] = lane_ld(a.x0 + gn, j, l8);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = (b + k < j1 && !done) ? b + k : a.P.n;                       // finished / beyond the tile: the sink row
        const double tt = 2.0 * xp[k] - x[k];
        lane_st(a.xa + gn, j, l8) = restart ? xp[k] : fma(oml, x0[k] - tt, tt);
        lane_st(a.x0w + gn, (restart && b + k < j1 && !done) ? b + k : a.P.n, l8) = xp[k];
      }
    }
    for (int b = i0; b < i1; b += U) {
      double yp[U], y[U], y0[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int i = min(b + k, i1 - 1);
        yp[k] = lane_ld(a.yp + gm, i, l8); y[k] = lane_ld(a.y_in + gm, i, l8); y0[k] = lane_ld(a.y0 + gm, i, l8);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const bool live = b + k < i1;
        const int i = (live && !done && !keep) ? b + k : a.P.m;
        const double tt = 2.0 * yp[k] - y[k];
        const double yn = keep ? y[k] : (restart ? yp[k] : fma(oml, y0[k] - tt, tt));
        lane_st(a.ya + gm, i, l8) = yn;
        lane_st(a.y0w + gm, (restart && live && !done && !keep) ? b + k : a.P.m, l8) = yp[k];
        const LaneVecD<NLP> al = ldu(reinterpret_cast<const LaneVecD<NLP> *>(a.P.rrec + (size_t)min(b + k, i1 - 1) * a.rrec_stride + a.ral_off));
        const double yw = live ? yn : 0.0;
#pragma unroll
        for (int l = 0; l < NLP; ++l) lp[l] = fma(al.v[l], yw, lp[l]);
      }
    }
  }
#pragma unroll
  for (int l = 0; l < NLP; ++l) red[wv][l][lane] = lp[l];
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int l = 0; l < NLP; ++l) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < kLaneWaves; ++w) t += red[w][l][lane];
      a.lpart[(((size_t)g * a.nwg + wg) * NLP + l) * 64 + lane] = t;
    }
  }
}

using LaneKernel = void (*)(LaneArgs);
template <int WC, int WR, int NLP, bool SHARED, bool QP>
LaneKernel lane_mode(int mode) {
  return mode == 0 ? &k_lane<WC, WR, NLP, SHARED, QP, 0> : (mode == 1 ? &k_lane<WC, WR, NLP, SHARED, QP, 1> : &k_lane<WC, WR, NLP, SHARED, QP, 2>);
}
template <int WC, int WR, int NLP>
LaneKernel lane_flags(bool shared, bool qp, int mode) {
  // soft rows (convex QP) are instantiated for the shared-bounds form only: a QP batch with per-scenario bounds keeps the two-launch form
  if (shared) return qp ? lane_mode<WC, WR, NLP, true, true>(mode) : lane_mode<WC, WR, NLP, true, false>(mode);
  return qp ? nullptr : lane_mode<WC, WR, NLP, false, false>(mode);
}
LaneKernel lane_pick(int wc, int wr, int nlp, bool shared, bool qp, int mode) {
  if (nlp == 4) {
    if (wc == 4 && wr == 4) return lane_flags<4, 4, 4>(shared, qp, mode);
    if (wc == 4 && wr == 8) return lane_flags<4, 8, 4>(shared, qp, mode);
    return lane_flags<8, 8, 4>(shared, qp, mode);
  }
  if (wc == 4 && wr == 4) return lane_flags<4, 4, 8>(shared, qp, mode);
  if (wc == 4 && wr == 8) return lane_flags<4, 8, 8>(shared, qp, mode);
  return lane_flags<8, 8, 8>(shared, qp, mode);
}

}  // namespace

// ---- host side ------------------------------------------------------------------------------------------------------------------------
hipError_t lane_create(const HostCSR &A_scaled, const HostCSR &AT_scaled, StreamSolver *S) {
  S->lane = nullptr;
  if (getenv("DSP_STREAM_NO_LANE") && atoi(getenv("DSP_STREAM_NO_LANE"))) return hipSuccess;
  // a group's block of one vector is addressed with 32-bit byte offsets (dsp_lane_tile.hpp: lane_ld)
  if ((uint64_t)(std::max(A_scaled.n, A_scaled.m) + 1) * 512ull >= (1ull << 32)) return hipSuccess;
  LaneState *L = new LaneState();
  L->plan = build_lane_plan(A_scaled, AT_scaled, std::max(4096, A_scaled.m / 4));
  const HostLanePlan &H = L->plan;
  // applicable when some tiling of a few dozen rows schedules within the ring budget (a matrix that is not banded does not)
  if (!H.ok || !build_lane_tiles(H, 64, lane_ch(H.WC, H.WR, H.NLP)).ok) { delete L; return hipSuccess; }
  hipError_t e;
  L->P.n = H.n; L->P.m = H.m; L->P.nl = H.nl;
  std::vector<uint8_t> is_long(H.n, 0);
  for (int j : H.long_id) is_long[j] = 1;
  std::vector<char> crec, rrec;
  L->rec_ring = 16;                                       // the usual ring; a tiling with another one repacks (lane_records_for)
  pack_lane_records(H, L->rec_ring, crec, rrec);
  const char *cdev = nullptr, *rdev = nullptr;
  if ((e = lane_up(L->allocs, crec, &cdev)) != hipSuccess || (e = lane_up(L->allocs, rrec, &rdev)) != hipSuccess ||
      (e = lane_up(L->allocs, H.long_id, &L->long_id)) != hipSuccess || (e = lane_up(L->allocs, is_long, &L->is_long)) != hipSuccess) {
    for (void *p : L->allocs) (void)hipFree(p);
    delete L;
    return e;
  }
  L->crec = const_cast<char *>(cdev); L->rrec = const_cast<char *>(rdev);
  L->P.crec = cdev; L->P.rrec = rdev;
  S->lane = L;
  return hipSuccess;
}

void lane_destroy(StreamSolver *S) {
  LaneState *L = S->lane;
  if (!L) return;
  for (void *p : L->allocs) (void)hipFree(p);
  for (void *p : L->W.allocs) (void)hipFree(p);
  if (L->flag_host) (void)hipHostFree(L->flag_host);
  if (L->sid) (void)hipFree(L->sid);
  if (L->stream) (void)hipStreamDestroy(L->stream);
  delete L;
  S->lane = nullptr;
}

// the records' gather indices are ring offsets: packed again when a tiling uses another ring than the last one did
// (*repacked: the records are fresh from the host plan - shared bounds +-inf, scale factors 1 - and need k_lane_fill_records again)
static hipError_t lane_records_for(LaneState *L, int ring, bool *repacked) {
  *repacked = false;
  if (ring == L->rec_ring) return hipSuccess;
  *repacked = true;
  std::vector<char> crec, rrec;
  pack_lane_records(L->plan, ring, crec, rrec);
  hipError_t e;
  if ((e = hipMemcpy(L->crec, crec.data(), crec.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
  if ((e = hipMemcpy(L->rrec, rrec.data(), rrec.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
  L->rec_ring = ring;
  return hipSuccess;
}

static hipError_t lane_tiling(LaneState *L, int rows_per_tile, LaneTiling **out) {
  // DSP_LANE_RING_MAX (development): largest ring the planner may use (8: less LDS per wave, more resident waves, emptier units)
  const int ring_max = getenv("DSP_LANE_RING_MAX") ? atoi(getenv("DSP_LANE_RING_MAX")) : kLaneMaxRing;
  // from 16 slots: with 8 the units of these matrices are 60 % full, with 16 80 - 90 % (17 - 40 % fewer units; r40x_lane_variants.log)
  const int ring_min = getenv("DSP_LANE_RING_MIN") ? atoi(getenv("DSP_LANE_RING_MIN")) : 16;
  const long long key = (long long)rows_per_tile * 4096 + std::min(ring_max, 63) * 64 + std::min(ring_min, 63);
  auto it = L->tilings.find(key);
  if (it == L->tilings.end()) {
    const HostLaneTiles T = build_lane_tiles(L->plan, rows_per_tile, lane_ch(L->plan.WC, L->plan.WR, L->plan.NLP), ring_min, ring_max);
    LaneTiling D;
    if (T.ok) {
      hipError_t e;
      if ((e = lane_up(L->allocs, T.tiles, &D.tiles)) != hipSuccess || (e = lane_up(L->allocs, T.units, &D.units)) != hipSuccess) return e;
      D.ntile = T.ntile; D.nwg = (T.ntile + kLaneWaves - 1) / kLaneWaves; D.ring = T.ring; D.rows_per_tile = rows_per_tile;
    }
    it = L->tilings.emplace(key, D).first;
  }
  *out = &it->second;
  return hipSuccess;
}

static hipError_t lane_workspace(LaneState *L, int B, int nwg, bool per_scenario_bounds, bool qp) {
  LaneWork &W = L->W;
  const int G = (B + 63) / 64;
  if (G <= W.G && nwg <= W.nwg_cap && (!per_scenario_bounds || W.per_scenario_bounds) && (!qp || W.qp)) { W.B = B; return hipSuccess; }
  for (void *p : W.allocs) (void)hipFree(p);
  W = LaneWork{};
  const size_t n1 = (size_t)L->P.n + 1, m1 = (size_t)L->P.m + 1, NLP = L->plan.NLP;
  hipError_t e;
  auto alloc = [&](size_t count, size_t elem, void **out) {
    hipError_t r = hipMalloc(out, std::max<size_t>(count * elem, 8));
    if (r == hipSuccess) { W.allocs.push_back(*out); r = hipMemset(*out, 0, std::max<size_t>(count * elem, 8)); }
    return r;
  };
  double **colv[] = {&W.xA, &W.xB, &W.x0, &W.c, &W.xp};
  double **rowv[] = {&W.yA, &W.yB, &W.y0, &W.yp};
  for (double **p : colv) if ((e = alloc((size_t)G * n1 * 64, 8, (void **)p)) != hipSuccess) return e;
  for (double **p : rowv) if ((e = alloc((size_t)G * m1 * 64, 8, (void **)p)) != hipSuccess) return e;
  if (per_scenario_bounds) {
    double **cb[] = {&W.lb, &W.ub}, **rb[] = {&W.rlo, &W.rhi};
    for (double **p : cb) if ((e = alloc((size_t)G * n1 * 64, 8, (void **)p)) != hipSuccess) return e;
    for (double **p : rb) if ((e = alloc((size_t)G * m1 * 64, 8, (void **)p)) != hipSuccess) return e;
  }
  if (qp && (e = alloc((size_t)G * m1 * 64, 8, (void **)&W.kap)) != hipSuccess) return e;
  double **lv[] = {&W.xbl, &W.xpl, &W.lbl, &W.ubl};
  for (double **p : lv) if ((e = alloc((size_t)G * NLP * 64, 8, (void **)p)) != hipSuccess) return e;
  if ((e = alloc((size_t)G * nwg * NLP * 64, 8, (void **)&W.lpart)) != hipSuccess) return e;
  if ((e = alloc((size_t)G * (nwg + NLP) * kLaneNQ * 64, 8, (void **)&W.partial)) != hipSuccess) return e;
  if ((e = alloc((size_t)G * kLaneNQ * 64, 8, (void **)&W.acc)) != hipSuccess) return e;
  if ((e = alloc((size_t)G * 64, 8, (void **)&W.tau)) != hipSuccess || (e = alloc((size_t)G * 64, 8, (void **)&W.sig)) != hipSuccess) return e;
  if ((e = alloc((size_t)G * 64, 4, (void **)&W.k)) != hipSuccess || (e = alloc((size_t)G * 64, 4, (void **)&W.done)) != hipSuccess ||
      (e = alloc((size_t)G * 64, 4, (void **)&W.mode)) != hipSuccess) return e;
  if ((e = alloc(1, 4, (void **)&W.flag)) != hipSuccess) return e;
  if (!L->flag_host && (e = hipHostMalloc((void **)&L->flag_host, sizeof(int))) != hipSuccess) return e;
  W.B = B; W.G = G; W.nwg_cap = nwg; W.per_scenario_bounds = per_scenario_bounds; W.qp = qp;
  return hipSuccess;
}

// Runs the iteration loop in the lane form on the state k_init / k_init_control left in the scenario-major workspace `a.W`, and leaves
// x+, y+ and the control blocks there for k_finalize.  *used = false: not applicable to this batch (the caller goes on with its own forms).
// `only`: the scenarios to run (the rest of the batch is finished: the interior-point form solved it) - the first phase packs exactly those.
hipError_t lane_run(StreamSolver *S, StreamArgs &a, hipStream_t st, int *periods_run, bool *used, const std::vector<int> *only) {
  *used = false;
  LaneState *L = S->lane;
  if (!L) return hipSuccess;
  const int B = a.b.B, n = L->P.n, m = L->P.m, NLP = L->plan.NLP;
  // Small batches stay with the workgroup-per-tile form of round 3 (dsp_stream.hip: k_fused_pre): a lane's walk through its tile costs
  // the same instruction stream whatever the batch - 36 us per iteration for 1 .. 32 scenarios, against 10 us (1 scenario) and
  // 28 - 32 us (16) there; from 32 scenarios on the lane form is ahead (64: 42 vs 90 us, 256: 165 vs 407 us; profiles/r40h_lane_rates.log,
  // r41a_lane_variants.log).  DSP_LANE_MIN_B: the threshold (development).
  const int min_b = getenv("DSP_LANE_MIN_B") ? atoi(getenv("DSP_LANE_MIN_B")) : 32;
  // a subset worth packing: as many scenarios as the form wants of a batch; fewer go to the other forms, whose kernels leave the finished
  // scenarios of the batch alone (one scenario: 10 us per iteration there against 36 us for a lane group here)
  const bool subset = only && (int)only->size() >= min_b;
  if (only && !subset) return hipSuccess;
  if (!subset && B < min_b) return hipSuccess;
  const bool qp = a.b.row_compliance != nullptr;
  hipError_t e;
  const int rows_env = getenv("DSP_LANE_ROWS") ? atoi(getenv("DSP_LANE_ROWS")) : 0;          // rows per tile (development; read per solve)
  const int waves_env = getenv("DSP_LANE_WAVES") ? atoi(getenv("DSP_LANE_WAVES")) : 0;
  const int graph_env = getenv("DSP_LANE_GRAPH") ? atoi(getenv("DSP_LANE_GRAPH")) : 1;
  // DSP_LANE_COMPACT=0: keep every scenario's lane to the end of the solve (measurement)
  const bool compact = !(getenv("DSP_LANE_COMPACT") && atoi(getenv("DSP_LANE_COMPACT")) == 0);
  const int ch = lane_ch(L->plan.WC, L->plan.WR, NLP);
  const int CREC = lane_crec(L->plan.WC), RREC = lane_rrec(L->plan.WR, NLP);
  const int C = a.opt.check_every > 0 ? a.opt.check_every : 64;
  const int max_periods = (a.opt.max_iter + C - 1) / C;
  const int poll = 4;
  const dim3 tb(256);

  // A solve runs in PHASES.  The first one gives every scenario of the batch a lane; whenever enough scenarios have finished to
  // free a quarter of the groups (at least one), the iterate goes back to the scenario-major workspace and the scenarios still
  // iterating are packed into fewer groups - a new tiling, a new graph: a launch costs what its groups cost, and a batch of year-long
  // LPs finishes over a 4 x range of iteration counts (35 k .. 139 k on the wind + battery family).  `ids`: slot -> scenario of the
  // current phase (empty: identity).
  std::vector<int> ids;
  int nact = B, period = 0;
  if (subset) { ids = *only; nact = (int)ids.size(); }
  S->last_phases = 0;
  bool shared = true, first = true;
  LaneTiling *Tprev = nullptr;            // the tiling the previous phase ran on
  bool count_phase = true;
  for (;;) {
    const int G = (nact + 63) / 64;
    // tiles: one wave per SIMD (1024 tiles) for a single group of 40 scenarios and more (~50 rows per wave), two waves per SIMD (2048
    // tiles) otherwise - fewer scenarios (less to fetch per unit: the second wave hides more than the extra halo rows cost), 2 - 3
    // groups (7 % / 2 % ahead), 8 and more (3 - 4 %).  Four groups: a tie in time (172 vs 172 us, 164 vs 156 - 172 across boxes), and
    // 1024 tiles move 1.01 x the algorithmic bytes against 1.12 x: 1024.  (profiles/r40y_, r40z_, r41a_lane_variants.log,
    // r41l_large_batches.log, r41m_mid_batches.log, r41n_*)
    const int want_waves = waves_env > 0 ? waves_env : (((G == 1 && nact >= 40) || G == 4) ? 1024 : 2048);
    int rows = rows_env > 0 ? rows_env : (int)(((int64_t)m * G + want_waves - 1) / want_waves);
    rows = std::max(rows, 3 * ch);
    rows = (rows + ch - 1) / ch * ch;
    LaneTiling *T = nullptr;
    if ((e = lane_tiling(L, rows, &T)) != hipSuccess) return e;
    // Not applicable (no schedule for this tile size, or the check kernel's three windows + record stages of four waves beyond a CU's
    // LDS - kLaneMaxRing keeps them below: defensive): in the first phase nothing has been touched and the caller's other forms take
    // the batch; a LATER phase - its tile size differs with the number of groups - keeps the tiling the previous phase ran on (the
    // scenario-major workspace holds a consistent iterate; the packed groups run on somewhat larger tiles than they would have got).
    if (T->ntile == 0 || (size_t)kLaneWaves * (3 * (size_t)T->ring * 64 * sizeof(double) + kLaneStageBytes) > 160u * 1024u) {
      if (first) return hipSuccess;
      T = Tprev;
    }
    Tprev = T;
    bool repacked = false;
    if ((e = lane_records_for(L, T->ring, &repacked)) != hipSuccess) return e;
    if ((e = lane_workspace(L, nact, T->nwg, !shared, qp)) != hipSuccess) return e;
    if (first) {
      // bounds: one template for the whole batch (stride 0), or equal apart from the long columns' (checked on the device)
      shared = (!a.b.var_lb || a.b.var_lb_stride == 0) && (!a.b.var_ub || a.b.var_ub_stride == 0) &&
               (!a.b.row_lb || a.b.row_lb_stride == 0) && (!a.b.row_ub || a.b.row_ub_stride == 0);
      if (!shared && B > 1) {
        if ((e = hipMemsetAsync(L->W.flag, 0, sizeof(int), st)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_lane_bounds_differ, dim3((n + 255) / 256, B - 1), dim3(256), 0, st, a.W.lb, a.W.ub, n, B, L->is_long, L->W.flag);
        hipLaunchKernelGGL(k_lane_bounds_differ, dim3((m + 255) / 256, B - 1), dim3(256), 0, st, a.W.rlo, a.W.rhi, m, B, (const uint8_t *)nullptr, L->W.flag);
        if ((e = hipMemcpyAsync(L->flag_host, L->W.flag, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        shared = *L->flag_host == 0;
      } else shared = true;
      if (!shared) {
        if (qp) return hipSuccess;                                     // (no instantiation: the two-launch form takes it)
        if ((e = lane_workspace(L, nact, T->nwg, true, qp)) != hipSuccess) return e;
      }
    }
    LaneWork &W = L->W;
    LaneKernel kern[3];
    for (int mode = 0; mode < 3; ++mode) if (!(kern[mode] = lane_pick(L->plan.WC, L->plan.WR, NLP, shared, qp, mode))) return first ? hipSuccess : hipErrorUnknown;
    *used = true;
    if (count_phase) ++S->last_phases;           // (a phase that only follows a certificate sequence is not a packing: not counted)
    count_phase = true;
    S->last_bytes_per_iteration = (size_t)8 * (shared ? 4 * (size_t)n + 3 * (size_t)m : 6 * (size_t)n + 5 * (size_t)m) + (qp ? 8 * (size_t)m : 0);
    const int *sid = nullptr;
    if (!ids.empty()) {
      if (L->sid_cap < B) {
        if (L->sid) (void)hipFree(L->sid);
        L->sid = nullptr; L->sid_cap = 0;
        if ((e = hipMalloc((void **)&L->sid, (size_t)B * sizeof(int))) != hipSuccess) return e;
        L->sid_cap = B;
      }
      if ((e = hipMemcpyAsync(L->sid, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;           // (`ids` is pageable and changes below)
      sid = L->sid;
    }

    LaneArgs la{};
    la.P = L->P; la.P.tiles = T->tiles; la.P.units = T->units; la.P.ntile = T->ntile; la.P.ring_mask = T->ring - 1;
    la.NLP = NLP; la.nwg = T->nwg; la.nslot = T->nwg + NLP; la.B = nact; la.kofs = 0; la.sums_only = 0; la.sid = sid;
    la.long_id = L->long_id;
    la.x0 = W.x0; la.c = W.c; la.y0 = W.y0; la.lb = W.lb; la.ub = W.ub; la.rlo = W.rlo; la.rhi = W.rhi; la.kap = W.kap;
    la.xbl = W.xbl; la.xpl = W.xpl; la.lbl = W.lbl; la.ubl = W.ubl; la.xp = W.xp; la.yp = W.yp;
    la.lpart = W.lpart; la.partial = W.partial; la.acc = W.acc; la.tau = W.tau; la.sig = W.sig; la.k = W.k; la.done = W.done; la.mode = W.mode;
    la.x0w = W.x0; la.y0w = W.y0; la.xa = W.xA; la.ya = W.yA;
    la.ctrl = a.W.ctrl; la.ndone = a.W.ndone; la.opt = a.opt; la.eta = a.eta; la.col_scale = S->P.col_scale;
    la.rrec_stride = RREC; la.ral_off = L->plan.WR * 12 + 16;
    la.iters = C;

    // ---- in: scenario-major workspace -> lane layout (iterate, anchors, x+ / y+ of the last check, objective, bounds) ----------------
    const dim3 gcol((n + 63) / 64, G), grow((m + 63) / 64, G);
    auto in = [&](const double *src, int len, double *dst) {
      hipLaunchKernelGGL(k_lane_in, len == n ? gcol : grow, tb, 0, st, src, (size_t)len, len, nact, sid, 0.0, dst);
    };
    in(a.W.x, n, W.xA); in(a.W.x0, n, W.x0); in(a.W.c, n, W.c); in(a.W.xp, n, W.xp);
    in(a.W.y, m, W.yA); in(a.W.y0, m, W.y0); in(a.W.yp, m, W.yp);
    if (!shared) { in(a.W.lb, n, W.lb); in(a.W.ub, n, W.ub); in(a.W.rlo, m, W.rlo); in(a.W.rhi, m, W.rhi); }
    if (qp) in(a.W.kap, m, W.kap);
    if (first || repacked) {      // (a.W.lb .. a.W.rhi: scenario 0's scaled bounds in the scenario-major workspace, valid for the whole solve)
      hipLaunchKernelGGL(k_lane_fill_records, dim3((n + 255) / 256), tb, 0, st, L->crec, CREC, L->plan.WC * 12, L->plan.WC * 12 + 16,
                         (const double *)a.W.lb, (const double *)a.W.ub, S->P.col_scale, n);
      hipLaunchKernelGGL(k_lane_fill_records, dim3((m + 255) / 256), tb, 0, st, L->rrec, RREC, L->plan.WR * 12, L->plan.WR * 12 + 16 + NLP * 8,
                         (const double *)a.W.rlo, (const double *)a.W.rhi, S->P.row_scale, m);
    }
    if ((e = hipMemsetAsync(W.partial, 0, (size_t)G * la.nslot * kLaneNQ * 64 * sizeof(double), st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_lane_setup, dim3(G), dim3(64), 0, st, la, (const double *)a.W.lb, (const double *)a.W.ub);

    const dim3 g_tile(T->nwg, G), b_tile(kLaneWaves * 64);
    const size_t lds_ring = (size_t)kLaneWaves * T->ring * 64 * sizeof(double);
    const size_t lds_red = (size_t)kLaneWaves * std::max(NLP, 8) * 64 * sizeof(double);
    const size_t lds_stage = (size_t)kLaneWaves * kLaneStageBytes;
    const size_t lds[3] = {std::max(2 * lds_ring + lds_stage, lds_red), std::max(3 * lds_ring + lds_stage, lds_red), std::max(2 * lds_ring + lds_stage, lds_red)};
    for (int mode = 0; mode < 3; ++mode)
      if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern[mode]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds[mode])) != hipSuccess) return e;
    // the long columns' partial sums of the y the phase starts from (the iterate itself stays where it is)
    {
      LaneArgs q = la; q.x_in = W.xA; q.y_in = W.yA; q.sums_only = 1;
      if (NLP == 4) hipLaunchKernelGGL(k_lane_apply<4>, g_tile, b_tile, 0, st, q);
      else hipLaunchKernelGGL(k_lane_apply<8>, g_tile, b_tile, 0, st, q);
    }

    const dim3 g_long(std::max(1, L->P.nl), G), tl(kLongWaves * 64);
    // One check period = 2 (C - 1) + 7 launches with the same arguments every time (a period starts in buffer A and k_lane_apply
    // leaves the iterate there): captured once into a hipGraph and replayed - a year-long solve is 10^3 .. 10^4 periods, and below
    // ~30 us per iteration the host cannot enqueue two launches per iteration fast enough.  The legacy default stream cannot be
    // captured: the loop then runs on the handle's own stream between two host synchronisations (the call is blocking anyway).
    auto enqueue_period = [&](hipStream_t s) {
      double *xcur = W.xA, *ycur = W.yA, *xalt = W.xB, *yalt = W.yB;
      for (int u = 0; u < C - 1; ++u) {
        LaneArgs q = la; q.kofs = u; q.x_in = xcur; q.y_in = ycur; q.x_out = xalt; q.y_out = yalt;
        if (L->P.nl) hipLaunchKernelGGL(k_lane_long<0>, g_long, tl, 0, s, q);
        hipLaunchKernelGGL(kern[0], g_tile, b_tile, lds[0], s, q);
        std::swap(xcur, xalt); std::swap(ycur, yalt);
      }
      LaneArgs q = la; q.kofs = 0; q.x_in = xcur; q.y_in = ycur; q.x_out = xalt; q.y_out = yalt;
      if (L->P.nl) hipLaunchKernelGGL(k_lane_long<1>, g_long, tl, 0, s, q);
      hipLaunchKernelGGL(kern[1], g_tile, b_tile, lds[1], s, q);
      hipLaunchKernelGGL(kern[2], g_tile, b_tile, lds[2], s, q);
      if (L->P.nl) hipLaunchKernelGGL(k_lane_long<2>, g_long, tl, 0, s, q);
      hipLaunchKernelGGL(k_lane_sum, dim3(13, G), tl, 0, s, q);
      hipLaunchKernelGGL(k_lane_decide, dim3(G), dim3(64), 0, s, q);
      LaneArgs r = la; r.x_in = xcur; r.y_in = ycur;
      if (NLP == 4) hipLaunchKernelGGL(k_lane_apply<4>, g_tile, b_tile, 0, s, r);
      else hipLaunchKernelGGL(k_lane_apply<8>, g_tile, b_tile, 0, s, r);
    };
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st) (void)hipStreamIsCapturing(st, &cap);
    const bool use_graph = graph_env && max_periods - period > 2 && cap == hipStreamCaptureStatusNone;
    hipStream_t ls = st;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (use_graph) {
      if (!L->stream && (e = hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking)) != hipSuccess) return e;
      ls = L->stream;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;                    // everything enqueued so far precedes the loop
      if ((e = hipStreamBeginCapture(ls, hipStreamCaptureModeThreadLocal)) != hipSuccess) return e;
      enqueue_period(ls);
      if ((e = hipStreamEndCapture(ls, &graph)) != hipSuccess) return e;
      if ((e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)) != hipSuccess) { (void)hipGraphDestroy(graph); return e; }
    }
    // scenarios still iterating below which the phase ends: they fit G - max(1, G / 4) groups
    const int shrink_at = (compact && G > 1) ? 64 * (G - std::max(1, G / 4)) : -1;
    bool finished = false, shrink = false, certify = false;
    for (; period < max_periods;) {
      if (use_graph) { if ((e = hipGraphLaunch(exec, ls)) != hipSuccess) break; }
      else enqueue_period(ls);
      ++period;
      if (period % poll == 0 || period == max_periods) {
        if ((e = hipMemcpyAsync(S->ndone_host, a.W.ndone, 2 * sizeof(int), hipMemcpyDeviceToHost, ls)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(ls)) != hipSuccess) break;
        if (S->ndone_host[0] >= B) { finished = true; break; }
        if (B - S->ndone_host[0] <= shrink_at && period < max_periods) { shrink = true; break; }
        // Suspects (relative gap >= 1/2 after 2048 iterations: control_decide): the infeasibility / unboundedness certificates are
        // evaluated on the scenario-major workspace (stream_certify, dsp_stream.hip), so the phase ends here, the iterate goes back
        // there, and the scenarios that are not certified go on in a new phase - at most every 16 check periods.
        if (S->ndone_host[1] && period % 16 == 0 && period < max_periods) { shrink = true; certify = true; break; }
      }
    }
    if (use_graph) {
      const hipError_t es = hipStreamSynchronize(ls);
      (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
      if (e == hipSuccess) e = es;
    }
    if (e != hipSuccess) return e;
#ifdef DSP_LANE_PROBE
    if (const char *path = getenv("DSP_LANE_PROBE_OUT")) {
      std::vector<unsigned long long> h((size_t)kProbeWaves * kProbeSlots);
      (void)hipStreamSynchronize(ls);
      if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_lane_probe), h.size() * sizeof(unsigned long long)) == hipSuccess) {
        if (FILE *f = fopen(path, "wb")) {
          const int nwg = (T->ntile + kLaneWaves - 1) / kLaneWaves;
          const int hdr[4] = {T->ntile, G, kLaneWaves, kProbeSlots};
          const size_t prow = std::min<size_t>((size_t)kProbeWaves, (size_t)nwg * kLaneWaves * G);
          fwrite(hdr, sizeof(int), 4, f); fwrite(h.data(), sizeof(unsigned long long), prow * kProbeSlots, f); fclose(f);
        }
      }
    }
#endif
    // ---- out: x+, y+ of the last check back to the scenario-major workspace (k_finalize unscales them); between two phases also the
    //      iterate and its anchors (buffer A: where k_lane_apply left them) -----------------------------------------------------------
    auto out = [&](const double *src, int len, double *dst) {
      hipLaunchKernelGGL(k_lane_out, len == n ? gcol : grow, tb, 0, st, src, len, nact, sid, dst);
    };
    out(W.xp, n, a.W.xp); out(W.yp, m, a.W.yp);
    if (!shrink) break;
    (void)finished;
    out(W.xA, n, a.W.x); out(W.x0, n, a.W.x0); out(W.yA, m, a.W.y); out(W.y0, m, a.W.y0);
    if (certify) {
      if ((e = stream_certify(S, a, st)) != hipSuccess) return e;
      count_phase = false;
    }
    // the scenarios that go on, in their order
    std::vector<int> done_h((size_t)G * 64);
    if ((e = hipMemcpyAsync(done_h.data(), W.done, done_h.size() * sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    // (after a certificate sequence the verdicts are in the scenarios' control blocks, not in the lanes' flags: a scenario that was just
    //  certified infeasible / unbounded must not take a lane in the next phase)
    std::vector<StreamCtrl> hc;
    if (certify) {
      hc.resize((size_t)B);
      if ((e = hipMemcpyAsync(hc.data(), a.W.ctrl, (size_t)B * sizeof(StreamCtrl), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    }
    std::vector<int> next;
    for (int slot = 0; slot < nact; ++slot) {
      const int sc = ids.empty() ? slot : ids[slot];
      if (!done_h[slot] && !(certify && hc[(size_t)sc].done)) next.push_back(sc);
    }
    if (next.empty()) break;
    ids.swap(next);
    nact = (int)ids.size();
    first = false;
  }
  *periods_run = period;
  return hipGetLastError();
}

}  // namespace dsp
