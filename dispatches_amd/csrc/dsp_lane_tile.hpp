// dsp_lane_tile.hpp — one lane's walk through one tile of the lane-per-scenario streaming iteration (plan: dsp_lane_plan.hpp,
// kernels: dsp_stream_lane.hip).  The SAME code is compiled for the device (one call per lane of a wave, `lane` = threadIdx & 63,
// rings in LDS) and for the host (tests/lane_harness.cpp calls it lane after lane on plain arrays): lanes never exchange iterates,
// so the host run is the kernel's arithmetic, operation for operation.
//
// Storage: every per-scenario vector is scenario-minor, v[(element) * 64 + lane] inside its group's block (the caller passes the
// group's base pointers).  Rings: ring[(index & mask) * 64 + lane].
//
// RECORDS.  What depends on the element only - matrix entries, their indices, shared bounds, scale factors - is one record per
// column (CREC bytes) and one per row (RREC bytes), contiguous in element order:
//     column  val[WC] | idx[WC] | lb, ub | col_scale, -            row  val[WR] | idx[WR] | rlo, rhi | al[NLP] | row_scale, -
// (idx: row / column indices of the SHORT entries, stored as the byte offset of the index's ring slot, (index & ring mask) << 9 - the
// records are packed for the ring of the tiling in use; padding entries repeat the first one with value 0; the sign bit of a column's
// idx[0] marks a long column; al: the row's coefficients on the long columns.)  A unit's records are one contiguous block per kind, so
// the wave fetches them with ONE 16-byte-per-lane load each, together with the unit's rows of x, c, y ... one unit ahead, parks them
// in its LDS stage when the unit's turn comes, and every lane reads the same 16 bytes back (broadcast ds_read_b128).  The first
// version read the records through the scalar unit: each record was a scalar-cache miss (~0.25 us) behind an s_waitcnt lgkmcnt(0)
// - scalar loads return out of order, so every use drains the whole queue - 17 such waits per unit of 4 columns + 4 rows: 4.5 us
// per unit, the whole launch latency-bound at 46 us whatever the batch (profiles/r40c_lane_kernel_stats_B64.csv).
// Host build: the records are read where they lie (no stage): lanes run one after the other there.
//
// MODE 0  plain iteration:  x+ = clip(x - tau (c - A^T y)),  xbar = 2 x+ - x,  y+ = prox(y - sig A xbar),  both reflected and
//         averaged with the anchors (x0, y0): x, y written to the other buffer; the long columns' partial sums of A^T y_new.
// MODE 1  check iteration:  x+, y+ themselves are written (xp, yp: no averaging) and the tile's terms of the fixed-point residual
//         and of the row part of the KKT sums are accumulated per lane (slots as k_check_rows of dsp_stream.hip); the long
//         columns' partial sums are those of A^T y+.
// MODE 2  column part of the KKT sums at (x+, y+): reduced costs of the own columns from yp (slots 8 .. 12).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define DSP_LANE_HD __host__ __device__ __forceinline__
#else
#define DSP_LANE_HD inline
#endif

namespace dsp {

// DSP_LANE_PROBE (measurement build only; tools/gpu_lane_probe.sh): every wave of the plain-iteration kernel leaves time stamps of
// its walk - start, per unit (before its loads are requested, after its arithmetic), before and after the workgroup's barrier - in a
// device array the host writes to $DSP_LANE_PROBE_OUT after the solve.  1: stamps only; 2: + a full wait after each unit's loads
// (splits a unit into "waiting for memory" and "arithmetic + LDS"); 1 and 2 walk with one register set, 3: stamps in the product's
// two-register-set walk.
#if defined(DSP_LANE_PROBE) && defined(__HIPCC__)
constexpr int kProbeWaves = 16384, kProbeSlots = 128;     // per wave: [0] wall clock at start, [1] clock at start, [2] units, [3] hw id,
                                                          // [4] end of walk, [5] after barrier, [6] end, [7] wall clock at end, [8 + 3 u ..] per unit
__device__ unsigned long long g_lane_probe[(size_t)kProbeWaves * kProbeSlots];
#endif

// (probe 3: stamps in the two-register-set walk - [8 + 3 u] before the next unit's rows are requested, [8 + 3 u + 2] after the unit's
//  arithmetic)
#if defined(DSP_LANE_PROBE) && defined(__HIP_DEVICE_COMPILE__)
#define DSP_LANE_STAMP(U, Q)                                                                                                              \
  do {                                                                                                                                    \
    if (DSP_LANE_PROBE >= 3 && MODE == 0 && lane == 0) {                                                                                  \
      const size_t pw_ = (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));                        \
      const int ps_ = 8 + 3 * ((U) - ubeg) + (Q);                                                                                         \
      if (pw_ < (size_t)kProbeWaves - 1 && ps_ < kProbeSlots) {                                                                           \
        g_lane_probe[pw_ * kProbeSlots + ps_] = clock64();                                                                                \
        if ((Q) == 2) g_lane_probe[pw_ * kProbeSlots + 2] = (unsigned long long)((U) - ubeg + 1);                                         \
      }                                                                                                                                   \
    }                                                                                                                                     \
  } while (0)
#else
#define DSP_LANE_STAMP(U, Q) do { } while (0)
#endif

template <int N> struct LaneVecD { double v[N]; };
template <int N> struct LaneVecI { int32_t v[N]; };
#if defined(__HIPCC__)
typedef uint32_t LaneQuad __attribute__((ext_vector_type(4)));      // 16 bytes of a record block (a register quad on the device)
#else
struct LaneQuad { uint32_t v[4]; };
#endif

constexpr int lane_crec(int wc) { return wc * 12 + 32; }
constexpr int lane_rrec(int wr, int nlp) { return wr * 12 + 32 + nlp * 8; }
constexpr int kLaneStageBytes = 2048;    // LDS stage per wave: 64 lanes x 16 bytes of column records, the same of row records

// wave-uniform load through the scalar unit (tile table, unit list: one per unit, requested a unit ahead)
template <class T>
DSP_LANE_HD T ldu(const T *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const T __attribute__((address_space(4))) CT;
  return *reinterpret_cast<CT *>(reinterpret_cast<uintptr_t>(p));
#else
  return *p;
#endif
}

// element (e, lane) of a group's block [elements][64] through a 32-bit byte offset on the block's wave-uniform base:
// global_load v, v_off, s[base] - one 32-bit add per access (e * 512 is scalar), no 64-bit vector address arithmetic.  A group's
// block of one vector is 512 bytes per element: 4 GiB = 8 M elements (checked by the host).
DSP_LANE_HD const double &lane_ld(const double *base, int e, uint32_t lane8) {
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + ((uint32_t)e * 512u + lane8));
}
DSP_LANE_HD double &lane_st(double *base, int e, uint32_t lane8) {
  return *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + ((uint32_t)e * 512u + lane8));
}

// (device) the value must be in its register HERE: the wait for its load is placed at this point of straight-line code, with an
// exact count, instead of inside the unit loop - where the merge over the back edge makes it s_waitcnt vmcnt(0), i.e. a wait for
// the prefetched rows of the next unit as well
DSP_LANE_HD void lane_pin(double &v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#else
  (void)v;
#endif
}

DSP_LANE_HD double lane_clamp(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }
DSP_LANE_HD double lane_fin0(double v) { return (fabs(v) < INFINITY) ? v : 0.0; }
DSP_LANE_HD bool lane_finite(double v) { return fabs(v) < INFINITY; }

struct LaneProblem {                 // per handle (+ the shared bounds of the solve, written into the records)
  int n, m, nl;
  const char *crec, *rrec;                       // [n] column records, [m] row records (+ 1 KiB of slack behind each)
  const int32_t *tiles, *units;                  // [ntile][16] (dsp_lane_plan.hpp: kLaneTileInts), [nunit][8]
  int ntile, ring_mask;
};

struct LaneGroup {                   // one scenario group's block of every array (pointers already offset to the group)
  const double *x_in, *y_in, *x0, *c, *y0;
  double *x_out, *y_out;
  const double *lb, *ub, *rlo, *rhi;             // per-scenario bounds (!SHARED)
  const double *kap;                             // soft rows' compliances (QP)
  const double *xbl;                             // [NLP][64] xbar of the long columns
  const double *xpl;                             // [NLP][64] x+ of the long columns (MODE 1)
  double *xp, *yp;                               // MODE 1 out; MODE 2 in
};

// active: the lane has a scenario that is still iterating.  Lanes without one take part in fetching and parking the unit's records
// (a wave's 16-byte-per-lane block), load what lane 0 loads (load_regs) and store nothing.
struct LaneScalars { double tau, sig, oml; bool done, active; };

// per-lane result of a tile: lp[l] = the tile's part of (A^T y)_long_l; v[0 .. 12] the check sums (MODE 1: 0 .. 7, MODE 2: 8 .. 12)
template <int NLP> struct LaneOut { double lp[NLP]; double v[13]; };

template <int WC, int WR, int NLP, int CH, bool SHARED, bool QP, int MODE>
struct LaneTile {
  static constexpr int CREC = lane_crec(WC), RREC = lane_rrec(WR, NLP);
  // the chains of a unit's CH slots are interleaved (the steps of `compute`): most overlap, most registers; two pairs one after
  // the other (128 VGPRs, 4 waves per SIMD) measured no faster (profiles/r40j_lane_variants.log)
  static constexpr int IL = CH;
  static_assert(CH * CREC <= 1024 && CH * RREC <= 1024, "a unit's records must fit one 16-byte-per-lane load");
  struct Unit { int ys0, nys, cx0, ncx, rd0, nrd; };
  struct Regs {
    double ys[CH], x[CH], c[CH], x0[CH], y0[CH];
    double lb[SHARED ? 1 : CH], ub[SHARED ? 1 : CH], rlo[SHARED ? 1 : CH], rhi[SHARED ? 1 : CH];
    double kap[QP ? CH : 1];
    double xp[MODE == 2 ? CH : 1];
    LaneQuad cq, rq;                              // this lane's 16 bytes of the unit's column / row records
  };

  static DSP_LANE_HD Unit load_unit(const LaneProblem &P, int u) {
    const LaneVecI<8> d = ldu(reinterpret_cast<const LaneVecI<8> *>(P.units) + u);
    Unit q;
    q.ys0 = d.v[0]; q.nys = d.v[1]; q.cx0 = d.v[2]; q.ncx = d.v[3]; q.rd0 = d.v[4]; q.nrd = d.v[5];
    return q;
  }

  // every global load of a unit, unconditionally, to clamped addresses (no data-dependent control flow between the loads)
  static DSP_LANE_HD void load_regs(const LaneProblem &P, const LaneGroup &G, const Unit &q, int lane, bool active, Regs &r) {
#if defined(__HIP_DEVICE_COMPILE__)
    r.cq = *reinterpret_cast<const LaneQuad *>(P.crec + ((uint32_t)q.cx0 * (uint32_t)CREC + (uint32_t)lane * 16u));
    if (MODE != 2) r.rq = *reinterpret_cast<const LaneQuad *>(P.rrec + ((uint32_t)q.rd0 * (uint32_t)RREC + (uint32_t)lane * 16u));
    // An idle lane loads what lane 0 loads - the same 8 bytes of the same row, already part of the active lanes' line: no traffic of
    // its own - instead of sitting behind a branch on EXEC: with the branch the number of loads in flight depends on the path, and
    // the compiler's wait before the first use of this unit's registers becomes a wait for every load issued so far (vmcnt(1) on the
    // merge; r40u), i.e. also for the rows of the NEXT unit that were requested to overlap with this one's arithmetic.
    const uint32_t l8 = active ? (uint32_t)lane * 8u : 0u;
#else
    (void)P;
    if (!active) return;
    const uint32_t l8 = (uint32_t)lane * 8u;
#endif
    auto ld = [&](const double *base, int e) -> double {
      return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (((uint32_t)e * 512u) | l8));
    };
#pragma unroll
    for (int k = 0; k < CH; ++k) r.ys[k] = ld(MODE == 2 ? G.yp : G.y_in, q.ys0 + (k < q.nys ? k : 0));
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int j = q.cx0 + (k < q.ncx ? k : 0);
      r.c[k] = ld(G.c, j);
      if (MODE == 2) r.xp[k] = ld(G.xp, j);
      else { r.x[k] = ld(G.x_in, j); r.x0[k] = ld(G.x0, j); }
      if (!SHARED) { r.lb[k] = ld(G.lb, j); r.ub[k] = ld(G.ub, j); }
    }
    if (MODE != 2) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int i = q.rd0 + (k < q.nrd ? k : 0);
        r.y0[k] = ld(G.y0, i);
        if (!SHARED) { r.rlo[k] = ld(G.rlo, i); r.rhi[k] = ld(G.rhi, i); }
        if (QP) r.kap[k] = ld(G.kap, i);
      }
    }
  }

  template <class T>
  static DSP_LANE_HD T rec(const char *base, int off) { return *reinterpret_cast<const T *>(base + off); }
  // ring slot of a record index: the records hold (index & ring mask) << 9 = the byte offset of the slot (dsp_lane_plan.hpp:
  // pack_lane_records; the ring is the tiling's, the records are packed for it) - one add per gather; bit 31 of a column's first
  // index flags a long column
  // (`winl` = the window's base + 8 * lane)
  static DSP_LANE_HD const double &slot(const char *winl, uint32_t off9) { return *reinterpret_cast<const double *>(winl + off9); }

  // ring layout: [0, R) y window, [R, 2R) xbar window, [2R, 3R) x+ window (MODE 1); `stage`: the wave's record stage (device).
  // Each phase (stage, primal, dual) is BRANCH-FREE over its CH slots - a slot beyond the unit's count repeats the unit's first
  // element: the same arithmetic on the same inputs, the same value stored to the same places; only sums are masked - and written
  // in STEPS over all CH slots: every record read, then every ring gather, then the arithmetic.  A wave is one in-order instruction
  // stream: written element by element (record -> index -> gather -> four dependent FMAs -> clamp -> store) every LDS round trip
  // of every element was exposed, 85 s_waitcnt per unit, and the launch was bound by that chain (37 us for a 28-row tile whatever
  // the batch: profiles/r40d_lane_kernel_stats_B64.csv, r40e_lane_variants.log); in steps the CH chains overlap.
  // Columns the tile does not own (halo, long) store their x to the SINK row (row n of every [n + 1][64] column block).
  static DSP_LANE_HD void compute(const LaneProblem &P, const LaneGroup &G, const Unit &q, const Regs &r, const LaneScalars &sc,
                                  int j0, int j1, int lane, double *ring, char *stage, const double (&xbl)[NLP], const double (&xpl)[NLP],
                                  LaneOut<NLP> &out) {
    const int M = P.ring_mask, R = M + 1;
    const uint32_t l8 = (uint32_t)lane * 8u;
    double *yr = ring + lane, *xr = ring + (size_t)R * 64 + lane, *xpr = ring + (size_t)2 * R * 64 + lane;
    const char *ywin = reinterpret_cast<const char *>(yr), *xwin = reinterpret_cast<const char *>(xr), *xpwin = reinterpret_cast<const char *>(xpr);
#if defined(__HIP_DEVICE_COMPILE__)
    // park the unit's records: lane L's 16 bytes at byte 16 L of each half of the stage; the LDS serves one wave's accesses in order
    *reinterpret_cast<LaneQuad *>(stage + lane * 16) = r.cq;
    if (MODE != 2) *reinterpret_cast<LaneQuad *>(stage + 1024 + lane * 16) = r.rq;
    const char *cbase = stage, *rbase = stage + 1024;
#else
    (void)stage;
    const char *cbase = P.crec + (size_t)q.cx0 * CREC, *rbase = P.rrec + (size_t)q.rd0 * RREC;
#endif
    if (!sc.active) return;
    if (q.nys > 0) {
#pragma unroll
      for (int k = 0; k < CH; ++k) yr[(size_t)((q.ys0 + (k < q.nys ? k : 0)) & M) * 64] = r.ys[k];
    }
    if (q.ncx > 0) {
#pragma unroll
     for (int k0 = 0; k0 < CH; k0 += IL) {
      LaneVecD<WC> av[IL];
      LaneVecI<WC> ix[IL];
      LaneVecD<2> bd[IL];
      double cs[IL], gy[IL][WC];
#pragma unroll
      for (int kk = 0; kk < IL; ++kk) {                               // step 1: the records
        const int k = k0 + kk;
        const char *cr = cbase + (k < q.ncx ? k : 0) * CREC;
        av[kk] = rec<LaneVecD<WC>>(cr, 0);
        ix[kk] = rec<LaneVecI<WC>>(cr, WC * 8);
        if (SHARED) bd[kk] = rec<LaneVecD<2>>(cr, WC * 12);
        else { bd[kk].v[0] = r.lb[k]; bd[kk].v[1] = r.ub[k]; }
        cs[kk] = MODE == 2 ? rec<double>(cr, WC * 12 + 16) : 1.0;
      }
#pragma unroll
      for (int kk = 0; kk < IL; ++kk)                                  // step 2: the gathers of y
#pragma unroll
        for (int e = 0; e < WC; ++e) gy[kk][e] = slot(ywin, (uint32_t)ix[kk].v[e] & (e == 0 ? 0x7fffffffu : 0xffffffffu));
#pragma unroll
      for (int kk = 0; kk < IL; ++kk) {                               // step 3: the arithmetic
        const int k = k0 + kk;
        const bool live = k < q.ncx;
        const int j = q.cx0 + (live ? k : 0);
        const bool own = j >= j0 && j < j1 && ix[kk].v[0] >= 0;         // sign bit of the first index: long column
        double aty = 0.0;
#pragma unroll
        for (int e = 0; e < WC; ++e) aty = fma(av[kk].v[e], gy[kk][e], aty);
        const double lbv = bd[kk].v[0], ubv = bd[kk].v[1];
        if (MODE == 2) {
          // reduced cost at y+ (kkt_col_terms of dsp_stream.hip); only the own columns count
          const double w = (own && live) ? 1.0 : 0.0;
          const double cj = r.c[k], xp = r.xp[k];
          const double rc = cj - aty;
          const double lp = lane_finite(lbv) ? fmax(rc, 0.0) : 0.0;
          const double lm = lane_finite(ubv) ? fmax(-rc, 0.0) : 0.0;
          const double dr = (rc - lp + lm) / cs[kk];
          out.v[8] = fma(w, dr * dr, out.v[8]);
          out.v[9] = fma(w, cj * xp, out.v[9]);
          out.v[10] = fma(w, lp * lane_fin0(lbv) - lm * lane_fin0(ubv), out.v[10]);
          out.v[11] = fma(w, fabs(cj * xp), out.v[11]);
          out.v[12] = fma(w, fabs(rc - lp + lm) * fabs(xp), out.v[12]);
        } else {
          const double x = r.x[k];
          const double xp = lane_clamp(fma(-sc.tau, r.c[k] - aty, x), lbv, ubv);
          const double tt = 2.0 * xp - x;
          xr[(size_t)(j & M) * 64] = tt;
          if (MODE == 1) xpr[(size_t)(j & M) * 64] = xp;
          const int jw = own ? j : P.n;                                     // not ours: the sink row
          if (MODE == 0) lane_st(G.x_out, jw, l8) = fma(sc.oml, r.x0[k] - tt, tt);
          else {
            lane_st(G.xp, (own && !sc.done) ? j : P.n, l8) = xp;
            const double w = (own && live) ? 1.0 : 0.0;
            const double dx = xp - x, d0 = xp - r.x0[k];
            out.v[0] = fma(w, dx * dx, out.v[0]);
            out.v[6] = fma(w, d0 * d0, out.v[6]);
          }
        }
      }
     }
    }
    if (MODE == 2 || q.nrd <= 0) return;
#pragma unroll
    for (int k0 = 0; k0 < CH; k0 += IL) {
      LaneVecD<WR> av[IL];
      LaneVecI<WR> ix[IL];
      LaneVecD<2> bd[IL];
      LaneVecD<NLP> al[IL];
      double rs[IL], gx[IL][WR], gxp[IL][MODE == 1 ? WR : 1], yv[IL];
#pragma unroll
      for (int kk = 0; kk < IL; ++kk) {                               // step 1: the records
        const int k = k0 + kk;
        const char *rr = rbase + (k < q.nrd ? k : 0) * RREC;
        av[kk] = rec<LaneVecD<WR>>(rr, 0);
        ix[kk] = rec<LaneVecI<WR>>(rr, WR * 8);
        if (SHARED) bd[kk] = rec<LaneVecD<2>>(rr, WR * 12);
        else { bd[kk].v[0] = r.rlo[k]; bd[kk].v[1] = r.rhi[k]; }
        al[kk] = rec<LaneVecD<NLP>>(rr, WR * 12 + 16);
        rs[kk] = MODE == 1 ? rec<double>(rr, WR * 12 + 16 + NLP * 8) : 1.0;
      }
#pragma unroll
      for (int kk = 0; kk < IL; ++kk) {                               // step 2: the gathers of xbar (and x+), the row's own y
        const int k = k0 + kk;
#pragma unroll
        for (int e = 0; e < WR; ++e) {
          gx[kk][e] = slot(xwin, (uint32_t)ix[kk].v[e]);
          if (MODE == 1) gxp[kk][e] = slot(xpwin, (uint32_t)ix[kk].v[e]);
        }
        yv[kk] = yr[(size_t)((q.rd0 + (k < q.nrd ? k : 0)) & M) * 64];
      }
#pragma unroll
      for (int kk = 0; kk < IL; ++kk) {                               // step 3: the arithmetic
        const int k = k0 + kk;
        const bool live = k < q.nrd;
        const int i = q.rd0 + (live ? k : 0);
        double ax = 0.0, axp = 0.0;
#pragma unroll
        for (int e = 0; e < WR; ++e) {
          ax = fma(av[kk].v[e], gx[kk][e], ax);
          if (MODE == 1) axp = fma(av[kk].v[e], gxp[kk][e], axp);
        }
#pragma unroll
        for (int l = 0; l < NLP; ++l) {
          ax = fma(al[kk].v[l], xbl[l], ax);
          if (MODE == 1) axp = fma(al[kk].v[l], xpl[l], axp);
        }
        const double rlo = bd[kk].v[0], rhi = bd[kk].v[1];
        const double y = yv[kk];
        const double gyy = fma(-sc.sig, ax, y);
        double yp = gyy - lane_clamp(gyy, -sc.sig * rhi, -sc.sig * rlo);
        const double kp = QP ? r.kap[k] : 0.0;
        if (QP) yp /= fma(sc.sig, kp, 1.0);                          // soft rows: proximal shrink (kappa = 0: hard row)
        if (MODE == 0) {
          const double tt = 2.0 * yp - y;
          const double yn = fma(sc.oml, r.y0[k] - tt, tt);
          lane_st(G.y_out, i, l8) = yn;
          const double yw = live ? yn : 0.0;
#pragma unroll
          for (int l = 0; l < NLP; ++l) out.lp[l] = fma(al[kk].v[l], yw, out.lp[l]);
        } else {
          lane_st(G.yp, sc.done ? P.m : i, l8) = yp;                  // (a finished scenario keeps the x+, y+ it finished with)
          const double w = live ? 1.0 : 0.0;
          const double yw = live ? yp : 0.0;
#pragma unroll
          for (int l = 0; l < NLP; ++l) out.lp[l] = fma(al[kk].v[l], yw, out.lp[l]);
          const double dy = yp - y;
          const double nsadx = -sc.sig * (ax - axp);                 // -sig A (x+ - x)   (xbar - x+ = x+ - x)
          out.v[1] = fma(w, dy * fma(2.0, nsadx, dy), out.v[1]);
          double viol_s = fmax(rlo - axp, 0.0) + fmax(axp - rhi, 0.0);
          double dobj = fmax(yp, 0.0) * lane_fin0(rlo) - fmax(-yp, 0.0) * lane_fin0(rhi);
          double soft = 0.0;
          if (QP && kp > 0.0) {                                      // soft row: no violation, quadratic terms of both objectives
            const double dev = axp - rlo;
            soft = 0.5 * dev * dev / kp;
            dobj -= 0.5 * kp * yp * yp;
            viol_s = 0.0;
          }
          out.v[7] = fma(w, soft, out.v[7]);
          out.v[4] = fma(w, dobj, out.v[4]);
          const double viol = viol_s / rs[kk];
          out.v[2] = fma(w, viol * viol, out.v[2]);
          out.v[3] = fma(w, fabs(yp) * viol_s, out.v[3]);
          const double d0 = yp - r.y0[k];
          out.v[5] = fma(w, d0 * d0, out.v[5]);
        }
      }
    }
  }

  // the whole tile: units [ubeg, uend), the next unit's rows requested before the current one is computed (two register sets,
  // the loop unrolled by two: no copies between them)
  static DSP_LANE_HD void run(const LaneProblem &P, const LaneGroup &G, int tile, int lane, const LaneScalars &sc_in, double *ring, char *stage,
                              LaneOut<NLP> &out) {
    LaneScalars sc = sc_in;
    // the rings start from zeros: a padding entry of an empty vector multiplies whatever its slot holds by 0 (first thing: nothing
    // here waits for a load)
    const int R = P.ring_mask + 1;
    for (int s = 0; s < (MODE == 1 ? 3 : 2) * R; ++s) ring[(size_t)s * 64 + lane] = 0.0;
    // the tile's entry brings the walk's first unit with it (one 64-byte scalar load)
    const LaneVecI<16> tp = ldu(reinterpret_cast<const LaneVecI<16> *>(P.tiles) + tile);
    const int j0 = tp.v[2], j1 = tp.v[3], ubeg = tp.v[4], uend = tp.v[5];
    Unit first;
    first.ys0 = tp.v[8]; first.nys = tp.v[9]; first.cx0 = tp.v[10]; first.ncx = tp.v[11]; first.rd0 = tp.v[12]; first.nrd = tp.v[13];
    const uint32_t l8 = (uint32_t)lane * 8u;
    double xbl[NLP], xpl[NLP];
#pragma unroll
    for (int l = 0; l < NLP; ++l) {
      xbl[l] = (MODE == 2 || !sc.active) ? 0.0 : lane_ld(G.xbl, l, l8);
      xpl[l] = (MODE == 1 && sc.active) ? lane_ld(G.xpl, l, l8) : 0.0;
      out.lp[l] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < 13; ++q) out.v[q] = 0.0;
#if defined(DSP_LANE_NO_PREFETCH) || (defined(DSP_LANE_PROBE) && DSP_LANE_PROBE < 3)
    // one register set, a unit's rows requested right before its arithmetic (measurement variant; the probe build)
    for (int u = ubeg; u < uend; ++u) {
#if defined(DSP_LANE_PROBE) && defined(__HIP_DEVICE_COMPILE__)
      unsigned long long *pw = g_lane_probe + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kProbeSlots;
      const bool pon = MODE == 0 && lane == 0 && (size_t)(pw - g_lane_probe) < (size_t)(kProbeWaves - 1) * kProbeSlots && 8 + 3 * (u - ubeg) + 2 < kProbeSlots;
      if (pon) pw[8 + 3 * (u - ubeg)] = clock64();
#endif
      const Unit qa = u == ubeg ? first : load_unit(P, u);
      Regs ra;
      load_regs(P, G, qa, lane, sc.active, ra);
#if defined(DSP_LANE_PROBE) && defined(__HIP_DEVICE_COMPILE__)
      if (DSP_LANE_PROBE >= 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (pon) pw[8 + 3 * (u - ubeg) + 1] = clock64(); }
#endif
      compute(P, G, qa, ra, sc, j0, j1, lane, ring, stage, xbl, xpl, out);
#if defined(DSP_LANE_PROBE) && defined(__HIP_DEVICE_COMPILE__)
      if (pon) { pw[8 + 3 * (u - ubeg) + 2] = clock64(); pw[2] = (unsigned long long)(u - ubeg + 1); }
#endif
    }
#else
    // Two register sets: the rows of unit u + 1 are requested before the arithmetic of unit u (its descriptor a unit earlier still),
    // the loop unrolled by two so that no register moves between the sets.  This pays only since the loads are unconditional
    // (load_regs: an idle lane's go to one line): behind a branch on EXEC the wait before unit u's first use was a wait for u + 1's
    // rows as well.  B = 64: 47.3 -> 43.2 us, B = 16: 45.2 -> 37.3 us per iteration (profiles/r41a_lane_variants.log).
    Unit qa = first, qb = load_unit(P, ubeg + 1 < uend ? ubeg + 1 : ubeg), qn = qa;
    Regs ra, rb;
    load_regs(P, G, qa, lane, sc.active, ra);
#pragma unroll
    for (int l = 0; l < NLP; ++l) { lane_pin(xbl[l]); if (MODE == 1) lane_pin(xpl[l]); }
    lane_pin(sc.tau); lane_pin(sc.sig); lane_pin(sc.oml);
    for (int u = ubeg; u < uend; u += 2) {
      qn = load_unit(P, u + 2 < uend ? u + 2 : u);                // descriptors two units ahead, rows one unit ahead
      DSP_LANE_STAMP(u, 0);
      if (u + 1 < uend) load_regs(P, G, qb, lane, sc.active, rb);
      compute(P, G, qa, ra, sc, j0, j1, lane, ring, stage, xbl, xpl, out);
      DSP_LANE_STAMP(u, 2);
      if (u + 1 >= uend) break;
      qa = qn;
      qn = load_unit(P, u + 3 < uend ? u + 3 : u);
      DSP_LANE_STAMP(u + 1, 0);
      if (u + 2 < uend) load_regs(P, G, qa, lane, sc.active, ra);
      compute(P, G, qb, rb, sc, j0, j1, lane, ring, stage, xbl, xpl, out);
      DSP_LANE_STAMP(u + 1, 2);
      qb = qn;
    }
#endif
  }
};

}  // namespace dsp
