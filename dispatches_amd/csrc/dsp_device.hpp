// dsp_device.hpp — plain structs shared by the kernels (dsp_kernels.hip) and the C ABI (dsp_capi.hip).
#pragma once
#ifdef __HIPCC_RTC__     /* run-time compilation (hiprtc, dsp_rtc.cpp): no system headers, the HIP device API is built in */
#ifndef DSP_RTC_TYPES
#define DSP_RTC_TYPES
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t; typedef unsigned long size_t;
#ifndef INFINITY
#define INFINITY (__builtin_inff())
#endif
#ifndef NAN
#define NAN (__builtin_nanf(""))
#endif
#endif
#else
#include <hip/hip_runtime.h>

#include <cstdint>
#endif

#include "../../include/dsp_hip.h"
#include "dsp_shapes.hpp"

namespace dsp {

constexpr int kMaxLong = 32;   // long vectors (longer than the ELL width) per orientation

// One matrix entry as the kernels read it from LDS with a single ds_read_b128:
// value + BYTE offset of the multiplied vector element inside the wave's exchange buffer.
struct __attribute__((aligned(16))) Entry {
  double v;
  uint32_t off;
  uint32_t pad;
};

struct LongList {
  int count;
  int owner[kMaxLong];   // vector (column / row) index the partial sums belong to
  int start[kMaxLong];   // offset into the tail arrays (multiple of 64)
  int len[kMaxLong];     // padded length (multiple of 64)
};

// Device-resident, scenario-independent data of one (flowsheet, horizon).
// ELL layout: entry (e, slot q, lane) at ((e * S + q) * 64 + lane), S = CPL (A^T, columns) or RPL (A, rows):
// for a fixed e the S slots of a lane are independent multiply-add chains.
struct DeviceProblem {
  int n, m;
  int n_pad, m_pad;            // CPL*64, RPL*64
  int Wc, Wr;                  // ELL widths of A^T (columns) and A (rows)
  int ellc_entries, ellr_entries, tailc_entries, tailr_entries;
  const Entry *ellc, *ellr, *tailc, *tailr;                        // scaled matrix  D_r A D_c
  const Entry *ellc_unscaled, *ellr_unscaled, *tailc_unscaled, *tailr_unscaled;
  const double *col_scale, *row_scale;                             // D_c [n], D_r [m]
  LongList long_c, long_r;
  // register-resident-matrix layout (dsp_prepare.hpp: SortedLayout / SlotELL), everything in POSITION space
  const Entry *mr_ellc, *mr_ellr, *mr_tailc, *mr_tailr;            // slot-major per-slot-width ELL of the scaled matrix
  int mr_tailc_entries, mr_tailr_entries;
  const int *mr_colat, *mr_rowat;                                  // [n_pad] / [m_pad] position -> column / row id, -1 = pad
  unsigned mr_wc_pack, mr_wr_pack;                                 // per-slot widths, 4 bits each
  const int32_t *mr_slot_x, *mr_slot_y;                            // [n_pad] / [m_pad] LDS slot of each exchange-buffer position
  LongList mr_long_c, mr_long_r;                                   // owner = POSITION
  // unscaled copies in the same layout + the LDS byte offset of every NATURAL element (streaming SpMV step)
  const Entry *mr_ellc_unscaled, *mr_ellr_unscaled, *mr_tailc_unscaled, *mr_tailr_unscaled;
  const int32_t *mr_nat_slot_x, *mr_nat_slot_y;                    // [n_pad] / [m_pad]  8 * slot(position(column j / row i))
};

// internal status written by the simplex kernel for scenarios it leaves to the PDLP kernel (never reaches the caller)
#define DSP_STATUS_UNSOLVED 99
// internal status of a register-resident solve kernel for a scenario whose objectives keep drifting apart (an LP without a solution is
// suspected): the certificate pass - the generic kernel, which evaluates the certificates - takes it from there (never reaches the caller)
#define DSP_STATUS_SUSPECT 98

struct SolveArgs {
  DeviceProblem P;
  dsp_batch b;
  int skip_solved;             // 1 = only scenarios whose status is DSP_STATUS_UNSOLVED are solved (after the simplex pass);
                               // 2 = only DSP_STATUS_SUSPECT ones, continued from the iterate in b.x / b.y (certificate pass)
                               // 3 = only OPTIMAL ones flagged DSP_FLAG_OBJ_WAIVED, from a cold start; results replace the earlier
                               //     ones only when certified (re-certification pass, dsp_options::recertify_passes)
  int waves_per_block;
  double eta;
  dsp_options opt;
  int *queue;                  // device work-queue head of this launch (zeroed on the stream right before it)
  unsigned queue_base;         //   scenario = ticket - queue_base (0 since the heads are reset per launch)
  const int *unsolved;         // skip_solved: number of scenarios the simplex pass left unsolved (0 = nothing to do)
  int matreg;                  // 1 = register-resident-matrix specialisation of the kernel
  int qp;                      // 1 = soft rows present (b.row_compliance): QP instantiation
  double *trace;               // development (-DDSP_KKT_TRACE, DSP_TRACE_SCENARIO): [4096][12] KKT history of one scenario
  int trace_scenario;
  int *suspects;               // scenarios the register-resident kernel left DSP_STATUS_SUSPECT (0 = the certificate pass has nothing to do);
                               // suspects[2]: scenarios currently flagged DSP_FLAG_OBJ_WAIVED (0 = the re-certification passes have nothing to do)
};

// Layout token of the kernarg buffer: a run-time compiled kernel (dsp_rtc.hpp) receives SolveArgs as raw bytes and is compiled
// from whatever csrc/ is on disk, so the library and the code object must agree on the structure AND on the development
// switches that change what the kernel does with it.  The token is compiled into both sides (the code object exports it as
// `dsp_rtc_layout_token`); a mismatch - a stale libdsp_hip.so next to newer sources, a -DDSP_KKT_TRACE build of one side
// only - refuses the run-time kernel instead of running it on a mis-read buffer.
#ifdef DSP_KKT_TRACE
#define DSP_SW_TRACE 1ull
#else
#define DSP_SW_TRACE 0ull
#endif
#ifdef DSP_PROF
#define DSP_SW_PROF 2ull
#else
#define DSP_SW_PROF 0ull
#endif
#ifdef DSP_CLOCKS
#define DSP_SW_CLOCKS 4ull
#else
#define DSP_SW_CLOCKS 0ull
#endif
#ifdef DSP_LEGACY_PULL
#define DSP_SW_PULL 8ull
#else
#define DSP_SW_PULL 0ull
#endif
#ifdef DSP_NO_JUMP
#define DSP_SW_NOJUMP 16ull
#else
#define DSP_SW_NOJUMP 0ull
#endif
constexpr unsigned long long kBuildSwitches = DSP_SW_TRACE | DSP_SW_PROF | DSP_SW_CLOCKS | DSP_SW_PULL | DSP_SW_NOJUMP;
constexpr unsigned long long kSolveArgsToken =
    ((unsigned long long)sizeof(SolveArgs) << 40) ^ ((unsigned long long)__builtin_offsetof(SolveArgs, opt) << 28) ^
    ((unsigned long long)__builtin_offsetof(SolveArgs, queue) << 16) ^ ((unsigned long long)sizeof(DeviceProblem) << 52) ^
    ((unsigned long long)DSP_VERSION << 8) ^ kBuildSwitches;

// in-wave dense simplex for tiny LPs (dsp_simplex.hip)
struct SimplexArgs {
  int n, m;
  int row_stride;              // doubles per tableau row in LDS (odd)
  int max_pivots;
  const double *A_dense;       // [m][n] scaled matrix D_r A D_c, row-major
  const double *col_scale, *row_scale;
  dsp_batch b;
  double tol_p, tol_d, tol_piv;
  int *unsolved;               // incremented for every scenario left to the PDLP kernel
  int debug_keep;              // development (DSP_SX_DEBUG=1): status 51 / 52 instead of the hand-over
  // warm start from the final basis of the previous solve of the same scenario on this handle (dsp_options::simplex_warm):
  int warm;                    // 0 off; 1 start from the saved state where there is one, save the final state; 2 cold start, save
  double *warm_T;              // [B][m][n + m] tableau of the final basis
  int *warm_basis;             // [B][m] variable of every row
  unsigned char *warm_upper;   // [B][n + m] 1 = nonbasic at its upper bound
  int *warm_valid;             // [B] 1 = the state above is that of a certified optimal vertex
};

struct SpmvArgs {
  DeviceProblem P;
  int B;
  int waves_per_block;
  const double *X, *Y;
  double *AX, *ATY;
};

// padded register-resident specialisations: every slot kPadWidth entries wide (LP shapes without a tight specialisation)
constexpr int kPadWidth = 4;
inline unsigned uniform_pack(int slots, int w) { unsigned p = 0; for (int q = 0; q < slots; ++q) p |= (unsigned)w << (4 * q); return p; }

#ifndef __HIPCC_RTC__
hipError_t launch_solve(int cpl, int rpl, const SolveArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);
int matreg_available(int cpl, int rpl, unsigned wc_pack, unsigned wr_pack, bool lng, bool qp = false);   // 0 / 1
hipError_t occupancy_solve(int cpl, int rpl, const SolveArgs &a, int block_threads, size_t lds, int *blocks_per_cu);
size_t simplex_lds_bytes(int n, int m, int *row_stride);
hipError_t launch_simplex(const SimplexArgs &a, int grid, size_t lds, hipStream_t st);
// float32-iterate solve (dsp_qp.hip); returns the launch geometry it chose
hipError_t launch_solve_f32(int cpl, int rpl, const SolveArgs &a, int num_cus, size_t lds_limit, hipStream_t st, int *grid,
                            int *threads, size_t *lds);
// bid-curve points of a solved batch (dsp_bids.hip)
hipError_t launch_bid_points(const dsp_bid_request &rq, hipStream_t st);
hipError_t launch_spmv(int cpl, int rpl, const SpmvArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st);

#endif

}  // namespace dsp
