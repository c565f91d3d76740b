/* dsp_hip.h — C ABI of the MI355X batched dispatch-LP solver (libdsp_hip.so).
 *
 * Drop-in seam: the reference has no FFI; its solve boundary is the Python call
 *     solver.solve(model, tee=...)                       (UPSTREAM pyomo SolverFactory object)
 * made from Bidder.compute_day_ahead_bids / compute_real_time_bids and Tracker.track_market_dispatch
 * (reference call sites: dispatches/case_studies/renewables_case/run_double_loop_battery.py:123,222-285,
 *  dispatches/case_studies/renewables_case/tests/test_multiperiod_wind_battery_doubleloop.py:47,81,140,160,235,
 *  dispatches/workflow/parametrized_bidder.py:73-119).  Where Pyomo writes an LP/NL file and spawns
 * cbc / ipopt, a binding of this library hands over the flattened standard form
 *
 *     min c.x + c0   s.t.  row_lb <= A x <= row_ub,   var_lb <= x <= var_ub          (one scenario)
 *
 * or, for the convex QPs of BASELINE config 5 (quadratic ramp cost), the same with SOFT rows (dsp_batch::row_compliance):
 *
 *     min c.x + c0 + sum_{i soft} (a_i.x - b_i)^2 / (2 kappa_i)      s.t. the hard rows and the column bounds
 *
 * once (dsp_create: the CSR of A shared by every scenario) and then, per call, B dense per-scenario vectors
 * (dsp_solve).  Plain C, opaque handle, caller-owned buffers, int return codes, stream-ordered.
 * All per-scenario pointers in dsp_batch / dsp_spmv_step are DEVICE pointers (e.g. torch.Tensor.data_ptr());
 * everything in dsp_lp_desc is a HOST pointer that is copied during dsp_create.
 */
#ifndef DSP_HIP_H
#define DSP_HIP_H

#ifndef __HIPCC_RTC__   /* hiprtc has no system headers; the kernel sources define the fixed-width types themselves */
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define DSP_VERSION 12

/* return codes (0 = ok, < 0 = API misuse / HIP error; text via dsp_strerror) */
#define DSP_OK                 0
#define DSP_ERR_INVALID       -1   /* bad argument (NULL, negative size, unsorted CSR, ...)            */
#define DSP_ERR_TOO_LARGE     -2   /* LP beyond what the layouts support (see dsp_strerror)            */
#define DSP_ERR_HIP           -3   /* a HIP runtime call failed (dsp_last_hip_error())                */
#define DSP_ERR_NO_DEVICE     -4   /* no gfx950 device visible                                        */
#define DSP_ERR_ALLOC         -5

/* per-scenario termination status written to status[B] */
#define DSP_STATUS_OPTIMAL            0
#define DSP_STATUS_ITERATION_LIMIT    1
#define DSP_STATUS_PRIMAL_INFEASIBLE  2   /* no point satisfies rows and bounds.  Reported (i) for crossed bounds in the input (var_lb > var_ub or
                                             row_lb > row_ub: before any iteration, x / y / obj = NaN), (ii) by the in-wave simplex when its
                                             phase 1 ends at a certified optimum clearly outside the bounds (x / y / obj = NaN), (iii) by the
                                             PDLP kernels - fused and HBM-resident - when the displacement T(z) - z of the iteration is a
                                             FARKAS RAY to dsp_options::eps_infeasible: a row multiplier dy (signs admitted by the finite row
                                             bounds) whose reduced costs -A'dy are absorbed by finite column bounds and whose bound value
                                             dy+.row_lb - dy-.row_ub + rc+.var_lb - rc-.var_ub is positive (x / y hold the last iterate)       */
#define DSP_STATUS_DUAL_INFEASIBLE    3   /* the objective is unbounded below along a recession direction of the feasible set (or the dual is
                                             infeasible): reported by the PDLP kernels when the primal part dx of the displacement, clipped to
                                             the recession cone of the column bounds, has c.dx < 0 and A dx inside the recession cone of the
                                             rows to dsp_options::eps_infeasible (x / y hold the last iterate).  An LP that is infeasible or
                                             unbounded by a margin below that tolerance still ends at ITERATION_LIMIT                          */
#define DSP_STATUS_NUMERICAL          4   /* NaN in the input or NaN / Inf met in the iteration                */

/* per-scenario flag bits written to flags[B] */
#define DSP_FLAG_OBJ_WAIVED   1   /* status OPTIMAL with both feasibility tests at eps_rel, but the objective-error bound
                                     only within 10 eps_obj (or the classic relative gap at eps_rel): bound AND objective stagnated
                                     (dsp_options::polish_patience) or the iteration stalled twice (stall_rescue).  The 1e-6
                                     objective accuracy is NOT certified for this scenario (measured: still within 3e-7);
                                     typically objectives that are the small difference of terms ~1e3-1e6 times larger   */
#define DSP_FLAG_STALL_RESCUE 2   /* the primal weight was reset once by the stall rescue                              */
#define DSP_FLAG_POLISH       4   /* the weight guard was tightened in the polish phase (dsp_options::polish_patience) */

typedef struct dsp_handle dsp_handle;

/* One scenario's LP template.  Replaces the model Pyomo would write to disk on every solve. */
typedef struct dsp_lp_desc {
  int32_t n;                 /* columns                                  */
  int32_t m;                 /* rows                                     */
  int64_t nnz;               /* nonzeros of A                            */
  const int32_t *A_rowptr;   /* [m+1]  CSR, column indices ascending within a row */
  const int32_t *A_colidx;   /* [nnz]  */
  const double  *A_val;      /* [nnz]  */
  const double  *col_scale;  /* [n] or NULL: variable scaling factors, the typical magnitude of each column (x_j = s_j x~_j with
                                x~ of order one) - what IDAES models carry as `iscale.set_scaling_factor` (there as 1 / s).  Applied
                                before the library's own equilibration; inputs and outputs stay in the caller's units.  The
                                wind + battery and wind + PEM bidding LPs (kW, MW, kWh of throughput: 3-8 decades) need 20-60 %
                                fewer iterations with the ranges their bounds imply (dispatches_amd/lp.py: implied_column_ranges) */
} dsp_lp_desc;

/* Solver options (restarted, reflected Halpern PDHG with ray jumps).  Fill with
 * dsp_default_options() and override fields. */
typedef struct dsp_options {
  double  eps_rel;           /* relative KKT tolerance: primal and dual residual (and, with eps_obj = 0, the relative
                                gap)                                                      default 1e-9   */
  double  eps_obj;           /* objective accuracy: the error bound |gap| + sum|y||row violation| + sum|dual residual||x|
                                of the returned point is <= eps_obj (1 + |c.x + c0|) - half of the 1e-6 parity contract by
                                default; it replaces the relative-gap test (c.x without the model constant is ~500x the
                                objective here, which made that test a 3x tighter duplicate and the one every straggler
                                hung on).  0 = classic PDLP tests (eps_rel on the relative gap)   default 5e-7   */
  int32_t max_iter;          /* iteration limit per scenario                          default 200000 */
  int32_t check_every;       /* restart / ray-jump test period (1 SpMV + 1 reduction); 0 = automatic: 16, or 32 for LPs with
                                more than 8 owned elements per lane in the fused kernel (the rare blocks of their check spill), 64 on the
                                streaming path                                            default 0      */
  double  restart_sufficient;/* beta_1: restart when r <= beta_1 r0                   default 0.2    */
  double  restart_necessary; /* beta_2: ... or r <= beta_2 r0 and r increased         default 0.8    */
  double  restart_artificial;/* beta_3: ... or k >= beta_3 * total iterations.  0 = automatic: 0.2 in the fused float64
                                kernels (0.3 with soft rows), 0.36 on the streaming / block-resident and float32 paths.
                                Shorter epochs cut the mean iteration count of every flowsheet by 10-28 % (0.36 -> 0.2:
                                profiles/r04j..r04m scans); they also update the weight more often, hence the gentler
                                pid_kp that goes with them                                   default 0      */
  double  pid_kp;            /* proportional gain of the primal-weight controller; 0 = automatic: 0.6 in the fused float64
                                kernels, 0.7 elsewhere (max_dlog_weight = 0 switches the controller off)  default 0 */
  double  max_dlog_weight;   /* clamp on |delta log(primal weight)| per restart       default log(30)*/
  double  step_scale;        /* eta = step_scale / ||A_scaled||_2                     default 0.998  */
  double  weight_guard;      /* keeps the primal weight where step x rounding noise stays below eps / guard:
                                w >= guard eta 1.1e-16 |c|max / (eps (1+|q|)) (and the mirror bound); 0 = off  default 4 */
  double  jump_steady;       /* ray jump: attempt when |r - r_prev| <= jump_steady r  default 0.05   */
  double  jump_tol;          /* ... and ||T(T z)-2T z+z|| <= jump_tol ||T(T z)-T z||  default 3e-3   */
  double  jump_min;          /* ... and the ray stays >= jump_min steps in its piece  default 4      */
  int32_t ray_jumps;         /* 0 = off, 1 = ray jumps, 2 = + chaining (a landing point is tested again at its
                                first check; measured worse on full batches)           default 1      */
  int32_t ruiz_iters;        /* Ruiz passes before Pock-Chambolle (create time)       default 10     */
  int32_t waves_per_block;   /* scenarios per workgroup (1 wave each); 0 = auto                      */
  int32_t kkt_every;         /* the KKT / termination test (7 reductions + two SpMVs) runs at least every
                                kkt_every-th check (kkt_gate = 0: exactly every kkt_every-th)   default 32     */
  int32_t no_matreg;         /* 1 = never use the register-resident-matrix kernel (create time)  default 0 */
  int32_t geo_iters;         /* geometric-mean equilibration passes BEFORE Ruiz (create time): balances
                                unit-mix rows such as P_T[MW] = 1e-3 (G + O)[kW]; helps the tracking LPs and the
                                nuclear flowsheet, hurts wind+battery bidding           default 0      */
  double  kkt_gate;          /* > 0: the KKT test is scheduled from the (free) fixed-point residual r of the restart
                                test: a test that fails by the factor rho = worst criterion / its limit arms the next
                                one for r <= r_now min(1, kkt_gate / rho); the first test runs at the 4th check and
                                kkt_every bounds the gap.  Only moves WHEN termination is detected, never the iterates.
                                0 = fixed cadence                                         default 16     */
  int32_t stall_rescue;      /* > 0: a restart forced by the artificial criterion alone after >= stall_rescue
                                iterations without the residual decaying means the iteration sits on its rounding
                                floor.  The first time, if the primal weight is within 30x of its rounding guard
                                (weight_guard), the movement-ratio controller has driven it away (seen: gap stuck
                                at 1e-7 relative, residuals at 1e-12) and the weight is pulled back to the
                                geometric mean of its value and the initial weight.  From the second time on the
                                eps_obj tests are waived and the scenario terminates on the eps_rel tests alone:
                                an objective that is the difference of terms 1e5 times larger (near-zero-price
                                days) cannot be resolved to eps_obj in double precision.  The threshold matters:
                                scenarios that converge with a weight at the guard do so within ~10 k iterations
                                and a rescue after 1000 slowed ~70 of the 4096 48-h scenarios 10-25x; 4000 (first
                                possible trigger at iteration 4000 / restart_artificial = 11-20 k) touches none of them.  0 = off
                                                                                          default 4000   */
  int32_t no_simplex;        /* 1 = never use the in-wave dense simplex (create time).  LPs with n + m <= 128 and m <= 64
                                (the hourly real-time-bid and tracking LPs: 24 of the 25 solves of a simulated day) are
                                solved by a bounded-variable primal simplex on the dense tableau, one LP per wave,
                                ~(n + m) / 2 pivots instead of thousands of first-order iterations, and the vertex is
                                certified against the original rows; whatever it does not certify falls through to the
                                PDLP kernel in the same call                                  default 0      */
  double  jump_rel;          /* ray jump: ... and the ray stays >= jump_rel * (iterations since the anchor was
                                last reset) steps in its piece.  A jump resets the Halpern anchor; when the pieces
                                are short the solver otherwise jumps at every opportunity (one jump per ~40
                                iterations, thousands per scenario) and never gets the averaged iteration going:
                                such scenarios were the 10-60x stragglers of every batch        default 3    */
  int32_t precision;         /* 0 = float64 everywhere (the parity path).  1 = float32 iterates, matrix and SpMVs with
                                float64 reductions / KKT tests (dsp_qp.hip: one scenario per wave, no ray jumps), for the
                                fp64-vs-fp32 tolerance sweep of BASELINE config 5 - it cannot reach the 1e-6 objective
                                contract on these LPs (bench.py --workload qp_sweep)             default 0    */
  int32_t polish_patience;   /* > 0: once both feasibility tests hold and the objective-error bound is within 10x of its limit,
                                no 2x improvement of the bound for this many iterations starts the near-miss logic: a weight
                                that sits at its rounding guard gets the guard tightened 4x on the noisy side (at most 3
                                times); after 4x this many iterations without improvement a scenario whose primal objective
                                has not moved by more than eps_obj / 10 either is accepted and flagged DSP_FLAG_OBJ_WAIVED.  Such scenarios - rounding
                                floors and slow drifts along nearly flat directions, the primal objective long converged -
                                were the slowest of every batch (30-58 k iterations).  0 = off       default 1024 */
  int32_t no_rtc;            /* 1 = never compile at run time (create time).  An LP without an ahead-of-time register-resident
                                specialisation (other horizons, other flowsheets, QP variants) gets its tight instantiation
                                compiled by hiprtc in dsp_create: 2-3 s once, then a disk cache ($DSP_RTC_CACHE or
                                ~/.cache/dsp_hip, keyed by shape and a hash of the kernel sources).  Needs libhiprtc.so and
                                the kernel sources ($DSP_KERNEL_SRC or csrc/ next to the library) at run time; without them
                                the padded / LDS-matrix ahead-of-time kernels are used (dsp_rtc_message says why)  default 0 */
  int32_t no_interior_point; /* 1 = never use the interior-point form of the HBM-resident path.  LPs of that path whose normal matrix is
                                TIME-BANDED (half-bandwidth <= 8 in the given row order once at most 4 wide columns - design variables,
                                periodic conditions - are set aside: the year-long price-taker LPs) are solved by a primal-dual
                                interior-point method with exact banded factorisations, one lane per scenario (csrc/dsp_ipm.hip): ~100
                                Newton iterations instead of ~75 k first-order iterations; the factorisations and solves run
                                TIME-PARALLEL from 512 rows on (up to 64 partitions of the horizon, their separators as a
                                block-tridiagonal system: csrc/dsp_ipm_seq.hpp; dsp_stats::stream_phases = partitions).  Scenarios it does not finish (free columns,
                                numerical breakdown, 250 iterations or dsp_options::max_iter if smaller) run the PDHG forms - those scenarios only,
                                the ones it solved stay solved (dsp_stats::ipm_solved) -, batches with soft rows as before; statuses
                                2 / 3 come from the PDHG forms.  `iters` counts Newton iterations for scenarios it solved; warm starts
                                (x0 / y0) are ignored by this form                                                     default 0 */
  double  eps_infeasible;    /* > 0: infeasibility / unboundedness certificates (DSP_STATUS_PRIMAL_INFEASIBLE / DUAL_INFEASIBLE, ABI 9).  On
                                an LP without a solution the PDHG operator has no fixed point, T(z) - z tends to a ray, one of the
                                objectives runs away and the relative gap |c.x - dual objective| / (1 + |c.x| + |dual objective|) tends
                                to 1.  A KKT test that finds the gap >= 1/2 after the 8th check (HBM-resident path: after 2048
                                iterations) marks the scenario SUSPECT - no feasible scenario of the reference's bidding LPs is there at
                                that stage, so feasible batches pay one comparison per KKT test.  For suspects the two certificates are
                                evaluated on the displacement (dy as a Farkas ray, dx clipped to the recession cone of the bounds as a
                                direction of unbounded descent: two products + one reduction) and the scenario ends when
                                    |dual residual of the ray| (1 + |bounds|) <= eps_infeasible * (bound value of the ray)        or
                                    |recession violation of A dx| (1 + |c|) <= eps_infeasible * (-c.dx)
                                (scaled space; both are proofs up to the tolerance, whatever the iterate).  Where: the generic (LDS-
                                matrix) kernels at their restarts; the register-resident kernels only watch the gap and hand a scenario
                                that stays suspect over three KKT tests to a second launch of the generic kernel, which continues from
                                its iterate (that launch returns at once when there is no suspect); the HBM-resident path in a
                                certificate sequence the host enqueues at most every 16 check periods while suspects exist.  Reference
                                behaviour: the solver's termination condition, on which the callers act (case_studies/renewables_case/
                                solar_battery_hydrogen.py:451-458).  0 = off                                        default 1e-6 */
  int32_t recertify_passes;       /* N = 1 .. 3 (ABI 11, fused path): scenarios the solve accepted WITHOUT a certified objective accuracy
                                (DSP_FLAG_OBJ_WAIVED) are solved again ON THE DEVICE, from a cold start, under up to N other restart /
                                weight-controller settings; a certified optimum replaces the flagged point and clears the flag, anything
                                else leaves both in place.  Each pass is one more launch that returns at once while no scenario of the
                                batch is flagged.  For callers that never read the flags back between solves - the rolling double loop
                                replays its simulated days from hipGraphs; the reference re-solves nothing, its solver either converges
                                or the run stops (idaes Bidder: `assert_optimal_termination`).  0 = off (a synchronous caller re-solves
                                flagged scenarios itself: hip_solver.HipPdlpSolver)                                  default 0    */
  int32_t simplex_warm;      /* in-wave simplex (tiny LPs): 1 = start every scenario from the final basis of ITS previous solve on this handle where
                                one is saved (the hourly LPs of a plant's rolling loop share one matrix and differ in costs, bounds and right-hand
                                sides: 2 - 4 pivots from the previous hour's basis against ~26 from the slack basis) and save the final basis;
                                2 = start from the slack basis, save (the first hour of a day: bounds the drift of a tableau carried over);
                                a warm attempt that does not end in a certified optimum is repeated from the slack basis.  The caller keeps
                                scenario k the same plant from call to call.  0 = off                                     default 0    */
  int32_t warm_patience;     /* > 0 (ABI 12, fused path): a scenario started from dsp_batch::x0 / y0 / primal_weight that has not terminated after
                                this many iterations starts again from the cold point (x = clamp(0), y = 0, automatic primal weight) inside the
                                same launch; its iteration count goes on.  A warm start shortens the mean of a rolling day-ahead solve
                                (3650 -> 2244 iterations) but a few scenarios per thousand do far worse from yesterday's point than from
                                zero - one plant-day in 16 384 ran into the iteration limit; with a patience of ~1.5 x the cold mean their cost
                                is bounded by patience + a cold solve.  0 = off                                           default 0    */
} dsp_options;

/* The per-call data of B scenarios.  c is required; every other input may be NULL (= no bound: -inf / +inf,
 * offset 0) and each has its own scenario stride in ELEMENTS: stride 0 broadcasts one template vector to all
 * scenarios, stride n (or m, or 1 for obj_offset) is a dense array.  x0 / y0 (optional warm start) and x / y use
 * dense strides n / m. */
typedef struct dsp_batch {
  int32_t B;
  int32_t reserved;
  const double *c;          int64_t c_stride;
  const double *var_lb;     int64_t var_lb_stride;
  const double *var_ub;     int64_t var_ub_stride;
  const double *row_lb;     int64_t row_lb_stride;
  const double *row_ub;     int64_t row_ub_stride;
  const double *obj_offset; int64_t obj_offset_stride;  /* c0: only used to scale the eps_obj tests */
  /* NULL = LP.  Otherwise [m] (stride 0) or [B][m] (stride m) compliances kappa_i >= 0.  A row with kappa_i > 0 is SOFT:
     it needs row_lb = row_ub = b_i (finite) and is not a constraint but the objective term (a_i.x - b_i)^2 / (2 kappa_i)
     - a convex quadratic objective x'Qx / 2 in FACTORED form, Q = sum_i a_i a_i' / kappa_i (the ramp cost
     (rho / 2) sum_t (P_T[t] - P_T[t-1])^2 is T - 1 such rows with kappa = 1 / rho).  By convex duality the term is an equality row
     whose multiplier pays kappa_i y_i^2 / 2, so the solver's dual step of that row becomes the proximal step
     y+ = (y - sigma (a_i.xbar - b_i)) / (1 + sigma kappa_i) and nothing else changes; y_i = -(a_i.x - b_i) / kappa_i at the
     optimum, obj[] includes the quadratic term.  (SURVEY.md 8(b) proposed Q in CSR; a general Q would need a proximal
     step with Q in the PRIMAL, which was measured 10-20x slower on these problems: tools/pdqp_proto.py.)
     Fused kernels (no vectors longer than the ELL width; the in-wave simplex is skipped) and the HBM-resident streaming path. */
  const double *row_compliance; int64_t row_compliance_stride;
  const double *x0;         /* [B][n] or NULL */
  const double *y0;         /* [B][m] or NULL */
  double  *primal_weight;   /* [B] in/out or NULL: > 0 on entry = initial primal weight of the scenario (rolling-
                               horizon carry-over), anything else = automatic; holds the final weight on exit */
  double  *x;               /* [B][n]  primal solution                                              */
  double  *y;               /* [B][m]  row duals (sign: y >= 0 active lower side, y <= 0 upper)     */
  double  *obj;             /* [B]     c.x at the returned x (the caller adds its own c0)           */
  int32_t *status;          /* [B]     DSP_STATUS_*                                                 */
  int32_t *iters;           /* [B] or NULL   iterations used                                        */
  int32_t *jumps;           /* [B] or NULL   ray jumps taken                                        */
  int32_t *flags;           /* [B] or NULL   DSP_FLAG_* bits of the scenario's solve                */
} dsp_batch;

/* Aggregate statistics of one dsp_solve call (filled after the call's stream work completes when
 * `sync_stats` is non-zero; otherwise only the launch geometry is filled). */
typedef struct dsp_stats {
  int64_t total_iterations;  /* sum over scenarios                                  */
  int32_t max_iterations;    /* slowest scenario                                    */
  int32_t n_optimal;         /* scenarios with status 0                             */
  int32_t grid_blocks;       /* launch geometry                                     */
  int32_t block_threads;
  int32_t lds_bytes;         /* dynamic LDS per block                               */
  int32_t cols_per_lane;     /* CPL template parameter chosen                       */
  int32_t rows_per_lane;     /* RPL template parameter chosen                       */
  float   kernel_ms;         /* hipEvent time of the solve kernel on `stream` (sync_stats only) */
  int32_t matreg;            /* 1 = the register-resident-matrix specialisation ran          */
  int32_t lds_conflicts_identity; /* simulated extra LDS cycles per iteration of the gathers, identity layout */
  int32_t lds_conflicts_chosen;   /* ... with the slot permutation chosen at create time          */
  int32_t simplex;                /* 1 = the in-wave simplex pass ran first (iters[] then counts pivots for the
                                     scenarios it solved)                                           */
  int32_t streaming;              /* 1 = LP beyond the register/LDS-resident kernels (n > 640 or m > 384): the HBM-resident
                                     PDLP ran (state streamed from HBM every iteration: dsp_stream.hip)    */
  int64_t stream_bytes_per_iteration;  /* streaming path: algorithmic HBM bytes per scenario and plain iteration OF THE FORM THAT RAN
                                          (stream_form): 8 (4 n + 3 m) one-launch forms with shared bounds, 8 (6 n + 5 m) with
                                          per-scenario bounds, 8 (8 n + 6 m) two-launch form (+ 8 m with soft rows)             */
  int32_t quadratic;              /* 1 = soft rows present (QP variant of the kernel ran)                */
  int32_t precision;              /* precision the iterates were held in (dsp_options::precision)        */
  int32_t rtc;                    /* 1 = the kernel that ran was compiled at run time for this LP's shape (dsp_options::no_rtc) */
  int32_t stream_form;            /* streaming path, which form of the iteration ran (ABI 7; was reserved): DSP_STREAM_FORM_* */
  int32_t stream_phases;          /* streaming path, lane form (ABI 8): phases the solve ran in = 1 + the number of times the scenarios
                                     still iterating were packed into fewer groups of 64 lanes after others had finished; interior-point form
                                     (round 5): the time partitions its banded factorisations and solves ran in (1 = sequential walks);
                                     0 otherwise */
  int32_t ipm_solved;             /* streaming path (ABI 11): scenarios the interior-point form solved.  stream_form = DSP_STREAM_FORM_IPM with
                                     ipm_solved < B: the others (given up on: an LP without a solution, a breakdown) were handed - they
                                     alone, packed into as few lane groups as they need - to the PDHG forms, whose statuses and
                                     certificates they carry; iters[] counts Newton iterations for the former, PDHG iterations for the latter */
  int32_t reserved1;
} dsp_stats;

/* dsp_stats::stream_form */
#define DSP_STREAM_FORM_NONE       0   /* not the streaming path */
#define DSP_STREAM_FORM_TWO_LAUNCH 1   /* k_primal + k_dual_halpern per iteration: 8 n + 6 m doubles per scenario-iteration */
#define DSP_STREAM_FORM_TILE       2   /* round 3: one launch per iteration, a workgroup per tile of rows x 2 scenarios (k_fused_pre / k_fused) */
#define DSP_STREAM_FORM_LANE       3   /* round 4: scenario-minor storage, a lane per scenario walks a tile (k_lane; batches of 32 scenarios and more) */
#define DSP_STREAM_FORM_BLOCK      4   /* mid-size LPs: the whole solve in one launch, one workgroup per scenario, state in LDS (k_block_solve) */
#define DSP_STREAM_FORM_IPM        5   /* round 5: interior point with banded LDL' factorisations, one lane per scenario (csrc/dsp_ipm.hip) */

void dsp_default_options(dsp_options *opt);

/* Build the shared, device-resident problem data for one (flowsheet, horizon): diagonal preconditioner
 * (Ruiz + Pock-Chambolle), scaled A in lane-major ELL + long-vector form for A and A^T, step size.
 * `opt` may be NULL. */
int dsp_create(const dsp_lp_desc *desc, int device, const dsp_options *opt, dsp_handle **out);

/* Solve the B scenarios of `batch`.  hipStream: a hipStream_t (NULL = default stream).  The call enqueues
 * work and returns; if `stats` is non-NULL and sync_stats != 0 it synchronises the stream and fills the
 * statistics.  `opt` may be NULL (= the options given to dsp_create).
 * Fused kernels and in-wave simplex (n <= 640, m <= 384): stream-ordered, capturable into a hipGraph, several calls of one
 * handle may be in flight on different streams.  HBM-resident streaming path (larger LPs, dsp_stats::streaming): BLOCKING and
 * one call at a time per handle - the per-scenario state (13 vectors of n / m doubles per scenario) is a per-handle
 * workspace and the host polls a finished-counter while it enqueues the iteration launches; concurrent callers are
 * serialised by a mutex inside the handle, the call returns after its stream work has completed, and it cannot be captured
 * into a hipGraph. */
int dsp_solve(dsp_handle *h, const dsp_batch *batch, const dsp_options *opt, dsp_stats *stats, int sync_stats,
              void *hipStream);

/* The PDLP SpMV step in streaming form (vectors in HBM): AX[B][m] = A X[b],  ATY[B][n] = A^T Y[b] with the
 * UNSCALED A.  This is the kernel the HBM roofline of SURVEY.md 8(d) is quoted on
 * (bytes = B*2*8*(n+m) + shared CSR).  A MEASUREMENT entry point - no solve calls it - compiled for the LP shapes of the benchmark
 * workloads (wind + battery 24 / 48 h, nuclear 24 / 48 h, wind + PEM 48 h); DSP_ERR_INVALID for any other LP, DSP_ERR_TOO_LARGE for
 * an LP of the streaming path (whose iteration kernels are its own: csrc/dsp_stream*.hip). */
int dsp_spmv_step(dsp_handle *h, int32_t B, const double *X, const double *Y, double *AX, double *ATY,
                  void *hipStream);

/* Rolling-horizon hand-off of the wind + battery double loop ON THE DEVICE (reference: MultiPeriodWindBattery.update_model,
 * wind_battery_double_loop.py:181-274, and Bidder._pass_price_forecasts, done per plant and hour by Python there): one launch
 * rewrites the per-plant objective entries, the mutable bounds and the realised state of B plants between two solves.
 * All pointers are DEVICE pointers; `model` describes one of the hourly LPs (real-time bidding / tracking model). */
typedef struct dsp_wb_model {
  double *c, *lb, *ub, *rlo, *rhi;     /* [B][n] / [B][m] per-plant vectors of the LP (the dsp_batch inputs of its solves)     */
  const double *base_c;                /* [n]  cost vector without prices                                                      */
  const double *x;                     /* [B][n] solution of its last solve                                                    */
  int32_t n, m, T;                     /* columns, rows, horizon (T <= 8)                                                      */
  int32_t soc_init, thr_init;          /* columns fixed to the realised state of charge / energy throughput                   */
  int32_t soc0, thr0;                  /* columns holding the state after the first period                                    */
  int32_t wind_cols[8];                /* wind production column of every period (upper bound = availability)                 */
  int32_t pt_cols[8][2];               /* P_T[t] = 1e-3 (x[a] + x[b])                                                          */
  int32_t pda_cols[8];                 /* day-ahead power column of every period (bidding models), -1 otherwise               */
  int32_t track_rows[8];               /* dispatch rows (tracking model), -1 otherwise                                        */
  double wind_kw;
  double *c0;                          /* [B] objective constant of every plant (dsp_batch::obj_offset of its solves) or NULL    */
  double c0_base, waste_per_kw;        /* c0 = c0_base + waste_per_kw * sum_t (wind availability of period t)   (ABI 11)         */
  const int32_t *status, *flags;       /* [B] outputs of its last solve or NULL: the phase after the solve folds them into
                                          dsp_wb_state::bad / uncertified (ABI 11; six tensor launches per solve otherwise)       */
} dsp_wb_model;

typedef struct dsp_wb_state {
  int32_t B, N;                        /* plants; length of the price / capacity-factor series                                */
  const int64_t *start;                /* [B] first hour of every plant's year in the series                                  */
  int64_t *hour;                       /* [1] the clock (hours since the start): advanced by phase 2                         */
  const double *da_series, *rt_series, *cf_series;   /* [N]                                                                   */
  double *soc, *thr;                   /* [B] realised state                                                                  */
  const double *da_offer, *da_prices;  /* [B][24] cleared day-ahead dispatch and prices of the current day                    */
  double *delivered, *revenue, *energy_mwh;          /* [B]                                                                   */
  uint8_t *bad;                        /* [1] or NULL: set to 1 when a solve left a status other than optimal                     */
  int64_t *uncertified;                /* [1] or NULL: + the scenarios a solve left flagged DSP_FLAG_OBJ_WAIVED                   */
} dsp_wb_state;

/* phase 0: before the real-time bidding solve of hour-of-day k  (prices, state, wind availability, day-ahead power fixed to the
 *          cleared dispatch for the hours of the horizon inside the cleared day) on `rt`;
 * phase 1: between the solves: real-time offer = SCED dispatch (stub market) -> dispatch rows, state and wind of `tr`;
 * phase 2: after the tracking solve: delivered power, realised state rounded to 2 dp, revenue, energy, clock + 1. */
int dsp_wb_rolling_update(const dsp_wb_state *st, const dsp_wb_model *rt, const dsp_wb_model *tr, int32_t phase, int32_t k,
                          void *hipStream);

/* The same hour-by-hour hand-off for ANY flowsheet of the reference, described instead of hard-coded (round 6; dispatches_amd/rolling_flowsheets.py):
 * power output P_T[t] = sum_e pt_coef[t][e] x[pt_cols[t][e]] + pt_const[t] (<= 2 terms per period: every P_T of the reference is one or two
 * columns - wind_battery_double_loop.py:175, wind_PEM_double_loop.py:172, nuclear_flowsheet_multiperiod_class.py:211), up to two state
 * columns re-fixed to the tracker's realised values rounded to `state_scale` (100: 2 dp, wind + battery :194-200; 1: an integer, the nuclear
 * tank holdup :232; none: wind + PEM), optional wind availability columns with their curtailment constant, horizons up to 16 periods. */
#define DSP_LOOP_MAX_T 16
typedef struct dsp_loop_model {
  double *c, *lb, *ub, *rlo, *rhi;     /* [B][n] / [B][m] per-plant vectors of the LP                                             */
  const double *base_c;                /* [n] cost vector without prices                                                          */
  const double *x;                     /* [B][n] solution of its last solve                                                       */
  double *c0;                          /* [B] objective constant of every plant (dsp_batch::obj_offset)                           */
  int32_t n, m, T, n_state;
  int32_t pt_cols[16][2];  /* -1 = no such term                                                                       */
  double  pt_coef[16][2];
  double  pt_const[16];
  int32_t pda_cols[16];    /* day-ahead power column of every period (bidding models), -1 otherwise                   */
  int32_t track_rows[16];  /* dispatch rows (tracking model), -1 otherwise                                            */
  int32_t wind_cols[16];   /* wind production column of every period, -1 = the flowsheet has no wind                  */
  int32_t state_init[2], state_real[2];/* columns fixed to the realised state / holding it after the first period                */
  double  wind_kw, c0_base, waste_per_kw;
  const int32_t *status, *flags;       /* [B] outputs of its last solve or NULL (as dsp_wb_model)                                 */
} dsp_loop_model;

typedef struct dsp_loop_state {
  int32_t B, N;
  const int64_t *start;                /* [B]                                                                                    */
  int64_t *hour;                       /* [1] the clock: advanced by phase 2                                                      */
  const double *da_series, *rt_series, *cf_series;   /* [N] (cf_series NULL without wind)                                         */
  double *state;                       /* [B][n_state] realised state                                                             */
  double state_scale[2];
  const double *da_offer, *da_prices;  /* [B][24]                                                                                */
  double *delivered, *revenue, *energy_mwh;          /* [B]                                                                       */
  uint8_t *bad;                        /* as dsp_wb_state                                                                         */
  int64_t *uncertified;
} dsp_loop_state;

/* phases as dsp_wb_rolling_update: 0 before the real-time bidding solve of hour-of-day k, 1 between the solves, 2 after the tracking solve */
int dsp_loop_update(const dsp_loop_state *st, const dsp_loop_model *rt, const dsp_loop_model *tr, int32_t phase, int32_t k, void *hipStream);

/* The scenario half of the Bidder's bid assembly ON THE DEVICE (reference: idaes Bidder._assemble_bids as DISPATCHES drives it -
 * dispatches/workflow/coordinator.py hands its bids to Prescient; golden values
 * dispatches/case_studies/renewables_case/tests/test_multiperiod_wind_battery_doubleloop.py:245-250): per hour t the pairs
 * (power[s][t], price[s][t]) of all scenarios, both numbers rounded to cents exactly as Python's round(v, 2) rounds them, pairs with
 * rounded power < p_min, non-finite numbers or ok[s] == 0 dropped, the highest price kept per distinct power, sorted by power.
 * One launch (one workgroup per hour: exact rounding, an LDS bitonic sort, an ordered compaction), no synchronisation.
 * power[s][t] = x[s][col[t][0]]                                                  (terms 0: the day-ahead power columns)
 *             = x[s][col[t][0]] * val[t][0] + constant[t]                        (terms 1)
 *             = (x[s][col[t][0]] * val[t][0] + x[s][col[t][1]] * val[t][1]) + constant[t]   (terms 2: P_T of the real-time models)
 * out[t][0] = number k of distinct powers of hour t, out[t][1 .. k] = the points: high 32 bits power cents, low 32 bits price cents
 * (both signed).  DSP_ERR_TOO_LARGE for B > DSP_BID_MAX_SCENARIOS or T > DSP_BID_MAX_HOURS (the caller keeps its own path then). */
#define DSP_BID_MAX_HOURS 64
#define DSP_BID_MAX_SCENARIOS 16384
typedef struct dsp_bid_request {
  int32_t B, T;                        /* scenarios, hours                                                                     */
  int32_t ldx, ldp;                    /* doubles per scenario in x / in price                                                 */
  int32_t terms;                       /* 0, 1 or 2 (above)                                                                    */
  int32_t reserved;
  const double *x;                     /* [B][ldx] DEVICE: the solution of the batch (dsp_batch::x of its solve)               */
  const double *price;                 /* [B][ldp] DEVICE: the energy prices the curve is built on, hour t in column t         */
  const uint8_t *ok;                   /* [B] DEVICE or NULL: 0 = the scenario offers nothing (not solved to optimality)       */
  int64_t *out;                        /* [T][B + 1] DEVICE                                                                    */
  double p_min;
  int32_t col[DSP_BID_MAX_HOURS][2];
  double val[DSP_BID_MAX_HOURS][2];
  double constant[DSP_BID_MAX_HOURS];
} dsp_bid_request;
int dsp_bid_points(const dsp_bid_request *rq, void *hipStream);

/* Introspection */
int dsp_get_dims(const dsp_handle *h, int32_t *n, int32_t *m, int64_t *nnz);
/* run-time specialisation (dsp_options::no_rtc): compile one shape without a GPU (returns the code size, 0 + reason in msg);
 * why a handle runs on an ahead-of-time kernel instead ("" if it was specialised) */
int dsp_rtc_compile_check(int cpl, int rpl, int has_long, unsigned wc_pack, unsigned wr_pack, int qp, char *msg, int msg_len);
const char *dsp_rtc_message(const dsp_handle *h);
int dsp_get_scaling(const dsp_handle *h, double *row_scale /*[m]*/, double *col_scale /*[n]*/, double *step_eta);

int dsp_destroy(dsp_handle *h);
const char *dsp_strerror(int code);
int dsp_last_hip_error(void);
int dsp_version(void);
/* The first 16 hex digits of the SHA-256 over the library's sources (csrc/ *.hip, *.hpp and this header, in name order, each preceded
 * by its base name) as they were when it was compiled ("unknown" for a build that did not pass -DDSP_SOURCE_HASH).  A binding that sits next to the sources compares it with the files it sees and refuses a stale binary
 * (dispatches_amd/hip_solver.py::load_library); bench.py prints it, so that a measured number names the code that produced it. */
const char *dsp_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* DSP_HIP_H */
