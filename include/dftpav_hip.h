/*
 * dftpav_hip.h — C-ABI of the MI355X-native batched MINCO / L-BFGS trajectory solver.
 *
 * This is the drop-in boundary for Dftpav's traj_planner solve path.  The
 * reference has no FFI layer: the boundary there is the C++ class
 * PolyTrajOptimizer (traj_planner/include/plan_manage/traj_optimizer.h:100-120)
 * plus the L-BFGS callback typedef (include/geo_utils2d/lbfgs.hpp:200-202).
 * Every entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain C PODs, caller-allocated buffers, no exceptions, no logging.
 *   - all reals are IEEE fp64 (the reference computes in double throughout,
 *     common/basics/basics.h:29 `decimal_t = double`).
 *   - 2x3 boundary states are column-major [px,py, vx,vy, ax,ay]
 *     (Eigen::MatrixXd(2,3) default storage, traj_optimizer.cpp:8).
 *   - return value: DFTPAV_OK (0) or a negative DFTPAV_E_* code.  Per-trajectory
 *     solver status uses the reference's lbfgs enum values (lbfgs.hpp:135-184).
 *   - one handle = one HIP stream + its device buffers; single owner thread
 *     (PolyTrajOptimizer is not re-entrant either, traj_optimizer.h:80-92).
 */
#ifndef DFTPAV_HIP_H
#define DFTPAV_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library return codes ------------------------------------------------ */
#define DFTPAV_OK 0
#define DFTPAV_E_INVALID (-1)     /* bad argument / size mismatch (traj_optimizer.cpp:26-48) */
#define DFTPAV_E_MINI_T (-2)      /* initTs.min < mini_T          (traj_optimizer.cpp:30-33) */
#define DFTPAV_E_ONE_PIECE (-3)   /* a segment with <2 pieces     (traj_optimizer.cpp:38-41) */
#define DFTPAV_E_NO_DEVICE (-4)   /* no HIP device / kernel image not loadable: never falls back to CPU */
#define DFTPAV_E_HIP (-5)         /* a HIP runtime call failed (see dftpav_last_error) */
#define DFTPAV_E_UNSUPPORTED (-6) /* layout exceeds a compiled limit */
#define DFTPAV_E_COMM (-7)        /* RCCL is not loadable, or one of its calls failed (see dftpav_last_error) */

/* ---- per-trajectory solver status: values of lbfgs.hpp:135-184 ----------- */
enum {
  DFTPAV_LBFGS_CONVERGENCE = 0,
  DFTPAV_LBFGS_STOP = 1,
  DFTPAV_LBFGS_CANCELED = 2,
  DFTPAV_LBFGSERR_UNKNOWNERROR = -1024,
  DFTPAV_LBFGSERR_INVALID_N = -1023,
  DFTPAV_LBFGSERR_INVALID_MEMSIZE = -1022,
  DFTPAV_LBFGSERR_INVALID_GEPSILON = -1021,
  DFTPAV_LBFGSERR_INVALID_TESTPERIOD = -1020,
  DFTPAV_LBFGSERR_INVALID_DELTA = -1019,
  DFTPAV_LBFGSERR_INVALID_MINSTEP = -1018,
  DFTPAV_LBFGSERR_INVALID_MAXSTEP = -1017,
  DFTPAV_LBFGSERR_INVALID_FDECCOEFF = -1016,
  DFTPAV_LBFGSERR_INVALID_SCURVCOEFF = -1015,
  DFTPAV_LBFGSERR_INVALID_MACHINEPREC = -1014,
  DFTPAV_LBFGSERR_INVALID_MAXLINESEARCH = -1013,
  DFTPAV_LBFGSERR_INVALID_FUNCVAL = -1012,
  DFTPAV_LBFGSERR_MINIMUMSTEP = -1011,
  DFTPAV_LBFGSERR_MAXIMUMSTEP = -1010,
  DFTPAV_LBFGSERR_MAXIMUMLINESEARCH = -1009,
  DFTPAV_LBFGSERR_MAXIMUMITERATION = -1008,
  DFTPAV_LBFGSERR_WIDTHTOOSMALL = -1007,
  DFTPAV_LBFGSERR_INVALIDPARAMETERS = -1006,
  DFTPAV_LBFGSERR_INCREASEGRADIENT = -1005
};

/* ---- constants of the optimiser ------------------------------------------
 * Replaces PolyTrajOptimizer::setParam (traj_optimizer.cpp:1713-1781), the
 * protobuf OptCfg (proto/minco_config.proto:67-99, values
 * config/minco_config.pb.txt:65-100), common::VehicleParam defaults
 * (common/basics/semantics.h:66-76) and the constants hard-coded in
 * traj_optimizer.h:68 / traj_optimizer.cpp:127-134,197. */
typedef struct dftpav_params {
  int traj_resolution;      /* K   : samples-1 on interior pieces        (pb.txt:66) */
  int des_traj_resolution;  /* K_d : samples-1 on first/last piece       (pb.txt:67) */
  double wei_obs;           /* wei_sta_obs  1000                         (pb.txt:68) */
  double wei_surround;      /* wei_dyn_obs  5000                         (pb.txt:69) */
  double wei_feas;          /* 2500                                      (pb.txt:70) */
  double wei_sqrvar;        /* read, unused by the live path             (pb.txt:71) */
  double wei_time;          /* 500                                       (pb.txt:72) */
  double surround_clearance;/* dyn_obs_clearance 0.4                     (pb.txt:73) */
  double half_margin;       /* 0.15 footprint inflation                  (pb.txt:74) */
  double max_forward_vel, max_forward_acc, max_forward_cur;   /* 5 8 1 (pb.txt:83-85) */
  double max_backward_vel, max_backward_acc, max_backward_cur;/* 2 4 1 (pb.txt:87-89) */
  double max_latacc;        /* computed, penalty disabled in the reference (traj_optimizer.cpp:666-678) */
  double max_phidot;        /* idem                                        (traj_optimizer.cpp:707-773) */
  int gear_opt;             /* GearOpt true                              (pb.txt:95) */
  double non_sinv;          /* 0.24, hard-coded traj_optimizer.h:68 */
  double mini_T;            /* 0.1                                       (pb.txt:99) */
  double fail_cost;         /* 50000, traj_optimizer.cpp:197 */
  /* raw vehicle (semantics.h:66-76); inflated by 2*half_margin inside, as setParam does */
  double veh_width, veh_length, veh_wheel_base, veh_d_cr;
  /* L-BFGS (lbfgs.hpp:15-129 defaults overridden at traj_optimizer.cpp:127-134) */
  int lbfgs_mem_size;       /* 256 (pb.txt:96) */
  int lbfgs_past;           /* 3   (pb.txt:97) */
  double lbfgs_delta;       /* 1e-4 (pb.txt:98) */
  double lbfgs_g_epsilon;   /* 1e-16 */
  int lbfgs_max_iterations; /* 12000 */
  int lbfgs_max_linesearch; /* 64 */
  double lbfgs_min_step;    /* 1e-32 */
  double lbfgs_max_step;    /* 1e20 */
  double lbfgs_f_dec_coeff; /* 1e-4 */
  double lbfgs_s_curv_coeff;/* 0.9 */
  double lbfgs_cautious_factor; /* 1e-6 */
  double lbfgs_machine_prec;    /* 1e-16 */
} dftpav_params;

/* Fills the live values of the reference (files cited on each field above). */
void dftpav_default_params(dftpav_params *p);

/* ---- moving obstacles ----------------------------------------------------
 * Replaces PolyTrajOptimizer::setSurroundTrajs(SurroundTrajData*)
 * (traj_optimizer.h:108, traj_optimizer.cpp:1916) and the fields of
 * plan_utils::LocalTrajData the hot path reads (traj_container.hpp:28-38):
 * traj (pieces), duration, start_time.  A piece is {duration, 2x6 coeffMat}
 * with column 0 multiplying t^5 (poly_traj_utils.hpp:77-87,993); coeffs are
 * stored column-major per piece: [x5,y5, x4,y4, ... x0,y0]. All obstacles of
 * one set share `n_pieces` rows in the flattened arrays via piece_offsets. */
typedef struct dftpav_surround {
  int S;                       /* number of moving obstacles (0 = none) */
  const int *piece_offsets;    /* [S+1] prefix offsets into durations/coeffs */
  const double *durations;     /* [piece_offsets[S]] */
  const double *coeffs;        /* [piece_offsets[S]][12] */
  const double *total_duration;/* [S]  LocalTrajData::duration */
  const double *start_time;    /* [S]  LocalTrajData::start_time */
} dftpav_surround;

/* ---- structure shared by every trajectory of a batch ----------------------
 * M gear segments (`trajnum`), N_i pieces each, singul_i = +1 forward / -1
 * reverse (traj_optimizer.cpp:13-23).  n = 2*sum(N_i-1) + M + 3*(M-1)
 * decision variables laid out [P_0..P_{M-1} | tau | gear xy | gear angle]
 * (traj_optimizer.cpp:80-115). */
typedef struct dftpav_layout {
  int M;
  const int *piece_nums; /* [M] */
  const int *singuls;    /* [M] */
  int H;                 /* half-planes per constraint point (4 on the live path, traj_manager.cpp:1225) */
} dftpav_layout;

/* ---- inputs of OptimizeTrajectory for B trajectories ----------------------
 * Replaces the argument list of PolyTrajOptimizer::OptimizeTrajectory
 * (traj_optimizer.h:118-120).  Every array is trajectory-major. */
typedef struct dftpav_batch_data {
  const double *ini_states; /* [B][M][6]  iniStates   */
  const double *fin_states; /* [B][M][6]  finStates   */
  const double *inner_pts;  /* [B][2*sum(N_i-1)] initInnerPts, segment after segment, (x,y) per waypoint */
  const double *init_Ts;    /* [B][M]     initTs (segment totals, each >= mini_T) */
  const double *corridor;   /* [B][Npts][H][4] hPoly columns (n_x,n_y,p_x,p_y), outward normal,
                               NOT necessarily unit (normalised inside, traj_optimizer.cpp:49-52);
                               Npts = sum_i (N_i-2)(K+1)+2(K_d+1) in sampling order */
  double t_now;             /* `now`  (traj_optimizer.h:120) */
  double help_eps;          /* `help_eps` -> epis (live value 0.0, traj_manager.cpp:610) */
} dftpav_batch_data;

typedef struct dftpav_handle dftpav_handle;
typedef struct dftpav_batch dftpav_batch;

/* number of decision variables / constraint points of a layout */
int dftpav_num_vars(const dftpav_layout *l);
int dftpav_num_points(const dftpav_params *p, const dftpav_layout *l);

/* Replaces PolyTrajOptimizer construction + setParam (traj_optimizer.h:100).
 * device: HIP ordinal.  Fails with DFTPAV_E_NO_DEVICE when no gfx950 device
 * is usable — there is no CPU fallback in this library. */
int dftpav_create(const dftpav_params *params, int device, dftpav_handle **out);
void dftpav_destroy(dftpav_handle *h);
const char *dftpav_last_error(const dftpav_handle *h);

/* Replaces setSurroundTrajs (traj_optimizer.h:108). s==NULL or s->S==0 clears
 * it (surround_trajs_ == NULL, traj_optimizer.cpp:636). Data is copied.
 * Limits (the reference loops over surround_trajs_->size() without one): at most DFTPAV_MAX_SURROUND obstacles with
 * DFTPAV_MAX_SURROUND_PIECES pieces in all -- a larger set is refused here and by dftpav_set_surround_wire
 * (DFTPAV_E_UNSUPPORTED, the installed set is kept); and (constraint points of the layout) x S <= 65535, which only a
 * solve / eval / validation of a batch can check (DFTPAV_E_UNSUPPORTED there).  dftpav_fit_surround fits and installs
 * a set of any size (its result can be read back with dftpav_get_surround); beyond the limits above the solver refuses
 * it in the same way. */
#define DFTPAV_MAX_SURROUND 16
#define DFTPAV_MAX_SURROUND_PIECES 512
int dftpav_set_surround(dftpav_handle *h, const dftpav_surround *s);

/* ---- front-end resampling: from a searched path to the solver's arguments ----
 * Replaces KinoAstar::getKinoNode from SampleTraj on (kino_astar.cpp:606-743:
 * gear segmentation, trapezoid time allocation, flat boundary states) and the
 * resampling of TrajPlanner::RunMINCOParking (traj_manager.cpp:531-568, through
 * KinoAstar::evaluatePos, kino_astar.cpp:468-521) for n_hyp hypotheses. */
typedef struct dftpav_frontend_params {
  double max_forward_vel, max_forward_acc;   /* 5.0, 8.0  (minco_config.pb.txt:77-78) */
  double max_backward_vel, max_backward_acc; /* 2.0, 4.0  (pb.txt:79-80) */
  double non_siguav;                         /* 0.2       (kino_astar.h:207) */
  double wheel_base;                         /* 2.85      (semantics.h:68) */
  double piece_duration;                     /* 1.0       (pb.txt:76 traj_piece_duration) */
  int traj_res, dense_traj_res;              /* samples per piece: inner pieces / first and last piece (pb.txt:66-67) */
} dftpav_frontend_params;
/* Caller-allocated outputs, padded: per hypothesis up to max_seg gear segments,
 * per segment up to max_pieces pieces and max_states constraint-point poses. */
typedef struct dftpav_frontend_out {
  int max_seg, max_pieces, max_states;
  int *n_seg;         /* [n_hyp]  segments found (may exceed max_seg: then only the first max_seg are written) */
  int *singul;        /* [n_hyp][max_seg]  +1 forward / -1 reverse */
  int *piece_nums;    /* [n_hyp][max_seg] */
  double *piece_dt;   /* [n_hyp][max_seg]  timePerPiece; initTs = piece_dt * piece_nums */
  double *ini_states; /* [n_hyp][max_seg][6]  col-major 2x3 flat state (p, v, a) */
  double *fin_states; /* [n_hyp][max_seg][6] */
  double *inner_pts;  /* [n_hyp][max_seg][max_pieces-1][2] */
  int *n_states;      /* [n_hyp][max_seg]  poses produced (may exceed max_states: then only the first are written) */
  double *states;     /* [n_hyp][max_seg][max_states][3]  statelist of getRectangleConst: (x, y, yaw) */
} dftpav_frontend_out;
/* paths: [n_hyp][max_path][3] poses (x, y, yaw in (-pi, pi]) of which path_len[h] >= 2 are used;
 * start_states / end_states: [n_hyp][4] (x, y, yaw, v); start_ctrl: [n_hyp][2] (steer, acceleration). */
int dftpav_frontend_resample(dftpav_handle *h, const dftpav_frontend_params *fp, const double *paths, const int *path_len,
                             int max_path, const double *start_states, const double *end_states, const double *start_ctrl,
                             int n_hyp, const dftpav_frontend_out *out);

/* ---- seeded restart sampler: the batch axis (SURVEY.md §8(d) "Restarts", §8(f)-3) ----
 * No reference counterpart (the reference plans one trajectory per cycle).  Every
 * hypothesis h (inner waypoints [n_inner] = x0, y0, x1, y1, ...; M segment
 * durations) yields n_restarts trajectories, b = h * n_restarts + r: r == 0 is the
 * hypothesis itself, r > 0 has its waypoints moved by N(0, sigma^2) per coordinate
 * and every duration scaled by U[dur_lo, dur_hi].  The generator (SplitMix64
 * streams keyed by (seed, h, r), Box-Muller) is defined in
 * dftpav_amd/csrc/restart.hip; the same (seed, h, r) gives the same bits anywhere. */
int dftpav_sample_restarts(dftpav_handle *h, const double *inner_pts, const double *durations, int n_hyp, int n_restarts,
                           int n_inner, int M, double sigma, double dur_lo, double dur_hi, unsigned long long seed,
                           double *out_inner_pts, double *out_durations);

/* ---- moving-obstacle trajectory fitting ----
 * Replaces TrajPlanner::ConverSurroundTrajFromPoints (traj_manager.cpp:743-789,
 * with state_to_flat_output :139-158) followed by setSurroundTrajs: S predicted
 * state sequences, states [S][n_states][7] = (x, y, angle, velocity,
 * acceleration, curvature, time_stamp) per state, are each fitted on the device
 * with a uniform-time minimum-jerk trajectory of n_states - 1 pieces and
 * installed as the handle's moving obstacles (S == 0 clears them). */
int dftpav_fit_surround(dftpav_handle *h, const double *states, int S, int n_states);
/* Reads back the installed moving obstacles in the layout of dftpav_surround
 * (any pointer may be NULL; call once with the arrays NULL to get S and the
 * total number of pieces). */
int dftpav_get_surround(dftpav_handle *h, int *S, int *n_pieces, int *piece_offsets, double *durations, double *coeffs,
                        double *total_duration, double *start_time);

/* ---- safe-corridor generation, the step before the solve ----
 * Replaces map_itf_->GetObstacleMap + the map queries of getRectangleConst
 * (traj_manager.cpp:1216-1217, map_adapter.cpp:93-97, semantics.h:351-358):
 * a GridMapND<uint8_t, 2>, cell (ix, iy) at cells[ix + size_x * iy], centred at
 * origin + (ix, iy) * resolution, 80 = OCCUPIED.  The cells are copied. */
typedef struct dftpav_grid_map {
  const unsigned char *cells;
  int size_x, size_y;
  double resolution;
  double origin_x, origin_y;
} dftpav_grid_map;
int dftpav_set_grid_map(dftpav_handle *h, const dftpav_grid_map *map);

/* Replaces TrajPlanner::getRectangleConst (traj_manager.cpp:1213-1469): one
 * vehicle-aligned rectangle per state (x, y, yaw), grown cell by cell on the
 * map of dftpav_set_grid_map with the raw vehicle size of dftpav_params.
 * hpoly: [n_states][4][4], per state the four columns (n_x, n_y, p_x, p_y) of
 * hPoly (traj_manager.cpp:1442-1465) — the layout dftpav_batch_data.corridor
 * takes for H = 4. */
int dftpav_corridor_rectangles(dftpav_handle *h, const double *states, int n_states, double *hpoly);
/* Duration of the last corridor kernel on the device (HIP events), without the copies. */
int dftpav_corridor_last_ms(dftpav_handle *h, float *ms);

/* Device-resident batch of B trajectories with a common layout.  The launch shape (threads per trajectory, workgroups per
 * CU, what is staged in LDS) is chosen here, for layouts with many constraint points with the obstacle set that is
 * installed on the handle at this moment (dftpav_set_surround / dftpav_fit_surround before dftpav_batch_create); a set
 * installed later still works, with the shape chosen without it. */
int dftpav_batch_create(dftpav_handle *h, const dftpav_layout *layout, int B, dftpav_batch **out);
/* The same with the residency plan chosen by the caller instead of by B: 0 = one workgroup per CU, the trajectory's
 * half-planes and the MINCO operators staged in LDS (lowest latency of one solve; the default for B <= #CUs), 1 = two per
 * CU, 2 = four per CU (highest throughput; the default for large B), -1 = by B.  For callers that keep several small
 * batches in flight on several handles (a PolyTrajOptimizer per planner thread, traj_optimizer.h:80-92). */
int dftpav_batch_create_shaped(dftpav_handle *h, const dftpav_layout *layout, int B, int residency, dftpav_batch **out);
void dftpav_batch_destroy(dftpav_batch *b);

/* The setup half of OptimizeTrajectory (traj_optimizer.cpp:26-115): validates,
 * normalises corridor normals, clamps boundary |v|,|a|, packs x0, uploads to
 * HBM.  After this call the batch is resident; solve/eval touch no host data. */
int dftpav_batch_upload(dftpav_batch *b, const dftpav_batch_data *d);

/* The floating-point order of the solve path for this batch (eval, solve, coeffs; default DFTPAV_ORDER_DEVICE).
 *
 * DFTPAV_ORDER_DEVICE: the throughput kernels.  They reassociate three sums of the reference -- the per-piece sums of the
 *   penalty gradient, the MINCO solves (a dense operator for the banded substitution) and the dot products of L-BFGS --
 *   so a single evaluation equals the reference's to rounding (<= 1e-11) and, the solver being chaotic, a whole solve
 *   equals it statistically.
 * DFTPAV_ORDER_REFERENCE: every sum in the order PolyTrajOptimizer executes it (traj_optimizer.cpp:486-705 sample ->
 *   vertex -> plane accumulation, poly_traj_utils.hpp:805-852 banded substitutions, lbfgs.hpp:716-739 two-loop with
 *   sequential dot products), no fused multiply-adds: the mode of a drop-in that must reproduce the CPU planner's
 *   decision exactly, and the parity proof of the other one.  THE CONTRACT, precisely: final x, cost, status, iterations
 *   and evaluations are bit-equal to those of the reference's PROGRAM evaluated with sequential reductions (dot products,
 *   norms and matrix products as one chain from their first term), no FMA contraction and -- where the program calls libm --
 *   correctly rounded calls.  That program is what the CPU restatement executes (oracle/dftpav_oracle.c, orders 0 / 2: test
 *   infrastructure).  The reference itself cannot be built in this project's image (it needs Eigen, ROS and protobuf-generated
 *   code) and holds no golden vectors: PARITY IS UNPINNED against an upstream binary.  Such a binary, built against a real
 *   Eigen, vectorises reductions into 2- or 4-lane partial sums and WILL differ in last bits -- as two such builds differ from
 *   each other -- and this solver turns a last-bit difference into a different, statistically equal answer.
 *     - one gear segment, no moving obstacles: the reference's program has no libm call inside the loop; this mode
 *       returns the bits of the restatement's order 0;
 *     - gear shifts: the reference calls libm's cos / sin of every junction angle per evaluation
 *       (traj_optimizer.cpp:273-282, 311-318), whose bits depend on the host (glibc's are not correctly rounded and are
 *       IFUNC-dispatched by CPU model): this mode uses the CORRECTLY ROUNDED cos / sin (cr_trig.h); it equals the
 *       result of the same program over a host libm whenever that libm rounded every junction angle's cos / sin correctly;
 *     - moving obstacles: libm's exp (40 times) and log (9 times) per (constraint point, obstacle) pair and pow(|v|, 3)
 *       (traj_optimizer.cpp:1686-1707, poly_traj_utils.hpp:109) are likewise the correctly rounded exp / log / x^3;
 *     - gear shifts TOGETHER WITH moving obstacles -- the reference's live call, traj_manager.cpp:604-610 -- are covered,
 *       including trajtimes[i] = duration of segment i - 1 (traj_optimizer.cpp:230-234) and the extra gdT addends per
 *       previous segment (:1674-1676);
 *     - with libm calls in the loop the yardstick is the restatement's order 2 (the same statements with exp / log / pow / sin /
 *       cos from binary128): whole solves of this mode are bit-equal to it on all of the three cases above;
 *     - limits: n <= 256 variables, H <= 12 half-planes, 5 H + S + 4 <= 64 terms per constraint point, every gear segment
 *       >= 2 pieces, 159 KB of LDS for one trajectory's state; otherwise DFTPAV_E_UNSUPPORTED, order unchanged.  (n > 64 or H > 5 --
 *       a plan of more than 32 pieces, corridors that are not rectangles -- run in a slow generic kernel: the plain L-BFGS step
 *       over vectors in LDS, twelve plane slots per point.)  Choose the order again after the number of
 *       obstacles on the handle changed.  dftpav_batch_trace* is a device-order facility (DFTPAV_E_UNSUPPORTED here).
 *     - one step of the path is bit-equal by MEASUREMENT, not by proof: a division by a stored quantity (the diagonals of the
 *       band factorisation, y.s of a stored pair) is a multiplication by its reciprocal with one residual correction, which
 *       is the correctly rounded quotient whenever the first product is a faithful rounding of it (Markstein); no
 *       counter-example in 2^31 random pairs nor in any solve compared with the restatement, none ruled out.  The
 *       environment variable DFTPAV_REF_EXACT_DIV=1 makes the recursion divide (slower; the bits have never differed).
 *   Launch shape by batch size (or by the residency hint of dftpav_batch_create_shaped: 2 = many batches in flight): up to five
 *   trajectories per CU one workgroup each (lowest latency); beyond, one gear segment of <= 16 pieces and n <= 32 without moving
 *   obstacles -- BASELINE configs[2] / [3] -- takes the QUAD shape (solver_ref4.hip): FOUR trajectories per wave, one per row of
 *   16 lanes, a piece per lane, 16 trajectories per CU, the rows popping trajectories from the batch's ring (35-38 k solves/s on a
 *   stream of 4096-batches on MI355X); up to four gear segments of <= 16 pieces in all and n <= 48 without moving obstacles -- BASELINE
 *   configs[1], the reference's live layouts -- the same shape with the segments' pieces side by side on a row (solver_ref4m.hip;
 *   22-25 k solves/s on such a stream); everything else one WAVE per trajectory, eight per CU (solver_ref.hip).  A batch that has
 *   the device to itself (the default; dftpav_batch_set_hand_over(b, 0) announces others behind it) takes every wave slot and
 *   hands its last trajectories to the WAVE shape.  Every shape returns the same bits. */
#define DFTPAV_ORDER_DEVICE 0
#define DFTPAV_ORDER_REFERENCE 1
int dftpav_batch_set_order(dftpav_batch *b, int order);
int dftpav_batch_get_order(const dftpav_batch *b);

/* Decision vectors x0 packed by upload ([B][n], host copy). */
/* RunMINCOParking's pair getRectangleConst(statelist) -> OptimizeTrajectory(..., hPoly_container, ...)
 * (traj_manager.cpp:569-610) without the rectangles leaving the device: the
 * corridor of every trajectory of the batch is generated from its constraint-
 * point poses, states [B][Npts][3] (x, y, yaw), straight into the solver's own
 * layout.  Use with dftpav_batch_upload(d) where d->corridor == NULL.  H must be 4.
 * ORDER: dftpav_batch_upload first, then this call, then solve / eval.  Every upload with d->corridor == NULL
 * invalidates the half-planes of the previous cycle on purpose (they belong to the previous poses): a solve that
 * follows an upload without a new corridor returns DFTPAV_E_INVALID rather than using stale half-planes. */
int dftpav_batch_corridor_from_states(dftpav_batch *b, const double *states);
/* The same when the batch is n_restarts restarts of each hypothesis (trajectory
 * t = hypothesis * n_restarts + restart, as dftpav_sample_restarts lays them out)
 * and the restarts share their hypothesis' corridor, as the path they were
 * sampled from does: states [B / n_restarts][Npts][3]. */
int dftpav_batch_corridor_from_hypotheses(dftpav_batch *b, const double *states, int n_restarts);

int dftpav_batch_get_x0(dftpav_batch *b, double *x0);

/* L1 cut (unit-test boundary) == PolyTrajOptimizer::costFunctionCallback
 * (traj_optimizer.cpp:206-350, type lbfgs.hpp:200-202) for B decision
 * vectors at once: x [B][n] -> f [B], g [B][n].  Host buffers. */
int dftpav_batch_eval(dftpav_batch *b, const double *x, double *f, double *g);

/* Observation of one trajectory's solve, the role of lbfgs_progress_t
 * (lbfgs.hpp:242-249; the reference passes NULL, traj_optimizer.cpp:164) at
 * the granularity of the evaluation callback (lbfgs.hpp:200-202): during the
 * following solves every cost/gradient evaluation of trajectory `traj` is
 * recorded, up to max_evals of them (0 switches it off).  dftpav_batch_get_trace
 * returns n_evals rows of 3 n + 4 doubles: the evaluated x [n], its gradient
 * g [n], the search direction d [n] in force, then f, the trial step stp, the
 * iteration k and the evaluation's number within its line search (row 0 is the
 * evaluation of x0: d, stp and the count are 0).  Used by the lockstep parity
 * test against the reference's line search and two-loop recursion. */
int dftpav_batch_trace(dftpav_batch *b, int traj, int max_evals);
int dftpav_batch_get_trace(dftpav_batch *b, double *rows, int *n_evals);
/* The same for `count` consecutive trajectories from `first` at once (one block of records each), and the read-out of one
 * of them.  Device-order solves only (a reference-order solve needs no such check: it is the reference's sequence). */
int dftpav_batch_trace_range(dftpav_batch *b, int first, int count, int max_evals);
int dftpav_batch_get_trace_of(dftpav_batch *b, int traj, double *rows, int *n_evals);

/* L2 cut (production) == lbfgs::lbfgs_optimize driven from
 * OptimizeTrajectory (traj_optimizer.cpp:159-166, lbfgs.hpp:440-751): one
 * persistent kernel launch runs every trajectory's whole L-BFGS solve on the
 * device, starting from the resident x0.  Asynchronous on the handle's stream. */
int dftpav_batch_solve_async(dftpav_batch *b);
int dftpav_batch_sync(dftpav_batch *b);

/* Results (any pointer may be NULL):
 *   x          [B][n]  final decision vectors (x after lbfgs_optimize returns)
 *   final_cost [B]     `final_cost` (traj_optimizer.cpp:160)
 *   status     [B]     lbfgs_optimize return code (enum above)
 *   success    [B]     flag_success of OptimizeTrajectory (traj_optimizer.cpp:176-201)
 *   iters      [B]     L-BFGS iterations k
 *   evals      [B]     cost/gradient evaluations (iter_num_, traj_optimizer.cpp:334)
 *   hist_sum   [B]     sum over iterations of the history depth used by the
 *                      two-loop recursion (input of the bytes model)
 *   latency_us [B]     in-kernel wall time of each trajectory's solve (100 MHz
 *                      constant counter; OPT.cpp:157-169 computes the same and drops it) */
int dftpav_batch_results(dftpav_batch *b, double *x, double *final_cost, int *status,
                         int *success, int *iters, int *evals, long long *hist_sum,
                         double *latency_us);

/* Chained solves: throughput mode for a stream of equally shaped batches (restarts of successive planning
 * cycles).  The iteration counts of a batch are heavy-tailed: when its queue launch is down to the last
 * trajectories, a plain solve finishes them on a nearly empty device.  dftpav_batch_solve_chained(b, prev) starts
 * b like dftpav_batch_solve_async but (1) leaves b's own last trajectories suspended and (2) takes over the ones
 * `prev` left suspended, so that they are worked off inside b's full-occupancy phase.  Results are bit-identical
 * to a plain solve.  `prev` is complete when this call's work is (stream order); b is complete after the next
 * chained solve that names it as `prev`, or after dftpav_batch_finish(b) -- which dftpav_batch_sync / _results /
 * _pack_results / _coeffs / _validate / _sample_states call by themselves.  b and prev must be different batches
 * of the same handle with the same layout, size, parameters and plan; otherwise (or with prev == NULL, or for
 * batches too small to be scheduled) the call degrades to finishing prev and solving b unchained.  New inputs
 * for a batch (dftpav_batch_upload, dftpav_batch_corridor_from_*) discard what it still had suspended. */
int dftpav_batch_solve_chained(dftpav_batch *b, dftpav_batch *prev);
int dftpav_batch_finish(dftpav_batch *b);

/* Timing across solves and handles: dftpav_mark records one of a handle's two marker events on its stream,
 * dftpav_marks_elapsed_ms returns the device time between a marker of one handle and a marker of another (same
 * device) after waiting for the second -- the HIP-event clock bench.py puts around its timed region. */
int dftpav_mark(dftpav_handle *h, int slot);
int dftpav_marks_elapsed_ms(dftpav_handle *from, int from_slot, dftpav_handle *to, int to_slot, float *ms);

/* The end game of a scheduled solve: once no more than `hand_over` trajectories of the batch are unfinished they
 * leave the queue launch and finish in the latency shape (default: one per CU; negative restores it).  0 keeps every
 * trajectory in the queue launch to its end -- the setting for a stream of batches solved alternately on TWO handles
 * (two HIP streams): the queue launch of the next batch then fills the workgroup slots the previous one frees while
 * it thins out, which keeps the device full without any hand-over (bench.py's default). */
int dftpav_batch_set_hand_over(dftpav_batch *b, int hand_over);

/* Multi-GPU hand-off: one 16-byte record {f64 final_cost, i32 status, i32 iters} per trajectory — what the single
 * all-gather of SURVEY §8(e) carries.  The solve kernels write a trajectory's record in their epilogue, the moment it
 * finishes; there is no packing kernel between the solve and the collective (it used to wait up to 110 ms for a workgroup
 * slot behind the other stream's persistent workgroups).
 *   dftpav_batch_pack_results: copies the records into caller-owned DEVICE memory, asynchronously on the handle's stream.
 *   dftpav_batch_records:      waits for the solve and hands them over in HOST memory [B][16]: the epilogues write every record a
 *                              second time into pinned host memory of the batch, so nothing runs on the device behind the solve (the
 *                              runtime's device-to-host copy is a blit kernel that queues behind other streams' persistent waves). */
int dftpav_batch_pack_results(dftpav_batch *b, void *device_dst);
int dftpav_batch_records(dftpav_batch *b, void *host_dst);

/* The collective itself behind the C-ABI, for a C++ host (the reference's caller is one: TrajPlanner::RunMINCOParking,
 * traj_manager.cpp:608-610) that shards its restarts / hypotheses over the GPUs of a node, one process (or thread) and one
 * dftpav_handle per GPU: ONE RCCL all-gather of the 16-byte records over xGMI, enqueued on the handle's stream.
 *   dftpav_comm_available   1 where RCCL is loadable (dlopen only; every rank may ask), else 0
 *   dftpav_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId) and hands the bytes to the other ranks by whatever
 *                           channel the host has (MPI, a socket, torch.distributed ...)
 *   dftpav_comm_create      every rank, collectively: ncclCommInitRank on the handle's device
 *   dftpav_comm_share       another handle (= HIP stream) of the same process and device uses the owner's communicator: a host
 *                           with k batches in flight on k handles sets up one communicator per rank, not k.  The communicator
 *                           is held by all of them and destroyed when the last lets go (dftpav_comm_destroy / dftpav_destroy,
 *                           any order); every rank issues its collectives in the same order (round-robin over the handles does).
 *                           The holders may live on different host threads: the library serialises its calls on the shared
 *                           communicator with a lock (RCCL takes no concurrent enqueues); the ORDER is still the host's to keep
 *   dftpav_comm_layout      the contiguous shard [first, first + count) of a rank out of global_B trajectories, and `block` =
 *                           the largest shard: the gathered buffer holds nranks blocks of `block` records, rank r's shard at
 *                           the start of block r (the pad, at most one record, is zero)
 *   dftpav_batch_allgather_results   all-gathers this rank's records (written by the solve kernels' epilogues: the send
 *                           buffer is the batch's own record array) into all_records (DEVICE memory, nranks * block * 16 bytes);
 *                           asynchronous: the caller synchronises the handle's stream (dftpav_batch_sync)
 * RCCL is loaded on first use (librccl.so.1); DFTPAV_E_COMM where it is absent or a call fails.
 *
 * HARD REQUIREMENT for more than four handles with work in flight in one process (strong scaling keeps up to 16 steps in
 * flight per GPU): the environment variable GPU_MAX_HW_QUEUES must be set to at least that number BEFORE the HIP runtime
 * starts (the default is 4 hardware queues per process: the fifth stream's launch waits for one of the first four to drain,
 * which halves the throughput of 8 batches in flight and lets a collective queue behind an unrelated batch). */
#define DFTPAV_UNIQUE_ID_BYTES 128
int dftpav_comm_available(void);
int dftpav_comm_unique_id(void *id128);
int dftpav_comm_create(dftpav_handle *h, int nranks, int rank, const void *id128);
int dftpav_comm_destroy(dftpav_handle *h);
int dftpav_comm_share(dftpav_handle *h, dftpav_handle *owner);
int dftpav_comm_layout(int global_B, int nranks, int rank, int *first, int *count, int *block);
int dftpav_batch_allgather_results(dftpav_batch *b, int global_B, void *all_records);

/* Replaces getMinJerkOptPtr()[i].getCoeffs()/getDt() (traj_optimizer.h:112,
 * poly_traj_utils.hpp:1069-1074): regenerates the piece coefficients from the
 * final x on the device.  coeffs [B][Ntot][6][2] (row k multiplies s^k, then
 * x/y), piece_dt [B][M]. */
int dftpav_batch_coeffs(dftpav_batch *b, double *coeffs, double *piece_dt);

/* Duration in ms of the last solve of this batch, measured with HIP events recorded
 * on the handle's stream around its launches.  After dftpav_batch_solve_chained(b, prev)
 * this is the GPU time of the call: b's queue launch, including the trajectories adopted
 * from prev; a later dftpav_batch_finish(b) or adoption of b's stragglers moves the end
 * event to where b became complete. */
int dftpav_batch_last_solve_ms(dftpav_batch *b, float *ms);

/* ---- validation of the result, the step after the solve ----
 * Replaces the collision re-check of TrajPlannerServer::CheckReplan
 * (traj_server_ros.cpp:385-397) for every trajectory of a solved batch: each
 * segment is sampled at t = 0, sample_dt, ... < duration (0.05 in the
 * reference), the vehicle outline at spacing vertex_res (0.1, shapes.h:201)
 * is tested against the map of dftpav_set_grid_map
 * (semantic_map_manager.cc:639-662).  collision[t] = 1 if any sample of
 * trajectory t collides, first_sample[t] = index of the first such sample
 * counted over the segments in order (-1 if none).  Either output may be NULL.
 * dftpav_corridor_last_ms reports the kernel's duration afterwards. */
int dftpav_batch_validate(dftpav_batch *b, double sample_dt, double vertex_res, int *collision, int *first_sample);

/* ---- read-out of the result: Trajectory::GetState over a time grid ----
 * Replaces Trajectory::GetState (poly_traj_utils.hpp:378-406, with
 * Piece::getStateExpPos :303-340) called the way the server plays a plan back
 * (TrajPlannerServer::PublishData, traj_server_ros.cpp:244-259): the gear
 * segments of a trajectory follow one another in time as
 * TrajContainer::addSingulTraj chains them (traj_container.hpp:58-73,
 * traj_manager.cpp:617-624; time 0 = start of the first segment), sample k
 * carries the time stamp t0 + k * sample_dt and is read from the first segment
 * whose end_time is not <= that time; past the last segment nothing is
 * published.  states [B][n_samples][8] = {time_stamp, x, y, angle, curvature,
 * velocity, acceleration, steer} (the fields of common::State the server
 * publishes), rows past n_valid[t] are zero; n_valid [B] may be NULL.
 * filter_singularity != 0 applies TrajPlannerServer::FilterSingularityState
 * (traj_server_ros.cpp:335-356) along each trajectory, the history being the
 * previous sample.  Needs a solved batch. */
int dftpav_batch_sample_states(dftpav_batch *b, double t0, double sample_dt, int n_samples, int filter_singularity,
                               double *states, int *n_valid);

/* ---- one planning cycle, stream-ordered ------------
 * Replaces the body of TrajPlanner::RunMINCOParking from getRectangleConst on
 * (traj_manager.cpp:551-626) together with the consumers of its result: the
 * collision re-check of CheckReplan (traj_server_ros.cpp:385-397) and the state
 * playback (traj_server_ros.cpp:244-259,335-356).  After the boundary states,
 * waypoints and durations of `d` are uploaded (d->corridor is ignored), every
 * stage is enqueued on the handle's stream without the host in between:
 *   rectangles of every hypothesis from the installed map
 *     (states [B / n_restarts][Npts][3], as dftpav_batch_corridor_from_hypotheses)
 *   -> solve -> coefficients of the solutions
 *   -> collision re-check every check_dt seconds (outline points every vertex_res m)
 *   -> states at t0 + k * state_dt, k < n_samples (as dftpav_batch_sample_states).
 * dftpav_plan_cycle returns once everything is enqueued; dftpav_plan_cycle_fetch
 * waits for the stream and copies out whatever is asked for (any pointer may be
 * NULL): the arrays of dftpav_batch_results, dftpav_batch_validate and
 * dftpav_batch_sample_states, bit for bit what the separate calls return. */
int dftpav_plan_cycle(dftpav_batch *b, const dftpav_batch_data *d, const double *states, int n_restarts, double check_dt,
                      double vertex_res, double t0, double state_dt, int n_samples, int filter_singularity);
int dftpav_plan_cycle_fetch(dftpav_batch *b, double *x, double *final_cost, int *status, int *success, int *iters, int *collision,
                            int *first_sample, double *states, int *n_valid);

/* ---- serialised form of a trajectory -----------------------
 * The reference declares traj_planner/msg/PolyTraj.msg:1-9 (drone_id, traj_id,
 * start_time, order, float32 coefficients, durations) and never uses it; plans
 * and predicted obstacle motions travel as plan_utils::LocalTrajData
 * (traj_container.hpp:28-38) inside one process.  This is the same content as
 * bytes, in fp64 so that a trajectory survives the trip bit for bit
 * (little-endian, every field naturally aligned):
 *
 *   header  (32 B)  char magic[4] = "DPTJ"; u16 version = 1; u8 order = 5; u8 dim = 2;
 *                   i32 drone_id; i32 traj_id; i32 n_segments; i32 reserved = 0; f64 start_time
 *   segment (24 B)  i32 singul; i32 n_pieces; f64 start_time; f64 duration   (LocalTrajData of one gear segment,
 *                   start/duration as addSingulTraj computes them from header.start_time)
 *   piece  (104 B)  f64 duration; f64 coeff[12]   CoefficientMat column-major, column 0 multiplies t^5:
 *                   x5,y5, x4,y4, ... x0,y0 (poly_traj_utils.hpp:77-87, 993) — the layout of dftpav_surround.coeffs
 *
 * dftpav_wire_size: bytes of a trajectory with these segments (0 on bad input).
 * dftpav_wire_pack: one trajectory from the arrays dftpav_batch_coeffs returns for it (coeffs [Ntot][6][2], piece_dt [M]).
 * dftpav_wire_info: validates a blob, returns its header fields and the total piece count (any output may be NULL).
 * dftpav_wire_unpack: singuls/piece_nums/seg_start/seg_duration [n_segments], durations [pieces], coeffs [pieces][12].
 * dftpav_set_surround_wire: installs S blobs as the moving obstacles (== dftpav_set_surround; the segments of a
 *   blob are joined into one obstacle trajectory that starts at the blob's start_time).  The reference builds its
 *   obstacle model with getTraj(1) (traj_manager.cpp:726,775), so a blob with a reverse segment is refused.
 * Pure host code except the last one. */
size_t dftpav_wire_size(int n_segments, const int *piece_nums);
int dftpav_wire_pack(const dftpav_layout *layout, const double *coeffs, const double *piece_dt, int drone_id, int traj_id,
                     double start_time, void *buf, size_t capacity, size_t *written);
int dftpav_wire_info(const void *buf, size_t size, int *drone_id, int *traj_id, double *start_time, int *n_segments,
                     int *n_pieces);
int dftpav_wire_unpack(const void *buf, size_t size, int *singuls, int *piece_nums, double *seg_start, double *seg_duration,
                       double *durations, double *coeffs);
int dftpav_set_surround_wire(dftpav_handle *h, const void *const *bufs, const size_t *sizes, int S);

/* ---- Reeds-Shepp shots: analytic hypotheses ------------------
 * Replaces KinoAstar::computeShotTraj and is_shot_sucess (kino_astar.cpp:304-345) for n (from, to) pose
 * pairs (x, y, yaw): the shortest Reeds-Shepp path of turning radius 1 / max_cur
 * (ompl::base::ReedsSheppStateSpace, kino_astar.cpp:423 -- OMPL is not part of the reference tree; the
 * published algorithm is restated in dftpav_amd/csrc/rs_math.h), its length (ReedsSheppStateSpace::distance),
 * word (type 0..17, the rows of OMPL's reedsSheppPathType table) and signed segment lengths in units of the
 * turning radius, the poses at l = 0, checkl, checkl + checkl, ... <= length (ReedsSheppStateSpace::
 * interpolate; 0.2 in the reference, minco_config.pb.txt:59) and, if `collides` is given, whether any of
 * them collides on the map of dftpav_set_grid_map (CheckCollisionUsingPosAndYaw, outline spacing vertex_res).
 * samples [n][max_samples][3]: at most max_samples poses are stored per pair, n_samples reports how many the
 * loop visits.  Any output may be NULL.  The sampled path is what dftpav_frontend_resample takes as a searched
 * path (the reference appends it to the searched prefix, kino_astar.cpp:585-599). */
int dftpav_reeds_shepp_shots(dftpav_handle *h, const double *from, const double *to, int n, double max_cur, double checkl,
                             int max_samples, double vertex_res, double *length, int *type, double *seg, double *samples,
                             int *n_samples, int *collides);

/* One-shot convenience == OptimizeTrajectory for B trajectories. */
int dftpav_solve_batch(dftpav_handle *h, const dftpav_layout *layout, int B,
                       const dftpav_batch_data *d, double *x, double *final_cost,
                       int *status, int *success, int *iters, int *evals);

/* The HIP stream of the handle as an opaque pointer (for callers that want to
 * order their own work after a solve). */
void *dftpav_stream(dftpav_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* DFTPAV_HIP_H */
