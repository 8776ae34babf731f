/*
 * dftpav_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * fp64 CPU restatement of Dftpav's traj_planner solve path, written from the
 * reference sources (file:line cited on every function in dftpav_oracle.c):
 *   src/Plan/traj_planner/src/traj_optimizer.cpp
 *   src/Plan/traj_planner/include/plan_utils/poly_traj_utils.hpp
 *   src/Plan/traj_planner/include/geo_utils2d/lbfgs.hpp
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or known-answer
 * fixtures for this path, and it cannot be built in this project's image (it
 * needs Eigen, ROS and protobuf-generated code).  The restatement is held by
 * its citations, by properties -- finite-difference gradients, MINCO
 * invariants, adjoint-vs-FD (tests/test_oracle_*.py) -- and by vectors it wrote
 * itself (tests/golden).  (Rounds 3-5 compared it, bit for bit, with the
 * reference's sources compiled against stand-in Eigen / ROS / protobuf headers
 * written here; that is not a reference build and is retired: oracle/pyref.py.)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library.  The product (dftpav_amd/, libdftpav_hip.so) never
 * links, imports or executes it.
 */
#ifndef DFTPAV_ORACLE_H
#define DFTPAV_ORACLE_H

#include "../include/dftpav_hip.h" /* POD types only: dftpav_params, dftpav_surround */

#ifdef __cplusplus
extern "C" {
#endif

/* One trajectory-optimisation problem = the arguments of
 * PolyTrajOptimizer::OptimizeTrajectory (traj_optimizer.h:118-120). */
typedef struct oracle_problem {
  int M;                    /* trajnum */
  const int *piece_nums;    /* [M] */
  const int *singuls;       /* [M] */
  const double *ini_states; /* [M][6] col-major 2x3 */
  const double *fin_states; /* [M][6] */
  const double *inner_pts;  /* concatenated 2x(N_i-1), col-major */
  const double *init_Ts;    /* [M] */
  int H;                    /* half-planes per point */
  const double *corridor;   /* [Npts][H][4] (n_x,n_y,p_x,p_y), un-normalised */
  double t_now;
  double help_eps;
  const dftpav_surround *surround; /* NULL = no moving obstacles */
} oracle_problem;

/* values of config/minco_config.pb.txt:65-100 etc., restated independently of the product */
void oracle_default_params(dftpav_params *p);

int oracle_num_vars(const oracle_problem *pb);
int oracle_num_points(const dftpav_params *p, const oracle_problem *pb);

/* opaque prepared problem (private normalised corridor copy, clamped states, LU factors) */
typedef struct oracle_ctx oracle_ctx;
/* returns NULL and sets *err (DFTPAV_E_*) on the validation failures of traj_optimizer.cpp:26-48 */
oracle_ctx *oracle_prepare(const dftpav_params *p, const oracle_problem *pb, int *err);
void oracle_free(oracle_ctx *c);
/* Summation order used by oracle_eval / oracle_solve on this problem:
 *   0 = LITERAL (default): every statement in the reference's program order
 *       (banded LU substitution, += chains over samples, sequential dots).
 *   1 = DEVICE ORDER: the same algorithm with the summation orders the gfx950
 *       kernel uses (dense MINCO operator built from that same banded LU,
 *       per-sample subtotals chained per piece, 64-lane butterfly dots,
 *       portable sin/cos/exp/log) — see dftpav_oracle_dev.cpp.  The kernel is
 *       required to match this mode BIT FOR BIT; mode 0 cross-checks mode 1 to
 *       rounding level on single evaluations.
 *   2 = LITERAL with CORRECTLY ROUNDED cos / sin of the junction angles
 *       (OPT:276-281, 312-317) in place of libm's: the program the
 *       reference-order device kernel runs on layouts with a gear shift, where
 *       libm's own bits are a property of the host CPU.  Identical to mode 0 on
 *       single-segment layouts (no such call). */
void oracle_set_order(oracle_ctx *c, int order);
/* x0 packing of traj_optimizer.cpp:96-115 */
void oracle_pack_x0(const oracle_ctx *c, double *x0);
/* costFunctionCallback, traj_optimizer.cpp:206-350 */
double oracle_eval(oracle_ctx *c, const double *x, double *g);
/* cost split of the last eval: [0]=jerk, [1]=time, [2]=corridor, [3]=surround, [4]=feasibility */
void oracle_last_cost_terms(const oracle_ctx *c, double out[5]);
/* piece coefficients of the last eval: coeffs [Ntot][6][2], piece_dt [M] */
void oracle_last_coeffs(const oracle_ctx *c, double *coeffs, double *piece_dt);

typedef struct oracle_result {
  double final_cost;
  int status;   /* lbfgs return code */
  int success;  /* flag_success, traj_optimizer.cpp:176-201 */
  int iters;    /* k */
  int evals;    /* iter_num_ */
  long long hist_sum; /* sum of `bound` over two-loop recursions */
} oracle_result;

/* lbfgs_optimize on the prepared problem, traj_optimizer.cpp:127-201; x in/out */
void oracle_solve(oracle_ctx *c, double *x, oracle_result *r);

/* prepare + pack + solve for a batch of B problems sharing the layout, arrays
 * trajectory-major exactly as dftpav_batch_data; OpenMP over trajectories when
 * nthreads > 1.  Used by bench.py's cpu_baseline leg. */
int oracle_solve_batch(const dftpav_params *p, const dftpav_layout *l, int B,
                       const dftpav_batch_data *d, const dftpav_surround *s, int nthreads, int order,
                       double *x, double *final_cost, int *status, int *success, int *iters,
                       int *evals, long long *hist_sum, double *seconds_each);

/* the same loop with a choice of what runs per trajectory (bench.py's parity legs):
 *   ORACLE_OP_SOLVE    pack x0, lbfgs_optimize                      (== oracle_solve_batch)
 *   ORACLE_OP_RESTART  lbfgs_optimize from the x handed in          (x in/out)
 *   ORACLE_OP_EVAL     costFunctionCallback at the x handed in: f -> final_cost, g -> g_out [B][n] */
enum { ORACLE_OP_SOLVE = 0, ORACLE_OP_RESTART = 1, ORACLE_OP_EVAL = 2 };
int oracle_batch_op(const dftpav_params *p, const dftpav_layout *l, int B, const dftpav_batch_data *d,
                    const dftpav_surround *s, int nthreads, int order, int op, double *x, double *g_out, double *final_cost,
                    int *status, int *success, int *iters, int *evals, long long *hist_sum, double *seconds_each);

/* ---- generic pieces exposed for unit tests ------------------------------- */
typedef double (*oracle_eval_fn)(void *instance, const double *x, double *g, int n);
/* lbfgs::lbfgs_optimize, lbfgs.hpp:440-751 (stepbound/progress callbacks NULL as at traj_optimizer.cpp:163-164) */
int oracle_lbfgs(int n, double *x, double *f, oracle_eval_fn fn, void *instance,
                 const dftpav_params *p, int *iters, int *evals, long long *hist_sum);

/* MinJerkOpt (poly_traj_utils.hpp:855-1095) stand-alone: generate + getTrajJerkCost;
 * inPs 2x(N-1) col-major, head/tail 2x3 col-major, coeffs out [N][6][2] */
double oracle_minco_generate(int N, const double *inPs, double dT, const double *head,
                             const double *tail, double *coeffs);
/* dense restatement of A_N^{-1} restricted to the N+5 non-zero RHS rows
 * (rows 0,1,2, 6i+5, 6N-3..6N-1): out [6N][N+5] row-major.  Computed with the
 * banded LU of poly_traj_utils.hpp:776-826, one unit vector at a time. */
void oracle_minco_operator(int N, double *out);

/* positiveSmoothedL1, traj_optimizer.cpp:783-806 */
void oracle_smoothed_l1(double x, double *f, double *df);
/* VirtualT2RealT / RealT2VirtualT, traj_optimizer.cpp:360-379 */
double oracle_virtual_to_real_T(const dftpav_params *p, double vt);
double oracle_real_to_virtual_T(const dftpav_params *p, double rt);

#ifdef __cplusplus
}
#endif
#endif
