/* oracle_internal.h — TEST INFRASTRUCTURE. Private structs shared by the literal
 * restatement (dftpav_oracle.c) and the device-order replay (dftpav_oracle_dev.cpp). */
#ifndef DFTPAV_ORACLE_INTERNAL_H
#define DFTPAV_ORACLE_INTERNAL_H
#include "dftpav_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int N, lowerBw, upperBw;
  double *ptr;
} banded_t;

typedef struct {
  int N;
  double headPVA[6], tailPVA[6]; /* col-major 2x3 */
  double *b, *c, *adj, *gdC;     /* 6N x 2 */
  banded_t A;
  double t[6], tInv[6];
  double gdT;
  double gdHead[6], gdTail[6]; /* 2x3 col-major */
  double *gdP;                 /* 2 x (N-1) col-major */
} minjerk_t;

typedef struct {
  int n_pieces;
  const double *durs;
  const double *coeffs; /* [n_pieces][12] */
  double duration, start_time;
} sur_traj_t;

struct oracle_ctx {
  dftpav_params P;
  int M, H, n, Npts_total, Ntot;
  int *piece_nums, *singuls, *pt_offset; /* [M], [M], [M+1] */
  double *iniS, *finS;                   /* clamped copies [M][6] */
  double *inner_pts, *init_Ts;
  double *cfgHs;                         /* normalised copy [Npts][H][4] */
  double t_now, epis;
  minjerk_t *mj;                         /* jerkOpt_container */
  int S;
  sur_traj_t *sur;
  double *sur_durs, *sur_coeffs;
  /* footprint, OPT:1749-1775 */
  double veh_length_infl, veh_width_infl;
  double vec_le[5][2], vec_lo[5][2];
  int evals;
  double cost_terms[5];
  int order;   /* 0 literal (reference statement order), 1 device order (replays the kernel) */
  void *dev;   /* device-order state, dftpav_oracle_dev.cpp */
};

/* device-order replay (dftpav_oracle_dev.cpp) */
void oracle_dev_init(oracle_ctx *c);
void oracle_dev_free(oracle_ctx *c);
double oracle_dev_eval(oracle_ctx *c, const double *x, double *g);
void oracle_dev_solve(oracle_ctx *c, double *x, oracle_result *r);
void oracle_dev_coeffs(const oracle_ctx *c, double *coeffs, double *piece_dt);

#ifdef __cplusplus
}
#endif
#endif
