// ref_order_common.h -- what the reference-order kernels share (solver_ref.hip: TEAM / WAVE shapes; solver_ref4.hip: QUAD shape):
// the address-space typedefs, div_by_rcp, the DPP helpers, the packed sweep tables, positiveSmoothedL1, the moving-obstacle
// terms and the per-point tests / records (point_masks, point_emit) -- every expression in the order the reference executes it.
// Moved out of solver_ref.hip unchanged (round 6).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_types.h"
#include "cr_trig.h"

namespace dftpav {
namespace reford {

// ------------------------------------------------------------------ helpers
typedef double __attribute__((address_space(3))) *ldsd_t;
typedef const double __attribute__((address_space(3))) *ldscd_t;
typedef int __attribute__((address_space(3))) *ldsi_t;
typedef const double __attribute__((address_space(1))) *gcd_t;
typedef double __attribute__((address_space(1))) *gd_t;

// a / b from y = 1 / b (Markstein's correction step): q0 = a y, r = a - b q0 (exact in an FMA), q = q0 + r y.  With y the correctly
// rounded reciprocal this is the correctly rounded quotient whenever q0 is a FAITHFUL rounding of a / b (Markstein's theorem); RN(a y)
// can be up to ~1.5 ulp off, so the theorem does not cover every operand pair and the claim made here is an EMPIRICAL one: equal to
// the division on 2^31 random pairs on gfx950 (solver.hip), on every solve ever compared with the reference build (the GPU fuzz: 16 197
// solves; every bench run: 64 + 685 sampled), and on a test that runs the recursion with true divisions beside it.  A pair that
// broke it would show as one differing bit in one alpha.  DFTPAV_REF_EXACT_DIV=1 (and any divisor outside [2^-500, 2^500], below)
// takes the true division; the direction of the sweeps' diagonals is checked on the host.  Nothing may under- or overflow on the
// way, which takes a divisor or a dividend beyond 2^+-500.  The divisors are stored quantities: the y . s of a stored pair only has to exceed a `cau` that can be
// tiny, so the solver notes the first one outside [2^-500, 2^500] in its state (iSLOWDIV) and runs the recursion with true
// divisions (EXACT = true) from then on, as the reference does; the LU diagonals of the band system are checked on the host
// (reference_order_tables refuses a system with such a diagonal).  Dividends -- sums of products of O(1e-30 .. 1e20) quantities
// even at the far trial points of a line search -- stay inside that range by a hundred orders of magnitude and are not tested
// (a per-division range test cost 9 % of the sweeps and 7 % of the recursion); the one observable difference of the
// reciprocal route is the sign of a zero: -0.0 / b for b > 0 comes out as +0.0.
template <bool EXACT = false>
__device__ __forceinline__ double div_by_rcp(double a, double b, double y) {
  if (EXACT) return a / b;
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  return __builtin_fma(r, y, q0);
}
// is the reciprocal route good for this divisor?
__host__ __device__ __forceinline__ bool rcp_route_ok(double b) {
  const double ab = __builtin_fabs(b);
  return ab >= 0x1p-500 && ab <= 0x1p500;
}
template <int CTRL> __device__ __forceinline__ double mov_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// max over the 64 lanes (order-free: the maximum has no rounding), the same value in every lane
__device__ __forceinline__ double wave_max64(double v) {
  v = fmax(v, mov_dpp<0xB1>(v));
  v = fmax(v, mov_dpp<0x4E>(v));
  v = fmax(v, mov_dpp<0x141>(v));
  v = fmax(v, mov_dpp<0x140>(v));
  {
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
  }
  {
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
  }
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// scalar solver state (as solver.hip keeps it)
enum { sFX = 0, sFINIT, sDGINIT, sDGTEST, sDSTEST, sMU, sNU, sSTP, sSTEP, sF, sPF0 /* ..+7 */, sGDT = 18, sCOST0, sCOST2, sENERGY, sNUM = 24 };
enum { iCOUNT = 0, iBRACKT, iTOUCHED, iK, iEND, iBOUND, iEVALS, iRET, iPHASE, iACTION, iHISTLO, iHISTHI, iSLOWDIV, iNUM = 16 };
enum { kActEval = 0, kActDone = 1 };

// optional in-kernel phase timer (thread 0, shader clock; DevBatch::prof == nullptr turns it off), slots as solver.hip's:
// 0 right-hand side + BandedSystem::solve, 1 coefficients + jerk terms, 2 constraint points, 3 numbering + chains,
// 4 calGrads_PT (solveAdj), 5 gradient assembly, 6 line search, 7 history update, 8 two-loop recursion
struct Prof {
  long long *acc;
  long long last;
  bool on;
  __device__ inline void start(bool leader, long long *row, bool resume) { // leader: the team's timing lane, profiling on
    on = leader;
    acc = row;
    if (on && !resume)
      for (int i = 0; i < 12; i++) acc[i] = 0;
    last = on ? clock64() : 0;
  }
  __device__ inline void count(int i, long long v) {
    if (on) acc[i] += v;
  }
  __device__ inline void tick(int i) {
    if (on) {
      const long long t = clock64();
      acc[i] += t - last;
      last = t;
    }
  }
};

constexpr int kRec = 16;            // doubles per term record: 12 entries of gdC, gdT, cost, 2 more gdT addends of a moving-obstacle term
constexpr int kListCapTeam = 1024;  // active terms chained per window (TEAM shape)
constexpr int kSerialMax = 512;     // WAVE shape: evaluations with up to this many active terms chain them in one pass on 16 lanes
constexpr int kRecWave = 48;        // WAVE shape: records kept in LDS per evaluation (LDS is what limits the trajectories per CU)

typedef unsigned long long mask_t;  // active terms of a constraint point, bit t = term t (5 H + S + 4 <= 64 terms)
typedef unsigned short __attribute__((address_space(3))) *ldsh_t;

// One lane, one dimension: a substitution sweep over the 6N rows of the band system, row by row.  Row i (ascending
// sweeps: i = 0, 1, ...; descending: i = 6N-1, ...) takes its updates  acc -= c[k] * (result of the k-th row of its
// window)  in the order the reference's column loops apply them to it (k = 0..5; ascending: rows i-6 .. i-1, descending:
// rows i+6 .. i+1), skipping exact zeros as the reference does (`if (a != 0.0)`), then -- sweeps 1 and 2 -- divides by the
// diagonal.  The six previous results live in registers (rows are taken six at a time, so the window is indexed
// statically).  The LU factors of the MINCO band are sparse (3.2 non-zeros per row of L, 1.75 of U) and away from the two
// ends of the system the pattern repeats with the pieces: kInterior_(sweep, i mod 6) below (the host checks it against the
// factors it uploads, capi.cpp: reference_order_tables), so the rows of the middle blocks compute their non-zero terms only,
// without a test; the first and the last block test every coefficient.
// Table row of a sweep: the six coefficients, then (diagonal, 1 / diagonal).
//   sweep 0: solve, forward (L)   1: solve, backward (U, / diagonal)   2: solveAdj, forward (U^T, / diagonal)   3: solveAdj, backward (L^T)
__host__ __device__ constexpr int kInterior_(int q, int r) {
  constexpr int t[4][6] = {{0x3f, 0x1f, 0x0f, 0x00, 0x00, 0x3e}, {0x00, 0x18, 0x30, 0x31, 0x21, 0x06}, {0x00, 0x00, 0x00, 0x35, 0x3b, 0x30}, {0x03, 0x07, 0x0f, 0x1e, 0x3c, 0x38}};
  return t[q][r];
}
typedef double __attribute__((ext_vector_type(2))) v2d_t;
typedef const v2d_t __attribute__((address_space(3))) *ldscv2_t;
// The table of one sweep of a segment of N pieces, blocks of six rows in the order the sweep TRAVERSES them (descending
// sweeps: row n6-1 first):
//   block 0 and block N-1 (the ends of the system): six rows of 8 doubles -- the six coefficients, the diagonal, 1 / diagonal;
//   blocks 1 .. N-2 (the interior): pk_size(Q) doubles -- only the coefficients the pattern kInterior_(Q, .) keeps, in (row, k)
//   order, then (diagonal, 1 / diagonal) of the six rows for the sweeps that divide.
// 384 + 88 (N - 2) doubles per segment instead of 192 N: 12.9 KB instead of 24.6 KB for 16 pieces, and 10-12 LDS reads per
// interior block instead of 24.
__host__ __device__ constexpr int pk_popc6(int m) { return (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1) + ((m >> 4) & 1) + ((m >> 5) & 1); }
__host__ __device__ constexpr int pk_mask(int Q, int r) { // traversal row r of an interior block
  return kInterior_(Q, (Q == 1 || Q == 3) ? 5 - r : r);
}
__host__ __device__ constexpr int pk_off(int Q, int r, int k) { // position of coefficient k of traversal row r inside the block
  int o = 0;
  for (int rr = 0; rr < r; rr++) o += pk_popc6(pk_mask(Q, rr));
  for (int kk = 0; kk < k; kk++) o += (pk_mask(Q, r) >> kk) & 1;
  return o;
}
__host__ __device__ constexpr int pk_ncoef(int Q) { return pk_off(Q, 6, 0); }
__host__ __device__ constexpr int pk_diag0(int Q) { return (pk_ncoef(Q) + 1) & ~1; } // (diagonal, 1 / diagonal) pairs start on an even slot
__host__ __device__ constexpr int pk_size(int Q) { return (Q == 1 || Q == 2) ? pk_diag0(Q) + 12 : ((pk_ncoef(Q) + 1) & ~1); }
__host__ __device__ constexpr int pk_sweep_doubles(int Q, int N) { return 96 + (N > 2 ? (N - 2) * pk_size(Q) : 0); }
__host__ __device__ constexpr int pk_sweep_offset(int Q, int N) { // start of sweep Q inside a segment's tables
  int o = 0;
  for (int q = 0; q < Q; q++) o += pk_sweep_doubles(q, N);
  return o;
}
__host__ __device__ constexpr int pk_segment_doubles(int N) { return pk_sweep_offset(4, N); }

struct SweepBlk { // a block of the ends: whole rows
  v2d_t c[6][4]; // (c0,c1) (c2,c3) (c4,c5) (diagonal, 1 / diagonal)
  double bi[6];
};
template <int Q> struct PackBlk { // an interior block
  v2d_t c[pk_size(Q) / 2];
  double bi[6];
  __device__ __forceinline__ double at(int o) const { return (o & 1) ? c[o >> 1].y : c[o >> 1].x; }
};
template <int Q>
__device__ __forceinline__ void load_end(SweepBlk &R, ldscd_t blk, ldscd_t b, int n6, int d, int i0) {
  constexpr bool DESC = Q == 1 || Q == 3;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int i = DESC ? n6 - 1 - (i0 + r) : i0 + r;
    const ldscv2_t a = (ldscv2_t)(blk + 8 * r);
#pragma unroll
    for (int q = 0; q < 4; q++) R.c[r][q] = a[q];
    R.bi[r] = b[2 * i + d];
  }
}
template <int Q>
__device__ __forceinline__ void load_pack(PackBlk<Q> &R, ldscd_t blk, ldscd_t b, int n6, int d, int i0) {
  constexpr bool DESC = Q == 1 || Q == 3;
  const ldscv2_t a = (ldscv2_t)blk;
#pragma unroll
  for (int q = 0; q < pk_size(Q) / 2; q++) R.c[q] = a[q];
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int i = DESC ? n6 - 1 - (i0 + r) : i0 + r;
    R.bi[r] = b[2 * i + d];
  }
}
// six rows of an end block: every coefficient is tested, as the reference does (`if (a != 0.0)`)
template <int Q>
__device__ __forceinline__ void rows_end(const SweepBlk &R, ldsd_t b, int n6, int d, int i0, double (&w)[6]) {
  constexpr bool DESC = Q == 1 || Q == 3, DIV = Q == 1 || Q == 2;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int i = DESC ? n6 - 1 - (i0 + r) : i0 + r;
    double acc = R.bi[r];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double ck = (k & 1) ? R.c[r][k >> 1].y : R.c[r][k >> 1].x;
      const double t = ck * w[(r + k) % 6];
      acc = ck != 0.0 ? acc - t : acc;
    }
    if (DIV) acc = div_by_rcp(acc, R.c[r][3].x, R.c[r][3].y);
    w[r] = acc;
    b[2 * i + d] = acc;
  }
}
// six rows of an interior block: the non-zero terms only, no test
template <int Q>
__device__ __forceinline__ void rows_pack(const PackBlk<Q> &R, ldsd_t b, int n6, int d, int i0, double (&w)[6]) {
  constexpr bool DESC = Q == 1 || Q == 3, DIV = Q == 1 || Q == 2;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int i = DESC ? n6 - 1 - (i0 + r) : i0 + r;
    constexpr int dummy = 0;
    (void)dummy;
    double acc = R.bi[r];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (pk_mask(Q, r) & (1 << k)) acc = acc - R.at(pk_off(Q, r, k)) * w[(r + k) % 6];
    if (DIV) acc = div_by_rcp(acc, R.at(pk_diag0(Q) + 2 * r), R.at(pk_diag0(Q) + 2 * r + 1));
    w[r] = acc;
    b[2 * i + d] = acc;
  }
}
// tab: this sweep's table (see above); b: the right-hand side / solution [n6][2]; d: the lane's dimension
template <int Q>
__device__ __forceinline__ void sweep(ldscd_t tab, ldsd_t b, int n6, int d) {
  double w[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const int N = n6 / 6;
  SweepBlk E;
  load_end<Q>(E, tab, b, n6, d, 0);
  ldscd_t ip = tab + 48; // interior blocks
  PackBlk<Q> A, B;
  if (N > 2) load_pack<Q>(A, ip, b, n6, d, 6);
  rows_end<Q>(E, b, n6, d, 0, w);
  // interior blocks 1 .. N-2, two per turn; the next block is requested before a block is worked on
  int k = 1;
  for (; k + 1 <= N - 2; k += 2) {
    load_pack<Q>(B, ip + (k) * pk_size(Q), b, n6, d, 6 * (k + 1));
    rows_pack<Q>(A, b, n6, d, 6 * k, w);
    if (k + 2 <= N - 2) load_pack<Q>(A, ip + (k + 1) * pk_size(Q), b, n6, d, 6 * (k + 2));
    else load_end<Q>(E, ip + (N - 2) * pk_size(Q), b, n6, d, n6 - 6);
    rows_pack<Q>(B, b, n6, d, 6 * (k + 1), w);
  }
  if (k <= N - 2) { // one interior block left (A holds it)
    load_end<Q>(E, ip + (N - 2) * pk_size(Q), b, n6, d, n6 - 6);
    rows_pack<Q>(A, b, n6, d, 6 * k, w);
  } else if (N <= 2) {
    load_end<Q>(E, ip, b, n6, d, n6 - 6);
  }
  rows_end<Q>(E, b, n6, d, n6 - 6, w);
}

// positiveSmoothedL1, traj_optimizer.cpp:783-806
__device__ __forceinline__ void smoothed_l1(double x, double &f, double &df) {
  const double pe = 1.0e-4;
  const double half = 0.5 * pe;
  const double f3c = 1.0 / (pe * pe);
  const double f4c = -0.5 * f3c / pe;
  const double d2c = 3.0 * f3c;
  const double d3c = 4.0 * f4c;
  if (x < pe) {
    f = (f4c * x + f3c) * x * x * x;
    df = (d3c * x + d2c) * x * x;
  } else {
    f = x - half;
    df = 1.0;
  }
}

// ------------------------------------------------ moving obstacles: dynamicObsGradCostP (traj_optimizer.cpp:1311-1684)
// Obstacle trajectories as the reference evaluates them (poly_traj_utils.hpp:77-112, 179-211, 510-528): the walk of
// locatePieceIdx, Horner-free power sums, getR / getRdot.  coeffs: 2 x 6 column-major, column 0 multiplies t^5.
struct SurTraj {
  const double *durs, *coeffs;
  int n_pieces;
  double duration, start_time;
};
__device__ inline void piece_getPos(const double *cm, double t, double out[2]) {
  out[0] = 0.0;
  out[1] = 0.0;
  double tn = 1.0;
  for (int i = 5; i >= 0; i--) {
    out[0] += tn * cm[2 * i + 0];
    out[1] += tn * cm[2 * i + 1];
    tn *= t;
  }
}
__device__ inline void piece_getdSigma(const double *cm, double t, double out[2]) {
  out[0] = 0.0;
  out[1] = 0.0;
  double tn = 1.0;
  int n = 1;
  for (int i = 4; i >= 0; i--) {
    out[0] += n * tn * cm[2 * i + 0];
    out[1] += n * tn * cm[2 * i + 1];
    tn *= t;
    n++;
  }
}
__device__ inline void piece_getddSigma(const double *cm, double t, double out[2]) {
  out[0] = 0.0;
  out[1] = 0.0;
  double tn = 1.0;
  int m = 1, n = 2;
  for (int i = 3; i >= 0; i--) {
    out[0] += m * n * tn * cm[2 * i + 0];
    out[1] += m * n * tn * cm[2 * i + 1];
    tn *= t;
    m++;
    n++;
  }
}
__device__ inline void piece_getR(const double *cm, double t, double R[4]) {
  double v[2];
  piece_getdSigma(cm, t, v);
  const double nv = sqrt(v[0] * v[0] + v[1] * v[1]);
  const int singul = 1; // obstacle trajectories are built with getTraj(1), traj_manager.cpp:775
  R[0] = singul * v[0] / nv;
  R[1] = singul * -v[1] / nv;
  R[2] = singul * v[1] / nv;
  R[3] = singul * v[0] / nv;
}
__device__ inline void piece_getRdot(const double *cm, double t, double Rd[4]) {
  double v[2], a[2];
  piece_getdSigma(cm, t, v);
  piece_getddSigma(cm, t, a);
  const double nv = sqrt(v[0] * v[0] + v[1] * v[1]);
  const double nv3 = crt::cube_cr(nv); // the reference: pow(nv, 3)
  const double va = v[0] * a[0] + v[1] * a[1];
  const int singul = 1;
  const double ta[4] = {a[0], -a[1], a[1], a[0]};
  const double tv[4] = {v[0], -v[1], v[1], v[0]};
  for (int k = 0; k < 4; k++) Rd[k] = singul * (ta[k] / nv - tv[k] / nv3 * va);
}
__device__ inline int traj_locate(const double *durs, int N, double &t) { // Trajectory::locatePieceIdx
  int idx;
  double dur;
  for (idx = 0; idx < N && t > (dur = durs[idx]); idx++) t -= dur;
  if (idx == N) {
    idx--;
    t += durs[idx];
  }
  return idx;
}
__device__ inline void traj_getPos(const SurTraj *s, double t, double o[2]) {
  const int i = traj_locate(s->durs, s->n_pieces, t);
  piece_getPos(s->coeffs + 12 * i, t, o);
}
__device__ inline void traj_getdSigma(const SurTraj *s, double t, double o[2]) {
  const int i = traj_locate(s->durs, s->n_pieces, t);
  piece_getdSigma(s->coeffs + 12 * i, t, o);
}
__device__ inline void traj_getddSigma(const SurTraj *s, double t, double o[2]) {
  const int i = traj_locate(s->durs, s->n_pieces, t);
  piece_getddSigma(s->coeffs + 12 * i, t, o);
}
__device__ inline void traj_getR(const SurTraj *s, double t, double R[4]) {
  const int i = traj_locate(s->durs, s->n_pieces, t);
  piece_getR(s->coeffs + 12 * i, t, R);
}
__device__ inline void traj_getRdot(const SurTraj *s, double t, double R[4]) {
  const int i = traj_locate(s->durs, s->n_pieces, t);
  piece_getRdot(s->coeffs + 12 * i, t, R);
}
// 2 x 2 helpers, m = {m00, m01, m10, m11}
__device__ inline void mat_vec(const double m[4], const double v[2], double o[2]) {
  o[0] = m[0] * v[0] + m[1] * v[1];
  o[1] = m[2] * v[0] + m[3] * v[1];
}
__device__ inline void mat_mat(const double a[4], const double b[4], double o[4]) {
  o[0] = a[0] * b[0] + a[1] * b[2];
  o[1] = a[0] * b[1] + a[1] * b[3];
  o[2] = a[2] * b[0] + a[3] * b[2];
  o[3] = a[2] * b[1] + a[3] * b[3];
}
// the piece Trajectory::locatePieceIdx stops at, from the host's theta table (device_types.h: DevSurround::theta; traj_math.h:
// sur_index): the number of pieces k of obstacle u with t > theta[k]
__device__ inline int ref_sur_index(const DevSurround &S, int u, double t) {
  const int p0 = S.piece_off[u], np = S.piece_off[u + 1] - p0;
  const double *th = S.theta + p0;
  int g = (int)(t * S.rate(u));
  g = g < 0 ? 0 : (g > np - 1 ? np - 1 : g);
  const double ta = th[g], tb = th[g > 0 ? g - 1 : 0];
  if (t > ta) {
    g++;
    while (g < np && t > th[g]) g++;
  } else if (g > 0 && !(t > tb)) {
    g--;
    while (g > 0 && !(t > th[g - 1])) g--;
  }
  return g;
}
// The local time locatePieceIdx leaves for a time t that stops at piece idx (idx from the theta table): the reference's subtractions
// t -= duration_0, t -= duration_1, ... in their order -- but with the durations requested in blocks in front of them instead of one
// dependent load per comparison (the comparisons only decide where the walk stops, and that is known).  traj_math.h: sur_local.
__device__ inline int ref_sur_local(const double *durs, int np, int idx, double &t) {
  const int nsub = idx < np ? idx : np;
  int k = 0;
  for (; k + 8 <= nsub; k += 8) {
    const double d0 = durs[k], d1 = durs[k + 1], d2 = durs[k + 2], d3 = durs[k + 3], d4 = durs[k + 4], d5 = durs[k + 5], d6 = durs[k + 6], d7 = durs[k + 7];
    t -= d0; t -= d1; t -= d2; t -= d3; t -= d4; t -= d5; t -= d6; t -= d7;
  }
  for (; k + 4 <= nsub; k += 4) {
    const double d0 = durs[k], d1 = durs[k + 1], d2 = durs[k + 2], d3 = durs[k + 3];
    t -= d0; t -= d1; t -= d2; t -= d3;
  }
  for (; k < nsub; k++) t -= durs[k];
  if (idx == np) {
    idx--;
    t += durs[idx];
  }
  return idx;
}
// log_sum_exp, traj_optimizer.cpp:1686-1707 (mutates all_dists into the exp weights); exp / log correctly rounded
__device__ inline double lse_cr(double alpha, double *all_dists, int n, double *exp_sum) {
  double d_0 = all_dists[0];
  if (alpha > 0) {
    for (int j = 1; j < n; j++)
      if (all_dists[j] > d_0) d_0 = all_dists[j];
  } else {
    for (int j = 1; j < n; j++)
      if (all_dists[j] < d_0) d_0 = all_dists[j];
  }
  *exp_sum = 0;
  for (int j = 0; j < n; j++) {
    all_dists[j] = crt::exp_cr(alpha * (all_dists[j] - d_0));
    *exp_sum += all_dists[j];
  }
  return crt::log_cr(*exp_sum) / alpha + d_0;
}
// The obstacle loop of dynamicObsGradCostP for one constraint point, statement by statement.  Every obstacle with a positive
// penalty writes a record (term t_first + sur_id) and sets its bit in `mask`; the point's penalty -- the inner sum over the
// obstacles, which the reference adds to costs(1) once per point -- goes into slot [13] of the first such record.
// trajtime: what the reference passes for gear segment trajid, trajtimes[trajid] = 0 for the first segment and the DURATION OF
// THE PREVIOUS SEGMENT (not the time since the start) for the others (traj_optimizer.cpp:230-234, 291, 1367-1369).
// One obstacle of the loop of dynamicObsGradCostP for one constraint point, statement by statement.  Returns 0 when the reference
// `continue`s (the distance gate, the bound, costp <= 0); else writes the pair's record (term t_first + sur_id; slot [13] = 0.0) and
// returns 1 with pen = what the point's penalty gets from this obstacle.  GATE_ONLY: stops after the last test that needs no exponential
// (1: the pair goes on, nothing written) -- the TEAM shape collects the pairs that pass and evaluates them densely packed.
template <bool GATE_ONLY>
__device__ __forceinline__ int surround_one(const DevParams &P, const DevSurround &S, int sur_id, double t_now, double omg, double step, double t,
                                            const double beta0[6], const double beta1[6], double gama, int pieceid, int trajres, const double sigma[2],
                                            const double dsigma[2], const double ddsigma[2], const double ego_R[4], int singul_, int trajid, double trajtime,
                                            int Nseg, int t_first, gd_t rec, double &pen) {
  const double B_h[4] = {0.0, -1.0, 1.0, 0.0}, B_hT[4] = {0.0, 1.0, -1.0, 0.0}; // traj_optimizer.cpp:1741-1742
  const double alpha = 100.0, d_min = P.surround_clearance + crt::log_cr(8.0) / alpha; // traj_optimizer.cpp:1336 (the reference: std::log(8.0))
  double temp0 = sqrt(dsigma[0] * dsigma[0] + dsigma[1] * dsigma[1]);
  double temp0_reci = (temp0 != 0.0) ? 1.0 / temp0 : 0.0;
  double temp3 = temp0_reci * temp0_reci;
  const int nE = 4, nO = 4;
  pen = 0.0;
  {
    const SurTraj st_{S.durations + S.piece_off[sur_id], S.coeffs + 12 * (size_t)S.piece_off[sur_id], S.piece_off[sur_id + 1] - S.piece_off[sur_id], S.total[sur_id], S.start[sur_id]};
    const SurTraj *st = &st_;
    double offsettime = t_now - st->start_time + trajtime; // OPT:1367-1369
    double pt_time = offsettime + t;
    double surround_p[2], surround_v[2], surround_a[2];
    // Every evaluator the reference calls at pt_time (getPos, getdSigma, getddSigma, getR, getRdot) starts with the same
    // locatePieceIdx walk: it is taken ONCE here (round 6; seven walks of up to 30 dependent subtractions per pair until then) --
    // the same piece, the same local time, the same bits.  And before any walk: the obstacle's piece at pt_time is known from the
    // host's theta table (the index the walk stops at, found by bisection over the doubles with that same walk), and a point
    // farther than the gate's radius from the box that holds that whole piece fails the reference's distance gate (:1393) for
    // certain -- the pair is dropped as the reference drops it, without evaluating anything (solver.hip does the same).
    double tloc = pt_time;
    int iloc = -1;
    if (pt_time < st->duration) {
      if (S.has_theta() && S.has_bbox() && pt_time >= 0.0) {
        const int ig = ref_sur_index(S, sur_id, pt_time);
        if (ig < st->n_pieces && S.far_from_piece(S.piece_off[sur_id] + ig, sigma, P.veh_length_infl * 1.5 + 1e-6)) return 0;
        iloc = ref_sur_local(st->durs, st->n_pieces, ig, tloc); // (the walk's subtractions without its dependent loads)
      } else {
        iloc = traj_locate(st->durs, st->n_pieces, tloc);
      }
      piece_getPos(st->coeffs + 12 * iloc, tloc, surround_p);
      piece_getdSigma(st->coeffs + 12 * iloc, tloc, surround_v);
      piece_getddSigma(st->coeffs + 12 * iloc, tloc, surround_a);
    } else { // OPT:1379-1389
      double vd[2], pd[2];
      double tend = st->duration;
      const int iend = traj_locate(st->durs, st->n_pieces, tend); // (one walk for the three evaluators at the obstacle's end)
      piece_getddSigma(st->coeffs + 12 * iend, tend, surround_a);
      double exceed_time = pt_time - st->duration;
      piece_getdSigma(st->coeffs + 12 * iend, tend, vd);
      surround_v[0] = vd[0] + exceed_time * surround_a[0];
      surround_v[1] = vd[1] + exceed_time * surround_a[1];
      piece_getPos(st->coeffs + 12 * iend, tend, pd);
      surround_p[0] = pd[0] + exceed_time * vd[0] + 0.5 * surround_a[0] * exceed_time * exceed_time;
      surround_p[1] = pd[1] + exceed_time * vd[1] + 0.5 * surround_a[1] * exceed_time * exceed_time;
    }
    {
      double dx = surround_p[0] - sigma[0], dy = surround_p[1] - sigma[1];
      if (sqrt(dx * dx + dy * dy) > P.veh_length_infl * 1.5) return 0; // OPT:1393
    }
    double surround_R[4];
    if (iloc < 0) iloc = traj_locate(st->durs, st->n_pieces, tloc); // (pt_time beyond the obstacle's duration: the last piece, extrapolated)
    piece_getR(st->coeffs + 12 * iloc, tloc, surround_R); // OPT:1410: getR(pt_time)

    double surround2ego_sum_exp_vec[4], d_U[4], d_U_tilde[4], d_E_tilde[4];
    double ego_normal[4][2], vec_d_Uo_e[4][4], F_delta_le_v[4][4], F_le_v[4][4];
    for (int e = 0; e < nE; e++) { // OPT:1417-1461
      const double *le = P.vec_le[e];
      double delta_le[2] = {P.vec_le[e + 1][0] - le[0], P.vec_le[e + 1][1] - le[1]};
      double delta_le_norm = sqrt(delta_le[0] * delta_le[0] + delta_le[1] * delta_le[1]);
      double delta_le_norm_inverse = 1 / delta_le_norm;
      double Rdl[2], Rle[2];
      mat_vec(ego_R, delta_le, Rdl);
      mat_vec(ego_R, le, Rle);
      // F(l) = singul*[l,Bl]^T*temp0_reci - dsigma*(R l)^T*temp3
      {
        double LT[4] = {delta_le[0], delta_le[1], -delta_le[1], delta_le[0]};
        double *F = F_delta_le_v[e];
        F[0] = singul_ * LT[0] * temp0_reci - dsigma[0] * Rdl[0] * temp3;
        F[1] = singul_ * LT[1] * temp0_reci - dsigma[0] * Rdl[1] * temp3;
        F[2] = singul_ * LT[2] * temp0_reci - dsigma[1] * Rdl[0] * temp3;
        F[3] = singul_ * LT[3] * temp0_reci - dsigma[1] * Rdl[1] * temp3;
      }
      {
        double LT[4] = {le[0], le[1], -le[1], le[0]};
        double *F = F_le_v[e];
        F[0] = singul_ * LT[0] * temp0_reci - dsigma[0] * Rle[0] * temp3;
        F[1] = singul_ * LT[1] * temp0_reci - dsigma[0] * Rle[1] * temp3;
        F[2] = singul_ * LT[2] * temp0_reci - dsigma[1] * Rle[0] * temp3;
        F[3] = singul_ * LT[3] * temp0_reci - dsigma[1] * Rle[1] * temp3;
      }
      double BR[4], H_tilde[2];
      mat_mat(B_h, ego_R, BR);
      mat_vec(BR, delta_le, H_tilde);
      H_tilde[0] *= delta_le_norm_inverse;
      H_tilde[1] *= delta_le_norm_inverse;
      ego_normal[e][0] = H_tilde[0];
      ego_normal[e][1] = H_tilde[1];
      double w[2] = {surround_p[0] - sigma[0] - Rle[0], surround_p[1] - sigma[1] - Rle[1]};
      double d_U_e_tilde = H_tilde[0] * w[0] + H_tilde[1] * w[1];
      double HtR[2] = {H_tilde[0] * surround_R[0] + H_tilde[1] * surround_R[2],
                       H_tilde[0] * surround_R[1] + H_tilde[1] * surround_R[3]};
      for (int o = 0; o < nO; o++) {
        const double *lo = P.vec_le[o];
        vec_d_Uo_e[e][o] = HtR[0] * lo[0] + HtR[1] * lo[1];
      }
      d_U_tilde[e] = d_U_e_tilde; // (its log_sum_exp follows the bound below: the same values in another instruction order)
    }

    double ego2surround_sum_exp_vec[4], d_E[4];
    double surround_normal[4][2], vec_d_Ee_o[4][4];
    for (int o = 0; o < nO; o++) { // OPT:1464-1496
      const double *lo = P.vec_le[o];
      double delta_lo[2] = {P.vec_le[o + 1][0] - lo[0], P.vec_le[o + 1][1] - lo[1]};
      double delta_lo_norm = sqrt(delta_lo[0] * delta_lo[0] + delta_lo[1] * delta_lo[1]);
      double delta_lo_norm_inverse = 1 / delta_lo_norm;
      double BR[4], H_tilde[2], Rlo[2];
      mat_mat(B_h, surround_R, BR);
      mat_vec(BR, delta_lo, H_tilde);
      H_tilde[0] *= delta_lo_norm_inverse;
      H_tilde[1] *= delta_lo_norm_inverse;
      surround_normal[o][0] = H_tilde[0];
      surround_normal[o][1] = H_tilde[1];
      mat_vec(surround_R, lo, Rlo);
      double w[2] = {sigma[0] - surround_p[0] - Rlo[0], sigma[1] - surround_p[1] - Rlo[1]};
      double d_E_o_tilde = H_tilde[0] * w[0] + H_tilde[1] * w[1];
      double HtR[2] = {H_tilde[0] * ego_R[0] + H_tilde[1] * ego_R[2], H_tilde[0] * ego_R[1] + H_tilde[1] * ego_R[3]};
      for (int e = 0; e < nE; e++) {
        const double *le = P.vec_le[e];
        vec_d_Ee_o[o][e] = HtR[0] * le[0] + HtR[1] * le[1];
      }
      d_E_tilde[o] = d_E_o_tilde;
    }
    {
      // A bound before any exponential (the correctly rounded ones are double-double series).  With m_k = min_j v_kj:
      // log_sum_exp(-alpha, v_k) lies in [m_k - ln 4 / alpha, m_k] and log_sum_exp(alpha, d) >= max_k d_k, hence
      //     d_value_test = d_min - log_sum_exp(alpha, d_test)  <=  d_min + ln 4 / alpha - max_k (m_k + t_k);
      // below -1e-9 (the roundings of the full evaluation are 1e-14) the reference's `if (costp <= 0) continue` is taken.
      double best = -1.0e300;
      for (int k = 0; k < 4; k++) {
        double mU = vec_d_Uo_e[k][0], mE = vec_d_Ee_o[k][0];
        for (int j = 1; j < 4; j++) {
          mU = vec_d_Uo_e[k][j] < mU ? vec_d_Uo_e[k][j] : mU;
          mE = vec_d_Ee_o[k][j] < mE ? vec_d_Ee_o[k][j] : mE;
        }
        const double a = mU + d_U_tilde[k], b = mE + d_E_tilde[k];
        best = a > best ? a : best;
        best = b > best ? b : best;
      }
      if (d_min + 1.38629436111989061883e+00 / alpha - best < -1.0e-9) return 0;
    }
    if (GATE_ONLY) return 1; // (the cheap part ends here: the pair goes on to its forty exponentials)
    for (int e = 0; e < nE; e++) {
      double exp_sum;
      d_U[e] = lse_cr(-alpha, vec_d_Uo_e[e], nO, &exp_sum) + d_U_tilde[e];
      surround2ego_sum_exp_vec[e] = exp_sum;
    }
    for (int o = 0; o < nO; o++) {
      double exp_sum;
      d_E[o] = lse_cr(-alpha, vec_d_Ee_o[o], nE, &exp_sum) + d_E_tilde[o];
      ego2surround_sum_exp_vec[o] = exp_sum;
    }

    double d_test[8];
    for (int e = 0; e < 4; e++) d_test[e] = d_U[e];
    for (int o = 0; o < 4; o++) d_test[4 + o] = d_E[o];
    double exp_sum_d = 0;
    double d_value_test = d_min - lse_cr(alpha, d_test, 8, &exp_sum_d); // OPT:1498-1502
    double costp = d_value_test;
    if (costp <= 0) return 0;
    double pena, penaD;
    smoothed_l1(costp, pena, penaD);
    pen = omg * step * P.wei_surround * pena; // (what the point's penalty gets from this obstacle; the caller adds them in obstacle order)

    // dG/dsigma, OPT:1511-1523
    double pGs[2] = {0.0, 0.0};
    for (int e = 0; e < nE; e++) {
      double w = d_test[e] / exp_sum_d;
      pGs[0] -= w * (-ego_normal[e][0]);
      pGs[1] -= w * (-ego_normal[e][1]);
    }
    for (int o = 0; o < nO; o++) {
      double w = d_test[o + nE] / exp_sum_d;
      pGs[0] -= w * surround_normal[o][0];
      pGs[1] -= w * surround_normal[o][1];
    }

    // dG/dsigma', OPT:1528-1573
    double pGds[2] = {0.0, 0.0};
    for (int e = 0; e < nE; e++) {
      const double *F_delta_le = F_delta_le_v[e], *F_le = F_le_v[e];
      const double *le = P.vec_le[e];
      double delta_le[2] = {P.vec_le[e + 1][0] - le[0], P.vec_le[e + 1][1] - le[1]};
      double dln = sqrt(delta_le[0] * delta_le[0] + delta_le[1] * delta_le[1]);
      double d_Uo_e_exp_sum = surround2ego_sum_exp_vec[e];
      double Rle[2];
      mat_vec(ego_R, le, Rle);
      double u[2] = {-surround_p[0] + sigma[0] + Rle[0], -surround_p[1] + sigma[1] + Rle[1]};
      double FB[4], t1[2], FlB[4], FlBR[4], t2[2];
      mat_mat(F_delta_le, B_h, FB);
      mat_vec(FB, u, t1);
      mat_mat(F_le, B_h, FlB);
      mat_mat(FlB, ego_R, FlBR);
      mat_vec(FlBR, delta_le, t2);
      double pdU[2] = {(t1[0] - t2[0]) / dln, (t1[1] - t2[1]) / dln};
      double FBT[4];
      mat_mat(F_delta_le, B_hT, FBT);
      for (int o = 0; o < nO; o++) {
        double d_Uo_e = vec_d_Uo_e[e][o];
        double Rlo[2], q[2];
        mat_vec(surround_R, P.vec_le[o], Rlo);
        mat_vec(FBT, Rlo, q);
        q[0] /= dln;
        q[1] /= dln;
        double w = d_Uo_e / d_Uo_e_exp_sum;
        pdU[0] += w * q[0];
        pdU[1] += w * q[1];
      }
      double w = d_test[e] / exp_sum_d;
      pGds[0] -= w * pdU[0];
      pGds[1] -= w * pdU[1];
    }
    for (int o = 0; o < nO; o++) {
      const double *lo = P.vec_le[o];
      double delta_lo[2] = {P.vec_le[o + 1][0] - lo[0], P.vec_le[o + 1][1] - lo[1]};
      double dln = sqrt(delta_lo[0] * delta_lo[0] + delta_lo[1] * delta_lo[1]);
      double d_Ee_o_exp_sum = ego2surround_sum_exp_vec[o];
      double pdE[2] = {0.0, 0.0};
      for (int e = 0; e < nE; e++) {
        const double *F_le = F_le_v[e];
        double d_Ee_o = vec_d_Ee_o[o][e];
        double FB[4], FBR[4], q[2];
        mat_mat(F_le, B_h, FB);
        mat_mat(FB, surround_R, FBR);
        mat_vec(FBR, delta_lo, q);
        q[0] /= dln;
        q[1] /= dln;
        double w = d_Ee_o / d_Ee_o_exp_sum;
        pdE[0] += w * q[0];
        pdE[1] += w * q[1];
      }
      double w = d_test[o + nE] / exp_sum_d;
      pGds[0] -= w * pdE[0];
      pGds[1] -= w * pdE[1];
    }

    // dG/dt_bar, OPT:1578-1580
    double pGtbar = (pGs[0] * dsigma[0] + pGs[1] * dsigma[1]) + (pGds[0] * ddsigma[0] + pGds[1] * ddsigma[1]);

    // dG/dt_hat, OPT:1586-1646
    double pGthat = 0.0;
    double Rud[4];
    piece_getRdot(st->coeffs + 12 * iloc, tloc, Rud); // OPT:1599: getRdot(pt_time)
    for (int e = 0; e < nE; e++) {
      double d_Uo_e_exp_sum = surround2ego_sum_exp_vec[e];
      const double *Hn = ego_normal[e];
      double acc = Hn[0] * surround_v[0] + Hn[1] * surround_v[1];
      double HtRd[2] = {Hn[0] * Rud[0] + Hn[1] * Rud[2], Hn[0] * Rud[1] + Hn[1] * Rud[3]};
      for (int o = 0; o < nO; o++) {
        const double *lo = P.vec_le[o];
        double pt = HtRd[0] * lo[0] + HtRd[1] * lo[1];
        double d_Uo_e = vec_d_Uo_e[e][o];
        acc += d_Uo_e / d_Uo_e_exp_sum * pt;
      }
      pGthat -= d_test[e] / exp_sum_d * acc;
    }
    for (int o = 0; o < nO; o++) {
      double d_Ee_o_exp_sum = ego2surround_sum_exp_vec[o];
      const double *lo = P.vec_le[o];
      double delta_lo[2] = {P.vec_le[o + 1][0] - lo[0], P.vec_le[o + 1][1] - lo[1]};
      double dln = sqrt(delta_lo[0] * delta_lo[0] + delta_lo[1] * delta_lo[1]);
      double BRd[4], BR[4], a1[2], a2[2], Rlo[2], Rdlo[2];
      mat_mat(B_h, Rud, BRd);
      mat_vec(BRd, delta_lo, a1);
      mat_mat(B_h, surround_R, BR);
      mat_vec(BR, delta_lo, a2);
      mat_vec(surround_R, lo, Rlo);
      mat_vec(Rud, lo, Rdlo);
      double w1[2] = {sigma[0] - surround_p[0] - Rlo[0], sigma[1] - surround_p[1] - Rlo[1]};
      double w2[2] = {-surround_v[0] - Rdlo[0], -surround_v[1] - Rdlo[1]};
      double acc = ((a1[0] / dln) * w1[0] + (a1[1] / dln) * w1[1]) + ((a2[0] / dln) * w2[0] + (a2[1] / dln) * w2[1]);
      for (int e = 0; e < nE; e++) {
        double d_Ee_o = vec_d_Ee_o[o][e];
        double Rle[2];
        mat_vec(ego_R, P.vec_le[e], Rle);
        double r1[2] = {Rle[0] * B_h[0] + Rle[1] * B_h[2], Rle[0] * B_h[1] + Rle[1] * B_h[3]};
        double r2[2] = {r1[0] * Rud[0] + r1[1] * Rud[2], r1[0] * Rud[1] + r1[1] * Rud[3]};
        double pt = (r2[0] * delta_lo[0] + r2[1] * delta_lo[1]) / dln;
        acc += d_Ee_o / d_Ee_o_exp_sum * pt;
      }
      pGthat -= d_test[o + nE] / exp_sum_d * acc;
    }

    // accumulate, OPT:1649-1676
    double gradViolaPt = gama * pGtbar;
    double scale = omg * step * P.wei_surround * penaD;
    gd_t r_ = rec + (size_t)(t_first + sur_id) * kRec;
    for (int k = 0; k < 6; k++) {
      r_[2 * k + 0] = scale * (beta0[k] * pGs[0] + beta1[k] * pGds[0]);
      r_[2 * k + 1] = scale * (beta0[k] * pGs[1] + beta1[k] * pGds[1]);
    }
    // the `gdT +=` of traj_optimizer.cpp:1663-1676, kept apart: the chain adds them one after the other -- [12], then
    // [14] * pieceid (the reference's product  omg * step * wei * grad_prev_t * penaD * pieceid  evaluates left to right, so its
    // last factor can be applied by the chain lane), then [15], then `trajid` times [14] * piece_num_container[trajid]
    // (:1674-1676: the loop over the previous segments adds to THIS segment's gdT)
    r_[12] = omg * P.wei_surround * (pena / trajres + penaD * gradViolaPt * step);
    r_[14] = omg * step * P.wei_surround * pGthat * penaD;
    r_[15] = omg * step * P.wei_surround * gama * pGthat * penaD;
    r_[13] = 0.0;
    return 1;
    }
}
// The obstacle loop of dynamicObsGradCostP for one constraint point.  Every obstacle with a positive
// penalty writes a record (term t_first + sur_id) and sets its bit in `mask`; the point's penalty -- the inner sum over the
// obstacles, which the reference adds to costs(1) once per point -- goes into slot [13] of the first such record.
// trajtime: what the reference passes for gear segment trajid, trajtimes[trajid] = 0 for the first segment and the DURATION OF
// THE PREVIOUS SEGMENT (not the time since the start) for the others (traj_optimizer.cpp:230-234, 291, 1367-1369).
__device__ __noinline__ mask_t surround_terms(const DevParams &P, const DevSurround &S, double t_now, double omg, double step, double t,
                                                const double beta0[6], const double beta1[6], double gama, int pieceid, int trajres,
                                                const double sigma[2], const double dsigma[2], const double ddsigma[2], const double ego_R[4],
                                                int singul_, int trajid, double trajtime, int Nseg, int t_first, gd_t rec) {
  mask_t mask = 0ull;
  int first_active = -1;
  double totalPenalty = 0.0;
  for (int sur_id = 0; sur_id < S.S; sur_id++) {
    double pen;
    if (!surround_one<false>(P, S, sur_id, t_now, omg, step, t, beta0, beta1, gama, pieceid, trajres, sigma, dsigma, ddsigma, ego_R, singul_, trajid, trajtime, Nseg,
                             t_first, rec, pen))
      continue;
    totalPenalty += pen;
    if (first_active < 0) first_active = sur_id;
    mask |= (mask_t)1 << (t_first + sur_id);
  }
  if (first_active >= 0) rec[(size_t)(t_first + first_active) * kRec + 13] = totalPenalty;
  return mask;
}

// inclusive prefix sum over the 64 lanes (row_shr 1, 2, 4, 8 inside the rows of 16, then the rows' totals by row_bcast 15 / 31)
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // last lane of rows 0 / 2 onto rows 1 / 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // lane 31 onto rows 2 and 3
  return v;
}

// ------------------------------------------------ one constraint point (traj_optimizer.cpp:499-705)
// What a constraint point keeps between its tests (point_masks: which terms are active) and its records (point_emit: what an
// active term adds).  Most points have no active term at all (24 of 14 784 terms per evaluation on BASELINE configs[2]), so
// everything only a record needs is formed in point_emit.
// (kept small: it is live in every lane across the numbering of a round; what can be formed again from it with the same
// expressions -- the powers of s1, R * vertex, the half-planes themselves -- is)
struct PtState {
  double s1, alpha, omg, step, sg;
  double dsigma[2], ddsigma[2];
  double z_h0 /* 1 / |dsigma| */, z_h1, z_h2, z_h3, z1, z_h4;
  double vel2_reci, vel2_reci_e, vel3_2_reci_e;
  double violaVel, violaAcc, violaCurL, violaCurR;
  double bp0, bp1; // sigma
  int K;
};

// Point j of piece i (K intervals, offset s1 = the running sum of traj_optimizer.cpp:513, taken from the table): the state and
// the mask of active terms -- term v H + k: vertex v against half-plane k (:592-634); 5 H + s: moving obstacle s (:636-638,
// whose records surround_terms writes to `sur_rec` [S][kRec] at once: its test IS its cost); then velocity, acceleration,
// curvature left / right (:642-705).  pl: the point's half-planes (load_planes), (n_x, n_y, p_x, p_y) of plane k at 4 k.
// cor: &corridor[b][0][pt] (component-major, pitch NptsPad); planes past H are never used.
// HMAX: the plane slots a point keeps in registers -- 5 for every kernel of the live path (H = 4 rectangles), 12 for the generic TEAM
// kernel that takes whatever the term mask can number (5 H + S + 4 <= 64)
template <int HMAX = 5>
__device__ __forceinline__ void load_planes(gcd_t cor, size_t pitch, int H, double (&pl)[4 * HMAX]) {
#pragma unroll
  for (int k = 0; k < HMAX; k++) {
#pragma unroll
    for (int q = 0; q < 4; q++) pl[4 * k + q] = k < H ? cor[(size_t)(4 * k + q) * pitch] : 0.0; // (uniform: planes past H are not fetched)
  }
}
// SKIP_SUR: the moving-obstacle terms are left to the caller (the TEAM shape collects the pairs that pass the cheap tests and
// evaluates them densely packed); the term numbers stay those of a layout WITH obstacles
template <bool SUR, int HMAX = 5, bool SKIP_SUR = false>
__device__ __forceinline__ mask_t point_masks(const DevParams &P, const double cc_[12], int i, int N, int j, int K, double step, double s1, int singul_,
                                            double epis, int H, const double (&pl)[4 * HMAX], gd_t sur_rec, const DevSurround &S, double t_now, double t_piece,
                                            int trajid, double trajtime, PtState &st) {
  double cc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) cc[k] = cc_[k];
  const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
  const double beta0[6] = {1.0, s1, s2, s3, s4, s5};
  const double beta1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
  const double beta2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
  const double beta3[6] = {0.0, 0.0, 0.0, 6.0, 24.0 * s1, 60.0 * s2};
  const double alpha = 1.0 / K * j;
  double sigma[2] = {0, 0}, dsigma[2] = {0, 0}, ddsigma[2] = {0, 0}, dddsigma[2] = {0, 0};
#pragma unroll
  for (int k = 0; k < 6; k++)
#pragma unroll
    for (int d = 0; d < 2; d++) {
      sigma[d] += cc[2 * k + d] * beta0[k];
      dsigma[d] += cc[2 * k + d] * beta1[k];
      ddsigma[d] += cc[2 * k + d] * beta2[k];
      dddsigma[d] += cc[2 * k + d] * beta3[k];
    }
  const double omg = (j == 0 || j == K) ? 0.5 : 1.0;
  double z_h0 = sqrt(dsigma[0] * dsigma[0] + dsigma[1] * dsigma[1]);
  const double z_h1 = ddsigma[0] * dsigma[0] + ddsigma[1] * dsigma[1];
  const double z_h2 = dddsigma[0] * dsigma[0] + dddsigma[1] * dsigma[1];
  const double z_h3 = ddsigma[1] * dsigma[0] + (-ddsigma[0]) * dsigma[1];  // ddsigma^T B_h dsigma, :529
  const double z1 = dddsigma[1] * dsigma[0] + (-dddsigma[0]) * dsigma[1];  // :538
  if (z_h0 < 1e-4 || (j == 0 && i == 0) || (i == N - 1 && j == K)) return 0ull; // :550-553

  const double max_vel = singul_ > 0 ? P.max_vel[0] : P.max_vel[1];
  const double max_acc = singul_ > 0 ? P.max_acc[0] : P.max_acc[1];
  const double max_cur = singul_ > 0 ? P.max_cur[0] : P.max_cur[1];
  const double sg = (double)singul_;

  const double vel2_reci = 1.0 / (z_h0 * z_h0);
  // (epis is 0.0 on the live path, traj_manager.cpp:610: x + 0.0 == x for every x >= 0, so the second quotient is the first)
  const double vel2_reci_e = epis == 0.0 ? vel2_reci : 1.0 / (z_h0 * z_h0 + epis);
  const double vel3_2_reci_e = vel2_reci_e * sqrt(vel2_reci_e);
  z_h0 = 1.0 / z_h0;
  const double z_h4 = z_h1 * vel2_reci;
  const double violaVel = 1.0 / vel2_reci - max_vel * max_vel;
  const double acc2 = z_h1 * z_h1 * vel2_reci;
  const double cur = z_h3 * vel3_2_reci_e;
  const double violaAcc = acc2 - max_acc * max_acc;
  const double violaCurL = cur - max_cur;
  const double violaCurR = -cur - max_cur;

  const double ego_R[4] = {sg * dsigma[0] * z_h0, sg * -dsigma[1] * z_h0, sg * dsigma[1] * z_h0, sg * dsigma[0] * z_h0}; // :581-583

  mask_t mask = 0ull;
  // ---- corridor: for (auto le : vec_le_) for (k < corr_k), traj_optimizer.cpp:592-634: the 5 H tests (term v H + k: the order
  // of the reference's nested loops)
  double pn0[HMAX], pn1[HMAX], pq0[HMAX], pq1[HMAX];
#pragma unroll
  for (int k = 0; k < HMAX; k++) {
    pn0[k] = pl[4 * k + 0];
    pn1[k] = pl[4 * k + 1];
    pq0[k] = pl[4 * k + 2];
    pq1[k] = pl[4 * k + 3];
  }
  // (vec_le_ holds the first vertex twice, traj_optimizer.cpp:1765-1775, and the reference tests it twice: the fifth vertex's
  // tests are the first's, expression for expression -- their bits are copied, not recomputed; capi.cpp fills vec_le[4] from
  // vec_le[0].  H <= 5: 5 H <= 25 tests, collected in 32 bits; the generic kernel's H <= 12 takes the 64-bit mask.)
  typename std::conditional<(HMAX > 5), mask_t, unsigned>::type cm = 0;
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const double le0 = P.vec_le[v][0], le1 = P.vec_le[v][1];
    const double rl0 = ego_R[0] * le0 + ego_R[1] * le1;
    const double rl1 = ego_R[2] * le0 + ego_R[3] * le1;
    const double bpt0 = sigma[0] + rl0, bpt1 = sigma[1] + rl1;
#pragma unroll
    for (int k = 0; k < HMAX; k++) {
      const double violaPos = pn0[k] * (bpt0 - pq0[k]) + pn1[k] * (bpt1 - pq1[k]);
      if (k < H && violaPos > 0) cm |= (decltype(cm))1 << (v * H + k);
    }
  }
  cm |= (cm & (((decltype(cm))1 << H) - 1)) << (4 * H);
  mask = (mask_t)cm;
  // ---- moving obstacles, traj_optimizer.cpp:636-638 (terms 5 H .. 5 H + S - 1)
  if (SUR && !SKIP_SUR && S.S > 0)
    mask |= surround_terms(P, S, t_now, omg, step, t_piece + step * j, beta0, beta1, alpha, i, K, sigma, dsigma, ddsigma, ego_R, singul_, trajid, trajtime,
                           N, 5 * H, sur_rec - (size_t)(5 * H) * kRec);
  const int t0 = 5 * H + (SUR ? S.S : 0);
  if (violaVel > 0.0) mask |= (mask_t)1 << t0;        // :642
  if (violaAcc > 0.0) mask |= (mask_t)1 << (t0 + 1);  // :655
  if (violaCurL > 0.0) mask |= (mask_t)1 << (t0 + 2); // :684
  if (violaCurR > 0.0) mask |= (mask_t)1 << (t0 + 3); // :695
  st.s1 = s1; st.alpha = alpha; st.omg = omg; st.step = step; st.sg = sg;
  st.dsigma[0] = dsigma[0]; st.dsigma[1] = dsigma[1]; st.ddsigma[0] = ddsigma[0]; st.ddsigma[1] = ddsigma[1];
  st.z_h0 = z_h0; st.z_h1 = z_h1; st.z_h2 = z_h2; st.z_h3 = z_h3; st.z1 = z1; st.z_h4 = z_h4;
  st.vel2_reci = vel2_reci; st.vel2_reci_e = vel2_reci_e; st.vel3_2_reci_e = vel3_2_reci_e;
  st.violaVel = violaVel; st.violaAcc = violaAcc; st.violaCurL = violaCurL; st.violaCurR = violaCurR;
  st.bp0 = sigma[0]; st.bp1 = sigma[1];
  st.K = K;
  return mask;
}

// The record of an active static term t (not a moving-obstacle term) of a point: what the term adds to gdC (12), gdT [12] and the
// cost [13], exactly the expressions of traj_optimizer.cpp:600-705.  t0 = 5 H + S: the first feasibility term.  R: any pointer
// type (global records of the TEAM shape, LDS / flat records of the WAVE shape).
// planes(k, n0, n1, q0, q1): the half-plane k of this point (what point_masks tested)
template <typename R, typename PF>
__device__ __forceinline__ void point_emit_pf(const DevParams &P, const PtState &st, int t, int H, int t0, PF planes, R r_) {
  const double s1 = st.s1;
  const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1; // the expressions of point_masks: the same bits
  const double beta0[6] = {1.0, s1, s2, s3, s4, s5};
  const double beta1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
  const double beta2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
  const double alpha = st.alpha, omg = st.omg, step = st.step, sg = st.sg;
  const double *dsigma = st.dsigma, *ddsigma = st.ddsigma;
  const double z_h0 = st.z_h0, z_h1 = st.z_h1, z_h2 = st.z_h2, z_h3 = st.z_h3, z1 = st.z1, z_h4 = st.z_h4;
  const double vel2_reci = st.vel2_reci, vel2_reci_e = st.vel2_reci_e, vel3_2_reci_e = st.vel3_2_reci_e;
  const int K = st.K;
  if (t < 5 * H) { // ---- corridor: vertex v against half-plane k (:600-634)
    int v = 0;
#pragma unroll
    for (int q = 1; q < 5; q++) v += t >= q * H ? 1 : 0;
    const int k = t - v * H;
    // the half-plane and the vertex of this term (they are what point_masks tested)
    double on0, on1, q0, q1;
    planes(k, on0, on1, q0, q1);
    double le0 = P.vec_le[0][0], le1 = P.vec_le[0][1];
#pragma unroll
    for (int q = 1; q < 5; q++) {
      le0 = v == q ? P.vec_le[q][0] : le0;
      le1 = v == q ? P.vec_le[q][1] : le1;
    }
    const double ego_R[4] = {sg * dsigma[0] * z_h0, sg * -dsigma[1] * z_h0, sg * dsigma[1] * z_h0, sg * dsigma[0] * z_h0}; // :581-583
    const double Rle0 = ego_R[0] * le0 + ego_R[1] * le1;
    const double Rle1 = ego_R[2] * le0 + ego_R[3] * le1;
    const double temp_a[4] = {ddsigma[0], -ddsigma[1], ddsigma[1], ddsigma[0]};
    const double temp_v[4] = {dsigma[0], -dsigma[1], dsigma[1], dsigma[0]};
    double R_dot[4];
#pragma unroll
    for (int q = 0; q < 4; q++) R_dot[q] = sg * (temp_a[q] * z_h0 - temp_v[q] * vel2_reci * z_h0 * z_h1);
    const double bpt0 = st.bp0 + Rle0, bpt1 = st.bp1 + Rle1;
    const double violaPos = on0 * (bpt0 - q0) + on1 * (bpt1 - q1); // the expression of the test: > 0 here
    const double tl[4] = {le0, -le1, le1, le0};
    double pena, penaD;
    smoothed_l1(violaPos, pena, penaD);
    double Mm[4];
    Mm[0] = sg * tl[0] * z_h0 - Rle0 * dsigma[0] * vel2_reci;
    Mm[1] = sg * tl[1] * z_h0 - Rle0 * dsigma[1] * vel2_reci;
    Mm[2] = sg * tl[2] * z_h0 - Rle1 * dsigma[0] * vel2_reci;
    Mm[3] = sg * tl[3] * z_h0 - Rle1 * dsigma[1] * vel2_reci;
    const double w0 = dsigma[0] + (R_dot[0] * le0 + R_dot[1] * le1);
    const double w1 = dsigma[1] + (R_dot[2] * le0 + R_dot[3] * le1);
    const double gradViolaPt = (alpha * on0) * w0 + (alpha * on1) * w1;
    const double sc = omg * step * P.wei_obs * penaD;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const double b1n0 = beta1[r] * on0, b1n1 = beta1[r] * on1;
      const double g0 = beta0[r] * on0 + (b1n0 * Mm[0] + b1n1 * Mm[2]);
      const double g1 = beta0[r] * on1 + (b1n0 * Mm[1] + b1n1 * Mm[3]);
      r_[2 * r + 0] = sc * g0;
      r_[2 * r + 1] = sc * g1;
    }
    r_[12] = omg * P.wei_obs * (penaD * gradViolaPt * step + pena / K);
    r_[13] = omg * step * P.wei_obs * pena;
    return;
  }
  const int f = t - t0;
  if (f == 0) { // :642-653
    double pena, penaD;
    smoothed_l1(st.violaVel, pena, penaD);
    const double gradViolaVt = 2.0 * alpha * z_h1;
    const double sc = omg * step * P.wei_feas * penaD;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      r_[2 * r + 0] = sc * (2.0 * beta1[r] * dsigma[0]);
      r_[2 * r + 1] = sc * (2.0 * beta1[r] * dsigma[1]);
    }
    r_[12] = omg * P.wei_feas * (penaD * gradViolaVt * step + pena / K);
    r_[13] = omg * step * P.wei_feas * pena;
  } else if (f == 1) { // :655-665
    double pena, penaD;
    smoothed_l1(st.violaAcc, pena, penaD);
    const double u0 = z_h4 * ddsigma[0] - z_h4 * z_h4 * dsigma[0], u1 = z_h4 * ddsigma[1] - z_h4 * z_h4 * dsigma[1];
    const double sqn = ddsigma[0] * ddsigma[0] + ddsigma[1] * ddsigma[1];
    const double gradViolaAt = 2.0 * alpha * (z_h4 * (sqn + z_h2) - z_h4 * z_h4 * z_h1);
    const double sc = omg * step * P.wei_feas * penaD;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      r_[2 * r + 0] = sc * (2.0 * beta1[r] * u0 + 2.0 * beta2[r] * z_h4 * dsigma[0]);
      r_[2 * r + 1] = sc * (2.0 * beta1[r] * u1 + 2.0 * beta2[r] * z_h4 * dsigma[1]);
    }
    r_[12] = omg * P.wei_feas * (penaD * gradViolaAt * step + pena / K);
    r_[13] = omg * step * P.wei_feas * pena;
  } else { // ---- curvature, :684-705 (f == 2: left, f == 3: right)
    const double ku0 = vel3_2_reci_e * ddsigma[1] - 3 * vel3_2_reci_e * vel2_reci_e * z_h3 * dsigma[0];
    const double ku1 = vel3_2_reci_e * -ddsigma[0] - 3 * vel3_2_reci_e * vel2_reci_e * z_h3 * dsigma[1];
    const double kt = alpha * vel3_2_reci_e * (z1 - 3 * vel2_reci_e * z_h3 * z_h1);
    double pena, penaD;
    smoothed_l1(f == 2 ? st.violaCurL : st.violaCurR, pena, penaD);
    const double sc = omg * step * P.wei_feas * 10.0 * penaD;
    if (f == 2) {
#pragma unroll
      for (int r = 0; r < 6; r++) {
        const double kw0 = -((beta2[r] * vel3_2_reci_e) * dsigma[1]), kw1 = (beta2[r] * vel3_2_reci_e) * dsigma[0];
        r_[2 * r + 0] = sc * (beta1[r] * ku0 + kw0);
        r_[2 * r + 1] = sc * (beta1[r] * ku1 + kw1);
      }
      r_[12] = omg * P.wei_feas * 10.0 * (penaD * kt * step + pena / K);
    } else {
#pragma unroll
      for (int r = 0; r < 6; r++) {
        const double kw0 = -((beta2[r] * vel3_2_reci_e) * dsigma[1]), kw1 = (beta2[r] * vel3_2_reci_e) * dsigma[0];
        r_[2 * r + 0] = sc * -(beta1[r] * ku0 + kw0);
        r_[2 * r + 1] = sc * -(beta1[r] * ku1 + kw1);
      }
      r_[12] = omg * P.wei_feas * 10.0 * (penaD * (-kt) * step + pena / K);
    }
    r_[13] = omg * step * P.wei_feas * 10.0 * pena;
  }
}

// One (constraint point, obstacle) pair from the point's kept state: the arguments point_masks hands to surround_terms, formed again with
// the same expressions (the same bits).  lp: the piece's index inside its segment, N: the segment's pieces, t_piece: the piece's start time.
template <bool GATE_ONLY>
__device__ __forceinline__ int point_surround_one(const DevParams &P, const DevSurround &S, const PtState &st, int sur_id, double t_now, int j, int lp, int N,
                                                  double t_piece, int singul_, int trajid, double trajtime, int t_first, gd_t rec, double &pen) {
  const double s1 = st.s1, s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
  const double beta0[6] = {1.0, s1, s2, s3, s4, s5};
  const double beta1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
  const double sigma[2] = {st.bp0, st.bp1};
  const double sg = st.sg, z_h0 = st.z_h0;
  const double ego_R[4] = {sg * st.dsigma[0] * z_h0, sg * -st.dsigma[1] * z_h0, sg * st.dsigma[1] * z_h0, sg * st.dsigma[0] * z_h0};
  return surround_one<GATE_ONLY>(P, S, sur_id, t_now, st.omg, st.step, t_piece + st.step * j, beta0, beta1, st.alpha, lp, st.K, sigma, st.dsigma, st.ddsigma, ego_R,
                                 singul_, trajid, trajtime, N, t_first, rec, pen);
}

// the half-plane fetched again from the corridor (TEAM / WAVE shapes: the point's planes are not kept across the numbering)
template <typename R>
__device__ __forceinline__ void point_emit(const DevParams &P, const PtState &st, int t, int H, int t0, gcd_t cor, size_t pitch, R r_) {
  point_emit_pf(P, st, t, H, t0,
                [&](int k, double &on0, double &on1, double &q0, double &q1) {
                  on0 = cor[(size_t)(4 * k + 0) * pitch];
                  on1 = cor[(size_t)(4 * k + 1) * pitch];
                  q0 = cor[(size_t)(4 * k + 2) * pitch];
                  q1 = cor[(size_t)(4 * k + 3) * pitch];
                },
                r_);
}

// TEAM shape: the point's tests, then a record per active term in the point's own slots rec[t][kRec] (global scratch)
template <bool SUR, int HMAX = 5>
__device__ __forceinline__ mask_t point_terms(const DevParams &P, const double cc_[12], int i, int N, int j, int K, double step, double s1,
                                            int singul_, double epis, int H, gcd_t cor, size_t pitch, gd_t rec, const DevSurround &S,
                                            double t_now, double t_piece, int trajid, double trajtime) {
  PtState st;
  const int nS = SUR ? S.S : 0, tS0 = 5 * H, t0 = tS0 + nS;
  double pl[4 * HMAX];
  load_planes<HMAX>(cor, pitch, H, pl);
  const mask_t mask = point_masks<SUR, HMAX>(P, cc_, i, N, j, K, step, s1, singul_, epis, H, pl, rec + (size_t)tS0 * kRec, S, t_now, t_piece, trajid, trajtime, st);
  for (mask_t m = mask; m;) {
    const int t = __builtin_ctzll(m);
    m &= m - 1;
    if (t >= tS0 && t < t0) continue; // a moving-obstacle term: surround_terms has written its record
    point_emit(P, st, t, H, t0, cor, pitch, rec + (size_t)t * kRec);
  }
  return mask;
}

// ---- the sequential sums as DPP chains (see solver_ref.hip: seq_sum_dpp): v_fmac_f64_dpp ... row_newbcast:K makes every lane of a
// row of 16 add term K of that row -- fma(p, 1.0, acc) is acc + p rounded once, the bits of the addition
#define DFTPAV_FMAC_BCAST(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define DFTPAV_FMAC_BCAST16 \
  DFTPAV_FMAC_BCAST(0) DFTPAV_FMAC_BCAST(1) DFTPAV_FMAC_BCAST(2) DFTPAV_FMAC_BCAST(3) DFTPAV_FMAC_BCAST(4) DFTPAV_FMAC_BCAST(5) DFTPAV_FMAC_BCAST(6) \
  DFTPAV_FMAC_BCAST(7) DFTPAV_FMAC_BCAST(8) DFTPAV_FMAC_BCAST(9) DFTPAV_FMAC_BCAST(10) DFTPAV_FMAC_BCAST(11) DFTPAV_FMAC_BCAST(12) \
  DFTPAV_FMAC_BCAST(13) DFTPAV_FMAC_BCAST(14) DFTPAV_FMAC_BCAST(15)
#define DFTPAV_FMAC_BCAST8 \
  DFTPAV_FMAC_BCAST(0) DFTPAV_FMAC_BCAST(1) DFTPAV_FMAC_BCAST(2) DFTPAV_FMAC_BCAST(3) DFTPAV_FMAC_BCAST(4) DFTPAV_FMAC_BCAST(5) DFTPAV_FMAC_BCAST(6) \
  DFTPAV_FMAC_BCAST(7)

// ------------------------------------------------ the ring of a scheduled launch
// Work ring of a scheduled launch (source 1): DevBatch::queue / qctl, the protocol of solver.hip's queue_pop / queue_push --
// CAS on the head, release / acquire on the published tail, device-scope fences, because the next slice of a trajectory may
// run behind another XCD's L2.  Popped and pushed by lane 0 of a wave.
__device__ inline int ring_pop(unsigned *ctl, const int *ring, int cap) {
  while (true) {
    const unsigned h = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned t = __hip_atomic_load(&ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (h >= t) return -1;
    if (atomicCAS(&ctl[0], h, h + 1) == h) {
      const int id = __hip_atomic_load(&ring[h % (unsigned)cap], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      return id;
    }
  }
}
__device__ inline void ring_push(unsigned *ctl, int *ring, int cap, int id) {
  __threadfence();
  const unsigned t = atomicAdd(&ctl[2], 1u);
  __hip_atomic_store(&ring[t % (unsigned)cap], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) { // publish in reservation order
    unsigned expect = t;
    if (__hip_atomic_compare_exchange_strong(&ctl[1], &expect, t + 1, __ATOMIC_RELEASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
  }
}
// (s, y) pairs of the history, (ys, 1 / ys) of a stored pair
typedef double __attribute__((ext_vector_type(2))) d2_t;
typedef const d2_t __attribute__((address_space(1))) *gcd2_t;
typedef d2_t __attribute__((address_space(1))) *gd2_t;
} // namespace reford
} // namespace dftpav
