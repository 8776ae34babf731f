// solver.hip — gfx950 kernels of the batched MINCO / L-BFGS trajectory solver.
//
// One workgroup owns one trajectory for its whole life: the L-BFGS outer loop,
// the Lewis-Overton line search, every cost/gradient evaluation and the
// two-loop recursion all run inside a single launch with no host round trip.
//
//   * threads <-> constraint points (the sample loop of
//     PolyTrajOptimizer::addPVAGradCost2CT, traj_optimizer.cpp:486-779), a few
//     points per thread so that the workgroup stays at <= 8 waves and every wave
//     gets the full 256-VGPR budget (the per-point mathematics is register hungry
//     in fp64; spilling it costs far more than the second point)
//   * MINCO forward/adjoint banded solves (poly_traj_utils.hpp:805-852) are
//     applied as the precomputed dense operator A_N^{-1}|_{N+5 columns}, staged in
//     LDS: the band matrix depends only on N (poly_traj_utils.hpp:895-947) and
//     only N+5 RHS rows are ever non-zero (poly_traj_utils.hpp:968-977), so both
//     solves become small lane-parallel mat-vecs instead of a 6N-step substitution
//   * per-sample gradients are kept as d/dsigma, d/dsigma', d/dsigma'' (6 values)
//     and expanded onto the 6x2 piece coefficients by a transposed LDS reduction
//     with a fixed summation order
//   * the whole L-BFGS state lives in LDS and wave 0 advances it as a state
//     machine between evaluations (one decision variable per lane, DPP /
//     permlane-swap butterflies for the dot products, history columns streamed
//     from global memory through a register prefetch ring); nothing but LDS
//     offsets is live in registers across an evaluation
//
// All arithmetic is fp64 with contraction off; every sum has a defined order
// that oracle/dftpav_oracle_dev.cpp replays, and the two agree bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "device_types.h"
#include "e4_plan.h"
#include "traj_math.h"

namespace dftpav {

// ------------------------------------------------------------------ LDS carve
// scalar L-BFGS state kept in Smem::st (doubles) and Smem::ist (ints)
enum { sFX = 0, sFINIT, sDGINIT, sDGTEST, sDSTEST, sMU, sNU, sSTP, sSTEP, sF, sPF0 /* .. sPF0+7 */, sNUM = 24 };
enum { iCOUNT = 0, iBRACKT, iTOUCHED, iK, iEND, iBOUND, iEVALS, iRET, iPHASE, iACTION, iHISTLO, iHISTHI, iCUR, iNUM = 16 };
enum { kActEval = 0, kActDone = 1 };

struct Smem {
  double *x, *xp, *g, *gp, *d;
  double *bnd;  // [2][M][6] iniStates, finStates of the trajectory being solved
  double *seg;  // [M][16]  0:T 1:dt 2..7:t^k 8..13:t^-k
  double *spow; // [M][2][Kmax+1] the accumulated sample offsets (s1 += step, traj_optimizer.cpp:513), for K and for Kd
  double *rhs;  // [rhs_tot][2]
  double *b, *c, *gdC; // [6*Ntot][2]
  double *adj;  // [rhs_tot][2]
  double *gsum;  // [e4_groups][16] the 14 per-piece sums of every group of constraint points (E4)
  double *lpart; // [14][e4_lcap + 2] contributions of the leftover points of one round
  double *dpart; // [14][T + 2] contributions of T (point, moving obstacle) pairs (kernels with moving obstacles only)
  double *tpc;   // [Ntot] SampleIn::t_piece of the pieces (moving obstacles only)
  double *shead; // [kSurHead] per-obstacle scalars of the moving obstacles (see SurLds)
  double *sdur;  // [sur_np] piece durations of the moving obstacles: Trajectory::locatePieceIdx (poly_traj_utils.hpp:510-528)
                 // walks them one dependent load after the other -- from LDS that is ~100 cycles a step instead of an L2 round trip
  double *pE, *pGsm, *pGdT, *pCost; // [Ntot]
  double *alpha; // [mem] alpha of the two-loop recursion (lbfgs.hpp:725); ys and 1 / ys of the stored pairs live in DevBatch::histR
  double *st;     // [sNUM] scalar solver state
  double *segsum; // [M][8] per-segment sums: 0 jerk energy, 1 penalty cost, 2 d(jerk)/dT, 3 penalty gdT, 4 chain-rule gdT
  double *opM, *opMT; // operators of all segments back to back (only when D.op_in_lds)
  double *cor;        // [4H][NptsPad] half-planes of this trajectory (only when D.cor_in_lds)
  int *ist;     // [iNUM]
  int *gtab;    // [e4_groups] piece | (first j) << 16 of a group of constraint points
  int *ltab;    // [e4_left] piece | j << 16 of a leftover point
  int *wtab;    // [e4_rounds][T/64][3] what a wave does in a round: kind, base, count (e4_plan.h)
  int *rtab;    // [e4_rounds][2] leftovers of a round, first index
  int *pgrp;    // [Ntot][4] first group, groups, first leftover, leftovers of a piece
  int *dinfo;   // [Npts + 1] moving obstacles near a point (bits 16..31) | index of its first pair (kernels with moving obstacles only)
  int *pcinfo;  // [Ntot][8] pt0, K, tab, segment, lp, N, singul, operator offset
  int *rowinfo; // [rhs_tot][4] segment, column, N, first piece of the segment
};

__host__ __device__ inline size_t op_doubles(const DevLayout &L) {
  size_t n = 0;
  for (int i = 0; i < L.M; i++) n += (size_t)6 * L.piece_nums[i] * (L.piece_nums[i] + 5);
  return n;
}
// The LDS copy of the transposed operator pads its rows by kOpTPad doubles: the adjoint reads 8 columns at the
// same row index per instruction, and 6 N doubles is a multiple of the bank period for N = 16 or 32 (8-way conflict).
constexpr int kOpTPad = 4;
__host__ __device__ inline size_t opT_lds_doubles(const DevLayout &L) {
  size_t n = 0;
  for (int i = 0; i < L.M; i++) n += (size_t)(6 * L.piece_nums[i] + kOpTPad) * (L.piece_nums[i] + 5);
  return n;
}
constexpr int kSurHead = 160;    // per-obstacle scalars in front of them: total [16], start [16], pieces per second [16], piece_off [17 ints
                                 // in 16 doubles], end state [16][6]
// The moving obstacles' tables as the evaluation reads them (traj_math.h: the view SV of sur_locate and its callers): every
// table staged in LDS is reached through a pointer that carries the LDS address space, so the accesses are ds_read
// instructions that go out together.  Through generic pointers (DevSurround, with its pointers redirected to LDS) every one
// of them was a flat load followed by a wait for both memory counters: the six loads of a coefficient block and the
// steps of the search cost a full round trip each, 7 800 cycles per (point, obstacle) in the gate.
typedef const double __attribute__((address_space(3))) *lds_cd_t;
typedef const int __attribute__((address_space(3))) *lds_ci_t;
struct SurLds {
  int S;
  lds_ci_t piece_off;                        // [S + 1]
  lds_cd_t durations, theta, total, start;   // [np], [np], [S], [S]
  lds_cd_t rate_;                            // [S] pieces per second (the search's starting guess)
  lds_cd_t bbox_;                            // [np][4] piece boxes (DevSurround::bbox)
  lds_cd_t end_;                             // [S][6] position, velocity, acceleration at the end of the trajectory
  lds_cd_t coef_lds;                         // [np][12] when they fit (kSurCoefLds)
  const double __attribute__((address_space(1))) *coef_glb; // the same blocks in global memory otherwise
  bool theta_on, coef_in_lds, bbox_on;
  __device__ __forceinline__ bool has_theta() const { return theta_on; }
  __device__ __forceinline__ double rate(int u) const { return rate_[u]; }
  __device__ __forceinline__ bool has_bbox() const { return bbox_on; }
  __device__ __forceinline__ void load_box(int k, double bb[4]) const {
    const lds_cd_t q = bbox_ + 4 * k;
    bb[0] = q[0]; bb[1] = q[1]; bb[2] = q[2]; bb[3] = q[3];
  }
  __device__ __forceinline__ bool far_from_piece(int k, const double sigma[2], double r) const {
    if (!bbox_on) return false; // uniform
    const lds_cd_t bb = bbox_ + 4 * k;
    return sigma[0] < bb[0] - r || sigma[0] > bb[1] + r || sigma[1] < bb[2] - r || sigma[1] > bb[3] + r;
  }
  __device__ __forceinline__ void end_state(int u, double pd[2], double vd[2], double ad[2]) const {
    const lds_cd_t q = end_ + 6 * u;
    pd[0] = q[0]; pd[1] = q[1]; vd[0] = q[2]; vd[1] = q[3]; ad[0] = q[4]; ad[1] = q[5];
  }
  __device__ __forceinline__ void load_piece(int k, double c[12]) const {
    if (coef_in_lds) { // uniform
      const lds_cd_t cm = coef_lds + 12 * k;
#pragma unroll
      for (int i = 0; i < 12; i++) c[i] = cm[i];
    } else {
      const double __attribute__((address_space(1))) *cm = coef_glb + 12 * (size_t)k;
#pragma unroll
      for (int i = 0; i < 12; i++) c[i] = cm[i];
    }
  }
};
__host__ __device__ inline size_t smem_doubles(const DevLayout &L, int mem, int groups, int lcap, int T, bool op_lds, bool cor_lds, int sur_np,
                                               bool sur_coef) {
  const bool sur = sur_np > 0;
  size_t n = 0;
  n += 5 * (size_t)L.npad;
  n += (size_t)L.M * 12;
  n += (size_t)L.M * 16;
  n += (size_t)L.M * 2 * (L.Kmax + 1);
  n += (size_t)L.rhs_tot * 2;
  n += 3 * (size_t)L.Ntot * 12;
  n += (size_t)L.rhs_tot * 2;
  {
    // the static stage's group sums + leftover staging and the pair staging of the moving-obstacle stage share their space
    // (the static chain pass has consumed the former before the first pair is written)
    const size_t a = 16 * (size_t)groups + 14 * (size_t)(lcap + 2), b = sur ? 14 * (size_t)(T + 2) : 0;
    n += a > b ? a : b;
  }
  if (sur) n += kSurHead + 6 * (size_t)sur_np + (sur_coef ? 12 * (size_t)sur_np : 0) + (size_t)L.Ntot;
  n += 4 * (size_t)L.Ntot;
  n += (size_t)mem;
  n += sNUM;
  n += (size_t)L.M * 8;
  if (op_lds) n += op_doubles(L) + opT_lds_doubles(L);
  if (cor_lds) n += (size_t)4 * L.H * (((size_t)L.Npts + 63) / 64 * 64);
  return n;
}
__host__ __device__ inline size_t smem_ints(const DevLayout &L, int rounds, int groups, int left, int T, bool sur) {
  return (sur ? (size_t)L.Npts + 2 : 0) + iNUM + (size_t)(groups > 0 ? groups : 1) + (size_t)(left > 0 ? left : 1) + (size_t)rounds * (T / kWave) * 3 +
         (size_t)rounds * 2 + 4 * (size_t)L.Ntot + 8 * (size_t)L.Ntot + 4 * (size_t)L.rhs_tot;
}

E4Sizes e4_sizes(const DevLayout &L, int threads) {
  const E4Plan pl = build_e4_plan(L, threads);
  return E4Sizes{pl.rounds, pl.groups, pl.left, pl.lcap};
}

size_t solver_lds_bytes(const DevLayout &L, const DevParams &P, int threads, bool op_lds, bool cor_lds, int sur_np, bool sur_coef) {
  const bool sur = sur_np > 0;
  const E4Sizes z = e4_sizes(L, threads);
  return smem_doubles(L, P.mem_size, z.groups, z.lcap, threads, op_lds, cor_lds, sur_np, sur_coef) * sizeof(double) + smem_ints(L, z.rounds, z.groups, z.left, threads, sur) * sizeof(int);
}

int solver_threads(const DevLayout &L, int shape) {
  // Every stage is a strided loop, so any multiple of 64 works; the choice trades the latency of one
  // solve against how many trajectories a CU holds (256 VGPRs per lane => 8 waves per CU).  Measured on
  // 528-point problems (scripts/profile_phases.py):
  //   shape 0, <= 1 trajectory per CU : about two constraint points per thread, up to 8 waves
  //   shape 1, <= 2 per CU            : 4 waves, two workgroups resident per CU
  //   shape 2, more                   : 1 wave, eight workgroups resident per CU (the 256-VGPR budget allows 8 waves)
  int T;
  if (shape == 2) {
    T = kWave; // one wave per trajectory: no wave of a workgroup ever waits for the serial part of another
  } else if (shape == 1) {
    T = 256;
    if (L.Npts <= 128) T = 128;
  } else {
    T = ((L.Npts + 1) / 2 + kWave - 1) / kWave * kWave;
    int need = 16 * L.Ntot; // the reduction stages like 16 threads per piece
    if (need > 512) need = 512;
    if (T < need) T = (need + kWave - 1) / kWave * kWave;
  }
  if (T < 128 && shape != 2) T = 128;
  if (T > 512) T = 512;
  return T;
}

__device__ inline void carve(Smem &s, double *base, const DevLayout &L, int mem, int T, int rounds, int groups, int left, int lcap, bool op_lds,
                             bool cor_lds, int sur_np, bool sur_coef) {
  const bool sur = sur_np > 0;
  double *p = base;
  s.x = p; p += L.npad;
  s.xp = p; p += L.npad;
  s.g = p; p += L.npad;
  s.gp = p; p += L.npad;
  s.d = p; p += L.npad;
  s.bnd = p; p += L.M * 12;
  s.seg = p; p += L.M * 16;
  s.spow = p; p += L.M * 2 * (L.Kmax + 1);
  s.rhs = p; p += L.rhs_tot * 2;
  s.b = p; p += L.Ntot * 12;
  s.c = p; p += L.Ntot * 12;
  s.gdC = p; p += L.Ntot * 12;
  s.adj = p; p += L.rhs_tot * 2;
  s.gsum = p;
  s.lpart = p + 16 * groups;
  s.dpart = p; // shares the space of gsum / lpart (see smem_doubles)
  {
    const int a = 16 * groups + 14 * (lcap + 2), b = sur ? 14 * (T + 2) : 0;
    p += a > b ? a : b;
  }
  s.shead = p; // total / start / piece_off of the moving obstacles
  if (sur) p += kSurHead;
  s.sdur = p; // durations, their thresholds (DevSurround::theta), the piece boxes, then the coefficient blocks if they fit
  if (sur) p += 6 * sur_np + (sur_coef ? 12 * sur_np : 0);
  s.tpc = p; // start time of every piece inside its segment (SampleIn::t_piece)
  if (sur) p += L.Ntot;
  s.pE = p; p += L.Ntot;
  s.pGsm = p; p += L.Ntot;
  s.pGdT = p; p += L.Ntot;
  s.pCost = p; p += L.Ntot;
  s.alpha = p; p += mem;
  s.st = p; p += sNUM;
  s.segsum = p; p += L.M * 8;
  s.opM = p;
  s.opMT = p;
  if (op_lds) {
    size_t nop = op_doubles(L);
    s.opMT = p + nop;
    p += nop + opT_lds_doubles(L);
  }
  s.cor = p;
  if (cor_lds) p += (size_t)4 * L.H * ((L.Npts + 63) / 64 * 64);
  s.ist = reinterpret_cast<int *>(p);
  s.gtab = s.ist + iNUM;
  s.ltab = s.gtab + (groups > 0 ? groups : 1);
  s.wtab = s.ltab + (left > 0 ? left : 1);
  s.rtab = s.wtab + rounds * (T / kWave) * 3;
  s.pgrp = s.rtab + rounds * 2;
  s.pcinfo = s.pgrp + 4 * L.Ntot;
  s.rowinfo = s.pcinfo + 8 * L.Ntot;
  s.dinfo = s.rowinfo + 4 * L.rhs_tot;
}

// ------------------------------------------------------------ device helpers
// Cross-lane butterfly over a wave, pairing lanes at distance 1, 2, 4, 8, 16, 32
// in that order: two quad permutes, row_half_mirror and row_mirror (after the
// quad steps every lane of a quad holds the same value, so mirroring pairs the
// same partial sums an xor would), then gfx950's v_permlane16_swap /
// v_permlane32_swap.  3 VALU instructions per level, no LDS traffic.  LV is the
// number of levels: vectors of n <= 16 / 32 / 64 elements use 4 / 5 / 6, the
// upper lanes holding zeros are simply never folded in.  fp addition is
// commutative, so every participating lane ends with the same bits; the CPU
// oracle's device-order mode replays this tree (oracle/dftpav_oracle_dev.cpp).
template <int CTRL>
__device__ inline double mov_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ inline void swap16(double v, double &x, double &y) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  x = __hiloint2double(b[0], a[0]);
  y = __hiloint2double(b[1], a[1]);
}
__device__ inline void swap32(double v, double &x, double &y) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  x = __hiloint2double(b[0], a[0]);
  y = __hiloint2double(b[1], a[1]);
}
template <int LV>
__device__ inline double wave_sum_raw(double v) {
  double x, y;
  v += mov_dpp<0xB1>(v);  // quad_perm [1,0,3,2]
  v += mov_dpp<0x4E>(v);  // quad_perm [2,3,0,1]
  v += mov_dpp<0x141>(v); // row_half_mirror
  v += mov_dpp<0x140>(v); // row_mirror
  if (LV >= 5) {
    swap16(v, x, y);
    v = x + y;
  }
  if (LV >= 6) {
    swap32(v, x, y);
    v = x + y;
  }
  return v;
}
template <int LV>
__device__ inline double wave_max_raw(double v) {
  double x, y;
  v = fmax(v, mov_dpp<0xB1>(v));
  v = fmax(v, mov_dpp<0x4E>(v));
  v = fmax(v, mov_dpp<0x141>(v));
  v = fmax(v, mov_dpp<0x140>(v));
  if (LV >= 5) {
    swap16(v, x, y);
    v = fmax(x, y);
  }
  if (LV >= 6) {
    swap32(v, x, y);
    v = fmax(x, y);
  }
  return v;
}

// The trimmed butterflies leave lanes beyond 2^LV with partial results; every use in the solver
// logic wants one value for the whole wave (the lanes also take branches on it), so lane 0's
// result is broadcast.
__device__ inline double first_lane_f64(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int LV>
__device__ inline double wave_sum(double v) {
  return first_lane_f64(wave_sum_raw<LV>(v));
}
template <int LV>
__device__ inline double wave_max(double v) {
  return first_lane_f64(wave_max_raw<LV>(v));
}

// Sixteen sums over the lanes of a 16-lane row (LV = 4) or of two adjacent rows (LV = 5) with shared butterflies: at
// distance 1 a lane keeps the eight values whose index bit 0 equals its k0 and hands the other eight to its partner,
// at distance 2 it keeps four (bit 1 = k1), at distance 4 (row_half_mirror) two (bit 2 = k2), at distance 8
// (row_mirror) one (bit 3 = k3); LV = 5 adds the two rows (v_permlane16_swap).  The k's are chosen so that mirror
// partners agree on the lower ones: k0 = b0^b2, k1 = b1^b2, k2 = b2^b3, k3 = b3 (b = lane bits).  The lane pairs added
// at every distance are those of the plain butterfly (wave_sum_raw), so each total has its bits; 15 exchanged values
// instead of 64.  Lane l ends with the total of value reduce16_index(l).
__device__ inline int reduce16_index(int lane) {
  const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1;
  return (b0 ^ b2) | ((b1 ^ b2) << 1) | ((b2 ^ b3) << 2) | (b3 << 3);
}
template <int LV>
__device__ __forceinline__ double reduce16(const double (&v)[16], int lane) {
  const bool k0 = ((lane ^ (lane >> 2)) & 1) != 0;
  const bool k1 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0;
  const bool k2 = (((lane >> 2) ^ (lane >> 3)) & 1) != 0;
  const bool k3 = ((lane >> 3) & 1) != 0;
  double w[8], z[4], y[2];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const double keep = k0 ? v[2 * k + 1] : v[2 * k], send = k0 ? v[2 * k] : v[2 * k + 1];
    w[k] = keep + mov_dpp<0xB1>(send); // quad_perm [1,0,3,2]
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double keep = k1 ? w[2 * k + 1] : w[2 * k], send = k1 ? w[2 * k] : w[2 * k + 1];
    z[k] = keep + mov_dpp<0x4E>(send); // quad_perm [2,3,0,1]
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const double keep = k2 ? z[2 * k + 1] : z[2 * k], send = k2 ? z[2 * k] : z[2 * k + 1];
    y[k] = keep + mov_dpp<0x141>(send); // row_half_mirror
  }
  double r;
  {
    const double keep = k3 ? y[1] : y[0], send = k3 ? y[0] : y[1];
    r = keep + mov_dpp<0x140>(send); // row_mirror
  }
  if (LV >= 5) {
    double a, b;
    swap16(r, a, b);
    r = a + b;
  }
  return r;
}

// optional in-kernel phase timer (thread 0, shader clock); D.prof == nullptr turns it off
struct Prof {
  // The twelve accumulators live in the trajectory's row of DevBatch::prof, not in registers: as a member array they
  // were 24 VGPRs held through the whole kernel for a debugging feature.
  long long *acc; // &prof[b][0], or anything when off
  long long last;
  bool on;
  __device__ inline void start(bool enable, long long *row, bool resume) {
    on = enable && threadIdx.x == 0;
    acc = row;
    if (on && !resume)
      for (int i = 0; i < 12; i++) acc[i] = 0;
    last = on ? clock64() : 0;
  }
  __device__ inline void tick(int i) {
    if (on) {
      long long t = clock64();
      acc[i] += t - last;
      last = t;
    }
  }
};
enum { kPE1 = 0, kPE2, kPE3S, kPE4R, kPE5, kPE6, kPLS, kPHIST, kPLOOP, kPX, kPMISC };

// half-planes of one constraint point held in registers (up to 4 planes)
struct RegPlanes {
  const double *c; // 16 values: [plane][n0,n1,q0,q1]
  __device__ inline void operator()(int k, double &n0, double &n1, double &q0, double &q1) const {
    n0 = c[4 * k];
    n1 = c[4 * k + 1];
    q0 = c[4 * k + 2];
    q1 = c[4 * k + 3];
  }
};
// strided plane loader over the component-major corridor of one trajectory
// half-plane k of one constraint point; the pointer carries its address space so that the loads are
// global_load / ds_read rather than flat_load (a generic pointer cannot tell the compiler which)
template <class P> struct PitchedPlanes {
  P base;       // &corridor[b][0][pt]
  size_t pitch; // NptsPad
  __device__ inline void operator()(int k, double &n0, double &n1, double &q0, double &q1) const {
    n0 = base[(size_t)(4 * k + 0) * pitch];
    n1 = base[(size_t)(4 * k + 1) * pitch];
    q0 = base[(size_t)(4 * k + 2) * pitch];
    q1 = base[(size_t)(4 * k + 3) * pitch];
  }
};
typedef const double __attribute__((address_space(1))) *cor_g_t;
typedef const double __attribute__((address_space(3))) *cor_l_t;
typedef PitchedPlanes<cor_g_t> GlobalPlanes;
typedef PitchedPlanes<cor_l_t> LdsPlanes;

// ----------------------------------------------------- cost + gradient
// PolyTrajOptimizer::costFunctionCallback (traj_optimizer.cpp:206-350) for the
// decision vector x (LDS) of trajectory b; writes g (LDS) and f (sm.st[sF]).
// D is read through scalar loads (uniform); per-lane lookups go through the LDS tables.
// Row of the dense MINCO operator times the right-hand side (E2) / column times the scaled gradient (E5).
// All operator values of a chunk are requested back to back before the first one is used: at two waves per
// SIMD nothing else hides an L2 (or LDS) round trip, so a loop that waits per element is a chain of them.
// Out-of-range slots re-read the last valid element and are masked at use; the sums keep their order.
typedef const double __attribute__((address_space(1))) *opg_t; // operator in global memory: global_load, vmcnt only
constexpr int kOpChunk = 24;
template <class P> __device__ __forceinline__ double op_row_dot(P Mrow, const double *rh, int ncol) {
  double acc = 0.0;
  for (int c0 = 0; c0 < ncol; c0 += kOpChunk) {
    double mv[kOpChunk], rv[kOpChunk];
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const int col = c0 + j < ncol ? c0 + j : ncol - 1;
      mv[j] = Mrow[col];
    }
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const int col = c0 + j < ncol ? c0 + j : ncol - 1;
      rv[j] = rh[2 * col];
    }
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const double t = fma_(mv[j], rv[j], acc);
      acc = c0 + j < ncol ? t : acc;
    }
  }
  return acc;
}
// the same row against both dimensions of the right-hand side (rh[2 col], rh[2 col + 1]): the operator values
// are fetched once for the two sums
template <class P> __device__ __forceinline__ void op_row_dot2(P Mrow, const double *rh, int ncol, double &ax, double &ay) {
  double acc0 = 0.0, acc1 = 0.0;
  for (int c0 = 0; c0 < ncol; c0 += kOpChunk) {
    double mv[kOpChunk], r0[kOpChunk], r1[kOpChunk];
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const int col = c0 + j < ncol ? c0 + j : ncol - 1;
      mv[j] = Mrow[col];
    }
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const int col = c0 + j < ncol ? c0 + j : ncol - 1;
      r0[j] = rh[2 * col];
      r1[j] = rh[2 * col + 1];
    }
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const double t0 = fma_(mv[j], r0[j], acc0), t1 = fma_(mv[j], r1[j], acc1);
      acc0 = c0 + j < ncol ? t0 : acc0;
      acc1 = c0 + j < ncol ? t1 : acc1;
    }
  }
  ax = acc0;
  ay = acc1;
}
// lane q of a quad sums rows q, q + 4, q + 8, ... of one operator column against gc[2 r] * tInv[r mod 6]
template <class P> __device__ __forceinline__ double op_col_dot(P MT, const double *gc, const double *tInv, int q, int nrow) {
  double acc = 0.0;
  for (int r0 = q; r0 < nrow; r0 += 4 * kOpChunk) {
    double mv[kOpChunk], gv[kOpChunk], tv[kOpChunk];
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const int r = r0 + 4 * j < nrow ? r0 + 4 * j : q;
      mv[j] = MT[r];
    }
    // the power of 1 / dt that goes with row r is tInv[r mod 6]; rows advance by 4, so a lane only ever meets three of them
    // (r0 mod 6, +4, +2): three reads in front instead of one per row (the per-row reads came out as a chain of LDS
    // round trips behind the operator loads)
    const int k0 = r0 % 6;
    const double t3[3] = {tInv[k0], tInv[k0 + 4 >= 6 ? k0 - 2 : k0 + 4], tInv[k0 + 2 >= 6 ? k0 - 4 : k0 + 2]};
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const int r = r0 + 4 * j < nrow ? r0 + 4 * j : q;
      gv[j] = gc[2 * r];
      tv[j] = t3[j % 3];
    }
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const double t = fma_(mv[j], gv[j] * tv[j], acc);
      acc = r0 + 4 * j < nrow ? t : acc;
    }
  }
  return acc;
}

// the same column against both dimensions of the gradient (gc[2 r], gc[2 r + 1])
template <class P>
__device__ __forceinline__ void op_col_dot2(P MT, const double *gc, const double *tInv, int q, int nrow, double &ax, double &ay) {
  double acc0 = 0.0, acc1 = 0.0;
  for (int r0 = q; r0 < nrow; r0 += 4 * kOpChunk) {
    double mv[kOpChunk], g0[kOpChunk], g1[kOpChunk], tv[kOpChunk];
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const int r = r0 + 4 * j < nrow ? r0 + 4 * j : q;
      mv[j] = MT[r];
    }
    const int k0 = r0 % 6; // see op_col_dot
    const double t3[3] = {tInv[k0], tInv[k0 + 4 >= 6 ? k0 - 2 : k0 + 4], tInv[k0 + 2 >= 6 ? k0 - 4 : k0 + 2]};
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const int r = r0 + 4 * j < nrow ? r0 + 4 * j : q;
      g0[j] = gc[2 * r];
      g1[j] = gc[2 * r + 1];
      tv[j] = t3[j % 3];
    }
#pragma unroll
    for (int j = 0; j < kOpChunk; j++) {
      const double t0 = fma_(mv[j], g0[j] * tv[j], acc0), t1 = fma_(mv[j], g1[j] * tv[j], acc1);
      acc0 = r0 + 4 * j < nrow ? t0 : acc0;
      acc1 = r0 + 4 * j < nrow ? t1 : acc1;
    }
  }
  ax = acc0;
  ay = acc1;
}

// the constraint point a lane evaluates in a round of the E4 plan (e4_plan.h): piece | j << 16, or -1
__device__ inline int e4_lane_point(const Smem &sm, int kind, int base, int count, int lane) {
  if (kind > 0) { // a wave of groups: `count` groups of `kind` consecutive points each
    const int g = lane / kind;
    if (g >= count) return -1;
    return sm.gtab[base + g] + ((lane & (kind - 1)) << 16);
  }
  if (kind == 0) return lane < count ? sm.ltab[base + lane] : -1; // a wave of leftover points
  return -1;
}

// Segment durations of the decision vector x (VirtualT2RealT, traj_optimizer.cpp:371-379), the piece duration and its
// powers (poly_traj_utils.hpp:961-966) into sm.seg, one lane per segment.  Wave 0 runs it as soon as an x to be evaluated
// is in place (before the first evaluation of a pass; at the end of lbfgs_advance), so that the evaluation starts with
// them instead of every right-hand-side lane dividing for itself.
__device__ inline void prep_durations(const DevBatch &D, const Smem &sm, const double *x, int lane) {
  const DevLayout &L = D.L;
  if (lane < L.M) {
    const int sg = lane;
    int N = 0;
    for (int s_ = 0; s_ < L.M; s_++) N = (s_ == sg) ? L.piece_nums[s_] : N;
    const double Tr = virtual_to_real(x[L.x_tau0 + sg], D.P.mini_T);
    const double dt = Tr / N;
    double *s = sm.seg + sg * 16;
    s[0] = Tr;
    s[1] = dt;
    duration_powers(dt, s + 2);
  }
}

// One (constraint point, obstacle) pair that passed the gate: dynamicObsGradCostP's body (traj_math.h: dynamic_pair, 9.2 k
// instructions) and the pair's 14 contributions to its piece, stored at dst[k * stride].  As a function of its own (tried in
// round 4, DFTPAV_PAIR_OUT_OF_LINE=1): the ~110 doubles live through its log-sum-exp stage do not fit 256 registers either way --
// the callee spills 229 vector registers with 816 B of scratch where the inlined form spills 188 with 432 B -- and the static
// samples, the gate and the chain passes around the call got 25-50 % slower: 479 against 369 ms per 1024 (configs[4]).  Inlined
// is the default.
#ifndef DFTPAV_PAIR_OUT_OF_LINE
#define DFTPAV_PAIR_OUT_OF_LINE 0
#endif
__device__ __attribute__((noinline)) void pair_eval(const DevParams &P, const SurLds &surL, const SampleIn &in, int u, double *dst, int stride) {
  double o[8], v[14];
  dynamic_pair_math(P, surL, in, u, o);
  point_contributions(in.s1, o, v);
#pragma unroll
  for (int k = 0; k < 14; k++) dst[k * stride] = v[k];
}

template <bool SUR>
__device__ __forceinline__ void block_eval(const DevBatch &D, const double *cor_b, const Smem &sm, const double *x, double *g,
                                           Prof &pr) {
  // cor_b: half-planes of this trajectory, &corridor[b][0][0] of the batch it belongs to
  const DevLayout &L = D.L; // uniform accesses only (scalar loads)
  const DevParams &P = D.P;
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
  const int M = L.M, Ntot = L.Ntot, Npts = L.Npts, rhs_tot = L.rhs_tot, Kmax1 = L.Kmax + 1;
  const double *iniS = sm.bnd, *finS = sm.bnd + 6 * M; // boundary states of this trajectory, staged with x

  // ---- E1: MINCO right-hand sides and the sample offsets.  The segment durations T, dt and the powers of
  // dt are already in sm.seg: prep_durations formed them on wave 0 as soon as this x was written.
  {
    const int n_rhs = 2 * rhs_tot;
    const int n_rhs_pad = (n_rhs + 63) & ~63; // the per-segment workers start on a wave of their own
    for (int w = tid; w < n_rhs_pad + 2 * M; w += T) {
      if (w >= n_rhs && w < n_rhs_pad) continue;
      if (w < n_rhs) {
        int row = w >> 1, d = w & 1;
        const int *ri = sm.rowinfo + 4 * row;
        int sg = ri[0], col = ri[1], N = ri[2];
        const double dt = sm.seg[sg * 16 + 1];
        // Every entry is one stored value (a waypoint or junction position from x, or a boundary state) times
        // 1, dt or dt^2 (poly_traj_utils.hpp:968-977), so it is formed without branching: the lanes of a wave hold
        // all kinds of rows and a branch per kind serialises them (each with its own LDS round trip).  x * 1.0 is x.
        // Only the junction velocities of a gear shift (traj_optimizer.cpp:273-282) keep a branch: they need cos / sin.
        const bool head = col < 3, tail = col >= N + 2;
        const int k = head ? col : (tail ? col - (N + 2) : 0);
        const bool junction = (head && sg > 0) || (tail && sg < M - 1);
        const bool from_x = (!head && !tail) || (k == 0 && junction);
        int ix = head ? L.x_gear0 + 2 * (sg - 1) + d : (tail ? L.x_gear0 + 2 * sg + d : ri[3] + 2 * (col - 3) + d);
        ix = from_x ? ix : 0; // ri[3] = offset of the segment's waypoints inside x
        const int ib = (tail ? 6 * M : 0) + sg * 6 + 2 * k + d; // sm.bnd: iniS of every segment, then finS
        const double xv = x[ix], bv = sm.bnd[ib];
        const double scale = k == 0 ? 1.0 : (k == 1 ? dt : dt * dt);
        double v = (from_x ? xv : bv) * scale;
        if (k == 1 && junction) {
          const double th = x[L.x_ang0 + (head ? sg - 1 : sg)];
          const double hv = head ? (d == 0 ? -P.non_sinv * p_cos(th) : -P.non_sinv * p_sin(th))
                                 : (d == 0 ? P.non_sinv * p_cos(th) : P.non_sinv * p_sin(th));
          v = hv * dt;
        }
        sm.rhs[w] = v;
      } else {
        int q = w - n_rhs_pad; // 2 workers per segment: the offset tables for K and Kd
        int sg = q >> 1, which = q & 1;
        const double dt = sm.seg[sg * 16 + 1];
        int K = which ? L.Kd : L.K;
        double step = dt / K;
        // offsets s1 = 0, += step, ... (traj_optimizer.cpp:513): one dependent chain of additions by definition; four
        // links at a time are formed in registers and their stores trail behind them
        typedef double __attribute__((address_space(3))) *ldsw_t;
        ldsw_t tab = (ldsw_t)(sm.spow + (size_t)(sg * 2 + which) * Kmax1);
        double s1 = 0.0;
        int j = 0;
        for (; j + 4 <= K + 1; j += 4) {
          const double a0 = s1, a1 = a0 + step, a2 = a1 + step, a3 = a2 + step;
          s1 = a3 + step;
          tab[j] = a0;
          tab[j + 1] = a1;
          tab[j + 2] = a2;
          tab[j + 3] = a3;
        }
        for (; j <= K; j++) {
          tab[j] = s1;
          s1 += step;
        }
      }
    }
    __syncthreads(); // the right-hand sides for E2; the offsets (their powers are formed per point, E4) for E4
  }
  pr.tick(kPE1);

  // ---- E2: b = A^{-1} rhs (dense operator), c = b * t^-k   (MinJerkOpt::generate, poly_traj_utils.hpp:979-984)
  // 16 threads per piece, 12 of them active: (row k, dimension d); narrow workgroups: 8 per piece, 6 active, a
  // thread forms both dimensions of its row from one fetch of the operator values (one pass for 16 pieces)
  if (T <= 128) {
    for (int w = tid; w < 8 * Ntot; w += T) {
      const int p = w >> 3, k = w & 7;
      if (k < 6) {
        const int *pc = sm.pcinfo + 8 * p;
        const int sg = pc[3], lp = pc[4], N = pc[5];
        int r0 = 0; // first RHS row of the segment
        for (int s = 0; s < M; s++) r0 = (s == sg) ? L.seg_rhs0[s] : r0;
        const double *rh = sm.rhs + 2 * r0;
        const size_t roff = (size_t)(6 * lp + k) * (N + 5);
        double ax, ay;
        if (D.op_in_lds) {
          op_row_dot2(sm.opM + pc[7] + roff, rh, N + 5, ax, ay);
        } else {
          const double *Mop = D.opM[0];
          for (int s = 1; s < M; s++) Mop = (s == sg) ? D.opM[s] : Mop;
          op_row_dot2((opg_t)Mop + roff, rh, N + 5, ax, ay);
        }
        const double tk = sm.seg[sg * 16 + 8 + k];
        sm.b[12 * p + 2 * k] = ax;
        sm.b[12 * p + 2 * k + 1] = ay;
        sm.c[12 * p + 2 * k] = ax * tk;
        sm.c[12 * p + 2 * k + 1] = ay * tk;
      }
    }
  } else
  for (int w = tid; w < 16 * Ntot; w += T) {
    int p = w >> 4, q = w & 15;
    if (q < 12) {
      int k = q >> 1, d = q & 1;
      const int *pc = sm.pcinfo + 8 * p;
      int sg = pc[3], lp = pc[4], N = pc[5];
      int r0 = 0; // first RHS row of the segment
      for (int s = 0; s < M; s++) r0 = (s == sg) ? L.seg_rhs0[s] : r0;
      const double *rh = sm.rhs + 2 * r0 + d;
      const size_t roff = (size_t)(6 * lp + k) * (N + 5);
      double acc;
      if (D.op_in_lds) {
        acc = op_row_dot(sm.opM + pc[7] + roff, rh, N + 5);
      } else {
        const double *Mop = D.opM[0];
        for (int s = 1; s < M; s++) Mop = (s == sg) ? D.opM[s] : Mop;
        acc = op_row_dot((opg_t)Mop + roff, rh, N + 5);
      }
      sm.b[12 * p + q] = acc;
      sm.c[12 * p + q] = acc * sm.seg[sg * 16 + 8 + k];
    }
  }
  __syncthreads();
  pr.tick(kPE2);

  // ---- E3: jerk energy and its partials per piece (poly_traj_utils.hpp:998-1035)
  for (int p = tid; p < Ntot; p += T) {
    int sg = sm.pcinfo[8 * p + 3];
    double en, gsm;
    piece_smoothness(sm.c + 12 * p, sm.seg + sg * 16 + 2, en, gsm, sm.gdC + 12 * p);
    sm.pE[p] = en;
    sm.pGsm[p] = gsm;
    sm.pGdT[p] = 0.0;
    sm.pCost[p] = 0.0;
  }

  // ---- E4: penalty integral over the constraint points (traj_optimizer.cpp:486-779)
  // Lanes take points by the plan of e4_plan.h.  A group of 32 (16) consecutive points of one piece sits in 32 (16)
  // adjacent lanes: each lane turns its point's subtotals into the 14 contributions to the piece (12 entries of gdC,
  // gdT, cost) and the group sums them with one cross-lane tree (reduce16) -- no LDS round trip, no barrier.  Points
  // that fill no group ("leftovers") are packed densely, their contributions staged in LDS.  After the last group round
  // (and after every round with leftovers) one chain pass adds, per piece and output in this order: the value E3 left,
  // the group sums, the leftovers in point order.
  {
    const int nwv = T >> 6, wv = tid >> 6;
    const int lstride = D.e4_lcap + 2;
    bool groups_added = false;
    for (int r = 0; r < D.e4_rounds; r++) {
      const int kind = __builtin_amdgcn_readfirstlane(sm.wtab[(r * nwv + wv) * 3]);
      const int wbase = __builtin_amdgcn_readfirstlane(sm.wtab[(r * nwv + wv) * 3 + 1]);
      const int wcount = __builtin_amdgcn_readfirstlane(sm.wtab[(r * nwv + wv) * 3 + 2]);
      const int info = e4_lane_point(sm, kind, wbase, wcount, lane); // piece | j << 16 of this lane's point, -1: none
      const int nleft = sm.rtab[2 * r], lbase = sm.rtab[2 * r + 1];
      if (kind >= 0) { // wave-uniform: this wave has points in this round
        double o[8];
        double s1 = 0.0;
        if (info >= 0) {
          const int p = info & 0xffff;
          const int *pc = sm.pcinfo + 8 * p;
          SampleIn in;
          in.j = info >> 16;
          in.K = pc[1];
          const int sg = pc[3];
          in.lp = pc[4];
          in.N = pc[5];
          in.singul = pc[6];
          in.dt = sm.seg[sg * 16 + 1];
          in.s1 = s1 = sm.spow[(size_t)pc[2] * Kmax1 + in.j];
          in.cc = sm.c + 12 * p;
          in.epis = D.epis;
          in.H = L.H;
          in.trajid = sg;
          in.trajtime = sg == 0 ? 0.0 : sm.seg[(sg > 0 ? sg - 1 : 0) * 16]; // trajtimes[trajid] = T_{i-1}, traj_optimizer.cpp:230-234
          in.t_now = D.t_now;
          {
            const int pt = pc[0] + in.j;
            const double *cb = cor_b + pt;
            if (D.cor_in_lds) {
              LdsPlanes pl{(cor_l_t)(sm.cor + pt), (size_t)((Npts + 63) / 64 * 64)};
              if (L.H <= 4) sample_point_math<false, 4>(P, D.sur, in, pl, o);
              else sample_point_math<false, 0>(P, D.sur, in, pl, o);
            } else if (L.H <= 4) {
              double ccl[12]; // the piece's coefficients first (LDS), then the half-planes (global memory)
              load_piece_coeffs(in.cc, ccl);
              in.cc = ccl;
              double cor[16]; // all half-plane loads issued up front (unconditionally: rows past 4 H re-read row 0 and
                              // are never used), consumed after the state evaluation
              const cor_g_t cg = (cor_g_t)cb;
#pragma unroll
              for (int k = 0; k < 16; k++) cor[k] = cg[(size_t)(k < 4 * L.H ? k : 0) * D.NptsPad];
              RegPlanes pl{cor};
              sample_point_math<false, 4>(P, D.sur, in, pl, o);
            } else {
              GlobalPlanes pl{(cor_g_t)cb, (size_t)D.NptsPad};
              sample_point_math<false, 0>(P, D.sur, in, pl, o);
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) o[k] = 0.0;
        }
        double v[16];
        point_contributions(s1, o, v);
        v[14] = 0.0;
        v[15] = 0.0;
        if (kind > 0) { // a wave of groups
          double tot;
          if (kind == 32) tot = reduce16<5>(v, lane);
          else tot = reduce16<4>(v, lane);
          const int idx = reduce16_index(lane);
          if (info >= 0 && (lane & (kind - 1)) < 16 && idx < 14) {
            sm.gsum[(wbase + lane / kind) * 16 + idx] = tot;
          }
        } else if (info >= 0) { // a wave of leftovers
          const int li = wbase - lbase + lane;
#pragma unroll
          for (int k = 0; k < 14; k++) sm.lpart[k * lstride + li] = v[k];
        }
      }
      pr.tick(kPE3S);
      const bool chain = nleft > 0 || r == D.e4_rounds - 1; // uniform
      if (chain) {
        __syncthreads();
        for (int w = tid; w < 16 * Ntot; w += T) {
          const int p = w >> 4, q = w & 15;
          if (q >= 14) continue;
          const int *pg = sm.pgrp + 4 * p;
          double acc = q < 12 ? sm.gdC[12 * p + q] : (q == 12 ? sm.pGdT[p] : sm.pCost[p]);
          if (!groups_added)
            for (int gi = 0; gi < pg[1]; gi++) acc += sm.gsum[(pg[0] + gi) * 16 + q];
          const int l0 = pg[2] > lbase ? pg[2] : lbase;
          const int l1 = pg[2] + pg[3] < lbase + nleft ? pg[2] + pg[3] : lbase + nleft;
          for (int l = l0; l < l1; l++) acc += sm.lpart[q * lstride + (l - lbase)];
          if (q < 12) sm.gdC[12 * p + q] = acc;
          else if (q == 12) sm.pGdT[p] = acc;
          else sm.pCost[p] = acc;
        }
        groups_added = true;
        __syncthreads();
        pr.tick(kPE4R);
      }
    }
  }


  // ---- E4, moving obstacles (traj_optimizer.cpp:636-638): one (constraint point, obstacle) PAIR per lane, and only the
  // pairs that pass the distance gate of traj_optimizer.cpp:1393.  First every point marks the obstacles near it (a bit
  // mask); an exclusive prefix sum over the points numbers the marked pairs in (point, obstacle) order; then the pairs
  // are evaluated T at a time, densely packed -- a lane finds its pair by bisection in the prefix sums -- and after
  // every T pairs a chain pass adds their contributions to the pieces in that order.
  if (SUR) {
    const int dstride = T + 2;
    SurLds surL;
    surL.S = D.sur.S;
    surL.total = (lds_cd_t)sm.shead;
    surL.start = (lds_cd_t)(sm.shead + 16);
    surL.rate_ = (lds_cd_t)(sm.shead + 32);
    surL.end_ = (lds_cd_t)(sm.shead + 64);
    surL.piece_off = (lds_ci_t)reinterpret_cast<int *>(sm.shead + 48);
    surL.durations = (lds_cd_t)sm.sdur;
    surL.theta = (lds_cd_t)(sm.sdur + D.sur_np);
    surL.theta_on = D.sur.theta != nullptr;
    surL.bbox_ = (lds_cd_t)(sm.sdur + 2 * D.sur_np);
    surL.bbox_on = D.sur.bbox != nullptr;
    surL.coef_lds = (lds_cd_t)(sm.sdur + 6 * D.sur_np);
    surL.coef_glb = (const double __attribute__((address_space(1))) *)D.sur.coeffs;
    surL.coef_in_lds = D.sur_coef_lds != 0;
    for (int p = tid; p < Ntot; p += T) { // the pieces' start times: the reference's running sum, once per piece instead of per point
      const int *pc = sm.pcinfo + 8 * p;
      sm.tpc[p] = piece_start_time(sm.seg[pc[3] * 16 + 1], pc[4]);
    }
    __syncthreads();
    for (int r = 0; r < D.e4_rounds; r++) {
      const int wv_ = tid >> 6, nwv_ = T >> 6;
      const int info = e4_lane_point(sm, sm.wtab[(r * nwv_ + wv_) * 3], sm.wtab[(r * nwv_ + wv_) * 3 + 1], sm.wtab[(r * nwv_ + wv_) * 3 + 2], lane);
      if (info >= 0) {
        const int p = info & 0xffff;
        const int *pc = sm.pcinfo + 8 * p;
        SampleIn in;
        in.j = info >> 16;
        in.K = pc[1];
        const int sg = pc[3];
        in.lp = pc[4];
        in.N = pc[5];
        in.singul = pc[6];
        in.dt = sm.seg[sg * 16 + 1];
        in.s1 = sm.spow[(size_t)pc[2] * Kmax1 + in.j];
        in.cc = sm.c + 12 * p;
        in.epis = D.epis;
        in.H = L.H;
        in.trajid = sg;
        in.trajtime = sg == 0 ? 0.0 : sm.seg[(sg > 0 ? sg - 1 : 0) * 16];
        in.t_piece = sm.tpc[p];
        in.t_now = D.t_now;
        sm.dinfo[pc[0] + in.j] = (int)(dynamic_gate_mask(P, surL, in) << 16);
      }
    }
    __syncthreads();
    if (tid < 64) { // exclusive prefix sum of the pair counts: a contiguous run of points per lane, then across the lanes
      const int seg = (Npts + 63) >> 6, start = lane * seg;
      int sum = 0;
      for (int i = start; i < start + seg && i < Npts; i++) sum += __builtin_popcount((unsigned)sm.dinfo[i] >> 16);
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      int run = incl - sum;
      for (int i = start; i < start + seg && i < Npts; i++) {
        const unsigned m = (unsigned)sm.dinfo[i];
        sm.dinfo[i] = (int)((m & 0xffff0000u) | (unsigned)run);
        run += __builtin_popcount(m >> 16);
      }
      if (lane == 63) sm.dinfo[Npts] = incl;
    }
    __syncthreads();
    pr.tick(kPX);
    const int n_pairs = sm.dinfo[Npts];
    for (int c0 = 0; c0 < n_pairs; c0 += T) {
      const int i = c0 + tid;
      const int last = (n_pairs - c0 < T ? n_pairs - c0 : T) - 1; // the thread of the batch's last pair
      if (i < n_pairs) {
        int lo = 0, hi = Npts; // the last point whose first pair index is <= i: it holds pair i
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if ((sm.dinfo[mid] & 0xffff) <= i) lo = mid;
          else hi = mid;
        }
        const int pt = lo;
        const unsigned w = (unsigned)sm.dinfo[pt];
        int kth = i - (int)(w & 0xffffu);
        unsigned mask = w >> 16;
        for (; kth > 0; kth--) mask &= mask - 1; // drop the kth lowest set bits
        const int u = __builtin_ctz(mask);
        int plo = 0, phi = Ntot; // the piece of the point
        while (phi - plo > 1) {
          const int mid = (plo + phi) >> 1;
          if (sm.pcinfo[8 * mid] <= pt) plo = mid;
          else phi = mid;
        }
        const int p = plo;
        // pairs are numbered in point order: the batch touches the pieces from its first pair's to its last pair's, and the
        // chain pass below visits only those (the two padding doubles at the end of dpart's first row carry the bounds)
        if (tid == 0) sm.dpart[T] = (double)p;
        if (tid == last) sm.dpart[T + 1] = (double)p;
        const int *pc = sm.pcinfo + 8 * p;
        SampleIn in;
        in.j = pt - pc[0];
        in.K = pc[1];
        const int sg = pc[3];
        in.lp = pc[4];
        in.N = pc[5];
        in.singul = pc[6];
        in.dt = sm.seg[sg * 16 + 1];
        in.s1 = sm.spow[(size_t)pc[2] * Kmax1 + in.j];
        in.cc = sm.c + 12 * p;
        in.epis = D.epis;
        in.H = L.H;
        in.trajid = sg;
        in.trajtime = sg == 0 ? 0.0 : sm.seg[(sg > 0 ? sg - 1 : 0) * 16];
        in.t_piece = sm.tpc[p];
        in.t_now = D.t_now;
#if DFTPAV_PAIR_OUT_OF_LINE
        {
          // copies, so that only they escape to the call (the kernel's own surL / in stay in registers)
          const SurLds sl = surL;
          const SampleIn in2 = in;
          pair_eval(P, sl, in2, u, sm.dpart + tid, dstride);
        }
#else
        double o[8], v[14];
        dynamic_pair_math(P, surL, in, u, o);
        point_contributions(in.s1, o, v);
#pragma unroll
        for (int k = 0; k < 14; k++) sm.dpart[k * dstride + tid] = v[k];
#endif
      }
      pr.tick(kPMISC);
      __syncthreads();
      const int p_first = (int)sm.dpart[T], p_last = (int)sm.dpart[T + 1];
      for (int w = 16 * p_first + tid; w < 16 * (p_last + 1); w += T) {
        const int p = w >> 4, q = w & 15;
        if (q >= 14) continue;
        const int *pc = sm.pcinfo + 8 * p;
        const int pt0 = pc[0], pt1 = pt0 + pc[1] + 1;
        int i0 = sm.dinfo[pt0] & 0xffff, i1 = pt1 < Npts ? (sm.dinfo[pt1] & 0xffff) : n_pairs;
        i0 = i0 > c0 ? i0 : c0;
        i1 = i1 < c0 + T ? i1 : c0 + T;
        if (i1 <= i0) continue;
        double acc = q < 12 ? sm.gdC[12 * p + q] : (q == 12 ? sm.pGdT[p] : sm.pCost[p]);
        {
          // the chain itself is sequential by definition; its LDS reads are not: eight at a time in front of their additions
          const double *src = sm.dpart + q * dstride - c0;
          int i2 = i0;
          for (; i2 + 8 <= i1; i2 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[i2 + u];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
          }
          for (; i2 < i1; i2++) acc += src[i2];
        }
        if (q < 12) sm.gdC[12 * p + q] = acc;
        else if (q == 12) sm.pGdT[p] = acc;
        else sm.pCost[p] = acc;
      }
      __syncthreads();
      pr.tick(kPE4R);
    }
  }

  // ---- E5: adjoint through A^{-T} (MinJerkOpt::calGrads_PT, poly_traj_utils.hpp:1037-1064)
  // waves [0, span/64): 4 lanes per output, lane q sums rows q, q+4, q+8, ... in order, then (p0+p1)+(p2+p3)
  // next wave: chain-rule duration terms per piece, summed per segment by a 64-lane butterfly
  // next wave: the other per-piece quantities (jerk energy, penalty cost, gdT parts), same butterfly
  {
    // narrow workgroups: a quad forms both dimensions of its row from one fetch of the operator column, which
    // halves the quads (one pass of 128 threads for 21 rows) and leaves one light task per wave for the second pass
    const bool both = T <= 128;
    const int n_out = both ? rhs_tot : 2 * rhs_tot;
    const int span = ((4 * n_out + 63) >> 6) << 6; // whole waves: the quad butterfly needs all 4 lanes alive
    for (int w = tid; w < span + 128; w += T) {
      if (w < span) {
        int o = w >> 2, q = w & 3;
        double acc = 0.0, acc2 = 0.0;
        if (o < n_out) {
          int row = both ? o : o >> 1, d = both ? 0 : o & 1;
          const int *ri = sm.rowinfo + 4 * row;
          int sg = ri[0], col = ri[1], N = ri[2];
          int p0 = 0, ooff = 0; // ooff: offset of the segment's operator inside the padded LDS copy
          for (int s = 0, a = 0; s < M; s++) {
            p0 = (s == sg) ? L.seg_piece0[s] : p0;
            ooff = (s == sg) ? a : ooff;
            a += (6 * L.piece_nums[s] + kOpTPad) * (L.piece_nums[s] + 5);
          }
          const double *gc = sm.gdC + 12 * p0 + d;
          const double *tInv = sm.seg + sg * 16 + 8;
          const size_t coff = (size_t)col * 6 * N;
          if (both) {
            if (D.op_in_lds) {
              op_col_dot2(sm.opMT + ooff + (size_t)col * (6 * N + kOpTPad), gc, tInv, q, 6 * N, acc, acc2);
            } else {
              const double *MTb = D.opMT[0];
              for (int s = 1; s < M; s++) MTb = (s == sg) ? D.opMT[s] : MTb;
              op_col_dot2((opg_t)MTb + coff, gc, tInv, q, 6 * N, acc, acc2);
            }
          } else if (D.op_in_lds) {
            acc = op_col_dot(sm.opMT + ooff + (size_t)col * (6 * N + kOpTPad), gc, tInv, q, 6 * N);
          } else {
            const double *MTb = D.opMT[0];
            for (int s = 1; s < M; s++) MTb = (s == sg) ? D.opMT[s] : MTb;
            acc = op_col_dot((opg_t)MTb + coff, gc, tInv, q, 6 * N);
          }
        }
        acc += mov_dpp<0xB1>(acc);
        acc += mov_dpp<0x4E>(acc);
        if (both) {
          acc2 += mov_dpp<0xB1>(acc2);
          acc2 += mov_dpp<0x4E>(acc2);
          if (o < n_out && q == 0) {
            sm.adj[2 * o] = acc;
            sm.adj[2 * o + 1] = acc2;
          }
        } else if (o < n_out && q == 0) {
          sm.adj[o] = acc;
        }
      } else if (w < span + 64) {
        int ln = w - span;
        for (int sg = 0; sg < M; sg++) {
          const double *tInv = sm.seg + sg * 16 + 8;
          double gdtInv[6] = {0.0, -1.0 * tInv[2], -2.0 * tInv[3], -3.0 * tInv[4], -4.0 * tInv[5],
                              -5.0 * tInv[5] * tInv[1]}; // poly_traj_utils.hpp:1054-1060
          double v = 0.0;
          for (int p = L.seg_piece0[sg] + ln; p < L.seg_piece0[sg + 1]; p += 64) {
            const double *gc = sm.gdC + 12 * p;
            const double *bb = sm.b + 12 * p;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) acc += gdtInv[k] * (gc[2 * k] * bb[2 * k] + gc[2 * k + 1] * bb[2 * k + 1]);
            v += acc;
          }
          v = wave_sum<6>(v);
          if (ln == 0) sm.segsum[sg * 8 + 4] = v;
        }
      } else {
        int ln = w - span - 64;
        for (int sg = 0; sg < M; sg++) {
          double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
          for (int p = L.seg_piece0[sg] + ln; p < L.seg_piece0[sg + 1]; p += 64) {
            v0 += sm.pE[p];
            v1 += sm.pCost[p];
            v2 += sm.pGsm[p];
            v3 += sm.pGdT[p];
          }
          v0 = wave_sum<6>(v0);
          v1 = wave_sum<6>(v1);
          v2 = wave_sum<6>(v2);
          v3 = wave_sum<6>(v3);
          if (ln == 0) {
            sm.segsum[sg * 8 + 0] = v0;
            sm.segsum[sg * 8 + 1] = v1;
            sm.segsum[sg * 8 + 2] = v2;
            sm.segsum[sg * 8 + 3] = v3;
          }
        }
      }
    }
  }
  __syncthreads();
  pr.tick(kPE5);

  // ---- E6: assemble g and f (traj_optimizer.cpp:299-344)
  for (int e = tid; e < L.n + 1; e += T) {
    if (e == L.n) {
      double sm_cost = 0.0, pen = 0.0, tc = 0.0;
      for (int sg = 0; sg < M; sg++) {
        sm_cost += sm.segsum[sg * 8 + 0];
        pen += sm.segsum[sg * 8 + 1];
        tc += sm.seg[sg * 16] * P.wei_time;
      }
      sm.st[sF] = sm_cost + tc + pen;
    } else if (e < L.x_tau0) { // waypoints: gdP = rows 6i+5 of the adjoint
      int sg = 0, x0 = 0, r0 = 0;
      for (int s = 0; s < M; s++) {
        bool in = e >= L.seg_x0[s];
        sg = in ? s : sg;
        x0 = in ? L.seg_x0[s] : x0;
        r0 = in ? L.seg_rhs0[s] : r0;
      }
      int q = e - x0;
      int wp = q >> 1, d = q & 1;
      g[e] = sm.adj[2 * (r0 + 3 + wp) + d];
    } else if (e < L.x_gear0) { // tau: VirtualTGradCost, traj_optimizer.cpp:405-419
      int sg = e - L.x_tau0;
      int N = 0, r0 = 0;
      for (int s = 0; s < M; s++) {
        N = (s == sg) ? L.piece_nums[s] : N;
        r0 = (s == sg) ? L.seg_rhs0[s] : r0;
      }
      const double *seg = sm.seg + sg * 16;
      double dt = seg[1];
      double gdT = 0.0;
      gdT += sm.segsum[sg * 8 + 2];
      gdT += sm.segsum[sg * 8 + 3];
      // boundary-scaling terms, poly_traj_utils.hpp:1050-1053 (with the junction-overridden head/tail v)
      const double *ad = sm.adj + 2 * r0;
      double hv[2], tv[2];
      if (sg > 0) {
        double th = x[L.x_ang0 + sg - 1];
        hv[0] = -P.non_sinv * p_cos(th);
        hv[1] = -P.non_sinv * p_sin(th);
      } else {
        hv[0] = iniS[sg * 6 + 2];
        hv[1] = iniS[sg * 6 + 3];
      }
      if (sg < M - 1) {
        double th = x[L.x_ang0 + sg];
        tv[0] = P.non_sinv * p_cos(th);
        tv[1] = P.non_sinv * p_sin(th);
      } else {
        tv[0] = finS[sg * 6 + 2];
        tv[1] = finS[sg * 6 + 3];
      }
      int rt = N + 2; // first tail row
      gdT += hv[0] * ad[2 * 1] + hv[1] * ad[2 * 1 + 1];
      gdT += (iniS[sg * 6 + 4] * ad[2 * 2] + iniS[sg * 6 + 5] * ad[2 * 2 + 1]) * 2.0 * dt;
      gdT += tv[0] * ad[2 * (rt + 1)] + tv[1] * ad[2 * (rt + 1) + 1];
      gdT += (finS[sg * 6 + 4] * ad[2 * (rt + 2)] + finS[sg * 6 + 5] * ad[2 * (rt + 2) + 1]) * 2.0 * dt;
      gdT += sm.segsum[sg * 8 + 4];
      g[e] = (gdT / N + P.wei_time) * virtual_to_real_grad(x[e]);
    } else { // gear position (traj_optimizer.cpp:307-320) and gear angle
      bool is_ang = e >= L.x_ang0;
      int i = is_ang ? e - L.x_ang0 : (e - L.x_gear0) >> 1;
      int d = (e - L.x_gear0) & 1;
      int Ni = 0, r0i = 0, r0n = 0;
      for (int s = 0; s < M; s++) {
        Ni = (s == i) ? L.piece_nums[s] : Ni;
        r0i = (s == i) ? L.seg_rhs0[s] : r0i;
        r0n = (s == i + 1) ? L.seg_rhs0[s] : r0n;
      }
      double v = 0.0;
      if (P.gear_opt) {
        if (!is_ang) {
          v += sm.adj[2 * (r0i + Ni + 2) + d] * 1.0; // gdTail.col(0) of segment i
          v += sm.adj[2 * (r0n + 0) + d] * 1.0;      // gdHead.col(0) of segment i+1
        } else {
          double th = x[e];
          double dti = sm.seg[i * 16 + 1], dtn = sm.seg[(i + 1) * 16 + 1];
          double ft0 = sm.adj[2 * (r0i + Ni + 3) + 0] * dti, ft1 = sm.adj[2 * (r0i + Ni + 3) + 1] * dti;
          double hd0 = sm.adj[2 * (r0n + 1) + 0] * dtn, hd1 = sm.adj[2 * (r0n + 1) + 1] * dtn;
          v += ft0 * (-P.non_sinv * p_sin(th)) + ft1 * (P.non_sinv * p_cos(th));
          v += hd0 * (P.non_sinv * p_sin(th)) + hd1 * (-P.non_sinv * p_cos(th));
        }
      }
      g[e] = v;
    }
  }
  __syncthreads();
  pr.tick(kPE6);
}

// ------------------------------------------------------------- two-loop
// Two-loop recursion (lbfgs.hpp:716-739) for n <= 64: one element of the
// direction per lane.  History columns, 1/ys and alpha are streamed through a
// PF-deep register ring so that the loads of step i+PF are in flight while step
// i reduces.  Ring refills are unconditional loads from always-valid addresses
// (lane clamped into the padded row, slot into the ring) followed by a select:
// no divergent region, so the loads stay in flight instead of being waited on
// at a branch join; slot indices advance with wrap-around (no integer division).
typedef const double __attribute__((address_space(1))) *gptr_t; // global (not flat) loads: vmcnt only
typedef double __attribute__((address_space(1))) *gwptr_t;

// ------------------------------------------- two-loop recursion, n <= 64
// One element of the direction per lane.  The reference's step (lbfgs.hpp:722-726)
//     alpha_t = s_t.d / ys_t ;  d -= alpha_t y_t
// is a chain of ~25 dependent instructions (lane multiply, cross-lane reduction, divide, axpy).  It
// is run in blocks of kLoopBlock stored pairs: the block's dot products are all taken against d as
// it stands at the start of the block (independent reductions, they overlap in the pipeline), and
// what the block's earlier steps would have removed from d is restored from the stored products of
// neighbouring pairs,   s_t.(d - sum_u alpha_u y_u) = s_t.d - sum_u alpha_u (s_t.y_u),
// so only one multiply-add of a step waits for the previous step: the division by ys_t is a product
// with the stored 1 / ys_t, applied to s_t.d once and folded into the stored products.  histU / histV
// hold (s_j.y_k) / ys of the pair that owns the row, for the kLoopBlock-1 pairs on either side of a pair;
// they are written once, when the newer pair of the two is stored (lbfgs_advance).  oracle/dftpav_oracle_dev.cpp replays the same
// sequence.
constexpr int kLoopBlock = 8;
__device__ inline double readlane_f64(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// Cross-lane instructions are the expensive ones here (a DPP move or v_readlane costs ~10 cycles of
// issue against ~4 for an fp64 FMA, scripts/ubench.hip), so the block's eight reductions share their
// butterflies: at distance 1 a lane keeps the four sums whose index bit 0 matches its own and hands
// the other four to its partner, at distance 2 it keeps two, at distance 4 one -- 7 exchanged values
// instead of 24 -- and the remaining distances fold that one value as before.  The pairs added at
// every distance are exactly those of wave_sum_raw (fp addition is commutative), so every sum has the
// bits of lane_dot in the oracle.  Lane l ends up holding the total of step step_of_lane(l); the
// mirrors used at distance 4 and 8 complement the lower lane bits, which the index map absorbs.
__device__ inline int step_of_lane(int l) {
  const int b0 = l & 1, b1 = (l >> 1) & 1, b2 = (l >> 2) & 1, b3 = (l >> 3) & 1;
  return ((b2 ^ b3) << 2) | ((b1 ^ b2) << 1) | (b0 ^ b2);
}
__host__ __device__ constexpr int lane_of_step(int u) { // a lane (< 8) holding step u
  return (((u >> 2) & 1) << 2) | ((((u >> 1) ^ (u >> 2)) & 1) << 1) | ((u ^ (u >> 2)) & 1);
}
template <int LV>
__device__ __forceinline__ double wave_sum8_transposed(const double (&v)[kLoopBlock], int lane) {
  asm volatile("" : "+v"(lane)); // keeps the three masks local (see first_loop_block)
  const bool k0 = ((lane ^ (lane >> 2)) & 1) != 0;        // bit 0 of my step
  const bool k1 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0; // bit 1
  const bool k2 = (((lane >> 2) ^ (lane >> 3)) & 1) != 0; // bit 2
  double w[4], z[2];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double keep = k0 ? v[2 * k + 1] : v[2 * k], send = k0 ? v[2 * k] : v[2 * k + 1];
    w[k] = keep + mov_dpp<0xB1>(send); // quad_perm [1,0,3,2]
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const double keep = k1 ? w[2 * k + 1] : w[2 * k], send = k1 ? w[2 * k] : w[2 * k + 1];
    z[k] = keep + mov_dpp<0x4E>(send); // quad_perm [2,3,0,1]
  }
  double r;
  {
    const double keep = k2 ? z[1] : z[0], send = k2 ? z[0] : z[1];
    r = keep + mov_dpp<0x141>(send); // row_half_mirror
  }
  r += mov_dpp<0x140>(r); // row_mirror
  double x, y;
  if (LV >= 5) {
    swap16(r, x, y);
    r = x + y;
  }
  if (LV >= 6) {
    swap32(r, x, y);
    r = x + y;
  }
  return r;
}

// one block of kLoopBlock consecutive steps: the history columns in registers; the per-step scalars
// live in the lanes that own the step (step_of_lane): ys, 1/ys and the products with the block's
// earlier steps, coef[u] = s_step . y_u (first loop) or y_step . s_u (second loop)
struct HistBlock {
  double s[kLoopBlock], y[kLoopBlock];
  double coef[kLoopBlock - 1];
  double al; // second loop: alpha of the step this lane owns (broadcast per step with a v_readlane pair, as the first loop
             // does with the alpha it has just formed; eight uniform copies per block would cost 14 more registers per
             // block, which pushes the function into callee-saved registers: 8 KB of scratch traffic per call)
};
// the three LDS arrays the recursion touches, as pointers that carry their address space (the recursion is an
// out-of-line function: a generic pointer would turn every access into a flat one)
typedef double __attribute__((address_space(3))) *lds_rw_t;
struct LoopLds {
  gptr_t ysr;     // [mem][2] (ys, 1 / ys) of the stored pairs, global memory (written once per pair, lbfgs_advance)
  lds_rw_t alpha; // [mem]
};
// loads the block whose first step sits in slot `jl`, walking downwards (DIR = -1) or upwards (+1)
// with wrap-around; unconditional loads from always-valid addresses, nothing consumes them here
template <int DIR, bool LOOP2>
__device__ __forceinline__ void load_block(HistBlock &R, const LoopLds &sm, gptr_t hS, gptr_t hY, gptr_t hB, int npad, int m, int ln,
                                           int lane, int &jl) {
  typedef const char __attribute__((address_space(1))) *gbytes_t;
  const int st = step_of_lane(lane);
  int js = jl + DIR * st; // |offset| < kLoopBlock <= m
  js = js < 0 ? js + m : (js >= m ? js - m : js);
  // Entry 7 of a histU / histV row is never written and stays 0.0 (the buffers are cleared when the batch is created):
  // the steps a lane does not wait for (u >= its own) get a zero product, so the correction below is an unconditional
  // multiply-add that leaves the lane's sum as it is -- no compare / select pair in the dependent chain of a step.
  // Addresses are a uniform 64-bit base plus a 32-bit lane offset (the saddr form of global_load: no 64-bit vector
  // arithmetic per load).
  const unsigned row = (unsigned)js * 64u;
#pragma unroll
  for (int u = 0; u < kLoopBlock - 1; u++) {
    const unsigned idx = st > u ? (unsigned)(st - u - 1) : 7u;
    R.coef[u] = *(gptr_t)((gbytes_t)hB + (row + idx * 8u));
  }
  if (LOOP2) R.al = sm.alpha[js];
  const unsigned lnoff = (unsigned)ln * 16u;
#pragma unroll
  for (int q = 0; q < kLoopBlock; q++) {
    {
      // element e of pair j is the double2 (s, y) at [(j * npad + e)]: one 16-byte load instead of two 8-byte ones
      typedef double __attribute__((ext_vector_type(2))) d2_t;
      // (a trajectory's history is mem * npad * 16 bytes, far below 4 GB: the whole offset fits 32 bits)
      const unsigned off = (unsigned)jl * ((unsigned)npad * 16u) + lnoff;
      const d2_t sy = *(const d2_t __attribute__((address_space(1))) *)((gbytes_t)hS + off);
      R.s[q] = sy.x;
      R.y[q] = sy.y;
    }
    if (DIR < 0) jl = jl == 0 ? m - 1 : jl - 1;
    else jl = jl == m - 1 ? 0 : jl + 1;
  }
}
// Makes every loaded register of the block a use: the wait for the block's loads lands here, before the
// next block's loads are issued, so it can only be a wait for this block (the compiler's counter for
// outstanding loads is conservative around the loop back edge and would otherwise wait for both).
__device__ __forceinline__ void pin_block(HistBlock &R) {
#pragma unroll
  for (int q = 0; q < kLoopBlock; q++) {
    asm volatile("" : "+v"(R.s[q]));
    asm volatile("" : "+v"(R.y[q]));
  }
#pragma unroll
  for (int u = 0; u < kLoopBlock - 1; u++) asm volatile("" : "+v"(R.coef[u]));
}
template <int LV, bool FULL>
__device__ __forceinline__ void first_loop_block(const HistBlock &R, const LoopLds &sm, int i0, int nb, int m, bool act, int lane,
                                                 int &j, double &dreg) {
  int jown = j - step_of_lane(lane); // slot of the step this lane owns (the block starts at slot j and walks downwards)
  jown = jown < 0 ? jown + m : jown;
  typedef const char __attribute__((address_space(1))) *gbytes_t;
  const double ri_ = *(gptr_t)((gbytes_t)sm.ysr + ((unsigned)jown * 16u + 8u)); // 1 / lm_ys of my step; first used after the reduction below
  // Lanes at and past n hold zeros: their direction element enters as 0.0 and the rows of the history are zero
  // there (pad elements are never written), so no lane is masked in the products or in the updates below.
  double v[kLoopBlock];
#pragma unroll
  for (int q = 0; q < kLoopBlock; q++) v[q] = R.s[q] * dreg; // lm_s.col(j).dot(d), steps past nb unused
  // ... / lm_ys(j) as a product with the stored reciprocal, taken once for the whole sum: the stored neighbour products
  // coef[] carry the same factor (lbfgs_advance), so the chain of a step is one v_readlane pair and one multiply-add
  double acc = wave_sum8_transposed<LV>(v, lane) * ri_;
  int st = step_of_lane(lane);
  asm volatile("" : "+v"(st));
#pragma unroll
  for (int u = 0; u < kLoopBlock; u++) {
    if (FULL || i0 + u < nb) { // uniform
      const double au = readlane_f64(acc, lane_of_step(u));     // alpha of step u for everybody (its owner's sum is complete)
      if (u < kLoopBlock - 1) acc = __builtin_fma(-au, R.coef[u], acc); // coef[u] == 0.0 for the lanes whose step is <= u
      dreg = __builtin_fma(-au, R.y[u], dreg);
    }
  }
  // lanes 0..7 own one step each; their sum stopped changing at their own step: it is their alpha
  const double mine = acc;
  int js = j - st;
  js = js < 0 ? js + m : js;
  if (lane < kLoopBlock && (FULL || i0 + st < nb)) sm.alpha[js] = mine;
  const int done = FULL ? kLoopBlock : nb - i0;
  j -= done;
  j = j < 0 ? j + m : j;
}
template <int LV, bool FULL>
__device__ __forceinline__ void second_loop_block(const HistBlock &R, const LoopLds &sm, int i0, int nb, int m, bool act, int lane, int &j2,
                                                  double &dreg) {
  int jown = j2 + step_of_lane(lane); // the block starts at slot j2 and walks upwards
  jown = jown >= m ? jown - m : jown;
  typedef const char __attribute__((address_space(1))) *gbytes_t;
  const double ri_ = *(gptr_t)((gbytes_t)sm.ysr + ((unsigned)jown * 16u + 8u));
  j2 += kLoopBlock;
  j2 = j2 >= m ? j2 - m : j2;
  double v[kLoopBlock];
#pragma unroll
  for (int q = 0; q < kLoopBlock; q++) v[q] = R.y[q] * dreg; // lm_y.col(j).dot(d), steps past nb unused
  double acc = wave_sum8_transposed<LV>(v, lane) * ri_;
  int st = step_of_lane(lane);
  asm volatile("" : "+v"(st));
#pragma unroll
  for (int u = 0; u < kLoopBlock; u++) {
    if (FULL || i0 + u < nb) { // uniform
      const double bu = readlane_f64(acc, lane_of_step(u));  // beta of step u
      const double au = readlane_f64(R.al, lane_of_step(u)); // alpha of step u (first loop), from the lane that owns the step
      const double df = au - bu;
      if (u < kLoopBlock - 1) acc = __builtin_fma(df, R.coef[u], acc); // coef[u] == 0.0 for the lanes whose step is <= u
      dreg = __builtin_fma(df, R.s[u], dreg);
    }
  }
}
template <int LV>
__device__ __forceinline__ void first_loop_step(const HistBlock &R, const LoopLds &sm, int i0, int nb, int m, bool act, int lane,
                                                int &j, double &dreg) {
  if (i0 + kLoopBlock <= nb) first_loop_block<LV, true>(R, sm, i0, nb, m, act, lane, j, dreg);
  else if (i0 < nb) first_loop_block<LV, false>(R, sm, i0, nb, m, act, lane, j, dreg);
}
template <int LV>
__device__ __forceinline__ void second_loop_step(const HistBlock &R, const LoopLds &sm, int i0, int nb, int m, bool act, int lane, int &j2,
                                                 double &dreg) {
  if (i0 + kLoopBlock <= nb) second_loop_block<LV, true>(R, sm, i0, nb, m, act, lane, j2, dreg);
  else if (i0 < nb) second_loop_block<LV, false>(R, sm, i0, nb, m, act, lane, j2, dreg);
}

// History columns (global memory: they live in L2 / Infinity Cache) are double-buffered in registers:
// the loads of block k+1 are in flight while block k is reduced, and no wait is
// ever placed right after a load.  Only lanes below 2^LV take part (the trimmed butterfly leaves the
// others with partial sums that are never used: their direction element is masked).
//
// Out of line on purpose: inlined into the solve kernel the recursion shares one register allocation with
// the per-point evaluation, and an unrelated edit there moved spills into this loop (290 -> 386 cycles per
// history step).  As a function it is allocated on its own.  Arguments arrive in vector registers; the
// uniform ones are moved to scalar registers on entry.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> __device__ __forceinline__ T uni_ptr(T p) {
  unsigned long long a = (unsigned long long)p;
  unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
  unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (T)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ lds_rw_t uni_lds(lds_rw_t p) {
#if defined(__HIP_DEVICE_COMPILE__) // LDS pointers are 32 bits wide on the device
  return (lds_rw_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)p);
#else
  return p;
#endif
}
template <int LV>
__device__ __attribute__((noinline)) double two_loop_lane(gptr_t h_ysr, lds_rw_t l_alpha, gptr_t hS, gptr_t hY,
                                                          gptr_t hU, gptr_t hV, int npad, int n, int m, int nb, int ne,
                                                          double ys_new, double yy_new, int lane, double dreg) {
  hS = uni_ptr(hS); hY = uni_ptr(hY); hU = uni_ptr(hU); hV = uni_ptr(hV);
  const LoopLds sm{uni_ptr(h_ysr), uni_lds(l_alpha)};
  npad = uni(npad); n = uni(n); m = uni(m); ne = uni(ne);
  const bool act = lane < n;
  const int ln = lane < (1 << LV) ? lane : 0; // rows are zero from element n on: the lanes of the butterfly load their own
  nb = __builtin_amdgcn_readfirstlane(nb);
  constexpr int PB = kLoopBlock;
  HistBlock A, B;
  // ---- first loop: newest -> oldest (slots ne-1, ne-2, ...)
  int j = ne == 0 ? m - 1 : ne - 1; // slot of the next step
  int jl = j;
  load_block<-1, false>(A, sm, hS, hY, hU, npad, m, ln, lane, jl);
  for (int i0 = 0; i0 < nb; i0 += 2 * PB) {
    pin_block(A);
    load_block<-1, false>(B, sm, hS, hY, hU, npad, m, ln, lane, jl);
    first_loop_step<LV>(A, sm, i0, nb, m, act, lane, j, dreg);
    pin_block(B);
    load_block<-1, false>(A, sm, hS, hY, hU, npad, m, ln, lane, jl);
    first_loop_step<LV>(B, sm, i0 + PB, nb, m, act, lane, j, dreg);
  }
  dreg *= ys_new / yy_new;
  // ---- second loop: oldest -> newest, starting one past the slot the first loop ended on
  jl = j == m - 1 ? 0 : j + 1;
  int j2 = jl; // slot of the first step of the block being reduced
  load_block<+1, true>(A, sm, hS, hY, hV, npad, m, ln, lane, jl);
  for (int i0 = 0; i0 < nb; i0 += 2 * PB) {
    pin_block(A);
    load_block<+1, true>(B, sm, hS, hY, hV, npad, m, ln, lane, jl);
    second_loop_step<LV>(A, sm, i0, nb, m, act, lane, j2, dreg);
    pin_block(B);
    load_block<+1, true>(A, sm, hS, hY, hV, npad, m, ln, lane, jl);
    second_loop_step<LV>(B, sm, i0 + PB, nb, m, act, lane, j2, dreg);
  }
  return dreg;
}

// dot products of wave 0 over LDS vectors: lane l accumulates elements l, l+64, ... from 0.0, then the butterfly
template <int LV>
__device__ inline double wave_dot(const double *a, const double *b, int n, int lane) {
  double acc = 0.0;
  for (int e = lane; e < n; e += 64) acc += a[e] * b[e];
  return wave_sum<LV>(acc);
}

// ---------------------------------------------------- L-BFGS state machine
// Start of an outer iteration (lbfgs.hpp:559-574 and the head of
// line_search_lewisoverton, lbfgs.hpp:290-315): xp = x, gp = g, dginit, first trial point.
// Returns false when the line search cannot start (negative return code in iRET).
template <int LV>
__device__ __forceinline__ bool begin_iteration(const DevParams &P, const Smem &sm, int n, int lane) {
  double acc = 0.0;
  for (int e = lane; e < n; e += 64) {
    sm.xp[e] = sm.x[e];
    double gv = sm.g[e];
    sm.gp[e] = gv;
    acc += gv * sm.d[e];
  }
  double dginit = wave_sum<LV>(acc);
  double step = sm.st[sSTEP];
  if (!(step > 0.0)) {
    if (lane == 0) sm.ist[iRET] = -1006; // LBFGSERR_INVALIDPARAMETERS
    return false;
  }
  if (0.0 < dginit) {
    if (lane == 0) sm.ist[iRET] = -1005; // LBFGSERR_INCREASEGRADIENT
    return false;
  }
  if (lane == 0) {
    sm.st[sFINIT] = sm.st[sFX];
    sm.st[sDGINIT] = dginit;
    sm.st[sDGTEST] = P.f_dec_coeff * dginit;
    sm.st[sDSTEST] = P.s_curv_coeff * dginit;
    sm.st[sMU] = 0.0;
    sm.st[sNU] = P.max_step;
    sm.st[sSTP] = step;
    sm.ist[iCOUNT] = 0;
    sm.ist[iBRACKT] = 0;
    sm.ist[iTOUCHED] = 0;
  }
  for (int e = lane; e < n; e += 64) sm.x[e] = sm.xp[e] + step * sm.d[e];
  return true;
}

// Everything lbfgs_optimize does between two evaluations (lbfgs.hpp:524-745 with the line
// search of lbfgs.hpp:312-389 unrolled into it).  Runs on wave 0, every lane computing the same
// scalars from LDS; sets iACTION to kActEval (a new trial x is in sm.x) or kActDone.
template <int LV>
__device__ __forceinline__ void lbfgs_advance(const DevBatch &D, const Smem &sm, double *hS_b, double *hU_b, double *hV_b, double *hR_b, int lane,
                                              Prof &pr) {
  // hS_b / hU_b / hV_b: this trajectory's history blocks inside the batch it belongs to
  const DevParams &P = D.P;
  const int n = D.L.n, m = P.mem_size, npad = D.L.npad;
  const double f = sm.st[sF];
  int action = kActEval;
  if (sm.ist[iPHASE] == 0) {
    // ---- after the first evaluation: lbfgs.hpp:524-551
    double gmax = 0.0, xmax = 0.0, dd = 0.0;
    for (int e = lane; e < n; e += 64) {
      double gv = sm.g[e];
      sm.d[e] = -gv;
      gmax = fmax(gmax, fabs(gv));
      xmax = fmax(xmax, fabs(sm.x[e]));
      dd += gv * gv;
    }
    gmax = wave_max<LV>(gmax);
    xmax = wave_max<LV>(xmax);
    dd = wave_sum<LV>(dd);
    if (lane == 0) {
      sm.st[sFX] = f;
      sm.st[sPF0] = f;
      sm.ist[iEVALS] = 1;
      sm.ist[iEND] = 0;
      sm.ist[iBOUND] = 0;
      sm.ist[iHISTLO] = 0;
      sm.ist[iHISTHI] = 0;
      sm.ist[iPHASE] = 1;
    }
    if (gmax / fmax(1.0, xmax) < P.g_epsilon) {
      if (lane == 0) {
        sm.ist[iRET] = 0; // LBFGS_CONVERGENCE
        sm.ist[iK] = 0;
      }
      action = kActDone;
    } else {
      if (lane == 0) {
        sm.st[sSTEP] = 1.0 / sqrt(dd);
        sm.ist[iK] = 1;
      }
      if (!begin_iteration<LV>(P, sm, n, lane)) action = kActDone; // x == xp, g == gp here: nothing to revert
    }
    if (lane == 0) sm.ist[iACTION] = action;
    pr.tick(kPMISC);
    return;
  }

  // ---- after a line-search trial: lbfgs.hpp:317-389
  const double fx = f;
  const double finit = sm.st[sFINIT];
  double stp = sm.st[sSTP];
  int count = sm.ist[iCOUNT] + 1;
  int ls = 0;
  bool decided = false; // true: the search ended (ls holds count or an error), false: try another step
  if (lane == 0) {
    sm.st[sFX] = fx;
    sm.ist[iEVALS] = sm.ist[iEVALS] + 1;
    sm.ist[iCOUNT] = count;
  }
  if (isinf(fx) || isnan(fx)) {
    ls = -1012; // LBFGSERR_INVALID_FUNCVAL
    decided = true;
  } else if (P.past > 0 && fabs(finit - fx) / (fabs(finit) + 1.0) < P.delta / P.past) { // lbfgs.hpp:326-329
    ls = count;
    decided = true;
  } else {
    double mu = sm.st[sMU], nu = sm.st[sNU];
    bool brackt = sm.ist[iBRACKT] != 0;
    if (fx > finit + stp * sm.st[sDGTEST]) {
      nu = stp;
      brackt = true;
    } else {
      double gs = wave_dot<LV>(sm.g, sm.d, n, lane);
      if (gs < sm.st[sDSTEST]) {
        mu = stp;
      } else {
        ls = count;
        decided = true;
      }
    }
    if (!decided) {
      if (P.max_linesearch <= count) {
        ls = -1009; // LBFGSERR_MAXIMUMLINESEARCH
        decided = true;
      } else if (brackt && (nu - mu) < P.machine_prec * nu) {
        ls = -1007; // LBFGSERR_WIDTHTOOSMALL
        decided = true;
      } else {
        if (brackt) stp = 0.5 * (mu + nu);
        else stp *= 2.0;
        if (stp < P.min_step) {
          ls = -1011; // LBFGSERR_MINIMUMSTEP
          decided = true;
        } else if (stp > P.max_step) {
          if (sm.ist[iTOUCHED]) {
            ls = -1010; // LBFGSERR_MAXIMUMSTEP
            decided = true;
          } else {
            if (lane == 0) sm.ist[iTOUCHED] = 1;
            stp = P.max_step;
          }
        }
      }
    }
    if (lane == 0) {
      sm.st[sMU] = mu;
      sm.st[sNU] = nu;
      sm.ist[iBRACKT] = brackt ? 1 : 0;
      sm.st[sSTP] = stp;
    }
    if (!decided) { // next trial point
      for (int e = lane; e < n; e += 64) sm.x[e] = sm.xp[e] + stp * sm.d[e];
      if (lane == 0) sm.ist[iACTION] = kActEval;
      pr.tick(kPLS);
      return;
    }
  }
  // the search ended; `step` takes the last trial step (lbfgs.hpp:574 passes it by reference)
  if (lane == 0) sm.st[sSTEP] = stp;
  if (ls < 0) { // revert x and g, keep fx (lbfgs.hpp:604-611)
    for (int e = lane; e < n; e += 64) {
      sm.x[e] = sm.xp[e];
      sm.g[e] = sm.gp[e];
    }
    if (lane == 0) {
      sm.ist[iRET] = ls;
      sm.ist[iACTION] = kActDone;
    }
    return;
  }

  // ---- convergence / stopping tests (lbfgs.hpp:628-666)
  int k = sm.ist[iK];
  {
    double gmax = 0.0, xmax = 0.0;
    for (int e = lane; e < n; e += 64) {
      gmax = fmax(gmax, fabs(sm.g[e]));
      xmax = fmax(xmax, fabs(sm.x[e]));
    }
    gmax = wave_max<LV>(gmax);
    xmax = wave_max<LV>(xmax);
    const int kGoOn = 12345;
    int ret = kGoOn;
    if (gmax / fmax(1.0, xmax) < P.g_epsilon) {
      ret = 0; // LBFGS_CONVERGENCE
    } else {
      if (0 < P.past) {
        int slot = k % P.past;
        if (P.past <= k) {
          double rate = fabs(sm.st[sPF0 + slot] - fx) / fmax(1.0, fabs(fx));
          if (rate < P.delta) ret = 1; // LBFGS_STOP
        }
        if (ret == kGoOn && lane == 0) sm.st[sPF0 + slot] = fx;
      }
      if (ret == kGoOn && P.max_iterations != 0 && P.max_iterations <= k) ret = -1008; // LBFGSERR_MAXIMUMITERATION
    }
    if (ret != kGoOn) {
      if (lane == 0) {
        sm.ist[iRET] = ret;
        sm.ist[iACTION] = kActDone;
      }
      return;
    }
  }
  ++k;
  if (lane == 0) sm.ist[iK] = k;
  pr.tick(kPLS);

  // ---- history update + two-loop recursion (lbfgs.hpp:676-740)
  // s and y of a pair are interleaved element by element (histY == histS + 1, element stride 2)
  double *hS = hS_b;
  double *hY = hS + 1;
  const int end = sm.ist[iEND];
  int bound = sm.ist[iBOUND];
  {
    double *sc = hS + (size_t)end * npad * 2;
    double ys = 0.0, yy = 0.0, ss = 0.0, gpgp = 0.0, ylane = 0.0;
    for (int e = lane; e < n; e += 64) {
      double sv = sm.x[e] - sm.xp[e];
      double yv = sm.g[e] - sm.gp[e];
      ylane = yv; // n <= 64: this lane's only element
      {
        typedef double __attribute__((ext_vector_type(2))) d2_t;
        d2_t sy;
        sy.x = sv;
        sy.y = yv;
        *reinterpret_cast<d2_t *>(sc + 2 * e) = sy; // the pair as one 16-byte store (yc == sc + 1)
      }
      ys += yv * sv;
      yy += yv * yv;
      ss += sv * sv;
      double gpv = sm.gp[e];
      gpgp += gpv * gpv;
      sm.d[e] = -sm.g[e];
    }
    ys = wave_sum<LV>(ys);
    yy = wave_sum<LV>(yy);
    ss = wave_sum<LV>(ss);
    gpgp = wave_sum<LV>(gpgp);
    if (lane == 0) {
      typedef double __attribute__((ext_vector_type(2))) d2r_t;
      d2r_t yr;
      yr.x = ys;
      yr.y = 1.0 / ys;
      *reinterpret_cast<d2r_t *>(hR_b + 2 * (size_t)end) = yr; // read back by this wave in the recursion below (fenced there)
    }
    double cau = ss * sqrt(gpgp) * P.cautious_factor;
    pr.tick(kPHIST);
    if (ys > cau) {
      ++bound;
      bound = m < bound ? m : bound;
      int ne = end + 1 == m ? 0 : end + 1;
      if (n <= 64 && m >= kLoopBlock) {
        // products of the new y with the s of the kLoopBlock-1 pairs before it (histU / histV)
        double *hU = hU_b, *hV = hV_b;
        {
          const bool act = lane < n;
          const int ln = act ? lane : 0;
          const int nold = bound - 1 < kLoopBlock - 1 ? bound - 1 : kLoopBlock - 1;
          double sv[kLoopBlock - 1], rio[kLoopBlock - 1];
          int o = end;
#pragma unroll
          for (int dd = 0; dd < kLoopBlock - 1; dd++) {
            o = o == 0 ? m - 1 : o - 1;
            sv[dd] = ((gptr_t)hS)[((size_t)o * npad + ln) * 2];
            rio[dd] = ((gptr_t)hR_b)[2 * (size_t)o + 1]; // 1 / ys of the older pair (slots never stored: unused below)
          }
          const double rin = 1.0 / ys; // what lane 0 stored above
          o = end;
#pragma unroll
          for (int dd = 0; dd < kLoopBlock - 1; dd++) {
            o = o == 0 ? m - 1 : o - 1;
            if (dd < nold) {
              double v = wave_sum_raw<LV>((act ? sv[dd] : 0.0) * ylane);
              if (lane == 0) { // each product is stored times 1 / ys of the pair whose step reads it (first / second loop)
                hU[(size_t)o * 8 + dd] = v * rio[dd];
                hV[(size_t)end * 8 + dd] = v * rin;
              }
            }
          }
          __threadfence_block(); // lane 0 wrote them, every lane of this wave reads them below
        }
        // the newest column was written by these same lanes: program order makes it visible to them
        double dreg = lane < n ? sm.d[lane] : 0.0;
        dreg = two_loop_lane<LV>((gptr_t)hR_b, (lds_rw_t)sm.alpha, (gptr_t)hS, (gptr_t)hY, (gptr_t)hU,
                                 (gptr_t)hV, npad, n, m, bound, ne, ys, yy, lane, dreg);
        if (lane < n) sm.d[lane] = dreg;
      } else {
        int j = ne;
        for (int i = 0; i < bound; ++i) {
          j = j == 0 ? m - 1 : j - 1;
          const double *sj = hS + (size_t)j * npad * 2, *yj = hY + (size_t)j * npad * 2;
          double acc = 0.0;
          for (int e = lane; e < n; e += 64) acc += sj[2 * e] * sm.d[e];
          acc = wave_sum<LV>(acc);
          double a = acc / ((gptr_t)hR_b)[2 * (size_t)j];
          if (lane == 0) sm.alpha[j] = a;
          double na = -a;
          for (int e = lane; e < n; e += 64) sm.d[e] += na * yj[2 * e];
        }
        double sc0 = ys / yy;
        for (int e = lane; e < n; e += 64) sm.d[e] *= sc0;
        for (int i = 0; i < bound; ++i) {
          const double *sj = hS + (size_t)j * npad * 2, *yj = hY + (size_t)j * npad * 2;
          double acc = 0.0;
          for (int e = lane; e < n; e += 64) acc += yj[2 * e] * sm.d[e];
          acc = wave_sum<LV>(acc);
          double beta = acc / ((gptr_t)hR_b)[2 * (size_t)j];
          double cf = sm.alpha[j] - beta;
          for (int e = lane; e < n; e += 64) sm.d[e] += cf * sj[2 * e];
          j = j == m - 1 ? 0 : j + 1;
        }
      }
      if (lane == 0) {
        sm.ist[iEND] = ne;
        sm.ist[iBOUND] = bound;
        long long hs = ((long long)sm.ist[iHISTHI] << 32) | (unsigned int)sm.ist[iHISTLO];
        hs += bound;
        sm.ist[iHISTLO] = (int)(hs & 0xffffffffLL);
        sm.ist[iHISTHI] = (int)(hs >> 32);
      }
    }
  }
  if (lane == 0) sm.st[sSTEP] = 1.0; // lbfgs.hpp:743
  pr.tick(kPLOOP);
  // wave 0 wrote sSTEP / d through LDS; begin_iteration reads them back in program order
  bool ok = begin_iteration<LV>(P, sm, n, lane);
  if (lane == 0) sm.ist[iACTION] = ok ? kActEval : kActDone;
  pr.tick(kPLS);
}

// ------------------------------------------------------------- the solver
// Work queue of a time-sliced launch (SchedArgs::source == 1): a ring of trajectory ids, popped and
// pushed by thread 0 of a workgroup.  Everything a suspended trajectory owns (its state record, the
// L-BFGS history) is written before the push and read after the pop, with device-scope fences in
// between: the next slice may run on another XCD, behind another L2.
__device__ inline int queue_pop(unsigned *ctl, const int *ring, int cap) {
  while (true) {
    unsigned h = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned t = __hip_atomic_load(&ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (h >= t) return -1;
    if (atomicCAS(&ctl[0], h, h + 1) == h) {
      int id = __hip_atomic_load(&ring[h % (unsigned)cap], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      return id;
    }
  }
}
__device__ inline void queue_push(unsigned *ctl, int *ring, int cap, int id) {
  __threadfence();
  unsigned t = atomicAdd(&ctl[2], 1u);
  __hip_atomic_store(&ring[t % (unsigned)cap], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // publish in reservation order, with release semantics: the ring entry (and everything fenced above) is visible
  // to whoever acquire-loads the new tail in queue_pop, also behind another XCD's L2
  while (true) {
    unsigned expect = t;
    if (__hip_atomic_compare_exchange_strong(&ctl[1], &expect, t + 1, __ATOMIC_RELEASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
  }
}

// dftpav_batch_trace: the evaluation that just finished (x, g in LDS, f in st[sF]) of the traced trajectory
__device__ inline void trace_eval(const DevBatch &Db, const Smem &sm, int b, int tid, int T) {
  if (Db.trace == nullptr || b < Db.trace_b || b >= Db.trace_b + Db.trace_n) return; // uniform
  const int idx = sm.ist[iPHASE] == 0 ? 0 : sm.ist[iEVALS];
  if (idx >= Db.trace_cap) return;
  const int n = Db.L.n, npad = Db.L.npad;
  double *tr0 = Db.trace + (size_t)(b - Db.trace_b) * (8 + (size_t)Db.trace_cap * (3 * npad + 8)); // this trajectory's block
  double *rec = tr0 + 8 + (size_t)idx * (3 * npad + 8);
  for (int e = tid; e < n; e += T) {
    rec[e] = sm.x[e];
    rec[npad + e] = sm.g[e];
    rec[2 * npad + e] = idx == 0 ? 0.0 : sm.d[e];
  }
  if (tid == 0) {
    rec[3 * npad + 0] = sm.st[sF];
    rec[3 * npad + 1] = idx == 0 ? 0.0 : sm.st[sSTP];
    rec[3 * npad + 2] = (double)sm.ist[iK];
    rec[3 * npad + 3] = idx == 0 ? 0.0 : (double)(sm.ist[iCOUNT] + 1);
    tr0[0] = (double)(idx + 1);
  }
  __syncthreads(); // wave 0 rewrites x / g / d next (every condition above is uniform)
}

// solver state of one trajectory <-> its record in DevBatch::state (everything lbfgs_advance keeps in LDS)
__device__ inline void state_io(const DevBatch &D, const Smem &sm, int b, int tid, int T, bool save) {
  const int npad = D.L.npad;
  double *rec = D.state + (size_t)b * D.state_stride;
  double *vecs[5] = {sm.x, sm.xp, sm.g, sm.gp, sm.d};
  for (int w = tid; w < 5 * npad; w += T) {
    int a = w / npad, e = w - a * npad;
    if (save) rec[w] = vecs[a][e];
    else vecs[a][e] = rec[w];
  }
  double *r2 = rec + 5 * npad;
  for (int w = tid; w < sNUM; w += T) {
    if (save) r2[w] = sm.st[w];
    else sm.st[w] = r2[w];
  }
  int *ri = reinterpret_cast<int *>(r2 + 24);
  for (int w = tid; w < iNUM; w += T) {
    if (save) ri[w] = sm.ist[w];
    else sm.ist[w] = ri[w];
  }
  // (ys and 1 / ys of the stored pairs live in DevBatch::histR, the history in histS: nothing more to move)
}

template <bool SUR, int LV, int MAXT>
__global__ void __launch_bounds__(MAXT) solver_kernel(const DevBatch *__restrict__ Dp, int mode, SchedArgs sched) {
  extern __shared__ double lds_raw[];
  // D0: the launched batch (queue, launch shape, role tables).  Inside a pass D is the batch the trajectory of
  // that pass belongs to -- D0, or for an adopted straggler of a chained solve the previous batch (sched.alt),
  // which has the same layout, parameters and shape by construction (dftpav_batch_solve_chained checks).
  const DevBatch &D0 = *Dp;
  if (mode == kModeSolve && sched.source >= 2) { // list launches: one workgroup per list entry, the rest leave at once
    const unsigned cnt = __hip_atomic_load(&D0.qctl[sched.source == 2 ? 4 : 5], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x >= cnt) return;
  }
  const DevBatch &D = D0; // prologue only; shadowed per pass below
  const DevLayout &L = D.L;
  const int tid = threadIdx.x, T = blockDim.x;
  const int lane = tid & 63;
  const int n = L.n;
  Smem sm;
  carve(sm, lds_raw, L, D.P.mem_size, T, D.e4_rounds, D.e4_groups, D.e4_left, D.e4_lcap, D.op_in_lds != 0, D.cor_in_lds != 0, SUR ? D.sur_np : 0,
        SUR && D.sur_coef_lds != 0);
  Prof pr;

  // ---- one-time staging: role tables and operators (the same for every trajectory of the batch)
  for (int i = tid; i < D.e4_groups; i += T) sm.gtab[i] = D.e4_gtab[i];
  for (int i = tid; i < D.e4_left; i += T) sm.ltab[i] = D.e4_ltab[i];
  for (int i = tid; i < D.e4_rounds * (T >> 6) * 3; i += T) sm.wtab[i] = D.e4_wave[i];
  for (int i = tid; i < D.e4_rounds * 2; i += T) sm.rtab[i] = D.e4_round[i];
  for (int i = tid; i < 4 * L.Ntot; i += T) sm.pgrp[i] = D.e4_piece[i];
  if (SUR) {
    for (int i = tid; i < D.sur.S; i += T) {
      sm.shead[i] = D.sur.total[i];
      sm.shead[16 + i] = D.sur.start[i];
      sm.shead[32 + i] = D.sur.rate(i);
    }
    for (int i = tid; i <= D.sur.S; i += T) reinterpret_cast<int *>(sm.shead + 48)[i] = D.sur.piece_off[i];
    for (int i = tid; i < D.sur.S; i += T) { // the obstacles' end states (dyn_obstacle_near), from the global tables
      double pd[2], vd[2], ad[2];
      sur_end_state(D.sur, i, pd, vd, ad);
      double *q = sm.shead + 64 + 6 * i;
      q[0] = pd[0]; q[1] = pd[1]; q[2] = vd[0]; q[3] = vd[1]; q[4] = ad[0]; q[5] = ad[1];
    }
  }
  if (SUR)
    for (int i = tid; i < D.sur_np; i += T) {
      sm.sdur[i] = D.sur.durations[i];
      if (D.sur.theta != nullptr) sm.sdur[D.sur_np + i] = D.sur.theta[i];
    }
  if (SUR && D.sur.bbox != nullptr)
    for (int i = tid; i < 4 * D.sur_np; i += T) sm.sdur[2 * D.sur_np + i] = D.sur.bbox[i];
  if (SUR && D.sur_coef_lds != 0)
    for (int i = tid; i < 12 * D.sur_np; i += T) sm.sdur[6 * D.sur_np + i] = D.sur.coeffs[i];
  for (int p = tid; p < L.Ntot; p += T) {
    int sg = 0, p0 = 0, N = 0, pt0s = 0, sgl = 0, ooff = 0;
    for (int s = 0, a = 0; s < L.M; s++) {
      bool in = p >= L.seg_piece0[s];
      sg = in ? s : sg;
      p0 = in ? L.seg_piece0[s] : p0;
      N = in ? L.piece_nums[s] : N;
      pt0s = in ? L.seg_pt0[s] : pt0s;
      sgl = in ? L.singuls[s] : sgl;
      ooff = in ? a : ooff;
      a += 6 * L.piece_nums[s] * (L.piece_nums[s] + 5);
    }
    int lp = p - p0;
    bool edge = (lp == 0 || lp == N - 1);
    int *pc = sm.pcinfo + 8 * p;
    // first constraint point of piece p: pieces of a segment are [Kd+1, K+1, ..., K+1, Kd+1] long
    pc[0] = pt0s + (lp == 0 ? 0 : (L.Kd + 1) + (lp - 1) * (L.K + 1));
    pc[1] = edge ? L.Kd : L.K;
    pc[2] = sg * 2 + (edge ? 1 : 0);
    pc[3] = sg;
    pc[4] = lp;
    pc[5] = N;
    pc[6] = sgl;
    pc[7] = ooff;
  }
  for (int row = tid; row < L.rhs_tot; row += T) {
    int sg = 0, r0 = 0, N = 0, x0 = 0;
    for (int s = 0; s < L.M; s++) {
      bool in = row >= L.seg_rhs0[s];
      sg = in ? s : sg;
      r0 = in ? L.seg_rhs0[s] : r0;
      N = in ? L.piece_nums[s] : N;
      x0 = in ? L.seg_x0[s] : x0;
    }
    int *ri = sm.rowinfo + 4 * row;
    ri[0] = sg;
    ri[1] = row - r0;
    ri[2] = N;
    ri[3] = x0;
  }
  if (D.op_in_lds) {
    int off = 0, offT = 0;
    for (int sg = 0; sg < L.M; sg++) {
      const int N = L.piece_nums[sg], cnt = 6 * N * (N + 5);
      for (int i = tid; i < cnt; i += T) {
        sm.opM[off + i] = D.opM[sg][i];
        const int col = i / (6 * N), r = i - col * 6 * N;
        sm.opMT[offT + col * (6 * N + kOpTPad) + r] = D.opMT[sg][i];
      }
      off += cnt;
      offT += (6 * N + kOpTPad) * (N + 5);
    }
  }

  // ---- one trajectory (or one slice of one) per pass
  for (int pass = 0;; pass++) {
    __syncthreads(); // the previous pass is done with LDS
    if (tid == 0) {
      int id = -1;
      if (mode != kModeSolve || sched.source == 0) {
        id = pass == 0 ? (int)blockIdx.x : -1;
      } else if (sched.source == 1) {
        id = queue_pop(D0.qctl, D0.queue, D0.qcap);
      } else if (pass == 0) {
        id = (sched.source == 2 ? D0.stragglers : D0.stragglers2)[blockIdx.x]; // blockIdx.x < count, checked on entry
        __threadfence();
      }
      sm.ist[iCUR] = id;
    }
    __syncthreads();
    // the same word for every lane, and known to the compiler as such: everything derived from it -- the trajectory's
    // base pointers, the fields of its batch descriptor -- lives in scalar registers and is fetched by scalar loads
    const int cur = __builtin_amdgcn_readfirstlane(sm.ist[iCUR]);
    if (cur < 0) break;
    const bool adopted = (cur & kAltTag) != 0;
    const int b = cur & (kAltTag - 1);
    // Db: the batch this trajectory's buffers live in.  Everything else (layout, parameters, launch shape,
    // operators) is read from D0 = D, whose loads the compiler may keep in scalar registers across passes.
    const DevBatch &Db = adopted ? *sched.alt : D0;
    const double *cor_b = Db.corridor + (size_t)b * L.H * 4 * D.NptsPad;
    double *hS_b = Db.histS + (size_t)b * D.P.mem_size * L.npad * 2;
    double *hU_b = Db.histU + (size_t)b * D.P.mem_size * 8, *hV_b = Db.histV + (size_t)b * D.P.mem_size * 8;
    double *hR_b = Db.histR + (size_t)b * D.P.mem_size * 2;
    const long long tick0 = wall_clock64();
    const bool resume = mode == kModeSolve && sched.source != 0 && Db.sflag[b] == 1;
    pr.start(Db.prof != nullptr && mode == kModeSolve, Db.prof + (size_t)b * 12, resume);
    __syncthreads(); // everybody has read iCUR before the state is restored over it

    // ---- per-trajectory staging: decision vector (or the suspended state) and half-planes
    if (resume) {
      state_io(Db, sm, b, tid, T, false);
    } else {
      const double *xsrc = (mode == kModeSolve) ? Db.x0 : (mode == kModeEval ? Db.x_in : Db.x_out);
      for (int e = tid; e < n; e += T) sm.x[e] = xsrc[(size_t)b * n + e];
      if (tid < iNUM) sm.ist[tid] = 0;
    }
    for (int w = tid; w < 12 * L.M; w += T)
      sm.bnd[w] = w < 6 * L.M ? Db.iniS[(size_t)b * L.M * 6 + w] : Db.finS[(size_t)b * L.M * 6 + (w - 6 * L.M)];
    if (D.cor_in_lds) { // the only read of the corridor from HBM: it stays in LDS for the whole pass
      const double *cb = Db.corridor + (size_t)b * L.H * 4 * D.NptsPad;
      const int pitch = (L.Npts + 63) / 64 * 64;
      for (int k = 0; k < 4 * L.H; k++)
        for (int pt = tid; pt < L.Npts; pt += T) sm.cor[k * pitch + pt] = cb[(size_t)k * D.NptsPad + pt];
    }
    __syncthreads();
    const int k_start = sm.ist[iK];
    if (tid < 64) prep_durations(D, sm, sm.x, lane);
    __syncthreads();

    block_eval<SUR>(D, cor_b, sm, sm.x, sm.g, pr); // x0, or the trial point the trajectory was suspended on
    if (mode == kModeSolve) trace_eval(Db, sm, b, tid, T);

    if (mode == kModeEval) {
      for (int e = tid; e < n; e += T) Db.g_out[(size_t)b * n + e] = sm.g[e];
      if (tid == 0) Db.f_eval[b] = sm.st[sF];
      return;
    }
    if (mode == kModeCoeffs) {
      for (int w = tid; w < 12 * L.Ntot; w += T) Db.coef_out[(size_t)b * 12 * L.Ntot + w] = sm.c[w];
      for (int sg = tid; sg < L.M; sg += T) Db.dt_out[(size_t)b * L.M + sg] = sm.seg[sg * 16 + 1];
      return;
    }

    // ---- lbfgs_optimize (lbfgs.hpp:440-751): wave 0 advances the solver state between evaluations
    bool finished = true;
    while (true) {
      if (tid < 64) {
        // the serial part of the trajectory: the other waves of the workgroup wait for it, the waves this one shares its
        // SIMD with belong to other trajectories -- let the arbiter prefer it
        __builtin_amdgcn_s_setprio(3);
        lbfgs_advance<LV>(D, sm, hS_b, hU_b, hV_b, hR_b, lane, pr);
        if (sm.ist[iACTION] == kActEval) prep_durations(D, sm, sm.x, lane); // the trial point is in place
        __builtin_amdgcn_s_setprio(0);
      }
      __syncthreads();
      if (sm.ist[iACTION] == kActDone) break;
      if (sched.source == 1 && sched.slice > 0 && sm.ist[iK] - k_start >= sched.slice) { // uniform
        finished = false;
        break;
      }
      block_eval<SUR>(D, cor_b, sm, sm.x, sm.g, pr);
      trace_eval(Db, sm, b, tid, T);
    }

    const long long spent = wall_clock64() - tick0;
    if (finished) {
      for (int e = tid; e < n; e += T) Db.x_out[(size_t)b * n + e] = sm.x[e];
      if (tid == 0) {
        const double fx = sm.st[sFX];
        const int ret = sm.ist[iRET];
        Db.f_out[b] = fx;
        Db.status[b] = ret;
        Db.iters[b] = sm.ist[iK];
        Db.evals[b] = sm.ist[iEVALS];
        Db.hist_sum[b] = ((long long)sm.ist[iHISTHI] << 32) | (unsigned int)sm.ist[iHISTLO];
        {
          double *rec = reinterpret_cast<double *>(Db.records + (size_t)16 * b); // the all-gather record, ready when the trajectory is
          rec[0] = fx;
          int *ri = reinterpret_cast<int *>(rec + 1);
          ri[0] = ret;
          ri[1] = sm.ist[iK];
        }
        if (Db.records_host != nullptr) { // (see device_types.h)
          double *rh = reinterpret_cast<double *>(Db.records_host + (size_t)16 * b);
          rh[0] = fx;
          int *rj = reinterpret_cast<int *>(rh + 1);
          rj[0] = ret;
          rj[1] = sm.ist[iK];
        }
        Db.ticks[b] = (resume ? Db.ticks[b] : 0) + spent; // time in service
        // flag_success, traj_optimizer.cpp:176-201
        int ok = (ret == 0 || ret == 1 || ret == 2 || ret == -1008 || ret == -1009) ? 1 : 0;
        if (fx >= D.P.fail_cost) ok = 0;
        Db.success[b] = ok;
        if (sched.source != 0) {
          Db.sflag[b] = 2;
          atomicSub(&Db.qctl[3], 1u);
        }
      }
    } else {
      state_io(Db, sm, b, tid, T, true);
      __syncthreads(); // all of the record is written (and fenced by thread 0 below) before the id is handed on
      if (tid == 0) {
        Db.ticks[b] = (resume ? Db.ticks[b] : 0) + spent;
        Db.sflag[b] = 1;
        // the end game is decided by the launched batch's own unfinished count; an adopted trajectory caught by
        // it goes to its own batch's second list (finished by a list launch of that batch right after this one)
        unsigned left = __hip_atomic_load(&D0.qctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left <= (unsigned)sched.hand_over) {
          __threadfence();
          if (adopted) {
            unsigned slot = atomicAdd(&Db.qctl[5], 1u);
            Db.stragglers2[slot] = b;
          } else {
            unsigned slot = atomicAdd(&Db.qctl[4], 1u);
            Db.stragglers[slot] = b;
          }
          __threadfence();
        } else {
          queue_push(D0.qctl, D0.queue, D0.qcap, cur);
        }
      }
    }
    if (mode != kModeSolve || sched.source != 1) break;
  }
}

// Chained solves: the stragglers the previous batch's queue launch left behind join the queue of the next
// batch (tagged), so that the long tail of one batch is worked off inside the full-occupancy phase of the next
// instead of on a nearly empty device.  One workgroup; runs between the queue reset and the queue launch.
__global__ void adopt_kernel(unsigned *ctl, int *queue, int qcap, unsigned *prev_ctl, const int *prev_list) {
  const unsigned cnt = prev_ctl[4], tail = ctl[1];
  for (unsigned i = threadIdx.x; i < cnt; i += blockDim.x) queue[(tail + i) % (unsigned)qcap] = prev_list[i] | kAltTag;
  __syncthreads();
  if (threadIdx.x == 0) {
    ctl[1] = tail + cnt;
    ctl[2] = tail + cnt;
    prev_ctl[4] = 0; // consumed
    prev_ctl[5] = 0; // the second list starts empty
  }
}
hipError_t launch_adopt(const DevBatch &D, const DevBatch &prev, hipStream_t stream) {
  hipLaunchKernelGGL(adopt_kernel, dim3(1), dim3(256), 0, stream, D.qctl, D.queue, D.qcap, prev.qctl, prev.stragglers);
  return hipGetLastError();
}

// ------------------------------------------------------------- host launchers
template <bool SUR, int LV, int MAXT>
static hipError_t launch_variant(const DevBatch *d_dev, int grid, int mode, int threads, size_t lds, SchedArgs sched, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solver_kernel<SUR, LV, MAXT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((solver_kernel<SUR, LV, MAXT>), dim3(grid), dim3(threads), lds, stream, d_dev, mode, sched);
  return hipGetLastError();
}
template <bool SUR>
static hipError_t launch_lv(int n, const DevBatch *d_dev, int grid, int mode, int threads, size_t lds, SchedArgs sched, hipStream_t stream) {
  if (n <= 16) return launch_variant<SUR, 4, 512>(d_dev, grid, mode, threads, lds, sched, stream);
  if (n <= 32) return launch_variant<SUR, 5, 512>(d_dev, grid, mode, threads, lds, sched, stream);
  return launch_variant<SUR, 6, 512>(d_dev, grid, mode, threads, lds, sched, stream);
}

// d_dev: device copy of the DevBatch `D` describes; grid: workgroups to launch (D.B unless the launch is scheduled)
hipError_t launch_solver(const DevBatch &D, const DevBatch *d_dev, int mode, int threads, int grid, SchedArgs sched,
                         hipStream_t stream) {
  size_t lds = solver_lds_bytes(D.L, D.P, threads, D.op_in_lds != 0, D.cor_in_lds != 0, D.sur.S > 0 ? D.sur_np : 0,
                                D.sur.S > 0 && D.sur_coef_lds != 0);
  if (std::getenv("DFTPAV_VERBOSE"))
    std::fprintf(stderr, "[dftpav] launch mode %d: grid %d x %d threads, %zu B of LDS (%d workgroups per CU by LDS), obstacle coefficients in %s\n", mode, grid,
                 threads, lds, (int)((160 * 1024) / (lds + 64)), D.sur.S > 0 ? (D.sur_coef_lds ? "LDS" : "global memory") : "-");
  if (D.sur.S > 0) return launch_lv<true>(D.L.n, d_dev, grid, mode, threads, lds, sched, stream);
  return launch_lv<false>(D.L.n, d_dev, grid, mode, threads, lds, sched, stream);
}

} // namespace dftpav
