// solver.hip — gfx950 kernels of the batched MINCO / L-BFGS trajectory solver.
//
// One workgroup owns one trajectory for its whole life: the L-BFGS outer loop,
// the Lewis-Overton line search, every cost/gradient evaluation and the
// two-loop recursion all run inside a single launch with no host round trip.
//
//   * threads <-> constraint points (the sample loop of
//     PolyTrajOptimizer::addPVAGradCost2CT, traj_optimizer.cpp:486-779)
//   * MINCO forward/adjoint banded solves (poly_traj_utils.hpp:805-852) are
//     applied as the precomputed dense operator A_N^{-1}|_{N+5 columns}: the band
//     matrix depends only on N (poly_traj_utils.hpp:895-947) and only N+5 RHS
//     rows are ever non-zero (poly_traj_utils.hpp:968-977), so both solves become
//     small lane-parallel mat-vecs instead of a 6N-step sequential substitution
//   * per-sample gradients are kept as d/dsigma, d/dsigma', d/dsigma'' (6 values)
//     and expanded onto the 6x2 piece coefficients by a transposed LDS reduction
//     with a fixed summation order (deterministic run to run)
//   * wave 0 runs the L-BFGS vector algebra (n <= 256 decision variables, lanes
//     over elements, cross-lane reductions) while the other waves wait at the
//     workgroup barrier
//
// All arithmetic is fp64, as in the reference.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "traj_math.h"

namespace dftpav {

// ------------------------------------------------------------------ LDS carve
struct Smem {
  double *x, *xp, *g, *gp, *d;
  double *seg;  // [M][16]  0:T 1:dt 2..7:t^k 8..13:t^-k
  double *stab; // [M][2][Kmax+1] accumulated sample offsets (s1 += step, traj_optimizer.cpp:513)
  double *rhs;  // [rhs_tot][2]
  double *b, *c, *gdC; // [6*Ntot][2]
  double *adj;  // [rhs_tot][2]
  double *part; // [8][T]
  double *pE, *pGsm, *pGdT, *pCost, *pChain; // [Ntot]
  double *ys, *alpha; // [mem]
  double *scal; // [16]
  int *flag;    // [8]
};

__host__ __device__ inline size_t smem_doubles(const DevLayout &L, int mem, int T) {
  size_t n = 0;
  n += 5 * (size_t)L.npad;
  n += (size_t)L.M * 16;
  n += (size_t)L.M * 2 * (L.Kmax + 1);
  n += (size_t)L.rhs_tot * 2;
  n += 3 * (size_t)L.Ntot * 12;
  n += (size_t)L.rhs_tot * 2;
  n += 8 * (size_t)T;
  n += 5 * (size_t)L.Ntot;
  n += 2 * (size_t)mem;
  n += 16;
  return n;
}

size_t solver_lds_bytes(const DevLayout &L, const DevParams &P, int threads) {
  return smem_doubles(L, P.mem_size, threads) * sizeof(double) + 8 * sizeof(int);
}

int solver_threads(const DevLayout &L) {
  // one pass over the constraint points when they fit a workgroup, otherwise
  // the fewest passes with the least idle lanes
  int passes = (L.Npts + kMaxThreads - 1) / kMaxThreads;
  int per = (L.Npts + passes - 1) / passes;
  int T = ((per + kWave - 1) / kWave) * kWave;
  int need = 14 * L.Ntot; // the reduction stage likes 14 threads per piece
  if (T < need) T = ((need + kWave - 1) / kWave) * kWave;
  if (T < 128) T = 128;
  if (T > kMaxThreads) T = kMaxThreads;
  return T;
}

__device__ inline void carve(Smem &s, double *base, const DevLayout &L, int mem, int T) {
  double *p = base;
  s.x = p; p += L.npad;
  s.xp = p; p += L.npad;
  s.g = p; p += L.npad;
  s.gp = p; p += L.npad;
  s.d = p; p += L.npad;
  s.seg = p; p += L.M * 16;
  s.stab = p; p += L.M * 2 * (L.Kmax + 1);
  s.rhs = p; p += L.rhs_tot * 2;
  s.b = p; p += L.Ntot * 12;
  s.c = p; p += L.Ntot * 12;
  s.gdC = p; p += L.Ntot * 12;
  s.adj = p; p += L.rhs_tot * 2;
  s.part = p; p += 8 * T;
  s.pE = p; p += L.Ntot;
  s.pGsm = p; p += L.Ntot;
  s.pGdT = p; p += L.Ntot;
  s.pCost = p; p += L.Ntot;
  s.pChain = p; p += L.Ntot;
  s.ys = p; p += mem;
  s.alpha = p; p += mem;
  s.scal = p; p += 16;
  s.flag = reinterpret_cast<int *>(p);
}

// ------------------------------------------------------------ device helpers
__device__ inline int seg_of_piece(const DevLayout &L, int p) {
  int s = 0;
  while (s + 1 < L.M && p >= L.seg_piece0[s + 1]) ++s;
  return s;
}

// Cross-lane butterfly over the 64 lanes of a wave: v += v(lane ^ o) for o = 32,16,..,1.
// fp addition is commutative, so every lane ends with the same bits; the CPU
// oracle's device-order mode replays exactly this tree (oracle/dftpav_oracle_dev.cpp).
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// strided plane loader over the component-major corridor of one trajectory
struct GlobalPlanes {
  const double *base; // &corridor[b][0][pt]
  size_t pitch;       // NptsPad
  __device__ inline void operator()(int k, double &n0, double &n1, double &q0, double &q1) const {
    n0 = base[(size_t)(4 * k + 0) * pitch];
    n1 = base[(size_t)(4 * k + 1) * pitch];
    q0 = base[(size_t)(4 * k + 2) * pitch];
    q1 = base[(size_t)(4 * k + 3) * pitch];
  }
};

// constraint point `pt` of trajectory b -> {dJ/dsigma, dJ/dsigma', dJ/dsigma'', gdT, cost}
template <bool SUR>
__device__ inline void sample_point(const DevBatch &D, int b, const Smem &sm, int pt, double out[8]) {
  const DevLayout &L = D.L;
  int p = D.pt_piece[pt];
  SampleIn in;
  in.j = D.pt_j[pt];
  int sg = seg_of_piece(L, p);
  in.lp = p - L.seg_piece0[sg];
  in.N = L.piece_nums[sg];
  bool edge = (in.lp == 0 || in.lp == in.N - 1);
  in.K = edge ? L.Kd : L.K;
  in.dt = sm.seg[sg * 16 + 1];
  in.s1 = sm.stab[(sg * 2 + (edge ? 1 : 0)) * (L.Kmax + 1) + in.j];
  in.cc = sm.c + 12 * p;
  in.singul = L.singuls[sg];
  in.epis = D.epis;
  in.H = L.H;
  in.trajid = sg;
  in.trajtime = sg == 0 ? 0.0 : sm.seg[(sg - 1) * 16]; // trajtimes[trajid] = T_{i-1}, traj_optimizer.cpp:230-234
  in.t_now = D.t_now;
  GlobalPlanes pl{D.corridor + (size_t)b * L.H * 4 * D.NptsPad + pt, (size_t)D.NptsPad};
  sample_point_math<SUR>(D.P, D.sur, in, pl, out);
}

// ----------------------------------------------------- cost + gradient
// PolyTrajOptimizer::costFunctionCallback (traj_optimizer.cpp:206-350) for the
// decision vector x (LDS) of trajectory b; writes g (LDS), returns f to every thread.
template <bool SUR>
__device__ double block_eval(const DevBatch &D, int b, const Smem &sm, const double *x, double *g) {
  const DevLayout &L = D.L;
  const DevParams &P = D.P;
  const int tid = threadIdx.x, T = blockDim.x;
  const double *iniS = D.iniS + (size_t)b * L.M * 6;
  const double *finS = D.finS + (size_t)b * L.M * 6;

  // ---- E1: segment durations, sample-offset tables, MINCO right-hand sides
  for (int w = tid; w < L.M + 2 * L.M + 2 * L.rhs_tot; w += T) {
    if (w < L.M) {
      int sg = w;
      double Tr = virtual_to_real(x[L.x_tau0 + sg], P.mini_T);
      double dt = Tr / L.piece_nums[sg];
      double *s = sm.seg + sg * 16;
      s[0] = Tr;
      s[1] = dt;
      duration_powers(dt, s + 2); // poly_traj_utils.hpp:961-966
    } else if (w < 3 * L.M) {
      int q = w - L.M, sg = q >> 1, which = q & 1;
      int K = which ? L.Kd : L.K;
      double Tr = virtual_to_real(x[L.x_tau0 + sg], P.mini_T);
      double dt = Tr / L.piece_nums[sg];
      double step = dt / K;
      double *tab = sm.stab + (sg * 2 + which) * (L.Kmax + 1);
      double s1 = 0.0;
      for (int j = 0; j <= K; j++) {
        tab[j] = s1;
        s1 += step; // traj_optimizer.cpp:513
      }
    } else {
      int q = w - 3 * L.M;
      int row = q >> 1, d = q & 1;
      int sg = 0;
      while (sg + 1 < L.M && row >= L.seg_rhs0[sg + 1]) ++sg;
      int col = row - L.seg_rhs0[sg];
      int N = L.piece_nums[sg];
      double Tr = virtual_to_real(x[L.x_tau0 + sg], P.mini_T);
      double dt = Tr / N;
      double v;
      if (col < 3) { // head p, v*dt, a*dt^2 (poly_traj_utils.hpp:969-971), junction override traj_optimizer.cpp:273-277
        if (col == 0) {
          v = sg > 0 ? x[L.x_gear0 + 2 * (sg - 1) + d] : iniS[sg * 6 + d];
        } else if (col == 1) {
          double hv;
          if (sg > 0) {
            double th = x[L.x_ang0 + sg - 1];
            hv = d == 0 ? -P.non_sinv * p_cos(th) : -P.non_sinv * p_sin(th);
          } else {
            hv = iniS[sg * 6 + 2 + d];
          }
          v = hv * dt;
        } else {
          v = iniS[sg * 6 + 4 + d] * (dt * dt);
        }
      } else if (col < N + 2) {
        v = x[L.seg_x0[sg] + 2 * (col - 3) + d];
      } else { // tail, poly_traj_utils.hpp:975-977, junction override traj_optimizer.cpp:278-282
        int k = col - (N + 2);
        if (k == 0) {
          v = sg < L.M - 1 ? x[L.x_gear0 + 2 * sg + d] : finS[sg * 6 + d];
        } else if (k == 1) {
          double tv;
          if (sg < L.M - 1) {
            double th = x[L.x_ang0 + sg];
            tv = d == 0 ? P.non_sinv * p_cos(th) : P.non_sinv * p_sin(th);
          } else {
            tv = finS[sg * 6 + 2 + d];
          }
          v = tv * dt;
        } else {
          v = finS[sg * 6 + 4 + d] * (dt * dt);
        }
      }
      sm.rhs[2 * row + d] = v;
    }
  }
  __syncthreads();

  // ---- E2: b = A^{-1} rhs (dense operator), c = b * t^-k   (MinJerkOpt::generate, poly_traj_utils.hpp:979-984)
  for (int w = tid; w < 12 * L.Ntot; w += T) {
    int r = w >> 1, d = w & 1;
    int p = r / 6, k = r - 6 * p;
    int sg = seg_of_piece(L, p);
    int N = L.piece_nums[sg];
    int lr = r - 6 * L.seg_piece0[sg];
    const double *Mrow = D.opM[sg] + (size_t)lr * (N + 5);
    const double *rh = sm.rhs + 2 * L.seg_rhs0[sg] + d;
    double acc = 0.0;
    for (int col = 0; col < N + 5; col++) acc += Mrow[col] * rh[2 * col];
    sm.b[w] = acc;
    sm.c[w] = acc * sm.seg[sg * 16 + 8 + k];
  }
  __syncthreads();

  // ---- E3: jerk energy and its partials per piece (poly_traj_utils.hpp:998-1035)
  for (int p = tid; p < L.Ntot; p += T) {
    int sg = seg_of_piece(L, p);
    double en, gsm;
    piece_smoothness(sm.c + 12 * p, sm.seg + sg * 16 + 2, en, gsm, sm.gdC + 12 * p);
    sm.pE[p] = en;
    sm.pGsm[p] = gsm;
    sm.pGdT[p] = 0.0;
    sm.pCost[p] = 0.0;
  }

  // ---- E4: penalty integral over the constraint points (traj_optimizer.cpp:486-779)
  for (int base = 0; base < L.Npts; base += T) {
    int pt = base + tid;
    double o[8];
    if (pt < L.Npts) {
      sample_point<SUR>(D, b, sm, pt, o);
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) o[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) sm.part[k * T + tid] = o[k];
    __syncthreads();
    // transposed reduction: 12 threads per piece expand (A,B,C) onto gdC, 2 more sum gdT and cost
    int lim = base + T < L.Npts ? base + T : L.Npts;
    for (int w = tid; w < 14 * L.Ntot; w += T) {
      int p = w / 14, q = w - 14 * p;
      int sg = seg_of_piece(L, p);
      int lp = p - L.seg_piece0[sg];
      int N = L.piece_nums[sg];
      bool edge = (lp == 0 || lp == N - 1);
      int K = edge ? L.Kd : L.K;
      // first constraint point of piece p: pieces of a segment are [Kd+1, K+1, ..., K+1, Kd+1] long
      int pt0 = L.seg_pt0[sg] + (lp == 0 ? 0 : (L.Kd + 1) + (lp - 1) * (L.K + 1));
      int j0 = base > pt0 ? base - pt0 : 0;
      int j1 = (pt0 + K + 1 < lim ? pt0 + K + 1 : lim) - pt0; // exclusive
      if (j1 <= j0) continue;
      const double *tab = sm.stab + (sg * 2 + (edge ? 1 : 0)) * (L.Kmax + 1);
      if (q < 12) {
        int k = q >> 1, d = q & 1;
        const double *pa = sm.part + (0 + d) * T + (pt0 - base);
        const double *pb = sm.part + (2 + d) * T + (pt0 - base);
        const double *pc = sm.part + (4 + d) * T + (pt0 - base);
        double acc = sm.gdC[12 * p + q]; // continue the chain that starts at the smoothness gradient
        for (int j = j0; j < j1; j++) {
          double b0, b1, b2;
          beta_row(k, tab[j], b0, b1, b2);
          acc += b0 * pa[j] + b1 * pb[j] + b2 * pc[j];
        }
        sm.gdC[12 * p + q] = acc;
      } else {
        const double *pv = sm.part + (q == 12 ? 6 : 7) * T + (pt0 - base);
        double acc = q == 12 ? sm.pGdT[p] : sm.pCost[p];
        for (int j = j0; j < j1; j++) acc += pv[j];
        if (q == 12) sm.pGdT[p] = acc;
        else sm.pCost[p] = acc;
      }
    }
    __syncthreads();
  }

  // ---- E5: adjoint through A^{-T} (MinJerkOpt::calGrads_PT, poly_traj_utils.hpp:1037-1064)
  for (int w = tid; w < 2 * L.rhs_tot + L.Ntot; w += T) {
    if (w < 2 * L.rhs_tot) {
      int row = w >> 1, d = w & 1;
      int sg = 0;
      while (sg + 1 < L.M && row >= L.seg_rhs0[sg + 1]) ++sg;
      int col = row - L.seg_rhs0[sg];
      int N = L.piece_nums[sg];
      const double *MT = D.opMT[sg] + (size_t)col * 6 * N;
      const double *gc = sm.gdC + 12 * L.seg_piece0[sg] + d;
      const double *tInv = sm.seg + sg * 16 + 8;
      double acc = 0.0;
      for (int p = 0; p < N; p++) {
#pragma unroll
        for (int k = 0; k < 6; k++) acc += MT[6 * p + k] * (gc[2 * (6 * p + k)] * tInv[k]);
      }
      sm.adj[w] = acc;
    } else {
      int p = w - 2 * L.rhs_tot;
      int sg = seg_of_piece(L, p);
      const double *tInv = sm.seg + sg * 16 + 8;
      double gdtInv[6] = {0.0, -1.0 * tInv[2], -2.0 * tInv[3], -3.0 * tInv[4], -4.0 * tInv[5],
                          -5.0 * tInv[5] * tInv[1]}; // poly_traj_utils.hpp:1054-1060
      const double *gc = sm.gdC + 12 * p;
      const double *bb = sm.b + 12 * p;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) acc += gdtInv[k] * (gc[2 * k] * bb[2 * k] + gc[2 * k + 1] * bb[2 * k + 1]);
      sm.pChain[p] = acc;
    }
  }
  __syncthreads();

  // ---- E6: assemble g and f (traj_optimizer.cpp:299-344)
  for (int e = tid; e < L.n + 1; e += T) {
    if (e == L.n) {
      double sm_cost = 0.0, pen = 0.0, tc = 0.0;
      for (int sg = 0; sg < L.M; sg++) {
        double en = 0.0, pc = 0.0;
        for (int p = L.seg_piece0[sg]; p < L.seg_piece0[sg + 1]; p++) {
          en += sm.pE[p];
          pc += sm.pCost[p];
        }
        sm_cost += en;
        pen += pc;
        tc += sm.seg[sg * 16] * P.wei_time;
      }
      sm.scal[0] = sm_cost + tc + pen;
    } else if (e < L.x_tau0) { // waypoints: gdP = rows 6i+5 of the adjoint
      int sg = 0;
      while (sg + 1 < L.M && e >= L.seg_x0[sg + 1]) ++sg;
      int q = e - L.seg_x0[sg];
      int wp = q >> 1, d = q & 1;
      g[e] = sm.adj[2 * (L.seg_rhs0[sg] + 3 + wp) + d];
    } else if (e < L.x_gear0) { // tau: VirtualTGradCost, traj_optimizer.cpp:405-419
      int sg = e - L.x_tau0;
      int N = L.piece_nums[sg];
      const double *seg = sm.seg + sg * 16;
      double dt = seg[1];
      double gdT = 0.0;
      for (int p = L.seg_piece0[sg]; p < L.seg_piece0[sg + 1]; p++) gdT += sm.pGsm[p];
      for (int p = L.seg_piece0[sg]; p < L.seg_piece0[sg + 1]; p++) gdT += sm.pGdT[p];
      // boundary-scaling terms, poly_traj_utils.hpp:1050-1053 (with the junction-overridden head/tail v)
      const double *ad = sm.adj + 2 * L.seg_rhs0[sg];
      double hv[2], tv[2];
      if (sg > 0) {
        double th = x[L.x_ang0 + sg - 1];
        hv[0] = -P.non_sinv * p_cos(th);
        hv[1] = -P.non_sinv * p_sin(th);
      } else {
        hv[0] = iniS[sg * 6 + 2];
        hv[1] = iniS[sg * 6 + 3];
      }
      if (sg < L.M - 1) {
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
      for (int p = L.seg_piece0[sg]; p < L.seg_piece0[sg + 1]; p++) gdT += sm.pChain[p];
      g[e] = (gdT / N + P.wei_time) * virtual_to_real_grad(x[e]);
    } else if (e < L.x_ang0) { // gear position, traj_optimizer.cpp:307-320
      int q = e - L.x_gear0;
      int i = q >> 1, d = q & 1;
      double v = 0.0;
      if (P.gear_opt) {
        int Ni = L.piece_nums[i];
        v += sm.adj[2 * (L.seg_rhs0[i] + Ni + 2) + d] * 1.0; // gdTail.col(0) of segment i
        v += sm.adj[2 * (L.seg_rhs0[i + 1] + 0) + d] * 1.0;  // gdHead.col(0) of segment i+1
      }
      g[e] = v;
    } else { // gear angle
      int i = e - L.x_ang0;
      double v = 0.0;
      if (P.gear_opt) {
        double th = x[e];
        int Ni = L.piece_nums[i];
        double dti = sm.seg[i * 16 + 1], dtn = sm.seg[(i + 1) * 16 + 1];
        double ft0 = sm.adj[2 * (L.seg_rhs0[i] + Ni + 3) + 0] * dti, ft1 = sm.adj[2 * (L.seg_rhs0[i] + Ni + 3) + 1] * dti;
        double hd0 = sm.adj[2 * (L.seg_rhs0[i + 1] + 1) + 0] * dtn, hd1 = sm.adj[2 * (L.seg_rhs0[i + 1] + 1) + 1] * dtn;
        v += ft0 * (-P.non_sinv * p_sin(th)) + ft1 * (P.non_sinv * p_cos(th));
        v += hd0 * (P.non_sinv * p_sin(th)) + hd1 * (-P.non_sinv * p_cos(th));
      }
      g[e] = v;
    }
  }
  __syncthreads();
  return sm.scal[0];
}

// ------------------------------------------------------------- the solver
enum LsState { kLsContinue = 0, kLsDone = 1 };

// lbfgs_optimize (lbfgs.hpp:440-751) + line_search_lewisoverton (lbfgs.hpp:276-390).
// Vector algebra on wave 0: element e of an n-vector lives at LDS index e, lanes stride over e.
template <bool SUR>
__global__ void __launch_bounds__(kMaxThreads) solver_kernel(DevBatch D, int mode) {
  extern __shared__ double lds_raw[];
  const DevLayout &L = D.L;
  const DevParams &P = D.P;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, T = blockDim.x;
  const int lane = tid & 63;
  const bool w0 = tid < 64;
  const int n = L.n;
  Smem sm;
  carve(sm, lds_raw, L, P.mem_size, T);
  const long long tick0 = wall_clock64();

  const double *xsrc = (mode == kModeSolve) ? D.x0 : (mode == kModeEval ? D.x_in : D.x_out);
  for (int e = tid; e < n; e += T) sm.x[e] = xsrc[(size_t)b * n + e];
  __syncthreads();

  double fx = block_eval<SUR>(D, b, sm, sm.x, sm.g);

  if (mode == kModeEval) {
    for (int e = tid; e < n; e += T) D.g_out[(size_t)b * n + e] = sm.g[e];
    if (tid == 0) D.f_out[b] = fx;
    return;
  }
  if (mode == kModeCoeffs) {
    for (int w = tid; w < 12 * L.Ntot; w += T) D.coef_out[(size_t)b * 12 * L.Ntot + w] = sm.c[w];
    for (int sg = tid; sg < L.M; sg += T) D.dt_out[(size_t)b * L.M + sg] = sm.seg[sg * 16 + 1];
    return;
  }

  // ---------------- L-BFGS state (uniform across the workgroup)
  const int m = P.mem_size;
  double *hS = D.histS + (size_t)b * m * L.npad;
  double *hY = D.histY + (size_t)b * m * L.npad;
  int evals = 1;
  int k = 0, end = 0, bound = 0, ret = 0;
  long long hist_sum = 0;
  double step = 0.0;
  double pf[8]; // past <= 8
  pf[0] = fx;

  // initial direction and stationarity test, lbfgs.hpp:533-547
  if (w0) {
    double gmax = 0.0, xmax = 0.0, dd = 0.0;
    for (int e = lane; e < n; e += 64) {
      double gv = sm.g[e];
      sm.d[e] = -gv;
      gmax = fmax(gmax, fabs(gv));
      xmax = fmax(xmax, fabs(sm.x[e]));
      dd += gv * gv;
    }
    gmax = wave_max(gmax);
    xmax = wave_max(xmax);
    dd = wave_sum(dd);
    if (lane == 0) {
      sm.flag[0] = (gmax / fmax(1.0, xmax) < P.g_epsilon) ? 1 : 0;
      sm.scal[1] = 1.0 / sqrt(dd);
    }
  }
  __syncthreads();
  bool done = sm.flag[0] != 0;
  if (done) {
    ret = 0; // LBFGS_CONVERGENCE
  } else {
    step = sm.scal[1];
    k = 1;
  }
  __syncthreads();

  while (!done) {
    // ---- start of an outer iteration: xp = x, gp = g, line-search setup (lbfgs.hpp:559-574, 290-309)
    double finit = fx;
    double dginit = 0.0;
    if (w0) {
      double acc = 0.0;
      for (int e = lane; e < n; e += 64) {
        sm.xp[e] = sm.x[e];
        double gv = sm.g[e];
        sm.gp[e] = gv;
        acc += gv * sm.d[e];
      }
      acc = wave_sum(acc);
      if (lane == 0) sm.scal[2] = acc;
    }
    __syncthreads();
    dginit = sm.scal[2];
    int ls = 0;
    bool ls_fail = false;
    if (!(step > 0.0)) {
      ls = -1006; // LBFGSERR_INVALIDPARAMETERS
      ls_fail = true;
    } else if (0.0 < dginit) {
      ls = -1005; // LBFGSERR_INCREASEGRADIENT
      ls_fail = true;
    }
    const double dgtest = P.f_dec_coeff * dginit;
    const double dstest = P.s_curv_coeff * dginit;
    int count = 0;
    bool brackt = false, touched = false;
    double mu = 0.0, nu = P.max_step;
    double stp = step;

    // ---- line search (lbfgs.hpp:312-389)
    while (!ls_fail) {
      for (int e = tid; e < n; e += T) sm.x[e] = sm.xp[e] + stp * sm.d[e];
      __syncthreads();
      fx = block_eval<SUR>(D, b, sm, sm.x, sm.g);
      ++count;
      ++evals;
      if (isinf(fx) || isnan(fx)) {
        ls = -1012; // LBFGSERR_INVALID_FUNCVAL
        break;
      }
      if (P.past > 0 && fabs(finit - fx) / (fabs(finit) + 1.0) < P.delta / P.past) { // lbfgs.hpp:326-329
        ls = count;
        break;
      }
      if (fx > finit + stp * dgtest) {
        nu = stp;
        brackt = true;
      } else {
        if (w0) {
          double acc = 0.0;
          for (int e = lane; e < n; e += 64) acc += sm.g[e] * sm.d[e];
          acc = wave_sum(acc);
          if (lane == 0) sm.scal[3] = acc;
        }
        __syncthreads();
        double gs = sm.scal[3];
        __syncthreads();
        if (gs < dstest) {
          mu = stp;
        } else {
          ls = count;
          break;
        }
      }
      if (P.max_linesearch <= count) {
        ls = -1009; // LBFGSERR_MAXIMUMLINESEARCH
        break;
      }
      if (brackt && (nu - mu) < P.machine_prec * nu) {
        ls = -1007; // LBFGSERR_WIDTHTOOSMALL
        break;
      }
      if (brackt) stp = 0.5 * (mu + nu);
      else stp *= 2.0;
      if (stp < P.min_step) {
        ls = -1011; // LBFGSERR_MINIMUMSTEP
        break;
      }
      if (stp > P.max_step) {
        if (touched) {
          ls = -1010; // LBFGSERR_MAXIMUMSTEP
          break;
        }
        touched = true;
        stp = P.max_step;
      }
    }
    step = stp;

    if (ls < 0) { // revert x and g, keep fx (lbfgs.hpp:604-611)
      for (int e = tid; e < n; e += T) {
        sm.x[e] = sm.xp[e];
        sm.g[e] = sm.gp[e];
      }
      ret = ls;
      break;
    }

    // ---- convergence / stopping tests (lbfgs.hpp:628-666)
    if (w0) {
      double gmax = 0.0, xmax = 0.0;
      for (int e = lane; e < n; e += 64) {
        gmax = fmax(gmax, fabs(sm.g[e]));
        xmax = fmax(xmax, fabs(sm.x[e]));
      }
      gmax = wave_max(gmax);
      xmax = wave_max(xmax);
      if (lane == 0) sm.flag[1] = (gmax / fmax(1.0, xmax) < P.g_epsilon) ? 1 : 0;
    }
    __syncthreads();
    if (sm.flag[1]) {
      ret = 0; // LBFGS_CONVERGENCE
      break;
    }
    if (0 < P.past) {
      if (P.past <= k) {
        double rate = fabs(pf[k % P.past] - fx) / fmax(1.0, fabs(fx));
        if (rate < P.delta) {
          ret = 1; // LBFGS_STOP
          break;
        }
      }
      pf[k % P.past] = fx;
    }
    if (P.max_iterations != 0 && P.max_iterations <= k) {
      ret = -1008; // LBFGSERR_MAXIMUMITERATION
      break;
    }
    ++k;

    // ---- history update + two-loop recursion on wave 0 (lbfgs.hpp:676-740)
    if (w0) {
      double *sc = hS + (size_t)end * L.npad, *yc = hY + (size_t)end * L.npad;
      double ys = 0.0, yy = 0.0, ss = 0.0, gpgp = 0.0;
      for (int e = lane; e < n; e += 64) {
        double sv = sm.x[e] - sm.xp[e];
        double yv = sm.g[e] - sm.gp[e];
        sc[e] = sv;
        yc[e] = yv;
        ys += yv * sv;
        yy += yv * yv;
        ss += sv * sv;
        double gpv = sm.gp[e];
        gpgp += gpv * gpv;
        sm.d[e] = -sm.g[e];
      }
      ys = wave_sum(ys);
      yy = wave_sum(yy);
      ss = wave_sum(ss);
      gpgp = wave_sum(gpgp);
      if (lane == 0) sm.ys[end] = ys;
      double cau = ss * sqrt(gpgp) * P.cautious_factor;
      int nb = bound, ne = end;
      if (ys > cau) {
        ++nb;
        nb = m < nb ? m : nb;
        ne = (end + 1) % m;
        // make this wave's own global stores visible to its own loads below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        int j = ne;
        for (int i = 0; i < nb; ++i) {
          j = (j + m - 1) % m;
          const double *sj = hS + (size_t)j * L.npad, *yj = hY + (size_t)j * L.npad;
          double acc = 0.0;
          for (int e = lane; e < n; e += 64) acc += sj[e] * sm.d[e];
          acc = wave_sum(acc);
          double ysj = (j == end) ? ys : sm.ys[j];
          double a = acc / ysj;
          if (lane == 0) sm.alpha[j] = a;
          double na = -a;
          for (int e = lane; e < n; e += 64) sm.d[e] += na * yj[e];
        }
        double sc0 = ys / yy;
        for (int e = lane; e < n; e += 64) sm.d[e] *= sc0;
        for (int i = 0; i < nb; ++i) {
          const double *sj = hS + (size_t)j * L.npad, *yj = hY + (size_t)j * L.npad;
          double acc = 0.0;
          for (int e = lane; e < n; e += 64) acc += yj[e] * sm.d[e];
          acc = wave_sum(acc);
          double ysj = (j == end) ? ys : sm.ys[j];
          double beta = acc / ysj;
          double cf = sm.alpha[j] - beta;
          for (int e = lane; e < n; e += 64) sm.d[e] += cf * sj[e];
          j = (j + 1) % m;
        }
      }
      if (lane == 0) {
        sm.flag[2] = nb;
        sm.flag[3] = ne;
      }
    }
    __syncthreads();
    {
      int nb = sm.flag[2], ne = sm.flag[3];
      if (ne != end) hist_sum += nb;
      bound = nb;
      end = ne;
    }
    step = 1.0;
    __syncthreads();
  }

  __syncthreads();
  for (int e = tid; e < n; e += T) D.x_out[(size_t)b * n + e] = sm.x[e];
  if (tid == 0) {
    D.f_out[b] = fx;
    D.status[b] = ret;
    D.iters[b] = k;
    D.evals[b] = evals;
    D.hist_sum[b] = hist_sum;
    D.ticks[b] = wall_clock64() - tick0;
    // flag_success, traj_optimizer.cpp:176-201
    int ok = (ret == 0 || ret == 1 || ret == 2 || ret == -1008 || ret == -1009) ? 1 : 0;
    if (fx >= P.fail_cost) ok = 0;
    D.success[b] = ok;
  }
}

// {f64 cost, i32 status, i32 iters} records for the all-gather of SURVEY §8(e)
__global__ void pack_results_kernel(const double *f, const int *status, const int *iters, int B, unsigned char *dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    double *rec = reinterpret_cast<double *>(dst + (size_t)16 * i);
    rec[0] = f[i];
    int *ri = reinterpret_cast<int *>(rec + 1);
    ri[0] = status[i];
    ri[1] = iters[i];
  }
}
hipError_t launch_pack(const DevBatch &D, void *dst, hipStream_t stream) {
  hipLaunchKernelGGL(pack_results_kernel, dim3((D.B + 255) / 256), dim3(256), 0, stream, D.f_out, D.status, D.iters, D.B,
                     static_cast<unsigned char *>(dst));
  return hipGetLastError();
}

// ------------------------------------------------------------- host launchers
hipError_t launch_solver(const DevBatch &D, int mode, int threads, hipStream_t stream) {
  size_t lds = solver_lds_bytes(D.L, D.P, threads);
  hipError_t e;
  if (D.sur.S > 0) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solver_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(solver_kernel<true>, dim3(D.B), dim3(threads), lds, stream, D, mode);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solver_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(solver_kernel<false>, dim3(D.B), dim3(threads), lds, stream, D, mode);
  }
  return hipGetLastError();
}

} // namespace dftpav
