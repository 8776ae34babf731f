// dense_dir.h -- the L-BFGS search direction d = -H g from a DENSE form of H instead of the two-loop recursion.
//
// EXPERIMENTAL (round 4, built and validated on the CPU only; the device path is off unless a batch asks for it through the
// debug hook dftpav_debug_set_direction): see DESIGN.md section 8.
//
// Why.  The reference keeps m = 256 pairs (s_j, y_j) (lbfgs_mem_size, pb.txt:96) for problems of n = 31 .. 63 variables, and
// its two-loop recursion (lbfgs.hpp:716-739) walks them one after the other: 2 * bound DEPENDENT steps per iteration, each a dot
// product over n <= 64 elements -- on the device a cross-lane reduction of 300 cycles that nothing else can hide (a third of
// the throughput kernel's time at 4096 trajectories, half of it for one trajectory alone).  With n << m the operator itself is
// small: n x n.  H_k = T_{k-1} o ... o T_{k-bound}(gamma I) with T_j(X) = V_j^T X V_j + rho_j s_j s_j^T, V_j = I - rho_j y_j s_j^T
// (Nocedal & Wright 7.19, the operator the two-loop recursion applies).  A composition of such maps is again of the form
// X -> A^T X A + C with A = V_old ... V_new (n x n) and C (n x n); composing with one more pair on the right is a rank-1 / rank-2
// update, O(n^2), and
//     H g = gamma A^T (A g) + C g
// is three matrix-vector products whose n results are INDEPENDENT: lane L owns row L, one chain of n multiply-adds, no
// cross-lane reduction at all.  The window (the oldest pair leaves when the 257th arrives) is kept with the two-stack queue
// of sliding-window aggregation: the newest pairs are folded into a running `back` aggregate; when the front runs dry the
// surviving pairs are turned, newest to oldest, into SUFFIX aggregates (one O(n^2) step each, m of them once every m
// iterations; kept as checkpoints + one block, see kBlock below), and dropping the oldest pair is stepping to the next suffix.  With front F = (A_f, C_f) and back B = (A_b, C_b):
//     H g = A_b^T ( gamma A_f^T (A_f (A_b g)) + C_f (A_b g) ) + C_b g.
// Numerically this is the BFGS update in product form: measured against the two-loop recursion in 80-bit arithmetic on real
// solves the direction differs by 1e-14 (median) .. 2e-10 (worst seen) relative, where the fp64 two-loop recursion itself is at
// 1e-15 .. 3e-13 (tests/test_dense_direction.py).  It is a different device order: same mathematics, its own bits, its own oracle
// mode (order 3).
//
// This header is the ARITHMETIC, shared by the kernel (one lane = one index L) and the device-order oracle (a loop over L):
// every value is one chain of fused multiply-adds from 0.0 in index order, so the two sides agree bit for bit by construction.
// Layout of one aggregate ("entry"), pitch np >= n:  Acm[k * np + i] = A[i][k],  Arm[i * np + k] = A[i][k],  Ccm[k * np + i]
// = C[i][k] -- in each, the lanes of a wave read consecutive doubles.  A is kept twice because both A v (lane = row) and
// A^T v (lane = column) are wanted; an element is computed by the same expression in both copies, so they hold the same bits.
#pragma once
#include "traj_math.h"

namespace dftpav {
namespace dense {

DFTPAV_HD inline int pitch(int n) { return (n + 7) & ~7; }
DFTPAV_HD inline size_t entry_doubles(int n) { return 3 * (size_t)pitch(n) * pitch(n); }

// the matrices live in HBM: on the device their pointers carry the global address space (global_load / global_store instead of
// flat accesses, which would wait on the LDS counter as well)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) double mat_t;
#else
typedef double mat_t;
#endif
struct Entry {
  mat_t *acm, *arm, *ccm;
  int np;
};
DFTPAV_HD inline Entry entry_at(double *base, int n, size_t index) {
  const int np = pitch(n);
  mat_t *p = (mat_t *)(base + index * entry_doubles(n));
  Entry e = {p, p + (size_t)np * np, p + 2 * (size_t)np * np, np};
  return e;
}

// sum_k M[k * np + L] * v[k]: lane L's chain, k ascending from 0.0.  The matrix lives in HBM / L2: its elements are requested
// eight at a time in front of the eight multiply-adds that consume them (the chain itself is the same, in the same order)
#ifndef DFTPAV_DENSE_MATVEC_BATCH
#define DFTPAV_DENSE_MATVEC_BATCH 8 // a tuning knob for scripts/build_variant.sh (the chains, hence the bits, do not depend on it)
#endif
constexpr int kMatvecBatch = DFTPAV_DENSE_MATVEC_BATCH;
DFTPAV_HD inline double lane_matvec(const mat_t *M, int np, int n, int L, const double *v) {
  double acc = 0.0;
  for (int k = 0; k < n; k += kMatvecBatch) {
    double mv[kMatvecBatch], vv[kMatvecBatch];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < kMatvecBatch; u++) {
      const int kk = k + u < n ? k + u : n - 1; // past the end: a valid address, the value is not used
      mv[u] = M[(size_t)kk * np + L];
      vv[u] = v[kk];
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < kMatvecBatch; u++)
      if (k + u < n) acc = fma_(mv[u], vv[u], acc);
  }
  return acc;
}

// A = I, C = 0 (lane L writes what it owns)
DFTPAV_HD inline void set_identity(const Entry &e, int n, int L) {
  for (int k = 0; k < n; k++) {
    e.acm[(size_t)k * e.np + L] = k == L ? 1.0 : 0.0;
    e.arm[(size_t)k * e.np + L] = k == L ? 1.0 : 0.0;
    e.ccm[(size_t)k * e.np + L] = 0.0;
  }
}

// ---- a new pair joins the back aggregate on the right: A <- A V, C <- V^T C V + rho s s^T, V = I - rho y s^T
//   a = A y, c = C y (lane_matvec on acm / ccm), yCy = y . c (the wave's sum, formed by the caller), beta = rho (rho yCy) + rho
//   A'[i][k] = A[i][k] - (rho a_i) s_k
//   C'[i][k] = C[i][k] - (rho s_i) c_k - (rho c_i) s_k + (beta s_i) s_k        (in this order)
// s, c, ra = rho * a: the whole vectors (LDS on the device); the _L values are lane L's own
#ifndef DFTPAV_DENSE_UPDATE_BATCH
#define DFTPAV_DENSE_UPDATE_BATCH 4 // likewise
#endif
constexpr int kUpdateBatch = DFTPAV_DENSE_UPDATE_BATCH;
DFTPAV_HD inline double push_beta(double rho, double yCy) { return fma_(rho, rho * yCy, rho); }
DFTPAV_HD inline void push_update(const Entry &e, int n, int L, const double *s, const double *c, const double *ra, double beta, double rho) {
  const double s_L = s[L], ra_L = ra[L];
  const double rs_L = rho * s_L, rc_L = rho * c[L], bs_L = beta * s_L;
  for (int k = 0; k < n; k += kUpdateBatch) { // the elements of kUpdateBatch steps are read before the first is written back
    double a1[kUpdateBatch], a2[kUpdateBatch], c1[kUpdateBatch];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < kUpdateBatch; u++) {
      const size_t at = (size_t)(k + u < n ? k + u : n - 1) * e.np + L;
      a1[u] = e.acm[at];
      a2[u] = e.arm[at];
      c1[u] = e.ccm[at];
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < kUpdateBatch; u++) {
      if (k + u < n) {
        const size_t at = (size_t)(k + u) * e.np + L;
        const double s_k = s[k + u];
        e.acm[at] = fma_(-ra_L, s_k, a1[u]);                                          // A[L][k]
        e.arm[at] = fma_(-ra[k + u], s_L, a2[u]);                                     // A[k][L]
        e.ccm[at] = fma_(bs_L, s_k, fma_(-rc_L, s_k, fma_(-rs_L, c[k + u], c1[u])));  // C[L][k]
      }
    }
  }
}

// ---- one step of the rebuild: the suffix aggregate that starts at pair j from the one that starts behind it,
//   (V_j, rho_j s_j s_j^T) o (A, C) = (V_j A, rho_j (A^T s_j)(A^T s_j)^T + C);   w = A^T s_j (lane_matvec on arm, or s itself
//   when the aggregate behind is the identity: in == nullptr)
//   C'[i][k] = C[i][k] + (rho w_i) w_k,      A'[i][k] = A[i][k] - (rho y_i) w_k
DFTPAV_HD inline double rebuild_w(const Entry *in, int n, int L, const double *s) { return in ? lane_matvec(in->arm, in->np, n, L, s) : s[L]; }
DFTPAV_HD inline void rebuild_step(const Entry *in, const Entry &out, int n, int L, const double *y, const double *w, double rho) {
  const double w_L = w[L], ry_L = rho * y[L], rw_L = rho * w_L;
  for (int k = 0; k < n; k += kUpdateBatch) {
    double a1[kUpdateBatch], a2[kUpdateBatch], c1[kUpdateBatch];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < kUpdateBatch; u++) {
      const int kk = k + u < n ? k + u : n - 1;
      const size_t at = (size_t)kk * out.np + L;
      a1[u] = in ? in->acm[at] : (kk == L ? 1.0 : 0.0);
      a2[u] = in ? in->arm[at] : (kk == L ? 1.0 : 0.0);
      c1[u] = in ? in->ccm[at] : 0.0;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < kUpdateBatch; u++) {
      if (k + u < n) {
        const size_t at = (size_t)(k + u) * out.np + L;
        const double w_k = w[k + u];
        out.ccm[at] = fma_(rw_L, w_k, c1[u]);                  // C[L][k]
        out.acm[at] = fma_(-ry_L, w_k, a1[u]);                 // A[L][k]
        out.arm[at] = fma_(-(rho * y[k + u]), w_L, a2[u]);     // A[k][L]
      }
    }
  }
}

// ---- the direction.  u = A_b g; with a front: v = A_f u, t = gamma A_f^T v + C_f u, else t = gamma u; d = -(A_b^T t + C_b g).
// Each stage needs the whole vector of the stage before (LDS on the device): the caller stores a stage's n results, then
// runs the next.  These are lane L's parts.
DFTPAV_HD inline double dir_u(const Entry &b, int n, int L, const double *g) { return lane_matvec(b.acm, b.np, n, L, g); }
DFTPAV_HD inline double dir_v(const Entry &f, int n, int L, const double *u) { return lane_matvec(f.acm, f.np, n, L, u); }
DFTPAV_HD inline double dir_t_front(const Entry &f, int n, int L, const double *u, const double *v, double gamma) {
  return fma_(gamma, lane_matvec(f.arm, f.np, n, L, v), lane_matvec(f.ccm, f.np, n, L, u));
}
DFTPAV_HD inline double dir_d(const Entry &b, int n, int L, const double *t, const double *g) {
  return -(lane_matvec(b.arm, b.np, n, L, t) + lane_matvec(b.ccm, b.np, n, L, g));
}

// ---- when the dense form is used.  Forming H explicitly costs accuracy where H is ill-conditioned: a pair with a small
// curvature y.s has a large |V| = |y| |s| / (y.s), and products of such factors lose what the two-loop recursion (which only
// ever applies them to one vector) keeps.  Measured over 3 000 random layouts, 587 k directions (scripts/fuzz_dense_cpu.py)
// against the recursion in 80-bit arithmetic: WITHOUT a gate 70 layouts exceed 1e-8 and five 1e-4 (worst 0.4: gear-shift
// layouts, a window shorter than n); neither the cancellation of the last product nor the secant equation of the newest pair
// tells such a direction from a good one, the largest |V| of the window does: below 1e3 the dense direction stays within
// 5e-10, below 1e4 within 8e-6 (where the fp64 recursion itself is at 2.5e-6 in the worst layout), and 3e-8 on every BASELINE
// configuration.  So the direction of an iteration comes from the dense form while  max_j |V_j| < kNuGate  over the window,
// and from the plain recursion otherwise (2.4 % of the fuzz's directions; none on BASELINE configs[0..4]; the aggregates are
// kept up to date either way).  Gate 3e3: three layouts above 1e-7 instead of seven, but 22 % of configs[1]'s iterations on
// the slow path; gate 1e3: two, 16 % of all.
constexpr double kNuGate = 1.0e4;
DFTPAV_HD inline double pair_nu(double ys, double yy, double ss) { return sqrt(yy * ss) / ys; }

// ---- the queue's bookkeeping (the same on both sides).  fpos: index of the oldest surviving pair among the m positions of the
// window as it stood at the last rebuild; m: no front (nothing rebuilt yet, or every suffix used up).
// A pair is accepted while the window is full (bound_before == m): returns true when the front must be rebuilt first.
DFTPAV_HD inline bool needs_rebuild(int fpos, int m) { return fpos >= m; }

// The m - 1 suffix aggregates of a rebuild are not all kept (24 KB each at n = 31): the rebuild stores every kBlock-th of them
// (checkpoints) and the first block; when the front pointer enters the next block, that block is formed again from its
// checkpoint by the very steps the rebuild took -- the same operations on the same operands, so the same bits -- at twice the
// rebuild's arithmetic and 1 / 8 of its memory (34 entries per trajectory at m = 256 instead of 256).
// Entries of one trajectory:  [0] back   [1 .. kBlock] suffixes of the current block (position p at 1 + p % kBlock)
//                             [1 + kBlock] the rebuild's running aggregate   [2 + kBlock + j] checkpoint = suffix of position (j + 1) kBlock
constexpr int kBlock = 16;
DFTPAV_HD inline int checkpoints(int m) { return (m - 1) / kBlock; } // positions kBlock, 2 kBlock, ... <= m - 1
DFTPAV_HD inline size_t entries_per_trajectory(int m) { return 2 + (size_t)kBlock + (size_t)checkpoints(m); }
DFTPAV_HD inline size_t idx_block(int p) { return 1 + (size_t)(p % kBlock); }
DFTPAV_HD inline size_t idx_running() { return 1 + (size_t)kBlock; }
DFTPAV_HD inline size_t idx_checkpoint(int p) { return 1 + (size_t)kBlock + (size_t)(p / kBlock); } // p a positive multiple of kBlock
// One rebuild step, position p of the window of size m: which entry it reads (-1: the identity) and which it writes.
//   whole pass (p = m - 1 .. kBlock): a running aggregate, copied out at the block boundaries;
//   a block (p = min((q + 1) kBlock, m) - 1 .. max(q kBlock, 1)): from the checkpoint at its end (or the identity) into the block's slots
DFTPAV_HD inline void pass_step_io(int p, int m, long long &in, size_t &out) {
  in = p + 1 >= m ? -1LL : ((p + 1) % kBlock == 0 ? (long long)idx_checkpoint(p + 1) : (long long)idx_running());
  out = p % kBlock == 0 ? idx_checkpoint(p) : idx_running();
}
DFTPAV_HD inline void block_step_io(int p, int m, long long &in, size_t &out) {
  in = p + 1 >= m ? -1LL : ((p + 1) % kBlock == 0 ? (long long)idx_checkpoint(p + 1) : (long long)idx_block(p + 1));
  out = idx_block(p);
}
DFTPAV_HD inline int block_first(int q) { return q * kBlock > 1 ? q * kBlock : 1; }
DFTPAV_HD inline int block_last(int q, int m) { return ((q + 1) * kBlock < m ? (q + 1) * kBlock : m) - 1; }

} // namespace dense
} // namespace dftpav
