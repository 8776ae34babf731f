// solver_ref4.hip — the reference order in the QUAD shape: FOUR trajectories per wave, one per row of 16 lanes (gfx950).
//
// solver_ref.hip's WAVE shape gives a trajectory a wave of its own.  Its sequential sums are chains of v_fmac_f64_dpp
// row_newbcast:K, and such an instruction costs the same whether one row of the wave wants its result or all four do; its band
// substitutions run on 2 of 64 lanes.  A wave there serves ONE trajectory per instruction, and the kernel is issue-bound
// (profiles/r05_*: 15.0 M vector instructions per solve, two-loop recursion at its issue rate).  Here a row of 16 lanes is a
// trajectory, so every chain instruction, every step of a substitution and every step of the line search serves four:
//
//   * lane l of a row OWNS PIECE l of its trajectory (one gear segment of at most 16 pieces): the piece's six rows of the band
//     system (both dimensions), its coefficients c, its gdC and its adjoint live in that lane's REGISTERS -- no LDS copy of
//     b / c / gdC / adj at all (they were 6.1 of the WAVE shape's 18.2 KB per trajectory);
//   * BandedSystem::solve / solveAdj (poly_traj_utils.hpp:805-852): the row sweeps of solver_ref.hip, piece by piece -- at step s
//     the lane that owns block s takes the six results of the previous block from its neighbour's registers (DPP row_shr / row_shl)
//     and computes its own six rows, both dimensions side by side; the row's multiply-subtract pairs in the reference's order
//     (16 pieces: the two dimensions of a piece on two lanes, six rows per step -- sweep4_split below);
//   * addPVAGradCost2CT (traj_optimizer.cpp:486-705): lane l walks the K + 1 constraint points of ITS piece in order (the running
//     s1 += step of :513 is a register), and what an active term adds to gdC goes straight into the lane's own gdC registers --
//     the order a gdC entry receives its additions in is (point, term) order within the piece, which is all the reference's order
//     says about it.  No 16-double records, no chain pass.  Only what the term adds to the segment's gdT and to the two costs
//     (three doubles) is parked, per piece, and chained afterwards over the pieces in order.  The active terms themselves are
//     EVALUATED densely packed: listed by the lanes that find them, taken 24 at a time one per lane, their results handed back to
//     the owners through LDS in list order (quad_common.h: DenseLds; q4_eval: flush);
//   * the corridor is read from a copy laid out [component][j][piece]: the 16 lanes of a row read 128 contiguous bytes;
//   * lbfgs_optimize / line_search_lewisoverton (lbfgs.hpp:276-390, 440-751): solver_ref.hip's lbfgs_advance, per row; a vector of
//     n <= 32 variables is two registers per lane (elements l and 16 + l), a sequential dot product is the 32-step DPP chain --
//     each row chains its own sixteen lanes, no permlane swap -- and the two-loop recursion (:716-739) runs the four rows' history
//     steps in the same instructions (a row that is still in its line search, or has the shorter history, is masked).
//
// Same bits as the TEAM / WAVE shapes (every sum is the same chain; tests/test_gpu_reference_order.py compares the shapes with one
// another, with the restatement and with the golden vectors).  Scope: one gear segment, N <= 16 pieces, n <= 32, no moving
// obstacles, H <= 5 -- the BASELINE configs[2] / [3] workload; everything else stays with solver_ref.hip.
//
// Residency: ONE wave per SIMD (the kernel takes 460 registers; built for two waves per SIMD it spilled 205 of them and its scratch
// traffic alone was HBM-sized), four waves = 16 trajectories per CU (8 in the WAVE shape), each wave at the speed of a wave that has
// its SIMD to itself; 5.7 KB of LDS per trajectory (x, g, the boundary states, the scalars of the line search, alpha[mem], the first
// seven parked terms of every piece) and 3.9 KB per wave for the list of active terms, workgroups of one wave with a copy of the sweep
// tables each (39 KB; a wave that has nothing
// left to do frees its slot at once).  Scheduling as the WAVE shape: the rows pop trajectories from the batch's ring, run them a
// slice of evaluations and push them back unfinished, so that a wave's rows stay filled until the batch runs out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#include "ref_order_common.h"
#include "quad_common.h"

namespace dftpav {
namespace reford {

constexpr int kQLcap = 7; // parked terms of a piece kept in LDS (the rest in global scratch)
// Waves per SIMD the kernel is built for.  At 2 (256 registers) it spills 205 of them and its scratch traffic alone is HBM-sized
// (measured: the point loop 3 x slower than at 1); at 1 the allocator takes 455 registers, nothing goes to scratch, and four waves
// per CU hold 16 trajectories -- twice the WAVE shape's 8 -- each of them at the speed of a wave that has its SIMD to itself.
#ifndef DFTPAV_Q4_WAVES_PER_EU
#define DFTPAV_Q4_WAVES_PER_EU 1
#endif
constexpr int kQ4WavesPerCU = 4 * DFTPAV_Q4_WAVES_PER_EU;

// LDS of one trajectory (a row)
struct Q4 {
  ldsd_t xs, gs;  // [32] the trial point, the gradient (the interface between the evaluation and the solver)
  ldsd_t bnd;     // [12] iniS [6], finS [6] as uploaded (clamped)
  ldsd_t tw;      // [6] 1 / t^k of the piece duration of the evaluation in progress
  ldsd_t st;      // [sNUM]
  ldsd_t alpha;   // [mem]
  ldsd_t tl;      // [16][kQLcap][3] parked terms of a piece: what they add to gdT, to the corridor cost, to the feasibility cost
  ldsi_t ist;     // [iNUM]
  ldsi_t tcnt;    // [16] parked terms of a piece
};
__host__ __device__ inline size_t q4_team_doubles(int mem) { return 32 + 32 + 12 + 6 + sNUM + (size_t)mem + 16 * kQLcap * 3; }
__host__ __device__ inline size_t q4_team_bytes(int mem) { return (q4_team_doubles(mem) * sizeof(double) + (iNUM + 16) * sizeof(int) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t q4_shared_bytes(int N) { return ((size_t)pk_segment_doubles(N) * sizeof(double) + 15) & ~(size_t)15; }
__device__ inline void q4_carve(Q4 &q, char *team, int mem) {
  ldsd_t p = (ldsd_t)reinterpret_cast<double *>(team);
  q.xs = p; p += 32;
  q.gs = p; p += 32;
  q.bnd = p; p += 12;
  q.tw = p; p += 6;
  q.st = p; p += sNUM;
  q.alpha = p; p += mem;
  q.tl = p; p += 16 * kQLcap * 3;
  ldsi_t i = (ldsi_t)p;
  q.ist = i; i += iNUM;
  q.tcnt = i;
}

// 0.0 + p[0] + p[1] + ... + p[n-1] for a vector held as (element l, element 16 + l); elements from n on contribute -0.0, and
// x + (-0.0) == x for every x: the second half of the chain is left out when it has nothing but those
__device__ __forceinline__ double row_sum32(double p0, double p1, int n, int l) {
  double acc = row_chain16(0.0, l < n ? p0 : -0.0);
  if (n > 16) acc = row_chain16(acc, 16 + l < n ? p1 : -0.0); // (uniform)
  return acc;
}
// One substitution sweep over the 6 N rows of the band system (solver_ref.hip: sweep / rows_end / rows_pack -- the same rows, the
// same order of a row's updates), the rows in registers: bq[2 r + d] = row 6 l + r, dimension d, of this lane's piece.  Step s of
// the traversal belongs to piece s (ascending sweeps) or N - 1 - s (descending); the six previous results it starts from are the
// finished rows of the previous step's piece, i.e. the neighbour lane's registers.  tab: this sweep's table (blocks in traversal
// order: whole rows at the two ends, the interior pattern's coefficients in between).
template <int Q>
__device__ __forceinline__ void sweep4(ldscd_t tab, double (&bq)[12], int N, int l) {
  constexpr bool DESC = Q == 1 || Q == 3, DIV = Q == 1 || Q == 2;
  ldscd_t ip = tab + 48;
  // the table of this lane's own block (an interior block: pk_size(Q) doubles), read once, in front of the traversal
  v2d_t c[pk_size(Q) / 2];
  {
    const int sl = DESC ? N - 1 - l : l; // the step this lane's piece is taken in
    const int bi = sl >= 1 && sl <= N - 2 ? sl - 1 : 0;
    const ldscv2_t a = (ldscv2_t)(ip + bi * pk_size(Q));
    if (N > 2) {
#pragma unroll
      for (int u = 0; u < pk_size(Q) / 2; u++) c[u] = a[u];
    } else {
#pragma unroll
      for (int u = 0; u < pk_size(Q) / 2; u++) c[u] = v2d_t{0.0, 0.0};
    }
  }
#pragma unroll 1
  for (int s = 0; s < N; s++) {
    const int p = DESC ? N - 1 - s : s;
    // traversal row r of a block is natural row r (ascending) or 5 - r (descending) of the piece
    double w[6][2];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int d = 0; d < 2; d++) {
        w[r][d] = DESC ? nb_dpp<0x101>(bq[2 * (5 - r) + d]) : nb_dpp<0x111>(bq[2 * r + d]);
      }
    if (s == 0 || s == N - 1) { // (uniform) a block of the ends: every coefficient is tested, as the reference does (`if (a != 0.0)`)
      const ldscv2_t a = (ldscv2_t)(s == 0 ? tab : ip + (N - 2) * pk_size(Q));
      if (s == 0) { // (uniform) the traversal starts from six zeros, whatever the neighbour holds
#pragma unroll
        for (int r = 0; r < 6; r++) w[r][0] = w[r][1] = 0.0;
      }
      if (l == p) {
#pragma unroll
        for (int r = 0; r < 6; r++) {
          v2d_t ce[4];
#pragma unroll
          for (int u = 0; u < 4; u++) ce[u] = a[4 * r + u];
          const int rr = DESC ? 5 - r : r;
#pragma unroll
          for (int d = 0; d < 2; d++) {
            double acc = bq[2 * rr + d];
#pragma unroll
            for (int k = 0; k < 6; k++) {
              const double ck = (k & 1) ? ce[k >> 1].y : ce[k >> 1].x;
              const double t = ck * w[(r + k) % 6][d];
              acc = ck != 0.0 ? acc - t : acc;
            }
            if (DIV) acc = div_by_rcp(acc, ce[3].x, ce[3].y);
            w[r][d] = acc;
            bq[2 * rr + d] = acc;
          }
        }
      }
    } else { // an interior block: the non-zero terms only, no test
      if (l == p) {
        auto at = [&](int o) { return (o & 1) ? c[o >> 1].y : c[o >> 1].x; };
#pragma unroll
        for (int r = 0; r < 6; r++) {
          const int rr = DESC ? 5 - r : r;
#pragma unroll
          for (int d = 0; d < 2; d++) {
            double acc = bq[2 * rr + d];
#pragma unroll
            for (int k = 0; k < 6; k++)
              if (pk_mask(Q, r) & (1 << k)) acc = acc - at(pk_off(Q, r, k)) * w[(r + k) % 6][d];
            if (DIV) acc = div_by_rcp(acc, at(pk_diag0(Q) + 2 * r), at(pk_diag0(Q) + 2 * r + 1));
            w[r][d] = acc;
            bq[2 * rr + d] = acc;
          }
        }
      }
    }
  }
}

// The same sweep for N = 16 with the two DIMENSIONS of a piece on two LANES (round 6).  In sweep4 the lane that owns block s computes its
// twelve rows -- six per dimension, two independent chains -- while fifteen lanes wait; here a lane holds two SETS of six rows:
//   X = (piece l, dimension 0) for l < 8, (piece l - 8, dimension 1) for l >= 8;   Y = (piece l + 8, dimension 1) for l < 8, (piece l, dimension 0) for l >= 8
// so that block p of both dimensions is in one set on the two lanes p mod 8 and p mod 8 + 8, and a step of the traversal is SIX rows of
// one instruction stream for both.  Pieces 0 .. 7 are in X, 8 .. 15 in Y: an ascending sweep runs eight steps on X, then eight on Y (a
// descending one Y, then X); the previous block's results are the neighbour lane's rows of the same set (lanes 7 / 8 and 15 / 0 are not
// neighbours in the chains they split: harmless, a chain's first block starts from zeros), except at the ninth step, which takes them from
// the other set one lane around the row (row_ror).  Per row the same products and differences in the same order: the same bits.
__device__ __forceinline__ void q4_split(const double (&bq)[12], double (&X)[6], double (&Y)[6], int l) {
  const bool lo = (l & 8) == 0;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const double own = bq[2 * r], far = nb_dpp<0x128>(bq[2 * r + 1]); // dimension 1 of piece (l + 8) mod 16
    X[r] = lo ? own : far;
    Y[r] = lo ? far : own;
  }
}
__device__ __forceinline__ void q4_unsplit(const double (&X)[6], const double (&Y)[6], double (&bq)[12], int l) {
  const bool lo = (l & 8) == 0;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    bq[2 * r] = lo ? X[r] : Y[r];
    bq[2 * r + 1] = nb_dpp<0x128>(lo ? Y[r] : X[r]); // this piece's dimension 1, from the lane eight around the row
  }
}
template <int Q>
__device__ __forceinline__ void sweep4_split(ldscd_t tab, double (&X)[6], double (&Y)[6], int l) {
  constexpr bool DESC = Q == 1 || Q == 3, DIV = Q == 1 || Q == 2;
  constexpr int N = 16;
  ldscd_t ip = tab + 48;
  const int l8 = l & 7;
  // eight steps of the traversal on one set: pieces p0, p0 + dp, ...; first: the rows the first step starts from
  auto phase = [&](double (&R)[6], int s0, const double (&first)[6]) {
    // the table of this lane's own block of the set (an interior block: pk_size(Q) doubles), read once
    v2d_t c[pk_size(Q) / 2];
    {
      const int sl = s0 + (DESC ? 7 - l8 : l8); // the step this lane's block of the set is taken in
      const int bi = sl >= 1 && sl <= N - 2 ? sl - 1 : 0;
      const ldscv2_t a = (ldscv2_t)(ip + bi * pk_size(Q));
#pragma unroll
      for (int u = 0; u < pk_size(Q) / 2; u++) c[u] = a[u];
    }
#pragma unroll 1
    for (int s = s0; s < s0 + 8; s++) {
      const int p8 = (DESC ? N - 1 - s : s) & 7;
      double w[6];
#pragma unroll
      for (int r = 0; r < 6; r++) w[r] = DESC ? nb_dpp<0x101>(R[5 - r]) : nb_dpp<0x111>(R[r]);
      if (s == s0) { // (uniform)
#pragma unroll
        for (int r = 0; r < 6; r++) w[r] = first[r];
      }
      if (l8 != p8) continue;
      if (s == 0 || s == N - 1) { // a block of the ends: every coefficient is tested, as the reference does (`if (a != 0.0)`)
        const ldscv2_t a = (ldscv2_t)(s == 0 ? tab : ip + (N - 2) * pk_size(Q));
#pragma unroll
        for (int r = 0; r < 6; r++) {
          v2d_t ce[4];
#pragma unroll
          for (int u = 0; u < 4; u++) ce[u] = a[4 * r + u];
          const int rr = DESC ? 5 - r : r;
          double acc = R[rr];
#pragma unroll
          for (int k = 0; k < 6; k++) {
            const double ck = (k & 1) ? ce[k >> 1].y : ce[k >> 1].x;
            const double t = ck * w[(r + k) % 6];
            acc = ck != 0.0 ? acc - t : acc;
          }
          if (DIV) acc = div_by_rcp(acc, ce[3].x, ce[3].y);
          w[r] = acc;
          R[rr] = acc;
        }
      } else { // an interior block: the non-zero terms only, no test
        auto at = [&](int o) { return (o & 1) ? c[o >> 1].y : c[o >> 1].x; };
#pragma unroll
        for (int r = 0; r < 6; r++) {
          const int rr = DESC ? 5 - r : r;
          double acc = R[rr];
#pragma unroll
          for (int k = 0; k < 6; k++)
            if (pk_mask(Q, r) & (1 << k)) acc = acc - at(pk_off(Q, r, k)) * w[(r + k) % 6];
          if (DIV) acc = div_by_rcp(acc, at(pk_diag0(Q) + 2 * r), at(pk_diag0(Q) + 2 * r + 1));
          w[r] = acc;
          R[rr] = acc;
        }
      }
    }
  };
  const double zeros[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double mid[6]; // what the ninth step starts from: the eighth block's results, from the other set one lane around the row
  if (!DESC) {
    phase(X, 0, zeros);
#pragma unroll
    for (int r = 0; r < 6; r++) mid[r] = nb_dpp<0x121>(X[r]); // lane 8 <- lane 7 (piece 7, dimension 0), lane 0 <- lane 15 (piece 7, dimension 1)
    phase(Y, 8, mid);
  } else {
    phase(Y, 0, zeros);
#pragma unroll
    for (int r = 0; r < 6; r++) mid[r] = nb_dpp<0x12F>(Y[5 - r]); // lane 7 <- lane 8 (piece 8, dimension 0), lane 15 <- lane 0 (piece 8, dimension 1)
    phase(X, 8, mid);
  }
}

// ------------------------------------------------ costFunctionCallback (traj_optimizer.cpp:206-350), one gear segment
// x (q.xs) -> g (q.gs), returns f.  The statements are solver_ref.hip's ref_eval, stage by stage; what changes is where a value
// lives.  cor: &cor_t[b][0][0][l] (component pitch cpitch = 16 (Kmax + 1), a round's 16 pieces contiguous); ovf: this
// trajectory's global scratch for the parked terms beyond the LDS window.
// FAST: the live path's constants known at compile time -- H = 4 half-planes per point (rectangles, traj_manager.cpp:1225) and
// help_eps = 0.0 (:610): the fifth plane slot and the second reciprocal of the curvature term drop out of the point loop.
template <bool FAST, bool DENSE>
__device__ __forceinline__ double q4_eval(const DevBatch &D, const Q4 &q, const DenseLds &dl, ldscd_t tab, gcd_t cor, size_t cpitch, gd_t ovf, int l, Prof &pr) {
  const DevLayout &L = D.L;
  const DevParams &P = D.P;
  constexpr bool SPLIT = DENSE; // (the round's two late changes share the switch DFTPAV_REF_QUAD_DENSE: the sweeps of 16 pieces with a dimension per lane)
  const int N = L.Ntot, H = FAST ? 4 : L.H, nterm = 5 * H + 4, t0 = 5 * H;
  const double epis = FAST ? 0.0 : D.epis;
  // ---- durations (VirtualT2RealT, :371-379), their powers (poly_traj_utils.hpp:961-966)
  const double vt = q.xs[L.x_tau0];
  const double Tr = vt > 0.0 ? ((0.5 * vt + 1.0) * vt + 1.0) + P.mini_T : 1.0 / ((0.5 * vt - 1.0) * vt + 1.0) + P.mini_T;
  const double t1 = Tr / N, t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
  pr.tick(10); // (what a pass of the kernel's loop does around the evaluation)
  // ---- right-hand sides (poly_traj_utils.hpp:968-977): the rows of this lane's piece
  double bq[12];
#pragma unroll
  for (int u = 0; u < 12; u++) bq[u] = 0.0;
  if (l == 0) {
#pragma unroll
    for (int d = 0; d < 2; d++) {
      bq[0 + d] = q.bnd[d];
      bq[2 + d] = q.bnd[2 + d] * t1;
      bq[4 + d] = q.bnd[4 + d] * t2;
    }
  }
  if (l == N - 1) {
#pragma unroll
    for (int d = 0; d < 2; d++) {
      bq[6 + d] = q.bnd[6 + d];
      bq[8 + d] = q.bnd[8 + d] * t1;
      bq[10 + d] = q.bnd[10 + d] * t2;
    }
  } else if (l < N - 1) {
    bq[10] = q.xs[2 * l];
    bq[11] = q.xs[2 * l + 1];
  }
  // ---- BandedSystem::solve (poly_traj_utils.hpp:805-826)
  if (N == 16 && SPLIT) { // (uniform)
    double X[6], Y[6];
    q4_split(bq, X, Y, l);
    sweep4_split<0>(tab, X, Y, l);
    sweep4_split<1>(tab + pk_sweep_offset(1, 16), X, Y, l);
    q4_unsplit(X, Y, bq, l);
  } else {
    sweep4<0>(tab, bq, N, l);
    sweep4<1>(tab + pk_sweep_offset(1, N), bq, N, l);
  }
  pr.tick(0);
  // ---- c = b * tInv (:979-984)
  double cc[12];
  {
    const double tI[6] = {1.0 / 1.0, 1.0 / t1, 1.0 / t2, 1.0 / t3, 1.0 / t4, 1.0 / t5};
#pragma unroll
    for (int u = 0; u < 12; u++) cc[u] = bq[u] * tI[u >> 1];
    if (l == 0) {
#pragma unroll
      for (int u = 0; u < 6; u++) q.tw[u] = tI[u]; // kept for calGrads_PT (six divisions, not six registers across the point loop)
    }
  }
  // ---- initSmGradCost / getTrajJerkCost per piece (poly_traj_utils.hpp:998-1035)
  double gdC[12], pE, pG;
  {
    const double *c = cc;
    const double t[6] = {1.0, t1, t2, t3, t4, t5};
    const double n33 = c[6] * c[6] + c[7] * c[7], n44 = c[8] * c[8] + c[9] * c[9], n55 = c[10] * c[10] + c[11] * c[11];
    const double d43 = c[8] * c[6] + c[9] * c[7], d53 = c[10] * c[6] + c[11] * c[7], d54 = c[10] * c[8] + c[11] * c[9];
    pE = 36.0 * n33 * t[1] + 144.0 * d43 * t[2] + 192.0 * n44 * t[3] + 240.0 * d53 * t[3] + 720.0 * d54 * t[4] + 720.0 * n55 * t[5];
    pG = 36.0 * n33 + 288.0 * d43 * t[1] + 576.0 * n44 * t[2] + 720.0 * d53 * t[2] + 2880.0 * d54 * t[3] + 3600.0 * n55 * t[4];
#pragma unroll
    for (int d = 0; d < 2; d++) {
      const double c3 = c[6 + d], c4 = c[8 + d], c5 = c[10 + d];
      gdC[10 + d] = 240.0 * c3 * t[3] + 720.0 * c4 * t[4] + 1440.0 * c5 * t[5];
      gdC[8 + d] = 144.0 * c3 * t[2] + 384.0 * c4 * t[3] + 720.0 * c5 * t[4];
      gdC[6 + d] = 72.0 * c3 * t[1] + 144.0 * c4 * t[2] + 240.0 * c5 * t[3];
      gdC[d] = 0.0;
      gdC[2 + d] = 0.0;
      gdC[4 + d] = 0.0;
    }
  }
  pr.tick(1);
  // ---- the constraint points of this lane's piece, in order (traj_optimizer.cpp:486-705)
  const bool piece = l < N;
  const int Kl = (l == 0 || l == N - 1) ? L.Kd : L.K;
  const int pt0 = l == 0 ? 0 : (L.Kd + 1) + (l - 1) * (L.K + 1); // the piece's first constraint point
  const double step = t1 / Kl;
  const int singul_ = L.singuls[0];
  double s1 = 0.0;
  int cnt = 0;
  const gd_t ovf_l = ovf + (size_t)pt0 * nterm * 3;
  // a round's half-planes are requested one round ahead (the copy holds zeros where a piece has no such point: every lane
  // loads, whatever its piece): 33 rounds of a dependent HBM round trip each were 3/4 of this kernel's time
  double pl[20];
  load_planes(cor, cpitch, H, pl);
  // (the first round's half-planes are waited for HERE: left pending into the loop, they make the compiler wait for ALL vector loads in front of
  // every round's first use of a half-plane -- its count of the loads in flight merges to zero at the loop's head -- and that includes the
  // next round's, requested a few hundred instructions earlier: the request ahead bought half a round instead of a whole one)
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
  // what an active term adds: to the lane's gdC in place, to gdT and to its cost parked (the other cost gets -0.0: x + (-0.0) == x)
  auto take = [&](const double (&r12)[12], double e0, double e1, double e2) {
#pragma unroll
    for (int u = 0; u < 12; u++) gdC[u] += r12[u];
    if (cnt < kQLcap) {
      ldsd_t e = q.tl + (l * kQLcap + cnt) * 3;
      e[0] = e0;
      e[1] = e1;
      e[2] = e2;
    } else {
      gd_t e = ovf_l + (size_t)cnt * 3;
      e[0] = e0;
      e[1] = e1;
      e[2] = e2;
    }
    cnt++;
  };
  // DENSE: the list of the wave
  const int lane64 = (int)(threadIdx.x & 63);
  int listed = 0;       // (uniform) entries in the list
  unsigned mine = 0u;   // the list's slots that hold terms of this lane's piece
  // (the rows of a wave that have no trajectory are not here: EXEC holds whole rows of 16 lanes, any of the four)
  const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
  const int row_here[4] = {(int)(here & 1ull), (int)((here >> 16) & 1ull), (int)((here >> 32) & 1ull), (int)((here >> 48) & 1ull)};
  const int my_row = lane64 >> 4;
  const int rank = ((my_row > 0 ? row_here[0] : 0) + (my_row > 1 ? row_here[1] : 0) + (my_row > 2 ? row_here[2] : 0)) * 16 + l; // among the lanes here
  const int n_here = 16 * (row_here[0] + row_here[1] + row_here[2] + row_here[3]);
  // inclusive prefix sum of c over the lanes that are here, and its total (wave_incl_scan_i32 wants all 64 lanes)
  auto scan_here = [&](int c, int &total) {
    int v = c;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    const int t0_ = row_here[0] ? __builtin_amdgcn_readlane(v, 15) : 0, t1_ = row_here[1] ? __builtin_amdgcn_readlane(v, 31) : 0;
    const int t2_ = row_here[2] ? __builtin_amdgcn_readlane(v, 47) : 0, t3_ = row_here[3] ? __builtin_amdgcn_readlane(v, 63) : 0;
    total = t0_ + t1_ + t2_ + t3_;
    return v + (my_row > 0 ? t0_ : 0) + (my_row > 1 ? t1_ : 0) + (my_row > 2 ? t2_ : 0);
  };
  auto flush = [&]() {
    for (int first = 0; first < listed; first += n_here) { // (uniform; one pass unless a single row is here and the list is longer than 16)
    const int slot = first + rank;
    const bool work = slot < listed;
    const int id = dl.id[work ? slot : 0];
    const int o = id & 63, t = (id >> 6) & 63, jo = id >> 12;
    // the owner's piece: its coefficients and the piece duration of its row come over from the owner's registers
    double cco[12];
#pragma unroll
    for (int u = 0; u < 12; u++) cco[u] = __shfl(cc[u], o);
    const double t1o = __shfl(t1, o);
    const int lo = o & 15;
    const int Ko = (lo == 0 || lo == N - 1) ? L.Kd : L.K;
    const double stepo = t1o / Ko; // (the owner's expression: the same bits)
    if (work) {
      const double nopl[20] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      PtState ps;
      // the state of the owner's point by the expressions that found the term (its tests are not needed again: no half-planes)
      (void)point_masks<false>(P, cco, lo, N, jo, Ko, stepo, dl.s1[slot], singul_, epis, H, nopl, (gd_t) nullptr, D.sur, 0.0, 0.0, 0, 0.0, ps);
      double r_[16];
      const ldscd_t pk = dl.pl + 4 * slot;
      point_emit_pf(P, ps, t, H, t0, [&](int, double &on0, double &on1, double &q0, double &q1) { on0 = pk[0]; on1 = pk[1]; q0 = pk[2]; q1 = pk[3]; },
                    (double *)r_);
      const bool corr = t < t0;
      ldsd_t w = dl.out + 15 * slot;
#pragma unroll
      for (int u = 0; u < 13; u++) w[u] = r_[u];
      w[13] = corr ? r_[13] : -0.0;
      w[14] = corr ? -0.0 : r_[13];
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (unsigned mm = mine; mm;) { // the owner adds its terms in list order: (point, term) order
      const int i = __builtin_ctz(mm);
      mm &= mm - 1;
      ldscd_t w = dl.out + 15 * i;
      double r12[12];
#pragma unroll
      for (int u = 0; u < 12; u++) r12[u] = w[u];
      take(r12, w[12], w[13], w[14]);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // the list is written again below
    listed = 0;
    mine = 0u;
  };
  // (DENSE: one more round than the pieces have points -- it only empties the list: the emission's code exists once)
#pragma unroll 1
  for (int j = 0; j <= L.Kmax + (DENSE ? 1 : 0); j++) {
    const bool extra = j > L.Kmax; // (uniform)
    unsigned m = 0u;
    PtState pst;
    double nx[20];
    load_planes(cor + (size_t)(j < L.Kmax ? j + 1 : L.Kmax) * 16, cpitch, H, nx);
    if (piece && j <= Kl && !extra)
      m = (unsigned)point_masks<false>(P, cc, l, N, j, Kl, step, s1, singul_, epis, H, pl, (gd_t) nullptr, D.sur, 0.0, 0.0, 0, 0.0, pst);
    const double s1_pt = s1;
    s1 += step; // the running sum of traj_optimizer.cpp:513
    [[maybe_unused]] long long emit_t0 = 0;
    if (D.prof != nullptr) { // (profiling only) trips of the loop below for the wave: the longest list of active terms among its 64 points;
      // DFTPAV_PROF_EMIT_CYCLES (a build flag): the cycles of that loop instead -- the clock behind it is read when the wave has reconverged
#ifdef DFTPAV_PROF_EMIT_CYCLES
      emit_t0 = clock64();
#else
      const int pc = __builtin_popcount(m);
      int trips = 0;
      while (__builtin_amdgcn_ballot_w64(pc > trips) != 0ull) trips++;
      pr.count(11, trips);
#endif
    }
    auto plane_of = [&](int k, double &on0, double &on1, double &q0, double &q1) { // the planes point_masks tested, still in registers
      on0 = pl[0]; on1 = pl[1]; q0 = pl[2]; q1 = pl[3];
#pragma unroll
      for (int u = 1; u < 5; u++) {
        on0 = k == u ? pl[4 * u] : on0;
        on1 = k == u ? pl[4 * u + 1] : on1;
        q0 = k == u ? pl[4 * u + 2] : q0;
        q1 = k == u ? pl[4 * u + 3] : q1;
      }
    };
    bool in_place = !DENSE;
    if (DENSE) {
      const int c = __builtin_popcount(m);
      const bool any = __builtin_amdgcn_ballot_w64(c != 0) != 0ull; // (uniform) some point of this round has active terms
      int incl = 0, total = 0;
      if (any) incl = scan_here(c, total);
      // the list is emptied when this round's terms do not fit behind what it holds, and by the extra round
      if (listed > 0 && (extra || listed + total > kDense)) flush(); // (uniform)
      if (any && total > kDense) { // (uniform, rare) more than a list's worth in one round: in place, behind the earlier points' terms
        in_place = true;
      } else if (any) {
        int e = listed + incl - c;
        for (unsigned mm = m; mm;) {
          const int t = __builtin_ctz(mm);
          mm &= mm - 1;
          dl.id[e] = lane64 | (t << 6) | (j << 12);
          dl.s1[e] = s1_pt;
          if (t < t0) { // a corridor term: vertex v against half-plane k = t - v H
            int v = 0;
#pragma unroll
            for (int qv = 1; qv < 5; qv++) v += t >= qv * H ? 1 : 0;
            double on0, on1, q0, q1;
            plane_of(t - v * H, on0, on1, q0, q1);
            ldsd_t w = dl.pl + 4 * e;
            w[0] = on0;
            w[1] = on1;
            w[2] = q0;
            w[3] = q1;
          }
          mine |= 1u << e;
          e++;
        }
        listed += total;
      }
    }
    if (in_place) {
      for (unsigned mm = m; mm;) {
        const int t = __builtin_ctz(mm);
        mm &= mm - 1;
        double r_[16];
        point_emit_pf(P, pst, t, H, t0, plane_of, (double *)r_);
        const bool corr = t < t0;
        double r12[12];
#pragma unroll
        for (int u = 0; u < 12; u++) r12[u] = r_[u];
        take(r12, r_[12], corr ? r_[13] : -0.0, corr ? -0.0 : r_[13]);
      }
    }
#ifdef DFTPAV_PROF_EMIT_CYCLES
    if (D.prof != nullptr) pr.count(11, clock64() - emit_t0);
#endif
#pragma unroll
    for (int u = 0; u < 20; u++) pl[u] = nx[u];
  }
  q.tcnt[l] = piece ? cnt : 0;
  __threadfence_block(); // parked terms beyond the LDS window went to global memory; the counts are read by the other lanes
  pr.tick(2);
  // ---- the per-segment chains: `gdT +=`, `energy +=` over the pieces in order from 0.0, then the parked terms in (piece, point,
  // term) order (every lane of the row forms them for itself: the same bits)
  double gdT = row_chain16(0.0, piece ? pG : -0.0);
  const double en = row_chain16(0.0, piece ? pE : -0.0);
  double cost0 = 0.0, cost2 = 0.0;
  // (round 6, measured and taken out: the first three terms of a piece kept in the lane's registers and chained by DPP when no piece of the
  // row has more -- on this workload a piece that has terms has more than three)
  {
    int total = 0;
    for (int p = 0; p < N; p++) total += q.tcnt[p];
    pr.count(9, total);
    if (total > 0) {
      for (int p = 0; p < N; p++) {
        const int c = q.tcnt[p];
        if (__builtin_amdgcn_ballot_w64(c != 0) == 0ull) continue; // (uniform: no row of this wave has a term in this piece)
        const int pp0 = p == 0 ? 0 : (L.Kd + 1) + (p - 1) * (L.K + 1);
        const gcd_t og = (gcd_t)(ovf + (size_t)pp0 * nterm * 3);
        // the piece's LDS window in one go (the eight entries' reads go out together; what lies beyond the piece's count adds -0.0),
        // the four rows of the wave in step: a loop over the count would run each row's trips one after the other
        {
          double e[kQLcap][3];
#pragma unroll
          for (int i = 0; i < kQLcap; i++)
#pragma unroll
            for (int w = 0; w < 3; w++) e[i][w] = q.tl[(p * kQLcap + i) * 3 + w];
#pragma unroll
          for (int i = 0; i < kQLcap; i++) {
            gdT += i < c ? e[i][0] : -0.0;
            cost0 += i < c ? e[i][1] : -0.0;
            cost2 += i < c ? e[i][2] : -0.0;
          }
        }
        for (int i0 = kQLcap; i0 < c; i0 += 8) { // beyond the LDS window: eight terms' loads in flight, then their additions in order
          double e[8][3];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int i = i0 + u < c ? i0 + u : c - 1;
#pragma unroll
            for (int w = 0; w < 3; w++) e[u][w] = og[(size_t)i * 3 + w];
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            if (i0 + u < c) {
              gdT += e[u][0];
              cost0 += e[u][1];
              cost2 += e[u][2];
            }
          }
        }
      }
    }
  }
  pr.tick(3);
  // ---- calGrads_PT (poly_traj_utils.hpp:1037-1066): adj = gdC * tInv, solveAdj, the duration gradient
  double pA;
  double tI[6];
#pragma unroll
  for (int u = 0; u < 6; u++) tI[u] = q.tw[u];
  {
    const double gdtInv[6] = {0.0, -1.0 * tI[2], -2.0 * tI[3], -3.0 * tI[4], -4.0 * tI[5], -5.0 * tI[5] * tI[1]};
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double gdcol = gdC[2 * k] * bq[2 * k] + gdC[2 * k + 1] * bq[2 * k + 1];
      acc += gdtInv[k] * gdcol;
    }
    pA = acc;
  }
  double adj[12];
#pragma unroll
  for (int u = 0; u < 12; u++) adj[u] = gdC[u] * tI[u >> 1];
  if (N == 16 && SPLIT) { // (uniform)
    double X[6], Y[6];
    q4_split(adj, X, Y, l);
    sweep4_split<2>(tab + pk_sweep_offset(2, 16), X, Y, l);
    sweep4_split<3>(tab + pk_sweep_offset(3, 16), X, Y, l);
    q4_unsplit(X, Y, adj, l);
  } else {
    sweep4<2>(tab + pk_sweep_offset(2, N), adj, N, l);
    sweep4<3>(tab + pk_sweep_offset(3, N), adj, N, l);
  }
  pr.tick(4);
  // ---- gradient and cost (traj_optimizer.cpp:299-344)
  if (l < N - 1) { // gdP: rows 6 i + 5 of the adjoint
    q.gs[2 * l] = adj[10];
    q.gs[2 * l + 1] = adj[11];
  }
  {
    // the duration gradient (poly_traj_utils.hpp:1050-1064, VirtualTGradCost :405-419): the head terms live in lane 0, the tail
    // terms in lane N - 1 (every other lane hands the chain a -0.0)
    double hv[6], tv[6];
#pragma unroll
    for (int u = 0; u < 6; u++) {
      hv[u] = q.bnd[u];
      tv[u] = q.bnd[6 + u];
    }
    const double h1 = hv[2] * adj[2] + hv[3] * adj[3];
    const double h2 = (hv[4] * adj[4] + hv[5] * adj[5]) * 2.0 * t1;
    const double g1 = tv[2] * adj[8] + tv[3] * adj[9];
    const double g2 = (tv[4] * adj[10] + tv[5] * adj[11]) * 2.0 * t1;
    gdT = row_add_lane0(gdT, h1);
    gdT = row_add_lane0(gdT, h2);
    gdT = row_chain16(gdT, l == N - 1 ? g1 : -0.0);
    gdT = row_chain16(gdT, l == N - 1 ? g2 : -0.0);
    gdT = row_chain16(gdT, piece ? pA : -0.0);
    double gdVT2Rt;
    if (vt > 0) {
      gdVT2Rt = vt + 1.0;
    } else {
      const double denSqrt = (0.5 * vt - 1.0) * vt + 1.0;
      gdVT2Rt = (1.0 - vt) / (denSqrt * denSqrt);
    }
    if (l == 0) q.gs[L.x_tau0] = (gdT / N + P.wei_time) * gdVT2Rt;
  }
  double total_smcost = 0.0, total_timecost = 0.0, penalty_cost = 0.0;
  total_smcost += en;
  penalty_cost += (cost0 + 0.0) + cost2; // (the moving-obstacle cost of a segment without obstacles is its start value 0.0)
  total_timecost += Tr * P.wei_time;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // g is read by other lanes of the row
  pr.tick(5);
  return total_smcost + total_timecost + penalty_cost;
}

// ------------------------------------------------ L-BFGS, per row
struct QVec { // the solver's vectors that live in registers: element l and element 16 + l
  double xp0, xp1, gp0, gp1, d0, d1;
};

// Start of an outer iteration (lbfgs.hpp:559-574, 290-315): xp = x, gp = g, dginit = gp . d, first trial point
__device__ __forceinline__ bool q4_begin_iteration(const DevParams &P, const Q4 &q, QVec &v, int n, int l) {
  const bool e0 = l < n, e1 = 16 + l < n;
  v.xp0 = e0 ? q.xs[l] : 0.0;
  v.xp1 = e1 ? q.xs[16 + l] : 0.0;
  v.gp0 = e0 ? q.gs[l] : 0.0;
  v.gp1 = e1 ? q.gs[16 + l] : 0.0;
  const double dginit = row_sum32(v.gp0 * v.d0, v.gp1 * v.d1, n, l);
  const double step = q.st[sSTEP];
  if (!(step > 0.0)) {
    if (l == 0) q.ist[iRET] = -1006;
    return false;
  }
  if (0.0 < dginit) {
    if (l == 0) q.ist[iRET] = -1005;
    return false;
  }
  if (l == 0) {
    q.st[sFINIT] = q.st[sFX];
    q.st[sDGINIT] = dginit;
    q.st[sDGTEST] = P.f_dec_coeff * dginit;
    q.st[sDSTEST] = P.s_curv_coeff * dginit;
    q.st[sMU] = 0.0;
    q.st[sNU] = P.max_step;
    q.st[sSTP] = step;
    q.ist[iCOUNT] = 0;
    q.ist[iBRACKT] = 0;
    q.ist[iTOUCHED] = 0;
  }
  if (e0) q.xs[l] = v.xp0 + step * v.d0;
  if (e1) q.xs[16 + l] = v.xp1 + step * v.d1;
  return true;
}

// The two-loop recursion (lbfgs.hpp:716-739) over `bound` stored pairs of this row's trajectory, the newest in slot ne - 1;
// d = -g on entry (elements from n on: 0.0).  The four rows run their steps in the same instructions; bound, the ring position
// and the division mode belong to the row.  History rows: (s, y) interleaved per element at pitch npad; (ys, 1 / ys) per pair.
constexpr int kQB = 8; // stored pairs per register block (the next block is in flight while one is worked on)
struct QBlk {
  d2_t a[kQB], b[kQB]; // (s, y) of elements l and 16 + l
  d2_t yr[kQB];        // (ys, 1 / ys)
};
typedef double __attribute__((ext_vector_type(2))) qd2_t;
template <int DIR>
__device__ __forceinline__ void q4_load_blk(QBlk &R, gcd2_t hS, gcd2_t hR, int npad, int m, int la, int lb, int &jl) {
#pragma unroll
  for (int u = 0; u < kQB; u++) {
    const gcd2_t row = hS + (size_t)jl * npad + la; // (element 16 + l sits 16 entries on: rows are npad >= 32 long, what lies beyond n is never used)
    R.a[u] = row[0];
    R.b[u] = row[16];
    R.yr[u] = hR[jl];
    if (DIR < 0) jl = jl == 0 ? m - 1 : jl - 1;
    else jl = jl == m - 1 ? 0 : jl + 1;
  }
}
__device__ __forceinline__ void q4_pin_blk(QBlk &R) {
#pragma unroll
  for (int u = 0; u < kQB; u++)
    asm volatile("" : "+v"(R.a[u].x), "+v"(R.a[u].y), "+v"(R.b[u].x), "+v"(R.b[u].y), "+v"(R.yr[u].x), "+v"(R.yr[u].y));
}
template <bool EXACT>
__device__ __forceinline__ void q4_first_steps(const QBlk &R, const Q4 &q, int i0, int bound, int m, int n, int l, bool exact, int &j, double &d0, double &d1) {
#pragma unroll
  for (int u = 0; u < kQB; u++) {
    if (i0 + u < bound) { // (row-uniform)
      j = j == 0 ? m - 1 : j - 1;
      const double dot = row_sum32(R.a[u].x * d0, R.b[u].x * d1, n, l);
      // lm_alpha[j] = lm_s.col(j).dot(d) / lm_ys[j]  (EXACT: some row of the wave divides -- only its lanes take the division's result)
      const double a = EXACT && exact ? dot / R.yr[u].x : div_by_rcp<false>(dot, R.yr[u].x, R.yr[u].y);
      if (l == 0) q.alpha[j] = a;
      const double na = -a;
      d0 = d0 + na * R.a[u].y; // d += (-alpha) * lm_y.col(j)
      d1 = d1 + na * R.b[u].y;
    }
  }
}
template <bool EXACT>
__device__ __forceinline__ void q4_second_steps(const QBlk &R, const Q4 &q, int i0, int bound, int m, int n, int l, bool exact, int &j, double &d0, double &d1) {
#pragma unroll
  for (int u = 0; u < kQB; u++) {
    if (i0 + u < bound) { // (row-uniform)
      const double al = q.alpha[j];
      const double dot = row_sum32(R.a[u].y * d0, R.b[u].y * d1, n, l);
      const double beta = EXACT && exact ? dot / R.yr[u].x : div_by_rcp<false>(dot, R.yr[u].x, R.yr[u].y);
      const double cf = al - beta;
      d0 = d0 + cf * R.a[u].x; // d += (alpha - beta) * lm_s.col(j)
      d1 = d1 + cf * R.b[u].x;
      j = j == m - 1 ? 0 : j + 1;
    }
  }
}
template <bool EXACT>
__device__ __forceinline__ void q4_two_loop(const Q4 &q, gcd2_t hS, gcd2_t hR, int npad, int m, int n, int l, int bound, int ne, bool exact, double sc0, double &d0,
                                            double &d1) {
  const int la = l, lb = 16 + l;
  QBlk A, B;
  int j = ne;
  int jl = ne == 0 ? m - 1 : ne - 1;
  q4_load_blk<-1>(A, hS, hR, npad, m, la, lb, jl);
#pragma unroll 1
  for (int i0 = 0; i0 < bound; i0 += 2 * kQB) {
    q4_pin_blk(A);
    q4_load_blk<-1>(B, hS, hR, npad, m, la, lb, jl);
    q4_first_steps<EXACT>(A, q, i0, bound, m, n, l, exact, j, d0, d1);
    q4_pin_blk(B);
    q4_load_blk<-1>(A, hS, hR, npad, m, la, lb, jl);
    q4_first_steps<EXACT>(B, q, i0 + kQB, bound, m, n, l, exact, j, d0, d1);
  }
  d0 = d0 * sc0;
  d1 = d1 * sc0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // alpha written by lane 0 of the row, read by all below
  jl = j;
  q4_load_blk<+1>(A, hS, hR, npad, m, la, lb, jl);
#pragma unroll 1
  for (int i0 = 0; i0 < bound; i0 += 2 * kQB) {
    q4_pin_blk(A);
    q4_load_blk<+1>(B, hS, hR, npad, m, la, lb, jl);
    q4_second_steps<EXACT>(A, q, i0, bound, m, n, l, exact, j, d0, d1);
    q4_pin_blk(B);
    q4_load_blk<+1>(A, hS, hR, npad, m, la, lb, jl);
    q4_second_steps<EXACT>(B, q, i0 + kQB, bound, m, n, l, exact, j, d0, d1);
  }
}

// Everything lbfgs_optimize does between two evaluations (lbfgs.hpp:524-745 with the line search of :312-389 unrolled into it):
// solver_ref.hip's lbfgs_advance for the trajectory of this row; f: the cost of the evaluation just made; sets iACTION.
__device__ __forceinline__ void q4_advance(const DevBatch &D, const Q4 &q, QVec &v, double f, gd_t hS, gd_t hR, int l, Prof &pr) {
  const DevParams &P = D.P;
  const int n = D.L.n, m = P.mem_size, npad = D.L.npad;
  const bool e0 = l < n, e1 = 16 + l < n;
  int action = kActEval;
  if (q.ist[iPHASE] == 0) { // after the first evaluation: lbfgs.hpp:524-551
    const double g0 = e0 ? q.gs[l] : 0.0, g1 = e1 ? q.gs[16 + l] : 0.0;
    const double x0 = e0 ? q.xs[l] : 0.0, x1 = e1 ? q.xs[16 + l] : 0.0;
    v.d0 = e0 ? -g0 : 0.0;
    v.d1 = e1 ? -g1 : 0.0;
    const double gmax = row_max16(fmax(fabs(g0), fabs(g1))), xmax = row_max16(fmax(fabs(x0), fabs(x1)));
    const double dd = row_sum32((-g0) * (-g0), (-g1) * (-g1), n, l);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (l == 0) {
      q.st[sFX] = f;
      q.st[sPF0] = f;
      q.ist[iEVALS] = 1;
      q.ist[iEND] = 0;
      q.ist[iBOUND] = 0;
      q.ist[iHISTLO] = 0;
      q.ist[iHISTHI] = 0;
      q.ist[iPHASE] = 1;
    }
    if (gmax / fmax(1.0, xmax) < P.g_epsilon) {
      if (l == 0) {
        q.ist[iRET] = 0;
        q.ist[iK] = 0;
      }
      action = kActDone;
    } else {
      if (l == 0) {
        q.st[sSTEP] = 1.0 / sqrt(dd);
        q.ist[iK] = 1;
      }
      __threadfence_block();
      if (!q4_begin_iteration(P, q, v, n, l)) action = kActDone;
    }
    if (l == 0) q.ist[iACTION] = action;
    return;
  }

  // ---- after a line-search trial: lbfgs.hpp:317-389
  const double fx = f;
  const double finit = q.st[sFINIT];
  double stp = q.st[sSTP];
  const int count = q.ist[iCOUNT] + 1;
  int ls = 0;
  bool decided = false;
  const int evals_before = q.ist[iEVALS];
  __threadfence_block();
  if (l == 0) {
    q.st[sFX] = fx;
    q.ist[iEVALS] = evals_before + 1;
    q.ist[iCOUNT] = count;
  }
  if (isinf(fx) || isnan(fx)) {
    ls = -1012;
    decided = true;
  } else if (P.past > 0 && fabs(finit - fx) / (fabs(finit) + 1.0) < P.delta / P.past) { // lbfgs.hpp:326-329
    ls = count;
    decided = true;
  } else {
    double mu = q.st[sMU], nu = q.st[sNU];
    bool brackt = q.ist[iBRACKT] != 0;
    const int touched = q.ist[iTOUCHED];
    if (fx > finit + stp * q.st[sDGTEST]) {
      nu = stp;
      brackt = true;
    } else {
      const double g0 = e0 ? q.gs[l] : 0.0, g1 = e1 ? q.gs[16 + l] : 0.0;
      const double gs = row_sum32(g0 * v.d0, g1 * v.d1, n, l);
      if (gs < q.st[sDSTEST]) {
        mu = stp;
      } else {
        ls = count;
        decided = true;
      }
    }
    bool touch_now = false;
    if (!decided) {
      if (P.max_linesearch <= count) {
        ls = -1009;
        decided = true;
      } else if (brackt && (nu - mu) < P.machine_prec * nu) {
        ls = -1007;
        decided = true;
      } else {
        if (brackt) stp = 0.5 * (mu + nu);
        else stp *= 2.0;
        if (stp < P.min_step) {
          ls = -1011;
          decided = true;
        } else if (stp > P.max_step) {
          if (touched) {
            ls = -1010;
            decided = true;
          } else {
            touch_now = true;
            stp = P.max_step;
          }
        }
      }
    }
    __threadfence_block();
    if (l == 0) {
      q.st[sMU] = mu;
      q.st[sNU] = nu;
      q.ist[iBRACKT] = brackt ? 1 : 0;
      q.st[sSTP] = stp;
      if (touch_now) q.ist[iTOUCHED] = 1;
    }
    if (!decided) {
      if (e0) q.xs[l] = v.xp0 + stp * v.d0;
      if (e1) q.xs[16 + l] = v.xp1 + stp * v.d1;
      if (l == 0) q.ist[iACTION] = kActEval;
      pr.tick(6);
      return;
    }
  }
  if (l == 0) q.st[sSTEP] = stp; // lbfgs.hpp:574 passes `step` by reference
  if (ls < 0) { // lbfgs.hpp:604-611: x, g reverted; fx is not
    if (e0) {
      q.xs[l] = v.xp0;
      q.gs[l] = v.gp0;
    }
    if (e1) {
      q.xs[16 + l] = v.xp1;
      q.gs[16 + l] = v.gp1;
    }
    if (l == 0) {
      q.ist[iRET] = ls;
      q.ist[iACTION] = kActDone;
    }
    return;
  }

  // ---- convergence / stopping tests (lbfgs.hpp:628-666)
  const double x0 = e0 ? q.xs[l] : 0.0, x1 = e1 ? q.xs[16 + l] : 0.0;
  const double g0 = e0 ? q.gs[l] : 0.0, g1 = e1 ? q.gs[16 + l] : 0.0;
  int k = q.ist[iK];
  {
    const double gmax = row_max16(fmax(fabs(g0), fabs(g1))), xmax = row_max16(fmax(fabs(x0), fabs(x1)));
    const int kGoOn = 12345;
    int ret = kGoOn;
    if (gmax / fmax(1.0, xmax) < P.g_epsilon) {
      ret = 0;
    } else {
      if (0 < P.past) {
        const int slot = k % P.past;
        const double pf = q.st[sPF0 + slot];
        __threadfence_block();
        if (P.past <= k) {
          const double rate = fabs(pf - fx) / fmax(1.0, fabs(fx));
          if (rate < P.delta) ret = 1;
        }
        if (ret == kGoOn && l == 0) q.st[sPF0 + slot] = fx;
      }
      if (ret == kGoOn && P.max_iterations != 0 && P.max_iterations <= k) ret = -1008;
    }
    if (ret != kGoOn) {
      if (l == 0) {
        q.ist[iRET] = ret;
        q.ist[iACTION] = kActDone;
      }
      return;
    }
  }
  ++k;
  pr.tick(6);
  const int end = q.ist[iEND];
  int bound = q.ist[iBOUND];
  __threadfence_block();
  if (l == 0) q.ist[iK] = k;

  // ---- history update + two-loop recursion (lbfgs.hpp:676-740); (s, y) interleaved per element
  const double sv0 = e0 ? x0 - v.xp0 : 0.0, sv1 = e1 ? x1 - v.xp1 : 0.0;
  const double yv0 = e0 ? g0 - v.gp0 : 0.0, yv1 = e1 ? g1 - v.gp1 : 0.0;
  if (e0) {
    d2_t sy;
    sy.x = sv0;
    sy.y = yv0;
    ((gd2_t)hS)[(size_t)end * npad + l] = sy;
  }
  if (e1) {
    d2_t sy;
    sy.x = sv1;
    sy.y = yv1;
    ((gd2_t)hS)[(size_t)end * npad + 16 + l] = sy;
  }
  v.d0 = e0 ? -g0 : 0.0;
  v.d1 = e1 ? -g1 : 0.0;
  // the four dot products of lbfgs.hpp:683-694
  const double ys = row_sum32(yv0 * sv0, yv1 * sv1, n, l);
  const double yy = row_sum32(yv0 * yv0, yv1 * yv1, n, l);
  const double ss = row_sum32(sv0 * sv0, sv1 * sv1, n, l);
  const double gpgp = row_sum32(v.gp0 * v.gp0, v.gp1 * v.gp1, n, l);
  if (l == 0) {
    d2_t yr;
    yr.x = ys;
    yr.y = 1.0 / ys;
    ((gd2_t)hR)[end] = yr;
    if (!rcp_route_ok(ys)) q.ist[iSLOWDIV] = 1; // from here on the recursion divides (see div_by_rcp)
  }
  const double cau = ss * sqrt(gpgp) * P.cautious_factor;
  pr.tick(7);
  if (ys > cau) {
    ++bound;
    bound = m < bound ? m : bound;
    const int ne = end + 1 == m ? 0 : end + 1;
    __threadfence_block(); // the newest pair's row and (ys, 1 / ys) are read back below
    const bool exact = q.ist[iSLOWDIV] != 0;
    double d0 = v.d0, d1 = v.d1;
    // (the division mode is a wave-level choice of code: the reciprocal route has no branch in its steps)
    if (__builtin_amdgcn_ballot_w64(exact) != 0ull) q4_two_loop<true>(q, (gcd2_t)hS, (gcd2_t)hR, npad, m, n, l, bound, ne, exact, ys / yy, d0, d1);
    else q4_two_loop<false>(q, (gcd2_t)hS, (gcd2_t)hR, npad, m, n, l, bound, ne, exact, ys / yy, d0, d1);
    v.d0 = e0 ? d0 : 0.0;
    v.d1 = e1 ? d1 : 0.0;
    if (l == 0) {
      q.ist[iEND] = ne;
      q.ist[iBOUND] = bound;
      long long hs = ((long long)q.ist[iHISTHI] << 32) | (unsigned int)q.ist[iHISTLO];
      hs += bound;
      q.ist[iHISTLO] = (int)(hs & 0xffffffffLL);
      q.ist[iHISTHI] = (int)(hs >> 32);
    }
  }
  if (l == 0) q.st[sSTEP] = 1.0; // lbfgs.hpp:743
  pr.tick(8);
  __threadfence_block();
  const bool ok = q4_begin_iteration(P, q, v, n, l);
  if (l == 0) q.ist[iACTION] = ok ? kActEval : kActDone;
  pr.tick(6);
}

// solver state of a suspended trajectory <-> its record in DevBatch::state (the layout of solver_ref.hip's state_io: five vectors at
// pitch npad -- x, xp, g, gp, d --, the scalars, the integers)
__device__ inline void q4_state_io(const DevBatch &D, const Q4 &q, QVec &v, int b, int l, bool save) {
  const int n = D.L.n, npad = D.L.npad;
  double *rec = D.state + (size_t)b * D.state_stride;
  for (int h = 0; h < 2; h++) {
    const int e = 16 * h + l;
    if (e >= n) {
      if (!save) {
        q.xs[e] = 0.0;
        q.gs[e] = 0.0;
        if (h == 0) { v.xp0 = 0.0; v.gp0 = 0.0; v.d0 = 0.0; } else { v.xp1 = 0.0; v.gp1 = 0.0; v.d1 = 0.0; }
      }
      continue;
    }
    if (save) {
      rec[0 * npad + e] = q.xs[e];
      rec[1 * npad + e] = h == 0 ? v.xp0 : v.xp1;
      rec[2 * npad + e] = q.gs[e];
      rec[3 * npad + e] = h == 0 ? v.gp0 : v.gp1;
      rec[4 * npad + e] = h == 0 ? v.d0 : v.d1;
    } else {
      q.xs[e] = rec[0 * npad + e];
      q.gs[e] = rec[2 * npad + e];
      if (h == 0) {
        v.xp0 = rec[1 * npad + e];
        v.gp0 = rec[3 * npad + e];
        v.d0 = rec[4 * npad + e];
      } else {
        v.xp1 = rec[1 * npad + e];
        v.gp1 = rec[3 * npad + e];
        v.d1 = rec[4 * npad + e];
      }
    }
  }
  double *r2 = rec + 5 * npad;
  for (int w = l; w < sNUM; w += 16) {
    if (save) r2[w] = q.st[w];
    else q.st[w] = r2[w];
  }
  int *ri = reinterpret_cast<int *>(r2 + 24);
  for (int w = l; w < iNUM; w += 16) {
    if (save) ri[w] = q.ist[w];
    else q.ist[w] = ri[w];
  }
}

// ------------------------------------------------ the kernel
// source bit 0: the rows pop trajectories from the batch's ring (solves; DevBatch::queue as solver_ref.hip uses it) -- otherwise row
// r of wave w of workgroup i takes trajectory (i W + w) 4 + r; bit 1: test hook, true divisions in the recursion from the start.
// slice: evaluations of a wave after which its unfinished trajectories go back to the ring (all four rows together, so that the
// rows of a wave are refilled together and the last trajectories of a batch gather in few waves).  hand: see the slice's end.
template <bool FAST, bool DENSE>
__global__ void __launch_bounds__(256, DFTPAV_Q4_WAVES_PER_EU)
    ref4_kernel(const DevBatch *__restrict__ Dp, int mode, const double *__restrict__ tabs, const double *__restrict__ cor_t, double *__restrict__ scratch, int source,
                int slice, int hand) {
  extern __shared__ double lds_raw[];
  const DevBatch &D = *Dp;
  const DevLayout &L = D.L;
  const int tidb = threadIdx.x, Tb = blockDim.x, lane = tidb & 63, wv = tidb >> 6, W = Tb >> 6, row = lane >> 4, l = lane & 15;
  const int n = L.n, N = L.Ntot, H = L.H, mem = D.P.mem_size;
  Q4 q;
  q4_carve(q, reinterpret_cast<char *>(lds_raw) + q4_shared_bytes(N) + (size_t)(wv * 4 + row) * q4_team_bytes(mem), mem);
  DenseLds dl; // the wave's list of active terms (behind the rows of all waves)
  q4_carve_dense(dl, reinterpret_cast<char *>(lds_raw) + q4_shared_bytes(N) + (size_t)W * 4 * q4_team_bytes(mem) + (size_t)wv * q4_dense_bytes());
  const ldscd_t tab = (ldscd_t)lds_raw;
  for (int i = tidb; i < pk_segment_doubles(N); i += Tb) ((ldsd_t)lds_raw)[i] = tabs[i];
  __syncthreads(); // the only time the waves of the workgroup meet
  const bool ring = mode == kModeSolve && (source & 1) != 0;
  const bool force_exact_div = (source & 2) != 0;
  const int nterm = 5 * H + 4, JP = L.Kmax + 1;
  const size_t cpitch = (size_t)JP * 16;
  const size_t scratch_per_traj = (size_t)L.Npts * nterm * kRec + (size_t)L.Npts * nterm; // reference_order_scratch_per_traj, no obstacles
  Prof pr;
  pr.on = false;
  pr.acc = nullptr;
  pr.last = 0;
  QVec v{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int b = -1;
  bool act = false, resumed = false;
  long long tick0 = 0;
  int steps = 0;

  for (int pass = 0;; pass++) {
    // ---- rows without a trajectory take one
    if (!act) {
      int id = -1;
      if (ring) {
        for (int r = 0; r < 4; r++) // one row after the other (the rows of a wave would otherwise race each other for the head)
          if (row == r && l == 0) id = ring_pop(D.qctl, D.queue, D.qcap);
        id = __shfl(id, lane & 48);
      } else if (pass == 0) {
        id = ((int)blockIdx.x * W + wv) * 4 + row;
        if (id >= D.B) id = -1;
      }
      if (id >= 0) {
        b = id;
        act = true;
        resumed = ring && D.sflag[b] == 1;
        if (resumed) {
          q4_state_io(D, q, v, b, l, false);
        } else {
          const double *xsrc = (mode == kModeSolve) ? D.x0 : D.x_in;
          for (int h = 0; h < 2; h++) {
            const int e = 16 * h + l;
            q.xs[e] = e < n ? xsrc[(size_t)b * n + e] : 0.0;
            q.gs[e] = 0.0;
          }
          v = QVec{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
          q.ist[l] = (l == iSLOWDIV && force_exact_div) ? 1 : 0; // (iNUM == 16 lanes)
        }
        if (l < 12) q.bnd[l] = l < 6 ? D.iniS[(size_t)b * 6 + l] : D.finS[(size_t)b * 6 + (l - 6)];
        tick0 = wall_clock64();
        pr.start(D.prof != nullptr && mode == kModeSolve && l == 0, D.prof + (size_t)b * 12, resumed);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
    }
    if (__builtin_amdgcn_ballot_w64(act) == 0ull) break; // (uniform) this wave has nothing left to do
    if (act) {
      const gcd_t cor = (gcd_t)(cor_t + (size_t)b * L.H * 4 * cpitch + l);
      const gd_t ovf = (gd_t)(scratch + (size_t)b * scratch_per_traj);
      const double f = q4_eval<FAST, DENSE>(D, q, dl, tab, cor, cpitch, ovf, l, pr);
      if (mode == kModeEval) {
        for (int h = 0; h < 2; h++) {
          const int e = 16 * h + l;
          if (e < n) D.g_out[(size_t)b * n + e] = q.gs[e];
        }
        if (l == 0) D.f_eval[b] = f;
        act = false;
      } else {
        const gd_t hS = (gd_t)(D.histS + (size_t)b * mem * L.npad * 2);
        const gd_t hR = (gd_t)(D.histR + (size_t)b * mem * 2);
        q4_advance(D, q, v, f, hS, hR, l, pr);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (q.ist[iACTION] == kActDone) { // the epilogue of solver_ref.hip
          for (int h = 0; h < 2; h++) {
            const int e = 16 * h + l;
            if (e < n) D.x_out[(size_t)b * n + e] = q.xs[e];
          }
          if (l == 0) {
            const double fx = q.st[sFX];
            const int ret = q.ist[iRET];
            D.f_out[b] = fx;
            D.status[b] = ret;
            D.iters[b] = q.ist[iK];
            D.evals[b] = q.ist[iEVALS];
            D.hist_sum[b] = ((long long)q.ist[iHISTHI] << 32) | (unsigned int)q.ist[iHISTLO];
            {
              double *rec = reinterpret_cast<double *>(D.records + (size_t)16 * b); // the all-gather record
              rec[0] = fx;
              int *ri = reinterpret_cast<int *>(rec + 1);
              ri[0] = ret;
              ri[1] = q.ist[iK];
            }
            if (D.records_host != nullptr) { // (see device_types.h)
              double *rh = reinterpret_cast<double *>(D.records_host + (size_t)16 * b);
              rh[0] = fx;
              int *rj = reinterpret_cast<int *>(rh + 1);
              rj[0] = ret;
              rj[1] = q.ist[iK];
            }
            D.ticks[b] = (resumed ? D.ticks[b] : 0) + (wall_clock64() - tick0); // time in service
            int ok = (ret == 0 || ret == 1 || ret == 2 || ret == -1008 || ret == -1009) ? 1 : 0; // traj_optimizer.cpp:176-201
            if (fx >= D.P.fail_cost) ok = 0;
            D.success[b] = ok;
            if (ring) {
              D.sflag[b] = 2;
              atomicSub(&D.qctl[3], 1u);
            }
          }
          act = false;
        }
      }
    }
    if (!ring) {
      if (mode == kModeEval) break;
      continue;
    }
    // ---- end of a slice: the wave's unfinished trajectories go back to the ring
    steps++;
    if (slice > 0 && steps >= slice) { // (uniform)
      steps = 0;
      // hand: once no more than this many trajectories of the batch are unfinished, the waves put theirs back and LEAVE -- the host
      // has a launch of the WAVE shape (solver_ref.hip: a wave per trajectory, 2-3 x faster per iteration on a device that is
      // emptying) queued behind this one, which pops them from the same ring and resumes them from the same records
      const bool leave = hand > 0 && __hip_atomic_load(&D.qctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= (unsigned)hand;
      if (act) {
        q4_state_io(D, q, v, b, l, true);
        __threadfence(); // the record and the history rows of this slice are out before the id is handed on
        if (l == 0) {
          D.ticks[b] = (resumed ? D.ticks[b] : 0) + (wall_clock64() - tick0);
          D.sflag[b] = 1;
        }
      }
      for (int r = 0; r < 4; r++)
        if (act && row == r && l == 0) ring_push(D.qctl, D.queue, D.qcap, b);
      act = false;
      if (leave) break;
    }
  }
}

// the corridor of a batch [B][4 H][NptsPad] -> [B][4 H][Kmax + 1][16]: element (j, p) = the value at constraint point j of piece p
// (0.0 where the piece has no such point), so that the 16 lanes of a row read a round's half-planes as 128 contiguous bytes
__global__ void q4_corridor_kernel(const double *__restrict__ cor, double *__restrict__ out, int B, int H4, int NptsPad, int N, int K, int Kd, int JP) {
  const size_t total = (size_t)B * H4 * JP * 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(i & 15);
    const size_t r = i >> 4;
    const int j = (int)(r % JP);
    const size_t bc = r / JP; // b * H4 + component
    double v = 0.0;
    if (p < N) {
      const int Kp = (p == 0 || p == N - 1) ? Kd : K;
      const int pt0 = p == 0 ? 0 : (Kd + 1) + (p - 1) * (K + 1);
      if (j <= Kp) v = cor[bc * NptsPad + pt0 + j];
    }
    out[i] = v;
  }
}

} // namespace reford

// ---- host side
// what the layout must satisfy for the QUAD shape (header)
bool reference_order_quad_supported(const DevLayout &L, const DevParams &P, int S) {
  if (L.M != 1 || S != 0 || L.n > 32 || L.Ntot > 16 || L.Ntot < 2 || L.H < 1 || L.H > 5) return false;
  return reford::q4_shared_bytes(L.Ntot) + 4 * reford::q4_team_bytes(P.mem_size) + reford::q4_dense_bytes() <= 160 * 1024;
}
size_t reference_order_quad_corridor_doubles(const DevLayout &L, int B) { return (size_t)B * L.H * 4 * (L.Kmax + 1) * 16; }
// fills RefPlan for the QUAD shape: as many waves per workgroup (at most 4) and workgroups per CU as the LDS holds, eight waves
// per CU at most (256 registers)
void reference_order_quad_plan(const DevLayout &L, const DevParams &P, int B, int n_cu, RefPlan &pl) {
  // (a wave: its four rows and its list of active terms)
  const size_t shared = reford::q4_shared_bytes(L.Ntot), team = reford::q4_team_bytes(P.mem_size) + (reford::q4_dense_bytes() + 3) / 4, budget = 160 * 1024;
  int best_w = 1, best_wg = 1, best_res = 0;
  for (int w = std::min(4, reford::kQ4WavesPerCU); w >= 1; w--) {
    const size_t lds = shared + (size_t)w * 4 * team;
    if (lds > budget) continue;
    const int wg = (int)std::min<size_t>((size_t)(reford::kQ4WavesPerCU / w), budget / lds);
    if (wg * w >= best_res) { // ties: the smaller workgroup (it leaves sooner at the end of a launch)
      best_res = wg * w;
      best_w = w;
      best_wg = wg;
    }
  }
  if (const char *e = std::getenv("DFTPAV_REF_QUAD_WAVES")) { // developer knob: waves per workgroup
    const int w = std::atoi(e);
    if (w >= 1 && w <= 4 && shared + (size_t)w * 4 * team <= budget) {
      best_w = w;
      best_wg = (int)std::min<size_t>((size_t)(reford::kQ4WavesPerCU / w), budget / (shared + (size_t)w * 4 * team));
    }
  }
  pl.quad = 1;
  pl.wave = 1;
  pl.threads = 64 * best_w;
  pl.lds = shared + (size_t)best_w * 4 * team;
  pl.wg_per_cu = best_wg;
  // Persistent workgroups of a scheduled solve: a wave's rows are only refilled while the batch's ring holds waiting
  // trajectories, so a launch takes HALF as many rows as the batch has trajectories (two per row) -- the rows stay busy until half
  // of the batch is done, the rest gathers in ever fewer waves (slices), and the waves that leave make room for the next batch's
  // launch on another stream.  Measured on the bench's stream of 4096-batches, four in flight (gpurun_out/q4.log, round 6): 256 / 512 /
  // 768 waves per launch -> 22.7 / 28.0 / 27.1 k solves/s; a launch as wide as the device (every trajectory its own row from the
  // start, no refill): 19.3 k.
  const int per_wg = 4 * best_w;
  pl.slots = std::max(1, std::min(n_cu * best_wg, (B + 2 * per_wg - 1) / (2 * per_wg)));
  pl.slice = 64; // evaluations between two visits to the ring (64 / 256: 28.0 / 26.0 k solves/s)
  pl.slots_wide = n_cu * best_wg; // a batch with the device to itself: every wave slot
  pl.hand = 768;                  // ... and its last trajectories finish in the WAVE shape (launch_ref, capi.cpp)
  if (const char *e = std::getenv("DFTPAV_REF_QUAD_HANDOVER")) pl.hand = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("DFTPAV_REF_SLICE")) pl.slice = std::atoi(e);
  if (const char *e = std::getenv("DFTPAV_REF_SLOTS")) pl.slots = pl.slots_wide = std::max(1, std::atoi(e)); // developer knob: persistent workgroups
}
hipError_t launch_quad_corridor(const DevBatch &D, double *cor_t, hipStream_t stream) {
  const DevLayout &L = D.L;
  const size_t total = reference_order_quad_corridor_doubles(L, D.B);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(reford::q4_corridor_kernel, dim3(grid), dim3(256), 0, stream, D.corridor, cor_t, D.B, L.H * 4, D.NptsPad, L.Ntot, L.K, L.Kd, L.Kmax + 1);
  return hipGetLastError();
}
// scheduled != 0: a solve whose rows pop from the batch's ring (the caller has reset it)
// slots: persistent workgroups of this launch; hand: unfinished trajectories at which the waves leave theirs to a follow-up launch
hipError_t launch_solver_ref4(const DevBatch &D, const DevBatch *d_dev, int mode, const double *tabs, const double *cor_t, double *scratch, const RefPlan &pl,
                              int scheduled, int slots, int hand, hipStream_t stream) {
  const int W = pl.threads / 64;
  int grid = (D.B + 4 * W - 1) / (4 * W), source = 0, slice = 0;
  if (scheduled && mode == kModeSolve) {
    grid = slots < grid ? slots : grid;
    source = 1;
    slice = pl.slice;
  } else {
    hand = 0;
  }
  if (const char *e = std::getenv("DFTPAV_REF_EXACT_DIV"))
    if (std::atoi(e) != 0) source |= 2;
  if (std::getenv("DFTPAV_VERBOSE"))
    std::fprintf(stderr, "[dftpav] reference order, QUAD shape: grid %d x %d threads, %zu B of LDS, source %d slice %d hand-over at %d\n", grid, pl.threads, pl.lds, source, slice, hand);
  const bool fast = D.L.H == 4 && D.epis == 0.0 && !std::getenv("DFTPAV_REF_QUAD_GENERIC"); // the live path's constants (q4_eval)
  bool dense = true; // the active terms evaluated densely packed (DFTPAV_REF_QUAD_DENSE=0: in place, as until late in round 6)
  if (const char *e = std::getenv("DFTPAV_REF_QUAD_DENSE")) dense = std::atoi(e) != 0;
  using Kern = void (*)(const DevBatch *, int, const double *, const double *, double *, int, int, int);
  const Kern fn = fast ? (dense ? &reford::ref4_kernel<true, true> : &reford::ref4_kernel<true, false>)
                       : (dense ? &reford::ref4_kernel<false, true> : &reford::ref4_kernel<false, false>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(pl.threads), pl.lds, stream, d_dev, mode, tabs, cor_t, scratch, source, slice, hand);
  return hipGetLastError();
}

} // namespace dftpav
