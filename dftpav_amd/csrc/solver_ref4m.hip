// solver_ref4m.hip — the reference order in the QUAD shape for SEVERAL gear segments (gfx950): four trajectories per wave, one per
// row of 16 lanes, as solver_ref4.hip (read its header first) -- with the pieces of all gear segments laid out on the row's lanes
// one segment after the other, the junction variables (gear position, gear angle) of traj_optimizer.cpp:273-282 / 307-320, and vectors
// of up to 48 variables (three registers per lane: elements l, 16 + l, 32 + l; a sequential dot product is a 48-step chain).
// Scope: up to four gear segments, 16 pieces in all, n <= 48, no moving obstacles, H <= 5 -- BASELINE configs[1] (8 + 8 pieces with a
// gear shift, n = 33).  solver_ref4.hip stays what it is for one segment (the headline); everything else stays with solver_ref.hip.
//
// What changes against the one-segment kernel:
//   * a lane knows its segment: (sg, lp) = (segment, piece inside it), the segment's pieces N, its offset in x, its direction; the
//     piece duration, its powers and reciprocals are the segment's;
//   * cos / sin of the junction angles by the correctly rounded functions (cr_trig.h), one junction per lane, shared through LDS;
//     the boundary states in force (junction position from x, junction velocity from the angle) per segment in LDS;
//   * the four substitution sweeps run all segments side by side: at step s the lanes that own block s of THEIR segment work; a
//     segment's first block starts from zeros, whatever the neighbour lane (another segment's last piece) holds;
//   * the per-segment chains (gdT, energy, the parked terms' shares) are formed per segment with the other segments' lanes masked to
//     -0.0; the gradients of a junction's position and angle come from two neighbouring lanes (the last piece of one segment, the
//     first of the next: one DPP step).
// Same bits as the TEAM / WAVE shapes and the restatement with correctly rounded cos / sin (oracle order 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#include "ref_order_common.h"
#include "quad_common.h"

namespace dftpav {
namespace reford {

constexpr int kMLcap = 8;  // parked terms of a piece kept in LDS (the rest in global scratch)
constexpr int kMSeg = 4;   // gear segments
constexpr int kMV = 3;     // registers of a solver vector per lane (n <= 48)

struct Q4M {
  ldsd_t xs, gs;  // [48] the trial point, the gradient
  ldsd_t bnd;     // [kMSeg][12] iniS [6], finS [6] of each segment as uploaded (clamped)
  ldsd_t pva;     // [kMSeg][12] head / tail position, velocity, acceleration in force for this x (junction overrides)
  ldsd_t trig;    // [kMSeg][2] cos, sin of the junction angles
  ldsd_t tw;      // [kMSeg][8] 1 / t^k of the segment's piece duration (6), its real duration T (slot 6)
  ldsd_t st;      // [sNUM]
  ldsd_t alpha;   // [mem]
  ldsd_t tl;      // [16][kMLcap][3] parked terms of a piece
  ldsi_t ist;     // [iNUM]
  ldsi_t tcnt;    // [16]
};
__host__ __device__ inline size_t q4m_team_doubles(int mem) { return 48 + 48 + kMSeg * (12 + 12 + 2 + 8) + sNUM + (size_t)mem + 16 * kMLcap * 3; }
__host__ __device__ inline size_t q4m_team_bytes(int mem) { return (q4m_team_doubles(mem) * sizeof(double) + (iNUM + 16) * sizeof(int) + 15) & ~(size_t)15; }
// The segments' sweep tables in LDS: segments with as many pieces share one copy (BASELINE configs[1], 8 + 8 pieces: 8 KB, and with
// them a fourth wave per CU).  lds_off[sg]: where segment sg's tables start in LDS; returns the doubles they take in all.
__host__ __device__ inline int q4m_table_layout(const DevLayout &L, int (&lds_off)[kMSeg]) {
  int next = 0;
  for (int sg = 0; sg < kMSeg; sg++) lds_off[sg] = 0;
  for (int sg = 0; sg < L.M && sg < kMSeg; sg++) {
    int first = sg;
    for (int q = sg - 1; q >= 0; q--)
      if (L.piece_nums[q] == L.piece_nums[sg]) first = q;
    if (first == sg) {
      lds_off[sg] = next;
      next += pk_segment_doubles(L.piece_nums[sg]);
    } else {
      lds_off[sg] = lds_off[first];
    }
  }
  return next;
}
__host__ __device__ inline size_t q4m_table_doubles(const DevLayout &L) {
  int off[kMSeg];
  return (size_t)q4m_table_layout(L, off);
}
__host__ __device__ inline size_t q4m_shared_bytes(const DevLayout &L) { return (q4m_table_doubles(L) * sizeof(double) + 15) & ~(size_t)15; }
__device__ inline void q4m_carve(Q4M &q, char *team, int mem) {
  ldsd_t p = (ldsd_t)reinterpret_cast<double *>(team);
  q.xs = p; p += 48;
  q.gs = p; p += 48;
  q.bnd = p; p += 12 * kMSeg;
  q.pva = p; p += 12 * kMSeg;
  q.trig = p; p += 2 * kMSeg;
  q.tw = p; p += 8 * kMSeg;
  q.st = p; p += sNUM;
  q.alpha = p; p += mem;
  q.tl = p; p += 16 * kMLcap * 3;
  ldsi_t i = (ldsi_t)p;
  q.ist = i; i += iNUM;
  q.tcnt = i;
}

// what a lane knows about its piece
struct LaneSeg {
  int sg, lp, N;   // segment, piece inside it, pieces of the segment
  int x0;          // offset of the segment's waypoints in x
  int singul;
  int toff;        // offset (doubles) of the segment's sweep tables
  int pt0;         // the piece's first constraint point (trajectory-wide numbering)
  int Kl;          // its intervals
  bool piece;      // this lane has a piece at all
};
__host__ __device__ inline LaneSeg lane_segment(const DevLayout &L, int l) {
  LaneSeg s{0, 0, 2, 0, 1, 0, 0, 1, false};
  int off[kMSeg];
  q4m_table_layout(L, off);
  for (int q = 0; q < L.M; q++) {
    const bool in = l >= L.seg_piece0[q] && l < L.seg_piece0[q + 1];
    if (in) {
      s.sg = q;
      s.lp = l - L.seg_piece0[q];
      s.N = L.piece_nums[q];
      s.x0 = L.seg_x0[q];
      s.singul = L.singuls[q];
      s.toff = off[q];
      s.pt0 = L.seg_pt0[q] + (s.lp == 0 ? 0 : (L.Kd + 1) + (s.lp - 1) * (L.K + 1));
      s.piece = true;
    }
  }
  s.Kl = (s.lp == 0 || s.lp == s.N - 1) ? L.Kd : L.K;
  return s;
}

// 0.0 + p[0] + ... + p[n-1] for a vector held as (element l, 16 + l, 32 + l); elements from n on contribute -0.0 (x + (-0.0) == x).
// TAIL: what is known of n - 32 at compile time -- 1: n == 33 (BASELINE configs[1]: 8 + 8 pieces; the reference's live case 5 + 4 + 6):
// the third part of the chain is one addition, not sixteen of which fifteen add -0.0; 16: anything
template <int TAIL>
__device__ __forceinline__ double row_sum48(const double (&p)[kMV], int n, int l) {
  double acc = row_chain16(0.0, l < n ? p[0] : -0.0);
  if (n > 16) acc = row_chain16(acc, 16 + l < n ? p[1] : -0.0); // (uniform)
  if (TAIL == 1) acc = row_add_lane0(acc, p[2]);
  else if (n > 32) acc = row_chain16(acc, 32 + l < n ? p[2] : -0.0);
  return acc;
}

// One substitution sweep of every segment's band system, the segments side by side (solver_ref4.hip: sweep4).  tabq: the start of
// sweep Q inside this lane's segment's tables.  Nmax: the largest segment's pieces.
template <int Q>
__device__ __forceinline__ void sweep4m(ldscd_t tabq, double (&bq)[12], const LaneSeg &S, int Nmax) {
  constexpr bool DESC = Q == 1 || Q == 3, DIV = Q == 1 || Q == 2;
  const int N = S.N;
  ldscd_t ip = tabq + 48;
  v2d_t c[pk_size(Q) / 2];
  {
    const int sl = DESC ? N - 1 - S.lp : S.lp; // the step this lane's piece is taken in
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
  for (int s = 0; s < Nmax; s++) {
    const int p = DESC ? N - 1 - s : s;
    double w[6][2];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int d = 0; d < 2; d++) w[r][d] = DESC ? nb_dpp<0x101>(bq[2 * (5 - r) + d]) : nb_dpp<0x111>(bq[2 * r + d]);
    if (s == 0) { // (uniform) every segment's traversal starts from six zeros, whatever the neighbour lane holds
#pragma unroll
      for (int r = 0; r < 6; r++) w[r][0] = w[r][1] = 0.0;
    }
    const bool mine = S.piece && s < N && S.lp == p;
    if (mine && (s == 0 || s == N - 1)) { // a block of the ends: every coefficient is tested, as the reference does (`if (a != 0.0)`)
      const ldscv2_t a = (ldscv2_t)(s == 0 ? tabq : ip + (N - 2) * pk_size(Q));
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
    } else if (mine) { // an interior block: the non-zero terms only, no test
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

// ------------------------------------------------ costFunctionCallback (traj_optimizer.cpp:206-350), several gear segments
template <bool FAST>
__device__ __forceinline__ double q4m_eval(const DevBatch &D, const Q4M &q, ldscd_t tab, gcd_t cor, size_t cpitch, gd_t ovf, int l, const LaneSeg &S, int Nmax,
                                           Prof &pr) {
  const DevLayout &L = D.L;
  const DevParams &P = D.P;
  const int M = L.M, Ntot = L.Ntot, H = FAST ? 4 : L.H, nterm = 5 * H + 4, t0 = 5 * H;
  const double epis = FAST ? 0.0 : D.epis;
  const int N = S.N, sg = S.sg, lp = S.lp;
  const bool piece = S.piece;
  // ---- durations (VirtualT2RealT, :371-379), their powers (poly_traj_utils.hpp:961-966): the lane's own segment's
  const double vt = q.xs[L.x_tau0 + sg];
  const double Tr = vt > 0.0 ? ((0.5 * vt + 1.0) * vt + 1.0) + P.mini_T : 1.0 / ((0.5 * vt - 1.0) * vt + 1.0) + P.mini_T;
  const double t1 = Tr / N, t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
  // ---- cos / sin of the junction angles: the reference calls libm's (host-dependent bits); here the correctly rounded ones
  if (l < M - 1) {
    double sn, cs;
    crt::sincos(q.xs[L.x_ang0 + l], sn, cs);
    q.trig[2 * l] = cs;
    q.trig[2 * l + 1] = sn;
  }
  if (piece && lp == 0) q.tw[8 * sg + 6] = Tr;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  // ---- boundary states in force (IniS / FinS of :270-282): junction position from x, junction velocity from the angle
  if (l < M) {
    const int s_ = l;
    ldscd_t ini = q.bnd + 12 * s_, fin = ini + 6;
    ldsd_t hv = q.pva + 12 * s_, tv = hv + 6;
    for (int u = 0; u < 6; u++) {
      hv[u] = ini[u];
      tv[u] = fin[u];
    }
    if (s_ > 0) {
      hv[0] = q.xs[L.x_gear0 + 2 * (s_ - 1)];
      hv[1] = q.xs[L.x_gear0 + 2 * (s_ - 1) + 1];
      hv[2] = -P.non_sinv * q.trig[2 * (s_ - 1)];
      hv[3] = -P.non_sinv * q.trig[2 * (s_ - 1) + 1];
    }
    if (s_ < M - 1) {
      tv[0] = q.xs[L.x_gear0 + 2 * s_];
      tv[1] = q.xs[L.x_gear0 + 2 * s_ + 1];
      tv[2] = P.non_sinv * q.trig[2 * s_];
      tv[3] = P.non_sinv * q.trig[2 * s_ + 1];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  pr.tick(10);
  // ---- right-hand sides (poly_traj_utils.hpp:968-977): the rows of this lane's piece
  ldscd_t hvs = q.pva + 12 * sg, tvs = hvs + 6;
  double bq[12];
#pragma unroll
  for (int u = 0; u < 12; u++) bq[u] = 0.0;
  if (piece && lp == 0) {
#pragma unroll
    for (int d = 0; d < 2; d++) {
      bq[0 + d] = hvs[d];
      bq[2 + d] = hvs[2 + d] * t1;
      bq[4 + d] = hvs[4 + d] * t2;
    }
  }
  if (piece && lp == N - 1) {
#pragma unroll
    for (int d = 0; d < 2; d++) {
      bq[6 + d] = tvs[d];
      bq[8 + d] = tvs[2 + d] * t1;
      bq[10 + d] = tvs[4 + d] * t2;
    }
  } else if (piece && lp < N - 1) {
    bq[10] = q.xs[S.x0 + 2 * lp];
    bq[11] = q.xs[S.x0 + 2 * lp + 1];
  }
  // ---- BandedSystem::solve (poly_traj_utils.hpp:805-826)
  ldscd_t tabs_ = tab + S.toff;
  sweep4m<0>(tabs_, bq, S, Nmax);
  sweep4m<1>(tabs_ + pk_sweep_offset(1, N), bq, S, Nmax);
  pr.tick(0);
  // ---- c = b * tInv (:979-984)
  double cc[12];
  {
    const double tI[6] = {1.0 / 1.0, 1.0 / t1, 1.0 / t2, 1.0 / t3, 1.0 / t4, 1.0 / t5};
#pragma unroll
    for (int u = 0; u < 12; u++) cc[u] = bq[u] * tI[u >> 1];
    if (piece && lp == 0) {
#pragma unroll
      for (int u = 0; u < 6; u++) q.tw[8 * sg + u] = tI[u]; // kept for calGrads_PT
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
  const int Kl = S.Kl;
  const double step = t1 / Kl;
  double s1 = 0.0;
  int cnt = 0;
  const gd_t ovf_l = ovf + (size_t)S.pt0 * nterm * 3;
  double pl[20];
  load_planes(cor, cpitch, H, pl);
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the first round's half-planes are waited for here, not inside the loop (solver_ref4.hip)
#pragma unroll 1
  for (int j = 0; j <= L.Kmax; j++) {
    unsigned m = 0u;
    PtState pst;
    double nx[20];
    load_planes(cor + (size_t)(j < L.Kmax ? j + 1 : j) * 16, cpitch, H, nx);
    if (piece && j <= Kl)
      m = (unsigned)point_masks<false>(P, cc, lp, N, j, Kl, step, s1, S.singul, epis, H, pl, (gd_t) nullptr, D.sur, 0.0, 0.0, 0, 0.0, pst);
    s1 += step; // the running sum of traj_optimizer.cpp:513
    for (unsigned mm = m; mm;) {
      const int t = __builtin_ctz(mm);
      mm &= mm - 1;
      double r_[16];
      point_emit_pf(P, pst, t, H, t0,
                    [&](int k, double &on0, double &on1, double &q0, double &q1) { // the planes point_masks tested, still in registers
                      on0 = pl[0]; on1 = pl[1]; q0 = pl[2]; q1 = pl[3];
#pragma unroll
                      for (int u = 1; u < 5; u++) {
                        on0 = k == u ? pl[4 * u] : on0;
                        on1 = k == u ? pl[4 * u + 1] : on1;
                        q0 = k == u ? pl[4 * u + 2] : q0;
                        q1 = k == u ? pl[4 * u + 3] : q1;
                      }
                    },
                    (double *)r_);
#pragma unroll
      for (int u = 0; u < 12; u++) gdC[u] += r_[u];
      const bool corr = t < t0;
      const double e0 = r_[12], e1 = corr ? r_[13] : -0.0, e2 = corr ? -0.0 : r_[13];
      if (cnt < kMLcap) {
        ldsd_t e = q.tl + (l * kMLcap + cnt) * 3;
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
    }
#pragma unroll
    for (int u = 0; u < 20; u++) pl[u] = nx[u];
  }
  q.tcnt[l] = piece ? cnt : 0;
  __threadfence_block();
  pr.tick(2);
  // ---- the per-segment chains: `gdT +=`, `energy +=` over the segment's pieces in order from 0.0, then its parked terms in (piece,
  // point, term) order; the other segments' lanes hand a chain -0.0 (every lane of the row forms all of them: the same bits)
  double gdT[kMSeg], en[kMSeg], cost0[kMSeg], cost2[kMSeg];
#pragma unroll
  for (int s_ = 0; s_ < kMSeg; s_++) {
    gdT[s_] = en[s_] = cost0[s_] = cost2[s_] = 0.0;
    if (s_ < M) { // (uniform)
      gdT[s_] = row_chain16(0.0, piece && sg == s_ ? pG : -0.0);
      en[s_] = row_chain16(0.0, piece && sg == s_ ? pE : -0.0);
    }
  }
  {
    int total = 0;
    for (int p = 0; p < Ntot; p++) total += q.tcnt[p];
    pr.count(9, total);
    if (total > 0) {
      for (int p = 0; p < Ntot; p++) {
        const int c = q.tcnt[p];
        if (__builtin_amdgcn_ballot_w64(c != 0) == 0ull) continue; // (uniform)
        int sgp = 0, lpp = p;
        for (int s_ = 0; s_ < M; s_++)
          if (p >= L.seg_piece0[s_]) {
            sgp = s_;
            lpp = p - L.seg_piece0[s_];
          }
        const int pp0 = L.seg_pt0[sgp] + (lpp == 0 ? 0 : (L.Kd + 1) + (lpp - 1) * (L.K + 1));
        const gcd_t og = (gcd_t)(ovf + (size_t)pp0 * nterm * 3);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0; // the running sums of the piece's segment
#pragma unroll
        for (int s_ = 0; s_ < kMSeg; s_++) {
          a0 = sgp == s_ ? gdT[s_] : a0;
          a1 = sgp == s_ ? cost0[s_] : a1;
          a2 = sgp == s_ ? cost2[s_] : a2;
        }
        {
          double e[kMLcap][3];
#pragma unroll
          for (int i = 0; i < kMLcap; i++)
#pragma unroll
            for (int w = 0; w < 3; w++) e[i][w] = q.tl[(p * kMLcap + i) * 3 + w];
#pragma unroll
          for (int i = 0; i < kMLcap; i++) {
            a0 += i < c ? e[i][0] : -0.0;
            a1 += i < c ? e[i][1] : -0.0;
            a2 += i < c ? e[i][2] : -0.0;
          }
        }
        for (int i0 = kMLcap; i0 < c; i0 += 8) {
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
              a0 += e[u][0];
              a1 += e[u][1];
              a2 += e[u][2];
            }
          }
        }
#pragma unroll
        for (int s_ = 0; s_ < kMSeg; s_++) {
          gdT[s_] = sgp == s_ ? a0 : gdT[s_];
          cost0[s_] = sgp == s_ ? a1 : cost0[s_];
          cost2[s_] = sgp == s_ ? a2 : cost2[s_];
        }
      }
    }
  }
  pr.tick(3);
  // ---- calGrads_PT (poly_traj_utils.hpp:1037-1066): adj = gdC * tInv, solveAdj, the duration gradient
  double pA;
  double tI[6];
#pragma unroll
  for (int u = 0; u < 6; u++) tI[u] = q.tw[8 * sg + u];
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
  sweep4m<2>(tabs_ + pk_sweep_offset(2, N), adj, S, Nmax);
  sweep4m<3>(tabs_ + pk_sweep_offset(3, N), adj, S, Nmax);
  pr.tick(4);
  // ---- gradient and cost (traj_optimizer.cpp:299-344)
  if (piece && lp < N - 1) { // gdP: rows 6 i + 5 of the segment's adjoint
    q.gs[S.x0 + 2 * lp] = adj[10];
    q.gs[S.x0 + 2 * lp + 1] = adj[11];
  }
  {
    // the duration gradient of every segment (poly_traj_utils.hpp:1050-1064, VirtualTGradCost :405-419): its head terms live in
    // its first lane, its tail terms in its last (every other lane hands the chain a -0.0)
    const double h1 = hvs[2] * adj[2] + hvs[3] * adj[3];
    const double h2 = (hvs[4] * adj[4] + hvs[5] * adj[5]) * 2.0 * t1;
    const double g1 = tvs[2] * adj[8] + tvs[3] * adj[9];
    const double g2 = (tvs[4] * adj[10] + tvs[5] * adj[11]) * 2.0 * t1;
    double mine_gdT = 0.0;
    const int rb = (int)(threadIdx.x & 48); // the row's first lane
#pragma unroll
    for (int s_ = 0; s_ < kMSeg; s_++) {
      if (s_ < M) { // (uniform)
        int first = 0, last = 0;
        for (int q2 = 0; q2 < M; q2++) {
          first = q2 == s_ ? L.seg_piece0[q2] : first;
          last = q2 == s_ ? L.seg_piece0[q2 + 1] - 1 : last;
        }
        double a = gdT[s_];
        a += __shfl(h1, rb + first);
        a += __shfl(h2, rb + first);
        a += __shfl(g1, rb + last);
        a += __shfl(g2, rb + last);
        a = row_chain16(a, piece && sg == s_ ? pA : -0.0);
        gdT[s_] = a;
        mine_gdT = sg == s_ ? a : mine_gdT;
      }
    }
    double gdVT2Rt;
    if (vt > 0) {
      gdVT2Rt = vt + 1.0;
    } else {
      const double denSqrt = (0.5 * vt - 1.0) * vt + 1.0;
      gdVT2Rt = (1.0 - vt) / (denSqrt * denSqrt);
    }
    if (piece && lp == 0) q.gs[L.x_tau0 + sg] = (mine_gdT / N + P.wei_time) * gdVT2Rt;
  }
  {
    // junction i between segment i and i + 1 (traj_optimizer.cpp:307-320): gdTail of segment i (the neighbour lane: its last piece)
    // and gdHead of segment i + 1 (this lane: its first piece) -- poly_traj_utils.hpp:1045-1049: adj row * t^k
    double nadj[4], nt1;
#pragma unroll
    for (int u = 0; u < 4; u++) nadj[u] = nb_dpp<0x111>(adj[6 + u]); // the neighbour's rows 3 and 4
    nt1 = nb_dpp<0x111>(t1);
    if (piece && lp == 0 && sg > 0) {
      const int i = sg - 1;
      if (P.gear_opt) {
        const double fin0[2] = {nadj[0] * 1.0, nadj[1] * 1.0}, fin1[2] = {nadj[2] * nt1, nadj[3] * nt1};
        const double ini0[2] = {adj[0] * 1.0, adj[1] * 1.0}, ini1[2] = {adj[2] * t1, adj[3] * t1};
        const double cs = q.trig[2 * i], sn = q.trig[2 * i + 1];
        // grad is zeroed, then segment i adds its tail term, then segment i + 1 its head term (trajid ascending)
        for (int d = 0; d < 2; d++) {
          double v = 0.0;
          v += fin0[d];
          v += ini0[d];
          q.gs[L.x_gear0 + 2 * i + d] = v;
        }
        double va = 0.0;
        va += fin1[0] * (-P.non_sinv * sn) + fin1[1] * (P.non_sinv * cs);
        va += ini1[0] * (P.non_sinv * sn) + ini1[1] * (-P.non_sinv * cs);
        q.gs[L.x_ang0 + i] = va;
      } else { // gear_opt off: the junction variables keep a zero gradient
        q.gs[L.x_gear0 + 2 * i] = 0.0;
        q.gs[L.x_gear0 + 2 * i + 1] = 0.0;
        q.gs[L.x_ang0 + i] = 0.0;
      }
    }
  }
  // the cost: sums over the segments in order (:292-297, :328-330)
  double total_smcost = 0.0, total_timecost = 0.0, penalty_cost = 0.0;
#pragma unroll
  for (int s_ = 0; s_ < kMSeg; s_++)
    if (s_ < M) {
      total_smcost += en[s_];
      penalty_cost += (cost0[s_] + 0.0) + cost2[s_]; // (the moving-obstacle cost of a segment without obstacles is its start value 0.0)
    }
  for (int s_ = 0; s_ < M; s_++) total_timecost += q.tw[8 * s_ + 6] * P.wei_time;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // g is read by other lanes of the row
  pr.tick(5);
  return total_smcost + total_timecost + penalty_cost;
}

// ------------------------------------------------ L-BFGS, per row (solver_ref4.hip's, a vector in three registers)
struct QVecM {
  double xp[kMV], gp[kMV], d[kMV];
};
__device__ __forceinline__ void q4m_read(ldscd_t a, int n, int l, double (&o)[kMV]) {
#pragma unroll
  for (int h = 0; h < kMV; h++) o[h] = 16 * h + l < n ? a[16 * h + l] : 0.0;
}
template <int TAIL>
__device__ __forceinline__ bool q4m_begin_iteration(const DevParams &P, const Q4M &q, QVecM &v, int n, int l) {
  q4m_read(q.xs, n, l, v.xp);
  q4m_read(q.gs, n, l, v.gp);
  double pr_[kMV];
#pragma unroll
  for (int h = 0; h < kMV; h++) pr_[h] = v.gp[h] * v.d[h];
  const double dginit = row_sum48<TAIL>(pr_, n, l);
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
#pragma unroll
  for (int h = 0; h < kMV; h++)
    if (16 * h + l < n) q.xs[16 * h + l] = v.xp[h] + step * v.d[h];
  return true;
}

constexpr int kMB = 4; // stored pairs per register block
struct MBlk {
  d2_t e[kMB][kMV]; // (s, y) of elements l, 16 + l, 32 + l
  d2_t yr[kMB];     // (ys, 1 / ys)
};
template <int DIR, int TAIL>
__device__ __forceinline__ void q4m_load_blk(MBlk &R, gcd2_t hS, gcd2_t hR, int npad, int m, int l, int &jl) {
#pragma unroll
  for (int u = 0; u < kMB; u++) {
    const gcd2_t row = hS + (size_t)jl * npad + l; // (rows are npad >= 64 long: what lies beyond n is never used)
    R.e[u][0] = row[0];
    R.e[u][1] = row[16];
    // n == 33: the third register is ONE element, and every lane of the row asks for that one (16 bytes of the row, not 256: the history is
    // what this kernel reads from HBM).  (Masked or clamped loads of "what the row holds" for any n were tried, round 6: a select on the loaded
    // value waits for the load where it is issued -- a block ahead of its use -- and a branch per load breaks the block up: 241 -> 389 / 278 /
    // 268 ms per step of the stream.)
    R.e[u][2] = TAIL == 1 ? row[32 - l] : row[32];
    R.yr[u] = hR[jl];
    if (DIR < 0) jl = jl == 0 ? m - 1 : jl - 1;
    else jl = jl == m - 1 ? 0 : jl + 1;
  }
}
__device__ __forceinline__ void q4m_pin_blk(MBlk &R) {
#pragma unroll
  for (int u = 0; u < kMB; u++) {
#pragma unroll
    for (int h = 0; h < kMV; h++) asm volatile("" : "+v"(R.e[u][h].x), "+v"(R.e[u][h].y));
    asm volatile("" : "+v"(R.yr[u].x), "+v"(R.yr[u].y));
  }
}
template <bool EXACT, int TAIL>
__device__ __forceinline__ void q4m_first_steps(const MBlk &R, const Q4M &q, int i0, int bound, int m, int n, int l, bool exact, int &j, double (&d)[kMV]) {
#pragma unroll
  for (int u = 0; u < kMB; u++) {
    if (i0 + u < bound) { // (row-uniform)
      j = j == 0 ? m - 1 : j - 1;
      double pr_[kMV];
#pragma unroll
      for (int h = 0; h < kMV; h++) pr_[h] = R.e[u][h].x * d[h];
      const double dot = row_sum48<TAIL>(pr_, n, l);
      const double a = EXACT && exact ? dot / R.yr[u].x : div_by_rcp<false>(dot, R.yr[u].x, R.yr[u].y); // lm_alpha[j] = lm_s.col(j).dot(d) / lm_ys[j]
      if (l == 0) q.alpha[j] = a;
      const double na = -a;
#pragma unroll
      for (int h = 0; h < kMV; h++) d[h] = d[h] + na * R.e[u][h].y; // d += (-alpha) * lm_y.col(j)
    }
  }
}
template <bool EXACT, int TAIL>
__device__ __forceinline__ void q4m_second_steps(const MBlk &R, const Q4M &q, int i0, int bound, int m, int n, int l, bool exact, int &j, double (&d)[kMV]) {
#pragma unroll
  for (int u = 0; u < kMB; u++) {
    if (i0 + u < bound) { // (row-uniform)
      const double al = q.alpha[j];
      double pr_[kMV];
#pragma unroll
      for (int h = 0; h < kMV; h++) pr_[h] = R.e[u][h].y * d[h];
      const double dot = row_sum48<TAIL>(pr_, n, l);
      const double beta = EXACT && exact ? dot / R.yr[u].x : div_by_rcp<false>(dot, R.yr[u].x, R.yr[u].y);
      const double cf = al - beta;
#pragma unroll
      for (int h = 0; h < kMV; h++) d[h] = d[h] + cf * R.e[u][h].x; // d += (alpha - beta) * lm_s.col(j)
      j = j == m - 1 ? 0 : j + 1;
    }
  }
}
template <bool EXACT, int TAIL>
__device__ __forceinline__ void q4m_two_loop(const Q4M &q, gcd2_t hS, gcd2_t hR, int npad, int m, int n, int l, int bound, int ne, bool exact, double sc0,
                                             double (&d)[kMV]) {
  MBlk A, B;
  int j = ne;
  int jl = ne == 0 ? m - 1 : ne - 1;
  q4m_load_blk<-1, TAIL>(A, hS, hR, npad, m, l, jl);
#pragma unroll 1
  for (int i0 = 0; i0 < bound; i0 += 2 * kMB) {
    q4m_pin_blk(A);
    q4m_load_blk<-1, TAIL>(B, hS, hR, npad, m, l, jl);
    q4m_first_steps<EXACT, TAIL>(A, q, i0, bound, m, n, l, exact, j, d);
    q4m_pin_blk(B);
    q4m_load_blk<-1, TAIL>(A, hS, hR, npad, m, l, jl);
    q4m_first_steps<EXACT, TAIL>(B, q, i0 + kMB, bound, m, n, l, exact, j, d);
  }
#pragma unroll
  for (int h = 0; h < kMV; h++) d[h] = d[h] * sc0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // alpha written by lane 0 of the row, read by all below
  jl = j;
  q4m_load_blk<+1, TAIL>(A, hS, hR, npad, m, l, jl);
#pragma unroll 1
  for (int i0 = 0; i0 < bound; i0 += 2 * kMB) {
    q4m_pin_blk(A);
    q4m_load_blk<+1, TAIL>(B, hS, hR, npad, m, l, jl);
    q4m_second_steps<EXACT, TAIL>(A, q, i0, bound, m, n, l, exact, j, d);
    q4m_pin_blk(B);
    q4m_load_blk<+1, TAIL>(A, hS, hR, npad, m, l, jl);
    q4m_second_steps<EXACT, TAIL>(B, q, i0 + kMB, bound, m, n, l, exact, j, d);
  }
}

// Everything lbfgs_optimize does between two evaluations: solver_ref4.hip's q4_advance with three registers per vector
template <int TAIL>
__device__ __forceinline__ void q4m_advance(const DevBatch &D, const Q4M &q, QVecM &v, double f, gd_t hS, gd_t hR, int l, Prof &pr) {
  const DevParams &P = D.P;
  const int n = D.L.n, m = P.mem_size, npad = D.L.npad;
  int action = kActEval;
  auto vmax = [&](const double (&a)[kMV]) {
    double mx = 0.0;
#pragma unroll
    for (int h = 0; h < kMV; h++) mx = fmax(mx, fabs(a[h]));
    return row_max16(mx);
  };
  if (q.ist[iPHASE] == 0) { // after the first evaluation: lbfgs.hpp:524-551
    double g[kMV], x[kMV], sq[kMV];
    q4m_read(q.gs, n, l, g);
    q4m_read(q.xs, n, l, x);
#pragma unroll
    for (int h = 0; h < kMV; h++) {
      v.d[h] = 16 * h + l < n ? -g[h] : 0.0;
      sq[h] = (-g[h]) * (-g[h]);
    }
    const double gmax = vmax(g), xmax = vmax(x);
    const double dd = row_sum48<TAIL>(sq, n, l);
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
      if (!q4m_begin_iteration<TAIL>(P, q, v, n, l)) action = kActDone;
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
      double g[kMV], pr_[kMV];
      q4m_read(q.gs, n, l, g);
#pragma unroll
      for (int h = 0; h < kMV; h++) pr_[h] = g[h] * v.d[h];
      const double gs = row_sum48<TAIL>(pr_, n, l);
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
#pragma unroll
      for (int h = 0; h < kMV; h++)
        if (16 * h + l < n) q.xs[16 * h + l] = v.xp[h] + stp * v.d[h];
      if (l == 0) q.ist[iACTION] = kActEval;
      pr.tick(6);
      return;
    }
  }
  if (l == 0) q.st[sSTEP] = stp; // lbfgs.hpp:574 passes `step` by reference
  if (ls < 0) { // lbfgs.hpp:604-611: x, g reverted; fx is not
#pragma unroll
    for (int h = 0; h < kMV; h++)
      if (16 * h + l < n) {
        q.xs[16 * h + l] = v.xp[h];
        q.gs[16 * h + l] = v.gp[h];
      }
    if (l == 0) {
      q.ist[iRET] = ls;
      q.ist[iACTION] = kActDone;
    }
    return;
  }
  // ---- convergence / stopping tests (lbfgs.hpp:628-666)
  double x[kMV], g[kMV];
  q4m_read(q.xs, n, l, x);
  q4m_read(q.gs, n, l, g);
  int k = q.ist[iK];
  {
    const double gmax = vmax(g), xmax = vmax(x);
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
  // ---- history update + two-loop recursion (lbfgs.hpp:676-740)
  double sv[kMV], yv[kMV], p0[kMV], p1[kMV], p2[kMV], p3[kMV];
#pragma unroll
  for (int h = 0; h < kMV; h++) {
    const bool in = 16 * h + l < n;
    sv[h] = in ? x[h] - v.xp[h] : 0.0;
    yv[h] = in ? g[h] - v.gp[h] : 0.0;
    if (in) {
      d2_t sy;
      sy.x = sv[h];
      sy.y = yv[h];
      ((gd2_t)hS)[(size_t)end * npad + 16 * h + l] = sy;
    }
    v.d[h] = in ? -g[h] : 0.0;
    p0[h] = yv[h] * sv[h];
    p1[h] = yv[h] * yv[h];
    p2[h] = sv[h] * sv[h];
    p3[h] = v.gp[h] * v.gp[h];
  }
  const double ys = row_sum48<TAIL>(p0, n, l), yy = row_sum48<TAIL>(p1, n, l), ss = row_sum48<TAIL>(p2, n, l), gpgp = row_sum48<TAIL>(p3, n, l);
  if (l == 0) {
    d2_t yr;
    yr.x = ys;
    yr.y = 1.0 / ys;
    ((gd2_t)hR)[end] = yr;
    if (!rcp_route_ok(ys)) q.ist[iSLOWDIV] = 1;
  }
  const double cau = ss * sqrt(gpgp) * P.cautious_factor;
  pr.tick(7);
  if (ys > cau) {
    ++bound;
    bound = m < bound ? m : bound;
    const int ne = end + 1 == m ? 0 : end + 1;
    __threadfence_block(); // the newest pair's row and (ys, 1 / ys) are read back below
    const bool exact = q.ist[iSLOWDIV] != 0;
    double d[kMV];
#pragma unroll
    for (int h = 0; h < kMV; h++) d[h] = v.d[h];
    if (__builtin_amdgcn_ballot_w64(exact) != 0ull) q4m_two_loop<true, TAIL>(q, (gcd2_t)hS, (gcd2_t)hR, npad, m, n, l, bound, ne, exact, ys / yy, d);
    else q4m_two_loop<false, TAIL>(q, (gcd2_t)hS, (gcd2_t)hR, npad, m, n, l, bound, ne, exact, ys / yy, d);
#pragma unroll
    for (int h = 0; h < kMV; h++) v.d[h] = 16 * h + l < n ? d[h] : 0.0;
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
  const bool ok = q4m_begin_iteration<TAIL>(P, q, v, n, l);
  if (l == 0) q.ist[iACTION] = ok ? kActEval : kActDone;
  pr.tick(6);
}

// solver state of a suspended trajectory <-> its record in DevBatch::state (solver_ref.hip's state_io layout)
__device__ inline void q4m_state_io(const DevBatch &D, const Q4M &q, QVecM &v, int b, int l, bool save) {
  const int n = D.L.n, npad = D.L.npad;
  double *rec = D.state + (size_t)b * D.state_stride;
  for (int h = 0; h < kMV; h++) {
    const int e = 16 * h + l;
    if (e >= n) {
      if (!save) {
        q.xs[e] = 0.0;
        q.gs[e] = 0.0;
        v.xp[h] = v.gp[h] = v.d[h] = 0.0;
      }
      continue;
    }
    if (save) {
      rec[0 * npad + e] = q.xs[e];
      rec[1 * npad + e] = v.xp[h];
      rec[2 * npad + e] = q.gs[e];
      rec[3 * npad + e] = v.gp[h];
      rec[4 * npad + e] = v.d[h];
    } else {
      q.xs[e] = rec[0 * npad + e];
      v.xp[h] = rec[1 * npad + e];
      q.gs[e] = rec[2 * npad + e];
      v.gp[h] = rec[3 * npad + e];
      v.d[h] = rec[4 * npad + e];
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

// ------------------------------------------------ the kernel (solver_ref4.hip's loop)
template <bool FAST, int TAIL>
__global__ void __launch_bounds__(256, 1)
    ref4m_kernel(const DevBatch *__restrict__ Dp, int mode, const double *__restrict__ tabs, const double *__restrict__ cor_t, double *__restrict__ scratch, int source,
                 int slice, int hand) {
  extern __shared__ double lds_raw[];
  const DevBatch &D = *Dp;
  const DevLayout &L = D.L;
  const int tidb = threadIdx.x, Tb = blockDim.x, lane = tidb & 63, wv = tidb >> 6, W = Tb >> 6, row = lane >> 4, l = lane & 15;
  const int n = L.n, H = L.H, mem = D.P.mem_size;
  Q4M q;
  q4m_carve(q, reinterpret_cast<char *>(lds_raw) + q4m_shared_bytes(L) + (size_t)(wv * 4 + row) * q4m_team_bytes(mem), mem);
  const ldscd_t tab = (ldscd_t)lds_raw;
  {
    int off[kMSeg], goff = 0;
    q4m_table_layout(L, off);
    for (int sg = 0; sg < L.M; sg++) { // (the batch's buffer holds every segment's tables, one after the other: solver_ref.hip reads them so)
      const int sz = pk_segment_doubles(L.piece_nums[sg]);
      bool first = true;
      for (int q2 = 0; q2 < sg; q2++) first = first && L.piece_nums[q2] != L.piece_nums[sg];
      if (first)
        for (int i = tidb; i < sz; i += Tb) ((ldsd_t)lds_raw)[off[sg] + i] = tabs[goff + i];
      goff += sz;
    }
  }
  __syncthreads(); // the only time the waves of the workgroup meet
  const LaneSeg S = lane_segment(L, l);
  int Nmax = 2;
  for (int s_ = 0; s_ < L.M; s_++) Nmax = L.piece_nums[s_] > Nmax ? L.piece_nums[s_] : Nmax;
  const bool ring = mode == kModeSolve && (source & 1) != 0;
  const bool force_exact_div = (source & 2) != 0;
  const int nterm = 5 * H + 4, JP = L.Kmax + 1;
  const size_t cpitch = (size_t)JP * 16;
  const size_t scratch_per_traj = (size_t)L.Npts * nterm * kRec + (size_t)L.Npts * nterm; // reference_order_scratch_per_traj, no obstacles
  Prof pr;
  pr.on = false;
  pr.acc = nullptr;
  pr.last = 0;
  QVecM v{};
  int b = -1;
  bool act = false, resumed = false;
  long long tick0 = 0;
  int steps = 0;

  for (int pass = 0;; pass++) {
    if (!act) {
      int id = -1;
      if (ring) {
        for (int r = 0; r < 4; r++)
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
          q4m_state_io(D, q, v, b, l, false);
        } else {
          const double *xsrc = (mode == kModeSolve) ? D.x0 : D.x_in;
          for (int h = 0; h < kMV; h++) {
            const int e = 16 * h + l;
            q.xs[e] = e < n ? xsrc[(size_t)b * n + e] : 0.0;
            q.gs[e] = 0.0;
            v.xp[h] = v.gp[h] = v.d[h] = 0.0;
          }
          q.ist[l] = (l == iSLOWDIV && force_exact_div) ? 1 : 0; // (iNUM == 16 lanes)
        }
        for (int w = l; w < 12 * L.M; w += 16) {
          const int s_ = w / 12, u = w - 12 * s_;
          q.bnd[w] = u < 6 ? D.iniS[((size_t)b * L.M + s_) * 6 + u] : D.finS[((size_t)b * L.M + s_) * 6 + (u - 6)];
        }
        tick0 = wall_clock64();
        pr.start(D.prof != nullptr && mode == kModeSolve && l == 0, D.prof + (size_t)b * 12, resumed);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
    }
    if (__builtin_amdgcn_ballot_w64(act) == 0ull) break; // (uniform) this wave has nothing left to do
    if (act) {
      const gcd_t cor = (gcd_t)(cor_t + (size_t)b * L.H * 4 * cpitch + l);
      const gd_t ovf = (gd_t)(scratch + (size_t)b * scratch_per_traj);
      const double f = q4m_eval<FAST>(D, q, tab, cor, cpitch, ovf, l, S, Nmax, pr);
      if (mode == kModeEval) {
        for (int h = 0; h < kMV; h++) {
          const int e = 16 * h + l;
          if (e < n) D.g_out[(size_t)b * n + e] = q.gs[e];
        }
        if (l == 0) D.f_eval[b] = f;
        act = false;
      } else {
        const gd_t hS = (gd_t)(D.histS + (size_t)b * mem * L.npad * 2);
        const gd_t hR = (gd_t)(D.histR + (size_t)b * mem * 2);
        q4m_advance<TAIL>(D, q, v, f, hS, hR, l, pr);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (q.ist[iACTION] == kActDone) { // the epilogue of solver_ref.hip
          for (int h = 0; h < kMV; h++) {
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
    steps++;
    if (slice > 0 && steps >= slice) { // (uniform) end of a slice: the wave's unfinished trajectories go back to the ring
      steps = 0;
      const bool leave = hand > 0 && __hip_atomic_load(&D.qctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= (unsigned)hand;
      if (act) {
        q4m_state_io(D, q, v, b, l, true);
        __threadfence();
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

// the corridor of a batch [B][4 H][NptsPad] -> [B][4 H][Kmax + 1][16] for pieces of several segments: element (j, p) = the value at
// constraint point j of (trajectory-wide) piece p, 0.0 where the piece has no such point
__global__ void q4m_corridor_kernel(const double *__restrict__ corridor, double *__restrict__ out, int B, int NptsPad, DevLayout L) {
  const int H4 = L.H * 4, JP = L.Kmax + 1;
  const size_t total = (size_t)B * H4 * JP * 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(i & 15);
    const size_t r = i >> 4;
    const int j = (int)(r % JP);
    const size_t bc = r / JP; // b * H4 + component
    double val = 0.0;
    if (p < L.Ntot) {
      const LaneSeg S = lane_segment(L, p);
      if (j <= S.Kl) val = corridor[bc * NptsPad + S.pt0 + j];
    }
    out[i] = val;
  }
}

} // namespace reford

// ---- host side
bool reference_order_quadm_supported(const DevLayout &L, const DevParams &P, int S) {
  if (L.M < 1 || L.M > reford::kMSeg || S != 0 || L.n > 16 * reford::kMV || L.Ntot > 16 || L.H < 1 || L.H > 5) return false;
  for (int i = 0; i < L.M; i++)
    if (L.piece_nums[i] < 2) return false;
  return reford::q4m_shared_bytes(L) + 4 * reford::q4m_team_bytes(P.mem_size) <= 160 * 1024;
}
void reference_order_quadm_plan(const DevLayout &L, const DevParams &P, int B, int n_cu, RefPlan &pl) {
  const size_t shared = reford::q4m_shared_bytes(L), team = reford::q4m_team_bytes(P.mem_size), budget = 160 * 1024;
  int best_w = 1, best_wg = 1, best_res = 0;
  for (int w = 4; w >= 1; w--) {
    const size_t lds = shared + (size_t)w * 4 * team;
    if (lds > budget) continue;
    const int wg = (int)std::min<size_t>((size_t)(4 / w), budget / lds);
    if (wg * w >= best_res) {
      best_res = wg * w;
      best_w = w;
      best_wg = wg;
    }
  }
  if (const char *e = std::getenv("DFTPAV_REF_QUAD_WAVES")) { // developer knob: waves per workgroup
    const int w = std::atoi(e);
    if (w >= 1 && w <= 4 && shared + (size_t)w * 4 * team <= budget) {
      best_w = w;
      best_wg = (int)std::min<size_t>((size_t)(4 / w), budget / (shared + (size_t)w * 4 * team));
    }
  }
  pl.quad = 2;
  pl.wave = 1;
  pl.threads = 64 * best_w;
  pl.lds = shared + (size_t)best_w * 4 * team;
  pl.wg_per_cu = best_wg;
  const int per_wg = 4 * best_w;
  pl.slots = std::max(1, std::min(n_cu * best_wg, (B + 2 * per_wg - 1) / (2 * per_wg)));
  pl.slice = 64;
  pl.slots_wide = n_cu * best_wg;
  pl.hand = 2048; // (a batch alone on the device, configs[1] at 4096: 401 ms with the hand-over at 768 unfinished trajectories, 386 at 1280, 373-390 at 2048)
  if (const char *e = std::getenv("DFTPAV_REF_QUAD_HANDOVER")) pl.hand = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("DFTPAV_REF_SLICE")) pl.slice = std::atoi(e);
  if (const char *e = std::getenv("DFTPAV_REF_SLOTS")) pl.slots = pl.slots_wide = std::max(1, std::atoi(e));
}
hipError_t launch_quadm_corridor(const DevBatch &D, double *cor_t, hipStream_t stream) {
  const size_t total = (size_t)D.B * D.L.H * 4 * (D.L.Kmax + 1) * 16;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(reford::q4m_corridor_kernel, dim3(grid), dim3(256), 0, stream, D.corridor, cor_t, D.B, D.NptsPad, D.L);
  return hipGetLastError();
}
hipError_t launch_solver_ref4m(const DevBatch &D, const DevBatch *d_dev, int mode, const double *tabs, const double *cor_t, double *scratch, const RefPlan &pl,
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
    std::fprintf(stderr, "[dftpav] reference order, QUAD shape (several segments): grid %d x %d threads, %zu B of LDS, source %d slice %d hand-over at %d\n", grid,
                 pl.threads, pl.lds, source, slice, hand);
  const bool fast = D.L.H == 4 && D.epis == 0.0 && !std::getenv("DFTPAV_REF_QUAD_GENERIC");
  const bool tail1 = D.L.n == 33 && !std::getenv("DFTPAV_REF_QUAD_GENERIC");
  using Kern = void (*)(const DevBatch *, int, const double *, const double *, double *, int, int, int);
  const Kern fn = fast ? (tail1 ? &reford::ref4m_kernel<true, 1> : &reford::ref4m_kernel<true, 16>)
                       : (tail1 ? &reford::ref4m_kernel<false, 1> : &reford::ref4m_kernel<false, 16>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(pl.threads), pl.lds, stream, d_dev, mode, tabs, cor_t, scratch, source, slice, hand);
  return hipGetLastError();
}

} // namespace dftpav
