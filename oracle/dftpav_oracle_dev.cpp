/*
 * dftpav_oracle_dev.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * DEVICE-ORDER mode of the CPU oracle: the same algorithm as the literal
 * restatement in dftpav_oracle.c, with every sum taken in the order the gfx950
 * kernel (dftpav_amd/csrc/solver.hip) takes it, so that the kernel can be held
 * to BIT-EXACT agreement on whole L-BFGS solves.  Why that is needed: the
 * reference solver is chaotic (1 ulp on x0 moves its final cost by up to 16 %,
 * tests/test_oracle_sensitivity.py), so no tolerance on single evaluations can
 * guarantee the same iterate sequence; only an identical floating-point program can.
 *
 * What differs from literal mode (all reassociations of the same mathematics,
 * each checked against literal mode to ~1e-13 per evaluation by
 * tests/test_oracle_orders.py):
 *   - MINCO solve / adjoint solve (poly_traj_utils.hpp:979,1042): product with the
 *     dense operator A_N^{-1}|(N+5 columns), built with the reference's banded
 *     LU (oracle_minco_operator) — instead of running the substitution per call;
 *   - penalty gradient (traj_optimizer.cpp:611-705): per-point subtotals of
 *     d/dsigma, d/dsigma', d/dsigma'' chained onto the coefficients piece by
 *     piece, instead of one running += over all points;
 *   - the three cost classes are summed into one running total per point;
 *   - dot products / norms of L-BFGS (Eigen-internal order in the reference,
 *     unpinned): 64-lane strided partials + xor butterfly 1,2,4,8,16,32;
 *   - cos/sin/exp/log: the portable routines of traj_math.h (libm is not
 *     bit-reproducible across host and device).
 * The per-point mathematics is the shared header dftpav_amd/csrc/traj_math.h
 * (the oracle may include product headers, never the reverse); orchestration,
 * reductions and the L-BFGS driver below are written independently of the kernel.
 */
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle_internal.h"
#include "../dftpav_amd/csrc/traj_math.h"

using namespace dftpav;

namespace {

struct DevState {
  DevLayout L;
  DevParams P;
  DevSurround S{};
  std::vector<int> sur_off;
  std::vector<double> sur_total, sur_start;
  std::vector<std::vector<double>> opM, opMT; // per segment
  // work arrays (names follow the kernel's LDS carve)
  std::vector<double> seg, stab, rhs, b, c, gdC, adj, part, pE, pGsm, pGdT, pCost, pChain;
};

void fill_params(const dftpav_params &p, DevParams &P) { // == what dftpav_batch_create hands the kernel
  P.wei_obs = p.wei_obs;
  P.wei_surround = p.wei_surround;
  P.wei_feas = p.wei_feas;
  P.wei_time = p.wei_time;
  P.surround_clearance = p.surround_clearance;
  P.max_vel[0] = p.max_forward_vel; P.max_vel[1] = p.max_backward_vel;
  P.max_acc[0] = p.max_forward_acc; P.max_acc[1] = p.max_backward_acc;
  P.max_cur[0] = p.max_forward_cur; P.max_cur[1] = p.max_backward_cur;
  P.non_sinv = p.non_sinv;
  P.mini_T = p.mini_T;
  P.fail_cost = p.fail_cost;
  P.gear_opt = p.gear_opt;
  P.mem_size = p.lbfgs_mem_size;
  P.past = p.lbfgs_past;
  P.max_iterations = p.lbfgs_max_iterations;
  P.max_linesearch = p.lbfgs_max_linesearch;
  P.delta = p.lbfgs_delta;
  P.g_epsilon = p.lbfgs_g_epsilon;
  P.min_step = p.lbfgs_min_step;
  P.max_step = p.lbfgs_max_step;
  P.f_dec_coeff = p.lbfgs_f_dec_coeff;
  P.s_curv_coeff = p.lbfgs_s_curv_coeff;
  P.cautious_factor = p.lbfgs_cautious_factor;
  P.machine_prec = p.lbfgs_machine_prec;
}

inline int seg_of_piece(const DevLayout &L, int p) {
  int s = 0;
  while (s + 1 < L.M && p >= L.seg_piece0[s + 1]) ++s;
  return s;
}

// the kernel's wave_sum: lanes hold v[0..63]; v[i] += v[i ^ o] for o = 1, 2, 4, 8, 16, 32
// (quad permutes, row_half_mirror, row_mirror, permlane16_swap, permlane32_swap)
// The kernel folds only as many levels as the vector needs: 4 / 5 / 6 for n <= 16 / 32 / 64+.
static thread_local int g_levels = 6;
inline int levels_for(int n) { return n <= 16 ? 4 : (n <= 32 ? 5 : 6); }
inline double butterfly_sum(double v[64]) {
  double t[64];
  for (int o = 1; o < (1 << g_levels); o <<= 1) {
    for (int i = 0; i < 64; i++) t[i] = v[i] + v[i ^ o];
    std::memcpy(v, t, sizeof(t));
  }
  return v[0];
}
// dot(a,b) as wave 0 computes it: lane l accumulates elements l, l+64, ... from 0.0, then butterfly
inline double wave_dot(const double *a, const double *b, int n) {
  double v[64];
  for (int l = 0; l < 64; l++) {
    double acc = 0.0;
    for (int e = l; e < n; e += 64) acc += a[e] * b[e];
    v[l] = acc;
  }
  return butterfly_sum(v);
}
// dot inside the two-loop recursion when n <= 64: one element per lane, no leading 0.0 +
inline double lane_dot(const double *a, const double *b, int n) {
  double v[64];
  for (int l = 0; l < 64; l++) v[l] = l < n ? a[l] * b[l] : 0.0;
  return butterfly_sum(v);
}
inline double absmax(const double *a, int n) {
  double m = 0.0;
  for (int e = 0; e < n; e++) m = std::fmax(m, std::fabs(a[e]));
  return m;
}

struct HostPlanes {
  const double *p; // [H][4] normalised planes of one point
  inline void operator()(int k, double &n0, double &n1, double &q0, double &q1) const {
    n0 = p[4 * k];
    n1 = p[4 * k + 1];
    q0 = p[4 * k + 2];
    q1 = p[4 * k + 3];
  }
};

double dev_eval(oracle_ctx *c, DevState &D, const double *x, double *g) {
  const DevLayout &L = D.L;
  const DevParams &P = D.P;
  const double *iniS = c->iniS, *finS = c->finS;
  // ---- E1
  for (int sg = 0; sg < L.M; sg++) {
    double Tr = virtual_to_real(x[L.x_tau0 + sg], P.mini_T);
    double dt = Tr / L.piece_nums[sg];
    double *s = &D.seg[sg * 16];
    s[0] = Tr;
    s[1] = dt;
    duration_powers(dt, s + 2);
    for (int which = 0; which < 2; which++) {
      int K = which ? L.Kd : L.K;
      double step = dt / K;
      double *tab = &D.stab[(sg * 2 + which) * (L.Kmax + 1)];
      double s1 = 0.0;
      for (int j = 0; j <= K; j++) {
        tab[j] = s1;
        s1 += step;
      }
    }
    int N = L.piece_nums[sg];
    for (int col = 0; col < N + 5; col++)
      for (int d = 0; d < 2; d++) {
        double v;
        if (col < 3) {
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
        } else {
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
        D.rhs[2 * (L.seg_rhs0[sg] + col) + d] = v;
      }
  }
  // ---- E2
  for (int w = 0; w < 12 * L.Ntot; w++) {
    int r = w >> 1, d = w & 1;
    int p = r / 6, k = r - 6 * p;
    int sg = seg_of_piece(L, p);
    int N = L.piece_nums[sg];
    int lr = r - 6 * L.seg_piece0[sg];
    const double *Mrow = &D.opM[sg][(size_t)lr * (N + 5)];
    const double *rh = &D.rhs[2 * L.seg_rhs0[sg] + d];
    double acc = 0.0;
    for (int col = 0; col < N + 5; col++) acc = fma_(Mrow[col], rh[2 * col], acc);
    D.b[w] = acc;
    D.c[w] = acc * D.seg[sg * 16 + 8 + k];
  }
  // ---- E3
  for (int p = 0; p < L.Ntot; p++) {
    int sg = seg_of_piece(L, p);
    double en, gsm;
    piece_smoothness(&D.c[12 * p], &D.seg[sg * 16 + 2], en, gsm, &D.gdC[12 * p]);
    D.pE[p] = en;
    D.pGsm[p] = gsm;
    D.pGdT[p] = 0.0;
    D.pCost[p] = 0.0;
  }
  // ---- E4: per point the 14 contributions to its piece; per piece and output: the value E3 left, then the groups of 32
  // (16) consecutive points, each summed by the kernel's cross-lane tree (pairs at distance 1, 2, 4, 8, 16), then the
  // points that fill no group, in point order (device_types.h: e4_group_size; solver.hip: reduce16)
  {
    int pt = 0;
    for (int sg = 0; sg < L.M; sg++)
      for (int lp = 0; lp < L.piece_nums[sg]; lp++) {
        int N = L.piece_nums[sg];
        bool edge = (lp == 0 || lp == N - 1);
        int K = edge ? L.Kd : L.K;
        int p = L.seg_piece0[sg] + lp;
        const int G = e4_group_size(K + 1), nf = G ? (K + 1) / G : 0;
        std::vector<double> contrib((size_t)(K + 1) * 14), contrib_d; // contrib_d: 14 values per (point, obstacle) pair of the piece
        for (int j = 0; j <= K; j++, pt++) {
          SampleIn in;
          in.j = j;
          in.K = K;
          in.lp = lp;
          in.N = N;
          in.dt = D.seg[sg * 16 + 1];
          in.s1 = D.stab[(sg * 2 + (edge ? 1 : 0)) * (L.Kmax + 1) + j];
          in.cc = &D.c[12 * p];
          in.singul = L.singuls[sg];
          in.epis = c->epis;
          in.H = L.H;
          in.trajid = sg;
          in.trajtime = sg == 0 ? 0.0 : D.seg[(sg - 1) * 16];
          in.t_piece = piece_start_time(in.dt, lp);
          in.t_now = c->t_now;
          HostPlanes pl{c->cfgHs + (size_t)pt * L.H * 4};
          double o[8];
          sample_point_math<false, 0>(P, D.S, in, pl, o);
          point_contributions(in.s1, o, &contrib[(size_t)j * 14]);
          if (D.S.S > 0) { // the kernel's pair stage: every obstacle that passes the distance gate at this point, in order
            const unsigned mask = dynamic_gate_mask(P, D.S, in);
            for (int u = 0; u < D.S.S; u++)
              if (mask >> u & 1u) {
                double v[14];
                dynamic_pair_math(P, D.S, in, u, o);
                point_contributions(in.s1, o, v);
                contrib_d.insert(contrib_d.end(), v, v + 14);
              }
          }
        }
        for (int q = 0; q < 14; q++) {
          double acc = q < 12 ? D.gdC[12 * p + q] : (q == 12 ? D.pGdT[p] : D.pCost[p]);
          auto tree = [&](const std::vector<double> &cv, int g) {
            double v[32];
            for (int l = 0; l < G; l++) v[l] = cv[(size_t)(g * G + l) * 14 + q];
            for (int o = 1; o < G; o <<= 1) {
              double w[32];
              for (int l = 0; l < G; l++) w[l] = v[l] + v[l ^ o];
              for (int l = 0; l < G; l++) v[l] = w[l];
            }
            return v[0];
          };
          for (int g = 0; g < nf; g++) acc += tree(contrib, g);
          for (int j = nf * G; j <= K; j++) acc += contrib[(size_t)j * 14 + q];
          for (size_t i = 0; i < contrib_d.size() / 14; i++) acc += contrib_d[i * 14 + q]; // pairs in (point, obstacle) order
          if (q < 12) D.gdC[12 * p + q] = acc;
          else if (q == 12) D.pGdT[p] = acc;
          else D.pCost[p] = acc;
        }
      }
  }
  // ---- E5: four strided partial chains per output (rows q, q+4, ...), combined (p0+p1)+(p2+p3)
  for (int w = 0; w < 2 * L.rhs_tot; w++) {
    int row = w >> 1, d = w & 1;
    int sg = 0;
    while (sg + 1 < L.M && row >= L.seg_rhs0[sg + 1]) ++sg;
    int col = row - L.seg_rhs0[sg];
    int N = L.piece_nums[sg];
    const double *MT = &D.opMT[sg][(size_t)col * 6 * N];
    const double *gc = &D.gdC[12 * L.seg_piece0[sg] + d];
    const double *tInv = &D.seg[sg * 16 + 8];
    double part[4];
    for (int q = 0; q < 4; q++) {
      double acc = 0.0;
      for (int r = q; r < 6 * N; r += 4) acc = fma_(MT[r], gc[2 * r] * tInv[r % 6], acc);
      part[q] = acc;
    }
    D.adj[w] = (part[0] + part[1]) + (part[2] + part[3]);
  }
  // per-piece chain-rule terms, then every per-segment sum as the kernel takes it: lane l holds
  // 0.0 + the values of pieces l, l+64, ... of the segment, folded by the full 64-lane butterfly
  std::vector<double> segsum(L.M * 8, 0.0);
  {
    const int saved = g_levels;
    g_levels = 6;
    for (int sg = 0; sg < L.M; sg++) {
      const double *tInv = &D.seg[sg * 16 + 8];
      double gdtInv[6] = {0.0, -1.0 * tInv[2], -2.0 * tInv[3], -3.0 * tInv[4], -4.0 * tInv[5], -5.0 * tInv[5] * tInv[1]};
      double lanes[5][64];
      for (int l = 0; l < 64; l++) {
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
        for (int p = L.seg_piece0[sg] + l; p < L.seg_piece0[sg + 1]; p += 64) {
          const double *gc = &D.gdC[12 * p];
          const double *bb = &D.b[12 * p];
          double acc = 0.0;
          for (int k = 0; k < 6; k++) acc += gdtInv[k] * (gc[2 * k] * bb[2 * k] + gc[2 * k + 1] * bb[2 * k + 1]);
          D.pChain[p] = acc;
          v4 += acc;
          v0 += D.pE[p];
          v1 += D.pCost[p];
          v2 += D.pGsm[p];
          v3 += D.pGdT[p];
        }
        lanes[0][l] = v0; lanes[1][l] = v1; lanes[2][l] = v2; lanes[3][l] = v3; lanes[4][l] = v4;
      }
      for (int q = 0; q < 5; q++) segsum[sg * 8 + q] = butterfly_sum(lanes[q]);
    }
    g_levels = saved;
  }
  // ---- E6
  double f;
  {
    double sm_cost = 0.0, pen = 0.0, tc = 0.0;
    for (int sg = 0; sg < L.M; sg++) {
      sm_cost += segsum[sg * 8 + 0];
      pen += segsum[sg * 8 + 1];
      tc += D.seg[sg * 16] * P.wei_time;
    }
    f = sm_cost + tc + pen;
    c->cost_terms[0] = sm_cost;
    c->cost_terms[1] = tc;
    c->cost_terms[2] = pen; /* device order does not split the penalty classes */
    c->cost_terms[3] = 0.0;
    c->cost_terms[4] = 0.0;
  }
  for (int e = 0; e < L.n; e++) {
    if (e < L.x_tau0) {
      int sg = 0;
      while (sg + 1 < L.M && e >= L.seg_x0[sg + 1]) ++sg;
      int q = e - L.seg_x0[sg];
      int wp = q >> 1, d = q & 1;
      g[e] = D.adj[2 * (L.seg_rhs0[sg] + 3 + wp) + d];
    } else if (e < L.x_gear0) {
      int sg = e - L.x_tau0;
      int N = L.piece_nums[sg];
      const double *seg = &D.seg[sg * 16];
      double dt = seg[1];
      double gdT = 0.0;
      gdT += segsum[sg * 8 + 2];
      gdT += segsum[sg * 8 + 3];
      const double *ad = &D.adj[2 * L.seg_rhs0[sg]];
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
      int rt = N + 2;
      gdT += hv[0] * ad[2 * 1] + hv[1] * ad[2 * 1 + 1];
      gdT += (iniS[sg * 6 + 4] * ad[2 * 2] + iniS[sg * 6 + 5] * ad[2 * 2 + 1]) * 2.0 * dt;
      gdT += tv[0] * ad[2 * (rt + 1)] + tv[1] * ad[2 * (rt + 1) + 1];
      gdT += (finS[sg * 6 + 4] * ad[2 * (rt + 2)] + finS[sg * 6 + 5] * ad[2 * (rt + 2) + 1]) * 2.0 * dt;
      gdT += segsum[sg * 8 + 4];
      g[e] = (gdT / N + P.wei_time) * virtual_to_real_grad(x[e]);
    } else if (e < L.x_ang0) {
      int q = e - L.x_gear0;
      int i = q >> 1, d = q & 1;
      double v = 0.0;
      if (P.gear_opt) {
        int Ni = L.piece_nums[i];
        v += D.adj[2 * (L.seg_rhs0[i] + Ni + 2) + d] * 1.0;
        v += D.adj[2 * (L.seg_rhs0[i + 1] + 0) + d] * 1.0;
      }
      g[e] = v;
    } else {
      int i = e - L.x_ang0;
      double v = 0.0;
      if (P.gear_opt) {
        double th = x[e];
        int Ni = L.piece_nums[i];
        double dti = D.seg[i * 16 + 1], dtn = D.seg[(i + 1) * 16 + 1];
        double ft0 = D.adj[2 * (L.seg_rhs0[i] + Ni + 3) + 0] * dti, ft1 = D.adj[2 * (L.seg_rhs0[i] + Ni + 3) + 1] * dti;
        double hd0 = D.adj[2 * (L.seg_rhs0[i + 1] + 1) + 0] * dtn, hd1 = D.adj[2 * (L.seg_rhs0[i + 1] + 1) + 1] * dtn;
        v += ft0 * (-P.non_sinv * p_sin(th)) + ft1 * (P.non_sinv * p_cos(th));
        v += hd0 * (P.non_sinv * p_sin(th)) + hd1 * (-P.non_sinv * p_cos(th));
      }
      g[e] = v;
    }
  }
  return f;
}

} // namespace

extern "C" void oracle_dev_init(oracle_ctx *c) {
  DevState *D = new DevState();
  DevLayout &L = D->L;
  std::memset(&L, 0, sizeof(L));
  const dftpav_params &p = c->P;
  L.M = c->M;
  L.H = c->H;
  L.K = p.traj_resolution;
  L.Kd = p.des_traj_resolution;
  L.Kmax = L.K > L.Kd ? L.K : L.Kd;
  int xoff = 0, poff = 0, roff = 0, ptoff = 0;
  for (int i = 0; i < L.M; i++) {
    int N = c->piece_nums[i];
    L.piece_nums[i] = N;
    L.singuls[i] = c->singuls[i];
    L.seg_piece0[i] = poff;
    L.seg_x0[i] = xoff;
    L.seg_rhs0[i] = roff;
    L.seg_pt0[i] = ptoff;
    poff += N;
    xoff += 2 * (N - 1);
    roff += N + 5;
    ptoff += (N - 2) * (L.K + 1) + 2 * (L.Kd + 1);
  }
  L.seg_piece0[L.M] = poff;
  L.seg_rhs0[L.M] = roff;
  L.seg_pt0[L.M] = ptoff;
  L.Ntot = poff;
  L.rhs_tot = roff;
  L.Npts = ptoff;
  L.x_tau0 = xoff;
  L.x_gear0 = xoff + L.M;
  L.x_ang0 = L.x_gear0 + 2 * (L.M - 1);
  L.n = L.x_ang0 + (L.M - 1);
  L.npad = ((L.n + 63) / 64) * 64;
  fill_params(p, D->P);
  D->P.veh_length_infl = c->veh_length_infl;
  for (int k = 0; k < 5; k++) {
    D->P.vec_le[k][0] = c->vec_le[k][0];
    D->P.vec_le[k][1] = c->vec_le[k][1];
  }
  fill_footprint_edges(D->P);
  // moving obstacles
  std::memset(&D->S, 0, sizeof(D->S));
  if (c->S > 0) {
    D->sur_off.resize(c->S + 1);
    D->sur_total.resize(c->S);
    D->sur_start.resize(c->S);
    D->sur_off[0] = 0;
    for (int u = 0; u < c->S; u++) {
      D->sur_off[u + 1] = D->sur_off[u] + c->sur[u].n_pieces;
      D->sur_total[u] = c->sur[u].duration;
      D->sur_start[u] = c->sur[u].start_time;
    }
    D->S.S = c->S;
    D->S.piece_off = D->sur_off.data();
    D->S.durations = c->sur_durs;
    D->S.coeffs = c->sur_coeffs;
    D->S.total = D->sur_total.data();
    D->S.start = D->sur_start.data();
    D->S.theta = nullptr; // the oracle walks the pieces as the reference does (the kernels search a threshold table)
  }
  D->opM.resize(L.M);
  D->opMT.resize(L.M);
  for (int sg = 0; sg < L.M; sg++) {
    int N = L.piece_nums[sg];
    D->opM[sg].resize((size_t)6 * N * (N + 5));
    D->opMT[sg].resize((size_t)6 * N * (N + 5));
    oracle_minco_operator(N, D->opM[sg].data());
    for (int r = 0; r < 6 * N; r++)
      for (int col = 0; col < N + 5; col++) D->opMT[sg][(size_t)col * 6 * N + r] = D->opM[sg][(size_t)r * (N + 5) + col];
  }
  D->seg.assign(L.M * 16, 0.0);
  D->stab.assign((size_t)L.M * 2 * (L.Kmax + 1), 0.0);
  D->rhs.assign(L.rhs_tot * 2, 0.0);
  D->b.assign(L.Ntot * 12, 0.0);
  D->c.assign(L.Ntot * 12, 0.0);
  D->gdC.assign(L.Ntot * 12, 0.0);
  D->adj.assign(L.rhs_tot * 2, 0.0);
  D->part.assign((size_t)8 * L.Npts, 0.0);
  D->pE.assign(L.Ntot, 0.0);
  D->pGsm.assign(L.Ntot, 0.0);
  D->pGdT.assign(L.Ntot, 0.0);
  D->pCost.assign(L.Ntot, 0.0);
  D->pChain.assign(L.Ntot, 0.0);
  c->dev = D;
}

extern "C" void oracle_dev_free(oracle_ctx *c) {
  delete static_cast<DevState *>(c->dev);
  c->dev = nullptr;
}

extern "C" double oracle_dev_eval(oracle_ctx *c, const double *x, double *g) {
  g_levels = levels_for(static_cast<DevState *>(c->dev)->L.n);
  return dev_eval(c, *static_cast<DevState *>(c->dev), x, g);
}

extern "C" void oracle_dev_coeffs(const oracle_ctx *c, double *coeffs, double *piece_dt) {
  const DevState &D = *static_cast<const DevState *>(c->dev);
  std::memcpy(coeffs, D.c.data(), sizeof(double) * 12 * D.L.Ntot);
  for (int sg = 0; sg < D.L.M; sg++) piece_dt[sg] = D.seg[sg * 16 + 1];
}

// lbfgs_optimize (lbfgs.hpp:440-751) + line_search_lewisoverton (lbfgs.hpp:276-390)
// with the kernel's reduction order for every dot product.
static const int kLoopBlock = 8; // stored pairs per block of the two-loop recursion (solver.hip)
static const int kBand = kLoopBlock - 1;

extern "C" void oracle_dev_solve(oracle_ctx *c, double *x, oracle_result *res) {
  DevState &D = *static_cast<DevState *>(c->dev);
  const DevParams &P = D.P;
  const int n = D.L.n, m = P.mem_size;
  g_levels = levels_for(n);
  std::vector<double> g(n), xp(n), gp(n), d(n), ys_h(m, 0.0), ri_h(m, 0.0), alpha_h(m, 0.0);
  std::vector<double> hS((size_t)m * n, 0.0), hY((size_t)m * n, 0.0);
  std::vector<double> hU((size_t)m * 8, 0.0), hV((size_t)m * 8, 0.0); // products with the kBand neighbouring pairs
  double pf[8];
  int evals = 0, k = 0, end = 0, bound = 0, ret = 0;
  long long hist_sum = 0;
  double step = 0.0;
  double fx = dev_eval(c, D, x, g.data());
  evals = 1;
  pf[0] = fx;
  bool done;
  {
    double gmax = absmax(g.data(), n), xmax = absmax(x, n);
    double dd = wave_dot(g.data(), g.data(), n);
    for (int e = 0; e < n; e++) d[e] = -g[e];
    done = (gmax / std::fmax(1.0, xmax) < P.g_epsilon);
    if (done) {
      ret = 0;
    } else {
      step = 1.0 / std::sqrt(dd);
      k = 1;
    }
  }
  while (!done) {
    double finit = fx;
    for (int e = 0; e < n; e++) {
      xp[e] = x[e];
      gp[e] = g[e];
    }
    double dginit = wave_dot(g.data(), d.data(), n);
    int ls = 0;
    bool ls_fail = false;
    if (!(step > 0.0)) {
      ls = DFTPAV_LBFGSERR_INVALIDPARAMETERS;
      ls_fail = true;
    } else if (0.0 < dginit) {
      ls = DFTPAV_LBFGSERR_INCREASEGRADIENT;
      ls_fail = true;
    }
    const double dgtest = P.f_dec_coeff * dginit;
    const double dstest = P.s_curv_coeff * dginit;
    int count = 0;
    bool brackt = false, touched = false;
    double mu = 0.0, nu = P.max_step;
    double stp = step;
    while (!ls_fail) {
      for (int e = 0; e < n; e++) x[e] = xp[e] + stp * d[e];
      fx = dev_eval(c, D, x, g.data());
      ++count;
      ++evals;
      if (std::isinf(fx) || std::isnan(fx)) {
        ls = DFTPAV_LBFGSERR_INVALID_FUNCVAL;
        break;
      }
      if (P.past > 0 && std::fabs(finit - fx) / (std::fabs(finit) + 1.0) < P.delta / P.past) {
        ls = count;
        break;
      }
      if (fx > finit + stp * dgtest) {
        nu = stp;
        brackt = true;
      } else {
        double gs = wave_dot(g.data(), d.data(), n);
        if (gs < dstest) {
          mu = stp;
        } else {
          ls = count;
          break;
        }
      }
      if (P.max_linesearch <= count) {
        ls = DFTPAV_LBFGSERR_MAXIMUMLINESEARCH;
        break;
      }
      if (brackt && (nu - mu) < P.machine_prec * nu) {
        ls = DFTPAV_LBFGSERR_WIDTHTOOSMALL;
        break;
      }
      if (brackt) stp = 0.5 * (mu + nu);
      else stp *= 2.0;
      if (stp < P.min_step) {
        ls = DFTPAV_LBFGSERR_MINIMUMSTEP;
        break;
      }
      if (stp > P.max_step) {
        if (touched) {
          ls = DFTPAV_LBFGSERR_MAXIMUMSTEP;
          break;
        }
        touched = true;
        stp = P.max_step;
      }
    }
    step = stp;
    if (ls < 0) {
      for (int e = 0; e < n; e++) {
        x[e] = xp[e];
        g[e] = gp[e];
      }
      ret = ls;
      break;
    }
    {
      double gmax = absmax(g.data(), n), xmax = absmax(x, n);
      if (gmax / std::fmax(1.0, xmax) < P.g_epsilon) {
        ret = DFTPAV_LBFGS_CONVERGENCE;
        break;
      }
    }
    if (0 < P.past) {
      if (P.past <= k) {
        double rate = std::fabs(pf[k % P.past] - fx) / std::fmax(1.0, std::fabs(fx));
        if (rate < P.delta) {
          ret = DFTPAV_LBFGS_STOP;
          break;
        }
      }
      pf[k % P.past] = fx;
    }
    if (P.max_iterations != 0 && P.max_iterations <= k) {
      ret = DFTPAV_LBFGSERR_MAXIMUMITERATION;
      break;
    }
    ++k;
    {
      double *sc = &hS[(size_t)end * n], *yc = &hY[(size_t)end * n];
      for (int e = 0; e < n; e++) {
        sc[e] = x[e] - xp[e];
        yc[e] = g[e] - gp[e];
        d[e] = -g[e];
      }
      double ys = wave_dot(yc, sc, n);
      double yy = wave_dot(yc, yc, n);
      double ss = wave_dot(sc, sc, n);
      double gpgp = wave_dot(gp.data(), gp.data(), n);
      ys_h[end] = ys;
      ri_h[end] = 1.0 / ys;
      double cau = ss * std::sqrt(gpgp) * P.cautious_factor;
      if (ys > cau) {
        ++bound;
        bound = m < bound ? m : bound;
        end = (end + 1) % m;
        if (n <= 64 && m >= kLoopBlock) { // the kernel's condition for the blocked form (a block must fit the ring)
          // Two-loop recursion (lbfgs.hpp:716-739) in blocks of kLoopBlock stored pairs, as the kernel runs
          // it: the kLoopBlock dot products of a block are taken against the direction as it stands at the
          // start of the block, and the effect of the block's earlier steps on a later dot product is
          // restored from the stored products s_j.y_k of neighbouring pairs,
          //     s_t.(d - sum_u alpha_u y_u) = s_t.d - sum_u alpha_u (s_t.y_u),
          // so that only one multiply-add of a step depends on the previous step: the division by ys_t is a product with
          // the stored 1 / ys_t, applied to s_t.d once and folded into the stored neighbour products.
          const int cs = (end + m - 1) % m; // slot of the pair just stored
          for (int dd = 0; dd < kBand && dd < bound - 1; dd++) { // products of the new y with the kBand pairs before it
            const int o = (cs + m - 1 - dd) % m;
            const double v = lane_dot(&hS[(size_t)o * n], yc, n);
            hU[(size_t)o * 8 + dd] = v * ri_h[o];   // s_o . y_(dd+1 pairs after o), over ys_o
            hV[(size_t)cs * 8 + dd] = v * ri_h[cs]; // y_cs . s_(dd+1 pairs before cs), over ys_cs
          }
          auto nth_older = [&](int t) { return (cs - t % m + m) % m; }; // slot of the t-th pair before the newest
          double al[kLoopBlock], dt[kLoopBlock];
          for (int t0 = 0; t0 < bound; t0 += kLoopBlock) { // first loop, newest -> oldest
            const int cnt = std::min(kLoopBlock, bound - t0);
            for (int q = 0; q < cnt; q++) dt[q] = lane_dot(&hS[(size_t)nth_older(t0 + q) * n], d.data(), n);
            for (int q = 0; q < cnt; q++) {
              const int j = nth_older(t0 + q);
              double acc = dt[q] * ri_h[j];
              for (int u = 0; u < q; u++) acc = __builtin_fma(-al[u], hU[(size_t)j * 8 + (q - u - 1)], acc);
              al[q] = acc;
              alpha_h[j] = al[q];
            }
            for (int q = 0; q < cnt; q++) {
              const double *yj = &hY[(size_t)nth_older(t0 + q) * n];
              for (int e = 0; e < n; e++) d[e] = __builtin_fma(-al[q], yj[e], d[e]);
            }
          }
          const double sc0 = ys / yy;
          for (int e = 0; e < n; e++) d[e] *= sc0;
          double be[kLoopBlock];
          for (int v0 = 0; v0 < bound; v0 += kLoopBlock) { // second loop, oldest -> newest
            const int cnt = std::min(kLoopBlock, bound - v0);
            for (int q = 0; q < cnt; q++) dt[q] = lane_dot(&hY[(size_t)nth_older(bound - 1 - (v0 + q)) * n], d.data(), n);
            for (int q = 0; q < cnt; q++) {
              const int j = nth_older(bound - 1 - (v0 + q));
              double acc = dt[q] * ri_h[j];
              for (int u = 0; u < q; u++) acc = __builtin_fma(al[u] - be[u], hV[(size_t)j * 8 + (q - u - 1)], acc);
              be[q] = acc;
              al[q] = alpha_h[j];
            }
            for (int q = 0; q < cnt; q++) {
              const double cf = al[q] - be[q];
              const double *sj = &hS[(size_t)nth_older(bound - 1 - (v0 + q)) * n];
              for (int e = 0; e < n; e++) d[e] = __builtin_fma(cf, sj[e], d[e]);
            }
          }
        } else {
          int j = end;
          for (int i = 0; i < bound; ++i) {
            j = (j + m - 1) % m;
            const double *sj = &hS[(size_t)j * n], *yj = &hY[(size_t)j * n];
            double a = wave_dot(sj, d.data(), n) / ys_h[j];
            alpha_h[j] = a;
            double na = -a;
            for (int e = 0; e < n; e++) d[e] += na * yj[e];
          }
          double sc0 = ys / yy;
          for (int e = 0; e < n; e++) d[e] *= sc0;
          for (int i = 0; i < bound; ++i) {
            const double *sj = &hS[(size_t)j * n], *yj = &hY[(size_t)j * n];
            double beta = wave_dot(yj, d.data(), n) / ys_h[j];
            double cf = alpha_h[j] - beta;
            for (int e = 0; e < n; e++) d[e] += cf * sj[e];
            j = (j + 1) % m;
          }
        }
        hist_sum += bound;
      }
    }
    step = 1.0;
  }
  res->final_cost = fx;
  res->status = ret;
  res->iters = k;
  res->evals = evals;
  res->hist_sum = hist_sum;
  int ok = (ret == 0 || ret == 1 || ret == 2 || ret == DFTPAV_LBFGSERR_MAXIMUMITERATION ||
            ret == DFTPAV_LBFGSERR_MAXIMUMLINESEARCH) ? 1 : 0;
  if (fx >= P.fail_cost) ok = 0;
  res->success = ok;
  c->evals = evals;
}
