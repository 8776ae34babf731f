// solver_ref.hip — the solve path in the REFERENCE'S OWN floating-point order (gfx950).
//
// solver.hip reassociates three things of the reference to run fast: the per-piece sums of the penalty gradient (cross-lane
// trees), the MINCO solves (dense operator instead of the banded substitution) and the dot products of L-BFGS (butterflies,
// blocked two-loop recursion).  Its results therefore equal the reference's only to rounding per evaluation, and the solver
// being chaotic whole solves equal the reference's only statistically.  This kernel keeps every sum in the
// order the reference executes it, so that a whole solve -- final x, cost, status, iterations, evaluations -- has the
// reference's BITS (tests/test_gpu_reference_order.py compares with oracle/_ref, the reference's sources compiled here):
//
//   * MinJerkOpt::generate / calGrads_PT: the banded forward / backward substitutions of BandedSystem::solve / solveAdj
//     (poly_traj_utils.hpp:805-852) row by row, each row's updates in the reference's order (a row is the unit: its six
//     multiply-subtract pairs are those the reference's column loops apply to it, in that order);
//   * addPVAGradCost2CT (traj_optimizer.cpp:486-705): every constraint point is evaluated by its own lane, term by term
//     (vertex x half-plane, velocity, acceleration, curvature left / right) exactly as written; what an ACTIVE term adds to
//     gdC (12 entries), gdT and the cost is parked in a record, and chain lanes -- one per (piece, entry), one for gdT, one
//     per cost -- add the records in the reference's sample -> vertex -> plane order;
//   * lbfgs_optimize / line_search_lewisoverton (lbfgs.hpp:276-390, 440-751): sequential dot products (the products are
//     formed by the lanes, the sum is one chain from the first element) and the plain two-loop recursion (:716-739);
//   * no fused multiply-add anywhere except inside a division by a stored reciprocal (div_by_rcp: equal to a / b on everything
//     it has been compared on -- an empirical claim, see there; DFTPAV_REF_EXACT_DIV=1 divides); contraction is off for the file.
//
// Scope: n <= 64 decision variables, at most 64 terms per constraint point (5 H + S + 4), any number of gear segments with or
// without moving obstacles (the reference's live call, traj_manager.cpp:604-610, installs both).  With ONE gear segment and no
// moving obstacles the reference's program has no libm call inside the loop and the device reproduces its bits.  With gear
// shifts the reference calls libm's cos / sin of the junction angles in every evaluation, with moving obstacles exp / log /
// pow per (point, obstacle) pair -- bits that belong to the host (glibc's are not correctly rounded, IFUNC-dispatched by CPU
// model, and gcc fuses cos + sin into sincos, which differs from both): the kernel uses the CORRECTLY ROUNDED functions
// (cr_trig.h), i.e. runs the reference's program with those calls defined instead of implemented (oracle order 2 is that
// program on the CPU).
//
// Two launch shapes, same bits (no sum depends on the shape):
//   * TEAM: one workgroup of 2-4 waves per trajectory -- the parallel stages spread over the waves, the serial ones (row sweeps,
//     L-BFGS) on wave 0.  The latency shape: few trajectories, each as fast as possible.
//   * WAVE: one WAVE per trajectory, eight of them in a workgroup that shares the sweep tables in LDS (the only big LDS item
//     that does not depend on the trajectory); every wave keeps its own < 17 KB of state, so a CU holds 8 trajectories that
//     all advance, against 3 in the TEAM shape.  A solve is a chain of dependent fp64 operations whichever way it is cut --
//     throughput is residency / latency -- so this is the throughput shape.  The waves of a workgroup never meet after the
//     tables are staged: each pops trajectories from the batch's ring (DevBatch::queue, as solver.hip's scheduled launch
//     does), runs one for a slice of iterations and puts it back unfinished, so that all trajectories advance together
//     and the launch ends without a tail of long solves on an empty device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#include "device_types.h"
#include "cr_trig.h"

#ifndef DFTPAV_REF_TL_ATTR
#define DFTPAV_REF_TL_ATTR __noinline__   // the two-loop recursion as a function of its own (see two_loop)
#endif
#ifndef DFTPAV_REF_EVAL_ATTR
#define DFTPAV_REF_EVAL_ATTR __forceinline__
#endif

#include "ref_order_common.h"

namespace dftpav {
namespace reford {

// How a launch lays out its LDS.  Shared by the waves of a workgroup: the sweep tables and the piece table (they depend on
// the layout only).  Per team (TEAM: the workgroup; WAVE: each wave): everything else.
// The width of the sequential sums: the smallest of the instantiated CAPs that holds the n decision variables.  A chain runs
// to CAP (lanes from n on contribute -0.0), so a CAP just above n matters: BASELINE configs[1] (8 + 8 pieces with a gear shift)
// has n = 33 -- 40 dependent additions per sum instead of 64.
__host__ __device__ inline int ref_cap_of(int n) { return n <= 16 ? 16 : (n <= 32 ? 32 : (n <= 40 ? 40 : (n <= 48 ? 48 : 64))); }

// WAVE shape: the kernels up to this width are built for 256 registers (two waves per SIMD, eight trajectories per CU), the wider
// ones take what they need (378-414 registers, no spills, four per CU).  Round 5, measured on configs[1]'s gear shift at 4096
// (n = 33: the 40-term kernel, profiles/r05_reference_order_baseline_batches.txt): 256 registers with 176 spilled against 414
// with none -- 450 against 467 ms, same bits: residency wins, by 3.7 %.  48 and 64 terms stay wide (not measured without
// moving obstacles; with them the kernel needs 512 either way).
#ifndef DFTPAV_REF_NARROW_CAP
#define DFTPAV_REF_NARROW_CAP 40
#endif
constexpr int kNarrowCap = DFTPAV_REF_NARROW_CAP;

struct Shape {
  int wave;     // 1: one wave per trajectory
  int cap;      // 16 / 32 / 64 >= n: width of the sequential sums (the kernel's CAP)
  int nl;       // doubles per solver vector in LDS
  int mw;       // 32-bit words of a point's term mask (1: up to 32 terms, 2: up to 64)
  int pf16;     // the running numbers of the active terms fit 16 bits
  int list_cap; // window of the chain pass (TEAM)
  int nrec;     // WAVE: term records (and their list entries) of an evaluation kept in LDS
};
__host__ __device__ inline Shape make_shape(const DevLayout &L, int S, bool wave) {
  Shape sh;
  sh.wave = wave ? 1 : 0;
  sh.cap = ref_cap_of(L.n);
  sh.nl = (L.n + 15) & ~15;
  const int nterm = 5 * L.H + S + 4;
  sh.mw = nterm > 32 ? 2 : 1;
  sh.pf16 = (long long)L.Npts * nterm <= 65535 ? 1 : 0;
  sh.list_cap = wave ? 0 : kListCapTeam;
  sh.nrec = wave ? kRecWave : 0;
  return sh;
}

struct Sm {
  ldsd_t x, xp, g, gp, d;   // [nl]
  ldsd_t bnd;               // [M][12] iniS [6], finS [6] of each gear segment as uploaded (clamped)
  ldsd_t pva;               // [M][12] head / tail position, velocity, acceleration in force for this x (junction overrides, traj_optimizer.cpp:273-282)
  ldsd_t trig;              // [M][2] cos, sin of the junction angles (M - 1 of them)
  ldsd_t seg;               // [M][16] 0:T 1:dt 2..7:t^k 8..13:t^-k
  ldsd_t spow;              // [M][2][Kmax+1] the running sample offsets (s1 += step) for K and Kd
  ldsd_t b, c, gdC, adj;    // [6 Ntot][2]
  ldsd_t pE, pG, pA;        // [Ntot] per-piece energy, d(energy)/dT, chain-rule term of calGrads_PT
  ldscd_t tab;              // (shared) per segment [4][6N][8]: rows of the four substitution sweeps (six coefficients, diagonal, 1 / diagonal)
  ldsd_t segsum;            // [M][gNUM] per segment: gdT, corridor cost, feasibility cost, jerk energy, moving-obstacle cost
  ldsd_t dot;               // [4][cap] products of up to four sequential dot products
  ldsd_t alpha;             // [mem]
  ldsd_t st;                // [sNUM]
  ldsi_t ist;               // [iNUM]
  ldsi_t pinfo;             // (shared) [Ntot][4] segment, piece index inside it, first constraint point, intervals K
  ldsi_t pmask;             // TEAM: [Npts][mw] active terms of a constraint point (bit t = term t)
  ldsi_t pfirst;            // TEAM: [Npts + 1] index of a point's first active term in (point, term) order (16-bit entries if pf16)
  ldsi_t list;              // TEAM: [list_cap] (point << 6 | term) of the active terms of the current window; WAVE: [2][nrec]: that
                            //   (bit 31 = the first moving-obstacle term of its point), then the term's piece
  ldsi_t pstart;            // WAVE: [Ntot + 1] number of active terms in front of a piece
  ldsd_t lrec;              // WAVE: [nrec][kRec] the first records of the evaluation, in (point, term) order
  int mw, pf16, list_cap, nrec;
  __device__ __forceinline__ mask_t mask(int pt) const {
    return mw == 2 ? ((mask_t)(unsigned)pmask[2 * pt] | ((mask_t)(unsigned)pmask[2 * pt + 1] << 32)) : (mask_t)(unsigned)pmask[pt];
  }
  __device__ __forceinline__ void set_mask(int pt, mask_t m) const {
    if (mw == 2) {
      pmask[2 * pt] = (int)(unsigned)(m & 0xffffffffull);
      pmask[2 * pt + 1] = (int)(unsigned)(m >> 32);
    } else {
      pmask[pt] = (int)(unsigned)m;
    }
  }
  __device__ __forceinline__ int first(int i) const { return pf16 ? (int)((ldsh_t)pfirst)[i] : pfirst[i]; }
  __device__ __forceinline__ void set_first(int i, int v) const {
    if (pf16) ((ldsh_t)pfirst)[i] = (unsigned short)v;
    else pfirst[i] = v;
  }
};
enum { gGDT = 0, gCOST0, gCOST2, gENERGY, gCOST1, gNUM = 6 };

// doubles of the sweep tables of all segments of a layout (pk_segment_doubles)
__host__ __device__ inline size_t table_doubles(const DevLayout &L) {
  size_t t = 0;
  for (int sg = 0; sg < L.M; sg++) t += (size_t)(384 + 88 * (L.piece_nums[sg] - 2));
  return t;
}
// bytes of the part of the LDS the waves of a workgroup share / of one team's part (both multiples of 16)
__host__ __device__ inline size_t lds_shared_bytes(const DevLayout &L) {
  return (table_doubles(L) * sizeof(double) + 4 * (size_t)L.Ntot * sizeof(int) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t lds_team_doubles(const DevLayout &L, int mem, const Shape &sh) {
  return 5 * (size_t)sh.nl + (size_t)L.M * (12 + 12 + 2 + 16 + gNUM) + 2 * (size_t)L.M * (L.Kmax + 1) + (4 * 12 + 3) * (size_t)L.Ntot + 4 * (size_t)sh.cap +
         (size_t)mem + sNUM + (size_t)sh.nrec * kRec;
}
__host__ __device__ inline size_t lds_team_ints(const DevLayout &L, const Shape &sh) {
  if (sh.wave) return iNUM + (size_t)L.Ntot + 1 + 2 * (size_t)sh.nrec;
  const size_t pf = sh.pf16 ? ((size_t)L.Npts + 2) / 2 : (size_t)L.Npts + 1;
  return iNUM + (size_t)sh.mw * L.Npts + pf + (size_t)sh.list_cap;
}
__host__ __device__ inline size_t lds_team_bytes(const DevLayout &L, int mem, const Shape &sh) {
  return (lds_team_doubles(L, mem, sh) * sizeof(double) + lds_team_ints(L, sh) * sizeof(int) + 15) & ~(size_t)15;
}

// shared: start of the workgroup's LDS; team: start of this team's part
__device__ inline void carve(Sm &s, double *shared, double *team, const DevLayout &L, int mem, const Shape &sh) {
  const int M = L.M, Ntot = L.Ntot;
  s.tab = (ldscd_t)shared;
  s.pinfo = (ldsi_t)((ldsd_t)shared + table_doubles(L));
  ldsd_t p = (ldsd_t)team;
  s.x = p; p += sh.nl;
  s.xp = p; p += sh.nl;
  s.g = p; p += sh.nl;
  s.gp = p; p += sh.nl;
  s.d = p; p += sh.nl;
  s.bnd = p; p += 12 * M;
  s.pva = p; p += 12 * M;
  s.trig = p; p += 2 * M;
  s.seg = p; p += 16 * M;
  s.spow = p; p += 2 * M * (L.Kmax + 1);
  s.b = p; p += 12 * Ntot;
  s.c = p; p += 12 * Ntot;
  s.gdC = p; p += 12 * Ntot;
  s.adj = p; p += 12 * Ntot;
  s.pE = p; p += Ntot;
  s.pG = p; p += Ntot;
  s.pA = p; p += Ntot;
  s.segsum = p; p += gNUM * M;
  s.dot = p; p += 4 * sh.cap;
  s.alpha = p; p += mem;
  s.st = p; p += sNUM;
  s.lrec = p; p += (size_t)sh.nrec * kRec;
  ldsi_t q = (ldsi_t)p;
  s.ist = q; q += iNUM;
  if (sh.wave) {
    s.pstart = q; q += L.Ntot + 1;
    s.list = q;
    s.pmask = q; // (unused in this shape)
    s.pfirst = q;
  } else {
    s.pmask = q; q += sh.mw * L.Npts;
    s.pfirst = q; q += sh.pf16 ? (L.Npts + 2) / 2 : L.Npts + 1;
    s.list = q;
    s.pstart = q; // (unused in this shape)
  }
  s.mw = sh.mw;
  s.pf16 = sh.pf16;
  s.list_cap = sh.list_cap;
  s.nrec = sh.nrec;
}

// A team's barrier: the workgroup's in the TEAM shape; in the WAVE shape the team is one wave, whose LDS operations execute
// in program order -- all that is needed is that the compiler keeps them in that order.
template <bool WAVE> __device__ __forceinline__ void team_sync() {
  if (WAVE) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  else __syncthreads();
}

// ------------------------------------------------ costFunctionCallback (traj_optimizer.cpp:206-350)
// x -> g (LDS), f in st[sF].  rec_b: this trajectory's term records [Npts][nterm][kRec].  The gear segments are independent
// up to the sums of :292-297 and the junction variables' gradients (:307-320), so every stage runs them side by side.
template <bool SUR, bool WAVE, int HMAX = 5>
__device__ DFTPAV_REF_EVAL_ATTR void ref_eval(const DevBatch &D, gcd_t cor_b, gd_t rec_b, const Sm &sm, ldscd_t x, ldsd_t g, Prof &pr) {
  const DevLayout &L = D.L;
  const DevParams &P = D.P;
  // the team: the workgroup, or (WAVE) this wave alone.  Stages with two independent jobs give the second one to the lanes of
  // wave 1 in a workgroup and to the same lanes, afterwards, in a lone wave (u2: the index inside the second job).
  const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, T = WAVE ? 64 : (int)blockDim.x;
  const int u2 = WAVE ? tid : tid - 64;
  const int M = L.M, Ntot = L.Ntot, Npts = L.Npts, H = L.H, nS = SUR ? D.sur.S : 0, nterm = 5 * H + nS + 4, Kmax1 = L.Kmax + 1;

  // ---- durations (VirtualT2RealT, :371-379), their powers (poly_traj_utils.hpp:961-966); cos / sin of the junction angles
  if (tid < M) {
    const int sg = tid;
    const double vt = x[L.x_tau0 + sg];
    const double Tr = vt > 0.0 ? ((0.5 * vt + 1.0) * vt + 1.0) + P.mini_T : 1.0 / ((0.5 * vt - 1.0) * vt + 1.0) + P.mini_T;
    int N = 0;
    for (int q = 0; q < M; q++) N = q == sg ? L.piece_nums[q] : N;
    const double t1 = Tr / N, t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
    ldsd_t se = sm.seg + 16 * sg;
    se[0] = Tr;
    se[1] = t1;
    se[2] = 1.0; se[3] = t1; se[4] = t2; se[5] = t3; se[6] = t4; se[7] = t5;
    se[8] = 1.0 / 1.0; se[9] = 1.0 / t1; se[10] = 1.0 / t2; se[11] = 1.0 / t3; se[12] = 1.0 / t4; se[13] = 1.0 / t5;
  }
  if (u2 >= 0 && u2 < M - 1) {
    const int i = u2;
    double sn, cs;
    crt::sincos(x[L.x_ang0 + i], sn, cs); // the reference: libm's cos / sin (host-dependent bits); here the correctly rounded ones
    sm.trig[2 * i] = cs;
    sm.trig[2 * i + 1] = sn;
  }
  team_sync<WAVE>();
  // ---- boundary states in force (IniS / FinS of :270-282): junction position from x, junction velocity from the angle
  if (tid < M) {
    const int sg = tid;
    ldscd_t ini = sm.bnd + 12 * sg, fin = ini + 6;
    ldsd_t hv = sm.pva + 12 * sg, tv = hv + 6;
    for (int q = 0; q < 6; q++) {
      hv[q] = ini[q];
      tv[q] = fin[q];
    }
    if (sg > 0) {
      hv[0] = x[L.x_gear0 + 2 * (sg - 1)];
      hv[1] = x[L.x_gear0 + 2 * (sg - 1) + 1];
      hv[2] = -P.non_sinv * sm.trig[2 * (sg - 1)];
      hv[3] = -P.non_sinv * sm.trig[2 * (sg - 1) + 1];
    }
    if (sg < M - 1) {
      tv[0] = x[L.x_gear0 + 2 * sg];
      tv[1] = x[L.x_gear0 + 2 * sg + 1];
      tv[2] = P.non_sinv * sm.trig[2 * sg];
      tv[3] = P.non_sinv * sm.trig[2 * sg + 1];
    }
  }
  team_sync<WAVE>();
  // ---- right-hand sides (poly_traj_utils.hpp:968-977)
  for (int w = tid; w < 12 * Ntot; w += T) {
    const int p = w / 12, q = w - 12 * p, k = q >> 1, d = q & 1;
    const int sg = sm.pinfo[4 * p], lp = sm.pinfo[4 * p + 1];
    int N = 0, x0 = 0;
    for (int q2 = 0; q2 < M; q2++) {
      N = q2 == sg ? L.piece_nums[q2] : N;
      x0 = q2 == sg ? L.seg_x0[q2] : x0;
    }
    const double t1 = sm.seg[16 * sg + 3], t2 = sm.seg[16 * sg + 4];
    ldscd_t hv = sm.pva + 12 * sg, tv = hv + 6;
    double v = 0.0;
    if (lp == 0 && k < 3) v = k == 0 ? hv[d] : (k == 1 ? hv[2 + d] * t1 : hv[4 + d] * t2);
    else if (lp == N - 1 && k >= 3) v = k == 3 ? tv[d] : (k == 4 ? tv[2 + d] * t1 : tv[4 + d] * t2);
    else if (k == 5) v = x[x0 + 2 * lp + d];
    sm.b[w] = v;
  }
  team_sync<WAVE>();
  // ---- BandedSystem::solve (poly_traj_utils.hpp:805-826), one lane per (segment, dimension); second job: the running sample
  // offsets s1 += step (traj_optimizer.cpp:513), one lane per table
  if (tid < 2 * M) {
    const int sg = tid >> 1, d = tid & 1;
    int N = 0, p0 = 0;
    for (int q = 0; q < M; q++) {
      N = q == sg ? L.piece_nums[q] : N;
      p0 = q == sg ? L.seg_piece0[q] : p0;
    }
    int toff = 0;
    for (int q = 0; q < M; q++) toff += q < sg ? pk_segment_doubles(L.piece_nums[q]) : 0;
    ldscd_t tb = sm.tab + toff;
    sweep<0>(tb, sm.b + 12 * p0, 6 * N, d);
    sweep<1>(tb + pk_sweep_offset(1, N), sm.b + 12 * p0, 6 * N, d);
  }
  if (u2 >= 0 && u2 < 2 * M) {
    const int sg = u2 >> 1, which = u2 & 1;
    const int K = which ? L.Kd : L.K;
    const double step = sm.seg[16 * sg + 1] / K;
    ldsd_t tab = sm.spow + (2 * sg + which) * Kmax1;
    double s1 = 0.0;
    for (int j = 0; j <= K; j++) {
      tab[j] = s1;
      s1 += step;
    }
    if (which == 0 && nS > 0) { // start time of every piece: the running sum t += dt of traj_optimizer.cpp:775 (pA is free until calGrads_PT)
      int p0 = 0, p1 = 0;
      for (int q = 0; q < M; q++) {
        p0 = q == sg ? L.seg_piece0[q] : p0;
        p1 = q == sg ? L.seg_piece0[q + 1] : p1;
      }
      double tt = 0.0;
      for (int i = p0; i < p1; i++) {
        sm.pA[i] = tt;
        tt += sm.seg[16 * sg + 1];
      }
    }
  }
  team_sync<WAVE>();
  pr.tick(0);
  // ---- c = b * tInv (:979-984)
  for (int w = tid; w < 12 * Ntot; w += T) {
    const int p = w / 12, k = (w - 12 * p) >> 1;
    sm.c[w] = sm.b[w] * sm.seg[16 * sm.pinfo[4 * p] + 8 + k];
  }
  team_sync<WAVE>();
  // ---- initSmGradCost / getTrajJerkCost per piece (poly_traj_utils.hpp:998-1035); the sums over the pieces are chained below
  for (int i = tid; i < Ntot; i += T) {
    ldscd_t c = sm.c + 12 * i;
    ldscd_t t = sm.seg + 16 * sm.pinfo[4 * i] + 2;
    const double n33 = c[6] * c[6] + c[7] * c[7], n44 = c[8] * c[8] + c[9] * c[9], n55 = c[10] * c[10] + c[11] * c[11];
    const double d43 = c[8] * c[6] + c[9] * c[7], d53 = c[10] * c[6] + c[11] * c[7], d54 = c[10] * c[8] + c[11] * c[9];
    sm.pE[i] = 36.0 * n33 * t[1] + 144.0 * d43 * t[2] + 192.0 * n44 * t[3] + 240.0 * d53 * t[3] + 720.0 * d54 * t[4] + 720.0 * n55 * t[5];
    sm.pG[i] = 36.0 * n33 + 288.0 * d43 * t[1] + 576.0 * n44 * t[2] + 720.0 * d53 * t[2] + 2880.0 * d54 * t[3] + 3600.0 * n55 * t[4];
    ldsd_t gc = sm.gdC + 12 * i;
    for (int d = 0; d < 2; d++) {
      const double c3 = c[6 + d], c4 = c[8 + d], c5 = c[10 + d];
      gc[10 + d] = 240.0 * c3 * t[3] + 720.0 * c4 * t[4] + 1440.0 * c5 * t[5];
      gc[8 + d] = 144.0 * c3 * t[2] + 384.0 * c4 * t[3] + 720.0 * c5 * t[4];
      gc[6 + d] = 72.0 * c3 * t[1] + 144.0 * c4 * t[2] + 240.0 * c5 * t[3];
      gc[d] = 0.0;
      gc[2 + d] = 0.0;
      gc[4 + d] = 0.0;
    }
  }
  pr.tick(1);
  if (WAVE) {
  // ================= WAVE shape: tests, numbering and records in one pass over the points; chains from LDS
  // The points are taken 64 at a time IN ORDER, so the running number of active terms is known at the end of every round: a
  // lane numbers its point's active terms (exclusive prefix over the wave + the running base) and writes each record straight
  // to its place in (point, term) order -- the first `nrec` of an evaluation in LDS (an evaluation of BASELINE configs[2] has
  // 24 active terms on average), the rest in global scratch.  No mask table, no second pass.
  const int nrec = sm.nrec, tS0 = 5 * H, t0 = tS0 + nS;
  const gd_t stage_b = rec_b + (size_t)Npts * nterm * kRec;                         // moving obstacles: a point's records as surround_terms leaves them
  int *glist = reinterpret_cast<int *>((double *)(stage_b + (size_t)Npts * nS * kRec)); // entries beyond the LDS window
  int base = 0;
  // (requesting the half-planes one round ahead was tried twice: 40 more live registers; round 4: 100 -> 116 k cycles per evaluation,
  // round 5 on the DPP-chain kernel: 92.6 -> 112.7 k, 310 -> 317 ms per 4096 -- docs/HISTORY.md)
  for (int r0 = 0; r0 < Npts; r0 += 64) {
    const int pt = r0 + tid;
    const bool in = pt < Npts;
    mask_t m = 0ull;
    PtState st;
    int p = 0, j = 1;
    if (in) {
      double pl[20];
      load_planes(cor_b + pt, (size_t)D.NptsPad, H, pl);
      p = D.pt_piece[pt];
      j = D.pt_j[pt];
      const int sg = sm.pinfo[4 * p], lp = sm.pinfo[4 * p + 1], K = sm.pinfo[4 * p + 3];
      int N = 0, singul_ = 1;
      for (int q = 0; q < M; q++) {
        N = q == sg ? L.piece_nums[q] : N;
        singul_ = q == sg ? L.singuls[q] : singul_;
      }
      const bool edge = lp == 0 || lp == N - 1;
      double cc[12];
#pragma unroll
      for (int k = 0; k < 12; k++) cc[k] = sm.c[12 * p + k];
      const double step = sm.seg[16 * sg + 1] / K;
      const double s1 = sm.spow[(2 * sg + (edge ? 1 : 0)) * Kmax1 + j];
      const double trajtime = (SUR && sg > 0) ? sm.seg[16 * (sg - 1)] : 0.0;
      m = point_masks<SUR>(P, cc, lp, N, j, K, step, s1, singul_, D.epis, H, pl, stage_b + (size_t)pt * nS * kRec, D.sur, D.t_now, SUR ? sm.pA[p] : 0.0, sg,
                           trajtime, st);
    }
    const int c = __builtin_popcountll(m);
    if (__builtin_amdgcn_ballot_w64(c != 0) == 0ull) { // (uniform) nothing active in this round
      if (in && j == 0) sm.pstart[p] = base;
      continue;
    }
    const int incl = wave_incl_scan_i32(c);
    int e = base + incl - c;
    if (in && j == 0) sm.pstart[p] = e; // a piece's terms start where its first point's do
    if (SUR && c != 0) __threadfence_block(); // this lane reads its moving-obstacle records back below
    bool sur_seen = false;
    for (mask_t mm = m; mm;) {
      const int t = __builtin_ctzll(mm);
      mm &= mm - 1;
      const bool sur_term = t >= tS0 && t < t0;
      int entry = (pt << 6) | t;
      if (sur_term && !sur_seen) entry |= (int)0x80000000u; // carries the point's moving-obstacle penalty (costs(1) += once per point)
      sur_seen = sur_seen || sur_term;
      double *r_ = e < nrec ? (double *)(sm.lrec + (size_t)e * kRec) : (double *)(rec_b + (size_t)e * kRec);
      if (e < nrec) {
        sm.list[e] = entry;
        sm.list[nrec + e] = p; // its piece
      } else {
        glist[2 * e] = entry;
        glist[2 * e + 1] = p;
      }
      if (sur_term) {
        const gcd_t src = (gcd_t)(stage_b + ((size_t)pt * nS + (t - tS0)) * kRec);
#pragma unroll
        for (int q = 0; q < kRec; q++) r_[q] = src[q];
      } else {
        point_emit(P, st, t, H, t0, cor_b + pt, (size_t)D.NptsPad, r_);
      }
      e++;
    }
    base += __builtin_amdgcn_readlane(incl, 63);
  }
  if (tid == 0) sm.pstart[Ntot] = base;
  // the start values of the per-segment chains (`gdT +=`, `energy +=` over the pieces in order, from 0.0)
  if (tid < M) {
    const int sg = tid;
    int p0 = 0, p1 = 0;
    for (int q = 0; q < M; q++) {
      p0 = q == sg ? L.seg_piece0[q] : p0;
      p1 = q == sg ? L.seg_piece0[q + 1] : p1;
    }
    double gdT = 0.0, en = 0.0;
    for (int i = p0; i < p1; i++) {
      gdT += sm.pG[i];
      en += sm.pE[i];
    }
    sm.segsum[gNUM * sg + gGDT] = gdT;
    sm.segsum[gNUM * sg + gENERGY] = en;
    sm.segsum[gNUM * sg + gCOST0] = 0.0;
    sm.segsum[gNUM * sg + gCOST2] = 0.0;
    sm.segsum[gNUM * sg + gCOST1] = 0.0;
  }
  __threadfence_block(); // records beyond the LDS window went to global memory
  team_sync<WAVE>();
  pr.tick(2);
  pr.count(9, base); // active terms of this evaluation (a count, not cycles)
  pr.count(10, base > nrec ? 1 : 0);   // evaluations whose records do not all fit the LDS window ...
  pr.count(11, base > nrec ? base : 0); // ... and their active terms
  // ---- chains: lane (piece, entry) adds its piece's records in order; four more lanes per segment walk all of the segment's
  // for gdT, the corridor cost, the feasibility cost and the moving-obstacle cost (they go first: theirs are the long walks)
  if (base > 0 && base <= kSerialMax) {
    // Up to a few hundred active terms (nearly every evaluation; most have all their records in LDS): ONE pass over the terms in
    // order on 16 lanes -- lane q < 12 carries entry q of gdC of the piece the terms belong to, lane 12 that segment's gdT,
    // lane 13 its three costs -- each term one read and one addition per lane, the reads of sixteen terms in flight (the ones
    // beyond the LDS window come from L2: one round trip per sixteen terms for all the sums together); a change of piece
    // (segment) stores the sums and fetches the next piece's (segment's).  Same chains as the lanes per (piece, entry) below,
    // term after term.
    if (tid < 16) {
      constexpr int kU = 16;
      int curp = -1, cursg = -1, Nseg = 0;
      double acc = 0.0, c2 = 0.0, c1 = 0.0; // lane < 12: gdC entry; 12: gdT; 13: corridor cost (acc), feasibility (c2), moving obstacles (c1)
      for (int e0 = 0; e0 < base; e0 += kU) {
        int ent[kU], pc[kU];
        double v[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const int e = e0 + u < base ? e0 + u : base - 1;
          if (e < nrec) { // (uniform)
            ent[u] = sm.list[e];
            pc[u] = sm.list[nrec + e];
            v[u] = sm.lrec[(size_t)e * kRec + tid];
          } else {
            ent[u] = glist[2 * e];
            pc[u] = glist[2 * e + 1];
            v[u] = rec_b[(size_t)e * kRec + tid];
          }
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
          if (e0 + u >= base) break; // uniform
          const int p = pc[u];
          if (p != curp) { // uniform: the terms come piece after piece, segment after segment
            const int sg = sm.pinfo[4 * p];
            if (tid < 12) {
              if (curp >= 0) sm.gdC[12 * curp + tid] = acc;
              acc = sm.gdC[12 * p + tid];
            } else if (sg != cursg) {
              if (cursg >= 0) {
                if (tid == 12) sm.segsum[gNUM * cursg + gGDT] = acc;
                if (tid == 13) {
                  sm.segsum[gNUM * cursg + gCOST0] = acc;
                  sm.segsum[gNUM * cursg + gCOST2] = c2;
                  sm.segsum[gNUM * cursg + gCOST1] = c1;
                }
              }
              if (tid == 12) acc = sm.segsum[gNUM * sg + gGDT];
              if (tid == 13) {
                acc = sm.segsum[gNUM * sg + gCOST0];
                c2 = sm.segsum[gNUM * sg + gCOST2];
                c1 = sm.segsum[gNUM * sg + gCOST1];
              }
            }
            if (sg != cursg) {
              Nseg = 0;
              for (int q2 = 0; q2 < M; q2++) Nseg = q2 == sg ? L.piece_nums[q2] : Nseg;
            }
            curp = p;
            cursg = sg;
          }
          const int t = ent[u] & 63;
          const bool sur_term = t >= tS0 && t < t0;
          if (tid < 12) {
            acc += v[u];
          } else if (tid == 12) { // gdT: one `+=` per term; a moving-obstacle term: three, and one more per previous segment (traj_optimizer.cpp:1663-1676)
            acc += v[u];
            if (sur_term) {
              const int e = e0 + u;
              const double vb = e < nrec ? sm.lrec[(size_t)e * kRec + 14] : rec_b[(size_t)e * kRec + 14];
              const double vc = e < nrec ? sm.lrec[(size_t)e * kRec + 15] : rec_b[(size_t)e * kRec + 15];
              acc += vb * sm.pinfo[4 * p + 1]; // * pieceid
              acc += vc;
              const double prev = vb * Nseg; // * piece_num_container[trajid]
              for (int idx = 0; idx < cursg; idx++) acc += prev;
            }
          } else if (tid == 13) {
            if (t < tS0) acc += v[u];
            else if (t >= t0) c2 += v[u];
            else if (ent[u] < 0) c1 += v[u]; // costs(1) += the point's penalty, once per point: carried by its first active obstacle term
          }
        }
      }
      if (tid < 12) sm.gdC[12 * curp + tid] = acc;
      if (tid == 12) sm.segsum[gNUM * cursg + gGDT] = acc;
      if (tid == 13) {
        sm.segsum[gNUM * cursg + gCOST0] = acc;
        sm.segsum[gNUM * cursg + gCOST2] = c2;
        sm.segsum[gNUM * cursg + gCOST1] = c1;
      }
    }
    team_sync<WAVE>();
  } else if (base > 0) {
    const int n_chain = 4 * M + 12 * Ntot;
    for (int w = tid; w < n_chain; w += 64) {
      int e0, e1, q, kind = -1, csg = 0;
      ldsd_t dst;
      if (w >= 4 * M) {
        const int p = (w - 4 * M) / 12;
        q = (w - 4 * M) - 12 * p;
        e0 = sm.pstart[p];
        e1 = sm.pstart[p + 1];
        dst = sm.gdC + (w - 4 * M);
      } else {
        const int sg = w >> 2, kd = w & 3; // per segment: 0 gdT, 1 corridor cost, 2 feasibility cost, 3 moving-obstacle cost
        int a0 = 0, a1 = 0;
        for (int q2 = 0; q2 < M; q2++) {
          a0 = q2 == sg ? L.seg_piece0[q2] : a0;
          a1 = q2 == sg ? L.seg_piece0[q2 + 1] : a1;
        }
        e0 = sm.pstart[a0];
        e1 = sm.pstart[a1];
        q = kd == 0 ? 12 : 13;
        dst = sm.segsum + gNUM * sg + (kd == 0 ? gGDT : (kd == 1 ? gCOST0 : (kd == 2 ? gCOST2 : gCOST1)));
        kind = kd;
        csg = sg;
      }
      if (e1 <= e0) continue;
      double acc = *dst;
      if (kind < 0) { // an entry of gdC: every term of the piece, in order
        int e = e0;
        const int eL = e1 < nrec ? e1 : nrec; // the part in LDS
        for (; e < eL; e++) acc += sm.lrec[(size_t)e * kRec + q];
        for (; e + 8 <= e1; e += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = rec_b[(size_t)(e + u) * kRec + q];
#pragma unroll
          for (int u = 0; u < 8; u++) acc += v[u];
        }
        for (; e < e1; e++) acc += rec_b[(size_t)e * kRec + q];
      } else {
        int Nseg = 0;
        for (int q2 = 0; q2 < M; q2++) Nseg = q2 == csg ? L.piece_nums[q2] : Nseg;
        for (int eb = e0; eb < e1; eb += 8) { // what the next eight terms add is requested in front of the additions
          int ent[8];
          double va8[8], vb8[8], vc8[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int e = eb + u < e1 ? eb + u : e1 - 1;
            if (e < nrec) {
              ent[u] = sm.list[e];
              ldscd_t r_ = sm.lrec + (size_t)e * kRec;
              va8[u] = r_[q]; vb8[u] = r_[14]; vc8[u] = r_[15];
            } else {
              ent[u] = glist[2 * e];
              gcd_t r_ = (gcd_t)(rec_b + (size_t)e * kRec);
              va8[u] = r_[q]; vb8[u] = r_[14]; vc8[u] = r_[15];
            }
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
          if (eb + u >= e1) break;
          const int entry = ent[u];
          const double va = va8[u], vb = vb8[u], vc = vc8[u];
          const int t = entry & 63;
          const bool sur_term = t >= tS0 && t < t0;
          if (kind == 0) { // gdT: one `+=` per term; a moving-obstacle term: three, and one more per previous segment (traj_optimizer.cpp:1663-1676)
            acc += va;
            if (sur_term) {
              const int lp = sm.pinfo[4 * (int)D.pt_piece[(entry >> 6) & 0x1ffffff] + 1]; // pieceid
              acc += vb * lp;
              acc += vc;
              const double prev = vb * Nseg; // ... * piece_num_container[trajid]
              for (int idx = 0; idx < csg; idx++) acc += prev;
            }
          } else if (kind == 1) {
            if (t < tS0) acc += va;
          } else if (kind == 2) {
            if (t >= t0) acc += va;
          } else if (sur_term && entry < 0) { // costs(1) += the point's penalty, once per point: carried by its first active obstacle term
            acc += va;
          }
          }
        }
      }
      *dst = acc;
    }
    team_sync<WAVE>();
  }
  } else {
  // ================= TEAM shape
  // ---- the constraint points, each on a lane of its own
  // what a point needs from its piece and segment
  auto point_ctx = [&](int pt, int &p, int &j, int &sg, int &lp, int &K, int &N, int &singul_, double (&cc)[12], double &step, double &s1, double &trajtime) {
    p = D.pt_piece[pt];
    j = D.pt_j[pt];
    sg = sm.pinfo[4 * p];
    lp = sm.pinfo[4 * p + 1];
    K = sm.pinfo[4 * p + 3];
    N = 0;
    singul_ = 1;
    for (int q = 0; q < M; q++) {
      N = q == sg ? L.piece_nums[q] : N;
      singul_ = q == sg ? L.singuls[q] : singul_;
    }
    const bool edge = lp == 0 || lp == N - 1;
#pragma unroll
    for (int k = 0; k < 12; k++) cc[k] = sm.c[12 * p + k];
    step = sm.seg[16 * sg + 1] / K;
    s1 = sm.spow[(2 * sg + (edge ? 1 : 0)) * Kmax1 + j];
    // trajtimes[sg] of traj_optimizer.cpp:230-234: 0, then the real duration of the PREVIOUS segment
    trajtime = (SUR && sg > 0) ? sm.seg[16 * (sg - 1)] : 0.0;
  };
  if (SUR && nS > 0) {
    // Moving obstacles: a (point, obstacle) pair that reaches its forty correctly rounded exponentials costs 40 k cycles, and one such
    // lane holds its whole wave.  So the pairs are COLLECTED first -- every point runs its static tests and, per obstacle, the cheap tests
    // (hull box, distance gate, the bound before any exponential) -- and then evaluated densely packed, one pair per lane; a pair's record
    // has a place of its own, so who evaluates it changes nothing, and the point's penalty (the reference's sum over the obstacles in
    // order) is formed afterwards from the pairs' shares.  (round 6: configs[4] at 1024 1.90 -> see docs/HISTORY.md)
    const int tS0 = 5 * H, cap = sm.list_cap;
    int *npairs = (int *)(sm.ist + 15);
    if (tid == 0) *npairs = 0;
    team_sync<WAVE>();
    // a pair, evaluated in full by whoever holds it: the point's state is formed again from scratch (the same expressions, the same bits)
    auto pair_eval = [&](int pt, int sur_id) {
      int p, j, sg, lp, K, N, singul_;
      double cc[12], step, s1, trajtime;
      point_ctx(pt, p, j, sg, lp, K, N, singul_, cc, step, s1, trajtime);
      PtState st;
      double pl[4 * HMAX];
      load_planes<HMAX>(cor_b + pt, (size_t)D.NptsPad, H, pl);
      (void)point_masks<SUR, HMAX, true>(P, cc, lp, N, j, K, step, s1, singul_, D.epis, H, pl, (gd_t) nullptr, D.sur, D.t_now, sm.pA[p], sg, trajtime, st);
      const gd_t rec_pt = rec_b + (size_t)pt * nterm * kRec;
      double pen;
      if (point_surround_one<false>(P, D.sur, st, sur_id, D.t_now, j, lp, N, sm.pA[p], singul_, sg, trajtime, tS0, rec_pt, pen)) {
        rec_pt[(size_t)(tS0 + sur_id) * kRec + 13] = pen; // (the pair's share; the point's sum is formed below)
        const int bit = tS0 + sur_id;
        atomicOr((int *)(sm.pmask + (sm.mw == 2 ? 2 * pt + (bit >> 5) : pt)), 1 << (bit & 31));
      }
    };
    for (int pt = tid; pt < Npts; pt += T) {
      int p, j, sg, lp, K, N, singul_;
      double cc[12], step, s1, trajtime;
      point_ctx(pt, p, j, sg, lp, K, N, singul_, cc, step, s1, trajtime);
      PtState st;
      st.K = -1; // (stays -1 for a point the reference skips, traj_optimizer.cpp:550-553)
      double pl[4 * HMAX];
      load_planes<HMAX>(cor_b + pt, (size_t)D.NptsPad, H, pl);
      const mask_t mask = point_masks<SUR, HMAX, true>(P, cc, lp, N, j, K, step, s1, singul_, D.epis, H, pl, (gd_t) nullptr, D.sur, D.t_now, sm.pA[p], sg, trajtime, st);
      const gd_t rec_pt = rec_b + (size_t)pt * nterm * kRec;
      for (mask_t mm = mask; mm;) {
        const int t = __builtin_ctzll(mm);
        mm &= mm - 1;
        point_emit(P, st, t, H, tS0 + nS, cor_b + pt, (size_t)D.NptsPad, rec_pt + (size_t)t * kRec);
      }
      sm.set_mask(pt, mask);
      if (st.K >= 0) {
        for (int sur_id = 0; sur_id < nS; sur_id++) {
          double pen;
          if (!point_surround_one<true>(P, D.sur, st, sur_id, D.t_now, j, lp, N, sm.pA[p], singul_, sg, trajtime, tS0, rec_pt, pen)) continue;
          const int slot = atomicAdd(npairs, 1);
          if (slot < cap) sm.list[slot] = (pt << 4) | sur_id;
          else pair_eval(pt, sur_id); // (more pairs than the list holds: this one at once)
        }
      }
    }
    __threadfence_block();
    team_sync<WAVE>();
    const int np = *npairs < cap ? *npairs : cap;
    for (int q = tid; q < np; q += T) pair_eval(sm.list[q] >> 4, sm.list[q] & 15);
    __threadfence_block(); // the pairs' records and shares are read by other lanes below
    team_sync<WAVE>();
    // the point's penalty: the shares of its active obstacles added in obstacle order from 0.0, kept in slot [13] of the first one
    for (int pt = tid; pt < Npts; pt += T) {
      const mask_t m = sm.mask(pt);
      const unsigned sb = (unsigned)((m >> tS0) & (((mask_t)1 << nS) - 1ull));
      if (!sb) continue;
      const gd_t rec_pt = rec_b + (size_t)pt * nterm * kRec;
      double total = 0.0;
      for (int sur_id = 0; sur_id < nS; sur_id++)
        if (sb & (1u << sur_id)) {
          total += rec_pt[(size_t)(tS0 + sur_id) * kRec + 13];
          rec_pt[(size_t)(tS0 + sur_id) * kRec + 13] = 0.0;
        }
      rec_pt[(size_t)(tS0 + __builtin_ctz(sb)) * kRec + 13] = total;
    }
  } else {
    for (int pt = tid; pt < Npts; pt += T) {
      int p, j, sg, lp, K, N, singul_;
      double cc[12], step, s1, trajtime;
      point_ctx(pt, p, j, sg, lp, K, N, singul_, cc, step, s1, trajtime);
      sm.set_mask(pt, point_terms<SUR, HMAX>(P, cc, lp, N, j, K, step, s1, singul_, D.epis, H, cor_b + pt, (size_t)D.NptsPad,
                                             rec_b + (size_t)pt * nterm * kRec, D.sur, D.t_now, SUR ? sm.pA[p] : 0.0, sg, trajtime));
    }
  }
  __threadfence_block(); // the records are read back by other lanes of this team
  team_sync<WAVE>();
  pr.tick(2);
  // ---- number the active terms in (point, term) order: exclusive prefix sum of the counts (wave 0); second job: the start
  // values of the per-segment chains (`gdT +=`, `energy +=` over the pieces in order, from 0.0)
  if (tid < 64) {
    const int per = (Npts + 63) >> 6, start = tid * per;
    int sum = 0;
    for (int i = start; i < start + per && i < Npts; i++) sum += __builtin_popcountll(sm.mask(i));
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (tid >= o) incl += up;
    }
    int run = incl - sum;
    for (int i = start; i < start + per && i < Npts; i++) {
      sm.set_first(i, run);
      run += __builtin_popcountll(sm.mask(i));
    }
    if (tid == 63) sm.set_first(Npts, incl);
  }
  if (u2 >= 0 && u2 < M) {
    const int sg = u2;
    int p0 = 0, p1 = 0;
    for (int q = 0; q < M; q++) {
      p0 = q == sg ? L.seg_piece0[q] : p0;
      p1 = q == sg ? L.seg_piece0[q + 1] : p1;
    }
    double gdT = 0.0, en = 0.0;
    for (int i = p0; i < p1; i++) {
      gdT += sm.pG[i];
      en += sm.pE[i];
    }
    sm.segsum[gNUM * sg + gGDT] = gdT;
    sm.segsum[gNUM * sg + gENERGY] = en;
    sm.segsum[gNUM * sg + gCOST0] = 0.0;
    sm.segsum[gNUM * sg + gCOST2] = 0.0;
    sm.segsum[gNUM * sg + gCOST1] = 0.0;
  }
  team_sync<WAVE>();
  // ---- chains: the active terms in windows of list_cap; lane (piece, entry) adds its piece's records in order, four more
  // lanes per segment walk all of the segment's for gdT, the corridor cost, the feasibility cost and the moving-obstacle cost
  const int n_act = sm.first(Npts);
  const int n_chain = 12 * Ntot + 4 * M;
  const int cap = sm.list_cap;
  pr.tick(10);     // numbering
  pr.count(9, n_act); // active terms of this evaluation (a count, not cycles)
  for (int c0 = 0; c0 < n_act; c0 += cap) {
    const int c1 = c0 + cap < n_act ? c0 + cap : n_act;
    for (int pt = tid; pt < Npts; pt += T) {
      mask_t m = sm.mask(pt);
      int e = sm.first(pt);
      while (m) {
        const int t = __builtin_ctzll(m);
        m &= m - 1;
        if (e >= c0 && e < c1) sm.list[e - c0] = (pt << 6) | t;
        e++;
      }
    }
    team_sync<WAVE>();
    pr.tick(11);   // the window's list
    for (int w = tid; w < n_chain; w += T) {
      int e0, e1, q, kind = -1, csg = 0;
      ldsd_t dst;
      if (w < 12 * Ntot) {
        const int p = w / 12;
        q = w - 12 * p;
        const int pt0 = sm.pinfo[4 * p + 2], pt1 = pt0 + sm.pinfo[4 * p + 3] + 1;
        e0 = sm.first(pt0);
        e1 = sm.first(pt1);
        dst = sm.gdC + w;
      } else {
        const int v = w - 12 * Ntot, sg = v >> 2, kd = v & 3; // per segment: 0 gdT, 1 corridor cost, 2 feasibility cost, 3 moving-obstacle cost
        int a0 = 0, a1 = 0;
        for (int q2 = 0; q2 < M; q2++) {
          a0 = q2 == sg ? L.seg_pt0[q2] : a0;
          a1 = q2 == sg ? L.seg_pt0[q2 + 1] : a1;
        }
        e0 = sm.first(a0);
        e1 = sm.first(a1);
        q = kd == 0 ? 12 : 13;
        dst = sm.segsum + gNUM * sg + (kd == 0 ? gGDT : (kd == 1 ? gCOST0 : (kd == 2 ? gCOST2 : gCOST1)));
        kind = kd;
        csg = sg;
      }
      e0 = e0 > c0 ? e0 : c0;
      e1 = e1 < c1 ? e1 : c1;
      if (e1 <= e0) continue;
      double acc = *dst;
      const int tS0 = 5 * H, tS1 = 5 * H + nS;
      if (kind < 0) { // an entry of gdC: every term of the piece, in order
        int e = e0;
        for (; e + 8 <= e1; e += 8) {
          int id[8];
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) id[u] = sm.list[e + u - c0];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = rec_b[((size_t)(id[u] >> 6) * nterm + (id[u] & 63)) * kRec + q];
#pragma unroll
          for (int u = 0; u < 8; u++) acc += v[u];
        }
        for (; e < e1; e++) {
          const int id = sm.list[e - c0];
          acc += rec_b[((size_t)(id >> 6) * nterm + (id & 63)) * kRec + q];
        }
      } else {
        // the per-segment chains: what an entry adds depends on its kind of term; its values are requested eight entries at
        // a time in front of the additions (slots 14 / 15 only mean something for a moving-obstacle term and are only used there)
        const mask_t smask = nS > 0 ? (((mask_t)1 << nS) - 1ull) : 0ull;
        int Nseg = 0;
        for (int q2 = 0; q2 < M; q2++) Nseg = q2 == csg ? L.piece_nums[q2] : Nseg;
        for (int e = e0; e < e1; e += 8) {
          int id[8];
          double va[8], vb[8], vc[8];
          mask_t pm[8];
#pragma unroll
          for (int u = 0; u < 8; u++) id[u] = sm.list[(e + u < e1 ? e + u : e1 - 1) - c0];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            gcd_t r_ = (gcd_t)(rec_b + ((size_t)(id[u] >> 6) * nterm + (id[u] & 63)) * kRec);
            va[u] = r_[kind == 0 ? 12 : 13];
            vb[u] = r_[14];
            vc[u] = r_[15];
            pm[u] = sm.mask(id[u] >> 6);
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            if (e + u >= e1) break;
            const int t = id[u] & 63;
            const bool sur_term = t >= tS0 && t < tS1;
            if (kind == 0) { // gdT: one `+=` per term; a moving-obstacle term: three, and one more per previous segment (traj_optimizer.cpp:1663-1676)
              acc += va[u];
              if (sur_term) {
                const int lp = sm.pinfo[4 * (int)D.pt_piece[id[u] >> 6] + 1]; // pieceid
                acc += vb[u] * lp;
                acc += vc[u];
                const double prev = vb[u] * Nseg; // ... * piece_num_container[trajid]
                for (int idx = 0; idx < csg; idx++) acc += prev;
              }
            } else if (kind == 1) {
              if (t < tS0) acc += va[u];
            } else if (kind == 2) {
              if (t >= tS1) acc += va[u];
            } else if (sur_term) { // costs(1) += the point's penalty, once per point: carried by its first active obstacle term
              const mask_t sb = (pm[u] >> tS0) & smask;
              if ((int)__builtin_ctzll(sb) == t - tS0) acc += va[u];
            }
          }
        }
      }
      *dst = acc;
    }
    team_sync<WAVE>();
  }
  } // TEAM shape
  pr.tick(3);
  // ---- calGrads_PT (poly_traj_utils.hpp:1037-1066): adj = gdC * tInv, solveAdj, the duration gradient
  for (int w = tid; w < 12 * Ntot; w += T) {
    const int p = w / 12, k = (w - 12 * p) >> 1;
    sm.adj[w] = sm.gdC[w] * sm.seg[16 * sm.pinfo[4 * p] + 8 + k];
  }
  team_sync<WAVE>();
  if (tid < 2 * M) {
    const int sg = tid >> 1, d = tid & 1;
    int N = 0, p0 = 0;
    for (int q = 0; q < M; q++) {
      N = q == sg ? L.piece_nums[q] : N;
      p0 = q == sg ? L.seg_piece0[q] : p0;
    }
    int toff = 0;
    for (int q = 0; q < M; q++) toff += q < sg ? pk_segment_doubles(L.piece_nums[q]) : 0;
    ldscd_t tb = sm.tab + toff;
    sweep<2>(tb + pk_sweep_offset(2, N), sm.adj + 12 * p0, 6 * N, d);
    sweep<3>(tb + pk_sweep_offset(3, N), sm.adj + 12 * p0, 6 * N, d);
  }
  for (int i = u2; i >= 0 && i < Ntot; i += (WAVE ? 64 : T - 64)) { // the per-piece chain-rule terms (they only need gdC and b)
    ldscd_t tInv = sm.seg + 16 * sm.pinfo[4 * i] + 8;
    const double gdtInv[6] = {0.0, -1.0 * tInv[2], -2.0 * tInv[3], -3.0 * tInv[4], -4.0 * tInv[5], -5.0 * tInv[5] * tInv[1]};
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double gdcol = sm.gdC[12 * i + 2 * k] * sm.b[12 * i + 2 * k] + sm.gdC[12 * i + 2 * k + 1] * sm.b[12 * i + 2 * k + 1];
      acc += gdtInv[k] * gdcol;
    }
    sm.pA[i] = acc;
  }
  team_sync<WAVE>();
  pr.tick(4);
  // ---- gradient and cost (traj_optimizer.cpp:299-344)
  for (int e = tid; e < L.x_tau0; e += T) { // gdP of every segment: rows 6 i + 5 of its adjoint
    int x0 = 0, p0 = 0;
    for (int q = 0; q < M; q++) {
      const bool in = e >= L.seg_x0[q];
      x0 = in ? L.seg_x0[q] : x0;
      p0 = in ? L.seg_piece0[q] : p0;
    }
    const int w = e - x0;
    g[e] = sm.adj[12 * p0 + 2 * (6 * (w >> 1) + 5) + (w & 1)];
  }
  if (tid < M) { // the duration gradient of segment tid (poly_traj_utils.hpp:1050-1064, VirtualTGradCost :405-419)
    const int sg = tid;
    int N = 0, p0 = 0;
    for (int q = 0; q < M; q++) {
      N = q == sg ? L.piece_nums[q] : N;
      p0 = q == sg ? L.seg_piece0[q] : p0;
    }
    ldscd_t adj = sm.adj + 12 * p0;
    ldscd_t hv = sm.pva + 12 * sg, tv = hv + 6;
    const int n6 = 6 * N;
    const double t1 = sm.seg[16 * sg + 3];
    double gdT = sm.segsum[gNUM * sg + gGDT];
    gdT += hv[2] * adj[2 * 1] + hv[3] * adj[2 * 1 + 1];
    gdT += (hv[4] * adj[2 * 2] + hv[5] * adj[2 * 2 + 1]) * 2.0 * t1;
    gdT += tv[2] * adj[2 * (n6 - 2)] + tv[3] * adj[2 * (n6 - 2) + 1];
    gdT += (tv[4] * adj[2 * (n6 - 1)] + tv[5] * adj[2 * (n6 - 1) + 1]) * 2.0 * t1;
    for (int i = 0; i < N; i++) gdT += sm.pA[p0 + i];
    const double VT = x[L.x_tau0 + sg];
    double gdVT2Rt;
    if (VT > 0) {
      gdVT2Rt = VT + 1.0;
    } else {
      const double denSqrt = (0.5 * VT - 1.0) * VT + 1.0;
      gdVT2Rt = (1.0 - VT) / (denSqrt * denSqrt);
    }
    g[L.x_tau0 + sg] = (gdT / N + P.wei_time) * gdVT2Rt;
  }
  if (u2 >= 0 && u2 < M - 1 && P.gear_opt) { // junction i: position and angle (traj_optimizer.cpp:307-320)
    const int i = u2;
    int Ni = 0, p0i = 0, p0n = 0;
    for (int q = 0; q < M; q++) {
      Ni = q == i ? L.piece_nums[q] : Ni;
      p0i = q == i ? L.seg_piece0[q] : p0i;
      p0n = q == i + 1 ? L.seg_piece0[q] : p0n;
    }
    ldscd_t adji = sm.adj + 12 * p0i, adjn = sm.adj + 12 * p0n;
    const int r3 = 6 * Ni - 3;
    const double t1i = sm.seg[16 * i + 3], t1n = sm.seg[16 * (i + 1) + 3];
    // gdTail of segment i and gdHead of segment i + 1 (poly_traj_utils.hpp:1045-1049: adj row * t^k)
    const double fin0[2] = {adji[2 * r3] * 1.0, adji[2 * r3 + 1] * 1.0}, fin1[2] = {adji[2 * (r3 + 1)] * t1i, adji[2 * (r3 + 1) + 1] * t1i};
    const double ini0[2] = {adjn[0] * 1.0, adjn[1] * 1.0}, ini1[2] = {adjn[2] * t1n, adjn[3] * t1n};
    const double cs = sm.trig[2 * i], sn = sm.trig[2 * i + 1];
    // grad is zeroed, then segment i adds its tail term, then segment i + 1 its head term (trajid ascending)
    for (int d = 0; d < 2; d++) {
      double v = 0.0;
      v += fin0[d];
      v += ini0[d];
      g[L.x_gear0 + 2 * i + d] = v;
    }
    double va = 0.0;
    va += fin1[0] * (-P.non_sinv * sn) + fin1[1] * (P.non_sinv * cs);
    va += ini1[0] * (P.non_sinv * sn) + ini1[1] * (-P.non_sinv * cs);
    g[L.x_ang0 + i] = va;
  } else if (u2 >= 0 && u2 < M - 1) { // gear_opt off: the junction variables keep a zero gradient
    const int i = u2;
    g[L.x_gear0 + 2 * i] = 0.0;
    g[L.x_gear0 + 2 * i + 1] = 0.0;
    g[L.x_ang0 + i] = 0.0;
  }
  if (tid == (T > 128 ? 128 : 0)) { // the cost: sums over the segments in order (:292-297, :328-330)
    double total_smcost = 0.0, total_timecost = 0.0, penalty_cost = 0.0;
    for (int sg = 0; sg < M; sg++) {
      total_smcost += sm.segsum[gNUM * sg + gENERGY];
      penalty_cost += (sm.segsum[gNUM * sg + gCOST0] + sm.segsum[gNUM * sg + gCOST1]) + sm.segsum[gNUM * sg + gCOST2];
    }
    for (int sg = 0; sg < M; sg++) total_timecost += sm.seg[16 * sg] * P.wei_time;
    sm.st[sF] = total_smcost + total_timecost + penalty_cost;
  }
  team_sync<WAVE>();
  pr.tick(5);
}

// ------------------------------------------------ sequential sums on wave 0
// The products sit one per lane (lanes >= n hold anything); the sum is the chain 0.0 + p[0] + p[1] + ... every lane forms
// for itself from the LDS copy (broadcast reads), so all lanes end with the same bits.
// (LDS operations of one wave execute in order: the reads below see the writes above them without waiting for anything
// else -- a workgroup-scope fence here would also wait for the history rows that are in flight from global memory)
__device__ __forceinline__ void wave_lds_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
// CAP = 16 / 32 / 64 >= n values are fetched in one go (the reads go out together, the chain starts when the first
// arrives); lanes from n on contribute -0.0, and x + (-0.0) == x for EVERY x (both zeros included), so the chain may
// simply run to CAP.
// Round 5: the chain WITHOUT the trip through LDS.  v_fmac_f64 has a DPP form whose first operand
// can be lane K of the reader's own row of 16 (row_newbcast:K), and fma(p, 1.0, acc) is acc + p, rounded once: the same bits as
// the addition.  So the sixteen terms of row 0 are sixteen dependent one-instruction steps in which every lane of row 0 adds
// term K; terms 16 .. 31 (row 1) are first moved under row 0 by one v_permlane16_swap per register half, then chained the same
// way, rows 2 and 3 after a v_permlane32_swap.  No store, no fence, no sixteen broadcast reads: 832 -> 477 cycles per step of the
// two-loop recursion at 4096, 370 -> 316 ms per batch, the same bits.  (A VALU write followed by a DPP read of the register needs two wait
// states, and EXEC is not touched here: s_nop 1 in front, the inline-asm block is opaque to the hazard recogniser.)
#ifndef DFTPAV_REF_DPP_CHAIN
#define DFTPAV_REF_DPP_CHAIN 1
#endif
// row r of v moved under row 0: v_permlane16_swap exchanges vdst's odd rows with src's even rows, v_permlane32_swap vdst's upper
// half with src's lower half; with both operands = v the SECOND result holds v's row 1 (rows 2, 3) in row 0 (rows 0, 1)
__device__ __forceinline__ double row1_to_row0(double v) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double upper_to_lower(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[1], lo[1]);
}
template <int CAP>
__device__ __forceinline__ double seq_sum_dpp(double p, int n, int lane) {
  static_assert(CAP == 16 || CAP == 32 || CAP == 40 || CAP == 48 || CAP == 64, "whole rows, or half of row 2");
  const double v = lane < n ? p : -0.0; // terms n .. CAP-1 are -0.0, as in the LDS form (lanes >= CAP hold anything: nobody chains them)
  double acc = 0.0;
  const double one = 1.0;
  asm volatile("s_nop 1\n\t" DFTPAV_FMAC_BCAST16 : "+v"(acc) : "v"(v), "v"(one)); // terms 0 .. 15
  if (CAP >= 32) {
    const double w = row1_to_row0(v);
    asm volatile("s_nop 1\n\t" DFTPAV_FMAC_BCAST16 : "+v"(acc) : "v"(w), "v"(one)); // 16 .. 31
  }
  if (CAP > 32) {
    const double u = upper_to_lower(v); // rows 2, 3 under rows 0, 1
    if (CAP == 40) {
      asm volatile("s_nop 1\n\t" DFTPAV_FMAC_BCAST8 : "+v"(acc) : "v"(u), "v"(one)); // 32 .. 39
    } else {
      asm volatile("s_nop 1\n\t" DFTPAV_FMAC_BCAST16 : "+v"(acc) : "v"(u), "v"(one)); // 32 .. 47
    }
    if (CAP == 64) {
      const double w3 = row1_to_row0(u);
      asm volatile("s_nop 1\n\t" DFTPAV_FMAC_BCAST16 : "+v"(acc) : "v"(w3), "v"(one)); // 48 .. 63
    }
  }
  const int rl = __builtin_amdgcn_readfirstlane(__double2loint(acc)), rh = __builtin_amdgcn_readfirstlane(__double2hiint(acc));
  return __hiloint2double(rh, rl);
}
template <int CAP>
__device__ __forceinline__ double seq_sum(double p, int n, ldsd_t buf, int lane) {
#if DFTPAV_REF_DPP_CHAIN
  return seq_sum_dpp<CAP>(p, n, lane);
#endif
  if (lane < CAP) buf[lane] = lane < n ? p : -0.0;
  wave_lds_order();
  double s = 0.0;
  { // (the chain on one lane + a broadcast was measured slower: 637 against 546 cycles per history step)
    double v[CAP];
#pragma unroll
    for (int u = 0; u < CAP; u++) v[u] = buf[u];
#pragma unroll
    for (int u = 0; u < CAP; u++) s += v[u];
  }
  wave_lds_order(); // the buffer is free again
  return s;
}

// ---- the two-loop recursion's view of the history: blocks of kPB stored pairs in registers, the next block in flight while
// one is worked on
constexpr int kPB = 8;
struct HistBlk {
  d2_t sy[kPB]; // (s, y) element of this lane
  d2_t yr[kPB]; // (ys, 1 / ys) of the pair
};
template <int DIR>
__device__ __forceinline__ void load_blk(HistBlk &R, gcd2_t hS, gcd2_t hR, int npad, int m, int ln, int &jl) {
#pragma unroll
  for (int q = 0; q < kPB; q++) {
    R.sy[q] = hS[(size_t)jl * npad + ln];
    R.yr[q] = hR[jl];
    if (DIR < 0) jl = jl == 0 ? m - 1 : jl - 1;
    else jl = jl == m - 1 ? 0 : jl + 1;
  }
}
// makes every register of the block a use: the wait for its loads lands here, before the next block's loads are issued,
// so it is a wait for this block only (solver.hip: pin_block)
__device__ __forceinline__ void pin_blk(HistBlk &R) {
#pragma unroll
  for (int q = 0; q < kPB; q++) {
    asm volatile("" : "+v"(R.sy[q].x), "+v"(R.sy[q].y), "+v"(R.yr[q].x), "+v"(R.yr[q].y));
  }
}
// kPB steps of the first loop (lbfgs.hpp:722-726): alpha_j = s_j . d / ys_j ; d -= alpha_j y_j
template <int CAP, bool EXACT>
__device__ __forceinline__ void first_steps(const HistBlk &R, ldsd_t dot_buf, ldsd_t alpha_buf, int i0, int bound, int m, int n, int lane, int &j, double &dreg) {
#pragma unroll
  for (int u = 0; u < kPB; u++) {
    if (i0 + u < bound) { // uniform
      j = j == 0 ? m - 1 : j - 1;
      const double dot = seq_sum<CAP>(R.sy[u].x * dreg, n, dot_buf, lane);
      const double a = div_by_rcp<EXACT>(dot, R.yr[u].x, R.yr[u].y); // lm_alpha[j] = lm_s.col(j).dot(d) / lm_ys[j]
      if (lane == 0) alpha_buf[j] = a;
      const double na = -a;
      dreg = dreg + na * R.sy[u].y; // d += (-alpha) * lm_y.col(j)
    }
  }
}
// kPB steps of the second loop (lbfgs.hpp:732-738): beta = y_j . d / ys_j ; d += (alpha_j - beta) s_j
template <int CAP, bool EXACT>
__device__ __forceinline__ void second_steps(const HistBlk &R, ldsd_t dot_buf, ldsd_t alpha_buf, int i0, int bound, int m, int n, int lane, int &j, double &dreg) {
  double al[kPB];
  {
    int jj = j;
#pragma unroll
    for (int u = 0; u < kPB; u++) {
      al[u] = alpha_buf[jj];
      jj = jj == m - 1 ? 0 : jj + 1;
    }
  }
#pragma unroll
  for (int u = 0; u < kPB; u++) {
    if (i0 + u < bound) { // uniform
      const double dot = seq_sum<CAP>(R.sy[u].y * dreg, n, dot_buf, lane);
      const double beta = div_by_rcp<EXACT>(dot, R.yr[u].x, R.yr[u].y);
      const double cf = al[u] - beta;
      dreg = dreg + cf * R.sy[u].x; // d += (alpha - beta) * lm_s.col(j)
      j = j == m - 1 ? 0 : j + 1;
    }
  }
}

// The two-loop recursion (lbfgs.hpp:716-739) over `bound` stored pairs, the newest in slot ne - 1: d = -g on entry (lanes >= n:
// 0.0), H0 = ys / yy between the loops.  A function of its own: its registers -- two history blocks in flight, the 32 values of
// a sequential sum -- are then allocated for it alone, not squeezed between whatever the rest of the kernel keeps live (inlined,
// the cost of a history step moved between 850 and 1850 cycles with unrelated edits elsewhere in the kernel).
template <int CAP, bool EXACT>
__device__ DFTPAV_REF_TL_ATTR double two_loop(ldsd_t dot_buf_, ldsd_t alpha_buf_, gcd2_t cS_, gcd2_t cR_, int npad_, int m_, int n_, int lane, int bound_, int ne_,
                                             double dreg, double sc0) {
  // arguments of an out-of-line function arrive in vector registers: everything but the lane's own values is the same in every
  // lane -- say so, and the loop control, the addresses and the branches below are scalar again
  const int npad = __builtin_amdgcn_readfirstlane(npad_), m = __builtin_amdgcn_readfirstlane(m_), n = __builtin_amdgcn_readfirstlane(n_);
  const int bound = __builtin_amdgcn_readfirstlane(bound_), ne = __builtin_amdgcn_readfirstlane(ne_);
  const ldsd_t dot_buf = (ldsd_t)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)dot_buf_);
  const ldsd_t alpha_buf = (ldsd_t)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)alpha_buf_);
  const unsigned long long cS_u = (unsigned long long)cS_, cR_u = (unsigned long long)cR_;
  const gcd2_t cS = (gcd2_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(cS_u >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)(cS_u & 0xffffffffull)));
  const gcd2_t cR = (gcd2_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(cR_u >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)(cR_u & 0xffffffffull)));
  const int ln = lane < n ? lane : 0;
  HistBlk A, B;
  // first loop: newest -> oldest (slots ne-1, ne-2, ...)
  int j = ne;
  int jl = ne == 0 ? m - 1 : ne - 1;
  load_blk<-1>(A, cS, cR, npad, m, ln, jl);
  for (int i0 = 0; i0 < bound; i0 += 2 * kPB) {
    pin_blk(A);
    load_blk<-1>(B, cS, cR, npad, m, ln, jl);
    first_steps<CAP, EXACT>(A, dot_buf, alpha_buf, i0, bound, m, n, lane, j, dreg);
    pin_blk(B);
    load_blk<-1>(A, cS, cR, npad, m, ln, jl);
    first_steps<CAP, EXACT>(B, dot_buf, alpha_buf, i0 + kPB, bound, m, n, lane, j, dreg);
  }
  dreg = dreg * sc0;
  wave_lds_order(); // alpha written by lane 0, read by all below
  // second loop: oldest -> newest, from the slot the first loop ended on
  jl = j;
  load_blk<+1>(A, cS, cR, npad, m, ln, jl);
  for (int i0 = 0; i0 < bound; i0 += 2 * kPB) {
    pin_blk(A);
    load_blk<+1>(B, cS, cR, npad, m, ln, jl);
    second_steps<CAP, EXACT>(A, dot_buf, alpha_buf, i0, bound, m, n, lane, j, dreg);
    pin_blk(B);
    load_blk<+1>(A, cS, cR, npad, m, ln, jl);
    second_steps<CAP, EXACT>(B, dot_buf, alpha_buf, i0 + kPB, bound, m, n, lane, j, dreg);
  }
  return dreg;
}

// Start of an outer iteration (lbfgs.hpp:559-574, 290-315): xp = x, gp = g, dginit = gp . d, first trial point
template <int CAP>
__device__ __forceinline__ bool begin_iteration(const DevParams &P, const Sm &sm, int n, int lane) {
  double pr = 0.0;
  if (lane < n) {
    sm.xp[lane] = sm.x[lane];
    const double gv = sm.g[lane];
    sm.gp[lane] = gv;
    pr = gv * sm.d[lane];
  }
  const double dginit = seq_sum<CAP>(pr, n, sm.dot, lane);
  const double step = sm.st[sSTEP];
  if (!(step > 0.0)) {
    if (lane == 0) sm.ist[iRET] = -1006;
    return false;
  }
  if (0.0 < dginit) {
    if (lane == 0) sm.ist[iRET] = -1005;
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
  if (lane < n) sm.x[lane] = sm.xp[lane] + step * sm.d[lane];
  return true;
}

// Everything lbfgs_optimize does between two evaluations (lbfgs.hpp:524-745 with the line search of :312-389 unrolled into
// it), on wave 0, one decision variable per lane (n <= 64); sets iACTION.
template <int CAP>
__device__ __forceinline__ void lbfgs_advance(const DevBatch &D, const Sm &sm, gd_t hS, gd_t hR, int lane, Prof &pr) {
  const DevParams &P = D.P;
  const int n = D.L.n, m = P.mem_size, npad = D.L.npad;
  const double f = sm.st[sF];
  int action = kActEval;
  if (sm.ist[iPHASE] == 0) { // after the first evaluation: lbfgs.hpp:524-551
    double gv = 0.0, xv = 0.0;
    if (lane < n) {
      gv = sm.g[lane];
      xv = sm.x[lane];
      sm.d[lane] = -gv;
    }
    const double gmax = wave_max64(lane < n ? fabs(gv) : 0.0), xmax = wave_max64(lane < n ? fabs(xv) : 0.0);
    const double dd = seq_sum<CAP>((-gv) * (-gv), n, sm.dot, lane);
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
        sm.ist[iRET] = 0;
        sm.ist[iK] = 0;
      }
      action = kActDone;
    } else {
      if (lane == 0) {
        sm.st[sSTEP] = 1.0 / sqrt(dd);
        sm.ist[iK] = 1;
      }
      __threadfence_block();
      if (!begin_iteration<CAP>(P, sm, n, lane)) action = kActDone;
    }
    if (lane == 0) sm.ist[iACTION] = action;
    return;
  }

  // ---- after a line-search trial: lbfgs.hpp:317-389
  const double fx = f;
  const double finit = sm.st[sFINIT];
  double stp = sm.st[sSTP];
  const int count = sm.ist[iCOUNT] + 1;
  int ls = 0;
  bool decided = false;
  const int evals_before = sm.ist[iEVALS];
  __threadfence_block();
  if (lane == 0) {
    sm.st[sFX] = fx;
    sm.ist[iEVALS] = evals_before + 1;
    sm.ist[iCOUNT] = count;
  }
  if (isinf(fx) || isnan(fx)) {
    ls = -1012;
    decided = true;
  } else if (P.past > 0 && fabs(finit - fx) / (fabs(finit) + 1.0) < P.delta / P.past) { // lbfgs.hpp:326-329
    ls = count;
    decided = true;
  } else {
    double mu = sm.st[sMU], nu = sm.st[sNU];
    bool brackt = sm.ist[iBRACKT] != 0;
    const int touched = sm.ist[iTOUCHED];
    if (fx > finit + stp * sm.st[sDGTEST]) {
      nu = stp;
      brackt = true;
    } else {
      const double gs = seq_sum<CAP>(lane < n ? sm.g[lane] * sm.d[lane] : 0.0, n, sm.dot, lane);
      if (gs < sm.st[sDSTEST]) {
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
    if (lane == 0) {
      sm.st[sMU] = mu;
      sm.st[sNU] = nu;
      sm.ist[iBRACKT] = brackt ? 1 : 0;
      sm.st[sSTP] = stp;
      if (touch_now) sm.ist[iTOUCHED] = 1;
    }
    if (!decided) {
      if (lane < n) sm.x[lane] = sm.xp[lane] + stp * sm.d[lane];
      if (lane == 0) sm.ist[iACTION] = kActEval;
      pr.tick(6);
      return;
    }
  }
  if (lane == 0) sm.st[sSTEP] = stp; // lbfgs.hpp:574 passes `step` by reference
  if (ls < 0) { // lbfgs.hpp:604-611: x, g reverted; fx is not
    if (lane < n) {
      sm.x[lane] = sm.xp[lane];
      sm.g[lane] = sm.gp[lane];
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
    const double gmax = wave_max64(lane < n ? fabs(sm.g[lane]) : 0.0), xmax = wave_max64(lane < n ? fabs(sm.x[lane]) : 0.0);
    const int kGoOn = 12345;
    int ret = kGoOn;
    if (gmax / fmax(1.0, xmax) < P.g_epsilon) {
      ret = 0;
    } else {
      if (0 < P.past) {
        const int slot = k % P.past;
        const double pf = sm.st[sPF0 + slot];
        __threadfence_block();
        if (P.past <= k) {
          const double rate = fabs(pf - fx) / fmax(1.0, fabs(fx));
          if (rate < P.delta) ret = 1;
        }
        if (ret == kGoOn && lane == 0) sm.st[sPF0 + slot] = fx;
      }
      if (ret == kGoOn && P.max_iterations != 0 && P.max_iterations <= k) ret = -1008;
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
  pr.tick(6);
  const int end = sm.ist[iEND];
  int bound = sm.ist[iBOUND];
  __threadfence_block();
  if (lane == 0) sm.ist[iK] = k;

  // ---- history update + two-loop recursion (lbfgs.hpp:676-740); (s, y) interleaved per element as solver.hip stores them
  double sv = 0.0, yv = 0.0, gpv = 0.0;
  if (lane < n) {
    sv = sm.x[lane] - sm.xp[lane];
    yv = sm.g[lane] - sm.gp[lane];
    gpv = sm.gp[lane];
    d2_t sy;
    sy.x = sv;
    sy.y = yv;
    ((gd2_t)hS)[(size_t)end * npad + lane] = sy;
    sm.d[lane] = -sm.g[lane];
  }
  // the four dot products of lbfgs.hpp:683-694, their chains side by side
  double ys, yy, ss, gpgp;
  {
    if (lane < CAP) {
      sm.dot[lane] = yv * sv;
      sm.dot[CAP + lane] = yv * yv;
      sm.dot[2 * CAP + lane] = sv * sv;
      sm.dot[3 * CAP + lane] = gpv * gpv;
    }
    wave_lds_order();
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
    for (int e = 0; e < n; e++) {
      a0 += sm.dot[e];
      a1 += sm.dot[CAP + e];
      a2 += sm.dot[2 * CAP + e];
      a3 += sm.dot[3 * CAP + e];
    }
    wave_lds_order();
    ys = a0; yy = a1; ss = a2; gpgp = a3;
  }
  if (lane == 0) {
    d2_t yr;
    yr.x = ys;
    yr.y = 1.0 / ys;
    ((gd2_t)hR)[end] = yr;
    if (!rcp_route_ok(ys)) sm.ist[iSLOWDIV] = 1; // from here on the recursion divides (see div_by_rcp)
  }
  const double cau = ss * sqrt(gpgp) * P.cautious_factor;
  pr.tick(7);
  if (ys > cau) {
    ++bound;
    bound = m < bound ? m : bound;
    const int ne = end + 1 == m ? 0 : end + 1;
    __threadfence_block(); // lane 0's (ys, 1 / ys) of the newest pair is read by every lane below
    double dreg = lane < n ? -sm.g[lane] : 0.0;
    if (__builtin_expect(sm.ist[iSLOWDIV] != 0, 0)) // (uniform; lane 0 wrote it at most a fence ago)
      dreg = two_loop<CAP, true>(sm.dot, sm.alpha, (gcd2_t)hS, (gcd2_t)hR, npad, m, n, lane, bound, ne, dreg, ys / yy);
    else
      dreg = two_loop<CAP, false>(sm.dot, sm.alpha, (gcd2_t)hS, (gcd2_t)hR, npad, m, n, lane, bound, ne, dreg, ys / yy);
    if (lane < n) sm.d[lane] = dreg;
    if (lane == 0) {
      sm.ist[iEND] = ne;
      sm.ist[iBOUND] = bound;
      long long hs = ((long long)sm.ist[iHISTHI] << 32) | (unsigned int)sm.ist[iHISTLO];
      hs += bound;
      sm.ist[iHISTLO] = (int)(hs & 0xffffffffLL);
      sm.ist[iHISTHI] = (int)(hs >> 32);
    }
  }
  if (lane == 0) sm.st[sSTEP] = 1.0; // lbfgs.hpp:743
  pr.tick(8);
  __threadfence_block();
  const bool ok = begin_iteration<CAP>(P, sm, n, lane);
  if (lane == 0) sm.ist[iACTION] = ok ? kActEval : kActDone;
  pr.tick(6);
}


// ------------------------------------------------ the same step for ANY number of variables (TEAM shape, n > 64)
// lbfgs_advance keeps one decision variable per lane and chains a dot product through the lanes; a plan of more than 32 pieces
// (traj_manager.cpp:543 sets no limit, lbfgs.hpp:512-513 none either) has more variables than a wave has lanes.  This is the slow,
// plain form for those: the vectors live in LDS (they do anyway), a lane takes the elements lane, lane + 64, ..., and a sequential
// sum is what it is in the reference -- one chain over the n products, read back from LDS by every lane (the same bits in each).
// True divisions throughout.  Same state, same scalars, same history layout as lbfgs_advance: ring, slices and suspend / resume
// are not used in this path (it runs in the TEAM shape only).
__device__ __forceinline__ double gen_sum(ldsd_t buf, int n) { // 0.0 + buf[0] + buf[1] + ... (the products were written by this wave)
  wave_lds_order();
  double s_ = 0.0;
  for (int e = 0; e < n; e++) s_ += buf[e];
  wave_lds_order();
  return s_;
}
__device__ __forceinline__ bool gen_begin_iteration(const DevParams &P, const Sm &sm, int n, int lane) {
  for (int e = lane; e < n; e += 64) {
    sm.xp[e] = sm.x[e];
    const double gv = sm.g[e];
    sm.gp[e] = gv;
    sm.dot[e] = gv * sm.d[e];
  }
  const double dginit = gen_sum(sm.dot, n);
  const double step = sm.st[sSTEP];
  if (!(step > 0.0)) {
    if (lane == 0) sm.ist[iRET] = -1006;
    return false;
  }
  if (0.0 < dginit) {
    if (lane == 0) sm.ist[iRET] = -1005;
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
__device__ __noinline__ void lbfgs_advance_generic(const DevBatch &D, const Sm &sm, gd_t hS, gd_t hR, int lane, Prof &pr) {
  const DevParams &P = D.P;
  const int n = D.L.n, m = P.mem_size, npad = D.L.npad;
  const double f = sm.st[sF];
  int action = kActEval;
  auto vmax = [&](ldscd_t v) { // max |v[e]| (order-free)
    double mx = 0.0;
    for (int e = lane; e < n; e += 64) mx = fmax(mx, fabs(v[e]));
    return wave_max64(mx);
  };
  if (sm.ist[iPHASE] == 0) { // after the first evaluation: lbfgs.hpp:524-551
    for (int e = lane; e < n; e += 64) {
      const double gv = sm.g[e];
      sm.d[e] = -gv;
      sm.dot[e] = (-gv) * (-gv);
    }
    const double gmax = vmax(sm.g), xmax = vmax(sm.x);
    const double dd = gen_sum(sm.dot, n);
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
        sm.ist[iRET] = 0;
        sm.ist[iK] = 0;
      }
      action = kActDone;
    } else {
      if (lane == 0) {
        sm.st[sSTEP] = 1.0 / sqrt(dd);
        sm.ist[iK] = 1;
      }
      __threadfence_block();
      if (!gen_begin_iteration(P, sm, n, lane)) action = kActDone;
    }
    if (lane == 0) sm.ist[iACTION] = action;
    return;
  }
  // ---- after a line-search trial: lbfgs.hpp:317-389
  const double fx = f;
  const double finit = sm.st[sFINIT];
  double stp = sm.st[sSTP];
  const int count = sm.ist[iCOUNT] + 1;
  int ls = 0;
  bool decided = false;
  const int evals_before = sm.ist[iEVALS];
  __threadfence_block();
  if (lane == 0) {
    sm.st[sFX] = fx;
    sm.ist[iEVALS] = evals_before + 1;
    sm.ist[iCOUNT] = count;
  }
  if (isinf(fx) || isnan(fx)) {
    ls = -1012;
    decided = true;
  } else if (P.past > 0 && fabs(finit - fx) / (fabs(finit) + 1.0) < P.delta / P.past) { // lbfgs.hpp:326-329
    ls = count;
    decided = true;
  } else {
    double mu = sm.st[sMU], nu = sm.st[sNU];
    bool brackt = sm.ist[iBRACKT] != 0;
    const int touched = sm.ist[iTOUCHED];
    if (fx > finit + stp * sm.st[sDGTEST]) {
      nu = stp;
      brackt = true;
    } else {
      for (int e = lane; e < n; e += 64) sm.dot[e] = sm.g[e] * sm.d[e];
      const double gs = gen_sum(sm.dot, n);
      if (gs < sm.st[sDSTEST]) {
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
    if (lane == 0) {
      sm.st[sMU] = mu;
      sm.st[sNU] = nu;
      sm.ist[iBRACKT] = brackt ? 1 : 0;
      sm.st[sSTP] = stp;
      if (touch_now) sm.ist[iTOUCHED] = 1;
    }
    if (!decided) {
      for (int e = lane; e < n; e += 64) sm.x[e] = sm.xp[e] + stp * sm.d[e];
      if (lane == 0) sm.ist[iACTION] = kActEval;
      pr.tick(6);
      return;
    }
  }
  if (lane == 0) sm.st[sSTEP] = stp; // lbfgs.hpp:574 passes `step` by reference
  if (ls < 0) { // lbfgs.hpp:604-611: x, g reverted; fx is not
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
    const double gmax = vmax(sm.g), xmax = vmax(sm.x);
    const int kGoOn = 12345;
    int ret = kGoOn;
    if (gmax / fmax(1.0, xmax) < P.g_epsilon) {
      ret = 0;
    } else {
      if (0 < P.past) {
        const int slot = k % P.past;
        const double pf = sm.st[sPF0 + slot];
        __threadfence_block();
        if (P.past <= k) {
          const double rate = fabs(pf - fx) / fmax(1.0, fabs(fx));
          if (rate < P.delta) ret = 1;
        }
        if (ret == kGoOn && lane == 0) sm.st[sPF0 + slot] = fx;
      }
      if (ret == kGoOn && P.max_iterations != 0 && P.max_iterations <= k) ret = -1008;
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
  pr.tick(6);
  const int end = sm.ist[iEND];
  int bound = sm.ist[iBOUND];
  __threadfence_block();
  if (lane == 0) sm.ist[iK] = k;
  // ---- history update (lbfgs.hpp:676-694): s = x - xp, y = g - gp, the four dot products
  double ys, yy, ss, gpgp;
  {
    for (int e = lane; e < n; e += 64) {
      const double sv = sm.x[e] - sm.xp[e], yv = sm.g[e] - sm.gp[e];
      d2_t sy;
      sy.x = sv;
      sy.y = yv;
      ((gd2_t)hS)[(size_t)end * npad + e] = sy;
      sm.d[e] = -sm.g[e];
      sm.dot[e] = yv * sv;
    }
    ys = gen_sum(sm.dot, n);
    for (int e = lane; e < n; e += 64) {
      const double yv = sm.g[e] - sm.gp[e];
      sm.dot[e] = yv * yv;
    }
    yy = gen_sum(sm.dot, n);
    for (int e = lane; e < n; e += 64) {
      const double sv = sm.x[e] - sm.xp[e];
      sm.dot[e] = sv * sv;
    }
    ss = gen_sum(sm.dot, n);
    for (int e = lane; e < n; e += 64) sm.dot[e] = sm.gp[e] * sm.gp[e];
    gpgp = gen_sum(sm.dot, n);
  }
  if (lane == 0) {
    d2_t yr;
    yr.x = ys;
    yr.y = 1.0 / ys;
    ((gd2_t)hR)[end] = yr;
  }
  const double cau = ss * sqrt(gpgp) * P.cautious_factor;
  pr.tick(7);
  if (ys > cau) { // ---- the two-loop recursion (lbfgs.hpp:716-739), plain
    ++bound;
    bound = m < bound ? m : bound;
    const int ne = end + 1 == m ? 0 : end + 1;
    __threadfence_block(); // the newest pair's row and y . s are read back below
    const gcd2_t cS = (gcd2_t)hS, cR = (gcd2_t)hR;
    int j = ne;
    for (int i = 0; i < bound; i++) {
      j = j == 0 ? m - 1 : j - 1;
      for (int e = lane; e < n; e += 64) sm.dot[e] = cS[(size_t)j * npad + e].x * sm.d[e];
      const double a = gen_sum(sm.dot, n) / cR[j].x; // lm_alpha[j] = lm_s.col(j).dot(d) / lm_ys[j]
      if (lane == 0) sm.alpha[j] = a;
      const double na = -a;
      for (int e = lane; e < n; e += 64) sm.d[e] = sm.d[e] + na * cS[(size_t)j * npad + e].y; // d += (-alpha) * lm_y.col(j)
    }
    const double sc0 = ys / yy;
    for (int e = lane; e < n; e += 64) sm.d[e] = sm.d[e] * sc0;
    wave_lds_order();
    for (int i = 0; i < bound; i++) {
      for (int e = lane; e < n; e += 64) sm.dot[e] = cS[(size_t)j * npad + e].y * sm.d[e];
      const double beta = gen_sum(sm.dot, n) / cR[j].x;
      const double cf = sm.alpha[j] - beta;
      for (int e = lane; e < n; e += 64) sm.d[e] = sm.d[e] + cf * cS[(size_t)j * npad + e].x; // d += (alpha - beta) * lm_s.col(j)
      j = j == m - 1 ? 0 : j + 1;
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
  if (lane == 0) sm.st[sSTEP] = 1.0; // lbfgs.hpp:743
  pr.tick(8);
  __threadfence_block();
  const bool ok = gen_begin_iteration(P, sm, n, lane);
  if (lane == 0) sm.ist[iACTION] = ok ? kActEval : kActDone;
  pr.tick(6);
}

// ------------------------------------------------ the kernel
// solver state of a suspended trajectory <-> its record in DevBatch::state (layout as solver.hip's: five vectors at pitch
// npad, the scalars, the integers); the history and (ys, 1 / ys) of the stored pairs live in HBM anyway
__device__ inline void state_io(const DevBatch &D, const Sm &sm, int b, int lane, int nl, bool save) {
  const int n = D.L.n, npad = D.L.npad;
  double *rec = D.state + (size_t)b * D.state_stride;
  ldsd_t vecs[5] = {sm.x, sm.xp, sm.g, sm.gp, sm.d};
#pragma unroll
  for (int a = 0; a < 5; a++)
    for (int e = lane; e < nl; e += 64) {
      if (save) {
        if (e < n) rec[a * npad + e] = vecs[a][e];
      } else {
        vecs[a][e] = e < n ? rec[a * npad + e] : 0.0;
      }
    }
  double *r2 = rec + 5 * npad;
  for (int w = lane; w < sNUM; w += 64) {
    if (save) r2[w] = sm.st[w];
    else sm.st[w] = r2[w];
  }
  int *ri = reinterpret_cast<int *>(r2 + 24);
  for (int w = lane; w < iNUM; w += 64) {
    if (save) ri[w] = sm.ist[w];
    else sm.ist[w] = ri[w];
  }
}

// TEAM: one workgroup per trajectory (blockIdx.x).  WAVE: every wave of the workgroup is a team of its own; source 0: wave w of
// workgroup i takes trajectory i * W + w; source 1 (solves only): it pops trajectories from the ring until the ring is empty,
// runs each for `slice` iterations and pushes it back unfinished.
// Registers: 256 per lane (two waves per SIMD) for the kernels that fit them -- a second trajectory fills the issue slots the
// dependent chains of the first leave empty; the wide (n > 32) and the moving-obstacle kernels take 512.
// GEN (TEAM shape only): more variables than a wave has lanes (lbfgs_advance_generic) and / or more than five half-planes per point
// (twelve plane slots per point, the 64-bit test mask)
template <int CAP, bool SUR, bool WAVE, bool GEN = false>
__global__ void __launch_bounds__((WAVE && CAP <= kNarrowCap && !SUR) ? 512 : 256, (!WAVE && CAP <= 32 && !SUR) ? 2 : 1)
    ref_kernel(const DevBatch *__restrict__ Dp, int mode, const double *__restrict__ tabs, double *__restrict__ scratch, int source, int slice) {
  extern __shared__ double lds_raw[];
  const DevBatch &D = *Dp;
  const DevLayout &L = D.L;
  const int tidb = threadIdx.x, Tb = blockDim.x, lane = tidb & 63, wv = tidb >> 6, W = Tb >> 6;
  const int tid = WAVE ? lane : tidb, T = WAVE ? 64 : Tb; // inside the team
  const int n = L.n;
  const Shape sh = make_shape(L, SUR ? D.sur.S : 0, WAVE);
  Sm sm;
  {
    char *base = reinterpret_cast<char *>(lds_raw);
    char *team = base + lds_shared_bytes(L) + (WAVE ? (size_t)wv * lds_team_bytes(L, D.P.mem_size, sh) : 0);
    carve(sm, lds_raw, reinterpret_cast<double *>(team), L, D.P.mem_size, sh);
  }
  // ---- shared by the workgroup: the sweep tables and the piece table
  for (int i = tidb; i < (int)table_doubles(L); i += Tb) ((ldsd_t)sm.tab)[i] = tabs[i];
  for (int p = tidb; p < L.Ntot; p += Tb) { // piece -> segment, index inside it, first constraint point, intervals
    int sg = 0, p0 = 0, N = 0, pt0s = 0;
    for (int q = 0; q < L.M; q++) {
      const bool in = p >= L.seg_piece0[q];
      sg = in ? q : sg;
      p0 = in ? L.seg_piece0[q] : p0;
      N = in ? L.piece_nums[q] : N;
      pt0s = in ? L.seg_pt0[q] : pt0s;
    }
    const int lp = p - p0;
    sm.pinfo[4 * p] = sg;
    sm.pinfo[4 * p + 1] = lp;
    sm.pinfo[4 * p + 2] = pt0s + (lp == 0 ? 0 : (L.Kd + 1) + (lp - 1) * (L.K + 1)); // pieces of a segment are [Kd+1, K+1, ..., K+1, Kd+1] points long
    sm.pinfo[4 * p + 3] = (lp == 0 || lp == N - 1) ? L.Kd : L.K;
  }
  __syncthreads(); // the only time the waves of a WAVE-shaped workgroup meet
  const bool ring = WAVE && mode == kModeSolve && (source & 1) != 0;
  const bool force_exact_div = (source & 2) != 0; // test hook: the recursion with true divisions from the first iteration on
  const int nterm = 5 * L.H + (SUR ? D.sur.S : 0) + 4, nS_ = SUR ? D.sur.S : 0;
  const size_t scratch_per_traj = (size_t)L.Npts * nterm * kRec + (size_t)L.Npts * nS_ * kRec + (size_t)L.Npts * nterm;
  Prof pr;

  for (int pass = 0;; pass++) {
    int b;
    if (ring) {
      int id = -1;
      if (lane == 0) id = ring_pop(D.qctl, D.queue, D.qcap);
      b = __builtin_amdgcn_readfirstlane(id);
    } else {
      b = pass == 0 ? (WAVE ? (int)blockIdx.x * W + wv : (int)blockIdx.x) : -1;
      if (b >= D.B) b = -1;
    }
    if (b < 0) break;
    const bool resume = ring && D.sflag[b] == 1;
    if (resume) {
      state_io(D, sm, b, tid, sh.nl, false);
    } else {
      const double *xsrc = (mode == kModeSolve) ? D.x0 : (mode == kModeEval ? D.x_in : D.x_out);
      for (int e = tid; e < sh.nl; e += T) {
        sm.x[e] = e < n ? xsrc[(size_t)b * n + e] : 0.0;
        sm.xp[e] = 0.0;
        sm.g[e] = 0.0;
        sm.gp[e] = 0.0;
        sm.d[e] = 0.0;
      }
      if (tid < iNUM) sm.ist[tid] = (tid == iSLOWDIV && force_exact_div) ? 1 : 0;
    }
    for (int w = tid; w < 12 * L.M; w += T) {
      const int sg = w / 12, q = w - 12 * sg;
      sm.bnd[w] = q < 6 ? D.iniS[((size_t)b * L.M + sg) * 6 + q] : D.finS[((size_t)b * L.M + sg) * 6 + (q - 6)];
    }
    const gcd_t cor_b = (gcd_t)(D.corridor + (size_t)b * L.H * 4 * D.NptsPad);
    const gd_t rec_b = (gd_t)(scratch + (size_t)b * scratch_per_traj);
    const gd_t hS = (gd_t)(D.histS + (size_t)b * D.P.mem_size * L.npad * 2);
    const gd_t hR = (gd_t)(D.histR + (size_t)b * D.P.mem_size * 2);
    const long long tick0 = wall_clock64();
    pr.start(D.prof != nullptr && mode == kModeSolve && tid == 0, D.prof + (size_t)b * 12, resume);
    team_sync<WAVE>();
    const int k_start = sm.ist[iK];

    bool finished = true;
    while (true) { // (one call site of the evaluation: the kernel is instruction-cache-sized as it is)
      ref_eval<SUR, WAVE, GEN ? 12 : 5>(D, cor_b, rec_b, sm, sm.x, sm.g, pr); // x0 / the trial point the trajectory was suspended on / the next trial point
      if (mode == kModeEval) {
        for (int e = tid; e < n; e += T) D.g_out[(size_t)b * n + e] = sm.g[e];
        if (tid == 0) D.f_eval[b] = sm.st[sF];
        return;
      }
      if (mode == kModeCoeffs) {
        for (int w = tid; w < 12 * L.Ntot; w += T) D.coef_out[(size_t)b * 12 * L.Ntot + w] = sm.c[w];
        for (int sg = tid; sg < L.M; sg += T) D.dt_out[(size_t)b * L.M + sg] = sm.seg[16 * sg + 1];
        return;
      }
      if (tid < 64) {
        if constexpr (GEN) lbfgs_advance_generic(D, sm, hS, hR, lane, pr);
        else lbfgs_advance<CAP>(D, sm, hS, hR, lane, pr);
      }
      team_sync<WAVE>();
      if (sm.ist[iACTION] == kActDone) break;
      if (ring && slice > 0 && sm.ist[iK] - k_start >= slice) { // uniform
        finished = false;
        break;
      }
    }
    const long long spent = wall_clock64() - tick0;
    if (finished) {
      for (int e = tid; e < n; e += T) D.x_out[(size_t)b * n + e] = sm.x[e];
      if (tid == 0) {
        const double fx = sm.st[sFX];
        const int ret = sm.ist[iRET];
        D.f_out[b] = fx;
        D.status[b] = ret;
        D.iters[b] = sm.ist[iK];
        D.evals[b] = sm.ist[iEVALS];
        D.hist_sum[b] = ((long long)sm.ist[iHISTHI] << 32) | (unsigned int)sm.ist[iHISTLO];
        {
          double *rec = reinterpret_cast<double *>(D.records + (size_t)16 * b); // the all-gather record (as solver.hip's epilogue)
          rec[0] = fx;
          int *ri = reinterpret_cast<int *>(rec + 1);
          ri[0] = ret;
          ri[1] = sm.ist[iK];
        }
        if (D.records_host != nullptr) { // (see device_types.h)
          double *rh = reinterpret_cast<double *>(D.records_host + (size_t)16 * b);
          rh[0] = fx;
          int *rj = reinterpret_cast<int *>(rh + 1);
          rj[0] = ret;
          rj[1] = sm.ist[iK];
        }
        D.ticks[b] = (resume ? D.ticks[b] : 0) + spent; // time in service
        int ok = (ret == 0 || ret == 1 || ret == 2 || ret == -1008 || ret == -1009) ? 1 : 0; // traj_optimizer.cpp:176-201
        if (fx >= D.P.fail_cost) ok = 0;
        D.success[b] = ok;
        if (ring) {
          D.sflag[b] = 2;
          atomicSub(&D.qctl[3], 1u);
        }
      }
    } else {
      state_io(D, sm, b, tid, sh.nl, true);
      __threadfence(); // the record and the history rows of this slice are out before the id is handed on
      if (tid == 0) {
        D.ticks[b] = (resume ? D.ticks[b] : 0) + spent;
        D.sflag[b] = 1;
        ring_push(D.qctl, D.queue, D.qcap, b);
      }
    }
    if (!ring) break;
    team_sync<WAVE>(); // this pass is done with the team's LDS
  }
}

} // namespace reford

// ---- host side
// The file is compiled as two translation units, side by side (Makefile): DFTPAV_REF_PART=1 holds the kernels with CAP 16 / 32
// and everything that is not a kernel, DFTPAV_REF_PART=2 the kernels with CAP 40 / 48 / 64; 0 (default) = one unit with all.
#ifndef DFTPAV_REF_PART
#define DFTPAV_REF_PART 0
#endif
bool reference_order_quad_supported(const DevLayout &L, const DevParams &P, int S); // solver_ref4.hip
void reference_order_quad_plan(const DevLayout &L, const DevParams &P, int B, int n_cu, RefPlan &pl);
bool reference_order_quadm_supported(const DevLayout &L, const DevParams &P, int S); // solver_ref4m.hip: several gear segments
void reference_order_quadm_plan(const DevLayout &L, const DevParams &P, int B, int n_cu, RefPlan &pl);
#if DFTPAV_REF_PART != 2
// what the layout must satisfy for the reference-order kernel (solver_ref.hip header)
bool reference_order_supported(const DevLayout &L, const DevParams &P, int S) {
  if (L.M < 1 || L.n > 256 || L.Npts >= (1 << 25)) return false; // (n > 64: the plain L-BFGS step of lbfgs_advance_generic, TEAM shape; its sums use the 256-double buffer)
  if (L.H < 1 || L.H > 12) return false; // (H > 5: the generic TEAM kernel with twelve plane slots per point; rectangles: H = 4)
  if (S < 0 || 5 * L.H + S + 4 > 64) return false; // the mask of a point's active terms has 64 bits
  for (int i = 0; i < L.M; i++)
    if (L.piece_nums[i] < 2) return false;
  const reford::Shape sh = reford::make_shape(L, S, false);
  const size_t lds = reford::lds_shared_bytes(L) + reford::lds_team_bytes(L, P.mem_size, sh);
  return lds <= 160 * 1024 - 1024;
}
// doubles of term records a batch of B trajectories needs
// (per trajectory: the records [Npts][nterm][kRec] -- TEAM: a point's own slots; WAVE: in (point, term) order -- then, WAVE
// with moving obstacles, the staging of surround_terms [Npts][S][kRec], then the (entry, piece) pairs of the terms beyond the
// LDS window)
size_t reference_order_scratch_per_traj(const DevLayout &L, int S) {
  const size_t nterm = (size_t)(5 * L.H + S + 4);
  return (size_t)L.Npts * nterm * reford::kRec + (size_t)L.Npts * S * reford::kRec + (size_t)L.Npts * nterm;
}
size_t reference_order_scratch_doubles(const DevLayout &L, int B, int S) { return (size_t)B * reference_order_scratch_per_traj(L, S); }
// doubles of the sweep tables of a segment of N pieces as the kernel reads them (the tables of a layout's segments follow one
// another)
size_t reference_order_table_doubles(int N) { return (size_t)reford::pk_segment_doubles(N); }
// full: the four sweeps of a segment as [4][6N][8] (row i of a sweep: its six coefficients, the diagonal, 1 / diagonal) -> the
// kernel's layout (solver_ref.hip: "The table of one sweep"): blocks in traversal order, whole rows at the two ends, only the
// coefficients of the interior pattern in between
void reference_order_pack_tables(int N, const double *full, double *packed) {
  using namespace reford;
  const int n6 = 6 * N;
  size_t o = 0;
  for (int q = 0; q < 4; q++) {
    const bool desc = q == 1 || q == 3, div = q == 1 || q == 2;
    const double *t = full + (size_t)q * 8 * n6;
    auto row = [&](int blk, int r) { return desc ? n6 - 1 - (6 * blk + r) : 6 * blk + r; }; // natural row of traversal row r of block blk
    for (int r = 0; r < 6; r++) // first end block
      for (int k = 0; k < 8; k++) packed[o++] = t[8 * row(0, r) + k];
    const int mask_of[4][6] = {{pk_mask(0, 0), pk_mask(0, 1), pk_mask(0, 2), pk_mask(0, 3), pk_mask(0, 4), pk_mask(0, 5)},
                               {pk_mask(1, 0), pk_mask(1, 1), pk_mask(1, 2), pk_mask(1, 3), pk_mask(1, 4), pk_mask(1, 5)},
                               {pk_mask(2, 0), pk_mask(2, 1), pk_mask(2, 2), pk_mask(2, 3), pk_mask(2, 4), pk_mask(2, 5)},
                               {pk_mask(3, 0), pk_mask(3, 1), pk_mask(3, 2), pk_mask(3, 3), pk_mask(3, 4), pk_mask(3, 5)}};
    const int size_of[4] = {pk_size(0), pk_size(1), pk_size(2), pk_size(3)}, diag_of[4] = {pk_diag0(0), pk_diag0(1), pk_diag0(2), pk_diag0(3)};
    for (int blk = 1; blk <= N - 2; blk++) {
      const size_t o0 = o;
      for (int r = 0; r < 6; r++)
        for (int k = 0; k < 6; k++)
          if (mask_of[q][r] & (1 << k)) packed[o++] = t[8 * row(blk, r) + k];
      while (o < o0 + (size_t)(div ? diag_of[q] : size_of[q])) packed[o++] = 0.0;
      if (div)
        for (int r = 0; r < 6; r++) {
          packed[o++] = t[8 * row(blk, r) + 6];
          packed[o++] = t[8 * row(blk, r) + 7];
        }
    }
    if (N >= 2) // last end block
      for (int r = 0; r < 6; r++)
        for (int k = 0; k < 8; k++) packed[o++] = t[8 * row(N - 1, r) + k];
  }
}
// the non-zero pattern the middle blocks of a sweep assume (solver_ref.hip: kInterior_), for the host's check
int reference_order_interior_mask(int sweep, int row_mod_6) { return reford::kInterior_(sweep, row_mod_6); }

// The ring of a scheduled launch, reset on the stream in front of it: queue = every trajectory, flags cleared, counters {head 0, tail B,
// reserved B, unfinished B}.  ONE launch of workgroups of ONE wave each.  (Until round 6 this was two copies and a fill by the runtime:
// three blit kernels with workgroups of several waves.  In a stream of batches the device is full of persistent waves that hold a SIMD's
// whole register file each; a blit's workgroup was seen to wait 200-670 ms for a CU with room for all of its waves -- and the solve behind
// it with it -- while single SIMDs were free: rocprofv3 --kernel-trace of scripts/ref_stream_time.py, configs[1].)
namespace reford {
__global__ void __launch_bounds__(64) ring_reset_kernel(int *__restrict__ queue, int *__restrict__ sflag, unsigned *__restrict__ qctl, int B) {
  const int i = (int)(blockIdx.x * 64 + threadIdx.x);
  if (i < B) {
    queue[i] = i;
    sflag[i] = 0;
  }
  if (i < 8) qctl[i] = (i >= 1 && i <= 3) ? (unsigned)B : 0u;
}
} // namespace reford
hipError_t launch_ring_reset(const DevBatch &D, hipStream_t stream) {
  hipLaunchKernelGGL(reford::ring_reset_kernel, dim3((D.B + 63) / 64), dim3(64), 0, stream, D.queue, D.sflag, D.qctl, D.B);
  return hipGetLastError();
}

// The launch shape of a batch (see the header).  TEAM: four waves per trajectory while the batch leaves CUs to spare (the
// parallel stages finish sooner: 70 against 73 ms at batch 32, 133 against 140 at 256), two for more.  WAVE: as many waves per
// workgroup as keep the most trajectories resident on a CU -- 8 waves of 256 registers (4 for the kernels that take 512), the
// LDS of the shared tables plus a team's part per wave.
// throughput: the caller keeps many such batches in flight (dftpav_batch_create_shaped, residency 2) -- the throughput shapes whatever B
RefPlan reference_order_plan(const DevLayout &L, const DevParams &P, int S, int B, int n_cu, bool allow_quad, bool throughput) {
  RefPlan pl{};
  const bool narrow = reford::ref_cap_of(L.n) <= reford::kNarrowCap && S == 0; // the kernels built for 256 registers (two waves per SIMD)
  const int max_waves_cu = narrow ? 8 : 4;
  const reford::Shape sw = reford::make_shape(L, S, true);
  const size_t shared = reford::lds_shared_bytes(L), team_w = reford::lds_team_bytes(L, P.mem_size, sw);
  const size_t budget = 160 * 1024;
  int best_w = 0, best_res = 0, best_wg = 0;
  for (int w = max_waves_cu; w >= 1; w--) {
    const size_t lds = shared + (size_t)w * team_w;
    if (lds > budget) continue;
    const int wg = (int)std::min<size_t>((size_t)(max_waves_cu / w), budget / lds);
    if (wg * w >= best_res) { // ties: the smaller workgroup (its waves leave sooner at the end of a launch)
      best_res = wg * w;
      best_w = w;
      best_wg = wg;
    }
  }
  // up to four per CU the TEAM shape (128 threads, 34 KB of LDS with the compact tables) holds them all at once, each one faster:
  // 171 against 195 ms at 1024, 133 against 189 at 512; at 2048 the WAVE shape is ahead, 262 against 320 ms
  const bool wide_n = L.n > 64 || L.H > 5; // more variables than a wave has lanes, or more than five half-planes per point: the generic TEAM kernel, whatever B
  bool wave = best_w > 0 && !wide_n && (B > 5 * n_cu || throughput);
  if (const char *e = std::getenv("DFTPAV_REF_SHAPE")) { // developer knob: "team" / "wave"
    if (e[0] == 't') wave = false;
    if (e[0] == 'w' && best_w > 0 && !wide_n) wave = true;
  }
  if (const char *e = std::getenv("DFTPAV_REF_WAVES")) { // developer knob: waves per workgroup in the WAVE shape
    const int w = std::atoi(e);
    if (w >= 1 && w <= max_waves_cu && shared + (size_t)w * team_w <= budget) {
      best_w = w;
      best_wg = (int)std::min<size_t>((size_t)(max_waves_cu / w), budget / (shared + (size_t)w * team_w));
    }
  }
  bool quad = allow_quad && wave && reference_order_quad_supported(L, P, S);
  if (const char *e = std::getenv("DFTPAV_REF_SHAPE")) { // "quad": four trajectories per wave wherever the layout allows it
    if (e[0] == 'q') quad = allow_quad && reference_order_quad_supported(L, P, S);
    else quad = false;
  }
  pl.quad = 0;
  if (quad) {
    reference_order_quad_plan(L, P, B, n_cu, pl);
    return pl;
  }
  // several gear segments (solver_ref4m.hip)
  bool quadm = allow_quad && wave && L.M > 1 && reference_order_quadm_supported(L, P, S);
  if (const char *e = std::getenv("DFTPAV_REF_QUADM_OFF")) quadm = quadm && !(e[0] != 0 && e[0] != '0'); // developer knob: several segments stay with the WAVE shape
  if (const char *e = std::getenv("DFTPAV_REF_SHAPE")) {
    if (e[0] == 'q') quadm = allow_quad && reference_order_quadm_supported(L, P, S);
    else quadm = false;
  }
  if (quadm) {
    reference_order_quadm_plan(L, P, B, n_cu, pl);
    return pl;
  }
  pl.wave = wave ? 1 : 0;
  if (wave) {
    pl.threads = 64 * best_w;
    pl.lds = shared + (size_t)best_w * team_w;
    pl.wg_per_cu = best_wg;
    pl.slots = n_cu * best_wg; // persistent workgroups of a scheduled solve
    pl.slice = 128;
    if (const char *e = std::getenv("DFTPAV_REF_SLICE")) pl.slice = std::atoi(e);
    if (const char *e = std::getenv("DFTPAV_REF_SLOTS")) pl.slots = std::max(1, std::atoi(e)); // developer knob: persistent workgroups
  } else {
    int threads = B > 768 ? 128 : 256;
    if (const char *e = std::getenv("DFTPAV_REF_THREADS")) { // developer knob: whole waves, at most the launch bound
      const int t = std::atoi(e);
      if (t == 128 || t == 192 || t == 256) threads = t; // wave 1 has jobs of its own: at least two waves
    }
    const reford::Shape st = reford::make_shape(L, S, false);
    pl.threads = threads;
    pl.lds = shared + reford::lds_team_bytes(L, P.mem_size, st);
    pl.wg_per_cu = 0;
    pl.slots = 0;
    pl.slice = 0;
  }
  return pl;
}

#endif // DFTPAV_REF_PART != 2
template <int CAP, bool SUR, bool WAVE>
static hipError_t launch_ref_variant(const DevBatch *d_dev, int grid, int threads, size_t lds, int mode, const double *tabs, double *scratch, int source, int slice,
                                     hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&reford::ref_kernel<CAP, SUR, WAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((reford::ref_kernel<CAP, SUR, WAVE>), dim3(grid), dim3(threads), lds, stream, d_dev, mode, tabs, scratch, source, slice);
  return hipGetLastError();
}
template <int CAP>
hipError_t launch_ref_cap(bool sur, bool wave, const DevBatch *d_dev, int grid, int threads, size_t lds, int mode, const double *tabs, double *scratch,
                                 int source, int slice, hipStream_t stream) {
  if (sur) {
    if (wave) return launch_ref_variant<CAP, true, true>(d_dev, grid, threads, lds, mode, tabs, scratch, source, slice, stream);
    return launch_ref_variant<CAP, true, false>(d_dev, grid, threads, lds, mode, tabs, scratch, source, slice, stream);
  }
  if (wave) return launch_ref_variant<CAP, false, true>(d_dev, grid, threads, lds, mode, tabs, scratch, source, slice, stream);
  return launch_ref_variant<CAP, false, false>(d_dev, grid, threads, lds, mode, tabs, scratch, source, slice, stream);
}
#define DFTPAV_REF_CAP_ARGS bool, bool, const DevBatch *, int, int, size_t, int, const double *, double *, int, int, hipStream_t
#if DFTPAV_REF_PART == 1 // the wide kernels live in the other unit
extern template hipError_t launch_ref_cap<40>(DFTPAV_REF_CAP_ARGS);
extern template hipError_t launch_ref_cap<48>(DFTPAV_REF_CAP_ARGS);
extern template hipError_t launch_ref_cap<64>(DFTPAV_REF_CAP_ARGS);
#elif DFTPAV_REF_PART == 2
template hipError_t launch_ref_cap<40>(DFTPAV_REF_CAP_ARGS);
template hipError_t launch_ref_cap<48>(DFTPAV_REF_CAP_ARGS);
template hipError_t launch_ref_cap<64>(DFTPAV_REF_CAP_ARGS);
#endif
#if DFTPAV_REF_PART != 2
template <bool SUR>
static hipError_t launch_ref_variant_gen(const DevBatch *d_dev, int grid, int threads, size_t lds, int mode, const double *tabs, double *scratch, int source, int slice,
                                         hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&reford::ref_kernel<64, SUR, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((reford::ref_kernel<64, SUR, false, true>), dim3(grid), dim3(threads), lds, stream, d_dev, mode, tabs, scratch, source, slice);
  return hipGetLastError();
}
// scheduled != 0: a solve in the WAVE shape whose waves pop from the batch's ring (the caller has reset it)
hipError_t launch_solver_ref(const DevBatch &D, const DevBatch *d_dev, int mode, const double *tabs, double *scratch, const RefPlan &pl, int scheduled,
                             hipStream_t stream) {
  const bool wave = pl.wave != 0, sur = D.sur.S > 0;
  const int W = pl.threads / 64;
  int grid = wave ? (D.B + W - 1) / W : D.B, source = 0, slice = 0;
  if (wave && scheduled && mode == kModeSolve) {
    grid = pl.slots < grid ? pl.slots : grid;
    source = 1;
    slice = pl.slice;
  }
  if (const char *e = std::getenv("DFTPAV_REF_EXACT_DIV")) // test hook: true divisions in the recursion (its fallback for divisors beyond 2^+-500)
    if (std::atoi(e) != 0) source |= 2;
  if (std::getenv("DFTPAV_VERBOSE"))
    std::fprintf(stderr, "[dftpav] reference order, %s shape: grid %d x %d threads, %zu B of LDS, source %d slice %d\n", wave ? "WAVE" : "TEAM", grid, pl.threads,
                 pl.lds, source, slice);
  if (D.L.n > 64 || D.L.H > 5) { // (reference_order_plan keeps these in the TEAM shape)
    if (sur) return launch_ref_variant_gen<true>(d_dev, grid, pl.threads, pl.lds, mode, tabs, scratch, source, slice, stream);
    return launch_ref_variant_gen<false>(d_dev, grid, pl.threads, pl.lds, mode, tabs, scratch, source, slice, stream);
  }
  switch (reford::ref_cap_of(D.L.n)) {
  case 16: return launch_ref_cap<16>(sur, wave, d_dev, grid, pl.threads, pl.lds, mode, tabs, scratch, source, slice, stream);
  case 32: return launch_ref_cap<32>(sur, wave, d_dev, grid, pl.threads, pl.lds, mode, tabs, scratch, source, slice, stream);
  case 40: return launch_ref_cap<40>(sur, wave, d_dev, grid, pl.threads, pl.lds, mode, tabs, scratch, source, slice, stream);
  case 48: return launch_ref_cap<48>(sur, wave, d_dev, grid, pl.threads, pl.lds, mode, tabs, scratch, source, slice, stream);
  default: return launch_ref_cap<64>(sur, wave, d_dev, grid, pl.threads, pl.lds, mode, tabs, scratch, source, slice, stream);
  }
}
#endif // DFTPAV_REF_PART != 2

} // namespace dftpav
