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
#ifndef DFTPAV_REF_TL_MASK
#define DFTPAV_REF_TL_MASK 0
#endif
#ifndef DFTPAV_REF_EVAL_ATTR
#define DFTPAV_REF_EVAL_ATTR __forceinline__
#endif
#ifndef DFTPAV_REF_SUM_ALL_LANES
#define DFTPAV_REF_SUM_ALL_LANES 1 // measured: the chain on one lane + broadcast costs 637 cycles per history step against 546
#endif

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
__device__ __noinline__ mask_t surround_terms(const DevParams &P, const DevSurround &S, double t_now, double omg, double step, double t,
                                                const double beta0[6], const double beta1[6], double gama, int pieceid, int trajres,
                                                const double sigma[2], const double dsigma[2], const double ddsigma[2], const double ego_R[4],
                                                int singul_, int trajid, double trajtime, int Nseg, int t_first, gd_t rec) {
  const double B_h[4] = {0.0, -1.0, 1.0, 0.0}, B_hT[4] = {0.0, 1.0, -1.0, 0.0}; // traj_optimizer.cpp:1741-1742
  mask_t mask = 0ull;
  int first_active = -1;
  const double alpha = 100.0, d_min = P.surround_clearance + crt::log_cr(8.0) / alpha; // traj_optimizer.cpp:1336 (the reference: std::log(8.0))
  double temp0 = sqrt(dsigma[0] * dsigma[0] + dsigma[1] * dsigma[1]);
  double temp0_reci = (temp0 != 0.0) ? 1.0 / temp0 : 0.0;
  double temp3 = temp0_reci * temp0_reci;
  const int nE = 4, nO = 4;
  double totalPenalty = 0.0;

  for (int sur_id = 0; sur_id < S.S; sur_id++) {
    const SurTraj st_{S.durations + S.piece_off[sur_id], S.coeffs + 12 * (size_t)S.piece_off[sur_id], S.piece_off[sur_id + 1] - S.piece_off[sur_id], S.total[sur_id], S.start[sur_id]};
    const SurTraj *st = &st_;
    double offsettime = t_now - st->start_time + trajtime; // OPT:1367-1369
    double pt_time = offsettime + t;
    double surround_p[2], surround_v[2], surround_a[2];
    if (pt_time < st->duration) {
      traj_getPos(st, pt_time, surround_p);
      traj_getdSigma(st, pt_time, surround_v);
      traj_getddSigma(st, pt_time, surround_a);
    } else { // OPT:1379-1389
      double vd[2], pd[2];
      traj_getddSigma(st, st->duration, surround_a);
      double exceed_time = pt_time - st->duration;
      traj_getdSigma(st, st->duration, vd);
      surround_v[0] = vd[0] + exceed_time * surround_a[0];
      surround_v[1] = vd[1] + exceed_time * surround_a[1];
      traj_getPos(st, st->duration, pd);
      surround_p[0] = pd[0] + exceed_time * vd[0] + 0.5 * surround_a[0] * exceed_time * exceed_time;
      surround_p[1] = pd[1] + exceed_time * vd[1] + 0.5 * surround_a[1] * exceed_time * exceed_time;
    }
    {
      double dx = surround_p[0] - sigma[0], dy = surround_p[1] - sigma[1];
      if (sqrt(dx * dx + dy * dy) > P.veh_length_infl * 1.5) continue; // OPT:1393
    }
    double surround_R[4];
    traj_getR(st, pt_time, surround_R); // OPT:1410

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
      if (d_min + 1.38629436111989061883e+00 / alpha - best < -1.0e-9) continue;
    }
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
    if (costp <= 0) continue;
    double pena, penaD;
    smoothed_l1(costp, pena, penaD);
    totalPenalty += omg * step * P.wei_surround * pena;

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
    traj_getRdot(st, pt_time, Rud); // OPT:1599
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
__device__ __forceinline__ void load_planes(gcd_t cor, size_t pitch, int H, double pl[20]) {
#pragma unroll
  for (int k = 0; k < 5; k++) {
#pragma unroll
    for (int q = 0; q < 4; q++) pl[4 * k + q] = k < H ? cor[(size_t)(4 * k + q) * pitch] : 0.0; // (uniform: planes past H are not fetched)
  }
}
template <bool SUR>
__device__ __forceinline__ mask_t point_masks(const DevParams &P, const double cc_[12], int i, int N, int j, int K, double step, double s1, int singul_,
                                            double epis, int H, const double pl[20], gd_t sur_rec, const DevSurround &S, double t_now, double t_piece,
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
  double pn0[5], pn1[5], pq0[5], pq1[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    pn0[k] = pl[4 * k + 0];
    pn1[k] = pl[4 * k + 1];
    pq0[k] = pl[4 * k + 2];
    pq1[k] = pl[4 * k + 3];
  }
  // (vec_le_ holds the first vertex twice, traj_optimizer.cpp:1765-1775, and the reference tests it twice: the fifth vertex's
  // tests are the first's, expression for expression -- their bits are copied, not recomputed; capi.cpp fills vec_le[4] from
  // vec_le[0].  H <= 5, so 5 H <= 25 tests: collected in 32 bits.)
  unsigned cm = 0u;
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const double le0 = P.vec_le[v][0], le1 = P.vec_le[v][1];
    const double rl0 = ego_R[0] * le0 + ego_R[1] * le1;
    const double rl1 = ego_R[2] * le0 + ego_R[3] * le1;
    const double bpt0 = sigma[0] + rl0, bpt1 = sigma[1] + rl1;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const double violaPos = pn0[k] * (bpt0 - pq0[k]) + pn1[k] * (bpt1 - pq1[k]);
      if (k < H && violaPos > 0) cm |= 1u << (v * H + k);
    }
  }
  cm |= (cm & ((1u << H) - 1u)) << (4 * H);
  mask = (mask_t)cm;
  // ---- moving obstacles, traj_optimizer.cpp:636-638 (terms 5 H .. 5 H + S - 1)
  if (SUR && S.S > 0)
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
template <typename R>
__device__ __forceinline__ void point_emit(const DevParams &P, const PtState &st, int t, int H, int t0, gcd_t cor, size_t pitch, R r_) {
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
    // the half-plane and the vertex of this term, fetched again (they are what point_masks tested)
    const double on0 = cor[(size_t)(4 * k + 0) * pitch], on1 = cor[(size_t)(4 * k + 1) * pitch];
    const double q0 = cor[(size_t)(4 * k + 2) * pitch], q1 = cor[(size_t)(4 * k + 3) * pitch];
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

// TEAM shape: the point's tests, then a record per active term in the point's own slots rec[t][kRec] (global scratch)
template <bool SUR>
__device__ __forceinline__ mask_t point_terms(const DevParams &P, const double cc_[12], int i, int N, int j, int K, double step, double s1,
                                            int singul_, double epis, int H, gcd_t cor, size_t pitch, gd_t rec, const DevSurround &S,
                                            double t_now, double t_piece, int trajid, double trajtime) {
  PtState st;
  const int nS = SUR ? S.S : 0, tS0 = 5 * H, t0 = tS0 + nS;
  double pl[20];
  load_planes(cor, pitch, H, pl);
  const mask_t mask = point_masks<SUR>(P, cc_, i, N, j, K, step, s1, singul_, epis, H, pl, rec + (size_t)tS0 * kRec, S, t_now, t_piece, trajid, trajtime, st);
  for (mask_t m = mask; m;) {
    const int t = __builtin_ctzll(m);
    m &= m - 1;
    if (t >= tS0 && t < t0) continue; // a moving-obstacle term: surround_terms has written its record
    point_emit(P, st, t, H, t0, cor, pitch, rec + (size_t)t * kRec);
  }
  return mask;
}

// ------------------------------------------------ costFunctionCallback (traj_optimizer.cpp:206-350)
// x -> g (LDS), f in st[sF].  rec_b: this trajectory's term records [Npts][nterm][kRec].  The gear segments are independent
// up to the sums of :292-297 and the junction variables' gradients (:307-320), so every stage runs them side by side.
template <bool SUR, bool WAVE>
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
  for (int pt = tid; pt < Npts; pt += T) {
    const int p = D.pt_piece[pt], j = D.pt_j[pt];
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
    // trajtimes[sg] of traj_optimizer.cpp:230-234: 0, then the real duration of the PREVIOUS segment
    const double trajtime = (SUR && sg > 0) ? sm.seg[16 * (sg - 1)] : 0.0;
    sm.set_mask(pt, point_terms<SUR>(P, cc, lp, N, j, K, step, s1, singul_, D.epis, H, cor_b + pt, (size_t)D.NptsPad,
                                     rec_b + (size_t)pt * nterm * kRec, D.sur, D.t_now, SUR ? sm.pA[p] : 0.0, sg, trajtime));
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
#define DFTPAV_FMAC_BCAST(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define DFTPAV_FMAC_BCAST16 \
  DFTPAV_FMAC_BCAST(0) DFTPAV_FMAC_BCAST(1) DFTPAV_FMAC_BCAST(2) DFTPAV_FMAC_BCAST(3) DFTPAV_FMAC_BCAST(4) DFTPAV_FMAC_BCAST(5) DFTPAV_FMAC_BCAST(6) \
  DFTPAV_FMAC_BCAST(7) DFTPAV_FMAC_BCAST(8) DFTPAV_FMAC_BCAST(9) DFTPAV_FMAC_BCAST(10) DFTPAV_FMAC_BCAST(11) DFTPAV_FMAC_BCAST(12) \
  DFTPAV_FMAC_BCAST(13) DFTPAV_FMAC_BCAST(14) DFTPAV_FMAC_BCAST(15)
#define DFTPAV_FMAC_BCAST8 \
  DFTPAV_FMAC_BCAST(0) DFTPAV_FMAC_BCAST(1) DFTPAV_FMAC_BCAST(2) DFTPAV_FMAC_BCAST(3) DFTPAV_FMAC_BCAST(4) DFTPAV_FMAC_BCAST(5) DFTPAV_FMAC_BCAST(6) \
  DFTPAV_FMAC_BCAST(7)
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
#if DFTPAV_REF_SUM_ALL_LANES
  {
#else
  if (lane == 0) { // one lane reads and chains (an LDS read costs by the lanes it serves), the others take its result
#endif
    double v[CAP];
#pragma unroll
    for (int u = 0; u < CAP; u++) v[u] = buf[u];
#pragma unroll
    for (int u = 0; u < CAP; u++) s += v[u];
  }
  wave_lds_order(); // the buffer is free again
#if !DFTPAV_REF_SUM_ALL_LANES
  {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(s)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(s));
    s = __hiloint2double(hi, lo);
  }
#endif
  return s;
}

// ---- the two-loop recursion's view of the history: blocks of kPB stored pairs in registers, the next block in flight while
// one is worked on
constexpr int kPB = 8;
typedef double __attribute__((ext_vector_type(2))) d2_t;
typedef const d2_t __attribute__((address_space(1))) *gcd2_t;
typedef d2_t __attribute__((address_space(1))) *gd2_t;
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
#if DFTPAV_REF_TL_MASK
  // only the lanes of the sums take part (n <= CAP): an LDS read costs by the lanes it serves, and a step is 32 broadcast reads
  if (CAP < 64 && lane >= CAP) return dreg;
#endif
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

// ------------------------------------------------ the kernel
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
template <int CAP, bool SUR, bool WAVE>
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
#ifdef DFTPAV_REF_TWO_CALLS
    ref_eval<SUR, WAVE>(D, cor_b, rec_b, sm, sm.x, sm.g, pr);
    for (bool first = true;; first = false) {
      if (!first) ref_eval<SUR, WAVE>(D, cor_b, rec_b, sm, sm.x, sm.g, pr);
#else
    while (true) { // (one call site of the evaluation: the kernel is instruction-cache-sized as it is)
      ref_eval<SUR, WAVE>(D, cor_b, rec_b, sm, sm.x, sm.g, pr); // x0 / the trial point the trajectory was suspended on / the next trial point
#endif
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
      if (tid < 64) lbfgs_advance<CAP>(D, sm, hS, hR, lane, pr);
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
#if DFTPAV_REF_PART != 2
// what the layout must satisfy for the reference-order kernel (solver_ref.hip header)
bool reference_order_supported(const DevLayout &L, const DevParams &P, int S) {
  if (L.M < 1 || L.n > 64 || L.Npts >= (1 << 25)) return false;
  if (L.H < 1 || L.H > 5) return false; // a point's half-planes are held in five register slots (rectangles: H = 4)
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

// The launch shape of a batch (see the header).  TEAM: four waves per trajectory while the batch leaves CUs to spare (the
// parallel stages finish sooner: 70 against 73 ms at batch 32, 133 against 140 at 256), two for more.  WAVE: as many waves per
// workgroup as keep the most trajectories resident on a CU -- 8 waves of 256 registers (4 for the kernels that take 512), the
// LDS of the shared tables plus a team's part per wave.
RefPlan reference_order_plan(const DevLayout &L, const DevParams &P, int S, int B, int n_cu) {
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
  bool wave = best_w > 0 && B > 5 * n_cu;
  if (const char *e = std::getenv("DFTPAV_REF_SHAPE")) { // developer knob: "team" / "wave"
    if (e[0] == 't') wave = false;
    if (e[0] == 'w' && best_w > 0) wave = true;
  }
  if (const char *e = std::getenv("DFTPAV_REF_WAVES")) { // developer knob: waves per workgroup in the WAVE shape
    const int w = std::atoi(e);
    if (w >= 1 && w <= max_waves_cu && shared + (size_t)w * team_w <= budget) {
      best_w = w;
      best_wg = (int)std::min<size_t>((size_t)(max_waves_cu / w), budget / (shared + (size_t)w * team_w));
    }
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
