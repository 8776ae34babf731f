// traj_math.h — the per-constraint-point mathematics of the solve path, written
// once for host and device.
//
// Everything here is straight-line fp64 arithmetic built from + - * / sqrt, explicit
// fused multiply-adds (fma_) and comparisons only (transcendentals are the portable
// routines below, not libm),
// evaluated in the written left-to-right order with floating-point contraction
// disabled (-ffp-contract=off on both compilers).  gfx950 and x86-64 then
// produce the same bits (scripts/ieee_probe.hip: sqrt, /, 1/x, a*b+c identical
// on 4M random operands), which is what makes the L-BFGS iterate sequence of
// the GPU reproducible on a CPU: the reference solver is chaotic — one ulp on
// x0 changes its final cost by up to 16 % — so "same result"
// can only mean "same floating-point program".
//
// The kernels in solver.hip include this header; so does the *device-order*
// mode of the CPU oracle (oracle/dftpav_oracle_dev.cpp), which replays the
// kernel's summation order on the host.  The oracle's *literal* mode
// (oracle/dftpav_oracle.c) is independent code and cross-checks this file to
// rounding level.
#pragma once
#include "device_types.h"
#include "cr_trig.h"

// Keeps the instruction scheduler from interleaving the iterations of an unrolled loop (each iteration's temporaries then
// die before the next one starts).  Only where the kernel is out of registers; no effect on results, nothing on the host.
#if defined(__HIP_DEVICE_COMPILE__)
#define DFTPAV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define DFTPAV_SCHED_FENCE() ((void)0)
#endif

namespace dftpav {

// Fused multiply-add with a single rounding, written out where it is wanted: v_fma_f64 on gfx950 and the
// correctly rounded fma of the host (hardware or libm) — a defined IEEE operation with the same result
// everywhere, unlike compiler contraction, which stays off.  The per-point evaluation is made of
// multiply-add pairs; fusing them removes a quarter of its instructions and shortens every dependent chain.
DFTPAV_HD inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// a / b from the correctly rounded reciprocal y = 1 / b (Markstein): q0 = a y, r = a - b q0 (exact in an FMA),
// q = q0 + r y -- the correctly rounded quotient, the same bits as a / b (solver.hip: 2^31 pairs checked on
// gfx950), in 3 dependent instructions instead of the ~12 of the division expansion.  Used where one divisor
// serves many quotients.
DFTPAV_HD inline double div_rcp(double a, double b, double y) {
  const double q0 = a * y;
  const double r = fma_(-b, q0, a);
  return fma_(r, y, q0);
}

// ------------------------------------------------------------ portable math
DFTPAV_HD inline double p_abs(double x) { return x < 0.0 ? -x : x; }

// sin/cos after fdlibm's k_sin.c / k_cos.c polynomial kernels with a
// two-term Cody-Waite reduction by pi/2 for |x| < 2^20 (every heading angle of a
// sane trajectory) and the Payne-Hanek reduction of cr_trig.h beyond: the first
// trial point of a line search late in a hard solve can lie 1e10 away, junction
// angle included, and the reference evaluates a finite cost there and backs off
// (tests/test_gpu_parity.py::test_far_trial_points).  Accuracy ~1 ulp; what
// matters is that host and device run the same operations.
DFTPAV_HD inline void p_rem_pio2(double x, int &quad, double &y0, double &y1) {
  if (!(p_abs(x) < 0x1.0p+20)) {
    if (!(x - x == 0.0)) { // infinity, NaN: NaN, as libm
      quad = 0;
      y0 = x - x;
      y1 = 0.0;
      return;
    }
    crt::dd r;
    quad = crt::reduce_large(x, r) & 3;
    y0 = r.hi;
    y1 = r.lo;
    return;
  }
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;  // first 33 bits of pi/2
  const double pio2_1t = 6.07710050650619224932e-11; // pi/2 - pio2_1
  double fnr = x * invpio2;
  int n = (int)(fnr < 0.0 ? fnr - 0.5 : fnr + 0.5);
  double fn = (double)n;
  double r = x - fn * pio2_1;
  double w = fn * pio2_1t;
  y0 = r - w;
  y1 = (r - y0) - w;
  quad = n & 3;
}
DFTPAV_HD inline double p_ksin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
DFTPAV_HD inline double p_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}
DFTPAV_HD inline double p_sin(double x) {
  int q;
  double y0, y1;
  p_rem_pio2(x, q, y0, y1);
  switch (q) {
    case 0: return p_ksin(y0, y1);
    case 1: return p_kcos(y0, y1);
    case 2: return -p_ksin(y0, y1);
    default: return -p_kcos(y0, y1);
  }
}
DFTPAV_HD inline double p_cos(double x) {
  int q;
  double y0, y1;
  p_rem_pio2(x, q, y0, y1);
  switch (q) {
    case 0: return p_kcos(y0, y1);
    case 1: return -p_ksin(y0, y1);
    case 2: return -p_kcos(y0, y1);
    default: return p_ksin(y0, y1);
  }
}

// exp after fdlibm e_exp.c (argument reduction by ln2, degree-5 rational core);

// atan / atan2 after fdlibm's s_atan.c / e_atan2.c (argument reduction to [0, 7/16] by the four breakpoints,
// odd degree-21 polynomial), range tests written as comparisons.  Used for the heading of an output
// trajectory (Piece::getAngle, poly_traj_utils.hpp:237-244); inputs are finite.
DFTPAV_HD inline double p_atan(double x) {
  const double hi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
                        1.57079632679489655800e+00};
  const double lo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17,
                        6.12323399573676603587e-17};
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
               aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
               aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
  const bool neg = x < 0.0;
  double ax = neg ? -x : x;
  if (ax >= 7.378697629483820646e19) return neg ? -(hi[3] + lo[3]) : hi[3] + lo[3]; // |x| >= 2^66
  int id;
  if (ax < 0.4375) {
    if (ax < 1.862645149230957031e-09) return x; // |x| < 2^-29
    id = -1;
  } else if (ax < 1.1875) {
    if (ax < 0.6875) {
      id = 0;
      ax = (2.0 * ax - 1.0) / (2.0 + ax);
    } else {
      id = 1;
      ax = (ax - 1.0) / (ax + 1.0);
    }
  } else if (ax < 2.4375) {
    id = 2;
    ax = (ax - 1.5) / (1.0 + 1.5 * ax);
  } else {
    id = 3;
    ax = -1.0 / ax;
  }
  const double z = ax * ax, w = z * z;
  const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) {
    const double r = ax - ax * (s1 + s2);
    return neg ? -r : r;
  }
  const double r = hi[id] - ((ax * (s1 + s2) - lo[id]) - ax);
  return neg ? -r : r;
}
DFTPAV_HD inline double p_atan2(double y, double x) {
  const double pi = 3.1415926535897931160e+00, pi_lo = 1.2246467991473531772e-16, pi_o_2 = 1.5707963267948965580e+00;
  if (x == 1.0) return p_atan(y);
  const bool yneg = y < 0.0 || (y == 0.0 && 1.0 / y < 0.0), xneg = x < 0.0 || (x == 0.0 && 1.0 / x < 0.0);
  if (y == 0.0) return xneg ? (yneg ? -pi : pi) : y;
  if (x == 0.0) return yneg ? -pi_o_2 : pi_o_2;
  const double ay = yneg ? -y : y, ax = xneg ? -x : x;
  double z;
  if (ay > ax * 1.152921504606846976e18) z = pi_o_2 + 0.5 * pi_lo;       // |y/x| > 2^60
  else if (xneg && ay * 1.152921504606846976e18 < ax) z = 0.0;           // |y/x| < 2^-60, x < 0
  else z = p_atan(ay / ax);
  if (!xneg) return yneg ? -z : z;
  return yneg ? (z - pi_lo) - pi : pi - (z - pi_lo);
}

// the range here is alpha*(d - d0) <= 0 with |arg| up to a few hundred.
DFTPAV_HD inline double p_exp(double x) {
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x < -745.0) return 0.0;
  if (x > 709.0) return 1.0e308 * 1.0e308;
  double kr = invln2 * x;
  int k = (int)(kr < 0.0 ? kr - 0.5 : kr + 0.5);
  double t = (double)k;
  double hi = x - t * ln2HI;
  double lo = t * ln2LO;
  double xr = hi - lo;
  double tt = xr * xr;
  double c = xr - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  double y = 1.0 - ((lo - (xr * c) / (2.0 - c)) - hi);
  // scale by 2^k exactly (two steps keep the multiplier normal for k in [-1074, 1023])
  int k1 = k / 2, k2 = k - k1;
  union { double d; unsigned long long u; } a, b;
  a.u = (unsigned long long)(1023 + k1) << 52;
  b.u = (unsigned long long)(1023 + k2) << 52;
  return (y * a.d) * b.d;
}

// log after fdlibm e_log.c
DFTPAV_HD inline double p_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  union { double d; unsigned long long u; } v;
  v.d = x;
  int e = (int)((v.u >> 52) & 0x7ff) - 1023;
  v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; // mantissa in [1,2)
  double m = v.d;
  if (m > 1.41421356237309514547) { // sqrt(2): keep f = m-1 in [-0.293, 0.414]
    m = m * 0.5;
    e += 1;
  }
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  double dk = (double)e;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// ------------------------------------------------------------ scalar helpers
// positiveSmoothedL1, traj_optimizer.cpp:783-806
DFTPAV_HD inline void smoothed_l1(double x, double &f, double &df) {
  const double pe = 1.0e-4;
  const double half = 0.5 * pe;
  const double f3c = 1.0 / (pe * pe);
  const double f4c = -0.5 * f3c / pe;
  const double d2c = 3.0 * f3c;
  const double d3c = 4.0 * f4c;
  if (x < pe) {
    f = fma_(f4c, x, f3c) * x * x * x;
    df = fma_(d3c, x, d2c) * x * x;
  } else {
    f = x - half;
    df = 1.0;
  }
}

// VirtualT2RealT, traj_optimizer.cpp:371-379
DFTPAV_HD inline double virtual_to_real(double vt, double mini_T) {
  return vt > 0.0 ? ((0.5 * vt + 1.0) * vt + 1.0) + mini_T : 1.0 / ((0.5 * vt - 1.0) * vt + 1.0) + mini_T;
}
// d RealT / d VirtualT, traj_optimizer.cpp:405-416
DFTPAV_HD inline double virtual_to_real_grad(double VT) {
  if (VT > 0) return VT + 1.0;
  double den = (0.5 * VT - 1.0) * VT + 1.0;
  return (1.0 - VT) / (den * den);
}

// powers of the piece duration, poly_traj_utils.hpp:961-966. s[0..5] = t^k, s[6..11] = t^-k
DFTPAV_HD inline void duration_powers(double dt, double *s) {
  double t1 = dt, t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
  s[0] = 1.0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = t4; s[5] = t5;
  s[6] = 1.0 / 1.0; s[7] = 1.0 / t1; s[8] = 1.0 / t2; s[9] = 1.0 / t3; s[10] = 1.0 / t4; s[11] = 1.0 / t5;
}

// jerk energy of one piece and its partials (poly_traj_utils.hpp:998-1035).
// c: 6x2 block c[2k+d]; t: powers t^k; gc receives d(energy)/dc (rows 0..2 zero).
DFTPAV_HD inline void piece_smoothness(const double *c, const double *t, double &energy, double &gdT, double *gc) {
  double c3x = c[6], c3y = c[7], c4x = c[8], c4y = c[9], c5x = c[10], c5y = c[11];
  double n33 = c3x * c3x + c3y * c3y, n44 = c4x * c4x + c4y * c4y, n55 = c5x * c5x + c5y * c5y;
  double d43 = c4x * c3x + c4y * c3y, d53 = c5x * c3x + c5y * c3y, d54 = c5x * c4x + c5y * c4y;
  energy = 36.0 * n33 * t[1] + 144.0 * d43 * t[2] + 192.0 * n44 * t[3] + 240.0 * d53 * t[3] + 720.0 * d54 * t[4] +
           720.0 * n55 * t[5];
  gdT = 36.0 * n33 + 288.0 * d43 * t[1] + 576.0 * n44 * t[2] + 720.0 * d53 * t[2] + 2880.0 * d54 * t[3] +
        3600.0 * n55 * t[4];
  for (int d = 0; d < 2; d++) {
    double c3 = c[6 + d], c4 = c[8 + d], c5 = c[10 + d];
    gc[10 + d] = 240.0 * c3 * t[3] + 720.0 * c4 * t[4] + 1440.0 * c5 * t[5];
    gc[8 + d] = 144.0 * c3 * t[2] + 384.0 * c4 * t[3] + 720.0 * c5 * t[4];
    gc[6 + d] = 72.0 * c3 * t[1] + 144.0 * c4 * t[2] + 240.0 * c5 * t[3];
    gc[d] = 0.0;
    gc[2 + d] = 0.0;
    gc[4 + d] = 0.0;
  }
}

// k-th entries of beta0, beta1, beta2 at offset s1 (traj_optimizer.cpp:505-507)
DFTPAV_HD inline void beta_row(int k, double s1, double &b0, double &b1, double &b2) {
  double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
  switch (k) {
    case 0: b0 = 1.0; b1 = 0.0; b2 = 0.0; break;
    case 1: b0 = s1; b1 = 1.0; b2 = 0.0; break;
    case 2: b0 = s2; b1 = 2.0 * s1; b2 = 2.0; break;
    case 3: b0 = s3; b1 = 3.0 * s2; b2 = 6.0 * s1; break;
    case 4: b0 = s4; b1 = 4.0 * s3; b2 = 12.0 * s2; break;
    default: b0 = s5; b1 = 5.0 * s4; b2 = 20.0 * s3; break;
  }
}

// What one constraint point adds to its piece: the 12 entries of gdC (row k, dimension d at v[2 k + d]), gdT (v[12])
// and the cost (v[13]), from the point's subtotals o = {d/dsigma (2), d/dsigma' (2), d/dsigma'' (2), gdT, cost}:
// beta0[k] a_d + beta1[k] b_d + beta2[k] c_d with the beta vectors of traj_optimizer.cpp:505-507 at offset s1.
DFTPAV_HD inline void point_contributions(double s1, const double o[8], double v[14]) {
  const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
  const double b0[6] = {1.0, s1, s2, s3, s4, s5};
  const double b1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
  const double b2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
  for (int k = 0; k < 6; k++) {
    v[2 * k] = fma_(b2[k], o[4], fma_(b1[k], o[2], b0[k] * o[0]));
    v[2 * k + 1] = fma_(b2[k], o[5], fma_(b1[k], o[3], b0[k] * o[1]));
  }
  v[12] = o[6];
  v[13] = o[7];
}

// The 6x2 coefficient block of a piece into registers, all twelve reads in flight before the first is used (left to
// itself the compiler reads them pair by pair, each pair a round trip to LDS in front of the multiply-adds that want it).
DFTPAV_HD inline void load_piece_coeffs(const double *cc, double c[12]) {
  for (int k = 0; k < 12; k++) c[k] = cc[k];
#if defined(__HIP_DEVICE_COMPILE__)
  for (int k = 0; k < 12; k++) asm volatile("" : "+v"(c[k]));
#endif
}

// 2x2 helpers, m = {m00, m01, m10, m11}
DFTPAV_HD inline void mat_vec(const double m[4], const double v[2], double o[2]) {
  o[0] = m[0] * v[0] + m[1] * v[1];
  o[1] = m[2] * v[0] + m[3] * v[1];
}
DFTPAV_HD inline void mat_mat(const double a[4], const double b[4], double o[4]) {
  o[0] = a[0] * b[0] + a[1] * b[2];
  o[1] = a[0] * b[1] + a[1] * b[3];
  o[2] = a[2] * b[0] + a[3] * b[2];
  o[3] = a[2] * b[1] + a[3] * b[3];
}

// ------------------------------------- moving-obstacle trajectories (R12)
// Piece evaluators, poly_traj_utils.hpp:77-112,179-211; locatePieceIdx :510-528.
// The tables are reached through a view SV (DevSurround, or the kernels' SurLds whose pointers carry the LDS address
// space): S, piece_off, durations, theta / has_theta(), total, start, rate(u), load_piece(k, c), end_state(u, ...),
// far_from_piece(k, sigma, r), has_bbox(), load_box(k, bb).  The 2x6 block of the located
// piece is fetched into registers in one go (twelve independent loads, one wait) before any arithmetic touches it.
struct SurEval {
  double c[12]; // 2x6 col-major, col 0 = t^5
  double t;     // local time inside the piece
};
// idx = number of pieces k of obstacle u with t > theta[k] (theta is non-decreasing): the piece locatePieceIdx stops at.
// Started from a guess -- the index a trajectory of equal pieces would have -- and walked to the answer: for any guess
// the same idx as a bisection, but one round trip to the table instead of log2(np) dependent ones when the guess is
// right or off by one.  Needs S.has_theta().
template <class SV>
DFTPAV_HD inline int sur_index(const SV &S, int u, double t) {
  const int p0 = S.piece_off[u], np = S.piece_off[u + 1] - p0;
  const auto th = S.theta + p0;
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
// the local time inside piece idx by the reference's subtractions (their operands do not depend on a comparison any
// more, so the loads go out together instead of one round trip per piece), and the piece's coefficient block
template <class SV>
DFTPAV_HD inline void sur_local(const SV &S, int u, int idx, double t, SurEval &e) {
  const int p0 = S.piece_off[u], np = S.piece_off[u + 1] - p0;
  const auto durs = S.durations + p0;
  const int nsub = idx < np ? idx : np;
  int k = 0;
  for (; k + 8 <= nsub; k += 8) { // the loads of a block first, then the reference's subtractions in their order
    const double d0 = durs[k], d1 = durs[k + 1], d2 = durs[k + 2], d3 = durs[k + 3];
    const double d4 = durs[k + 4], d5 = durs[k + 5], d6 = durs[k + 6], d7 = durs[k + 7];
    t -= d0;
    t -= d1;
    t -= d2;
    t -= d3;
    t -= d4;
    t -= d5;
    t -= d6;
    t -= d7;
  }
  for (; k + 4 <= nsub; k += 4) {
    const double d0 = durs[k], d1 = durs[k + 1], d2 = durs[k + 2], d3 = durs[k + 3];
    t -= d0;
    t -= d1;
    t -= d2;
    t -= d3;
  }
  for (; k < nsub; k++) t -= durs[k];
  if (idx == np) {
    idx--;
    t += durs[idx];
  }
  S.load_piece(p0 + idx, e.c);
  e.t = t;
}
template <class SV>
DFTPAV_HD inline void sur_locate(const SV &S, int u, double t, SurEval &e) {
  if (S.has_theta()) {
    sur_local(S, u, sur_index(S, u, t), t, e);
    return;
  }
  const int p0 = S.piece_off[u], np = S.piece_off[u + 1] - p0;
  const auto durs = S.durations + p0;
  int idx;
  double dur = 0.0;
  for (idx = 0; idx < np && t > (dur = durs[idx]); idx++) t -= dur;
  if (idx == np) {
    idx--;
    t += durs[idx];
  }
  S.load_piece(p0 + idx, e.c);
  e.t = t;
}
DFTPAV_HD inline void piece_pos(const SurEval &e, double o[2]) {
  o[0] = 0.0; o[1] = 0.0;
  double tn = 1.0;
  for (int i = 5; i >= 0; i--) {
    o[0] += tn * e.c[2 * i];
    o[1] += tn * e.c[2 * i + 1];
    tn *= e.t;
  }
}
DFTPAV_HD inline void piece_vel(const SurEval &e, double o[2]) {
  o[0] = 0.0; o[1] = 0.0;
  double tn = 1.0;
  int n = 1;
  for (int i = 4; i >= 0; i--) {
    o[0] += n * tn * e.c[2 * i];
    o[1] += n * tn * e.c[2 * i + 1];
    tn *= e.t;
    n++;
  }
}
DFTPAV_HD inline void piece_acc(const SurEval &e, double o[2]) {
  o[0] = 0.0; o[1] = 0.0;
  double tn = 1.0;
  int m = 1, n = 2;
  for (int i = 3; i >= 0; i--) {
    o[0] += m * n * tn * e.c[2 * i];
    o[1] += m * n * tn * e.c[2 * i + 1];
    tn *= e.t;
    m++;
    n++;
  }
}

// log_sum_exp, traj_optimizer.cpp:1686-1707 (mutates v into the exp weights)
template <int NV>
DFTPAV_HD inline double log_sum_exp(double alpha, double *v, double &exp_sum) {
  double d0 = v[0];
  if (alpha > 0) {
    for (int j = 1; j < NV; j++) d0 = v[j] > d0 ? v[j] : d0;
  } else {
    for (int j = 1; j < NV; j++) d0 = v[j] < d0 ? v[j] : d0;
  }
  exp_sum = 0;
  for (int j = 0; j < NV; j++) {
    v[j] = p_exp(alpha * (v[j] - d0));
    exp_sum += v[j];
    DFTPAV_SCHED_FENCE();
  }
  return p_log(exp_sum) / alpha + d0;
}

// dynamicObsGradCostP, traj_optimizer.cpp:1311-1684, for ONE moving obstacle u (the body of its loop over the obstacles).
// dyn_obstacle_near: position / velocity / acceleration of obstacle u at the time of the constraint point
// (traj_optimizer.cpp:1367-1389) and the distance gate of :1393; false = the obstacle is skipped at this point.
struct DynObs {
  double sp[2], sv[2], sa[2];
  double pt_time;
};
// getPos / getVel / getAcc at the obstacle's total duration (traj_optimizer.cpp:1381-1383)
template <class SV>
DFTPAV_HD inline void sur_end_state(const SV &S, int u, double pd[2], double vd[2], double ad[2]) {
  SurEval e;
  sur_locate(S, u, S.total[u], e);
  piece_acc(e, ad);
  piece_vel(e, vd);
  piece_pos(e, pd);
}
DFTPAV_HD inline void DevSurround::end_state(int u, double pd[2], double vd[2], double ad[2]) const { sur_end_state(*this, u, pd, vd, ad); }
template <class SV>
DFTPAV_HD inline bool dyn_obstacle_near(const DevParams &P, const SV &S, int u, double t_now, double t, double trajtime,
                                        const double sigma[2], DynObs &ob) {
  double dur = S.total[u];
  double offsettime = t_now - S.start[u] + trajtime; // traj_optimizer.cpp:1367-1369
  double pt_time = offsettime + t;
  double *sp = ob.sp, *sv = ob.sv, *sa = ob.sa;
  if (pt_time < dur) {
    SurEval e;
    if (S.has_theta()) {
      // Before anything is evaluated: if the point is farther than the gate's radius from the box that holds the whole
      // piece the obstacle is on (DevSurround::bbox, with a margin far above rounding), the distance test below fails.
      // (only for pt_time >= 0: an obstacle whose trajectory starts after t_now is extrapolated backwards along the first
      // piece's quintic, Trajectory::locatePieceIdx returns piece 0 with a negative local time, traj_optimizer.cpp:1374-1378 --
      // that position is outside the hull box of the piece)
      const int idx = sur_index(S, u, pt_time);
      if (pt_time >= 0.0 && idx < S.piece_off[u + 1] - S.piece_off[u] && S.far_from_piece(S.piece_off[u] + idx, sigma, P.veh_length_infl * 1.5 + 1e-6))
        return false;
      sur_local(S, u, idx, pt_time, e);
    } else {
      sur_locate(S, u, pt_time, e);
    }
    piece_pos(e, sp);
    piece_vel(e, sv);
    piece_acc(e, sa);
  } else { // traj_optimizer.cpp:1379-1389
    // position, velocity, acceleration of the obstacle at the end of its trajectory: the same for every point
    // (sur_end_state below; the kernels form them once per launch and keep them beside the other tables)
    double vd[2], pd[2];
    S.end_state(u, pd, vd, sa);
    double ex = pt_time - dur;
    sv[0] = vd[0] + ex * sa[0];
    sv[1] = vd[1] + ex * sa[1];
    sp[0] = pd[0] + ex * vd[0] + 0.5 * sa[0] * ex * ex;
    sp[1] = pd[1] + ex * vd[1] + 0.5 * sa[1] * ex * ex;
  }
  ob.pt_time = pt_time;
  double dx = sp[0] - sigma[0], dy = sp[1] - sigma[1];
  return !(sqrt(dx * dx + dy * dy) > P.veh_length_infl * 1.5); // traj_optimizer.cpp:1393
}
// DevParams::edge_*: the expressions of traj_optimizer.cpp:1419-1421 (dl = vec_le[e+1] - vec_le[e], dl.norm()) evaluated once
// on the host; IEEE subtraction, multiplication, sqrt and division are correctly rounded there as on the device.
inline void fill_footprint_edges(DevParams &P) {
  for (int e = 0; e < 4; e++) {
    const double dx = P.vec_le[e + 1][0] - P.vec_le[e][0], dy = P.vec_le[e + 1][1] - P.vec_le[e][1];
    P.edge_d[e][0] = dx;
    P.edge_d[e][1] = dy;
    P.edge_len[e] = sqrt(dx * dx + dy * dy);
    P.edge_rlen[e] = 1 / P.edge_len[e];
  }
}

// Products with the constant quarter turn B_h = [0 -1; 1 0] (traj_optimizer.cpp:1330-1333) written out: every entry of such a
// product is one entry of the other factor, with or without its sign -- 0 * x + (-1) * y is -y exactly -- so no arithmetic
// is spent on them.  m = {m00, m01, m10, m11}.
DFTPAV_HD inline void bh_times(const double x[4], double o[4]) { // B_h x
  o[0] = -x[2]; o[1] = -x[3]; o[2] = x[0]; o[3] = x[1];
}
DFTPAV_HD inline void times_bh(const double x[4], double o[4]) { // x B_h
  o[0] = x[1]; o[1] = -x[0]; o[2] = x[3]; o[3] = -x[2];
}
DFTPAV_HD inline void times_bhT(const double x[4], double o[4]) { // x B_h^T
  o[0] = -x[1]; o[1] = x[0]; o[2] = -x[3]; o[3] = x[2];
}

// The penalty of obstacle u at the point and its gradients: adds d/dsigma into A, d/dsigma' into Bv, the duration
// gradient into gdT; returns the cost (0 when the smoothed distance stays above the clearance).
// Every quotient x / d whose divisor recurs (the edge lengths, the exponential sums, the obstacle's speed) is formed by
// div_rcp from one reciprocal of d: the correctly rounded quotient, i.e. the bits of x / d, in 3 instructions instead of
// the ~10 of a division.  Additions keep the reference's order.
template <class SV>
DFTPAV_HD inline double dynamic_pair(const DevParams &P, const SV &S, int u, const DynObs &ob, double omg, double step,
                                     double gama, int pieceid, int trajres, const double sigma[2],
                                     const double dsigma[2], const double ddsigma[2], const double ego_R[4],
                                     int singul_, int trajid, int Ntraj, double A[2], double Bv[2],
                                     double &gdT) {
  const double alpha = 100.0;
  const double ln8 = 2.07944154167983574766e+00; // std::log(8.0)
  const double d_min = P.surround_clearance + ln8 / alpha; // traj_optimizer.cpp:1336
  double temp0 = sqrt(dsigma[0] * dsigma[0] + dsigma[1] * dsigma[1]);
  double temp0_reci = (temp0 != 0.0) ? 1.0 / temp0 : 0.0;
  double temp3 = temp0_reci * temp0_reci;
  double totalPenalty = 0.0;
  const double(*vle)[2] = P.vec_le; // vec_lo_ == vec_le_, traj_optimizer.cpp:1769
  const double *sp = ob.sp, *sv = ob.sv;
  const double pt_time = ob.pt_time;
  {
    // the four edges of the footprint: direction, length, 1 / length (the same for the ego vehicle and the obstacle;
    // batch constants, so they arrive as scalars instead of being formed from vec_le for every pair)
    const double(*edl)[2] = P.edge_d;
    const double *dln = P.edge_len, *dlni = P.edge_rlen;
    // getR / getRdot extrapolate the last polynomial piece past the duration (traj_optimizer.cpp:1410,1599)
    double sR[4], Rud[4];
    {
      SurEval e;
      sur_locate(S, u, pt_time, e);
      double v[2], a[2];
      piece_vel(e, v);
      piece_acc(e, a);
      double nv = sqrt(v[0] * v[0] + v[1] * v[1]);
      const double rnv = 1.0 / nv;
      const double v0n = div_rcp(v[0], nv, rnv), v1n = div_rcp(v[1], nv, rnv);
      sR[0] = v0n; sR[1] = -v1n; sR[2] = v1n; sR[3] = v0n;
      double nv3 = nv * nv * nv; // pow(norm, 3), poly_traj_utils.hpp:109
      const double rnv3 = 1.0 / nv3;
      double va = v[0] * a[0] + v[1] * a[1];
      const double a0n = div_rcp(a[0], nv, rnv), a1n = div_rcp(a[1], nv, rnv);
      const double v0c = div_rcp(v[0], nv3, rnv3), v1c = div_rcp(v[1], nv3, rnv3);
      Rud[0] = (a0n - v0c * va);
      Rud[1] = (-a1n - (-v1c) * va);
      Rud[2] = (a1n - v1c * va);
      Rud[3] = (a0n - v0c * va);
    }
    DFTPAV_SCHED_FENCE();

    // d(R l)/d(sigma') for a body-frame vector l (the F matrices of traj_optimizer.cpp:1423-1440).  They are formed where
    // they are used (the gradient loops below) from the same expressions, instead of being kept for all four edges: 32
    // doubles fewer are live through the log-sum-exp stage, where the GPU kernel is out of registers.
    auto f_matrix = [&](const double l[2], const double Rl[2], double F[4]) {
      double LT[4] = {l[0], l[1], -l[1], l[0]};
      F[0] = singul_ * LT[0] * temp0_reci - dsigma[0] * Rl[0] * temp3;
      F[1] = singul_ * LT[1] * temp0_reci - dsigma[0] * Rl[1] * temp3;
      F[2] = singul_ * LT[2] * temp0_reci - dsigma[1] * Rl[0] * temp3;
      F[3] = singul_ * LT[3] * temp0_reci - dsigma[1] * Rl[1] * temp3;
    };
    double BRego[4], BRsur[4], BRud[4];
    bh_times(ego_R, BRego);
    bh_times(sR, BRsur);
    bh_times(Rud, BRud);
    double s2e_sum[4], d_test[8];
    double egoN[4][2], dUo[4][4];
    double dUt4[4], dEt4[4];
    // The geometry of the eight separating directions first (traj_optimizer.cpp:1417-1461, 1464-1496 without their
    // log_sum_exp), the 40 exponentials and 9 logarithms of the three-level soft min / max afterwards: the same values in
    // another instruction order.
    for (int e = 0; e < 4; e++) { // traj_optimizer.cpp:1417-1461
      const double *le = vle[e];
      const double dl[2] = {edl[e][0], edl[e][1]};
      double Rle[2];
      mat_vec(ego_R, le, Rle);
      double Ht[2];
      mat_vec(BRego, dl, Ht);
      Ht[0] *= dlni[e];
      Ht[1] *= dlni[e];
      egoN[e][0] = Ht[0];
      egoN[e][1] = Ht[1];
      double w[2] = {sp[0] - sigma[0] - Rle[0], sp[1] - sigma[1] - Rle[1]};
      dUt4[e] = Ht[0] * w[0] + Ht[1] * w[1];
      double HtR[2] = {Ht[0] * sR[0] + Ht[1] * sR[2], Ht[0] * sR[1] + Ht[1] * sR[3]};
      for (int o = 0; o < 4; o++) dUo[e][o] = HtR[0] * vle[o][0] + HtR[1] * vle[o][1];
    }
    double e2s_sum[4];
    double surN[4][2], dEe[4][4];
    for (int o = 0; o < 4; o++) { // traj_optimizer.cpp:1464-1496
      const double *lo = vle[o];
      const double dl[2] = {edl[o][0], edl[o][1]};
      double Ht[2], Rlo[2];
      mat_vec(BRsur, dl, Ht);
      Ht[0] *= dlni[o];
      Ht[1] *= dlni[o];
      surN[o][0] = Ht[0];
      surN[o][1] = Ht[1];
      mat_vec(sR, lo, Rlo);
      double w[2] = {sigma[0] - sp[0] - Rlo[0], sigma[1] - sp[1] - Rlo[1]};
      dEt4[o] = Ht[0] * w[0] + Ht[1] * w[1];
      double HtR[2] = {Ht[0] * ego_R[0] + Ht[1] * ego_R[2], Ht[0] * ego_R[1] + Ht[1] * ego_R[3]};
      for (int e = 0; e < 4; e++) dEe[o][e] = HtR[0] * vle[e][0] + HtR[1] * vle[e][1];
    }
    DFTPAV_SCHED_FENCE();
    {
      // A bound before any exponential.  With m_k = min_j v_kj: log_sum_exp(-alpha, v_k) lies in [m_k - ln 4 / alpha, m_k] (its sum
      // of four exponentials lies in [1, 4]) and log_sum_exp(alpha, d) >= max_k d_k, hence
      //     costp = d_min - log_sum_exp(alpha, d_test)  <=  d_min + ln 4 / alpha - max_k (m_k + t_k).
      // When that is below -1e-9 (the roundings of the full evaluation are 1e-14) the full evaluation returns 0.0 at
      // `costp <= 0` below -- and so does this, 40 exponentials and 9 logarithms earlier.  Two footprints whose centres pass the
      // distance gate of :1393 (7 m) are rarely within the 0.42 m at which the penalty starts.
      double best = -1.0e300;
      for (int k = 0; k < 4; k++) {
        double mU = dUo[k][0], mE = dEe[k][0];
        for (int j = 1; j < 4; j++) {
          mU = dUo[k][j] < mU ? dUo[k][j] : mU;
          mE = dEe[k][j] < mE ? dEe[k][j] : mE;
        }
        const double a = mU + dUt4[k], b = mE + dEt4[k];
        best = a > best ? a : best;
        best = b > best ? b : best;
      }
      const double ln4 = 1.38629436111989061883e+00;
      if (d_min + ln4 / alpha - best < -1.0e-9) return 0.0;
    }
    for (int e = 0; e < 4; e++) {
      double es;
      d_test[e] = log_sum_exp<4>(-alpha, dUo[e], es) + dUt4[e];
      s2e_sum[e] = es;
      DFTPAV_SCHED_FENCE();
    }
    for (int o = 0; o < 4; o++) {
      double es;
      d_test[4 + o] = log_sum_exp<4>(-alpha, dEe[o], es) + dEt4[o];
      e2s_sum[o] = es;
      DFTPAV_SCHED_FENCE();
    }
    DFTPAV_SCHED_FENCE();
    double exp_sum_d = 0;
    double costp = d_min - log_sum_exp<8>(alpha, d_test, exp_sum_d); // traj_optimizer.cpp:1498-1502
    if (costp <= 0) return 0.0;
    double pena, penaD;
    smoothed_l1(costp, pena, penaD);
    totalPenalty += omg * step * P.wei_surround * pena;

    // the weights of the three log-sum-exp levels, each quotient formed once
    const double r_esd = 1.0 / exp_sum_d;
    double wd[8];
    for (int k = 0; k < 8; k++) wd[k] = div_rcp(d_test[k], exp_sum_d, r_esd);
    for (int e = 0; e < 4; e++) {
      const double r = 1.0 / s2e_sum[e];
      for (int o = 0; o < 4; o++) dUo[e][o] = div_rcp(dUo[e][o], s2e_sum[e], r);
    }
    for (int o = 0; o < 4; o++) {
      const double r = 1.0 / e2s_sum[o];
      for (int e = 0; e < 4; e++) dEe[o][e] = div_rcp(dEe[o][e], e2s_sum[o], r);
    }
    DFTPAV_SCHED_FENCE();

    double pGs[2] = {0.0, 0.0}; // traj_optimizer.cpp:1511-1523
    for (int e = 0; e < 4; e++) {
      double w = wd[e];
      pGs[0] -= w * (-egoN[e][0]);
      pGs[1] -= w * (-egoN[e][1]);
    }
    for (int o = 0; o < 4; o++) {
      double w = wd[o + 4];
      pGs[0] -= w * surN[o][0];
      pGs[1] -= w * surN[o][1];
    }
    double pGds[2] = {0.0, 0.0}; // traj_optimizer.cpp:1528-1573
    for (int e = 0; e < 4; e++) {
      const double *le = vle[e];
      const double dl[2] = {edl[e][0], edl[e][1]};
      const double dn = dln[e], rdn = dlni[e];
      double Rle[2], Rdl[2], Fdl_e[4], Fl_e[4];
      mat_vec(ego_R, le, Rle);
      mat_vec(ego_R, dl, Rdl);
      f_matrix(dl, Rdl, Fdl_e);
      f_matrix(le, Rle, Fl_e);
      double uu[2] = {-sp[0] + sigma[0] + Rle[0], -sp[1] + sigma[1] + Rle[1]};
      double FB[4], t1[2], FlB[4], FlBR[4], t2[2];
      times_bh(Fdl_e, FB);
      mat_vec(FB, uu, t1);
      times_bh(Fl_e, FlB);
      mat_mat(FlB, ego_R, FlBR);
      mat_vec(FlBR, dl, t2);
      double pdU[2] = {div_rcp(t1[0] - t2[0], dn, rdn), div_rcp(t1[1] - t2[1], dn, rdn)};
      double FBT[4];
      times_bhT(Fdl_e, FBT);
      for (int o = 0; o < 4; o++) {
        double Rlo[2], q[2];
        mat_vec(sR, vle[o], Rlo);
        mat_vec(FBT, Rlo, q);
        q[0] = div_rcp(q[0], dn, rdn);
        q[1] = div_rcp(q[1], dn, rdn);
        double w = dUo[e][o];
        pdU[0] += w * q[0];
        pdU[1] += w * q[1];
        DFTPAV_SCHED_FENCE();
      }
      double w = wd[e];
      pGds[0] -= w * pdU[0];
      pGds[1] -= w * pdU[1];
      DFTPAV_SCHED_FENCE();
    }
    for (int o = 0; o < 4; o++) {
      const double dl[2] = {edl[o][0], edl[o][1]};
      const double dn = dln[o], rdn = dlni[o];
      double pdE[2] = {0.0, 0.0};
      for (int e = 0; e < 4; e++) {
        double FB[4], FBR[4], q[2], Rle[2], Fl_e[4];
        mat_vec(ego_R, vle[e], Rle);
        f_matrix(vle[e], Rle, Fl_e);
        times_bh(Fl_e, FB);
        mat_mat(FB, sR, FBR);
        mat_vec(FBR, dl, q);
        q[0] = div_rcp(q[0], dn, rdn);
        q[1] = div_rcp(q[1], dn, rdn);
        double w = dEe[o][e];
        pdE[0] += w * q[0];
        pdE[1] += w * q[1];
        DFTPAV_SCHED_FENCE();
      }
      double w = wd[o + 4];
      pGds[0] -= w * pdE[0];
      pGds[1] -= w * pdE[1];
      DFTPAV_SCHED_FENCE();
    }
    DFTPAV_SCHED_FENCE();
    double pGtbar = (pGs[0] * dsigma[0] + pGs[1] * dsigma[1]) + (pGds[0] * ddsigma[0] + pGds[1] * ddsigma[1]);

    double pGthat = 0.0; // traj_optimizer.cpp:1586-1646
    for (int e = 0; e < 4; e++) {
      const double *Hn = egoN[e];
      double acc = Hn[0] * sv[0] + Hn[1] * sv[1];
      double HtRd[2] = {Hn[0] * Rud[0] + Hn[1] * Rud[2], Hn[0] * Rud[1] + Hn[1] * Rud[3]};
      for (int o = 0; o < 4; o++) {
        double ptv = HtRd[0] * vle[o][0] + HtRd[1] * vle[o][1];
        acc += dUo[e][o] * ptv;
      }
      pGthat -= wd[e] * acc;
      DFTPAV_SCHED_FENCE();
    }
    for (int o = 0; o < 4; o++) {
      const double *lo = vle[o];
      const double dl[2] = {edl[o][0], edl[o][1]};
      const double dn = dln[o], rdn = dlni[o];
      double a1[2], a2[2], Rlo[2], Rdlo[2];
      mat_vec(BRud, dl, a1);
      mat_vec(BRsur, dl, a2);
      mat_vec(sR, lo, Rlo);
      mat_vec(Rud, lo, Rdlo);
      double w1[2] = {sigma[0] - sp[0] - Rlo[0], sigma[1] - sp[1] - Rlo[1]};
      double w2[2] = {-sv[0] - Rdlo[0], -sv[1] - Rdlo[1]};
      double acc = (div_rcp(a1[0], dn, rdn) * w1[0] + div_rcp(a1[1], dn, rdn) * w1[1]) +
                   (div_rcp(a2[0], dn, rdn) * w2[0] + div_rcp(a2[1], dn, rdn) * w2[1]);
      for (int e = 0; e < 4; e++) {
        double Rle[2];
        mat_vec(ego_R, vle[e], Rle);
        double r1[2] = {Rle[1], -Rle[0]}; // Rle^T B_h
        double r2[2] = {r1[0] * Rud[0] + r1[1] * Rud[2], r1[0] * Rud[1] + r1[1] * Rud[3]};
        double ptv = div_rcp(r2[0] * dl[0] + r2[1] * dl[1], dn, rdn);
        acc += dEe[o][e] * ptv;
        DFTPAV_SCHED_FENCE();
      }
      pGthat -= wd[o + 4] * acc;
      DFTPAV_SCHED_FENCE();
    }

    double gradViolaPt = gama * pGtbar; // traj_optimizer.cpp:1649-1676
    double scale = omg * step * P.wei_surround * penaD;
    A[0] += scale * pGs[0];
    A[1] += scale * pGs[1];
    Bv[0] += scale * pGds[0];
    Bv[1] += scale * pGds[1];
    gdT += omg * P.wei_surround * (pena / trajres + penaD * gradViolaPt * step);
    gdT += omg * step * P.wei_surround * pGthat * penaD * pieceid;
    gdT += omg * step * P.wei_surround * gama * pGthat * penaD;
    for (int idx = 0; idx < trajid; idx++) gdT += omg * step * P.wei_surround * pGthat * penaD * Ntraj;
  }
  return totalPenalty;
}

// -------------------------------------------------- one constraint point
struct SampleIn {
  const double *cc; // 6x2 coefficient block of the piece, cc[2k+d]
  double s1;        // accumulated sample offset (s1 += step, traj_optimizer.cpp:513)
  int j, K;         // sample index and resolution of the piece
  int lp, N;        // piece index inside its segment, pieces of the segment
  double dt;        // piece duration
  int singul;       // +1 forward / -1 reverse
  double epis;      // help_eps
  int H;            // half-planes per point
  int trajid;       // segment index (moving obstacles only)
  double trajtime;  // trajtimes[trajid] (traj_optimizer.cpp:230-234,291)
  double t_piece;   // start time of the piece inside its segment: t += getDt() for the lp pieces before it
                    // (traj_optimizer.cpp:775), the sum formed in that order once per piece (moving obstacles only)
  double t_now;
};

// The body of the j-loop of addPVAGradCost2CT (traj_optimizer.cpp:499-706) for one
// constraint point.  Gradients are returned with respect to sigma, sigma',
// sigma'' (each penalty's dviol/dc is beta0 a^T + beta1 b^T + beta2 c^T, so
// out = {a (2), b (2), c (2), gdT, cost}); the caller chains them onto the piece
// coefficients.  `plane(k, n0, n1, q0, q1)` loads half-plane k of this point.
// HU > 0: the half-plane loop is unrolled HU times with a guard on in.H (keeps a register-resident
// plane array statically indexed); HU == 0: plain runtime loop.
template <bool SUR, int HU, class PlaneLoader>
DFTPAV_HD inline void sample_point_math(const DevParams &P, const DevSurround &S, const SampleIn &in, PlaneLoader plane,
                                        double out[8]) {
  for (int k = 0; k < 8; k++) out[k] = 0.0;
  const int j = in.j, K = in.K, lp = in.lp, N = in.N;
  const double dt = in.dt;
  const double Kd = (double)K, rK = 1.0 / Kd; // one division for dt / K, 1 / K and every pena / K below
  const double step = div_rcp(dt, Kd, rK);
  const double s1 = in.s1;
  double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
  double beta0[6] = {1.0, s1, s2, s3, s4, s5};
  double beta1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
  double beta2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
  double alpha = rK * j;
  double cc[12];
  load_piece_coeffs(in.cc, cc);
  double sigma[2] = {0, 0}, dsigma[2] = {0, 0}, ddsigma[2] = {0, 0};
  for (int k = 0; k < 6; k++) {
    double c0 = cc[2 * k], c1 = cc[2 * k + 1];
    sigma[0] = fma_(c0, beta0[k], sigma[0]);
    sigma[1] = fma_(c1, beta0[k], sigma[1]);
    dsigma[0] = fma_(c0, beta1[k], dsigma[0]);
    dsigma[1] = fma_(c1, beta1[k], dsigma[1]);
    ddsigma[0] = fma_(c0, beta2[k], ddsigma[0]);
    ddsigma[1] = fma_(c1, beta2[k], ddsigma[1]);
  }
  double omg = (j == 0 || j == K) ? 0.5 : 1.0;
  double z_h0 = sqrt(fma_(dsigma[0], dsigma[0], dsigma[1] * dsigma[1]));
  double z_h1 = fma_(ddsigma[0], dsigma[0], ddsigma[1] * dsigma[1]);
  double z_h3 = fma_(ddsigma[1], dsigma[0], (-ddsigma[0]) * dsigma[1]);
  if (z_h0 < 1e-4 || (j == 0 && lp == 0) || (lp == N - 1 && j == K)) return; // traj_optimizer.cpp:550-553
  // sigma''' (traj_optimizer.cpp:508,519) only enters the gradients of the acceleration and curvature
  // penalties; its value does not depend on when it is evaluated, so it is formed only where one is active
  const double beta3[6] = {0.0, 0.0, 0.0, 6.0, 24.0 * s1, 60.0 * s2};

  const int singul_ = in.singul;
  // limits switch on the gear (traj_optimizer.cpp:448-457); selects keep the constants in scalar registers
  const bool fwd = singul_ > 0;
  double max_vel = fwd ? P.max_vel[0] : P.max_vel[1];
  double max_acc = fwd ? P.max_acc[0] : P.max_acc[1];
  double max_cur = fwd ? P.max_cur[0] : P.max_cur[1];
  double vel2_reci = 1.0 / (z_h0 * z_h0);
  // z^2 + 0.0 == z^2 exactly, so with help_eps == 0 (the live value, traj_manager.cpp:610) the second
  // reciprocal is the first one
  double vel2_reci_e = in.epis == 0.0 ? vel2_reci : 1.0 / fma_(z_h0, z_h0, in.epis);
  double vel3_2_reci_e = vel2_reci_e * sqrt(vel2_reci_e);
  z_h0 = 1.0 / z_h0;
  double z_h4 = z_h1 * vel2_reci;
  double violaVel = fma_(-max_vel, max_vel, 1.0 / vel2_reci);
  double acc2 = z_h1 * z_h1 * vel2_reci;
  double cur = z_h3 * vel3_2_reci_e;
  double violaAcc = fma_(-max_acc, max_acc, acc2);
  double violaCurL = cur - max_cur;
  double violaCurR = -cur - max_cur;
  double ego_R[4] = {singul_ * dsigma[0] * z_h0, singul_ * -dsigma[1] * z_h0, singul_ * dsigma[1] * z_h0,
                     singul_ * dsigma[0] * z_h0};
  double R_dot[4];
  {
    double ta[4] = {ddsigma[0], -ddsigma[1], ddsigma[1], ddsigma[0]};
    double tv[4] = {dsigma[0], -dsigma[1], dsigma[1], dsigma[0]};
    for (int k = 0; k < 4; k++) R_dot[k] = singul_ * fma_(ta[k], z_h0, -(tv[k] * vel2_reci * z_h0 * z_h1));
  }
  double A[2] = {0, 0}, Bv[2] = {0, 0}, Cv[2] = {0, 0}, gdT = 0.0, cost = 0.0;

  // ---- safe corridor, traj_optimizer.cpp:592-622 (5 footprint entries, vertex 0 repeated)
  // The body point of a vertex does not depend on the half-plane, and entry 4 of vec_le_ is entry 0
  // again (traj_optimizer.cpp:1772-1773): the four distinct body points are formed once and the
  // contribution of vertex 0 is accumulated a second time, in the order of the reference.
  double Rle[4][2], bpt[4][2];
  for (int v = 0; v < 4; v++) {
    const double le0 = P.vec_le[v][0], le1 = P.vec_le[v][1];
    Rle[v][0] = fma_(ego_R[0], le0, ego_R[1] * le1);
    Rle[v][1] = fma_(ego_R[2], le0, ego_R[3] * le1);
    bpt[v][0] = sigma[0] + Rle[v][0];
    bpt[v][1] = sigma[1] + Rle[v][1];
  }
  if (HU > 0) {
    // The 5 H tests first, collected in a bit mask (bit 5 k + vv, the order of the reference's nested loops),
    // then one pass over the set bits: on the GPU a lane only spends time on its own violated half-planes --
    // as nested loops every one of the 20 bodies ran for the whole wave if a single lane needed it.
    double pn0[HU > 0 ? HU : 1], pn1[HU > 0 ? HU : 1], pq0[HU > 0 ? HU : 1], pq1[HU > 0 ? HU : 1];
    unsigned active = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < HU; k++) {
      pn0[k] = pn1[k] = pq0[k] = pq1[k] = 0.0;
      if (k < in.H) {
        plane(k, pn0[k], pn1[k], pq0[k], pq1[k]);
        for (int vv = 0; vv < 5; vv++) {
          const int v = vv == 4 ? 0 : vv;
          const double violaPos = fma_(pn0[k], bpt[v][0] - pq0[k], pn1[k] * (bpt[v][1] - pq1[k]));
          if (violaPos > 0) active |= 1u << (5 * k + vv);
        }
      }
    }
    while (active) {
      const int bit = __builtin_ctz(active);
      active &= active - 1;
      const int k = (bit * 13) >> 6; // bit / 5 for bit < 32
      const int vv = bit - 5 * k;
      const int v = vv == 4 ? 0 : vv;
      double on0 = pn0[0], on1 = pn1[0], q0 = pq0[0], q1 = pq1[0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int t = 1; t < HU; t++) {
        on0 = k == t ? pn0[t] : on0;
        on1 = k == t ? pn1[t] : on1;
        q0 = k == t ? pq0[t] : q0;
        q1 = k == t ? pq1[t] : q1;
      }
      double bp0 = bpt[0][0], bp1 = bpt[0][1], rl0 = Rle[0][0], rl1 = Rle[0][1], le0 = P.vec_le[0][0], le1 = P.vec_le[0][1];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int t = 1; t < 4; t++) {
        bp0 = v == t ? bpt[t][0] : bp0;
        bp1 = v == t ? bpt[t][1] : bp1;
        rl0 = v == t ? Rle[t][0] : rl0;
        rl1 = v == t ? Rle[t][1] : rl1;
        le0 = v == t ? P.vec_le[t][0] : le0;
        le1 = v == t ? P.vec_le[t][1] : le1;
      }
      const double violaPos = fma_(on0, bp0 - q0, on1 * (bp1 - q1)); // the same expression as in the test: > 0 here
      double pena, penaD;
      smoothed_l1(violaPos, pena, penaD);
      double tl[4] = {le0, -le1, le1, le0};
      double Mm[4];
      Mm[0] = fma_(singul_ * tl[0], z_h0, -(rl0 * dsigma[0] * vel2_reci));
      Mm[1] = fma_(singul_ * tl[1], z_h0, -(rl0 * dsigma[1] * vel2_reci));
      Mm[2] = fma_(singul_ * tl[2], z_h0, -(rl1 * dsigma[0] * vel2_reci));
      Mm[3] = fma_(singul_ * tl[3], z_h0, -(rl1 * dsigma[1] * vel2_reci));
      double w[2] = {dsigma[0] + fma_(R_dot[0], le0, R_dot[1] * le1), dsigma[1] + fma_(R_dot[2], le0, R_dot[3] * le1)};
      double gradViolaPt = fma_(alpha * on0, w[0], (alpha * on1) * w[1]);
      double sc = omg * step * P.wei_obs * penaD;
      A[0] = fma_(sc, on0, A[0]);
      A[1] = fma_(sc, on1, A[1]);
      Bv[0] = fma_(sc, fma_(on0, Mm[0], on1 * Mm[2]), Bv[0]);
      Bv[1] = fma_(sc, fma_(on0, Mm[1], on1 * Mm[3]), Bv[1]);
      gdT = fma_(omg * P.wei_obs, fma_(penaD * gradViolaPt, step, div_rcp(pena, Kd, rK)), gdT);
      cost = fma_(omg * step * P.wei_obs, pena, cost);
    }
  } else {
  for (int k = 0; k < in.H; k++) {
    double on0, on1, q0, q1;
    plane(k, on0, on1, q0, q1);
    for (int vv = 0; vv < 5; vv++) {
      const int v = vv == 4 ? 0 : vv;
      double violaPos = fma_(on0, bpt[v][0] - q0, on1 * (bpt[v][1] - q1));
      if (violaPos > 0) {
        const double le0 = P.vec_le[v][0], le1 = P.vec_le[v][1];
        double pena, penaD;
        smoothed_l1(violaPos, pena, penaD);
        double tl[4] = {le0, -le1, le1, le0};
        double Mm[4];
        Mm[0] = fma_(singul_ * tl[0], z_h0, -(Rle[v][0] * dsigma[0] * vel2_reci));
        Mm[1] = fma_(singul_ * tl[1], z_h0, -(Rle[v][0] * dsigma[1] * vel2_reci));
        Mm[2] = fma_(singul_ * tl[2], z_h0, -(Rle[v][1] * dsigma[0] * vel2_reci));
        Mm[3] = fma_(singul_ * tl[3], z_h0, -(Rle[v][1] * dsigma[1] * vel2_reci));
        double w[2] = {dsigma[0] + fma_(R_dot[0], le0, R_dot[1] * le1), dsigma[1] + fma_(R_dot[2], le0, R_dot[3] * le1)};
        double gradViolaPt = fma_(alpha * on0, w[0], (alpha * on1) * w[1]);
        double sc = omg * step * P.wei_obs * penaD;
        A[0] = fma_(sc, on0, A[0]);
        A[1] = fma_(sc, on1, A[1]);
        Bv[0] = fma_(sc, fma_(on0, Mm[0], on1 * Mm[2]), Bv[0]);
        Bv[1] = fma_(sc, fma_(on0, Mm[1], on1 * Mm[3]), Bv[1]);
        gdT = fma_(omg * P.wei_obs, fma_(penaD * gradViolaPt, step, div_rcp(pena, Kd, rK)), gdT);
        cost = fma_(omg * step * P.wei_obs, pena, cost);
        }
      }
    }
  }

  // ---- velocity, traj_optimizer.cpp:642-653
  if (violaVel > 0.0) {
    double pena, penaD;
    smoothed_l1(violaVel, pena, penaD);
    double gradViolaVt = 2.0 * alpha * z_h1;
    double sc = omg * step * P.wei_feas * penaD;
    Bv[0] = fma_(sc, 2.0 * dsigma[0], Bv[0]);
    Bv[1] = fma_(sc, 2.0 * dsigma[1], Bv[1]);
    gdT = fma_(omg * P.wei_feas, fma_(penaD * gradViolaVt, step, div_rcp(pena, Kd, rK)), gdT);
    cost = fma_(omg * step * P.wei_feas, pena, cost);
  }
  // ---- longitudinal acceleration, traj_optimizer.cpp:655-665
  if (violaAcc > 0.0) {
    double pena, penaD;
    smoothed_l1(violaAcc, pena, penaD);
    double u0 = fma_(z_h4, ddsigma[0], -(z_h4 * z_h4 * dsigma[0]));
    double u1 = fma_(z_h4, ddsigma[1], -(z_h4 * z_h4 * dsigma[1]));
    double ddd0 = 0.0, ddd1 = 0.0;
    for (int k = 0; k < 6; k++) {
      ddd0 = fma_(in.cc[2 * k], beta3[k], ddd0);
      ddd1 = fma_(in.cc[2 * k + 1], beta3[k], ddd1);
    }
    double z_h2 = fma_(ddd0, dsigma[0], ddd1 * dsigma[1]);
    double sqn = fma_(ddsigma[0], ddsigma[0], ddsigma[1] * ddsigma[1]);
    double gradViolaAt = 2.0 * alpha * fma_(z_h4, sqn + z_h2, -(z_h4 * z_h4 * z_h1));
    double sc = omg * step * P.wei_feas * penaD;
    Bv[0] = fma_(sc, 2.0 * u0, Bv[0]);
    Bv[1] = fma_(sc, 2.0 * u1, Bv[1]);
    Cv[0] = fma_(sc, 2.0 * z_h4 * dsigma[0], Cv[0]);
    Cv[1] = fma_(sc, 2.0 * z_h4 * dsigma[1], Cv[1]);
    gdT = fma_(omg * P.wei_feas, fma_(penaD * gradViolaAt, step, div_rcp(pena, Kd, rK)), gdT);
    cost = fma_(omg * step * P.wei_feas, pena, cost);
  }
  // ---- curvature, two one-sided penalties weighted x10, traj_optimizer.cpp:684-705
  if (violaCurL > 0.0 || violaCurR > 0.0) {
    double ku0 = fma_(vel3_2_reci_e, ddsigma[1], -(3 * vel3_2_reci_e * vel2_reci_e * z_h3 * dsigma[0]));
    double ku1 = fma_(vel3_2_reci_e, -ddsigma[0], -(3 * vel3_2_reci_e * vel2_reci_e * z_h3 * dsigma[1]));
    double kw0 = -(vel3_2_reci_e * dsigma[1]);
    double kw1 = vel3_2_reci_e * dsigma[0];
    double ddd0 = 0.0, ddd1 = 0.0;
    for (int k = 0; k < 6; k++) {
      ddd0 = fma_(in.cc[2 * k], beta3[k], ddd0);
      ddd1 = fma_(in.cc[2 * k + 1], beta3[k], ddd1);
    }
    double z1 = fma_(ddd1, dsigma[0], (-ddd0) * dsigma[1]);
    double kt = alpha * vel3_2_reci_e * fma_(-(3 * vel2_reci_e * z_h3), z_h1, z1);
    if (violaCurL > 0.0) {
      double pena, penaD;
      smoothed_l1(violaCurL, pena, penaD);
      double sc = omg * step * P.wei_feas * 10.0 * penaD;
      Bv[0] = fma_(sc, ku0, Bv[0]);
      Bv[1] = fma_(sc, ku1, Bv[1]);
      Cv[0] = fma_(sc, kw0, Cv[0]);
      Cv[1] = fma_(sc, kw1, Cv[1]);
      gdT = fma_(omg * P.wei_feas * 10.0, fma_(penaD * kt, step, div_rcp(pena, Kd, rK)), gdT);
      cost = fma_(omg * step * P.wei_feas * 10.0, pena, cost);
    }
    if (violaCurR > 0.0) {
      double pena, penaD;
      smoothed_l1(violaCurR, pena, penaD);
      double sc = omg * step * P.wei_feas * 10.0 * penaD;
      Bv[0] = fma_(sc, -ku0, Bv[0]);
      Bv[1] = fma_(sc, -ku1, Bv[1]);
      Cv[0] = fma_(sc, -kw0, Cv[0]);
      Cv[1] = fma_(sc, -kw1, Cv[1]);
      gdT = fma_(omg * P.wei_feas * 10.0, fma_(penaD * (-kt), step, div_rcp(pena, Kd, rK)), gdT);
      cost = fma_(omg * step * P.wei_feas * 10.0, pena, cost);
    }
  }
  out[0] = A[0]; out[1] = A[1];
  out[2] = Bv[0]; out[3] = Bv[1];
  out[4] = Cv[0]; out[5] = Cv[1];
  out[6] = gdT;
  out[7] = cost;
}

// ---- the moving-obstacle term (traj_optimizer.cpp:636-638 -> dynamicObsGradCostP) as a pass of its own
// The kernels evaluate it per (constraint point, obstacle) PAIR, only for the pairs that pass the distance gate of
// traj_optimizer.cpp:1393, one pair per lane (fused into sample_point_math it spilled ~2500 VGPRs and every lane of a wave
// waited for the few that had an obstacle near).  The point's state is formed again from the same expressions as in
// sample_point_math; a pair's result has the layout of a point's, {d/dsigma (2), d/dsigma' (2), 0, 0, gdT, cost}, and is
// added to the per-piece sums after the static part, pairs in (point, obstacle) order (solver.hip, E4).
// SampleIn::t_piece: t += getDt() for the lp pieces before piece lp of a segment (traj_optimizer.cpp:775)
DFTPAV_HD inline double piece_start_time(double dt, int lp) {
  double t = 0.0;
  for (int p = 0; p < lp; p++) t += dt;
  return t;
}
struct DynPoint {
  double sigma[2], dsigma[2], ddsigma[2], ego_R[4];
  double omg, step, alpha, t;
  bool skip;
};
DFTPAV_HD inline void dynamic_point_state(const SampleIn &in, DynPoint &q) {
  const int j = in.j, K = in.K, lp = in.lp, N = in.N;
  const double dt = in.dt;
  const double Kd = (double)K, rK = 1.0 / Kd;
  q.step = div_rcp(dt, Kd, rK);
  const double s1 = in.s1;
  double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
  double beta0[6] = {1.0, s1, s2, s3, s4, s5};
  double beta1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
  double beta2[6] = {0.0, 0.0, 2.0, 6.0 * s1, 12.0 * s2, 20.0 * s3};
  q.alpha = rK * j;
  double cc[12];
  load_piece_coeffs(in.cc, cc);
  double sigma[2] = {0, 0}, dsigma[2] = {0, 0}, ddsigma[2] = {0, 0};
  for (int k = 0; k < 6; k++) {
    double c0 = cc[2 * k], c1 = cc[2 * k + 1];
    sigma[0] = fma_(c0, beta0[k], sigma[0]);
    sigma[1] = fma_(c1, beta0[k], sigma[1]);
    dsigma[0] = fma_(c0, beta1[k], dsigma[0]);
    dsigma[1] = fma_(c1, beta1[k], dsigma[1]);
    ddsigma[0] = fma_(c0, beta2[k], ddsigma[0]);
    ddsigma[1] = fma_(c1, beta2[k], ddsigma[1]);
  }
  q.omg = (j == 0 || j == K) ? 0.5 : 1.0;
  double z_h0 = sqrt(fma_(dsigma[0], dsigma[0], dsigma[1] * dsigma[1]));
  q.skip = z_h0 < 1e-4 || (j == 0 && lp == 0) || (lp == N - 1 && j == K); // traj_optimizer.cpp:550-553
  z_h0 = 1.0 / z_h0;
  const int singul_ = in.singul;
  q.ego_R[0] = singul_ * dsigma[0] * z_h0;
  q.ego_R[1] = singul_ * -dsigma[1] * z_h0;
  q.ego_R[2] = singul_ * dsigma[1] * z_h0;
  q.ego_R[3] = singul_ * dsigma[0] * z_h0;
  for (int d = 0; d < 2; d++) {
    q.sigma[d] = sigma[d];
    q.dsigma[d] = dsigma[d];
    q.ddsigma[d] = ddsigma[d];
  }
  q.t = in.t_piece + q.step * j;
}
// bit u set: obstacle u passes the distance gate at this point (no bits for a point the sample loop skips).
// Most (point, obstacle) combinations are rejected by the box of the piece the obstacle is on (dyn_obstacle_near).  That
// decision is taken here first for four obstacles at a time in straight-line code -- the guessed piece index is checked
// against its two thresholds, the box of that piece against the point -- so that the table reads of the four are in
// flight together instead of one dependent round trip after the other; only what is not rejected this way (guess off,
// obstacle past its end, point near the box) goes through dyn_obstacle_near, which takes the same decisions again.
template <class SV>
DFTPAV_HD inline unsigned dynamic_gate_mask(const DevParams &P, const SV &S, const SampleIn &in) {
  DynPoint q;
  dynamic_point_state(in, q);
  if (q.skip) return 0u;
  const int ns = S.S;
  unsigned need = ns >= 32 ? 0xffffffffu : (1u << ns) - 1u;
  if (S.has_theta() && S.has_bbox()) {
    const double r = P.veh_length_infl * 1.5 + 1e-6;
    for (int u0 = 0; u0 < ns; u0 += 4) {
      // three rounds of table reads for the four obstacles, each round's reads independent of one another; no branches
      // (bitwise combinations of the comparisons, every read from an always-valid address)
      int pz[4], gz[4];
      double ptz[4], totz[4];
      for (int i = 0; i < 4; i++) {
        const int u = u0 + i < ns ? u0 + i : ns - 1;
        ptz[i] = (in.t_now - S.start[u] + in.trajtime) + q.t; // as in dyn_obstacle_near
        totz[i] = S.total[u];
        const int p0 = S.piece_off[u], np = S.piece_off[u + 1] - p0;
        int g = (int)(ptz[i] * S.rate(u));
        g = g < 0 ? 0 : (g > np - 1 ? np - 1 : g);
        pz[i] = p0;
        gz[i] = g;
      }
      double taz[4], tbz[4], bbz[4][4];
      for (int i = 0; i < 4; i++) {
        taz[i] = S.theta[pz[i] + gz[i]];
        tbz[i] = S.theta[pz[i] + (gz[i] > 0 ? gz[i] - 1 : 0)];
        S.load_box(pz[i] + gz[i], bbz[i]);
      }
      for (int i = 0; i < 4; i++) {
        const bool exact = (!(ptz[i] > taz[i])) & ((gz[i] == 0) | (ptz[i] > tbz[i])); // sur_index returns g
        const bool far = (q.sigma[0] < bbz[i][0] - r) | (q.sigma[0] > bbz[i][1] + r) | (q.sigma[1] < bbz[i][2] - r) |
                         (q.sigma[1] > bbz[i][3] + r); // far_from_piece
        const bool rej = (u0 + i < ns) & (ptz[i] >= 0.0) & (ptz[i] < totz[i]) & exact & far; // a negative local time leaves the box
        need &= ~((rej ? 1u : 0u) << (u0 + i));
      }
    }
  }
  unsigned mask = 0u;
  for (int u = 0; u < ns; u++) {
    if (!((need >> u) & 1u)) continue;
    DynObs ob;
    if (dyn_obstacle_near(P, S, u, in.t_now, q.t, in.trajtime, q.sigma, ob)) mask |= 1u << u;
  }
  return mask;
}
// one (point, obstacle) pair that passed the gate
template <class SV>
DFTPAV_HD inline void dynamic_pair_math(const DevParams &P, const SV &S, const SampleIn &in, int u, double out[8]) {
  for (int k = 0; k < 8; k++) out[k] = 0.0;
  DynPoint q;
  dynamic_point_state(in, q);
  DynObs ob;
  if (q.skip || !dyn_obstacle_near(P, S, u, in.t_now, q.t, in.trajtime, q.sigma, ob)) return;
  double A[2] = {0, 0}, Bv[2] = {0, 0}, gdT = 0.0;
  const double cost = dynamic_pair(P, S, u, ob, q.omg, q.step, q.alpha, in.lp, in.K, q.sigma, q.dsigma, q.ddsigma, q.ego_R,
                                   in.singul, in.trajid, in.N, A, Bv, gdT);
  out[0] = A[0]; out[1] = A[1];
  out[2] = Bv[0]; out[3] = Bv[1];
  out[6] = gdT;
  out[7] = cost;
}

// -------------------------------------------- constant MINCO operator
// BandedSystem (poly_traj_utils.hpp:727-826) restated for the set-up step that
// builds the dense operator: A_N^{-1} applied to the N+5 unit vectors of the
// RHS rows that can be non-zero (poly_traj_utils.hpp:968-977).  Host only.
struct BandedLU {
  int N, lowerBw, upperBw;
  double *ptr;
  double &at(int i, int j) { return ptr[(i - j + upperBw) * N + j]; }
};
inline void banded_factorize(BandedLU &A) { // poly_traj_utils.hpp:776-800, no pivoting
  int N = A.N;
  for (int k = 0; k <= N - 2; k++) {
    int iM = k + A.lowerBw < N - 1 ? k + A.lowerBw : N - 1;
    double cVl = A.at(k, k);
    for (int i = k + 1; i <= iM; i++)
      if (A.at(i, k) != 0.0) A.at(i, k) /= cVl;
    int jM = k + A.upperBw < N - 1 ? k + A.upperBw : N - 1;
    for (int j = k + 1; j <= jM; j++) {
      cVl = A.at(k, j);
      if (cVl != 0.0)
        for (int i = k + 1; i <= iM; i++)
          if (A.at(i, k) != 0.0) A.at(i, j) -= A.at(i, k) * cVl;
    }
  }
}
inline void banded_solve1(BandedLU &A, double *b) { // poly_traj_utils.hpp:805-826, one column
  int N = A.N;
  for (int j = 0; j <= N - 1; j++) {
    int iM = j + A.lowerBw < N - 1 ? j + A.lowerBw : N - 1;
    for (int i = j + 1; i <= iM; i++)
      if (A.at(i, j) != 0.0) b[i] -= A.at(i, j) * b[j];
  }
  for (int j = N - 1; j >= 0; j--) {
    b[j] /= A.at(j, j);
    int iM = 0 > j - A.upperBw ? 0 : j - A.upperBw;
    for (int i = iM; i <= j - 1; i++)
      if (A.at(i, j) != 0.0) b[i] -= A.at(i, j) * b[j];
  }
}
// fills the band matrix of MinJerkOpt::reset (poly_traj_utils.hpp:895-947)
inline void minco_fill(BandedLU &A, int N) {
  A.at(0, 0) = 1.0;
  A.at(1, 1) = 1.0;
  A.at(2, 2) = 2.0;
  for (int i = 0; i < N - 1; i++) {
    int r = 6 * i;
    A.at(r + 3, r + 3) = 6.0; A.at(r + 3, r + 4) = 24.0; A.at(r + 3, r + 5) = 60.0; A.at(r + 3, r + 9) = -6.0;
    A.at(r + 4, r + 4) = 24.0; A.at(r + 4, r + 5) = 120.0; A.at(r + 4, r + 10) = -24.0;
    for (int k = 0; k < 6; k++) A.at(r + 5, r + k) = 1.0;
    for (int k = 0; k < 6; k++) A.at(r + 6, r + k) = 1.0;
    A.at(r + 6, r + 6) = -1.0;
    A.at(r + 7, r + 1) = 1.0; A.at(r + 7, r + 2) = 2.0; A.at(r + 7, r + 3) = 3.0; A.at(r + 7, r + 4) = 4.0;
    A.at(r + 7, r + 5) = 5.0; A.at(r + 7, r + 7) = -1.0;
    A.at(r + 8, r + 2) = 2.0; A.at(r + 8, r + 3) = 6.0; A.at(r + 8, r + 4) = 12.0; A.at(r + 8, r + 5) = 20.0;
    A.at(r + 8, r + 8) = -2.0;
  }
  int n6 = 6 * N;
  for (int k = 0; k < 6; k++) A.at(n6 - 3, n6 - 6 + k) = 1.0;
  for (int k = 1; k < 6; k++) A.at(n6 - 2, n6 - 6 + k) = (double)k;
  A.at(n6 - 1, n6 - 4) = 2.0; A.at(n6 - 1, n6 - 3) = 6.0; A.at(n6 - 1, n6 - 2) = 12.0; A.at(n6 - 1, n6 - 1) = 20.0;
}

} // namespace dftpav
