// cr_trig.h — correctly rounded sin and cos in fp64, for host and device.
//
// The reference-order kernel (solver_ref.hip) runs the reference's floating-point program.  With a gear shift that program
// calls libm's cos / sin of the junction angle in every evaluation (traj_optimizer.cpp:273-282, 311-318), and libm's
// results are a property of the host: glibc's sin / cos are not correctly rounded (on this image's glibc 2.35 they differ
// from the correctly rounded value for 1 argument in 1 000) and are IFUNC symbols whose variant -- SSE2 or AVX2+FMA, with
// different roundings -- is picked by the CPU.  The device therefore uses the one sin / cos that is DEFINED rather than
// implemented: the correctly rounded one.  Any correctly rounded libm (CORE-MATH, LLVM libc) gives these bits; glibc
// gives them for 999 arguments in 1 000.
//
// Method: x - k pi/2 with pi/2 in three 33-bit chunks and a tail (152 bits; k p_i exact for |k| < 2^20, used for |x| < 2^20;
// from there on the reduction of Payne and Hanek over 1312 bits of 2/pi, reduce_large below), carried in double-double;
// Taylor series of sin / cos on |r| <= pi/4 in double-double arithmetic (error-free sums and FMA products) with the coefficients 1 / n! as double-double constants, to ~2^-100; the high word of the normalised
// result is the correctly rounded value unless the true value lies within ~2^-100 relative of a rounding boundary
// (probability ~2^-47 per call; the known hardest cases of sin / cos in double need 2^-126).
// tests/test_cr_trig.py checks it against binary128 (libquadmath) on millions of arguments.
#pragma once
#include "device_types.h"

// (scripts/cr_quick_check.cpp counts how often the quick phases of exp / log / sincos hand over to the accurate ones)
#ifndef DFTPAV_CR_FALLBACK
#define DFTPAV_CR_FALLBACK(which) ((void)0)
#endif
// what the quick phases' rounding tests ask for, relative.  Their error bounds are 2^-67 (exp) and 2^-68 (log); built with a
// smaller number here, the check above finds no mismatch at 2^-67 (1.4e9 arguments), the first ones of exp at 2^-69 and of log at
// 2^-71 (2e8 arguments per range) -- the bounds are real, and 2^-64 is eight times the larger one (7e9 arguments: none).
#ifndef DFTPAV_CR_QUICK_REL
#define DFTPAV_CR_QUICK_REL 0x1.0p-64
#endif

namespace dftpav {
namespace crt {

struct dd {
  double hi, lo;
};
DFTPAV_HD inline dd two_sum(double a, double b) { // error-free a + b
  const double s = a + b;
  const double bb = s - a;
  return dd{s, (a - (s - bb)) + (b - bb)};
}
DFTPAV_HD inline dd fast_two_sum(double a, double b) { // |a| >= |b|
  const double s = a + b;
  return dd{s, b - (s - a)};
}
DFTPAV_HD inline dd two_prod(double a, double b) { // error-free a b
  const double p = a * b;
  return dd{p, __builtin_fma(a, b, -p)};
}
DFTPAV_HD inline dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = fast_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return fast_two_sum(s.hi, s.lo);
}
DFTPAV_HD inline dd dd_add_d(dd a, double b) {
  dd s = two_sum(a.hi, b);
  s.lo += a.lo;
  return fast_two_sum(s.hi, s.lo);
}
DFTPAV_HD inline dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return fast_two_sum(p.hi, p.lo);
}
DFTPAV_HD inline dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }

// 1 / n!, n = 2 .. 31 (index n - 2)
DFTPAV_HD inline dd inv_fact(int n) {
  const double t[30][2] = {
      {0x1.0000000000000p-1, 0x0.0p+0},           {0x1.5555555555555p-3, 0x1.5555555555555p-57},   {0x1.5555555555555p-5, 0x1.5555555555555p-59},
      {0x1.1111111111111p-7, 0x1.1111111111111p-63}, {0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65}, {0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73},
      {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76}, {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73}, {0x1.27e4fb7789f5cp-22, 0x1.cbbc05b4fa99ap-76},
      {0x1.ae64567f544e4p-26, -0x1.c062e06d1f209p-80}, {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83}, {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
      {0x1.93974a8c07c9dp-37, 0x1.05d6f8a2efd1fp-92}, {0x1.ae7f3e733b81fp-41, 0x1.1d8656b0ee8cbp-97}, {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101},
      {0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103}, {0x1.6827863b97d97p-53, 0x1.eec01221a8b0bp-107}, {0x1.2f49b46814157p-57, 0x1.2650f61dbdcb4p-112},
      {0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120}, {0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120}, {0x1.0ce396db7f853p-70, -0x1.aebcdbd20331cp-124},
      {0x1.761b41316381ap-75, -0x1.3423c7d91404fp-130}, {0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135}, {0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139},
      {0x1.88e85fc6a4e5ap-89, -0x1.71c37ebd16540p-143}, {0x1.d1ab1c2dccea3p-94, 0x1.054d0c78aea14p-149}, {0x1.0a18a2635085dp-98, 0x1.b9e2e28e1aa54p-153},
      {0x1.259f98b4358adp-103, 0x1.eaf8c39dd9bc5p-157}, {0x1.3932c5047d60ep-108, 0x1.832b7b530a627p-162}, {0x1.434d2e783f5bcp-113, 0x1.0b87b91be9affp-167}};
  return dd{t[n - 2][0], t[n - 2][1]};
}

// r = x - k pi/2 as a double-double, k = nearest integer to x 2/pi (|x| < 1.6e6); returns k
DFTPAV_HD inline int reduce(double x, dd &r) {
  const double two_over_pi = 0x1.45f306dc9c883p-1;
  const double p1 = 0x1.921fb54400000p+0, p2 = 0x1.0b4611a600000p-34, p3 = 0x1.3198a2e000000p-69, p4 = 0x1.b839a252049c1p-104;
  const double fk = x * two_over_pi;
  const int k = (int)(fk < 0.0 ? fk - 0.5 : fk + 0.5);
  const double kd = (double)k;
  dd a = two_sum(x, -(kd * p1)); // kd p1, kd p2, kd p3 are exact (33-bit chunks, |k| < 2^20)
  a = dd_add_d(a, -(kd * p2));
  a = dd_add_d(a, -(kd * p3));
  const dd t = two_prod(kd, p4);
  r = dd_add(a, dd_neg(t));
  return k;
}
// The same for every finite |x| >= 2^20 (Payne & Hanek): an L-BFGS line search may try a point 1e10 away, where the junction
// angle is 1e10 too (found by scripts/fuzz_reference_order.py: the literal program returns a finite cost there and backs off).
// |x| = m 2^E with m a 53-bit integer; of x 2/pi = m 2^E sum_j W[j] 2^(-32 (j+1)) the words j with E - 32 (j+1) >= 2 add
// multiples of 4 and are dropped, the next nine words (288 bits) are multiplied by m exactly in 32-bit limbs, the words after
// them add less than 2^-202.  The two bits above the binary point are k mod 4, the 320 bits below it the fraction f (made
// |f| <= 1/2 by rounding k), and r = f pi/2 in double-double keeps >= 107 significant bits after a cancellation of up to 62.
// (returned by value: on the device three registers, where a reference parameter of a function that is not inlined is a slot of
// the caller's stack -- scratch memory, and a kernel that needs none without it)
struct reduced {
  dd r;
  int k;
};
DFTPAV_HD __attribute__((noinline)) inline reduced reduce_large_v(double x) {
  dd r;
  const unsigned W[41] = { // 2/pi = 0.W[0] W[1] ... in base 2^32
      0xa2f9836eu, 0x4e441529u, 0xfc2757d1u, 0xf534ddc0u, 0xdb629599u, 0x3c439041u, 0xfe5163abu, 0xdebbc561u, 0xb7246e3au, 0x424dd2e0u, 0x06492eeau,
      0x09d1921cu, 0xfe1deb1cu, 0xb129a73eu, 0xe88235f5u, 0x2ebb4484u, 0xe99c7026u, 0xb45f7e41u, 0x3991d639u, 0x835339f4u, 0x9c845f8bu, 0xbdf9283bu,
      0x1ff897ffu, 0xde05980fu, 0xef2f118bu, 0x5a0a6d1fu, 0x6d367ecfu, 0x27cb09b7u, 0x4f463f66u, 0x9e5fea2du, 0x7527bac7u, 0xebe5f17bu, 0x3d0739f7u,
      0x8a5292eau, 0x6bfb5fb1u, 0x1f8d5d08u, 0x56033046u, 0xfc7b6babu, 0xf0cfbc20u, 0x9af4361du, 0xa9e39161u};
  union {
    double d;
    unsigned long long u;
  } v;
  v.d = x < 0.0 ? -x : x;
  const int E = (int)(v.u >> 52) - 1075;
  const unsigned long long m = (v.u & ((1ull << 52) - 1ull)) | (1ull << 52);
  const unsigned ml = (unsigned)m, mh = (unsigned)(m >> 32);
  const int j0 = E >= 2 ? (E - 2) >> 5 : 0;
  // P = m x (W[j0] .. W[j0+8]), little-endian limbs p[0..10]; p[0]'s unit is 2^(E - 32 (j0 + 9))
  unsigned p[11];
  for (int i = 0; i < 11; i++) p[i] = 0u;
  unsigned long long carry = 0ull;
  for (int i = 0; i < 9; i++) {
    const unsigned long long t = (unsigned long long)ml * W[j0 + 8 - i] + carry;
    p[i] = (unsigned)t;
    carry = t >> 32;
  }
  p[9] = (unsigned)carry;
  carry = 0ull;
  for (int i = 0; i < 9; i++) {
    const unsigned long long t = (unsigned long long)mh * W[j0 + 8 - i] + p[i + 1] + carry;
    p[i + 1] = (unsigned)t;
    carry = t >> 32;
  }
  p[10] = (unsigned)carry; // mh < 2^21: no further carry
  // the binary point is at bit 32 (j0 + 9) - E in [255, 320]: move it to bit 320, the bottom of p[10]
  const int sh = 320 - (32 * (j0 + 9) - E);
  const int a = sh >> 5, b = sh & 31;
  unsigned q[11];
  for (int i = 10; i >= 0; i--) { // static indices: the limbs stay in registers on the device
    const unsigned s0 = i >= 0 ? p[i] : 0u, s1 = i >= 1 ? p[i - 1] : 0u, s2 = i >= 2 ? p[i - 2] : 0u, s3 = i >= 3 ? p[i - 3] : 0u;
    const unsigned hi = a == 0 ? s0 : (a == 1 ? s1 : s2), lo = a == 0 ? s1 : (a == 1 ? s2 : s3);
    q[i] = b ? (hi << b) | (lo >> (32 - b)) : hi;
  }
  int k = (int)(q[10] & 3u);
  const bool up = (q[9] & 0x80000000u) != 0u; // f >= 1/2: k + 1 and f - 1 = -(2^320 - fraction)
  if (up) {
    k += 1;
    unsigned c = 1u;
    for (int i = 0; i < 10; i++) {
      const unsigned t = ~q[i] + c;
      c = (c && t == 0u) ? 1u : 0u;
      q[i] = t;
    }
  }
  // |f| as a double-double: the limbs from the top down (non-overlapping, each product exact; what a double-double cannot hold
  // is below 2^-106 of the leading limb)
  dd f{0.0, 0.0};
  double w = 0x1.0p-32;
  for (int i = 9; i >= 0; i--) {
    f = dd_add_d(f, (double)q[i] * w);
    w *= 0x1.0p-32;
  }
  const dd pio2{0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54};
  r = dd_mul(f, pio2);
  if (up) r = dd_neg(r);
  if (x < 0.0) { // x = -(k pi/2 + r)
    r = dd_neg(r);
    k = -k;
  }
  return reduced{r, k};
}
DFTPAV_HD inline int reduce_large(double x, dd &r) {
  const reduced t = reduce_large_v(x);
  r = t.r;
  return t.k;
}
// sin and cos of a double-double |r| <= ~pi/4 by their Taylor series in double-double (Horner in r^2)
DFTPAV_HD inline dd sin_dd(dd r) {
  const dd r2 = dd_mul(r, r);
  dd acc = inv_fact(31);
  for (int n = 29; n >= 3; n -= 2) acc = dd_add(inv_fact(n), dd_neg(dd_mul(acc, r2))); // 1/n! - r^2 (1/(n+2)! - ...)
  // sin r = r - r^3 (1/3! - r^2 (...)) = r (1 - r^2 acc)
  const dd t = dd_mul(dd_mul(acc, r2), r);
  return dd_add(r, dd_neg(t));
}
DFTPAV_HD inline dd cos_dd(dd r) {
  const dd r2 = dd_mul(r, r);
  dd acc = inv_fact(30);
  for (int n = 28; n >= 2; n -= 2) acc = dd_add(inv_fact(n), dd_neg(dd_mul(acc, r2))); // 1/n! - r^2 (1/(n+2)! - ...)
  // cos r = 1 - r^2 (1/2! - r^2 (...))
  return dd_add_d(dd_neg(dd_mul(acc, r2)), 1.0);
}

// Ziv's rounding test (used by every two-phase function below).  e (normalised) approximates a value v to within rel |e.hi|: when
// e.hi + (e.lo - d) and e.hi + (e.lo + d), d = rel |e.hi|, round to the same double, every number between them does (rounding is
// monotonic), v among them: that double IS v correctly rounded.  (The two inner sums are themselves rounded, by at most 2^-106 |e.hi|:
// the callers' rel leaves a factor of four for it.)
DFTPAV_HD inline bool round_if_certain(dd e, double rel, double &out) {
  const double d = rel * (e.hi < 0.0 ? -e.hi : e.hi);
  const double a = e.hi + (e.lo - d), b = e.hi + (e.lo + d);
  out = a;
  return a == b;
}

// correctly rounded sin x and cos x, in two phases (Ziv) since round 6 -- the junction angles of a gear-shift trajectory cost one call per
// evaluation, on one lane while its wave waits:
//   quick     w = r^2 in double-double; sin r = r (1 + w (-1/3! + w (1/5! + w (-1/7! + w Ts)))), cos r = 1 + w (-1/2! + w (1/4! + w (-1/6!
//             + w Tc))) with the three leading coefficients and the Horner steps in double-double and the tails Ts = 1/9! - w/11! + ...
//             - w^5/19!, Tc = 1/8! - w/10! + ... + w^6/20! in plain fp64 (|r| <= 0.786, w <= 0.617: w^4 Ts <= 2^-21 of sin r, w^4 Tc <= 2^-18
//             of cos r >= 0.707; a few ulp of them, the neglected w.lo, the series' remainders r^21/21!, r^22/22!: below 2^-68 relative);
//             the rounding test asks for 2^-64 of each of the two (built with 2^-72 the first wrong roundings appear in 5e8 arguments, with
//             2^-70 none: scripts/cr_quick_check.cpp).  One angle in 640 fails one of the two tests and takes
//   accurate  the Taylor series of both in double-double to ~2^-100 (all there was until round 6).
// Both return the correctly rounded values (tests/test_cr_trig.py: against binary128; scripts/cr_quick_check.cpp: against each other).
template <bool QUICK>
DFTPAV_HD inline void sincos_impl(double x, double &s, double &c) {
  if (x == 0.0) { // sin keeps the sign of zero
    s = x;
    c = 1.0;
    return;
  }
  if (!(x - x == 0.0)) { // infinity or NaN: NaN, as libm
    s = c = x - x;
    return;
  }
  dd r;
  const int k = (x < 0.0 ? -x : x) < 0x1.0p+20 ? reduce(x, r) : reduce_large(x, r);
  double sq = 0.0, cq = 0.0;
  bool quick = false;
  if (QUICK) {
    const dd w = dd_mul(r, r);
    const double wh = w.hi;
    double Ts = -0x1.2f49b46814157p-57;              // -1 / 19!
    Ts = __builtin_fma(Ts, wh, 0x1.952c77030ad4ap-49);  //  1 / 17!
    Ts = __builtin_fma(Ts, wh, -0x1.ae7f3e733b81fp-41); // -1 / 15!
    Ts = __builtin_fma(Ts, wh, 0x1.6124613a86d09p-33);  //  1 / 13!
    Ts = __builtin_fma(Ts, wh, -0x1.ae64567f544e4p-26); // -1 / 11!
    Ts = __builtin_fma(Ts, wh, 0x1.71de3a556c734p-19);  //  1 / 9!
    double Tq = 0x1.e542ba4020225p-62;               //  1 / 20!
    Tq = __builtin_fma(Tq, wh, -0x1.6827863b97d97p-53); // -1 / 18!
    Tq = __builtin_fma(Tq, wh, 0x1.ae7f3e733b81fp-45);  //  1 / 16!
    Tq = __builtin_fma(Tq, wh, -0x1.93974a8c07c9dp-37); // -1 / 14!
    Tq = __builtin_fma(Tq, wh, 0x1.1eed8eff8d898p-29);  //  1 / 12!
    Tq = __builtin_fma(Tq, wh, -0x1.27e4fb7789f5cp-22); // -1 / 10!
    Tq = __builtin_fma(Tq, wh, 0x1.a01a01a01a01ap-16);  //  1 / 8!
    dd a = dd_add_d(dd_neg(inv_fact(7)), wh * Ts);          // -1/7! + w Ts
    a = dd_add(inv_fact(5), dd_mul(a, w));                  //  1/5! + w (...)
    a = dd_add(dd_neg(inv_fact(3)), dd_mul(a, w));          // -1/3! + w (...)
    const dd sr = dd_add(r, dd_mul(dd_mul(a, w), r));       // r + r w (...)
    dd b = dd_add_d(dd_neg(inv_fact(6)), wh * Tq);          // -1/6! + w Tc
    b = dd_add(inv_fact(4), dd_mul(b, w));                  //  1/4! + w (...)
    b = dd_add(dd_neg(inv_fact(2)), dd_mul(b, w));          // -1/2! + w (...)
    const dd cr = dd_add_d(dd_mul(b, w), 1.0);              // 1 + w (...)
    quick = round_if_certain(sr, DFTPAV_CR_QUICK_REL, sq);
    quick = round_if_certain(cr, DFTPAV_CR_QUICK_REL, cq) && quick;
    if (!quick) DFTPAV_CR_FALLBACK(2);
  }
  if (!quick) {
    sq = sin_dd(r).hi;
    cq = cos_dd(r).hi;
  }
  switch (k & 3) {
    case 0: s = sq; c = cq; break;
    case 1: s = cq; c = -sq; break;
    case 2: s = -sq; c = -cq; break;
    default: s = -cq; c = sq; break;
  }
}
DFTPAV_HD inline void sincos(double x, double &s, double &c) { sincos_impl<true>(x, s, c); }

// ---------------------------------------------------------------- exp, log, x^3 (the moving-obstacle term of the reference:
// 40 exponentials and 9 logarithms per (constraint point, obstacle) pair, traj_optimizer.cpp:1686-1707; pow(|v|, 3) in
// Piece::getRdot, poly_traj_utils.hpp:109) -- correctly rounded for the same reason as sin / cos above.
DFTPAV_HD inline dd dd_mul_d(dd a, double b) {
  dd p = two_prod(a.hi, b);
  p.lo += a.lo * b;
  return fast_two_sum(p.hi, p.lo);
}
DFTPAV_HD inline dd dd_div(dd a, dd b) { // a / b to ~2^-104
  const double q0 = a.hi / b.hi;
  dd r = dd_add(a, dd_neg(dd_mul_d(b, q0)));
  const double q1 = r.hi / b.hi;
  r = dd_add(r, dd_neg(dd_mul_d(b, q1)));
  const double q2 = r.hi / b.hi;
  dd q = fast_two_sum(q0, q1);
  return dd_add_d(q, q2);
}
DFTPAV_HD inline double scale2(double v, int k) { // v 2^k, in two steps so that the multipliers stay normal
  const int k1 = k / 2, k2 = k - k1;
  union { double d; unsigned long long u; } a, b;
  a.u = (unsigned long long)(1023 + k1) << 52;
  b.u = (unsigned long long)(1023 + k2) << 52;
  return (v * a.d) * b.d;
}
// exp x: x = k ln2 + r (ln2 in three 33-bit chunks and a tail), then two phases (Ziv):
//   quick     s = r / 32 (|s| <= 0.0109): exp s = 1 + s + s^2 / 2 in double-double + s^3 (1/6 + s/24 + ... + s^5 / 8!) in plain fp64
//             (the tail is below 2^-22, its error -- a few ulp of it, the neglected s.lo, the series' remainder s^9 / 9! -- below
//             2^-72), five squarings (error x 32: 2^-67 relative); the rounding test asks for 2^-64.  It fails for about one
//             argument in 1 000, which then takes
//   accurate  exp(r / 16) by its Taylor series in double-double to ~2^-100, four squarings (all there was until round 5).
// Both return the correctly rounded value, so which phase answered cannot be seen in the result (tests/test_cr_trig.py: the two
// against each other and against binary128).
// Results below the normal range (x < -708.4) go through a second rounding in scale2 (the reference's sums absorb them:
// they are weights relative to a term that is exactly 1).
template <bool QUICK>
DFTPAV_HD inline double exp_cr_impl(double x) {
  if (x != x) return x; // NaN in, NaN out (before the conversion to int below, which is undefined for a NaN on the host)
  if (x < -745.2) return 0.0;
  if (x > 709.8) return 1.0e308 * 1.0e308;
  if (x == 0.0) return 1.0;
  const double inv_ln2 = 0x1.71547652b82fep+0;
  const double l1 = 0x1.62e42ff000000p-1, l2 = -0x1.718432a200000p-35, l3 = 0x1.3c76730000000p-69, l4 = 0x1.f97b57a079a19p-103;
  const double fk = x * inv_ln2;
  const int k = (int)(fk < 0.0 ? fk - 0.5 : fk + 0.5);
  const double kd = (double)k;
  dd r = two_sum(x, -(kd * l1)); // kd l1, kd l2, kd l3 are exact (|k| <= 1075)
  r = dd_add_d(r, -(kd * l2));
  r = dd_add_d(r, -(kd * l3));
  r = dd_add(r, dd_neg(two_prod(kd, l4)));
  if (QUICK) {
    const dd s{r.hi * 0.03125, r.lo * 0.03125}; // r / 32, exact
    const double sh = s.hi;
    double P = 0x1.a01a01a01a01ap-16;              // 1 / 8!
    P = __builtin_fma(P, sh, 0x1.a01a01a01a01ap-13); // 1 / 7!
    P = __builtin_fma(P, sh, 0x1.6c16c16c16c17p-10); // 1 / 6!
    P = __builtin_fma(P, sh, 0x1.1111111111111p-7);  // 1 / 5!
    P = __builtin_fma(P, sh, 0x1.5555555555555p-5);  // 1 / 4!
    P = __builtin_fma(P, sh, 0x1.5555555555555p-3);  // 1 / 3!
    const double T = sh * sh * sh * P;
    dd h = dd_mul(s, s);
    h.hi *= 0.5;
    h.lo *= 0.5;
    h = dd_add(s, h);
    h = dd_add_d(h, T);
    dd e = dd_add_d(h, 1.0);
    for (int q = 0; q < 5; q++) e = dd_mul(e, e);
    double out;
    if (round_if_certain(e, DFTPAV_CR_QUICK_REL, out)) return scale2(out, k);
    DFTPAV_CR_FALLBACK(0);
  }
  r.hi *= 0.0625; // r / 16, exact
  r.lo *= 0.0625;
  dd p = inv_fact(14);
  for (int n = 13; n >= 2; n--) p = dd_add(inv_fact(n), dd_mul(p, r));
  p = dd_add_d(dd_mul(p, r), 1.0); // 1 + r (1/2! + ...)  -> coefficient of r^1
  dd e = dd_add_d(dd_mul(p, r), 1.0);
  for (int q = 0; q < 4; q++) e = dd_mul(e, e);
  return scale2(e.hi, k);
}
DFTPAV_HD inline double exp_cr(double x) { return exp_cr_impl<true>(x); }
// log x: x = m 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh z, z = (m - 1) / (m + 1) in double-double (|z| <= 0.1716,
// w = z^2 <= 0.0295), again in two phases:
//   quick     2 z (1 + w (1/3 + w/5 + w^2 Q(w))) with Q = 1/7 + w/9 + ... + w^9 / 25 in plain fp64 and the rest in double-double:
//             w^3 Q is below 2^-18 of the result, its error (a few ulp of it, the remainder w^10 / 27) below 2^-68; e ln2 is at
//             least twice log m when e != 0, so nothing cancels; the rounding test asks for 2^-64;
//   accurate  the whole series in double-double (all there was until round 5).
template <bool QUICK>
DFTPAV_HD inline double log_cr_impl(double x) {
  const double t[21][2] = {
      {0x1.5555555555555p-2, 0x1.5555555555555p-56},  {0x1.999999999999ap-3, -0x1.999999999999ap-57}, {0x1.2492492492492p-3, 0x1.2492492492492p-57},
      {0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58},  {0x1.745d1745d1746p-4, -0x1.745d1745d1746p-59}, {0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58},
      {0x1.1111111111111p-4, 0x1.1111111111111p-60},  {0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61},  {0x1.af286bca1af28p-5, 0x1.af286bca1af28p-59},
      {0x1.8618618618618p-5, 0x1.8618618618618p-59},  {0x1.642c8590b2164p-5, 0x1.642c8590b2164p-60},  {0x1.47ae147ae147bp-5, -0x1.eb851eb851eb8p-61},
      {0x1.2f684bda12f68p-5, 0x1.2f684bda12f68p-59},  {0x1.1a7b9611a7b96p-5, 0x1.1a7b9611a7b96p-61},  {0x1.0842108421084p-5, 0x1.0842108421084p-60},
      {0x1.f07c1f07c1f08p-6, -0x1.f07c1f07c1f08p-61}, {0x1.d41d41d41d41dp-6, 0x1.0750750750750p-60},  {0x1.bacf914c1bad0p-6, -0x1.bacf914c1bad0p-60},
      {0x1.a41a41a41a41ap-6, 0x1.0690690690690p-60},  {0x1.8f9c18f9c18fap-6, -0x1.f3831f3831f38p-61}, {0x1.7d05f417d05f4p-6, 0x1.7d05f417d05f4p-62}};
  const double l1 = 0x1.62e42ff000000p-1, l2 = -0x1.718432a200000p-35, l3 = 0x1.3c76730000000p-69, l4 = 0x1.f97b57a079a19p-103;
  if (x == 1.0) return 0.0;
  if (!(x > 0.0)) return x == 0.0 ? -1.0 / 0.0 : (x - x) / 0.0;   // log 0 = -inf; negative or NaN: NaN (an overflowed penalty of a far trial point)
  if (x > 0x1.fffffffffffffp+1023) return x;                        // +inf
  union { double d; unsigned long long u; } v;
  int e = -1023;
  if (x < 0x1.0p-1022) { // subnormal: exact scaling into the normal range
    x *= 0x1.0p+54;
    e -= 54;
  }
  v.d = x;
  e += (int)((v.u >> 52) & 0x7ff);
  v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; // mantissa in [1, 2)
  double m = v.d;
  if (m > 0x1.6a09e667f3bcdp+0) { // sqrt(2)
    m *= 0.5;
    e += 1;
  }
  const dd num = dd{m - 1.0, 0.0};        // exact (m in [0.70, 1.42))
  const dd den = two_sum(m, 1.0);         // exact as a double-double
  const dd z = dd_div(num, den);
  const dd w = dd_mul(z, z);
  const double ed = (double)e;
  dd el = two_sum(ed * l1, ed * l2); // e ln2: exact products (|e| <= 1074)
  el = dd_add_d(el, ed * l3);
  el = dd_add(el, two_prod(ed, l4));
  if (QUICK) {
    const double wh = w.hi;
    double Q = t[11][0];                          // 1 / 25
    for (int n = 10; n >= 2; n--) Q = __builtin_fma(Q, wh, t[n][0]); // ... 1/9, 1/7
    const double U = wh * wh * Q;
    dd in = dd_mul(w, dd{t[1][0], t[1][1]});      // w / 5
    in = dd_add(in, dd{t[0][0], t[0][1]});        // + 1/3
    in = dd_add_d(in, U);
    dd sq = dd_add(z, dd_mul(dd_mul(in, w), z));  // z + z w (1/3 + w/5 + w^2 Q)
    sq.hi *= 2.0;
    sq.lo *= 2.0;
    double out;
    if (round_if_certain(dd_add(el, sq), DFTPAV_CR_QUICK_REL, out)) return out;
    DFTPAV_CR_FALLBACK(1);
  }
  dd acc = dd{t[20][0], t[20][1]};
  for (int n = 19; n >= 0; n--) acc = dd_add(dd{t[n][0], t[n][1]}, dd_mul(acc, w)); // 1/3 + w (1/5 + w (...))
  dd sres = dd_add(z, dd_mul(dd_mul(acc, w), z)); // z + z^3 (1/3 + ...)
  sres.hi *= 2.0;
  sres.lo *= 2.0;
  return dd_add(el, sres).hi;
}
DFTPAV_HD inline double log_cr(double x) { return log_cr_impl<true>(x); }
// x^3, correctly rounded (the reference: pow(x, 3))
DFTPAV_HD inline double cube_cr(double x) {
  const double plain = x * x * x;
  if (!(plain - plain == 0.0) || plain == 0.0) return plain; // overflow, NaN, a zero or an underflow to zero: the error-free products below would give NaN
  const dd p = two_prod(x, x);
  return dd_mul_d(p, x).hi;
}


// ---------------------------------------------------------------- atan, atan2, tan (the steps either side of the solve: the heading
// of a sampled state is atan2 of its velocity -- traj_server_ros.cpp:385-397, poly_traj_utils.hpp:378-406 --, the steering angle an
// atan, the front end's curvature a tan) -- correctly rounded for the same reason as sin / cos above: the bits of libm's belong to
// the host.
// atan of a double-double 0 <= t <= 1: t = c + d with c = k / 64 the nearest sixty-fourth, atan t = atan c + atan u,
// u = (t - c) / (1 + t c), |u| <= 1 / 128, atan u by its Taylor series in double-double (u^19 / 19 < 2^-137 u); atan(k / 64) from a
// table of binary128 values split in two doubles.  ~2^-102 relative.
DFTPAV_HD inline dd atan_tab(int k) {
  const double t[65][2] = {
      {0x0p+0, 0x0p+0},       {0x1.fff555bbb729bp-7, -0x1.220c39d4dff5p-61},       {0x1.ffd55bba97625p-6, -0x1.5ec431444912cp-60},
      {0x1.7fb818430da2ap-5, -0x1.86ef8f794f105p-63},       {0x1.ff55bb72cfdeap-5, -0x1.c934d86d23f1dp-60},       {0x1.3f59f0e7c559dp-4, 0x1.ac4ce285df847p-58},
      {0x1.7ee182602f10fp-4, -0x1.cfb654c0c3d98p-58},       {0x1.be39ebe6f07c3p-4, 0x1.f7b8f29a05987p-58},       {0x1.fd5ba9aac2f6ep-4, -0x1.cd37686760c17p-59},
      {0x1.1e1fafb043727p-3, -0x1.b485914dacf8cp-59},       {0x1.3d6eee8c6626cp-3, 0x1.61a3b0ce9281bp-57},       {0x1.5c9811e3ec26ap-3, -0x1.054ab2c010f3dp-58},
      {0x1.7b97b4bce5b02p-3, 0x1.347b0b4f881cap-58},       {0x1.9a6a8e96c8626p-3, 0x1.cf601e7b4348ep-59},       {0x1.b90d7529260a2p-3, 0x1.17b10d2e0e5aap-61},
      {0x1.d77d5df205736p-3, 0x1.c648d1534597ep-57},       {0x1.f5b75f92c80ddp-3, 0x1.8ab6e3cf7afbdp-57},       {0x1.09dc597d86362p-2, 0x1.62e47390cb865p-56},
      {0x1.18bf5a30bf178p-2, 0x1.30ca4748b1bf8p-57},       {0x1.278372057ef46p-2, -0x1.077cdd36dfc81p-56},       {0x1.362773707ebccp-2, -0x1.963a544b672d8p-57},
      {0x1.44aa436c2af0ap-2, -0x1.5d5e43c55b3bap-56},       {0x1.530ad9951cd4ap-2, -0x1.2566480884082p-57},       {0x1.614840309cfe2p-2, -0x1.a725715711fp-56},
      {0x1.6f61941e4def1p-2, -0x1.c63aae6f6e918p-56},       {0x1.7d5604b63b3f7p-2, 0x1.69c885c2b249ap-56},       {0x1.8b24d394a1b25p-2, 0x1.b6d0ba3748fa8p-56},
      {0x1.98cd5454d6b18p-2, 0x1.9e6c988fd0a77p-56},       {0x1.a64eec3cc23fdp-2, -0x1.24dec1b50b7ffp-56},       {0x1.b3a911da65c6cp-2, 0x1.ae187b1ca504p-56},
      {0x1.c0db4c94ec9fp-2, -0x1.cc1ce70934c34p-56},       {0x1.cde53432c1351p-2, -0x1.a2cfa4418f1adp-56},       {0x1.dac670561bb4fp-2, 0x1.a2b7f222f65e2p-56},
      {0x1.e77eb7f175a34p-2, 0x1.0e53dc1bf3435p-56},       {0x1.f40dd0b541418p-2, -0x1.a3992dc382a23p-57},       {0x1.0039c73c1a40cp-1, -0x1.b32c949c9d593p-55},
      {0x1.0657e94db30dp-1, -0x1.d5b495f6349e6p-56},       {0x1.0c6145b5b43dap-1, 0x1.974fa13b5404fp-58},       {0x1.1255d9bfbd2a9p-1, -0x1.2bdaee1c0ee35p-58},
      {0x1.1835a88be7c13p-1, 0x1.c621cec00c301p-55},       {0x1.1e00babdefeb4p-1, -0x1.928df287a668fp-58},       {0x1.23b71e2cc9e6ap-1, 0x1.c421c9f38224ep-57},
      {0x1.2958e59308e31p-1, -0x1.09e73b0c6c087p-56},       {0x1.2ee628406cbcap-1, 0x1.c5d5e9ff0cf8dp-55},       {0x1.345f01cce37bbp-1, 0x1.1021137c71102p-55},
      {0x1.39c391cd4171ap-1, -0x1.2304331d8bf46p-55},       {0x1.3f13fb89e96f4p-1, 0x1.ecf8b492644fp-56},       {0x1.445065b795b56p-1, -0x1.f76d0163f79c8p-56},
      {0x1.4978fa3269ee1p-1, 0x1.2419a87f2a458p-56},       {0x1.4e8de5bb6ec04p-1, 0x1.4a33dbeb3796cp-55},       {0x1.538f57b89061fp-1, -0x1.1bb74abda520cp-55},
      {0x1.587d81f732fbbp-1, -0x1.5e5c9d8c5a95p-56},       {0x1.5d58987169b18p-1, 0x1.0028e4bc5e7cap-57},       {0x1.6220d115d7b8ep-1, -0x1.2b785350ee8c1p-57},
      {0x1.66d663923e087p-1, -0x1.6ea6febe8bbbap-56},       {0x1.6b798920b3d99p-1, -0x1.a80386188c50ep-55},       {0x1.700a7c5784634p-1, -0x1.8c34d25aadef6p-56},
      {0x1.748978fba8e0fp-1, 0x1.7b2a6165884a2p-59},       {0x1.78f6bbd5d315ep-1, 0x1.406a08980374p-55},       {0x1.7d528289fa093p-1, 0x1.560821e2f3aa9p-55},
      {0x1.819d0b7158a4dp-1, -0x1.bf76229d3b917p-56},       {0x1.85d69576cc2c5p-1, 0x1.6b66e7fc8b8c4p-57},       {0x1.89ff5ff57f1f8p-1, -0x1.55b9a5e177a1bp-55},
      {0x1.8e17aa99cc05ep-1, -0x1.ec182ab042f61p-56},       {0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55},
  };
  return dd{t[k][0], t[k][1]};
}
DFTPAV_HD inline dd atan_dd01(dd t) {
  const int k = (int)(t.hi * 64.0 + 0.5);
  const double c = (double)k * 0.015625;
  dd u = t;
  if (k != 0) u = dd_div(dd_add_d(t, -c), dd_add_d(dd_mul_d(t, c), 1.0));
  const dd w = dd_mul(u, u);
  const double inv[9][2] = {{0x1.5555555555555p-2, 0x1.5555555555555p-56},  {0x1.999999999999ap-3, -0x1.999999999999ap-57}, {0x1.2492492492492p-3, 0x1.2492492492492p-57},
                            {0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58},  {0x1.745d1745d1746p-4, -0x1.745d1745d1746p-59}, {0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58},
                            {0x1.1111111111111p-4, 0x1.1111111111111p-60},  {0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61},  {0x1.af286bca1af28p-5, 0x1.af286bca1af28p-59}}; // 1/3 .. 1/19
  dd acc = dd{inv[8][0], inv[8][1]};
  for (int m = 7; m >= 0; m--) acc = dd_add(dd{inv[m][0], inv[m][1]}, dd_neg(dd_mul(acc, w))); // 1/(2m+3) - w (1/(2m+5) - ...)
  // atan u = u - u^3 (1/3 - w (1/5 - ...)) = u - u w acc
  const dd au = dd_add(u, dd_neg(dd_mul(dd_mul(acc, w), u)));
  return k != 0 ? dd_add(atan_tab(k), au) : au;
}
// atan of the quotient of two positive finite doubles, as a double-double in [0, pi / 2]
DFTPAV_HD inline dd atan_ratio_dd(double ay, double ax) {
  const dd pio2 = dd{0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54};
  if (ay <= ax) return atan_dd01(dd_div(dd{ay, 0.0}, dd{ax, 0.0}));
  return dd_add(pio2, dd_neg(atan_dd01(dd_div(dd{ax, 0.0}, dd{ay, 0.0}))));
}
// (cos x and sin x alone: the pair's code, one result used -- a caller that takes both of one argument pays for one)
DFTPAV_HD inline double cos(double x) {
  double s_, c_;
  sincos(x, s_, c_);
  return c_;
}
DFTPAV_HD inline double sin(double x) {
  double s_, c_;
  sincos(x, s_, c_);
  return s_;
}
// correctly rounded atan2(y, x), the special cases as C99 Annex F has them
DFTPAV_HD inline double atan2(double y, double x) {
  const double pi_hi = 0x1.921fb54442d18p+1, pio2_hi = 0x1.921fb54442d18p+0, pio4_hi = 0x1.921fb54442d18p-1, pi34_hi = 0x1.2d97c7f3321d2p+1;
  if (x != x || y != y) return x + y;
  const bool xneg = __builtin_signbit(x), yneg = __builtin_signbit(y);
  const double ax = xneg ? -x : x, ay = yneg ? -y : y;
  const bool xinf = ax > 0x1.fffffffffffffp+1023, yinf = ay > 0x1.fffffffffffffp+1023;
  double r;
  if (ay == 0.0) r = xneg ? pi_hi : 0.0;
  else if (ax == 0.0) r = pio2_hi;
  else if (xinf && yinf) r = xneg ? pi34_hi : pio4_hi;
  else if (xinf) r = xneg ? pi_hi : 0.0;
  else if (yinf) r = pio2_hi;
  else {
    // scale the pair so that neither the quotient nor the products inside dd_div leave the normal range (powers of two: exact)
    double sx = ax, sy = ay;
    const double big = sx > sy ? sx : sy;
    if (big > 0x1.0p+500) { sx *= 0x1.0p-600; sy *= 0x1.0p-600; }
    else if (big < 0x1.0p-500) { sx *= 0x1.0p+600; sy *= 0x1.0p+600; }
    const double small = sx < sy ? sx : sy, large = sx < sy ? sy : sx;
    if (small < large * 0x1.0p-110) { // the quotient is below 2^-110: atan q = q (1 - q^2 / 3 ...) rounds as q does
      const dd pi = dd{0x1.921fb54442d18p+1, 0x1.1a62633145c07p-53}, pio2 = dd{0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54};
      if (sy <= sx) {
        if (!xneg) r = ay / ax; // (the true quotient of the unscaled pair: correctly rounded by IEEE, gradual underflow included)
        else r = dd_add(pi, dd_neg(dd_div(dd{sy, 0.0}, dd{sx, 0.0}))).hi;
      } else {
        const dd q = dd_div(dd{sx, 0.0}, dd{sy, 0.0});
        r = (xneg ? dd_add(pio2, q) : dd_add(pio2, dd_neg(q))).hi;
      }
    } else {
      dd a = atan_ratio_dd(sy, sx);
      if (xneg) a = dd_add(dd{0x1.921fb54442d18p+1, 0x1.1a62633145c07p-53}, dd_neg(a));
      r = a.hi;
    }
  }
  return yneg ? -r : r;
}
// correctly rounded atan x
DFTPAV_HD inline double atan(double x) {
  if (x != x) return x;
  const bool neg = __builtin_signbit(x);
  const double ax = neg ? -x : x;
  double r;
  if (ax == 0.0) return x;
  if (ax > 0x1.0p+110) r = 0x1.921fb54442d18p+0;            // pi / 2 - 1 / x rounds to pi / 2
  else if (ax < 0x1.0p-55) r = ax;                           // x - x^3 / 3 rounds to x  (x^2 / 3 < 2^-111)
  else r = atan_ratio_dd(ax, 1.0).hi;
  return neg ? -r : r;
}
// correctly rounded tan x
DFTPAV_HD inline double tan(double x) {
  if (x == 0.0) return x;
  if (!(x - x == 0.0)) return x - x;
  dd r;
  const int k = (x < 0.0 ? -x : x) < 0x1.0p+20 ? reduce(x, r) : reduce_large(x, r);
  const dd sr = sin_dd(r), cr = cos_dd(r);
  return (k & 1) ? dd_neg(dd_div(cr, sr)).hi : dd_div(sr, cr).hi;
}

} // namespace crt
} // namespace dftpav
