// rs_math.h — shortest Reeds–Shepp path between two poses, written once for host and device.
//
// The reference takes this from a dependency that is not in its tree: KinoAstar builds
// `ompl::base::ReedsSheppStateSpace(1 / max_cur_)` (kino_astar.cpp:423) and uses its `distance` and
// `interpolate` for the analytic "shot" from a search node to the goal (kino_astar.cpp:327-345, 585-599).  OMPL
// is found with find_package, version unpinned (traj_planner/CMakeLists.txt); nothing of it is vendored.
// What follows is a RESTATEMENT OF OMPL'S IMPLEMENTATION (src/ompl/base/spaces/src/ReedsSheppStateSpace.cpp, BSD
// licence; written from knowledge of that file's structure, function by function: LpSpLp, LpSpRp, CSC, LpRmL, CCC,
// tauOmega, LpRupLumRm, LpRumLumRp, CCCC, LpRmSmLm, LpRmSmRm, CCSC, LpRmSLmRp, CCSCC, the `if (f(...) && Lmin > ...)`
// ladder and the 18-row path-type table) -- not only of the paper it implements: J. A. Reeds, L. A. Shepp, "Optimal
// paths for a car that goes both forwards and backwards", Pacific J. Math. 145 (1990) — the 48 words as 18
// path types x {time flip, reflection, backwards}, formulas 8.1-8.11 with the two corrections OMPL's source
// documents (8.3/8.4 and 8.11) — in OMPL's conventions: unit turning radius in normalised coordinates,
// segment lengths signed (negative = reverse), candidates tried in the order CSC, CCC, CCCC, CCSC, CCSCC,
// a candidate replacing the best one only if strictly shorter.  The independent check of this file is
// oracle/shot_oracle_literal.cpp, written from the paper with another organisation (base words x symmetries, every
// candidate validated by integration) and libm; tests/test_shot_oracle.py and tests/test_gpu_parity.py compare the two.
//
// `M` supplies sin / cos / atan2 / sqrt: the portable routines of traj_math.h on the device and in the
// oracle's device-order mode (bit-identical results), libm in the oracle's literal mode.  asin and acos are
// formed from atan2 and sqrt in both.  fmod by 2 pi is an exact operation (the remainder is representable);
// it is written with one FMA and a correction so that it does not depend on a library.
#pragma once
#include "traj_math.h"

namespace dftpav {
namespace rs {

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kZero = 10.0 * 2.220446049250313e-16; // 10 * DBL_EPSILON
enum Seg { NOP = 0, LEFT = 1, STRAIGHT = 2, RIGHT = 3 };

struct PortableMath {
  DFTPAV_HD static double sin(double x) { return p_sin(x); }
  DFTPAV_HD static double cos(double x) { return p_cos(x); }
  DFTPAV_HD static double atan2(double y, double x) { return p_atan2(y, x); }
};

// x - trunc(x / y) * y, exactly (y > 0, |x / y| far below 2^52): the quotient can be off by one when x / y
// rounds across an integer; the remainder of the neighbouring integer differs by exactly y
DFTPAV_HD inline double fmod_pos(double x, double y) {
  double q = x / y;
  q = q < 0.0 ? -floor(-q) : floor(q);
  double r = fma_(-q, y, x);
  if (x >= 0.0) {
    if (r < 0.0) r += y;
    else if (r >= y) r -= y;
  } else {
    if (r > 0.0) r -= y;
    else if (r <= -y) r += y;
  }
  return r;
}
DFTPAV_HD inline double mod2pi(double x) {
  double v = fmod_pos(x, kTwoPi);
  if (v < -kPi) v += kTwoPi;
  else if (v > kPi) v -= kTwoPi;
  return v;
}

struct Path {
  int type;         // row of kTypes
  double len[5];    // signed segment lengths, unit turning radius
  double total;     // sum of |len|
};

// segment kinds of the 18 path types
DFTPAV_HD inline int seg_kind(int type, int i) {
  // two bits per segment, five segments per type
  const unsigned short t[18] = {
      /* 0  L R L . . */ LEFT | RIGHT << 2 | LEFT << 4,
      /* 1  R L R . . */ RIGHT | LEFT << 2 | RIGHT << 4,
      /* 2  L R L R . */ LEFT | RIGHT << 2 | LEFT << 4 | RIGHT << 6,
      /* 3  R L R L . */ RIGHT | LEFT << 2 | RIGHT << 4 | LEFT << 6,
      /* 4  L R S L . */ LEFT | RIGHT << 2 | STRAIGHT << 4 | LEFT << 6,
      /* 5  R L S R . */ RIGHT | LEFT << 2 | STRAIGHT << 4 | RIGHT << 6,
      /* 6  L S R L . */ LEFT | STRAIGHT << 2 | RIGHT << 4 | LEFT << 6,
      /* 7  R S L R . */ RIGHT | STRAIGHT << 2 | LEFT << 4 | RIGHT << 6,
      /* 8  L R S R . */ LEFT | RIGHT << 2 | STRAIGHT << 4 | RIGHT << 6,
      /* 9  R L S L . */ RIGHT | LEFT << 2 | STRAIGHT << 4 | LEFT << 6,
      /* 10 R S R L . */ RIGHT | STRAIGHT << 2 | RIGHT << 4 | LEFT << 6,
      /* 11 L S L R . */ LEFT | STRAIGHT << 2 | LEFT << 4 | RIGHT << 6,
      /* 12 L S R . . */ LEFT | STRAIGHT << 2 | RIGHT << 4,
      /* 13 R S L . . */ RIGHT | STRAIGHT << 2 | LEFT << 4,
      /* 14 L S L . . */ LEFT | STRAIGHT << 2 | LEFT << 4,
      /* 15 R S R . . */ RIGHT | STRAIGHT << 2 | RIGHT << 4,
      /* 16 L R S L R */ LEFT | RIGHT << 2 | STRAIGHT << 4 | LEFT << 6 | RIGHT << 8,
      /* 17 R L S R L */ RIGHT | LEFT << 2 | STRAIGHT << 4 | RIGHT << 6 | LEFT << 8};
  return (t[type] >> (2 * i)) & 3;
}

template <class M> struct Solver {
  DFTPAV_HD static void polar(double x, double y, double &r, double &theta) {
    r = sqrt(x * x + y * y);
    theta = M::atan2(y, x);
  }
  DFTPAV_HD static double asin_(double x) { return M::atan2(x, sqrt((1.0 - x) * (1.0 + x))); }
  DFTPAV_HD static double acos_(double x) { return M::atan2(sqrt((1.0 - x) * (1.0 + x)), x); }
  DFTPAV_HD static void tau_omega(double u, double v, double xi, double eta, double phi, double &tau, double &omega) {
    const double delta = mod2pi(u - v), A = M::sin(u) - M::sin(delta), B = M::cos(u) - M::cos(delta) - 1.0;
    const double t1 = M::atan2(eta * A - xi * B, xi * A + eta * B), t2 = 2.0 * (M::cos(delta) - M::cos(v) - M::cos(u)) + 3.0;
    tau = t2 < 0.0 ? mod2pi(t1 + kPi) : mod2pi(t1);
    omega = mod2pi(tau - u + v - phi);
  }
  // formula 8.1
  DFTPAV_HD static bool LpSpLp(double x, double y, double phi, double &t, double &u, double &v) {
    polar(x - M::sin(phi), y - 1.0 + M::cos(phi), u, t);
    if (t >= -kZero) {
      v = mod2pi(phi - t);
      if (v >= -kZero) return true;
    }
    return false;
  }
  // formula 8.2
  DFTPAV_HD static bool LpSpRp(double x, double y, double phi, double &t, double &u, double &v) {
    double t1, u1;
    polar(x + M::sin(phi), y - 1.0 - M::cos(phi), u1, t1);
    u1 = u1 * u1;
    if (u1 >= 4.0) {
      u = sqrt(u1 - 4.0);
      const double theta = M::atan2(2.0, u);
      t = mod2pi(t1 + theta);
      v = mod2pi(t - phi);
      return t >= -kZero && v >= -kZero;
    }
    return false;
  }
  // formulas 8.3 / 8.4 (as corrected)
  DFTPAV_HD static bool LpRmL(double x, double y, double phi, double &t, double &u, double &v) {
    const double xi = x - M::sin(phi), eta = y - 1.0 + M::cos(phi);
    double u1, theta;
    polar(xi, eta, u1, theta);
    if (u1 <= 4.0) {
      u = -2.0 * asin_(0.25 * u1);
      t = mod2pi(theta + 0.5 * u + kPi);
      v = mod2pi(phi - t + u);
      return t >= -kZero && u <= kZero;
    }
    return false;
  }
  // formula 8.7
  DFTPAV_HD static bool LpRupLumRm(double x, double y, double phi, double &t, double &u, double &v) {
    const double xi = x + M::sin(phi), eta = y - 1.0 - M::cos(phi), rho = 0.25 * (2.0 + sqrt(xi * xi + eta * eta));
    if (rho <= 1.0) {
      u = acos_(rho);
      tau_omega(u, -u, xi, eta, phi, t, v);
      return t >= -kZero && v <= kZero;
    }
    return false;
  }
  // formula 8.8
  DFTPAV_HD static bool LpRumLumRp(double x, double y, double phi, double &t, double &u, double &v) {
    const double xi = x + M::sin(phi), eta = y - 1.0 - M::cos(phi), rho = (20.0 - xi * xi - eta * eta) / 16.0;
    if (rho >= 0.0 && rho <= 1.0) {
      u = -acos_(rho);
      if (u >= -0.5 * kPi) {
        tau_omega(u, u, xi, eta, phi, t, v);
        return t >= -kZero && v >= -kZero;
      }
    }
    return false;
  }
  // formula 8.9
  DFTPAV_HD static bool LpRmSmLm(double x, double y, double phi, double &t, double &u, double &v) {
    const double xi = x - M::sin(phi), eta = y - 1.0 + M::cos(phi);
    double rho, theta;
    polar(xi, eta, rho, theta);
    if (rho >= 2.0) {
      const double r = sqrt(rho * rho - 4.0);
      u = 2.0 - r;
      t = mod2pi(theta + M::atan2(r, -2.0));
      v = mod2pi(phi - 0.5 * kPi - t);
      return t >= -kZero && u <= kZero && v <= kZero;
    }
    return false;
  }
  // formula 8.10
  DFTPAV_HD static bool LpRmSmRm(double x, double y, double phi, double &t, double &u, double &v) {
    const double xi = x + M::sin(phi), eta = y - 1.0 - M::cos(phi);
    double rho, theta;
    polar(-eta, xi, rho, theta);
    if (rho >= 2.0) {
      t = theta;
      u = 2.0 - rho;
      v = mod2pi(t + 0.5 * kPi - phi);
      return t >= -kZero && u <= kZero && v <= kZero;
    }
    return false;
  }
  // formula 8.11 (as corrected)
  DFTPAV_HD static bool LpRmSLmRp(double x, double y, double phi, double &t, double &u, double &v) {
    const double xi = x + M::sin(phi), eta = y - 1.0 - M::cos(phi);
    double rho, theta;
    polar(xi, eta, rho, theta);
    if (rho >= 2.0) {
      u = 4.0 - sqrt(rho * rho - 4.0);
      if (u <= kZero) {
        t = mod2pi(M::atan2((4.0 - u) * xi - 2.0 * eta, -2.0 * xi + (u - 4.0) * eta));
        v = mod2pi(t - phi);
        return t >= -kZero && v >= -kZero;
      }
    }
    return false;
  }

  DFTPAV_HD static void take(Path &best, double &Lmin, double L, int type, double a, double b, double c, double d = 0.0, double e = 0.0) {
    best.type = type;
    best.len[0] = a; best.len[1] = b; best.len[2] = c; best.len[3] = d; best.len[4] = e;
    best.total = fabs(a) + fabs(b) + fabs(c) + fabs(d) + fabs(e);
    Lmin = L;
  }

  // the shortest path from (0, 0, 0) to (x, y, phi), unit turning radius
  DFTPAV_HD static Path shortest(double x, double y, double phi) {
    Path best;
    best.type = 0;
    for (int i = 0; i < 5; i++) best.len[i] = 0.0;
    best.total = 1.7976931348623157e308;
    double t, u, v, L, Lmin;
    // ---- CSC
    Lmin = best.total;
    if (LpSpLp(x, y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 14, t, u, v);
    if (LpSpLp(-x, y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 14, -t, -u, -v); // time flip
    if (LpSpLp(x, -y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 15, t, u, v);    // reflection
    if (LpSpLp(-x, -y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 15, -t, -u, -v); // both
    if (LpSpRp(x, y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 12, t, u, v);
    if (LpSpRp(-x, y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 12, -t, -u, -v);
    if (LpSpRp(x, -y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 13, t, u, v);
    if (LpSpRp(-x, -y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 13, -t, -u, -v);
    // ---- CCC
    Lmin = best.total;
    if (LpRmL(x, y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 0, t, u, v);
    if (LpRmL(-x, y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 0, -t, -u, -v);
    if (LpRmL(x, -y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 1, t, u, v);
    if (LpRmL(-x, -y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 1, -t, -u, -v);
    const double cphi = M::cos(phi), sphi = M::sin(phi);
    const double xb = x * cphi + y * sphi, yb = x * sphi - y * cphi; // the path run backwards
    if (LpRmL(xb, yb, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 0, v, u, t);
    if (LpRmL(-xb, yb, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 0, -v, -u, -t);
    if (LpRmL(xb, -yb, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 1, v, u, t);
    if (LpRmL(-xb, -yb, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 1, -v, -u, -t);
    // ---- CCCC
    Lmin = best.total;
    if (LpRupLumRm(x, y, phi, t, u, v) && Lmin > (L = fabs(t) + 2.0 * fabs(u) + fabs(v))) take(best, Lmin, L, 2, t, u, -u, v);
    if (LpRupLumRm(-x, y, -phi, t, u, v) && Lmin > (L = fabs(t) + 2.0 * fabs(u) + fabs(v))) take(best, Lmin, L, 2, -t, -u, u, -v);
    if (LpRupLumRm(x, -y, -phi, t, u, v) && Lmin > (L = fabs(t) + 2.0 * fabs(u) + fabs(v))) take(best, Lmin, L, 3, t, u, -u, v);
    if (LpRupLumRm(-x, -y, phi, t, u, v) && Lmin > (L = fabs(t) + 2.0 * fabs(u) + fabs(v))) take(best, Lmin, L, 3, -t, -u, u, -v);
    if (LpRumLumRp(x, y, phi, t, u, v) && Lmin > (L = fabs(t) + 2.0 * fabs(u) + fabs(v))) take(best, Lmin, L, 2, t, u, u, v);
    if (LpRumLumRp(-x, y, -phi, t, u, v) && Lmin > (L = fabs(t) + 2.0 * fabs(u) + fabs(v))) take(best, Lmin, L, 2, -t, -u, -u, -v);
    if (LpRumLumRp(x, -y, -phi, t, u, v) && Lmin > (L = fabs(t) + 2.0 * fabs(u) + fabs(v))) take(best, Lmin, L, 3, t, u, u, v);
    if (LpRumLumRp(-x, -y, phi, t, u, v) && Lmin > (L = fabs(t) + 2.0 * fabs(u) + fabs(v))) take(best, Lmin, L, 3, -t, -u, -u, -v);
    // ---- CCSC (the quarter turn is added to the three returned lengths)
    Lmin = best.total - 0.5 * kPi;
    const double hp = 0.5 * kPi;
    if (LpRmSmLm(x, y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 4, t, -hp, u, v);
    if (LpRmSmLm(-x, y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 4, -t, hp, -u, -v);
    if (LpRmSmLm(x, -y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 5, t, -hp, u, v);
    if (LpRmSmLm(-x, -y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 5, -t, hp, -u, -v);
    if (LpRmSmRm(x, y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 8, t, -hp, u, v);
    if (LpRmSmRm(-x, y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 8, -t, hp, -u, -v);
    if (LpRmSmRm(x, -y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 9, t, -hp, u, v);
    if (LpRmSmRm(-x, -y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 9, -t, hp, -u, -v);
    if (LpRmSmLm(xb, yb, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 6, v, u, -hp, t);
    if (LpRmSmLm(-xb, yb, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 6, -v, -u, hp, -t);
    if (LpRmSmLm(xb, -yb, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 7, v, u, -hp, t);
    if (LpRmSmLm(-xb, -yb, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 7, -v, -u, hp, -t);
    if (LpRmSmRm(xb, yb, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 10, v, u, -hp, t);
    if (LpRmSmRm(-xb, yb, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 10, -v, -u, hp, -t);
    if (LpRmSmRm(xb, -yb, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 11, v, u, -hp, t);
    if (LpRmSmRm(-xb, -yb, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 11, -v, -u, hp, -t);
    // ---- CCSCC (two quarter turns)
    Lmin = best.total - kPi;
    if (LpRmSLmRp(x, y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 16, t, -hp, u, -hp, v);
    if (LpRmSLmRp(-x, y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 16, -t, hp, -u, hp, -v);
    if (LpRmSLmRp(x, -y, -phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 17, t, -hp, u, -hp, v);
    if (LpRmSLmRp(-x, -y, phi, t, u, v) && Lmin > (L = fabs(t) + fabs(u) + fabs(v))) take(best, Lmin, L, 17, -t, hp, -u, hp, -v);
    return best;
  }

  // ReedsSheppStateSpace::reedsShepp(state1, state2): the goal in the frame of the start, turning radius rho
  DFTPAV_HD static Path between(const double from[3], const double to[3], double rho) {
    const double dx = to[0] - from[0], dy = to[1] - from[1], c = M::cos(from[2]), s = M::sin(from[2]);
    const double x = c * dx + s * dy, y = -s * dx + c * dy, phi = to[2] - from[2];
    return shortest(x / rho, y / rho, phi);
  }

  // ReedsSheppStateSpace::interpolate(from, path, t, state): the pose at fraction t of the path; yaw as SO2
  // enforceBounds leaves it, in [-pi, pi)
  DFTPAV_HD static void interpolate(const double from[3], const Path &path, double rho, double t, double out[3]) {
    double seg = t * path.total, x = 0.0, y = 0.0, yaw = from[2];
    for (int i = 0; i < 5 && seg > 0.0; i++) {
      double v;
      if (path.len[i] < 0.0) {
        v = -seg > path.len[i] ? -seg : path.len[i];
        seg += v;
      } else {
        v = seg < path.len[i] ? seg : path.len[i];
        seg -= v;
      }
      const double phi = yaw;
      switch (seg_kind(path.type, i)) {
        case LEFT:
          x = x + M::sin(phi + v) - M::sin(phi);
          y = y - M::cos(phi + v) + M::cos(phi);
          yaw = phi + v;
          break;
        case RIGHT:
          x = x - M::sin(phi - v) + M::sin(phi);
          y = y + M::cos(phi - v) - M::cos(phi);
          yaw = phi - v;
          break;
        case STRAIGHT:
          x = x + v * M::cos(phi);
          y = y + v * M::sin(phi);
          break;
        default: break;
      }
    }
    out[0] = x * rho + from[0];
    out[1] = y * rho + from[1];
    double w = fmod_pos(yaw, kTwoPi);
    if (w < -kPi) w += kTwoPi;
    else if (w >= kPi) w -= kTwoPi;
    out[2] = w;
  }
};

} // namespace rs
} // namespace dftpav
