// TEST INFRASTRUCTURE ONLY (see oracle/dftpav_oracle.h): an INDEPENDENT restatement of the shortest Reeds-Shepp path,
// SURVEY.md §8(f)-3.  It does not include dftpav_amd/csrc/rs_math.h (which the kernel and oracle/shot_oracle.cpp share, and
// which follows the structure of OMPL's ReedsSheppStateSpace.cpp); it is written from the paper —
//   J. A. Reeds, L. A. Shepp, "Optimal paths for a car that goes both forwards and backwards", Pacific J. Math. 145 (1990),
//   section 8: the eight base words with formulas 8.1-8.4, 8.7-8.11, and the three symmetries that generate the 48 —
// with libm (sin, cos, atan2, asin, acos, fmod, sqrt) and a different organisation: every base word is solved for each of
// the eight symmetry images of the goal, the solution is mapped back by the symmetry itself (time flip = negate the segment
// lengths, reflection = swap L and R, backwards = reverse the segment order), and a candidate only counts if INTEGRATING it
// from the origin actually lands on the goal pose.  The shortest valid candidate wins.
// What the reference calls: ompl::base::ReedsSheppStateSpace::distance / interpolate from KinoAstar::computeShotTraj
// (traj_planner/src/kino_astar.cpp:304-345); OMPL is not in the reference tree (parity against OMPL itself unpinned).
#include <cmath>
#include <cstddef>

namespace {
const double PI = 3.14159265358979323846;
enum { N = 0, L = 1, S = 2, R = 3 };

double wrap(double a) { // to [-pi, pi]
  double v = std::fmod(a, 2.0 * PI);
  if (v < -PI) v += 2.0 * PI;
  else if (v > PI) v -= 2.0 * PI;
  return v;
}
void to_polar(double x, double y, double &r, double &th) {
  r = std::sqrt(x * x + y * y);
  th = std::atan2(y, x);
}
const double TOL = 10.0 * 2.220446049250313e-16;

struct Word {
  int kind[5];
  double len[5];
  int n;
};

// ---- the eight base words (goal (x, y, phi) in units of the turning radius, start at the origin heading +x)
// each returns false when the word does not exist for this goal
bool w_LSL(double x, double y, double phi, Word &w) { // 8.1  L+ S+ L+
  double u, t;
  to_polar(x - std::sin(phi), y - 1.0 + std::cos(phi), u, t);
  if (t < -TOL) return false;
  double v = wrap(phi - t);
  if (v < -TOL) return false;
  w = Word{{L, S, L, N, N}, {t, u, v, 0, 0}, 3};
  return true;
}
bool w_LSR(double x, double y, double phi, Word &w) { // 8.2  L+ S+ R+
  double u1, t1;
  to_polar(x + std::sin(phi), y - 1.0 - std::cos(phi), u1, t1);
  double q = u1 * u1;
  if (q < 4.0) return false;
  double u = std::sqrt(q - 4.0);
  double t = wrap(t1 + std::atan2(2.0, u));
  double v = wrap(t - phi);
  if (t < -TOL || v < -TOL) return false;
  w = Word{{L, S, R, N, N}, {t, u, v, 0, 0}, 3};
  return true;
}
bool w_LRL(double x, double y, double phi, Word &w) { // 8.3 / 8.4  L+ R- L
  double xi = x - std::sin(phi), eta = y - 1.0 + std::cos(phi), u1, th;
  to_polar(xi, eta, u1, th);
  if (u1 > 4.0) return false;
  double u = -2.0 * std::asin(0.25 * u1);
  double t = wrap(th + 0.5 * u + PI);
  double v = wrap(phi - t + u);
  if (t < -TOL || u > TOL) return false;
  w = Word{{L, R, L, N, N}, {t, u, v, 0, 0}, 3};
  return true;
}
void tau_omega(double u, double v, double xi, double eta, double phi, double &tau, double &omega) { // 8.5 / 8.6
  double delta = wrap(u - v), A = std::sin(u) - std::sin(delta), B = std::cos(u) - std::cos(delta) - 1.0;
  double t1 = std::atan2(eta * A - xi * B, xi * A + eta * B);
  double t2 = 2.0 * (std::cos(delta) - std::cos(v) - std::cos(u)) + 3.0;
  tau = t2 < 0.0 ? wrap(t1 + PI) : wrap(t1);
  omega = wrap(tau - u + v - phi);
}
bool w_LRLR_a(double x, double y, double phi, Word &w) { // 8.7  L+ R+(u) L-(u) R-
  double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi);
  double rho = 0.25 * (2.0 + std::sqrt(xi * xi + eta * eta));
  if (rho > 1.0) return false;
  double u = std::acos(rho), t, v;
  tau_omega(u, -u, xi, eta, phi, t, v);
  if (t < -TOL || v > TOL) return false;
  w = Word{{L, R, L, R, N}, {t, u, -u, v, 0}, 4};
  return true;
}
bool w_LRLR_b(double x, double y, double phi, Word &w) { // 8.8  L+ R-(u) L-(u) R+
  double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi);
  double rho = (20.0 - xi * xi - eta * eta) / 16.0;
  if (rho < 0.0 || rho > 1.0) return false;
  double u = -std::acos(rho);
  if (u < -0.5 * PI) return false;
  double t, v;
  tau_omega(u, u, xi, eta, phi, t, v);
  if (t < -TOL || v < -TOL) return false;
  w = Word{{L, R, L, R, N}, {t, u, u, v, 0}, 4};
  return true;
}
bool w_LRSL(double x, double y, double phi, Word &w) { // 8.9  L+ R-(pi/2) S- L-
  double xi = x - std::sin(phi), eta = y - 1.0 + std::cos(phi), rho, th;
  to_polar(xi, eta, rho, th);
  if (rho < 2.0) return false;
  double r = std::sqrt(rho * rho - 4.0);
  double u = 2.0 - r;
  double t = wrap(th + std::atan2(r, -2.0));
  double v = wrap(phi - 0.5 * PI - t);
  if (t < -TOL || u > TOL || v > TOL) return false;
  w = Word{{L, R, S, L, N}, {t, -0.5 * PI, u, v, 0}, 4};
  return true;
}
bool w_LRSR(double x, double y, double phi, Word &w) { // 8.10  L+ R-(pi/2) S- R-
  double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi), rho, th;
  to_polar(-eta, xi, rho, th);
  if (rho < 2.0) return false;
  double t = th, u = 2.0 - rho;
  double v = wrap(t + 0.5 * PI - phi);
  if (t < -TOL || u > TOL || v > TOL) return false;
  w = Word{{L, R, S, R, N}, {t, -0.5 * PI, u, v, 0}, 4};
  return true;
}
bool w_LRSLR(double x, double y, double phi, Word &w) { // 8.11  L+ R-(pi/2) S- L-(pi/2) R+
  double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi), rho, th;
  to_polar(xi, eta, rho, th);
  if (rho < 2.0) return false;
  double u = 4.0 - std::sqrt(rho * rho - 4.0);
  if (u > TOL) return false;
  double t = wrap(std::atan2((4.0 - u) * xi - 2.0 * eta, -2.0 * xi + (u - 4.0) * eta));
  double v = wrap(t - phi);
  if (t < -TOL || v < -TOL) return false;
  w = Word{{L, R, S, L, R}, {t, -0.5 * PI, u, -0.5 * PI, v}, 5};
  return true;
}
typedef bool (*BaseWord)(double, double, double, Word &);
// in the order the families are usually listed: CSC, CCC, CCCC, CCSC, CCSCC
const BaseWord kBase[8] = {w_LSL, w_LSR, w_LRL, w_LRLR_a, w_LRLR_b, w_LRSL, w_LRSR, w_LRSLR};

// pose reached by driving `frac` of the word from (0, 0, yaw0): exact arcs of unit radius and straight lines
void drive(const Word &w, double yaw0, double upto, double &x, double &y, double &yaw) {
  x = 0.0;
  y = 0.0;
  yaw = yaw0;
  double left = upto; // arc length still to drive
  for (int i = 0; i < w.n && left > 0.0; i++) {
    double seg = std::fabs(w.len[i]);
    double go = seg < left ? seg : left;
    double v = w.len[i] < 0.0 ? -go : go; // signed
    left -= go;
    if (w.kind[i] == L) {
      x += std::sin(yaw + v) - std::sin(yaw);
      y += -std::cos(yaw + v) + std::cos(yaw);
      yaw += v;
    } else if (w.kind[i] == R) {
      x += -std::sin(yaw - v) + std::sin(yaw);
      y += std::cos(yaw - v) - std::cos(yaw);
      yaw -= v;
    } else if (w.kind[i] == S) {
      x += v * std::cos(yaw);
      y += v * std::sin(yaw);
    }
  }
}
double total(const Word &w) {
  double s = 0.0;
  for (int i = 0; i < w.n; i++) s += std::fabs(w.len[i]);
  return s;
}

// shortest valid word from the origin to (x, y, phi); returns its length (in turning radii)
double shortest(double x, double y, double phi, Word &best, int &n_valid) {
  double best_len = 1e300;
  n_valid = 0;
  const double xb = x * std::cos(phi) + y * std::sin(phi), yb = x * std::sin(phi) - y * std::cos(phi);
  for (int b = 0; b < 8; b++)
    for (int sym = 0; sym < 8; sym++) {
      const bool back = sym & 4, flip = sym & 1, refl = sym & 2;
      double gx = back ? xb : x, gy = back ? yb : y, gp = phi;
      if (flip) { gx = -gx; gp = -gp; }
      if (refl) { gy = -gy; gp = -gp; }
      Word w;
      if (!kBase[b](gx, gy, gp, w)) continue;
      // map the solution back through the symmetry
      if (flip) for (int i = 0; i < w.n; i++) w.len[i] = -w.len[i];
      if (refl) for (int i = 0; i < w.n; i++) w.kind[i] = w.kind[i] == L ? R : (w.kind[i] == R ? L : w.kind[i]);
      if (back)
        for (int i = 0; i < w.n / 2; i++) {
          int k = w.kind[i]; w.kind[i] = w.kind[w.n - 1 - i]; w.kind[w.n - 1 - i] = k;
          double l = w.len[i]; w.len[i] = w.len[w.n - 1 - i]; w.len[w.n - 1 - i] = l;
        }
      // the candidate has to land on the goal
      double ex, ey, eyaw;
      const double len = total(w);
      drive(w, 0.0, len * (1.0 + 1e-15) + 1e-300, ex, ey, eyaw);
      const double dyaw = wrap(eyaw - phi);
      if (std::fabs(ex - x) > 1e-8 || std::fabs(ey - y) > 1e-8 || std::fabs(dyaw) > 1e-8) continue;
      n_valid++;
      if (len < best_len) {
        best_len = len;
        best = w;
      }
    }
  return best_len;
}
} // namespace

// from / to [n][3] (x, y, yaw); rho: turning radius.  Outputs: length [n] (metres), kinds [n][5] (0 none, 1 left,
// 2 straight, 3 right), seg [n][5] signed segment lengths in turning radii, samples [n][max_samples][3] every `checkl`
// metres along the path as KinoAstar::computeShotTraj walks it (kino_astar.cpp:338), n_samples [n], n_valid [n] (how many
// of the 64 symmetry images produced a word that reaches the goal)
extern "C" void oracle_reeds_shepp_literal(const double *from, const double *to, int n, double rho, double checkl, int max_samples,
                                           double *length, int *kinds, double *seg, double *samples, int *n_samples, int *n_valid) {
  for (int i = 0; i < n; i++) {
    const double *f = from + 3 * (size_t)i, *t = to + 3 * (size_t)i;
    const double dx = t[0] - f[0], dy = t[1] - f[1], c = std::cos(f[2]), s = std::sin(f[2]);
    const double x = (c * dx + s * dy) / rho, y = (-s * dx + c * dy) / rho, phi = t[2] - f[2];
    Word w{};
    int nv = 0;
    const double len_r = shortest(x, y, wrap(phi), w, nv);
    const double len = rho * len_r;
    length[i] = len;
    n_valid[i] = nv;
    for (int k = 0; k < 5; k++) {
      kinds[5 * (size_t)i + k] = k < w.n ? w.kind[k] : 0;
      seg[5 * (size_t)i + k] = k < w.n ? w.len[k] : 0.0;
    }
    double *out = samples + (size_t)i * max_samples * 3;
    for (int k = 0; k < 3 * max_samples; k++) out[k] = 0.0;
    int cnt = 0;
    for (double l = 0.0; l <= len; l += checkl) {
      if (cnt < max_samples) {
        const double tt = l / len;
        double px, py, pyaw;
        if (tt >= 1.0) { px = t[0]; py = t[1]; pyaw = t[2]; }
        else if (tt <= 0.0) { px = f[0]; py = f[1]; pyaw = f[2]; }
        else {
          double ex, ey, eyaw;
          drive(w, f[2], tt * len_r, ex, ey, eyaw);
          px = ex * rho + f[0];
          py = ey * rho + f[1];
          pyaw = std::fmod(eyaw, 2.0 * PI); // SO(2) bounds: [-pi, pi)
          if (pyaw < -PI) pyaw += 2.0 * PI;
          else if (pyaw >= PI) pyaw -= 2.0 * PI;
        }
        out[3 * cnt] = px; out[3 * cnt + 1] = py; out[3 * cnt + 2] = pyaw;
      }
      cnt++;
    }
    n_samples[i] = cnt;
  }
}
