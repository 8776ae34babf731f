// TEST INFRASTRUCTURE ONLY (see oracle/dftpav_oracle.h): CPU restatement of the Reeds-Shepp shot of the front end,
// SURVEY.md §8(f)-3.
//
//   KinoAstar::computeShotTraj / is_shot_sucess       traj_planner/src/kino_astar.cpp:304-345
//   ompl::base::ReedsSheppStateSpace::distance / interpolate   (OMPL, not vendored, version unpinned: the published
//     algorithm is restated in dftpav_amd/csrc/rs_math.h, shared with the kernel; Reeds & Shepp 1990, formulas 8.1-8.11)
//   SemanticMapManager::CheckCollisionUsingPosAndYaw  semantic_map_manager.cc:639-662
//
// order 0: libm sin / cos / atan2, as OMPL and the reference use them.  order 1: the portable functions of
// traj_math.h, which is what the HIP kernel evaluates -- bit-identical to the GPU.  PARITY UNPINNED against OMPL
// itself (not vendored by the reference, absent here: the one step of SURVEY §8(f) that can only stay property-pinned; the
// collision test it ends in IS pinned, through validate_oracle.cpp); the pins are properties: the interpolated path ends on the goal, its pieces respect the
// turning radius, no other word is shorter, and the symmetries of the problem (tests/test_shot_oracle.py).
#include <cmath>
#include <cstdint>

#include "../dftpav_amd/csrc/rs_math.h"

namespace {

struct LibmMath {
  static double sin(double x) { return std::sin(x); }
  static double cos(double x) { return std::cos(x); }
  static double atan2(double y, double x) { return std::atan2(y, x); }
};

struct Grid {
  const unsigned char *data;
  int sx, sy;
  double res, ox, oy;
};
inline bool occupied(const Grid &g, double x, double y) {
  const double cx = std::round((x - g.ox) / g.res), cy = std::round((y - g.oy) / g.res);
  if (!(cx >= 0.0 && cx < (double)g.sx && cy >= 0.0 && cy < (double)g.sy)) return false;
  return g.data[(int)cx + g.sx * (int)cy] == 80;
}
inline bool edge_hits(const Grid &g, double ax, double ay, double bx, double by, double res) {
  const double dx = bx - ax, dy = by - ay;
  const double norm = std::sqrt(dx * dx + dy * dy);
  for (double dl = res; dl < norm; dl += res) {
    const double f = dl / norm;
    if (occupied(g, f * dx + ax, f * dy + ay)) return true;
  }
  return false;
}
inline bool pose_collides(const Grid &g, double px, double py, double yaw, double W, double L, double dcr, double vres, int order) {
  const double cs = order ? dftpav::p_cos(yaw) : std::cos(yaw), sn = order ? dftpav::p_sin(yaw) : std::sin(yaw);
  const double x = px + dcr * cs, y = py + dcr * sn;
  const double c1x = x + 0.5 * L * cs + 0.5 * W * sn, c1y = y + 0.5 * L * sn - 0.5 * W * cs;
  const double c2x = x + 0.5 * L * cs - 0.5 * W * sn, c2y = y + 0.5 * L * sn + 0.5 * W * cs;
  const double c3x = x - 0.5 * L * cs - 0.5 * W * sn, c3y = y - 0.5 * L * sn + 0.5 * W * cs;
  const double c4x = x - 0.5 * L * cs + 0.5 * W * sn, c4y = y - 0.5 * L * sn - 0.5 * W * cs;
  if (edge_hits(g, c1x, c1y, c2x, c2y, vres)) return true;
  if (edge_hits(g, c2x, c2y, c3x, c3y, vres)) return true;
  if (edge_hits(g, c3x, c3y, c4x, c4y, vres)) return true;
  if (edge_hits(g, c4x, c4y, c1x, c1y, vres)) return true;
  return occupied(g, c1x, c1y) || occupied(g, c2x, c2y) || occupied(g, c3x, c3y) || occupied(g, c4x, c4y);
}

template <class M>
void run(const unsigned char *grid, int size_x, int size_y, double resolution, double origin_x, double origin_y, const double *from,
         const double *to, int n, double rho, double checkl, int max_samples, double veh_width, double veh_length, double veh_dcr,
         double vertex_res, int order, double *length, int *type, double *seg, double *samples, int *n_samples, int *collides) {
  typedef dftpav::rs::Solver<M> RS;
  const Grid g{grid, size_x, size_y, resolution, origin_x, origin_y};
  for (int i = 0; i < n; i++) {
    const double *f = from + 3 * (size_t)i, *t = to + 3 * (size_t)i;
    const dftpav::rs::Path path = RS::between(f, t, rho);
    const double len = rho * path.total;
    length[i] = len;
    type[i] = path.type;
    for (int k = 0; k < 5; k++) seg[5 * (size_t)i + k] = path.len[k];
    double *out = samples + (size_t)i * max_samples * 3;
    for (int k = 0; k < 3 * max_samples; k++) out[k] = 0.0;
    int cnt = 0, hit = 0;
    for (double l = 0.0; l <= len; l += checkl) { // kino_astar.cpp:338
      if (cnt < max_samples) {
        double s[3];
        const double tt = l / len;
        if (tt >= 1.0) { s[0] = t[0]; s[1] = t[1]; s[2] = t[2]; }
        else if (tt <= 0.0) { s[0] = f[0]; s[1] = f[1]; s[2] = f[2]; }
        else RS::interpolate(f, path, rho, tt, s);
        out[3 * cnt] = s[0]; out[3 * cnt + 1] = s[1]; out[3 * cnt + 2] = s[2];
        if (grid && pose_collides(g, s[0], s[1], s[2], veh_width, veh_length, veh_dcr, vertex_res, order)) hit = 1;
      }
      cnt++;
    }
    n_samples[i] = cnt;
    if (collides) collides[i] = hit;
  }
}

} // namespace

extern "C" void oracle_reeds_shepp_shots(const unsigned char *grid, int size_x, int size_y, double resolution, double origin_x,
                                         double origin_y, const double *from, const double *to, int n, double rho, double checkl,
                                         int max_samples, double veh_width, double veh_length, double veh_dcr, double vertex_res,
                                         int order, double *length, int *type, double *seg, double *samples, int *n_samples,
                                         int *collides) {
  if (order)
    run<dftpav::rs::PortableMath>(grid, size_x, size_y, resolution, origin_x, origin_y, from, to, n, rho, checkl, max_samples,
                                  veh_width, veh_length, veh_dcr, vertex_res, order, length, type, seg, samples, n_samples, collides);
  else
    run<LibmMath>(grid, size_x, size_y, resolution, origin_x, origin_y, from, to, n, rho, checkl, max_samples, veh_width,
                  veh_length, veh_dcr, vertex_res, order, length, type, seg, samples, n_samples, collides);
}
