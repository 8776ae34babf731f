// TEST INFRASTRUCTURE ONLY (see oracle/dftpav_oracle.h): CPU restatement of the step that feeds the
// solve path its half-planes, SURVEY.md §8(f)-1.
//
//   TrajPlanner::getRectangleConst                 traj_planner/src/traj_manager.cpp:1213-1469
//   TrajPlannerAdapter::CheckIfCollisionUsingLine  traj_planner/src/map_adapter.cpp:117-129
//   GridMapND::CheckIfEqualUsingGlobalPosition     common/src/common/basics/semantics.cc:169-179,214-221,264-288
//   common::VehicleParam defaults                  common/inc/common/basics/semantics.h:66-76
//
// For every state (x, y, yaw) a vehicle-aligned rectangle is grown side by side (+dy, +dx, -dy, -dx, in
// turn) in steps of one grid cell until the U-shaped strip a step would add touches an occupied cell
// or the side has grown by 10 m; the rectangle is returned as four half-planes (outward normal,
// point on the edge), the 4x4 matrix the optimiser consumes (traj_manager.cpp:1442-1465).
//
// order 0: cos / sin of libm, as the reference.  order 1: the portable cos / sin of
// dftpav_amd/csrc/traj_math.h, which is what the HIP kernel evaluates; every other operation is a
// correctly rounded IEEE operation in the reference's order, so order 1 is bit-identical to the GPU.
// PINNED (round 5): order 0 is bit-equal to the reference's own code -- the cited functions cut verbatim out of
// /root/reference (oracle/ref_slices.py) and compiled into oracle/_ref/libdftpav_ref_next.so (oracle/ref_next_driver.cpp) --
// on the scenarios the GPU tests of this step use (tests/test_ref_pin.py::test_corridor_oracle_is_bit_equal_to_getRectangleConst).
#include <cmath>
#include <cstdint>

#include "../dftpav_amd/csrc/traj_math.h"
#include "step_trig.h"

namespace {

struct Grid {
  const unsigned char *data;
  int sx, sy;
  double res, ox, oy;
};

// GridMapND::CheckIfEqualUsingGlobalPosition(p, OCCUPIED): coord = round((p - origin) / resolution),
// out of range counts as free (semantics.cc:169-179, 214-221); dims_step = {1, size_x} (semantics.cc:303-311)
inline bool occupied(const Grid &g, double x, double y) {
  const double cx = std::round((x - g.ox) / g.res), cy = std::round((y - g.oy) / g.res);
  if (!(cx >= 0.0 && cx < (double)g.sx && cy >= 0.0 && cy < (double)g.sy)) return false;
  return g.data[(int)cx + g.sx * (int)cy] == 80; // GridMapND::OCCUPIED
}

// map_adapter.cpp:117-129
inline bool line_hits(const Grid &g, double p1x, double p1y, double p2x, double p2y, double checkl) {
  const double dx = p2x - p1x, dy = p2y - p1y;
  const double norm = std::sqrt(dx * dx + dy * dy);
  for (double dl = 0.0; dl < norm; dl += checkl) {
    const double px = dx * dl / norm + p1x, py = dy * dl / norm + p1y;
    if (occupied(g, px, py)) return true;
  }
  return occupied(g, p2x, p2y);
}

} // namespace

extern "C" void oracle_corridor_rectangles(const unsigned char *grid, int size_x, int size_y, double resolution, double origin_x,
                                           double origin_y, const double *states, int n, double veh_width, double veh_length,
                                           double veh_dcr, int order, double *hpoly) {
  const Grid g{grid, size_x, size_y, resolution, origin_x, origin_y};
  const double step = resolution * 1.0; // traj_manager.cpp:1218
  const double limit = 10.0;            // :1219
  const double checkl = resolution / 2.0;
  for (int i = 0; i < n; i++) {
    const double rx = states[3 * i], ry = states[3 * i + 1], yaw = states[3 * i + 2];
    const step_trig::Trig T{order};
    const double c = T.cos(yaw), s = T.sin(yaw);
    const double ns = -s;                              // egoR = [c -s; s c], :1233-1234
    auto at = [&](double sxp, double syp, double a, double b, double &ox, double &oy) { // pt + egoR * (a, b)
      ox = sxp + (c * a + ns * b);
      oy = syp + (s * a + c * b);
    };
    double sx = rx, sy = ry, W = veh_width, L = veh_length; // sourcePt, sourceVp
    const double dcr = veh_dcr;
    double expand[4] = {0.0, 0.0, 0.0, 0.0};
    bool open[4] = {true, true, true, true};
    while (open[0] || open[1] || open[2] || open[3]) { // NotFinishTable.norm() > 0
      for (int side = 0; side < 4; side++) {
        if (!open[side]) continue;
        double a1, b1, a2, b2, na1, nb1, na2, nb2; // body coordinates of point1, point2, newpoint1, newpoint2
        switch (side) {
          case 0: // +dy, :1311-1315
            a1 = L / 2.0 + dcr; b1 = W / 2.0; a2 = -L / 2.0 + dcr; b2 = W / 2.0;
            na1 = L / 2.0 + dcr; nb1 = W / 2.0 + step; na2 = -L / 2.0 + dcr; nb2 = W / 2.0 + step;
            break;
          case 1: // +dx, :1343-1347
            a1 = L / 2.0 + dcr; b1 = -W / 2.0; a2 = L / 2.0 + dcr; b2 = W / 2.0;
            na1 = step + L / 2.0 + dcr; nb1 = -W / 2.0; na2 = step + L / 2.0 + dcr; nb2 = W / 2.0;
            break;
          case 2: // -dy, :1375-1379
            a1 = -L / 2.0 + dcr; b1 = -W / 2.0; a2 = L / 2.0 + dcr; b2 = -W / 2.0;
            na1 = -L / 2.0 + dcr; nb1 = -W / 2.0 - step; na2 = L / 2.0 + dcr; nb2 = -W / 2.0 - step;
            break;
          default: // -dx, :1407-1411
            a1 = -L / 2.0 + dcr; b1 = W / 2.0; a2 = -L / 2.0 + dcr; b2 = -W / 2.0;
            na1 = -L / 2.0 + dcr - step; nb1 = W / 2.0; na2 = -L / 2.0 + dcr - step; nb2 = -W / 2.0;
            break;
        }
        double p1x, p1y, p2x, p2y, n1x, n1y, n2x, n2y;
        at(sx, sy, a1, b1, p1x, p1y);
        at(sx, sy, a2, b2, p2x, p2y);
        at(sx, sy, na1, nb1, n1x, n1y);
        at(sx, sy, na2, nb2, n2x, n2y);
        // point1 -> newpoint1 -> newpoint2 -> point2
        if (line_hits(g, p1x, p1y, n1x, n1y, checkl) || line_hits(g, n1x, n1y, n2x, n2y, checkl) ||
            line_hits(g, n2x, n2y, p2x, p2y, checkl)) {
          open[side] = false;
          continue;
        }
        expand[side] += step;
        if (expand[side] >= limit) { // the centre / size update is skipped on the closing step, :1332-1335
          open[side] = false;
          continue;
        }
        double ma, mb; // centre shift in body coordinates
        switch (side) {
          case 0: ma = 0.0; mb = step / 2.0; W = W + step; break;
          case 1: ma = step / 2.0; mb = 0.0; L = L + step; break;
          case 2: ma = 0.0; mb = -step / 2.0; W = W + step; break;
          default: ma = -step / 2.0; mb = 0.0; L = L + step; break;
        }
        at(sx, sy, ma, mb, sx, sy);
      }
    }
    // traj_manager.cpp:1442-1465: columns (normal; point), from the RAW pose and size plus the expansions
    double *H = hpoly + 16 * i;
    const double W0 = veh_width, L0 = veh_length;
    double px, py;
    at(rx, ry, L0 / 2.0 + dcr + expand[1], W0 / 2.0 + expand[0], px, py);
    H[0] = -s; H[1] = c; H[2] = px; H[3] = py;
    at(rx, ry, L0 / 2.0 + dcr + expand[1], -W0 / 2.0 - expand[2], px, py);
    H[4] = c; H[5] = s; H[6] = px; H[7] = py;
    at(rx, ry, -L0 / 2.0 + dcr - expand[3], -W0 / 2.0 - expand[2], px, py);
    H[8] = s; H[9] = -c; H[10] = px; H[11] = py;
    at(rx, ry, -L0 / 2.0 + dcr - expand[3], W0 / 2.0 + expand[0], px, py);
    H[12] = -c; H[13] = -s; H[14] = px; H[15] = py;
  }
}
