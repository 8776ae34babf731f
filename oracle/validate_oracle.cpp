// TEST INFRASTRUCTURE ONLY (see oracle/dftpav_oracle.h): CPU restatement of the step that consumes the
// solve path's result, SURVEY.md §8(f)-2 — the sampled collision re-check of an optimised trajectory.
//
//   TrajPlannerServer::CheckReplan, collision part   traj_planner/src/traj_server_ros.cpp:385-397
//   Trajectory::getPos / getAngle / locatePieceIdx    plan_utils/poly_traj_utils.hpp:510-528, 77-87, 179-192, 237-244
//   MinJerkOpt::getTraj (column order of a piece)     plan_utils/poly_traj_utils.hpp:987-997
//   SemanticMapManager::CheckCollisionUsingPosAndYaw  semantic_map_manager.cc:639-662
//   ShapeUtils::GetDenseVerticesOfOrientedBoundingBox common/src/common/basics/shapes.cc:110-149 (res = 0.1, shapes.h:200-201)
//   GridMapND::CheckIfEqualUsingGlobalPosition        common/src/common/basics/semantics.cc:169-179,214-221
//
// Every segment of a trajectory is sampled at t = 0, dt, dt + dt, ... < duration; at each sample the
// vehicle's outline (dense points along the four edges, then the corners) is tested against the
// occupancy grid.  The result per trajectory is whether any sample collides and the index of the first
// one (counted over the segments in order).
//
// order 0: libm cos / sin / atan2, as the reference.  order 1: the portable functions of traj_math.h that
// the HIP kernel evaluates; everything else is correctly rounded IEEE arithmetic in the reference's order,
// so order 1 is bit-identical to the GPU.  
// PINNED (round 5): order 0 is bit-equal to the reference's own code -- the cited functions cut verbatim out of
// /root/reference (oracle/ref_slices.py) and compiled into oracle/_ref/libdftpav_ref_next.so (oracle/ref_next_driver.cpp) --
// on the scenarios the GPU tests of this step use (tests/test_ref_pin.py::test_validate_oracle_is_bit_equal_to_CheckReplan).
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
inline bool occupied(const Grid &g, double x, double y) {
  const double cx = std::round((x - g.ox) / g.res), cy = std::round((y - g.oy) / g.res);
  if (!(cx >= 0.0 && cx < (double)g.sx && cy >= 0.0 && cy < (double)g.sy)) return false;
  return g.data[(int)cx + g.sx * (int)cy] == 80;
}

// one edge of GetDenseVerticesOfOrientedBoundingBox: dl = res, res + res, ... < |b - a|
inline bool edge_hits(const Grid &g, double ax, double ay, double bx, double by, double res) {
  const double dx = bx - ax, dy = by - ay;
  const double norm = std::sqrt(dx * dx + dy * dy);
  for (double dl = res; dl < norm; dl += res) {
    const double f = dl / norm;
    if (occupied(g, f * dx + ax, f * dy + ay)) return true;
  }
  return false;
}

// CheckCollisionUsingPosAndYaw (semantic_map_manager.cc:639-662)
inline bool pose_collides(const Grid &g, double px, double py, double yaw, double W, double L, double dcr, double vres, int order) {
  const step_trig::Trig T{order};
  const double cs = T.cos(yaw), sn = T.sin(yaw);
  const double x = px + dcr * cs, y = py + dcr * sn; // obb centre, :645-646
  // shapes.cc:116-127
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

} // namespace

// coeffs: [B][Ntot][6][2], entry [k][d] = coefficient of s^k (what dftpav_batch_coeffs / oracle coeffs return);
// piece_dt: [B][M] piece duration of each segment
extern "C" void oracle_validate_trajectories(const unsigned char *grid, int size_x, int size_y, double resolution, double origin_x,
                                             double origin_y, const double *coeffs, const double *piece_dt, const int *piece_nums,
                                             const int *singuls, int M, int B, double veh_width, double veh_length,
                                             double veh_dcr, double sample_dt, double vertex_res, int order, int *collision,
                                             int *first_sample) {
  const Grid g{grid, size_x, size_y, resolution, origin_x, origin_y};
  int Ntot = 0;
  for (int i = 0; i < M; i++) Ntot += piece_nums[i];
  for (int b = 0; b < B; b++) {
    const double *cb = coeffs + (size_t)b * Ntot * 12;
    int hit = 0, first = -1, counter = 0, p0 = 0;
    for (int i = 0; i < M && !hit; i++) {
      const int N = piece_nums[i];
      const double dtp = piece_dt[(size_t)b * M + i];
      double duration = 0.0; // Trajectory::getTotalDuration: the piece durations summed in order
      for (int p = 0; p < N; p++) duration += dtp;
      for (double t = 0.0; t < duration; t += sample_dt, counter++) { // traj_server_ros.cpp:387
        // locatePieceIdx, poly_traj_utils.hpp:510-528
        double tt = t;
        int idx = 0;
        while (idx < N && tt > dtp) {
          tt -= dtp;
          idx++;
        }
        if (idx == N) {
          idx--;
          tt += dtp;
        }
        const double *c = cb + (size_t)(p0 + idx) * 12; // c[2k + d]
        // Piece::getPos: columns from the constant term upwards, tn = 1, t, t*t, ...
        double px = 0.0, py = 0.0, tn = 1.0;
        for (int k = 0; k <= 5; k++) {
          px += tn * c[2 * k];
          py += tn * c[2 * k + 1];
          tn *= tt;
        }
        // Piece::getdSigma: n * tn * column, n = 1..5
        double vx = 0.0, vy = 0.0;
        tn = 1.0;
        for (int k = 1; k <= 5; k++) {
          vx += (double)k * tn * c[2 * k];
          vy += (double)k * tn * c[2 * k + 1];
          tn *= tt;
        }
        const double sg = (double)singuls[i];
        const double yaw = step_trig::Trig{order}.atan2(sg * vy, sg * vx);
        if (pose_collides(g, px, py, yaw, veh_width, veh_length, veh_dcr, vertex_res, order)) {
          hit = 1;
          first = counter;
          break;
        }
      }
      p0 += N;
    }
    collision[b] = hit;
    first_sample[b] = first;
  }
}
