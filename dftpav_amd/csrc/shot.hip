// Reeds-Shepp shots on the device (hypothesis generation): the analytic connection of a
// pose to the goal that the reference's front end tries from its search nodes.
//
//   KinoAstar::computeShotTraj / is_shot_sucess      traj_planner/src/kino_astar.cpp:304-345
//   ompl::base::ReedsSheppStateSpace(1 / max_cur_)   kino_astar.cpp:423 (OMPL is not vendored: rs_math.h restates
//     ::distance, ::interpolate                        the published algorithm behind it)
//   SemanticMapManager::CheckCollisionUsingPosAndYaw semantic_map_manager.cc:639-662 (through map_adapter.cpp:110-115)
//   ShapeUtils::GetDenseVerticesOfOrientedBoundingBox common/src/common/basics/shapes.cc:110-149
//
// One wave per (from, to) pair.  Every lane computes the shortest path (the 48 candidate words are straight-line
// code, nothing to share), lane 0 lays down the sample offsets l = 0, checkl, checkl + checkl, ... <= length as
// the reference's running sum, then the samples are independent: a lane interpolates its pose and, if a map is
// installed, walks the vehicle outline through the occupancy grid.  fp64, no contraction, portable
// sin / cos / atan2: bit-identical to oracle/shot_oracle.cpp in order 1.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "rs_math.h"

namespace dftpav {

struct ShotArgs {
  const double *from, *to; // [n][3]
  int n;
  double rho, checkl;
  int max_samples;
  // occupancy grid (cells == nullptr: no collision check)
  const unsigned char *cells;
  int size_x, size_y;
  double resolution, origin_x, origin_y;
  double veh_width, veh_length, veh_dcr;
  const double *v_tab; // res, res + res, ...: spacing of the outline points
  int n_v;
  double *length;  // [n]
  int *type;       // [n]
  double *seg;     // [n][5]
  double *samples; // [n][max_samples][3]
  int *n_samples;  // [n]
  int *collides;   // [n]
};

__device__ inline bool s_occupied(const ShotArgs &A, double x, double y) {
  const double cx = round((x - A.origin_x) / A.resolution), cy = round((y - A.origin_y) / A.resolution);
  if (!(cx >= 0.0 && cx < (double)A.size_x && cy >= 0.0 && cy < (double)A.size_y)) return false;
  return A.cells[(int)cx + A.size_x * (int)cy] == 80;
}
__device__ inline bool s_edge_hits(const ShotArgs &A, double ax, double ay, double bx, double by) {
  const double dx = bx - ax, dy = by - ay;
  const double norm = sqrt(dx * dx + dy * dy);
  for (int j = 0; j < A.n_v; j++) {
    const double dl = A.v_tab[j];
    if (!(dl < norm)) break;
    const double f = dl / norm;
    if (s_occupied(A, f * dx + ax, f * dy + ay)) return true;
  }
  return false;
}
__device__ inline bool s_pose_collides(const ShotArgs &A, double px, double py, double yaw) {
  const double cs = p_cos(yaw), sn = p_sin(yaw);
  const double W = A.veh_width, Lv = A.veh_length;
  const double x = px + A.veh_dcr * cs, y = py + A.veh_dcr * sn;
  const double c1x = x + 0.5 * Lv * cs + 0.5 * W * sn, c1y = y + 0.5 * Lv * sn - 0.5 * W * cs;
  const double c2x = x + 0.5 * Lv * cs - 0.5 * W * sn, c2y = y + 0.5 * Lv * sn + 0.5 * W * cs;
  const double c3x = x - 0.5 * Lv * cs - 0.5 * W * sn, c3y = y - 0.5 * Lv * sn + 0.5 * W * cs;
  const double c4x = x - 0.5 * Lv * cs + 0.5 * W * sn, c4y = y - 0.5 * Lv * sn - 0.5 * W * cs;
  return s_edge_hits(A, c1x, c1y, c2x, c2y) || s_edge_hits(A, c2x, c2y, c3x, c3y) || s_edge_hits(A, c3x, c3y, c4x, c4y) ||
         s_edge_hits(A, c4x, c4y, c1x, c1y) || s_occupied(A, c1x, c1y) || s_occupied(A, c2x, c2y) || s_occupied(A, c3x, c3y) ||
         s_occupied(A, c4x, c4y);
}

__global__ void __launch_bounds__(64) shot_kernel(ShotArgs A) {
  extern __shared__ double l_tab[]; // [max_samples]
  __shared__ int s_cnt, s_hit;
  const int i = blockIdx.x, lane = threadIdx.x;
  typedef rs::Solver<rs::PortableMath> RS;
  double from[3], to[3];
  for (int k = 0; k < 3; k++) {
    from[k] = A.from[3 * (size_t)i + k];
    to[k] = A.to[3 * (size_t)i + k];
  }
  const rs::Path path = RS::between(from, to, A.rho);
  const double len = A.rho * path.total; // ReedsSheppStateSpace::distance
  if (lane == 0) {
    A.length[i] = len;
    A.type[i] = path.type;
    for (int k = 0; k < 5; k++) A.seg[5 * (size_t)i + k] = path.len[k];
    int cnt = 0;
    for (double l = 0.0; l <= len; l += A.checkl) { // kino_astar.cpp:338
      if (cnt < A.max_samples) l_tab[cnt] = l;
      cnt++;
    }
    s_cnt = cnt;
    s_hit = 0;
  }
  __syncthreads();
  const int cnt = s_cnt, stored = cnt < A.max_samples ? cnt : A.max_samples;
  double *out = A.samples + (size_t)i * A.max_samples * 3;
  for (int k = lane; k < A.max_samples; k += 64) {
    double s[3] = {0.0, 0.0, 0.0};
    if (k < stored) {
      const double t = l_tab[k] / len;
      if (t >= 1.0) { // ReedsSheppStateSpace::interpolate: the end states are copied as they are
        s[0] = to[0]; s[1] = to[1]; s[2] = to[2];
      } else if (t <= 0.0) {
        s[0] = from[0]; s[1] = from[1]; s[2] = from[2];
      } else {
        RS::interpolate(from, path, A.rho, t, s);
      }
      if (A.cells != nullptr && s_pose_collides(A, s[0], s[1], s[2])) atomicOr(&s_hit, 1);
    }
    out[3 * k] = s[0];
    out[3 * k + 1] = s[1];
    out[3 * k + 2] = s[2];
  }
  __syncthreads();
  if (lane == 0) {
    A.n_samples[i] = cnt;
    if (A.collides) A.collides[i] = s_hit;
  }
}

hipError_t launch_shots(const double *from, const double *to, int n, double rho, double checkl, int max_samples,
                        const unsigned char *cells, int size_x, int size_y, double resolution, double origin_x, double origin_y,
                        double veh_width, double veh_length, double veh_dcr, const double *v_tab, int n_v, double *length, int *type,
                        double *seg, double *samples, int *n_samples, int *collides, hipStream_t stream) {
  ShotArgs A{from, to, n, rho, checkl, max_samples, cells, size_x, size_y, resolution, origin_x, origin_y, veh_width, veh_length,
             veh_dcr, v_tab, n_v, length, type, seg, samples, n_samples, collides};
  hipLaunchKernelGGL(shot_kernel, dim3(n), dim3(64), sizeof(double) * (size_t)max_samples, stream, A);
  return hipGetLastError();
}

} // namespace dftpav
