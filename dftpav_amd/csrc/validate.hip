// Collision re-check of optimised trajectories on the device: the step after the solve.
//
//   TrajPlannerServer::CheckReplan, collision part   traj_planner/src/traj_server_ros.cpp:385-397
//   Trajectory::getPos / getAngle / locatePieceIdx    plan_utils/poly_traj_utils.hpp:510-528, 77-87, 179-192, 237-244
//   SemanticMapManager::CheckCollisionUsingPosAndYaw  semantic_map_manager.cc:639-662
//   ShapeUtils::GetDenseVerticesOfOrientedBoundingBox common/src/common/basics/shapes.cc:110-149
//
// One workgroup per trajectory, one thread per time sample (t = 0, dt, dt + dt, ... < duration of each
// segment).  The samples are independent; a thread evaluates the pose, walks the ~140 outline points of
// the vehicle through the occupancy grid and, on a hit, lowers the trajectory's first-collision index with
// an atomic minimum — the reference stops at the first colliding sample, the minimum is the same answer.
// Both running sums of the reference (the sample times and the spacing along an edge) are tabulated on the
// host so that a thread sees exactly the value the sequential loop would have reached.  fp64, no
// contraction, portable cos / sin / atan2: bit-identical to oracle/validate_oracle.cpp in order 1.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "traj_math.h"

namespace dftpav {

struct ValidateArgs {
  const unsigned char *cells;
  int size_x, size_y;
  double resolution, origin_x, origin_y;
  const double *coeffs;   // [B][Ntot][6][2]
  const double *piece_dt; // [B][M]
  DevLayout L;
  int B;
  double veh_width, veh_length, veh_dcr;
  const double *t_tab; // 0, dt, dt + dt, ...
  int n_t;
  double sample_dt;
  const double *v_tab; // res, res + res, ...
  int n_v;
  int *collision, *first_sample; // [B]
};

__device__ inline bool v_occupied(const ValidateArgs &A, double x, double y) {
  const double cx = round((x - A.origin_x) / A.resolution), cy = round((y - A.origin_y) / A.resolution);
  if (!(cx >= 0.0 && cx < (double)A.size_x && cy >= 0.0 && cy < (double)A.size_y)) return false;
  return A.cells[(int)cx + A.size_x * (int)cy] == 80;
}
__device__ inline bool v_edge_hits(const ValidateArgs &A, double ax, double ay, double bx, double by) {
  const double dx = bx - ax, dy = by - ay;
  const double norm = sqrt(dx * dx + dy * dy);
  for (int j = 0; j < A.n_v; j++) {
    const double dl = A.v_tab[j];
    if (!(dl < norm)) break;
    const double f = dl / norm;
    if (v_occupied(A, f * dx + ax, f * dy + ay)) return true;
  }
  return false;
}

__global__ void __launch_bounds__(256) validate_kernel(ValidateArgs A) {
  __shared__ int s_count[kMaxSeg + 1]; // samples of the segments before segment i
  __shared__ double s_dur[kMaxSeg];
  __shared__ int s_first;
  const int b = blockIdx.x, tid = threadIdx.x;
  const DevLayout &L = A.L;
  const int M = L.M;
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < M; i++) {
      const double dtp = A.piece_dt[(size_t)b * M + i];
      double dur = 0.0; // Trajectory::getTotalDuration: piece durations summed in order
      for (int p = 0; p < L.piece_nums[i]; p++) dur += dtp;
      s_dur[i] = dur;
      // number of samples t_k < dur: the table is increasing; past its end the running sum is continued
      int lo = 0, hi = A.n_t;
      while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (A.t_tab[mid] < dur) lo = mid + 1;
        else hi = mid;
      }
      int cnt = lo;
      if (cnt == A.n_t) {
        for (double t = A.t_tab[A.n_t - 1] + A.sample_dt; t < dur; t += A.sample_dt) cnt++;
      }
      s_count[i] = acc;
      acc += cnt;
    }
    s_count[M] = acc;
    s_first = 0x7fffffff;
  }
  __syncthreads();
  const int total = s_count[M];
  const double *cb = A.coeffs + (size_t)b * L.Ntot * 12;
  for (int q = tid; q < total; q += blockDim.x) {
    int i = 0;
    while (i + 1 < M && q >= s_count[i + 1]) i++;
    const int k = q - s_count[i];
    double t;
    if (k < A.n_t) {
      t = A.t_tab[k];
    } else {
      t = A.t_tab[A.n_t - 1];
      for (int j = A.n_t - 1; j < k; j++) t += A.sample_dt;
    }
    const int N = L.piece_nums[i];
    const double dtp = A.piece_dt[(size_t)b * M + i];
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
    const double *c = cb + (size_t)(L.seg_piece0[i] + idx) * 12;
    double px = 0.0, py = 0.0, tn = 1.0;
#pragma unroll
    for (int kk = 0; kk <= 5; kk++) { // Piece::getPos
      px += tn * c[2 * kk];
      py += tn * c[2 * kk + 1];
      tn *= tt;
    }
    double vx = 0.0, vy = 0.0;
    tn = 1.0;
#pragma unroll
    for (int kk = 1; kk <= 5; kk++) { // Piece::getdSigma
      vx += (double)kk * tn * c[2 * kk];
      vy += (double)kk * tn * c[2 * kk + 1];
      tn *= tt;
    }
    const double sg = (double)L.singuls[i];
    const double yaw = crt::atan2(sg * vy, sg * vx); // (the reference: libm; here correctly rounded, as oracle order 2)
    // CheckCollisionUsingPosAndYaw, semantic_map_manager.cc:639-662 + shapes.cc:116-147
    double cs, sn;
    crt::sincos(yaw, sn, cs);
    const double W = A.veh_width, Lv = A.veh_length;
    const double x = px + A.veh_dcr * cs, y = py + A.veh_dcr * sn;
    const double c1x = x + 0.5 * Lv * cs + 0.5 * W * sn, c1y = y + 0.5 * Lv * sn - 0.5 * W * cs;
    const double c2x = x + 0.5 * Lv * cs - 0.5 * W * sn, c2y = y + 0.5 * Lv * sn + 0.5 * W * cs;
    const double c3x = x - 0.5 * Lv * cs - 0.5 * W * sn, c3y = y - 0.5 * Lv * sn + 0.5 * W * cs;
    const double c4x = x - 0.5 * Lv * cs + 0.5 * W * sn, c4y = y - 0.5 * Lv * sn - 0.5 * W * cs;
    const bool hit = v_edge_hits(A, c1x, c1y, c2x, c2y) || v_edge_hits(A, c2x, c2y, c3x, c3y) ||
                     v_edge_hits(A, c3x, c3y, c4x, c4y) || v_edge_hits(A, c4x, c4y, c1x, c1y) || v_occupied(A, c1x, c1y) ||
                     v_occupied(A, c2x, c2y) || v_occupied(A, c3x, c3y) || v_occupied(A, c4x, c4y);
    if (hit) atomicMin(&s_first, q);
  }
  __syncthreads();
  if (tid == 0) {
    const bool any = s_first != 0x7fffffff;
    A.collision[b] = any ? 1 : 0;
    A.first_sample[b] = any ? s_first : -1;
  }
}

hipError_t launch_validate(const unsigned char *cells, int size_x, int size_y, double resolution, double origin_x, double origin_y,
                           const double *coeffs, const double *piece_dt, const DevLayout &L, int B, double veh_width,
                           double veh_length, double veh_dcr, const double *t_tab, int n_t, double sample_dt, const double *v_tab,
                           int n_v, int *collision, int *first_sample, hipStream_t stream) {
  ValidateArgs A{cells, size_x, size_y, resolution, origin_x, origin_y, coeffs, piece_dt, L, B, veh_width, veh_length, veh_dcr,
                 t_tab, n_t, sample_dt, v_tab, n_v, collision, first_sample};
  hipLaunchKernelGGL(validate_kernel, dim3(B), dim3(256), 0, stream, A);
  return hipGetLastError();
}

} // namespace dftpav
