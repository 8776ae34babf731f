// Read-out of optimised trajectories on the device: Trajectory::GetState over a time
// grid for every trajectory of a solved batch, played back the way the server walks the gear segments.
//
//   Trajectory::GetState / locatePieceIdx / getTotalDuration   plan_utils/poly_traj_utils.hpp:378-406, 510-528, 425-434
//   Piece::getPos / getdSigma / getddSigma / getStateExpPos    plan_utils/poly_traj_utils.hpp:77-87, 179-211, 303-340
//   TrajContainer::addSingulTraj                               plan_utils/traj_container.hpp:58-73
//   TrajPlannerServer::PublishData / FilterSingularityState    traj_planner/src/traj_server_ros.cpp:244-259, 335-356
//
// One workgroup per trajectory, one thread per time sample: the samples of GetState are independent and a
// sample is 8 doubles, so the kernel is a pure HBM write stream (64 B per sample, the 12 coefficients of a
// piece come from L1/L2).  The server's singularity filter is the only sequential part: it holds the
// heading of a near-standstill sample at the previously *published* heading.  A sample moving faster than
// 0.1 m/s is never touched, so the chain only runs inside a run of consecutive slow samples; the first
// thread of each run replays its run in order, all runs in parallel.  fp64, no contraction, portable
// atan2 / atan: bit-identical to oracle/states_oracle.cpp in order 1.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "traj_math.h"

namespace dftpav {

struct StatesArgs {
  const double *coeffs;   // [B][Ntot][6][2]
  const double *piece_dt; // [B][M]
  DevLayout L;
  int B;
  double wheel_base, t0, sample_dt;
  int n_samples, filter;
  double *states; // [B][n_samples][8]
  int *n_valid;   // [B]
};

__device__ inline double s_normalize_angle(double theta) { // calculations.cc:18-23
  const double pi = 3.14159265358979323846;
  double tmp = theta;
  tmp -= (double)((theta >= pi) * 2) * pi;
  tmp += (double)((theta < -pi) * 2) * pi;
  return tmp;
}

__global__ void __launch_bounds__(256) states_kernel(StatesArgs A) {
  __shared__ double s_start[kMaxSeg], s_dur[kMaxSeg], s_end[kMaxSeg];
  __shared__ int s_valid;
  const int b = blockIdx.x, tid = threadIdx.x;
  const DevLayout &L = A.L;
  const int M = L.M;
  if (tid == 0) {
    double world = 0.0;
    for (int i = 0; i < M; i++) {
      const double dtp = A.piece_dt[(size_t)b * M + i];
      double d = 0.0;
      for (int p = 0; p < L.piece_nums[i]; p++) d += dtp;
      s_start[i] = world;
      s_dur[i] = d;
      s_end[i] = world + d;
      world = s_end[i];
    }
    s_valid = 0;
  }
  __syncthreads();
  const double *cb = A.coeffs + (size_t)b * L.Ntot * 12;
  double *out = A.states + (size_t)b * A.n_samples * 8;
  for (int k = tid; k < A.n_samples; k += blockDim.x) {
    double *s = out + 8 * (size_t)k;
    const double t = A.t0 + (double)k * A.sample_dt;
    int i = 0;
    while (i < M && s_end[i] <= t) i++;
    if (i >= M) {
#pragma unroll
      for (int q = 0; q < 8; q++) s[q] = 0.0;
      continue;
    }
    atomicMax(&s_valid, k + 1);
    double inner = t - s_start[i];
    if (inner > s_dur[i]) inner = s_dur[i];
    const int N = L.piece_nums[i];
    const double dtp = A.piece_dt[(size_t)b * M + i];
    int idx = 0;
    while (idx < N && inner > dtp) {
      inner -= dtp;
      idx++;
    }
    if (idx == N) {
      idx--;
      inner += dtp;
    }
    const double *c = cb + (size_t)(L.seg_piece0[i] + idx) * 12;
    double px = 0.0, py = 0.0, tn = 1.0;
#pragma unroll
    for (int q = 0; q <= 5; q++) {
      px += tn * c[2 * q];
      py += tn * c[2 * q + 1];
      tn *= inner;
    }
    double vx = 0.0, vy = 0.0;
    tn = 1.0;
#pragma unroll
    for (int q = 1; q <= 5; q++) {
      vx += (double)q * tn * c[2 * q];
      vy += (double)q * tn * c[2 * q + 1];
      tn *= inner;
    }
    double ax = 0.0, ay = 0.0;
    tn = 1.0;
#pragma unroll
    for (int q = 2; q <= 5; q++) {
      ax += (double)((q - 1) * q) * tn * c[2 * q];
      ay += (double)((q - 1) * q) * tn * c[2 * q + 1];
      tn *= inner;
    }
    const double sg = (double)L.singuls[i];
    const double angle = crt::atan2(sg * vy, sg * vx); // (the reference: libm; here correctly rounded, as oracle order 2)
    const double vel = sg * sqrt(vx * vx + vy * vy);
    double curv = 0.0, acc = 0.0, steer = 0.0;
    if (!(fabs(vel) < 1e-6)) {
      curv = (vx * ay - vy * ax) / crt::cube_cr(vel); // (the reference: pow(vel, 3) of libm; here the correctly rounded cube)
      acc = (vx * ax + vy * ay) / vel;
      steer = crt::atan(A.wheel_base * curv);
    }
    s[0] = t; s[1] = px; s[2] = py; s[3] = angle; s[4] = curv; s[5] = vel; s[6] = acc; s[7] = steer;
  }
  __syncthreads(); // the raw samples of this trajectory are visible to the whole workgroup
  const int valid = s_valid;
  if (tid == 0) A.n_valid[b] = valid;
  if (!A.filter) return;
  // FilterSingularityState: a slow sample (|v| < kBigEPS) takes the previous published heading when its own
  // differs from it by more than the steering limit allows.  Run heads replay their runs.
  const double max_rate = 0x1.fffffffffffffp-1 / 2.85 * 0.1; // tan(M_PI / 4) as glibc returns it
  for (int k = tid; k < valid; k += blockDim.x) {
    if (k == 0) continue; // empty history: the first sample is published as it is
    const bool slow = fabs(out[8 * (size_t)k + 5]) < 0.1;
    const bool prev_slow = k > 1 && fabs(out[8 * (size_t)(k - 1) + 5]) < 0.1; // sample 0 is never rewritten: it can feed, not join
    if (!slow || prev_slow) continue;
    double hist_angle = out[8 * (size_t)(k - 1) + 3], hist_t = out[8 * (size_t)(k - 1)];
    for (int j = k; j < valid && fabs(out[8 * (size_t)j + 5]) < 0.1; j++) {
      const double t = out[8 * (size_t)j];
      double angle = out[8 * (size_t)j + 3];
      const double max_change = max_rate * (t - hist_t);
      if (fabs(s_normalize_angle(angle - hist_angle)) > max_change) {
        angle = hist_angle;
        out[8 * (size_t)j + 3] = angle;
      }
      hist_angle = angle;
      hist_t = t;
    }
  }
}

hipError_t launch_states(const double *coeffs, const double *piece_dt, const DevLayout &L, int B, double wheel_base, double t0,
                         double sample_dt, int n_samples, int filter, double *states, int *n_valid, hipStream_t stream) {
  StatesArgs A{coeffs, piece_dt, L, B, wheel_base, t0, sample_dt, n_samples, filter, states, n_valid};
  hipLaunchKernelGGL(states_kernel, dim3(B), dim3(256), 0, stream, A);
  return hipGetLastError();
}

} // namespace dftpav
