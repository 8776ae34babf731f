// Moving-obstacle trajectory fitting on the device: predicted state sequences of the
// surrounding vehicles become the piecewise quintics the solve path's moving-obstacle penalty reads.
//
//   TrajPlanner::ConverSurroundTrajFromPoints   traj_planner/src/traj_manager.cpp:743-789
//   TrajPlanner::state_to_flat_output           traj_planner/src/traj_manager.cpp:139-158
//   MinJerkOpt::reset / generate / getTraj      plan_utils/poly_traj_utils.hpp:895-997
//
// One workgroup per obstacle.  The fit is one MINCO "generate": the right-hand side (boundary states and
// the predicted positions as waypoints) times the dense operator A_N^{-1}|columns (the same operator the
// solve path uses for its own segments, built on the host with the reference's banded LU), scaled by
// t^-k.  The result is written in the layout dftpav_set_surround takes, so the fitted obstacles are
// installed without leaving the device.  fp64, no contraction, portable cos / sin: bit-identical to
// oracle/fit_oracle.cpp in order 1.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "traj_math.h"

namespace dftpav {

struct FitArgs {
  const double *states; // [S][n_states][7]: x, y, angle, velocity, acceleration, curvature, time_stamp
  int S, n_states;
  const double *opM; // [6N][N+5], N = n_states - 1
  double *dur;       // [S][N]
  double *coef;      // [S][N][12], per piece [x5,y5, x4,y4, ... x0,y0]
  double *total;     // [S]
  double *start;     // [S]
};

// state_to_flat_output, traj_manager.cpp:139-158: columns p, v, a of the 2x3 flat output
__device__ inline void flat_output(const double *st, double out[6]) {
  double vel = st[3];
  const double angle = st[2], acc = st[4], cur = st[5];
  double c, s;
  crt::sincos(angle, s, c); // (the reference: libm; here correctly rounded, as oracle order 2)
  const double ns = -s;
  if (vel == 0.0) vel = 1e-5;
  out[0] = st[0];
  out[1] = st[1];
  out[2] = c * vel + ns * 0.0; // init_R * (vel, 0)
  out[3] = s * vel + c * 0.0;
  const double lat = cur * (vel * vel); // curvature * pow(vel, 2)
  out[4] = c * acc + ns * lat;          // init_R * (acc, curvature * vel^2)
  out[5] = s * acc + c * lat;
}

__global__ void __launch_bounds__(256) fit_kernel(FitArgs A) {
  extern __shared__ double lds[]; // rhs [N+5][2]
  const int o = blockIdx.x, tid = threadIdx.x;
  const int n = A.n_states, N = n - 1, nc = N + 5;
  const double *st = A.states + (size_t)o * n * 7;
  double *rhs = lds;
  __shared__ double s_dT;
  if (tid == 0) {
    // piece_dur_vec.sum() / pieceNum (traj_manager.cpp:773), durations summed in order
    double sum = 0.0;
    for (int i = 1; i < n; i++) sum += st[7 * i + 6] - st[7 * (i - 1) + 6];
    const double dT = sum / N;
    s_dT = dT;
    double head[6], tail[6];
    flat_output(st, head);
    flat_output(st + (size_t)(n - 1) * 7, tail);
    // rows that can be non-zero, in the operator's column order (poly_traj_utils.hpp:968-977)
    rhs[0] = head[0]; rhs[1] = head[1];
    rhs[2] = head[2] * dT; rhs[3] = head[3] * dT;
    rhs[4] = head[4] * (dT * dT); rhs[5] = head[5] * (dT * dT);
    rhs[2 * (N + 2)] = tail[0]; rhs[2 * (N + 2) + 1] = tail[1];
    rhs[2 * (N + 3)] = tail[2] * dT; rhs[2 * (N + 3) + 1] = tail[3] * dT;
    rhs[2 * (N + 4)] = tail[4] * (dT * dT); rhs[2 * (N + 4) + 1] = tail[5] * (dT * dT);
    double tot = 0.0; // Trajectory::getTotalDuration
    for (int p = 0; p < N; p++) tot += dT;
    A.total[o] = tot;
    A.start[o] = st[6];
  }
  for (int i = 1 + tid; i < n - 1; i += blockDim.x) { // inner waypoints, traj_manager.cpp:764-767
    rhs[2 * (2 + i)] = st[7 * i];
    rhs[2 * (2 + i) + 1] = st[7 * i + 1];
  }
  __syncthreads();
  const double dT = s_dT;
  double t[12];
  duration_powers(dT, t); // t[k] = dT^k, t[6 + k] = 1 / dT^k (poly_traj_utils.hpp:961-966)
  for (int w = tid; w < 12 * N; w += blockDim.x) {
    const int p = w / 12, q = w - 12 * p, k = q >> 1, d = q & 1;
    const double *Mrow = A.opM + (size_t)(6 * p + k) * nc;
    double acc = 0.0;
    for (int col = 0; col < nc; col++) acc += Mrow[col] * rhs[2 * col + d]; // b = A^-1 rhs, columns ascending
    const double c = acc * t[6 + k];                                        // c = b * t^-k (poly_traj_utils.hpp:981-984)
    A.coef[((size_t)o * N + p) * 12 + 2 * (5 - k) + d] = c;                 // getTraj: columns reversed (t^5 first)
  }
  for (int p = tid; p < N; p += blockDim.x) A.dur[(size_t)o * N + p] = dT;
}

hipError_t launch_fit(const double *states, int S, int n_states, const double *opM, double *dur, double *coef, double *total,
                      double *start, hipStream_t stream) {
  FitArgs A{states, S, n_states, opM, dur, coef, total, start};
  const size_t lds = sizeof(double) * 2 * (size_t)(n_states - 1 + 5);
  hipLaunchKernelGGL(fit_kernel, dim3(S), dim3(256), lds, stream, A);
  return hipGetLastError();
}

} // namespace dftpav
