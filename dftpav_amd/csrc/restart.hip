// Seeded restart sampler on the device: the batch axis of the solve path.
//
// The reference plans one trajectory per cycle; this drop-in solves B at a time, and the extra ones are
// restarts of a hypothesis: its inner waypoints perturbed by N(0, sigma^2) per coordinate and each segment
// duration scaled by U[lo, hi] (SURVEY §8(d) "Restarts"; restart 0 of a hypothesis is the unperturbed one).
// There is no reference code for this step — the definition is ours and it is stated here in full so that the
// oracle (oracle/restart_oracle.cpp) and anybody else can reproduce a batch from (seed, hypothesis, restart):
//
//   stream   s0 = SplitMix64 output of (seed, counter 1 + hypothesis * 65536 + restart): streams of different
//            (hypothesis, restart) are unrelated, not shifted copies of each other
//   u64 #k   SplitMix64: z = s0 + k * 0x9E3779B97F4A7C15; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9;
//                        z ^= z >> 27; z *= 0x94D049BB133111EB; z ^= z >> 31
//   u01(k)   ((u64 #k >> 11) + 0.5) * 2^-53                                                   in (0, 1)
//   normal   Box-Muller on the pair (u01(2j), u01(2j+1)): sqrt(-2 p_log(u1)) * p_cos(2 pi u2) for the x of
//            waypoint j, ... * p_sin(2 pi u2) for its y          (portable log / cos / sin of traj_math.h)
//   duration factor of segment i: lo + (hi - lo) * u01(2 * n_waypoints + i)
//
// One thread per (trajectory, waypoint or segment); fp64, no contraction: bit-identical to the oracle.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "traj_math.h"

namespace dftpav {

struct RestartArgs {
  const double *inner; // [n_hyp][n_inner] inner waypoints of the hypotheses (x0, y0, x1, y1, ...)
  const double *durs;  // [n_hyp][M] segment durations
  int n_hyp, n_restarts, n_inner, M;
  double sigma, lo, hi;
  unsigned long long seed;
  double *out_inner; // [n_hyp * n_restarts][n_inner], trajectory b = hypothesis * n_restarts + restart
  double *out_durs;  // [n_hyp * n_restarts][M]
};

__host__ __device__ inline unsigned long long splitmix64(unsigned long long s0, unsigned long long k) {
  unsigned long long z = s0 + k * 0x9E3779B97F4A7C15ull;
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
__host__ __device__ inline double u01(unsigned long long s0, unsigned long long k) {
  return ((double)(splitmix64(s0, k) >> 11) + 0.5) * 1.1102230246251565e-16; // 2^-53
}

__global__ void __launch_bounds__(256) restart_kernel(RestartArgs A) {
  const int nw = A.n_inner / 2;      // waypoints
  const int per = nw + A.M;          // work items per trajectory
  const long long total = (long long)A.n_hyp * A.n_restarts * per;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(w / per), q = (int)(w - (long long)b * per);
    const int hyp = b / A.n_restarts, r = b - hyp * A.n_restarts;
    const unsigned long long s0 = splitmix64(A.seed, 1ull + (unsigned long long)hyp * 65536ull + (unsigned long long)r);
    if (q < nw) {
      const double bx = A.inner[(size_t)hyp * A.n_inner + 2 * q], by = A.inner[(size_t)hyp * A.n_inner + 2 * q + 1];
      double dx = 0.0, dy = 0.0;
      if (r > 0) {
        const double u1 = u01(s0, 2ull * q), u2 = u01(s0, 2ull * q + 1);
        const double rad = sqrt(-2.0 * p_log(u1)), ang = 6.283185307179586476925 * u2;
        dx = A.sigma * (rad * p_cos(ang));
        dy = A.sigma * (rad * p_sin(ang));
      }
      A.out_inner[(size_t)b * A.n_inner + 2 * q] = bx + dx;
      A.out_inner[(size_t)b * A.n_inner + 2 * q + 1] = by + dy;
    } else {
      const int i = q - nw;
      double f = 1.0;
      if (r > 0) f = A.lo + (A.hi - A.lo) * u01(s0, 2ull * nw + i);
      A.out_durs[(size_t)b * A.M + i] = A.durs[(size_t)hyp * A.M + i] * f;
    }
  }
}

hipError_t launch_restarts(const double *inner, const double *durs, int n_hyp, int n_restarts, int n_inner, int M, double sigma,
                           double lo, double hi, unsigned long long seed, double *out_inner, double *out_durs, hipStream_t stream) {
  RestartArgs A{inner, durs, n_hyp, n_restarts, n_inner, M, sigma, lo, hi, seed, out_inner, out_durs};
  const long long total = (long long)n_hyp * n_restarts * (n_inner / 2 + M);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 65535) blocks = 65535;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(restart_kernel, dim3(blocks), dim3(256), 0, stream, A);
  return hipGetLastError();
}

} // namespace dftpav
