// TEST INFRASTRUCTURE ONLY (see oracle/dftpav_oracle.h): CPU restatement of the seeded restart sampler
// (dftpav_amd/csrc/restart.hip, SURVEY.md §8(d) "Restarts").  There is no reference code for this step; the
// definition is stated in restart.hip and written out again here, independently: SplitMix64 streams keyed by
// (seed, hypothesis, restart), Box-Muller normals with the portable log / cos / sin, uniform duration factors.
#include <cmath>
#include <cstdint>

#include "../dftpav_amd/csrc/traj_math.h"

static uint64_t mix(uint64_t s0, uint64_t k) {
  uint64_t z = s0 + k * UINT64_C(0x9E3779B97F4A7C15);
  z = (z ^ (z >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
  z = (z ^ (z >> 27)) * UINT64_C(0x94D049BB133111EB);
  return z ^ (z >> 31);
}
static double unit(uint64_t s0, uint64_t k) { return ((double)(mix(s0, k) >> 11) + 0.5) * std::ldexp(1.0, -53); }

extern "C" void oracle_sample_restarts(const double *inner, const double *durs, int n_hyp, int n_restarts, int n_inner, int M,
                                       double sigma, double lo, double hi, unsigned long long seed, double *out_inner,
                                       double *out_durs) {
  const int nw = n_inner / 2;
  for (int hyp = 0; hyp < n_hyp; hyp++)
    for (int r = 0; r < n_restarts; r++) {
      const size_t b = (size_t)hyp * n_restarts + r;
      const uint64_t s0 = mix((uint64_t)seed, 1 + (uint64_t)hyp * 65536 + (uint64_t)r); // the stream of (hyp, r)
      for (int j = 0; j < nw; j++) {
        double dx = 0.0, dy = 0.0;
        if (r > 0) {
          const double u1 = unit(s0, 2 * (uint64_t)j), u2 = unit(s0, 2 * (uint64_t)j + 1);
          const double rad = std::sqrt(-2.0 * dftpav::p_log(u1)), ang = 6.283185307179586476925 * u2;
          dx = sigma * (rad * dftpav::p_cos(ang));
          dy = sigma * (rad * dftpav::p_sin(ang));
        }
        out_inner[b * n_inner + 2 * j] = inner[(size_t)hyp * n_inner + 2 * j] + dx;
        out_inner[b * n_inner + 2 * j + 1] = inner[(size_t)hyp * n_inner + 2 * j + 1] + dy;
      }
      for (int i = 0; i < M; i++) {
        double f = 1.0;
        if (r > 0) f = lo + (hi - lo) * unit(s0, 2 * (uint64_t)nw + i);
        out_durs[b * M + i] = durs[(size_t)hyp * M + i] * f;
      }
    }
}
