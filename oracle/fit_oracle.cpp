// TEST INFRASTRUCTURE ONLY (see oracle/dftpav_oracle.h): CPU restatement of the moving-obstacle trajectory
// fit, SURVEY.md §8(f)-4.
//
//   TrajPlanner::ConverSurroundTrajFromPoints   traj_planner/src/traj_manager.cpp:743-789
//   TrajPlanner::state_to_flat_output           traj_planner/src/traj_manager.cpp:139-158
//   MinJerkOpt::reset / generate / getTraj      plan_utils/poly_traj_utils.hpp:895-997
//   BandedSystem::factorizeLU / solve           plan_utils/poly_traj_utils.hpp:776-826
//
// Each predicted state sequence (x, y, angle, velocity, acceleration, curvature, time_stamp) is fitted with a
// uniform-time minimum-jerk trajectory of n_states - 1 pieces through its positions, with the flat outputs
// of the first and last state as boundary conditions.
//
// order 0: as the reference — libm cos / sin, the banded system solved for the actual right-hand side.
// order 1: what the HIP kernel evaluates — portable cos / sin, the right-hand side times the dense operator
// (A_N^{-1} applied to unit vectors with the same banded LU), columns in ascending order.
// PINNED (round 5): order 0 is bit-equal to the reference's own code -- the cited functions cut verbatim out of
// /root/reference (oracle/ref_slices.py) and compiled into oracle/_ref/libdftpav_ref_next.so (oracle/ref_next_driver.cpp) --
// on the scenarios the GPU tests of this step use (tests/test_ref_pin.py::test_fit_oracle_is_bit_equal_to_ConverSurroundTrajFromPoints).
#include <cmath>
#include <vector>

#include "../dftpav_amd/csrc/traj_math.h"
#include "step_trig.h"

extern "C" void oracle_fit_surround(const double *states, int S, int n_states, int order, double *dur, double *coef, double *total,
                                    double *start) {
  using namespace dftpav;
  const int N = n_states - 1, n6 = 6 * N, nc = N + 5;
  std::vector<double> band((size_t)n6 * 13, 0.0);
  BandedLU A{n6, 6, 6, band.data()};
  minco_fill(A, N);
  banded_factorize(A);
  std::vector<double> op;
  if (order) { // the dense operator, as dftpav_amd/csrc/capi.cpp builds it
    op.assign((size_t)n6 * nc, 0.0);
    std::vector<double> col(n6);
    for (int c = 0; c < nc; c++) {
      const int row = c < 3 ? c : (c < N + 2 ? 6 * (c - 3) + 5 : n6 - 3 + (c - (N + 2)));
      std::fill(col.begin(), col.end(), 0.0);
      col[row] = 1.0;
      banded_solve1(A, col.data());
      for (int r = 0; r < n6; r++) op[(size_t)r * nc + c] = col[r];
    }
  }
  auto flat = [&](const double *st, double out[6]) { // state_to_flat_output
    double vel = st[3];
    const double angle = st[2], acc = st[4], cur = st[5];
    const step_trig::Trig T{order};
    const double c = T.cos(angle), s = T.sin(angle), ns = -s;
    if (vel == 0.0) vel = 1e-5;
    out[0] = st[0];
    out[1] = st[1];
    out[2] = c * vel + ns * 0.0;
    out[3] = s * vel + c * 0.0;
    const double lat = cur * (vel * vel);
    out[4] = c * acc + ns * lat;
    out[5] = s * acc + c * lat;
  };
  for (int o = 0; o < S; o++) {
    const double *st = states + (size_t)o * n_states * 7;
    double sum = 0.0;
    for (int i = 1; i < n_states; i++) sum += st[7 * i + 6] - st[7 * (i - 1) + 6];
    const double dT = sum / N;
    double head[6], tail[6];
    flat(st, head);
    flat(st + (size_t)(n_states - 1) * 7, tail);
    double t[12];
    duration_powers(dT, t);
    // right-hand side rows, poly_traj_utils.hpp:968-977
    std::vector<double> bx(n6, 0.0), by(n6, 0.0);
    bx[0] = head[0]; by[0] = head[1];
    bx[1] = head[2] * dT; by[1] = head[3] * dT;
    bx[2] = head[4] * (dT * dT); by[2] = head[5] * (dT * dT);
    for (int i = 1; i < n_states - 1; i++) {
      bx[6 * (i - 1) + 5] = st[7 * i];
      by[6 * (i - 1) + 5] = st[7 * i + 1];
    }
    bx[n6 - 3] = tail[0]; by[n6 - 3] = tail[1];
    bx[n6 - 2] = tail[2] * dT; by[n6 - 2] = tail[3] * dT;
    bx[n6 - 1] = tail[4] * (dT * dT); by[n6 - 1] = tail[5] * (dT * dT);
    std::vector<double> sx(n6), sy(n6);
    if (order) {
      std::vector<double> rx(nc), ry(nc);
      for (int c = 0; c < nc; c++) {
        const int row = c < 3 ? c : (c < N + 2 ? 6 * (c - 3) + 5 : n6 - 3 + (c - (N + 2)));
        rx[c] = bx[row];
        ry[c] = by[row];
      }
      for (int r = 0; r < n6; r++) {
        double ax = 0.0, ay = 0.0;
        for (int c = 0; c < nc; c++) {
          ax += op[(size_t)r * nc + c] * rx[c];
          ay += op[(size_t)r * nc + c] * ry[c];
        }
        sx[r] = ax;
        sy[r] = ay;
      }
    } else {
      sx = bx;
      sy = by;
      banded_solve1(A, sx.data());
      banded_solve1(A, sy.data());
    }
    for (int p = 0; p < N; p++) {
      for (int k = 0; k < 6; k++) { // c = b * t^-k, stored with the t^5 column first (getTraj, :987-997)
        coef[((size_t)o * N + p) * 12 + 2 * (5 - k) + 0] = sx[6 * p + k] * t[6 + k];
        coef[((size_t)o * N + p) * 12 + 2 * (5 - k) + 1] = sy[6 * p + k] * t[6 + k];
      }
      dur[(size_t)o * N + p] = dT;
    }
    double tot = 0.0;
    for (int p = 0; p < N; p++) tot += dT;
    total[o] = tot;
    start[o] = st[6];
  }
}
