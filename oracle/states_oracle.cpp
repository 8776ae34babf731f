// TEST INFRASTRUCTURE ONLY (see oracle/dftpav_oracle.h): CPU restatement of the read-out of an optimised
// trajectory, the other half of SURVEY.md §8(f)-2 — Trajectory::GetState over a time grid, played back the
// way the server walks the gear segments.
//
//   Trajectory::GetState / locatePieceIdx / getTotalDuration   plan_utils/poly_traj_utils.hpp:378-406, 510-528, 425-434
//   Piece::getPos / getdSigma / getddSigma / getStateExpPos    plan_utils/poly_traj_utils.hpp:77-87, 179-211, 303-340
//   TrajContainer::addSingulTraj (start / end time of a segment) plan_utils/traj_container.hpp:58-73, traj_manager.cpp:617-624
//   TrajPlannerServer::PublishData (which segment is played)   traj_planner/src/traj_server_ros.cpp:244-259
//   TrajPlannerServer::FilterSingularityState                  traj_planner/src/traj_server_ros.cpp:335-356
//   normalize_angle, kPi, kBigEPS                              common/src/common/math/calculations.cc:18-23, basics.h:76
//
// Sample k carries the time stamp t_k = t0 + k * dt (time 0 = start of the first segment).  The segment
// played at t is the first one whose end_time is not <= t (the server moves on by one segment per tick when
// `end_time <= t`; ticks are far denser than segments); past the last segment nothing is published, which
// ends the valid prefix.  With `filter` the heading of a near-standstill sample is held at the previous
// published heading when it jumps, exactly as the server does with its history's last entry.
//
// order 0: libm atan2 / atan / pow / tan, as the reference.  order 1: the portable atan2 / atan of
// traj_math.h, v*v*v for pow(v, 3) and the constant tan(M_PI / 4) = 0x1.fffffffffffffp-1 (what glibc returns),
// which is what the HIP kernel evaluates; everything else is correctly rounded IEEE arithmetic in the
// reference's order, so order 1 is bit-identical to the GPU.  
// PINNED (round 5): order 0 is bit-equal to the reference's own code -- the cited functions cut verbatim out of
// /root/reference (oracle/ref_slices.py) and compiled into oracle/_ref/libdftpav_ref_next.so (oracle/ref_next_driver.cpp) --
// on the scenarios the GPU tests of this step use (tests/test_ref_pin.py::test_states_oracle_is_bit_equal_to_GetState_and_the_servers_playback).
#include <cmath>
#include <cstdint>

#include "../dftpav_amd/csrc/traj_math.h"
#include "step_trig.h"

namespace {

constexpr double kPi = 3.14159265358979323846; // acos(-1.0), basics.h
constexpr double kTanQuarterPi = 0x1.fffffffffffffp-1;

inline double normalize_angle(double theta) { // calculations.cc:18-23
  double tmp = theta;
  tmp -= (double)((theta >= kPi) * 2) * kPi;
  tmp += (double)((theta < -kPi) * 2) * kPi;
  return tmp;
}

} // namespace

// coeffs: [B][Ntot][6][2], entry [k][d] = coefficient of t^k; piece_dt: [B][M]; states: [B][n_samples][8] =
// (time_stamp, x, y, angle, curvature, velocity, acceleration, steer); rows past n_valid[b] are zero.
extern "C" void oracle_sample_states(const double *coeffs, const double *piece_dt, const int *piece_nums, const int *singuls,
                                     int M, int B, double wheel_base, double t0, double sample_dt, int n_samples, int filter,
                                     int order, double *states, int *n_valid) {
  int Ntot = 0;
  for (int i = 0; i < M; i++) Ntot += piece_nums[i];
  for (int b = 0; b < B; b++) {
    const double *cb = coeffs + (size_t)b * Ntot * 12;
    double *out = states + (size_t)b * n_samples * 8;
    // addSingulTraj: duration = getTotalDuration (piece durations summed in order), end = start + duration,
    // the next segment starts at the previous end (traj_manager.cpp:619-623)
    double start[64], dur[64], end[64];
    double world = 0.0;
    for (int i = 0; i < M; i++) {
      double d = 0.0;
      for (int p = 0; p < piece_nums[i]; p++) d += piece_dt[(size_t)b * M + i];
      start[i] = world;
      dur[i] = d;
      end[i] = world + d;
      world = end[i];
    }
    int valid = 0;
    bool have_hist = false;
    double hist_t = 0.0, hist_angle = 0.0;
    for (int k = 0; k < n_samples; k++) {
      double *s = out + 8 * k;
      for (int q = 0; q < 8; q++) s[q] = 0.0;
      const double t = t0 + (double)k * sample_dt;
      int i = 0;
      while (i < M && end[i] <= t) i++;
      if (i >= M) continue; // exe_traj_index_ > final_traj_index_: nothing published
      valid = k + 1;
      // GetState(t - start_time)
      double inner = t - start[i];
      if (inner > dur[i]) inner = dur[i];
      const int N = piece_nums[i];
      const double dtp = piece_dt[(size_t)b * M + i];
      int idx = 0;
      while (idx < N && inner > dtp) {
        inner -= dtp;
        idx++;
      }
      if (idx == N) {
        idx--;
        inner += dtp;
      }
      int p0 = 0;
      for (int j = 0; j < i; j++) p0 += piece_nums[j];
      const double *c = cb + (size_t)(p0 + idx) * 12; // c[2k + d]
      double px = 0.0, py = 0.0, tn = 1.0;
      for (int q = 0; q <= 5; q++) { // getPos
        px += tn * c[2 * q];
        py += tn * c[2 * q + 1];
        tn *= inner;
      }
      double vx = 0.0, vy = 0.0;
      tn = 1.0;
      for (int q = 1; q <= 5; q++) { // getdSigma: n * tn * column
        vx += (double)q * tn * c[2 * q];
        vy += (double)q * tn * c[2 * q + 1];
        tn *= inner;
      }
      double ax = 0.0, ay = 0.0;
      tn = 1.0;
      for (int q = 2; q <= 5; q++) { // getddSigma: m * n * tn * column, m = q - 1, n = q
        ax += (double)((q - 1) * q) * tn * c[2 * q];
        ay += (double)((q - 1) * q) * tn * c[2 * q + 1];
        tn *= inner;
      }
      // getStateExpPos, poly_traj_utils.hpp:303-340
      const double sg = (double)singuls[i];
      double angle = step_trig::Trig{order}.atan2(sg * vy, sg * vx);
      const double vel = sg * std::sqrt(vx * vx + vy * vy);
      double curv = 0.0, acc = 0.0, steer = 0.0;
      if (!(std::fabs(vel) < 1e-6)) {
        const double v3 = step_trig::Trig{order}.cube(vel);
        curv = (vx * ay - vy * ax) / v3;
        acc = (vx * ax + vy * ay) / vel;
        steer = step_trig::Trig{order}.atan(wheel_base * curv);
      }
      if (filter && have_hist) { // FilterSingularityState, traj_server_ros.cpp:335-356
        const double duration = t - hist_t;
        const double max_rate = (order ? kTanQuarterPi : std::tan(kPi / 4.0)) / 2.85 * 0.1;
        const double max_change = max_rate * duration;
        if (std::fabs(vel) < 0.1 && std::fabs(normalize_angle(angle - hist_angle)) > max_change) angle = hist_angle;
      }
      have_hist = true;
      hist_t = t;
      hist_angle = angle;
      s[0] = t; s[1] = px; s[2] = py; s[3] = angle; s[4] = curv; s[5] = vel; s[6] = acc; s[7] = steer;
    }
    n_valid[b] = valid;
  }
}
