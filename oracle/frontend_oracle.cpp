// TEST INFRASTRUCTURE ONLY (see oracle/dftpav_oracle.h): CPU restatement of the step that turns a searched
// path into the solver's initial guess, SURVEY.md §8(f)-3 ("front-end resampling").
//
//   KinoAstar::getKinoNode, from SampleTraj on    traj_planner/src/kino_astar.cpp:606-743 (gear segmentation,
//                                                 trapezoid time allocation, 0.1 s samples, flat boundary states)
//   KinoAstar::evaluateDuration / evaluateLength  traj_planner/src/kino_astar.cpp:744-795
//   KinoAstar::evaluatePos                        traj_planner/src/kino_astar.cpp:468-521
//   KinoAstar::getFlatState                       traj_planner/src/kino_astar.cpp:834-857
//   TrajPlanner::RunMINCOParking, resampling      traj_planner/src/traj_manager.cpp:531-568
//
// Input per hypothesis: the sampled path (x, y, yaw) the search produced (its construction from the A* nodes and
// the Reeds-Shepp shot, kino_astar.cpp:566-605, needs OMPL and is not restated), the start / end states
// (x, y, yaw, v) and the start control (steer, acceleration).  Output per gear segment: direction, flat boundary
// states, number of pieces and piece duration, the inner waypoints and the constraint-point poses — the
// arguments of getRectangleConst and OptimizeTrajectory.
//
// order 0: libm cos / sin / tan, as the reference.  order 1: the portable cos / sin of traj_math.h (tan = sin / cos),
// which is what the HIP kernel evaluates; all else is correctly rounded IEEE arithmetic in the reference's order.
// PINNED (round 5): order 0 is bit-equal to the reference's own code -- the cited functions cut verbatim out of
// /root/reference (oracle/ref_slices.py) and compiled into oracle/_ref/libdftpav_ref_next.so (oracle/ref_next_driver.cpp) --
// on the scenarios the GPU tests of this step use (tests/test_ref_pin.py::test_frontend_oracle_is_bit_equal_to_getKinoNode_and_RunMINCOParking).
// (From SampleTraj on: the A* nodes and the OMPL shot that build SampleTraj are outside what can be compiled here.)
#include <algorithm>
#include <cmath>
#include <vector>

#include "../dftpav_amd/csrc/traj_math.h"
#include "../include/dftpav_hip.h"
#include "step_trig.h"

namespace {

struct Trig {
  int order;
  double c(double a) const { return step_trig::Trig{order}.cos(a); }
  double s(double a) const { return step_trig::Trig{order}.sin(a); }
  double t(double a) const { return step_trig::Trig{order}.tan(a); }
};

// kino_astar.cpp:744-762
double evaluate_duration(double length, double max_vel, double max_acc, double startV, double endV) {
  const double startv2 = startV * startV, endv2 = endV * endV, maxv2 = max_vel * max_vel; // pow(x, 2)
  const double critical_len = (maxv2 - startv2) / (2 * max_acc) + (maxv2 - endv2) / (2 * max_acc);
  if (length >= critical_len) return (max_vel - startV) / max_acc + (max_vel - endV) / max_acc + (length - critical_len) / max_vel;
  const double tmpv = std::sqrt(0.5 * (startv2 + endv2 + 2 * max_acc * length));
  return (tmpv - startV) / max_acc + (tmpv - endV) / max_acc;
}
// kino_astar.cpp:763-795
double evaluate_length(double curt, double locallength, double localtime, double max_vel, double max_acc, double startV, double endV) {
  (void)localtime;
  const double startv2 = startV * startV, endv2 = endV * endV, maxv2 = max_vel * max_vel;
  const double critical_len = (maxv2 - startv2) / (2 * max_acc) + (maxv2 - endv2) / (2 * max_acc);
  if (locallength >= critical_len) {
    const double t1 = (max_vel - startV) / max_acc;
    const double t2 = t1 + (locallength - critical_len) / max_vel;
    if (curt <= t1) return startV * curt + 0.5 * max_acc * (curt * curt);
    if (curt <= t2) return startV * t1 + 0.5 * max_acc * (t1 * t1) + (curt - t1) * max_vel;
    return startV * t1 + 0.5 * max_acc * (t1 * t1) + (t2 - t1) * max_vel + max_vel * (curt - t2) -
           0.5 * max_acc * ((curt - t2) * (curt - t2));
  }
  const double tmpv = std::sqrt(0.5 * (startv2 + endv2 + 2 * max_acc * locallength));
  const double tmpt = (tmpv - startV) / max_acc;
  if (curt <= tmpt) return startV * curt + 0.5 * max_acc * (curt * curt);
  return startV * tmpt + 0.5 * max_acc * (tmpt * tmpt) + tmpv * (curt - tmpt) - 0.5 * max_acc * ((curt - tmpt) * (curt - tmpt));
}

struct Shot {
  std::vector<int> index;   // shotindex
  std::vector<int> S;       // shot_SList
  std::vector<double> len;  // shot_lengthList
  std::vector<double> time; // shot_timeList
  double total = 0.0;       // totalTrajTime
};

inline double norm2(const double *a, const double *b) {
  const double dx = b[0] - a[0], dy = b[1] - a[1];
  return std::sqrt(dx * dx + dy * dy);
}

// the interpolation shared by getKinoNode and evaluatePos (kino_astar.cpp:503-516, 689-703)
inline void interpolate(const double *a, const double *b, double l1, double l, double out[3]) {
  const double l2 = l - l1;
  for (int d = 0; d < 3; d++) out[d] = l1 / l * a[d] + l2 / l * b[d];
  if (std::fabs(b[2] - a[2]) >= M_PI) {
    if (b[2] <= 0) out[2] = l1 / l * a[2] + l2 / l * (b[2] + 2 * M_PI);
    else if (a[2] <= 0) out[2] = l1 / l * (a[2] + 2 * M_PI) + l2 / l * b[2];
    // (neither: the reference reads an uninitialised value; the unwrapped interpolation is kept)
  }
}

// kino_astar.cpp:468-521
void evaluate_pos(const dftpav_frontend_params &fp, const double *P, const Shot &sh, double startvel, double endvel, double t,
                  double out[3]) {
  t = std::min<double>(std::max<double>(0, t), sh.total);
  int index = -1;
  double tmpT = 0, CutTime = 0;
  for (size_t i = 0; i < sh.time.size(); i++) {
    tmpT += sh.time[i];
    if (tmpT >= t) {
      index = (int)i;
      CutTime = t - tmpT + sh.time[i];
      break;
    }
  }
  double initv = fp.non_siguav, finv = fp.non_siguav;
  if (index == 0) initv = startvel;
  if (index == (int)sh.len.size() - 1) finv = endvel;
  const double localtime = sh.time[index], locallength = sh.len[index];
  const int front = sh.index[index], back = sh.index[index + 1];
  const double arclength = sh.S[index] > 0
                               ? evaluate_length(CutTime, locallength, localtime, fp.max_forward_vel, fp.max_forward_acc, initv, finv)
                               : evaluate_length(CutTime, locallength, localtime, fp.max_backward_vel, fp.max_backward_acc, initv, finv);
  double tmparc = 0;
  for (int i = front; i < back; i++) {
    tmparc += norm2(P + 3 * i, P + 3 * (i + 1));
    if (tmparc >= arclength) {
      const double l1 = tmparc - arclength;
      const double l = norm2(P + 3 * i, P + 3 * (i + 1));
      interpolate(P + 3 * i, P + 3 * (i + 1), l1, l, out);
      return;
    }
  }
  for (int d = 0; d < 3; d++) out[d] = P[3 * back + d];
}

// kino_astar.cpp:834-857, column-major 2x3
void flat_state(const dftpav_frontend_params &fp, const Trig &T, const double pose[3], double v, const double ctrl[2], int singul,
                double out[6]) {
  const double angle = pose[2];
  double vel = v;
  const double c = T.c(angle), s = T.s(angle), ns = -s;
  if (std::fabs(vel) <= fp.non_siguav) vel = singul * fp.non_siguav; // abs(vel) <= non_siguav
  else vel = singul * vel;
  out[0] = pose[0];
  out[1] = pose[1];
  out[2] = c * vel + ns * 0.0;
  out[3] = s * vel + c * 0.0;
  const double lat = T.t(ctrl[0]) / fp.wheel_base * (vel * vel);
  out[4] = c * ctrl[1] + ns * lat;
  out[5] = s * ctrl[1] + c * lat;
}

} // namespace

extern "C" void oracle_frontend_resample(const dftpav_frontend_params *fpp, const double *paths, const int *path_len, int max_path,
                                         const double *start_states, const double *end_states, const double *start_ctrl, int n_hyp,
                                         int order, const dftpav_frontend_out *out) {
  const dftpav_frontend_params &fp = *fpp;
  const Trig T{order};
  const int MS = out->max_seg, MP = out->max_pieces, MST = out->max_states;
  for (int h = 0; h < n_hyp; h++) {
    const double *P = paths + (size_t)h * max_path * 3;
    const int n = path_len[h];
    const double startvel = std::fabs(start_states[4 * h + 3]), endvel = std::fabs(end_states[4 * h + 3]);
    // ---- gear segmentation and time allocation, kino_astar.cpp:618-665
    Shot sh;
    double tmpl = 0;
    auto dir = [&](int i) {
      const double dx = P[3 * (i + 1)] - P[3 * i], dy = P[3 * (i + 1) + 1] - P[3 * i + 1];
      return dx * T.c(P[3 * i + 2]) + dy * T.s(P[3 * i + 2]) >= 0 ? 1 : -1;
    };
    auto dur = [&](double len, int S, double v0, double v1) {
      return S > 0 ? evaluate_duration(len, fp.max_forward_vel, fp.max_forward_acc, v0, v1)
                   : evaluate_duration(len, fp.max_backward_vel, fp.max_backward_acc, v0, v1);
    };
    int lastS = dir(0);
    sh.index.push_back(0);
    for (int i = 0; i < n - 1; i++) {
      const int curS = dir(i);
      if (curS * lastS >= 0) {
        tmpl += norm2(P + 3 * i, P + 3 * (i + 1));
      } else {
        sh.index.push_back(i);
        sh.S.push_back(lastS);
        sh.len.push_back(tmpl);
        sh.time.push_back(dur(tmpl, lastS, fp.non_siguav, fp.non_siguav));
        tmpl = norm2(P + 3 * i, P + 3 * (i + 1));
      }
      lastS = curS;
    }
    sh.S.push_back(lastS);
    sh.len.push_back(tmpl);
    sh.time.push_back(dur(tmpl, lastS, fp.non_siguav, fp.non_siguav));
    sh.index.push_back(n - 1);
    const int ns = (int)sh.time.size();
    if (ns >= 2) {
      sh.time[0] = dur(sh.len[0], sh.S[0], startvel, fp.non_siguav);
      sh.time[ns - 1] = dur(sh.len[ns - 1], sh.S[ns - 1], fp.non_siguav, endvel);
    } else {
      sh.time[0] = dur(sh.len[0], sh.S[0], startvel, endvel);
    }
    // ---- 0.1 s samples of every segment (only their time stamps reach RunMINCOParking), flat boundary states
    std::vector<double> init_total(ns, 0.0);
    out->n_seg[h] = ns;
    if (ns > MS) continue; // more gear changes than the caller takes: nothing is produced (as the kernel)
    for (int i = 0; i < ns && i < MS; i++) {
      double initv = fp.non_siguav, finv = fp.non_siguav;
      double ictrl[2] = {0.0, 0.0};
      const double fctrl[2] = {0.0, 0.0};
      if (i == 0) {
        initv = startvel;
        ictrl[0] = start_ctrl[2 * h];
        ictrl[1] = start_ctrl[2 * h + 1];
      }
      if (i == ns - 1) finv = endvel;
      const double locallength = sh.len[i];
      const int sig = sh.S[i];
      const int f0 = sh.index[i], f1 = sh.index[i + 1];
      std::vector<double> pts_t;
      double samplet, tmparc = 0;
      int index = 0;
      double sampletime = 0.1;
      if (sh.time[i] <= sampletime) sampletime = sh.time[i] / 2.0;
      for (samplet = sampletime; samplet < sh.time[i]; samplet += sampletime) {
        const double arc = sig > 0 ? evaluate_length(samplet, locallength, sh.time[i], fp.max_forward_vel, fp.max_forward_acc, initv, finv)
                                   : evaluate_length(samplet, locallength, sh.time[i], fp.max_backward_vel, fp.max_backward_acc, initv, finv);
        for (int k = index; k < (f1 - f0); k++) { // localTraj = SampleTraj[f0 .. f1]
          const double seg = norm2(P + 3 * (f0 + k), P + 3 * (f0 + k + 1));
          tmparc += seg;
          if (tmparc >= arc) {
            index = k;
            pts_t.push_back(sampletime);
            tmparc -= seg;
            break;
          }
        }
      }
      pts_t.push_back(sh.time[i] - (samplet - sampletime));
      for (double v : pts_t) init_total[i] += v; // initTotalduration, traj_manager.cpp:540-542
      out->singul[(size_t)h * MS + i] = sig;
      flat_state(fp, T, P + 3 * f0, initv, ictrl, sig, out->ini_states + ((size_t)h * MS + i) * 6);
      flat_state(fp, T, P + 3 * f1, finv, fctrl, sig, out->fin_states + ((size_t)h * MS + i) * 6);
    }
    sh.total = 0.0;
    for (double dt : sh.time) sh.total += dt;
    // ---- RunMINCOParking, traj_manager.cpp:531-568
    double basetime = 0.0;
    for (int i = 0; i < ns && i < MS; i++) {
      double timePerPiece = fp.piece_duration;
      const double initTotalduration = init_total[i];
      const int piece_nums = std::max(int(initTotalduration / timePerPiece + 0.5), 2);
      timePerPiece = initTotalduration / piece_nums;
      out->piece_nums[(size_t)h * MS + i] = piece_nums;
      out->piece_dt[(size_t)h * MS + i] = timePerPiece;
      double *inner = out->inner_pts + ((size_t)h * MS + i) * (size_t)(MP - 1) * 2;
      double *st = out->states + ((size_t)h * MS + i) * (size_t)MST * 3;
      int cnt = 0;
      double res_time = 0;
      for (int j = 0; j < piece_nums; j++) {
        const int resolution = (j == 0 || j == piece_nums - 1) ? fp.dense_traj_res : fp.traj_res;
        for (int k = 0; k <= resolution; k++) {
          const double t = basetime + res_time + 1.0 * k / resolution * timePerPiece;
          double pos[3];
          evaluate_pos(fp, P, sh, startvel, endvel, t, pos);
          if (cnt < MST) {
            st[3 * cnt] = pos[0];
            st[3 * cnt + 1] = pos[1];
            st[3 * cnt + 2] = pos[2];
          }
          cnt++;
          if (k == resolution && j != piece_nums - 1 && j < MP - 1) {
            inner[2 * j] = pos[0];
            inner[2 * j + 1] = pos[1];
          }
        }
        res_time += timePerPiece;
      }
      out->n_states[(size_t)h * MS + i] = cnt;
      basetime += initTotalduration;
    }
  }
}
