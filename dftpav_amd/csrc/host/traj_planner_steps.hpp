// traj_planner_steps.hpp — C++ host side of the steps around the solve: the member functions
// of TrajPlanner / KinoAstar / TrajPlannerServer that feed PolyTrajOptimizer::OptimizeTrajectory and consume its
// result, with the reference's names and argument meaning, over the C-ABI of include/dftpav_hip.h.  Header-only,
// no Eigen, no ROS; nothing here computes — every step runs on the GPU.
//
//   reference                                                         here (class plan_manage::TrajPlannerSteps)
//   map_itf_->GetObstacleMap(&grid_map)             MGR:1216           setObstacleMap(cells, size_x, size_y, resolution, origin)
//   KinoAstar::getKinoNode(flat_trajs) + the        KA:606-743,        getKinoNode(SampleTraj, start_state, end_state, start_ctrl)
//     resampling loop of RunMINCOParking            MGR:531-568          -> std::vector<FlatTrajData> (one per gear segment)
//   KinoAstar::computeShotTraj / is_shot_sucess     KA:304-345         computeShotTraj(state1, state2, path_list, len) / is_shot_sucess(state1, state2)
//   TrajPlanner::getRectangleConst(statelist)       MGR:1213-1469      getRectangleConst(statelist) -> hPolys_
//   TrajPlanner::ConverSurroundTrajFromPoints(...)  MGR:743-789        ConverSurroundTrajFromPoints(sur_trajs) (installs them)
//   collision part of CheckReplan                   SRV:385-397        CheckCollision(batch) -> per trajectory bool
//   Trajectory::GetState as PublishData plays it    PTU:378-406,       GetStates(batch, t0, dt, n) -> common::State rows per trajectory
//     back, FilterSingularityState                  SRV:244-259,335-356
//   LocalTrajData as bytes (PolyTraj.msg, unused)   msg/PolyTraj.msg   SerializeTraj(...) / setSurroundTrajsFromWire(blobs)
// (PTU = plan_utils/poly_traj_utils.hpp, MGR = traj_planner/src/traj_manager.cpp, KA = traj_planner/src/kino_astar.cpp, SRV = traj_planner/src/traj_server_ros.cpp)
#pragma once
#include <array>
#include <vector>

#include "../../../include/dftpav_hip.h"
#include "poly_traj_optimizer.hpp"

namespace plan_manage {

// common::State fields ConverSurroundTrajFromPoints reads (state.h): position, angle, velocity, acceleration,
// curvature, time_stamp
struct PredictedState {
  double x, y, angle, velocity, acceleration, curvature, time_stamp;
};

// common::State as Trajectory::GetState fills it (poly_traj_utils.hpp:388-404) — one row of dftpav_batch_sample_states
struct State {
  double time_stamp, x, y, angle, curvature, velocity, acceleration, steer;
};

// plan_utils::FlatTrajData (traj_container.hpp) plus what RunMINCOParking derives from it per gear segment
struct FlatTrajData {
  int singul = 1;
  Mat start_state{2, 3}, final_state{2, 3};
  int piece_nums = 0;
  double piece_duration = 0.0;               // timePerPiece; duration_container[i] = piece_duration * piece_nums
  Mat inner_pts;                             // 2 x (piece_nums - 1), ego_innerPs
  std::vector<std::array<double, 3>> states; // statelist handed to getRectangleConst
};

class TrajPlannerSteps {
 public:
  explicit TrajPlannerSteps(dftpav_handle *h) : h_(h) {}
  int last_error() const { return err_; }

  bool setObstacleMap(const unsigned char *cells, int size_x, int size_y, double resolution, double origin_x, double origin_y) {
    dftpav_grid_map m{cells, size_x, size_y, resolution, origin_x, origin_y};
    return ok(dftpav_set_grid_map(h_, &m));
  }

  // one hypothesis; the reference's members start_state_, end_state_, start_ctrl are arguments here
  bool getKinoNode(const std::vector<std::array<double, 3>> &SampleTraj, const std::array<double, 4> &start_state,
                   const std::array<double, 4> &end_state, const std::array<double, 2> &start_ctrl,
                   const dftpav_frontend_params &fp, std::vector<FlatTrajData> &flat_trajs) {
    flat_trajs.clear();
    const int MS = 8, MP = 64, MST = (MP - 2) * (fp.traj_res + 1) + 2 * (fp.dense_traj_res + 1);
    std::vector<int> n_seg(1), singul(MS), pieces(MS), n_states(MS);
    std::vector<double> dt(MS), ini(MS * 6), fin(MS * 6), inner((size_t)MS * (MP - 1) * 2), states((size_t)MS * MST * 3);
    dftpav_frontend_out out{MS, MP, MST, n_seg.data(), singul.data(), pieces.data(), dt.data(), ini.data(),
                            fin.data(), inner.data(), n_states.data(), states.data()};
    const int len = (int)SampleTraj.size();
    if (!ok(dftpav_frontend_resample(h_, &fp, &SampleTraj[0][0], &len, len, start_state.data(), end_state.data(),
                                     start_ctrl.data(), 1, &out)))
      return false;
    if (n_seg[0] > MS) {
      err_ = DFTPAV_E_UNSUPPORTED;
      return false;
    }
    for (int i = 0; i < n_seg[0]; i++) {
      FlatTrajData f;
      f.singul = singul[i];
      for (int k = 0; k < 6; k++) {
        f.start_state.a[k] = ini[(size_t)i * 6 + k];
        f.final_state.a[k] = fin[(size_t)i * 6 + k];
      }
      f.piece_nums = pieces[i];
      f.piece_duration = dt[i];
      f.inner_pts = Mat(2, pieces[i] - 1);
      for (int j = 0; j < pieces[i] - 1; j++)
        for (int d = 0; d < 2; d++) f.inner_pts(d, j) = inner[((size_t)i * (MP - 1) + j) * 2 + d];
      for (int s = 0; s < n_states[i] && s < MST; s++)
        f.states.push_back({states[((size_t)i * MST + s) * 3], states[((size_t)i * MST + s) * 3 + 1],
                            states[((size_t)i * MST + s) * 3 + 2]});
      flat_trajs.push_back(f);
    }
    return true;
  }

  // hPolys_: one 4x4 matrix per state, columns (n_x, n_y, p_x, p_y)
  bool getRectangleConst(const std::vector<std::array<double, 3>> &statelist) {
    hPolys_.assign(statelist.size(), Mat(4, 4));
    if (statelist.empty()) return true;
    std::vector<double> h(16 * statelist.size());
    if (!ok(dftpav_corridor_rectangles(h_, &statelist[0][0], (int)statelist.size(), h.data()))) return false;
    for (size_t i = 0; i < statelist.size(); i++) hPolys_[i].a.assign(h.begin() + 16 * i, h.begin() + 16 * (i + 1));
    return true;
  }
  const std::vector<Mat> &hPolys() const { return hPolys_; }

  // all sequences must have the same number of states (they do: the prediction horizon is common, MGR:743)
  bool ConverSurroundTrajFromPoints(const std::vector<std::vector<PredictedState>> &sur_trajs) {
    if (sur_trajs.empty()) return ok(dftpav_fit_surround(h_, nullptr, 0, 0)); // kWrongStatus in the reference: no obstacles
    const size_t n = sur_trajs[0].size();
    std::vector<double> st;
    for (const auto &t : sur_trajs) {
      if (t.size() != n) {
        err_ = DFTPAV_E_INVALID;
        return false;
      }
      for (const auto &s : t) st.insert(st.end(), {s.x, s.y, s.angle, s.velocity, s.acceleration, s.curvature, s.time_stamp});
    }
    return ok(dftpav_fit_surround(h_, st.data(), (int)sur_trajs.size(), (int)n));
  }

  // the Reeds-Shepp shot from a pose to the goal (turning radius 1 / max_cur, samples every checkl)
  bool computeShotTraj(const std::array<double, 3> &state1, const std::array<double, 3> &state2,
                       std::vector<std::array<double, 3>> &path_list, double &len, double max_cur = 1.0, double checkl = 0.2) {
    const int cap = 4096;
    std::vector<double> smp((size_t)cap * 3);
    int n = 0;
    path_list.clear();
    if (!ok(dftpav_reeds_shepp_shots(h_, state1.data(), state2.data(), 1, max_cur, checkl, cap, 0.1, &len, nullptr, nullptr, smp.data(),
                                     &n, nullptr)))
      return false;
    for (int i = 0; i < n && i < cap; i++) path_list.push_back({smp[3 * i], smp[3 * i + 1], smp[3 * i + 2]});
    return true;
  }
  // ... and whether it is free on the map of setObstacleMap
  bool is_shot_sucess(const std::array<double, 3> &state1, const std::array<double, 3> &state2, double max_cur = 1.0,
                      double checkl = 0.2) {
    int hit = 1;
    if (!ok(dftpav_reeds_shepp_shots(h_, state1.data(), state2.data(), 1, max_cur, checkl, 4096, 0.1, nullptr, nullptr, nullptr, nullptr,
                                     nullptr, &hit)))
      return false;
    return hit == 0;
  }

  // the collision loop of CheckReplan for every trajectory of a solved batch
  bool CheckCollision(dftpav_batch *batch, int B, std::vector<int> &is_collision) {
    is_collision.assign(B, 0);
    return ok(dftpav_batch_validate(batch, 0.05, 0.1, is_collision.data(), nullptr));
  }

  // the states the server would publish for every trajectory of a solved batch at t0, t0 + dt, ...;
  // states[t] holds the samples that fall inside trajectory t
  bool GetStates(dftpav_batch *batch, int B, double t0, double dt, int n_samples, std::vector<std::vector<State>> &states,
                 bool filter_singularity = true) {
    static_assert(sizeof(State) == 8 * sizeof(double), "State is 8 packed doubles");
    std::vector<State> flat((size_t)B * n_samples);
    std::vector<int> valid(B, 0);
    states.assign(B, {});
    if (!ok(dftpav_batch_sample_states(batch, t0, dt, n_samples, filter_singularity ? 1 : 0, &flat[0].time_stamp, valid.data())))
      return false;
    for (int t = 0; t < B; t++) states[t].assign(flat.begin() + (size_t)t * n_samples, flat.begin() + (size_t)t * n_samples + valid[t]);
    return true;
  }

  // one solved trajectory (its rows of dftpav_batch_coeffs) as a "DPTJ" blob
  bool SerializeTraj(const dftpav_layout &layout, const double *coeffs, const double *piece_dt, int drone_id, int traj_id,
                     double start_time, std::vector<unsigned char> &blob) {
    blob.assign(dftpav_wire_size(layout.M, layout.piece_nums), 0);
    size_t written = 0;
    if (blob.empty()) return ok(DFTPAV_E_INVALID);
    return ok(dftpav_wire_pack(&layout, coeffs, piece_dt, drone_id, traj_id, start_time, blob.data(), blob.size(), &written));
  }
  // setSurroundTrajs for obstacles that arrive serialised
  bool setSurroundTrajsFromWire(const std::vector<std::vector<unsigned char>> &blobs) {
    std::vector<const void *> ptr;
    std::vector<size_t> len;
    for (const auto &b : blobs) {
      ptr.push_back(b.data());
      len.push_back(b.size());
    }
    return ok(dftpav_set_surround_wire(h_, ptr.data(), len.data(), (int)blobs.size()));
  }

 private:
  bool ok(int rc) {
    err_ = rc;
    return rc == DFTPAV_OK;
  }
  dftpav_handle *h_;
  std::vector<Mat> hPolys_;
  int err_ = 0;
};

} // namespace plan_manage
