// C entry points of oracle/_ref/libdftpav_ref_next.so -- TEST INFRASTRUCTURE.
//
// The reference's OWN code of the steps either side of the solve path, compiled here so that the restatements
// oracle/{corridor,validate,states,fit,frontend}_oracle.cpp can be held bit-equal to it (tests/test_ref_pin.py):
//
//   (f)-1  TrajPlanner::getRectangleConst                   traj_planner/src/traj_manager.cpp:1213-1469
//          TrajPlannerAdapter::CheckIfCollisionUsingLine     traj_planner/src/map_adapter.cpp:117-129 (+ :93-97, :110-115)
//   (f)-2  Trajectory::GetState, getPos, getAngle            plan_utils/poly_traj_utils.hpp (whole header, as in libdftpav_ref.so)
//          TrajContainer::addSingulTraj                      plan_utils/traj_container.hpp (whole header)
//          TrajPlannerServer: playback tick, FilterSingularityState, CheckReplan's re-check loop
//                                                            traj_planner/src/traj_server_ros.cpp:248-259, 335-356, 385-397
//          SemanticMapManager::CheckCollisionUsingPosAndYaw / ...GlobalPosition      semantic_map_manager.cc:639-662, 710-715
//          ShapeUtils::GetDenseVerticesOfOrientedBoundingBox common/basics/shapes.cc:110-149
//          GridMapND (class + members), normalize_angle      common/basics/semantics.h:350-604, semantics.cc:130-322, calculations.cc:18-23
//   (f)-3  KinoAstar::getKinoNode from SampleTraj on, evaluatePos, evaluateDuration / evaluateLength, getFlatState
//                                                            traj_planner/src/kino_astar.cpp:613-743, 468-521, 744-795, 834-857
//          TrajPlanner::RunMINCOParking, the resampling loop  traj_planner/src/traj_manager.cpp:531-568, 573-577
//   (f)-4  TrajPlanner::ConverSurroundTrajFromPoints, state_to_flat_output           traj_manager.cpp:743-789, 139-158
//
// The files those functions live in cannot be compiled whole here (ROS, OMPL, PCL, OpenCV, the simulator's lane / vehicle
// library), so oracle/ref_slices.py cuts the functions out of /root/reference by line range, VERBATIM, into oracle/_ref/slices/
// (build products, git-ignored), and this file supplies what surrounds them: class declarations with the reference's member
// names (each citing the declaration it stands for) and the conversion between plain arrays and the reference's argument types.
// Headers that can be used whole are: common/basics/basics.h (simulator), plan_utils/poly_traj_utils.hpp, traj_container.hpp.
// This file computes nothing itself.  Not covered: the A* expansion and the OMPL Reeds-Shepp shot in front of SampleTraj
// (kino_astar.cpp:566-605; OMPL is not vendored -- oracle/shot_oracle*.cpp stay property-pinned).
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include <Eigen/Eigen>
#include <ros/ros.h>
#include <visualization_msgs/Marker.h>
#include "common/basics/basics.h"     // the simulator's own header (ErrorType, decimal_t, Vecf, vec_E, kBigEPS, kPi)
#include "common/basics/semantics.h"  // stand-in: common::VehicleParam with the reference's defaults
#include "common/state/state.h"       // stand-in: common::State
#include "plan_utils/traj_container.hpp"  // the reference's own header (pulls in poly_traj_utils.hpp)
#include "../include/dftpav_hip.h"    // dftpav_frontend_params / dftpav_frontend_out (the oracle's argument types)

// ---------------------------------------------------------------------------------------- simulator library (namespace common)
decimal_t normalize_angle(const decimal_t& theta);  // common/math/calculations.h
#include "_ref/slices/normalize_angle.inc"

namespace common {
#include "_ref/slices/gridmap_class.inc"
#include "_ref/slices/gridmap_members.inc"
template class GridMapND<uint8_t, 2>;  // semantics.cc:324

#include "_ref/slices/obb_struct.inc"
#include "_ref/slices/obb_ctors.inc"
class ShapeUtils {  // common/basics/shapes.h:156-: only the member the re-check reaches
 public:
#include "_ref/slices/dense_vertices_decl.inc"
};
#include "_ref/slices/dense_vertices.inc"
}  // namespace common

// ---------------------------------------------------------------------------------------- semantic_map_manager
namespace semantic_map_manager {
class SemanticMapManager {  // semantic_map_manager.h:30-: the obstacle map and the two queries on it
 public:
  using ObstacleMapType = uint8_t;                                  // :32
  using GridMap2D = common::GridMapND<ObstacleMapType, 2>;          // :33
  ErrorType CheckCollisionUsingGlobalPosition(const Vec2f &p_w, bool *res) const;  // :45-46
  ErrorType CheckCollisionUsingPosAndYaw(const common::VehicleParam &vehicle_param, const Eigen::Vector3d &state, bool *res);  // :52-54
  inline common::GridMapND<ObstacleMapType, 2> obstacle_map() const { return obstacle_map_; }  // :159-161
  common::GridMapND<ObstacleMapType, 2> obstacle_map_;              // :262
};
#include "_ref/slices/smm_pos_and_yaw.inc"
#include "_ref/slices/smm_global_position.inc"
}  // namespace semantic_map_manager

// ---------------------------------------------------------------------------------------- map_utils
namespace map_utils {
class TrajPlannerMapItf {  // map_utils/map_interface.h:31-70: the members the sliced code calls
 public:
  using ObstacleMapType = uint8_t;
  using State = common::State;
  using GridMap2D = common::GridMapND<ObstacleMapType, 2>;
  virtual ~TrajPlannerMapItf() {}
  virtual ErrorType GetObstacleMap(GridMap2D *grid_map) = 0;
  virtual ErrorType CheckIfCollisionUsingPosAndYaw(const common::VehicleParam &vehicle_param, const Eigen::Vector3d &state, bool *res) = 0;
  virtual ErrorType CheckIfCollisionUsingLine(const Eigen::Vector2d p1, const Eigen::Vector2d p2, bool *res, double checkl) = 0;
};
class TrajPlannerAdapter : public TrajPlannerMapItf {  // map_utils/map_adapter.h:29-
 public:
  using IntegratedMap = semantic_map_manager::SemanticMapManager;  // :31
  ErrorType GetObstacleMap(GridMap2D *grid_map) override;
  ErrorType CheckIfCollisionUsingPosAndYaw(const common::VehicleParam &vehicle_param, const Eigen::Vector3d &state, bool *res) override;
  ErrorType CheckIfCollisionUsingLine(const Eigen::Vector2d p1, const Eigen::Vector2d p2, bool *res, double checkl) override;
  std::shared_ptr<IntegratedMap> map_;
  bool is_valid_ = false;
};
#include "_ref/slices/adapter_obstacle_map.inc"
#include "_ref/slices/adapter_pos_and_yaw.inc"
#include "_ref/slices/adapter_line.inc"
}  // namespace map_utils

// ---------------------------------------------------------------------------------------- path_searching::KinoAstar
namespace path_searching {
class KinoAstar {  // path_searching/kino_astar.h:128-: member declarations cut from the header itself
 public:
#include "_ref/slices/kino_members_states.inc"
#include "_ref/slices/kino_members_limits.inc"
#include "_ref/slices/kino_members_shot.inc"
#include "_ref/slices/kino_members_flat.inc"
#include "_ref/slices/kino_members_profile.inc"
#include "_ref/slices/kino_members_vehicle.inc"
#include "_ref/slices/kino_members_total.inc"
  Eigen::Vector3d evaluatePos(double t);  // :250
  // getKinoNode (:241) from the line behind `SampleTraj = roughSampleList;` on: what precedes it builds SampleTraj from the A*
  // nodes and the OMPL shot
  void getKinoNodeFromSampleTraj(plan_utils::KinoTrajData &flat_trajs);
};
#include "_ref/slices/kino_evaluate_pos.inc"
void KinoAstar::getKinoNodeFromSampleTraj(plan_utils::KinoTrajData &flat_trajs)
{
#include "_ref/slices/kino_node_locals_a.inc"
#include "_ref/slices/kino_node_locals_b.inc"
#include "_ref/slices/kino_node_locals_c.inc"
#include "_ref/slices/kino_node_body.inc"
// (the slice ends with the function's closing brace)
#include "_ref/slices/kino_profile.inc"
#include "_ref/slices/kino_flat_state.inc"
}  // namespace path_searching

// ---------------------------------------------------------------------------------------- plan_manage::TrajPlanner
namespace plan_manage {
using State = common::State;  // traj_manager.h:57
struct SegmentRecord {        // (driver bookkeeping: what one pass of the resampling loop produced)
  int piece_nums;
  double timePerPiece;
  Eigen::MatrixXd innerPs;
  std::vector<Eigen::Vector3d> statelist;
};
class TrajPlanner {  // plan_manage/traj_manager.h:60-: the members the sliced functions touch
 public:
  using GridMap2D = common::GridMapND<uint8_t, 2>;  // :64
  double traj_piece_duration_;                        // :114
  int traj_res, dense_traj_res;                       // :115
  std::unique_ptr<path_searching::KinoAstar> kino_path_finder_;  // :133
  plan_utils::KinoTrajData kino_trajs_;               // :134
  map_utils::TrajPlannerMapItf *map_itf_ = nullptr;   // :165
  std::vector<Eigen::MatrixXd> hPolys_, display_hPolys_;  // :172
  ErrorType ConverSurroundTrajFromPoints(std::vector<std::vector<common::State>> sur_trajs, plan_utils::SurroundTrajData *surround_trajs_ptr);  // :199
  Eigen::MatrixXd state_to_flat_output(const State &state);  // :205
  ErrorType getRectangleConst(std::vector<Eigen::Vector3d> statelist);  // :219
  ros::Publisher DebugCorridorPub;                    // :234
  // RunMINCOParking (:509-641) up to the corridor call of every gear segment
  void resampleOfRunMINCOParking(std::vector<SegmentRecord> *segments);
};
#include "_ref/slices/state_to_flat_output.inc"
#include "_ref/slices/fit_surround.inc"
#include "_ref/slices/rectangle.inc"
void TrajPlanner::resampleOfRunMINCOParking(std::vector<SegmentRecord> *segments) {
#include "_ref/slices/resample_locals.inc"
#include "_ref/slices/resample_containers.inc"
#include "_ref/slices/resample_loop.inc"
    // (traj_manager.cpp:569-572 -- a print, a timer, getRectangleConst(statelist) and the push of its result -- are left out:
    // the corridor is pinned by ref_corridor_rectangles on its own)
    segments->push_back(SegmentRecord{piece_nums, timePerPiece, ego_innerPs, statelist});
#include "_ref/slices/resample_loop_tail.inc"
  }
}
}  // namespace plan_manage

// ---------------------------------------------------------------------------------------- plan_manage::TrajPlannerServer
namespace plan_manage {
using plan_utils::SingulTrajData;
// (driver bookkeeping) the re-check loop returns at the first colliding sample without saying which: the adapter the server holds
// counts the pose queries it forwards
struct CountingAdapter : map_utils::TrajPlannerAdapter {
  int pose_queries = 0;
  ErrorType CheckIfCollisionUsingPosAndYaw(const common::VehicleParam &vehicle_param, const Eigen::Vector3d &state, bool *res) override {
    ++pose_queries;
    return map_utils::TrajPlannerAdapter::CheckIfCollisionUsingPosAndYaw(vehicle_param, state, res);
  }
};
struct NoMutex {  // traj_server_ros.h:144 `std::mutex m`: the playback slice unlocks it on its early return
  void unlock() {}
};
class TrajPlannerServer {  // plan_utils/traj_server_ros.h:30-
 public:
  int final_traj_index_ = 0, exe_traj_index_ = 0;       // :98
  std::unique_ptr<SingulTrajData> executing_traj_;      // :99
  CountingAdapter map_adapter_;                         // :118 (map_utils::TrajPlannerAdapter)
  vec_E<common::State> ctrl_state_hist_;                // :134
  NoMutex m;                                            // :144
  common::VehicleParam vp_;                             // :148
  ErrorType FilterSingularityState(const vec_E<common::State> &hist, common::State *filter_state);  // :84-85
  // one tick of PublishData's trajectory feedback (traj_server_ros.cpp:244-259); *published = a state was produced
  void playbackTick(double t, common::State &state, bool *published);
  // CheckReplan's collision part (traj_server_ros.cpp:384-400)
  bool recheck();
};
#include "_ref/slices/server_filter.inc"
void TrajPlannerServer::playbackTick(double t, common::State &state, bool *published) {
  *published = false;
  state.time_stamp = t;  // traj_server_ros.cpp:246
#include "_ref/slices/server_playback.inc"
  *published = true;
}
bool TrajPlannerServer::recheck() {
  bool is_collision = false;  // traj_server_ros.cpp:364
#include "_ref/slices/server_recheck.inc"
  return false;               // :400
}
}  // namespace plan_manage

// ======================================================================================== plain-array entry points
namespace {
struct Quiet {  // the sliced code prints on std::cout
  std::streambuf *old;
  std::ostringstream sink;
  Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~Quiet() { std::cout.rdbuf(old); }
};

std::shared_ptr<semantic_map_manager::SemanticMapManager> make_map(const unsigned char *grid, int size_x, int size_y, double resolution,
                                                                   double origin_x, double origin_y) {
  auto smm = std::make_shared<semantic_map_manager::SemanticMapManager>();
  auto &g = smm->obstacle_map_;
  g.set_dims_size({{size_x, size_y}});
  g.set_dims_resolution({{resolution, resolution}});
  g.set_origin({{origin_x, origin_y}});
  g.set_data(std::vector<uint8_t>(grid, grid + (size_t)size_x * size_y));
  return smm;
}

// Trajectory of one gear segment from coefficients [N][6][2] (entry [k][d] = coefficient of t^k): the column order MinJerkOpt::getTraj
// produces (poly_traj_utils.hpp:987-997: t^5 first)
plan_utils::Trajectory make_trajectory(const double *c, int N, double piece_dt, int singul) {
  std::vector<double> durs(N, piece_dt);
  std::vector<plan_utils::CoefficientMat> mats(N);
  for (int p = 0; p < N; p++)
    for (int k = 0; k < 6; k++)
      for (int d = 0; d < 2; d++) mats[p](d, 5 - k) = c[(size_t)p * 12 + 2 * k + d];
  return plan_utils::Trajectory(durs, mats, singul);
}
bool default_vehicle(double w, double l, double dcr) {
  common::VehicleParam vp;
  return w == vp.width() && l == vp.length() && dcr == vp.d_cr();
}
}  // namespace

// == oracle_corridor_rectangles.  getRectangleConst builds its vehicles from `common::VehicleParam` defaults: other sizes are refused.
extern "C" int ref_corridor_rectangles(const unsigned char *grid, int size_x, int size_y, double resolution, double origin_x, double origin_y,
                                       const double *states, int n, double veh_width, double veh_length, double veh_dcr, int order,
                                       double *hpoly) {
  if (order != 0 || !default_vehicle(veh_width, veh_length, veh_dcr)) return -1;
  Quiet q;
  map_utils::TrajPlannerAdapter adapter;
  adapter.map_ = make_map(grid, size_x, size_y, resolution, origin_x, origin_y);
  adapter.is_valid_ = true;
  plan_manage::TrajPlanner planner;
  planner.map_itf_ = &adapter;
  std::vector<Eigen::Vector3d> statelist(n);
  for (int i = 0; i < n; i++) statelist[i] = Eigen::Vector3d(states[3 * i], states[3 * i + 1], states[3 * i + 2]);
  planner.getRectangleConst(statelist);
  if ((int)planner.hPolys_.size() != n) return -2;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 16; k++) hpoly[16 * (size_t)i + k] = planner.hPolys_[i].data()[k];
  return 0;
}

// == oracle_validate_trajectories.  The server samples at 0.05 s and the outline at the default 0.1 m: other values are refused.
extern "C" int ref_validate_trajectories(const unsigned char *grid, int size_x, int size_y, double resolution, double origin_x, double origin_y,
                                         const double *coeffs, const double *piece_dt, const int *piece_nums, const int *singuls, int M, int B,
                                         double veh_width, double veh_length, double veh_dcr, double sample_dt, double vertex_res, int order,
                                         int *collision, int *first_sample) {
  if (order != 0 || sample_dt != 0.05 || vertex_res != 0.1) return -1;
  Quiet q;
  int Ntot = 0;
  for (int i = 0; i < M; i++) Ntot += piece_nums[i];
  plan_manage::TrajPlannerServer server;
  server.map_adapter_.map_ = make_map(grid, size_x, size_y, resolution, origin_x, origin_y);
  server.map_adapter_.is_valid_ = true;
  server.vp_.set_width(veh_width);
  server.vp_.set_length(veh_length);
  server.vp_.set_d_cr(veh_dcr);
  for (int b = 0; b < B; b++) {
    plan_utils::TrajContainer container;
    double worldtime = 0.0;
    int p0 = 0;
    for (int i = 0; i < M; i++) {  // traj_manager.cpp:618-625
      container.addSingulTraj(make_trajectory(coeffs + ((size_t)b * Ntot + p0) * 12, piece_nums[i], piece_dt[(size_t)b * M + i], singuls[i]), worldtime);
      worldtime = container.singul_traj.back().end_time;
      p0 += piece_nums[i];
    }
    server.executing_traj_.reset(new plan_utils::SingulTrajData(container.singul_traj));
    // the loop returns at the first colliding sample: the index of that sample is the number of pose queries before it
    server.map_adapter_.pose_queries = 0;
    const bool hit = server.recheck();
    collision[b] = hit ? 1 : 0;
    first_sample[b] = hit ? server.map_adapter_.pose_queries - 1 : -1;
  }
  return 0;
}

// == oracle_sample_states
extern "C" int ref_sample_states(const double *coeffs, const double *piece_dt, const int *piece_nums, const int *singuls, int M, int B,
                                 double wheel_base, double t0, double sample_dt, int n_samples, int filter, int order, double *states,
                                 int *n_valid) {
  if (order != 0 || wheel_base != common::VehicleParam().wheel_base()) return -1;  // Piece::vp_ is a default VehicleParam
  Quiet q;
  int Ntot = 0;
  for (int i = 0; i < M; i++) Ntot += piece_nums[i];
  for (int b = 0; b < B; b++) {
    plan_utils::TrajContainer container;
    double worldtime = 0.0;
    int p0 = 0;
    for (int i = 0; i < M; i++) {  // traj_manager.cpp:618-625
      container.addSingulTraj(make_trajectory(coeffs + ((size_t)b * Ntot + p0) * 12, piece_nums[i], piece_dt[(size_t)b * M + i], singuls[i]), worldtime);
      worldtime = container.singul_traj.back().end_time;
      p0 += piece_nums[i];
    }
    plan_manage::TrajPlannerServer server;
    server.executing_traj_.reset(new plan_utils::SingulTrajData(container.singul_traj));
    server.exe_traj_index_ = 0;
    server.final_traj_index_ = M - 1;  // traj_server_ros.cpp:228
    double *out = states + (size_t)b * n_samples * 8;
    int valid = 0;
    for (int k = 0; k < n_samples; k++) {
      double *s = out + 8 * k;
      for (int j = 0; j < 8; j++) s[j] = 0.0;
      common::State state;
      bool published = false;
      if (!filter) server.ctrl_state_hist_.clear();  // an empty history makes FilterSingularityState return at once (:337-339)
      if (server.exe_traj_index_ <= server.final_traj_index_) server.playbackTick(t0 + (double)k * sample_dt, state, &published);
      if (!published) continue;
      valid = k + 1;
      s[0] = state.time_stamp; s[1] = state.vec_position[0]; s[2] = state.vec_position[1]; s[3] = state.angle;
      s[4] = state.curvature; s[5] = state.velocity; s[6] = state.acceleration; s[7] = state.steer;
    }
    n_valid[b] = valid;
  }
  return 0;
}

// == oracle_fit_surround.  states [S][n_states][7] = (x, y, angle, velocity, acceleration, curvature, time_stamp)
extern "C" int ref_fit_surround(const double *states, int S, int n_states, int order, double *dur, double *coef, double *total, double *start) {
  if (order != 0) return -1;
  Quiet q;
  std::vector<std::vector<common::State>> sur(S, std::vector<common::State>(n_states));
  for (int o = 0; o < S; o++)
    for (int i = 0; i < n_states; i++) {
      const double *st = states + ((size_t)o * n_states + i) * 7;
      common::State &s = sur[o][i];
      s.vec_position = Eigen::Vector2d(st[0], st[1]);
      s.angle = st[2]; s.velocity = st[3]; s.acceleration = st[4]; s.curvature = st[5]; s.time_stamp = st[6];
    }
  plan_manage::TrajPlanner planner;
  plan_utils::SurroundTrajData data;
  if (planner.ConverSurroundTrajFromPoints(sur, &data) != kSuccess) return -2;
  const int N = n_states - 1;
  for (int o = 0; o < S; o++) {
    const plan_utils::Trajectory &tr = data[o].traj;
    if (tr.getPieceNum() != N) return -3;
    for (int p = 0; p < N; p++) {
      dur[(size_t)o * N + p] = tr[p].getDuration();
      const plan_utils::CoefficientMat &cm = tr[p].getCoeffMat();
      for (int col = 0; col < 6; col++)
        for (int d = 0; d < 2; d++) coef[((size_t)o * N + p) * 12 + 2 * col + d] = cm(d, col);
    }
    total[o] = data[o].duration;
    start[o] = data[o].start_time;
  }
  return 0;
}

// == oracle_frontend_resample
extern "C" int ref_frontend_resample(const dftpav_frontend_params *fp, const double *paths, const int *path_len, int max_path,
                                     const double *start_states, const double *end_states, const double *start_ctrl, int n_hyp, int order,
                                     const dftpav_frontend_out *out) {
  if (order != 0) return -1;
  Quiet q;
  const int MS = out->max_seg, MP = out->max_pieces, MST = out->max_states;
  for (int h = 0; h < n_hyp; h++) {
    plan_manage::TrajPlanner planner;
    planner.kino_path_finder_.reset(new path_searching::KinoAstar());
    path_searching::KinoAstar &ka = *planner.kino_path_finder_;
    ka.max_forward_vel = fp->max_forward_vel; ka.max_forward_acc = fp->max_forward_acc;      // KinoAstar::init reads them from the config
    ka.max_backward_vel = fp->max_backward_vel; ka.max_backward_acc = fp->max_backward_acc;
    ka.non_siguav = fp->non_siguav;
    ka.vp_.set_wheel_base(fp->wheel_base);
    for (int k = 0; k < 4; k++) {
      ka.start_state_[k] = start_states[4 * h + k];
      ka.end_state_[k] = end_states[4 * h + k];
    }
    ka.start_ctrl = Eigen::Vector2d(start_ctrl[2 * h], start_ctrl[2 * h + 1]);
    ka.SampleTraj.clear();
    for (int i = 0; i < path_len[h]; i++) {
      const double *P = paths + ((size_t)h * max_path + i) * 3;
      ka.SampleTraj.push_back(Eigen::Vector3d(P[0], P[1], P[2]));
    }
    ka.getKinoNodeFromSampleTraj(planner.kino_trajs_);
    const int ns = (int)planner.kino_trajs_.size();
    out->n_seg[h] = ns;
    if (ns > MS) continue;
    planner.traj_piece_duration_ = fp->piece_duration;
    planner.traj_res = fp->traj_res;
    planner.dense_traj_res = fp->dense_traj_res;
    std::vector<plan_manage::SegmentRecord> segs;
    planner.resampleOfRunMINCOParking(&segs);
    for (int i = 0; i < ns; i++) {
      const plan_utils::FlatTrajData &ft = planner.kino_trajs_[i];
      const size_t hs = (size_t)h * MS + i;
      out->singul[hs] = ft.singul;
      for (int k = 0; k < 6; k++) {
        out->ini_states[hs * 6 + k] = ft.start_state.data()[k];
        out->fin_states[hs * 6 + k] = ft.final_state.data()[k];
      }
      out->piece_nums[hs] = segs[i].piece_nums;
      out->piece_dt[hs] = segs[i].timePerPiece;
      double *inner = out->inner_pts + hs * (size_t)(MP - 1) * 2;
      for (int j = 0; j < segs[i].piece_nums - 1 && j < MP - 1; j++) {
        inner[2 * j] = segs[i].innerPs(0, j);
        inner[2 * j + 1] = segs[i].innerPs(1, j);
      }
      double *st = out->states + hs * (size_t)MST * 3;
      const int cnt = (int)segs[i].statelist.size();
      for (int j = 0; j < cnt && j < MST; j++)
        for (int d = 0; d < 3; d++) st[3 * j + d] = segs[i].statelist[j][d];
      out->n_states[hs] = cnt;
    }
  }
  return 0;
}
