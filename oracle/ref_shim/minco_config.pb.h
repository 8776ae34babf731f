// Stand-in for the protobuf-generated minco_config.pb.h (reference: src/Plan/traj_planner/proto/minco_config.proto:68-103,
// message OptCfg inside message Config; protoc is not installed here).  Plain struct with the generated getters' names
// that PolyTrajOptimizer::setParam reads (traj_optimizer.cpp:1715-1736); defaults are the shipped configuration
// src/Plan/traj_planner/config/minco_config.pb.txt:65-100.  TEST INFRASTRUCTURE for oracle/_ref.
#pragma once
namespace planning {
namespace minco {
#define DFTPAV_PB_FIELD(type, name, dflt)           \
 private:                                           \
  type name##_ = dflt;                              \
 public:                                            \
  type name() const { return name##_; }             \
  void set_##name(type v) { name##_ = v; }
class OptCfg {
  DFTPAV_PB_FIELD(int, traj_resolution, 16)
  DFTPAV_PB_FIELD(int, des_traj_resolution, 32)
  DFTPAV_PB_FIELD(double, wei_sta_obs, 1000.0)
  DFTPAV_PB_FIELD(double, wei_dyn_obs, 5000.0)
  DFTPAV_PB_FIELD(double, wei_feas, 2500.0)
  DFTPAV_PB_FIELD(double, wei_sqrvar, 500.0)
  DFTPAV_PB_FIELD(double, wei_time, 500.0)
  DFTPAV_PB_FIELD(double, dyn_obs_clearance, 0.4)
  DFTPAV_PB_FIELD(double, half_margin, 0.15)
  DFTPAV_PB_FIELD(double, traj_piece_duration, 1.0)
  DFTPAV_PB_FIELD(double, max_frontend_forward_vel, 5.0)
  DFTPAV_PB_FIELD(double, max_frontend_forward_acc, 8.0)
  DFTPAV_PB_FIELD(double, max_frontend_backward_vel, 2.0)
  DFTPAV_PB_FIELD(double, max_frontend_backward_acc, 4.0)
  DFTPAV_PB_FIELD(double, max_frontend_cur, 1.0)
  DFTPAV_PB_FIELD(double, max_forward_vel, 5.0)
  DFTPAV_PB_FIELD(double, max_forward_acc, 8.0)
  DFTPAV_PB_FIELD(double, max_forward_cur, 1.0)
  DFTPAV_PB_FIELD(double, max_backward_vel, 2.0)
  DFTPAV_PB_FIELD(double, max_backward_acc, 4.0)
  DFTPAV_PB_FIELD(double, max_backward_cur, 1.0)
  DFTPAV_PB_FIELD(double, max_latacc, 5.0)
  DFTPAV_PB_FIELD(double, max_phidot, 10000.0)
  DFTPAV_PB_FIELD(double, max_nonsv, 0.1)
  DFTPAV_PB_FIELD(bool, gearopt, true)
  DFTPAV_PB_FIELD(int, lbfgs_memsize, 256)
  DFTPAV_PB_FIELD(int, lbfgs_past, 3)
  DFTPAV_PB_FIELD(double, lbfgs_delta, 0.0001)
  DFTPAV_PB_FIELD(double, mini_t, 0.1)
};
#undef DFTPAV_PB_FIELD
class Config {
 public:
  const OptCfg &opt_cfg() const { return opt_; }
  OptCfg *mutable_opt_cfg() { return &opt_; }
 private:
  OptCfg opt_;
};
}  // namespace minco
}  // namespace planning
