// poly_traj_optimizer.hpp — C++ host side of the drop-in: plan_manage::PolyTrajOptimizer with the
// reference's entry points (traj_planner/include/plan_manage/traj_optimizer.h:100-120), implemented
// over the C-ABI of include/dftpav_hip.h.  Header-only, no Eigen, no ROS: the ROS host passes
// Eigen::MatrixXd storage straight through (`Mat` below is column-major like Eigen's default, and
// INTEGRATION.md shows the three-line adapter).  The solve runs on the GPU; nothing here computes.
//
//   reference                                            here
//   setParam(ros::NodeHandle, planning::minco::Config)   setParam(const dftpav_params&)
//   setSurroundTrajs(plan_utils::SurroundTrajData*)      setSurroundTrajs(const SurroundSet*)
//   bool OptimizeTrajectory(iniStates, finStates,        same argument list, same bool / no-throw
//        initInnerPts, initTs, hPoly_container,           error behaviour (traj_optimizer.cpp:26-48,
//        singuls, now, help_eps)                          176-201)
//   getMinJerkOptPtr()                                    vector<MinJerkOptView>: getCoeffs/getDt/getTraj
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

#include "../../../include/dftpav_hip.h"

namespace plan_manage {

// column-major dense matrix (the storage order of Eigen::MatrixXd)
struct Mat {
  int rows = 0, cols = 0;
  std::vector<double> a;
  Mat() = default;
  Mat(int r, int c) : rows(r), cols(c), a((size_t)r * c, 0.0) {}
  double &operator()(int i, int j) { return a[(size_t)j * rows + i]; }
  double operator()(int i, int j) const { return a[(size_t)j * rows + i]; }
  const double *data() const { return a.data(); }
};

// plan_utils::LocalTrajData fields the hot path reads (traj_container.hpp:28-38)
struct SurroundTraj {
  std::vector<double> durations; // per piece
  std::vector<double> coeffs;    // per piece 2x6 column-major, column 0 = t^5 (poly_traj_utils.hpp:993)
  double duration = 0.0, start_time = 0.0;
};
typedef std::vector<SurroundTraj> SurroundSet;

// what callers read from getMinJerkOptPtr()[i] (poly_traj_utils.hpp:987-997, 1069-1074)
class MinJerkOptView {
 public:
  MinJerkOptView(int N, const double *c, double dt) : N_(N), c_(c, c + 12 * (size_t)N), dt_(dt) {}
  // (6N)x2, row 6i+k = k-th power coefficient of piece i
  Mat getCoeffs() const {
    Mat m(6 * N_, 2);
    for (int r = 0; r < 6 * N_; r++)
      for (int d = 0; d < 2; d++) m(r, d) = c_[2 * (size_t)r + d];
    return m;
  }
  double getDt() const { return dt_; }
  int getPieceNum() const { return N_; }
  // per piece the 2x6 coeffMat with column 0 = t^5, as Trajectory::emplace_back receives it
  std::vector<Mat> getTrajCoeffMats() const {
    std::vector<Mat> out;
    for (int i = 0; i < N_; i++) {
      Mat m(2, 6);
      for (int k = 0; k < 6; k++)
        for (int d = 0; d < 2; d++) m(d, 5 - k) = c_[2 * ((size_t)6 * i + k) + d];
      out.push_back(m);
    }
    return out;
  }

 private:
  int N_;
  std::vector<double> c_;
  double dt_;
};

class PolyTrajOptimizer {
 public:
  PolyTrajOptimizer() { dftpav_default_params(&params_); }
  ~PolyTrajOptimizer() {
    drop_batch();
    if (h_) dftpav_destroy(h_);
  }
  PolyTrajOptimizer(const PolyTrajOptimizer &) = delete;
  PolyTrajOptimizer &operator=(const PolyTrajOptimizer &) = delete;

  // traj_optimizer.h:100 — the ros::NodeHandle only served debug publishers
  void setParam(const dftpav_params &p) {
    params_ = p;
    drop_batch();
    if (h_) {
      dftpav_destroy(h_);
      h_ = nullptr;
    }
  }
  // traj_optimizer.h:108 — non-owning, must outlive OptimizeTrajectory like the reference's raw pointer
  void setSurroundTrajs(const SurroundSet *s) { surround_ = s; }
  int get_traj_resolution_() const { return params_.traj_resolution; }        // traj_optimizer.h:113
  int get_destraj_resolution_() const { return params_.des_traj_resolution; } // traj_optimizer.h:114
  const std::vector<MinJerkOptView> *getMinJerkOptPtr() const { return &mjo_; } // traj_optimizer.h:112
  // extras of the new build.  setReferenceOrder(true): solves run with every sum in the order the reference executes it and
  // return the bits of the reference's program with sequential reductions, no FMA and -- gear shifts, moving obstacles --
  // correctly rounded libm calls (include/dftpav_hip.h: dftpav_batch_set_order states the contract); where the layout is
  // outside its limits the throughput order runs and last_order() says so
  void setReferenceOrder(bool on) { want_reference_order_ = on; }
  int last_order() const { return order_; } // DFTPAV_ORDER_* of the last solve
  // status of the last solve
  int last_status() const { return status_; }
  double last_cost() const { return cost_; }
  int last_iterations() const { return iters_; }
  int last_error() const { return err_; } // DFTPAV_E_* of the last call (0 if it reached the solver)
  // the solved problem stays on the device until the next call, for the steps that consume it there
  // (TrajPlannerSteps::CheckCollision / GetStates); NULL before the first successful upload
  dftpav_batch *solved_batch() const { return batch_; }
  dftpav_handle *handle() const { return h_; }

  // traj_optimizer.h:118-120
  bool OptimizeTrajectory(const std::vector<Mat> &iniStates, const std::vector<Mat> &finStates,
                          std::vector<Mat> &initInnerPts, const std::vector<double> &initTs,
                          std::vector<std::vector<Mat>> &hPoly_container, std::vector<int> singuls, double now,
                          double help_eps) {
    err_ = DFTPAV_OK;
    drop_batch();
    const int M = (int)initInnerPts.size();
    if ((int)initTs.size() != M || M < 1) return fail(DFTPAV_E_INVALID); // traj_optimizer.cpp:26-29
    for (double T : initTs)
      if (T < params_.mini_T) return fail(DFTPAV_E_MINI_T); // traj_optimizer.cpp:30-33
    std::vector<int> piece_nums(M);
    int H = 0;
    for (int i = 0; i < M; i++) {
      if (initInnerPts[i].cols == 0) return fail(DFTPAV_E_ONE_PIECE); // traj_optimizer.cpp:38-41
      piece_nums[i] = initInnerPts[i].cols + 1;
      size_t need = (size_t)(piece_nums[i] - 2) * (params_.traj_resolution + 1) + 2 * (params_.des_traj_resolution + 1);
      if (hPoly_container[i].size() != need) return fail(DFTPAV_E_INVALID); // traj_optimizer.cpp:44-48
      for (const Mat &h : hPoly_container[i]) H = h.cols > H ? h.cols : H;
    }
    // flatten (B = 1)
    std::vector<double> ini, fin, inner, cor;
    for (int i = 0; i < M; i++) {
      ini.insert(ini.end(), iniStates[i].a.begin(), iniStates[i].a.end());
      fin.insert(fin.end(), finStates[i].a.begin(), finStates[i].a.end());
      inner.insert(inner.end(), initInnerPts[i].a.begin(), initInnerPts[i].a.end());
      for (const Mat &h : hPoly_container[i]) {
        for (int k = 0; k < H; k++) {
          if (k < h.cols) {
            for (int r = 0; r < 4; r++) cor.push_back(h(r, k));
          } else { // pad with a half-plane that can never be violated
            cor.push_back(1.0); cor.push_back(0.0); cor.push_back(1.0e9); cor.push_back(0.0);
          }
        }
      }
    }
    if (!h_) {
      int rc = dftpav_create(&params_, 0, &h_);
      if (rc != DFTPAV_OK) return fail(rc); // no GPU: fails, never computes on the host
    }
    // moving obstacles
    {
      dftpav_surround s{};
      std::vector<int> off{0};
      std::vector<double> durs, coefs, tot, st;
      if (surround_ && !surround_->empty()) {
        for (const SurroundTraj &t : *surround_) {
          durs.insert(durs.end(), t.durations.begin(), t.durations.end());
          coefs.insert(coefs.end(), t.coeffs.begin(), t.coeffs.end());
          off.push_back((int)durs.size());
          tot.push_back(t.duration);
          st.push_back(t.start_time);
        }
        s.S = (int)surround_->size();
        s.piece_offsets = off.data();
        s.durations = durs.data();
        s.coeffs = coefs.data();
        s.total_duration = tot.data();
        s.start_time = st.data();
      }
      int rc = dftpav_set_surround(h_, s.S ? &s : nullptr);
      if (rc != DFTPAV_OK) return fail(rc);
    }
    dftpav_layout lay{M, piece_nums.data(), singuls.data(), H};
    dftpav_batch_data d{ini.data(), fin.data(), inner.data(), initTs.data(), cor.data(), now, help_eps};
    dftpav_batch *b = nullptr;
    int rc = dftpav_batch_create(h_, &lay, 1, &b);
    if (rc != DFTPAV_OK) return fail(rc);
    rc = dftpav_batch_upload(b, &d);
    order_ = DFTPAV_ORDER_DEVICE;
    if (rc == DFTPAV_OK && want_reference_order_ && dftpav_batch_set_order(b, DFTPAV_ORDER_REFERENCE) == DFTPAV_OK) order_ = DFTPAV_ORDER_REFERENCE;
    if (rc == DFTPAV_OK) rc = dftpav_batch_solve_async(b);
    int success = 0;
    if (rc == DFTPAV_OK) rc = dftpav_batch_results(b, nullptr, &cost_, &status_, &success, &iters_, nullptr, nullptr, nullptr);
    // results stay inside the optimiser until the next call (traj_optimizer.h:91,112)
    mjo_.clear();
    if (rc == DFTPAV_OK) {
      int Ntot = 0;
      for (int N : piece_nums) Ntot += N;
      std::vector<double> c((size_t)12 * Ntot), dt(M);
      rc = dftpav_batch_coeffs(b, c.data(), dt.data());
      int o = 0;
      for (int i = 0; i < M && rc == DFTPAV_OK; i++) {
        mjo_.emplace_back(piece_nums[i], c.data() + (size_t)12 * o, dt[i]);
        o += piece_nums[i];
      }
    }
    if (rc != DFTPAV_OK) {
      dftpav_batch_destroy(b);
      return fail(rc);
    }
    batch_ = b;
    return success != 0;
  }

 private:
  bool fail(int code) {
    err_ = code;
    return false;
  }
  void drop_batch() {
    if (batch_) dftpav_batch_destroy(batch_);
    batch_ = nullptr;
  }
  dftpav_batch *batch_ = nullptr;
  dftpav_params params_;
  dftpav_handle *h_ = nullptr;
  const SurroundSet *surround_ = nullptr;
  std::vector<MinJerkOptView> mjo_;
  int status_ = 0, iters_ = 0, err_ = 0, order_ = DFTPAV_ORDER_DEVICE;
  bool want_reference_order_ = false;
  double cost_ = 0.0;
};

} // namespace plan_manage
