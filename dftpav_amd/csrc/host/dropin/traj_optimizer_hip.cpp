// traj_optimizer_hip.cpp — the drop-in itself: the live path of Dftpav's
//   traj_planner/src/traj_optimizer.cpp
// re-implemented over the C-ABI of include/dftpav_hip.h, against the reference's OWN, UNMODIFIED class declaration
//   traj_planner/include/plan_manage/traj_optimizer.h:24-250
// A Dftpav maintainer builds this file INSTEAD of traj_optimizer.cpp and links libdftpav_hip.so; nothing else of the planner
// changes: TrajPlanner::RunMINCOParking (traj_manager.cpp:604-625) keeps calling
//   setSurroundTrajs(&surround_trajs)                                  traj_manager.cpp:604
//   OptimizeTrajectory(iniStates, finStates, innerPts, Ts, hPolys, singuls, now, eps)   traj_manager.cpp:608-610
//   getMinJerkOptPtr()->at(i).getTraj(singul)                          traj_manager.cpp:618-625
// and gets the answers from the GPU.  In this repository the file is compiled by oracle/Makefile.dropin against the header as
// it lies under /root/reference (plus the interface stand-ins of oracle/ref_shim for Eigen / ROS / protobuf, which this image
// lacks) and run by tests/test_gpu_dropin.py: the reference's PolyTrajOptimizer OBJECT, GPU-backed, returns the bits the
// reference build (oracle/_ref) returns.
//
// What the class declares and this file defines (the members the live path reaches):
//   setParam             traj_optimizer.cpp:1709-1770   config -> members, footprint inflation; + creates the HIP handle
//   setSurroundTrajs     traj_optimizer.cpp (setter)    + forwards the obstacle polynomials to dftpav_set_surround
//   OptimizeTrajectory   traj_optimizer.cpp:7-202       flatten the Eigen containers, solve on the device, refill
//                                                       jerkOpt_container so that getMinJerkOptPtr() keeps working
//   costFunctionCallback traj_optimizer.cpp:206-350     the unit-test cut: one evaluation on the device (dftpav_batch_eval)
// The class has no spare member for a handle and its header stays untouched, so the device state of an object lives in a
// registry keyed by the object's address.  ~PolyTrajOptimizer is inline and empty in the header: an entry is released when
// setParam is called again on the same address or at process exit.
//
// Floating-point order: DFTPAV_ORDER_REFERENCE unless the environment says DFTPAV_DROPIN_ORDER=device -- the planner solves
// one trajectory per 20 Hz cycle (traj_server_ros.cpp:100), and the reference order runs the reference's floating-point program:
// bit-equal to the reference's statements with SEQUENTIAL reductions (what the CPU restatement executes; for gear shifts / moving
// obstacles with correctly rounded cos / sin / exp / log / pow in place of libm's host-dependent ones).  A planner built against a
// real, vectorising Eigen reduces its dot products in another order and differs from this -- and from any other build of itself --
// by a perturbation of the last bit, which this chaotic solver amplifies to a different, statistically equal answer (measured:
// profiles/r05_eigen_redux_cpu.json).
//
// One trajectory per call leaves 255 of 256 CUs idle.  Opt-in, DFTPAV_DROPIN_RESTARTS=K (2 .. 1024): the call's problem goes
// into slot 0 of a batch of K, slots 1 .. K-1 are seeded restarts of it (dftpav_sample_restarts: waypoints moved by N(0, 0.3^2) m,
// segment durations scaled by U[0.8, 1.25] -- SURVEY.md section 8(d) "Restarts"), all solved in the same launch; with a map given
// (dftpav_dropin_set_map) the candidates are re-checked for collisions on the device (CheckReplan's loop, dftpav_batch_validate).
// The best successful, collision-free candidate fills jerkOpt_container; slot 0 -- the reference's own solve, same bits as
// without restarts -- stays readable through dftpav_dropin_last_solve, the choice through dftpav_dropin_last_choice.
// (every standard header the reference's headers pull in comes first: the access lift below must not reach libstdc++)
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

// MinJerkOpt keeps its coefficients private and has no setter (poly_traj_utils.hpp:861-877); a maintainer adds a three-line
// `setSolution(c, dT)`.  To leave the reference's headers untouched the access specifier is lifted for this one translation
// unit instead (it changes no layout and no code -- the same device oracle/ref_driver.cpp uses).
#define private public
#include "plan_manage/traj_optimizer.h"
#undef private

#include "dftpav_hip.h"

namespace {

struct Backend {
  dftpav_handle *h = nullptr;
  dftpav_batch *b = nullptr;
  dftpav_params prm{};
  // layout of the cached batch
  std::vector<int> piece_nums, singuls;
  int H = 0;
  int S_at_create = -1;
  // the last solve, as lbfgs_optimize leaves it in OptimizeTrajectory's locals (traj_optimizer.cpp:159-175)
  std::vector<double> x;
  double final_cost = 0.0;
  int status = 0, success = 0, iters = 0, evals = 0, order = DFTPAV_ORDER_REFERENCE; // order: the one the last solve RAN in
  int order_wanted = -1;            // the order the cached batch was created for (the cache key)
  bool ref_unsupported = false;     // this layout is outside the reference-order kernel's limits: do not ask again
  bool solved = false;
  // restarts (DFTPAV_DROPIN_RESTARTS): size of the cached batch; the candidate OptimizeTrajectory returned and what it cost
  int K = 1, chosen = 0, n_success = 0, n_colliding = 0;
  double chosen_cost = 0.0, solve_ms = 0.0;
  bool have_map = false;
  ~Backend() {
    if (b) dftpav_batch_destroy(b);
    if (h) dftpav_destroy(h);
  }
};

std::mutex g_mu;
std::map<const void *, Backend *> &registry() {
  static std::map<const void *, Backend *> r;
  return r;
}
Backend *backend_of(const void *obj, bool create) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto &r = registry();
  auto it = r.find(obj);
  if (it != r.end()) return it->second;
  if (!create) return nullptr;
  Backend *be = new Backend();
  r[obj] = be;
  return be;
}
void drop_backend(const void *obj) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto &r = registry();
  auto it = r.find(obj);
  if (it != r.end()) {
    delete it->second;
    r.erase(it);
  }
}
int wanted_restarts() {
  const char *e = std::getenv("DFTPAV_DROPIN_RESTARTS");
  const int k = e ? std::atoi(e) : 1;
  return k < 1 ? 1 : (k > 1024 ? 1024 : k);
}
int wanted_order() {
  const char *e = std::getenv("DFTPAV_DROPIN_ORDER");
  return (e && e[0] == 'd') ? DFTPAV_ORDER_DEVICE : DFTPAV_ORDER_REFERENCE;
}

}  // namespace

namespace plan_manage {

// traj_optimizer.cpp:1709-1770.  The members are filled as the reference fills them (other planner code reads them through
// get_traj_resolution_() / getsurroundClearance()); the same values go into dftpav_params, field by field as
// include/dftpav_hip.h documents them, and the device handle is created.
void PolyTrajOptimizer::setParam(ros::NodeHandle nh, planning::minco::Config cfg_) {
  (void)nh; // the reference advertises four debug topics here (:1741-1744); the drop-in publishes nothing
  const auto &o = cfg_.opt_cfg();
  dftpav_params p;
  dftpav_default_params(&p);
  p.traj_resolution = traj_resolution_ = o.traj_resolution();
  p.des_traj_resolution = destraj_resolution_ = o.des_traj_resolution();
  p.wei_obs = wei_obs_ = o.wei_sta_obs();
  p.wei_surround = wei_surround_ = o.wei_dyn_obs();
  p.wei_feas = wei_feas_ = o.wei_feas();
  p.wei_sqrvar = wei_sqrvar_ = o.wei_sqrvar();
  p.wei_time = wei_time_ = o.wei_time();
  p.surround_clearance = surround_clearance_ = o.dyn_obs_clearance();
  p.half_margin = half_margin = o.half_margin();
  p.max_phidot = max_phidot_ = o.max_phidot();
  p.max_forward_vel = max_forward_vel = o.max_forward_vel();
  p.max_backward_vel = max_backward_vel = o.max_backward_vel();
  p.max_forward_cur = max_forward_cur = o.max_forward_cur();
  p.max_backward_cur = max_backward_cur = o.max_backward_cur();
  p.max_forward_acc = max_forward_acc = o.max_forward_acc();
  p.max_backward_acc = max_backward_acc = o.max_backward_acc();
  p.max_latacc = max_latacc_ = o.max_latacc();
  GearOpt = o.gearopt();
  p.gear_opt = GearOpt ? 1 : 0;
  p.lbfgs_mem_size = memsize = o.lbfgs_memsize();
  p.lbfgs_past = past = o.lbfgs_past();
  p.lbfgs_delta = delta = o.lbfgs_delta();
  p.mini_T = mini_T = o.mini_t();
  p.non_sinv = non_sinv; // in-class initialiser, traj_optimizer.h:68
  // the library inflates the RAW vehicle by 2 * half_margin itself, as the reference does at :1746-1747
  p.veh_width = veh_param_.width();
  p.veh_length = veh_param_.length();
  p.veh_wheel_base = veh_param_.wheel_base();
  p.veh_d_cr = veh_param_.d_cr();
  B_h << 0, -1, 1, 0;
  veh_param_.set_width(veh_param_.width() + 2 * half_margin);
  veh_param_.set_length(veh_param_.length() + 2 * half_margin);
  L_ = veh_param_.wheel_base();

  drop_backend(this);
  Backend *be = backend_of(this, true);
  be->prm = p;
  int device = 0;
  if (const char *e = std::getenv("DFTPAV_DROPIN_DEVICE")) device = std::atoi(e);
  const int rc = dftpav_create(&p, device, &be->h);
  if (rc != DFTPAV_OK) {
    ROS_ERROR("dftpav_create failed: %d", rc); // no CPU path: OptimizeTrajectory will return false
    be->h = nullptr;
  }
}

void PolyTrajOptimizer::setSurroundTrajs(plan_utils::SurroundTrajData *surround_trajs_ptr) {
  surround_trajs_ = surround_trajs_ptr;
  Backend *be = backend_of(this, false);
  if (!be || !be->h) return;
  // LocalTrajData (traj_container.hpp:28-38): the fitted polynomial trajectory of one moving obstacle, its duration and clock
  std::vector<int> offs{0};
  std::vector<double> durs, coeffs, total, start;
  if (surround_trajs_ptr)
    for (size_t s = 0; s < surround_trajs_ptr->size(); s++) {
      const plan_utils::LocalTrajData &d = surround_trajs_ptr->at(s);
      const int np = d.traj.getPieceNum();
      for (int q = 0; q < np; q++) {
        const plan_utils::Piece &pc = d.traj[q];
        durs.push_back(pc.getDuration());
        const plan_utils::CoefficientMat &cm = pc.getCoeffMat(); // 2 x 6, column 0 multiplies t^5 (poly_traj_utils.hpp:993)
        coeffs.insert(coeffs.end(), cm.data(), cm.data() + 12);
      }
      offs.push_back(offs.back() + np);
      total.push_back(d.duration);
      start.push_back(d.start_time);
    }
  dftpav_surround S;
  S.S = (int)total.size();
  S.piece_offsets = offs.data();
  S.durations = durs.data();
  S.coeffs = coeffs.data();
  S.total_duration = total.data();
  S.start_time = start.data();
  const int rc = dftpav_set_surround(be->h, &S);
  if (rc != DFTPAV_OK) ROS_ERROR("dftpav_set_surround: %d %s", rc, dftpav_last_error(be->h));
}

// traj_optimizer.cpp:7-202
bool PolyTrajOptimizer::OptimizeTrajectory(const std::vector<Eigen::MatrixXd> &iniStates, const std::vector<Eigen::MatrixXd> &finStates,
                                           std::vector<Eigen::MatrixXd> &initInnerPts, const Eigen::VectorXd &initTs,
                                           std::vector<std::vector<Eigen::MatrixXd>> &hPoly_container, std::vector<int> singuls, double now,
                                           double help_eps) {
  Backend *be = backend_of(this, false);
  if (!be || !be->h) {
    ROS_ERROR("dftpav: no device handle (setParam not called, or no HIP device)");
    return false;
  }
  be->solved = false;
  trajnum = (int)initInnerPts.size();
  const int M = trajnum;
  // the size checks of :26-48 that concern the containers themselves; the rest (piece counts, corridor sizes, mini_T) is
  // checked by the library, which refuses with the corresponding error code
  if (M < 1 || (int)initTs.size() != M || (int)iniStates.size() != M || (int)finStates.size() != M || (int)hPoly_container.size() != M ||
      (int)singuls.size() != M) {
    ROS_ERROR("initTs.size()!=trajnum");
    return false;
  }
  std::vector<int> piece_nums(M);
  std::vector<double> ini, fin, inner, cor;
  int H = 0;
  for (int i = 0; i < M; i++) {
    piece_nums[i] = (int)initInnerPts[i].cols() + 1;
    if (iniStates[i].size() != 6 || finStates[i].size() != 6) return false;
    // Eigen is column-major: data() of a 2 x 3 state matrix is (p_x, p_y, v_x, v_y, a_x, a_y), the ABI layout
    ini.insert(ini.end(), iniStates[i].data(), iniStates[i].data() + 6);
    fin.insert(fin.end(), finStates[i].data(), finStates[i].data() + 6);
    inner.insert(inner.end(), initInnerPts[i].data(), initInnerPts[i].data() + initInnerPts[i].size());
    for (const Eigen::MatrixXd &hp : hPoly_container[i]) { // 4 x H, each column (n_x, n_y, p_x, p_y), traj_optimizer.h:77
      if (H == 0) H = (int)hp.cols();
      if ((int)hp.cols() != H || hp.rows() != 4) {
        ROS_ERROR("dftpav: corridor polygons of different sizes");
        return false;
      }
      cor.insert(cor.end(), hp.data(), hp.data() + 4 * H);
    }
  }
  // the members the rest of the class would have (debug read-outs, dead on the live path, keep consistent values)
  piece_num_container = piece_nums;
  singul_container = singuls;
  iniState_container = iniStates;
  finState_container = finStates;
  t_now_ = now;
  epis = help_eps;

  const int S_now = surround_trajs_ ? (int)surround_trajs_->size() : 0;
  const int order = wanted_order();
  const int K = wanted_restarts();
  if (!be->b || be->piece_nums != piece_nums || be->singuls != singuls || be->H != H || be->S_at_create != S_now || be->order_wanted != order ||
      be->K != K) {
    if (be->b) dftpav_batch_destroy(be->b);
    be->b = nullptr;
    dftpav_layout lay{M, piece_nums.data(), singuls.data(), H};
    int rc = dftpav_batch_create(be->h, &lay, K, &be->b);
    if (rc != DFTPAV_OK) {
      ROS_ERROR("dftpav_batch_create: %d %s", rc, dftpav_last_error(be->h));
      be->b = nullptr;
      return false;
    }
    be->piece_nums = piece_nums;
    be->singuls = singuls;
    be->H = H;
    be->S_at_create = S_now;
    be->order_wanted = order;
    be->ref_unsupported = false;
    be->K = K;
  }
  std::vector<double> Ts(initTs.data(), initTs.data() + M);
  if (K > 1) { // slot 0: the call's problem; slots 1 .. K-1: its seeded restarts, same boundary states and corridor
    const int n_inner = (int)inner.size();
    std::vector<double> rin((size_t)K * n_inner), rts((size_t)K * M);
    unsigned long long seed = 20240;
    if (const char *e = std::getenv("DFTPAV_DROPIN_SEED")) seed = std::strtoull(e, nullptr, 10);
    int rc = dftpav_sample_restarts(be->h, inner.data(), Ts.data(), 1, K, n_inner, M, 0.3, 0.8, 1.25, seed, rin.data(), rts.data());
    if (rc != DFTPAV_OK) {
      ROS_ERROR("dftpav_sample_restarts: %d %s", rc, dftpav_last_error(be->h));
      return false;
    }
    // a restart's duration must stay ABOVE mini_T: RealT2VirtualT (traj_optimizer.cpp:365-367) divides by T - mini_T, so a duration
    // clamped to exactly mini_T would start from a virtual time of -inf (a wasted slot at best)
    for (int k = 1; k < K; k++)
      for (int i = 0; i < M; i++) rts[(size_t)k * M + i] = std::max(rts[(size_t)k * M + i], mini_T * (1.0 + 1.0e-3) + 1.0e-6);
    auto tile = [&](std::vector<double> &v) {
      const size_t n1 = v.size();
      v.resize(n1 * K);
      for (int k = 1; k < K; k++) std::copy(v.begin(), v.begin() + n1, v.begin() + (size_t)k * n1);
    };
    tile(ini);
    tile(fin);
    tile(cor);
    inner.swap(rin);
    Ts.swap(rts);
  }
  dftpav_batch_data d;
  std::memset(&d, 0, sizeof(d));
  d.ini_states = ini.data();
  d.fin_states = fin.data();
  d.inner_pts = inner.data();
  d.init_Ts = Ts.data();
  d.corridor = cor.data();
  d.t_now = now;
  d.help_eps = help_eps;
  int rc = dftpav_batch_upload(be->b, &d);
  be->order = order;
  if (rc == DFTPAV_OK && order == DFTPAV_ORDER_REFERENCE && !be->ref_unsupported) {
    rc = dftpav_batch_set_order(be->b, DFTPAV_ORDER_REFERENCE);
    if (rc == DFTPAV_E_UNSUPPORTED) { // a layout outside the reference-order kernel's limits: the throughput order solves it
      rc = DFTPAV_OK;                 // (remembered with the cached batch: the 20 Hz cycle neither recreates it nor asks again)
      be->ref_unsupported = true;
    }
  }
  if (be->ref_unsupported) be->order = DFTPAV_ORDER_DEVICE;
  if (rc == DFTPAV_OK) rc = dftpav_batch_solve_async(be->b);
  dftpav_layout lay{M, piece_nums.data(), singuls.data(), H};
  const int n = dftpav_num_vars(&lay);
  variable_num_ = n;
  be->x.assign(n, 0.0);
  std::vector<double> xs((size_t)K * n), costs(K);
  std::vector<int> status(K), success(K), iters(K), evals(K), colliding(K, 0), first(K, -1);
  if (rc == DFTPAV_OK) rc = dftpav_batch_results(be->b, xs.data(), costs.data(), status.data(), success.data(), iters.data(), evals.data(), nullptr, nullptr);
  if (rc == DFTPAV_OK && K > 1 && be->have_map) rc = dftpav_batch_validate(be->b, 0.05, 0.1, colliding.data(), first.data()); // traj_server_ros.cpp:385-397
  if (rc != DFTPAV_OK) {
    ROS_ERROR("dftpav: %d %s", rc, dftpav_last_error(be->h));
    return false;
  }
  // slot 0 is the reference's own solve
  std::copy(xs.begin(), xs.begin() + n, be->x.begin());
  be->final_cost = costs[0];
  be->status = status[0];
  be->success = success[0];
  be->iters = iters[0];
  be->evals = evals[0];
  iter_num_ = be->iters;
  be->solved = true;
  // the candidate that is returned: the cheapest successful one that does not collide (slot 0 when there is none, or no restarts).
  // WITHOUT a map the restarts cannot be re-checked (their waypoints were moved by N(0, 0.3^2) m off the collision-free front-end
  // path): slot 0 is kept whenever it succeeded, a restart only stands in for a failed slot 0.
  int chosen = 0;
  const bool unchecked = K > 1 && !be->have_map;
  be->n_success = be->n_colliding = 0;
  for (int k = 0; k < K; k++) {
    be->n_success += success[k] != 0;
    be->n_colliding += colliding[k] != 0;
    const bool ok_k = success[k] != 0 && colliding[k] == 0, ok_c = success[chosen] != 0 && colliding[chosen] == 0;
    if (ok_k && (!ok_c || (costs[k] < costs[chosen] && !(unchecked && chosen == 0)))) chosen = k;
  }
  iter_num_ = iters[chosen]; // (the member describes what is returned; slot 0's own figures stay in dftpav_dropin_last_solve)
  be->chosen = chosen;
  be->chosen_cost = costs[chosen];
  {
    float ms = 0.0f;
    (void)dftpav_batch_last_solve_ms(be->b, &ms);
    be->solve_ms = ms;
  }

  // jerkOpt_container: what getMinJerkOptPtr() hands to traj_manager.cpp:618-625 -- per gear segment the coefficients and the
  // piece duration of the solution, regenerated on the device from the final x (dftpav_batch_coeffs)
  int Ntot = 0;
  for (int i = 0; i < M; i++) Ntot += piece_nums[i];
  std::vector<double> cf((size_t)12 * Ntot * K), dt((size_t)M * K);
  rc = dftpav_batch_coeffs(be->b, cf.data(), dt.data());
  if (rc != DFTPAV_OK) {
    ROS_ERROR("dftpav_batch_coeffs: %d %s", rc, dftpav_last_error(be->h));
    return false;
  }
  jerkOpt_container.clear();
  jerkOpt_container.resize(M);
  size_t off = (size_t)12 * Ntot * chosen;
  for (int i = 0; i < M; i++) {
    plan_utils::MinJerkOpt &mj = jerkOpt_container[i];
    const int N = piece_nums[i];
    mj.N = N;
    mj.c.resize(6 * N, 2);
    for (int r = 0; r < 6 * N; r++)
      for (int dd = 0; dd < 2; dd++) mj.c(r, dd) = cf[off + 2 * r + dd];
    off += (size_t)12 * N;
    mj.t(0) = 1.0;
    mj.t(1) = dt[(size_t)M * chosen + i];
    mj.headPVA = iniStates[i];
    mj.tailPVA = finStates[i];
  }
  return success[chosen] != 0;
}

// traj_optimizer.cpp:206-350 -- lbfgs_evaluate_t (lbfgs.hpp:200-202): one evaluation of the cost and its gradient at x for the
// problem the last OptimizeTrajectory installed, on the device
double PolyTrajOptimizer::costFunctionCallback(void *func_data, const Eigen::VectorXd &x, Eigen::VectorXd &grad) {
  PolyTrajOptimizer *opt = reinterpret_cast<PolyTrajOptimizer *>(func_data);
  Backend *be = backend_of(opt, false);
  double f = 0.0;
  if (!be || !be->b || dftpav_batch_eval(be->b, x.data(), &f, grad.data()) != DFTPAV_OK) return std::nan("");
  return f;
}

}  // namespace plan_manage

// What OptimizeTrajectory keeps in locals and the class cannot return (solution vector, final cost, solver status, counts):
// read by the tests' driver.  Returns 0 if the object has no finished solve.
extern "C" int dftpav_dropin_last_solve(const void *optimizer, int *n, const double **x, double *final_cost, int *status, int *iters, int *evals,
                                        int *order) {
  Backend *be = backend_of(optimizer, false);
  if (!be || !be->solved) return 0;
  if (n) *n = (int)be->x.size();
  if (x) *x = be->x.data();
  if (final_cost) *final_cost = be->final_cost;
  if (status) *status = be->status;
  if (iters) *iters = be->iters;
  if (evals) *evals = be->evals;
  if (order) *order = be->order;
  return 1;
}

// With DFTPAV_DROPIN_RESTARTS=K: which candidate the last OptimizeTrajectory returned (0 = the call's own problem), its cost, how
// many of the K succeeded / were rejected by the collision re-check, and the device time of the launch.  Returns K (0: no solve).
extern "C" int dftpav_dropin_last_choice(const void *optimizer, int *chosen, double *chosen_cost, int *n_success, int *n_colliding, double *solve_ms) {
  Backend *be = backend_of(optimizer, false);
  if (!be || !be->solved) return 0;
  if (chosen) *chosen = be->chosen;
  if (chosen_cost) *chosen_cost = be->chosen_cost;
  if (n_success) *n_success = be->n_success;
  if (n_colliding) *n_colliding = be->n_colliding;
  if (solve_ms) *solve_ms = be->solve_ms;
  return be->K;
}
// The occupancy grid the candidates are re-checked on (the planner has it: TrajPlannerMapItf::GetObstacleMap, map_interface.h:52);
// grid[ix + size_x * iy], 80 = occupied.  Without it the restarts are ranked by cost alone.
extern "C" int dftpav_dropin_set_map(const void *optimizer, const unsigned char *grid, int size_x, int size_y, double resolution, double origin_x,
                                     double origin_y) {
  Backend *be = backend_of(optimizer, false);
  if (!be || !be->h) return DFTPAV_E_INVALID;
  dftpav_grid_map m;
  std::memset(&m, 0, sizeof(m));
  m.cells = grid;
  m.size_x = size_x;
  m.size_y = size_y;
  m.resolution = resolution;
  m.origin_x = origin_x;
  m.origin_y = origin_y;
  const int rc = dftpav_set_grid_map(be->h, &m);
  be->have_map = rc == DFTPAV_OK;
  return rc;
}
