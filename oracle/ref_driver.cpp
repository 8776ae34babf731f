// C entry points of oracle/_ref/libdftpav_ref.so — TEST INFRASTRUCTURE.
//
// This file is the only code of ours inside the reference build: it converts plain arrays to the reference's argument
// types and calls the reference's own functions, compiled UNMODIFIED from /root/reference/src/Plan/traj_planner:
//   src/traj_optimizer.cpp                      PolyTrajOptimizer (OptimizeTrajectory, costFunctionCallback, penalties)
//   include/plan_utils/poly_traj_utils.hpp      BandedSystem, MinJerkOpt, Piece, Trajectory
//   include/geo_utils2d/lbfgs.hpp               lbfgs_optimize, line_search_lewisoverton
// against the interface stand-ins under oracle/ref_shim/ (Eigen, ROS, protobuf config: none is installed here).
// It computes nothing itself.  `#define private public` gives the tests access to costFunctionCallback and the time maps
// (private in traj_optimizer.h:125-148); it changes no layout and no code.
//
// With -DDFTPAV_DROPIN (oracle/Makefile.dropin) the same driver is linked against the DROP-IN's implementation of the class
// (dftpav_amd/csrc/host/dropin/traj_optimizer_hip.cpp over libdftpav_hip.so) instead of the reference's traj_optimizer.cpp:
// the same calls on the same object -- setParam, setSurroundTrajs, OptimizeTrajectory, costFunctionCallback,
// getMinJerkOptPtr -- then run on the GPU.  What OptimizeTrajectory keeps in locals comes from the drop-in's accessor there
// (no lbfgs_optimize call site to observe); the hooks on private helpers the drop-in does not have are left out.
#include <sstream>
#define private public
#include "plan_manage/traj_optimizer.h"
#undef private
#include "ref_hook.h"
#include "dftpav_oracle.h"

using plan_manage::PolyTrajOptimizer;
#ifdef DFTPAV_DROPIN
extern "C" int dftpav_dropin_last_solve(const void *optimizer, int *n, const double **x, double *final_cost, int *status, int *iters, int *evals, int *order);
extern "C" int dftpav_dropin_last_choice(const void *optimizer, int *chosen, double *chosen_cost, int *n_success, int *n_colliding, double *solve_ms);
extern "C" int dftpav_dropin_set_map(const void *optimizer, const unsigned char *grid, int size_x, int size_y, double resolution, double origin_x, double origin_y);
#endif

namespace {
struct Quiet {  // the reference prints its progress on std::cout (traj_optimizer.cpp:34,144,154,168)
  std::streambuf *old;
  std::ostringstream sink;
  Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~Quiet() { std::cout.rdbuf(old); }
};

struct RefCtx {
  PolyTrajOptimizer opt;
  dftpav_params prm;
  std::vector<Eigen::MatrixXd> ini, fin, inner;
  Eigen::VectorXd Ts;
  std::vector<std::vector<Eigen::MatrixXd>> hpolys;
  std::vector<int> singuls;
  plan_utils::SurroundTrajData sur;
  double t_now = 0.0, help_eps = 0.0;
  int n = 0;
  bool solved = false;
  bool success = false;
  dftpav_ref::SolveRecord rec;
};

void set_params(PolyTrajOptimizer &o, const dftpav_params &p) {
  planning::minco::Config cfg;
  planning::minco::OptCfg *c = cfg.mutable_opt_cfg();
  c->set_traj_resolution(p.traj_resolution);
  c->set_des_traj_resolution(p.des_traj_resolution);
  c->set_wei_sta_obs(p.wei_obs);
  c->set_wei_dyn_obs(p.wei_surround);
  c->set_wei_feas(p.wei_feas);
  c->set_wei_sqrvar(p.wei_sqrvar);
  c->set_wei_time(p.wei_time);
  c->set_dyn_obs_clearance(p.surround_clearance);
  c->set_half_margin(p.half_margin);
  c->set_max_phidot(p.max_phidot);
  c->set_max_forward_vel(p.max_forward_vel);
  c->set_max_backward_vel(p.max_backward_vel);
  c->set_max_forward_cur(p.max_forward_cur);
  c->set_max_backward_cur(p.max_backward_cur);
  c->set_max_forward_acc(p.max_forward_acc);
  c->set_max_backward_acc(p.max_backward_acc);
  c->set_max_latacc(p.max_latacc);
  c->set_gearopt(p.gear_opt != 0);
  c->set_lbfgs_memsize(p.lbfgs_mem_size);
  c->set_lbfgs_past(p.lbfgs_past);
  c->set_lbfgs_delta(p.lbfgs_delta);
  c->set_mini_t(p.mini_T);
  // the vehicle is a member with the defaults of semantics.h:66-76; setParam inflates it by half_margin
  o.veh_param_.set_width(p.veh_width);
  o.veh_param_.set_length(p.veh_length);
  o.veh_param_.set_wheel_base(p.veh_wheel_base);
  o.veh_param_.set_d_cr(p.veh_d_cr);
  o.non_sinv = p.non_sinv;  // in-class initialiser 0.24, traj_optimizer.h:68
  o.setParam(ros::NodeHandle(), cfg);
}

lbfgs::lbfgs_parameter_t lbfgs_params_of(const dftpav_params &p) {
  lbfgs::lbfgs_parameter_t q;  // defaults of lbfgs.hpp:15-129
  q.mem_size = p.lbfgs_mem_size;
  q.past = p.lbfgs_past;
  q.delta = p.lbfgs_delta;
  q.g_epsilon = p.lbfgs_g_epsilon;
  q.max_iterations = p.lbfgs_max_iterations;
  q.max_linesearch = p.lbfgs_max_linesearch;
  q.min_step = p.lbfgs_min_step;
  q.max_step = p.lbfgs_max_step;
  q.f_dec_coeff = p.lbfgs_f_dec_coeff;
  q.s_curv_coeff = p.lbfgs_s_curv_coeff;
  q.cautious_factor = p.lbfgs_cautious_factor;
  q.machine_prec = p.lbfgs_machine_prec;
  return q;
}

struct FnBridge {
  oracle_eval_fn fn;
  void *instance;
};
double bridge_eval(void *b, const Eigen::VectorXd &x, Eigen::VectorXd &g) {
  FnBridge *f = static_cast<FnBridge *>(b);
  return f->fn(f->instance, x.data(), g.data(), (int)x.size());
}
}  // namespace

extern "C" {

int ref_abi_version(void) { return 1; }
#ifdef DFTPAV_DROPIN
int ref_is_dropin(void) { return 1; }
// the drop-in's restarts (DFTPAV_DROPIN_RESTARTS): which candidate the last OptimizeTrajectory returned, and the map they are re-checked on
int ref_dropin_last_choice(void *h, int *chosen, double *chosen_cost, int *n_success, int *n_colliding, double *solve_ms) {
  return dftpav_dropin_last_choice(&static_cast<RefCtx *>(h)->opt, chosen, chosen_cost, n_success, n_colliding, solve_ms);
}
int ref_dropin_set_map(void *h, const unsigned char *grid, int size_x, int size_y, double resolution, double origin_x, double origin_y) {
  return dftpav_dropin_set_map(&static_cast<RefCtx *>(h)->opt, grid, size_x, size_y, resolution, origin_x, origin_y);
}
#else
int ref_is_dropin(void) { return 0; }
#endif

/* Builds the argument objects of OptimizeTrajectory (traj_optimizer.h:118-120) from the flat problem. */
void *ref_prepare(const dftpav_params *p, const oracle_problem *pb) {
  RefCtx *c = new RefCtx();
  c->prm = *p;
  set_params(c->opt, *p);
  const int M = pb->M;
  int npts_off = 0, inner_off = 0;
  c->Ts.resize(M);
  for (int i = 0; i < M; ++i) {
    const int N = pb->piece_nums[i];
    Eigen::MatrixXd s0(2, 3), s1(2, 3);
    std::memcpy(s0.data(), pb->ini_states + 6 * i, 6 * sizeof(double));
    std::memcpy(s1.data(), pb->fin_states + 6 * i, 6 * sizeof(double));
    c->ini.push_back(s0);
    c->fin.push_back(s1);
    Eigen::MatrixXd P(2, N - 1 > 0 ? N - 1 : 0);
    if (N > 1) std::memcpy(P.data(), pb->inner_pts + inner_off, sizeof(double) * 2 * (N - 1));
    inner_off += 2 * (N - 1 > 0 ? N - 1 : 0);
    c->inner.push_back(P);
    c->Ts(i) = pb->init_Ts[i];
    c->singuls.push_back(pb->singuls[i]);
    const int npts = (N - 2) * (p->traj_resolution + 1) + 2 * (p->des_traj_resolution + 1);
    std::vector<Eigen::MatrixXd> hp;
    for (int k = 0; k < npts; ++k) {
      Eigen::MatrixXd h(4, pb->H);  // each column (n_x, n_y, p_x, p_y), traj_optimizer.h:77
      std::memcpy(h.data(), pb->corridor + (size_t)(npts_off + k) * pb->H * 4, sizeof(double) * 4 * pb->H);
      hp.push_back(h);
    }
    npts_off += npts;
    c->hpolys.push_back(hp);
  }
  c->t_now = pb->t_now;
  c->help_eps = pb->help_eps;
  if (pb->surround && pb->surround->S > 0) {
    const dftpav_surround *s = pb->surround;
    for (int o = 0; o < s->S; ++o) {
      plan_utils::LocalTrajData d;
      std::vector<double> durs;
      std::vector<plan_utils::CoefficientMat> mats;
      for (int q = s->piece_offsets[o]; q < s->piece_offsets[o + 1]; ++q) {
        durs.push_back(s->durations[q]);
        plan_utils::CoefficientMat cm;
        std::memcpy(cm.data(), s->coeffs + 12 * q, 12 * sizeof(double));
        mats.push_back(cm);
      }
      d.traj = plan_utils::Trajectory(durs, mats, 1);  // obstacles are built with getTraj(1), traj_manager.cpp:726,775
      d.drone_id = o;
      d.traj_id = o;
      d.duration = s->total_duration[o];
      d.start_time = s->start_time[o];
      d.end_time = d.start_time + d.duration;
      d.start_pos = d.traj.getJuncPos(0);
      d.init_angle = 0.0;
      c->sur.push_back(d);
    }
    c->opt.setSurroundTrajs(&c->sur);
  }
  return c;
}

void ref_free(void *h) { delete static_cast<RefCtx *>(h); }

/* PolyTrajOptimizer::OptimizeTrajectory, traj_optimizer.cpp:7-202.  Returns its bool (1/0); the locals it discards are read
 * through the observation hook.  trace != 0 records every evaluation and iteration (ref_trace_*). */
int ref_optimize(void *h, int trace, double *x_out, double *final_cost, int *status, int *iters, int *evals) {
  RefCtx *c = static_cast<RefCtx *>(h);
  Quiet q;
  c->rec = dftpav_ref::SolveRecord();
  c->rec.trace = trace != 0;
  dftpav_ref::current_record() = &c->rec;
  std::vector<Eigen::MatrixXd> inner = c->inner;
  std::vector<std::vector<Eigen::MatrixXd>> hp = c->hpolys;
  bool ok = c->opt.OptimizeTrajectory(c->ini, c->fin, inner, c->Ts, hp, c->singuls, c->t_now, c->help_eps);
  dftpav_ref::current_record() = nullptr;
#ifdef DFTPAV_DROPIN
  {  // the drop-in has no lbfgs_optimize call site to observe: its accessor returns what the device solve left
    int n = 0, st = 0, it = 0, ev = 0, order = 0;
    const double *xs = nullptr;
    double fc = 0.0;
    c->solved = dftpav_dropin_last_solve(&c->opt, &n, &xs, &fc, &st, &it, &ev, &order) != 0;
    c->success = ok;
    c->n = c->opt.variable_num_;
    if (!c->solved) return ok ? 1 : 0;
    if (x_out) std::memcpy(x_out, xs, sizeof(double) * n);
    if (final_cost) *final_cost = fc;
    if (status) *status = st;
    if (iters) *iters = it;
    if (evals) *evals = ev;
    return ok ? 1 : 0;
  }
#endif
  c->solved = c->rec.have;
  c->success = ok;
  c->n = c->opt.variable_num_;
  if (!c->rec.have) return ok ? 1 : 0;  // refused before the solve (traj_optimizer.cpp:26-48)
  if (x_out) std::memcpy(x_out, c->rec.x.data(), sizeof(double) * c->rec.x.size());
  if (final_cost) *final_cost = c->rec.f;
  if (status) *status = c->rec.ret;
  // k at exit: a failed line search leaves before the progress report of its iteration (lbfgs.hpp:604-611)
  if (iters) *iters = (c->rec.ret < 0 && c->rec.ret != lbfgs::LBFGSERR_MAXIMUMITERATION) ? c->rec.last_progress_k + 1 : c->rec.last_progress_k;
  if (evals) *evals = c->rec.evals;
  return ok ? 1 : 0;
}
int ref_solved(void *h) { return static_cast<RefCtx *>(h)->solved ? 1 : 0; }
int ref_num_vars(void *h) { return static_cast<RefCtx *>(h)->opt.variable_num_; }

/* trace of the last ref_optimize(trace=1): sizes, then copies */
int ref_trace_num_evals(void *h) { return (int)static_cast<RefCtx *>(h)->rec.eval_f.size(); }
int ref_trace_num_iters(void *h) { return (int)static_cast<RefCtx *>(h)->rec.iter_fx.size(); }
void ref_trace_evals(void *h, double *x, double *g, double *f) {
  RefCtx *c = static_cast<RefCtx *>(h);
  if (x) std::memcpy(x, c->rec.eval_x.data(), sizeof(double) * c->rec.eval_x.size());
  if (g) std::memcpy(g, c->rec.eval_g.data(), sizeof(double) * c->rec.eval_g.size());
  if (f) std::memcpy(f, c->rec.eval_f.data(), sizeof(double) * c->rec.eval_f.size());
}
void ref_trace_iters(void *h, double *x, double *g, double *fx, double *step, int *k, int *ls) {
  RefCtx *c = static_cast<RefCtx *>(h);
  if (x) std::memcpy(x, c->rec.iter_x.data(), sizeof(double) * c->rec.iter_x.size());
  if (g) std::memcpy(g, c->rec.iter_g.data(), sizeof(double) * c->rec.iter_g.size());
  if (fx) std::memcpy(fx, c->rec.iter_fx.data(), sizeof(double) * c->rec.iter_fx.size());
  if (step) std::memcpy(step, c->rec.iter_step.data(), sizeof(double) * c->rec.iter_step.size());
  if (k) std::memcpy(k, c->rec.iter_k.data(), sizeof(int) * c->rec.iter_k.size());
  if (ls) std::memcpy(ls, c->rec.iter_ls.data(), sizeof(int) * c->rec.iter_ls.size());
}

/* PolyTrajOptimizer::costFunctionCallback, traj_optimizer.cpp:206-350, at an arbitrary x.  The set-up it depends on
 * (normalised corridor, clamped boundary states, MinJerkOpt::reset) lives in members that OptimizeTrajectory fills, so a
 * solve must have run on this object first (ref_optimize). */
double ref_eval(void *h, const double *x, double *g) {
  RefCtx *c = static_cast<RefCtx *>(h);
  if (!c->solved) return std::nan("");
  const int n = c->opt.variable_num_;
  Eigen::VectorXd xv(n), gv(n);
  std::memcpy(xv.data(), x, sizeof(double) * n);
  double f = PolyTrajOptimizer::costFunctionCallback(&c->opt, xv, gv);
  std::memcpy(g, gv.data(), sizeof(double) * n);
  return f;
}
/* coefficients the optimiser's MinJerkOpt objects hold after the last evaluation: [Ntot][6][2], piece_dt [M] */
void ref_last_coeffs(void *h, double *coeffs, double *piece_dt) {
  RefCtx *c = static_cast<RefCtx *>(h);
  size_t off = 0;
  for (size_t i = 0; i < c->opt.jerkOpt_container.size(); ++i) {
    const Eigen::MatrixXd &m = c->opt.jerkOpt_container[i].getCoeffs();
    for (Eigen::Index r = 0; r < m.rows(); ++r)
      for (int d = 0; d < 2; ++d) coeffs[off + 2 * r + d] = m(r, d);
    off += 2 * m.rows();
    piece_dt[i] = c->opt.jerkOpt_container[i].getDt();
  }
}

/* lbfgs::lbfgs_optimize (lbfgs.hpp:440-751) on a caller-supplied function; callbacks NULL as at traj_optimizer.cpp:163-164 */
int ref_lbfgs(int n, double *x, double *f, oracle_eval_fn fn, void *instance, const dftpav_params *p, int *iters, int *evals) {
  Eigen::VectorXd xv(n);
  std::memcpy(xv.data(), x, sizeof(double) * n);
  FnBridge b{fn, instance};
  dftpav_ref::SolveRecord rec;
  dftpav_ref::current_record() = &rec;
  int ret = lbfgs::lbfgs_optimize_observed(xv, *f, bridge_eval, nullptr, nullptr, &b, lbfgs_params_of(*p));
  dftpav_ref::current_record() = nullptr;
  std::memcpy(x, xv.data(), sizeof(double) * n);
  if (iters) *iters = (ret < 0 && ret != lbfgs::LBFGSERR_MAXIMUMITERATION) ? rec.last_progress_k + 1 : rec.last_progress_k;
  if (evals) *evals = rec.evals;
  return ret;
}

/* BandedSystem (poly_traj_utils.hpp:727-853): A is n x n dense row-major on input, bandwidths p, q; the right-hand sides
 * b [n][m] row-major are overwritten by the solution of A x = b (adj == 0) or A^T x = b (adj != 0). */
void ref_banded_solve(int n, int p, int q, const double *A, int m, double *b, int adj) {
  plan_utils::BandedSystem sys;
  sys.create(n, p, q);
  for (int i = 0; i < n; ++i)
    for (int j = std::max(0, i - p); j <= std::min(n - 1, i + q); ++j) sys(i, j) = A[(size_t)i * n + j];
  sys.factorizeLU();
  Eigen::MatrixXd B(n, m);
  for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) B(i, j) = b[(size_t)i * m + j];
  if (adj) sys.solveAdj(B); else sys.solve(B);
  for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) b[(size_t)i * m + j] = B(i, j);
  sys.destroy();
}

/* MinJerkOpt (poly_traj_utils.hpp:855-1095): reset, generate, initSmGradCost, getTrajJerkCost; with gdC_add != NULL that
 * array [6N][2] is added to gdC before calGrads_PT, whose outputs are returned.  Layouts as oracle_minco_generate. */
double ref_minco(int N, const double *inPs, double dT, const double *head, const double *tail, double *coeffs,
                 const double *gdC_add, double *gdP, double *gdHead, double *gdTail, double *gdT) {
  plan_utils::MinJerkOpt o;
  o.reset(N);
  Eigen::MatrixXd P(2, N - 1), h(2, 3), t(2, 3);
  std::memcpy(P.data(), inPs, sizeof(double) * 2 * (N - 1));
  std::memcpy(h.data(), head, sizeof(double) * 6);
  std::memcpy(t.data(), tail, sizeof(double) * 6);
  o.generate(P, dT, h, t);
  o.initSmGradCost();
  double e = o.getTrajJerkCost();
  const Eigen::MatrixXd &c = o.getCoeffs();
  if (coeffs) for (int r = 0; r < 6 * N; ++r) for (int d = 0; d < 2; ++d) coeffs[2 * r + d] = c(r, d);
  if (gdP) {
    if (gdC_add) for (int r = 0; r < 6 * N; ++r) for (int d = 0; d < 2; ++d) o.get_gdC()(r, d) += gdC_add[2 * r + d];
    o.calGrads_PT();
    Eigen::MatrixXd a = o.get_gdP(), b = o.get_gdHead(), cc = o.get_gdTail();
    std::memcpy(gdP, a.data(), sizeof(double) * 2 * (N - 1));
    std::memcpy(gdHead, b.data(), sizeof(double) * 6);
    std::memcpy(gdTail, cc.data(), sizeof(double) * 6);
    *gdT = o.get_gdT();
  }
  return e;
}

/* Trajectory / Piece evaluators used by the moving-obstacle term (poly_traj_utils.hpp:77-112,179-211,510-603) on obstacle
 * `o` of a prepared problem: out = {pos(2), dsigma(2), ddsigma(2), R(4 col-major), Rdot(4 col-major)} at time t */
void ref_surround_state(void *h, int o, double t, double *out) {
  RefCtx *c = static_cast<RefCtx *>(h);
  const plan_utils::Trajectory &tr = c->sur[o].traj;
  Eigen::Vector2d p = tr.getPos(t), v = tr.getdSigma(t), a = tr.getddSigma(t);
  Eigen::Matrix2d R = tr.getR(t), Rd = tr.getRdot(t);
  out[0] = p(0); out[1] = p(1); out[2] = v(0); out[3] = v(1); out[4] = a(0); out[5] = a(1);
  std::memcpy(out + 6, R.data(), 4 * sizeof(double));
  std::memcpy(out + 10, Rd.data(), 4 * sizeof(double));
}

#ifndef DFTPAV_DROPIN
/* positiveSmoothedL1, traj_optimizer.cpp:783-806 */
void ref_smoothed_l1(const dftpav_params *p, double x, double *f, double *df) {
  PolyTrajOptimizer o;
  o.positiveSmoothedL1(x, *f, *df);
}
/* VirtualT2RealT / RealT2VirtualT (traj_optimizer.cpp:360-379) are member templates defined inside the .cpp: no symbol to call.
 * They are pinned through their effects: the first evaluation point of a traced solve is the packed x0 (RealT2VirtualT,
 * :104), and ref_last_coeffs returns T_i / N_i of the evaluated x (VirtualT2RealT, :229,288). */
/* VirtualTGradCost (scalar overload), traj_optimizer.cpp:405-419 */
void ref_virtual_T_grad_cost(const dftpav_params *p, double RT, double VT, double gdRT, double *gdVT, double *costT) {
  PolyTrajOptimizer o;
  o.wei_time_ = p->wei_time;
  o.VirtualTGradCost(RT, VT, gdRT, *gdVT, *costT);
}
/* log_sum_exp, traj_optimizer.cpp:1686-1707: dists [n] is overwritten with the exponentials as the reference does */
double ref_log_sum_exp(double alpha, int n, double *dists, double *exp_sum) {
  PolyTrajOptimizer o;
  Eigen::VectorXd d(n);
  std::memcpy(d.data(), dists, sizeof(double) * n);
  double r = o.log_sum_exp(alpha, d, *exp_sum);
  std::memcpy(dists, d.data(), sizeof(double) * n);
  return r;
}
#endif  // !DFTPAV_DROPIN
}  // extern "C"
