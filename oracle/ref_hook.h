// Observation hook for oracle/_ref — TEST INFRASTRUCTURE.
//
// PolyTrajOptimizer::OptimizeTrajectory (traj_optimizer.cpp:7-202) keeps the solution vector, the final cost and the
// solver's return code in locals and returns a bool.  To read them without touching the reference's sources, this header
// is force-included (-include) in front of traj_optimizer.cpp: it pulls in the reference's own lbfgs.hpp first (its include
// guard then skips the later inclusion), defines a wrapper that calls the real lbfgs::lbfgs_optimize with the caller's
// arguments — plus an evaluation wrapper and a progress callback that only record and always return 0, which by
// lbfgs.hpp:617-624 leaves the control flow unchanged — and renames the one call site (traj_optimizer.cpp:159) to it.
#pragma once
#include <vector>
#include "geo_utils2d/lbfgs.hpp"

namespace dftpav_ref {
struct SolveRecord {
  bool have = false;
  int ret = 0;
  double f = 0.0;
  std::vector<double> x;            // solution vector as lbfgs_optimize leaves it
  int last_progress_k = 0;          // k of the last progress report (lbfgs.hpp:617-624)
  int evals = 0;
  bool trace = false;               // record every evaluation and every iteration
  std::vector<double> eval_x, eval_g, eval_f;        // per evaluation: x [n], g [n], f
  std::vector<double> iter_x, iter_g, iter_fx, iter_step;  // per accepted iteration
  std::vector<int> iter_k, iter_ls;
  int n = 0;
};
inline SolveRecord *&current_record() {
  static thread_local SolveRecord *r = nullptr;
  return r;
}
struct Forward {
  lbfgs::lbfgs_evaluate_t eval;
  void *instance;
};
inline double observed_evaluate(void *fw, const Eigen::VectorXd &x, Eigen::VectorXd &g) {
  Forward *f = static_cast<Forward *>(fw);
  double v = f->eval(f->instance, x, g);
  SolveRecord *r = current_record();
  if (r) {
    r->evals++;
    if (r->trace) {
      r->eval_x.insert(r->eval_x.end(), x.data(), x.data() + x.size());
      r->eval_g.insert(r->eval_g.end(), g.data(), g.data() + g.size());
      r->eval_f.push_back(v);
    }
  }
  return v;
}
inline int observed_progress(void *, const Eigen::VectorXd &x, const Eigen::VectorXd &g, const double fx, const double step,
                             const int k, const int ls) {
  SolveRecord *r = current_record();
  if (r) {
    r->last_progress_k = k;
    if (r->trace) {
      r->iter_x.insert(r->iter_x.end(), x.data(), x.data() + x.size());
      r->iter_g.insert(r->iter_g.end(), g.data(), g.data() + g.size());
      r->iter_fx.push_back(fx);
      r->iter_step.push_back(step);
      r->iter_k.push_back(k);
      r->iter_ls.push_back(ls);
    }
  }
  return 0;
}
}  // namespace dftpav_ref

namespace lbfgs {
inline int lbfgs_optimize_observed(Eigen::VectorXd &x, double &f, lbfgs_evaluate_t proc_evaluate, lbfgs_stepbound_t proc_stepbound,
                                   lbfgs_progress_t proc_progress, void *instance, const lbfgs_parameter_t &param) {
  dftpav_ref::SolveRecord *r = dftpav_ref::current_record();
  if (!r || proc_progress != nullptr) return lbfgs_optimize(x, f, proc_evaluate, proc_stepbound, proc_progress, instance, param);
  dftpav_ref::Forward fw{proc_evaluate, instance};
  r->n = (int)x.size();
  int ret = lbfgs_optimize(x, f, dftpav_ref::observed_evaluate, proc_stepbound, dftpav_ref::observed_progress, &fw, param);
  r->have = true;
  r->ret = ret;
  r->f = f;
  r->x.assign(x.data(), x.data() + x.size());
  return ret;
}
}  // namespace lbfgs
#ifdef DFTPAV_REF_RENAME_CALL_SITE
#define lbfgs_optimize lbfgs_optimize_observed
#endif
