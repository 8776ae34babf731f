// capi.cpp — host half of the C-ABI declared in include/dftpav_hip.h.
//
// Holds what PolyTrajOptimizer::OptimizeTrajectory does before and after the
// L-BFGS call (traj_optimizer.cpp:7-134, 176-201): validation, corridor normal
// normalisation, boundary clamping, decision-vector packing, status mapping.
// Everything between (the solve itself) is the kernel in solver.hip.
// There is deliberately no CPU fallback: without a usable HIP device every
// entry point that needs one fails with DFTPAV_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <atomic>
#include <mutex>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dftpav_hip.h"
#include "device_types.h"
#include "e4_plan.h"
#include "traj_math.h"
#include "cr_trig.h"

namespace dftpav {
hipError_t launch_solver(const DevBatch &D, const DevBatch *d_dev, int mode, int threads, int grid, SchedArgs sched,
                         hipStream_t stream);
hipError_t launch_corridor(const unsigned char *cells, const unsigned *bits, int size_x, int size_y, double resolution, double origin_x, double origin_y,
                           const double *states, int n, double veh_width, double veh_length, double veh_dcr, const double *dl,
                           int n_dl, double *hpoly, double *batch_cor, int Npts, int NptsPad, int replicate, hipStream_t stream);
hipError_t launch_frontend(const dftpav_frontend_params &fp, const double *paths, const int *path_len, int max_path,
                           const double *start_states, const double *end_states, const double *start_ctrl, int n_hyp,
                           const dftpav_frontend_out &out, hipStream_t stream);
hipError_t launch_restarts(const double *inner, const double *durs, int n_hyp, int n_restarts, int n_inner, int M, double sigma,
                           double lo, double hi, unsigned long long seed, double *out_inner, double *out_durs, hipStream_t stream);
hipError_t launch_fit(const double *states, int S, int n_states, const double *opM, double *dur, double *coef, double *total,
                      double *start, hipStream_t stream);
hipError_t launch_validate(const unsigned char *cells, int size_x, int size_y, double resolution, double origin_x, double origin_y,
                           const double *coeffs, const double *piece_dt, const DevLayout &L, int B, double veh_width,
                           double veh_length, double veh_dcr, const double *t_tab, int n_t, double sample_dt, const double *v_tab,
                           int n_v, int *collision, int *first_sample, hipStream_t stream);
hipError_t launch_states(const double *coeffs, const double *piece_dt, const DevLayout &L, int B, double wheel_base, double t0,
                         double sample_dt, int n_samples, int filter, double *states, int *n_valid, hipStream_t stream);
hipError_t launch_shots(const double *from, const double *to, int n, double rho, double checkl, int max_samples,
                        const unsigned char *cells, int size_x, int size_y, double resolution, double origin_x, double origin_y,
                        double veh_width, double veh_length, double veh_dcr, const double *v_tab, int n_v, double *length, int *type,
                        double *seg, double *samples, int *n_samples, int *collides, hipStream_t stream);
hipError_t launch_corridor_layout(const double *raw, double *out, int B, int Npts, int H, int NptsPad, hipStream_t stream);
hipError_t launch_adopt(const DevBatch &D, const DevBatch &prev, hipStream_t stream);
// solver_ref.hip: the same path in the reference's own floating-point order
bool reference_order_supported(const DevLayout &L, const DevParams &P, int S);
size_t reference_order_scratch_doubles(const DevLayout &L, int B, int S);
size_t reference_order_table_doubles(int N);
void reference_order_pack_tables(int N, const double *full, double *packed);
int reference_order_interior_mask(int sweep, int row_mod_6);
RefPlan reference_order_plan(const DevLayout &L, const DevParams &P, int S, int B, int n_cu, bool allow_quad = true, bool throughput = false);
// the QUAD shape (solver_ref4.hip): four trajectories per wave; its copy of the corridor and its launches
size_t reference_order_quad_corridor_doubles(const DevLayout &L, int B);
hipError_t launch_quad_corridor(const DevBatch &D, double *cor_t, hipStream_t stream);
hipError_t launch_ring_reset(const DevBatch &D, hipStream_t stream); // solver_ref.hip
hipError_t launch_quadm_corridor(const DevBatch &D, double *cor_t, hipStream_t stream); // solver_ref4m.hip: the QUAD shape for several gear segments (RefPlan::quad == 2)
hipError_t launch_solver_ref4m(const DevBatch &D, const DevBatch *d_dev, int mode, const double *tabs, const double *cor_t, double *scratch, const RefPlan &pl,
                               int scheduled, int slots, int hand, hipStream_t stream);
hipError_t launch_solver_ref4(const DevBatch &D, const DevBatch *d_dev, int mode, const double *tabs, const double *cor_t, double *scratch, const RefPlan &pl,
                              int scheduled, int slots, int hand, hipStream_t stream);
hipError_t launch_solver_ref(const DevBatch &D, const DevBatch *d_dev, int mode, const double *tabs, double *scratch, const RefPlan &pl, int scheduled,
                             hipStream_t stream);
}
using namespace dftpav;

// An RCCL communicator and the number of handles that hold it (its creator and the handles it was shared with, each on its own
// host thread at most): the communicator is destroyed by whichever of them lets go last, in whatever order they do.  An RCCL
// communicator does not take concurrent enqueues: `mu` is held around every call on it (the holders' threads serialise there;
// the ORDER of the collectives across ranks stays the host's business -- same order on every rank).  Taking and dropping a
// reference (share / destroy / create) happens under g_comm_mu, so a handle never reads another's comm_ref while that one
// lets go of it.
struct CommShared {
  void *comm;
  std::atomic<int> holders;
  std::mutex mu;
};
static std::mutex g_comm_mu;

struct dftpav_handle {
  dftpav_params params;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // moving obstacles (device copies)
  int S = 0;
  int sur_pieces = 0; // pieces of all obstacles together
  int sur_version = 0; // bumped by dftpav_set_surround so batches refresh their device descriptor
  int *d_sur_off = nullptr;
  double *d_sur_dur = nullptr, *d_sur_coef = nullptr, *d_sur_total = nullptr, *d_sur_start = nullptr, *d_sur_theta = nullptr, *d_sur_bbox = nullptr;
  // obstacle map of the corridor generator (device copy) and the table of sample offsets along a line
  dftpav_grid_map map{};
  unsigned char *d_cells = nullptr;
  unsigned *d_bits = nullptr; // one bit per cell, when the whole map fits in a quarter of the LDS
  double *d_dl = nullptr;
  int n_dl = 0;
  hipEvent_t cev0 = nullptr, cev1 = nullptr; // around the last corridor kernel
  hipEvent_t mark[2] = {nullptr, nullptr};   // dftpav_mark
  bool ctimed = false;
  std::vector<struct dftpav_batch *> batches; // every live batch of this handle (obstacle changes finish their chained stragglers)
  // RCCL communicator of dftpav_comm_create (one rank per handle = per GPU), and the staging block of this rank's records
  void *comm = nullptr;
  struct CommShared *comm_ref = nullptr; // the communicator's holders (dftpav_comm_share): destroyed when the last one lets go
  int comm_ranks = 0, comm_rank = 0;
  unsigned char *d_comm_send = nullptr;
  size_t comm_send_bytes = 0;
};

struct dftpav_batch {
  dftpav_handle *h = nullptr;
  int B = 0;
  DevLayout L{};
  DevParams P{};
  int threads = 0;
  bool op_in_lds = false, cor_in_lds = false;
  bool have_corridor = false; // set by dftpav_batch_upload (host corridor) or dftpav_batch_corridor_from_states
  // time-sliced scheduling (batches larger than the device holds at once): the queue launch runs in the
  // shape above, the stragglers it hands over finish in the latency shape below
  bool sched = false;
  int slots = 0, slice = 0, hand_over = 0;
  int threads2 = 0;
  bool op_in_lds2 = false, cor_in_lds2 = false;
  int *d_queue = nullptr, *d_stragglers = nullptr, *d_stragglers2 = nullptr, *d_sflag = nullptr, *d_iota = nullptr;
  int qcap = 0;
  bool pending = false; // a chained solve left this batch's stragglers for the next chained solve (or dftpav_batch_finish)
  unsigned *d_qctl = nullptr;
  double *d_state = nullptr;
  DevBatch *d_dev2 = nullptr;
  // E4 lane plans (e4_plan.h) of the two launch shapes: host copies of the sizes, device tables
  E4Sizes e4{}, e4b{};
  int *d_e4[2][5] = {{nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr}}; // gtab, ltab, wave, round, piece
  int NptsPad = 0;
  std::vector<double> x0_host;
  bool uploaded = false;
  double t_now = 0.0, epis = 0.0;
  // device buffers
  double *d_x0 = nullptr, *d_iniS = nullptr, *d_finS = nullptr, *d_corridor = nullptr;
  int16_t *d_pt_piece = nullptr, *d_pt_j = nullptr;
  double *d_opM[kMaxSeg] = {nullptr}, *d_opMT[kMaxSeg] = {nullptr};
  double *d_histS = nullptr, *d_histY = nullptr, *d_histU = nullptr, *d_histV = nullptr, *d_histR = nullptr;
  double *d_x_in = nullptr, *d_x_out = nullptr, *d_f = nullptr, *d_g = nullptr;
  int *d_status = nullptr, *d_success = nullptr, *d_iters = nullptr, *d_evals = nullptr;
  long long *d_hist = nullptr, *d_ticks = nullptr, *d_prof = nullptr;
  unsigned char *d_records = nullptr; // [B + 1][16] result records written by the solver's epilogue (+ one zero record of padding)
  unsigned char *h_records = nullptr; // [B][16] the same in pinned host memory, written by the epilogues too (DevBatch::records_host)
  DevBatch *d_dev = nullptr; // device copy of the launch descriptor
  int dev_version = -1;
  // pinned host staging of the two descriptors and the event behind their last copy: refreshing the device copies then
  // needs no stream synchronisation (dftpav_plan_cycle enqueues the corridor kernel in front of the solve and must not wait for it)
  DevBatch *h_stage = nullptr;
  hipEvent_t stage_ev = nullptr;
  bool stage_busy = false;
  bool prof_on = false;
  double *d_coef = nullptr, *d_dt = nullptr;
  double *d_f_eval = nullptr; // costs of dftpav_batch_eval (kept apart from the solve's final costs)
  double *d_trace = nullptr;  // dftpav_batch_trace
  double *d_cor_raw = nullptr; // the caller's hPoly columns as uploaded (normalised and laid out on the device)
  // dftpav_batch_set_order(DFTPAV_ORDER_REFERENCE): the substitution tables of the band system and the term records (solver_ref.hip)
  int order = DFTPAV_ORDER_DEVICE;
  int ref_S = 0; // moving obstacles on the handle when the reference order was chosen (the term records are sized for them)
  double *d_ref_tab = nullptr, *d_ref_scratch = nullptr;
  RefPlan ref_plan{}; // its launch shape (chosen with the order)
  RefPlan ref_plan_wt{}; // QUAD shape: the TEAM / WAVE plan of the same batch (what the QUAD kernel leaves to solver_ref.hip: the coefficient read-out)
  double *d_cor_t = nullptr; // QUAD shape: the corridor as [B][4 H][Kmax + 1][16] (solver_ref4.hip), refreshed when the corridor changes
  bool cor_t_dirty = true;
  bool coef_override = false; // test hook dftpav_debug_batch_set_coeffs: validate / sample_states take the coefficients as they are
  int residency = -1; // the caller's residency hint (dftpav_batch_create_shaped); 2 = many such batches in flight: the throughput shapes whatever B
  // dftpav_plan_cycle: work buffers that live from the call to dftpav_plan_cycle_fetch (reused by the next cycle)
  struct PlanCycle {
    double *d_poses = nullptr, *d_t = nullptr, *d_v = nullptr, *d_rd = nullptr;
    int *d_col = nullptr, *d_first = nullptr, *d_valid = nullptr;
    size_t n_poses = 0, n_t = 0, n_v = 0, n_rd = 0;
    std::vector<double> poses, tt, vv; // host sources of the asynchronous copies
    int n_samples = 0;
    bool in_flight = false;
  } pc;
  int trace_b = -1, trace_cap = 0, trace_n = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;  // a solve was enqueued: ev0 / ev1 are recorded
  bool solved = false; // results of a solve of the CURRENT inputs exist (cleared by dftpav_batch_upload)
};

#define HIPCHK(h, call)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return DFTPAV_E_HIP;                                                               \
    }                                                                                    \
  } while (0)

extern "C" int dftpav_comm_destroy(dftpav_handle *h);

// ------------------------------------------------------------------ params
extern "C" void dftpav_default_params(dftpav_params *p) {
  std::memset(p, 0, sizeof(*p));
  // config/minco_config.pb.txt:65-100
  p->traj_resolution = 16;
  p->des_traj_resolution = 32;
  p->wei_obs = 1000.0;
  p->wei_surround = 5000.0;
  p->wei_feas = 2500.0;
  p->wei_sqrvar = 500.0;
  p->wei_time = 500.0;
  p->surround_clearance = 0.4;
  p->half_margin = 0.15;
  p->max_forward_vel = 5.0;
  p->max_forward_acc = 8.0;
  p->max_forward_cur = 1.0;
  p->max_backward_vel = 2.0;
  p->max_backward_acc = 4.0;
  p->max_backward_cur = 1.0;
  p->max_latacc = 5.0;
  p->max_phidot = 10000.0;
  p->gear_opt = 1;
  p->non_sinv = 0.24; // traj_optimizer.h:68
  p->mini_T = 0.1;
  p->fail_cost = 50000.0; // traj_optimizer.cpp:197
  // common/basics/semantics.h:66-76
  p->veh_width = 1.90;
  p->veh_length = 4.88;
  p->veh_wheel_base = 2.85;
  p->veh_d_cr = 1.015;
  // traj_optimizer.cpp:127-134 over lbfgs.hpp:15-129
  p->lbfgs_mem_size = 256;
  p->lbfgs_past = 3;
  p->lbfgs_delta = 1.0e-4;
  p->lbfgs_g_epsilon = 1.0e-16;
  p->lbfgs_max_iterations = 12000;
  p->lbfgs_max_linesearch = 64;
  p->lbfgs_min_step = 1.0e-32;
  p->lbfgs_max_step = 1.0e+20;
  p->lbfgs_f_dec_coeff = 1.0e-4;
  p->lbfgs_s_curv_coeff = 0.9;
  p->lbfgs_cautious_factor = 1.0e-6;
  p->lbfgs_machine_prec = 1.0e-16;
}

extern "C" int dftpav_num_vars(const dftpav_layout *l) { // traj_optimizer.cpp:80-86
  if (!l || l->M < 1) return DFTPAV_E_INVALID;
  int n = 0;
  for (int i = 0; i < l->M; i++) n += 2 * (l->piece_nums[i] - 1);
  n += l->M;
  n += 2 * (l->M - 1);
  n += 1 * (l->M - 1);
  return n;
}

extern "C" int dftpav_num_points(const dftpav_params *p, const dftpav_layout *l) { // traj_optimizer.cpp:44
  if (!p || !l || l->M < 1) return DFTPAV_E_INVALID;
  int s = 0;
  for (int i = 0; i < l->M; i++)
    s += (l->piece_nums[i] - 2) * (p->traj_resolution + 1) + 2 * (p->des_traj_resolution + 1);
  return s;
}

// sizeof probes so the Python mirror of the PODs can be checked against the compiled layout
extern "C" int dftpav_abi_sizeof_params(void) { return (int)sizeof(dftpav_params); }
extern "C" int dftpav_abi_sizeof_layout(void) { return (int)sizeof(dftpav_layout); }
extern "C" int dftpav_abi_sizeof_batch_data(void) { return (int)sizeof(dftpav_batch_data); }
extern "C" int dftpav_abi_sizeof_surround(void) { return (int)sizeof(dftpav_surround); }

// ------------------------------------------------------------------ handle
extern "C" int dftpav_create(const dftpav_params *params, int device, dftpav_handle **out) {
  if (!params || !out) return DFTPAV_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DFTPAV_E_NO_DEVICE;
  if (params->lbfgs_past > 8 || params->lbfgs_past < 0) return DFTPAV_E_UNSUPPORTED;
  if (params->lbfgs_mem_size <= 0 || params->lbfgs_mem_size > 1024) return DFTPAV_E_UNSUPPORTED;
  if (hipSetDevice(device) != hipSuccess) return DFTPAV_E_NO_DEVICE;
  auto *h = new dftpav_handle();
  h->params = *params;
  h->device = device;
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return DFTPAV_E_NO_DEVICE;
  }
  *out = h;
  return DFTPAV_OK;
}

static int finish_pending(dftpav_batch *b);
// The obstacle set is captured by every batch's launch descriptor: suspended stragglers of a chained solve must finish
// against the obstacles they started with, so they are finished before the set changes.
static int finish_batches_of(dftpav_handle *h) {
  for (dftpav_batch *b : h->batches)
    if (int rc = finish_pending(b)) return rc;
  return DFTPAV_OK;
}

// DevSurround::theta: for piece k of an obstacle the largest double t with f_k(t) <= duration_k, where f_k is the
// reference's walk t -= d_0, ..., t -= d_{k-1} (poly_traj_utils.hpp:515-521) in fp64.  f_k is non-decreasing in t, so the
// boundary is found by bisection over the bit patterns of the non-negative doubles with that same arithmetic.  Returns
// false (no table: the kernels walk) if a duration is not positive and finite or the thresholds do not increase.
static bool build_theta(const std::vector<int> &off, const std::vector<double> &dur, std::vector<double> &theta) {
  theta.assign(dur.size(), 0.0);
  for (size_t u = 0; u + 1 < off.size(); u++) {
    const double *d = dur.data() + off[u];
    const int np = off[u + 1] - off[u];
    double prev = -1.0;
    for (int k = 0; k < np; k++) {
      if (!(d[k] > 0.0) || !std::isfinite(d[k])) return false;
      auto stops = [&](double t) {
        for (int i = 0; i < k; i++) t -= d[i];
        return !(t > d[k]);
      };
      unsigned long long lo = 0, hi; // bit patterns: stops(lo) holds, stops(hi) does not
      const double big = 1.0e300;
      std::memcpy(&hi, &big, 8);
      while (hi - lo > 1) {
        const unsigned long long mid = lo + (hi - lo) / 2;
        double t;
        std::memcpy(&t, &mid, 8);
        if (stops(t)) lo = mid;
        else hi = mid;
      }
      double th;
      std::memcpy(&th, &lo, 8);
      if (!(th >= prev)) return false;
      prev = th;
      theta[off[u] + k] = th;
    }
  }
  return true;
}
// DevSurround::bbox: per piece a box that contains the obstacle's position for every local time in [0, duration].  With
// t = duration * s the piece is a quintic in s on [0, 1]; its Bernstein coefficients B_i = sum_{k<=i} C(i,k)/C(5,k) a_k
// duration^k span a hull the curve cannot leave, so their minimum and maximum per axis bound it.  The box is widened by
// 1e-6 m (+ 1e-9 relative): orders of magnitude above the rounding of this computation, of the kernels' evaluation of the
// position and of a local time that leaves [0, duration] by an ulp.
static void build_piece_boxes(const std::vector<double> &dur, const std::vector<double> &coef, std::vector<double> &box) {
  static const double binom5[6] = {1, 5, 10, 10, 5, 1};
  const size_t np = dur.size();
  box.assign(4 * np, 0.0);
  for (size_t k = 0; k < np; k++) {
    const double *cm = coef.data() + 12 * k; // 2x6 col-major, col 0 multiplies t^5
    for (int d = 0; d < 2; d++) {
      double b[6], pw = 1.0;
      for (int q = 0; q < 6; q++) {
        b[q] = cm[2 * (5 - q) + d] * pw; // a_q duration^q
        pw *= dur[k];
      }
      double lo = 0, hi = 0, mag = 0;
      for (int i = 0; i < 6; i++) {
        double B = 0.0, cik = 1.0; // C(i, q), built up with q
        for (int q = 0; q <= i; q++) {
          B += cik / binom5[q] * b[q];
          cik = cik * (i - q) / (q + 1);
        }
        if (i == 0 || B < lo) lo = B;
        if (i == 0 || B > hi) hi = B;
        mag = std::fmax(mag, std::fabs(B));
      }
      const double m = 1e-6 + 1e-9 * mag;
      box[4 * k + 2 * d] = lo - m;
      box[4 * k + 2 * d + 1] = hi + m;
    }
  }
}
// test hook (host only, no device): the tables the host derives from a set of moving obstacles -- theta [np] (zeros and
// return value 0 if the walk has no threshold table) and the piece boxes [np][4]
extern "C" int dftpav_debug_surround_tables(int S, const int *piece_offsets, const double *durations, const double *coeffs, double *theta,
                                            double *boxes) {
  if (S <= 0 || !piece_offsets || !durations || !coeffs) return DFTPAV_E_INVALID;
  const int np = piece_offsets[S];
  std::vector<int> off(piece_offsets, piece_offsets + S + 1);
  std::vector<double> dur(durations, durations + np), coef(coeffs, coeffs + 12 * (size_t)np), th, box;
  const bool ok = build_theta(off, dur, th);
  build_piece_boxes(dur, coef, box);
  if (theta) std::memcpy(theta, th.data(), sizeof(double) * np);
  if (boxes) std::memcpy(boxes, box.data(), sizeof(double) * 4 * (size_t)np);
  return ok ? 1 : 0;
}
// test hook (host only, no device): the distance gate of traj_optimizer.cpp:1393 for `npts` ego positions `sigma` [npts][2] at
// local times `t` [npts] against obstacle u, decided twice by the code the kernels run -- with the host's tables (threshold
// search + piece boxes) and without them (the reference's walk).  out_with / out_without [npts]: 1 = the pair passes the gate.
extern "C" int dftpav_debug_gate(int S, const int *piece_offsets, const double *durations, const double *coeffs, const double *total,
                                 const double *start, int u, double t_now, double trajtime, double veh_length_infl, int npts,
                                 const double *sigma, const double *t, int *out_with, int *out_without) {
  if (S <= 0 || u < 0 || u >= S || !piece_offsets || !durations || !coeffs || !total || !start) return DFTPAV_E_INVALID;
  const int np = piece_offsets[S];
  std::vector<int> off(piece_offsets, piece_offsets + S + 1);
  std::vector<double> dur(durations, durations + np), coef(coeffs, coeffs + 12 * (size_t)np), th, box;
  if (!build_theta(off, dur, th)) return 0;
  build_piece_boxes(dur, coef, box);
  dftpav::DevParams P{};
  P.veh_length_infl = veh_length_infl;
  dftpav::DevSurround A{S, piece_offsets, durations, coeffs, total, start, th.data(), box.data()};
  dftpav::DevSurround W{S, piece_offsets, durations, coeffs, total, start, nullptr, nullptr};
  for (int i = 0; i < npts; i++) {
    dftpav::DynObs ob;
    out_with[i] = dftpav::dyn_obstacle_near(P, A, u, t_now, t[i], trajtime, sigma + 2 * i, ob) ? 1 : 0;
    out_without[i] = dftpav::dyn_obstacle_near(P, W, u, t_now, t[i], trajtime, sigma + 2 * i, ob) ? 1 : 0;
  }
  return 1;
}
static int upload_boxes(dftpav_handle *h, const std::vector<double> &dur, const std::vector<double> &coef) {
  for (double v : coef)
    if (!std::isfinite(v)) return DFTPAV_OK; // no table: nothing is skipped
  std::vector<double> box;
  build_piece_boxes(dur, coef, box);
  if (box.empty()) return DFTPAV_OK;
  HIPCHK(h, hipMalloc(&h->d_sur_bbox, sizeof(double) * box.size()));
  HIPCHK(h, hipMemcpy(h->d_sur_bbox, box.data(), sizeof(double) * box.size(), hipMemcpyHostToDevice));
  return DFTPAV_OK;
}
static int upload_theta(dftpav_handle *h, const std::vector<int> &off, const std::vector<double> &dur) {
  std::vector<double> theta;
  if (!build_theta(off, dur, theta) || theta.empty()) return DFTPAV_OK; // d_sur_theta stays null: the kernels walk
  HIPCHK(h, hipMalloc(&h->d_sur_theta, sizeof(double) * theta.size()));
  HIPCHK(h, hipMemcpy(h->d_sur_theta, theta.data(), sizeof(double) * theta.size(), hipMemcpyHostToDevice));
  return DFTPAV_OK;
}

static void free_surround(dftpav_handle *h) {
  if (h->d_sur_off) (void)hipFree(h->d_sur_off);
  if (h->d_sur_dur) (void)hipFree(h->d_sur_dur);
  if (h->d_sur_theta) (void)hipFree(h->d_sur_theta);
  if (h->d_sur_bbox) (void)hipFree(h->d_sur_bbox);
  if (h->d_sur_coef) (void)hipFree(h->d_sur_coef);
  if (h->d_sur_total) (void)hipFree(h->d_sur_total);
  if (h->d_sur_start) (void)hipFree(h->d_sur_start);
  h->d_sur_off = nullptr;
  h->d_sur_dur = h->d_sur_coef = h->d_sur_total = h->d_sur_start = h->d_sur_theta = h->d_sur_bbox = nullptr;
  h->S = 0;
  h->sur_pieces = 0;
}

extern "C" void dftpav_destroy(dftpav_handle *h) {
  if (!h) return;
  (void)dftpav_comm_destroy(h);
  (void)hipSetDevice(h->device);
  free_surround(h);
  if (h->d_cells) (void)hipFree(h->d_cells);
  if (h->d_bits) (void)hipFree(h->d_bits);
  if (h->d_dl) (void)hipFree(h->d_dl);
  for (hipEvent_t &e : h->mark)
    if (e) (void)hipEventDestroy(e);
  if (h->cev0) (void)hipEventDestroy(h->cev0);
  if (h->cev1) (void)hipEventDestroy(h->cev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" const char *dftpav_last_error(const dftpav_handle *h) { return h ? h->err.c_str() : "null handle"; }
extern "C" void *dftpav_stream(dftpav_handle *h) { return h ? (void *)h->stream : nullptr; }

static void minco_operator(int N, std::vector<double> &Mop, std::vector<double> &MopT);

extern "C" int dftpav_sample_restarts(dftpav_handle *h, const double *inner_pts, const double *durations, int n_hyp, int n_restarts,
                                      int n_inner, int M, double sigma, double dur_lo, double dur_hi, unsigned long long seed,
                                      double *out_inner_pts, double *out_durations) {
  if (!h || !inner_pts || !durations || !out_inner_pts || !out_durations || n_hyp < 0 || n_restarts < 1 || n_inner < 0 ||
      (n_inner & 1) || M < 1 || !(sigma >= 0.0) || !(dur_lo > 0.0) || !(dur_hi >= dur_lo))
    return DFTPAV_E_INVALID;
  if (n_hyp == 0) return DFTPAV_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t B = (size_t)n_hyp * n_restarts;
  double *d_in = nullptr, *d_du = nullptr, *d_oi = nullptr, *d_od = nullptr;
  int rc = DFTPAV_OK;
  auto chk = [&](hipError_t e) {
    if (e != hipSuccess && rc == DFTPAV_OK) {
      h->err = hipGetErrorString(e);
      rc = DFTPAV_E_HIP;
    }
  };
  chk(hipMalloc(&d_in, sizeof(double) * std::max<size_t>(1, (size_t)n_hyp * n_inner)));
  chk(hipMalloc(&d_du, sizeof(double) * (size_t)n_hyp * M));
  chk(hipMalloc(&d_oi, sizeof(double) * std::max<size_t>(1, B * n_inner)));
  chk(hipMalloc(&d_od, sizeof(double) * B * M));
  if (rc == DFTPAV_OK) {
    chk(hipMemcpyAsync(d_in, inner_pts, sizeof(double) * (size_t)n_hyp * n_inner, hipMemcpyHostToDevice, h->stream));
    chk(hipMemcpyAsync(d_du, durations, sizeof(double) * (size_t)n_hyp * M, hipMemcpyHostToDevice, h->stream));
    chk(launch_restarts(d_in, d_du, n_hyp, n_restarts, n_inner, M, sigma, dur_lo, dur_hi, seed, d_oi, d_od, h->stream));
    chk(hipMemcpyAsync(out_inner_pts, d_oi, sizeof(double) * B * n_inner, hipMemcpyDeviceToHost, h->stream));
    chk(hipMemcpyAsync(out_durations, d_od, sizeof(double) * B * M, hipMemcpyDeviceToHost, h->stream));
    chk(hipStreamSynchronize(h->stream));
  }
  for (double *p : {d_in, d_du, d_oi, d_od})
    if (p) (void)hipFree(p);
  return rc;
}

extern "C" int dftpav_frontend_resample(dftpav_handle *h, const dftpav_frontend_params *fp, const double *paths, const int *path_len,
                                        int max_path, const double *start_states, const double *end_states,
                                        const double *start_ctrl, int n_hyp, const dftpav_frontend_out *out) {
  if (!h || !fp || !paths || !path_len || !start_states || !end_states || !start_ctrl || !out || n_hyp < 0 || max_path < 2)
    return DFTPAV_E_INVALID;
  if (out->max_seg < 1 || out->max_seg > 16 || out->max_pieces < 2 || out->max_states < 1) return DFTPAV_E_UNSUPPORTED;
  if (fp->traj_res < 1 || fp->dense_traj_res < 1 || !(fp->piece_duration > 0.0)) return DFTPAV_E_INVALID;
  for (int i = 0; i < n_hyp; i++)
    if (path_len[i] < 2 || path_len[i] > max_path) return DFTPAV_E_INVALID;
  if (n_hyp == 0) return DFTPAV_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t nh = (size_t)n_hyp, MS = (size_t)out->max_seg, MP = (size_t)out->max_pieces, MST = (size_t)out->max_states;
  struct Buf {
    void **dev;
    const void *src; // host input (nullptr for outputs)
    void *dst;       // host output
    size_t bytes;
  };
  double *d_paths = nullptr, *d_ss = nullptr, *d_es = nullptr, *d_sc = nullptr;
  int *d_len = nullptr;
  dftpav_frontend_out D = *out; // device pointers below
  D.n_seg = nullptr; D.singul = nullptr; D.piece_nums = nullptr; D.piece_dt = nullptr; D.ini_states = nullptr;
  D.fin_states = nullptr; D.inner_pts = nullptr; D.n_states = nullptr; D.states = nullptr;
  Buf bufs[] = {
      {(void **)&d_paths, paths, nullptr, sizeof(double) * nh * max_path * 3},
      {(void **)&d_len, path_len, nullptr, sizeof(int) * nh},
      {(void **)&d_ss, start_states, nullptr, sizeof(double) * nh * 4},
      {(void **)&d_es, end_states, nullptr, sizeof(double) * nh * 4},
      {(void **)&d_sc, start_ctrl, nullptr, sizeof(double) * nh * 2},
      {(void **)&D.n_seg, nullptr, out->n_seg, sizeof(int) * nh},
      {(void **)&D.singul, nullptr, out->singul, sizeof(int) * nh * MS},
      {(void **)&D.piece_nums, nullptr, out->piece_nums, sizeof(int) * nh * MS},
      {(void **)&D.piece_dt, nullptr, out->piece_dt, sizeof(double) * nh * MS},
      {(void **)&D.ini_states, nullptr, out->ini_states, sizeof(double) * nh * MS * 6},
      {(void **)&D.fin_states, nullptr, out->fin_states, sizeof(double) * nh * MS * 6},
      {(void **)&D.inner_pts, nullptr, out->inner_pts, sizeof(double) * nh * MS * (MP - 1) * 2},
      {(void **)&D.n_states, nullptr, out->n_states, sizeof(int) * nh * MS},
      {(void **)&D.states, nullptr, out->states, sizeof(double) * nh * MS * MST * 3},
  };
  int rc = DFTPAV_OK;
  auto chk = [&](hipError_t e) {
    if (e != hipSuccess && rc == DFTPAV_OK) {
      h->err = hipGetErrorString(e);
      rc = DFTPAV_E_HIP;
    }
  };
  for (Buf &b : bufs) {
    if (!b.src && !b.dst) {
      rc = DFTPAV_E_INVALID;
      break;
    }
    chk(hipMalloc(b.dev, b.bytes));
    if (rc != DFTPAV_OK) break;
    if (b.src) chk(hipMemcpyAsync(*b.dev, b.src, b.bytes, hipMemcpyHostToDevice, h->stream));
    else chk(hipMemsetAsync(*b.dev, 0, b.bytes, h->stream));
  }
  if (rc == DFTPAV_OK) chk(launch_frontend(*fp, d_paths, d_len, max_path, d_ss, d_es, d_sc, n_hyp, D, h->stream));
  if (rc == DFTPAV_OK)
    for (Buf &b : bufs)
      if (b.dst) chk(hipMemcpyAsync(b.dst, *b.dev, b.bytes, hipMemcpyDeviceToHost, h->stream));
  chk(hipStreamSynchronize(h->stream));
  for (Buf &b : bufs)
    if (*b.dev) (void)hipFree(*b.dev);
  return rc;
}

// The solver numbers (constraint point, obstacle) pairs with 16 bits and keeps a 16-bit mask of obstacles per point; its
// tables of the obstacles' pieces live in LDS.  A set beyond that is refused where it is installed, not at the first solve.
static int check_surround_limits(dftpav_handle *h, int S, long long pieces) {
  if (S > DFTPAV_MAX_SURROUND || pieces > DFTPAV_MAX_SURROUND_PIECES) {
    h->err = "too many moving obstacles (at most DFTPAV_MAX_SURROUND = 16 with DFTPAV_MAX_SURROUND_PIECES = 512 pieces in all)";
    return DFTPAV_E_UNSUPPORTED;
  }
  return DFTPAV_OK;
}
extern "C" int dftpav_fit_surround(dftpav_handle *h, const double *states, int S, int n_states) {
  if (!h || (S > 0 && !states) || S < 0 || (S > 0 && n_states < 3)) return DFTPAV_E_INVALID;
  // (no limit here: the fit is also a service of its own, its result read back with dftpav_get_surround; a fitted set beyond
  // the solver's limits makes solve / eval / validate of a batch return DFTPAV_E_UNSUPPORTED, as the header says)
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = finish_batches_of(h)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_surround(h);
  h->sur_version++;
  if (S == 0) return DFTPAV_OK; // ConverSurroundTrajFromPoints returns without obstacles, traj_manager.cpp:750-752
  const int N = n_states - 1, np = S * N;
  std::vector<int> off(S + 1);
  for (int i = 0; i <= S; i++) off[i] = i * N;
  std::vector<double> Mop, MopT;
  minco_operator(N, Mop, MopT);
  double *d_states = nullptr, *d_op = nullptr;
  HIPCHK(h, hipMalloc(&h->d_sur_off, sizeof(int) * (S + 1)));
  HIPCHK(h, hipMalloc(&h->d_sur_dur, sizeof(double) * np));
  HIPCHK(h, hipMalloc(&h->d_sur_coef, sizeof(double) * 12 * np));
  HIPCHK(h, hipMalloc(&h->d_sur_total, sizeof(double) * S));
  HIPCHK(h, hipMalloc(&h->d_sur_start, sizeof(double) * S));
  HIPCHK(h, hipMalloc(&d_states, sizeof(double) * 7 * (size_t)S * n_states));
  if (hipMalloc(&d_op, sizeof(double) * Mop.size()) != hipSuccess) {
    (void)hipFree(d_states);
    h->err = "hipMalloc";
    return DFTPAV_E_HIP;
  }
  int rc = DFTPAV_OK;
  auto chk = [&](hipError_t e) {
    if (e != hipSuccess && rc == DFTPAV_OK) {
      h->err = hipGetErrorString(e);
      rc = DFTPAV_E_HIP;
    }
  };
  chk(hipMemcpyAsync(h->d_sur_off, off.data(), sizeof(int) * (S + 1), hipMemcpyHostToDevice, h->stream));
  chk(hipMemcpyAsync(d_states, states, sizeof(double) * 7 * (size_t)S * n_states, hipMemcpyHostToDevice, h->stream));
  chk(hipMemcpyAsync(d_op, Mop.data(), sizeof(double) * Mop.size(), hipMemcpyHostToDevice, h->stream));
  if (rc == DFTPAV_OK)
    chk(launch_fit(d_states, S, n_states, d_op, h->d_sur_dur, h->d_sur_coef, h->d_sur_total, h->d_sur_start, h->stream));
  chk(hipStreamSynchronize(h->stream)); // off / Mop live on this stack frame
  (void)hipFree(d_states);
  (void)hipFree(d_op);
  if (rc == DFTPAV_OK) {
    std::vector<double> dur(np);
    if (hipMemcpy(dur.data(), h->d_sur_dur, sizeof(double) * np, hipMemcpyDeviceToHost) != hipSuccess) rc = DFTPAV_E_HIP;
    if (rc == DFTPAV_OK) rc = upload_theta(h, off, dur);
    if (rc == DFTPAV_OK && h->d_sur_theta) {
      std::vector<double> coef(12 * (size_t)np);
      if (hipMemcpy(coef.data(), h->d_sur_coef, sizeof(double) * coef.size(), hipMemcpyDeviceToHost) != hipSuccess) rc = DFTPAV_E_HIP;
      if (rc == DFTPAV_OK) rc = upload_boxes(h, dur, coef);
    }
  }
  if (rc == DFTPAV_OK) {
    h->S = S;
    h->sur_pieces = np;
  } else {
    free_surround(h);
  }
  return rc;
}

extern "C" int dftpav_get_surround(dftpav_handle *h, int *S, int *n_pieces, int *piece_offsets, double *durations, double *coeffs,
                                   double *total_duration, double *start_time) {
  if (!h) return DFTPAV_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (S) *S = h->S;
  if (n_pieces) *n_pieces = h->sur_pieces;
  if (h->S == 0) return DFTPAV_OK;
  if (piece_offsets) HIPCHK(h, hipMemcpy(piece_offsets, h->d_sur_off, sizeof(int) * (h->S + 1), hipMemcpyDeviceToHost));
  if (durations) HIPCHK(h, hipMemcpy(durations, h->d_sur_dur, sizeof(double) * h->sur_pieces, hipMemcpyDeviceToHost));
  if (coeffs) HIPCHK(h, hipMemcpy(coeffs, h->d_sur_coef, sizeof(double) * 12 * h->sur_pieces, hipMemcpyDeviceToHost));
  if (total_duration) HIPCHK(h, hipMemcpy(total_duration, h->d_sur_total, sizeof(double) * h->S, hipMemcpyDeviceToHost));
  if (start_time) HIPCHK(h, hipMemcpy(start_time, h->d_sur_start, sizeof(double) * h->S, hipMemcpyDeviceToHost));
  return DFTPAV_OK;
}

// ------------------------------------------------- Reeds-Shepp shots
extern "C" int dftpav_reeds_shepp_shots(dftpav_handle *h, const double *from, const double *to, int n, double max_cur,
                                        double checkl, int max_samples, double vertex_res, double *length, int *type, double *seg,
                                        double *samples, int *n_samples, int *collides) {
  if (!h || n < 0 || !(max_cur > 0.0) || !(checkl > 0.0) || max_samples < 1 || max_samples > 4096) return DFTPAV_E_INVALID;
  if (n == 0) return DFTPAV_OK;
  if (!from || !to) return DFTPAV_E_INVALID;
  if (collides && (!h->d_cells || !(vertex_res > 0.0))) return DFTPAV_E_INVALID; // a collision check needs the map
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->cev0) HIPCHK(h, hipEventCreate(&h->cev0));
  if (!h->cev1) HIPCHK(h, hipEventCreate(&h->cev1));
  std::vector<double> vv; // spacing of the outline points as the reference's running sum (shapes.cc:128)
  if (collides) {
    const double longest = std::max(h->params.veh_length, h->params.veh_width) + 1.0;
    for (double dl = vertex_res; dl < longest; dl += vertex_res) vv.push_back(dl);
  }
  if (vv.empty()) vv.push_back(1.0);
  double *d_from = nullptr, *d_to = nullptr, *d_len = nullptr, *d_seg = nullptr, *d_smp = nullptr, *d_v = nullptr;
  int *d_type = nullptr, *d_ns = nullptr, *d_col = nullptr;
  int rc = DFTPAV_OK;
  auto chk = [&](hipError_t e) {
    if (e != hipSuccess && rc == DFTPAV_OK) {
      h->err = hipGetErrorString(e);
      rc = DFTPAV_E_HIP;
    }
  };
  const size_t nsmp = (size_t)n * max_samples * 3;
  chk(hipMalloc(&d_from, sizeof(double) * 3 * (size_t)n));
  chk(hipMalloc(&d_to, sizeof(double) * 3 * (size_t)n));
  chk(hipMalloc(&d_len, sizeof(double) * (size_t)n));
  chk(hipMalloc(&d_seg, sizeof(double) * 5 * (size_t)n));
  chk(hipMalloc(&d_smp, sizeof(double) * nsmp));
  chk(hipMalloc(&d_v, sizeof(double) * vv.size()));
  chk(hipMalloc(&d_type, sizeof(int) * (size_t)n));
  chk(hipMalloc(&d_ns, sizeof(int) * (size_t)n));
  chk(hipMalloc(&d_col, sizeof(int) * (size_t)n));
  if (rc == DFTPAV_OK) {
    chk(hipMemcpyAsync(d_from, from, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, h->stream));
    chk(hipMemcpyAsync(d_to, to, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, h->stream));
    chk(hipMemcpyAsync(d_v, vv.data(), sizeof(double) * vv.size(), hipMemcpyHostToDevice, h->stream));
    chk(hipEventRecord(h->cev0, h->stream));
    chk(launch_shots(d_from, d_to, n, 1.0 / max_cur, checkl, max_samples, collides ? h->d_cells : nullptr, h->map.size_x,
                     h->map.size_y, h->map.resolution, h->map.origin_x, h->map.origin_y, h->params.veh_width, h->params.veh_length,
                     h->params.veh_d_cr, d_v, (int)vv.size(), d_len, d_type, d_seg, d_smp, d_ns, d_col, h->stream));
    chk(hipEventRecord(h->cev1, h->stream));
    if (length) chk(hipMemcpyAsync(length, d_len, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    if (type) chk(hipMemcpyAsync(type, d_type, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    if (seg) chk(hipMemcpyAsync(seg, d_seg, sizeof(double) * 5 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    if (samples) chk(hipMemcpyAsync(samples, d_smp, sizeof(double) * nsmp, hipMemcpyDeviceToHost, h->stream));
    if (n_samples) chk(hipMemcpyAsync(n_samples, d_ns, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    if (collides) chk(hipMemcpyAsync(collides, d_col, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    chk(hipStreamSynchronize(h->stream));
    h->ctimed = rc == DFTPAV_OK;
  }
  for (void *p : {(void *)d_from, (void *)d_to, (void *)d_len, (void *)d_seg, (void *)d_smp, (void *)d_v, (void *)d_type, (void *)d_ns,
                  (void *)d_col})
    if (p) (void)hipFree(p);
  return rc;
}

extern "C" int dftpav_set_grid_map(dftpav_handle *h, const dftpav_grid_map *map) {
  if (!h || !map || !map->cells || map->size_x <= 0 || map->size_y <= 0 || !(map->resolution > 0.0)) return DFTPAV_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->d_cells) (void)hipFree(h->d_cells);
  if (h->d_bits) (void)hipFree(h->d_bits);
  if (h->d_dl) (void)hipFree(h->d_dl);
  h->d_cells = nullptr;
  h->d_bits = nullptr;
  h->d_dl = nullptr;
  const size_t ncell = (size_t)map->size_x * map->size_y;
  HIPCHK(h, hipMalloc(&h->d_cells, ncell));
  HIPCHK(h, hipMemcpy(h->d_cells, map->cells, ncell, hipMemcpyHostToDevice));
  h->map = *map;
  h->map.cells = nullptr;
  if (ncell <= (size_t)8 * 40 * 1024) { // <= 40 KB of bits per workgroup: four workgroups per CU
    std::vector<unsigned> bits((ncell + 31) / 32, 0u);
    for (size_t i = 0; i < ncell; i++)
      if (map->cells[i] == 80) bits[i >> 5] |= 1u << (i & 31);
    HIPCHK(h, hipMalloc(&h->d_bits, sizeof(unsigned) * bits.size()));
    HIPCHK(h, hipMemcpy(h->d_bits, bits.data(), sizeof(unsigned) * bits.size(), hipMemcpyHostToDevice));
  }
  // sample offsets of CheckIfCollisionUsingLine (map_adapter.cpp:119): dl = 0, then dl += checkl; the longest
  // segment is the far edge of a fully grown rectangle
  const double checkl = map->resolution / 2.0;
  const double longest = std::max(h->params.veh_length, h->params.veh_width) + 2.0 * (10.0 + map->resolution) + 1.0;
  std::vector<double> dl;
  for (double v = 0.0; v < longest; v += checkl) dl.push_back(v);
  h->n_dl = (int)dl.size();
  HIPCHK(h, hipMalloc(&h->d_dl, sizeof(double) * dl.size()));
  HIPCHK(h, hipMemcpy(h->d_dl, dl.data(), sizeof(double) * dl.size(), hipMemcpyHostToDevice));
  if (!h->cev0) HIPCHK(h, hipEventCreate(&h->cev0));
  if (!h->cev1) HIPCHK(h, hipEventCreate(&h->cev1));
  return DFTPAV_OK;
}

// uploads the states and runs the corridor kernel into `hpoly` (device, [n][16]) or into a batch's corridor
static int run_corridor(dftpav_handle *h, const double *states, int n_states, double *d_hpoly, double *batch_cor, int Npts,
                        int NptsPad, int replicate) {
  double *d_states = nullptr;
  HIPCHK(h, hipMalloc(&d_states, sizeof(double) * 3 * (size_t)n_states));
  int rc = DFTPAV_OK;
  auto chk = [&](hipError_t e) {
    if (e != hipSuccess && rc == DFTPAV_OK) {
      h->err = hipGetErrorString(e);
      rc = DFTPAV_E_HIP;
    }
  };
  chk(hipMemcpyAsync(d_states, states, sizeof(double) * 3 * (size_t)n_states, hipMemcpyHostToDevice, h->stream));
  chk(hipEventRecord(h->cev0, h->stream));
  if (rc == DFTPAV_OK)
    chk(launch_corridor(h->d_cells, h->d_bits, h->map.size_x, h->map.size_y, h->map.resolution, h->map.origin_x, h->map.origin_y, d_states,
                        n_states, h->params.veh_width, h->params.veh_length, h->params.veh_d_cr, h->d_dl, h->n_dl, d_hpoly,
                        batch_cor, Npts, NptsPad, replicate, h->stream));
  chk(hipEventRecord(h->cev1, h->stream));
  chk(hipStreamSynchronize(h->stream));
  h->ctimed = rc == DFTPAV_OK;
  (void)hipFree(d_states);
  return rc;
}

extern "C" int dftpav_corridor_last_ms(dftpav_handle *h, float *ms) {
  if (!h || !ms || !h->ctimed) return DFTPAV_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventElapsedTime(ms, h->cev0, h->cev1));
  return DFTPAV_OK;
}

extern "C" int dftpav_corridor_rectangles(dftpav_handle *h, const double *states, int n_states, double *hpoly) {
  if (!h || !states || !hpoly || n_states < 0) return DFTPAV_E_INVALID;
  if (!h->d_cells) return DFTPAV_E_INVALID; // no map
  if (n_states == 0) return DFTPAV_OK;
  HIPCHK(h, hipSetDevice(h->device));
  double *d_hpoly = nullptr;
  HIPCHK(h, hipMalloc(&d_hpoly, sizeof(double) * 16 * (size_t)n_states));
  int rc = run_corridor(h, states, n_states, d_hpoly, nullptr, 1, 1, 1);
  if (rc == DFTPAV_OK && hipMemcpy(hpoly, d_hpoly, sizeof(double) * 16 * (size_t)n_states, hipMemcpyDeviceToHost) != hipSuccess) {
    h->err = "hipMemcpy";
    rc = DFTPAV_E_HIP;
  }
  (void)hipFree(d_hpoly);
  return rc;
}

extern "C" int dftpav_set_surround(dftpav_handle *h, const dftpav_surround *s) {
  if (!h) return DFTPAV_E_INVALID;
  if (s && s->S > 0) {
    if (!s->piece_offsets) return DFTPAV_E_INVALID;
    if (int rc = check_surround_limits(h, s->S, s->piece_offsets[s->S])) return rc; // the installed set stays as it is
  }
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = finish_batches_of(h)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_surround(h);
  h->sur_version++;
  if (!s || s->S <= 0) return DFTPAV_OK;
  int S = s->S, np = s->piece_offsets[S];
  if (np <= 0 || s->piece_offsets[0] != 0) return DFTPAV_E_INVALID;
  for (int u = 0; u < S; u++) // every obstacle is a trajectory of at least one piece (Trajectory::locatePieceIdx has no empty case)
    if (s->piece_offsets[u + 1] <= s->piece_offsets[u]) return DFTPAV_E_INVALID;
  HIPCHK(h, hipMalloc(&h->d_sur_off, sizeof(int) * (S + 1)));
  HIPCHK(h, hipMalloc(&h->d_sur_dur, sizeof(double) * np));
  HIPCHK(h, hipMalloc(&h->d_sur_coef, sizeof(double) * 12 * np));
  HIPCHK(h, hipMalloc(&h->d_sur_total, sizeof(double) * S));
  HIPCHK(h, hipMalloc(&h->d_sur_start, sizeof(double) * S));
  HIPCHK(h, hipMemcpy(h->d_sur_off, s->piece_offsets, sizeof(int) * (S + 1), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_sur_dur, s->durations, sizeof(double) * np, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_sur_coef, s->coeffs, sizeof(double) * 12 * np, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_sur_total, s->total_duration, sizeof(double) * S, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_sur_start, s->start_time, sizeof(double) * S, hipMemcpyHostToDevice));
  if (int rc = upload_theta(h, std::vector<int>(s->piece_offsets, s->piece_offsets + S + 1), std::vector<double>(s->durations, s->durations + np)))
    return rc;
  if (h->d_sur_theta)
    if (int rc = upload_boxes(h, std::vector<double>(s->durations, s->durations + np), std::vector<double>(s->coeffs, s->coeffs + 12 * (size_t)np)))
      return rc;
  h->sur_pieces = np;
  h->S = S;
  return DFTPAV_OK;
}

// ------------------------------------------------- constant MINCO operator
// Columns of A_N^{-1} for the N+5 RHS rows that can be non-zero
// (poly_traj_utils.hpp:968-977): the reference's own banded LU
// (poly_traj_utils.hpp:776-826, restated in traj_math.h) applied to unit
// vectors, once per distinct N at batch creation.  fp64, fixed operation order:
// the operator's bits are part of the reproducible program (tests compare them
// with the oracle's).
static void minco_operator(int N, std::vector<double> &Mop, std::vector<double> &MopT) {
  const int n6 = 6 * N, nc = N + 5;
  std::vector<double> band((size_t)n6 * 13, 0.0);
  BandedLU A{n6, 6, 6, band.data()};
  minco_fill(A, N);
  banded_factorize(A);
  Mop.assign((size_t)n6 * nc, 0.0);
  MopT.assign((size_t)n6 * nc, 0.0);
  std::vector<double> col(n6);
  for (int c = 0; c < nc; c++) {
    int row = c < 3 ? c : (c < N + 2 ? 6 * (c - 3) + 5 : n6 - 3 + (c - (N + 2)));
    std::fill(col.begin(), col.end(), 0.0);
    col[row] = 1.0;
    banded_solve1(A, col.data());
    for (int r = 0; r < n6; r++) {
      Mop[(size_t)r * nc + c] = col[r];
      MopT[(size_t)c * n6 + r] = col[r];
    }
  }
}

// debug/test hook: the operator the kernels use for a segment of N pieces, [6N][N+5] row-major
extern "C" int dftpav_debug_minco_operator(int N, double *out) {
  if (N < 2 || !out) return DFTPAV_E_INVALID;
  std::vector<double> M, MT;
  minco_operator(N, M, MT);
  std::memcpy(out, M.data(), sizeof(double) * M.size());
  return DFTPAV_OK;
}

// -------------------------------------------------------------------- batch
static void fill_dev_params(const dftpav_params &p, DevParams &P) {
  P.wei_obs = p.wei_obs;
  P.wei_surround = p.wei_surround;
  P.wei_feas = p.wei_feas;
  P.wei_time = p.wei_time;
  P.surround_clearance = p.surround_clearance;
  P.max_vel[0] = p.max_forward_vel; P.max_vel[1] = p.max_backward_vel;
  P.max_acc[0] = p.max_forward_acc; P.max_acc[1] = p.max_backward_acc;
  P.max_cur[0] = p.max_forward_cur; P.max_cur[1] = p.max_backward_cur;
  P.non_sinv = p.non_sinv;
  P.mini_T = p.mini_T;
  P.fail_cost = p.fail_cost;
  // footprint, traj_optimizer.cpp:1749-1775
  double W = p.veh_width + 2 * p.half_margin, Lh = p.veh_length + 2 * p.half_margin, dcr = p.veh_d_cr;
  P.veh_length_infl = Lh;
  double le[4][2] = {{dcr + Lh / 2.0, W / 2.0}, {dcr + Lh / 2.0, -W / 2.0}, {dcr - Lh / 2.0, -W / 2.0},
                     {dcr - Lh / 2.0, W / 2.0}};
  for (int k = 0; k < 4; k++) { P.vec_le[k][0] = le[k][0]; P.vec_le[k][1] = le[k][1]; }
  P.vec_le[4][0] = le[0][0]; P.vec_le[4][1] = le[0][1];
  fill_footprint_edges(P);
  P.gear_opt = p.gear_opt;
  P.mem_size = p.lbfgs_mem_size;
  P.past = p.lbfgs_past;
  P.max_iterations = p.lbfgs_max_iterations;
  P.max_linesearch = p.lbfgs_max_linesearch;
  P.delta = p.lbfgs_delta;
  P.g_epsilon = p.lbfgs_g_epsilon;
  P.min_step = p.lbfgs_min_step;
  P.max_step = p.lbfgs_max_step;
  P.f_dec_coeff = p.lbfgs_f_dec_coeff;
  P.s_curv_coeff = p.lbfgs_s_curv_coeff;
  P.cautious_factor = p.lbfgs_cautious_factor;
  P.machine_prec = p.lbfgs_machine_prec;
}

extern "C" void dftpav_batch_destroy(dftpav_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->h->device);
  (void)hipStreamSynchronize(b->h->stream);
  void *ptrs[] = {b->d_x0, b->d_iniS, b->d_finS, b->d_corridor, b->d_pt_piece, b->d_pt_j, b->d_histS, b->d_histU, b->d_histV, b->d_histR,
                  b->d_x_in, b->d_x_out, b->d_f, b->d_g, b->d_status, b->d_success, b->d_iters, b->d_evals,
                  b->d_hist, b->d_ticks, b->d_prof, b->d_dev, b->d_coef, b->d_dt, b->d_records,
                  b->d_queue, b->d_stragglers, b->d_stragglers2, b->d_sflag, b->d_iota, b->d_qctl, b->d_state, b->d_dev2,
                  b->d_f_eval, b->d_trace, b->d_cor_raw, b->d_ref_tab, b->d_ref_scratch, b->d_cor_t, b->pc.d_poses, b->pc.d_t, b->pc.d_v, b->pc.d_rd, b->pc.d_col, b->pc.d_first,
                  b->pc.d_valid};
  {
    auto &v = b->h->batches;
    v.erase(std::remove(v.begin(), v.end(), b), v.end());
  }
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (b->h_records) (void)hipHostFree(b->h_records);
  for (int w = 0; w < 2; w++)
    for (int t = 0; t < 5; t++)
      if (b->d_e4[w][t]) (void)hipFree(b->d_e4[w][t]);
  for (int i = 0; i < kMaxSeg; i++) {
    if (b->d_opM[i]) (void)hipFree(b->d_opM[i]);
    if (b->d_opMT[i]) (void)hipFree(b->d_opMT[i]);
  }
  if (b->h_stage) (void)hipHostFree(b->h_stage);
  if (b->stage_ev) (void)hipEventDestroy(b->stage_ev);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  delete b;
}

// the kernels' view of a layout: offsets of the segments' pieces, waypoints, right-hand-side rows and constraint points, and of
// tau | gear xy | gear angle inside x (traj_optimizer.cpp:96-115)
static void fill_dev_layout(const dftpav_layout &layout, int K, int Kd, DevLayout &L) {
  L = DevLayout{};
  L.M = layout.M;
  L.H = layout.H;
  L.K = K;
  L.Kd = Kd;
  L.Kmax = L.K > L.Kd ? L.K : L.Kd;
  int xoff = 0, poff = 0, roff = 0, ptoff = 0;
  for (int i = 0; i < L.M; i++) {
    int N = layout.piece_nums[i];
    L.piece_nums[i] = N;
    L.singuls[i] = layout.singuls[i];
    L.seg_piece0[i] = poff;
    L.seg_x0[i] = xoff;
    L.seg_rhs0[i] = roff;
    L.seg_pt0[i] = ptoff;
    poff += N;
    xoff += 2 * (N - 1);
    roff += N + 5;
    ptoff += (N - 2) * (L.K + 1) + 2 * (L.Kd + 1);
  }
  L.seg_piece0[L.M] = poff;
  L.seg_rhs0[L.M] = roff;
  L.seg_pt0[L.M] = ptoff;
  L.Ntot = poff;
  L.rhs_tot = roff;
  L.Npts = ptoff;
  L.x_tau0 = xoff;
  L.x_gear0 = xoff + L.M;
  L.x_ang0 = L.x_gear0 + 2 * (L.M - 1);
  L.n = L.x_ang0 + (L.M - 1);
  L.npad = ((L.n + 63) / 64) * 64;
}
// test hook (host only): is the reference order available for this layout with S moving obstacles, and which launch shape would
// a batch of B trajectories take on a device of n_cu CUs?  out = {supported, wave, threads, workgroups per CU, persistent
// workgroups, slice, LDS bytes per workgroup, width of the sequential sums}
extern "C" int dftpav_debug_reference_plan(const dftpav_layout *layout, const dftpav_params *p, int S, int B, int n_cu, long long *out) {
  if (!layout || !p || !out || layout->M < 1 || layout->M > kMaxSeg || B < 1 || n_cu < 1) return DFTPAV_E_INVALID;
  DevLayout L;
  fill_dev_layout(*layout, p->traj_resolution, p->des_traj_resolution, L);
  DevParams P;
  fill_dev_params(*p, P);
  out[0] = reference_order_supported(L, P, S) ? 1 : 0;
  const RefPlan pl = reference_order_plan(L, P, S, B, n_cu);
  out[1] = pl.wave | (pl.quad << 1); // 0: TEAM, 1: WAVE, 3: QUAD
  out[2] = pl.threads;
  out[3] = pl.wg_per_cu;
  out[4] = pl.slots;
  out[5] = pl.slice;
  out[6] = (long long)pl.lds;
  out[7] = L.n <= 16 ? 16 : (L.n <= 32 ? 32 : (L.n <= 40 ? 40 : (L.n <= 48 ? 48 : 64)));
  return DFTPAV_OK;
}

static int batch_create_impl(dftpav_handle *h, const dftpav_layout *layout, int B, int residency, dftpav_batch **out);
extern "C" int dftpav_batch_create(dftpav_handle *h, const dftpav_layout *layout, int B, dftpav_batch **out) {
  return batch_create_impl(h, layout, B, -1, out);
}
// residency: -1 = by the batch size (dftpav_batch_create); 0 = one workgroup per CU (lowest latency of a solve), 1 = two,
// 2 = four per CU (highest throughput): for callers that keep several small batches in flight on several handles
extern "C" int dftpav_batch_create_shaped(dftpav_handle *h, const dftpav_layout *layout, int B, int residency, dftpav_batch **out) {
  if (residency < -1 || residency > 2) return DFTPAV_E_INVALID;
  return batch_create_impl(h, layout, B, residency, out);
}
static int batch_create_impl(dftpav_handle *h, const dftpav_layout *layout, int B, int residency, dftpav_batch **out) {
  if (!h || !layout || !out || B < 1) return DFTPAV_E_INVALID;
  *out = nullptr;
  if (layout->M < 1 || layout->M > kMaxSeg || layout->H < 1) return DFTPAV_E_UNSUPPORTED;
  for (int i = 0; i < layout->M; i++) {
    if (layout->piece_nums[i] < 2) return DFTPAV_E_ONE_PIECE; // "There is only a piece?", traj_optimizer.cpp:38-41
  }
  const dftpav_params &p = h->params;
  if (p.traj_resolution < 1 || p.des_traj_resolution < 1) return DFTPAV_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  auto *b = new dftpav_batch();
  b->h = h;
  b->B = B;
  b->residency = residency;
  DevLayout &L = b->L;
  fill_dev_layout(*layout, p.traj_resolution, p.des_traj_resolution, L);
  if (L.n > 256 || L.Ntot > 1024 || L.Npts > 32767) {
    delete b;
    return DFTPAV_E_UNSUPPORTED;
  }
  fill_dev_params(p, b->P);
  // Residency plan.  Few trajectories (<= one per CU): latency shape, operators and the corridor
  // staged in LDS.  Many: throughput shape, LDS kept small so that two workgroups share a CU.
  {
    int n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    int shape = B <= n_cu ? 0 : (B <= 2 * n_cu ? 1 : 2);
    if (residency >= 0) shape = residency;
    if (const char *e = std::getenv("DFTPAV_MODE")) shape = std::atoi(e); // 0 latency, 1 two per CU, 2 four per CU
    b->threads = solver_threads(L, shape);
    int per_cu = shape == 0 ? 1 : (shape == 1 ? 2 : 8);
    if (shape == 2) {
      // one wave per trajectory and eight per CU when a workgroup's LDS fits an eighth of the CU; layouts with more state
      // (many pieces or points: BASELINE configs[4]) take two waves per trajectory and as many workgroups as their LDS allows
      const size_t l64 = solver_lds_bytes(L, b->P, kWave, false, false, 0) + 64;
      if (l64 > 20 * 1024 || L.Npts > 1024) {
        // Two or four waves per trajectory, whichever keeps more waves resident on a CU with the obstacle set installed on
        // the handle right now (its tables are part of a workgroup's LDS): BASELINE configs[4] holds 3 workgroups of 128
        // threads (6 waves) or 2 of 256 (8 waves) per CU -- 415 against 355 ms per 1024.  Without obstacles both give 8
        // waves and the narrower workgroup wastes less in the serial phases.
        const int np = h->S > 0 ? h->sur_pieces : 0;
        size_t best_waves = 0;
        for (int waves = 2; waves <= 4; waves += 2) {
          const int T = waves * kWave;
          const size_t cap = 8 / (size_t)waves; // 256 VGPRs: two waves per SIMD
          auto resident = [&](bool coef) {
            return std::min<size_t>(cap, (160 * 1024) / (solver_lds_bytes(L, b->P, T, false, false, np, coef) + 64));
          };
          const size_t wg = std::max<size_t>(1, np > 0 && np <= kSurCoefLds && resident(true) == resident(false) ? resident(true) : resident(false));
          if (wg * waves > best_waves) {
            best_waves = wg * waves;
            b->threads = T;
            per_cu = (int)wg;
          }
        }
      } else if (residency < 0 && B < 2 * n_cu * per_cu) {
        // Fewer trajectories than two per one-wave slot: an isolated batch of this size ends with most of the device idle
        // behind its longest solves.  Two waves per trajectory on half as many slots finish each solve sooner and keep the
        // time-sliced queue busy: 14.7 k against 13.5 k solves/s at B = 1024, 17.2 k against 12.8 k at 1536, 17.6 k against
        // 15.0 k at 2048, 21.3 k against 19.8 k at 3072; equal at 4096, where the one-wave shape is 12 % ahead as soon as
        // batches follow one another on two streams (the bench).  A caller that streams smaller batches asks for the
        // one-wave shape with dftpav_batch_create_shaped(..., 2, ...).
        b->threads = 2 * kWave;
        const size_t l128 = solver_lds_bytes(L, b->P, b->threads, false, false, 0) + 64;
        per_cu = (int)std::min<size_t>(4, std::max<size_t>(1, (160 * 1024) / l128));
      }
    }
    if (const char *e = std::getenv("DFTPAV_THREADS")) b->threads = std::atoi(e);
    // LDS budget per workgroup: the whole CU, half of it, its share in the throughput shape
    const size_t budget = shape == 0 ? 158 * 1024 : (shape == 1 ? 78 * 1024 : (size_t)(160 * 1024) / per_cu - 512);
    b->op_in_lds = solver_lds_bytes(L, b->P, b->threads, true, false, 512) + 64 <= budget;
    b->cor_in_lds = solver_lds_bytes(L, b->P, b->threads, b->op_in_lds, true, 512) + 64 <= budget;
    if (const char *e = std::getenv("DFTPAV_LDS")) { // bit 0 operators, bit 1 corridor
      int v = std::atoi(e);
      b->op_in_lds = (v & 1) != 0;
      b->cor_in_lds = (v & 2) != 0;
    }
    // More trajectories than resident workgroups: solve lengths differ several-fold, and a launch of one
    // workgroup per trajectory ends with a long tail of half-empty CUs.  Instead `slots` persistent
    // workgroups take trajectories from a queue, run them `slice` iterations at a time and put the
    // unfinished ones back, so all trajectories advance together; when no more than `hand_over` are left
    // they are finished by a second launch in the latency shape (one wide workgroup per CU).
    b->slots = n_cu * per_cu;
    // iterations per slice: long slices cost fewer suspensions, short ones balance the end of a solve better; layouts with
    // many constraint points (costly iterations, few trajectories per slot) take the short ones (measured)
    b->slice = L.Npts > 1024 ? (b->threads >= 4 * kWave ? 32 : 48) : 128;
    b->hand_over = n_cu;
    if (const char *e = std::getenv("DFTPAV_SLOTS")) b->slots = std::atoi(e);
    if (const char *e = std::getenv("DFTPAV_SLICE")) b->slice = std::atoi(e);
    if (const char *e = std::getenv("DFTPAV_HANDOVER")) b->hand_over = std::atoi(e);
    b->sched = B >= b->slots && b->slots > 0 && b->slice > 0;
    if (const char *e = std::getenv("DFTPAV_SCHED")) b->sched = std::atoi(e) != 0 && b->slots > 0 && b->slice > 0;
    if (b->hand_over > B) b->hand_over = B;
    b->threads2 = solver_threads(L, 0);
    b->op_in_lds2 = solver_lds_bytes(L, b->P, b->threads2, true, false, 512) + 64 <= 158 * 1024;
    b->cor_in_lds2 = solver_lds_bytes(L, b->P, b->threads2, b->op_in_lds2, true, 512) + 64 <= 158 * 1024;
  }
  size_t lds = solver_lds_bytes(L, b->P, b->threads, b->op_in_lds, b->cor_in_lds, 512) + 64;
  if (lds > 160 * 1024 || b->threads < 64 || b->threads > 512 || b->threads % 64) {
    delete b;
    return DFTPAV_E_UNSUPPORTED;
  }
  b->NptsPad = ((L.Npts + 63) / 64) * 64;
  const int n = L.n, M = L.M;
  int rc = DFTPAV_OK;
  auto fail = [&](int code) {
    dftpav_batch_destroy(b);
    return code;
  };
#define BCHK(call)                                                             \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);              \
      return fail(DFTPAV_E_HIP);                                               \
    }                                                                          \
  } while (0)
  BCHK(hipMalloc(&b->d_x0, sizeof(double) * (size_t)B * n));
  BCHK(hipMalloc(&b->d_iniS, sizeof(double) * (size_t)B * M * 6));
  BCHK(hipMalloc(&b->d_finS, sizeof(double) * (size_t)B * M * 6));
  BCHK(hipMalloc(&b->d_corridor, sizeof(double) * (size_t)B * L.H * 4 * b->NptsPad));
  BCHK(hipMemset(b->d_corridor, 0, sizeof(double) * (size_t)B * L.H * 4 * b->NptsPad));
  BCHK(hipMalloc(&b->d_pt_piece, sizeof(int16_t) * L.Npts));
  BCHK(hipMalloc(&b->d_pt_j, sizeof(int16_t) * L.Npts));
  // s and y of a stored pair are interleaved element by element: one 16-byte load fetches both (solver.hip, load_block)
  BCHK(hipMalloc(&b->d_histS, sizeof(double) * 2 * (size_t)B * b->P.mem_size * L.npad));
  b->d_histY = b->d_histS + 1; // alias into the same allocation, never freed on its own
  // never-written slots are read (and discarded) by the unconditional prefetch loads: keep them finite
  BCHK(hipMemset(b->d_histS, 0, sizeof(double) * 2 * (size_t)B * b->P.mem_size * L.npad));
  BCHK(hipMalloc(&b->d_histU, sizeof(double) * (size_t)B * b->P.mem_size * 8));
  BCHK(hipMalloc(&b->d_histV, sizeof(double) * (size_t)B * b->P.mem_size * 8));
  BCHK(hipMemset(b->d_histU, 0, sizeof(double) * (size_t)B * b->P.mem_size * 8));
  BCHK(hipMemset(b->d_histV, 0, sizeof(double) * (size_t)B * b->P.mem_size * 8));
  BCHK(hipMalloc(&b->d_histR, sizeof(double) * (size_t)B * b->P.mem_size * 2));
  BCHK(hipMemset(b->d_histR, 0, sizeof(double) * (size_t)B * b->P.mem_size * 2));
  BCHK(hipMalloc(&b->d_x_in, sizeof(double) * (size_t)B * n));
  BCHK(hipMalloc(&b->d_x_out, sizeof(double) * (size_t)B * n));
  BCHK(hipMalloc(&b->d_f, sizeof(double) * (size_t)B));
  BCHK(hipMalloc(&b->d_f_eval, sizeof(double) * (size_t)B));
  BCHK(hipMalloc(&b->d_g, sizeof(double) * (size_t)B * n));
  BCHK(hipMalloc(&b->d_status, sizeof(int) * (size_t)B));
  BCHK(hipMalloc(&b->d_success, sizeof(int) * (size_t)B));
  BCHK(hipMalloc(&b->d_iters, sizeof(int) * (size_t)B));
  BCHK(hipMalloc(&b->d_evals, sizeof(int) * (size_t)B));
  BCHK(hipMalloc(&b->d_hist, sizeof(long long) * (size_t)B));
  BCHK(hipMalloc(&b->d_ticks, sizeof(long long) * (size_t)B));
  BCHK(hipMalloc(&b->d_prof, sizeof(long long) * (size_t)B * 12));
  BCHK(hipMalloc(&b->d_records, (size_t)16 * (B + 1)));
  BCHK(hipMemset(b->d_records, 0, (size_t)16 * (B + 1)));
  // (pinned host memory the kernels can write; without it dftpav_batch_records copies from the device)
  if (std::getenv("DFTPAV_RECORDS_ON_HOST_OFF") == nullptr && hipHostMalloc(reinterpret_cast<void **>(&b->h_records), (size_t)16 * B, hipHostMallocDefault) == hipSuccess) {
    std::memset(b->h_records, 0, (size_t)16 * B);
  } else {
    (void)hipGetLastError();
    b->h_records = nullptr;
  }
  BCHK(hipMalloc(&b->d_dev, sizeof(DevBatch)));
  BCHK(hipMalloc(&b->d_dev2, sizeof(DevBatch)));
  if (b->sched) {
    const size_t stride = (size_t)solver_state_doubles(L, b->P);
    b->qcap = 2 * B; // own trajectories + the stragglers adopted from a previous batch of the same size
    BCHK(hipMalloc(&b->d_queue, sizeof(int) * (size_t)b->qcap));
    BCHK(hipMalloc(&b->d_stragglers, sizeof(int) * (size_t)B));
    BCHK(hipMalloc(&b->d_stragglers2, sizeof(int) * (size_t)B));
    BCHK(hipMalloc(&b->d_sflag, sizeof(int) * (size_t)B));
    BCHK(hipMalloc(&b->d_iota, sizeof(int) * (size_t)B));
    BCHK(hipMalloc(&b->d_qctl, sizeof(unsigned) * 16)); // [0..7] live counters, [8..15] their initial values
    {
      // queue = all trajectories: {head 0, published B, reserved B, unfinished B, stragglers 0}
      const unsigned ctl0[8] = {0u, (unsigned)B, (unsigned)B, (unsigned)B, 0u, 0u, 0u, 0u};
      BCHK(hipMemcpy(b->d_qctl + 8, ctl0, sizeof(ctl0), hipMemcpyHostToDevice));
    }
    BCHK(hipMalloc(&b->d_state, sizeof(double) * stride * (size_t)B));
    std::vector<int> iota(B);
    for (int i = 0; i < B; i++) iota[i] = i;
    BCHK(hipMemcpy(b->d_iota, iota.data(), sizeof(int) * (size_t)B, hipMemcpyHostToDevice));
  }
  BCHK(hipMalloc(&b->d_coef, sizeof(double) * (size_t)B * 12 * L.Ntot));
  BCHK(hipMalloc(&b->d_dt, sizeof(double) * (size_t)B * M));
  BCHK(hipEventCreate(&b->ev0));
  BCHK(hipEventCreate(&b->ev1));
  // constraint point -> (piece, j) tables, the pointid order of traj_optimizer.cpp:486-514
  {
    std::vector<int16_t> pp(L.Npts), pj(L.Npts);
    int pt = 0;
    for (int sg = 0; sg < M; sg++)
      for (int lp = 0; lp < L.piece_nums[sg]; lp++) {
        int K = (lp == 0 || lp == L.piece_nums[sg] - 1) ? L.Kd : L.K;
        for (int j = 0; j <= K; j++, pt++) {
          pp[pt] = (int16_t)(L.seg_piece0[sg] + lp);
          pj[pt] = (int16_t)j;
        }
      }
    BCHK(hipMemcpy(b->d_pt_piece, pp.data(), sizeof(int16_t) * L.Npts, hipMemcpyHostToDevice));
    BCHK(hipMemcpy(b->d_pt_j, pj.data(), sizeof(int16_t) * L.Npts, hipMemcpyHostToDevice));
  }
  // E4 lane plans of the two launch shapes
  for (int which = 0; which < 2; which++) {
    const E4Plan pl = build_e4_plan(L, which == 0 ? b->threads : b->threads2);
    (which == 0 ? b->e4 : b->e4b) = E4Sizes{pl.rounds, pl.groups, pl.left, pl.lcap};
    const std::vector<int> *tabs[5] = {&pl.gtab, &pl.ltab, &pl.wave, &pl.round, &pl.piece};
    for (int t = 0; t < 5; t++) {
      BCHK(hipMalloc(&b->d_e4[which][t], sizeof(int) * tabs[t]->size()));
      BCHK(hipMemcpy(b->d_e4[which][t], tabs[t]->data(), sizeof(int) * tabs[t]->size(), hipMemcpyHostToDevice));
    }
  }
  for (int sg = 0; sg < M; sg++) {
    int N = L.piece_nums[sg];
    std::vector<double> Mop, MopT;
    minco_operator(N, Mop, MopT);
    BCHK(hipMalloc(&b->d_opM[sg], sizeof(double) * Mop.size()));
    BCHK(hipMalloc(&b->d_opMT[sg], sizeof(double) * MopT.size()));
    BCHK(hipMemcpy(b->d_opM[sg], Mop.data(), sizeof(double) * Mop.size(), hipMemcpyHostToDevice));
    BCHK(hipMemcpy(b->d_opMT[sg], MopT.data(), sizeof(double) * MopT.size(), hipMemcpyHostToDevice));
  }
#undef BCHK
  (void)rc;
  h->batches.push_back(b);
  *out = b;
  return DFTPAV_OK;
}

// RealT2VirtualT, traj_optimizer.cpp:360-369
static double real_to_virtual(double rt, double mini_T) {
  return rt > 1.0 + mini_T ? (std::sqrt(2.0 * rt - 1.0 - 2 * mini_T) - 1.0)
                           : (1.0 - std::sqrt(2.0 / (rt - mini_T) - 1.0));
}

static void clamp_col(double *col, double lim) { // traj_optimizer.cpp:65-76
  double nrm = std::sqrt(col[0] * col[0] + col[1] * col[1]);
  if (nrm >= lim) {
    double nx = col[0] / nrm, ny = col[1] / nrm;
    col[0] = nx * (lim - 1.0e-2);
    col[1] = ny * (lim - 1.0e-2);
  }
}

extern "C" int dftpav_batch_upload(dftpav_batch *b, const dftpav_batch_data *d) {
  if (!b || !d || !d->ini_states || !d->fin_states || !d->inner_pts || !d->init_Ts) return DFTPAV_E_INVALID;
  b->pending = false; // new inputs: whatever a chained solve of the old ones left suspended is moot
  dftpav_handle *h = b->h;
  const dftpav_params &p = h->params;
  const DevLayout &L = b->L;
  const int B = b->B, M = L.M, n = L.n;
  const int ninner = L.x_tau0;
  // initTs.minCoeff() < mini_T, traj_optimizer.cpp:30-33
  for (size_t i = 0; i < (size_t)B * M; i++)
    if (d->init_Ts[i] < p.mini_T) return DFTPAV_E_MINI_T;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<double> ini(d->ini_states, d->ini_states + (size_t)B * M * 6);
  std::vector<double> fin(d->fin_states, d->fin_states + (size_t)B * M * 6);
  for (int t = 0; t < B; t++)
    for (int i = 0; i < M; i++) {
      double mv = L.singuls[i] > 0 ? p.max_forward_vel : p.max_backward_vel;
      double ma = L.singuls[i] > 0 ? p.max_forward_acc : p.max_backward_acc;
      double *I = ini.data() + ((size_t)t * M + i) * 6, *F = fin.data() + ((size_t)t * M + i) * 6;
      clamp_col(I + 2, mv);
      clamp_col(F + 2, mv);
      clamp_col(I + 4, ma);
      clamp_col(F + 4, ma);
    }
  // x0 packing, traj_optimizer.cpp:96-115
  b->x0_host.assign((size_t)B * n, 0.0);
  for (int t = 0; t < B; t++) {
    double *x = b->x0_host.data() + (size_t)t * n;
    std::memcpy(x, d->inner_pts + (size_t)t * ninner, sizeof(double) * ninner);
    for (int i = 0; i < M; i++) x[L.x_tau0 + i] = real_to_virtual(d->init_Ts[(size_t)t * M + i], p.mini_T);
    for (int i = 0; i < M - 1; i++) {
      const double *F = fin.data() + ((size_t)t * M + i) * 6;
      x[L.x_gear0 + 2 * i + 0] = F[0];
      x[L.x_gear0 + 2 * i + 1] = F[1];
      x[L.x_ang0 + i] = std::atan2(F[3], F[2]);
    }
  }
  // corridor: the private normalised copy of traj_optimizer.cpp:15,49-52 is made on the device (corridor_layout_kernel):
  // the caller's columns go up as they are, one copy, and are normalised and transposed there
  HIPCHK(h, hipMemcpy(b->d_x0, b->x0_host.data(), sizeof(double) * (size_t)B * n, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(b->d_iniS, ini.data(), sizeof(double) * ini.size(), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(b->d_finS, fin.data(), sizeof(double) * fin.size(), hipMemcpyHostToDevice));
  // a new upload without half-planes does not inherit the previous cycle's: they must follow from
  // dftpav_batch_corridor_from_states / _from_hypotheses before the next solve
  b->have_corridor = false;
  if (d->corridor) {
    const size_t nraw = (size_t)B * L.Npts * L.H * 4;
    if (!b->d_cor_raw) HIPCHK(h, hipMalloc(&b->d_cor_raw, sizeof(double) * nraw));
    HIPCHK(h, hipMemcpyAsync(b->d_cor_raw, d->corridor, sizeof(double) * nraw, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_corridor_layout(b->d_cor_raw, b->d_corridor, B, L.Npts, L.H, b->NptsPad, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream)); // the caller's buffer is free again when this returns
    b->have_corridor = true;
    b->cor_t_dirty = true;
  }
  b->t_now = d->t_now;
  b->epis = d->help_eps;
  b->uploaded = true;
  b->solved = false;
  b->coef_override = false;
  b->dev_version = -1; // t_now / help_eps live in the device copy of the launch descriptor: refresh it
  return DFTPAV_OK;
}

extern "C" int dftpav_batch_corridor_from_hypotheses(dftpav_batch *b, const double *states, int n_restarts) {
  if (!b || !states || n_restarts < 1 || b->B % n_restarts) return DFTPAV_E_INVALID;
  b->pending = false; // as dftpav_batch_upload
  dftpav_handle *h = b->h;
  if (!h->d_cells) return DFTPAV_E_INVALID;       // no map
  if (b->L.H != 4) return DFTPAV_E_UNSUPPORTED;   // rectangles
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int rc = run_corridor(h, states, (b->B / n_restarts) * b->L.Npts, nullptr, b->d_corridor, b->L.Npts, b->NptsPad, n_restarts);
  if (rc == DFTPAV_OK) {
    b->have_corridor = true;
    b->cor_t_dirty = true;
  }
  return rc;
}
extern "C" int dftpav_batch_corridor_from_states(dftpav_batch *b, const double *states) {
  return dftpav_batch_corridor_from_hypotheses(b, states, 1);
}

extern "C" int dftpav_batch_get_x0(dftpav_batch *b, double *x0) {
  if (!b || !x0 || !b->uploaded) return DFTPAV_E_INVALID;
  std::memcpy(x0, b->x0_host.data(), sizeof(double) * b->x0_host.size());
  return DFTPAV_OK;
}

// Whether the obstacles' coefficient blocks (96 B per piece) are staged in LDS for a launch shape: only if that does not
// cost a resident workgroup -- BASELINE configs[4] at 128 threads holds 3 workgroups per CU without them and 2 with them,
// and one more workgroup hides far more latency than the LDS copy saves.
static int sur_coef_in_lds(const dftpav_batch *b, int threads, bool op, bool cor) {
  const dftpav_handle *h = b->h;
  if (h->S <= 0 || h->sur_pieces > kSurCoefLds) return 0;
  const size_t by_waves = std::max<size_t>(1, 512 / (size_t)threads);
  auto resident = [&](bool coef) {
    const size_t lds = solver_lds_bytes(b->L, b->P, threads, op, cor, h->sur_pieces, coef) + 64;
    return std::min<size_t>(by_waves, (160 * 1024) / lds);
  };
  return resident(true) == resident(false) ? 1 : 0;
}

static DevBatch make_dev(dftpav_batch *b) {
  DevBatch D{};
  D.L = b->L;
  D.P = b->P;
  D.B = b->B;
  D.x0 = b->d_x0;
  D.iniS = b->d_iniS;
  D.finS = b->d_finS;
  D.corridor = b->d_corridor;
  D.NptsPad = b->NptsPad;
  D.pt_piece = b->d_pt_piece;
  D.pt_j = b->d_pt_j;
  D.op_in_lds = b->op_in_lds ? 1 : 0;
  D.cor_in_lds = b->cor_in_lds ? 1 : 0;
  D.e4_rounds = b->e4.rounds;
  D.e4_groups = b->e4.groups;
  D.e4_left = b->e4.left;
  D.e4_lcap = b->e4.lcap;
  D.e4_gtab = b->d_e4[0][0];
  D.e4_ltab = b->d_e4[0][1];
  D.e4_wave = b->d_e4[0][2];
  D.e4_round = b->d_e4[0][3];
  D.e4_piece = b->d_e4[0][4];
  int off = 0;
  for (int i = 0; i < kMaxSeg; i++) {
    D.opM[i] = b->d_opM[i];
    D.opMT[i] = b->d_opMT[i];
    D.op_off[i] = off;
    if (i < b->L.M) off += 6 * b->L.piece_nums[i] * (b->L.piece_nums[i] + 5);
  }
  dftpav_handle *h = b->h;
  D.sur.S = h->S;
  D.sur_np = h->sur_pieces;
  D.sur_coef_lds = sur_coef_in_lds(b, b->threads, b->op_in_lds, b->cor_in_lds);
  D.sur.piece_off = h->d_sur_off;
  D.sur.durations = h->d_sur_dur;
  D.sur.theta = h->d_sur_theta;
  D.sur.bbox = h->d_sur_bbox;
  D.sur.coeffs = h->d_sur_coef;
  D.sur.total = h->d_sur_total;
  D.sur.start = h->d_sur_start;
  D.t_now = b->t_now;
  D.epis = b->epis;
  D.histS = b->d_histS;
  D.histY = b->d_histY;
  D.histU = b->d_histU;
  D.histV = b->d_histV;
  D.histR = b->d_histR;
  D.queue = b->d_queue;
  D.qctl = b->d_qctl;
  D.stragglers = b->d_stragglers;
  D.stragglers2 = b->d_stragglers2;
  D.qcap = b->qcap;
  D.state = b->d_state;
  D.sflag = b->d_sflag;
  D.state_stride = solver_state_doubles(b->L, b->P);
  D.x_in = b->d_x_in;
  D.x_out = b->d_x_out;
  D.f_out = b->d_f;
  D.g_out = b->d_g;
  D.f_eval = b->d_f_eval;
  D.status = b->d_status;
  D.success = b->d_success;
  D.iters = b->d_iters;
  D.evals = b->d_evals;
  D.hist_sum = b->d_hist;
  D.ticks = b->d_ticks;
  D.records = b->d_records;
  D.records_host = b->h_records;
  D.prof = b->prof_on ? b->d_prof : nullptr;
  D.coef_out = b->d_coef;
  D.dt_out = b->d_dt;
  D.trace = b->d_trace;
  D.trace_b = b->trace_b;
  D.trace_cap = b->trace_cap;
  D.trace_n = b->trace_n;
  return D;
}

// refreshes the device copy of the launch descriptor when something it captures changed
static int sync_dev(dftpav_batch *b, DevBatch &D) {
  dftpav_handle *h = b->h;
  // the kernel numbers (constraint point, obstacle) pairs with 16 bits and keeps a 16-bit mask of obstacles per point
  if (h->S > 16 || (long long)b->L.Npts * h->S > 65535 || h->sur_pieces > 512) {
    h->err = "too many moving obstacles for this layout (S <= 16, Npts * S <= 65535, <= 512 pieces in all)";
    return DFTPAV_E_UNSUPPORTED;
  }
  if (b->order == DFTPAV_ORDER_REFERENCE && h->S != b->ref_S) { // the obstacle set changed after the order was chosen
    h->err = "reference order: the number of moving obstacles changed -- choose the order again (dftpav_batch_set_order)";
    return DFTPAV_E_UNSUPPORTED;
  }
  D = make_dev(b);
  int version = h->sur_version * 4 + (b->prof_on ? 1 : 0) + (b->uploaded ? 2 : 0);
  if (version != b->dev_version) {
    if (!b->h_stage) HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&b->h_stage), 2 * sizeof(DevBatch), hipHostMallocDefault));
    if (!b->stage_ev) HIPCHK(h, hipEventCreateWithFlags(&b->stage_ev, hipEventDisableTiming));
    if (b->stage_busy) HIPCHK(h, hipEventSynchronize(b->stage_ev)); // the previous copies out of the staging area (long done)
    b->h_stage[0] = D;
    DevBatch &D2 = b->h_stage[1];
    D2 = D; // the same batch in the latency shape (follow-up launch of a scheduled solve)
    D2.op_in_lds = b->op_in_lds2 ? 1 : 0;
    D2.cor_in_lds = b->cor_in_lds2 ? 1 : 0;
    D2.sur_coef_lds = sur_coef_in_lds(b, b->threads2, b->op_in_lds2, b->cor_in_lds2);
    D2.e4_rounds = b->e4b.rounds;
    D2.e4_groups = b->e4b.groups;
    D2.e4_left = b->e4b.left;
    D2.e4_lcap = b->e4b.lcap;
    D2.e4_gtab = b->d_e4[1][0];
    D2.e4_ltab = b->d_e4[1][1];
    D2.e4_wave = b->d_e4[1][2];
    D2.e4_round = b->d_e4[1][3];
    D2.e4_piece = b->d_e4[1][4];
    HIPCHK(h, hipMemcpyAsync(b->d_dev, &b->h_stage[0], sizeof(DevBatch), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(b->d_dev2, &b->h_stage[1], sizeof(DevBatch), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipEventRecord(b->stage_ev, h->stream));
    b->stage_busy = true;
    b->dev_version = version;
  }
  return DFTPAV_OK;
}

// debug/test hook: switch the in-kernel phase profiler on/off and read it back ([B][12] shader clocks)
extern "C" int dftpav_debug_profile(dftpav_batch *b, int enable, long long *out) {
  if (!b) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  HIPCHK(h, hipSetDevice(h->device));
  if (out) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, b->d_prof, sizeof(long long) * (size_t)b->B * 12, hipMemcpyDeviceToHost));
  }
  b->prof_on = enable != 0;
  return DFTPAV_OK;
}

// Records every evaluation of one trajectory during the following solves (what lbfgs_optimize shows its progress callback,
// lbfgs.hpp:242-249,617-624, but per evaluation): used by the lockstep parity test against the reference's line search.
extern "C" int dftpav_batch_trace_range(dftpav_batch *b, int first, int count, int max_evals) {
  if (!b || max_evals < 0 || (max_evals > 0 && (first < 0 || count < 1 || first + count > b->B))) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  if (max_evals > 0 && b->order == DFTPAV_ORDER_REFERENCE) { // solver_ref.hip records nothing: say so instead of returning empty traces
    h->err = "dftpav_batch_trace: not available in the reference order (the lockstep replay is a check of the device order)";
    return DFTPAV_E_UNSUPPORTED;
  }
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = finish_pending(b)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (b->d_trace) {
    HIPCHK(h, hipFree(b->d_trace));
    b->d_trace = nullptr;
  }
  b->trace_b = -1;
  b->trace_cap = 0;
  b->trace_n = 0;
  if (max_evals > 0) {
    const size_t nd = (size_t)count * (8 + (size_t)max_evals * (3 * (size_t)b->L.npad + 8));
    HIPCHK(h, hipMalloc(&b->d_trace, sizeof(double) * nd));
    HIPCHK(h, hipMemset(b->d_trace, 0, sizeof(double) * nd));
    b->trace_b = first;
    b->trace_n = count;
    b->trace_cap = max_evals;
  }
  b->dev_version = -1;
  return DFTPAV_OK;
}
extern "C" int dftpav_batch_trace(dftpav_batch *b, int traj, int max_evals) { return dftpav_batch_trace_range(b, traj, 1, max_evals); }

extern "C" int dftpav_batch_get_trace_of(dftpav_batch *b, int traj, double *out, int *n_evals) {
  if (!b || !n_evals || !b->d_trace || traj < b->trace_b || traj >= b->trace_b + b->trace_n) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = finish_pending(b)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const int n = b->L.n, npad = b->L.npad;
  const size_t stride = 3 * (size_t)npad + 8;
  const size_t block = 8 + (size_t)b->trace_cap * stride;
  std::vector<double> raw(block);
  HIPCHK(h, hipMemcpy(raw.data(), b->d_trace + (size_t)(traj - b->trace_b) * block, sizeof(double) * block, hipMemcpyDeviceToHost));
  int cnt = (int)raw[0];
  if (cnt > b->trace_cap) cnt = b->trace_cap;
  *n_evals = cnt;
  if (out)
    for (int i = 0; i < cnt; i++) {
      const double *r = raw.data() + 8 + (size_t)i * stride;
      double *o = out + (size_t)i * (3 * n + 4);
      std::memcpy(o, r, sizeof(double) * n);
      std::memcpy(o + n, r + npad, sizeof(double) * n);
      std::memcpy(o + 2 * n, r + 2 * npad, sizeof(double) * n);
      std::memcpy(o + 3 * n, r + 3 * npad, sizeof(double) * 4);
    }
  return DFTPAV_OK;
}
extern "C" int dftpav_batch_get_trace(dftpav_batch *b, double *out, int *n_evals) {
  if (!b) return DFTPAV_E_INVALID;
  return dftpav_batch_get_trace_of(b, b->trace_b, out, n_evals);
}

// every launch of the solve kernel for a batch goes through here: the reference-order kernel when the batch asks for it
static hipError_t launch_ref(dftpav_batch *b, const DevBatch &D, int mode, int scheduled) {
  if (b->ref_plan.quad && mode != kModeCoeffs) {
    if (b->cor_t_dirty) { // the QUAD shape reads its own layout of the corridor
      const hipError_t e = b->ref_plan.quad == 2 ? launch_quadm_corridor(D, b->d_cor_t, b->h->stream) : launch_quad_corridor(D, b->d_cor_t, b->h->stream);
      if (e != hipSuccess) return e;
      b->cor_t_dirty = false;
    }
    // A batch that has the device to itself (the default; dftpav_batch_set_hand_over(b, 0) says that other batches follow on other
    // streams) takes every wave slot and hands its last trajectories to the WAVE shape: a launch of solver_ref.hip's kernel queued
    // behind this one pops them from the same ring and resumes them from the same records (a wave per trajectory is 2-3 x faster
    // per iteration once the device is emptying).  In a stream of batches the launch is half as wide as the batch (solver_ref4.hip).
    const bool alone = b->hand_over != 0 && scheduled && mode == kModeSolve;
    const int slots = alone ? b->ref_plan.slots_wide : b->ref_plan.slots;
    const int hand = alone && b->ref_plan_wt.wave ? std::min(b->ref_plan.hand, b->B / 2) : 0;
    hipError_t e = b->ref_plan.quad == 2
                       ? launch_solver_ref4m(D, b->d_dev, mode, b->d_ref_tab, b->d_cor_t, b->d_ref_scratch, b->ref_plan, scheduled, slots, hand, b->h->stream)
                       : launch_solver_ref4(D, b->d_dev, mode, b->d_ref_tab, b->d_cor_t, b->d_ref_scratch, b->ref_plan, scheduled, slots, hand, b->h->stream);
    if (e == hipSuccess && hand > 0) e = launch_solver_ref(D, b->d_dev, kModeSolve, b->d_ref_tab, b->d_ref_scratch, b->ref_plan_wt, 1, b->h->stream);
    return e;
  }
  return launch_solver_ref(D, b->d_dev, mode, b->d_ref_tab, b->d_ref_scratch, b->ref_plan.quad ? b->ref_plan_wt : b->ref_plan, scheduled, b->h->stream);
}
static hipError_t launch_for(dftpav_batch *b, const DevBatch &D, int mode) {
  if (b->order == DFTPAV_ORDER_REFERENCE) return launch_ref(b, D, mode, 0);
  return launch_solver(D, b->d_dev, mode, b->threads, b->B, SchedArgs{0, 0, 0, nullptr}, b->h->stream);
}

// The coefficient tables of the four substitution sweeps of BandedSystem::solve / solveAdj (poly_traj_utils.hpp:805-852) for
// a segment of N pieces, row-oriented: row i of a sweep takes tab[i][0..5] against its six predecessors in the order the
// reference's column loops reach it (solver_ref.hip, sweep).  From the reference's own LU (banded_factorize, traj_math.h).
//   [0] solve, forward:     L(i, i-6+k)      [1] solve, backward:    U(i, i+6-k), then / U(i,i)
//   [2] solveAdj, forward:  U(i-6+k, i), then / U(i,i)               [3] solveAdj, backward: L(i+6-k, i)
//   each row 8 doubles: the six coefficients, U(i,i), 1 / U(i,i).
// Returns false if a row of a middle block (rows 6 .. 6N-7) has another non-zero pattern than the kernel assumes there.
static bool reference_order_tables(int N, std::vector<double> &out) {
  const int n6 = 6 * N;
  std::vector<double> band((size_t)n6 * 13, 0.0);
  BandedLU A{n6, 6, 6, band.data()};
  minco_fill(A, N);
  banded_factorize(A);
  out.assign((size_t)(4 * 48) * N, 0.0); // the full form [4][6N][8]; reference_order_pack_tables makes the kernel's
  double *t[4];
  for (int q = 0; q < 4; q++) t[q] = out.data() + (size_t)q * 8 * n6;
  bool ok = true;
  for (int i = 0; i < n6; i++) {
    for (int k = 0; k < 6; k++) {
      const int jl = i - 6 + k, jh = i + 6 - k;
      if (jl >= 0) {
        t[0][8 * i + k] = A.at(i, jl);
        t[2][8 * i + k] = A.at(jl, i);
      }
      if (jh <= n6 - 1) {
        t[1][8 * i + k] = A.at(i, jh);
        t[3][8 * i + k] = A.at(jh, i);
      }
    }
    for (int q = 0; q < 4; q++) {
      int m = 0;
      for (int k = 0; k < 6; k++)
        if (t[q][8 * i + k] != 0.0) m |= 1 << k;
      if (i >= 6 && i < n6 - 6 && m != reference_order_interior_mask(q, i % 6)) ok = false;
      t[q][8 * i + 6] = A.at(i, i);
      { // the kernel divides by the diagonal through its reciprocal (solver_ref.hip: div_by_rcp): the bits of the division as long
        // as the diagonal is an ordinary number -- O(1) for every MINCO system; anything else is refused here
        const double dg = std::fabs(A.at(i, i));
        if (!(dg >= 0x1p-500 && dg <= 0x1p500)) ok = false;
        t[q][8 * i + 7] = 1.0 / A.at(i, i);
      }
    }
  }
  return ok;
}
// test hook (host only): the correctly rounded sin / cos the reference-order kernel uses for the junction angles (cr_trig.h),
// run on the host for n arguments
extern "C" int dftpav_debug_cr_sincos(int n, const double *x, double *s, double *c) {
  if (n < 0 || !x || !s || !c) return DFTPAV_E_INVALID;
  for (int i = 0; i < n; i++) dftpav::crt::sincos(x[i], s[i], c[i]);
  return DFTPAV_OK;
}
// test hook (host only): which = 0 exp, 1 log, 2 x^3 -- the correctly rounded functions of cr_trig.h for n arguments; 3 / 4: exp / log
// by their accurate phase alone (what the quick phase with its rounding test stands in front of)
extern "C" int dftpav_debug_cr_fn(int which, int n, const double *x, double *y) {
  if (n < 0 || !x || !y || which < 0 || which > 6) return DFTPAV_E_INVALID;
  for (int i = 0; i < n; i++) {
    switch (which) {
      case 0: y[i] = dftpav::crt::exp_cr(x[i]); break;
      case 1: y[i] = dftpav::crt::log_cr(x[i]); break;
      case 2: y[i] = dftpav::crt::cube_cr(x[i]); break;
      case 3: y[i] = dftpav::crt::exp_cr_impl<false>(x[i]); break;
      case 4: y[i] = dftpav::crt::log_cr_impl<false>(x[i]); break;
      case 5: y[i] = dftpav::crt::atan(x[i]); break;   // (the step kernels' reference order: validate / states / frontend)
      default: y[i] = dftpav::crt::tan(x[i]); break;
    }
  }
  return DFTPAV_OK;
}
extern "C" int dftpav_debug_cr_atan2(int n, const double *y, const double *x, double *out) {
  if (n < 0 || !x || !y || !out) return DFTPAV_E_INVALID;
  for (int i = 0; i < n; i++) out[i] = dftpav::crt::atan2(y[i], x[i]);
  return DFTPAV_OK;
}
// test hook (host only): the sweep tables of a segment of N pieces, [4][6N][8]; returns 1 if the middle blocks have the assumed pattern
extern "C" int dftpav_debug_reference_tables(int N, double *out) {
  if (N < 2) return DFTPAV_E_INVALID;
  std::vector<double> tab;
  const bool ok = reference_order_tables(N, tab);
  if (out) std::memcpy(out, tab.data(), sizeof(double) * tab.size());
  return ok ? 1 : 0;
}

// test hook (host only): the same tables in the layout the kernel reads (solver_ref.hip: "The table of one sweep"): blocks in
// traversal order, whole rows at the two ends, only the coefficients of the interior pattern in between; *n_doubles = their size
extern "C" int dftpav_debug_reference_tables_packed(int N, double *out, int *n_doubles) {
  if (N < 2 || !n_doubles) return DFTPAV_E_INVALID;
  *n_doubles = (int)reference_order_table_doubles(N);
  if (!out) return DFTPAV_OK;
  std::vector<double> full;
  const bool ok = reference_order_tables(N, full);
  reference_order_pack_tables(N, full.data(), out);
  return ok ? 1 : 0;
}

// the ring, the flags and the state records of a scheduled solve, for a batch whose device-order plan did not need them
static hipError_t ensure_ring_buffers(dftpav_batch *b) {
  if (b->d_queue && b->d_sflag && b->d_iota && b->d_qctl && b->d_state) return hipSuccess;
  const int B = b->B;
  const size_t stride = (size_t)solver_state_doubles(b->L, b->P);
  int *q = nullptr, *fl = nullptr, *io = nullptr;
  unsigned *ctl = nullptr;
  double *st = nullptr;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) {
    if (e == hipSuccess && r != hipSuccess) e = r;
  };
  const int qcap = 2 * B;
  chk(hipMalloc(&q, sizeof(int) * (size_t)qcap));
  chk(hipMalloc(&fl, sizeof(int) * (size_t)B));
  chk(hipMalloc(&io, sizeof(int) * (size_t)B));
  chk(hipMalloc(&ctl, sizeof(unsigned) * 16));
  chk(hipMalloc(&st, sizeof(double) * stride * (size_t)B));
  if (e == hipSuccess) {
    const unsigned ctl0[8] = {0u, (unsigned)B, (unsigned)B, (unsigned)B, 0u, 0u, 0u, 0u};
    chk(hipMemcpy(ctl + 8, ctl0, sizeof(ctl0), hipMemcpyHostToDevice));
    std::vector<int> iota(B);
    for (int i = 0; i < B; i++) iota[i] = i;
    chk(hipMemcpy(io, iota.data(), sizeof(int) * (size_t)B, hipMemcpyHostToDevice));
  }
  if (e != hipSuccess) {
    for (void *p : {(void *)q, (void *)fl, (void *)io, (void *)ctl, (void *)st})
      if (p) (void)hipFree(p);
    (void)hipGetLastError();
    return e;
  }
  // (a batch either has all of them -- its device-order plan is scheduled -- or none)
  b->d_queue = q;
  b->d_sflag = fl;
  b->d_iota = io;
  b->d_qctl = ctl;
  b->d_state = st;
  b->qcap = qcap;
  b->dev_version = -1; // the device descriptor carries these pointers
  return hipSuccess;
}

extern "C" int dftpav_batch_set_order(dftpav_batch *b, int order) {
  if (!b || (order != DFTPAV_ORDER_DEVICE && order != DFTPAV_ORDER_REFERENCE)) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  if (order == b->order && (order == DFTPAV_ORDER_DEVICE || b->ref_S == h->S)) return DFTPAV_OK;
  if (order == DFTPAV_ORDER_REFERENCE && b->d_trace) { // the reference-order kernel does not record evaluations
    h->err = "reference order: dftpav_batch_trace is a device-order facility -- switch the trace off first";
    return DFTPAV_E_UNSUPPORTED;
  }
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = finish_pending(b)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (order == DFTPAV_ORDER_REFERENCE) {
    if (!reference_order_supported(b->L, b->P, h->S)) {
      h->err = "reference order: n <= 256 variables, H <= 12 half-planes, 5 H + S + 4 <= 64 terms per point, every gear segment >= 2 pieces, 159 KB of LDS";
      return DFTPAV_E_UNSUPPORTED;
    }
    {
      int n_cu = 256;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
      const RefPlan pl = reference_order_plan(b->L, b->P, h->S, b->B, n_cu, true, b->residency == 2);
      if (pl.wave && ensure_ring_buffers(b) != hipSuccess) {
        h->err = "reference order: no device memory for the ring of this batch";
        return DFTPAV_E_HIP;
      }
      b->ref_plan = pl;
      if (pl.quad) {
        b->ref_plan_wt = reference_order_plan(b->L, b->P, h->S, b->B, n_cu, false, true); // (the WAVE shape whatever B: it finishes the QUAD shape's last trajectories)
        if (!b->d_cor_t && hipMalloc(&b->d_cor_t, sizeof(double) * reference_order_quad_corridor_doubles(b->L, b->B)) != hipSuccess) {
          (void)hipGetLastError();
          h->err = "reference order: no device memory for the QUAD shape's copy of the corridor";
          return DFTPAV_E_HIP;
        }
        b->cor_t_dirty = true;
      }
    }
    if (!b->d_ref_tab || !b->d_ref_scratch || b->ref_S != h->S) {
      std::vector<double> tab; // the tables of the segments, one after the other
      for (int sg = 0; sg < b->L.M; sg++) {
        std::vector<double> one;
        if (!reference_order_tables(b->L.piece_nums[sg], one)) {
          h->err = "reference order: the LU factors of this band system do not have the pattern the kernel assumes";
          return DFTPAV_E_UNSUPPORTED;
        }
        std::vector<double> packed(reference_order_table_doubles(b->L.piece_nums[sg]));
        reference_order_pack_tables(b->L.piece_nums[sg], one.data(), packed.data());
        tab.insert(tab.end(), packed.begin(), packed.end());
      }
      double *d_tab = nullptr, *d_scr = nullptr;
      if (hipMalloc(&d_tab, sizeof(double) * tab.size()) != hipSuccess ||
          hipMalloc(&d_scr, sizeof(double) * reference_order_scratch_doubles(b->L, b->B, h->S)) != hipSuccess ||
          hipMemcpy(d_tab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
        if (d_tab) (void)hipFree(d_tab);
        if (d_scr) (void)hipFree(d_scr);
        (void)hipGetLastError();
        h->err = "reference order: no device memory for the term records of this batch";
        return DFTPAV_E_HIP; // the order stays as it was, the batch usable
      }
      if (b->d_ref_tab) (void)hipFree(b->d_ref_tab);
      if (b->d_ref_scratch) (void)hipFree(b->d_ref_scratch);
      b->d_ref_tab = d_tab;
      b->d_ref_scratch = d_scr;
      b->ref_S = h->S;
    }
  }
  b->order = order;
  b->solved = false;
  return DFTPAV_OK;
}
extern "C" int dftpav_batch_get_order(const dftpav_batch *b) { return b ? b->order : DFTPAV_E_INVALID; }

extern "C" int dftpav_batch_eval(dftpav_batch *b, const double *x, double *f, double *g) {
  if (!b || !x || !b->uploaded || !b->have_corridor) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  const size_t nb = (size_t)b->B * b->L.n;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(b->d_x_in, x, sizeof(double) * nb, hipMemcpyHostToDevice, h->stream));
  DevBatch D;
  if (int rc = sync_dev(b, D)) return rc;
  HIPCHK(h, launch_for(b, D, kModeEval));
  if (f) HIPCHK(h, hipMemcpyAsync(f, b->d_f_eval, sizeof(double) * b->B, hipMemcpyDeviceToHost, h->stream));
  if (g) HIPCHK(h, hipMemcpyAsync(g, b->d_g, sizeof(double) * nb, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFTPAV_OK;
}

// the follow-up launch of a scheduled solve: the stragglers in the latency shape
static int launch_stragglers(dftpav_batch *b, const DevBatch &D, int source) {
  dftpav_handle *h = b->h;
  DevBatch D2 = D;
  D2.op_in_lds = b->op_in_lds2 ? 1 : 0;
  D2.cor_in_lds = b->cor_in_lds2 ? 1 : 0;
  D2.sur_coef_lds = sur_coef_in_lds(b, b->threads2, b->op_in_lds2, b->cor_in_lds2);
  D2.e4_rounds = b->e4b.rounds;
  D2.e4_groups = b->e4b.groups;
  D2.e4_left = b->e4b.left;
  D2.e4_lcap = b->e4b.lcap;
  D2.e4_gtab = b->d_e4[1][0];
  D2.e4_ltab = b->d_e4[1][1];
  D2.e4_wave = b->d_e4[1][2];
  D2.e4_round = b->d_e4[1][3];
  D2.e4_piece = b->d_e4[1][4];
  HIPCHK(h, launch_solver(D2, b->d_dev2, kModeSolve, b->threads2, b->hand_over, SchedArgs{source, 0, 0, nullptr}, h->stream));
  return DFTPAV_OK;
}

// a chained solve left the stragglers of `b` suspended: finish them now (no-op otherwise)
static int finish_pending(dftpav_batch *b) {
  if (!b->pending) return DFTPAV_OK;
  dftpav_handle *h = b->h;
  HIPCHK(h, hipSetDevice(h->device));
  DevBatch D;
  if (int rc = sync_dev(b, D)) return rc;
  if (b->hand_over > 0)
    if (int rc = launch_stragglers(b, D, 2)) return rc;
  HIPCHK(h, hipEventRecord(b->ev1, h->stream));
  b->pending = false;
  return DFTPAV_OK;
}

static bool chain_compatible(const dftpav_batch *a, const dftpav_batch *b) {
  return a->h == b->h && a->sched && b->sched && a->B == b->B && a->threads == b->threads &&
         a->op_in_lds == b->op_in_lds && a->cor_in_lds == b->cor_in_lds && a->threads2 == b->threads2 &&
         a->hand_over == b->hand_over && a->hand_over > 0 && a->NptsPad == b->NptsPad && a->prof_on == b->prof_on &&
         std::memcmp(&a->L, &b->L, sizeof(DevLayout)) == 0 && std::memcmp(&a->P, &b->P, sizeof(DevParams)) == 0 &&
         a->t_now == b->t_now && a->epis == b->epis;
}

static int solve_impl(dftpav_batch *b, dftpav_batch *prev, bool chained) {
  if (!b || !b->uploaded || !b->have_corridor || prev == b) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  HIPCHK(h, hipSetDevice(h->device));
  b->pending = false; // a new solve of this batch supersedes whatever it had suspended
  if (prev && prev->pending && !(chained && chain_compatible(b, prev)))
    if (int rc = finish_pending(prev)) return rc;
  DevBatch D;
  if (int rc = sync_dev(b, D)) return rc;
  HIPCHK(h, hipEventRecord(b->ev0, h->stream));
  if (b->order == DFTPAV_ORDER_REFERENCE) {
    // (the QUAD shape always runs from the ring: its rows take a new trajectory as soon as one ends)
    if (b->ref_plan.wave && b->ref_plan.slots > 0 && b->ref_plan.slice > 0 && (b->ref_plan.quad || b->ref_plan.slots * (b->ref_plan.threads / 64) < b->B)) {
      // more trajectories than resident waves: persistent workgroups whose waves pop trajectories from the ring and run them a
      // slice of iterations at a time (solver_ref.hip); queue = all trajectories, flags cleared, counters reset on the stream
      HIPCHK(h, launch_ring_reset(D, h->stream));
      HIPCHK(h, launch_ref(b, D, kModeSolve, 1));
    } else {
      HIPCHK(h, launch_for(b, D, kModeSolve)); // every trajectory has its team from the start
    }
  } else if (!b->sched) {
    HIPCHK(h, launch_solver(D, b->d_dev, kModeSolve, b->threads, b->B, SchedArgs{0, 0, 0, nullptr}, h->stream));
  } else {
    // queue = all trajectories, flags cleared, counters reset: device-to-device, nothing waits on the host
    HIPCHK(h, hipMemcpyAsync(b->d_queue, b->d_iota, sizeof(int) * (size_t)b->B, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(b->d_sflag, 0, sizeof(int) * (size_t)b->B, h->stream));
    HIPCHK(h, hipMemcpyAsync(b->d_qctl, b->d_qctl + 8, sizeof(unsigned) * 8, hipMemcpyDeviceToDevice, h->stream));
    const int grid = b->slots < b->B ? b->slots : b->B;
    const bool adopt = chained && prev && prev->pending;
    if (adopt) {
      DevBatch Dprev;
      if (int rc = sync_dev(prev, Dprev)) return rc;
      HIPCHK(h, launch_adopt(D, Dprev, h->stream));
      HIPCHK(h, launch_solver(D, b->d_dev, kModeSolve, b->threads, grid, SchedArgs{1, b->slice, b->hand_over, prev->d_dev}, h->stream));
      // whatever of the adopted trajectories met this batch's end game: finished in the latency shape (normally none:
      // the workgroups of this launch leave at once)
      if (int rc = launch_stragglers(prev, Dprev, 3)) return rc;
      HIPCHK(h, hipEventRecord(prev->ev1, h->stream));
      prev->pending = false;
    } else {
      HIPCHK(h, launch_solver(D, b->d_dev, kModeSolve, b->threads, grid, SchedArgs{1, b->slice, b->hand_over, nullptr}, h->stream));
    }
    if (chained && b->hand_over > 0) {
      b->pending = true;
    } else if (b->hand_over > 0) {
      if (int rc = launch_stragglers(b, D, 2)) return rc;
    }
  }
  HIPCHK(h, hipEventRecord(b->ev1, h->stream));
  b->timed = true;
  b->solved = true;
  b->coef_override = false;
  return DFTPAV_OK;
}

extern "C" int dftpav_mark(dftpav_handle *h, int slot) {
  if (!h || slot < 0 || slot > 1) return DFTPAV_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->mark[slot]) HIPCHK(h, hipEventCreate(&h->mark[slot]));
  HIPCHK(h, hipEventRecord(h->mark[slot], h->stream));
  return DFTPAV_OK;
}

extern "C" int dftpav_marks_elapsed_ms(dftpav_handle *from, int from_slot, dftpav_handle *to, int to_slot, float *ms) {
  if (!from || !to || !ms || from_slot < 0 || from_slot > 1 || to_slot < 0 || to_slot > 1 || !from->mark[from_slot] ||
      !to->mark[to_slot] || from->device != to->device)
    return DFTPAV_E_INVALID;
  HIPCHK(to, hipSetDevice(to->device));
  HIPCHK(to, hipEventSynchronize(to->mark[to_slot]));
  HIPCHK(to, hipEventElapsedTime(ms, from->mark[from_slot], to->mark[to_slot]));
  return DFTPAV_OK;
}

extern "C" int dftpav_batch_set_hand_over(dftpav_batch *b, int hand_over) {
  if (!b) return DFTPAV_E_INVALID;
  if (int rc = finish_pending(b)) return rc;
  if (hand_over < 0) { // back to the plan's default: one trajectory per CU
    int n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, b->h->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    hand_over = n_cu;
  }
  b->hand_over = hand_over < b->B ? hand_over : b->B;
  return DFTPAV_OK;
}

extern "C" int dftpav_batch_solve_async(dftpav_batch *b) { return solve_impl(b, nullptr, false); }

extern "C" int dftpav_batch_solve_chained(dftpav_batch *b, dftpav_batch *prev) { return solve_impl(b, prev, true); }

extern "C" int dftpav_batch_finish(dftpav_batch *b) {
  if (!b) return DFTPAV_E_INVALID;
  return finish_pending(b);
}

extern "C" int dftpav_batch_sync(dftpav_batch *b) {
  if (!b) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  if (int rc = finish_pending(b)) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFTPAV_OK;
}

extern "C" int dftpav_batch_last_solve_ms(dftpav_batch *b, float *ms) {
  if (!b || !ms || !b->timed) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventSynchronize(b->ev1));
  HIPCHK(h, hipEventElapsedTime(ms, b->ev0, b->ev1));
  return DFTPAV_OK;
}

extern "C" int dftpav_batch_results(dftpav_batch *b, double *x, double *final_cost, int *status, int *success,
                                    int *iters, int *evals, long long *hist_sum, double *latency_us) {
  if (!b || !b->solved) return DFTPAV_E_INVALID; // nothing solved since the last upload
  dftpav_handle *h = b->h;
  const int B = b->B;
  if (int rc = finish_pending(b)) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (x) HIPCHK(h, hipMemcpy(x, b->d_x_out, sizeof(double) * (size_t)B * b->L.n, hipMemcpyDeviceToHost));
  if (final_cost) HIPCHK(h, hipMemcpy(final_cost, b->d_f, sizeof(double) * B, hipMemcpyDeviceToHost));
  if (status) HIPCHK(h, hipMemcpy(status, b->d_status, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (success) HIPCHK(h, hipMemcpy(success, b->d_success, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (iters) HIPCHK(h, hipMemcpy(iters, b->d_iters, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (evals) HIPCHK(h, hipMemcpy(evals, b->d_evals, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (hist_sum) HIPCHK(h, hipMemcpy(hist_sum, b->d_hist, sizeof(long long) * B, hipMemcpyDeviceToHost));
  if (latency_us) {
    std::vector<long long> t(B);
    HIPCHK(h, hipMemcpy(t.data(), b->d_ticks, sizeof(long long) * B, hipMemcpyDeviceToHost));
    for (int i = 0; i < B; i++) latency_us[i] = (double)t[i] * 0.01; // wall_clock64: 100 MHz
  }
  return DFTPAV_OK;
}

// ------------------------------------------------------------------ RCCL behind the C-ABI
// The one collective of the path: an all-gather of 16-byte result records over xGMI.  RCCL is loaded at the first use
// (dlopen by its soname: inside a process that already holds an RCCL -- PyTorch ships one -- this is that same copy, so a
// process never runs two), which keeps the library loadable where no RCCL is installed: only these entry points fail there.
namespace {
struct RcclUniqueId {
  char internal[DFTPAV_UNIQUE_ID_BYTES];
};
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(RcclUniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};
RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    api.GetUniqueId = reinterpret_cast<int (*)(RcclUniqueId *)>(dlsym(api.lib, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<int (*)(void **, int, RcclUniqueId, int)>(dlsym(api.lib, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(api.lib, "ncclCommDestroy"));
    api.AllGather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(dlsym(api.lib, "ncclAllGather"));
    api.GetErrorString = reinterpret_cast<const char *(*)(int)>(dlsym(api.lib, "ncclGetErrorString"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather;
  });
  return api;
}
constexpr int kNcclUint8 = 1; // ncclUint8 == ncclChar + 1 (rccl.h)
} // namespace
#define RCCLCHK(h, call)                                                                                        \
  do {                                                                                                          \
    int e_ = (call);                                                                                            \
    if (e_ != 0) {                                                                                              \
      (h)->err = std::string(#call) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e_) : "rccl error"); \
      return DFTPAV_E_COMM;                                                                                     \
    }                                                                                                           \
  } while (0)

extern "C" int dftpav_comm_unique_id(void *id) {
  if (!id) return DFTPAV_E_INVALID;
  if (!rccl().ok) return DFTPAV_E_COMM;
  RcclUniqueId u;
  if (rccl().GetUniqueId(&u) != 0) return DFTPAV_E_COMM;
  std::memcpy(id, u.internal, DFTPAV_UNIQUE_ID_BYTES);
  return DFTPAV_OK;
}
// Is RCCL loadable here?  (dlopen + dlsym only: no bootstrap root is started, unlike dftpav_comm_unique_id.)
extern "C" int dftpav_comm_available(void) { return rccl().ok ? 1 : 0; }
// lets go of h's reference to its communicator; g_comm_mu is held by the caller, h's stream is drained
static void comm_release_locked(dftpav_handle *h) {
  if (h->comm_ref && h->comm_ref->holders.fetch_sub(1) == 1) { // the last holder (every holder has drained its own stream)
    {
      std::lock_guard<std::mutex> lk(h->comm_ref->mu);
      (void)rccl().CommDestroy(h->comm_ref->comm);
    }
    delete h->comm_ref;
  }
  h->comm = nullptr;
  h->comm_ref = nullptr;
}
extern "C" int dftpav_comm_destroy(dftpav_handle *h) {
  if (!h) return DFTPAV_E_INVALID;
  if (h->comm) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    std::lock_guard<std::mutex> reg(g_comm_mu);
    comm_release_locked(h);
  }
  if (h->d_comm_send) (void)hipFree(h->d_comm_send);
  h->d_comm_send = nullptr;
  h->comm_send_bytes = 0;
  h->comm_ranks = 0;
  return DFTPAV_OK;
}
extern "C" int dftpav_comm_create(dftpav_handle *h, int nranks, int rank, const void *unique_id) {
  if (!h || nranks < 1 || rank < 0 || rank >= nranks || !unique_id) return DFTPAV_E_INVALID;
  if (!rccl().ok) {
    h->err = "RCCL (librccl.so.1) is not loadable";
    return DFTPAV_E_COMM;
  }
  if (int rc = dftpav_comm_destroy(h)) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  RcclUniqueId u;
  std::memcpy(u.internal, unique_id, DFTPAV_UNIQUE_ID_BYTES);
  RCCLCHK(h, rccl().CommInitRank(&h->comm, nranks, u, rank));
  {
    std::lock_guard<std::mutex> reg(g_comm_mu);
    h->comm_ref = new CommShared;
    h->comm_ref->comm = h->comm;
    h->comm_ref->holders.store(1);
  }
  h->comm_ranks = nranks;
  h->comm_rank = rank;
  return DFTPAV_OK;
}
// Several handles (= HIP streams) of one process on one communicator: a host that keeps k batches in flight on k handles sets
// ONE communicator up per rank instead of k (k ncclCommInitRank rendezvous and k sets of RCCL buffers per rank otherwise).
// RCCL orders successive operations of a communicator among the streams they are enqueued on; what the host owes it is the
// same order of collectives on every rank -- which a round-robin over the handles is.
extern "C" int dftpav_comm_share(dftpav_handle *h, dftpav_handle *owner) {
  if (!h || !owner || h == owner) return DFTPAV_E_INVALID;
  // h's stream is drained BEFORE the registry lock is taken (its collectives may still be in flight on the communicator it gives up)
  if (h->comm) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
  }
  std::lock_guard<std::mutex> reg(g_comm_mu);
  if (owner->comm && owner->comm_ref && owner->comm_ref == h->comm_ref) return DFTPAV_OK; // already the same communicator
  // the owner is checked and the new reference taken FIRST: a share that fails leaves h as it was (round 5 released h's own
  // communicator before the check -- a failed share could then destroy it on this rank alone and hang the other ranks)
  if (!owner->comm || !owner->comm_ref || owner->device != h->device) { // read under the lock: the owner may be letting go
    h->err = "dftpav_comm_share: the other handle needs a communicator (dftpav_comm_create, or shared itself) on the same device";
    return DFTPAV_E_INVALID;
  }
  owner->comm_ref->holders.fetch_add(1);
  if (h->comm) comm_release_locked(h); // ... only then the old one goes
  if (h->d_comm_send) (void)hipFree(h->d_comm_send); // (sized for the communicator it belonged to)
  h->d_comm_send = nullptr;
  h->comm_send_bytes = 0;
  h->comm = owner->comm;
  h->comm_ref = owner->comm_ref;
  h->comm_ranks = owner->comm_ranks;
  h->comm_rank = owner->comm_rank;
  return DFTPAV_OK;
}
extern "C" int dftpav_comm_layout(int global_B, int nranks, int rank, int *first, int *count, int *block) {
  if (global_B < 1 || nranks < 1 || rank < 0 || rank >= nranks) return DFTPAV_E_INVALID;
  const long long lo = (long long)global_B * rank / nranks, hi = (long long)global_B * (rank + 1) / nranks;
  if (first) *first = (int)lo;
  if (count) *count = (int)(hi - lo);
  if (block) *block = (global_B + nranks - 1) / nranks; // the largest shard: every rank's block in the gathered buffer
  return DFTPAV_OK;
}
extern "C" int dftpav_batch_allgather_results(dftpav_batch *b, int global_B, void *all_records) {
  if (!b || !all_records || !b->uploaded || !b->solved) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  if (!h->comm) {
    h->err = "dftpav_comm_create first";
    return DFTPAV_E_INVALID;
  }
  int first = 0, count = 0, block = 0;
  if (int rc = dftpav_comm_layout(global_B, h->comm_ranks, h->comm_rank, &first, &count, &block)) return rc;
  if (count != b->B) {
    h->err = "this batch is not the shard dftpav_comm_layout assigns to the rank";
    return DFTPAV_E_INVALID;
  }
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = finish_pending(b)) return rc;
  const size_t bytes = (size_t)block * 16;
  if (block <= b->B + 1) {
    // the send buffer IS the batch's record array (its epilogue-written records, one zero record of padding behind them for
    // the ranks whose shard is one short of the block): nothing of ours runs between the solve and the collective
    std::lock_guard<std::mutex> lk(h->comm_ref->mu);
    RCCLCHK(h, rccl().AllGather(b->d_records, all_records, bytes, kNcclUint8, h->comm, h->stream));
    return DFTPAV_OK;
  }
  if (h->comm_send_bytes < bytes) {
    if (h->d_comm_send) (void)hipFree(h->d_comm_send);
    h->d_comm_send = nullptr;
    HIPCHK(h, hipMalloc(&h->d_comm_send, bytes));
    h->comm_send_bytes = bytes;
  }
  HIPCHK(h, hipMemsetAsync(h->d_comm_send, 0, bytes, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_comm_send, b->d_records, (size_t)16 * count, hipMemcpyDeviceToDevice, h->stream));
  std::lock_guard<std::mutex> lk(h->comm_ref->mu);
  RCCLCHK(h, rccl().AllGather(h->d_comm_send, all_records, bytes, kNcclUint8, h->comm, h->stream));
  return DFTPAV_OK;
}

extern "C" int dftpav_batch_pack_results(dftpav_batch *b, void *device_dst) {
  if (!b || !device_dst || !b->solved) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  if (int rc = finish_pending(b)) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  // the records are written by the solver's epilogue as trajectories finish: a copy on the stream, no kernel of ours
  HIPCHK(h, hipMemcpyAsync(device_dst, b->d_records, (size_t)16 * b->B, hipMemcpyDeviceToDevice, h->stream));
  return DFTPAV_OK;
}

// The 16-byte records of the last solve on the host: waits for the solve; no work on the device behind it
extern "C" int dftpav_batch_records(dftpav_batch *b, void *host_dst) {
  if (!b || !host_dst || !b->solved) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  if (int rc = finish_pending(b)) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  // The epilogues wrote every record a second time into pinned host memory of the batch (DevBatch::records_host): when the stream has
  // drained they are there.  (A copy from the device -- to pageable or to pinned memory -- is a blit KERNEL of the runtime,
  // __amd_rocclr_copyBuffer in a kernel trace; in a stream of batches it queued for a CU behind the other streams' persistent waves and was
  // seen to take 240-300 ms per delivery on configs[1], the host's loop with it: round 6.)
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (b->h_records) {
    std::memcpy(host_dst, b->h_records, (size_t)16 * b->B);
    return DFTPAV_OK;
  }
  HIPCHK(h, hipMemcpyAsync(host_dst, b->d_records, (size_t)16 * b->B, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return DFTPAV_OK;
}

// test hook: the steps after the solve (dftpav_batch_validate, dftpav_batch_sample_states) on GIVEN coefficients [B][Ntot][6][2] and piece
// durations [B][M] instead of a solution's -- so that the kernels can be held against committed vectors of arbitrary trajectories
// (tests/golden/steps.npz).  Cleared by the next upload or solve.
extern "C" int dftpav_debug_batch_set_coeffs(dftpav_batch *b, const double *coeffs, const double *piece_dt) {
  if (!b || !coeffs || !piece_dt) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(b->d_coef, coeffs, sizeof(double) * (size_t)b->B * 12 * b->L.Ntot, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(b->d_dt, piece_dt, sizeof(double) * (size_t)b->B * b->L.M, hipMemcpyHostToDevice));
  b->coef_override = true;
  b->uploaded = true;
  b->solved = true;
  return DFTPAV_OK;
}
extern "C" int dftpav_batch_coeffs(dftpav_batch *b, double *coeffs, double *piece_dt) {
  if (!b || !b->uploaded || !b->solved) return DFTPAV_E_INVALID; // the coefficients are those of the solution x
  dftpav_handle *h = b->h;
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = finish_pending(b)) return rc;
  DevBatch D;
  if (int rc = sync_dev(b, D)) return rc;
  HIPCHK(h, launch_for(b, D, kModeCoeffs));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (coeffs)
    HIPCHK(h, hipMemcpy(coeffs, b->d_coef, sizeof(double) * (size_t)b->B * 12 * b->L.Ntot, hipMemcpyDeviceToHost));
  if (piece_dt) HIPCHK(h, hipMemcpy(piece_dt, b->d_dt, sizeof(double) * (size_t)b->B * b->L.M, hipMemcpyDeviceToHost));
  return DFTPAV_OK;
}

extern "C" int dftpav_batch_validate(dftpav_batch *b, double sample_dt, double vertex_res, int *collision, int *first_sample) {
  if (!b || !b->uploaded || !b->solved || !(sample_dt > 0.0) || !(vertex_res > 0.0)) return DFTPAV_E_INVALID; // nothing solved yet
  dftpav_handle *h = b->h;
  if (!h->d_cells) return DFTPAV_E_INVALID; // no map
  HIPCHK(h, hipSetDevice(h->device));
  // coefficients and piece durations of the solutions, regenerated on the device from x (as dftpav_batch_coeffs)
  if (int rc = finish_pending(b)) return rc;
  DevBatch D;
  if (int rc = sync_dev(b, D)) return rc;
  if (!b->coef_override) HIPCHK(h, launch_for(b, D, kModeCoeffs));
  // the two running sums of the reference, tabulated: sample times (traj_server_ros.cpp:387) and the spacing of the
  // outline points (shapes.cc:128)
  std::vector<double> tt, vv;
  {
    double t = 0.0;
    for (int k = 0; k < 4096; k++, t += sample_dt) tt.push_back(t);
    const double longest = std::max(h->params.veh_length, h->params.veh_width) + 1.0;
    for (double dl = vertex_res; dl < longest; dl += vertex_res) vv.push_back(dl);
    if (vv.empty()) vv.push_back(vertex_res);
  }
  double *d_t = nullptr, *d_v = nullptr;
  int *d_col = nullptr, *d_first = nullptr;
  int rc = DFTPAV_OK;
  auto chk = [&](hipError_t e) {
    if (e != hipSuccess && rc == DFTPAV_OK) {
      h->err = hipGetErrorString(e);
      rc = DFTPAV_E_HIP;
    }
  };
  chk(hipMalloc(&d_t, sizeof(double) * tt.size()));
  chk(hipMalloc(&d_v, sizeof(double) * vv.size()));
  chk(hipMalloc(&d_col, sizeof(int) * (size_t)b->B));
  chk(hipMalloc(&d_first, sizeof(int) * (size_t)b->B));
  if (rc == DFTPAV_OK) {
    chk(hipMemcpyAsync(d_t, tt.data(), sizeof(double) * tt.size(), hipMemcpyHostToDevice, h->stream));
    chk(hipMemcpyAsync(d_v, vv.data(), sizeof(double) * vv.size(), hipMemcpyHostToDevice, h->stream));
    chk(hipEventRecord(h->cev0, h->stream));
    chk(launch_validate(h->d_cells, h->map.size_x, h->map.size_y, h->map.resolution, h->map.origin_x, h->map.origin_y, b->d_coef,
                        b->d_dt, b->L, b->B, h->params.veh_width, h->params.veh_length, h->params.veh_d_cr, d_t, (int)tt.size(),
                        sample_dt, d_v, (int)vv.size(), d_col, d_first, h->stream));
    chk(hipEventRecord(h->cev1, h->stream));
    if (collision) chk(hipMemcpyAsync(collision, d_col, sizeof(int) * (size_t)b->B, hipMemcpyDeviceToHost, h->stream));
    if (first_sample) chk(hipMemcpyAsync(first_sample, d_first, sizeof(int) * (size_t)b->B, hipMemcpyDeviceToHost, h->stream));
    chk(hipStreamSynchronize(h->stream));
    h->ctimed = rc == DFTPAV_OK;
  }
  for (void *p : {(void *)d_t, (void *)d_v, (void *)d_col, (void *)d_first})
    if (p) (void)hipFree(p);
  return rc;
}

extern "C" int dftpav_batch_sample_states(dftpav_batch *b, double t0, double sample_dt, int n_samples, int filter_singularity,
                                          double *states, int *n_valid) {
  if (!b || !b->uploaded || !b->solved || !(sample_dt > 0.0) || n_samples <= 0 || !states) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = finish_pending(b)) return rc;
  DevBatch D;
  if (int rc = sync_dev(b, D)) return rc;
  if (!b->coef_override) HIPCHK(h, launch_for(b, D, kModeCoeffs));
  double *d_states = nullptr;
  int *d_valid = nullptr;
  int rc = DFTPAV_OK;
  auto chk = [&](hipError_t e) {
    if (e != hipSuccess && rc == DFTPAV_OK) {
      h->err = hipGetErrorString(e);
      rc = DFTPAV_E_HIP;
    }
  };
  if (!h->cev0) chk(hipEventCreate(&h->cev0));
  if (!h->cev1) chk(hipEventCreate(&h->cev1));
  const size_t nst = (size_t)b->B * (size_t)n_samples * 8;
  chk(hipMalloc(&d_states, sizeof(double) * nst));
  chk(hipMalloc(&d_valid, sizeof(int) * (size_t)b->B));
  if (rc == DFTPAV_OK) {
    chk(hipEventRecord(h->cev0, h->stream));
    chk(launch_states(b->d_coef, b->d_dt, b->L, b->B, h->params.veh_wheel_base, t0, sample_dt, n_samples, filter_singularity != 0,
                      d_states, d_valid, h->stream));
    chk(hipEventRecord(h->cev1, h->stream));
    chk(hipMemcpyAsync(states, d_states, sizeof(double) * nst, hipMemcpyDeviceToHost, h->stream));
    if (n_valid) chk(hipMemcpyAsync(n_valid, d_valid, sizeof(int) * (size_t)b->B, hipMemcpyDeviceToHost, h->stream));
    chk(hipStreamSynchronize(h->stream));
    h->ctimed = rc == DFTPAV_OK;
  }
  if (d_states) (void)hipFree(d_states);
  if (d_valid) (void)hipFree(d_valid);
  return rc;
}

// ------------------------------------------------- one planning cycle, stream-ordered
// TrajPlanner::RunMINCOParking from getRectangleConst on (traj_manager.cpp:551-626) and the consumers of its result
// (CheckReplan's collision re-check, traj_server_ros.cpp:385-397; the state playback, :244-259,335-356) as ONE enqueue:
// upload of the boundary states / waypoints / durations, then on the handle's stream and without the host in between
//   constraint-point poses -> rectangles of every hypothesis (corridor.hip) -> solve (solver.hip) -> coefficients of the
//   solutions -> collision re-check (validate.hip) -> state read-out (states.hip).
// dftpav_plan_cycle returns when everything is enqueued; dftpav_plan_cycle_fetch waits and copies the results out.
static int grow(dftpav_handle *h, void **p, size_t *have, size_t want, size_t elem) {
  if (*have >= want && *p) return DFTPAV_OK;
  if (*p) HIPCHK(h, hipFree(*p));
  *p = nullptr;
  HIPCHK(h, hipMalloc(p, elem * want));
  *have = want;
  return DFTPAV_OK;
}

extern "C" int dftpav_plan_cycle(dftpav_batch *b, const dftpav_batch_data *d, const double *states, int n_restarts, double check_dt,
                                 double vertex_res, double t0, double state_dt, int n_samples, int filter_singularity) {
  if (!b || !d || !states || n_restarts < 1 || b->B % n_restarts || !(check_dt > 0.0) || !(vertex_res > 0.0) || !(state_dt > 0.0) ||
      n_samples < 1)
    return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  if (!h->d_cells) return DFTPAV_E_INVALID;     // no map
  if (b->L.H != 4) return DFTPAV_E_UNSUPPORTED; // rectangles
  dftpav_batch_data dd = *d;
  dd.corridor = nullptr; // the half-planes come from the map
  if (int rc = dftpav_batch_upload(b, &dd)) return rc; // waits for the previous cycle of this handle, then copies the small inputs
  HIPCHK(h, hipSetDevice(h->device));
  auto &pc = b->pc;
  const int B = b->B, n_hyp = B / n_restarts;
  const size_t n_poses = (size_t)n_hyp * b->L.Npts;
  pc.poses.assign(states, states + 3 * n_poses);
  {
    void *p = pc.d_poses;
    if (int rc = grow(h, &p, &pc.n_poses, 3 * n_poses, sizeof(double))) return rc;
    pc.d_poses = (double *)p;
  }
  if (!h->cev0) HIPCHK(h, hipEventCreate(&h->cev0));
  if (!h->cev1) HIPCHK(h, hipEventCreate(&h->cev1));
  HIPCHK(h, hipMemcpyAsync(pc.d_poses, pc.poses.data(), sizeof(double) * 3 * n_poses, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, launch_corridor(h->d_cells, h->d_bits, h->map.size_x, h->map.size_y, h->map.resolution, h->map.origin_x, h->map.origin_y,
                            pc.d_poses, (int)n_poses, h->params.veh_width, h->params.veh_length, h->params.veh_d_cr, h->d_dl, h->n_dl,
                            nullptr, b->d_corridor, b->L.Npts, b->NptsPad, n_restarts, h->stream));
  b->have_corridor = true;
  b->cor_t_dirty = true;
  if (int rc = solve_impl(b, nullptr, false)) return rc;
  DevBatch D;
  if (int rc = sync_dev(b, D)) return rc;
  HIPCHK(h, launch_for(b, D, kModeCoeffs));
  // the two running sums of the reference, tabulated (as dftpav_batch_validate)
  pc.tt.clear();
  pc.vv.clear();
  {
    double t = 0.0;
    for (int k = 0; k < 4096; k++, t += check_dt) pc.tt.push_back(t);
    const double longest = std::max(h->params.veh_length, h->params.veh_width) + 1.0;
    for (double dl = vertex_res; dl < longest; dl += vertex_res) pc.vv.push_back(dl);
    if (pc.vv.empty()) pc.vv.push_back(vertex_res);
  }
  {
    void *p = pc.d_t;
    if (int rc = grow(h, &p, &pc.n_t, pc.tt.size(), sizeof(double))) return rc;
    pc.d_t = (double *)p;
    p = pc.d_v;
    if (int rc = grow(h, &p, &pc.n_v, pc.vv.size(), sizeof(double))) return rc;
    pc.d_v = (double *)p;
    size_t nb = pc.d_col ? (size_t)B : 0;
    p = pc.d_col;
    if (int rc = grow(h, &p, &nb, (size_t)B, sizeof(int))) return rc;
    pc.d_col = (int *)p;
    nb = pc.d_first ? (size_t)B : 0;
    p = pc.d_first;
    if (int rc = grow(h, &p, &nb, (size_t)B, sizeof(int))) return rc;
    pc.d_first = (int *)p;
    nb = pc.d_valid ? (size_t)B : 0;
    p = pc.d_valid;
    if (int rc = grow(h, &p, &nb, (size_t)B, sizeof(int))) return rc;
    pc.d_valid = (int *)p;
    p = pc.d_rd;
    if (int rc = grow(h, &p, &pc.n_rd, (size_t)B * n_samples * 8, sizeof(double))) return rc;
    pc.d_rd = (double *)p;
  }
  HIPCHK(h, hipMemcpyAsync(pc.d_t, pc.tt.data(), sizeof(double) * pc.tt.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(pc.d_v, pc.vv.data(), sizeof(double) * pc.vv.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, launch_validate(h->d_cells, h->map.size_x, h->map.size_y, h->map.resolution, h->map.origin_x, h->map.origin_y, b->d_coef, b->d_dt,
                            b->L, b->B, h->params.veh_width, h->params.veh_length, h->params.veh_d_cr, pc.d_t, (int)pc.tt.size(), check_dt,
                            pc.d_v, (int)pc.vv.size(), pc.d_col, pc.d_first, h->stream));
  HIPCHK(h, launch_states(b->d_coef, b->d_dt, b->L, b->B, h->params.veh_wheel_base, t0, state_dt, n_samples, filter_singularity != 0,
                          pc.d_rd, pc.d_valid, h->stream));
  pc.n_samples = n_samples;
  pc.in_flight = true;
  return DFTPAV_OK;
}

extern "C" int dftpav_plan_cycle_fetch(dftpav_batch *b, double *x, double *final_cost, int *status, int *success, int *iters, int *collision,
                                       int *first_sample, double *states, int *n_valid) {
  if (!b || !b->pc.in_flight) return DFTPAV_E_INVALID;
  dftpav_handle *h = b->h;
  if (int rc = dftpav_batch_results(b, x, final_cost, status, success, iters, nullptr, nullptr, nullptr)) return rc; // waits for the stream
  const size_t B = (size_t)b->B;
  if (collision) HIPCHK(h, hipMemcpy(collision, b->pc.d_col, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (first_sample) HIPCHK(h, hipMemcpy(first_sample, b->pc.d_first, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (states) HIPCHK(h, hipMemcpy(states, b->pc.d_rd, sizeof(double) * B * b->pc.n_samples * 8, hipMemcpyDeviceToHost));
  if (n_valid) HIPCHK(h, hipMemcpy(n_valid, b->pc.d_valid, sizeof(int) * B, hipMemcpyDeviceToHost));
  b->pc.in_flight = false;
  return DFTPAV_OK;
}

// ------------------------------------------------- serialised trajectories (include/dftpav_hip.h, "DPTJ" v1)
namespace {
constexpr size_t kWireHeader = 32, kWireSegment = 24, kWirePiece = 104;
template <class T> inline void put(unsigned char *&p, T v) {
  std::memcpy(p, &v, sizeof(T));
  p += sizeof(T);
}
template <class T> inline T get(const unsigned char *&p) {
  T v;
  std::memcpy(&v, p, sizeof(T));
  p += sizeof(T);
  return v;
}
} // namespace

extern "C" size_t dftpav_wire_size(int n_segments, const int *piece_nums) {
  if (n_segments <= 0 || n_segments > kMaxSeg || !piece_nums) return 0;
  size_t n = kWireHeader;
  for (int i = 0; i < n_segments; i++) {
    if (piece_nums[i] <= 0) return 0;
    n += kWireSegment + kWirePiece * (size_t)piece_nums[i];
  }
  return n;
}

extern "C" int dftpav_wire_pack(const dftpav_layout *layout, const double *coeffs, const double *piece_dt, int drone_id,
                                int traj_id, double start_time, void *buf, size_t capacity, size_t *written) {
  if (!layout || !coeffs || !piece_dt || !buf) return DFTPAV_E_INVALID;
  const size_t need = dftpav_wire_size(layout->M, layout->piece_nums);
  if (need == 0 || capacity < need) return DFTPAV_E_INVALID;
  unsigned char *p = (unsigned char *)buf;
  std::memcpy(p, "DPTJ", 4);
  p += 4;
  put<unsigned short>(p, 1);
  put<unsigned char>(p, 5);
  put<unsigned char>(p, 2);
  put<int>(p, drone_id);
  put<int>(p, traj_id);
  put<int>(p, layout->M);
  put<int>(p, 0);
  put<double>(p, start_time);
  double world = start_time; // addSingulTraj: each segment starts where the previous one ended
  int piece = 0;
  for (int i = 0; i < layout->M; i++) {
    const int N = layout->piece_nums[i];
    double dur = 0.0; // getTotalDuration: piece durations summed in order
    for (int q = 0; q < N; q++) dur += piece_dt[i];
    put<int>(p, layout->singuls ? layout->singuls[i] : 1);
    put<int>(p, N);
    put<double>(p, world);
    put<double>(p, dur);
    world = world + dur;
    for (int q = 0; q < N; q++, piece++) {
      put<double>(p, piece_dt[i]);
      const double *c = coeffs + (size_t)piece * 12; // [k][d], k = power
      for (int k = 5; k >= 0; k--) {                   // column 0 of CoefficientMat multiplies t^5
        put<double>(p, c[2 * k]);
        put<double>(p, c[2 * k + 1]);
      }
    }
  }
  if (written) *written = need;
  return DFTPAV_OK;
}

extern "C" int dftpav_wire_info(const void *buf, size_t size, int *drone_id, int *traj_id, double *start_time, int *n_segments,
                                int *n_pieces) {
  if (!buf || size < kWireHeader) return DFTPAV_E_INVALID;
  const unsigned char *p = (const unsigned char *)buf;
  if (std::memcmp(p, "DPTJ", 4) != 0) return DFTPAV_E_INVALID;
  p += 4;
  if (get<unsigned short>(p) != 1 || get<unsigned char>(p) != 5 || get<unsigned char>(p) != 2) return DFTPAV_E_INVALID;
  const int did = get<int>(p), tid = get<int>(p), M = get<int>(p);
  (void)get<int>(p);
  const double st = get<double>(p);
  if (M <= 0 || M > kMaxSeg) return DFTPAV_E_INVALID;
  size_t off = kWireHeader;
  int pieces = 0;
  for (int i = 0; i < M; i++) {
    if (size < off + kWireSegment) return DFTPAV_E_INVALID;
    const unsigned char *q = (const unsigned char *)buf + off;
    const int sg = get<int>(q), N = get<int>(q);
    if ((sg != 1 && sg != -1) || N <= 0 || N > 4096) return DFTPAV_E_INVALID;
    off += kWireSegment + kWirePiece * (size_t)N;
    pieces += N;
  }
  if (size < off) return DFTPAV_E_INVALID;
  if (drone_id) *drone_id = did;
  if (traj_id) *traj_id = tid;
  if (start_time) *start_time = st;
  if (n_segments) *n_segments = M;
  if (n_pieces) *n_pieces = pieces;
  return DFTPAV_OK;
}

extern "C" int dftpav_wire_unpack(const void *buf, size_t size, int *singuls, int *piece_nums, double *seg_start,
                                  double *seg_duration, double *durations, double *coeffs) {
  int M = 0;
  if (int rc = dftpav_wire_info(buf, size, nullptr, nullptr, nullptr, &M, nullptr)) return rc;
  const unsigned char *p = (const unsigned char *)buf + kWireHeader;
  int piece = 0;
  for (int i = 0; i < M; i++) {
    const int sg = get<int>(p), N = get<int>(p);
    const double st = get<double>(p), du = get<double>(p);
    if (singuls) singuls[i] = sg;
    if (piece_nums) piece_nums[i] = N;
    if (seg_start) seg_start[i] = st;
    if (seg_duration) seg_duration[i] = du;
    for (int q = 0; q < N; q++, piece++) {
      const double d = get<double>(p);
      if (durations) durations[piece] = d;
      for (int k = 0; k < 12; k++) {
        const double c = get<double>(p);
        if (coeffs) coeffs[(size_t)piece * 12 + k] = c;
      }
    }
  }
  return DFTPAV_OK;
}

extern "C" int dftpav_set_surround_wire(dftpav_handle *h, const void *const *bufs, const size_t *sizes, int S) {
  if (!h) return DFTPAV_E_INVALID;
  if (S <= 0) return dftpav_set_surround(h, nullptr);
  if (!bufs || !sizes) return DFTPAV_E_INVALID;
  std::vector<int> off(1, 0);
  std::vector<double> dur, coef, total, start;
  for (int s = 0; s < S; s++) {
    int M = 0, np = 0;
    double st = 0.0;
    if (int rc = dftpav_wire_info(bufs[s], sizes[s], nullptr, nullptr, &st, &M, &np)) return rc;
    std::vector<int> sg(M), pn(M);
    const size_t at = dur.size();
    dur.resize(at + np);
    coef.resize((at + np) * 12);
    if (int rc = dftpav_wire_unpack(bufs[s], sizes[s], sg.data(), pn.data(), nullptr, nullptr, dur.data() + at, coef.data() + at * 12))
      return rc;
    for (int i = 0; i < M; i++)
      if (sg[i] != 1) return DFTPAV_E_INVALID; // the obstacle model is forward-only (traj_manager.cpp:726,775)
    double tot = 0.0; // LocalTrajData::duration = Trajectory::getTotalDuration of the joined pieces
    for (int q = 0; q < np; q++) tot += dur[at + q];
    off.push_back((int)(at + np));
    total.push_back(tot);
    start.push_back(st);
  }
  dftpav_surround sur{S, off.data(), dur.data(), coef.data(), total.data(), start.data()};
  return dftpav_set_surround(h, &sur);
}

extern "C" int dftpav_solve_batch(dftpav_handle *h, const dftpav_layout *layout, int B, const dftpav_batch_data *d,
                                  double *x, double *final_cost, int *status, int *success, int *iters, int *evals) {
  dftpav_batch *b = nullptr;
  int rc = dftpav_batch_create(h, layout, B, &b);
  if (rc != DFTPAV_OK) return rc;
  rc = dftpav_batch_upload(b, d);
  if (rc == DFTPAV_OK) rc = dftpav_batch_solve_async(b);
  if (rc == DFTPAV_OK) rc = dftpav_batch_results(b, x, final_cost, status, success, iters, evals, nullptr, nullptr);
  dftpav_batch_destroy(b);
  return rc;
}

