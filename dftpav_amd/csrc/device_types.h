// device_types.h — PODs passed by value from the host C-ABI layer to the gfx950 kernels.
#pragma once
#include <cstdint>

#ifndef DFTPAV_HD
#if defined(__HIPCC__)
#define DFTPAV_HD __host__ __device__
#else
#define DFTPAV_HD
#endif
#endif

namespace dftpav {

constexpr int kMaxSeg = 8;        // gear segments per trajectory (trajnum)
constexpr int kWave = 64;         // CDNA wavefront
constexpr int kMaxThreads = 1024; // workgroup size limit

// Subset of dftpav_params the kernels read (traj_optimizer.cpp:1713-1736).
struct DevParams {
  double wei_obs, wei_surround, wei_feas, wei_time;
  double surround_clearance;
  double max_vel[2], max_acc[2], max_cur[2]; // [0]=forward (singul>0) [1]=backward, traj_optimizer.cpp:448-457
  double non_sinv, mini_T, fail_cost;
  double veh_length_infl;    // inflated length, gate of traj_optimizer.cpp:1393
  double vec_le[5][2];       // inflated footprint, first vertex repeated (traj_optimizer.cpp:1765-1775)
  // the four edges of that footprint as dynamicObsGradCostP forms them for every point (traj_optimizer.cpp:1419-1421,
  // 1466-1468): direction vec_le[e+1] - vec_le[e], its norm, 1 / norm.  Batch constants, filled by fill_footprint_edges.
  double edge_d[4][2], edge_len[4], edge_rlen[4];
  int gear_opt;
  // lbfgs_parameter_t (lbfgs.hpp:15-129) as set at traj_optimizer.cpp:127-134
  int mem_size, past, max_iterations, max_linesearch;
  double delta, g_epsilon, min_step, max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec;
};

// Structure shared by all trajectories of a batch.
struct DevLayout {
  int M, H, n, npad;           // npad = n rounded up to 64 (history row pitch)
  int Ntot, Npts, K, Kd, Kmax; // pieces, constraint points, resolutions
  int rhs_tot;                 // sum_i (N_i + 5): non-zero RHS rows of the MINCO systems
  int piece_nums[kMaxSeg], singuls[kMaxSeg];
  int seg_piece0[kMaxSeg + 1]; // first global piece of segment
  int seg_x0[kMaxSeg];         // offset of P_i inside x
  int seg_rhs0[kMaxSeg + 1];   // offset of the segment's RHS rows
  int seg_pt0[kMaxSeg + 1];    // first constraint point of segment
  int x_tau0, x_gear0, x_ang0; // offsets of tau | gear xy | gear angle inside x
};

struct DevSurround {
  int S;
  const int *piece_off;    // [S+1]
  const double *durations; // [np]
  const double *coeffs;    // [np][12], col 0 multiplies t^5 (poly_traj_utils.hpp:993)
  const double *total;     // [S]
  const double *start;     // [S]
  // [np] or nullptr.  theta[k] of an obstacle = the largest t for which Trajectory::locatePieceIdx (poly_traj_utils.hpp:510-528,
  // a walk of dependent subtractions t -= duration) stops at piece k or earlier, found on the host by bisection over the
  // doubles with that same walk (capi.cpp, build_theta): the piece index is then a search in theta, and only the local
  // time is formed by the reference's subtractions.  Same result for every t by construction; nullptr = walk.
  const double *theta;
  // [np][4] or nullptr: {x_min, x_max, y_min, y_max} of a box that contains piece k of an obstacle over its whole duration
  // (the hull of its Bernstein coefficients, widened by 1e-6 m; capi.cpp, build_piece_boxes).  Only ever used to skip a
  // (constraint point, obstacle) pair whose distance is certainly above the gate of traj_optimizer.cpp:1393, before the
  // obstacle's position is evaluated: with or without the table the gate passes the same pairs.
  const double *bbox;
  // the accessors traj_math.h uses (the kernels have a second view of the same tables with pointers that carry the
  // LDS address space, solver.hip: SurLds)
  DFTPAV_HD bool has_theta() const { return theta != nullptr; }
  DFTPAV_HD bool has_bbox() const { return bbox != nullptr; }
  DFTPAV_HD void load_box(int k, double bb[4]) const {
    for (int i = 0; i < 4; i++) bb[i] = bbox[4 * (size_t)k + i];
  }
  DFTPAV_HD bool far_from_piece(int k, const double sigma[2], double r) const {
    if (bbox == nullptr) return false;
    const double *bb = bbox + 4 * (size_t)k;
    return sigma[0] < bb[0] - r || sigma[0] > bb[1] + r || sigma[1] < bb[2] - r || sigma[1] > bb[3] + r;
  }
  // pieces per second of obstacle u: only the starting guess of the search in theta (any value gives the same index)
  DFTPAV_HD double rate(int u) const { return (double)(piece_off[u + 1] - piece_off[u]) / total[u]; }
  DFTPAV_HD inline void end_state(int u, double pd[2], double vd[2], double ad[2]) const; // traj_math.h
  DFTPAV_HD void load_piece(int k, double c[12]) const {
    const double *cm = coeffs + 12 * (size_t)k;
    for (int i = 0; i < 12; i++) c[i] = cm[i];
  }
};

// ---- how the constraint points of a trajectory are mapped onto the lanes of a workgroup (solver.hip, E4)
// The K+1 points of a piece are taken in groups of G consecutive points, G = 32 or 16 lanes of one wave: a group's 14
// per-piece sums (12 entries of gdC, gdT, cost) are formed by a cross-lane tree inside the wave, no LDS round trip.
// What does not fill a group (point 33 of a 33-point piece; every point of a piece shorter than 16) is a "leftover":
// evaluated in densely packed lanes, its contributions staged in LDS and chained onto the piece in point order.
// Per piece and output the sum is   start value + group 0 + group 1 + ... + leftovers in point order.
inline int e4_group_size(int points) { return points >= 32 ? 32 : (points >= 16 ? 16 : 0); }

// Everything one launch needs. Device pointers.
struct DevBatch {
  DevLayout L;
  DevParams P;
  int B;
  // problem data (resident after dftpav_batch_upload)
  const double *x0;       // [B][n]
  const double *iniS;     // [B][M][6] clamped (traj_optimizer.cpp:55-76)
  const double *finS;     // [B][M][6]
  const double *corridor; // [B][H*4][NptsPad] normalised, component-major for coalesced loads
  int NptsPad;
  // layout tables
  const int16_t *pt_piece; // [Npts] global piece index of a constraint point
  const int16_t *pt_j;     // [Npts] sample index j inside the piece
  const double *opM[kMaxSeg];  // A_N^{-1} restricted to the N+5 non-zero RHS rows, [6N][N+5] row-major
  const double *opMT[kMaxSeg]; // its transpose [N+5][6N]
  int op_in_lds;               // operators are staged in LDS at kernel start
  int cor_in_lds;              // the trajectory's half-planes are staged in LDS at kernel start
  // E4 lane plan for this launch shape (see e4_group_size): rounds of blockDim slots
  int e4_rounds, e4_groups, e4_left, e4_lcap; // rounds; groups and leftover points per trajectory; leftover capacity of the LDS staging
  const int *e4_gtab;          // [e4_groups] piece | (first j) << 16 of a group
  const int *e4_ltab;          // [e4_left] piece | j << 16 of a leftover point
  const int *e4_wave;          // [e4_rounds][blockDim / 64][3] per wave: kind (32, 16: group size; 0: leftovers; -1: idle), base (first
                               // group id / first leftover index), count (groups / leftover points it holds)
  const int *e4_round;         // [e4_rounds][2] leftovers evaluated in the round (0 = none: no chain pass after it), index of the first
  const int *e4_piece;         // [Ntot][4] first group, groups, first leftover, leftovers of a piece
  int op_off[kMaxSeg];         // offset (doubles) of each segment's operator inside the LDS copy
  DevSurround sur;
  int sur_np;                  // pieces of all moving obstacles together (their durations are staged in LDS)
  int sur_coef_lds;            // their 2x6 coefficient blocks are staged in LDS as well (chosen per launch shape so that
                               // it does not cost a resident workgroup, capi.cpp: sur_coef_in_lds)
  double t_now, epis;
  // L-BFGS history workspace (lm_s, lm_y of lbfgs.hpp:512-513), one slab per trajectory
  double *histS, *histY; // one buffer [B][mem][npad][2]: (s, y) interleaved per element, histY == histS + 1
  // products of neighbouring stored pairs, [B][mem][8] (solver.hip, two_loop_lane):
  //   histU[j][d] = s_j . y_(the d+1-th pair after j),  histV[j][d] = y_j . s_(the d+1-th pair before j)
  double *histU, *histV;
  double *histR; // [B][mem][2] (ys, 1 / ys) of the stored pairs (lbfgs.hpp:685 lm_ys and its reciprocal)
  // time-sliced scheduling of batches larger than the device holds at once (solver.hip, solver_kernel)
  int *queue;          // [B] ring of trajectories waiting for a workgroup
  unsigned *qctl;      // [0] head  [1] published tail  [2] reserved tail  [3] unfinished  [4] number of stragglers
  int *stragglers;     // [B] trajectories handed to the follow-up launch (count in qctl[4])
  int *stragglers2;    // [B] adopted trajectories a chained launch could not finish (count in qctl[5])
  int qcap;            // entries of `queue` (B own trajectories + room for adopted ones)
  double *state;       // [B][state_stride] solver state of a suspended trajectory
  int *sflag;          // [B] 0 fresh, 1 suspended (state valid), 2 finished
  int state_stride;
  // in/out
  const double *x_in; // eval mode: [B][n]
  double *x_out;      // [B][n]
  double *f_out;      // [B]
  double *g_out;      // eval mode: [B][n]
  double *f_eval;     // eval mode: [B] (apart from f_out, which holds the final costs of the last solve)
  int *status, *success, *iters, *evals;
  long long *hist_sum;
  // [B + 1] 16-byte result records {f64 final cost, i32 status, i32 iterations} (what the all-gather carries),
  // written by the solver's epilogue when a trajectory finishes -- no packing kernel between the solve and the collective
  unsigned char *records;
  // the same records once more in PINNED HOST memory (or nullptr): a single-GPU caller reads them there after the solve without a copy on
  // the device -- the runtime's device-to-host copy of 64 KB is a blit kernel, and in a stream of batches a kernel of several waves queues for
  // a CU behind the other streams' persistent waves (capi.cpp: dftpav_batch_records)
  unsigned char *records_host;
  long long *ticks; // per-trajectory solve time in wall_clock64 ticks (100 MHz)
  long long *prof;  // optional [B][12] shader-clock phase profile (nullptr = off)
  double *coef_out; // [B][Ntot][6][2]
  double *dt_out;   // [B][M]
  // optional record of every evaluation of trajectories trace_b .. trace_b + trace_n - 1 (dftpav_batch_trace_range; nullptr =
  // off), per trajectory: 8 doubles of header ([0] = records written), then trace_cap records of 3 npad + 8 doubles: x, g, d,
  // {f, stp, k, count}
  double *trace;
  int trace_b, trace_cap, trace_n;
};

enum KernelMode { kModeSolve = 0, kModeEval = 1, kModeCoeffs = 2 };

// how a solve launch picks its trajectories
constexpr int kAltTag = 1 << 30; // queue entry of a trajectory that belongs to SchedArgs::alt, not to the launched batch
struct SchedArgs {
  int source;     // 0: workgroup i solves trajectory i   1: pops from DevBatch::queue until it is empty
                  // 2: workgroup i resumes DevBatch::stragglers[i] (i < qctl[4])   3: ... stragglers2[i] (i < qctl[5])
  int slice;      // iterations after which an unfinished trajectory is suspended (source 1; 0 = never)
  int hand_over;  // source 1: once this few trajectories are unfinished, suspended ones go to `stragglers`
  const DevBatch *alt; // source 1, chained solves: descriptor of the previous batch, whose stragglers sit in
                       // this batch's queue tagged with kAltTag (same layout, parameters and launch shape)
};
// doubles of solver state per suspended trajectory
inline int solver_state_doubles(const DevLayout &L, const DevParams &P) { (void)P; return 5 * L.npad + 24 + 8; }

// launch shape of a reference-order batch (solver_ref.hip: reference_order_plan)
struct RefPlan {
  int wave;      // 1: one wave per trajectory, several per workgroup (throughput); 0: one workgroup per trajectory (latency)
  int quad;      // 1 (with wave = 1): FOUR trajectories per wave, one per row of 16 lanes (solver_ref4.hip)
  int threads;   // workgroup size
  int wg_per_cu; // WAVE shape: resident workgroups per CU
  int slots;     // WAVE shape: persistent workgroups of a scheduled solve
  int slice;     // WAVE shape: iterations after which an unfinished trajectory goes back to the ring
  int slots_wide; // QUAD shape: persistent workgroups of a launch that has the device to itself (dftpav_batch_set_hand_over != 0)
  int hand;       // QUAD shape, such a launch: unfinished trajectories at which its waves leave theirs to a follow-up launch in the WAVE shape
  size_t lds;    // dynamic LDS per workgroup
};

// host-side E4 lane plan of a layout for a workgroup size (tables of DevBatch::e4_*)
struct E4Sizes {
  int rounds, groups, left, lcap;
};
E4Sizes e4_sizes(const DevLayout &L, int threads);
// size in bytes of the dynamic LDS a launch needs
constexpr int kSurCoefLds = 170; // most obstacle pieces whose coefficient blocks are ever staged in LDS (16 KB)
size_t solver_lds_bytes(const DevLayout &L, const DevParams &P, int threads, bool op_lds, bool cor_lds, int sur_np, bool sur_coef = false);
// picks the workgroup size for a layout; shape 0/1/2 = at most one / two / more trajectories per CU
int solver_threads(const DevLayout &L, int shape);

} // namespace dftpav
