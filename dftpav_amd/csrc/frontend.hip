// Front-end resampling on the device: from a searched path to the arguments of
// getRectangleConst and OptimizeTrajectory.
//
//   KinoAstar::getKinoNode, from SampleTraj on    traj_planner/src/kino_astar.cpp:606-743
//   KinoAstar::evaluateDuration / evaluateLength  traj_planner/src/kino_astar.cpp:744-795
//   KinoAstar::evaluatePos                        traj_planner/src/kino_astar.cpp:468-521
//   KinoAstar::getFlatState                       traj_planner/src/kino_astar.cpp:834-857
//   TrajPlanner::RunMINCOParking, resampling      traj_planner/src/traj_manager.cpp:531-568
//
// One workgroup per hypothesis.  Thread 0 walks the path once: gear segmentation (a segment ends where the
// direction of travel flips against the heading), trapezoid time allocation, the 0.1 s sampling loop of
// getKinoNode (only the sum of its time stamps survives into RunMINCOParking, but whether a sample is kept
// depends on the walk, so the walk is replayed) and the piece count / duration of every segment.  Then all
// threads resample: one constraint-point pose per thread, each an independent evaluatePos — locate the
// segment, invert the trapezoid profile, walk the segment's arc length, interpolate.  Every running sum of
// the reference (arc length along the path, res_time, basetime) is rebuilt by the same additions in the
// same order.  fp64, no contraction, portable cos / sin: bit-identical to oracle/frontend_oracle.cpp in order 1.
#include <hip/hip_runtime.h>

#include "../../include/dftpav_hip.h"
#include "device_types.h"
#include "traj_math.h"

namespace dftpav {

constexpr int kFeMaxSeg = 16;
constexpr double kPi = 3.14159265358979323846; // M_PI

struct FeArgs {
  dftpav_frontend_params fp;
  const double *paths;
  const int *path_len;
  int max_path;
  const double *start_states, *end_states, *start_ctrl;
  int n_hyp;
  dftpav_frontend_out out; // device pointers
};

__device__ inline double fe_duration(double length, double max_vel, double max_acc, double startV, double endV) {
  const double startv2 = startV * startV, endv2 = endV * endV, maxv2 = max_vel * max_vel;
  const double critical_len = (maxv2 - startv2) / (2 * max_acc) + (maxv2 - endv2) / (2 * max_acc);
  if (length >= critical_len) return (max_vel - startV) / max_acc + (max_vel - endV) / max_acc + (length - critical_len) / max_vel;
  const double tmpv = sqrt(0.5 * (startv2 + endv2 + 2 * max_acc * length));
  return (tmpv - startV) / max_acc + (tmpv - endV) / max_acc;
}
__device__ inline double fe_length(double curt, double locallength, double max_vel, double max_acc, double startV, double endV) {
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
  const double tmpv = sqrt(0.5 * (startv2 + endv2 + 2 * max_acc * locallength));
  const double tmpt = (tmpv - startV) / max_acc;
  if (curt <= tmpt) return startV * curt + 0.5 * max_acc * (curt * curt);
  return startV * tmpt + 0.5 * max_acc * (tmpt * tmpt) + tmpv * (curt - tmpt) - 0.5 * max_acc * ((curt - tmpt) * (curt - tmpt));
}
__device__ inline double fe_norm2(const double *a, const double *b) {
  const double dx = b[0] - a[0], dy = b[1] - a[1];
  return sqrt(dx * dx + dy * dy);
}
__device__ inline void fe_flat_state(const dftpav_frontend_params &fp, const double *pose, double v, double steer, double accel,
                                     int singul, double *out) {
  const double angle = pose[2];
  double c, s;
  crt::sincos(angle, s, c); // (the reference: libm; here correctly rounded, as oracle order 2)
  const double ns = -s;
  double vel = fabs(v) <= fp.non_siguav ? singul * fp.non_siguav : singul * v;
  out[0] = pose[0];
  out[1] = pose[1];
  out[2] = c * vel + ns * 0.0;
  out[3] = s * vel + c * 0.0;
  const double lat = crt::tan(steer) / fp.wheel_base * (vel * vel);
  out[4] = c * accel + ns * lat;
  out[5] = s * accel + c * lat;
}

__global__ void __launch_bounds__(256) frontend_kernel(FeArgs A) {
  __shared__ int s_index[kFeMaxSeg + 1], s_S[kFeMaxSeg], s_pieces[kFeMaxSeg], s_first[kFeMaxSeg + 1];
  __shared__ double s_len[kFeMaxSeg], s_time[kFeMaxSeg], s_dt[kFeMaxSeg], s_base[kFeMaxSeg];
  __shared__ double s_total;
  __shared__ int s_ns;
  const int h = blockIdx.x, tid = threadIdx.x;
  const dftpav_frontend_params &fp = A.fp;
  const dftpav_frontend_out &O = A.out;
  const int MS = O.max_seg, MP = O.max_pieces, MST = O.max_states;
  const double *P = A.paths + (size_t)h * A.max_path * 3;
  const int n = A.path_len[h];
  const double startvel = fabs(A.start_states[4 * h + 3]), endvel = fabs(A.end_states[4 * h + 3]);
  if (tid == 0) {
    // ---- gear segmentation and time allocation, kino_astar.cpp:618-665
    int ns = 0;
    double tmpl = 0;
    auto dir = [&](int i) {
      const double dx = P[3 * (i + 1)] - P[3 * i], dy = P[3 * (i + 1) + 1] - P[3 * i + 1];
      double cy, sy;
      crt::sincos(P[3 * i + 2], sy, cy);
      return dx * cy + dy * sy >= 0 ? 1 : -1;
    };
    auto dur = [&](double len, int S, double v0, double v1) {
      return S > 0 ? fe_duration(len, fp.max_forward_vel, fp.max_forward_acc, v0, v1)
                   : fe_duration(len, fp.max_backward_vel, fp.max_backward_acc, v0, v1);
    };
    int lastS = dir(0);
    s_index[0] = 0;
    for (int i = 0; i < n - 1; i++) {
      const int curS = dir(i);
      if (curS * lastS >= 0) {
        tmpl += fe_norm2(P + 3 * i, P + 3 * (i + 1));
      } else {
        if (ns < kFeMaxSeg) {
          s_index[ns + 1] = i;
          s_S[ns] = lastS;
          s_len[ns] = tmpl;
          s_time[ns] = dur(tmpl, lastS, fp.non_siguav, fp.non_siguav);
        }
        ns++;
        tmpl = fe_norm2(P + 3 * i, P + 3 * (i + 1));
      }
      lastS = curS;
    }
    if (ns < kFeMaxSeg) {
      s_S[ns] = lastS;
      s_len[ns] = tmpl;
      s_time[ns] = dur(tmpl, lastS, fp.non_siguav, fp.non_siguav);
      s_index[ns + 1] = n - 1;
    }
    ns++;
    O.n_seg[h] = ns;
    if (ns > kFeMaxSeg || ns > MS) ns = 0; // more gear changes than the solve path takes: nothing is produced
    s_ns = ns;
    if (ns >= 2) {
      s_time[0] = dur(s_len[0], s_S[0], startvel, fp.non_siguav);
      s_time[ns - 1] = dur(s_len[ns - 1], s_S[ns - 1], fp.non_siguav, endvel);
    } else if (ns == 1) {
      s_time[0] = dur(s_len[0], s_S[0], startvel, endvel);
    }
    // ---- per segment: the 0.1 s walk (sum of the kept time stamps), boundary states, piece count and duration
    double basetime = 0.0;
    int first = 0;
    for (int i = 0; i < ns; i++) {
      double initv = fp.non_siguav, finv = fp.non_siguav;
      double steer = 0.0, accel = 0.0;
      if (i == 0) {
        initv = startvel;
        steer = A.start_ctrl[2 * h];
        accel = A.start_ctrl[2 * h + 1];
      }
      if (i == ns - 1) finv = endvel;
      const int sig = s_S[i], f0 = s_index[i], f1 = s_index[i + 1];
      const double locallength = s_len[i], T = s_time[i];
      const double mv = sig > 0 ? fp.max_forward_vel : fp.max_backward_vel, ma = sig > 0 ? fp.max_forward_acc : fp.max_backward_acc;
      double samplet, tmparc = 0, init_total = 0.0;
      int index = 0;
      double sampletime = 0.1;
      if (T <= sampletime) sampletime = T / 2.0;
      for (samplet = sampletime; samplet < T; samplet += sampletime) {
        const double arc = fe_length(samplet, locallength, mv, ma, initv, finv);
        for (int k = index; k < f1 - f0; k++) {
          const double seg = fe_norm2(P + 3 * (f0 + k), P + 3 * (f0 + k + 1));
          tmparc += seg;
          if (tmparc >= arc) {
            index = k;
            init_total += sampletime; // traj_pts.push_back(.., sampletime), summed in order by RunMINCOParking
            tmparc -= seg;
            break;
          }
        }
      }
      init_total += T - (samplet - sampletime);
      O.singul[(size_t)h * MS + i] = sig;
      fe_flat_state(fp, P + 3 * f0, initv, steer, accel, sig, O.ini_states + ((size_t)h * MS + i) * 6);
      fe_flat_state(fp, P + 3 * f1, finv, 0.0, 0.0, sig, O.fin_states + ((size_t)h * MS + i) * 6);
      // traj_manager.cpp:543-546
      int piece_nums = (int)(init_total / fp.piece_duration + 0.5);
      piece_nums = piece_nums > 2 ? piece_nums : 2;
      const double dtp = init_total / piece_nums;
      s_pieces[i] = piece_nums;
      s_dt[i] = dtp;
      s_base[i] = basetime;
      s_first[i] = first;
      first += (piece_nums - 2) * (fp.traj_res + 1) + 2 * (fp.dense_traj_res + 1);
      O.piece_nums[(size_t)h * MS + i] = piece_nums;
      O.piece_dt[(size_t)h * MS + i] = dtp;
      O.n_states[(size_t)h * MS + i] = (piece_nums - 2) * (fp.traj_res + 1) + 2 * (fp.dense_traj_res + 1);
      basetime += init_total;
    }
    s_first[ns] = first;
    double tot = 0.0; // totalTrajTime, kino_astar.cpp:739-742
    for (int i = 0; i < ns; i++) tot += s_time[i];
    s_total = tot;
  }
  __syncthreads();
  const int ns = s_ns;
  const int total_states = ns > 0 ? s_first[ns] : 0;
  // ---- resampling, traj_manager.cpp:551-568: one evaluatePos per thread
  for (int q = tid; q < total_states; q += blockDim.x) {
    int i = 0;
    while (i + 1 < ns && q >= s_first[i + 1]) i++;
    const int local = q - s_first[i];
    const int pieces = s_pieces[i], K = fp.traj_res, Kd = fp.dense_traj_res;
    int j, k;
    if (local < Kd + 1) {
      j = 0;
      k = local;
    } else if (local < (Kd + 1) + (pieces - 2) * (K + 1)) {
      j = 1 + (local - (Kd + 1)) / (K + 1);
      k = (local - (Kd + 1)) - (j - 1) * (K + 1);
    } else {
      j = pieces - 1;
      k = local - (Kd + 1) - (pieces - 2) * (K + 1);
    }
    const int resolution = (j == 0 || j == pieces - 1) ? Kd : K;
    const double dtp = s_dt[i];
    double res_time = 0;
    for (int jj = 0; jj < j; jj++) res_time += dtp; // res_time += ego_piece_dur_vec[i]
    double t = s_base[i] + res_time + 1.0 * k / resolution * dtp;
    // ---- KinoAstar::evaluatePos, kino_astar.cpp:468-521
    t = fmin(fmax(0.0, t), s_total);
    int index = -1;
    double tmpT = 0, CutTime = 0;
    for (int s = 0; s < ns; s++) {
      tmpT += s_time[s];
      if (tmpT >= t) {
        index = s;
        CutTime = t - tmpT + s_time[s];
        break;
      }
    }
    double initv = fp.non_siguav, finv = fp.non_siguav;
    if (index == 0) initv = startvel;
    if (index == ns - 1) finv = endvel;
    const int front = s_index[index], back = s_index[index + 1];
    const int sig = s_S[index];
    const double arclength = fe_length(CutTime, s_len[index], sig > 0 ? fp.max_forward_vel : fp.max_backward_vel,
                                       sig > 0 ? fp.max_forward_acc : fp.max_backward_acc, initv, finv);
    double pos[3] = {P[3 * back], P[3 * back + 1], P[3 * back + 2]};
    double tmparc = 0;
    for (int p = front; p < back; p++) {
      const double *a = P + 3 * p, *b = P + 3 * (p + 1);
      const double l = fe_norm2(a, b);
      tmparc += l;
      if (tmparc >= arclength) {
        const double l1 = tmparc - arclength, l2 = l - l1;
        for (int d = 0; d < 3; d++) pos[d] = l1 / l * a[d] + l2 / l * b[d];
        if (fabs(b[2] - a[2]) >= kPi) {
          if (b[2] <= 0) pos[2] = l1 / l * a[2] + l2 / l * (b[2] + 2 * kPi);
          else if (a[2] <= 0) pos[2] = l1 / l * (a[2] + 2 * kPi) + l2 / l * b[2];
        }
        break;
      }
    }
    if (local < MST) {
      double *st = O.states + (((size_t)h * MS + i) * (size_t)MST + local) * 3;
      st[0] = pos[0];
      st[1] = pos[1];
      st[2] = pos[2];
    }
    if (k == resolution && j != pieces - 1 && j < MP - 1) {
      double *inner = O.inner_pts + ((size_t)h * MS + i) * (size_t)(MP - 1) * 2;
      inner[2 * j] = pos[0];
      inner[2 * j + 1] = pos[1];
    }
  }
}

hipError_t launch_frontend(const dftpav_frontend_params &fp, const double *paths, const int *path_len, int max_path,
                           const double *start_states, const double *end_states, const double *start_ctrl, int n_hyp,
                           const dftpav_frontend_out &out, hipStream_t stream) {
  FeArgs A{fp, paths, path_len, max_path, start_states, end_states, start_ctrl, n_hyp, out};
  hipLaunchKernelGGL(frontend_kernel, dim3(n_hyp), dim3(256), 0, stream, A);
  return hipGetLastError();
}

} // namespace dftpav
