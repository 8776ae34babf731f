// Rectangle safe-corridor generation on the device: the step that produces the
// half-planes the solve path consumes.
//
//   TrajPlanner::getRectangleConst                 traj_planner/src/traj_manager.cpp:1213-1469
//   TrajPlannerAdapter::CheckIfCollisionUsingLine  traj_planner/src/map_adapter.cpp:117-129
//   GridMapND::CheckIfEqualUsingGlobalPosition     common/src/common/basics/semantics.cc:169-179,214-221
//
// One wavefront per state (x, y, yaw).  The growth of the rectangle is a sequential decision process
// (side after side, one cell at a time, at most 4 x 34 steps), but every decision is "does any sample
// of three line segments fall into an occupied cell": the lanes take the samples (64 per pass, the
// long edge needs up to three passes) and one ballot answers it.  The sample offsets dl = 0, +checkl,
// +checkl, ... are the reference's running sum, tabulated once on the host so that lane k sees
// exactly the k-th value of that sum.  The map is byte work out of L2 (a 120 m x 120 m map at 0.3 m
// is 160 KB); the kernel is bound by the ~400 dependent decisions per state, not by bandwidth.
// fp64 throughout, no contraction: bit-identical to oracle/corridor_oracle.cpp in order 1.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "traj_math.h"

namespace dftpav {

struct CorridorArgs {
  const unsigned char *cells;
  const unsigned *bits; // the same map, one bit per cell (set = OCCUPIED), or nullptr when it does not fit in LDS
  int size_x, size_y;
  double resolution, origin_x, origin_y;
  double res_rcp; // 1.0 / resolution
  const double *states; // [n][3]
  int n;
  double veh_width, veh_length, veh_dcr;
  const double *dl; // running sum 0, checkl, checkl + checkl, ...
  int n_dl;
  double *hpoly; // [n][4][4], or nullptr:
  // the solve path's own layout, [trajectory][4 * plane + component][NptsPad] with unit normals
  // (traj_optimizer.cpp:49-52), state i being point i % Npts of trajectory i / Npts
  double *batch_cor;
  int Npts, NptsPad;
  int replicate; // every trajectory i / Npts is written `replicate` times: trajectories t * replicate + r (restarts share a corridor)
};

// GridMapND::CheckIfEqualUsingGlobalPosition(p, OCCUPIED): coord = round((p - origin) / resolution), out of range
// counts as free (semantics.cc:169-179, 214-221).  BITS: the map is the 1-bit-per-cell copy staged in LDS.
// a / b from the correctly rounded reciprocal y = 1/b (Markstein): the correctly rounded quotient, the same
// bits as a / b (checked on 2^31 pairs on gfx950, see solver.hip), in 3 instructions instead of ~14
__device__ inline double div_by_rcp(double a, double b, double y) {
  double q0 = a * y;
  double r = __builtin_fma(-b, q0, a);
  return __builtin_fma(r, y, q0);
}
template <bool BITS>
__device__ inline bool cell_occupied(const CorridorArgs &A, const unsigned *bits, double x, double y) {
  const double cx = round(div_by_rcp(x - A.origin_x, A.resolution, A.res_rcp));
  const double cy = round(div_by_rcp(y - A.origin_y, A.resolution, A.res_rcp));
  if (!(cx >= 0.0 && cx < (double)A.size_x && cy >= 0.0 && cy < (double)A.size_y)) return false;
  const int idx = (int)cx + A.size_x * (int)cy;
  if (BITS) return (bits[idx >> 5] >> (idx & 31)) & 1u;
  return A.cells[idx] == 80; // GridMapND::OCCUPIED
}

// One growth step asks whether any sample of point1 -> newpoint1 -> newpoint2 -> point2 is occupied
// (three CheckIfCollisionUsingLine calls, map_adapter.cpp:117-129: samples at dl = 0, checkl, 2 checkl, ... < length,
// then the end point).  The answer is the OR over all samples, so they are taken together: the two short
// segments (one cell long) and the three end points in the first pass, the long edge 64 samples per pass.
template <bool BITS>
__device__ inline bool strip_hits(const CorridorArgs &A, const unsigned *bits, double p1x, double p1y, double n1x, double n1y,
                                  double n2x, double n2y, double p2x, double p2y, int lane) {
  // ---- pass 0: lanes 0..15 segment point1->newpoint1, 16..31 segment newpoint2->point2 (sample k = lane & 15),
  //              lanes 32, 33, 34 the end points newpoint1, newpoint2, point2
  {
    const bool second = (lane & 16) != 0;
    const double ax = second ? n2x : p1x, ay = second ? n2y : p1y, bx = second ? p2x : n1x, by = second ? p2y : n1y;
    const double dx = bx - ax, dy = by - ay;
    const double norm = sqrt(dx * dx + dy * dy), nrcp = 1.0 / norm;
    const int k = lane & 15;
    bool hit = false;
    if (lane < 32) {
      const double dl = k < A.n_dl ? A.dl[k] : norm;
      if (dl < norm)
        hit = cell_occupied<BITS>(A, bits, div_by_rcp(dx * dl, norm, nrcp) + ax, div_by_rcp(dy * dl, norm, nrcp) + ay);
    } else if (lane == 32) {
      hit = cell_occupied<BITS>(A, bits, n1x, n1y);
    } else if (lane == 33) {
      hit = cell_occupied<BITS>(A, bits, n2x, n2y);
    } else if (lane == 34) {
      hit = cell_occupied<BITS>(A, bits, p2x, p2y);
    }
    if (__ballot(hit) != 0) return true;
  }
  // ---- the long edge newpoint1 -> newpoint2 (a short segment is one cell = 2 checkl long, traj_manager.cpp:1218,
  //      1316: it never has more than 3 samples, so pass 0 covered it)
  {
    const double dx = n2x - n1x, dy = n2y - n1y;
    const double norm = sqrt(dx * dx + dy * dy), nrcp = 1.0 / norm;
    for (int base = 0; base < A.n_dl; base += 64) {
      const int k = base + lane;
      const double dl = k < A.n_dl ? A.dl[k] : norm;
      const bool active = dl < norm;
      if (__ballot(active) == 0) break;
      bool hit = false;
      if (active)
        hit = cell_occupied<BITS>(A, bits, div_by_rcp(dx * dl, norm, nrcp) + n1x, div_by_rcp(dy * dl, norm, nrcp) + n1y);
      if (__ballot(hit) != 0) return true;
    }
  }
  return false;
}

template <bool BITS>
__global__ void __launch_bounds__(256) corridor_kernel(CorridorArgs A) {
  extern __shared__ unsigned lds_bits[];
  const int lane = threadIdx.x & 63;
  if (BITS) { // the whole map, one bit per cell
    const int words = (A.size_x * A.size_y + 31) >> 5;
    for (int w = threadIdx.x; w < words; w += blockDim.x) lds_bits[w] = A.bits[w];
    __syncthreads();
  }
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= A.n) return;
  const double rx = A.states[3 * i], ry = A.states[3 * i + 1], yaw = A.states[3 * i + 2];
  double c, s;
  crt::sincos(yaw, s, c); // (the reference: libm cos / sin; here the correctly rounded ones, as oracle order 2)
  const double ns = -s; // egoR = [c -s; s c], traj_manager.cpp:1233-1234
  const double step = A.resolution * 1.0, limit = 10.0; // :1218-1219
  const double dcr = A.veh_dcr;
  double sx = rx, sy = ry, W = A.veh_width, L = A.veh_length; // sourcePt, sourceVp
  double expand[4] = {0.0, 0.0, 0.0, 0.0};
  int open = 0xF;
  while (open) { // NotFinishTable.norm() > 0
#pragma unroll
    for (int side = 0; side < 4; side++) {
      if (!(open & (1 << side))) continue;
      double a1, b1, a2, b2, na1, nb1, na2, nb2; // body coordinates of point1, point2, newpoint1, newpoint2
      if (side == 0) { // +dy, :1311-1315
        a1 = L / 2.0 + dcr; b1 = W / 2.0; a2 = -L / 2.0 + dcr; b2 = W / 2.0;
        na1 = L / 2.0 + dcr; nb1 = W / 2.0 + step; na2 = -L / 2.0 + dcr; nb2 = W / 2.0 + step;
      } else if (side == 1) { // +dx, :1343-1347
        a1 = L / 2.0 + dcr; b1 = -W / 2.0; a2 = L / 2.0 + dcr; b2 = W / 2.0;
        na1 = step + L / 2.0 + dcr; nb1 = -W / 2.0; na2 = step + L / 2.0 + dcr; nb2 = W / 2.0;
      } else if (side == 2) { // -dy, :1375-1379
        a1 = -L / 2.0 + dcr; b1 = -W / 2.0; a2 = L / 2.0 + dcr; b2 = -W / 2.0;
        na1 = -L / 2.0 + dcr; nb1 = -W / 2.0 - step; na2 = L / 2.0 + dcr; nb2 = -W / 2.0 - step;
      } else { // -dx, :1407-1411
        a1 = -L / 2.0 + dcr; b1 = W / 2.0; a2 = -L / 2.0 + dcr; b2 = -W / 2.0;
        na1 = -L / 2.0 + dcr - step; nb1 = W / 2.0; na2 = -L / 2.0 + dcr - step; nb2 = -W / 2.0;
      }
      const double p1x = sx + (c * a1 + ns * b1), p1y = sy + (s * a1 + c * b1);
      const double p2x = sx + (c * a2 + ns * b2), p2y = sy + (s * a2 + c * b2);
      const double n1x = sx + (c * na1 + ns * nb1), n1y = sy + (s * na1 + c * nb1);
      const double n2x = sx + (c * na2 + ns * nb2), n2y = sy + (s * na2 + c * nb2);
      if (strip_hits<BITS>(A, lds_bits, p1x, p1y, n1x, n1y, n2x, n2y, p2x, p2y, lane)) {
        open &= ~(1 << side);
        continue;
      }
      expand[side] += step;
      if (expand[side] >= limit) { // the centre / size update is skipped on the closing step, :1332-1335
        open &= ~(1 << side);
        continue;
      }
      double ma, mb;
      if (side == 0) { ma = 0.0; mb = step / 2.0; W = W + step; }
      else if (side == 1) { ma = step / 2.0; mb = 0.0; L = L + step; }
      else if (side == 2) { ma = 0.0; mb = -step / 2.0; W = W + step; }
      else { ma = -step / 2.0; mb = 0.0; L = L + step; }
      const double nx = sx + (c * ma + ns * mb), ny = sy + (s * ma + c * mb);
      sx = nx;
      sy = ny;
    }
  }
  if (lane == 0) { // traj_manager.cpp:1442-1465: (normal; point) columns from the RAW pose and size
    double Hl[16];
    double *H = A.hpoly ? A.hpoly + 16 * (size_t)i : Hl;
    const double W0 = A.veh_width, L0 = A.veh_length;
    double a, b;
    a = L0 / 2.0 + dcr + expand[1]; b = W0 / 2.0 + expand[0];
    H[0] = -s; H[1] = c; H[2] = rx + (c * a + ns * b); H[3] = ry + (s * a + c * b);
    a = L0 / 2.0 + dcr + expand[1]; b = -W0 / 2.0 - expand[2];
    H[4] = c; H[5] = s; H[6] = rx + (c * a + ns * b); H[7] = ry + (s * a + c * b);
    a = -L0 / 2.0 + dcr - expand[3]; b = -W0 / 2.0 - expand[2];
    H[8] = s; H[9] = -c; H[10] = rx + (c * a + ns * b); H[11] = ry + (s * a + c * b);
    a = -L0 / 2.0 + dcr - expand[3]; b = W0 / 2.0 + expand[0];
    H[12] = -c; H[13] = -s; H[14] = rx + (c * a + ns * b); H[15] = ry + (s * a + c * b);
    if (!A.hpoly) { // what dftpav_batch_upload does with a host corridor: normalise, component-major
      const int t = i / A.Npts, pt = i - t * A.Npts;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double nrm = sqrt(H[4 * k] * H[4 * k] + H[4 * k + 1] * H[4 * k + 1]);
        H[4 * k] = H[4 * k] / nrm;
        H[4 * k + 1] = H[4 * k + 1] / nrm;
      }
      for (int r = 0; r < A.replicate; r++) {
        double *dst = A.batch_cor + ((size_t)t * A.replicate + r) * 16 * A.NptsPad + pt;
#pragma unroll
        for (int k = 0; k < 16; k++) dst[(size_t)k * A.NptsPad] = H[k];
      }
    }
  }
}

hipError_t launch_corridor(const unsigned char *cells, const unsigned *bits, int size_x, int size_y, double resolution, double origin_x, double origin_y,
                           const double *states, int n, double veh_width, double veh_length, double veh_dcr, const double *dl,
                           int n_dl, double *hpoly, double *batch_cor, int Npts, int NptsPad, int replicate, hipStream_t stream) {
  CorridorArgs A{cells, bits, size_x, size_y, resolution, origin_x, origin_y, 1.0 / resolution, states, n, veh_width, veh_length, veh_dcr, dl, n_dl,
                 hpoly, batch_cor, Npts, NptsPad, replicate};
  const int waves_per_block = 4;
  const dim3 grid((n + waves_per_block - 1) / waves_per_block), block(64 * waves_per_block);
  if (bits) {
    const size_t lds = (((size_t)size_x * size_y + 31) / 32) * sizeof(unsigned);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&corridor_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(corridor_kernel<true>, grid, block, lds, stream, A);
  } else {
    hipLaunchKernelGGL(corridor_kernel<false>, grid, block, 0, stream, A);
  }
  return hipGetLastError();
}

// The set-up of OptimizeTrajectory on the corridor (traj_optimizer.cpp:15,49-52): a private copy whose normals are
// normalised, here written straight into the solver's layout [trajectory][plane * 4 + component][point] (lanes = points read
// contiguous doubles).  raw: the caller's hPoly columns [B][Npts][H][4] (n_x, n_y, p_x, p_y) as uploaded.  One thread per
// (trajectory, point, plane); the division and the square root are the IEEE operations the host code used before
// (contraction off), so the bits are those of the former host loop.
__global__ void corridor_layout_kernel(const double *__restrict__ raw, double *__restrict__ out, int B, int Npts, int H, int NptsPad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * Npts * H;
  if (i >= total) return;
  const int k = (int)(i % H);
  const size_t tp = i / H;
  const int pt = (int)(tp % Npts);
  const size_t t = tp / Npts;
  const double *col = raw + i * 4;
  const double c0 = col[0], c1 = col[1], c2 = col[2], c3 = col[3];
  const double nrm = sqrt(c0 * c0 + c1 * c1);
  double *dst = out + t * (size_t)H * 4 * NptsPad + (size_t)(4 * k) * NptsPad + pt;
  dst[0] = c0 / nrm;
  dst[(size_t)NptsPad] = c1 / nrm;
  dst[(size_t)2 * NptsPad] = c2;
  dst[(size_t)3 * NptsPad] = c3;
}
hipError_t launch_corridor_layout(const double *raw, double *out, int B, int Npts, int H, int NptsPad, hipStream_t stream) {
  const size_t total = (size_t)B * Npts * H;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(corridor_layout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, raw, out, B, Npts, H, NptsPad);
  return hipGetLastError();
}

} // namespace dftpav