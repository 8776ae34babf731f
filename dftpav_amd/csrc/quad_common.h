// quad_common.h -- row-of-16-lanes primitives of the QUAD shapes of the reference order (solver_ref4.hip: one gear segment;
// solver_ref4m.hip: several): a row of a wave is a trajectory, its sixteen lanes are its pieces.
#pragma once
#include "ref_order_common.h"

namespace dftpav {
namespace reford {

// acc + v[lane 0 of the row] + v[lane 1] + ... + v[lane 15]: sixteen dependent additions, every lane of a row ends with its
// row's sum.  (fma(v, 1.0, acc) == acc + v rounded once.  A VALU write followed by a DPP read needs two wait states and the asm
// block is opaque to the hazard recogniser: s_nop 1 in front.)
__device__ __forceinline__ double row_chain16(double acc, double v) {
  const double one = 1.0;
  asm volatile("s_nop 1\n\t" DFTPAV_FMAC_BCAST16 : "+v"(acc) : "v"(v), "v"(one));
  return acc;
}
// acc + v[lane 0 of the row]
__device__ __forceinline__ double row_add_lane0(double acc, double v) {
  const double one = 1.0;
  asm volatile("s_nop 1\n\t" DFTPAV_FMAC_BCAST(0) : "+v"(acc) : "v"(v), "v"(one));
  return acc;
}
// max over the 16 lanes of a row (order-free), the same value in every lane of the row
__device__ __forceinline__ double row_max16(double v) {
  v = fmax(v, mov_dpp<0xB1>(v));
  v = fmax(v, mov_dpp<0x4E>(v));
  v = fmax(v, mov_dpp<0x141>(v));
  v = fmax(v, mov_dpp<0x140>(v));
  return v;
}
// the neighbour's value: lane l - 1 (row_shr:1) / lane l + 1 (row_shl:1) of the same row; 0.0 where there is none
template <int CTRL> __device__ __forceinline__ double nb_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}


// DENSE emission (round 6): the active terms the lanes of a wave find in the point loop are listed -- (owner lane, point, term, the
// point's running offset s1, the half-plane of a corridor term) -- and evaluated kDense at a time, one per lane, by whichever lanes
// the list numbers: the owner's coefficients come over by ds_bpermute, the point's state is formed again by the same expressions
// (the same bits), what the term adds goes back through LDS and the owner adds its terms in list order = (point, term) order.  In
// place, a lane with k active terms ran k trips of the emission while the other 63 waited: 26-29 trips of ~2.2 k cycles per evaluation
// of four trajectories with three lanes active on average (55-65 k cycles, 15 % of a pass).
// (solver_ref4m.hip, several gear segments: tried and not kept -- its kernel has no registers left for the list's code: 64 bytes of
// scratch per lane and 165 -> 171 ms per step of configs[1]'s stream)
constexpr int kDense = 24;
struct DenseLds {
  ldsi_t id;   // [kDense] owner lane | term << 6 | point << 12
  ldsd_t s1;   // [kDense]
  ldsd_t pl;   // [kDense][4] the half-plane of a corridor term
  ldsd_t out;  // [kDense][15] gdC (12), gdT, corridor cost, feasibility cost
};
__host__ __device__ inline size_t q4_dense_bytes() { return ((size_t)kDense * (4 + 8 + 32 + 120) + 15) & ~(size_t)15; }
__device__ inline void q4_carve_dense(DenseLds &d, char *base) {
  ldsd_t p = (ldsd_t)reinterpret_cast<double *>(base);
  d.s1 = p; p += kDense;
  d.pl = p; p += 4 * kDense;
  d.out = p; p += 15 * kDense;
  d.id = (ldsi_t)p;
}

} // namespace reford
} // namespace dftpav
