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

} // namespace reford
} // namespace dftpav
