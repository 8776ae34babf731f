// oracle/step_trig.h -- TEST INFRASTRUCTURE: the libm calls of the step oracles (corridor / validate / states / frontend / fit), by order.
//   order 0: libm, as the reference calls it (bits that belong to the host: glibc is not correctly rounded);
//   order 1: the functions the HIP kernels call (dftpav_amd/csrc/cr_trig.h, double-double) compiled for the host -- the device's
//            program replayed, for locating a difference;
//   order 2: the CORRECTLY ROUNDED functions, from binary128 (libquadmath) -- an implementation that shares nothing with the
//            kernels'.  The kernels' results must equal order 2's bit for bit: discrete outputs (rectangles, collision flags,
//            first colliding sample, valid counts) included.
#pragma once
#include <cmath>
#include <quadmath.h>

#include "../dftpav_amd/csrc/cr_trig.h"

namespace step_trig {
struct Trig {
  int order;
  double cos(double a) const { return order == 0 ? std::cos(a) : (order == 1 ? dftpav::crt::cos(a) : (double)cosq((__float128)a)); }
  double sin(double a) const { return order == 0 ? std::sin(a) : (order == 1 ? dftpav::crt::sin(a) : (double)sinq((__float128)a)); }
  double tan(double a) const { return order == 0 ? std::tan(a) : (order == 1 ? dftpav::crt::tan(a) : (double)tanq((__float128)a)); }
  double atan(double a) const { return order == 0 ? std::atan(a) : (order == 1 ? dftpav::crt::atan(a) : (double)atanq((__float128)a)); }
  double atan2(double y, double x) const {
    return order == 0 ? std::atan2(y, x) : (order == 1 ? dftpav::crt::atan2(y, x) : (double)atan2q((__float128)y, (__float128)x));
  }
  double cube(double v) const { // the reference: pow(v, 3)
    return order == 0 ? std::pow(v, 3) : (order == 1 ? dftpav::crt::cube_cr(v) : (double)((__float128)v * (__float128)v * (__float128)v));
  }
};
} // namespace step_trig
