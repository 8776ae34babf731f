// Shadows geo_utils2d/root_finder.hpp (polynomial root finding; needs Eigen's FFT and eigenvalue modules).  The solve path
// includes it through poly_traj_utils.hpp:14 and calls nothing in it.  TEST INFRASTRUCTURE for oracle/_ref.
#pragma once
