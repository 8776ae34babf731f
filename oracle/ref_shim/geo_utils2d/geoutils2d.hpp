// Shadows geo_utils2d/geoutils2d.hpp (vertex enumeration of polytopes via quickhull + sdlp).  traj_optimizer.h:11 includes
// it for the inline helper extractVs (traj_optimizer.h:188-235), which nothing on the solve path calls; the declaration
// below lets that helper compile, calling it aborts.  TEST INFRASTRUCTURE for oracle/_ref.
#pragma once
#include <Eigen/Eigen>
#include <cstdlib>
namespace geoutils {
inline bool enumerateVs(const Eigen::MatrixXd &, Eigen::MatrixXd &) { std::abort(); }
}  // namespace geoutils
