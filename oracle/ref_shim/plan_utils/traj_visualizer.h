// Shadows plan_utils/traj_visualizer.h (rviz publishing; needs tf, decomp_util, decomp_ros_utils).  traj_optimizer.h:15
// includes it only for the decomp types that PolyTrajOptimizer::displayBugPoly (traj_optimizer.cpp:1895-1914) names; that
// function is never called on the solve path.  TEST INFRASTRUCTURE for oracle/_ref.
#pragma once
#include <Eigen/Eigen>
#include <vector>
#include <visualization_msgs/Marker.h>
#include "plan_utils/traj_container.hpp"
template <typename T> using vec_E = std::vector<T>;
struct Hyperplane2D {
  Hyperplane2D() = default;
  Hyperplane2D(const Eigen::Vector2d &p, const Eigen::Vector2d &n) : p_(p), n_(n) {}
  Eigen::Vector2d p_, n_;
};
struct Polyhedron2D {
  void add(const Hyperplane2D &h) { hs.push_back(h); }
  std::vector<Hyperplane2D> hs;
};
namespace decomp_ros_msgs {
struct PolyhedronArray { std_msgs::Header header; };
}  // namespace decomp_ros_msgs
namespace DecompROS {
inline decomp_ros_msgs::PolyhedronArray polyhedron_array_to_ros(const vec_E<Polyhedron2D> &) { return decomp_ros_msgs::PolyhedronArray(); }
}  // namespace DecompROS
