// Stand-in for common/basics/semantics.h: only common::VehicleParam, with the defaults of the reference
// (src/Sim/core/common/inc/common/basics/semantics.h:32-76).  The real header drags in the simulator's lane / shape /
// state library, none of which the solve path touches.  TEST INFRASTRUCTURE for oracle/_ref.
#pragma once
#include "common/state/state.h"
namespace common {
class VehicleParam {
 public:
  double width() const { return width_; }
  double length() const { return length_; }
  double wheel_base() const { return wheel_base_; }
  double front_suspension() const { return front_suspension_; }
  double rear_suspension() const { return rear_suspension_; }
  double max_steering_angle() const { return max_steering_angle_; }
  double max_longitudinal_acc() const { return max_longitudinal_acc_; }
  double max_lateral_acc() const { return max_lateral_acc_; }
  double d_cr() const { return d_cr_; }
  void set_width(const double v) { width_ = v; }
  void set_length(const double v) { length_ = v; }
  void set_wheel_base(const double v) { wheel_base_ = v; }
  void set_front_suspension(const double v) { front_suspension_ = v; }
  void set_rear_suspension(const double v) { rear_suspension_ = v; }
  void set_max_steering_angle(const double v) { max_steering_angle_ = v; }
  void set_max_longitudinal_acc(const double v) { max_longitudinal_acc_ = v; }
  void set_max_lateral_acc(const double v) { max_lateral_acc_ = v; }
  void set_d_cr(const double v) { d_cr_ = v; }
 private:
  double width_ = 1.90;
  double length_ = 4.88;
  double wheel_base_ = 2.85;
  double front_suspension_ = 0.93;
  double rear_suspension_ = 1.10;
  double max_steering_angle_ = 45.0;
  double max_longitudinal_acc_ = 2.0;
  double max_lateral_acc_ = 2.0;
  double d_cr_ = 1.015;
};
}  // namespace common
