// Stand-in for common/state/state.h (reference: src/Sim/core/common/inc/common/state/state.h:7-33): the same fields,
// without the simulator's basics.h.  TEST INFRASTRUCTURE for oracle/_ref.
#pragma once
#include <Eigen/Eigen>
typedef double decimal_t;
namespace common {
struct State {
  decimal_t time_stamp{0.0};
  Eigen::Vector2d vec_position{Eigen::Vector2d::Zero()};
  decimal_t angle{0.0};
  decimal_t curvature{0.0};
  decimal_t velocity{0.0};
  decimal_t acceleration{0.0};
  decimal_t steer{0.0};
};
}  // namespace common
