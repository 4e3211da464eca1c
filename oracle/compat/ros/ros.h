// ORACLE -- TEST INFRASTRUCTURE ONLY.  The only ROS symbol bavoxel.hpp touches is ros::Time::now()
// inside the dead `left_evaluate` (src/benchmark/bavoxel.hpp:183,275).
#ifndef BALM_COMPAT_ROS
#define BALM_COMPAT_ROS
#include <chrono>
namespace ros {
struct Time {
  double t;
  static Time now() { Time x; x.t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); return x; }
  double toSec() const { return t; }
};
}  // namespace ros
#endif
