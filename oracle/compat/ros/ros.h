// ORACLE -- TEST INFRASTRUCTURE ONLY.  The only ROS symbol bavoxel.hpp touches is ros::Time::now()
// inside the dead `left_evaluate` (src/benchmark/bavoxel.hpp:183,275); benchmark_virtual.cpp also
// names NodeHandle / Publisher / init / spin in its display code and main(), which are never run
// by the oracle (empty shells below).
#ifndef BALM_COMPAT_ROS
#define BALM_COMPAT_ROS
#include <chrono>
#include <string>
namespace ros {
struct Time {
  double t;
  static Time now() { Time x; x.t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); return x; }
  double toSec() const { return t; }
};
struct Publisher {
  template <class M> void publish(const M &) const {}
};
struct NodeHandle {
  template <class M> Publisher advertise(const std::string &, int) { return Publisher(); }
  template <class T> void param(const std::string &, T &out, const T &dflt) { out = dflt; }
};
inline void init(int &, char **, const std::string &) {}
inline void spin() {}
}  // namespace ros
#endif
