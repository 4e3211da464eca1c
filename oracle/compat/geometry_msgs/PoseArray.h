// ORACLE -- TEST INFRASTRUCTURE ONLY.  Plain structs for benchmark_virtual.cpp's display code.
#ifndef BALM_COMPAT_GEOMETRY_MSGS_POSEARRAY
#define BALM_COMPAT_GEOMETRY_MSGS_POSEARRAY
#include <sensor_msgs/PointCloud2.h>
#include <vector>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseArray { std_msgs::Header header; std::vector<Pose> poses; };
}  // namespace geometry_msgs
#endif
