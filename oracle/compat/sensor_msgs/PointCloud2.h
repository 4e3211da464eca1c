// ORACLE -- TEST INFRASTRUCTURE ONLY.  Empty shell: benchmark_virtual.cpp's display code names it.
#ifndef BALM_COMPAT_SENSOR_MSGS_PC2
#define BALM_COMPAT_SENSOR_MSGS_PC2
#include <ros/ros.h>
#include <string>
namespace std_msgs { struct Header { std::string frame_id; ros::Time stamp; }; }
namespace sensor_msgs { struct PointCloud2 { std_msgs::Header header; }; }
#endif
