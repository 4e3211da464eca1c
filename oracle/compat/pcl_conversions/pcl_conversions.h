// ORACLE -- TEST INFRASTRUCTURE ONLY.  Empty shell of pcl::toROSMsg (display code only).
#ifndef BALM_COMPAT_PCL_CONVERSIONS
#define BALM_COMPAT_PCL_CONVERSIONS
#include <pcl/point_cloud.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl { template <class C> void toROSMsg(const C &, sensor_msgs::PointCloud2 &) {} }
#endif
