// ORACLE -- TEST INFRASTRUCTURE ONLY.  Stand-in for the one PCL point type the reference uses
// (include/tools.hpp:22 `typedef pcl::PointXYZINormal PointType;`): same field names, float32.
#ifndef BALM_COMPAT_PCL_POINT_TYPES
#define BALM_COMPAT_PCL_POINT_TYPES
namespace pcl {
struct PointXYZINormal {
  union { float data[4]; struct { float x, y, z; }; };
  union { float data_n[4]; struct { float normal_x, normal_y, normal_z; }; };
  union { struct { float intensity, curvature; }; float data_c[4]; };
  PointXYZINormal() {
    data[0] = data[1] = data[2] = 0; data[3] = 1;
    data_n[0] = data_n[1] = data_n[2] = data_n[3] = 0;
    data_c[0] = data_c[1] = data_c[2] = data_c[3] = 0;
  }
};
}  // namespace pcl
#endif
