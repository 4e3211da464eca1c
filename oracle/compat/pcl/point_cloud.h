// ORACLE -- TEST INFRASTRUCTURE ONLY.  Stand-in for pcl::PointCloud<T> (container surface only).
#ifndef BALM_COMPAT_PCL_POINT_CLOUD
#define BALM_COMPAT_PCL_POINT_CLOUD
#include <cstdint>
#include <memory>
#include <vector>
namespace pcl {
template <class P>
class PointCloud {
 public:
  typedef std::shared_ptr<PointCloud<P>> Ptr;
  typedef std::shared_ptr<const PointCloud<P>> ConstPtr;
  std::vector<P> points;
  uint32_t width = 0, height = 0;
  void push_back(const P &p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = height = 0; }
  void reserve(size_t n) { points.reserve(n); }
  void swap(PointCloud &o) { points.swap(o.points); std::swap(width, o.width); std::swap(height, o.height); }
  P &operator[](size_t i) { return points[i]; }
  const P &operator[](size_t i) const { return points[i]; }
  PointCloud &operator+=(const PointCloud &o) { points.insert(points.end(), o.points.begin(), o.points.end()); width = (uint32_t)points.size(); height = 1; return *this; }
  typename std::vector<P>::iterator begin() { return points.begin(); }
  typename std::vector<P>::iterator end() { return points.end(); }
  typename std::vector<P>::const_iterator begin() const { return points.begin(); }
  typename std::vector<P>::const_iterator end() const { return points.end(); }
};
}  // namespace pcl
#endif
