// oracle/shim: minimal PCL stand-in (TEST INFRASTRUCTURE): only the cloud container PlaneMatcher.cpp iterates
#pragma once
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZRGB { float x, y, z; unsigned rgba; };
template <typename T> struct PointCloud {
    typedef std::shared_ptr<PointCloud<T>> Ptr;
    std::vector<T> points;
};
}  // namespace pcl
