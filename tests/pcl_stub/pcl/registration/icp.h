// TEST-ONLY stand-in for the slice of pcl::IterativeClosestPoint that pclomp/gicp_omp.h touches (see
// registration.h in this directory).  Not part of the product.
#pragma once
#include <pcl/registration/registration.h>

#ifndef PCL_ERROR
#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
#endif

namespace pcl {
template <class PointSource, class PointTarget>
class IterativeClosestPoint : public Registration<PointSource, PointTarget> {
 public:
  IterativeClosestPoint() : corr_dist_threshold_(std::sqrt(std::numeric_limits<double>::max())) {}
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  double getMaxCorrespondenceDistance() const { return corr_dist_threshold_; }

 protected:
  double corr_dist_threshold_;
};
}  // namespace pcl
