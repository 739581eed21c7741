// TEST-ONLY stand-in for the slice of pcl::VoxelGrid that pclomp/voxel_grid_covariance_omp.h touches (see
// ../registration/registration.h: written from the public API documentation, never installed).
#pragma once
#include <pcl/registration/registration.h>

namespace pcl {
template <class PointT>
class VoxelGrid {
 public:
  typedef PointCloud<PointT> PointCloudT;
  typedef typename PointCloudT::Ptr PointCloudPtr;
  typedef typename PointCloudT::ConstPtr PointCloudConstPtr;
  VoxelGrid() : downsample_all_data_(true), save_leaf_layout_(false), filter_name_("VoxelGrid") {
    leaf_size_.setZero();
    inverse_leaf_size_.setZero();
    min_b_.setZero();
    max_b_.setZero();
    div_b_.setZero();
    divb_mul_.setZero();
  }
  virtual ~VoxelGrid() {}
  void setInputCloud(const PointCloudConstPtr& c) { input_ = c; }
  void setLeafSize(float lx, float ly, float lz) {
    leaf_size_[0] = lx; leaf_size_[1] = ly; leaf_size_[2] = lz; leaf_size_[3] = 1.f;
    for (int k = 0; k < 4; k++) inverse_leaf_size_[k] = 1.f / leaf_size_[k];
  }
  const std::string& getClassName() const { return filter_name_; }

 protected:
  PointCloudConstPtr input_;
  Eigen::Vector4f leaf_size_;
  Eigen::Array4f inverse_leaf_size_;
  bool downsample_all_data_, save_leaf_layout_;
  Eigen::Vector4i min_b_, max_b_, div_b_, divb_mul_;
  std::string filter_name_;
};
}  // namespace pcl
