// TEST-ONLY stand-ins for the slice of PCL / Eigen / boost that pclomp/ndt_omp.h touches.
// This image has no PCL, Eigen or Boost; these few declarations let the adapter header be
// compile- and run-checked here.  Written from the public API documentation, not from PCL source;
// they are NOT part of the product and are never installed.
#pragma once
#include <cmath>
#include <cstdio>
#include <limits>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define PCL_WARN(...) std::fprintf(stderr, __VA_ARGS__)

namespace boost {
template <class T>
using shared_ptr = std::shared_ptr<T>;
}

namespace Eigen {
template <class S, int R, int C>
struct Matrix {
  S v[R * C];
  S* data() { return v; }
  const S* data() const { return v; }
  S& operator()(int r, int c) { return v[c * R + r]; }  // column-major, like Eigen's default
  const S& operator()(int r, int c) const { return v[c * R + r]; }
  S& operator()(int i) { return v[i]; }
  const S& operator()(int i) const { return v[i]; }
  static Matrix Identity() {
    Matrix m;
    std::memset(m.v, 0, sizeof(m.v));
    for (int i = 0; i < (R < C ? R : C); i++) m.v[i * R + i] = S(1);
    return m;
  }
  bool operator!=(const Matrix& o) const { return std::memcmp(v, o.v, sizeof(v)) != 0; }
  S& operator[](int i) { return v[i]; }
  const S& operator[](int i) const { return v[i]; }
  static Matrix Zero() {
    Matrix m;
    std::memset(m.v, 0, sizeof(m.v));
    return m;
  }
  void setZero() { std::memset(v, 0, sizeof(v)); }
};
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<float, 4, 1> Array4f;
typedef Matrix<int, 4, 1> Vector4i;
struct VectorXf {  // dynamic-size float vector: resize / size / operator[] only
  std::vector<float> v;
  void resize(int n) { v.assign(static_cast<size_t>(n), 0.f); }
  int size() const { return static_cast<int>(v.size()); }
  float& operator[](int i) { return v[static_cast<size_t>(i)]; }
  const float& operator[](int i) const { return v[static_cast<size_t>(i)]; }
};
template <class T>
using aligned_allocator = std::allocator<T>;
struct Affine3f {
  Matrix4f m;
  Matrix4f& matrix() { return m; }
};
}  // namespace Eigen

namespace pcl {
struct PointXYZ {
  float x, y, z, pad;  // 16 bytes, data[3] padding
};
struct PointXYZI {
  float x, y, z, pad;
  float intensity, pad2[3];  // 32 bytes
};

template <class PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT> > Ptr;
  typedef std::shared_ptr<const PointCloud<PointT> > ConstPtr;
  std::vector<PointT> points;
  unsigned width = 0, height = 1;
  bool is_dense = true;
  size_t size() const { return points.size(); }
};

template <class PointSource, class PointTarget>
class Registration {
 public:
  typedef PointCloud<PointSource> PointCloudSource;
  typedef PointCloud<PointTarget> PointCloudTarget;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  typedef typename PointCloudTarget::ConstPtr PointCloudTargetConstPtr;
  typedef std::shared_ptr<Registration<PointSource, PointTarget> > Ptr;

  Registration()
      : nr_iterations_(0), max_iterations_(10), transformation_epsilon_(0.0), converged_(false) {
    final_transformation_ = transformation_ = previous_transformation_ = Eigen::Matrix4f::Identity();
  }
  virtual ~Registration() {}
  virtual void setInputSource(const PointCloudSourceConstPtr& c) { input_ = c; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& c) { target_ = c; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  bool hasConverged() const { return converged_; }
  Eigen::Matrix4f getFinalTransformation() const { return final_transformation_; }
  void align(PointCloudSource& output) { align(output, Eigen::Matrix4f::Identity()); }
  void align(PointCloudSource& output, const Eigen::Matrix4f& guess) {
    if (!input_ || !target_) return;
    output.points = input_->points;
    output.width = static_cast<unsigned>(output.points.size());
    output.is_dense = input_->is_dense;
    converged_ = false;
    final_transformation_ = transformation_ = previous_transformation_ = Eigen::Matrix4f::Identity();
    for (auto& p : output.points) p.pad = 1.0f;
    computeTransformation(output, guess);
  }

 protected:
  virtual void computeTransformation(PointCloudSource& output, const Eigen::Matrix4f& guess) = 0;
  std::string reg_name_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  int nr_iterations_, max_iterations_;
  Eigen::Matrix4f final_transformation_, transformation_, previous_transformation_;
  double transformation_epsilon_;
  bool converged_;
};
}  // namespace pcl
