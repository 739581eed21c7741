// pclomp/gicp_omp.h -- header-only adapter: pclomp::GeneralizedIterativeClosestPoint re-declared on top
// of the MI355X C-ABI (include/gicp_mi355.h, libndt_mi355.so).
//
// Replaces, with the same include path, class name, template parameters and public method surface:
//   /root/reference/ndt_omp/include/pclomp/gicp_omp.h:52-378 (+ gicp_omp_impl.hpp and
//   src/pclomp/gicp_omp.cpp, which this header makes unnecessary).
// Its one caller builds unchanged against it: ndt_omp/apps/align.cpp:84-86 creates the object, hands it
// over as pcl::Registration<...>::Ptr and calls setInputTarget / setInputSource / align /
// getFitnessScore through the base class -- so the class derives from pcl::IterativeClosestPoint
// (hence pcl::Registration) and overrides the virtual setInput* and computeTransformation.
//
// setSourceCovariances / setTargetCovariances (:165-168,186-189) are carried over (gicp_set_*_covariances).
// Not carried over: the protected per-point helpers (computeCovariances,
// mahalanobis(), computeRDerivative, estimateRigidTransformationBFGS, the BFGS functor), which live
// behind the C-ABI.  Needs PCL at compile time like the original; compile- and run-checked here against
// the stand-ins of tests/pcl_stub/ (test-only).
#ifndef PCL_GICP_OMP_MI355_H_
#define PCL_GICP_OMP_MI355_H_

#include <pcl/registration/icp.h>

#include <cstddef>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "gicp_mi355.h"

namespace pclomp {

template <typename PointSource, typename PointTarget>
class GeneralizedIterativeClosestPoint : public pcl::IterativeClosestPoint<PointSource, PointTarget> {
 public:
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
#if defined(PCL_VERSION_CALC)
#if PCL_VERSION >= PCL_VERSION_CALC(1, 10, 0)
#define GICP_MI355_PCL_SHARED_PTR 1
#endif
#endif
  using MatricesVector = std::vector<Eigen::Matrix3d, Eigen::aligned_allocator<Eigen::Matrix3d> >;  // gicp_omp.h:96
#ifdef GICP_MI355_PCL_SHARED_PTR
  using MatricesVectorPtr = pcl::shared_ptr<MatricesVector>;
  using MatricesVectorConstPtr = pcl::shared_ptr<const MatricesVector>;
  using Ptr = pcl::shared_ptr<GeneralizedIterativeClosestPoint<PointSource, PointTarget> >;
  using ConstPtr = pcl::shared_ptr<const GeneralizedIterativeClosestPoint<PointSource, PointTarget> >;
#else
  using MatricesVectorPtr = boost::shared_ptr<MatricesVector>;
  using MatricesVectorConstPtr = boost::shared_ptr<const MatricesVector>;
  using Ptr = boost::shared_ptr<GeneralizedIterativeClosestPoint<PointSource, PointTarget> >;
  using ConstPtr = boost::shared_ptr<const GeneralizedIterativeClosestPoint<PointSource, PointTarget> >;
#endif

  /** gicp_omp.h:106-122 (the GICP-specific defaults live in gicp_create()). */
  GeneralizedIterativeClosestPoint() : k_correspondences_(20), rotation_epsilon_(2e-3), max_inner_iterations_(20), handle_(nullptr) {
    reg_name_ = "GeneralizedIterativeClosestPoint";
    check(gicp_create(default_device(), &handle_), "gicp_create");
    max_iterations_ = 200;
    transformation_epsilon_ = 5e-4;
    corr_dist_threshold_ = 5.;
  }
  GeneralizedIterativeClosestPoint(const GeneralizedIterativeClosestPoint&) = delete;  // the reference holds it by Ptr only
  GeneralizedIterativeClosestPoint& operator=(const GeneralizedIterativeClosestPoint&) = delete;
  virtual ~GeneralizedIterativeClosestPoint() { gicp_destroy(handle_); }

  /** :128-143 */
  inline void setInputSource(const PointCloudSourceConstPtr& cloud) override {
    if (cloud->points.empty()) {
      PCL_ERROR("[pcl::%s::setInputSource] Invalid or empty point cloud dataset given!\n", reg_name_.c_str());
      return;
    }
    pcl::IterativeClosestPoint<PointSource, PointTarget>::setInputSource(cloud);
    input_covariances_.reset();
    source_cov_dirty_ = false;  // setting the cloud resets the handle's matrices too
    check(gicp_set_input_source(handle_, cloud->points.data(), cloud->points.size(), sizeof(PointSource)), "gicp_set_input_source");
  }
  /** :165-168 -- used by the next align instead of the k-NN covariances, until setInputSource */
  inline void setSourceCovariances(const MatricesVectorPtr& covariances) {
    input_covariances_ = covariances;
    source_cov_dirty_ = true;
  }
  /** :156-160 */
  inline void setInputTarget(const PointCloudTargetConstPtr& target) override {
    pcl::IterativeClosestPoint<PointSource, PointTarget>::setInputTarget(target);
    target_covariances_.reset();
    target_cov_dirty_ = false;
    check(gicp_set_input_target(handle_, target->points.data(), target->points.size(), sizeof(PointTarget)), "gicp_set_input_target");
  }
  /** :186-189 */
  inline void setTargetCovariances(const MatricesVectorPtr& covariances) {
    target_covariances_ = covariances;
    target_cov_dirty_ = true;
  }

  inline void setRotationEpsilon(double epsilon) { rotation_epsilon_ = epsilon; }  // :213
  inline double getRotationEpsilon() { return rotation_epsilon_; }                 // :219
  void setCorrespondenceRandomness(int k) { k_correspondences_ = k; }              // :229
  int getCorrespondenceRandomness() { return k_correspondences_; }                 // :235
  void setMaximumOptimizerIterations(int max) { max_inner_iterations_ = max; }     // :241
  int getMaximumOptimizerIterations() { return max_inner_iterations_; }            // :247

  /** pcl::Registration::getFitnessScore(max_range) of the last align, on the GPU; hides the base's
   *  non-virtual KD-tree version for callers that hold the derived type. */
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double fitness = 0;
    check(gicp_get_fitness_score(handle_, max_range, &fitness), "gicp_get_fitness_score");
    return fitness;
  }

  gicp_handle native_handle() const { return handle_; }

 protected:
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::reg_name_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::input_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::target_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::nr_iterations_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::max_iterations_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::previous_transformation_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::final_transformation_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::transformation_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::transformation_epsilon_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::converged_;
  using pcl::IterativeClosestPoint<PointSource, PointTarget>::corr_dist_threshold_;

  /** :338, gicp_omp_impl.hpp:372-517.  pcl::Registration::align has already copied input_ to output and
   *  reset the transforms; everything else happens behind the C-ABI. */
  void computeTransformation(PointCloudSource& output, const Eigen::Matrix4f& guess) override {
    check(gicp_set_correspondence_randomness(handle_, k_correspondences_), "gicp_set_correspondence_randomness");
    gicp_set_rotation_epsilon(handle_, rotation_epsilon_);
    gicp_set_maximum_optimizer_iterations(handle_, max_inner_iterations_);
    gicp_set_transformation_epsilon(handle_, transformation_epsilon_);
    gicp_set_maximum_iterations(handle_, max_iterations_);
    gicp_set_max_correspondence_distance(handle_, corr_dist_threshold_);
    // caller-supplied covariances (gicp_omp_impl.hpp:386-397: the class computes its own k-NN covariances whenever the
    // pointer is missing or the vector empty -- so a null / empty setter call CLEARS what an earlier call supplied).
    // Pushed only when a setter ran since the last align: the matrices are flattened and uploaded once, not per align.
    if (target_cov_dirty_) push_covariances(target_covariances_, false);
    if (source_cov_dirty_) push_covariances(input_covariances_, true);
    target_cov_dirty_ = source_cov_dirty_ = false;
    int conv = 0, iters = 0;
    float final_T[16];
    std::vector<float> moved(input_->points.size() * 4);
    check(gicp_align(handle_, guess.data(), final_T, &conv, &iters, moved.data()), "gicp_align");
    for (int i = 0; i < 16; i++) final_transformation_.data()[i] = final_T[i];
    if (output.points.size() != input_->points.size()) output.points.resize(input_->points.size());
    for (size_t i = 0; i < output.points.size(); i++) {  // xyz only: the other fields stay as align() copied them
      output.points[i].x = moved[4 * i];
      output.points[i].y = moved[4 * i + 1];
      output.points[i].z = moved[4 * i + 2];
    }
    converged_ = conv != 0;
    nr_iterations_ = iters;
  }

  int k_correspondences_;
  double rotation_epsilon_;
  int max_inner_iterations_;
  MatricesVectorPtr input_covariances_, target_covariances_;
  bool source_cov_dirty_ = false, target_cov_dirty_ = false;  // a setter ran since the handle last saw the matrices

 private:
  static int default_device() {
    const char* v = std::getenv("NDT_MI355_DEVICE");
    return v ? std::atoi(v) : 0;
  }
  void push_covariances(const MatricesVectorPtr& c, bool source) {
    if (!c || c->empty()) {  // back to the class's own k-NN covariances
      check(source ? gicp_set_source_covariances(handle_, nullptr, 0) : gicp_set_target_covariances(handle_, nullptr, 0),
            source ? "gicp_set_source_covariances" : "gicp_set_target_covariances");
      return;
    }
    std::vector<double> flat(c->size() * 9);
    for (size_t i = 0; i < c->size(); i++)
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) flat[i * 9 + r * 3 + k] = (*c)[i](r, k);
    check(source ? gicp_set_source_covariances(handle_, flat.data(), c->size()) : gicp_set_target_covariances(handle_, flat.data(), c->size()),
          source ? "gicp_set_source_covariances" : "gicp_set_target_covariances");
  }
  static void check(ndt_status s, const char* what) {
    // the reference's only error channels are PCL_ERROR and hasConverged(); a missing GPU or an invalid
    // cloud is reported loudly instead of silently not converging
    if (s != NDT_OK) throw std::runtime_error(std::string(what) + ": " + ndt_last_error());
  }

  gicp_handle handle_;
};

}  // namespace pclomp

#endif  // PCL_GICP_OMP_MI355_H_
