// pclomp/ndt_omp.h -- header-only adapter: the class ToySLAM's nodes include, re-declared on top
// of the MI355X C-ABI (include/ndt_mi355.h, libndt_mi355.so).
//
// Replaces, with the same include path, class name, template parameters, public data member and
// method surface:  /root/reference/ndt_omp/include/pclomp/ndt_omp.h:50-504
// (+ ndt_omp_impl.hpp, voxel_grid_covariance_omp*.h, src/pclomp/ndt_omp.cpp, which this header
// makes unnecessary: nothing is compiled into a libndt_omp any more -- link libndt_mi355.so).
//
// Callers that build unchanged against it: ndt_omp/apps/align.cpp:88-105 (passes the object as
// pcl::Registration<...>::Ptr, so the class derives from pcl::Registration and overrides
// computeTransformation), lidar_subscriber/src/ndt_omp_mapping_node.cpp:55-62,151-169 (returns the
// object BY VALUE -> copy constructor shares the device grid), ndt_rosbag_mapping_node.cpp:100-141,
// ndt_omp_node.cpp:101-125.
//
// Needs PCL (pcl::Registration, pcl::PointCloud) and Eigen at compile time like the original; the
// numerics live behind the C-ABI, so no OpenMP, no FLANN.  In this repository it is compile- and
// run-checked against the minimal PCL/Eigen stand-ins of tests/pcl_stub/ (test-only).
#ifndef PCL_REGISTRATION_NDT_OMP_MI355_H_
#define PCL_REGISTRATION_NDT_OMP_MI355_H_

#include <pcl/registration/registration.h>

#include <cstddef>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "ndt_mi355.h"

namespace pclomp {

enum NeighborSearchMethod { KDTREE, DIRECT26, DIRECT7, DIRECT1 };  // ndt_omp.h:52-57

template <typename PointSource, typename PointTarget>
class NormalDistributionsTransform : public pcl::Registration<PointSource, PointTarget> {
 protected:
  typedef typename pcl::Registration<PointSource, PointTarget>::PointCloudSource PointCloudSource;
  typedef typename PointCloudSource::Ptr PointCloudSourcePtr;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  typedef typename pcl::Registration<PointSource, PointTarget>::PointCloudTarget PointCloudTarget;
  typedef typename PointCloudTarget::Ptr PointCloudTargetPtr;
  typedef typename PointCloudTarget::ConstPtr PointCloudTargetConstPtr;

 public:
#if defined(PCL_VERSION_CALC)
#if PCL_VERSION >= PCL_VERSION_CALC(1, 10, 0)
#define NDT_MI355_PCL_SHARED_PTR 1
#endif
#endif
#ifdef NDT_MI355_PCL_SHARED_PTR
  typedef pcl::shared_ptr<NormalDistributionsTransform<PointSource, PointTarget> > Ptr;
  typedef pcl::shared_ptr<const NormalDistributionsTransform<PointSource, PointTarget> > ConstPtr;
#else
  typedef boost::shared_ptr<NormalDistributionsTransform<PointSource, PointTarget> > Ptr;
  typedef boost::shared_ptr<const NormalDistributionsTransform<PointSource, PointTarget> > ConstPtr;
#endif

  /** ctor defaults of ndt_omp_impl.hpp:47-76 live in ndt_create(). */
  NormalDistributionsTransform() : handle_(nullptr), source_dirty_(false), trans_probability_(0), search_method(DIRECT7) {
    reg_name_ = "NormalDistributionsTransform";
    check(ndt_create(default_device(), &handle_), "ndt_create");
    transformation_epsilon_ = 0.1;  // :71
    max_iterations_ = 35;           // :72
  }

  /** Value semantics (ndt_omp_mapping_node.cpp:151-169): the copy shares the immutable device
   *  grid and source cloud, and carries the parameters and the last result. */
  NormalDistributionsTransform(const NormalDistributionsTransform& o)
      : pcl::Registration<PointSource, PointTarget>(o), handle_(nullptr), uploaded_source_(o.uploaded_source_),
        source_dirty_(o.source_dirty_), trans_probability_(o.trans_probability_), search_method(o.search_method) {
    check(ndt_clone(o.handle_, &handle_), "ndt_clone");
  }
  NormalDistributionsTransform& operator=(const NormalDistributionsTransform& o) {
    if (this != &o) {
      pcl::Registration<PointSource, PointTarget>::operator=(o);
      ndt_handle h = nullptr;
      check(ndt_clone(o.handle_, &h), "ndt_clone");
      ndt_destroy(handle_);
      handle_ = h;
      uploaded_source_ = o.uploaded_source_;
      source_dirty_ = o.source_dirty_;
      trans_probability_ = o.trans_probability_;
      search_method = o.search_method;
    }
    return *this;
  }
  virtual ~NormalDistributionsTransform() { ndt_destroy(handle_); }

  void setNumThreads(int n) { ndt_set_num_threads(handle_, n); }  // :115-117 (no effect on the GPU)

  /** :122-127 -- uploads the cloud and builds the voxel grid on the GPU (init()). */
  inline void setInputTarget(const PointCloudTargetConstPtr& cloud) {
    pcl::Registration<PointSource, PointTarget>::setInputTarget(cloud);
    const ndt_status s = ndt_set_input_target(handle_, cloud->points.data(), cloud->points.size(), sizeof(PointTarget),
                                              cloud->is_dense ? 1 : 0);
    if (s == NDT_ERR_GRID_OVERFLOW)
      PCL_WARN("[pclomp::NormalDistributionsTransform] %s\n", ndt_last_error());  // _impl.hpp:79-84 warns, keeps going
    else
      check(s, "ndt_set_input_target");
  }

  /** pcl::Registration::setInputSource is virtual; the reference does not override it and reads
   *  *input_ at every align (ndt_omp_impl.hpp:833).  Here the points live on the GPU, so every call
   *  marks the source for upload at the next align -- also when the pointer is the one already
   *  uploaded (a caller may have refilled the cloud in place).  Between two align calls with no
   *  setInputSource in between the upload is reused, as PCL reuses its own KD-tree. */
  inline void setInputSource(const PointCloudSourceConstPtr& cloud) {
    pcl::Registration<PointSource, PointTarget>::setInputSource(cloud);
    source_dirty_ = true;
  }

  inline void setResolution(float resolution) {  // :132-142 (rebuild rule is inside the library)
    if (input_) sync_source();                   // the reference tests input_, the SOURCE
    check(ndt_set_resolution(handle_, resolution), "ndt_set_resolution");
  }
  inline float getResolution() const { return ndt_get_resolution(handle_); }
  inline double getStepSize() const { return ndt_get_step_size(handle_); }
  inline void setStepSize(double step_size) { ndt_set_step_size(handle_, step_size); }
  inline double getOutlierRatio() const { return ndt_get_outlier_ratio(handle_); }
  inline void setOutlierRatio(double outlier_ratio) { ndt_set_outlier_ratio(handle_, outlier_ratio); }
  inline void setNeighborhoodSearchMethod(NeighborSearchMethod method) { search_method = method; }
  inline double getTransformationProbability() const { return trans_probability_; }
  inline int getFinalNumIteration() const { return nr_iterations_; }

  /** :215-234 */
  static void convertTransform(const Eigen::Matrix<double, 6, 1>& x, Eigen::Affine3f& trans) {
    float T[16];
    const double p[6] = {x(0), x(1), x(2), x(3), x(4), x(5)};
    ndt_host_pose_to_matrix(p, T);
    Eigen::Matrix4f m;
    for (int i = 0; i < 16; i++) m.data()[i] = T[i];
    trans.matrix() = m;
  }
  static void convertTransform(const Eigen::Matrix<double, 6, 1>& x, Eigen::Matrix4f& trans) {
    float T[16];
    const double p[6] = {x(0), x(1), x(2), x(3), x(4), x(5)};
    ndt_host_pose_to_matrix(p, T);
    for (int i = 0; i < 16; i++) trans.data()[i] = T[i];
  }

  /** :238, ndt_omp_impl.hpp:935-983 */
  double calculateScore(const PointCloudSource& cloud) const {
    double score = 0;
    ndt_set_neighborhood_search_method(handle_, static_cast<int>(search_method));
    check(ndt_calculate_score(handle_, cloud.points.data(), cloud.points.size(), sizeof(PointSource), &score),
          "ndt_calculate_score");
    return score;
  }

  /** pcl::Registration::getFitnessScore(max_range) of the last align, on the GPU (exact nearest
   *  neighbour over the target's voxel grid).  Hides the base's non-virtual KD-tree version for
   *  callers that hold the derived type (ndt_rosbag_mapping_node.cpp:133); through a
   *  pcl::Registration pointer (apps/align.cpp:24-33) PCL's own implementation still answers. */
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double fitness = 0;
    check(ndt_get_fitness_score(handle_, max_range, &fitness), "ndt_get_fitness_score");
    return fitness;
  }

  /** Access for callers that want the batch / device entry points of the C-ABI. */
  ndt_handle native_handle() const { return handle_; }

 protected:
  using pcl::Registration<PointSource, PointTarget>::reg_name_;
  using pcl::Registration<PointSource, PointTarget>::input_;
  using pcl::Registration<PointSource, PointTarget>::target_;
  using pcl::Registration<PointSource, PointTarget>::nr_iterations_;
  using pcl::Registration<PointSource, PointTarget>::max_iterations_;
  using pcl::Registration<PointSource, PointTarget>::previous_transformation_;
  using pcl::Registration<PointSource, PointTarget>::final_transformation_;
  using pcl::Registration<PointSource, PointTarget>::transformation_;
  using pcl::Registration<PointSource, PointTarget>::transformation_epsilon_;
  using pcl::Registration<PointSource, PointTarget>::converged_;

  virtual void computeTransformation(PointCloudSource& output) {  // :262-266
    computeTransformation(output, Eigen::Matrix4f::Identity());
  }

  /** :272-273, ndt_omp_impl.hpp:80-171.  pcl::Registration::align has already copied input_ to
   *  output and reset the transforms; everything else happens behind the C-ABI. */
  virtual void computeTransformation(PointCloudSource& output, const Eigen::Matrix4f& guess) {
    sync_source();
    ndt_set_transformation_epsilon(handle_, transformation_epsilon_);
    ndt_set_maximum_iterations(handle_, max_iterations_);
    ndt_set_neighborhood_search_method(handle_, static_cast<int>(search_method));
    int conv = 0, iters = 0;
    float final_T[16];
    if (output.points.size() != input_->points.size()) output.points.resize(input_->points.size());
    check(ndt_align(handle_, guess.data(), final_T, &conv, &iters, &trans_probability_, output.points.data(),
                    sizeof(PointSource)),
          "ndt_align");
    for (int i = 0; i < 16; i++) final_transformation_.data()[i] = final_T[i];
    converged_ = conv != 0;
    nr_iterations_ = iters;
  }

 private:
  static int default_device() {
    const char* v = std::getenv("NDT_MI355_DEVICE");
    return v ? std::atoi(v) : 0;
  }
  static void check(ndt_status s, const char* what) {
    // The reference has no error channel (PCL_WARN + hasConverged()).  A missing GPU is not a
    // condition it could ever meet, so it is reported loudly instead of silently not converging.
    if (s != NDT_OK) throw std::runtime_error(std::string(what) + ": " + ndt_last_error());
  }
  void sync_source() {
    if (input_ && (source_dirty_ || input_ != uploaded_source_)) {
      check(ndt_set_input_source(handle_, input_->points.data(), input_->points.size(), sizeof(PointSource)),
            "ndt_set_input_source");
      uploaded_source_ = input_;
      source_dirty_ = false;
    }
  }

  ndt_handle handle_;
  PointCloudSourceConstPtr uploaded_source_;
  bool source_dirty_;  // setInputSource since the last upload
  double trans_probability_;

 public:
  NeighborSearchMethod search_method;  // public data member, ndt_omp.h:499

  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
};

}  // namespace pclomp

#endif  // PCL_REGISTRATION_NDT_OMP_MI355_H_
