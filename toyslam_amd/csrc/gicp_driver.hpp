// gicp_driver.hpp -- host side of the GICP row: the outer correspondence loop of
// pclomp::GeneralizedIterativeClosestPoint::computeTransformation (reference
// ndt_omp/include/pclomp/gicp_omp_impl.hpp:372-517) and the 6-parameter BFGS it runs per iteration
// (estimateRigidTransformationBFGS :181-238 on [PCL 1.10] registration/bfgs.h).  Scalar f64 code,
// PCL-free and Eigen-free; every sum over points comes from the device through `Backend`.
#pragma once

namespace gicp {

// ctor defaults, gicp_omp.h:106-122
struct Params {
  int k_correspondences = 20;
  double gicp_epsilon = 0.001;
  double rotation_epsilon = 2e-3;
  double transformation_epsilon = 5e-4;
  double corr_dist_threshold = 5.0;
  int max_iterations = 200;
  int max_inner_iterations = 20;
};

// raw sums of OptimizationFunctorWithIndices over the current correspondences
struct FunctorSums {
  double f;     // sum of res' M res
  double g[3];  // sum of M res
  double R[9];  // sum of p_src (M res)', row-major
  double m;     // number of correspondences
};

class Backend {
 public:
  virtual ~Backend() {}
  // correspondence + Mahalanobis step for transformation_ (row-major 4x4) and the f64 rotation of
  // transformation_ * guess (:411-419); asynchronous
  virtual bool correspond(const float transformation[16], const double R[9]) = 0;
  // mode 0: operator() (only f and m are meaningful), 1: df, 2: fdf; T = applyState(x), row-major 4x4
  virtual bool sums(int mode, const float T[16], FunctorSums& out) = 0;
};

struct Result {
  float final_T[16];  // row-major
  bool converged = false;
  bool backend_failed = false;
  int nr_iterations = 0;
  int n_f = 0, n_df = 0, n_fdf = 0;
  int correspondences = 0;
};

// applyState on the identity (:519-532): [Eigen] AngleAxisf(z) * AngleAxisf(y) * AngleAxisf(x) through f32
// quaternions; row-major 4x4
void apply_state(const double x[6], float T[16]);
// computeRDerivative (:119-178): fills g[3..5]
void rotation_gradient(const double x[6], const double R[9], double g[6]);

// guess: row-major 4x4
Result run(const Params& prm, const float guess[16], Backend& dev);

}  // namespace gicp
