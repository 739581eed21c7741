// gicp_oracle.hpp -- TEST INFRASTRUCTURE ONLY (parity oracle for the GICP row).
//
// CPU restatement of pclomp::GeneralizedIterativeClosestPoint (reference
// ndt_omp/include/pclomp/gicp_omp.h, gicp_omp_impl.hpp) together with the third-party pieces it runs
// on that are NOT in the reference tree and are restated from their published algorithms:
//   [PCL 1.10]  pcl::Registration::align pre-amble, pcl::BFGS (registration/bfgs.h, itself a port of
//               GSL's vector_bfgs2 minimiser with Fletcher's line search), KdTreeFLANN exact k-NN
//               (L2_Simple f32 distances), transformPointCloud;
//   [Eigen 3.3] JacobiSVD of a symmetric 3x3 (through the symmetric eigen-decomposition),
//               AngleAxisf products (quaternions), Matrix3d::inverse.
// Only tests/ may build, link or run this.  PARITY STATUS: "parity unpinned" -- the reference has
// neither tests nor published numbers for GICP (the README table lists NDT only) and cannot be
// built here; the oracle is pinned by analytic checks only (gradient vs finite differences,
// closed-form covariances of planar patches, recovery of a known transform).
#pragma once
#include <array>
#include <cstddef>
#include <vector>

#include "ndt_oracle.hpp"

namespace oracle {

// ctor defaults, gicp_omp.h:106-122
struct GicpParams {
  int k_correspondences = 20;
  double gicp_epsilon = 0.001;
  double rotation_epsilon = 2e-3;
  double transformation_epsilon = 5e-4;
  double corr_dist_threshold = 5.0;
  int max_iterations = 200;
  int max_inner_iterations = 20;
};

struct GicpResult {
  float final_T[4][4];
  bool converged;
  int nr_iterations;
  int n_f, n_df, n_fdf;  // functor calls over the whole align
  int last_correspondences;
};

// exact k nearest neighbours of every `query` point among `cloud` ([FLANN] L2_Simple, f32), ordered by
// (distance, index); out_idx / out_d2 are [n_query][k]
void knn_exact(const std::vector<Pt>& cloud, const std::vector<Pt>& query, int k, std::vector<int>& out_idx,
               std::vector<float>& out_d2);

class GICP {
 public:
  GicpParams prm;
  std::vector<Pt> target, source;
  std::vector<M3> target_cov, source_cov;  // computed lazily by align (gicp_omp_impl.hpp:385-397)

  void set_target(const std::vector<Pt>& t) { target = t; target_cov.clear(); }  // gicp_omp.h:156-160
  void set_source(const std::vector<Pt>& s) { source = s; source_cov.clear(); }  // :128-143

  // computeCovariances, gicp_omp_impl.hpp:48-116
  static bool covariances(const std::vector<Pt>& cloud, int k, double gicp_epsilon, std::vector<M3>& out);

  // pcl::Registration::align pre-amble + computeTransformation (:372-517)
  GicpResult align(const float guess[4][4], std::vector<Pt>* output);

  // --- pieces of one outer iteration, public for the kernel-level parity tests ---
  // correspondence + Mahalanobis step (:419-456); `output` is the guess-transformed source
  int correspond(const std::vector<Pt>& output, const float transformation[4][4], const float guess[4][4]);
  std::vector<int> corr_src, corr_tgt;            // sorted by source index (:461-474)
  std::vector<std::array<float, 9>> mahalanobis;  // row-major 3x3 block of mahalanobis_[i], per source point
  const std::vector<Pt>* opt_src = nullptr;       // tmp_src_ of the functor
  // OptimizationFunctorWithIndices (:241-368): operator(), df, fdf
  double functor_f(const double x[6]) const;
  void functor_df(const double x[6], double g[6]) const;
  void functor_fdf(const double x[6], double& f, double g[6]) const;
  // the un-normalised sums behind them at a given T: out = f, g[3], R[9] row-major, count (mode 0: f32 form, f and count only)
  void functor_raw(int mode, const float T[4][4], double out[14]) const;

  // applyState on the identity (:519-532), computeRDerivative (:119-178)
  static void apply_state(const double x[6], float T[4][4]);
  static void r_derivative(const double x[6], const double R[3][3], double g[6]);

  mutable int n_f = 0, n_df = 0, n_fdf = 0;

 private:
  // estimateRigidTransformationBFGS (:181-238); false = NotEnoughPointsException
  bool estimate_bfgs(float transformation[4][4]);
};

}  // namespace oracle
