/* gicp_mi355.h -- C-ABI of the MI355X GICP core (libndt_mi355.so), SURVEY 8(f) row N4.
 *
 * Drop-in boundary for pclomp::GeneralizedIterativeClosestPoint<PointT,PointT>, the second registration
 * class of the reference's libndt_omp (ndt_omp/include/pclomp/gicp_omp.h, gicp_omp_impl.hpp; used by
 * ndt_omp/apps/align.cpp:80-86).  Plain pointers and sizes only; every entry point names the reference
 * member it replaces.  include/pclomp/gicp_omp.h maps the PCL class onto these 1:1.
 *
 * Status codes and the error string are those of ndt_mi355.h (ndt_last_error()).  A handle is
 * thread-compatible: one thread at a time.
 */
#ifndef GICP_MI355_H_
#define GICP_MI355_H_

#include "ndt_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gicp_context* gicp_handle;

/* ctor, gicp_omp.h:106-122: k_correspondences 20, gicp_epsilon 0.001, rotation_epsilon 2e-3,
 * max_inner_iterations 20, max_iterations 200, transformation_epsilon 5e-4, corr_dist_threshold 5 m */
ndt_status gicp_create(int device, gicp_handle* out);
void gicp_destroy(gicp_handle h);

/* setCorrespondenceRandomness (gicp_omp.h:229): 1 <= k <= 64 on this implementation */
ndt_status gicp_set_correspondence_randomness(gicp_handle h, int k);
/* setRotationEpsilon (:213) */
ndt_status gicp_set_rotation_epsilon(gicp_handle h, double eps);
/* setMaximumOptimizerIterations (:241) */
ndt_status gicp_set_maximum_optimizer_iterations(gicp_handle h, int n);
/* pcl::Registration::setTransformationEpsilon / setMaximumIterations / setMaxCorrespondenceDistance */
ndt_status gicp_set_transformation_epsilon(gicp_handle h, double eps);
ndt_status gicp_set_maximum_iterations(gicp_handle h, int n);
ndt_status gicp_set_max_correspondence_distance(gicp_handle h, double d);

/* setInputTarget (gicp_omp.h:156-160) / setInputSource (:128-143): host buffers of n points, xyz as three
 * f32 at the start of each stride_bytes record.  Drops the cloud's covariances, as the reference does.
 * Points must be finite (pcl::KdTreeFLANN requires it of its queries); NDT_ERR_INVALID otherwise. */
ndt_status gicp_set_input_target(gicp_handle h, const void* pts, size_t n, size_t stride_bytes);
ndt_status gicp_set_input_source(gicp_handle h, const void* pts, size_t n, size_t stride_bytes);
/* setSourceCovariances / setTargetCovariances (gicp_omp.h:165-168,186-189): one 3x3 f64 matrix per point of the cloud set
 * before ([n][9] row-major; symmetric, the upper triangle is used) instead of the k-NN covariances computeCovariances
 * (gicp_omp_impl.hpp:48-116) would produce; setting the cloud again resets them, n == 0 / NULL clears them. */
ndt_status gicp_set_source_covariances(gicp_handle h, const double* cov, size_t n);
ndt_status gicp_set_target_covariances(gicp_handle h, const double* cov, size_t n);

/* pcl::Registration::align(output, guess) -> computeTransformation (gicp_omp_impl.hpp:372-517).
 * guess / final_T: column-major 4x4 f32 (Eigen::Matrix4f::data()), guess may be NULL (identity).
 * out_cloud: NULL or n_source records of stride 16 bytes (x, y, z, 1). */
ndt_status gicp_align(gicp_handle h, const float* guess, float* final_T, int* converged, int* n_iterations, void* out_cloud);

/* hasConverged / getFinalTransformation state of the last align */
ndt_status gicp_get_result(gicp_handle h, float* final_T, int* converged, int* n_iterations);
/* pcl::Registration::getFitnessScore(max_range) after align */
ndt_status gicp_get_fitness_score(gicp_handle h, double max_range, double* fitness);
/* functor calls of the last align (operator(), df, fdf) and its last correspondence count */
ndt_status gicp_get_stats(gicp_handle h, int* n_f, int* n_df, int* n_fdf, int* correspondences);

/* --- inspection entry points (parity tests) ---------------------------------------------- */
/* computeCovariances (gicp_omp_impl.hpp:48-116) of the target (which = 0) or the source (1):
 * cov [n][9] row-major f64; nn_idx / nn_d2 optional [n][k] (ascending distance, then index). */
ndt_status gicp_covariances(gicp_handle h, int which, double* cov, int* nn_idx, float* nn_d2);
/* One correspondence step (:405-456) for `transformation` (column-major, NULL = identity) on the source
 * moved by `guess`: corr[i] = target index or -1, maha [n_source][9] row-major f32. */
ndt_status gicp_step_correspond(gicp_handle h, const float* guess, const float* transformation, int* corr, float* maha,
                                int* n_correspondences);
/* OptimizationFunctorWithIndices (:241-368) at x over the correspondences of the last step:
 * mode 0 operator() -> *f; 1 df -> g[6]; 2 fdf -> *f, g[6]. */
ndt_status gicp_step_functor(gicp_handle h, int mode, const double* x, double* f, double* g);
/* applyState on the identity (:519-532): column-major 4x4 */
void gicp_host_apply_state(const double* x, float* T);

#ifdef __cplusplus
}
#endif
#endif /* GICP_MI355_H_ */
