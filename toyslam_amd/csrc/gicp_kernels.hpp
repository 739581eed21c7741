// gicp_kernels.hpp -- launchers of the GICP kernels (gicp_kernels.hip), gfx950 only.
// GICP = pclomp::GeneralizedIterativeClosestPoint (reference ndt_omp/include/pclomp/gicp_omp.h,
// gicp_omp_impl.hpp), SURVEY 8(f) row N4.
#pragma once
#include <hip/hip_runtime.h>

#include "ndt_kernels.hpp"

namespace gicp {

using ndt::PointIndex;  // a cloud + the voxel index K1 built over it (ndt_kernels.hpp)

using ndt::launch_gather_points;

constexpr int kMaxK = 64;           // k_correspondences_ supported by the LDS candidate lists
constexpr int kFunctorValues = 14;  // f, g_t[3], R[9] (row-major), correspondence count
constexpr int kFunctorMaxBlocks = 1024;

// computeCovariances (gicp_omp_impl.hpp:48-116): cov6[i] = xx,xy,xz,yy,yz,zz of the regularised
// covariance of point i.  nn_idx / nn_d2 (optional, [n][k]): the neighbours, ascending (distance, index).
hipError_t launch_knn_covariances(const PointIndex& ix, int k, double gicp_epsilon, double* cov6, int* nn_idx, float* nn_d2,
                                  hipStream_t stream);

// One outer iteration's correspondence step (:405-456): query = T * output[i]; corr[i] = nearest target
// index if its squared distance < dist_threshold else -1; maha9[i] = (R C1 R^T + C2)^-1 as f32 (row-major).
struct Rot3d {
  double m[9];
};
hipError_t launch_correspond(const float4* output, int n, const float* T12, const Rot3d& R, const PointIndex& tgt,
                             const double* cov_src6, const double* cov_tgt6, double dist_threshold, int* corr, float* maha9,
                             hipStream_t stream);

// OptimizationFunctorWithIndices (:241-368) over the current correspondences.  mode 0 = operator()
// (f32 quadratic form), 1 / 2 = df / fdf (f64), 3 = operator() in slot 0 together with df's gradient sums.  One launch: per-block rows -> ticket -> the last block
// sums them in a fixed order and publishes kFunctorValues raw sums as a tagged row (ndt_device.hpp
// publish_row_tagged) into pinned host memory.  counter: one zero-initialised u32, reset by the kernel.
int functor_blocks(int n);
hipError_t launch_functor(int mode, const float4* output, int n, const float4* tgt, const int* corr, const float* maha9,
                          const float* T12, int n_blocks, double* partials, unsigned* counter, double* out_row,
                          unsigned long long seq, hipStream_t stream);

// Persistent objective server (one launch per BFGS run), see gicp_kernels.hip.  mailbox: 256 + 128 bytes of fine-grained
// device memory laid out like the NDT server's (the host posts with ndt::server_post: kind = functor mode 0 / 1 / 3, or
// ndt::kServerCmdExit); counter: kGicpServerParts * 32 zeroed u32; out_rows: kGicpServerParts tagged rows of pinned host memory.
constexpr int kGicpServerParts = 8;
int server_blocks(int n);
hipError_t launch_server(const float4* output, int n, const float4* tgt, const int* corr, const float* maha9, void* mailbox,
                         int n_blocks, double* partials, unsigned* counter, double* out_rows, unsigned long long first_seq,
                         unsigned long long idle_ticks, hipStream_t stream);

}  // namespace gicp
