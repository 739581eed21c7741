// ndt_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  See ndt_oracle.hpp for the rules.
// Build: g++ -O3 -fopenmp -msse4.2 -ffp-contract=off (reference flags:
// RELEASE + SSE4.2, ndt_omp/CMakeLists.txt:10-15; no FMA on that target).
#include "ndt_oracle.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace oracle {

// ===========================================================================
// [Eigen] restatements
// ===========================================================================

// Cyclic Jacobi on the symmetric matrix defined by the LOWER triangle of `a`
// (SelfAdjointEigenSolver reads only the lower part).  Any backward-stable
// symmetric solver gives the same eigenvalues to O(eps*|A|) and the same
// V*L*V^-1, which is all voxel_grid_covariance_omp_impl.hpp:333-356 consumes.
void eig3_sym(const M3& a, V3& evals, M3& evecs) {
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = (i >= j) ? a.m[i][j] : a.m[j][i];
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
    double diag = std::fabs(A[0][0]) + std::fabs(A[1][1]) + std::fabs(A[2][2]);
    if (off <= 1e-300 || off <= diag * 1e-18) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {  // A <- A * G
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- G^T * A
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int order[3] = {0, 1, 2};
  std::sort(order, order + 3, [&](int x, int y) { return A[x][x] < A[y][y]; });
  for (int j = 0; j < 3; j++) {
    evals.v[j] = A[order[j]][order[j]];
    for (int i = 0; i < 3; i++) evecs.m[i][j] = V[i][order[j]];
  }
}

// Eigen/src/LU/InverseImpl.h, compute_inverse<.,.,3>: cofactors of column 0,
// det = cofactors_col0 . col(0), result(j,i) = cofactor<i,j> * (1/det).
M3 inv3(const M3& a) {
  auto cof = [&](int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
  };
  double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
  double det = (c00 * a.m[0][0] + c10 * a.m[1][0]) + c20 * a.m[2][0];
  double invdet = 1.0 / det;
  M3 r;
  r.m[0][0] = c00 * invdet;
  r.m[0][1] = c10 * invdet;
  r.m[0][2] = c20 * invdet;
  r.m[1][0] = cof(0, 1) * invdet;
  r.m[1][1] = cof(1, 1) * invdet;
  r.m[1][2] = cof(2, 1) * invdet;
  r.m[2][0] = cof(0, 2) * invdet;
  r.m[2][1] = cof(1, 2) * invdet;
  r.m[2][2] = cof(2, 2) * invdet;
  return r;
}

// One-sided (Hestenes) Jacobi SVD, then x = V * S^+ * U^T * b with Eigen's
// rank rule (SVDBase::rank(): sigma_i >= max(sigma_0 * diagSize*eps, DBL_MIN)).
// ndt_omp_impl.hpp:127-129.
void svd6_solve(const double H[36], const double b[6], double x[6]) {
  double W[6][6], V[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      W[i][j] = H[i * 6 + j];
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 5; p++)
      for (int q = p + 1; q < 6; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 6; k++) {
          alpha += W[k][p] * W[k][p];
          beta += W[k][q] * W[k][q];
          gamma += W[k][p] * W[k][q];
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 6; k++) {
          double wp = W[k][p], wq = W[k][q];
          W[k][p] = c * wp - s * wq;
          W[k][q] = s * wp + c * wq;
          double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq;
          V[k][q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sig[6], smax = 0;
  for (int j = 0; j < 6; j++) {
    double n2 = 0;
    for (int k = 0; k < 6; k++) n2 += W[k][j] * W[k][j];
    sig[j] = std::sqrt(n2);
    smax = std::max(smax, sig[j]);
  }
  double thr = std::max(smax * 6.0 * std::numeric_limits<double>::epsilon(),
                        std::numeric_limits<double>::min());
  for (int i = 0; i < 6; i++) x[i] = 0;
  for (int j = 0; j < 6; j++) {
    if (!(sig[j] >= thr)) continue;  // also drops NaN columns
    double ub = 0;
    for (int k = 0; k < 6; k++) ub += W[k][j] * b[k];  // sigma_j * (u_j . b)
    double coef = ub / (sig[j] * sig[j]);
    for (int i = 0; i < 6; i++) x[i] += V[i][j] * coef;
  }
  // NaN in H or b must surface as NaN (reference exits on delta_p_norm != itself)
  for (int i = 0; i < 36; i++)
    if (H[i] != H[i]) x[0] = H[i];
  for (int i = 0; i < 6; i++)
    if (b[i] != b[i]) x[0] = b[i];
}

// Eigen/src/Geometry/Transform.h: Affine-mode rotation() = polar factor of the
// linear part (computeRotationScaling via f32 JacobiSVD).  Restated as the
// Newton polar iteration in f64, rounded to f32 (differs from Eigen's f32 SVD by
// O(1e-7); exact for exactly-orthonormal input such as Identity).
// Eigen/src/Geometry/EulerAngles.h (3.3.7), a0,a1,a2 = 0,1,2  => odd=0,i=0,j=1,k=2.
// ndt_omp_impl.hpp:103-111.
void euler_xyz_from_matrix(const float T[4][4], float ang[3]) {
  M3 X;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) X.m[i][j] = T[i][j];
  for (int it = 0; it < 30; it++) {
    M3 Xi = inv3(X);
    double diff = 0;
    M3 Y;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        Y.m[i][j] = 0.5 * (X.m[i][j] + Xi.m[j][i]);
        diff = std::max(diff, std::fabs(Y.m[i][j] - X.m[i][j]));
      }
    X = Y;
    if (diff < 1e-15) break;
  }
  float m[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m[i][j] = static_cast<float>(X.m[i][j]);
  const float PI_F = static_cast<float>(3.141592653589793238462643383279502884L);
  float res[3];
  res[0] = std::atan2(m[1][2], m[2][2]);
  float c2 = std::sqrt(m[0][0] * m[0][0] + m[0][1] * m[0][1]);
  if (res[0] > 0.0f) {  // !odd && res[0] > 0
    res[0] -= PI_F;
    res[1] = std::atan2(-m[0][2], -c2);
  } else {
    res[1] = std::atan2(-m[0][2], c2);
  }
  float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * m[2][0] - c1 * m[1][0], c1 * m[1][1] - s1 * m[2][1]);
  ang[0] = -res[0];
  ang[1] = -res[1];
  ang[2] = -res[2];
}

// Eigen AngleAxis<float>::toRotationMatrix for a unit axis, then
// Transform*=rotation as f32 3x3 products (k-ordered sums, no FMA).
// ndt_omp_impl.hpp:146-149, 827-830; ndt_omp.h:215-222.
static void angle_axis_unit(int axis, float angle, float R[3][3]) {
  float ax[3] = {0, 0, 0};
  ax[axis] = 1.0f;
  float s = std::sin(angle), c = std::cos(angle);
  float sin_axis[3] = {s * ax[0], s * ax[1], s * ax[2]};
  float cos1_axis[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
  float tmp;
  tmp = cos1_axis[0] * ax[1];
  R[0][1] = tmp - sin_axis[2];
  R[1][0] = tmp + sin_axis[2];
  tmp = cos1_axis[0] * ax[2];
  R[0][2] = tmp + sin_axis[1];
  R[2][0] = tmp - sin_axis[1];
  tmp = cos1_axis[1] * ax[2];
  R[1][2] = tmp - sin_axis[0];
  R[2][1] = tmp + sin_axis[0];
  for (int i = 0; i < 3; i++) R[i][i] = cos1_axis[i] * ax[i] + c;
}

static void mul33f(const float A[3][3], const float B[3][3], float C[3][3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i][j] = (A[i][0] * B[0][j] + A[i][1] * B[1][j]) + A[i][2] * B[2][j];
}

void pose_to_matrix(const double p[6], float T[4][4]) {
  float Rx[3][3], Ry[3][3], Rz[3][3], A[3][3], B[3][3];
  angle_axis_unit(0, static_cast<float>(p[3]), Rx);
  angle_axis_unit(1, static_cast<float>(p[4]), Ry);
  angle_axis_unit(2, static_cast<float>(p[5]), Rz);
  mul33f(Rx, Ry, A);  // (Translation * Rx) has linear = Rx exactly
  mul33f(A, Rz, B);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[i][j] = B[i][j];
    T[i][3] = static_cast<float>(p[i]);
    T[3][i] = 0.0f;
  }
  T[3][3] = 1.0f;
}

// [PCL 1.10] common/impl/transforms.hpp, detail::Transformer<float>::se3 (SSE2).
void transform_cloud(const std::vector<Pt>& in, std::vector<Pt>& out, const float T[4][4]) {
  if (&in != &out) out.resize(in.size());
  for (size_t i = 0; i < in.size(); i++) {
    float x = in[i].x, y = in[i].y, z = in[i].z;
    Pt o;
    o.x = x * T[0][0] + (y * T[0][1] + (z * T[0][2] + T[0][3]));
    o.y = x * T[1][0] + (y * T[1][1] + (z * T[1][2] + T[1][3]));
    o.z = x * T[2][0] + (y * T[2][1] + (z * T[2][2] + T[2][3]));
    o.w = x * T[3][0] + (y * T[3][1] + (z * T[3][2] + T[3][3]));
    out[i] = o;
  }
}

bool voxel_grid_filter(const std::vector<Pt>& in, bool is_dense, float leaf, std::vector<Pt>& out) {
  out.clear();
  if (in.empty()) return true;
  const float inv = 1.0f / leaf;
  float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const Pt& p : in) {
    if (!is_dense && (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))) continue;
    const float v[3] = {p.x, p.y, p.z};
    for (int k = 0; k < 3; k++) {
      min_p[k] = std::min(min_p[k], v[k]);
      max_p[k] = std::max(max_p[k], v[k]);
    }
  }
  if (!(min_p[0] <= max_p[0])) return true;
  int64_t d[3];
  for (int k = 0; k < 3; k++) d[k] = static_cast<int64_t>((max_p[k] - min_p[k]) * inv) + 1;
  if (d[0] * d[1] * d[2] > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) {
    out = in;  // "Leaf size is too small for the input dataset": output = *input_
    return false;
  }
  int min_b[3], div_b[3];
  for (int k = 0; k < 3; k++) {
    min_b[k] = static_cast<int>(std::floor(min_p[k] * inv));
    div_b[k] = static_cast<int>(std::floor(max_p[k] * inv)) - min_b[k] + 1;
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  std::vector<std::pair<unsigned, unsigned>> iv;  // (voxel idx, point index)
  iv.reserve(in.size());
  for (size_t i = 0; i < in.size(); i++) {
    const Pt& p = in[i];
    if (!is_dense && (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))) continue;
    const int i0 = static_cast<int>(std::floor(p.x * inv) - static_cast<float>(min_b[0]));
    const int i1 = static_cast<int>(std::floor(p.y * inv) - static_cast<float>(min_b[1]));
    const int i2 = static_cast<int>(std::floor(p.z * inv) - static_cast<float>(min_b[2]));
    iv.emplace_back(static_cast<unsigned>(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), static_cast<unsigned>(i));
  }
  std::sort(iv.begin(), iv.end());
  for (size_t a = 0; a < iv.size();) {
    size_t b = a;
    float sx = 0, sy = 0, sz = 0;
    while (b < iv.size() && iv[b].first == iv[a].first) {
      sx += in[iv[b].second].x;
      sy += in[iv[b].second].y;
      sz += in[iv[b].second].z;
      b++;
    }
    const float n = static_cast<float>(b - a);
    out.push_back(Pt{sx / n, sy / n, sz / n, 1.0f});
    a = b;
  }
  return true;
}

// ===========================================================================
// VoxelGrid  (voxel_grid_covariance_omp_impl.hpp)
// ===========================================================================

// [PCL] VoxelGrid::setLeafSize: inverse_leaf_size_ = 1 / leaf_size_ in f32.
void VoxelGrid::set_leaf_size(float l) {
  for (int k = 0; k < 3; k++) {
    leaf_size[k] = l;
    inv_leaf_size[k] = 1.0f / l;
  }
}

// applyFilter, voxel_grid_covariance_omp_impl.hpp:48-370 (no field filter,
// downsample_all_data_ = false as the ctor sets, .h:215).
void VoxelGrid::build(const std::vector<Pt>& cloud, bool is_dense) {
  leaves.clear();
  centroids.clear();
  centroid_leaf_idx.clear();
  overflow = false;
  if (cloud.empty()) return;  // :54-60 (no input) -- empty cloud leaves an empty grid

  // [PCL] getMinMax3D(cloud, min_p, max_p)  (:72)
  float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const Pt& p : cloud) {
    if (!is_dense && (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))) continue;
    const float v[3] = {p.x, p.y, p.z};
    for (int k = 0; k < 3; k++) {
      min_p[k] = std::min(min_p[k], v[k]);
      max_p[k] = std::max(max_p[k], v[k]);
    }
  }
  // :75-84 overflow guard
  int64_t d[3];
  for (int k = 0; k < 3; k++) d[k] = static_cast<int64_t>((max_p[k] - min_p[k]) * inv_leaf_size[k]) + 1;
  if (d[0] * d[1] * d[2] > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) {
    overflow = true;
    return;
  }
  // :87-103
  for (int k = 0; k < 3; k++) {
    min_b[k] = static_cast<int>(std::floor(min_p[k] * inv_leaf_size[k]));
    max_b[k] = static_cast<int>(std::floor(max_p[k] * inv_leaf_size[k]));
    div_b[k] = max_b[k] - min_b[k] + 1;
  }
  divb_mul[0] = 1;
  divb_mul[1] = div_b[0];
  divb_mul[2] = div_b[0] * div_b[1];

  // first pass :209-263
  for (const Pt& p : cloud) {
    if (!is_dense && (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))) continue;
    int ijk0 = static_cast<int>(std::floor(p.x * inv_leaf_size[0]) - static_cast<float>(min_b[0]));
    int ijk1 = static_cast<int>(std::floor(p.y * inv_leaf_size[1]) - static_cast<float>(min_b[1]));
    int ijk2 = static_cast<int>(std::floor(p.z * inv_leaf_size[2]) - static_cast<float>(min_b[2]));
    int idx = ijk0 * divb_mul[0] + ijk1 * divb_mul[1] + ijk2 * divb_mul[2];
    Leaf& leaf = leaves[static_cast<size_t>(idx)];
    const double pt[3] = {p.x, p.y, p.z};
    for (int i = 0; i < 3; i++) leaf.mean[i] += pt[i];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) leaf.cov.m[i][j] += pt[i] * pt[j];
    leaf.centroid[0] += p.x;
    leaf.centroid[1] += p.y;
    leaf.centroid[2] += p.z;
    ++leaf.nr_points;
  }

  // second pass :282-367
  for (auto& kv : leaves) {
    Leaf& leaf = kv.second;
    for (int k = 0; k < 4; k++) leaf.centroid[k] /= static_cast<float>(leaf.nr_points);
    double pt_sum[3] = {leaf.mean[0], leaf.mean[1], leaf.mean[2]};
    for (int k = 0; k < 3; k++) leaf.mean[k] /= leaf.nr_points;
    if (leaf.nr_points < min_points_per_voxel) continue;

    centroids.push_back(Pt{leaf.centroid[0], leaf.centroid[1], leaf.centroid[2], 1.0f});
    centroid_leaf_idx.push_back(static_cast<int>(kv.first));
    leaf.in_centroids = true;  // stays in the KD-tree even if rejected below (trap 7)

    // :329-330   cov_ started at Identity (trap 1)
    const double n = leaf.nr_points;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        leaf.cov.m[i][j] = (leaf.cov.m[i][j] - 2 * (pt_sum[i] * leaf.mean[j])) / n + leaf.mean[i] * leaf.mean[j];
    const double f = (leaf.nr_points - 1.0) / leaf.nr_points;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) leaf.cov.m[i][j] *= f;

    // :333-341
    V3 ev;
    eig3_sym(leaf.cov, ev, leaf.evecs);
    if (ev.v[0] < 0 || ev.v[1] < 0 || ev.v[2] <= 0) {
      leaf.nr_points = -1;
      continue;
    }
    // :345-357
    double min_ev = min_covar_eigvalue_mult * ev.v[2];
    if (ev.v[0] < min_ev) {
      ev.v[0] = min_ev;
      if (ev.v[1] < min_ev) ev.v[1] = min_ev;
      M3 vinv = inv3(leaf.evecs), vl, c;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) vl.m[i][j] = leaf.evecs.m[i][j] * ev.v[j];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          c.m[i][j] = (vl.m[i][0] * vinv.m[0][j] + vl.m[i][1] * vinv.m[1][j]) + vl.m[i][2] * vinv.m[2][j];
      leaf.cov = c;
    }
    for (int k = 0; k < 3; k++) leaf.evals[k] = ev.v[k];
    // :359-364
    leaf.icov = inv3(leaf.cov);
    double mx = -DBL_MAX, mn = DBL_MAX;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        mx = std::max(mx, leaf.icov.m[i][j]);
        mn = std::min(mn, leaf.icov.m[i][j]);
      }
    if (mx == std::numeric_limits<float>::infinity() || mn == -std::numeric_limits<float>::infinity())
      leaf.nr_points = -1;
  }
}

// getNeighborhoodAtPoint (:373-404) + wrappers (:407-442).
// [PCL] getAllNeighborCellIndices(): 3x3x3 offsets with i outer, j, k inner over
// {-1,0,1}, the centre (0,0,0) removed.
int VoxelGrid::neighbors(const Pt& p, SearchMethod m, const Leaf** out, float radius) const {
  if (m == KDTREE) {
    // radiusSearch (voxel_grid_covariance_omp.h:476-505): [PCL] KdTreeFLANN over voxel_centroids_
    // (f32 x,y,z), [FLANN] L2_Simple distance accumulated in f32, RadiusResultSet keeps
    // dist < radius^2 (strict), results sorted by distance.  A centroid lies inside its voxel, so
    // every hit is in the 3x3x3 cells around the query; rejected leaves (nr_points == -1) are
    // still in the centroid cloud and ARE returned (trap 7).
    if (leaves.empty()) return 0;
    const int ijk[3] = {static_cast<int>(std::floor(p.x / leaf_size[0])), static_cast<int>(std::floor(p.y / leaf_size[1])),
                        static_cast<int>(std::floor(p.z / leaf_size[2]))};
    const float r2 = static_cast<float>(static_cast<double>(radius) * static_cast<double>(radius));
    struct Hit { float d; size_t key; const Leaf* leaf; };
    Hit hits[27];
    int n = 0;
    for (int a = -1; a <= 1; a++)
      for (int b = -1; b <= 1; b++)
        for (int c = -1; c <= 1; c++) {
          const int q[3] = {ijk[0] + a, ijk[1] + b, ijk[2] + c};
          bool inside = true;
          for (int k = 0; k < 3; k++)
            if (q[k] < min_b[k] || q[k] > max_b[k]) inside = false;
          if (!inside) continue;
          size_t key = 0;
          for (int k = 0; k < 3; k++) key += static_cast<size_t>(q[k] - min_b[k]) * divb_mul[k];
          const Leaf* lp = find_leaf(key);
          if (!lp || !lp->in_centroids) continue;
          const Leaf& lf = *lp;
          const float dx = p.x - lf.centroid[0], dy = p.y - lf.centroid[1], dz = p.z - lf.centroid[2];
          float d = 0.0f;
          d += dx * dx;
          d += dy * dy;
          d += dz * dz;
          if (d < r2) hits[n++] = Hit{d, key, &lf};
        }
    std::sort(hits, hits + n, [](const Hit& x, const Hit& y) { return x.d < y.d || (x.d == y.d && x.key < y.key); });
    for (int i = 0; i < n; i++) out[i] = hits[i].leaf;
    return n;
  }
  static const int rel7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  int rel26[26][3];
  const int (*rel)[3];
  int nrel;
  if (m == DIRECT1) {
    rel = rel7;
    nrel = 1;
  } else if (m == DIRECT26) {
    int c = 0;
    for (int i = -1; i <= 1; i++)
      for (int j = -1; j <= 1; j++)
        for (int k = -1; k <= 1; k++) {
          if (i == 0 && j == 0 && k == 0) continue;
          rel26[c][0] = i;
          rel26[c][1] = j;
          rel26[c][2] = k;
          c++;
        }
    rel = rel26;
    nrel = 26;
  } else {  // DIRECT7 and the `default:` label (ndt_omp_impl.hpp:240-243)
    rel = rel7;
    nrel = 7;
  }
  if (leaves.empty()) return 0;
  // :379-381  note the DIVISION by leaf size (trap 2)
  int ijk[3] = {static_cast<int>(std::floor(p.x / leaf_size[0])), static_cast<int>(std::floor(p.y / leaf_size[1])),
                static_cast<int>(std::floor(p.z / leaf_size[2]))};
  int n = 0;
  for (int ni = 0; ni < nrel; ni++) {
    bool inside = true;
    for (int k = 0; k < 3; k++) {
      if (!(min_b[k] - ijk[k] <= rel[ni][k] && max_b[k] - ijk[k] >= rel[ni][k])) inside = false;
    }
    if (!inside) continue;
    int key = 0;
    for (int k = 0; k < 3; k++) key += (ijk[k] + rel[ni][k] - min_b[k]) * divb_mul[k];
    const Leaf* lp = find_leaf(static_cast<size_t>(key));
    if (lp && lp->nr_points >= min_points_per_voxel) out[n++] = lp;
  }
  return n;
}

// ===========================================================================
// NDT  (ndt_omp.h, ndt_omp_impl.hpp)
// ===========================================================================

void VoxelGrid::build_dense() {
  dense.clear();
  if (leaves.empty() || overflow) return;
  const size_t n_cells = static_cast<size_t>(div_b[0]) * div_b[1] * div_b[2];
  if (n_cells > (size_t(1) << 28)) return;  // keep the map for absurdly sparse grids
  dense.assign(n_cells, nullptr);
  for (const auto& kv : leaves)
    if (kv.first < n_cells) dense[kv.first] = &kv.second;
}

void NDT::set_target(const std::vector<Pt>& t, bool is_dense) {
  target = t;
  target_dense = is_dense;
  grid.set_leaf_size(resolution);  // init(), ndt_omp.h:276-283
  grid.build(target, target_dense);
  grid.dense.clear();
  if (optimised) grid.build_dense();
}

void NDT::set_resolution(float r) {  // ndt_omp.h:132-142 (tests input_, the SOURCE)
  if (resolution != r) {
    resolution = r;
    if (!source.empty()) {
      grid.set_leaf_size(resolution);
      grid.build(target, target_dense);
      grid.dense.clear();
      if (optimised) grid.build_dense();
    }
  }
}

void NDT::compute_gauss() {  // ndt_omp_impl.hpp:86-93
  double c1 = 10 * (1 - outlier_ratio);
  double c2 = outlier_ratio / std::pow(static_cast<double>(resolution), 3);
  gauss_d3 = -std::log(c2);
  gauss_d1 = -std::log(c1 + c2) - gauss_d3;
  gauss_d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - gauss_d3) / gauss_d1);
}

// ndt_omp_impl.hpp:288-395
void NDT::compute_angle_derivatives(const double p[6]) {
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
  if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
  if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }

  const double J[8][3] = {
      {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},  // a
      {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},    // b
      {(-sy * cz), sy * sz, cy},                                           // c
      {sx * cy * cz, (-sx * cy * sz), sx * sy},                            // d
      {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},                         // e
      {(-cy * sz), (-cy * cz), 0},                                         // f
      {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},            // g
      {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};            // h
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 3; c++) {
      j_ang_d[r][c] = J[r][c];
      j_ang[r][c] = static_cast<float>(J[r][c]);
    }
    j_ang[r][3] = 0.0f;
  }
  // f64 vectors :351-371 ; f32 rows :374-393.  Row 6 (d1) z-component:
  // -sy in the f64 vector (:361), +sy in the f32 matrix (:383)  -- trap 3.
  const double Hh[15][3] = {
      {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},     // a2
      {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},  // a3
      {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},                        // b2
      {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},                        // b3
      {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},            // c2
      {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},            // c3
      {(-cy * cz), (cy * sz), (-sy)},                                      // d1 (f64)
      {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},                        // d2
      {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},                       // d3
      {(sy * sz), (sy * cz), 0},                                           // e1
      {(-sx * cy * sz), (-sx * cy * cz), 0},                               // e2
      {(cx * cy * sz), (cx * cy * cz), 0},                                 // e3
      {(-cy * cz), (cy * sz), 0},                                          // f1
      {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},           // f2
      {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};          // f3
  for (int r = 0; r < 16; r++)
    for (int c = 0; c < 4; c++) h_ang[r][c] = 0.0f;
  for (int r = 0; r < 15; r++)
    for (int c = 0; c < 3; c++) {
      h_ang_d[r][c] = Hh[r][c];
      h_ang[r][c] = static_cast<float>(Hh[r][c]);
    }
  h_ang[6][2] = static_cast<float>(sy);  // :383
}

namespace {

// computePointDerivatives, f32 overload (ndt_omp_impl.hpp:398-440).
// J is the 3x6 live part of point_gradient_ (row 3 is zero); Hh[i] (i=3,4,5 <->
// blocks 3,4,5) holds the 3 live rows of each 4x6 block of point_hessian_.
struct PointDerivF {
  float J[3][6];
  float HE[3][3][6];  // [block-3][row][col]
};

inline void point_derivatives_f32(const float j_ang[8][4], const float h_ang[16][4], const double x[3], PointDerivF& d) {
  const float x4[3] = {static_cast<float>(x[0]), static_cast<float>(x[1]), static_cast<float>(x[2])};
  float xj[8], xh[15];
  for (int r = 0; r < 8; r++) xj[r] = (j_ang[r][0] * x4[0] + j_ang[r][1] * x4[1]) + j_ang[r][2] * x4[2];
  for (int r = 0; r < 15; r++) xh[r] = (h_ang[r][0] * x4[0] + h_ang[r][1] * x4[1]) + h_ang[r][2] * x4[2];
  std::memset(&d, 0, sizeof(d));
  d.J[0][0] = d.J[1][1] = d.J[2][2] = 1.0f;
  d.J[1][3] = xj[0];
  d.J[2][3] = xj[1];
  d.J[0][4] = xj[2];
  d.J[1][4] = xj[3];
  d.J[2][4] = xj[4];
  d.J[0][5] = xj[5];
  d.J[1][5] = xj[6];
  d.J[2][5] = xj[7];
  const float a[3] = {0, xh[0], xh[1]}, b[3] = {0, xh[2], xh[3]}, c[3] = {0, xh[4], xh[5]};
  const float dd[3] = {xh[6], xh[7], xh[8]}, e[3] = {xh[9], xh[10], xh[11]}, f[3] = {xh[12], xh[13], xh[14]};
  for (int r = 0; r < 3; r++) {
    d.HE[0][r][3] = a[r];
    d.HE[1][r][3] = b[r];
    d.HE[2][r][3] = c[r];
    d.HE[0][r][4] = b[r];
    d.HE[1][r][4] = dd[r];
    d.HE[2][r][4] = e[r];
    d.HE[0][r][5] = c[r];
    d.HE[1][r][5] = e[r];
    d.HE[2][r][5] = f[r];
  }
}

// updateDerivatives (ndt_omp_impl.hpp:484-537).  f32 math, f64 accumulation.
// Products of small fixed-size f32 matrices are k-ordered sums (Eigen's
// coefficient order); the all-zero 4th row/column contributes exact zeros.
inline double update_derivatives(double g_acc[6], double H_acc[36], const PointDerivF& d, const double x_trans[3],
                                 const M3& c_inv, double gauss_d1, double gauss_d2_d, bool compute_hessian) {
  const float x4[3] = {static_cast<float>(x_trans[0]), static_cast<float>(x_trans[1]), static_cast<float>(x_trans[2])};
  float c[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c[i][j] = static_cast<float>(c_inv.m[i][j]);
  const float gauss_d2 = static_cast<float>(gauss_d2_d);

  float xc[3];  // x_trans4 * c_inv4
  for (int j = 0; j < 3; j++) xc[j] = (x4[0] * c[0][j] + x4[1] * c[1][j]) + x4[2] * c[2][j];
  float q = (x4[0] * xc[0] + x4[1] * xc[1]) + x4[2] * xc[2];
  float e_x_cov_x = std::exp(-gauss_d2 * q * 0.5f);  // f32 overload (see DESIGN.md "exp ambiguity")
  float score_inc = static_cast<float>(-gauss_d1 * e_x_cov_x);
  e_x_cov_x = gauss_d2 * e_x_cov_x;
  if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) return 0;
  e_x_cov_x = static_cast<float>(e_x_cov_x * gauss_d1);

  float CJ[3][6];  // c_inv4 * point_gradient4
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 6; k++) CJ[i][k] = (c[i][0] * d.J[0][k] + c[i][1] * d.J[1][k]) + c[i][2] * d.J[2][k];
  float gk[6];  // x_trans4 * CJ
  for (int k = 0; k < 6; k++) gk[k] = (x4[0] * CJ[0][k] + x4[1] * CJ[1][k]) + x4[2] * CJ[2][k];
  for (int k = 0; k < 6; k++) g_acc[k] += static_cast<double>(e_x_cov_x * gk[k]);

  if (compute_hessian) {
    float JCJ[6][6];  // point_gradient4^T * CJ
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++) JCJ[a][b] = (d.J[0][a] * CJ[0][b] + d.J[1][a] * CJ[1][b]) + d.J[2][a] * CJ[2][b];
    for (int i = 0; i < 6; i++) {
      float xH[6];
      for (int j = 0; j < 6; j++) {
        if (i < 3)
          xH[j] = 0.0f;
        else
          xH[j] = (xc[0] * d.HE[i - 3][0][j] + xc[1] * d.HE[i - 3][1][j]) + xc[2] * d.HE[i - 3][2][j];
      }
      for (int j = 0; j < 6; j++)
        H_acc[i * 6 + j] += static_cast<double>(e_x_cov_x * (-gauss_d2 * gk[i] * gk[j] + xH[j] + JCJ[j][i]));
    }
  }
  return score_inc;
}

}  // namespace

// computeDerivatives, ndt_omp_impl.hpp:179-285.  Kept structurally faithful
// (per-evaluation allocation + zero fill of N x 344 B, guided schedule, serial
// ordered reduce) because this is also the CPU baseline.
double NDT::compute_derivatives(double g[6], double H[36], const std::vector<Pt>& trans_cloud, const double p[6],
                                bool compute_hessian) {
  n_evals++;
  const size_t N = source.size();
  for (int i = 0; i < 6; i++) g[i] = 0;
  for (int i = 0; i < 36; i++) H[i] = 0;
  double score = 0;

  if (optimised) {  // "optimised CPU" baseline: same arithmetic per neighbour, no per-point result arrays
    compute_angle_derivatives(p);
    long long nn_total = 0;
    const int nthreads = std::max(1, num_threads);
#pragma omp parallel num_threads(nthreads)
    {
      double s_loc = 0, g_loc[6] = {0, 0, 0, 0, 0, 0}, H_loc[36];
      long long nn_loc = 0;
      for (int k = 0; k < 36; k++) H_loc[k] = 0;
#pragma omp for schedule(static)
      for (size_t idx = 0; idx < N; idx++) {
        const Pt x_trans_pt = trans_cloud[idx];
        const Leaf* nb[27];
        const int n_nb = grid.neighbors(x_trans_pt, search_method, nb, resolution);
        if (n_nb == 0) continue;
        const double x[3] = {source[idx].x, source[idx].y, source[idx].z};
        PointDerivF d;
        point_derivatives_f32(j_ang, h_ang, x, d);  // once per point
        for (int ni = 0; ni < n_nb; ni++) {
          double x_trans[3] = {x_trans_pt.x, x_trans_pt.y, x_trans_pt.z};
          for (int k = 0; k < 3; k++) x_trans[k] -= nb[ni]->mean[k];
          s_loc += update_derivatives(g_loc, H_loc, d, x_trans, nb[ni]->icov, gauss_d1, gauss_d2, compute_hessian);
        }
        nn_loc += n_nb;
      }
#pragma omp critical
      {
        score += s_loc;
        for (int k = 0; k < 6; k++) g[k] += g_loc[k];
        for (int k = 0; k < 36; k++) H[k] += H_loc[k];
        nn_total += nn_loc;
      }
    }
    mean_neighbors = N ? static_cast<double>(nn_total) / N : 0.0;
    return score;
  }

  std::vector<double> scores(N);
  std::vector<double> grads(N * 6);
  std::vector<double> hess(N * 36);
  std::vector<int> nn(N);
  for (size_t i = 0; i < N; i++) {
    scores[i] = 0;
    for (int k = 0; k < 6; k++) grads[i * 6 + k] = 0;
    for (int k = 0; k < 36; k++) hess[i * 36 + k] = 0;
  }
  compute_angle_derivatives(p);  // :200

  const int nthreads = std::max(1, num_threads);
#pragma omp parallel for num_threads(nthreads) schedule(guided, 8)
  for (size_t idx = 0; idx < N; idx++) {
    const Pt x_trans_pt = trans_cloud[idx];
    const Leaf* nb[27];
    int n_nb = grid.neighbors(x_trans_pt, search_method, nb, resolution);
    double score_pt = 0, g_pt[6] = {0, 0, 0, 0, 0, 0}, H_pt[36];
    for (int k = 0; k < 36; k++) H_pt[k] = 0;
    PointDerivF d;
    for (int ni = 0; ni < n_nb; ni++) {
      const Leaf* cell = nb[ni];
      const Pt x_pt = source[idx];
      const double x[3] = {x_pt.x, x_pt.y, x_pt.z};
      double x_trans[3] = {x_trans_pt.x, x_trans_pt.y, x_trans_pt.z};
      for (int k = 0; k < 3; k++) x_trans[k] -= cell->mean[k];  // :262 (f64)
      point_derivatives_f32(j_ang, h_ang, x, d);                // :267
      score_pt += update_derivatives(g_pt, H_pt, d, x_trans, cell->icov, gauss_d1, gauss_d2, compute_hessian);
    }
    scores[idx] = score_pt;
    for (int k = 0; k < 6; k++) grads[idx * 6 + k] = g_pt[k];
    for (int k = 0; k < 36; k++) hess[idx * 36 + k] = H_pt[k];
    nn[idx] = n_nb;
  }
  // :278-282
  long long nn_total = 0;
  for (size_t i = 0; i < N; i++) {
    score += scores[i];
    for (int k = 0; k < 6; k++) g[k] += grads[i * 6 + k];
    for (int k = 0; k < 36; k++) H[k] += hess[i * 36 + k];
    nn_total += nn[i];
  }
  mean_neighbors = N ? static_cast<double>(nn_total) / N : 0.0;
  return score;
}

// computeHessian + updateHessian + f64 computePointDerivatives,
// ndt_omp_impl.hpp:540-645, 443-481.
void NDT::compute_hessian(double H[36], const std::vector<Pt>& trans_cloud) {
  n_hess++;
  for (int i = 0; i < 36; i++) H[i] = 0;
  const size_t N = source.size();
  // serial in the reference (and here by default: one thread, same summation order); the optimised
  // baseline spreads the points over the threads
  const int nthreads = optimised ? std::max(1, num_threads) : 1;
#pragma omp parallel num_threads(nthreads)
  {
  double Hl[36];
  for (int i = 0; i < 36; i++) Hl[i] = 0;
#pragma omp for schedule(static)
  for (size_t idx = 0; idx < N; idx++) {
    const Pt x_trans_pt = trans_cloud[idx];
    const Leaf* nb[27];
    int n_nb = grid.neighbors(x_trans_pt, search_method, nb, resolution);
    for (int ni = 0; ni < n_nb; ni++) {
      const Leaf* cell = nb[ni];
      const double x[3] = {source[idx].x, source[idx].y, source[idx].z};
      double xt[3] = {x_trans_pt.x, x_trans_pt.y, x_trans_pt.z};
      for (int k = 0; k < 3; k++) xt[k] -= cell->mean[k];
      const M3& C = cell->icov;
      // f64 computePointDerivatives :443-481
      double J[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
      auto dot3 = [&](const double a[3], const double b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; };
      J[1][3] = dot3(x, j_ang_d[0]);
      J[2][3] = dot3(x, j_ang_d[1]);
      J[0][4] = dot3(x, j_ang_d[2]);
      J[1][4] = dot3(x, j_ang_d[3]);
      J[2][4] = dot3(x, j_ang_d[4]);
      J[0][5] = dot3(x, j_ang_d[5]);
      J[1][5] = dot3(x, j_ang_d[6]);
      J[2][5] = dot3(x, j_ang_d[7]);
      double xh[15];
      for (int r = 0; r < 15; r++) xh[r] = dot3(x, h_ang_d[r]);
      const double a[3] = {0, xh[0], xh[1]}, b[3] = {0, xh[2], xh[3]}, c[3] = {0, xh[4], xh[5]};
      const double dd[3] = {xh[6], xh[7], xh[8]}, e[3] = {xh[9], xh[10], xh[11]}, f[3] = {xh[12], xh[13], xh[14]};
      double HE[6][3][6];
      std::memset(HE, 0, sizeof(HE));
      for (int r = 0; r < 3; r++) {
        HE[3][r][3] = a[r]; HE[4][r][3] = b[r]; HE[5][r][3] = c[r];
        HE[3][r][4] = b[r]; HE[4][r][4] = dd[r]; HE[5][r][4] = e[r];
        HE[3][r][5] = c[r]; HE[4][r][5] = e[r]; HE[5][r][5] = f[r];
      }
      // updateHessian :613-645
      auto matvec = [&](const double v[3], double o[3]) {
        for (int i = 0; i < 3; i++) o[i] = (C.m[i][0] * v[0] + C.m[i][1] * v[1]) + C.m[i][2] * v[2];
      };
      double Cx[3];
      matvec(xt, Cx);
      double e_x_cov_x = gauss_d2 * std::exp(-gauss_d2 * dot3(xt, Cx) / 2);
      if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
      e_x_cov_x *= gauss_d1;
      for (int i = 0; i < 6; i++) {
        double Ji[3] = {J[0][i], J[1][i], J[2][i]}, cov_dxd_pi[3];
        matvec(Ji, cov_dxd_pi);
        for (int j = 0; j < 6; j++) {
          double Jj[3] = {J[0][j], J[1][j], J[2][j]}, CJj[3], Hb[3] = {HE[i][0][j], HE[i][1][j], HE[i][2][j]}, CHb[3];
          matvec(Jj, CJj);
          matvec(Hb, CHb);
          Hl[i * 6 + j] += e_x_cov_x * (-gauss_d2 * dot3(xt, cov_dxd_pi) * dot3(xt, CJj) + dot3(xt, CHb) + dot3(Jj, cov_dxd_pi));
        }
      }
    }
  }
#pragma omp critical
  for (int i = 0; i < 36; i++) H[i] += Hl[i];
  }
}

// calculateScore, ndt_omp_impl.hpp:935-983
double NDT::calculate_score(const std::vector<Pt>& trans_cloud) const {
  double score = 0;
  for (size_t idx = 0; idx < trans_cloud.size(); idx++) {
    const Pt x_trans_pt = trans_cloud[idx];
    const Leaf* nb[27];
    int n_nb = grid.neighbors(x_trans_pt, search_method, nb, resolution);
    for (int ni = 0; ni < n_nb; ni++) {
      const Leaf* cell = nb[ni];
      double xt[3] = {x_trans_pt.x, x_trans_pt.y, x_trans_pt.z};
      for (int k = 0; k < 3; k++) xt[k] -= cell->mean[k];
      const M3& C = cell->icov;
      double Cx[3];
      for (int i = 0; i < 3; i++) Cx[i] = (C.m[i][0] * xt[0] + C.m[i][1] * xt[1]) + C.m[i][2] * xt[2];
      double e_x_cov_x = std::exp(-gauss_d2 * ((xt[0] * Cx[0] + xt[1] * Cx[1]) + xt[2] * Cx[2]) / 2);
      double score_inc = -gauss_d1 * e_x_cov_x - gauss_d3;
      score += score_inc / n_nb;
    }
  }
  return score / static_cast<double>(trans_cloud.size());
}

// ---- More-Thuente (ndt_omp_impl.hpp:648-769, ndt_omp.h:430-447) ------------
namespace {
inline double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
inline double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

bool update_interval_mt(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t,
                        double f_t, double g_t) {
  if (f_t > f_l) {  // U1
    a_u = a_t; f_u = f_t; g_u = g_t;
    return false;
  } else if (g_t * (a_l - a_t) > 0) {  // U2
    a_l = a_t; f_l = f_t; g_l = g_t;
    return false;
  } else if (g_t * (a_l - a_t) < 0) {  // U3
    a_u = a_l; f_u = f_l; g_u = g_l;
    a_l = a_t; f_l = f_t; g_l = g_t;
    return false;
  }
  return true;
}

double trial_value_selection_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t,
                                double f_t, double g_t) {
  if (f_t > f_l) {  // case 1
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    if (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {  // case 2
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    if (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (std::fabs(g_t) <= std::fabs(g_l)) {  // case 3
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double a_t_next = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
    if (a_t > a_l) return std::min(a_t + 0.66 * (a_u - a_t), a_t_next);
    return std::max(a_t + 0.66 * (a_u - a_t), a_t_next);
  }
  // case 4
  double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
  double w = std::sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}
}  // namespace

// computeStepLengthMT, ndt_omp_impl.hpp:772-932
double NDT::step_length_mt(const double x[6], double step_dir[6], double step_init, double step_max, double step_min,
                           double& score, double g[6], double H[36], std::vector<Pt>& trans_cloud) {
  double phi_0 = -score;
  double d_phi_0 = 0;
  for (int i = 0; i < 6; i++) d_phi_0 += g[i] * step_dir[i];
  d_phi_0 = -d_phi_0;
  double x_t[6];
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) return 0;
    d_phi_0 *= -1;
    for (int i = 0; i < 6; i++) step_dir[i] *= -1;
  }
  const int max_step_iterations = 10;
  int step_iterations = 0;
  const double mu = 1.e-4, nu = 0.9;
  double a_l = 0, a_u = 0;
  double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu);
  double g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
  double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu);
  double g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
  bool interval_converged = (step_max - step_min) < 0, open_interval = true;
  double a_t = step_init;
  a_t = std::min(a_t, step_max);
  a_t = std::max(a_t, step_min);
  for (int i = 0; i < 6; i++) x_t[i] = x[i] + step_dir[i] * a_t;
  pose_to_matrix(x_t, final_transformation);
  transform_cloud(source, trans_cloud, final_transformation);
  score = compute_derivatives(g, H, trans_cloud, x_t, true);
  double phi_t = -score;
  double d_phi_t = 0;
  for (int i = 0; i < 6; i++) d_phi_t += g[i] * step_dir[i];
  d_phi_t = -d_phi_t;
  double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
  double d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);

  while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    if (open_interval)
      a_t = trial_value_selection_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
    else
      a_t = trial_value_selection_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_t = std::min(a_t, step_max);
    a_t = std::max(a_t, step_min);
    for (int i = 0; i < 6; i++) x_t[i] = x[i] + step_dir[i] * a_t;
    pose_to_matrix(x_t, final_transformation);
    transform_cloud(source, trans_cloud, final_transformation);
    score = compute_derivatives(g, H, trans_cloud, x_t, false);
    phi_t = -score;
    d_phi_t = 0;
    for (int i = 0; i < 6; i++) d_phi_t += g[i] * step_dir[i];
    d_phi_t = -d_phi_t;
    psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
    d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = false;
      f_l = f_l + phi_0 - mu * d_phi_0 * a_l;
      g_l = g_l + mu * d_phi_0;
      f_u = f_u + phi_0 - mu * d_phi_0 * a_u;
      g_u = g_u + mu * d_phi_0;
    }
    if (open_interval)
      interval_converged = update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
    else
      interval_converged = update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    step_iterations++;
  }
  if (step_iterations) compute_hessian(H, trans_cloud);  // :928-929
  return a_t;
}

// [PCL] Registration::align pre-amble (registration.hpp) + computeTransformation
// (ndt_omp_impl.hpp:80-171).
AlignResult NDT::align(const float guess[4][4], std::vector<Pt>* output_out) {
  AlignResult res;
  n_evals = 0;
  n_hess = 0;
  const size_t N = source.size();
  std::vector<Pt> output = source;  // copy input -> output
  bool converged = false;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) final_transformation[i][j] = (i == j) ? 1.0f : 0.0f;
  for (size_t i = 0; i < N; i++) output[i].w = 1.0f;  // data[3] = 1  (trap 12)

  int nr_iterations = 0;
  compute_gauss();
  bool guess_is_identity = true;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (guess[i][j] != ((i == j) ? 1.0f : 0.0f)) guess_is_identity = false;
  if (!guess_is_identity) {  // :95-101
    std::memcpy(final_transformation, guess, sizeof(float) * 16);
    transform_cloud(output, output, guess);
  }
  // :103-111
  double p[6], delta_p[6], g[6], H[36];
  float ang[3];
  euler_xyz_from_matrix(final_transformation, ang);
  for (int i = 0; i < 3; i++) {
    p[i] = final_transformation[i][3];
    p[3 + i] = ang[i];
  }
  double score = compute_derivatives(g, H, output, p, true);  // :119
  double trans_probability = 0;
  bool early_return = false;
  while (!converged) {
    double neg_g[6];
    for (int i = 0; i < 6; i++) neg_g[i] = -g[i];
    svd6_solve(H, neg_g, delta_p);  // :127-129
    double nrm2 = 0;
    for (int i = 0; i < 6; i++) nrm2 += delta_p[i] * delta_p[i];
    double delta_p_norm = std::sqrt(nrm2);
    if (delta_p_norm == 0 || delta_p_norm != delta_p_norm) {  // :134-139
      trans_probability = score / static_cast<double>(N);
      converged = (delta_p_norm == delta_p_norm);
      early_return = true;
      break;
    }
    for (int i = 0; i < 6; i++) delta_p[i] /= delta_p_norm;  // normalize()
    delta_p_norm = step_length_mt(p, delta_p, delta_p_norm, step_size, transformation_epsilon / 2, score, g, H, output);
    for (int i = 0; i < 6; i++) delta_p[i] *= delta_p_norm;
    for (int i = 0; i < 6; i++) p[i] = p[i] + delta_p[i];
    if (nr_iterations > max_iterations || (nr_iterations && (std::fabs(delta_p_norm) < transformation_epsilon)))
      converged = true;  // :158-162 (trap 9)
    nr_iterations++;
  }
  if (!early_return) trans_probability = score / static_cast<double>(N);  // :170
  std::memcpy(res.final_T, final_transformation, sizeof(float) * 16);
  res.converged = converged;
  res.nr_iterations = nr_iterations;
  res.trans_probability = trans_probability;
  res.n_evals = n_evals;
  res.n_hessian_recomputes = n_hess;
  if (output_out) *output_out = output;
  return res;
}

}  // namespace oracle
