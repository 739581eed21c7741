// gicp_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  See gicp_oracle.hpp.
// Line references are to /root/reference/ndt_omp/include/pclomp/gicp_omp_impl.hpp unless marked.
#include "gicp_oracle.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <unordered_map>

namespace oracle {

// ---------------------------------------------------------------------------
// exact k-NN ([PCL] KdTreeFLANN::nearestKSearch semantics: exact, ascending distance)
// ---------------------------------------------------------------------------
namespace {

inline float dist2_l2simple(const Pt& a, const Pt& b) {  // [FLANN] L2_Simple<float>: result += diff*diff, in order
  const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  float r = dx * dx;
  r += dy * dy;
  r += dz * dz;
  return r;
}

struct CellGrid {
  double h = 1.0;
  std::int64_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  std::unordered_map<std::uint64_t, std::vector<int>> cells;
  static std::uint64_t key(std::int64_t i, std::int64_t j, std::int64_t k) {
    return (static_cast<std::uint64_t>(i & 0x1FFFFF) << 42) | (static_cast<std::uint64_t>(j & 0x1FFFFF) << 21) |
           static_cast<std::uint64_t>(k & 0x1FFFFF);
  }
  std::int64_t cell_of(float v) const { return static_cast<std::int64_t>(std::floor(static_cast<double>(v) / h)); }
  void build(const std::vector<Pt>& cloud) {
    // cell edge from the occupied volume: about 2 points per cell for a volumetric cloud; surface-like
    // clouds end up with more, which only costs time
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (const Pt& p : cloud) {
      const double c[3] = {p.x, p.y, p.z};
      for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], c[a]); mx[a] = std::max(mx[a], c[a]); }
    }
    double vol = 1.0;
    for (int a = 0; a < 3; a++) vol *= std::max(mx[a] - mn[a], 1e-3);
    h = std::cbrt(vol * 2.0 / std::max<size_t>(cloud.size(), 1));
    const double ext = std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2], 1e-3});
    h = std::max(h, ext / 1000.0);  // 21-bit keys are ample
    // flat or thin clouds make the volume guess far too fine (every point alone in its cell, searches walk
    // thousands of empty cells): coarsen until an occupied cell holds a few points on average
    for (int pass = 0; pass < 12; pass++) {
      cells.clear();
      for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<std::int64_t>::max(); hi[a] = std::numeric_limits<std::int64_t>::min(); }
      for (size_t i = 0; i < cloud.size(); i++) {
        const std::int64_t c[3] = {cell_of(cloud[i].x), cell_of(cloud[i].y), cell_of(cloud[i].z)};
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
        cells[key(c[0], c[1], c[2])].push_back(static_cast<int>(i));
      }
      if (cells.size() * 3 <= cloud.size() || cells.size() <= 8) break;
      h *= 1.6;
    }
  }
};

struct Cand {
  float d;
  int idx;
  bool operator<(const Cand& o) const { return d < o.d || (d == o.d && idx < o.idx); }
};

void knn_one(const CellGrid& g, const std::vector<Pt>& cloud, const Pt& q, int k, std::vector<Cand>& best) {
  best.clear();
  std::int64_t c[3] = {g.cell_of(q.x), g.cell_of(q.y), g.cell_of(q.z)};
  std::int64_t far = 0;  // shells needed to cover every occupied cell
  for (int a = 0; a < 3; a++) far = std::max({far, c[a] - g.lo[a], g.hi[a] - c[a]});
  auto offer = [&](int idx) {
    const Cand cd{dist2_l2simple(q, cloud[idx]), idx};
    if (static_cast<int>(best.size()) < k) {
      best.insert(std::upper_bound(best.begin(), best.end(), cd), cd);
    } else if (cd < best.back()) {
      best.pop_back();
      best.insert(std::upper_bound(best.begin(), best.end(), cd), cd);
    }
  };
  for (std::int64_t r = 0; r <= far; r++) {
    if (r > 24) {  // an isolated query: walking ever larger shells costs more than looking at every point
      best.clear();
      for (size_t i = 0; i < cloud.size(); i++) offer(static_cast<int>(i));
      return;
    }
    for (std::int64_t dz = -r; dz <= r; dz++)
      for (std::int64_t dy = -r; dy <= r; dy++)
        for (std::int64_t dx = -r; dx <= r; dx++) {
          if (std::max({std::llabs(dx), std::llabs(dy), std::llabs(dz)}) != r) continue;
          const std::int64_t x = c[0] + dx, y = c[1] + dy, z = c[2] + dz;
          if (x < g.lo[0] || x > g.hi[0] || y < g.lo[1] || y > g.hi[1] || z < g.lo[2] || z > g.hi[2]) continue;
          auto it = g.cells.find(CellGrid::key(x, y, z));
          if (it == g.cells.end()) continue;
          for (int idx : it->second) offer(idx);
        }
    // every unvisited point is at least r cells away from the query
    if (static_cast<int>(best.size()) == k) {
      const double reach = static_cast<double>(r) * g.h * (1.0 - 1e-6);
      if (static_cast<double>(best.back().d) < reach * reach) break;
    }
  }
}

}  // namespace

void knn_exact(const std::vector<Pt>& cloud, const std::vector<Pt>& query, int k, std::vector<int>& out_idx,
               std::vector<float>& out_d2) {
  CellGrid g;
  g.build(cloud);
  out_idx.assign(query.size() * static_cast<size_t>(k), -1);
  out_d2.assign(query.size() * static_cast<size_t>(k), std::numeric_limits<float>::infinity());
#pragma omp parallel
  {
    std::vector<Cand> best;
#pragma omp for schedule(dynamic, 64)
    for (long long i = 0; i < static_cast<long long>(query.size()); i++) {
      knn_one(g, cloud, query[i], k, best);
      for (size_t j = 0; j < best.size(); j++) {
        out_idx[static_cast<size_t>(i) * k + j] = best[j].idx;
        out_d2[static_cast<size_t>(i) * k + j] = best[j].d;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// computeCovariances, :48-116
// ---------------------------------------------------------------------------
bool GICP::covariances(const std::vector<Pt>& cloud, int k, double gicp_epsilon, std::vector<M3>& out) {
  if (k > static_cast<int>(cloud.size())) return false;  // :53-57 (PCL_ERROR + return)
  out.resize(cloud.size());
  std::vector<int> nn;
  std::vector<float> d2;
  knn_exact(cloud, cloud, k, nn, d2);
#pragma omp parallel for
  for (long long i = 0; i < static_cast<long long>(cloud.size()); i++) {
    double mean[3] = {0, 0, 0};
    M3 cov{};
    for (int j = 0; j < k; j++) {  // :81-95: f32 products, f64 sums, neighbours in ascending distance
      const Pt& pt = cloud[nn[static_cast<size_t>(i) * k + j]];
      mean[0] += pt.x;
      mean[1] += pt.y;
      mean[2] += pt.z;
      cov.m[0][0] += pt.x * pt.x;
      cov.m[1][0] += pt.y * pt.x;
      cov.m[1][1] += pt.y * pt.y;
      cov.m[2][0] += pt.z * pt.x;
      cov.m[2][1] += pt.z * pt.y;
      cov.m[2][2] += pt.z * pt.z;
    }
    for (int a = 0; a < 3; a++) mean[a] /= static_cast<double>(k);
    for (int a = 0; a < 3; a++)  // :99-105
      for (int l = 0; l <= a; l++) {
        cov.m[a][l] /= static_cast<double>(k);
        cov.m[a][l] -= mean[a] * mean[l];
        cov.m[l][a] = cov.m[a][l];
      }
    // :108-120 [Eigen] JacobiSVD of a symmetric matrix: U = eigenvectors, singular values = |eigenvalues|
    // in descending order; the two largest become 1, the smallest gicp_epsilon
    V3 ev;
    M3 U;
    eig3_sym(cov, ev, U);
    int order[3] = {2, 1, 0};  // eig3_sym: ascending eigenvalues
    std::stable_sort(order, order + 3, [&](int a, int b) { return std::fabs(ev.v[a]) > std::fabs(ev.v[b]); });
    M3 rec{};
    for (int s = 0; s < 3; s++) {
      const int c = order[s];
      const double v = (s == 2) ? gicp_epsilon : 1.0;
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) rec.m[a][b] += (v * U.m[a][c]) * U.m[b][c];
    }
    out[i] = rec;
  }
  return true;
}

// ---------------------------------------------------------------------------
// applyState on the identity, :519-532.  [Eigen] AngleAxisf * AngleAxisf is a quaternion product and
// the Matrix3f is Quaternion::toRotationMatrix(); f32 throughout.
// ---------------------------------------------------------------------------
namespace {
struct Qf { float w, x, y, z; };
Qf quat_axis(int axis, float angle) {  // Quaternion = AngleAxis
  const float ha = 0.5f * angle;
  Qf q{std::cos(ha), 0.0f, 0.0f, 0.0f};
  const float s = std::sin(ha);
  (axis == 0 ? q.x : axis == 1 ? q.y : q.z) = s * 1.0f;
  return q;
}
Qf quat_mul(const Qf& a, const Qf& b) {
  return Qf{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
}  // namespace

void GICP::apply_state(const double x[6], float T[4][4]) {
  const Qf q = quat_mul(quat_mul(quat_axis(2, static_cast<float>(x[5])), quat_axis(1, static_cast<float>(x[4]))),
                        quat_axis(0, static_cast<float>(x[3])));
  const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  const float R[3][3] = {{1.0f - (tyy + tzz), txy - twz, txz + twy},
                         {txy + twz, 1.0f - (txx + tzz), tyz - twx},
                         {txz - twy, tyz + twx, 1.0f - (txx + tyy)}};
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[r][c] = R[r][c];  // R * I
    T[r][3] = static_cast<float>(x[r]);             // 0 + T
    T[3][r] = 0.0f;
  }
  T[3][3] = 1.0f;
}

// computeRDerivative, :119-178
void GICP::r_derivative(const double x[6], const double R[3][3], double g[6]) {
  const double phi = x[3], theta = x[4], psi = x[5];
  const double cphi = std::cos(phi), sphi = std::sin(phi);
  const double ctheta = std::cos(theta), stheta = std::sin(theta);
  const double cpsi = std::cos(psi), spsi = std::sin(psi);
  double dphi[3][3], dth[3][3], dpsi[3][3];
  dphi[0][0] = 0; dphi[1][0] = 0; dphi[2][0] = 0;
  dphi[0][1] = sphi * spsi + cphi * cpsi * stheta;
  dphi[1][1] = -cpsi * sphi + cphi * spsi * stheta;
  dphi[2][1] = cphi * ctheta;
  dphi[0][2] = cphi * spsi - cpsi * sphi * stheta;
  dphi[1][2] = -cphi * cpsi - sphi * spsi * stheta;
  dphi[2][2] = -ctheta * sphi;
  dth[0][0] = -cpsi * stheta; dth[1][0] = -spsi * stheta; dth[2][0] = -ctheta;
  dth[0][1] = cpsi * ctheta * sphi; dth[1][1] = ctheta * sphi * spsi; dth[2][1] = -sphi * stheta;
  dth[0][2] = cphi * cpsi * ctheta; dth[1][2] = cphi * ctheta * spsi; dth[2][2] = -cphi * stheta;
  dpsi[0][0] = -ctheta * spsi; dpsi[1][0] = cpsi * ctheta; dpsi[2][0] = 0;
  dpsi[0][1] = -cphi * cpsi - sphi * spsi * stheta; dpsi[1][1] = -cphi * spsi + cpsi * sphi * stheta; dpsi[2][1] = 0;
  dpsi[0][2] = cpsi * sphi - cphi * spsi * stheta; dpsi[1][2] = sphi * spsi + cphi * cpsi * stheta; dpsi[2][2] = 0;
  auto inner = [&](const double A[3][3]) {  // matricesInnerProd, gicp_omp.h:318-327
    double r = 0.0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r += A[j][i] * R[i][j];
    return r;
  };
  g[3] = inner(dphi);
  g[4] = inner(dth);
  g[5] = inner(dpsi);
}

// ---------------------------------------------------------------------------
// OptimizationFunctorWithIndices, :241-368.  base_transformation_ is the identity for the whole
// align (:399), so transformation_matrix = T(x) and the rotation-gradient point is p_src itself.
// ---------------------------------------------------------------------------
namespace {
inline void mat4f_vec(const float T[4][4], const Pt& p, float out[4]) {  // [Eigen] Matrix4f * Vector4f, column by column
  for (int r = 0; r < 4; r++) out[r] = ((T[r][0] * p.x + T[r][1] * p.y) + T[r][2] * p.z) + T[r][3] * 1.0f;
}
}  // namespace

static double functor_f_sum(const GICP& s, const float T[4][4]) {
  const int m = static_cast<int>(s.corr_src.size());
  double f = 0.0;
  for (int i = 0; i < m; i++) {
    const Pt& ps = (*s.opt_src)[s.corr_src[i]];
    const Pt& pt = s.target[s.corr_tgt[i]];
    float pp[4];
    mat4f_vec(T, ps, pp);
    const float res[3] = {pp[0] - pt.x, pp[1] - pt.y, pp[2] - pt.z};  // 4th component 1 - 1 = 0
    const std::array<float, 9>& M = s.mahalanobis[s.corr_src[i]];
    float mr[3];
    for (int r = 0; r < 3; r++) mr[r] = (M[r * 3 + 0] * res[0] + M[r * 3 + 1] * res[1]) + M[r * 3 + 2] * res[2];
    // [Eigen] 4-wide dot: (p0 + p2) + (p1 + p3), p3 = 0
    const float ret = (res[0] * mr[0] + res[2] * mr[2]) + res[1] * mr[1];
    f += static_cast<double>(ret);
  }
  return f;
}

double GICP::functor_f(const double x[6]) const {  // operator(), :241-274: f32 quadratic form, f64 sum
  n_f++;
  float T[4][4];
  apply_state(x, T);
  return functor_f_sum(*this, T) / static_cast<int>(corr_src.size());
}

namespace {
struct Sums { double f = 0, g[3] = {0, 0, 0}, R[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; };
}

static void functor_sums(const GICP& s, const float T[4][4], Sums& acc) {
  const int m = static_cast<int>(s.corr_src.size());
  for (int i = 0; i < m; i++) {
    const Pt& ps = (*s.opt_src)[s.corr_src[i]];
    const Pt& pt = s.target[s.corr_tgt[i]];
    float pp[4];
    mat4f_vec(T, ps, pp);
    const double res[3] = {static_cast<double>(pp[0] - pt.x), static_cast<double>(pp[1] - pt.y),
                           static_cast<double>(pp[2] - pt.z)};  // f32 difference, then widened (:299, :343)
    const std::array<float, 9>& M = s.mahalanobis[s.corr_src[i]];
    double temp[3];
    for (int r = 0; r < 3; r++)
      temp[r] = (static_cast<double>(M[r * 3 + 0]) * res[0] + static_cast<double>(M[r * 3 + 1]) * res[1]) +
                static_cast<double>(M[r * 3 + 2]) * res[2];
    acc.f += (res[0] * temp[0] + res[1] * temp[1]) + res[2] * temp[2];
    const double p3[3] = {ps.x, ps.y, ps.z};  // base_transformation_ (identity) * p_src
    for (int r = 0; r < 3; r++) {
      acc.g[r] += temp[r];
      for (int c = 0; c < 3; c++) acc.R[r][c] += p3[r] * temp[c];
    }
  }
}

void GICP::functor_raw(int mode, const float T[4][4], double out[14]) const {
  for (int i = 0; i < 14; i++) out[i] = 0.0;
  out[13] = static_cast<double>(corr_src.size());
  if (mode == 0) {
    out[0] = functor_f_sum(*this, T);
    return;
  }
  Sums acc;
  functor_sums(*this, T, acc);
  out[0] = acc.f;
  for (int r = 0; r < 3; r++) {
    out[1 + r] = acc.g[r];
    for (int c = 0; c < 3; c++) out[4 + r * 3 + c] = acc.R[r][c];
  }
}

void GICP::functor_df(const double x[6], double g[6]) const {  // :277-331
  n_df++;
  float T[4][4];
  apply_state(x, T);
  Sums acc;
  functor_sums(*this, T, acc);
  const int m = static_cast<int>(corr_src.size());
  for (int r = 0; r < 3; r++) g[r] = acc.g[r] * (2.0 / m);
  double R[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[r][c] = acc.R[r][c] * (2.0 / m);
  r_derivative(x, R, g);
}

void GICP::functor_fdf(const double x[6], double& f, double g[6]) const {  // :334-368
  n_fdf++;
  float T[4][4];
  apply_state(x, T);
  Sums acc;
  functor_sums(*this, T, acc);
  const int m = static_cast<int>(corr_src.size());
  f = acc.f / static_cast<double>(m);
  for (int r = 0; r < 3; r++) g[r] = acc.g[r] * (2.0 / m);
  double R[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[r][c] = acc.R[r][c] * (2.0 / m);
  r_derivative(x, R, g);
}

// ---------------------------------------------------------------------------
// [PCL 1.10] BFGS<FunctorType> (registration/bfgs.h): GSL's vector_bfgs2 with Fletcher's line
// search.  Restated with its PCL-specific conditions: the cubic interpolation is only taken when
// fpb == fpa (bfgs.h tests `!(fpb != fpa)` where GSL tests "fpb is a number"), and the quadratic
// branch requires `c > a` where GSL requires c > 0.
// ---------------------------------------------------------------------------
namespace {

enum BfgsStatus { NegativeGradientEpsilon = -3, NotStarted = -2, Running = -1, Success = 0, NoProgress = 1 };

struct Vec6 {
  double v[6];
  double dot(const Vec6& o) const {
    double s = 0;
    for (int i = 0; i < 6; i++) s += v[i] * o.v[i];
    return s;
  }
  double norm() const { return std::sqrt(dot(*this)); }
};

class Bfgs {
 public:
  // Parameters of estimateRigidTransformationBFGS, gicp_omp_impl.hpp:212-217; the rest are PCL's defaults
  double sigma = 0.01, rho = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 1.0;
  int order = 3, bracket_iters = 100, section_iters = 100;
  explicit Bfgs(const GICP& fn) : fn_(fn) {}

  BfgsStatus minimize_init(Vec6& x) {
    iter_ = 0;
    delta_f_ = 0;
    for (double& d : dx_.v) d = 0;
    fn_.functor_fdf(x.v, f_, gradient_.v);
    x0_ = x;
    g0_ = gradient_;
    g0norm_ = g0_.norm();
    for (int i = 0; i < 6; i++) p_.v[i] = gradient_.v[i] * -1 / g0norm_;
    pnorm_ = p_.norm();
    fp0_ = -g0norm_;
    x_alpha_ = x0_; x_cache_key_ = 0;
    f_alpha_ = f_; f_cache_key_ = 0;
    g_alpha_ = g0_; g_cache_key_ = 0;
    df_alpha_ = slope(); df_cache_key_ = 0;
    return NotStarted;
  }

  BfgsStatus minimize_one_step(Vec6& x) {
    double alpha = 0.0, alpha1;
    const double f0 = f_;
    if (pnorm_ == 0.0 || g0norm_ == 0.0 || fp0_ == 0) {
      for (double& d : dx_.v) d = 0;
      return NoProgress;
    }
    if (delta_f_ < 0) {
      const double del = std::max(-delta_f_, 10 * std::numeric_limits<double>::epsilon() * std::fabs(f0));
      alpha1 = std::min(1.0, 2.0 * del / (-fp0_));
    } else {
      alpha1 = std::fabs(step_size);
    }
    const BfgsStatus status = line_search(alpha1, alpha);
    if (status != Success) return status;
    update_position(alpha, x, f_, gradient_);
    delta_f_ = f_ - f0;
    {
      Vec6 dx0, dg0;
      for (int i = 0; i < 6; i++) dx0.v[i] = x.v[i] - x0_.v[i];
      dx_ = dx0;
      for (int i = 0; i < 6; i++) dg0.v[i] = gradient_.v[i] - g0_.v[i];
      const double dxg = dx0.dot(gradient_), dgg = dg0.dot(gradient_), dxdg = dx0.dot(dg0), dgnorm = dg0.norm();
      double A, B;
      if (dxdg != 0) {
        B = dxg / dxdg;
        A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
      } else {
        B = 0;
        A = 0;
      }
      for (int i = 0; i < 6; i++) p_.v[i] = -A * dx0.v[i];
      for (int i = 0; i < 6; i++) p_.v[i] += gradient_.v[i];
      for (int i = 0; i < 6; i++) p_.v[i] += -B * dg0.v[i];
    }
    g0_ = gradient_;
    x0_ = x;
    g0norm_ = g0_.norm();
    pnorm_ = p_.norm();
    const double dir = (p_.dot(gradient_) > 0) ? -1.0 : 1.0;
    for (int i = 0; i < 6; i++) p_.v[i] *= dir / pnorm_;
    pnorm_ = p_.norm();
    fp0_ = p_.dot(g0_);
    change_direction();
    return Success;
  }

  BfgsStatus test_gradient(double epsilon) const {
    if (epsilon < 0) return NegativeGradientEpsilon;
    return gradient_.norm() < epsilon ? Success : Running;
  }

 private:
  const GICP& fn_;
  int iter_ = 0;
  double f_ = 0, delta_f_ = 0, g0norm_ = 0, pnorm_ = 0, fp0_ = 0;
  Vec6 gradient_, x0_, g0_, p_, dx_;
  Vec6 x_alpha_, g_alpha_;
  double f_alpha_ = 0, df_alpha_ = 0;
  double x_cache_key_ = 0, f_cache_key_ = 0, g_cache_key_ = 0, df_cache_key_ = 0;

  void move_to(double alpha) {
    if (alpha == x_cache_key_) return;
    for (int i = 0; i < 6; i++) x_alpha_.v[i] = x0_.v[i] + alpha * p_.v[i];
    x_cache_key_ = alpha;
  }
  double slope() const { return g_alpha_.dot(p_); }
  double apply_f(double alpha) {
    if (alpha == f_cache_key_) return f_alpha_;
    move_to(alpha);
    f_alpha_ = fn_.functor_f(x_alpha_.v);
    f_cache_key_ = alpha;
    return f_alpha_;
  }
  double apply_df(double alpha) {
    if (alpha == df_cache_key_) return df_alpha_;
    move_to(alpha);
    if (alpha != g_cache_key_) {
      fn_.functor_df(x_alpha_.v, g_alpha_.v);
      g_cache_key_ = alpha;
    }
    df_alpha_ = slope();
    df_cache_key_ = alpha;
    return df_alpha_;
  }
  void apply_fdf(double alpha, double& f, double& df) {
    if (alpha == f_cache_key_ && alpha == df_cache_key_) {
      f = f_alpha_;
      df = df_alpha_;
      return;
    }
    if (alpha == f_cache_key_ || alpha == df_cache_key_) {
      f = apply_f(alpha);
      df = apply_df(alpha);
      return;
    }
    move_to(alpha);
    fn_.functor_fdf(x_alpha_.v, f_alpha_, g_alpha_.v);
    f_cache_key_ = alpha;
    g_cache_key_ = alpha;
    df_alpha_ = slope();
    df_cache_key_ = alpha;
    f = f_alpha_;
    df = df_alpha_;
  }
  void update_position(double alpha, Vec6& x, double& f, Vec6& g) {
    double fa, dfa;
    apply_fdf(alpha, fa, dfa);
    f = fa;
    x = x_alpha_;
    g = g_alpha_;
  }
  void change_direction() {
    x_alpha_ = x0_; x_cache_key_ = 0;
    f_cache_key_ = 0; f_alpha_ = f_;
    g_alpha_ = g0_; g_cache_key_ = 0;
    df_alpha_ = slope(); df_cache_key_ = 0;
  }

  static double poly3(double c0, double c1, double c2, double c3, double y) { return c0 + y * (c1 + y * (c2 + y * c3)); }
  static void check_extremum(double c0, double c1, double c2, double c3, double x, double& xmin, double& fmin) {
    const double y = poly3(c0, c1, c2, c3, x);
    if (y < fmin) { xmin = x; fmin = y; }
  }

  double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax) const {
    double y, ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a), fmin;
    if (ymin > ymax) std::swap(ymin, ymax);
    if (order > 2 && !(fpb != fpa) && fpb != std::numeric_limits<double>::infinity()) {
      fpa = fpa * (b - a);
      fpb = fpb * (b - a);
      const double eta = 3 * (fb - fa) - 2 * fpa - fpb, xi = fpa + fpb - 2 * (fb - fa);
      const double c0 = fa, c1 = fpa, c2 = eta, c3 = xi;
      y = ymin;
      fmin = poly3(c0, c1, c2, c3, ymin);
      check_extremum(c0, c1, c2, c3, ymax, y, fmin);
      // roots of c1 + 2 c2 y + 3 c3 y^2
      const double qa = 3 * c3, qb = 2 * c2, qc = c1;
      if (qa != 0) {
        const double disc = qb * qb - 4 * qa * qc;
        if (disc >= 0) {
          const double sq = std::sqrt(disc);
          double y0 = (-qb - sq) / (2 * qa), y1 = (-qb + sq) / (2 * qa);
          if (y0 > y1) std::swap(y0, y1);
          if (y0 > ymin && y0 < ymax) check_extremum(c0, c1, c2, c3, y0, y, fmin);
          if (y1 > ymin && y1 < ymax) check_extremum(c0, c1, c2, c3, y1, y, fmin);
        }
      } else if (qb != 0) {
        const double y0 = -qc / qb;
        if (y0 > ymin && y0 < ymax) check_extremum(c0, c1, c2, c3, y0, y, fmin);
      }
    } else {
      fpa = fpa * (b - a);
      const double fl = fa + ymin * (fpa + ymin * (fb - fa - fpa));
      const double fh = fa + ymax * (fpa + ymax * (fb - fa - fpa));
      const double c = 2 * (fb - fa - fpa);
      y = ymin;
      fmin = fl;
      if (fh < fmin) { y = ymax; fmin = fh; }
      if (c > a) {
        const double z = -fpa / c;
        if (z > ymin && z < ymax) {
          const double f = fa + z * (fpa + z * (fb - fa - fpa));
          if (f < fmin) { y = z; fmin = f; }
        }
      }
    }
    return a + y * (b - a);
  }

  BfgsStatus line_search(double alpha1, double& alpha_new) {
    double f0, fp0, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
    double alpha = alpha1, alpha_prev = 0.0;
    double a, b, fa, fb, fpa, fpb;
    int i = 0;
    apply_fdf(0.0, f0, fp0);
    falpha_prev = f0;
    fpalpha_prev = fp0;
    a = 0.0; b = alpha;
    fa = f0; fb = 0.0;
    fpa = fp0; fpb = 0.0;
    while (i++ < bracket_iters) {  // bracketing
      falpha = apply_f(alpha);
      if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
        a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
        b = alpha; fb = falpha; fpb = std::numeric_limits<double>::quiet_NaN();
        break;
      }
      fpalpha = apply_df(alpha);
      if (std::fabs(fpalpha) <= -sigma * fp0) {
        alpha_new = alpha;
        return Success;
      }
      if (fpalpha >= 0) {
        a = alpha; fa = falpha; fpa = fpalpha;
        b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
        break;
      }
      delta = alpha - alpha_prev;
      alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta);
      alpha_prev = alpha;
      falpha_prev = falpha;
      fpalpha_prev = fpalpha;
      alpha = alpha_next;
    }
    while (i++ < section_iters) {  // sectioning
      delta = b - a;
      alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta);
      falpha = apply_f(alpha);
      if ((a - alpha) * fpa <= std::numeric_limits<double>::epsilon()) return NoProgress;
      if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
        b = alpha; fb = falpha; fpb = std::numeric_limits<double>::quiet_NaN();
      } else {
        fpalpha = apply_df(alpha);
        if (std::fabs(fpalpha) <= -sigma * fp0) {
          alpha_new = alpha;
          return Success;
        }
        if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
          b = a; fb = fa; fpb = fpa;
          a = alpha; fa = falpha; fpa = fpalpha;
        } else {
          a = alpha; fa = falpha; fpa = fpalpha;
        }
      }
    }
    return Success;
  }
};

}  // namespace

// estimateRigidTransformationBFGS, :181-238
bool GICP::estimate_bfgs(float transformation[4][4]) {
  if (corr_src.size() < 4) return false;  // NotEnoughPointsException
  Vec6 x;
  x.v[0] = transformation[0][3];
  x.v[1] = transformation[1][3];
  x.v[2] = transformation[2][3];
  x.v[3] = std::atan2(transformation[2][1], transformation[2][2]);  // float arguments: atan2f / asinf
  x.v[4] = std::asin(-transformation[2][0]);
  x.v[5] = std::atan2(transformation[1][0], transformation[0][0]);
  const double gradient_tol = 1e-2;
  Bfgs bfgs(*this);
  int inner = 0;
  int result = bfgs.minimize_init(x);
  result = Running;
  do {
    inner++;
    result = bfgs.minimize_one_step(x);
    if (result) break;
    result = bfgs.test_gradient(gradient_tol);  // PCL < 1.11 branch (:227-231)
  } while (result == Running && inner < prm.max_inner_iterations);
  if (result == NoProgress || result == Success || inner == prm.max_inner_iterations) {
    apply_state(x.v, transformation);  // setIdentity + applyState
    return true;
  }
  return false;  // SolverDidntConvergeException -- unreachable with the statuses above
}

// correspondences and Mahalanobis matrices of one outer iteration, :405-474
int GICP::correspond(const std::vector<Pt>& output, const float transformation[4][4], const float guess[4][4]) {
  const size_t N = output.size();
  if (mahalanobis.size() != N) mahalanobis.assign(N, std::array<float, 9>{1, 0, 0, 0, 1, 0, 0, 0, 1});  // :382
  double R[3][3];
  for (int i = 0; i < 3; i++)  // transform_R = transformation_ * guess in f64, :411-417
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < 4; k++) s += static_cast<double>(transformation[i][k]) * static_cast<double>(guess[k][j]);
      R[i][j] = s;
    }
  std::vector<Pt> query(N);
  for (size_t i = 0; i < N; i++) {
    float q[4];
    mat4f_vec(transformation, output[i], q);
    query[i] = Pt{q[0], q[1], q[2], q[3]};
  }
  std::vector<int> nn;
  std::vector<float> d2;
  knn_exact(target, query, 1, nn, d2);
  const double dist_threshold = prm.corr_dist_threshold * prm.corr_dist_threshold;
  corr_src.clear();
  corr_tgt.clear();
  for (size_t i = 0; i < N; i++) {
    if (nn[i] < 0) continue;
    if (static_cast<double>(d2[i]) < dist_threshold) {
      const M3& C1 = source_cov[i];
      const M3& C2 = target_cov[nn[i]];
      M3 M{}, temp{};
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) M.m[r][c] = (R[r][0] * C1.m[0][c] + R[r][1] * C1.m[1][c]) + R[r][2] * C1.m[2][c];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
          temp.m[r][c] = ((M.m[r][0] * R[c][0] + M.m[r][1] * R[c][1]) + M.m[r][2] * R[c][2]) + C2.m[r][c];
      const M3 inv = inv3(temp);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) mahalanobis[i][r * 3 + c] = static_cast<float>(inv.m[r][c]);
      corr_src.push_back(static_cast<int>(i));
      corr_tgt.push_back(nn[i]);
    }
  }
  return static_cast<int>(corr_src.size());
}

// [PCL] Registration::align pre-amble + computeTransformation, :372-517
GicpResult GICP::align(const float guess[4][4], std::vector<Pt>* output_out) {
  GicpResult res{};
  n_f = n_df = n_fdf = 0;
  std::vector<Pt> output = source;  // align(): output = *input_, data[3] = 1
  for (Pt& p : output) p.w = 1.0f;
  float transformation[4][4], previous[4][4];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) transformation[r][c] = previous[r][c] = (r == c) ? 1.0f : 0.0f;
  mahalanobis.assign(source.size(), std::array<float, 9>{1, 0, 0, 0, 1, 0, 0, 0, 1});
  bool ok = true;
  if (target_cov.empty()) ok = covariances(target, prm.k_correspondences, prm.gicp_epsilon, target_cov) && ok;
  if (source_cov.empty()) ok = covariances(source, prm.k_correspondences, prm.gicp_epsilon, source_cov) && ok;
  int nr_iterations = 0;
  bool converged = false;
  transform_cloud(output, output, guess);  // :403
  opt_src = &output;
  while (ok && !converged) {
    res.last_correspondences = correspond(output, transformation, guess);
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) previous[r][c] = transformation[r][c];
    if (!estimate_bfgs(transformation)) break;  // catch (pcl::PCLException&) { break; }
    double delta = 0.0;
    for (int k = 0; k < 4; k++)
      for (int l = 0; l < 4; l++) {
        const double ratio = (k < 3 && l < 3) ? 1.0 / prm.rotation_epsilon : 1.0 / prm.transformation_epsilon;
        const double c_delta = ratio * std::fabs(static_cast<double>(previous[k][l] - transformation[k][l]));
        if (c_delta > delta) delta = c_delta;
      }
    nr_iterations++;
    if (nr_iterations >= prm.max_iterations || delta < 1) {
      converged = true;
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) previous[r][c] = transformation[r][c];
    }
  }
  // final_transformation_ = previous_transformation_ * guess  ([Eigen] Matrix4f product, f32)
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++)
      res.final_T[r][c] = ((previous[r][0] * guess[0][c] + previous[r][1] * guess[1][c]) + previous[r][2] * guess[2][c]) +
                          previous[r][3] * guess[3][c];
  res.converged = converged;
  res.nr_iterations = nr_iterations;
  res.n_f = n_f;
  res.n_df = n_df;
  res.n_fdf = n_fdf;
  if (output_out) {
    std::vector<Pt> in = source;
    transform_cloud(in, *output_out, res.final_T);
  }
  opt_src = nullptr;
  return res;
}

}  // namespace oracle
