// ndt_driver.cpp -- host-side scalar driver of the MI355X NDT core.
// Compiled with -ffp-contract=off: the reference target (SSE4.2, no FMA,
// ndt_omp/CMakeLists.txt:10-15) never fuses, and the f32 pose->matrix
// composition below is meant to be rounding-identical to it.
#include "ndt_driver.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace ndt {

// ---------------------------------------------------------------------------
Gauss gauss_constants(float resolution, double outlier_ratio) {
  // ndt_omp_impl.hpp:86-93 (eq. 6.8 [Magnusson 2009])
  Gauss k;
  const double c1 = 10 * (1 - outlier_ratio);
  const double c2 = outlier_ratio / std::pow(static_cast<double>(resolution), 3);
  k.d3 = -std::log(c2);
  k.d1 = -std::log(c1 + c2) - k.d3;
  k.d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - k.d3) / k.d1);
  return k;
}

// ---------------------------------------------------------------------------
// 6x6 SVD solve.  One-sided Jacobi on the columns of H (A*V = U*S), then
// x = sum_j v_j (u_j.b)/s_j over singular values above Eigen's threshold
// max(s_max * 6*eps, DBL_MIN)  (SVDBase::rank / solve).
void solve6(const double H[36], const double b[6], double x[6]) {
  double a[6][6];  // a[j] = column j of the working matrix
  double v[6][6];  // v[j] = column j of V
  bool has_nan = false;
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      a[c][r] = H[r * 6 + c];
      v[c][r] = (r == c) ? 1.0 : 0.0;
      if (H[r * 6 + c] != H[r * 6 + c]) has_nan = true;
    }
  for (int k = 0; k < 6; k++)
    if (b[k] != b[k]) has_nan = true;
  if (has_nan) {
    for (int k = 0; k < 6; k++) x[k] = std::numeric_limits<double>::quiet_NaN();
    return;
  }
  // Fast path.  JacobiSVD::solve returns the minimum-norm least-squares solution, which for a matrix
  // of full numerical rank IS H^-1 b.  Gaussian elimination with complete pivoting both computes that
  // and certifies the conditioning: it is used only when every pivot is above 1e-5 of the largest, so
  // that (a) no singular value can be anywhere near Eigen's rank threshold (6 eps s_max) and (b) the
  // elimination and the SVD agree to ~1e-10 relative -- on worse-conditioned Hessians (registrations
  // started far off with DIRECT26, say) the two backward-stable answers differ by cond * eps, enough
  // to send the line search down another path than the reference's, so those take the SVD below, as
  // do rank-deficient and zero matrices.  ~0.2 us against ~5 us, once per Newton iteration on the
  // registration's serial path.
  constexpr double kPivotFloor = 1e-5;
  {
    double m[6][7];
    for (int r = 0; r < 6; r++) {
      for (int c = 0; c < 6; c++) m[r][c] = H[r * 6 + c];
      m[r][6] = b[r];
    }
    int col_of[6] = {0, 1, 2, 3, 4, 5};
    double p_max = 0.0, p_min = std::numeric_limits<double>::infinity();
    bool ok = true;
    for (int k = 0; k < 6 && ok; k++) {
      int pr = k, pc = k;
      double best = -1.0;
      for (int r = k; r < 6; r++)
        for (int c = k; c < 6; c++)
          if (std::fabs(m[r][c]) > best) {
            best = std::fabs(m[r][c]);
            pr = r;
            pc = c;
          }
      p_max = std::max(p_max, best);
      p_min = std::min(p_min, best);
      if (!(best > kPivotFloor * p_max) || !(best > std::numeric_limits<double>::min())) {
        ok = false;
        break;
      }
      if (pr != k)
        for (int c = 0; c < 7; c++) std::swap(m[pr][c], m[k][c]);
      if (pc != k) {
        for (int r = 0; r < 6; r++) std::swap(m[r][pc], m[r][k]);
        std::swap(col_of[pc], col_of[k]);
      }
      const double inv = 1.0 / m[k][k];
      for (int r = k + 1; r < 6; r++) {
        const double f = m[r][k] * inv;
        if (f == 0.0) continue;
        for (int c = k + 1; c < 7; c++) m[r][c] -= f * m[k][c];
      }
    }
    if (ok && p_min > kPivotFloor * p_max) {
      double y[6];
      for (int k = 5; k >= 0; k--) {
        double acc = m[k][6];
        for (int c = k + 1; c < 6; c++) acc -= m[k][c] * y[c];
        y[k] = acc / m[k][k];
      }
      for (int k = 0; k < 6; k++) x[col_of[k]] = y[k];
      return;
    }
  }
  const double tol = 4.0 * std::numeric_limits<double>::epsilon();
  for (int sweep = 0; sweep < 64; sweep++) {
    int n_rot = 0;
    for (int i = 0; i < 5; i++)
      for (int j = i + 1; j < 6; j++) {
        double aii = 0, ajj = 0, aij = 0;
        for (int k = 0; k < 6; k++) {
          aii += a[i][k] * a[i][k];
          ajj += a[j][k] * a[j][k];
          aij += a[i][k] * a[j][k];
        }
        if (aij == 0.0 || std::fabs(aij) <= tol * 0.0625 * std::sqrt(aii) * std::sqrt(ajj)) continue;
        n_rot++;
        const double tau = (ajj - aii) / (2.0 * aij);
        const double t = std::copysign(1.0, tau) / (std::fabs(tau) + std::hypot(1.0, tau));
        const double cs = 1.0 / std::hypot(1.0, t), sn = cs * t;
        for (int k = 0; k < 6; k++) {
          const double ai = a[i][k], aj = a[j][k];
          a[i][k] = cs * ai - sn * aj;
          a[j][k] = sn * ai + cs * aj;
          const double vi = v[i][k], vj = v[j][k];
          v[i][k] = cs * vi - sn * vj;
          v[j][k] = sn * vi + cs * vj;
        }
      }
    if (n_rot == 0) break;
  }
  double s2[6], s_max = 0;
  for (int j = 0; j < 6; j++) {
    s2[j] = 0;
    for (int k = 0; k < 6; k++) s2[j] += a[j][k] * a[j][k];
    s_max = std::max(s_max, std::sqrt(s2[j]));
  }
  const double thr = std::max(s_max * 6.0 * std::numeric_limits<double>::epsilon(), std::numeric_limits<double>::min());
  for (int k = 0; k < 6; k++) x[k] = 0.0;
  for (int j = 0; j < 6; j++) {
    if (!(std::sqrt(s2[j]) >= thr)) continue;
    double ab = 0;
    for (int k = 0; k < 6; k++) ab += a[j][k] * b[k];
    const double w = ab / s2[j];
    for (int k = 0; k < 6; k++) x[k] += v[j][k] * w;
  }
}

// ---------------------------------------------------------------------------
namespace {
struct Mat3f {
  float m[3][3];
};

// Eigen::AngleAxisf(angle, Unit{X,Y,Z}).toRotationMatrix(), all terms kept
// (the zero products are exact, the diagonal (1-c)*1*1 + c is not always 1).
Mat3f unit_axis_rotation(int axis, float angle) {
  float u[3] = {0.f, 0.f, 0.f};
  u[axis] = 1.f;
  const float s = std::sin(angle), c = std::cos(angle), omc = 1.0f - c;
  const float su[3] = {s * u[0], s * u[1], s * u[2]};
  const float cu[3] = {omc * u[0], omc * u[1], omc * u[2]};
  Mat3f r;
  float t = cu[0] * u[1];
  r.m[0][1] = t - su[2];
  r.m[1][0] = t + su[2];
  t = cu[0] * u[2];
  r.m[0][2] = t + su[1];
  r.m[2][0] = t - su[1];
  t = cu[1] * u[2];
  r.m[1][2] = t - su[0];
  r.m[2][1] = t + su[0];
  r.m[0][0] = cu[0] * u[0] + c;
  r.m[1][1] = cu[1] * u[1] + c;
  r.m[2][2] = cu[2] * u[2] + c;
  return r;
}

Mat3f mul(const Mat3f& a, const Mat3f& b) {
  Mat3f r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
  return r;
}
}  // namespace

void pose_to_matrix(const double p[6], float T[16]) {
  const Mat3f R = mul(mul(unit_axis_rotation(0, static_cast<float>(p[3])), unit_axis_rotation(1, static_cast<float>(p[4]))),
                      unit_axis_rotation(2, static_cast<float>(p[5])));
  for (int c = 0; c < 3; c++) {
    for (int r = 0; r < 3; r++) T[c * 4 + r] = R.m[r][c];
    T[c * 4 + 3] = 0.0f;
  }
  T[12] = static_cast<float>(p[0]);
  T[13] = static_cast<float>(p[1]);
  T[14] = static_cast<float>(p[2]);
  T[15] = 1.0f;
}

// Transform<float,3,Affine>::rotation() is the polar factor U*V^T of the linear
// part (Eigen Transform.h computeRotationScaling, f32 JacobiSVD); then
// MatrixBase::eulerAngles(0,1,2) (Eigen 3.3.7 EulerAngles.h).
void matrix_to_pose(const float T[16], double p[6]) {
  // rotation(): the orthogonal polar factor of the linear part.  Eigen gets it from an f32 JacobiSVD
  // (U V^T), i.e. the exact factor plus a few ulps of that solver's own rounding noise, which cannot be
  // restated without Eigen.  The neutral choice is the correctly rounded factor: Newton's polar
  // iteration X <- (X + X^-T) / 2 in f64 (quadratically convergent, exact fixed point for an
  // orthonormal input such as Identity), rounded to f32.
  double X[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) X[r][c] = T[c * 4 + r];
  for (int it = 0; it < 30; it++) {
    const double c00 = X[1][1] * X[2][2] - X[1][2] * X[2][1], c01 = X[1][2] * X[2][0] - X[1][0] * X[2][2],
                 c02 = X[1][0] * X[2][1] - X[1][1] * X[2][0];
    const double det = (X[0][0] * c00 + X[0][1] * c01) + X[0][2] * c02;
    if (!(std::fabs(det) > 0.0)) break;  // singular linear part: leave it as it is
    const double inv = 1.0 / det;
    // cofactor matrix / det = X^-T
    const double XiT[3][3] = {
        {c00 * inv, c01 * inv, c02 * inv},
        {(X[0][2] * X[2][1] - X[0][1] * X[2][2]) * inv, (X[0][0] * X[2][2] - X[0][2] * X[2][0]) * inv, (X[0][1] * X[2][0] - X[0][0] * X[2][1]) * inv},
        {(X[0][1] * X[1][2] - X[0][2] * X[1][1]) * inv, (X[0][2] * X[1][0] - X[0][0] * X[1][2]) * inv, (X[0][0] * X[1][1] - X[0][1] * X[1][0]) * inv}};
    double diff = 0.0;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        const double y = 0.5 * (X[r][c] + XiT[r][c]);
        diff = std::max(diff, std::fabs(y - X[r][c]));
        X[r][c] = y;
      }
    if (diff < 1e-15) break;
  }
  float R[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[r][c] = static_cast<float>(X[r][c]);
  // eulerAngles(0,1,2): odd = 0, i = 0, j = 1, k = 2
  const float pi_f = static_cast<float>(3.141592653589793238462643383279502884L);
  float e0 = std::atan2(R[1][2], R[2][2]), e1;
  const float c2 = std::sqrt(R[0][0] * R[0][0] + R[0][1] * R[0][1]);
  if (e0 > 0.0f) {
    e0 -= pi_f;
    e1 = std::atan2(-R[0][2], -c2);
  } else {
    e1 = std::atan2(-R[0][2], c2);
  }
  const float s1 = std::sin(e0), c1 = std::cos(e0);
  const float e2 = std::atan2(s1 * R[2][0] - c1 * R[1][0], c1 * R[1][1] - s1 * R[2][1]);
  p[0] = T[12];
  p[1] = T[13];
  p[2] = T[14];
  p[3] = -e0;
  p[4] = -e1;
  p[5] = -e2;
}

// ---------------------------------------------------------------------------
void chain_pose(const float a[16], const float b[16], float out[16]) {
  float r[16];
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++)
      r[j * 4 + i] = ((a[0 * 4 + i] * b[j * 4 + 0] + a[1 * 4 + i] * b[j * 4 + 1]) + a[2 * 4 + i] * b[j * 4 + 2]) + a[3 * 4 + i] * b[j * 4 + 3];
  std::memcpy(out, r, sizeof(r));
}

void snapped_cos_sin(const double p[6], double cs[6]) {
  for (int k = 0; k < 3; k++) {
    if (std::fabs(p[3 + k]) < 10e-5) {
      cs[k] = 1.0;
      cs[3 + k] = 0.0;
    } else {
      cs[k] = std::cos(p[3 + k]);
      cs[3 + k] = std::sin(p[3 + k]);
    }
  }
}

void angle_derivatives(const double p[6], AngleDerivs& o) {
  // ndt_omp_impl.hpp:292-326: |angle| < 10e-5 snaps to cos = 1, sin = 0
  double c[3], s[3];
  for (int k = 0; k < 3; k++) {
    if (std::fabs(p[3 + k]) < 10e-5) {
      c[k] = 1.0;
      s[k] = 0.0;
    } else {
      c[k] = std::cos(p[3 + k]);
      s[k] = std::sin(p[3 + k]);
    }
  }
  const double cx = c[0], cy = c[1], cz = c[2], sx = s[0], sy = s[1], sz = s[2];
  // :329-346
  const double J[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy},
                          {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},
                          {-sy * cz, sy * sz, cy},
                          {sx * cy * cz, -sx * cy * sz, sx * sy},
                          {-cx * cy * cz, cx * cy * sz, -cx * sy},
                          {-cy * sz, -cy * cz, 0},
                          {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0},
                          {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};
  // :351-393, rows a2 a3 b2 b3 c2 c3 d1 d2 d3 e1 e2 e3 f1 f2 f3
  const double Hm[15][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy},
                            {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},
                            {cx * cy * cz, -cx * cy * sz, cx * sy},
                            {sx * cy * cz, -sx * cy * sz, sx * sy},
                            {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0},
                            {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0},
                            {-cy * cz, cy * sz, -sy},
                            {-sx * sy * cz, sx * sy * sz, sx * cy},
                            {cx * sy * cz, -cx * sy * sz, -cx * cy},
                            {sy * sz, sy * cz, 0},
                            {-sx * cy * sz, -sx * cy * cz, 0},
                            {cx * cy * sz, cx * cy * cz, 0},
                            {-cy * cz, cy * sz, 0},
                            {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0},
                            {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0}};
  for (int r = 0; r < 8; r++)
    for (int k = 0; k < 3; k++) {
      o.jd[r][k] = J[r][k];
      o.j[r][k] = static_cast<float>(J[r][k]);
    }
  for (int r = 0; r < 15; r++)
    for (int k = 0; k < 3; k++) {
      o.hd[r][k] = Hm[r][k];
      o.h[r][k] = static_cast<float>(Hm[r][k]);
    }
  // the f32 matrix stores +sy in d1 (ndt_omp_impl.hpp:383) while the f64 vector
  // has -sy (:361): reproduced, not fixed.
  o.h[6][2] = static_cast<float>(sy);
}

// ---------------------------------------------------------------------------
// More-Thuente helpers (ndt_omp_impl.hpp:648-769; ndt_omp.h:430-447)
namespace {
inline double aux_psi(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
inline double aux_dpsi(double g_a, double g_0, double mu) { return g_a - mu * g_0; }
constexpr double kMu = 1.e-4, kNu = 0.9;
constexpr int kMaxStepIterations = 10;

struct Bracket {
  double a, f, g;
};

// cubic minimiser through (lo, hi) -- Sun & Yuan eq. 2.4.52/2.4.56
inline double cubic_min(const Bracket& lo, double a_t, double f_t, double g_t) {
  const double z = 3 * (f_t - lo.f) / (a_t - lo.a) - g_t - lo.g;
  const double w = std::sqrt(z * z - g_t * lo.g);
  return lo.a + (a_t - lo.a) * (w - lo.g - z) / (g_t - lo.g + 2 * w);
}

double select_trial(const Bracket& l, const Bracket& u, double a_t, double f_t, double g_t) {
  if (f_t > l.f) {  // case 1
    const double a_c = cubic_min(l, a_t, f_t, g_t);
    const double a_q = l.a - 0.5 * (l.a - a_t) * l.g / (l.g - (l.f - f_t) / (l.a - a_t));
    return (std::fabs(a_c - l.a) < std::fabs(a_q - l.a)) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * l.g < 0) {  // case 2
    const double a_c = cubic_min(l, a_t, f_t, g_t);
    const double a_s = l.a - (l.a - a_t) / (l.g - g_t) * l.g;
    return (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) ? a_c : a_s;
  }
  if (std::fabs(g_t) <= std::fabs(l.g)) {  // case 3
    const double a_c = cubic_min(l, a_t, f_t, g_t);
    const double a_s = l.a - (l.a - a_t) / (l.g - g_t) * l.g;
    const double next = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
    const double lim = a_t + 0.66 * (u.a - a_t);
    return (a_t > l.a) ? std::min(lim, next) : std::max(lim, next);
  }
  return cubic_min(u, a_t, f_t, g_t);  // case 4
}

// returns true when the interval has converged
bool update_interval(Bracket& l, Bracket& u, double a_t, double f_t, double g_t) {
  if (f_t > l.f) {
    u = {a_t, f_t, g_t};
    return false;
  }
  const double sgn = g_t * (l.a - a_t);
  if (sgn > 0) {
    l = {a_t, f_t, g_t};
    return false;
  }
  if (sgn < 0) {
    u = l;
    l = {a_t, f_t, g_t};
    return false;
  }
  return true;
}

bool is_identity16(const float* m) {
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++)
      if (m[c * 4 + r] != ((r == c) ? 1.0f : 0.0f)) return false;
  return true;
}
}  // namespace

// ---------------------------------------------------------------------------
void ScanSolver::start(const float* guess, size_t n_source, const SolverParams& prm) {
  prm_ = prm;
  n_source_ = n_source;
  nr_iterations = 0;
  converged = false;
  trans_probability = 0;
  n_evals = n_hess = 0;
  // pcl::Registration::align resets final_transformation_ to Identity; a
  // non-identity guess replaces it (ndt_omp_impl.hpp:95-101)
  for (int i = 0; i < 16; i++) final_T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  if (guess && !is_identity16(guess)) std::memcpy(final_T, guess, sizeof(final_T));
  matrix_to_pose(final_T, p_);  // :103-111
  req_.kind = EVAL_WITH_HESSIAN;  // :119
  std::memcpy(req_.T, final_T, sizeof(final_T));
  std::memcpy(req_.p, p_, sizeof(p_));
  state_ = S_INIT;
}

void ScanSolver::finish(bool conv) {
  converged = conv;
  trans_probability = score_ / static_cast<double>(n_source_);  // :136, :170
  req_.kind = EVAL_NONE;
  state_ = S_DONE;
}

void ScanSolver::issue_trial(EvalKind kind) {
  for (int i = 0; i < 6; i++) x_t_[i] = x_[i] + dir_[i] * a_t_;
  pose_to_matrix(x_t_, final_T);  // :827-830 / :871-874
  req_.kind = kind;
  std::memcpy(req_.T, final_T, sizeof(final_T));
  std::memcpy(req_.p, x_t_, sizeof(x_t_));
}

void ScanSolver::newton_top() {
  for (;;) {
    double neg_g[6], delta[6];
    for (int i = 0; i < 6; i++) neg_g[i] = -g_[i];
    solve6(H_, neg_g, delta);  // :127-129
    double n2 = 0;
    for (int i = 0; i < 6; i++) n2 += delta[i] * delta[i];
    const double norm = std::sqrt(n2);
    if (norm == 0 || norm != norm) {  // :134-139
      finish(norm == norm);
      return;
    }
    for (int i = 0; i < 6; i++) dir_[i] = delta[i] / norm;
    // ---- computeStepLengthMT prologue, :777-837
    std::memcpy(x_, p_, sizeof(p_));
    step_max_ = prm_.step_size;
    step_min_ = prm_.trans_eps / 2;
    phi_0_ = -score_;
    double gd = 0;
    for (int i = 0; i < 6; i++) gd += g_[i] * dir_[i];
    d_phi_0_ = -gd;
    if (d_phi_0_ >= 0) {
      if (d_phi_0_ == 0) {  // "return 0": zero step, no evaluation
        a_t_ = 0;
        if (nr_iterations > prm_.max_iter || (nr_iterations && (std::fabs(a_t_) < prm_.trans_eps))) {
          nr_iterations++;
          finish(true);
          return;
        }
        nr_iterations++;
        continue;
      }
      d_phi_0_ = -d_phi_0_;
      for (int i = 0; i < 6; i++) dir_[i] = -dir_[i];
    }
    step_iterations_ = 0;
    a_l_ = a_u_ = 0;
    f_l_ = f_u_ = aux_psi(0, phi_0_, phi_0_, d_phi_0_, kMu);
    g_l_ = g_u_ = aux_dpsi(d_phi_0_, d_phi_0_, kMu);
    interval_converged_ = (step_max_ - step_min_) < 0;
    open_interval_ = true;
    a_t_ = std::max(std::min(norm, step_max_), step_min_);
    issue_trial(EVAL_WITH_HESSIAN);
    state_ = S_MT_FIRST;
    return;
  }
}

void ScanSolver::mt_check() {
  // loop condition of :850
  if (!interval_converged_ && step_iterations_ < kMaxStepIterations &&
      !(psi_t_ <= 0 && d_phi_t_ <= -kNu * d_phi_0_)) {
    Bracket l{a_l_, f_l_, g_l_}, u{a_u_, f_u_, g_u_};
    a_t_ = open_interval_ ? select_trial(l, u, a_t_, psi_t_, d_psi_t_) : select_trial(l, u, a_t_, phi_t_, d_phi_t_);
    a_t_ = std::max(std::min(a_t_, step_max_), step_min_);
    issue_trial(EVAL_NO_HESSIAN);  // :881
    state_ = S_MT_LOOP;
    return;
  }
  if (step_iterations_) {  // :928-929
    req_.kind = EVAL_HESSIAN_F64;
    std::memcpy(req_.T, final_T, sizeof(final_T));
    std::memcpy(req_.p, x_t_, sizeof(x_t_));
    state_ = S_MT_HESS;
    return;
  }
  mt_finish();
}

void ScanSolver::mt_finish() {
  // back in computeTransformation, :143-164
  for (int i = 0; i < 6; i++) p_[i] = p_[i] + dir_[i] * a_t_;
  const bool stop = nr_iterations > prm_.max_iter || (nr_iterations && (std::fabs(a_t_) < prm_.trans_eps));
  nr_iterations++;
  if (stop) {
    finish(true);
    return;
  }
  newton_top();
}

void ScanSolver::feed(const EvalResult& r) {
  switch (state_) {
    case S_INIT:
      n_evals++;
      score_ = r.score;
      std::memcpy(g_, r.g, sizeof(g_));
      std::memcpy(H_, r.H, sizeof(H_));
      newton_top();
      break;
    case S_MT_FIRST:
    case S_MT_LOOP: {
      n_evals++;
      score_ = r.score;
      std::memcpy(g_, r.g, sizeof(g_));
      if (state_ == S_MT_FIRST)
        std::memcpy(H_, r.H, sizeof(H_));
      else
        std::memset(H_, 0, sizeof(H_));  // compute_hessian=false leaves hessian.setZero() (:187)
      phi_t_ = -score_;
      double gd = 0;
      for (int i = 0; i < 6; i++) gd += g_[i] * dir_[i];
      d_phi_t_ = -gd;
      psi_t_ = aux_psi(a_t_, phi_t_, phi_0_, d_phi_0_, kMu);
      d_psi_t_ = aux_dpsi(d_phi_t_, d_phi_0_, kMu);
      if (state_ == S_MT_LOOP) {
        if (open_interval_ && (psi_t_ <= 0 && d_psi_t_ >= 0)) {  // :894-905
          open_interval_ = false;
          f_l_ = f_l_ + phi_0_ - kMu * d_phi_0_ * a_l_;
          g_l_ = g_l_ + kMu * d_phi_0_;
          f_u_ = f_u_ + phi_0_ - kMu * d_phi_0_ * a_u_;
          g_u_ = g_u_ + kMu * d_phi_0_;
        }
        Bracket l{a_l_, f_l_, g_l_}, u{a_u_, f_u_, g_u_};
        interval_converged_ = open_interval_ ? update_interval(l, u, a_t_, psi_t_, d_psi_t_)
                                             : update_interval(l, u, a_t_, phi_t_, d_phi_t_);
        a_l_ = l.a; f_l_ = l.f; g_l_ = l.g;
        a_u_ = u.a; f_u_ = u.f; g_u_ = u.g;
        step_iterations_++;
      }
      mt_check();
      break;
    }
    case S_MT_HESS:
      n_hess++;
      std::memcpy(H_, r.H, sizeof(H_));
      mt_finish();
      break;
    case S_DONE:
      break;
  }
}

}  // namespace ndt
