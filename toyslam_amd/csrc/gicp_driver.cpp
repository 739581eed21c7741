// gicp_driver.cpp -- see gicp_driver.hpp.  Built with -ffp-contract=off like ndt_driver.cpp: the
// reference target (x86-64, SSE4.2) has no fused multiply-add.
#include "gicp_driver.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace gicp {

namespace {

constexpr int kDim = 6;
using Vec = double[kDim];

inline double dot6(const double* a, const double* b) {
  double s = 0.0;
  for (int i = 0; i < kDim; i++) s += a[i] * b[i];
  return s;
}
inline double norm6(const double* a) { return std::sqrt(dot6(a, a)); }
inline void copy6(double* d, const double* s) {
  for (int i = 0; i < kDim; i++) d[i] = s[i];
}

// f32 unit quaternion of a rotation about one coordinate axis ([Eigen] Quaternion = AngleAxis)
struct Quat {
  float w, x, y, z;
};
Quat axis_quat(int axis, float angle) {
  const float half = 0.5f * angle;
  const float c = std::cos(half), s = std::sin(half);
  Quat q{c, 0.f, 0.f, 0.f};
  if (axis == 0) q.x = s;
  else if (axis == 1) q.y = s;
  else q.z = s;
  return q;
}
Quat operator*(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

// The objective as BFGS sees it: value, gradient, and their bookkeeping of device evaluations.
class Objective {
 public:
  Objective(Backend& dev, Result& stats) : dev_(dev), stats_(stats) {}
  bool ok() const { return ok_; }
  double count() const { return m_; }

  double value(const double* x) {  // operator(), gicp_omp_impl.hpp:241-274
    FunctorSums s;
    fetch(0, x, s);
    stats_.n_f++;
    return s.f / static_cast<double>(static_cast<int>(s.m));
  }
  void gradient(const double* x, double* g) {  // df, :277-331
    FunctorSums s;
    fetch(1, x, s);
    stats_.n_df++;
    finish_gradient(x, s, g);
  }
  void both(const double* x, double& f, double* g) {  // fdf, :334-368
    FunctorSums s;
    fetch(2, x, s);
    stats_.n_fdf++;
    f = s.f / static_cast<double>(static_cast<int>(s.m));
    finish_gradient(x, s, g);
  }

 private:
  Backend& dev_;
  Result& stats_;
  bool ok_ = true;
  double m_ = 0;

  void fetch(int mode, const double* x, FunctorSums& s) {
    float T[16];
    apply_state(x, T);
    if (!dev_.sums(mode, T, s)) {
      ok_ = false;
      s = FunctorSums{};
    }
    m_ = s.m;
  }
  static void finish_gradient(const double* x, const FunctorSums& s, double* g) {
    const int m = static_cast<int>(s.m);
    const double scale = 2.0 / m;
    for (int i = 0; i < 3; i++) g[i] = s.g[i] * scale;
    double R[9];
    for (int i = 0; i < 9; i++) R[i] = s.R[i] * scale;
    rotation_gradient(x, R, g);
  }
};

// [PCL 1.10] BFGS (registration/bfgs.h; GSL vector_bfgs2 + Fletcher's line search) for six
// parameters.  PCL's own conditions are kept where they differ from GSL: the cubic interpolation
// needs fpb == fpa (bfgs.h: `!(fpb != fpa)`), the quadratic one needs c > a.
enum Status { kNegativeGradientEpsilon = -3, kNotStarted = -2, kRunning = -1, kSuccess = 0, kNoProgress = 1 };

class Minimizer {
 public:
  // estimateRigidTransformationBFGS parameters (:212-217) over PCL's defaults
  double sigma = 0.01, rho = 0.01, tau1 = 9.0, tau2 = 0.05, tau3 = 0.5, step_size = 1.0;
  int order = 3, bracket_iters = 100, section_iters = 100;

  explicit Minimizer(Objective& fn) : fn_(fn) {}

  void init(double* x) {
    delta_f_ = 0.0;
    fn_.both(x, f_, grad_);
    copy6(x0_, x);
    copy6(g0_, grad_);
    g0norm_ = norm6(g0_);
    for (int i = 0; i < kDim; i++) p_[i] = grad_[i] * -1 / g0norm_;
    pnorm_ = norm6(p_);
    fp0_ = -g0norm_;
    reset_line();
  }

  Status step(double* x) {
    const double f0 = f_;
    if (pnorm_ == 0.0 || g0norm_ == 0.0 || fp0_ == 0) return kNoProgress;
    double alpha1;
    if (delta_f_ < 0) {
      const double del = std::max(-delta_f_, 10 * std::numeric_limits<double>::epsilon() * std::fabs(f0));
      alpha1 = std::min(1.0, 2.0 * del / (-fp0_));
    } else {
      alpha1 = std::fabs(step_size);
    }
    double alpha = 0.0;
    const Status st = line_search(alpha1, alpha);
    if (st != kSuccess) return st;
    // updatePosition
    double fa, dfa;
    eval_fdf(alpha, fa, dfa);
    f_ = fa;
    copy6(x, xa_);
    copy6(grad_, ga_);
    delta_f_ = f_ - f0;
    // memoryless BFGS direction: p' = g1 - A dx - B dg
    Vec dx, dg;
    for (int i = 0; i < kDim; i++) dx[i] = x[i] - x0_[i];
    for (int i = 0; i < kDim; i++) dg[i] = grad_[i] - g0_[i];
    const double dxg = dot6(dx, grad_), dgg = dot6(dg, grad_), dxdg = dot6(dx, dg), dgnorm = norm6(dg);
    double A = 0, B = 0;
    if (dxdg != 0) {
      B = dxg / dxdg;
      A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
    }
    for (int i = 0; i < kDim; i++) p_[i] = -A * dx[i];
    for (int i = 0; i < kDim; i++) p_[i] += grad_[i];
    for (int i = 0; i < kDim; i++) p_[i] += -B * dg[i];
    copy6(g0_, grad_);
    copy6(x0_, x);
    g0norm_ = norm6(g0_);
    pnorm_ = norm6(p_);
    const double dir = (dot6(p_, grad_) > 0) ? -1.0 : 1.0;
    for (int i = 0; i < kDim; i++) p_[i] *= dir / pnorm_;
    pnorm_ = norm6(p_);
    fp0_ = dot6(p_, g0_);
    reset_line();
    return kSuccess;
  }

  Status test_gradient(double epsilon) const {
    if (epsilon < 0) return kNegativeGradientEpsilon;
    return norm6(grad_) < epsilon ? kSuccess : kRunning;
  }

 private:
  Objective& fn_;
  double f_ = 0, delta_f_ = 0, g0norm_ = 0, pnorm_ = 0, fp0_ = 0;
  Vec grad_, x0_, g0_, p_;
  // the objective along x0 + alpha p, with GSL's one-entry caches
  Vec xa_, ga_;
  double fa_ = 0, dfa_ = 0, key_x_ = 0, key_f_ = 0, key_g_ = 0, key_df_ = 0;

  void reset_line() {  // changeDirection
    copy6(xa_, x0_);
    key_x_ = 0;
    fa_ = f_;
    key_f_ = 0;
    copy6(ga_, g0_);
    key_g_ = 0;
    dfa_ = dot6(ga_, p_);
    key_df_ = 0;
  }
  void move_to(double alpha) {
    if (alpha == key_x_) return;
    for (int i = 0; i < kDim; i++) xa_[i] = x0_[i] + alpha * p_[i];
    key_x_ = alpha;
  }
  double eval_f(double alpha) {
    if (alpha == key_f_) return fa_;
    move_to(alpha);
    fa_ = fn_.value(xa_);
    key_f_ = alpha;
    return fa_;
  }
  double eval_df(double alpha) {
    if (alpha == key_df_) return dfa_;
    move_to(alpha);
    if (alpha != key_g_) {
      fn_.gradient(xa_, ga_);
      key_g_ = alpha;
    }
    dfa_ = dot6(ga_, p_);
    key_df_ = alpha;
    return dfa_;
  }
  void eval_fdf(double alpha, double& f, double& df) {
    if (alpha == key_f_ && alpha == key_df_) {
      f = fa_;
      df = dfa_;
      return;
    }
    if (alpha == key_f_ || alpha == key_df_) {
      f = eval_f(alpha);
      df = eval_df(alpha);
      return;
    }
    move_to(alpha);
    fn_.both(xa_, fa_, ga_);
    key_f_ = alpha;
    key_g_ = alpha;
    dfa_ = dot6(ga_, p_);
    key_df_ = alpha;
    f = fa_;
    df = dfa_;
  }

  struct Cubic {
    double c0, c1, c2, c3;
    double at(double y) const { return c0 + y * (c1 + y * (c2 + y * c3)); }
    void lower(double y, double& ybest, double& fbest) const {
      const double v = at(y);
      if (v < fbest) {
        ybest = y;
        fbest = v;
      }
    }
  };

  // minimiser of the interpolant through (a, fa, fpa), (b, fb[, fpb]) within [xmin, xmax]
  double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax) const {
    double ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
    if (ymin > ymax) std::swap(ymin, ymax);
    double y, fmin;
    if (order > 2 && !(fpb != fpa) && fpb != std::numeric_limits<double>::infinity()) {
      fpa = fpa * (b - a);
      fpb = fpb * (b - a);
      const Cubic c{fa, fpa, 3 * (fb - fa) - 2 * fpa - fpb, fpa + fpb - 2 * (fb - fa)};
      y = ymin;
      fmin = c.at(ymin);
      c.lower(ymax, y, fmin);
      // stationary points: c1 + 2 c2 y + 3 c3 y^2 = 0
      const double q2 = 3 * c.c3, q1 = 2 * c.c2, q0 = c.c1;
      if (q2 != 0) {
        const double disc = q1 * q1 - 4 * q2 * q0;
        if (disc >= 0) {
          const double root = std::sqrt(disc);
          double ya = (-q1 - root) / (2 * q2), yb = (-q1 + root) / (2 * q2);
          if (ya > yb) std::swap(ya, yb);
          if (ya > ymin && ya < ymax) c.lower(ya, y, fmin);
          if (yb > ymin && yb < ymax) c.lower(yb, y, fmin);
        }
      } else if (q1 != 0) {
        const double ya = -q0 / q1;
        if (ya > ymin && ya < ymax) c.lower(ya, y, fmin);
      }
    } else {
      fpa = fpa * (b - a);
      const double curv = fb - fa - fpa;
      const double fl = fa + ymin * (fpa + ymin * curv);
      const double fh = fa + ymax * (fpa + ymax * curv);
      const double c = 2 * curv;
      y = ymin;
      fmin = fl;
      if (fh < fmin) {
        y = ymax;
        fmin = fh;
      }
      if (c > a) {
        const double z = -fpa / c;
        if (z > ymin && z < ymax) {
          const double fz = fa + z * (fpa + z * curv);
          if (fz < fmin) {
            y = z;
            fmin = fz;
          }
        }
      }
    }
    return a + y * (b - a);
  }

  Status line_search(double alpha1, double& alpha_new) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    double f0, fp0;
    eval_fdf(0.0, f0, fp0);
    double alpha = alpha1, alpha_prev = 0.0;
    double falpha, falpha_prev = f0, fpalpha, fpalpha_prev = fp0;
    double a = 0.0, b = alpha, fa = f0, fb = 0.0, fpa = fp0, fpb = 0.0;
    int i = 0;
    while (i++ < bracket_iters) {
      falpha = eval_f(alpha);
      if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {  // Fletcher's rho test
        a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
        b = alpha; fb = falpha; fpb = nan;
        break;
      }
      fpalpha = eval_df(alpha);
      if (std::fabs(fpalpha) <= -sigma * fp0) {  // sigma test
        alpha_new = alpha;
        return kSuccess;
      }
      if (fpalpha >= 0) {
        a = alpha; fa = falpha; fpa = fpalpha;
        b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
        break;
      }
      const double delta = alpha - alpha_prev;
      const double next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta,
                                      alpha + tau1 * delta);
      alpha_prev = alpha;
      falpha_prev = falpha;
      fpalpha_prev = fpalpha;
      alpha = next;
    }
    while (i++ < section_iters) {
      const double delta = b - a;
      alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta);
      falpha = eval_f(alpha);
      if ((a - alpha) * fpa <= std::numeric_limits<double>::epsilon()) return kNoProgress;  // roundoff prevents progress
      if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
        b = alpha; fb = falpha; fpb = nan;
      } else {
        fpalpha = eval_df(alpha);
        if (std::fabs(fpalpha) <= -sigma * fp0) {
          alpha_new = alpha;
          return kSuccess;
        }
        if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
          b = a; fb = fa; fpb = fpa;
        }
        a = alpha; fa = falpha; fpa = fpalpha;
      }
    }
    return kSuccess;
  }
};

// estimateRigidTransformationBFGS, :181-238.  false = one of its exceptions (fewer than 4
// correspondences / solver did not converge) or a device failure.
bool estimate(const Params& prm, Backend& dev, Result& stats, float transformation[16]) {
  double x[kDim];
  x[0] = transformation[3];
  x[1] = transformation[7];
  x[2] = transformation[11];
  x[3] = std::atan2(transformation[9], transformation[10]);  // f32 atan2 / asin of the f32 matrix
  x[4] = std::asin(-transformation[8]);
  x[5] = std::atan2(transformation[4], transformation[0]);
  Objective fn(dev, stats);
  Minimizer bfgs(fn);
  bfgs.init(x);
  if (!fn.ok()) return false;
  stats.correspondences = static_cast<int>(fn.count());
  if (stats.correspondences < 4) {  // NotEnoughPointsException: thrown before the reference evaluates anything
    stats.n_fdf--;
    return false;
  }
  const double gradient_tol = 1e-2;
  int inner = 0;
  int result = kRunning;
  do {
    inner++;
    result = bfgs.step(x);
    if (!fn.ok()) return false;
    if (result) break;
    result = bfgs.test_gradient(gradient_tol);  // the PCL < 1.11 branch (:227-231)
  } while (result == kRunning && inner < prm.max_inner_iterations);
  if (result == kNoProgress || result == kSuccess || inner == prm.max_inner_iterations) {
    apply_state(x, transformation);
    return true;
  }
  return false;
}

}  // namespace

void apply_state(const double x[6], float T[16]) {
  const Quat q = axis_quat(2, static_cast<float>(x[5])) * axis_quat(1, static_cast<float>(x[4])) *
                 axis_quat(0, static_cast<float>(x[3]));
  // [Eigen] QuaternionBase::toRotationMatrix
  const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  T[0] = 1.0f - (tyy + tzz); T[1] = txy - twz;          T[2] = txz + twy;           T[3] = static_cast<float>(x[0]);
  T[4] = txy + twz;          T[5] = 1.0f - (txx + tzz); T[6] = tyz - twx;           T[7] = static_cast<float>(x[1]);
  T[8] = txz - twy;          T[9] = tyz + twx;          T[10] = 1.0f - (txx + tyy); T[11] = static_cast<float>(x[2]);
  T[12] = 0.0f; T[13] = 0.0f; T[14] = 0.0f; T[15] = 1.0f;
}

void rotation_gradient(const double x[6], const double R[9], double g[6]) {
  const double phi = x[3], theta = x[4], psi = x[5];
  const double cphi = std::cos(phi), sphi = std::sin(phi);
  const double cth = std::cos(theta), sth = std::sin(theta);
  const double cpsi = std::cos(psi), spsi = std::sin(psi);
  // dR/dphi, dR/dtheta, dR/dpsi of R = Rz(psi) Ry(theta) Rx(phi), row-major
  const double d_phi[9] = {0.0, sphi * spsi + cphi * cpsi * sth,  cphi * spsi - cpsi * sphi * sth,
                           0.0, -cpsi * sphi + cphi * spsi * sth, -cphi * cpsi - sphi * spsi * sth,
                           0.0, cphi * cth,                       -cth * sphi};
  const double d_theta[9] = {-cpsi * sth, cpsi * cth * sphi, cphi * cpsi * cth,
                             -spsi * sth, cth * sphi * spsi, cphi * cth * spsi,
                             -cth,        -sphi * sth,       -cphi * sth};
  const double d_psi[9] = {-cth * spsi, -cphi * cpsi - sphi * spsi * sth, cpsi * sphi - cphi * spsi * sth,
                           cpsi * cth,  -cphi * spsi + cpsi * sphi * sth, sphi * spsi + cphi * cpsi * sth,
                           0.0,         0.0,                              0.0};
  auto inner = [&](const double* D) {  // matricesInnerProd (gicp_omp.h:318-327): sum_ij D(j,i) R(i,j)
    double r = 0.0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r += D[j * 3 + i] * R[i * 3 + j];
    return r;
  };
  g[3] = inner(d_phi);
  g[4] = inner(d_theta);
  g[5] = inner(d_psi);
}

Result run(const Params& prm, const float guess[16], Backend& dev) {
  Result res;
  float transformation[16], previous[16];
  for (int i = 0; i < 16; i++) transformation[i] = previous[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  while (!res.converged) {
    double R[9];  // rotation of transformation_ * guess, f64 (:411-419)
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0.0;
        for (int k = 0; k < 4; k++) s += static_cast<double>(transformation[i * 4 + k]) * static_cast<double>(guess[k * 4 + j]);
        R[i * 3 + j] = s;
      }
    if (!dev.correspond(transformation, R)) {
      res.backend_failed = true;
      break;
    }
    for (int i = 0; i < 16; i++) previous[i] = transformation[i];
    if (!estimate(prm, dev, res, transformation)) break;  // catch (pcl::PCLException&) { break; }
    double delta = 0.0;  // :482-494
    for (int k = 0; k < 4; k++)
      for (int l = 0; l < 4; l++) {
        const double ratio = (k < 3 && l < 3) ? 1.0 / prm.rotation_epsilon : 1.0 / prm.transformation_epsilon;
        const double c_delta = ratio * std::fabs(static_cast<double>(previous[k * 4 + l] - transformation[k * 4 + l]));
        if (c_delta > delta) delta = c_delta;
      }
    res.nr_iterations++;
    if (res.nr_iterations >= prm.max_iterations || delta < 1) {
      res.converged = true;
      for (int i = 0; i < 16; i++) previous[i] = transformation[i];
    }
  }
  // final_transformation_ = previous_transformation_ * guess  ([Eigen] Matrix4f product)
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++)
      res.final_T[r * 4 + c] = ((previous[r * 4] * guess[c] + previous[r * 4 + 1] * guess[4 + c]) + previous[r * 4 + 2] * guess[8 + c]) +
                               previous[r * 4 + 3] * guess[12 + c];
  return res;
}

}  // namespace gicp
