// ndt_driver.hpp -- host side of the MI355X NDT core: scalar math around the
// GPU evaluations (PCL-free, Eigen-free).
//
// The Newton / More-Thuente driver of pclomp::NormalDistributionsTransform
// (reference ndt_omp_impl.hpp:80-171, 648-932) is written as a RESUMABLE STATE
// MACHINE: it emits one evaluation request at a time and is fed the result.
// A single registration pumps one solver; map-build batches pump many solvers
// in lock-step with one fused GPU launch per step.
#pragma once
#include <cstddef>

namespace ndt {

// Gaussian fitting constants, ndt_omp_impl.hpp:86-93.
struct Gauss {
  double d1, d2, d3;
};
Gauss gauss_constants(float resolution, double outlier_ratio);

// Eigen::JacobiSVD<Matrix6d>(H, FullU|FullV).solve(b)  (ndt_omp_impl.hpp:127-129):
// minimum-norm least-squares solution with Eigen's default rank threshold.
void solve6(const double H[36], const double b[6], double x[6]);

// Translation3f * AngleAxisf(X) * AngleAxisf(Y) * AngleAxisf(Z) in f32
// (ndt_omp_impl.hpp:146-149, 827-830).  T is column-major 4x4.
void pose_to_matrix(const double p[6], float T[16]);
// translation + rotation().eulerAngles(0,1,2)  (ndt_omp_impl.hpp:103-111).
void matrix_to_pose(const float T[16], double p[6]);

// pose * transform of the mapping nodes (ndt_omp_mapping_node.cpp:88-99, ndt_rosbag_mapping_node.cpp:62-68):
// [Eigen] fixed-size Matrix4f product, column-major, each entry ((a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j) + a_i3 b_3j.
void chain_pose(const float a[16], const float b[16], float out[16]);

// computeAngleDerivatives, ndt_omp_impl.hpp:288-395.  j/h are the f32 matrices
// (h row 6 carries +sy, :383); jd/hd the f64 vectors used by computeHessian
// (hd row 6 carries -sy, :361).
struct AngleDerivs {
  float j[8][3];
  float h[15][3];
  double jd[8][3];
  double hd[15][3];
};
void angle_derivatives(const double p[6], AngleDerivs& out);
// cos / sin of roll, pitch, yaw with the reference's snap (|angle| < 10e-5 -> 1, 0): cx cy cz sx sy sz
void snapped_cos_sin(const double p[6], double cs[6]);

enum EvalKind { EVAL_WITH_HESSIAN = 0, EVAL_NO_HESSIAN = 1, EVAL_HESSIAN_F64 = 2, EVAL_NONE = 3 };

struct EvalRequest {
  EvalKind kind;
  float T[16];  // column-major transform to apply to the source
  double p[6];  // pose the angle derivatives are taken at
};

struct EvalResult {
  double score;
  double g[6];
  double H[36];  // row-major; ignored for EVAL_NO_HESSIAN
};

struct SolverParams {
  float resolution = 1.0f;
  double step_size = 0.1;
  double outlier_ratio = 0.55;
  double trans_eps = 0.1;
  int max_iter = 35;
};

class ScanSolver {
 public:
  void start(const float* guess /*16 col-major or nullptr*/, size_t n_source, const SolverParams& prm);
  bool done() const { return state_ == S_DONE; }
  const EvalRequest& request() const { return req_; }
  void feed(const EvalResult& r);

  // results (valid once done())
  float final_T[16];
  bool converged = false;
  int nr_iterations = 0;
  double trans_probability = 0;
  int n_evals = 0, n_hess = 0;

 private:
  enum State { S_INIT, S_MT_FIRST, S_MT_LOOP, S_MT_HESS, S_DONE };
  State state_ = S_DONE;
  EvalRequest req_;
  SolverParams prm_;
  size_t n_source_ = 0;
  // Newton state
  double p_[6], score_ = 0, g_[6], H_[36];
  // line-search state (computeStepLengthMT locals)
  double x_[6], dir_[6], x_t_[6];
  double phi_0_, d_phi_0_, a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, step_min_, step_max_;
  double phi_t_, d_phi_t_, psi_t_, d_psi_t_;
  bool interval_converged_, open_interval_;
  int step_iterations_;

  void newton_top();
  void mt_check();
  void mt_finish();
  void issue_trial(EvalKind kind);
  void finish(bool converged);
};

}  // namespace ndt
