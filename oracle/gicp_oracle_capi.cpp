// gicp_oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY: extern "C" shim over oracle::GICP for ctypes.
#include <cstring>
#include <vector>

#include "gicp_oracle.hpp"

using namespace oracle;

namespace {
std::vector<Pt> pts_of(const float* p, size_t n, size_t stride_floats) {
  std::vector<Pt> v(n);
  for (size_t i = 0; i < n; i++) v[i] = Pt{p[i * stride_floats], p[i * stride_floats + 1], p[i * stride_floats + 2], 1.0f};
  return v;
}
void from_colmajor(const float* m, float T[4][4]) {
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) T[r][c] = m ? m[c * 4 + r] : (r == c ? 1.0f : 0.0f);
}
void to_colmajor(const float T[4][4], float* m) {
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) m[c * 4 + r] = T[r][c];
}
struct Session {
  GICP g;
  std::vector<Pt> output;  // guess-transformed source of the step-wise interface
  float guess[4][4];
};
}  // namespace

extern "C" {

void* gicp_oracle_create() { return new Session(); }
void gicp_oracle_destroy(void* h) { delete static_cast<Session*>(h); }

void gicp_oracle_set_params(void* h, int k, double gicp_eps, double rot_eps, double trans_eps, double corr_dist, int max_iter,
                            int max_inner) {
  GicpParams& p = static_cast<Session*>(h)->g.prm;
  p.k_correspondences = k;
  p.gicp_epsilon = gicp_eps;
  p.rotation_epsilon = rot_eps;
  p.transformation_epsilon = trans_eps;
  p.corr_dist_threshold = corr_dist;
  p.max_iterations = max_iter;
  p.max_inner_iterations = max_inner;
}
void gicp_oracle_set_target(void* h, const float* pts, size_t n, size_t stride_floats) {
  static_cast<Session*>(h)->g.set_target(pts_of(pts, n, stride_floats));
}
void gicp_oracle_set_source(void* h, const float* pts, size_t n, size_t stride_floats) {
  static_cast<Session*>(h)->g.set_source(pts_of(pts, n, stride_floats));
}

// setTargetCovariances / setSourceCovariances (gicp_omp.h:165-168,186-189): caller-supplied [n][9] row-major matrices replace the
// lazily computed ones until the cloud is set again (which == 0 target, 1 source; n == 0 clears them)
void gicp_oracle_set_covariances(void* h, int which, const double* cov, size_t n) {
  std::vector<M3>& dst = which == 0 ? static_cast<Session*>(h)->g.target_cov : static_cast<Session*>(h)->g.source_cov;
  dst.resize(n);
  if (n) std::memcpy(dst.data(), cov, n * sizeof(M3));
}

// covariances of a cloud: out [n][9] row-major; returns 0 when k exceeds the cloud
int gicp_oracle_covariances(const float* pts, size_t n, size_t stride_floats, int k, double eps, double* out) {
  std::vector<M3> cov;
  if (!GICP::covariances(pts_of(pts, n, stride_floats), k, eps, cov)) return 0;
  std::memcpy(out, cov.data(), n * sizeof(M3));
  return 1;
}

void gicp_oracle_knn(const float* cloud, size_t n, const float* query, size_t nq, int k, int* idx, float* d2) {
  std::vector<int> oi;
  std::vector<float> od;
  knn_exact(pts_of(cloud, n, 4), pts_of(query, nq, 4), k, oi, od);
  std::memcpy(idx, oi.data(), oi.size() * sizeof(int));
  std::memcpy(d2, od.data(), od.size() * sizeof(float));
}

// stats: nr_iterations, n_f, n_df, n_fdf, last correspondences
void gicp_oracle_align(void* h, const float* guess_colmajor, float* final_T_colmajor, int* converged, int* stats,
                       float* out_cloud /*ns x 4 or NULL*/) {
  Session* s = static_cast<Session*>(h);
  float guess[4][4];
  from_colmajor(guess_colmajor, guess);
  std::vector<Pt> out;
  const GicpResult r = s->g.align(guess, out_cloud ? &out : nullptr);
  to_colmajor(r.final_T, final_T_colmajor);
  *converged = r.converged ? 1 : 0;
  stats[0] = r.nr_iterations;
  stats[1] = r.n_f;
  stats[2] = r.n_df;
  stats[3] = r.n_fdf;
  stats[4] = r.last_correspondences;
  if (out_cloud) std::memcpy(out_cloud, out.data(), out.size() * sizeof(Pt));
}

// ---- step-wise interface for the kernel-level parity tests ----
// covariances of both clouds + guess-transformed source; then correspond(transformation) and functor(x)
int gicp_oracle_prepare(void* h, const float* guess_colmajor) {
  Session* s = static_cast<Session*>(h);
  from_colmajor(guess_colmajor, s->guess);
  GICP& g = s->g;
  bool ok = true;
  if (g.target_cov.empty()) ok = GICP::covariances(g.target, g.prm.k_correspondences, g.prm.gicp_epsilon, g.target_cov) && ok;
  if (g.source_cov.empty()) ok = GICP::covariances(g.source, g.prm.k_correspondences, g.prm.gicp_epsilon, g.source_cov) && ok;
  s->output = g.source;
  for (Pt& p : s->output) p.w = 1.0f;
  transform_cloud(s->output, s->output, s->guess);
  g.opt_src = &s->output;
  g.mahalanobis.clear();
  return ok ? 1 : 0;
}
// returns the number of correspondences; tgt_idx[i] = -1 where source point i has none; maha [ns][9]
int gicp_oracle_correspond(void* h, const float* transformation_colmajor, int* tgt_idx, float* maha) {
  Session* s = static_cast<Session*>(h);
  float T[4][4];
  from_colmajor(transformation_colmajor, T);
  const int m = s->g.correspond(s->output, T, s->guess);
  for (size_t i = 0; i < s->output.size(); i++) tgt_idx[i] = -1;
  for (int i = 0; i < m; i++) tgt_idx[s->g.corr_src[i]] = s->g.corr_tgt[i];
  std::memcpy(maha, s->g.mahalanobis.data(), s->output.size() * 9 * sizeof(float));
  return m;
}
// mode 0: operator() -> f ; 1: df -> g ; 2: fdf -> f, g
void gicp_oracle_functor(void* h, int mode, const double* x, double* f, double* g) {
  Session* s = static_cast<Session*>(h);
  if (mode == 0) *f = s->g.functor_f(x);
  else if (mode == 1) s->g.functor_df(x, g);
  else s->g.functor_fdf(x, *f, g);
}
void gicp_oracle_apply_state(const double* x, float* T_colmajor) {
  float T[4][4];
  GICP::apply_state(x, T);
  to_colmajor(T, T_colmajor);
}

}  // extern "C"
