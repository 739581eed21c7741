// tests/driver_fuzz.cpp -- the host Newton / More-Thuente state machine (ndt_driver.cpp) fed random,
// degenerate and non-finite evaluation results, built with ASan + UBSan by tests/test_host_logic.py:
// every run must terminate (bounded number of requests) without undefined behaviour.
//   driver_fuzz [runs]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>

#include "ndt_driver.hpp"

int main(int argc, char** argv) {
  const int runs = argc > 1 ? std::atoi(argv[1]) : 20000;
  std::mt19937 rng(11);
  std::normal_distribution<double> nd;
  std::uniform_real_distribution<double> un(0.0, 1.0);
  long long total_requests = 0, max_requests = 0;
  int converged = 0;
  for (int r = 0; r < runs; r++) {
    ndt::SolverParams prm;
    prm.resolution = static_cast<float>(0.25 + 3.0 * un(rng));
    prm.step_size = 0.01 + un(rng);
    prm.outlier_ratio = 0.1 + 0.8 * un(rng);
    prm.trans_eps = (r % 7 == 0) ? 0.0 : std::pow(10.0, -6.0 * un(rng));
    prm.max_iter = static_cast<int>(un(rng) * 40);
    float guess[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (r % 3 == 0) {
      double p[6];
      for (int k = 0; k < 6; k++) p[k] = nd(rng) * (k < 3 ? 5.0 : 0.5);
      ndt::pose_to_matrix(p, guess);
    }
    ndt::ScanSolver s;
    s.start((r % 5 == 0) ? nullptr : guess, static_cast<size_t>(un(rng) * 100000), prm);
    // a random quadratic bowl (sometimes indefinite / singular) plus noise, with rare poisoned results
    double A[36], c[6];
    for (int i = 0; i < 6; i++) {
      c[i] = nd(rng);
      for (int j = 0; j <= i; j++) A[i * 6 + j] = A[j * 6 + i] = nd(rng) * ((r % 11 == 0 && i == j) ? 0.0 : 1.0) - (i == j ? 3.0 : 0.0);
    }
    long long requests = 0;
    while (!s.done()) {
      if (++requests > 100000) {
        std::fprintf(stderr, "run %d does not terminate\n", r);
        return 1;
      }
      const ndt::EvalRequest& q = s.request();
      ndt::EvalResult res;
      double x[6];
      for (int k = 0; k < 6; k++) x[k] = q.p[k] - c[k];
      res.score = 0;
      for (int i = 0; i < 6; i++) {
        res.g[i] = 0;
        for (int j = 0; j < 6; j++) {
          res.g[i] += A[i * 6 + j] * x[j];
          res.H[i * 6 + j] = A[i * 6 + j] + 1e-9 * nd(rng);
        }
        res.score += 0.5 * x[i] * res.g[i];
      }
      const double roll = un(rng);
      if (roll < 0.002) res.score = std::numeric_limits<double>::quiet_NaN();
      else if (roll < 0.004) res.g[static_cast<int>(un(rng) * 6) % 6] = std::numeric_limits<double>::infinity();
      else if (roll < 0.006) for (double& v : res.H) v = 0.0;
      else if (roll < 0.008) res.H[7] = std::numeric_limits<double>::quiet_NaN();
      else if (roll < 0.010) { for (double& v : res.g) v = 0.0; }
      s.feed(res);
    }
    total_requests += requests;
    if (requests > max_requests) max_requests = requests;
    converged += s.converged ? 1 : 0;
    if (s.nr_iterations > prm.max_iter + 2 || s.nr_iterations < 0) {
      std::fprintf(stderr, "run %d: %d iterations with max_iter %d\n", r, s.nr_iterations, prm.max_iter);
      return 1;
    }
  }
  std::printf("driver fuzz: %d runs, %lld evaluations (max %lld in one run), %d converged, no crash\n", runs, total_requests,
              max_requests, converged);
  return 0;
}
