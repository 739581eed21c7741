// host-to-host ndt_voxel_grid_filter at the nodes' size, timed from C++ (no Python allocations in the way):
//   g++ -O2 -std=c++17 -Iinclude tools/probes/time_filter.cpp -o /tmp/time_filter -Ltoyslam_amd -lndt_mi355 -Wl,-rpath,$PWD/toyslam_amd && /tmp/time_filter
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "ndt_mi355.h"
int main() {
  const size_t n = 60000;
  std::mt19937 rng(3);
  std::uniform_real_distribution<float> u(-30.f, 30.f), sheet(-0.05f, 0.05f);
  std::vector<float> in(4 * n), out(4 * n);
  for (size_t i = 0; i < n; i++) {  // mostly sheets: a filtered size like a lidar sweep's
    in[4 * i] = u(rng); in[4 * i + 1] = u(rng); in[4 * i + 2] = (i % 3) ? sheet(rng) + float(i % 5) : u(rng) * 0.2f; in[4 * i + 3] = 1.f;
  }
  ndt_handle h;
  if (ndt_create(0, &h) != NDT_OK || ndt_warm_up(h, 65536) != NDT_OK) { std::fprintf(stderr, "%s\n", ndt_last_error()); return 1; }
  size_t m = 0;
  std::vector<double> t;
  for (int it = 0; it < 40; it++) {
    const auto t0 = std::chrono::steady_clock::now();
    if (ndt_voxel_grid_filter(h, in.data(), n, 16, 1, 0.5f, out.data(), 16, &m) != NDT_OK) { std::fprintf(stderr, "%s\n", ndt_last_error()); return 1; }
    t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  std::sort(t.begin(), t.end());
  std::printf("{\"points\": %zu, \"kept\": %zu, \"host_to_host_us_median\": %.1f, \"min\": %.1f}\n", n, m, t[t.size() / 2], t[0]);
  ndt_destroy(h);
  return 0;
}
