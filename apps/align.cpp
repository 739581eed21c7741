// apps/align.cpp -- the pclomp part of the reference's benchmark program (ndt_omp/apps/align.cpp) written
// against the C-ABI alone (no PCL, no ROS): load two PCD files, 0.1 m VoxelGrid down-sample, then GICP at
// class defaults (align.cpp:84-86) and NDT with KDTREE / DIRECT7 / DIRECT1 at resolution 1.0 (:88-105): one
// registration timed, ten more timed, and the fitness score -- the three numbers per method the reference
// prints and its README tabulates (ndt_omp/README.md:13-46).  The stock pcl::GICP / pcl::NDT runs and the
// visualiser of the original are not here.
//
//   align target.pcd source.pcd [leaf_size (0.1; 0 = no down-sample)]
#include <chrono>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gicp_mi355.h"
#include "ndt_mi355.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    if ((call) != NDT_OK) {                                                  \
      std::fprintf(stderr, "%s failed: %s\n", #call, ndt_last_error());      \
      return 1;                                                              \
    }                                                                        \
  } while (0)

struct Pt {
  float x, y, z, w;
};

static int load(ndt_handle h, const char* path, float leaf, std::vector<Pt>& out) {
  size_t n = 0;
  int fields = 0, kind = 0, dense = 1;
  CHECK(ndt_pcd_read_header(path, &n, &fields, &kind));
  std::vector<Pt> raw(n ? n : 1);
  CHECK(ndt_pcd_read_xyz(path, raw.data(), n, sizeof(Pt), &n, &dense));
  if (!(leaf > 0)) {
    raw.resize(n);
    out.swap(raw);
    return 0;
  }
  out.resize(n ? n : 1);
  size_t m = 0;
  CHECK(ndt_voxel_grid_filter(h, raw.data(), n, sizeof(Pt), dense, leaf, out.data(), sizeof(Pt), &m));  // align.cpp:60-69
  out.resize(m);
  std::printf("%s: %zu points (%d fields, %s) -> %zu after the %.2f m voxel grid\n", path, n, fields,
              kind == 0 ? "ascii" : kind == 1 ? "binary" : "binary_compressed", m, leaf);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::printf("usage: align target.pcd source.pcd [leaf_size]\n");
    return 0;
  }
  const float leaf = argc > 3 ? static_cast<float>(std::atof(argv[3])) : 0.1f;
  ndt_handle h = nullptr;
  CHECK(ndt_create(0, &h));
  std::vector<Pt> target, source;
  if (load(h, argv[1], leaf, target) || load(h, argv[2], leaf, source)) return 1;

  using clock = std::chrono::steady_clock;
  auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  std::vector<Pt> aligned(source.size() ? source.size() : 1);
  {  // align.cpp:84-86
    std::printf("--- gicp_mi355 ---\n");
    gicp_handle g = nullptr;
    CHECK(gicp_create(0, &g));
    CHECK(gicp_set_input_target(g, target.data(), target.size(), sizeof(Pt)));
    CHECK(gicp_set_input_source(g, source.data(), source.size(), sizeof(Pt)));
    float T[16];
    int converged = 0, iterations = 0;
    const auto t1 = clock::now();
    CHECK(gicp_align(g, nullptr, T, &converged, &iterations, aligned.data()));
    const auto t2 = clock::now();
    std::printf("single : %.3f[msec]\n", ms(t1, t2));
    for (int i = 0; i < 10; i++) CHECK(gicp_align(g, nullptr, T, &converged, &iterations, aligned.data()));
    const auto t3 = clock::now();
    std::printf("10times: %.3f[msec]\n", ms(t2, t3));
    double fitness = 0;
    CHECK(gicp_get_fitness_score(g, DBL_MAX, &fitness));
    std::printf("fitness: %.6g\n", fitness);
    int n_f = 0, n_df = 0, n_fdf = 0, corr = 0;
    CHECK(gicp_get_stats(g, &n_f, &n_df, &n_fdf, &corr));
    std::printf("converged: %d  iterations: %d  objective evaluations: %d  correspondences: %d\n", converged, iterations,
                n_f + n_df + n_fdf, corr);
    std::printf("T:");
    for (int r = 0; r < 4; r++) {
      for (int c = 0; c < 4; c++) std::printf(" %.9g", T[c * 4 + r]);
      std::printf(r < 3 ? " |" : "\n\n");
    }
    gicp_destroy(g);
  }

  CHECK(ndt_set_resolution(h, 1.0f));  // align.cpp:96
  const struct { const char* name; int method; } methods[] = {{"KDTREE", NDT_KDTREE}, {"DIRECT7", NDT_DIRECT7}, {"DIRECT1", NDT_DIRECT1}};
  for (const auto& m : methods) {
    std::printf("--- ndt_mi355 (%s) ---\n", m.name);
    CHECK(ndt_set_neighborhood_search_method(h, m.method));
    CHECK(ndt_set_input_target(h, target.data(), target.size(), sizeof(Pt), 1));  // align(): setInputTarget / setInputSource
    CHECK(ndt_set_input_source(h, source.data(), source.size(), sizeof(Pt)));
    float T[16];
    int converged = 0, iterations = 0;
    double prob = 0;
    const auto t1 = clock::now();
    CHECK(ndt_align(h, nullptr, T, &converged, &iterations, &prob, aligned.data(), sizeof(Pt)));
    const auto t2 = clock::now();
    std::printf("single : %.3f[msec]\n", ms(t1, t2));
    for (int i = 0; i < 10; i++) CHECK(ndt_align(h, nullptr, T, &converged, &iterations, &prob, aligned.data(), sizeof(Pt)));
    const auto t3 = clock::now();
    std::printf("10times: %.3f[msec]\n", ms(t2, t3));
    double fitness = 0;
    CHECK(ndt_get_fitness_score(h, DBL_MAX, &fitness));
    std::printf("fitness: %.6g\n", fitness);
    std::printf("converged: %d  iterations: %d  transformation_probability: %.6g\n", converged, iterations, prob);
    std::printf("T:");
    for (int r = 0; r < 4; r++) {
      for (int c = 0; c < 4; c++) std::printf(" %.9g", T[c * 4 + r]);
      std::printf(r < 3 ? " |" : "\n\n");
    }
  }
  ndt_destroy(h);
  return 0;
}
