// apps/map_sequence.cpp -- the processing loop of the reference's mapping node
// (lidar_subscriber/src/ndt_omp_mapping_node.cpp) written against the C-ABI alone (no ROS, no PCL): read the
// numbered cloud_N.pcd scans of a directory in order (process_new_clouds :110-136), voxel-filter each
// (load_and_filter_cloud :138-149), register every scan against its predecessor with the node's NDT settings
// (initialize_ndt :55-62, align_consecutive_clouds :151-169), chain the pose (process_available_clouds :70-100) and
// accumulate the global map (update_global_map :195-211).  What the node publishes as ROS messages is printed.
// Scans are read ahead in the background while the GPU works on the previous one.
//
// With a fourth argument "rosbag" the loop is the other mapping node's (lidar_subscriber/src/ndt_rosbag_mapping_node.cpp:46-75,
// with the scans of the directory standing in for the bag's messages): leaf 0.3 (:87), every registration starts from
// the previous one's result (pres_transform, :63,127), its fitness score is printed (:130), a registration that did not
// converge counts as identity (:137-140), and pose, trajectory and global map are updated after every scan (:64-68).
//
//   map_sequence <pcd_directory> [voxel_leaf_size (0.5 | 0.3)] [global_map_out.pcd | -] [rosbag]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ndt_mi355.h"

#define CHECK(call)                                                     \
  do {                                                                  \
    if ((call) != NDT_OK) {                                             \
      std::fprintf(stderr, "%s failed: %s\n", #call, ndt_last_error()); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

struct Pt {
  float x, y, z, w;
};

static void print_matrix(const char* title, const float* T) {
  std::printf("%s\n", title);
  for (int r = 0; r < 4; r++) std::printf("  %.9g %.9g %.9g %.9g\n", T[r], T[4 + r], T[8 + r], T[12 + r]);
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("usage: map_sequence <pcd_directory> [voxel_leaf_size] [global_map_out.pcd]\n");
    return 0;
  }
  const bool rosbag = argc > 4 && std::strcmp(argv[4], "rosbag") == 0;
  const float voxel_leaf_size = argc > 2 ? static_cast<float>(std::atof(argv[2])) : (rosbag ? 0.3f : 0.5f);  // :44 / rosbag :87
  ndt_handle h = nullptr;
  CHECK(ndt_create(0, &h));
  CHECK(ndt_set_resolution(h, 1.0f));  // initialize_parameters / initialize_ndt, :37-62
  CHECK(ndt_set_step_size(h, 0.1));
  CHECK(ndt_set_transformation_epsilon(h, 0.01));
  CHECK(ndt_set_maximum_iterations(h, 64));
  CHECK(ndt_set_num_threads(h, 40));
  CHECK(ndt_set_neighborhood_search_method(h, NDT_DIRECT7));

  ndt_pcd_sequence_handle seq = nullptr;
  CHECK(ndt_pcd_sequence_open(argv[1], &seq));

  const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::vector<Pt> previous, current;      // clouds_[current_index_ - 1], clouds_[current_index_]
  std::vector<std::vector<float>> trajectory;  // trajectory_
  std::vector<float> pose(identity, identity + 16), pres_transform(identity, identity + 16);  // rosbag node :33,95
  size_t loaded = 0;                      // clouds_.size()
  size_t registered = 0, not_converged = 0;
  double t_filter = 0, t_align = 0, t_map = 0;
  using clock = std::chrono::steady_clock;
  auto since = [](clock::time_point a) { return std::chrono::duration<double, std::milli>(clock::now() - a).count(); };
  const auto t_begin = clock::now();

  for (;;) {  // the node polls the directory once per second (:28-34); here: until a poll brings nothing new
    size_t fresh = 0;
    CHECK(ndt_pcd_sequence_poll(seq, loaded, &fresh));
    if (fresh == 0) break;
    for (;;) {
      const void* raw = nullptr;
      size_t n = 0;
      int dense = 1, number = -1;
      const ndt_status s = ndt_pcd_sequence_next(seq, &raw, &n, &dense, &number);
      if (s != NDT_OK) {  // loadPCDFile == -1 -> nullptr -> skipped (:140, :128)
        std::fprintf(stderr, "skipped: %s\n", ndt_last_error());
        continue;
      }
      if (!raw) break;
      // load_and_filter_cloud, :142-148
      auto t0 = clock::now();
      current.resize(n ? n : 1);
      size_t m = 0;
      const ndt_status fs = ndt_voxel_grid_filter(h, raw, n, sizeof(Pt), dense, voxel_leaf_size, current.data(), sizeof(Pt), &m);
      if (fs != NDT_OK && fs != NDT_ERR_GRID_OVERFLOW) {
        std::fprintf(stderr, "voxel filter failed: %s\n", ndt_last_error());
        return 1;
      }
      current.resize(m);
      t_filter += since(t0);
      if (current.empty()) continue;  // :128 -- empty clouds are not kept
      loaded++;
      std::printf("Loaded cloud_%d.pcd (%zu points)\n", number, current.size());
      if (loaded == 1) {  // load_initial_clouds, :64-68
        t0 = clock::now();
        int overflowed = 0;
        CHECK(ndt_map_update(h, current.data(), current.size(), sizeof(Pt), 1, identity, 0.5f, &overflowed));
        t_map += since(t0);
      } else {  // process_available_clouds, :70-100
        t0 = clock::now();
        CHECK(ndt_set_input_target(h, previous.data(), previous.size(), sizeof(Pt), 1));
        CHECK(ndt_set_input_source(h, current.data(), current.size(), sizeof(Pt)));
        float T[16];
        int converged = 0, iterations = 0;
        double probability = 0;
        CHECK(ndt_align(h, rosbag ? pres_transform.data() : nullptr, T, &converged, &iterations, &probability, nullptr, 0));
        t_align += since(t0);
        registered++;
        if (rosbag) {  // perform_registration :119-141, then :63-68
          double fitness = 0;
          CHECK(ndt_get_fitness_score(h, 1.7976931348623157e308, &fitness));
          std::printf("fitness: %.9g (%d iterations%s)\n", fitness, iterations, converged ? "" : ", not converged");
          if (!converged) {
            not_converged++;
            for (int i = 0; i < 16; i++) T[i] = identity[i];
          }
          pres_transform.assign(T, T + 16);
          ndt_host_chain_pose(pose.data(), T, pose.data());
          trajectory.push_back(pose);
          t0 = clock::now();
          int overflowed = 0;
          CHECK(ndt_map_update(h, current.data(), current.size(), sizeof(Pt), 1, pose.data(), 0.5f, &overflowed));  // map_voxel, :88
          t_map += since(t0);
        } else if (converged) {
          char title[96];
          std::snprintf(title, sizeof(title), "Transform %zu to %zu: (%d iterations)", loaded - 2, loaded - 1, iterations);
          print_matrix(title, T);
          std::vector<float> global(T, T + 16);
          if (!trajectory.empty()) ndt_host_chain_pose(trajectory.back().data(), T, global.data());  // trajectory_.back() * transform
          trajectory.push_back(global);
          t0 = clock::now();
          int overflowed = 0;
          CHECK(ndt_map_update(h, current.data(), current.size(), sizeof(Pt), 1, global.data(), 0.5f, &overflowed));  // :204: leaf fixed at 0.5
          t_map += since(t0);
        } else {
          not_converged++;
        }
      }
      previous.swap(current);
    }
  }

  size_t map_points = 0;
  CHECK(ndt_map_size(h, &map_points));
  std::printf("\nclouds %zu  registrations %zu (not converged %zu)  global map %zu points\n", loaded, registered, not_converged, map_points);
  for (size_t i = 0; i < trajectory.size(); i++) {
    char title[64];
    std::snprintf(title, sizeof(title), "trajectory[%zu]:", i);
    print_matrix(title, trajectory[i].data());
  }
  std::printf("time: total %.2f ms  (prefilter %.2f, set inputs + align %.2f, map update %.2f; file reading overlapped)\n",
              since(t_begin), t_filter, t_align, t_map);
  if (argc > 3 && std::strcmp(argv[3], "-") != 0 && map_points) {
    std::vector<Pt> map(map_points);
    CHECK(ndt_map_get(h, map.data(), sizeof(Pt)));
    CHECK(ndt_pcd_write_xyz(argv[3], map.data(), map.size(), sizeof(Pt), 1));
    std::printf("global map written to %s\n", argv[3]);
  }
  ndt_pcd_sequence_close(seq);
  ndt_destroy(h);
  return 0;
}
