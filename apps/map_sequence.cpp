// apps/map_sequence.cpp -- the processing loop of the reference's mapping node
// (lidar_subscriber/src/ndt_omp_mapping_node.cpp) written against the C-ABI alone (no ROS, no PCL): read the
// numbered cloud_N.pcd scans of a directory in order (process_new_clouds :110-136), voxel-filter each
// (load_and_filter_cloud :138-149), register every scan against its predecessor with the node's NDT settings
// (initialize_ndt :55-62, align_consecutive_clouds :151-169), chain the pose (process_available_clouds :70-100) and
// accumulate the global map (update_global_map :195-211).  What the node publishes as ROS messages is printed.
//
// With a fourth argument "rosbag" the loop is the other mapping node's (lidar_subscriber/src/ndt_rosbag_mapping_node.cpp:46-75,
// with the scans of the directory standing in for the bag's messages): leaf 0.3 (:87), every registration starts from
// the previous one's result (pres_transform, :63,127), its fitness score is printed (:130), a registration that did not
// converge counts as identity (:137-140), and pose, trajectory and global map are updated after every scan (:64-68).
//
// The node does everything for a scan one step after the other, and so does this program by default (files are read and
// parsed ahead in the background).  With a fifth argument "pipeline" the steps of CONSECUTIVE scans overlap (the results are
// the same, bit for bit -- every step runs the same kernels on the same data):
//   prep thread      voxel filter of scan k+1, its upload as the next source, its voxel grid as the target after that, on a
//                    prep handle of its own (NDT_PIPELINE_PARTITION=1: on the side partition of the CUs, ndt_set_cu_partition)
//   this thread      registration of scan k (inputs taken over with ndt_share_input_target / ndt_share_input_source: no
//                    copy, no rebuild), pose chain, printing
//   map thread       global map update of scan k-1 (the map is not an input of any registration)
// Measured at the nodes' size (40 scans of 60 k raw points, tools/time_map_sequence.py, round 3): 1.9 ms per scan one step
// after the other, 2.7 ms overlapped, 3.7 ms overlapped on CU partitions -- the steps are short host-paced sequences of
// pageable copies and small launches, and two threads driving them get in each other's way; hence not the default.
//
//   map_sequence <pcd_directory> [voxel_leaf_size (0.5 | 0.3)] [global_map_out.pcd | -] [rosbag | node] [serial | pipeline] [host]
// ("host": the serial loop with every cloud passing through host buffers, as in rounds 1-3; the default keeps them in HBM)
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
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

using clock_type = std::chrono::steady_clock;
static double since(clock_type::time_point a) { return std::chrono::duration<double, std::milli>(clock_type::now() - a).count(); }

static const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

// NDT_PIPELINE_PARTITION=1: the pipeline's handles live on CU partitions instead of sharing the whole device
static bool use_partitions() {
  const char* v = std::getenv("NDT_PIPELINE_PARTITION");
  return v && std::atoi(v) != 0;
}

static int configure(ndt_handle h) {  // initialize_parameters / initialize_ndt, :37-62
  CHECK(ndt_set_resolution(h, 1.0f));
  CHECK(ndt_set_step_size(h, 0.1));
  CHECK(ndt_set_transformation_epsilon(h, 0.01));
  CHECK(ndt_set_maximum_iterations(h, 64));
  CHECK(ndt_set_num_threads(h, 40));
  CHECK(ndt_set_neighborhood_search_method(h, NDT_DIRECT7));
  return 0;
}

// a small blocking queue
template <class T>
class Channel {
 public:
  void put(T v) {
    {
      std::lock_guard<std::mutex> g(m_);
      q_.push_back(std::move(v));
    }
    cv_.notify_one();
  }
  T take() {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return !q_.empty(); });
    T v = std::move(q_.front());
    q_.pop_front();
    return v;
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<T> q_;
};

// what the two loops share: the state of the node and what happens with a registration's result
struct Node {
  bool rosbag = false;
  std::vector<std::vector<float>> trajectory;                     // trajectory_
  std::vector<float> pose = std::vector<float>(kIdentity, kIdentity + 16);
  std::vector<float> pres_transform = std::vector<float>(kIdentity, kIdentity + 16);  // rosbag node :33,95
  size_t loaded = 0, registered = 0, not_converged = 0;
  double t_filter = 0, t_align = 0, t_map = 0, t_wait = 0;  // t_wait: for the next file (the reader runs ahead in the background)

  // after align: -> whether the scan goes into the global map, and with which pose
  bool after_align(ndt_handle h, float* T, int converged, int iterations, std::vector<float>& map_pose, std::string& err) {
    registered++;
    if (rosbag) {  // perform_registration :119-141, then :63-68
      double fitness = 0;
      if (ndt_get_fitness_score(h, 1.7976931348623157e308, &fitness) != NDT_OK) {
        err = ndt_last_error();
        return false;
      }
      std::printf("fitness: %.9g (%d iterations%s)\n", fitness, iterations, converged ? "" : ", not converged");
      if (!converged) {
        not_converged++;
        for (int i = 0; i < 16; i++) T[i] = kIdentity[i];
      }
      pres_transform.assign(T, T + 16);
      ndt_host_chain_pose(pose.data(), T, pose.data());
      trajectory.push_back(pose);
      map_pose = pose;
      return true;
    }
    if (converged) {
      char title[96];
      std::snprintf(title, sizeof(title), "Transform %zu to %zu: (%d iterations)", loaded - 2, loaded - 1, iterations);
      print_matrix(title, T);
      std::vector<float> global(T, T + 16);
      if (!trajectory.empty()) ndt_host_chain_pose(trajectory.back().data(), T, global.data());  // trajectory_.back() * transform
      trajectory.push_back(global);
      map_pose = global;
      return true;
    }
    not_converged++;
    return false;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// the node's loop as it stands, one handle, one step after the other -- with every cloud staying in HBM (the default):
// the raw scan goes up once (by the reader thread, ahead of time: ndt_pcd_sequence_stage), the prefilter leaves its result on the device as an
// ndt_cloud, and that one object is the source of this registration, the target of the next one and what the map update
// adds -- no download, no second or third upload / repack / bounding-box pass of the same points
// ---------------------------------------------------------------------------------------------------------------------
static int run_resident(Node& node, ndt_pcd_sequence_handle seq, float voxel_leaf_size, ndt_handle h, size_t first_poll) {
  // The prefilter of scan k + 1 is BEGUN (queued on the handle's filter stream: ndt_cloud_voxel_filter_begin -- its input's
  // boxes were computed by the reader, so nothing has to come back from the device first) before scan k is registered,
  // and ended when scan k + 1's turn comes: the node's steps in the node's order, the GPU working on two of them at once.
  struct Raw {
    ndt_cloud view = nullptr;  // the staged scan (the sequence's memory)
    size_t n = 0;
    int dense = 1, number = -1;
  };
  const char* ov_env = std::getenv("MAP_SEQUENCE_OVERLAP");
  const bool overlap = ov_env && std::atoi(ov_env) != 0;  // (measured: no gain, the loop is bound by the GPU's time; off)
  ndt_cloud previous = nullptr;  // clouds_[current_index_ - 1]
  Raw begun;                     // the scan whose prefilter is under way
  bool have_begun = false, queue_dry = first_poll == 0;
  int rc = 0;
  // the next queued scan -> its prefilter begun; false: nothing queued right now
  auto begin_next = [&]() -> int {
    for (;;) {
      Raw r;
      const void* host = nullptr;
      const auto t_next = clock_type::now();
      const ndt_status s = ndt_pcd_sequence_next_cloud(seq, &r.view, &host, &r.n, &r.dense, &r.number);
      node.t_wait += since(t_next);
      if (s != NDT_OK) {  // loadPCDFile == -1 -> nullptr -> skipped (:140, :128)
        std::fprintf(stderr, "skipped: %s\n", ndt_last_error());
        continue;
      }
      if (!r.view) {
        queue_dry = true;
        return 0;
      }
      const auto t0 = clock_type::now();
      CHECK(ndt_cloud_voxel_filter_begin(h, r.view, r.dense, voxel_leaf_size));
      node.t_filter += since(t0);
      begun = r;
      have_begun = true;
      return 0;
    }
  };
  for (; !rc;) {
    if (!have_begun) {
      if (queue_dry) {  // process_new_clouds looks at the directory (the first look ran before the device's start-up)
        size_t fresh = 0;
        CHECK(ndt_pcd_sequence_poll(seq, node.loaded, &fresh));
        if (fresh == 0) break;
        queue_dry = false;
      }
      if (begin_next()) return 1;
      if (!have_begun) continue;  // (every queued file was unreadable: look again)
    }
    // ---- load_and_filter_cloud of this scan ends (:142-148) ...
    auto t0 = clock_type::now();
    ndt_cloud current = nullptr;
    int overflowed = 0;
    size_t m = 0;
    if (ndt_cloud_voxel_filter_end(h, &current, &overflowed) != NDT_OK || ndt_cloud_size(current, &m) != NDT_OK) {
      std::fprintf(stderr, "voxel filter failed: %s\n", ndt_last_error());
      return 1;
    }
    node.t_filter += since(t0);
    const int number = begun.number;
    ndt_cloud_release(begun.view);
    have_begun = false;
    // ... and the next scan's begins.  Not across a look at the directory: process_new_clouds counts the clouds KEPT so far,
    // and whether the scan in flight is kept (not empty) is known only when its filter has ended.
    if (overlap && !queue_dry && begin_next()) return 1;
    if (m == 0) {  // :128 -- empty clouds are not kept
      ndt_cloud_release(current);
      continue;
    }
    node.loaded++;
    std::printf("Loaded cloud_%d.pcd (%zu points)\n", number, m);
    auto step = [&]() -> int {
      if (node.loaded == 1) {  // load_initial_clouds, :64-68
        t0 = clock_type::now();
        int ov = 0;
        CHECK(ndt_map_update_cloud(h, current, 1, kIdentity, 0.5f, &ov));
        node.t_map += since(t0);
        return 0;
      }
      t0 = clock_type::now();  // process_available_clouds, :70-100
      CHECK(ndt_set_input_target_cloud(h, previous, 1));
      CHECK(ndt_set_input_source_cloud(h, current));
      float T[16];
      int converged = 0, iterations = 0;
      double probability = 0;
      CHECK(ndt_align(h, node.rosbag ? node.pres_transform.data() : nullptr, T, &converged, &iterations, &probability, nullptr, 0));
      node.t_align += since(t0);
      std::vector<float> map_pose;
      std::string err;
      const bool into_map = node.after_align(h, T, converged, iterations, map_pose, err);
      if (!err.empty()) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
      }
      if (into_map) {
        t0 = clock_type::now();
        int ov = 0;
        CHECK(ndt_map_update_cloud(h, current, 1, map_pose.data(), 0.5f, &ov));  // :204 / map_voxel :88: leaf fixed at 0.5
        node.t_map += since(t0);
      }
      return 0;
    };
    rc = step();
    ndt_cloud_release(previous);
    previous = current;
  }
  if (have_begun) {  // (left over after an error)
    ndt_cloud c = nullptr;
    (void)ndt_cloud_voxel_filter_end(h, &c, nullptr);
    ndt_cloud_release(c);
    ndt_cloud_release(begun.view);
  }
  ndt_cloud_release(previous);
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// the same loop through host buffers, as a caller that holds its clouds in host memory (PCL) has to run it
// ---------------------------------------------------------------------------------------------------------------------
static int run_serial(Node& node, ndt_pcd_sequence_handle seq, float voxel_leaf_size, ndt_handle h) {
  std::vector<Pt> previous, current;  // clouds_[current_index_ - 1], clouds_[current_index_]
  for (;;) {  // the node polls the directory once per second (:28-34); here: until a poll brings nothing new
    size_t fresh = 0;
    CHECK(ndt_pcd_sequence_poll(seq, node.loaded, &fresh));
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
      auto t0 = clock_type::now();
      current.resize(n ? n : 1);
      size_t m = 0;
      const ndt_status fs = ndt_voxel_grid_filter(h, raw, n, sizeof(Pt), dense, voxel_leaf_size, current.data(), sizeof(Pt), &m);
      if (fs != NDT_OK && fs != NDT_ERR_GRID_OVERFLOW) {
        std::fprintf(stderr, "voxel filter failed: %s\n", ndt_last_error());
        return 1;
      }
      current.resize(m);
      node.t_filter += since(t0);
      if (current.empty()) continue;  // :128 -- empty clouds are not kept
      node.loaded++;
      std::printf("Loaded cloud_%d.pcd (%zu points)\n", number, current.size());
      if (node.loaded == 1) {  // load_initial_clouds, :64-68
        t0 = clock_type::now();
        int overflowed = 0;
        CHECK(ndt_map_update(h, current.data(), current.size(), sizeof(Pt), 1, kIdentity, 0.5f, &overflowed));
        node.t_map += since(t0);
      } else {  // process_available_clouds, :70-100
        t0 = clock_type::now();
        CHECK(ndt_set_input_target(h, previous.data(), previous.size(), sizeof(Pt), 1));
        CHECK(ndt_set_input_source(h, current.data(), current.size(), sizeof(Pt)));
        float T[16];
        int converged = 0, iterations = 0;
        double probability = 0;
        CHECK(ndt_align(h, node.rosbag ? node.pres_transform.data() : nullptr, T, &converged, &iterations, &probability, nullptr, 0));
        node.t_align += since(t0);
        std::vector<float> map_pose;
        std::string err;
        const bool into_map = node.after_align(h, T, converged, iterations, map_pose, err);
        if (!err.empty()) {
          std::fprintf(stderr, "%s\n", err.c_str());
          return 1;
        }
        if (into_map) {
          t0 = clock_type::now();
          int overflowed = 0;
          CHECK(ndt_map_update(h, current.data(), current.size(), sizeof(Pt), 1, map_pose.data(), 0.5f, &overflowed));  // :204 / map_voxel :88: leaf fixed at 0.5
          node.t_map += since(t0);
        }
      }
      previous.swap(current);
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// the same loop with the steps of consecutive scans overlapped
// ---------------------------------------------------------------------------------------------------------------------
struct Prepared {
  ndt_handle h = nullptr;   // prep handle holding the filtered scan as input source AND as (built) input target
  std::vector<Pt> cloud;    // the filtered scan on the host (for the map update)
  int number = -1;
  double filter_ms = 0;
  bool end = false;
  std::string error;
};
struct MapJob {
  std::vector<Pt> cloud;
  std::vector<float> pose;
  bool end = false;
};

static int run_pipelined(Node& node, ndt_pcd_sequence_handle seq, float voxel_leaf_size, ndt_handle h, ndt_handle map_handle) {
  // three prep handles rotate: one being prepared, one holding the current source, one holding the current target
  constexpr int kPrep = 3;
  ndt_handle prep[kPrep] = {nullptr, nullptr, nullptr};
  for (int i = 0; i < kPrep; i++) {
    CHECK(ndt_create(0, &prep[i]));
    if (configure(prep[i])) return 1;
    if (use_partitions()) CHECK(ndt_set_cu_partition(prep[i], 2));
  }
  Channel<ndt_handle> free_prep;
  for (int i = 0; i < kPrep; i++) free_prep.put(prep[i]);
  Channel<Prepared> prepared;
  Channel<MapJob> map_jobs;
  std::string map_error;
  double map_ms = 0;

  std::thread prep_thread([&] {
    size_t kept = 0;  // clouds_.size() as process_new_clouds sees it
    for (;;) {
      size_t fresh = 0;
      if (ndt_pcd_sequence_poll(seq, kept, &fresh) != NDT_OK) {
        Prepared e;
        e.end = true;
        e.error = ndt_last_error();
        prepared.put(std::move(e));
        return;
      }
      if (fresh == 0) break;
      for (;;) {
        const void* raw = nullptr;
        size_t n = 0;
        int dense = 1, number = -1;
        const ndt_status s = ndt_pcd_sequence_next(seq, &raw, &n, &dense, &number);
        if (s != NDT_OK) {
          std::fprintf(stderr, "skipped: %s\n", ndt_last_error());
          continue;
        }
        if (!raw) break;
        Prepared p;
        p.number = number;
        p.h = free_prep.take();
        const auto t0 = clock_type::now();
        p.cloud.resize(n ? n : 1);
        size_t m = 0;
        const ndt_status fs = ndt_voxel_grid_filter(p.h, raw, n, sizeof(Pt), dense, voxel_leaf_size, p.cloud.data(), sizeof(Pt), &m);
        if (fs != NDT_OK && fs != NDT_ERR_GRID_OVERFLOW) {
          p.end = true;
          p.error = std::string("voxel filter failed: ") + ndt_last_error();
          prepared.put(std::move(p));
          return;
        }
        p.cloud.resize(m);
        p.filter_ms = since(t0);
        if (p.cloud.empty()) {  // :128 -- empty clouds are not kept
          free_prep.put(p.h);
          continue;
        }
        kept++;
        // the scan is the source of this registration and the target of the next one
        if (ndt_set_input_source(p.h, p.cloud.data(), p.cloud.size(), sizeof(Pt)) != NDT_OK ||
            ndt_set_input_target(p.h, p.cloud.data(), p.cloud.size(), sizeof(Pt), 1) != NDT_OK) {
          p.end = true;
          p.error = std::string("preparing the inputs failed: ") + ndt_last_error();
          prepared.put(std::move(p));
          return;
        }
        prepared.put(std::move(p));
      }
    }
    Prepared e;
    e.end = true;
    prepared.put(std::move(e));
  });
  std::thread map_thread([&] {
    for (;;) {
      MapJob j = map_jobs.take();
      if (j.end) return;
      if (!map_error.empty()) continue;
      const auto t0 = clock_type::now();
      int overflowed = 0;
      if (ndt_map_update(map_handle, j.cloud.data(), j.cloud.size(), sizeof(Pt), 1, j.pose.data(), 0.5f, &overflowed) != NDT_OK)
        map_error = ndt_last_error();
      map_ms += since(t0);
    }
  });

  int rc = 0;
  bool producer_done = false;  // the prep thread's end message has been taken: nothing more will come
  Prepared previous;
  for (;;) {
    Prepared cur = prepared.take();
    if (cur.end) {
      producer_done = true;
      if (!cur.error.empty()) {
        std::fprintf(stderr, "%s\n", cur.error.c_str());
        rc = 1;
      }
      break;
    }
    node.t_filter += cur.filter_ms;
    node.loaded++;
    std::printf("Loaded cloud_%d.pcd (%zu points)\n", cur.number, cur.cloud.size());
    if (node.loaded == 1) {  // load_initial_clouds, :64-68
      MapJob j;
      j.cloud = cur.cloud;
      j.pose.assign(kIdentity, kIdentity + 16);
      map_jobs.put(std::move(j));
    } else {
      const auto t0 = clock_type::now();
      float T[16];
      int converged = 0, iterations = 0;
      double probability = 0;
      if (ndt_share_input_target(h, previous.h) != NDT_OK || ndt_share_input_source(h, cur.h) != NDT_OK ||
          ndt_align(h, node.rosbag ? node.pres_transform.data() : nullptr, T, &converged, &iterations, &probability, nullptr, 0) != NDT_OK) {
        std::fprintf(stderr, "registration failed: %s\n", ndt_last_error());
        rc = 1;
        free_prep.put(cur.h);
        break;
      }
      node.t_align += since(t0);
      std::vector<float> map_pose;
      std::string err;
      const bool into_map = node.after_align(h, T, converged, iterations, map_pose, err);
      if (!err.empty()) {
        std::fprintf(stderr, "%s\n", err.c_str());
        rc = 1;
        free_prep.put(cur.h);
        break;
      }
      if (into_map) {
        MapJob j;
        j.cloud = cur.cloud;
        j.pose = map_pose;
        map_jobs.put(std::move(j));
      }
    }
    if (previous.h) free_prep.put(previous.h);  // its grid stays alive for as long as the registration handle shares it
    previous = std::move(cur);
  }
  if (rc && !producer_done) {  // let the prep thread run dry: every handle it may be waiting for goes back to it
    if (previous.h) free_prep.put(previous.h);
    for (;;) {
      Prepared p = prepared.take();
      if (p.end) break;
      free_prep.put(p.h);
    }
  }
  MapJob stop;
  stop.end = true;
  map_jobs.put(std::move(stop));
  prep_thread.join();
  map_thread.join();
  node.t_map += map_ms;
  if (!map_error.empty()) {
    std::fprintf(stderr, "map update failed: %s\n", map_error.c_str());
    rc = 1;
  }
  for (int i = 0; i < kPrep; i++) ndt_destroy(prep[i]);
  return rc;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("usage: map_sequence <pcd_directory> [voxel_leaf_size] [global_map_out.pcd | -] [rosbag | node] [serial | pipeline] [host]\n");
    return 0;
  }
  Node node;
  node.rosbag = argc > 4 && std::strcmp(argv[4], "rosbag") == 0;
  const bool serial = !(argc > 5 && std::strcmp(argv[5], "pipeline") == 0);
  const bool host_clouds = argc > 6 && std::strcmp(argv[6], "host") == 0;  // serial only: every cloud through host buffers
  const float voxel_leaf_size = argc > 2 ? static_cast<float>(std::atof(argv[2])) : (node.rosbag ? 0.3f : 0.5f);  // :44 / rosbag :87
  ndt_handle h = nullptr, map_handle = nullptr;
  CHECK(ndt_create(0, &h));
  if (configure(h)) return 1;
  if (!serial) {
    if (use_partitions()) CHECK(ndt_set_cu_partition(h, 1));
    CHECK(ndt_create(0, &map_handle));
    if (use_partitions()) CHECK(ndt_set_cu_partition(map_handle, 2));
  }

  ndt_pcd_sequence_handle seq = nullptr;
  CHECK(ndt_pcd_sequence_open(argv[1], &seq));
  // a node constructs its objects (and a GPU library loads its code, creates its streams, page-locks its slots) before the
  // first scan arrives: not part of any scan's time
  const auto t_warm = clock_type::now();
  size_t first_poll = 0;
  if (serial && !host_clouds) {
    CHECK(ndt_pcd_sequence_stage(seq, 0));             // the reader uploads every scan as soon as it has parsed it
    CHECK(ndt_pcd_sequence_poll(seq, 0, &first_poll));  // process_new_clouds' first look at the directory: the reads start now
  }
  CHECK(ndt_warm_up(h, 65536));  // (a node knows its sensor: the reference's scans are lidar sweeps of some ten thousand points)
  if (map_handle) CHECK(ndt_warm_up(map_handle, 65536));
  const double warm_ms = since(t_warm);
  const auto t_begin = clock_type::now();
  const int rc = !serial ? run_pipelined(node, seq, voxel_leaf_size, h, map_handle)
                         : host_clouds ? run_serial(node, seq, voxel_leaf_size, h) : run_resident(node, seq, voxel_leaf_size, h, first_poll);
  if (rc) return rc;

  ndt_handle mh = serial ? h : map_handle;
  size_t map_points = 0;
  CHECK(ndt_map_size(mh, &map_points));
  std::printf("\nclouds %zu  registrations %zu (not converged %zu)  global map %zu points\n", node.loaded, node.registered, node.not_converged, map_points);
  for (size_t i = 0; i < node.trajectory.size(); i++) {
    char title[64];
    std::snprintf(title, sizeof(title), "trajectory[%zu]:", i);
    print_matrix(title, node.trajectory[i].data());
  }
  std::printf("start-up (device, code object, page-locked slots; ndt_warm_up): %.1f ms, not in the times below\n", warm_ms);
  std::printf("time: total %.2f ms  (prefilter %.2f, %s %.2f, map update %.2f, waiting for the next file %.2f; %s)\n", since(t_begin), node.t_filter,
              serial ? "set inputs + align" : "take inputs over + align", node.t_align, node.t_map, node.t_wait,
              !serial ? "file reading, prefilter + input preparation and map update overlapped with the registrations"
                      : host_clouds ? "file reading overlapped; clouds through host buffers" : "file reading overlapped; clouds resident in HBM");
  if (argc > 3 && std::strcmp(argv[3], "-") != 0 && map_points) {
    std::vector<Pt> map(map_points);
    CHECK(ndt_map_get(mh, map.data(), sizeof(Pt)));
    CHECK(ndt_pcd_write_xyz(argv[3], map.data(), map.size(), sizeof(Pt), 1));
    std::printf("global map written to %s\n", argv[3]);
  }
  ndt_pcd_sequence_close(seq);
  ndt_destroy(h);
  if (map_handle) ndt_destroy(map_handle);
  return 0;
}
