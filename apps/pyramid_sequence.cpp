// apps/pyramid_sequence.cpp -- multi-resolution NDT over a streamed sequence of PCD scans (BASELINE configs[4]) against the
// C-ABI alone: no Python, no PCL.  Three resident voxel grids over one target (2.0 -> 1.0 -> 0.5 m by default), one handle
// per grid; every numbered scan of a directory (the mapping node's cloud_N.pcd convention, ndt_omp_mapping_node.cpp:110-136,
// 231-239) is registered coarse to fine, each level's final transformation being the next level's initial guess -- the
// align(output, guess) chaining of ndt_rosbag_mapping_node.cpp:120-144 (pres_transform, :63,130) applied across levels.
//
// Pipeline (what toyslam_amd/pyramid.py sequences from Python, here in C++):
//   reader threads   ndt_pcd_sequence_*: files k+1 ... read and parsed ahead into page-locked buffers
//   upload thread    scan k+1: host -> HBM + spatial ordering on a DONOR handle with a stream of its own (with
//                    NDT_PIPELINE_PARTITION=1 on the side partition of the CUs, ndt_set_cu_partition(h, 2))
//   this thread      scan k: the three level handles take the donor's cloud over (ndt_share_input_source: one upload
//                    serves all levels) and register it level by level
//
//   pyramid_sequence <target.pcd> <pcd_directory> [levels, e.g. 2.0,1.0,0.5] [serial]
// "serial": read -> upload -> register strictly one after the other (same results, bit for bit).
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

using clock_type = std::chrono::steady_clock;
static double ms_since(clock_type::time_point a) { return std::chrono::duration<double, std::milli>(clock_type::now() - a).count(); }

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

struct Upload {
  ndt_handle donor = nullptr;
  int number = -1;
  size_t points = 0;
  double upload_ms = 0;
  bool end = false;
  std::string error;
};

static void print_matrix(const float* T) {
  for (int r = 0; r < 4; r++) std::printf("  %.9g %.9g %.9g %.9g\n", T[r], T[4 + r], T[8 + r], T[12 + r]);
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::printf("usage: pyramid_sequence <target.pcd> <pcd_directory> [levels, e.g. 2.0,1.0,0.5] [serial]\n");
    return 0;
  }
  std::vector<float> resolutions;
  {
    std::string spec = argc > 3 ? argv[3] : "2.0,1.0,0.5";
    size_t pos = 0;
    while (pos < spec.size()) {
      const size_t comma = spec.find(',', pos);
      const std::string tok = spec.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
      const float r = static_cast<float>(std::atof(tok.c_str()));
      if (!(r > 0)) {
        std::fprintf(stderr, "bad resolution '%s'\n", tok.c_str());
        return 1;
      }
      resolutions.push_back(r);
      if (comma == std::string::npos) break;
      pos = comma + 1;
    }
  }
  const bool serial = argc > 4 && std::strcmp(argv[4], "serial") == 0;
  // NDT_PIPELINE_PARTITION=1: level handles on the registration partition of the CUs, donors on the side partition
  // (ndt_set_cu_partition).  Off by default: measured slower on this workload (NOTES.md, round 3).
  const bool partitions = std::getenv("NDT_PIPELINE_PARTITION") && std::atoi(std::getenv("NDT_PIPELINE_PARTITION")) != 0;

  // ---- the target and its grids (resident for the whole sequence)
  size_t n_tgt = 0;
  int tgt_dense = 1;
  int n_fields = 0, data_kind = 0;
  CHECK(ndt_pcd_read_header(argv[1], &n_tgt, &n_fields, &data_kind));
  std::vector<float> tgt_buf(4 * n_tgt);
  float* tgt = tgt_buf.data();
  CHECK(ndt_pcd_read_xyz(argv[1], tgt, n_tgt, 16, &n_tgt, &tgt_dense));
  std::vector<ndt_handle> level(resolutions.size(), nullptr);
  const auto t_grids = clock_type::now();
  for (size_t l = 0; l < level.size(); l++) {
    CHECK(ndt_create(0, &level[l]));
    if (partitions) CHECK(ndt_set_cu_partition(level[l], 1));
    CHECK(ndt_set_resolution(level[l], resolutions[l]));
    CHECK(ndt_set_neighborhood_search_method(level[l], NDT_DIRECT7));
    CHECK(ndt_set_transformation_epsilon(level[l], 0.01));
    CHECK(ndt_set_maximum_iterations(level[l], 35));
    CHECK(ndt_set_step_size(level[l], 0.1));
    CHECK(ndt_set_input_target(level[l], tgt, n_tgt, 16, tgt_dense));
  }
  std::vector<float>().swap(tgt_buf);
  std::printf("target %zu points, %zu grids built in %.1f ms\n", n_tgt, level.size(), ms_since(t_grids));

  ndt_handle donors[2] = {nullptr, nullptr};
  for (auto& d : donors) {
    CHECK(ndt_create(0, &d));
    if (partitions) CHECK(ndt_set_cu_partition(d, 2));
  }
  ndt_pcd_sequence_handle seq = nullptr;
  CHECK(ndt_pcd_sequence_open(argv[2], &seq));
  size_t n_files = 0;
  CHECK(ndt_pcd_sequence_poll(seq, 0, &n_files));

  Channel<ndt_handle> free_donors;
  Channel<Upload> ready;
  free_donors.put(donors[0]);
  free_donors.put(donors[1]);
  auto upload_next = [&](Upload& u) -> bool {  // false: nothing left (or an error, in u.error)
    for (;;) {
      const void* raw = nullptr;
      size_t n = 0;
      int dense = 1, number = -1;
      const ndt_status s = ndt_pcd_sequence_next(seq, &raw, &n, &dense, &number);
      if (s != NDT_OK) {
        std::fprintf(stderr, "skipped: %s\n", ndt_last_error());
        continue;
      }
      if (!raw) return false;
      u.donor = free_donors.take();
      const auto t0 = clock_type::now();
      if (ndt_set_input_source(u.donor, raw, n, 16) != NDT_OK) {  // returns when the page-locked buffer is free again
        u.error = ndt_last_error();
        return false;
      }
      u.upload_ms = ms_since(t0);
      u.number = number;
      u.points = n;
      return true;
    }
  };
  std::thread uploader;
  if (!serial) {
    uploader = std::thread([&] {
      for (;;) {
        Upload u;
        if (!upload_next(u)) {
          u.end = true;
          ready.put(std::move(u));
          return;
        }
        ready.put(std::move(u));
      }
    });
  }

  int rc = 0;
  bool producer_done = false;  // the uploader's end message has been taken: nothing more will come
  size_t n_scans = 0;
  double wait_ms = 0, upload_ms = 0;
  std::vector<double> level_ms(level.size(), 0.0);
  long long evals = 0;
  const auto t_begin = clock_type::now();
  for (;;) {
    Upload u;
    const auto tw = clock_type::now();
    if (serial) {
      if (!upload_next(u)) u.end = true;
    } else {
      u = ready.take();
    }
    if (u.end) {
      producer_done = true;
      if (!u.error.empty()) {
        std::fprintf(stderr, "upload failed: %s\n", u.error.c_str());
        rc = 1;
      }
      break;
    }
    wait_ms += ms_since(tw) - (serial ? u.upload_ms : 0.0);
    upload_ms += u.upload_ms;
    std::printf("scan cloud_%d.pcd (%zu points)\n", u.number, u.points);
    float T[16];
    const float* guess = nullptr;
    for (size_t l = 0; l < level.size() && !rc; l++) {
      const auto t0 = clock_type::now();
      int converged = 0, iterations = 0, n_evals = 0, n_hess = 0;
      double probability = 0, nn = 0;
      if (ndt_share_input_source(level[l], u.donor) != NDT_OK ||
          ndt_align(level[l], guess, T, &converged, &iterations, &probability, nullptr, 0) != NDT_OK ||
          ndt_get_stats(level[l], &n_evals, &n_hess, &nn) != NDT_OK) {
        std::fprintf(stderr, "registration failed: %s\n", ndt_last_error());
        rc = 1;
        break;
      }
      level_ms[l] += ms_since(t0);
      evals += n_evals + n_hess;
      std::printf("level %g: iterations %d converged %d\n", resolutions[l], iterations, converged);
      print_matrix(T);
      guess = T;  // this level's result is the next level's initial guess
    }
    free_donors.put(u.donor);
    if (rc) break;
    n_scans++;
  }
  const double total_ms = ms_since(t_begin);
  if (uploader.joinable()) {
    if (rc && !producer_done) {  // let the uploader run dry: every donor it uploads into comes straight back to it
      for (;;) {
        Upload u = ready.take();
        if (u.end) break;
        free_donors.put(u.donor);
      }
    }
    uploader.join();
  }
  if (!rc && n_scans) {
    std::printf("\nscans %zu of %zu files  %.1f scans/s  (%.2f ms per scan: waiting for the upload %.2f, levels", n_scans, n_files,
                n_scans / (total_ms * 1e-3), total_ms / n_scans, wait_ms / n_scans);
    for (size_t l = 0; l < level.size(); l++) std::printf(" %.2f", level_ms[l] / n_scans);
    std::printf("; upload itself %.2f ms per scan%s; %.1f evaluations per scan)\n", upload_ms / n_scans, serial ? "" : ", overlapped",
                static_cast<double>(evals) / n_scans);
  }
  ndt_pcd_sequence_close(seq);
  for (auto h : level) ndt_destroy(h);
  for (auto d : donors) ndt_destroy(d);
  return rc;
}
