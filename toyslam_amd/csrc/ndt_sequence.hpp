// ndt_sequence.hpp -- a directory of numbered PCD scans read in order with the NEXT file parsed in the
// background (host, no PCL, no ROS): the file handling of the reference's mapping node
// (lidar_subscriber/src/ndt_omp_mapping_node.cpp:110-136 process_new_clouds, :231-239
// extract_file_number) behind the C-ABI, row N3 of the scope table.
#pragma once
#include <cstddef>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ndt {

// the number after the last underscore of a file stem, std::stoi semantics, -1 when there is none (:231-239)
int extract_file_number(const std::string& stem);

class PcdSequence {
 public:
  // alloc / release of the kSlots scan buffers (pinned host memory when a GPU is there)
  using Alloc = std::function<void*(size_t)>;
  using Release = std::function<void(void*)>;
  PcdSequence(std::string directory, Alloc alloc, Release release);
  ~PcdSequence();
  PcdSequence(const PcdSequence&) = delete;
  PcdSequence& operator=(const PcdSequence&) = delete;

  // process_new_clouds (:110-136): appends the *.pcd files whose number is >= loaded_clouds + 1, ascending by
  // number.  Returns the number of files appended, -1 when the directory cannot be read (err set).
  int poll(size_t loaded_clouds, std::string& err);
  size_t pending() const { return queue_.size() - cursor_; }

  struct Scan {
    const void* pts = nullptr;  // x, y, z, 1.0f records of 16 bytes; valid until the next call of next()
    size_t n = 0;
    int is_dense = 1;
    int file_number = -1;
    const char* path = nullptr;
    int slot = -1;  // which of the kSlots buffers holds it (on_read's first argument)
    // bounding boxes of the scan, computed by the reading thread: [0] NaN coordinates dropped (pcl::getMinMax3D of a dense
    // cloud), [1] over the finite points only; min > max = no such point
    float bb_min[2][3], bb_max[2][3];
  };
  // called by the reading thread when a file has been read and parsed into slot `slot` (n records at buf): the C-ABI's
  // staging of scans into HBM hangs on it
  using OnRead = std::function<void(int slot, const void* buf, size_t n)>;
  void set_on_read(OnRead f) { on_read_ = std::move(f); }
  // The next queued file (0), nothing queued (1), or a file that cannot be read (2: err set, the file is skipped,
  // as load_and_filter_cloud's nullptr is).  While the caller works on a scan the following kSlots - 1 files are being
  // read and parsed by a pool of kSlots - 1 background threads that live as long as the sequence (a 2M-point scan takes
  // longer to read and parse than to register; a thread per FILE cost the caller ~50 us of thread creation per scan, as
  // much as a node-sized scan's whole prefilter).
  int next(Scan& out, std::string& err);
  static constexpr size_t kSlots = 6;  // (ten: no faster -- the readers share the host's memory bandwidth: 2 M-point sequence 229-237 scans/s on the box that gave six 194-281)

 private:
  struct Entry {
    std::string path;
    int number;
  };
  struct Slot {
    void* buf = nullptr;
    size_t cap_points = 0;
    size_t n = 0;
    int dense = 1;
    int status = 0;
    std::string err;
    float bb_min[2][3], bb_max[2][3];
  };
  void start_read(size_t index);
  void top_up(size_t limit);
  void worker();
  std::vector<std::thread> workers_;
  std::deque<std::packaged_task<void()>> jobs_;
  std::mutex jobs_mu_;
  std::condition_variable jobs_cv_;
  bool stop_ = false;
  OnRead on_read_;
  std::string dir_;
  Alloc alloc_;
  Release release_;
  std::vector<Entry> queue_;
  size_t cursor_ = 0;
  Slot slots_[kSlots];
  std::future<void> inflight_[kSlots];            // the read into slot k ...
  size_t inflight_index_[kSlots] = {};  // ... is of queue entry inflight_index_[k] (valid while the future is)
  size_t read_ahead_ = 0;                         // queue entries below this index have been started
};

}  // namespace ndt
