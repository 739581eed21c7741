// ndt_sequence.cpp -- see ndt_sequence.hpp.
#include "ndt_sequence.hpp"

#include <emmintrin.h>

#include <algorithm>
#include <cerrno>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <filesystem>
#include <utility>

#include "ndt_pcd.hpp"

namespace ndt {

// both bounding boxes of n (x, y, z, .) records: min / max are exact and order-free, so these are the boxes the device's
// box kernel reduces to (ndt_grid_kernels.hip k_bbox16)
static void host_boxes16(const float* p, size_t n, float bb_min[2][3], float bb_max[2][3]) {
  __m128 mn0 = _mm_set1_ps(FLT_MAX), mx0 = _mm_set1_ps(-FLT_MAX), mn1 = mn0, mx1 = mx0;
  const __m128 abs_mask = _mm_castsi128_ps(_mm_set1_epi32(0x7fffffff)), inf = _mm_set1_ps(INFINITY);
  for (size_t i = 0; i < n; i++) {
    const __m128 v = _mm_loadu_ps(p + 4 * i);
    mn0 = _mm_min_ps(v, mn0);  // (min / max hand back their SECOND operand when the first is NaN)
    mx0 = _mm_max_ps(v, mx0);
    if ((_mm_movemask_ps(_mm_cmplt_ps(_mm_and_ps(v, abs_mask), inf)) & 7) == 7) {
      mn1 = _mm_min_ps(v, mn1);
      mx1 = _mm_max_ps(v, mx1);
    }
  }
  alignas(16) float a[4], b[4], c[4], d[4];
  _mm_store_ps(a, mn0); _mm_store_ps(b, mx0); _mm_store_ps(c, mn1); _mm_store_ps(d, mx1);
  for (int k = 0; k < 3; k++) {
    bb_min[0][k] = a[k]; bb_max[0][k] = b[k];
    bb_min[1][k] = c[k]; bb_max[1][k] = d[k];
  }
}

int extract_file_number(const std::string& stem) {
  const size_t underscore = stem.find_last_of('_');
  if (underscore == std::string::npos) return -1;
  // std::stoi: optional white space, optional sign, then at least one digit; trailing text is ignored;
  // no digits or out of int range -> exception -> the node's catch (...) -> -1
  const char* s = stem.c_str() + underscore + 1;
  char* end = nullptr;
  errno = 0;
  const long v = std::strtol(s, &end, 10);
  if (end == s || errno == ERANGE || v < static_cast<long>(INT32_MIN) || v > static_cast<long>(INT32_MAX)) return -1;
  return static_cast<int>(v);
}

PcdSequence::PcdSequence(std::string directory, Alloc alloc, Release release)
    : dir_(std::move(directory)), alloc_(std::move(alloc)), release_(std::move(release)) {}

void PcdSequence::worker() {
  for (;;) {
    std::packaged_task<void()> job;
    {
      std::unique_lock<std::mutex> lk(jobs_mu_);
      jobs_cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
      if (jobs_.empty()) return;  // (stop_, and nothing left to read)
      job = std::move(jobs_.front());
      jobs_.pop_front();
    }
    job();
  }
}

PcdSequence::~PcdSequence() {
  for (std::future<void>& f : inflight_)
    if (f.valid()) f.wait();
  {
    std::lock_guard<std::mutex> g(jobs_mu_);
    stop_ = true;
  }
  jobs_cv_.notify_all();
  for (std::thread& t : workers_) t.join();
  for (Slot& s : slots_)
    if (s.buf) release_(s.buf);
}

int PcdSequence::poll(size_t loaded_clouds, std::string& err) {
  namespace fs = std::filesystem;
  std::error_code ec;
  fs::directory_iterator it(dir_, ec);
  if (ec) {
    err = "cannot read directory " + dir_ + ": " + ec.message();
    return -1;
  }
  std::vector<Entry> fresh;
  const long long threshold = static_cast<long long>(loaded_clouds) + 1;
  for (const fs::directory_entry& e : it) {
    if (e.path().extension() != ".pcd") continue;
    const int number = extract_file_number(e.path().stem().string());
    if (number < threshold) continue;
    // a file already queued and not yet handed out is not queued twice
    bool queued = false;
    for (size_t q = cursor_; q < queue_.size(); q++) queued = queued || queue_[q].path == e.path().string();
    if (!queued) fresh.push_back(Entry{e.path().string(), number});
  }
  std::stable_sort(fresh.begin(), fresh.end(), [](const Entry& a, const Entry& b) { return a.number < b.number; });
  for (Entry& e : fresh) queue_.push_back(std::move(e));
  top_up(cursor_ + kSlots - 1);  // the first files are on their way before anybody asks for them (one slot stays with the scan handed out last)
  return static_cast<int>(fresh.size());
}

void PcdSequence::start_read(size_t index) {
  if (index >= queue_.size()) return;
  Slot* slot = &slots_[index % kSlots];
  const std::string path = queue_[index].path;
  inflight_index_[index % kSlots] = index;
  if (workers_.empty())
    for (size_t w = 0; w + 1 < kSlots; w++) workers_.emplace_back([this] { worker(); });
  std::packaged_task<void()> job([this, slot, path] {
    slot->status = 0;
    slot->err.clear();
    slot->n = 0;
    try {
      size_t n_points = 0;
      int fields = 0, kind = 0;
      if (pcd_read_header(path.c_str(), &n_points, &fields, &kind, slot->err)) {
        slot->status = 2;
        return;
      }
      if (n_points > slot->cap_points || !slot->buf) {
        if (slot->buf) release_(slot->buf);
        slot->cap_points = std::max<size_t>(n_points + n_points / 4, 1024);
        slot->buf = alloc_(slot->cap_points * 16);
        if (!slot->buf) {
          slot->cap_points = 0;
          slot->err = "out of memory for a scan buffer";
          slot->status = 2;
          return;
        }
      }
      if (pcd_read_xyz(path.c_str(), slot->buf, slot->cap_points, 16, &slot->n, &slot->dense, slot->err)) {
        slot->status = 2;
      } else {
        if (on_read_) on_read_(static_cast<int>(slot - slots_), slot->buf, slot->n);  // (the copy to the device runs beside the pass below)
        host_boxes16(static_cast<const float*>(slot->buf), slot->n, slot->bb_min, slot->bb_max);
      }
    } catch (const std::exception& e) {
      slot->err = std::string("PCD: ") + e.what();
      slot->status = 2;
    }
  });
  inflight_[index % kSlots] = job.get_future();
  {
    std::lock_guard<std::mutex> g(jobs_mu_);
    jobs_.push_back(std::move(job));
  }
  jobs_cv_.notify_one();
}

void PcdSequence::top_up(size_t limit) {
  if (read_ahead_ < cursor_) read_ahead_ = cursor_;
  for (; read_ahead_ < queue_.size() && read_ahead_ < limit; read_ahead_++) {
    std::future<void>& f = inflight_[read_ahead_ % kSlots];
    if (f.valid()) f.wait();  // (an older read into this slot: long finished, its scan handed out)
    start_read(read_ahead_);
  }
}

int PcdSequence::next(Scan& out, std::string& err) {
  out = Scan{};
  if (cursor_ >= queue_.size()) return 1;
  // The scan handed out by the previous call is released by this one, so every slot but the one handed out now may hold
  // a read in flight: entries cursor_ .. cursor_ + kSlots - 1 (the first of them is waited for below).
  top_up(cursor_ + kSlots);
  const size_t mine = cursor_++;
  std::future<void>& fm = inflight_[mine % kSlots];
  if (fm.valid()) fm.wait();
  const Slot& slot = slots_[mine % kSlots];
  out.file_number = queue_[mine].number;
  out.path = queue_[mine].path.c_str();
  if (slot.status) {
    err = queue_[mine].path + ": " + slot.err;
    return 2;
  }
  out.pts = slot.buf;
  out.slot = static_cast<int>(mine % kSlots);
  std::memcpy(out.bb_min, slot.bb_min, sizeof(out.bb_min));
  std::memcpy(out.bb_max, slot.bb_max, sizeof(out.bb_max));
  out.n = slot.n;
  out.is_dense = slot.dense;
  return 0;
}

}  // namespace ndt
