// ndt_io.hip -- PCD files, numbered scan sequences and PointCloud2-style repacking behind the C-ABI (row N3 of the scope table; host only).
// (split out of the former single C-ABI unit; shared state in ndt_internal.hpp)
#include "ndt_internal.hpp"

#include <mutex>

extern "C" {

// ---- N3: PCD files -------------------------------------------------------------
ndt_status ndt_pcd_read_header(const char* path, size_t* n_points, int* n_fields, int* data_kind) {
  if (!path) return fail(NDT_ERR_INVALID, "null path");
  std::string err;
  try {
    if (ndt::pcd_read_header(path, n_points, n_fields, data_kind, err)) return fail(NDT_ERR_INVALID, err);
  } catch (const std::exception& e) {  // nothing C++ crosses the C boundary
    return fail(NDT_ERR_INVALID, std::string("PCD: ") + e.what());
  }
  return NDT_OK;
}
ndt_status ndt_pcd_read_xyz(const char* path, void* out, size_t capacity_points, size_t stride_bytes, size_t* n_points,
                            int* is_dense) {
  if (!path || (capacity_points && !out) || stride_bytes < 12) return fail(NDT_ERR_INVALID, "bad arguments");
  std::string err;
  try {
    if (ndt::pcd_read_xyz(path, out, capacity_points, stride_bytes, n_points, is_dense, err)) return fail(NDT_ERR_INVALID, err);
  } catch (const std::exception& e) {
    return fail(NDT_ERR_INVALID, std::string("PCD: ") + e.what());
  }
  return NDT_OK;
}
ndt_status ndt_pcd_write_xyz(const char* path, const void* pts, size_t n, size_t stride_bytes, int binary) {
  if (!path || (n && !pts) || stride_bytes < 12) return fail(NDT_ERR_INVALID, "bad arguments");
  std::string err;
  try {
    if (ndt::pcd_write_xyz(path, pts, n, stride_bytes, binary, err)) return fail(NDT_ERR_INVALID, err);
  } catch (const std::exception& e) {
    return fail(NDT_ERR_INVALID, std::string("PCD: ") + e.what());
  }
  return NDT_OK;
}

// ---- N3: numbered scans of a directory, read ahead into page-locked buffers -------------------
struct ndt_pcd_sequence {
  std::unique_ptr<ndt::PcdSequence> seq;
  // ndt_pcd_sequence_stage: every scan also goes up to HBM as soon as it has been read -- by the reading thread, on a copy
  // stream of the sequence's own -- so that the caller finds it there (ndt_pcd_sequence_next_device)
  int stage_device = -1;
  hipStream_t copy_stream = nullptr;
  void* dev[ndt::PcdSequence::kSlots] = {};
  size_t dev_cap[ndt::PcdSequence::kSlots] = {};
  hipEvent_t ready[ndt::PcdSequence::kSlots] = {};
  bool staged[ndt::PcdSequence::kSlots] = {};
  std::mutex mu;  // the copy stream's queue and the slots' device buffers (several reading threads)
  ~ndt_pcd_sequence() {
    seq.reset();  // (joins the reading threads: nothing stages any more)
    if (stage_device >= 0) {
      (void)hipSetDevice(stage_device);
      if (copy_stream) (void)hipStreamSynchronize(copy_stream);
      for (size_t k = 0; k < ndt::PcdSequence::kSlots; k++) {
        if (dev[k]) (void)hipFree(dev[k]);
        if (ready[k]) (void)hipEventDestroy(ready[k]);
      }
      if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
  }
};

namespace {
// Page-locked scan buffers outlive the sequence that asked for them: page-locking a 2 M-point scan's 40 MB costs ~10 ms and
// giving it back ~6 ms, and a sequence takes six of them -- a node that opens a sequence per batch of files (or a benchmark
// per pass: 16 scans of 60 ms) paid more for its buffers than for the registrations.  Blocks go back to a process-wide list
// (at most NDT_PINNED_CACHE_MB megabytes, default 512; 0 = none kept) and are handed out again to whoever asks for that
// much or a little less; whatever the list holds at exit goes with the process.
class PinnedCache {
 public:
  static PinnedCache& instance() {
    static PinnedCache* c = new PinnedCache();  // (never destroyed: the runtime may be gone before static destructors run)
    return *c;
  }
  void* take(size_t bytes) {
    {
      std::lock_guard<std::mutex> g(mu_);
      size_t best = free_.size();
      for (size_t k = 0; k < free_.size(); k++)
        if (free_[k].bytes >= bytes && free_[k].bytes <= bytes + bytes / 2 + (1u << 20) && (best == free_.size() || free_[k].bytes < free_[best].bytes)) best = k;
      if (best != free_.size()) {
        const Block b = free_[best];
        free_.erase(free_.begin() + static_cast<long>(best));
        kept_ -= b.bytes;
        out_.push_back(b);
        return b.p;
      }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(mu_);
    out_.push_back(Block{p, bytes});
    return p;
  }
  void give(void* p) {
    if (!p) return;
    Block b{p, 0};
    {
      std::lock_guard<std::mutex> g(mu_);
      for (size_t k = 0; k < out_.size(); k++)
        if (out_[k].p == p) {
          b = out_[k];
          out_.erase(out_.begin() + static_cast<long>(k));
          break;
        }
      if (b.bytes && kept_ + b.bytes <= cap_) {
        kept_ += b.bytes;
        free_.push_back(b);
        return;
      }
    }
    (void)hipHostFree(p);
  }

 private:
  struct Block {
    void* p;
    size_t bytes;
  };
  PinnedCache() {
    const char* v = std::getenv("NDT_PINNED_CACHE_MB");
    cap_ = static_cast<size_t>(v ? std::max(0, std::atoi(v)) : 512) << 20;
  }
  std::mutex mu_;
  std::vector<Block> free_, out_;
  size_t kept_ = 0, cap_ = 0;
};
}  // namespace

ndt_status ndt_pcd_sequence_open(const char* directory, ndt_pcd_sequence_handle* out) {
  if (!directory || !out) return fail(NDT_ERR_INVALID, "bad arguments");
  // page-locked when a device is there (the scans go straight into ndt_set_input_* / ndt_voxel_grid_filter uploads),
  // pageable otherwise -- reading files needs no GPU
  const bool pinned = usable_devices() > 0;
  auto seq = new ndt_pcd_sequence();
  if (pinned)
    seq->seq.reset(new ndt::PcdSequence(directory, [](size_t bytes) { return PinnedCache::instance().take(bytes); },
                                        [](void* p) { PinnedCache::instance().give(p); }));
  else
    seq->seq.reset(new ndt::PcdSequence(directory, [](size_t bytes) { return std::malloc(bytes); }, [](void* p) { std::free(p); }));
  *out = seq;
  return NDT_OK;
}

ndt_status ndt_pcd_sequence_poll(ndt_pcd_sequence_handle s, size_t loaded_clouds, size_t* n_new_files) {
  if (!s) return fail(NDT_ERR_INVALID, "null");
  std::string err;
  try {
    const int n = s->seq->poll(loaded_clouds, err);
    if (n < 0) return fail(NDT_ERR_INVALID, err);
    if (n_new_files) *n_new_files = static_cast<size_t>(n);
  } catch (const std::exception& e) {
    return fail(NDT_ERR_INVALID, std::string("directory listing: ") + e.what());
  }
  return NDT_OK;
}

ndt_status ndt_pcd_sequence_next(ndt_pcd_sequence_handle s, const void** pts, size_t* n, int* is_dense, int* file_number) {
  if (!s || !pts || !n) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt::PcdSequence::Scan scan;
  std::string err;
  const int rc = s->seq->next(scan, err);
  *pts = scan.pts;
  *n = scan.n;
  if (is_dense) *is_dense = scan.is_dense;
  if (file_number) *file_number = scan.file_number;
  if (rc == 2) return fail(NDT_ERR_INVALID, err);
  return NDT_OK;
}

ndt_status ndt_pcd_sequence_stage(ndt_pcd_sequence_handle s, int device) {
  if (!s || device < 0) return fail(NDT_ERR_INVALID, "bad arguments");
  if (device >= usable_devices()) return fail(NDT_ERR_NO_DEVICE, "no such device");
  if (s->stage_device >= 0) return s->stage_device == device ? NDT_OK : fail(NDT_ERR_INVALID, "the sequence already stages to another device");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
  for (size_t k = 0; k < ndt::PcdSequence::kSlots; k++) HIP_TRY(hipEventCreateWithFlags(&s->ready[k], hipEventDisableTiming));
  s->stage_device = device;
  ndt_pcd_sequence* self = s;
  s->seq->set_on_read([self](int slot, const void* buf, size_t n) {
    std::lock_guard<std::mutex> g(self->mu);
    self->staged[slot] = false;
    if (hipSetDevice(self->stage_device) != hipSuccess) return;
    const size_t bytes = std::max<size_t>(n, 1) * 16;
    if (self->dev_cap[slot] < bytes) {
      if (self->dev[slot]) (void)hipFree(self->dev[slot]);  // (synchronises: rare, the buffers grow to the scans' size once)
      self->dev[slot] = nullptr;
      self->dev_cap[slot] = 0;
      if (hipMalloc(&self->dev[slot], bytes + bytes / 4) != hipSuccess) return;
      self->dev_cap[slot] = bytes + bytes / 4;
    }
    if (n && hipMemcpyAsync(self->dev[slot], buf, n * 16, hipMemcpyHostToDevice, self->copy_stream) != hipSuccess) return;
    if (hipEventRecord(self->ready[slot], self->copy_stream) != hipSuccess) return;
    self->staged[slot] = true;
  });
  return NDT_OK;
}

ndt_status ndt_pcd_sequence_next_device(ndt_pcd_sequence_handle s, const void** d_pts, const void** host_pts, size_t* n, int* is_dense,
                                        int* file_number) {
  if (!s || !d_pts || !n) return fail(NDT_ERR_INVALID, "bad arguments");
  if (s->stage_device < 0) return fail(NDT_ERR_INVALID, "ndt_pcd_sequence_stage has not been called");
  ndt::PcdSequence::Scan scan;
  std::string err;
  const int rc = s->seq->next(scan, err);
  *d_pts = nullptr;
  if (host_pts) *host_pts = scan.pts;
  *n = scan.n;
  if (is_dense) *is_dense = scan.is_dense;
  if (file_number) *file_number = scan.file_number;
  if (rc == 2) return fail(NDT_ERR_INVALID, err);
  if (rc == 0 && scan.pts) {
    bool ok;
    {
      std::lock_guard<std::mutex> g(s->mu);
      ok = s->staged[scan.slot];
    }
    if (!ok) return fail(NDT_ERR_HIP, "staging the scan to the device failed");
    HIP_TRY(hipSetDevice(s->stage_device));
    HIP_TRY(hipEventSynchronize(s->ready[scan.slot]));
    *d_pts = s->dev[scan.slot];
  }
  return NDT_OK;
}

ndt_status ndt_pcd_sequence_next_cloud(ndt_pcd_sequence_handle s, ndt_cloud* cloud, const void** host_pts, size_t* n, int* is_dense,
                                       int* file_number) {
  if (!s || !cloud || !n) return fail(NDT_ERR_INVALID, "bad arguments");
  *cloud = nullptr;
  if (s->stage_device < 0) return fail(NDT_ERR_INVALID, "ndt_pcd_sequence_stage has not been called");
  ndt::PcdSequence::Scan scan;
  std::string err;
  const int rc = s->seq->next(scan, err);
  if (host_pts) *host_pts = scan.pts;
  *n = scan.n;
  if (is_dense) *is_dense = scan.is_dense;
  if (file_number) *file_number = scan.file_number;
  if (rc == 2) return fail(NDT_ERR_INVALID, err);
  if (rc == 0 && scan.pts) {
    bool ok;
    {
      std::lock_guard<std::mutex> g(s->mu);
      ok = s->staged[scan.slot];
    }
    if (!ok) return fail(NDT_ERR_HIP, "staging the scan to the device failed");
    HIP_TRY(hipSetDevice(s->stage_device));
    HIP_TRY(hipEventSynchronize(s->ready[scan.slot]));
    // a view of the staged records with the boxes the reading thread computed (the sequence owns the memory: valid until
    // the next call, like the host records)
    auto c = std::make_shared<DeviceCloud>();
    c->pts.borrow(static_cast<float4*>(s->dev[scan.slot]), scan.n);
    c->n = scan.n;
    c->device = s->stage_device;
    std::memcpy(c->bb_min, scan.bb_min, sizeof(c->bb_min));
    std::memcpy(c->bb_max, scan.bb_max, sizeof(c->bb_max));
    *cloud = new ndt_cloud_s{c};
  }
  return NDT_OK;
}

void ndt_pcd_sequence_close(ndt_pcd_sequence_handle s) { delete s; }

int ndt_host_extract_file_number(const char* file_stem) { return file_stem ? ndt::extract_file_number(file_stem) : -1; }

ndt_status ndt_host_repack_fields(const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                                  void* out_xyz1, int* is_dense) {
  if ((n && (!data || !out_xyz1)) || point_step < 12) return fail(NDT_ERR_INVALID, "bad arguments");
  for (size_t off : {off_x, off_y, off_z})
    if (off + sizeof(float) > point_step) return fail(NDT_ERR_INVALID, "field offset outside the point record");
  const unsigned char* src = static_cast<const unsigned char*>(data);
  float* dst = static_cast<float*>(out_xyz1);
  bool finite = true;
  for (size_t i = 0; i < n; i++) {
    const unsigned char* rec = src + i * point_step;
    float v[3];
    std::memcpy(&v[0], rec + off_x, sizeof(float));  // unaligned-safe
    std::memcpy(&v[1], rec + off_y, sizeof(float));
    std::memcpy(&v[2], rec + off_z, sizeof(float));
    finite = finite && std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]);
    dst[4 * i] = v[0];
    dst[4 * i + 1] = v[1];
    dst[4 * i + 2] = v[2];
    dst[4 * i + 3] = 1.0f;
  }
  if (is_dense) *is_dense = finite ? 1 : 0;
  return NDT_OK;
}

}  // extern "C"
