// ndt_internal.hpp -- state shared by the translation units behind the C-ABI (include/ndt_mi355.h,
// include/gicp_mi355.h): error reporting, the caching device allocator, device clouds / grids, the handle
// itself, and the helpers one unit offers the others.  Not installed; nothing here crosses the C-ABI.
//   ndt_handle.hip : handle lifetime, parameters, results, profiling switches, host-only scalar exports
//   ndt_grid.hip   : cloud upload, bounding boxes, spatial ordering, K1 target grid build (dense / sparse index),
//                    N1 voxel filter, N2 map accumulation, getFitnessScore, calculateScore, grid inspection
//   ndt_eval.hip   : one evaluation (launch path), the persistent evaluation server's host side (mailbox protocol),
//                    ndt_align, ndt_eval*, diagnostics and self-tests
//   ndt_batch.hip  : lock-step batches (ndt_align_batch*), the RCCL communicator (ndt_comm_*), ndt_set_allreduce
//   ndt_io.hip     : PCD files, numbered scan sequences, PointCloud2-style repacking (host)
//   gicp_capi.hip  : the GICP row
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <thread>
#include <cfloat>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <mutex>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "gicp_driver.hpp"
#include "gicp_kernels.hpp"
#include "gicp_mi355.h"
#include "ndt_driver.hpp"
#include "ndt_kernels.hpp"
#include "ndt_pcd.hpp"
#include "ndt_sequence.hpp"
#include "ndt_mi355.h"

namespace ndtc {

extern thread_local std::string g_last_error;

inline ndt_status fail(ndt_status s, const std::string& msg) {
  g_last_error = msg;
  return s;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(NDT_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(_e));          \
  } while (0)

// Caching device allocator: setInputTarget / setInputSource run once per scan in the nodes, and a
// dozen hipMalloc/hipFree pairs per call (~100 us each) would dominate the GPU time of the grid
// build.  Freed blocks go to a free list keyed by (device, STREAM, rounded size class) and are reused
// only by work queued on the same stream: a block may be released while the kernels that use it are
// still in flight (the grid build does not wait for the GPU), and stream order is what makes the
// next user safe.  The calling thread's current stream is set by every API entry (ensure_device).
// The cache is trimmed (hipFree, which synchronises) when it exceeds kPoolTrimBytes.
extern thread_local hipStream_t tls_pool_stream;

class DevPool {
 public:
  static DevPool& instance() {
    static DevPool p;
    return p;
  }
  static size_t size_class(size_t bytes) {
    if (bytes < 512) return 512;
    size_t p2 = 512;
    while (p2 * 2 <= bytes) p2 *= 2;
    const size_t step = p2 / 8;
    return (bytes + step - 1) / step * step;
  }
  using Key = std::tuple<int, hipStream_t, size_t>;
  hipError_t alloc(size_t bytes, void** out, size_t* got) {
    const size_t cls = size_class(bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
      std::lock_guard<std::mutex> g(m_);
      auto it = free_.find(Key(dev, tls_pool_stream, cls));
      if (it != free_.end()) {
        *out = it->second;
        *got = cls;
        cached_ -= cls;
        free_.erase(it);
        return hipSuccess;
      }
    }
    hipError_t e = hipMalloc(out, cls);
    if (e != hipSuccess) {  // retry once with an empty cache
      trim(0);
      e = hipMalloc(out, cls);
    }
    *got = cls;
    return e;
  }
  void release(void* p, size_t cls) {
    if (!p) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool over = false;
    {
      std::lock_guard<std::mutex> g(m_);
      free_.insert(std::make_pair(Key(dev, tls_pool_stream, cls), p));
      cached_ += cls;
      over = cached_ > kPoolTrimBytes;
    }
    if (over) trim(kPoolTrimBytes / 2);
  }
  // Streams that have been destroyed (forget_stream) and not created again since (adopt_stream: a new stream may get an old
  // one's handle value): an ndt_cloud that outlives a handle it was made or read on must neither wait for such a stream nor
  // give its memory to that stream's pool.
  void adopt_stream(hipStream_t st) {
    std::lock_guard<std::mutex> g(m_);
    retired_.erase(st);
  }
  bool retired(hipStream_t st) {
    std::lock_guard<std::mutex> g(m_);
    return retired_.count(st) != 0;
  }
  // blocks cached for a stream that is about to be destroyed: give them back
  void forget_stream(hipStream_t st) {
    std::lock_guard<std::mutex> g(m_);
    retired_.insert(st);
    for (auto it = free_.begin(); it != free_.end();) {
      if (std::get<1>(it->first) == st) {
        (void)hipFree(it->second);
        cached_ -= std::get<2>(it->first);
        it = free_.erase(it);
      } else {
        ++it;
      }
    }
  }
  void trim(size_t keep) {
    std::lock_guard<std::mutex> g(m_);
    for (auto it = free_.begin(); it != free_.end() && cached_ > keep;) {
      (void)hipFree(it->second);
      cached_ -= std::get<2>(it->first);
      it = free_.erase(it);
    }
  }

 private:
  static constexpr size_t kPoolTrimBytes = size_t(16) << 30;
  std::mutex m_;
  std::multimap<Key, void*> free_;
  std::set<hipStream_t> retired_;
  size_t cached_ = 0;
};

// grow-only device buffer on top of the pool
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;        // elements the caller may use
  size_t cls_bytes = 0;  // pool size class actually held
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  void release() {
    if (p && cls_bytes) DevPool::instance().release(p, cls_bytes);  // (cls_bytes == 0: borrowed memory, not ours to free)
    p = nullptr;
    cap = 0;
    cls_bytes = 0;
  }
  // refer to n elements of memory somebody else owns and keeps alive (clouds handed over by reference)
  void borrow(T* ptr, size_t n) {
    release();
    p = ptr;
    cap = n;
    cls_bytes = 0;
  }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(cap, o.cap);
    std::swap(cls_bytes, o.cls_bytes);
  }
  hipError_t reserve(size_t n) {
    if (n <= cap && p && cls_bytes) return hipSuccess;
    release();
    void* q = nullptr;
    size_t got = 0;
    hipError_t e = DevPool::instance().alloc(std::max<size_t>(n, 1) * sizeof(T), &q, &got);
    if (e == hipSuccess) {
      p = static_cast<T*>(q);
      cls_bytes = got;
      cap = got / sizeof(T);
    }
    return e;
  }
};

struct DeviceCloud {
  DevBuf<float4> pts;     // caller's order (align's output cloud keeps it)
  DevBuf<float4> sorted;  // lattice-cell order, what the derivative kernels read
  size_t n = 0;
  // bounding boxes computed during the upload (k_repack_bbox): [0] over the non-NaN points (the
  // is_dense rule of getMinMax3D), [1] over the finite points (!is_dense); min > max = no such point
  float bb_min[2][3] = {{FLT_MAX, FLT_MAX, FLT_MAX}, {FLT_MAX, FLT_MAX, FLT_MAX}};
  float bb_max[2][3] = {{-FLT_MAX, -FLT_MAX, -FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
  size_t n_sorted = 0;    // finite points only
  std::vector<size_t> scan_counts;  // batch uploads: finite points of each scan ...
  std::vector<size_t> scan_starts;  // ... and where its ordered segment starts in `sorted`
  const float4* k2_pts() const { return n_sorted ? sorted.p : pts.p; }
  int k2_n() const { return static_cast<int>(n_sorted ? n_sorted : n); }
  // An ndt_cloud (a cloud the caller holds as an object and hands to several consumers): the stream it was made on -- its
  // memory goes back to THAT stream's pool, whichever thread or handle drops the last reference -- and the other streams
  // it has been read on (waited for before the memory is given back).  Null for the handles' own uploads.
  int device = -1;
  hipStream_t made_on = nullptr;
  std::vector<hipStream_t> used_on;
  std::shared_ptr<DeviceCloud> parent;  // a view of another cloud's points (pts borrowed) with an ordered copy of its own
  ~DeviceCloud() {
    if (!made_on) return;
    (void)hipSetDevice(device);
    DevPool& pool = DevPool::instance();
    for (hipStream_t s : used_on)
      if (s != made_on && !pool.retired(s)) (void)hipStreamSynchronize(s);  // (a destroyed stream's work is over: its handle waited)
    const hipStream_t keep = tls_pool_stream;
    if (pool.retired(made_on)) {  // the handle that made it is gone: to the default pool, behind a device-wide wait
      (void)hipDeviceSynchronize();
      tls_pool_stream = nullptr;
    } else {
      tls_pool_stream = made_on;
    }
    pts.release();
    sorted.release();
    tls_pool_stream = keep;
  }
};


// An enqueued voxel filter: where its count and the rows of its result's boxes arrive (page-locked), what the host decided
struct FilterPending {
  float* rows = nullptr;     // [64][12] per-block rows of the result's bounding boxes
  unsigned* tot = nullptr;   // [3]: points binned, voxels, -
  size_t n_max = 0, fixed_n = 0;
  bool from_device = false;  // the count is tot[1] (else fixed_n: empty input, or the input copied through)
  bool overflow = false;
};

// Immutable once built (shared between cloned handles).
struct DeviceGrid {
  ndt::GridGeom geom{};
  float resolution = 0;
  int min_pts = 6;
  double eig_ratio = 0.01;
  bool empty = true;
  size_t n_leaves = 0, n_cand = 0, n_valid = 0;
  std::shared_ptr<DeviceCloud> target;  // kept for the dump pass
  DevBuf<int> lut;
  DevBuf<ndt::VoxelRec> recs;
  DevBuf<ndt::VoxelSide> centroids;  // per record: voxel centroid (KDTREE search) + f64 inverse covariance
  DevBuf<int> leaf_cell, leaf_count, leaf_rec, sorted_idx;
  DevBuf<unsigned> leaf_start;
  size_t n_sorted = 0;  // target points that landed in a voxel (finite ones)
  DevBuf<unsigned> counts;      // device copy of {n_sorted, n_leaves, n_cand, n_valid}
  bool counts_known = true;     // host copies above are current (grid_counts() fetches them lazily)
  // getFitnessScore's nearest-neighbour search: cell -> leaf ordinal (or -1), built on first use
  std::mutex fit_mu;
  DevBuf<uint2> cell_range;  // per cell: its segment of cell_pts (count 0 = empty)
  DevBuf<float4> cell_pts;  // the target points in cell order (ndt_search.hpp scans them)
  DevBuf<int> row_any;      // per x-row of cells: occupied or not
  bool have_cell2leaf = false;
  // bucket-form build: records still numbered the way k1_finalize numbers them (slot = segment start / min_pts: gaps,
  // bucket by bucket); maybe_compact_records makes them dense and cell-ordered once the grid is seen to be reused
  bool compact_pending = false;
  int n_registrations = 0;
  // bucket-form build (ndt_kernels.hip): kept until the leaf arrays have been written (grid_counts)
  bool leaves_pending = false;
  ndt::GridBuildPlan plan{};
  DevBuf<float4> bpts;
  bool index_form = false;  // (NDT_K1_INDEX=1: bpts holds point indices)
  DevBuf<unsigned> bucket_base;
  ndt::GridView view() const {
    ndt::GridView v;
    v.lut = lut.p;
    v.recs = recs.p;
    v.centroids = centroids.p;
    v.g = geom;
    return v;
  }
};


}  // namespace ndtc
using namespace ndtc;

// the C-ABI's ndt_cloud
struct ndt_cloud_s {
  std::shared_ptr<DeviceCloud> c;
};

struct ndt_context;
namespace ndtc {
void comm_release(ndt_context* h);  // ndt_batch.hip
}

struct ndt_context {
  int device = 0;
  bool device_ready = false;
  hipStream_t stream = nullptr;
  bool stream_masked[3] = {false, false, false};
  hipStream_t partition_stream[3] = {nullptr, nullptr, nullptr};  // the streams this handle has had, by CU partition (ndt_set_cu_partition switches, nothing is destroyed before the handle is)
  // parameters (ctor defaults ndt_omp_impl.hpp:47-76, voxel_grid_covariance_omp.h:208-223)
  float resolution = 1.0f;
  double step_size = 0.1, outlier_ratio = 0.55, trans_eps = 0.1;
  int max_iter = 35, search = NDT_DIRECT7, num_threads = 1, min_pts = 6;
  double eig_ratio = 0.01;
  // inputs
  std::shared_ptr<DeviceCloud> target, source;
  int target_dense = 1;
  std::shared_ptr<DeviceGrid> grid;
  // scratch
  DevBuf<double> partials;
  DevBuf<unsigned> ticket;  // zero between launches (reset by the last block of the fused kernel)
  DevBuf<double> batch_out;
  DevBuf<ndt::ScanDesc> descs;
  void* batch_pinned = nullptr;  // pinned staging: [n_scans] ScanDesc + [3 n_scans] int
  size_t batch_pinned_bytes = 0;
  DevBuf<float4> out_cloud;
  void* out_pinned = nullptr;  // page-locked staging of the aligned cloud on its way to the caller
  float4* server_out_host = nullptr;  // set by ndt_align before the server starts: the server writes the cloud there too
  bool server_wrote_host = false;
  size_t out_pinned_bytes = 0;
  DevBuf<unsigned char> staging;
  double* host_result = nullptr;  // pinned, kEvalStride doubles (+ batch rows)
  float* bbox_rows = nullptr;     // pinned, per-block bounding-box rows of the last upload (k_repack_bbox)
  unsigned long long* bbox_tagged = nullptr;  // pinned: the same rows as self-validating words, polled (clouds by reference)
  unsigned bbox_tag = 0;
  // small host clouds (the mapping nodes' 16 k-point scans): repacked to float4 and bounded ON THE HOST into one of these
  // page-locked slots and DMA'd from there -- no repack kernel, no wait for the device (upload_cloud)
  static constexpr int kStageSlots = 4;
  static constexpr size_t kStageSlotPoints = 65536;
  float* stage_host[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_done[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};  // the slot's last DMA has been read
  int stage_next = 0;
  double* host_pub = nullptr;     // pinned, tagged publication row of the single-scan paths (ndt_kernels.hip publish_row_tagged)
  size_t host_result_rows = 0;
  unsigned long long eval_seq = 0;
  double t_launch = 0, t_wait = 0, t_solver = 0, t_fill = 0, t_gap = 0;  // NDT_TIMING=1 accounting (seconds; t_gap: NDT_TIMING=2)
  // results
  float final_T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  int converged = 0, nr_iterations = 0;
  double trans_probability = 0;
  int n_evals = 0, n_hess = 0;
  double mean_neighbors = 0;
  size_t out_n = 0;
  // persistent evaluation server (single-scan align)
  bool server_running = false;
  int server_blocks = 0;  // grid of the running server: min(16, blocks) part rows come back per evaluation
  // N2: accumulated global map (dense float4, HBM resident)
  DevBuf<float4> map_pts;   // the map, with room behind it: the next scan is transformed straight into its tail
  DevBuf<float4> map_alt;   // where the next filter pass writes (the two swap roles)
  size_t map_n = 0;
  int map_dense = 1;
  DeviceCloud map_boxes;    // bounding boxes of the map as the last filter pass left it (bb_min / bb_max only)
  bool map_boxes_known = false;
  // The map lives on a stream of its own: an update is queued there and NOT waited for -- the next registration does not
  // read the map -- until somebody needs the map or its size (map_complete): the next update, ndt_map_size / _get.
  hipStream_t map_stream = nullptr;
  hipEvent_t map_ready = nullptr;       // recorded on the handle's stream: the scan the update reads is complete
  bool map_pending = false;
  FilterPending map_filter;
  std::shared_ptr<DeviceCloud> map_scan;  // the scan a queued update reads (kept until the update has been waited for)
  float* filter_slots = nullptr;        // page-locked: [3] x (64 x 12 rows + count words): slot 0 N1, slot 1 the map, slot 2 a begun N1
  // ndt_cloud_voxel_filter_begin / _end: one prefilter queued on a stream of its own (beside a registration on the handle's)
  hipStream_t filter_stream = nullptr;
  bool n1_pending = false;
  FilterPending n1_filter;
  std::shared_ptr<DeviceCloud> n1_in, n1_out;
  int voxel_index = 0;              // ndt_set_voxel_index: 0 automatic, 1 dense table, 2 sparse (sorted build + hash look-up)
  bool index_only = false;  // GICP's point index: cells and their point lists only, no per-voxel statistics
  int persistent = -1;  // -1 = default (NDT_PERSISTENT / on), 0 = launch per evaluation, 1 = server
  // Two command mailboxes, used by alternate server instances: a server told to finish (transform +
  // exit) is not waited for, and the next instance's first command must not overwrite the line the
  // old one may still be reading.
  void* server_host_mbs = nullptr;  // 2 command mailboxes the host writes: host-visible device memory (large BAR) or pinned host memory
  bool server_mbs_on_device = false;
  void* server_host_mb = nullptr;   // the running (or next) instance's mailbox
  int server_flip = 0;
  DevBuf<unsigned char> server_dev_mb;
  DevBuf<unsigned> server_counter;  // two sets of kServerCounterWords, used alternately (server_start)
  int server_counter_set = 0;
  DevBuf<unsigned long long> server_dbg;  // diagnostics only (ndt_diag_server_roundtrip)
  bool server_want_dbg = false;
  int cu_count = 0;      // CUs this handle's stream may use (its partition's)
  int cu_total = 0;      // CUs of the device
  int cu_partition = 0;  // ndt_set_cu_partition: 0 whole device, 1 registration partition, 2 side partition
  // live kernel timing (HIP events on `stream`)
  bool profiling = false;       // mode 1: one launch per evaluation, an event pair around each
  bool profile_server = false;  // mode 2: the persistent kernel of each registration between one event pair
  bool server_timed = false;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
  static constexpr int kProfSlots = 7;  // 0-2 evaluation kinds / lock-step kernels, 3 server launch, 4-6 exchange step (ndt_mi355.h)
  long long prof_n[kProfSlots] = {0, 0, 0, 0, 0, 0, 0};
  double prof_ms[kProfSlots] = {0, 0, 0, 0, 0, 0, 0};
  // collective hook
  ndt_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int allreduce_on_device = 0;
  // native RCCL communicator (ndt_comm_*, ndt_batch.hip): sharded lock-step batches, point-sharded scans
  void* comm = nullptr;  // ncclComm_t
  int comm_rank = 0, comm_world = 1;
  long long comm_collectives = 0;
  int batch_lock_steps = 0;  // lock-steps of the last ndt_align_batch*
  int batch_groups = 1;      // independent lock-step groups the last batch ran as
  bool is_batch_worker = false;
  int batch_groups_wanted = 0;  // ndt_set_batch_groups: 0 = automatic
  std::vector<ndt_context*> batch_workers;  // worker handles of those groups (own stream and staging each; grid shared)

  ~ndt_context() {
    for (ndt_context* w : batch_workers) delete w;
    batch_workers.clear();
    ndtc::comm_release(this);
    if (stream) {  // nothing of this handle may still be running when its buffers go back to the pool
      (void)hipSetDevice(device);
      (void)hipStreamSynchronize(stream);
      tls_pool_stream = stream;
    }
    if (map_stream) {  // the map's buffers belong to the map stream's pool
      (void)hipStreamSynchronize(map_stream);
      const hipStream_t keep = tls_pool_stream;
      tls_pool_stream = map_stream;
      map_pts.release();
      map_alt.release();
      tls_pool_stream = keep;
      DevPool::instance().forget_stream(map_stream);
      (void)hipStreamDestroy(map_stream);
      if (map_ready) (void)hipEventDestroy(map_ready);
    }
    if (filter_stream) {
      (void)hipStreamSynchronize(filter_stream);
      n1_in.reset();
      n1_out.reset();
      DevPool::instance().forget_stream(filter_stream);
      (void)hipStreamDestroy(filter_stream);
    }
    if (filter_slots) (void)hipHostFree(filter_slots);
    release_buffers();
    if (host_result) (void)hipHostFree(host_result);
    if (host_pub) (void)hipHostFree(host_pub);
    if (out_pinned) (void)hipHostFree(out_pinned);
    if (bbox_rows) (void)hipHostFree(bbox_rows);
    if (bbox_tagged) (void)hipHostFree(bbox_tagged);
    for (int k = 0; k < kStageSlots; k++) {
      if (stage_host[k]) (void)hipHostFree(stage_host[k]);
      if (stage_done[k]) (void)hipEventDestroy(stage_done[k]);
    }
    if (server_host_mbs) (void)(server_mbs_on_device ? hipFree(server_host_mbs) : hipHostFree(server_host_mbs));
    if (batch_pinned) (void)hipHostFree(batch_pinned);
    if (ev_a) (void)hipEventDestroy(ev_a);
    if (ev_b) (void)hipEventDestroy(ev_b);
    if (ev_c) (void)hipEventDestroy(ev_c);
    if (ev_d) (void)hipEventDestroy(ev_d);
    for (hipStream_t& ps : partition_stream) {
      if (ps && ps != stream) {
        DevPool::instance().forget_stream(ps);
        (void)hipStreamDestroy(ps);
      }
      ps = nullptr;
    }
    if (stream) {
      DevPool::instance().forget_stream(stream);
      (void)hipStreamDestroy(stream);
    }
    tls_pool_stream = nullptr;
  }
  void release_buffers();
};

namespace ndtc {
// ---- ndt_handle.hip
int usable_devices();
int side_cus();
ndt_status ensure_device(ndt_context* h);
ndt_status ensure_host_rows(ndt_context* h, size_t rows);
ndt::SolverParams solver_params(const ndt_context* h);
// ---- ndt_grid.hip
ndt_status upload_cloud(ndt_context* h, const void* pts, size_t n, size_t stride, bool on_device,
                        std::shared_ptr<DeviceCloud>& out, bool by_reference = false);
struct BBox {
  float mn[3], mx[3];
};
BBox bbox_of(const DeviceCloud& c, int dense);
ndt_status bbox_compute(ndt_context* h, const float4* d_pts, int n, int dense, BBox& out);
ndt_status order_range(ndt_context* h, const float4* d_pts, size_t n, float pitch, float4* d_out, size_t* n_out,
                       const BBox* known_bbox = nullptr);
ndt_status order_batch(ndt_context* h, DeviceCloud* c, const size_t* offsets, size_t n_scans);
ndt_status order_cloud(ndt_context* h, DeviceCloud* c, const size_t* offsets, size_t n_scans);
ndt_status build_grid(ndt_context* h);
ndt_status maybe_compact_records(ndt_context* h, bool eager);
ndt_status grid_counts(ndt_context* h, DeviceGrid* g);
ndt_status ensure_cell2leaf(ndt_context* h, DeviceGrid* g);
float index_slack(const DeviceGrid* g);
ndt_status download_records(ndt_context* h, const float4* d_src, size_t n, void* out, size_t out_stride);
void fill_point_index(const DeviceGrid* g, ndt::PointIndex& ix);
ndt_status fitness_impl(ndt_context* h, const float4* d_src, int n, const float* T_colmajor, double max_range, double* fitness);
ndt_status filter_slots(ndt_handle h, int which, FilterPending& P);
ndt_status voxel_filter_enqueue(ndt_handle h, hipStream_t st, const float4* d_in, size_t n, int is_dense, float leaf, float4* d_out,
                                const BBox& bb, FilterPending& P);
void voxel_filter_finish(const FilterPending& P, size_t* n_out, DeviceCloud* boxes);
// out_boxes: the result's bounding boxes as DeviceCloud keeps them ([2][3] min, [2][3] max), or null
ndt_status voxel_filter_device(ndt_handle h, const float4* d_in, size_t n, int is_dense, float leaf, float4* d_out,
                               size_t* n_out, bool* overflow, const BBox* known_bbox = nullptr, DeviceCloud* out_boxes = nullptr);
// ---- ndt_eval.hip
void colmajor_to_T12(const float* m, float* T12);
float kd_radius2(float resolution);
void fill_eval_params(const ndt::EvalRequest& rq, const ndt::Gauss& gs, float kd_r2, ndt::EvalParams& P);
void fill_h64_params(const ndt::EvalRequest& rq, const ndt::Gauss& gs, float kd_r2, ndt::Hess64Params& P);
void unpack_row(const double* row, bool have_h, ndt::EvalResult& r, double* nn);
ndt_status check_ready(ndt_context* h);
ndt_status evaluate_single(ndt_context* h, const ndt::EvalRequest& rq, ndt::EvalResult& res, double* nn_total);
ndt_status server_stop(ndt_context* h);
ndt_status server_start(ndt_context* h);
void server_finish(ndt_context* h, const float* T_colmajor);
ndt_status server_evaluate(ndt_context* h, const ndt::EvalRequest& rq, const ndt::Gauss& gs, ndt::EvalResult& res,
                           double* nn_total, bool* served);
bool server_enabled();
// ---- ndt_batch.hip
ndt_status comm_allreduce(ndt_context* h, double* d_buf, size_t n_doubles);
void server_mark(ndt_context* h, bool running);

// tagged publication row (ndt_kernels.hip publish_row_tagged): 64 words, each (half of a value << 32) |
// low 32 bits of the sequence number; complete when every word carries the tag
inline bool pub_ready(const double* pub, unsigned long long seq) {
  const volatile unsigned long long* w = reinterpret_cast<const volatile unsigned long long*>(pub);
  const unsigned tag = static_cast<unsigned>(seq);
  for (int i = ndt::kPublishSlots - 1; i >= 0; i--)
    if (static_cast<unsigned>(w[i]) != tag) return false;
  return true;
}
inline void pub_gather(const double* pub, double* row) {
  std::atomic_thread_fence(std::memory_order_acquire);
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(pub);
  for (int k = 0; k < ndt::kEvalStride; k++) {
    const unsigned long long bits = (w[2 * k] >> 32) | ((w[2 * k + 1] >> 32) << 32);
    std::memcpy(&row[k], &bits, sizeof(double));
  }
}
}  // namespace ndtc
