// gicp_capi.hip -- C-ABI glue of the GICP row (include/gicp_mi355.h).  It reuses the NDT units' device pool,
// cloud upload and K1 grid build (ndt_internal.hpp) -- the
// voxel index K1 produces over a cloud is the search structure of GICP's nearest-neighbour queries.
//
// A gicp_context owns two ndt_contexts used purely as index holders: `tgt` (the GICP target as its
// target cloud + grid) and `src` (the GICP *source* as ITS target cloud + grid, for the source's own
// k-NN covariances).  All GICP kernels run on tgt's stream.
#include "ndt_internal.hpp"

namespace {

struct GicpDevice;

}  // namespace

struct gicp_context {
  ndt_context tgt, src;
  gicp::Params prm;
  bool have_tgt = false, have_src = false;
  bool have_cov_tgt = false, have_cov_src = false;
  bool user_cov_tgt = false, user_cov_src = false;  // supplied through gicp_set_*_covariances
  DevBuf<double> cov_tgt, cov_src;  // [n][6]
  DevBuf<float4> output;            // the source moved by the guess
  DevBuf<int> corr;
  DevBuf<float> maha;
  DevBuf<double> partials;
  DevBuf<unsigned> counter;
  DevBuf<float4> out_cloud;
  DevBuf<int> nn_idx;
  DevBuf<float> nn_d2;
  double* host_pub = nullptr;  // pinned tagged publication row
  void* out_pinned = nullptr;  // page-locked staging of the aligned cloud
  // persistent objective server (one launch per BFGS run): two alternating command mailboxes in host-visible device
  // memory, shard counters, the pinned part rows
  void* srv_mbs = nullptr;
  int srv_flip = 0;
  bool srv_tried = false;
  double* srv_rows = nullptr;
  DevBuf<unsigned> srv_counter;
  int srv_blocks = 0;
  bool src_cov_pending = false;
  float hint_leaf[2] = {0.f, 0.f};  // index leaf the previous target / source ended up with, and its size
  size_t hint_n[2] = {0, 0};
  hipEvent_t ev_src = nullptr;  // the source's covariance pass (on the source index's stream) -> the main stream
  size_t out_pinned_bytes = 0;
  unsigned long long seq = 0;
  float guess_rm[16];          // guess of the current align / step, row-major
  bool step_ready = false;
  // results
  float final_T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // column-major
  int converged = 0, nr_iterations = 0, n_f = 0, n_df = 0, n_fdf = 0, correspondences = 0;

  ~gicp_context() {
    if (tgt.device_ready) {
      (void)hipSetDevice(tgt.device);
      (void)hipStreamSynchronize(tgt.stream);
      tls_pool_stream = tgt.stream;
    }
    cov_tgt.release(); cov_src.release(); output.release(); corr.release(); maha.release(); partials.release();
    counter.release(); out_cloud.release(); nn_idx.release(); nn_d2.release();
    if (host_pub) (void)hipHostFree(host_pub);
    if (out_pinned) (void)hipHostFree(out_pinned);
    if (ev_src) (void)hipEventDestroy(ev_src);
    if (srv_mbs) (void)hipFree(srv_mbs);
    if (srv_rows) (void)hipHostFree(srv_rows);
    srv_counter.release();
  }
};

namespace {

// what the index leaf size aims for (points per occupied cell; NDT_GICP_PPC overrides it for tuning runs): with the
// margin bound most queries finish inside their own cell, so fuller cells cost little and far queries need fewer shells
static const double kGicpPointsPerCell = std::getenv("NDT_GICP_PPC") ? std::atof(std::getenv("NDT_GICP_PPC")) : 12.0;
constexpr long long kGicpMaxCells = 1ll << 26;       // dense cell table budget (256 MB of int)

// Builds the voxel index of a host cloud on `c`: finite check, upload, leaf size from the cloud's own
// density (volume guess first, then corrected once from the measured points per occupied cell --
// scans are surfaces, so occupancy grows with the square of the leaf).
ndt_status gicp_build_index(ndt_context* c, const void* pts, size_t n, size_t stride, float* hint_leaf, size_t* hint_n) {
  if (!pts || n == 0) return fail(NDT_ERR_INVALID, "invalid or empty point cloud dataset given");
  if (stride < 12 || stride % 4) return fail(NDT_ERR_INVALID, "stride_bytes must be a multiple of 4 and >= 12");
  double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  const unsigned char* base = static_cast<const unsigned char*>(pts);
  for (size_t i = 0; i < n; i++) {
    float p[3];
    std::memcpy(p, base + i * stride, sizeof(p));
    if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])))
      return fail(NDT_ERR_INVALID, "GICP needs finite points (point " + std::to_string(i) + " is not)");
    for (int k = 0; k < 3; k++) {
      mn[k] = std::min(mn[k], static_cast<double>(p[k]));
      mx[k] = std::max(mx[k], static_cast<double>(p[k]));
    }
  }
  ndt_status s = ensure_device(c);
  if (s) return s;
  c->target_dense = 1;
  c->min_pts = 1;
  c->index_only = true;
  s = upload_cloud(c, pts, n, stride, false, c->target);
  if (s) return s;
  double ext[3], vol = 1.0, ext_max = 0.0;
  for (int k = 0; k < 3; k++) {
    ext[k] = std::max(mx[k] - mn[k], 1e-3);
    vol *= ext[k];
    ext_max = std::max(ext_max, ext[k]);
  }
  auto clamp_leaf = [&](double leaf) {
    leaf = std::max(leaf, 1e-4);
    for (int it = 0; it < 64; it++) {  // keep the dense cell table within budget
      const double cells = (ext[0] / leaf + 2) * (ext[1] / leaf + 2) * (ext[2] / leaf + 2);
      if (cells <= static_cast<double>(kGicpMaxCells)) break;
      leaf *= 1.26;
    }
    return static_cast<float>(leaf);
  };
  float leaf = clamp_leaf(std::cbrt(vol * kGicpPointsPerCell / static_cast<double>(n)));
  // consecutive scans of a sequence have about the same density: start from the leaf the previous cloud of about this size
  // ended up with (usually right at once; the search results do not depend on the leaf, only the time does)
  if (*hint_leaf > 0 && n >= *hint_n - *hint_n / 4 && n <= *hint_n + *hint_n / 4) leaf = clamp_leaf(*hint_leaf);
  for (int pass = 0; pass < 4; pass++) {
    c->resolution = leaf;
    s = build_grid(c);
    if (s) return s;
    s = grid_counts(c, c->grid.get());
    if (s) return s;
    const double per_cell = static_cast<double>(n) / static_cast<double>(std::max<size_t>(c->grid->n_leaves, 1));
    if (per_cell <= 2.0 * kGicpPointsPerCell && (per_cell >= 0.4 * kGicpPointsPerCell || c->grid->n_leaves <= 8)) break;
    // too coarse (dense surfaces) or too fine (flat / thin clouds, where the volume guess means little)
    const float next = clamp_leaf(static_cast<double>(leaf) * std::sqrt(kGicpPointsPerCell / per_cell));
    if (next < 0.9f * leaf || next > 1.1f * leaf) leaf = next;
    else break;
  }
  if (std::getenv("NDT_GICP_DEBUG"))
    std::fprintf(stderr, "[gicp index] n=%zu leaf=%.4f cells=%lld (%d x %d x %d) occupied=%zu\n", n, static_cast<double>(leaf),
                 c->grid->geom.n_cells, c->grid->geom.div_b[0], c->grid->geom.div_b[1], c->grid->geom.div_b[2], c->grid->n_leaves);
  *hint_leaf = leaf;
  *hint_n = n;
  return ensure_cell2leaf(c, c->grid.get());
}

gicp::PointIndex gicp_index_of(const ndt_context* c) {
  gicp::PointIndex ix;
  fill_point_index(c->grid.get(), ix);
  return ix;
}

void rowmajor_from_colmajor(const float* cm, float* rm) {
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) rm[r * 4 + c] = cm ? cm[c * 4 + r] : (r == c ? 1.0f : 0.0f);
}
void colmajor_from_rowmajor(const float* rm, float* cm) {
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) cm[c * 4 + r] = rm[r * 4 + c];
}

ndt_status gicp_ready(gicp_context* h) {
  if (!h->have_tgt) return fail(NDT_ERR_NO_INPUT, "no target cloud set");
  if (!h->have_src) return fail(NDT_ERR_NO_INPUT, "no source cloud set");
  ndt_status s = ensure_device(&h->tgt);  // current device + this thread's pool stream
  if (s) return s;
  if (!h->host_pub) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->host_pub), ndt::kPublishSlots * sizeof(double), hipHostMallocDefault));
    std::memset(h->host_pub, 0, ndt::kPublishSlots * sizeof(double));
  }
  if (!h->counter.p) {
    HIP_TRY(h->counter.reserve(1));
    HIP_TRY(hipMemsetAsync(h->counter.p, 0, sizeof(unsigned), h->tgt.stream));
  }
  return NDT_OK;
}

// computeCovariances of one cloud (lazily, gicp_omp_impl.hpp:385-397)
ndt_status gicp_cloud_covariances(gicp_context* h, int which, bool want_neighbors) {
  ndt_context* c = which == 0 ? &h->tgt : &h->src;
  DevBuf<double>& cov = which == 0 ? h->cov_tgt : h->cov_src;
  bool& have = which == 0 ? h->have_cov_tgt : h->have_cov_src;
  if (have && !want_neighbors) return NDT_OK;
  const size_t n = c->target->n;
  const int k = h->prm.k_correspondences;
  if (k > static_cast<int>(n))  // :53-57: PCL_ERROR and return, the covariances stay empty
    return fail(NDT_ERR_INVALID, "number of points in cloud (" + std::to_string(n) + ") is less than k_correspondences_ (" +
                                     std::to_string(k) + ")");
  HIP_TRY(cov.reserve(n * 6));
  if (want_neighbors) {
    HIP_TRY(h->nn_idx.reserve(n * static_cast<size_t>(k)));
    HIP_TRY(h->nn_d2.reserve(n * static_cast<size_t>(k)));
  }
  // the source's pass runs on the source index's own stream, next to the target's (two independent kernels of a few
  // thousand waves each); everything that follows on the main stream waits for it through an event
  hipStream_t st = (which == 1) ? h->src.stream : h->tgt.stream;
  HIP_TRY(gicp::launch_knn_covariances(gicp_index_of(c), k, h->prm.gicp_epsilon, cov.p, want_neighbors ? h->nn_idx.p : nullptr,
                                       want_neighbors ? h->nn_d2.p : nullptr, st));
  if (which == 1) {
    if (!h->ev_src) HIP_TRY(hipEventCreateWithFlags(&h->ev_src, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(h->ev_src, h->src.stream));
    h->src_cov_pending = true;  // gicp_join_source makes the main stream wait -- after the target's pass has been queued
  }
  have = true;
  return NDT_OK;
}

ndt_status gicp_join_source(gicp_context* h) {
  if (h->src_cov_pending) {
    HIP_TRY(hipStreamWaitEvent(h->tgt.stream, h->ev_src, 0));
    h->src_cov_pending = false;
  }
  return NDT_OK;
}

// covariances of both clouds + the guess-moved source (:385-403)
ndt_status gicp_prepare(gicp_context* h, const float* guess_cm) {
  ndt_status s = gicp_ready(h);
  if (s) return s;
  s = gicp_cloud_covariances(h, 1, false);  // (source first: it runs on its own stream while the target's is queued here)
  if (s) return s;
  s = gicp_cloud_covariances(h, 0, false);
  if (s) return s;
  s = gicp_join_source(h);
  if (s) return s;
  const size_t n = h->src.target->n;
  rowmajor_from_colmajor(guess_cm, h->guess_rm);
  HIP_TRY(h->output.reserve(n));
  HIP_TRY(h->corr.reserve(n));
  HIP_TRY(h->maha.reserve(n * 9));
  HIP_TRY(h->partials.reserve(static_cast<size_t>(gicp::kFunctorMaxBlocks) * ndt::kEvalStride));
  // pcl::transformPointCloud(output, output, guess), :403
  HIP_TRY(ndt::launch_transform(h->src.target->pts.p, static_cast<int>(n), h->guess_rm, h->output.p, h->tgt.stream));
  return NDT_OK;
}

struct GicpDevice : gicp::Backend {
  gicp_context* h;
  std::string error;
  // The line search evaluates operator() and then, if the step passes Fletcher's rho test, df at the very same
  // point (gicp_driver.cpp line_search): the operator() launch also accumulates df's sums, and the df request that
  // follows is answered from here without a launch.
  bool fuse = std::getenv("NDT_GICP_NO_FUSE") == nullptr;
  int evals_served = 0;
  bool have_grad = false;
  float grad_T[16];
  gicp::FunctorSums grad_sums;
  explicit GicpDevice(gicp_context* ctx) : h(ctx) {}

  bool correspond(const float transformation[16], const double R[9]) override {
    gicp::Rot3d rot;
    for (int i = 0; i < 9; i++) rot.m[i] = R[i];
    have_grad = false;
    server_stop();  // the correspondences change: the next BFGS run gets a fresh server behind this kernel
    const double thr = h->prm.corr_dist_threshold * h->prm.corr_dist_threshold;  // :401
    const hipError_t e = gicp::launch_correspond(h->output.p, static_cast<int>(h->src.target->n), transformation, rot,
                                                 gicp_index_of(&h->tgt), h->cov_src.p, h->cov_tgt.p, thr, h->corr.p, h->maha.p,
                                                 h->tgt.stream);
    if (e != hipSuccess) {
      error = std::string("correspondence kernel: ") + hipGetErrorString(e);
      return false;
    }
    return true;
  }

  // ---- persistent objective server -------------------------------------------------------------------------
  void* mailbox() const { return static_cast<unsigned char*>(h->srv_mbs) + static_cast<size_t>(h->srv_flip) * ndt::server_mailbox_bytes(); }
  bool server_available() {
    if (h->srv_tried) return h->srv_mbs != nullptr;
    h->srv_tried = true;
    const char* v = std::getenv("NDT_GICP_SERVER");
    if (v && std::atoi(v) == 0) return false;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->tgt.device) != hipSuccess || prop.isLargeBar == 0) return false;  // the direct mailbox needs the BAR
    const size_t mb = ndt::server_mailbox_bytes();
    if (hipExtMallocWithFlags(&h->srv_mbs, 2 * mb, hipDeviceMallocFinegrained) != hipSuccess) {
      h->srv_mbs = nullptr;
      (void)hipGetLastError();
      return false;
    }
    if (hipMemset(h->srv_mbs, 0, 2 * mb) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->srv_rows), gicp::kGicpServerParts * ndt::kPublishSlots * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        h->srv_counter.reserve(32 * gicp::kGicpServerParts) != hipSuccess) {
      (void)hipFree(h->srv_mbs);
      h->srv_mbs = nullptr;
      (void)hipGetLastError();
      return false;
    }
    std::memset(h->srv_rows, 0, gicp::kGicpServerParts * ndt::kPublishSlots * sizeof(double));
    return true;
  }
  bool server_start() {
    server_mark(&h->tgt, true);  // this device's turn among the persistent kernels of the process
    h->srv_flip ^= 1;
    ndt::server_reset_mailbox(mailbox());
    const int n = static_cast<int>(h->src.target->n);
    h->srv_blocks = gicp::server_blocks(n);
    hipError_t e = hipMemsetAsync(h->srv_counter.p, 0, 32 * gicp::kGicpServerParts * sizeof(unsigned), h->tgt.stream);
    if (e == hipSuccess) e = h->partials.reserve(static_cast<size_t>(std::max(h->srv_blocks, gicp::kFunctorMaxBlocks)) * ndt::kEvalStride);
    if (e == hipSuccess)
      e = gicp::launch_server(h->output.p, n, h->tgt.target->pts.p, h->corr.p, h->maha.p, mailbox(), h->srv_blocks, h->partials.p,
                              h->srv_counter.p, h->srv_rows, h->seq + 1, 2000000ull /* 20 ms */, h->tgt.stream);
    if (e != hipSuccess) {
      server_mark(&h->tgt, false);
      error = std::string("objective server: ") + hipGetErrorString(e);
      return false;
    }
    return true;
  }
  void server_stop() {  // the last command: everything queued later on the stream runs behind the exiting kernel
    if (!h->tgt.server_running) return;
    ndt::server_post(mailbox(), ++h->seq, ndt::kServerCmdExit, nullptr, nullptr);
    server_mark(&h->tgt, false);
  }
  // one evaluation through the server; false = it has left (idle time-out): the caller uses the launch path
  bool server_sums(int launch_mode, const float T[16], double row[ndt::kEvalStride]) {
    const unsigned long long seq = ++h->seq;
    ndt::server_post(mailbox(), seq, launch_mode, T, nullptr);
    const int n_parts = std::min(gicp::kGicpServerParts, h->srv_blocks);
    unsigned arrived = 0;
    const unsigned all = (1u << n_parts) - 1u;
    unsigned spins = 0;
    for (;;) {
      for (int p = 0; p < n_parts; p++)
        if (!(arrived & (1u << p)) && pub_ready(h->srv_rows + static_cast<size_t>(p) * ndt::kPublishSlots, seq)) arrived |= 1u << p;
      if (arrived == all) break;
      __builtin_ia32_pause();
      if ((++spins & 0x3FFF) == 0 && (ndt::server_dead_word(mailbox()) != 0 || hipStreamQuery(h->tgt.stream) != hipErrorNotReady)) {
        server_mark(&h->tgt, false);
        (void)hipStreamSynchronize(h->tgt.stream);
        return false;
      }
    }
    double part[ndt::kEvalStride];
    for (int k = 0; k < ndt::kEvalStride; k++) row[k] = 0.0;
    for (int p = 0; p < gicp::kGicpServerParts; p++) {  // second stage of k_functor's fixed-order sum
      if (p < n_parts) pub_gather(h->srv_rows + static_cast<size_t>(p) * ndt::kPublishSlots, part);
      for (int k = 0; k < ndt::kEvalStride; k++) row[k] += (p < n_parts) ? part[k] : 0.0;
    }
    return true;
  }

  bool sums(int mode, const float T[16], gicp::FunctorSums& out) override {
    if (mode == 1 && have_grad && std::memcmp(T, grad_T, sizeof(grad_T)) == 0) {
      out = grad_sums;
      return true;
    }
    const int n = static_cast<int>(h->src.target->n);
    const int launch_mode = (mode == 0 && fuse) ? 3 : mode;
    double row[ndt::kEvalStride];
    bool have_row = false;
    if (server_available()) {
      if (!h->tgt.server_running && !server_start()) return false;
      // liveness test hook: a host that goes quiet in the middle of a BFGS run (the server's patience is 20 ms)
      static const int stall_ms = std::getenv("NDT_GICP_TEST_STALL_MS") ? std::atoi(std::getenv("NDT_GICP_TEST_STALL_MS")) : 0;
      if (stall_ms > 0 && ++evals_served == 5) std::this_thread::sleep_for(std::chrono::milliseconds(stall_ms));
      have_row = server_sums(launch_mode, T, row);
    }
    if (!have_row) {
      const unsigned long long seq = ++h->seq;
      const hipError_t e = gicp::launch_functor(launch_mode, h->output.p, n, h->tgt.target->pts.p, h->corr.p, h->maha.p, T,
                                                gicp::functor_blocks(n), h->partials.p, h->counter.p, h->host_pub, seq, h->tgt.stream);
      if (e != hipSuccess) {
        error = std::string("functor kernel: ") + hipGetErrorString(e);
        return false;
      }
      // the row arrives as 64 self-validating words; poll, and look at the stream now and then so that a
      // failed launch cannot hang the caller
      for (unsigned long long spins = 1; !pub_ready(h->host_pub, seq); spins++) {
        if ((spins & 0xfffff) == 0) {
          const hipError_t q = hipStreamQuery(h->tgt.stream);
          if (q == hipSuccess) {
            if (pub_ready(h->host_pub, seq)) break;
            error = "functor kernel finished without publishing its result";
            return false;
          }
          if (q != hipErrorNotReady) {
            error = std::string("functor kernel: ") + hipGetErrorString(q);
            return false;
          }
        }
      }
      pub_gather(h->host_pub, row);
    }
    out.f = row[0];
    for (int i = 0; i < 3; i++) out.g[i] = row[1 + i];
    for (int i = 0; i < 9; i++) out.R[i] = row[4 + i];
    out.m = row[13];
    if (launch_mode == 3) {  // slot 0 is operator()'s value; keep the gradient sums for the df that follows
      have_grad = true;
      std::memcpy(grad_T, T, sizeof(grad_T));
      grad_sums = out;
    }
    return true;
  }
  ~GicpDevice() override { server_stop(); }
};

}  // namespace

extern "C" {

ndt_status gicp_create(int device, gicp_handle* out) {
  if (!out || device < 0) return fail(NDT_ERR_INVALID, "bad arguments");
  gicp_context* h = new gicp_context();
  h->tgt.device = device;
  h->src.device = device;
  *out = h;
  return NDT_OK;
}

void gicp_destroy(gicp_handle h) {
  if (!h) return;
  for (ndt_context* c : {&h->tgt, &h->src})
    if (c->device_ready) {
      (void)hipSetDevice(c->device);
      (void)hipStreamSynchronize(c->stream);
    }
  delete h;
}

ndt_status gicp_set_correspondence_randomness(gicp_handle h, int k) {
  if (!h || k < 1 || k > gicp::kMaxK) return fail(NDT_ERR_INVALID, "k_correspondences must be in [1, 64]");
  // (covariances already computed stay, as in the reference: computeTransformation only computes them while they are
  // empty, gicp_omp_impl.hpp:386-397, and only setInputSource / setInputTarget reset them)
  h->prm.k_correspondences = k;
  return NDT_OK;
}
ndt_status gicp_set_rotation_epsilon(gicp_handle h, double eps) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  h->prm.rotation_epsilon = eps;
  return NDT_OK;
}
ndt_status gicp_set_maximum_optimizer_iterations(gicp_handle h, int n) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  h->prm.max_inner_iterations = n;
  return NDT_OK;
}
ndt_status gicp_set_transformation_epsilon(gicp_handle h, double eps) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  h->prm.transformation_epsilon = eps;
  return NDT_OK;
}
ndt_status gicp_set_maximum_iterations(gicp_handle h, int n) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  h->prm.max_iterations = n;
  return NDT_OK;
}
ndt_status gicp_set_max_correspondence_distance(gicp_handle h, double d) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  h->prm.corr_dist_threshold = d;
  return NDT_OK;
}

ndt_status gicp_set_input_target(gicp_handle h, const void* pts, size_t n, size_t stride_bytes) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  h->have_tgt = false;
  h->have_cov_tgt = h->user_cov_tgt = false;  // target_covariances_.reset()
  h->step_ready = false;
  if (h->tgt.device_ready) {  // kernels of an earlier align may still read the old index
    HIP_TRY(hipSetDevice(h->tgt.device));
    HIP_TRY(hipStreamSynchronize(h->tgt.stream));
  }
  const ndt_status s = gicp_build_index(&h->tgt, pts, n, stride_bytes, &h->hint_leaf[0], &h->hint_n[0]);
  if (s) return s;
  h->have_tgt = true;
  return NDT_OK;
}

ndt_status gicp_set_input_source(gicp_handle h, const void* pts, size_t n, size_t stride_bytes) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  h->have_src = false;
  h->have_cov_src = h->user_cov_src = false;  // input_covariances_.reset()
  h->step_ready = false;
  if (h->tgt.device_ready) {
    HIP_TRY(hipSetDevice(h->tgt.device));
    HIP_TRY(hipStreamSynchronize(h->tgt.stream));
  }
  const ndt_status s = gicp_build_index(&h->src, pts, n, stride_bytes, &h->hint_leaf[1], &h->hint_n[1]);
  if (s) return s;
  HIP_TRY(hipStreamSynchronize(h->src.stream));  // the index is read from tgt's stream from here on
  h->have_src = true;
  return NDT_OK;
}

ndt_status gicp_align(gicp_handle h, const float* guess, float* final_T, int* converged, int* n_iterations, void* out_cloud) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  ndt_status s = gicp_prepare(h, guess);
  if (s) return s;
  h->step_ready = false;
  GicpDevice dev(h);
  const gicp::Result r = gicp::run(h->prm, h->guess_rm, dev);
  dev.server_stop();  // before anything else is queued on the stream or waited for: the server would sit out its patience
  if (r.backend_failed || !dev.error.empty()) return fail(NDT_ERR_HIP, dev.error.empty() ? "device failure" : dev.error);
  colmajor_from_rowmajor(r.final_T, h->final_T);
  h->converged = r.converged ? 1 : 0;
  h->nr_iterations = r.nr_iterations;
  h->n_f = r.n_f;
  h->n_df = r.n_df;
  h->n_fdf = r.n_fdf;
  h->correspondences = r.correspondences;
  if (final_T) std::memcpy(final_T, h->final_T, sizeof(h->final_T));
  if (converged) *converged = h->converged;
  if (n_iterations) *n_iterations = h->nr_iterations;
  if (out_cloud) {  // pcl::transformPointCloud(*input_, output, final_transformation_), :513-516
    const size_t n = h->src.target->n;
    HIP_TRY(h->out_cloud.reserve(n));
    HIP_TRY(ndt::launch_transform(h->src.target->pts.p, static_cast<int>(n), r.final_T, h->out_cloud.p, h->tgt.stream));
    // through page-locked staging: a D2H copy into the caller's pageable buffer is staged by the runtime in small pieces
    const size_t bytes = n * sizeof(float4);
    if (h->out_pinned_bytes < bytes) {
      if (h->out_pinned) (void)hipHostFree(h->out_pinned);
      h->out_pinned = nullptr;
      h->out_pinned_bytes = 0;
      HIP_TRY(hipHostMalloc(&h->out_pinned, bytes + bytes / 4, hipHostMallocDefault));
      h->out_pinned_bytes = bytes + bytes / 4;
    }
    HIP_TRY(hipMemcpyAsync(h->out_pinned, h->out_cloud.p, bytes, hipMemcpyDeviceToHost, h->tgt.stream));
    HIP_TRY(hipStreamSynchronize(h->tgt.stream));
    std::memcpy(out_cloud, h->out_pinned, bytes);
  }
  return NDT_OK;
}

ndt_status gicp_get_result(gicp_handle h, float* final_T, int* converged, int* n_iterations) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  if (final_T) std::memcpy(final_T, h->final_T, sizeof(h->final_T));
  if (converged) *converged = h->converged;
  if (n_iterations) *n_iterations = h->nr_iterations;
  return NDT_OK;
}

ndt_status gicp_get_fitness_score(gicp_handle h, double max_range, double* fitness) {
  if (!h || !fitness) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = gicp_ready(h);
  if (s) return s;
  return fitness_impl(&h->tgt, h->src.target->pts.p, static_cast<int>(h->src.target->n), h->final_T, max_range, fitness);
}

ndt_status gicp_get_stats(gicp_handle h, int* n_f, int* n_df, int* n_fdf, int* correspondences) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  if (n_f) *n_f = h->n_f;
  if (n_df) *n_df = h->n_df;
  if (n_fdf) *n_fdf = h->n_fdf;
  if (correspondences) *correspondences = h->correspondences;
  return NDT_OK;
}

// setTargetCovariances / setSourceCovariances (gicp_omp.h:165-168,186-189): caller-supplied covariances take the place of
// the k-NN ones until the cloud is set again.  cov: [n][9] row-major 3x3 (symmetric: the upper triangle is kept).
static ndt_status gicp_set_covariances(gicp_handle h, int which, const double* cov, size_t n) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  if (!(which == 0 ? h->have_tgt : h->have_src)) return fail(NDT_ERR_NO_INPUT, "set the cloud before its covariances");
  bool& have = which == 0 ? h->have_cov_tgt : h->have_cov_src;
  bool& user = which == 0 ? h->user_cov_tgt : h->user_cov_src;
  if (!cov || n == 0) {  // an empty vector: computed again by the next align (:386,392)
    have = user = false;
    return NDT_OK;
  }
  ndt_context* c = which == 0 ? &h->tgt : &h->src;
  if (n != c->target->n) return fail(NDT_ERR_INVALID, "one covariance per point of the cloud is required");
  ndt_status s = ensure_device(&h->tgt);
  if (s) return s;
  std::vector<double> c6(n * 6);
  for (size_t i = 0; i < n; i++) {
    const double* m = cov + i * 9;
    double* o = &c6[i * 6];
    o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = m[4]; o[4] = m[5]; o[5] = m[8];
  }
  DevBuf<double>& dst = which == 0 ? h->cov_tgt : h->cov_src;
  HIP_TRY(dst.reserve(n * 6));
  HIP_TRY(hipStreamSynchronize(h->tgt.stream));  // kernels of an earlier align may still read the old covariances
  HIP_TRY(hipMemcpy(dst.p, c6.data(), n * 6 * sizeof(double), hipMemcpyHostToDevice));
  have = user = true;
  h->step_ready = false;
  return NDT_OK;
}
ndt_status gicp_set_target_covariances(gicp_handle h, const double* cov, size_t n) { return gicp_set_covariances(h, 0, cov, n); }
ndt_status gicp_set_source_covariances(gicp_handle h, const double* cov, size_t n) { return gicp_set_covariances(h, 1, cov, n); }

ndt_status gicp_covariances(gicp_handle h, int which, double* cov, int* nn_idx, float* nn_d2) {
  if (!h || !cov || which < 0 || which > 1) return fail(NDT_ERR_INVALID, "bad arguments");
  if (!(which == 0 ? h->have_tgt : h->have_src)) return fail(NDT_ERR_NO_INPUT, "cloud not set");
  ndt_status s = ensure_device(&h->tgt);
  if (s) return s;
  // inspection: the k-NN covariances for the CURRENT k (caller-supplied ones are returned as they are; neighbours cannot
  // be asked for then)
  if (which == 0 ? h->user_cov_tgt : h->user_cov_src) {
    if (nn_idx || nn_d2) return fail(NDT_ERR_INVALID, "caller-supplied covariances have no neighbour lists");
  } else {
    (which == 0 ? h->have_cov_tgt : h->have_cov_src) = false;
  }
  const bool want_nn = nn_idx && nn_d2;
  s = gicp_cloud_covariances(h, which, want_nn);
  if (s) return s;
  s = gicp_join_source(h);
  if (s) return s;
  ndt_context* c = which == 0 ? &h->tgt : &h->src;
  const size_t n = c->target->n;
  std::vector<double> c6(n * 6);
  HIP_TRY(hipMemcpyAsync(c6.data(), (which == 0 ? h->cov_tgt : h->cov_src).p, n * 6 * sizeof(double), hipMemcpyDeviceToHost, h->tgt.stream));
  if (want_nn) {
    const size_t k = static_cast<size_t>(h->prm.k_correspondences);
    HIP_TRY(hipMemcpyAsync(nn_idx, h->nn_idx.p, n * k * sizeof(int), hipMemcpyDeviceToHost, h->tgt.stream));
    HIP_TRY(hipMemcpyAsync(nn_d2, h->nn_d2.p, n * k * sizeof(float), hipMemcpyDeviceToHost, h->tgt.stream));
  }
  HIP_TRY(hipStreamSynchronize(h->tgt.stream));
  for (size_t i = 0; i < n; i++) {
    const double* s6 = &c6[i * 6];
    double* o = cov + i * 9;
    o[0] = s6[0]; o[1] = s6[1]; o[2] = s6[2];
    o[3] = s6[1]; o[4] = s6[3]; o[5] = s6[4];
    o[6] = s6[2]; o[7] = s6[4]; o[8] = s6[5];
  }
  return NDT_OK;
}

ndt_status gicp_step_correspond(gicp_handle h, const float* guess, const float* transformation, int* corr, float* maha,
                                int* n_correspondences) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  ndt_status s = gicp_prepare(h, guess);
  if (s) return s;
  float T[16];
  rowmajor_from_colmajor(transformation, T);
  double R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double acc = 0.0;
      for (int k = 0; k < 4; k++) acc += static_cast<double>(T[i * 4 + k]) * static_cast<double>(h->guess_rm[k * 4 + j]);
      R[i * 3 + j] = acc;
    }
  GicpDevice dev(h);
  if (!dev.correspond(T, R)) return fail(NDT_ERR_HIP, dev.error);
  const size_t n = h->src.target->n;
  std::vector<int> c(n);
  HIP_TRY(hipMemcpyAsync(c.data(), h->corr.p, n * sizeof(int), hipMemcpyDeviceToHost, h->tgt.stream));
  if (maha) HIP_TRY(hipMemcpyAsync(maha, h->maha.p, n * 9 * sizeof(float), hipMemcpyDeviceToHost, h->tgt.stream));
  HIP_TRY(hipStreamSynchronize(h->tgt.stream));
  int m = 0;
  for (size_t i = 0; i < n; i++) m += c[i] >= 0;
  if (corr) std::memcpy(corr, c.data(), n * sizeof(int));
  if (n_correspondences) *n_correspondences = m;
  h->step_ready = true;
  return NDT_OK;
}

ndt_status gicp_step_functor(gicp_handle h, int mode, const double* x, double* f, double* g) {
  if (!h || !x || mode < 0 || mode > 2) return fail(NDT_ERR_INVALID, "bad arguments");
  if (!h->step_ready) return fail(NDT_ERR_NO_INPUT, "gicp_step_correspond has not run");
  ndt_status s = gicp_ready(h);
  if (s) return s;
  GicpDevice dev(h);
  float T[16];
  gicp::apply_state(x, T);
  gicp::FunctorSums sums;
  if (!dev.sums(mode, T, sums)) return fail(NDT_ERR_HIP, dev.error);
  const int m = static_cast<int>(sums.m);
  if (mode != 1 && f) *f = sums.f / static_cast<double>(m);
  if (mode != 0 && g) {
    for (int i = 0; i < 3; i++) g[i] = sums.g[i] * (2.0 / m);
    double R[9];
    for (int i = 0; i < 9; i++) R[i] = sums.R[i] * (2.0 / m);
    gicp::rotation_gradient(x, R, g);
  }
  return NDT_OK;
}

void gicp_host_apply_state(const double* x, float* T) {
  float rm[16];
  gicp::apply_state(x, rm);
  colmajor_from_rowmajor(rm, T);
}

}  // extern "C"
