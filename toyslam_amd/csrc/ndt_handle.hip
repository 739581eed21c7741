// ndt_handle.hip -- handle lifetime, parameters, results, profiling switches, host-only scalar exports of the C-ABI (include/ndt_mi355.h).
// There is deliberately no CPU fallback: without a usable gfx950 device every compute entry point returns NDT_ERR_NO_DEVICE.
// (split out of the former single C-ABI unit; shared state in ndt_internal.hpp)
#include "ndt_internal.hpp"

void ndt_context::release_buffers() {
  target.reset();
  source.reset();
  map_scan.reset();
  grid.reset();
  partials.release();
  ticket.release();
  batch_out.release();
  descs.release();
  out_cloud.release();
  staging.release();
  map_pts.release();
  map_alt.release();
  server_dev_mb.release();
  server_counter.release();
  server_dbg.release();
}

namespace ndtc {

thread_local std::string g_last_error;
thread_local hipStream_t tls_pool_stream = nullptr;

int usable_devices() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// The handle's stream, on its CU partition (ndt_set_cu_partition): the whole device, the registration partition (all CUs
// but the last side_cus()) or the side partition (those).  hipExtStreamCreateWithCUMask takes one bit per CU; which
// physical CU a bit stands for is the driver's business -- all that matters here is that the two partitions are
// complementary.  If the masked stream cannot be had the handle runs unpartitioned (cu_count says which it is).
int side_cus() {
  static const int v = [] {
    const char* e = getenv("NDT_SIDE_CUS");
    return e ? std::max(8, std::min(128, atoi(e))) : 32;
  }();
  return v;
}

static ndt_status create_stream(ndt_context* h) {
  const int total = h->cu_total > 0 ? h->cu_total : 256;
  h->cu_count = total;
  if (h->cu_partition != 0 && total > 2 * side_cus()) {
    const int side = side_cus(), lo = h->cu_partition == 1 ? 0 : total - side, hi = h->cu_partition == 1 ? total - side : total;
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (int c = lo; c < hi; c++) mask[c / 32] |= 1u << (c % 32);
    if (hipExtStreamCreateWithCUMask(&h->stream, static_cast<uint32_t>(mask.size()), mask.data()) == hipSuccess) {
      h->cu_count = hi - lo;
      DevPool::instance().adopt_stream(h->stream);
      return NDT_OK;
    }
    (void)hipGetLastError();
    h->stream = nullptr;
  }
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  DevPool::instance().adopt_stream(h->stream);
  return NDT_OK;
}

ndt_status ensure_device(ndt_context* h) {
  if (h->device_ready) {
    HIP_TRY(hipSetDevice(h->device));
    tls_pool_stream = h->stream;
    return NDT_OK;
  }
  const int n = usable_devices();
  if (n <= 0 || h->device >= n)
    return fail(NDT_ERR_NO_DEVICE, "no usable HIP device (this library has no CPU fallback)");
  HIP_TRY(hipSetDevice(h->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, h->device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(NDT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  h->cu_total = prop.multiProcessorCount;
  ndt_status cs = create_stream(h);
  if (cs) return cs;
  h->device_ready = true;
  tls_pool_stream = h->stream;
  return NDT_OK;
}

ndt_status ensure_host_rows(ndt_context* h, size_t rows) {
  if (rows <= h->host_result_rows) return NDT_OK;
  if (h->host_result) (void)hipHostFree(h->host_result);
  h->host_result = nullptr;
  h->host_result_rows = 0;
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->host_result), rows * ndt::kEvalStride * sizeof(double),
                        hipHostMallocDefault));
  if (!h->host_pub) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->host_pub), ndt::kServerParts * ndt::kPublishSlots * sizeof(double), hipHostMallocDefault));
    std::memset(h->host_pub, 0, ndt::kServerParts * ndt::kPublishSlots * sizeof(double));
  }
  h->host_result_rows = rows;
  return NDT_OK;
}

ndt::SolverParams solver_params(const ndt_context* h) {
  ndt::SolverParams sp;
  sp.resolution = h->resolution;
  sp.step_size = h->step_size;
  sp.outlier_ratio = h->outlier_ratio;
  sp.trans_eps = h->trans_eps;
  sp.max_iter = h->max_iter;
  return sp;
}

}  // namespace ndtc

extern "C" {

const char* ndt_last_error(void) { return g_last_error.c_str(); }

int ndt_device_count(void) { return usable_devices(); }

ndt_status ndt_create(int device, ndt_handle* out) {
  if (!out || device < 0) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_context* h = new ndt_context();
  h->device = device;
  *out = h;
  return NDT_OK;
}

ndt_status ndt_clone(ndt_handle src, ndt_handle* out) {
  if (!src || !out) return fail(NDT_ERR_INVALID, "bad arguments");
  if (src->device_ready) {  // the shared grid / clouds may still be under construction on the source's stream
    HIP_TRY(hipSetDevice(src->device));
    HIP_TRY(hipStreamSynchronize(src->stream));
  }
  ndt_context* h = new ndt_context();
  h->device = src->device;
  h->resolution = src->resolution;
  h->step_size = src->step_size;
  h->outlier_ratio = src->outlier_ratio;
  h->trans_eps = src->trans_eps;
  h->max_iter = src->max_iter;
  h->search = src->search;
  h->num_threads = src->num_threads;
  h->persistent = src->persistent;
  h->cu_partition = src->cu_partition;
  h->voxel_index = src->voxel_index;
  h->min_pts = src->min_pts;
  h->eig_ratio = src->eig_ratio;
  h->target = src->target;
  h->source = src->source;
  h->target_dense = src->target_dense;
  h->grid = src->grid;
  std::memcpy(h->final_T, src->final_T, sizeof(h->final_T));
  h->converged = src->converged;
  h->nr_iterations = src->nr_iterations;
  h->trans_probability = src->trans_probability;
  h->n_evals = src->n_evals;
  h->n_hess = src->n_hess;
  h->mean_neighbors = src->mean_neighbors;
  *out = h;
  return NDT_OK;
}

void ndt_destroy(ndt_handle h) {
  if (!h) return;
  if (h->device_ready) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);  // nothing of this handle may still be running when its buffers return to the pool
  }
  delete h;
}

ndt_status ndt_set_resolution(ndt_handle h, float resolution) {
  if (!h || !(resolution > 0)) return fail(NDT_ERR_INVALID, "bad resolution");
  // ndt_omp.h:132-142 -- rebuilds only when a SOURCE (input_) is set
  if (h->resolution != resolution) {
    h->resolution = resolution;
    if (h->source && h->target) return build_grid(h);
  }
  return NDT_OK;
}
ndt_status ndt_set_step_size(ndt_handle h, double v) { if (!h) return fail(NDT_ERR_INVALID, "null"); h->step_size = v; return NDT_OK; }
ndt_status ndt_set_outlier_ratio(ndt_handle h, double v) { if (!h) return fail(NDT_ERR_INVALID, "null"); h->outlier_ratio = v; return NDT_OK; }
ndt_status ndt_set_transformation_epsilon(ndt_handle h, double v) { if (!h) return fail(NDT_ERR_INVALID, "null"); h->trans_eps = v; return NDT_OK; }
ndt_status ndt_set_maximum_iterations(ndt_handle h, int v) { if (!h) return fail(NDT_ERR_INVALID, "null"); h->max_iter = v; return NDT_OK; }
ndt_status ndt_set_neighborhood_search_method(ndt_handle h, int m) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  h->search = m;  // unknown values behave like DIRECT7: the reference's `default:` label
  return NDT_OK;
}
ndt_status ndt_set_evaluation_path(ndt_handle h, int persistent) { if (!h) return fail(NDT_ERR_INVALID, "null"); h->persistent = persistent ? 1 : 0; return NDT_OK; }
ndt_status ndt_set_num_threads(ndt_handle h, int n) { if (!h) return fail(NDT_ERR_INVALID, "null"); h->num_threads = n; return NDT_OK; }
ndt_status ndt_set_min_points_per_voxel(ndt_handle h, int n) {
  if (!h) return fail(NDT_ERR_INVALID, "null");
  h->min_pts = (n > 2) ? n : 3;  // voxel_grid_covariance_omp.h:227-239
  return NDT_OK;
}
ndt_status ndt_set_cov_eig_value_inflation_ratio(ndt_handle h, double r) { if (!h) return fail(NDT_ERR_INVALID, "null"); h->eig_ratio = r; return NDT_OK; }
float ndt_get_resolution(ndt_handle h) { return h ? h->resolution : 0.f; }
double ndt_get_step_size(ndt_handle h) { return h ? h->step_size : 0.0; }
double ndt_get_outlier_ratio(ndt_handle h) { return h ? h->outlier_ratio : 0.0; }

ndt_status ndt_get_result(ndt_handle h, float* final_transformation, int* has_converged, int* final_num_iteration,
                          double* transformation_probability) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  if (final_transformation) std::memcpy(final_transformation, h->final_T, sizeof(h->final_T));
  if (has_converged) *has_converged = h->converged;
  if (final_num_iteration) *final_num_iteration = h->nr_iterations;
  if (transformation_probability) *transformation_probability = h->trans_probability;
  return NDT_OK;
}

ndt_status ndt_get_stats(ndt_handle h, int* n_evals, int* n_hess, double* mean_neighbors) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  if (n_evals) *n_evals = h->n_evals;
  if (n_hess) *n_hess = h->n_hess;
  if (mean_neighbors) *mean_neighbors = h->mean_neighbors;
  return NDT_OK;
}

ndt_status ndt_set_cu_partition(ndt_handle h, int partition) {
  if (!h || partition < 0 || partition > 2) return fail(NDT_ERR_INVALID, "bad arguments");
  if (h->cu_partition == partition) return NDT_OK;
  const int old_partition = h->cu_partition;
  h->cu_partition = partition;
  if (!h->device_ready) return NDT_OK;  // the stream is created on first use
  HIP_TRY(hipSetDevice(h->device));
  ndt_status s = server_stop(h);
  if (s) return s;
  HIP_TRY(hipStreamSynchronize(h->stream));
  // The handle keeps the streams it has had, one per partition, and switches between them: destroying a stream here --
  // hipStreamDestroy of the plain stream of a handle whose other handle had been fed from pageable memory on a CU-masked
  // stream by a second host thread -- never returned (ROCm 7.2; tools/test_switches.sh: NDT_HOST_STAGE_MAX=0, NDT_SORT_SOURCE=1).
  // Buffers the handle still holds go back to the pool of the stream they were allocated for; only this handle uses it.
  const int total = h->cu_total > 0 ? h->cu_total : 256;
  h->partition_stream[old_partition] = h->stream;
  h->stream_masked[old_partition] = h->cu_count != total;
  h->stream = h->partition_stream[partition];
  if (h->stream) {
    h->cu_count = h->stream_masked[partition] ? (partition == 1 ? total - side_cus() : side_cus()) : total;
  } else {
    const int old_count = h->cu_count;
    s = create_stream(h);
    if (s || !h->stream) {  // no stream for the new partition: the handle stays where it was
      h->stream = h->partition_stream[old_partition];
      h->cu_partition = old_partition;
      h->cu_count = old_count;
      tls_pool_stream = h->stream;
      return s ? s : fail(NDT_ERR_HIP, "no stream for the CU partition");
    }
    h->stream_masked[partition] = h->cu_count != total;
    h->partition_stream[partition] = h->stream;
  }
  tls_pool_stream = h->stream;
  return s;
}

ndt_status ndt_get_cu_partition(ndt_handle h, int* partition, int* n_cus) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  ndt_status s = ensure_device(h);
  if (s) return s;
  if (partition) *partition = h->cu_partition;
  if (n_cus) *n_cus = h->cu_count;
  return NDT_OK;
}

ndt_status ndt_profile_enable(ndt_handle h, int on) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  if (on) {
    ndt_status s = ensure_device(h);
    if (s) return s;
    if (!h->ev_a) HIP_TRY(hipEventCreate(&h->ev_a));
    if (!h->ev_b) HIP_TRY(hipEventCreate(&h->ev_b));
    if (!h->ev_c) HIP_TRY(hipEventCreate(&h->ev_c));
    if (!h->ev_d) HIP_TRY(hipEventCreate(&h->ev_d));
  }
  h->profiling = on == 1;
  h->profile_server = on == 2;
  return NDT_OK;
}

ndt_status ndt_profile_read(ndt_handle h, int kind, long long* n_launches, double* total_ms, int reset) {
  if (!h || kind < 0 || kind >= ndt_context::kProfSlots) return fail(NDT_ERR_INVALID, "bad arguments");
  if (n_launches) *n_launches = h->prof_n[kind];
  if (total_ms) *total_ms = h->prof_ms[kind];
  if (reset) {
    h->prof_n[kind] = 0;
    h->prof_ms[kind] = 0;
  }
  return NDT_OK;
}

// ---- host-only pieces (no GPU) -------------------------------------------
void ndt_host_solve6(const double* H, const double* b, double* x) { ndt::solve6(H, b, x); }
void ndt_host_pose_to_matrix(const double* p, float* T) { ndt::pose_to_matrix(p, T); }
void ndt_host_matrix_to_pose(const float* T, double* p) { ndt::matrix_to_pose(T, p); }
void ndt_host_angle_derivatives(const double* p, float* j_ang, float* h_ang, double* j_ang_d, double* h_ang_d) {
  ndt::AngleDerivs ad;
  ndt::angle_derivatives(p, ad);
  if (j_ang) std::memcpy(j_ang, ad.j, sizeof(ad.j));
  if (h_ang) std::memcpy(h_ang, ad.h, sizeof(ad.h));
  if (j_ang_d) std::memcpy(j_ang_d, ad.jd, sizeof(ad.jd));
  if (h_ang_d) std::memcpy(h_ang_d, ad.hd, sizeof(ad.hd));
}
void ndt_host_gauss(float resolution, double outlier_ratio, double* d) {
  const ndt::Gauss g = ndt::gauss_constants(resolution, outlier_ratio);
  d[0] = g.d1;
  d[1] = g.d2;
  d[2] = g.d3;
}

ndt_status ndt_host_run_driver(ndt_eval_cb cb, void* user, size_t n_source, const float* guess, float resolution,
                               double step_size, double outlier_ratio, double trans_eps, int max_iter,
                               float* final_transformation, int* has_converged, int* final_num_iteration,
                               double* transformation_probability, int* n_evals, int* n_hess) {
  if (!cb) return fail(NDT_ERR_INVALID, "null callback");
  ndt::SolverParams sp;
  sp.resolution = resolution;
  sp.step_size = step_size;
  sp.outlier_ratio = outlier_ratio;
  sp.trans_eps = trans_eps;
  sp.max_iter = max_iter;
  ndt::ScanSolver solver;
  solver.start(guess, n_source, sp);
  while (!solver.done()) {
    const ndt::EvalRequest& rq = solver.request();
    ndt::EvalResult r;
    std::memset(&r, 0, sizeof(r));
    if (cb(user, rq.kind, rq.T, rq.p, &r.score, r.g, r.H)) return fail(NDT_ERR_INVALID, "evaluator callback failed");
    solver.feed(r);
  }
  if (final_transformation) std::memcpy(final_transformation, solver.final_T, 16 * sizeof(float));
  if (has_converged) *has_converged = solver.converged ? 1 : 0;
  if (final_num_iteration) *final_num_iteration = solver.nr_iterations;
  if (transformation_probability) *transformation_probability = solver.trans_probability;
  if (n_evals) *n_evals = solver.n_evals;
  if (n_hess) *n_hess = solver.n_hess;
  return NDT_OK;
}

}  // extern "C"
