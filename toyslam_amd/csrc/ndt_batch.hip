// ndt_batch.hip -- lock-step batches (map-build mode: many sources against the one target) and the multi-GPU exchange step.
// (split out of the former single C-ABI unit; shared state in ndt_internal.hpp)
#include "ndt_internal.hpp"

namespace ndtc {

// Small spinning worker pool for the per-step host work of a lock-step batch (one Newton /
// More-Thuente state machine per scan: 6x6 SVD solves, pose -> matrix, angle tables).  Threads
// live for one ndt_align_batch call.
class StepPool {
 public:
  explicit StepPool(int n_threads) : n_(std::max(1, n_threads)) {
    for (int t = 1; t < n_; t++) workers_.emplace_back([this, t] { loop(t); });
  }
  ~StepPool() {
    stop_.store(true, std::memory_order_release);
    gen_.fetch_add(1, std::memory_order_acq_rel);
    for (auto& w : workers_) w.join();
  }
  // runs fn(i) for i in [0, count), statically partitioned; returns when all are done
  template <class F>
  void run(size_t count, const F& fn) {
    if (n_ == 1 || count < 32) {
      for (size_t i = 0; i < count; i++) fn(i);
      return;
    }
    job_ = [&](int t) {
      const size_t lo = count * t / n_, hi = count * (t + 1) / n_;
      for (size_t i = lo; i < hi; i++) fn(i);
    };
    pending_.store(n_ - 1, std::memory_order_release);
    gen_.fetch_add(1, std::memory_order_acq_rel);
    job_(0);
    while (pending_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  }

 private:
  void loop(int t) {
    unsigned long long seen = 0;
    for (;;) {
      unsigned spins = 0;
      while (gen_.load(std::memory_order_acquire) == seen) {
        __builtin_ia32_pause();
        if (++spins > 20000) { std::this_thread::yield(); spins = 0; }
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_.load(std::memory_order_acquire)) return;
      job_(t);
      pending_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  int n_;
  std::vector<std::thread> workers_;
  std::function<void(int)> job_;
  std::atomic<unsigned long long> gen_{0};
  std::atomic<int> pending_{0};
  std::atomic<bool> stop_{false};
};

}  // namespace ndtc

extern "C" {

// ---- batch ---------------------------------------------------------------
static ndt_status align_batch_impl(ndt_handle h, const void* pts, const size_t* offsets, size_t n_scans, size_t stride,
                                   bool on_device, const float* guesses, float* final_T, int* conv, int* iters,
                                   double* tprob) {
  if (!h || !offsets) return fail(NDT_ERR_INVALID, "bad arguments");
  if (!h->grid || !h->target) return fail(NDT_ERR_NO_INPUT, "no input target");
  if (n_scans == 0) return NDT_OK;
  if (n_scans > 65535) return fail(NDT_ERR_INVALID, "at most 65535 scans per batch");
  for (size_t k = 0; k < n_scans; k++)
    if (offsets[k + 1] < offsets[k]) return fail(NDT_ERR_INVALID, "offsets must be non-decreasing");
  std::shared_ptr<DeviceCloud> cloud;
  const unsigned char* base = static_cast<const unsigned char*>(pts) + offsets[0] * stride;
  const size_t total = offsets[n_scans] - offsets[0];
  ndt_status s = upload_cloud(h, base, total, stride, on_device, cloud);
  if (s) return s;
  s = order_cloud(h, cloud.get(), offsets, n_scans);
  if (s) return s;
  const bool use_sorted = cloud->n_sorted > 0 && !cloud->scan_counts.empty();
  const float4* batch_pts = use_sorted ? cloud->sorted.p : cloud->pts.p;
  s = ensure_host_rows(h, n_scans);
  if (s) return s;
  const ndt::Gauss gs = ndt::gauss_constants(h->resolution, h->outlier_ratio);
  std::vector<ndt::ScanSolver> solvers(n_scans);
  // per-step descriptors live in pinned host memory: the H2D copies are then truly asynchronous
  const size_t pinned_need = n_scans * sizeof(ndt::ScanDesc) + 4 * n_scans * sizeof(int);  // + per-kind and all-kinds active lists
  if (pinned_need > h->batch_pinned_bytes) {
    if (h->batch_pinned) (void)hipHostFree(h->batch_pinned);
    h->batch_pinned = nullptr;
    h->batch_pinned_bytes = 0;
    HIP_TRY(hipHostMalloc(&h->batch_pinned, pinned_need, hipHostMallocDefault));
    h->batch_pinned_bytes = pinned_need;
  }
  ndt::ScanDesc* descs = static_cast<ndt::ScanDesc*>(h->batch_pinned);
  int* active = reinterpret_cast<int*>(descs + n_scans);
  size_t max_n = 0;
  for (size_t k = 0; k < n_scans; k++) {
    const size_t cnt = offsets[k + 1] - offsets[k];
    solvers[k].start(guesses ? guesses + 16 * k : nullptr, cnt, solver_params(h));
    descs[k].offset = static_cast<int>(use_sorted ? cloud->scan_starts[k] : offsets[k] - offsets[0]);
    descs[k].count = static_cast<int>(use_sorted ? cloud->scan_counts[k] : cnt);
    descs[k].pad = 0;
    max_n = std::max(max_n, cnt);
  }
  // rows of partials reserved per scan; the blocks actually used per scan follow the number of
  // scans that want the same kind of evaluation in a step (few active scans -> more blocks each)
  const int max_blocks = ndt::derivative_blocks(static_cast<int>(max_n), h->search);
  constexpr int kBlockBudget = 4096;
  HIP_TRY(h->partials.reserve(n_scans * max_blocks * ndt::kEvalStride));
  HIP_TRY(h->batch_out.reserve(n_scans * ndt::kEvalStride));
  HIP_TRY(h->descs.reserve((pinned_need + sizeof(ndt::ScanDesc) - 1) / sizeof(ndt::ScanDesc)));  // descriptors + the 3 active lists
  const ndt::GridView gv = h->grid->view();
  const bool degenerate = h->grid->empty;
  static const int n_host_threads = [] {
    const char* v = getenv("NDT_HOST_THREADS");
    if (v) return std::max(1, atoi(v));
    return static_cast<int>(std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 2)));
  }();
  StepPool pool(n_scans >= 32 ? n_host_threads : 1);
  static const bool batch_timing = [] { const char* v = getenv("NDT_TIMING"); return v && atoi(v) != 0; }();
  double t_fill = 0, t_gpu = 0, t_feed = 0;
  int n_steps = 0;
  // ndt_get_stats after a batch: scan evaluations (f32 kinds) / f64 Hessian recomputes of all scans, neighbours per point
  std::vector<double> nn_row(n_scans, 0.0);
  double nn_sum = 0, pts_sum = 0;
  long long evals_f32 = 0, evals_h64 = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  for (;;) {
    const auto tb0 = now();
    int n_act[3] = {0, 0, 0};
    for (size_t k = 0; k < n_scans; k++) {
      if (solvers[k].done()) {
        descs[k].kind = ndt::EVAL_NONE;
        continue;
      }
      const int kind = solvers[k].request().kind;
      descs[k].kind = kind;
      active[kind * n_scans + n_act[kind]++] = static_cast<int>(k);
    }
    if (n_act[0] + n_act[1] + n_act[2] == 0) break;
    int nblk_kind[3];
    for (int c = 0; c < 3; c++) nblk_kind[c] = std::max(1, std::min(max_blocks, kBlockBudget / std::max(1, n_act[c])));
    // scans asking for different kinds in the same step: one launch over all of them
    const int n_live = n_act[0] + n_act[1] + n_act[2];
    const bool mixed = (n_act[0] != n_live && n_act[1] != n_live && n_act[2] != n_live) && ndt::derivative_variant() == 0;
    if (mixed) {
      int* all = active + 3 * n_scans;
      int m = 0;
      for (int c = 0; c < 3; c++) {
        nblk_kind[c] = std::max(1, std::min(max_blocks, kBlockBudget / n_live));
        for (int i = 0; i < n_act[c]; i++) all[m++] = active[c * n_scans + i];
      }
    }
    pool.run(n_scans, [&](size_t k) {  // per-scan parameter tables (sin/cos, pose -> matrix)
      if (descs[k].kind == ndt::EVAL_NONE) return;
      const ndt::EvalRequest& rq = solvers[k].request();
      descs[k].pad = nblk_kind[descs[k].kind];
      if (rq.kind == ndt::EVAL_HESSIAN_F64) fill_h64_params(rq, gs, kd_radius2(h->resolution), descs[k].P64);
      else fill_eval_params(rq, gs, kd_radius2(h->resolution), descs[k].P);
    });
    const auto tb1 = now();
    if (degenerate) {
      std::memset(h->host_result, 0, n_scans * ndt::kEvalStride * sizeof(double));
    } else {
      // one H2D copy: descriptors and the three active lists are contiguous in the pinned block
      HIP_TRY(hipMemcpyAsync(h->descs.p, descs, pinned_need, hipMemcpyHostToDevice, h->stream));
      const int* d_active = reinterpret_cast<const int*>(h->descs.p + n_scans);
      ndt::EvalParams dummy = {};
      ndt::Hess64Params dummy64 = {};
      if (h->profiling) HIP_TRY(hipEventRecord(h->ev_a, h->stream));
      if (mixed) {
        HIP_TRY(ndt::launch_batch_step(batch_pts, gv, h->search, h->descs.p, d_active + 3 * n_scans, n_live, max_blocks, nblk_kind[0], h->partials.p, h->stream));
      } else {
        if (n_act[0]) HIP_TRY(ndt::launch_derivatives(batch_pts, 0, gv, dummy, h->search, true, h->descs.p, d_active, n_act[0], max_blocks, nblk_kind[0], h->partials.p, h->stream));
        if (n_act[1]) HIP_TRY(ndt::launch_derivatives(batch_pts, 0, gv, dummy, h->search, false, h->descs.p, d_active + n_scans, n_act[1], max_blocks, nblk_kind[1], h->partials.p, h->stream));
        if (n_act[2]) HIP_TRY(ndt::launch_hessian64(batch_pts, 0, gv, dummy64, h->search, h->descs.p, d_active + 2 * n_scans, n_act[2], max_blocks, nblk_kind[2], h->partials.p, h->stream));
      }
      if (h->profiling) HIP_TRY(hipEventRecord(h->ev_b, h->stream));
      if (h->allreduce) {
        HIP_TRY(ndt::launch_reduce(h->partials.p, max_blocks, static_cast<int>(n_scans), h->descs.p, h->batch_out.p, h->stream));
        if (h->allreduce_on_device) {
          HIP_TRY(hipStreamSynchronize(h->stream));
          if (h->allreduce(h->batch_out.p, n_scans * ndt::kEvalStride, 1, h->allreduce_user))
            return fail(NDT_ERR_COMM, "allreduce callback failed");
        }
        HIP_TRY(hipMemcpyAsync(h->host_result, h->batch_out.p, n_scans * ndt::kEvalStride * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (!h->allreduce_on_device) {
          if (h->allreduce(h->host_result, n_scans * ndt::kEvalStride, 0, h->allreduce_user))
            return fail(NDT_ERR_COMM, "allreduce callback failed");
        }
      } else {
        // the reduce kernel writes every live scan's row and then its sequence word (slot 31)
        // straight into pinned host memory; poll those instead of a D2H copy + stream synchronise
        const unsigned long long seq = ++h->eval_seq;
        HIP_TRY(ndt::launch_reduce(h->partials.p, max_blocks, static_cast<int>(n_scans), h->descs.p, h->host_result, h->stream, seq));
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        for (size_t k = 0; k < n_scans; k++) {
          if (descs[k].kind == ndt::EVAL_NONE) continue;
          volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(h->host_result + k * ndt::kEvalStride) + (ndt::kEvalStride - 1);
          while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFF) == 0) {
              if (hipStreamQuery(h->stream) != hipErrorNotReady) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
                return fail(NDT_ERR_HIP, "batch step finished without publishing its results");
              }
              if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20))
                return fail(NDT_ERR_HIP, "timed out waiting for the batch step");
            }
          }
        }
      }
    }
    const auto tb2 = now();
    if (h->profiling && !degenerate) {  // ndt_profile_enable(h, 1): the derivative kernels of this lock-step (slot 0)
      float ms = 0;
      HIP_TRY(hipEventSynchronize(h->ev_b));
      HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
      h->prof_n[0]++;
      h->prof_ms[0] += ms;
    }
    pool.run(n_scans, [&](size_t k) {  // Newton / More-Thuente step of every live scan
      if (descs[k].kind == ndt::EVAL_NONE) return;
      ndt::EvalResult r;
      unpack_row(h->host_result + k * ndt::kEvalStride, descs[k].kind != ndt::EVAL_NO_HESSIAN, r, &nn_row[k]);
      solvers[k].feed(r);
    });
    for (size_t k = 0; k < n_scans; k++) {
      if (descs[k].kind == ndt::EVAL_NONE) continue;
      if (descs[k].kind == ndt::EVAL_HESSIAN_F64) {
        evals_h64++;
      } else {
        evals_f32++;
        nn_sum += nn_row[k];
        pts_sum += static_cast<double>(offsets[k + 1] - offsets[k]);
      }
    }
    const auto tb3 = now();
    static const bool step_dump = [] { const char* v = getenv("NDT_TIMING"); return v && atoi(v) >= 2; }();
    if (step_dump) std::fprintf(stderr, "[step %d] act H=%d noH=%d h64=%d blocks/scan=%d/%d/%d gpu=%.1fus\n", n_steps, n_act[0], n_act[1], n_act[2], nblk_kind[0], nblk_kind[1], nblk_kind[2], secs(tb1, tb2) * 1e6);
    t_fill += secs(tb0, tb1);
    t_gpu += secs(tb1, tb2);
    t_feed += secs(tb2, tb3);
    n_steps++;
  }
  if (batch_timing)
    std::fprintf(stderr, "[ndt batch timing] scans=%zu steps=%d fill=%.1fus gpu(launch+wait)=%.1fus feed=%.1fus per step\n", n_scans,
                 n_steps, t_fill / std::max(1, n_steps) * 1e6, t_gpu / std::max(1, n_steps) * 1e6, t_feed / std::max(1, n_steps) * 1e6);
  h->n_evals = static_cast<int>(std::min<long long>(evals_f32, INT32_MAX));
  h->n_hess = static_cast<int>(std::min<long long>(evals_h64, INT32_MAX));
  h->mean_neighbors = pts_sum > 0 ? nn_sum / pts_sum : 0.0;
  for (size_t k = 0; k < n_scans; k++) {
    if (final_T) std::memcpy(final_T + 16 * k, solvers[k].final_T, 16 * sizeof(float));
    if (conv) conv[k] = solvers[k].converged ? 1 : 0;
    if (iters) iters[k] = solvers[k].nr_iterations;
    if (tprob) tprob[k] = solvers[k].trans_probability;
  }
  return NDT_OK;
}

ndt_status ndt_align_batch(ndt_handle h, const void* pts, const size_t* offsets, size_t n_scans, size_t stride,
                           const float* guesses, float* final_T, int* conv, int* iters, double* tprob) {
  return align_batch_impl(h, pts, offsets, n_scans, stride, false, guesses, final_T, conv, iters, tprob);
}
ndt_status ndt_align_batch_device(ndt_handle h, const void* pts, const size_t* offsets, size_t n_scans, size_t stride,
                                  const float* guesses, float* final_T, int* conv, int* iters, double* tprob) {
  return align_batch_impl(h, pts, offsets, n_scans, stride, true, guesses, final_T, conv, iters, tprob);
}

ndt_status ndt_set_allreduce(ndt_handle h, ndt_allreduce_fn fn, void* user, int on_device) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  h->allreduce = fn;
  h->allreduce_user = user;
  h->allreduce_on_device = on_device;
  return NDT_OK;
}

}  // extern "C"
