// ndt_batch.hip -- lock-step batches (map-build mode: many sources against the one target) and the multi-GPU exchange step.
// (split out of the former single C-ABI unit; shared state in ndt_internal.hpp)
#include "ndt_internal.hpp"

#include <condition_variable>
#include <dlfcn.h>
#include <sched.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is dlopen'ed (no link-time dependency)

namespace ndtc {

// Small worker pool for the per-step host work of a lock-step batch (one Newton / More-Thuente
// state machine per scan: 6x6 SVD solves, pose -> matrix, angle tables).  Threads live for one
// ndt_align_batch call.  An idle worker spins for about 50 us (the gap between the two jobs of one lock-step is
// shorter than that when the step is host-bound) and then BLOCKS on a condition variable: a lock-step whose kernels
// run for hundreds of microseconds must not burn a core per worker meanwhile -- on a box whose cgroup grants a few
// CPUs to eight ranks the spinning is what exhausts the CFS quota and parks the polling thread.
class StepPool {
 public:
  explicit StepPool(int n_threads) : n_(std::max(1, n_threads)) {
    for (int t = 1; t < n_; t++) workers_.emplace_back([this, t] { loop(t); });
  }
  ~StepPool() {
    stop_.store(true, std::memory_order_release);
    post();
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
    post();
    job_(0);
    while (pending_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  }

 private:
  void post() {
    // post: gen_++ then read sleepers_; a worker: sleepers_++ then read gen_ -- a store-then-load handshake on two words,
    // which only sequential consistency orders on every architecture
    gen_.fetch_add(1, std::memory_order_seq_cst);
    if (sleepers_.load(std::memory_order_seq_cst) > 0) {
      std::lock_guard<std::mutex> g(mu_);  // a sleeper re-checks gen_ under this lock before it waits: no lost wake-up
      cv_.notify_all();
    }
  }
  void loop(int t) {
    unsigned long long seen = 0;
    for (;;) {
      const auto idle_since = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (gen_.load(std::memory_order_acquire) == seen) {
        __builtin_ia32_pause();
        if ((++spins & 63) == 0 && std::chrono::steady_clock::now() - idle_since > std::chrono::microseconds(50)) {
          std::unique_lock<std::mutex> lk(mu_);
          sleepers_.fetch_add(1, std::memory_order_seq_cst);
          cv_.wait(lk, [&] { return gen_.load(std::memory_order_seq_cst) != seen; });
          sleepers_.fetch_sub(1, std::memory_order_acq_rel);
        }
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
  std::atomic<int> sleepers_{0};
  std::atomic<bool> stop_{false};
  std::mutex mu_;
  std::condition_variable cv_;
};

// ---- how many host threads a rank may use -----------------------------------------------------------
// CPUs this process can really run on: its affinity mask, cut down to the cgroup's CPU bandwidth (cgroup v2
// cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us: threads beyond the quota are throttled by the kernel, not run),
// shared between the ranks of this node (LOCAL_WORLD_SIZE: one process per GPU).
static double read_cgroup_quota() {
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0};
    double per = 0;
    const int n = std::fscanf(f, "%63s %lf", q, &per);
    std::fclose(f);
    if (n == 2 && std::strcmp(q, "max") != 0 && per > 0) return std::atof(q) / per;
    if (n >= 1) return 0.0;  // "max": no limit
  }
  double quota = -1, per = 0;
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    if (std::fscanf(f, "%lf", &quota) != 1) quota = -1;
    std::fclose(f);
  }
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
    if (std::fscanf(f, "%lf", &per) != 1) per = 0;
    std::fclose(f);
  }
  return (quota > 0 && per > 0) ? quota / per : 0.0;
}

void host_thread_budget(int* affinity_cpus, double* quota_cpus, int* local_world) {
  cpu_set_t set;
  CPU_ZERO(&set);
  int aff = 0;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) aff = CPU_COUNT(&set);
  if (aff <= 0) aff = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
  int lw = 1;
  if (const char* v = getenv("LOCAL_WORLD_SIZE")) lw = std::max(1, atoi(v));
  if (affinity_cpus) *affinity_cpus = aff;
  if (quota_cpus) *quota_cpus = read_cgroup_quota();
  if (local_world) *local_world = lw;
}

// pure: the plan for a given budget.  share = min(affinity, floor(quota)) / local_world, at least 1.  One of the
// share is the thread that drives the batch (launches, polls); the pool gets the rest up to 16 threads (the per-step
// host work of 512 solvers stops scaling there), the batch groups at most one host thread per CPU of the share.
void host_thread_plan(int affinity_cpus, double quota_cpus, int local_world, int* pool_threads, int* max_groups) {
  int cpus = std::max(1, affinity_cpus);
  if (quota_cpus > 0) cpus = std::min(cpus, std::max(1, static_cast<int>(std::floor(quota_cpus))));
  const int share = std::max(1, cpus / std::max(1, local_world));
  if (pool_threads) *pool_threads = std::max(1, std::min(16, share / 2));
  if (max_groups) *max_groups = std::max(1, std::min(8, share));
}

}  // namespace ndtc

// ---- RCCL over xGMI (one process per GPU) -------------------------------------------------------------
// librccl is loaded on first use (dlopen by soname: inside a process that already holds an RCCL -- e.g. PyTorch's --
// the loader hands back that same copy, bound to the same HIP runtime this library is bound to).
namespace ndtc {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;  // optional
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

static Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      const char* e = dlerror();
      r.error = std::string("cannot load librccl: ") + (e ? e : "?");
      return;
    }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) r.error = "librccl lacks an expected symbol";
  });
  return r.error.empty() ? &r : nullptr;
}

static ndt_status rccl_fail(const char* what, ncclResult_t e) {
  Rccl* r = rccl();
  return fail(NDT_ERR_COMM, std::string(what) + " failed: " + (r ? r->GetErrorString(e) : "librccl unavailable"));
}

void comm_release(ndt_context* h) {
  if (!h->comm) return;
  if (Rccl* r = rccl()) (void)r->CommDestroy(static_cast<ncclComm_t>(h->comm));
  h->comm = nullptr;
  h->comm_rank = 0;
  h->comm_world = 1;
}

// in-place SUM of n f64 on the handle's stream (stream-ordered: nothing waits on the host)
ndt_status comm_allreduce(ndt_context* h, double* d_buf, size_t n) {
  Rccl* r = rccl();
  if (!r || !h->comm) return fail(NDT_ERR_COMM, "no communicator");
  const ncclResult_t e = r->AllReduce(d_buf, d_buf, n, ncclDouble, ncclSum, static_cast<ncclComm_t>(h->comm), h->stream);
  if (e != ncclSuccess) return rccl_fail("ncclAllReduce", e);
  h->comm_collectives++;
  return NDT_OK;
}

static int poll_timeout_s() {
  static const int v = [] { const char* e = getenv("NDT_BATCH_TIMEOUT_S"); return e ? std::max(1, atoi(e)) : 60; }();
  return v;
}

// waits until slot 31 of every listed row of the pinned result block carries `seq`
template <class LiveFn>
static ndt_status poll_rows(ndt_context* h, size_t n_rows, unsigned long long seq, const LiveFn& live) {
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (size_t k = 0; k < n_rows; k++) {
    if (!live(k)) continue;
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(h->host_result + k * ndt::kEvalStride) + (ndt::kEvalStride - 1);
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
      __builtin_ia32_pause();
      if ((++spins & 0xFFFF) == 0) {
        if (hipStreamQuery(h->stream) != hipErrorNotReady) {
          HIP_TRY(hipStreamSynchronize(h->stream));
          if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
          return fail(NDT_ERR_HIP, "batch step finished without publishing its results");
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(poll_timeout_s())) {
          // a peer never joined the collective: the enqueued ncclAllReduce and everything behind it would sit on the
          // stream for ever (and ndt_destroy / ndt_comm_destroy synchronise that stream) -- abort the communicator so
          // that the stream drains; the handle is left without one
          if (h->comm) {
            Rccl* r = rccl();
            if (r && r->CommAbort) (void)r->CommAbort(static_cast<ncclComm_t>(h->comm));
            h->comm = nullptr;
            h->comm_rank = 0;
            h->comm_world = 1;
            return fail(NDT_ERR_COMM, "timed out waiting for the exchange step (a rank never joined the all-reduce); communicator aborted");
          }
          return fail(NDT_ERR_HIP, "timed out waiting for the batch step");
        }
      }
    }
  }
  return NDT_OK;
}

}  // namespace ndtc

extern "C" {

// ---- batch ---------------------------------------------------------------
// Lock-step registration of `total` scans of which this rank holds scans [first, first + n_local) (plain batch:
// first = 0, n_local = total).  Every rank steps all `total` Newton / More-Thuente state machines; a rank evaluates
// only the scans it holds, the rows of the others stay zero, and ONE in-place SUM all-reduce of the [total][32] f64
// buffer per lock-step gives every rank every row (the exchange step of north_star's map-build mode).
static ndt_status align_batch_impl(ndt_handle h, const void* pts, const size_t* offsets, size_t n_local, size_t stride,
                                   bool on_device, const float* guesses, float* final_T, int* conv, int* iters,
                                   double* tprob, size_t first = 0, size_t total = 0) {
  if (!h || !offsets) return fail(NDT_ERR_INVALID, "bad arguments");
  if (!h->grid || !h->target) return fail(NDT_ERR_NO_INPUT, "no input target");
  const bool sharded = total != 0;
  if (!sharded) total = n_local;
  if (first + n_local > total) return fail(NDT_ERR_INVALID, "scan range outside the batch");
  const bool exchange = h->comm != nullptr || h->allreduce != nullptr;
  if (sharded && n_local != total && !exchange)
    return fail(NDT_ERR_COMM, "a sharded batch needs a communicator (ndt_comm_init_rank) or an all-reduce hook");
  if (total == 0) return NDT_OK;
  if (total > 65535) return fail(NDT_ERR_INVALID, "at most 65535 scans per batch");
  for (size_t k = 0; k < n_local; k++)
    if (offsets[k + 1] < offsets[k]) return fail(NDT_ERR_INVALID, "offsets must be non-decreasing");
  ndt_status s = ensure_device(h);
  if (s) return s;
  if (!h->is_batch_worker) {  // many registrations against this grid: dense, cell-ordered records (+8 % on the batch kernels)
    s = maybe_compact_records(h, true);
    if (s) return s;
  }
  std::shared_ptr<DeviceCloud> cloud;
  const size_t n_pts = n_local ? offsets[n_local] - offsets[0] : 0;
  const unsigned char* base = n_local ? static_cast<const unsigned char*>(pts) + offsets[0] * stride : nullptr;
  s = upload_cloud(h, base, n_pts, stride, on_device, cloud);
  if (s) return s;
  if (n_local) {
    s = order_cloud(h, cloud.get(), offsets, n_local);
    if (s) return s;
  }
  const bool use_sorted = cloud->n_sorted > 0 && !cloud->scan_counts.empty();
  const float4* batch_pts = use_sorted ? cloud->sorted.p : cloud->pts.p;
  s = ensure_host_rows(h, total);
  if (s) return s;
  HIP_TRY(h->batch_out.reserve(total * ndt::kEvalStride));
  const ndt::Gauss gs = ndt::gauss_constants(h->resolution, h->outlier_ratio);
  auto is_local = [&](size_t g) { return g >= first && g < first + n_local; };

  // one exchange of the packed rows: device buffer -> [all-reduce] -> pinned host rows (every row, tagged with seq)
  auto exchange_rows = [&](const std::function<bool(size_t)>& live) -> ndt_status {
    if (h->comm) {
      ndt_status sc = comm_allreduce(h, h->batch_out.p, total * ndt::kEvalStride);
      if (sc) return sc;
      if (h->profiling) HIP_TRY(hipEventRecord(h->ev_c, h->stream));
      const unsigned long long seq = ++h->eval_seq;
      HIP_TRY(ndt::launch_publish_rows(h->batch_out.p, static_cast<int>(total), h->host_result, seq, h->stream));
      if (h->profiling) HIP_TRY(hipEventRecord(h->ev_d, h->stream));
      return poll_rows(h, total, seq, live);
    }
    // caller-supplied collective (host tests over gloo; torch.distributed on the device buffer): not stream-ordered
    if (h->allreduce_on_device) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->allreduce(h->batch_out.p, total * ndt::kEvalStride, 1, h->allreduce_user)) return fail(NDT_ERR_COMM, "allreduce callback failed");
    }
    HIP_TRY(hipMemcpyAsync(h->host_result, h->batch_out.p, total * ndt::kEvalStride * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!h->allreduce_on_device && h->allreduce(h->host_result, total * ndt::kEvalStride, 0, h->allreduce_user))
      return fail(NDT_ERR_COMM, "allreduce callback failed");
    return NDT_OK;
  };

  // every rank needs every scan's point count (transformation_probability = score / N): one exchange up front
  std::vector<size_t> counts(total, 0);
  for (size_t k = 0; k < n_local; k++) counts[first + k] = offsets[k + 1] - offsets[k];
  if (sharded && exchange) {
    std::vector<double> rows(total * ndt::kEvalStride, 0.0);
    for (size_t k = 0; k < n_local; k++) rows[(first + k) * ndt::kEvalStride] = static_cast<double>(counts[first + k]);
    HIP_TRY(hipMemcpyAsync(h->batch_out.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));  // `rows` is pageable and goes out of scope
    s = exchange_rows([](size_t) { return true; });
    if (s) return s;
    for (size_t g = 0; g < total; g++) counts[g] = static_cast<size_t>(h->host_result[g * ndt::kEvalStride]);
  }

  std::vector<ndt::ScanSolver> solvers(total);
  // per-step descriptors live in pinned host memory: the H2D copies are then truly asynchronous
  const size_t pinned_need = (total * sizeof(ndt::ScanDesc) + 4 * total * sizeof(int) + 15) & ~static_cast<size_t>(15);  // + per-kind and all-kinds active lists; whole 16-B words
  if (pinned_need > h->batch_pinned_bytes) {
    if (h->batch_pinned) (void)hipHostFree(h->batch_pinned);
    h->batch_pinned = nullptr;
    h->batch_pinned_bytes = 0;
    HIP_TRY(hipHostMalloc(&h->batch_pinned, pinned_need, hipHostMallocDefault));
    h->batch_pinned_bytes = pinned_need;
  }
  ndt::ScanDesc* descs = static_cast<ndt::ScanDesc*>(h->batch_pinned);
  int* active = reinterpret_cast<int*>(descs + total);
  std::vector<int> live_kind(total, ndt::EVAL_NONE);  // what every scan's solver asked for this step, local or not
  size_t max_n = 0;
  for (size_t g = 0; g < total; g++) {
    solvers[g].start(guesses ? guesses + 16 * g : nullptr, counts[g], solver_params(h));
    descs[g].offset = 0;
    descs[g].count = 0;
    descs[g].pad = 0;
    descs[g].kind = ndt::EVAL_NONE;
    if (is_local(g)) {
      const size_t k = g - first;
      descs[g].offset = static_cast<int>(use_sorted ? cloud->scan_starts[k] : offsets[k] - offsets[0]);
      descs[g].count = static_cast<int>(use_sorted ? cloud->scan_counts[k] : counts[g]);
      max_n = std::max(max_n, counts[g]);
    }
  }
  // rows of partials reserved per scan; the blocks actually used per scan follow the number of
  // scans that want the same kind of evaluation in a step (few active scans -> more blocks each)
  const int max_blocks = ndt::batch_blocks(static_cast<int>(max_n));
  HIP_TRY(h->partials.reserve(total * max_blocks * ndt::kEvalStride));
  HIP_TRY(h->descs.reserve((pinned_need + sizeof(ndt::ScanDesc) - 1) / sizeof(ndt::ScanDesc)));  // descriptors + the 3 active lists
  const ndt::GridView gv = h->grid->view();
  const bool degenerate = h->grid->empty;
  static const int n_host_threads = [] {
    const char* v = getenv("NDT_HOST_THREADS");
    if (v) return std::max(1, atoi(v));
    int aff = 1, lw = 1, pool = 1;
    double quota = 0;
    host_thread_budget(&aff, &quota, &lw);
    host_thread_plan(aff, quota, lw, &pool, nullptr);
    return pool;
  }();
  StepPool pool((total >= 32 && !h->is_batch_worker) ? n_host_threads : 1);  // grouped batches: the groups are the host parallelism
  static const bool batch_timing = [] { const char* v = getenv("NDT_TIMING"); return v && atoi(v) != 0; }();
  double t_fill = 0, t_gpu = 0, t_feed = 0;
  int n_steps = 0;
  // ndt_get_stats after a batch: scan evaluations (f32 kinds) / f64 Hessian recomputes of all scans, neighbours per point
  std::vector<double> nn_row(total, 0.0);
  double nn_sum = 0, pts_sum = 0;
  long long evals_f32 = 0, evals_h64 = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  for (;;) {
    const auto tb0 = now();
    int n_act[3] = {0, 0, 0};
    int n_live_all = 0;
    for (size_t g = 0; g < total; g++) {
      descs[g].kind = ndt::EVAL_NONE;
      live_kind[g] = ndt::EVAL_NONE;
      if (solvers[g].done()) continue;
      const int kind = solvers[g].request().kind;
      live_kind[g] = kind;
      n_live_all++;
      if (!is_local(g)) continue;  // somebody else's scan: its row arrives with the exchange
      descs[g].kind = kind;
      active[kind * total + n_act[kind]++] = static_cast<int>(g);
    }
    if (n_live_all == 0) break;
    // scans asking for different kinds in the same step: one launch over all of them
    const int n_live = n_act[0] + n_act[1] + n_act[2];
    const bool mixed = (n_act[0] != n_live && n_act[1] != n_live && n_act[2] != n_live);
    if (mixed) {
      int* all = active + 3 * total;
      int m = 0;
      for (int c = 0; c < 3; c++)
        for (int i = 0; i < n_act[c]; i++) all[m++] = active[c * total + i];
    }
    pool.run(total, [&](size_t g) {  // per-scan parameter tables (sin/cos, pose -> matrix)
      if (descs[g].kind == ndt::EVAL_NONE) return;
      const ndt::EvalRequest& rq = solvers[g].request();
      descs[g].pad = ndt::batch_blocks(descs[g].count);  // the scan's own block count: its sums do not depend on the batch around it
      if (rq.kind == ndt::EVAL_HESSIAN_F64) fill_h64_params(rq, gs, kd_radius2(h->resolution), descs[g].P64);
      else fill_eval_params(rq, gs, kd_radius2(h->resolution), descs[g].P);
    });
    const auto tb1 = now();
    if (degenerate && !exchange) {
      std::memset(h->host_result, 0, total * ndt::kEvalStride * sizeof(double));
    } else if (degenerate) {
      // this rank's target has no voxel: its rows are zero, but it still joins the exchange -- the other ranks' grids
      // need not be empty (every rank is SUPPOSED to hold the same target; a rank that does not must not hang the rest)
      HIP_TRY(hipMemsetAsync(h->batch_out.p, 0, total * ndt::kEvalStride * sizeof(double), h->stream));
      s = exchange_rows([&](size_t g) { return live_kind[g] != ndt::EVAL_NONE; });
      if (s) return s;
    } else {
      // one H2D copy: descriptors and the three active lists are contiguous in the pinned block
      // (by a kernel reading the page-locked block: a few KB are in HBM before a DMA engine would have started)
      static const bool desc_dma = [] { const char* v = getenv("NDT_BATCH_DESC_DMA"); return v && atoi(v) != 0; }();
      if (desc_dma) HIP_TRY(hipMemcpyAsync(h->descs.p, descs, pinned_need, hipMemcpyHostToDevice, h->stream));
      else HIP_TRY(ndt::launch_copy_records(reinterpret_cast<const float4*>(descs), reinterpret_cast<float4*>(h->descs.p), static_cast<int>(pinned_need / 16), h->stream));
      const int* d_active = reinterpret_cast<const int*>(h->descs.p + total);
      ndt::EvalParams dummy = {};
      ndt::Hess64Params dummy64 = {};
      if (exchange) HIP_TRY(hipMemsetAsync(h->batch_out.p, 0, total * ndt::kEvalStride * sizeof(double), h->stream));
      if (h->profiling) HIP_TRY(hipEventRecord(h->ev_a, h->stream));
      if (mixed) {
        HIP_TRY(ndt::launch_batch_step(batch_pts, gv, h->search, h->descs.p, d_active + 3 * total, n_live, max_blocks, max_blocks, h->partials.p, h->stream));
      } else {
        if (n_act[0]) HIP_TRY(ndt::launch_derivatives(batch_pts, 0, gv, dummy, h->search, true, h->descs.p, d_active, n_act[0], max_blocks, max_blocks, h->partials.p, h->stream));
        if (n_act[1]) HIP_TRY(ndt::launch_derivatives(batch_pts, 0, gv, dummy, h->search, false, h->descs.p, d_active + total, n_act[1], max_blocks, max_blocks, h->partials.p, h->stream));
        if (n_act[2]) HIP_TRY(ndt::launch_hessian64(batch_pts, 0, gv, dummy64, h->search, h->descs.p, d_active + 2 * total, n_act[2], max_blocks, max_blocks, h->partials.p, h->stream));
      }
      if (h->profiling) HIP_TRY(hipEventRecord(h->ev_b, h->stream));
      if (exchange) {
        // rows of this rank's live scans; everything else stays zero for the SUM
        if (n_live) HIP_TRY(ndt::launch_reduce(h->partials.p, max_blocks, static_cast<int>(total), h->descs.p, h->batch_out.p, h->stream));
        s = exchange_rows([&](size_t g) { return live_kind[g] != ndt::EVAL_NONE; });
        if (s) return s;
      } else {
        // the reduce kernel writes every live scan's row and then its sequence word (slot 31)
        // straight into pinned host memory; poll those instead of a D2H copy + stream synchronise
        const unsigned long long seq = ++h->eval_seq;
        HIP_TRY(ndt::launch_reduce(h->partials.p, max_blocks, static_cast<int>(total), h->descs.p, h->host_result, h->stream, seq));
        s = poll_rows(h, total, seq, [&](size_t g) { return live_kind[g] != ndt::EVAL_NONE; });
        if (s) return s;
      }
    }
    const auto tb2 = now();
    if (h->profiling && !degenerate) {  // ndt_profile_enable(h, 1): the derivative kernels of this lock-step (slot 0)
      float ms = 0;
      HIP_TRY(hipEventSynchronize(h->ev_b));
      HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
      h->prof_n[0]++;
      h->prof_ms[0] += ms;
      if (h->comm) {  // ... the exchange: k_reduce + ncclAllReduce (slot 4), k_publish_rows (slot 5)
        HIP_TRY(hipEventSynchronize(h->ev_d));
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_b, h->ev_c));
        h->prof_n[4]++;
        h->prof_ms[4] += ms;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_c, h->ev_d));
        h->prof_n[5]++;
        h->prof_ms[5] += ms;
      }
      h->prof_n[6]++;  // slot 6: host wall time of the lock-step (descriptor fill + launches + wait + solver steps), ms
      h->prof_ms[6] += secs(tb0, now()) * 1e3;
    }
    static const int trace_scan = [] { const char* v = getenv("NDT_BATCH_TRACE"); return v ? atoi(v) : -1; }();
    if (trace_scan >= 0 && static_cast<size_t>(trace_scan) < total && live_kind[trace_scan] != ndt::EVAL_NONE) {
      const ndt::EvalRequest& rq = solvers[trace_scan].request();
      const double* row = h->host_result + static_cast<size_t>(trace_scan) * ndt::kEvalStride;
      std::fprintf(stderr, "[trace] kind %d p %.17g %.17g %.17g %.17g %.17g %.17g -> score %.17g g0 %.17g g5 %.17g H00 %.17g nn %.17g\n", static_cast<int>(rq.kind),
                   rq.p[0], rq.p[1], rq.p[2], rq.p[3], rq.p[4], rq.p[5], row[0], row[1], row[6], row[7], row[28]);
    }
    pool.run(total, [&](size_t g) {  // Newton / More-Thuente step of every live scan
      if (live_kind[g] == ndt::EVAL_NONE) return;
      ndt::EvalResult r;
      unpack_row(h->host_result + g * ndt::kEvalStride, live_kind[g] != ndt::EVAL_NO_HESSIAN, r, &nn_row[g]);
      solvers[g].feed(r);
    });
    for (size_t g = 0; g < total; g++) {
      if (live_kind[g] == ndt::EVAL_NONE) continue;
      if (live_kind[g] == ndt::EVAL_HESSIAN_F64) {
        evals_h64++;
      } else {
        evals_f32++;
        nn_sum += nn_row[g];
        pts_sum += static_cast<double>(counts[g]);
      }
    }
    const auto tb3 = now();
    static const bool step_dump = [] { const char* v = getenv("NDT_TIMING"); return v && atoi(v) >= 2; }();
    if (step_dump) std::fprintf(stderr, "[step %d] act H=%d noH=%d h64=%d blocks/scan<=%d gpu=%.1fus\n", n_steps, n_act[0], n_act[1], n_act[2], max_blocks, secs(tb1, tb2) * 1e6);
    t_fill += secs(tb0, tb1);
    t_gpu += secs(tb1, tb2);
    t_feed += secs(tb2, tb3);
    n_steps++;
  }
  if (batch_timing)
    std::fprintf(stderr, "[ndt batch timing] scans=%zu (local %zu) steps=%d fill=%.1fus gpu(launch+wait)=%.1fus feed=%.1fus per step\n", total, n_local,
                 n_steps, t_fill / std::max(1, n_steps) * 1e6, t_gpu / std::max(1, n_steps) * 1e6, t_feed / std::max(1, n_steps) * 1e6);
  h->n_evals = static_cast<int>(std::min<long long>(evals_f32, INT32_MAX));
  h->n_hess = static_cast<int>(std::min<long long>(evals_h64, INT32_MAX));
  h->mean_neighbors = pts_sum > 0 ? nn_sum / pts_sum : 0.0;
  h->batch_lock_steps = n_steps;
  for (size_t g = 0; g < total; g++) {
    if (final_T) std::memcpy(final_T + 16 * g, solvers[g].final_T, 16 * sizeof(float));
    if (conv) conv[g] = solvers[g].converged ? 1 : 0;
    if (iters) iters[g] = solvers[g].nr_iterations;
    if (tprob) tprob[g] = solvers[g].trans_probability;
  }
  return NDT_OK;
}

// A batch without an exchange step is run as several INDEPENDENT lock-step groups, each on a worker handle of its own
// (own stream, own staging buffers, the target grid shared) driven by a host thread of its own: while one group's
// kernels run, the other groups' hosts step their Newton / More-Thuente state machines, upload descriptors and queue
// their next launches -- the per-step host time (~30-70 us: descriptor fill, H2D, launches, the polled result) that a
// single lock-step loop leaves the GPU idle for disappears behind the other groups' kernels, and the tail of a group
// (few live scans) shares the chip with full steps of the others.  A scan's registration does not depend on its
// group's other members.  NDT_BATCH_GROUPS overrides the group count (1 = the single loop).
static ndt_status align_batch_grouped(ndt_handle h, const void* pts, const size_t* offsets, size_t n_scans, size_t stride,
                                      bool on_device, const float* guesses, float* final_T, int* conv, int* iters,
                                      double* tprob) {
  static const int forced = [] { const char* v = getenv("NDT_BATCH_GROUPS"); return v ? std::max(1, atoi(v)) : 0; }();
  const bool exchange = h && (h->comm != nullptr || h->allreduce != nullptr);
  size_t groups = (h && h->batch_groups_wanted > 0) ? static_cast<size_t>(h->batch_groups_wanted)
                  : forced                           ? static_cast<size_t>(forced)
                                                     : (n_scans >= 192 ? 4 : n_scans >= 16 ? 2 : 1);  // (same-box A/B, two / four groups: 64 scans 5.9k / 5.6k reg/s, 96 6.25k / 6.1k, 128 equal, 512 6.4k / 6.6k; one loop: 32 scans 3.8k against 5.4k as two)
  groups = std::min(groups, std::max<size_t>(1, n_scans / 4));
  static const int max_groups = [] {
    int aff = 1, lw = 1, mg = 1;
    double quota = 0;
    host_thread_budget(&aff, &quota, &lw);
    host_thread_plan(aff, quota, lw, nullptr, &mg);
    return mg;
  }();
  if (!(h && h->batch_groups_wanted > 0) && !forced) groups = std::min(groups, static_cast<size_t>(max_groups));  // a host thread per group
  // (event pairs around the kernels of a lock-step -- ndt_profile_enable(1) -- only mean something without overlap)
  if (!h || !offsets || exchange || groups <= 1 || !h->grid || !h->target || h->profiling)
    return align_batch_impl(h, pts, offsets, n_scans, stride, on_device, guesses, final_T, conv, iters, tprob);
  ndt_status s0 = ensure_device(h);
  if (s0) return s0;
  s0 = maybe_compact_records(h, true);  // before the groups' worker handles share the grid
  if (s0) return s0;
  HIP_TRY(hipStreamSynchronize(h->stream));  // the shared grid may still be under construction on h's stream
  while (h->batch_workers.size() < groups) {
    ndt_context* w = new ndt_context();
    w->device = h->device;
    w->is_batch_worker = true;
    h->batch_workers.push_back(w);
  }
  std::vector<ndt_status> st(groups, NDT_OK);
  std::vector<std::string> msg(groups);
  std::vector<std::thread> threads;
  for (size_t g = 0; g < groups; g++) {
    ndt_context* w = h->batch_workers[g];
    w->resolution = h->resolution;
    w->step_size = h->step_size;
    w->outlier_ratio = h->outlier_ratio;
    w->trans_eps = h->trans_eps;
    w->max_iter = h->max_iter;
    w->search = h->search;
    w->min_pts = h->min_pts;
    w->eig_ratio = h->eig_ratio;
    w->target = h->target;
    w->target_dense = h->target_dense;
    w->grid = h->grid;
    const size_t lo = n_scans * g / groups, hi = n_scans * (g + 1) / groups;
    threads.emplace_back([=, &st, &msg] {
      ndt_status s = align_batch_impl(w, pts, offsets + lo, hi - lo, stride, on_device, guesses ? guesses + 16 * lo : nullptr,
                             final_T ? final_T + 16 * lo : nullptr, conv ? conv + lo : nullptr, iters ? iters + lo : nullptr,
                             tprob ? tprob + lo : nullptr);
      st[g] = s;
      if (s) msg[g] = ndt_last_error();
    });
  }
  for (auto& t : threads) t.join();
  // the workers' rows have been polled, so their streams are idle: drop their references to the target and its grid (a
  // caller that now sets a new target must not find hundreds of MB of the old one pinned until the next batch)
  for (size_t g = 0; g < groups; g++) {
    ndt_context* w = h->batch_workers[g];
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    w->target.reset();
    w->grid.reset();
  }
  long long ne = 0, nh = 0;
  double nn_w = 0, pts_w = 0;
  int steps = 0;
  for (size_t g = 0; g < groups; g++) {
    ndt_context* w = h->batch_workers[g];
    if (st[g]) return fail(st[g], msg[g]);
    const size_t lo = n_scans * g / groups, hi = n_scans * (g + 1) / groups;
    const double evals_pts = static_cast<double>(w->n_evals) * (hi > lo ? static_cast<double>(offsets[hi] - offsets[lo]) / static_cast<double>(hi - lo) : 0.0);
    ne += w->n_evals;
    nh += w->n_hess;
    nn_w += w->mean_neighbors * evals_pts;
    pts_w += evals_pts;
    steps = std::max(steps, w->batch_lock_steps);
  }
  h->n_evals = static_cast<int>(std::min<long long>(ne, INT32_MAX));
  h->n_hess = static_cast<int>(std::min<long long>(nh, INT32_MAX));
  h->mean_neighbors = pts_w > 0 ? nn_w / pts_w : 0.0;
  h->batch_lock_steps = steps;
  h->batch_groups = static_cast<int>(groups);
  return NDT_OK;
}

ndt_status ndt_align_batch(ndt_handle h, const void* pts, const size_t* offsets, size_t n_scans, size_t stride,
                           const float* guesses, float* final_T, int* conv, int* iters, double* tprob) {
  return align_batch_grouped(h, pts, offsets, n_scans, stride, false, guesses, final_T, conv, iters, tprob);
}
ndt_status ndt_align_batch_device(ndt_handle h, const void* pts, const size_t* offsets, size_t n_scans, size_t stride,
                                  const float* guesses, float* final_T, int* conv, int* iters, double* tprob) {
  return align_batch_grouped(h, pts, offsets, n_scans, stride, true, guesses, final_T, conv, iters, tprob);
}
ndt_status ndt_align_batch_sharded(ndt_handle h, const void* pts, const size_t* offsets, size_t n_local, size_t first_scan,
                                   size_t total_scans, size_t stride, const float* guesses, float* final_T, int* conv, int* iters,
                                   double* tprob) {
  if (total_scans == 0) return fail(NDT_ERR_INVALID, "total_scans must be > 0");
  return align_batch_impl(h, pts, offsets, n_local, stride, false, guesses, final_T, conv, iters, tprob, first_scan, total_scans);
}
ndt_status ndt_align_batch_sharded_device(ndt_handle h, const void* d_pts, const size_t* offsets, size_t n_local, size_t first_scan,
                                          size_t total_scans, size_t stride, const float* guesses, float* final_T, int* conv,
                                          int* iters, double* tprob) {
  if (total_scans == 0) return fail(NDT_ERR_INVALID, "total_scans must be > 0");
  return align_batch_impl(h, d_pts, offsets, n_local, stride, true, guesses, final_T, conv, iters, tprob, first_scan, total_scans);
}

void ndt_host_thread_budget(int* affinity_cpus, double* quota_cpus, int* local_world_size) {
  host_thread_budget(affinity_cpus, quota_cpus, local_world_size);
}
void ndt_host_thread_plan(int affinity_cpus, double quota_cpus, int local_world_size, int* pool_threads, int* max_batch_groups) {
  host_thread_plan(affinity_cpus, quota_cpus, local_world_size, pool_threads, max_batch_groups);
}

ndt_status ndt_set_batch_groups(ndt_handle h, int n_groups) {
  if (!h || n_groups < 0) return fail(NDT_ERR_INVALID, "bad arguments");
  h->batch_groups_wanted = n_groups;
  return NDT_OK;
}

ndt_status ndt_set_allreduce(ndt_handle h, ndt_allreduce_fn fn, void* user, int on_device) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  h->allreduce = fn;
  h->allreduce_user = user;
  h->allreduce_on_device = on_device;
  return NDT_OK;
}

// ---- communicator ------------------------------------------------------------------------------------
ndt_status ndt_comm_get_unique_id(void* id_out) {
  if (!id_out) return fail(NDT_ERR_INVALID, "null id buffer");
  static_assert(sizeof(ncclUniqueId) == NDT_COMM_ID_BYTES, "NDT_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
  Rccl* r = rccl();
  if (!r) return fail(NDT_ERR_COMM, "librccl is not available");
  ncclUniqueId id;
  const ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) return rccl_fail("ncclGetUniqueId", e);
  std::memcpy(id_out, &id, sizeof(id));
  return NDT_OK;
}

ndt_status ndt_comm_init_rank(ndt_handle h, const void* id, int rank, int world_size) {
  if (!h || !id || world_size < 1 || rank < 0 || rank >= world_size) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = ensure_device(h);
  if (s) return s;
  Rccl* r = rccl();
  if (!r) return fail(NDT_ERR_COMM, "librccl is not available");
  comm_release(h);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  const ncclResult_t e = r->CommInitRank(&c, world_size, uid, rank);  // on the handle's device (set by ensure_device)
  if (e != ncclSuccess) return rccl_fail("ncclCommInitRank", e);
  h->comm = c;
  h->comm_rank = rank;
  h->comm_world = world_size;
  h->comm_collectives = 0;
  return NDT_OK;
}

ndt_status ndt_comm_destroy(ndt_handle h) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  if (h->device_ready) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  comm_release(h);
  return NDT_OK;
}

ndt_status ndt_comm_stats(ndt_handle h, int* rank, int* world_size, long long* n_collectives, int* lock_steps) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  if (rank) *rank = h->comm ? h->comm_rank : -1;
  if (world_size) *world_size = h->comm ? h->comm_world : 0;
  if (n_collectives) *n_collectives = h->comm_collectives;
  if (lock_steps) *lock_steps = h->batch_lock_steps;
  return NDT_OK;
}

}  // extern "C"
