// ndt_eval.hip -- one evaluation of a single scan (launch path), the host side of the persistent evaluation server (mailbox protocol,
// see ndt_latency.hip), ndt_align, the ndt_eval* inspection entry points, diagnostics and self-tests.
// (split out of the former single C-ABI unit; shared state in ndt_internal.hpp)
#include "ndt_internal.hpp"

#include <atomic>

namespace ndtc {

void colmajor_to_T12(const float* m, float* T12) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) T12[r * 4 + c] = m[c * 4 + r];
}

// squared KDTREE radius as [PCL]/[FLANN] see it: radius = resolution_ (f32 -> f64), squared, then f32
float kd_radius2(float resolution) {
  const double r = static_cast<double>(resolution);
  return static_cast<float>(r * r);
}

void fill_eval_params(const ndt::EvalRequest& rq, const ndt::Gauss& gs, float kd_r2, ndt::EvalParams& P) {
  colmajor_to_T12(rq.T, P.T);
  ndt::AngleDerivs ad;
  ndt::angle_derivatives(rq.p, ad);
  std::memcpy(P.j, ad.j, sizeof(P.j));
  std::memcpy(P.h, ad.h, sizeof(P.h));
  P.d1 = gs.d1;
  P.d2 = static_cast<float>(gs.d2);
  std::memcpy(&P.pad, &kd_r2, sizeof(float));
}

void fill_h64_params(const ndt::EvalRequest& rq, const ndt::Gauss& gs, float kd_r2, ndt::Hess64Params& P) {
  colmajor_to_T12(rq.T, P.T);
  ndt::AngleDerivs ad;
  ndt::angle_derivatives(rq.p, ad);
  std::memcpy(P.jd, ad.jd, sizeof(P.jd));
  std::memcpy(P.hd, ad.hd, sizeof(P.hd));
  P.d1 = gs.d1;
  P.d2 = gs.d2;
  P.r2 = kd_r2;
}

// packed row -> EvalResult
void unpack_row(const double* row, bool have_h, ndt::EvalResult& r, double* nn) {
  r.score = row[0];
  for (int k = 0; k < 6; k++) r.g[k] = row[1 + k];
  std::memset(r.H, 0, sizeof(r.H));
  if (have_h) {
    int idx = 7;
    for (int i = 0; i < 6; i++)
      for (int j = i; j < 6; j++) {
        r.H[i * 6 + j] = row[idx];
        r.H[j * 6 + i] = row[idx];
        idx++;
      }
  }
  if (nn) *nn = row[28];
}

ndt_status check_ready(ndt_context* h) {
  if (!h->grid || !h->target) return fail(NDT_ERR_NO_INPUT, "no input target: call ndt_set_input_target first");
  if (!h->source) return fail(NDT_ERR_NO_INPUT, "no input source: call ndt_set_input_source first");
  return ensure_device(h);
}

// one evaluation of a single scan; blocks until the result is on the host
ndt_status evaluate_single(ndt_context* h, const ndt::EvalRequest& rq, ndt::EvalResult& res, double* nn_total) {
  const int n = h->source->k2_n();
  const float4* src = h->source->k2_pts();
  const ndt::Gauss gs = ndt::gauss_constants(h->resolution, h->outlier_ratio);
  ndt_status s = ensure_host_rows(h, 1);
  if (s) return s;
  if (h->grid->empty || ((h->source->n == 0 || n == 0) && !h->comm)) {  // nothing contributes (an empty SHARD still joins the collective)
    std::memset(&res, 0, sizeof(res));
    if (nn_total) *nn_total = 0;
    return NDT_OK;
  }
  static const bool spin_wait = [] { const char* v = getenv("NDT_SPIN_WAIT"); return v ? atoi(v) != 0 : true; }();
  static const bool fuse = [] { const char* v = getenv("NDT_K2_FUSED"); return v ? atoi(v) != 0 : true; }();
  const bool fused = fuse && spin_wait && rq.kind != ndt::EVAL_HESSIAN_F64 && ndt::derivative_variant() == 0 && !h->allreduce && !h->comm;
  const int nblk = fused ? ndt::fused_blocks(n, h->cu_count, h->cu_partition != 0) : ndt::derivative_blocks(n, h->search);
  HIP_TRY(h->partials.reserve(static_cast<size_t>(nblk) * ndt::kEvalStride));
  if (!h->ticket.p) {
    HIP_TRY(h->ticket.reserve(32 * 17));  // top counter + 16 shard counters, one per 128-B line (k_derivatives_fused)
    HIP_TRY(hipMemsetAsync(h->ticket.p, 0, 32 * 17 * sizeof(unsigned), h->stream));
  }
  const ndt::GridView gv = h->grid->view();
  const auto tp0 = std::chrono::steady_clock::now();
  unsigned long long seq = 0;
  if (h->profiling) HIP_TRY(hipEventRecord(h->ev_a, h->stream));
  if (rq.kind == ndt::EVAL_HESSIAN_F64) {
    ndt::Hess64Params P;
    fill_h64_params(rq, gs, kd_radius2(h->resolution), P);
    HIP_TRY(ndt::launch_hessian64(src, n, gv, P, h->search, nullptr, nullptr, 1, nblk, nblk, h->partials.p, h->stream));
  } else {
    ndt::EvalParams P;
    fill_eval_params(rq, gs, kd_radius2(h->resolution), P);
    if (fused) {
      seq = ++h->eval_seq;
      HIP_TRY(ndt::launch_derivatives_fused(src, n, gv, P, h->search, rq.kind == ndt::EVAL_WITH_HESSIAN, nblk, ndt::points_per_block(n, h->cu_count),
                                            h->partials.p, h->ticket.p, h->host_pub, seq, h->stream));
    } else {
      HIP_TRY(ndt::launch_derivatives(src, n, gv, P, h->search, rq.kind == ndt::EVAL_WITH_HESSIAN, nullptr, nullptr, 1, nblk, nblk,
                                      h->partials.p, h->stream));
    }
  }
  if (h->profiling) HIP_TRY(hipEventRecord(h->ev_b, h->stream));
  if (spin_wait && !h->profiling) {
    // Latency path: the result row and then a sequence number are written straight into pinned
    // host memory; poll it instead of paying a stream synchronisation per evaluation.
    if (!fused) {
      seq = ++h->eval_seq;
      if (h->comm) {
        // point-sharded scan: this rank's row -> in-place SUM over the ranks (RCCL, same stream) -> pinned host row
        HIP_TRY(h->batch_out.reserve(ndt::kEvalStride));
        HIP_TRY(ndt::launch_reduce(h->partials.p, nblk, 1, nullptr, h->batch_out.p, h->stream));
        ndt_status sc = comm_allreduce(h, h->batch_out.p, ndt::kEvalStride);
        if (sc) return sc;
        HIP_TRY(ndt::launch_publish_rows(h->batch_out.p, 1, h->host_result, seq, h->stream));
      } else {
        HIP_TRY(ndt::launch_reduce(h->partials.p, nblk, 1, nullptr, h->host_result, h->stream, seq));
      }
    }
    const auto tp1 = std::chrono::steady_clock::now();
    h->t_launch += std::chrono::duration<double>(tp1 - tp0).count();
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(h->host_result) + (ndt::kEvalStride - 1);
    auto arrived = [&] { return fused ? pub_ready(h->host_pub, seq) : __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq; };
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (!arrived()) {
      __builtin_ia32_pause();
      if ((++spins & 0xFFFF) == 0) {
        if (hipStreamQuery(h->stream) != hipErrorNotReady) {  // finished (or failed) without the flag
          HIP_TRY(hipStreamSynchronize(h->stream));
          if (arrived()) break;
          return fail(NDT_ERR_HIP, "evaluation finished without publishing its result");
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20))
          return fail(NDT_ERR_HIP, "timed out waiting for the evaluation result");
      }
    }
    if (fused) pub_gather(h->host_pub, h->host_result);
    h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp1).count();
  } else {
    if (!fused && h->comm) {
      HIP_TRY(h->batch_out.reserve(ndt::kEvalStride));
      HIP_TRY(ndt::launch_reduce(h->partials.p, nblk, 1, nullptr, h->batch_out.p, h->stream));
      ndt_status sc = comm_allreduce(h, h->batch_out.p, ndt::kEvalStride);
      if (sc) return sc;
      HIP_TRY(hipMemcpyAsync(h->host_result, h->batch_out.p, ndt::kEvalStride * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    } else if (!fused) {
      HIP_TRY(ndt::launch_reduce(h->partials.p, nblk, 1, nullptr, h->host_result, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (fused) {
      if (!pub_ready(h->host_pub, seq)) return fail(NDT_ERR_HIP, "evaluation finished without publishing its result");
      pub_gather(h->host_pub, h->host_result);
    }
  }
  if (h->profiling) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    h->prof_n[rq.kind]++;
    h->prof_ms[rq.kind] += ms;
  }
  if (h->allreduce && !h->comm) {  // point-sharded scan, caller-supplied collective: sum the packed row across ranks
    if (h->allreduce(h->host_result, ndt::kEvalStride, 0, h->allreduce_user))
      return fail(NDT_ERR_COMM, "allreduce callback failed");
  }
  unpack_row(h->host_result, rq.kind != ndt::EVAL_NO_HESSIAN, res, nn_total);
  return NDT_OK;
}

// ---- persistent evaluation server (see ndt_latency.hip) -----------------------------------------
// One server per device at a time inside this process: a server needs ALL its blocks resident to
// finish a round, and two of them launched from different host threads could each hold part of the
// CUs and wait for the rest (they would recover through their time-outs, ~100 ms later).  Handles
// therefore take turns at registration granularity.  (Across processes the time-out path remains.)
std::mutex& server_device_mutex(int device) {
  static std::mutex m[64];
  return m[device & 63];
}
// registrations (ndt_align) in flight per device, this process
std::atomic<int>& active_aligns(int device) {
  static std::atomic<int> n[64];
  return n[device & 63];
}
void server_mark(ndt_context* h, bool running) {
  if (running && !h->server_running) server_device_mutex(h->device).lock();
  if (!running && h->server_running) server_device_mutex(h->device).unlock();
  h->server_running = running;
}

bool server_enabled() {
  static const bool on = [] { const char* v = getenv("NDT_PERSISTENT"); return v ? atoi(v) != 0 : true; }();
  return on;
}

ndt_status server_stop(ndt_context* h) {
  if (!h->server_running) return NDT_OK;
  ndt::server_post(h->server_host_mb, ++h->eval_seq, ndt::kServerCmdExit, nullptr, nullptr);
  server_mark(h, false);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return NDT_OK;
}

ndt_status server_start(ndt_context* h) {
  if (h->server_running) return NDT_OK;
  server_mark(h, true);  // takes this device's turn BEFORE anything is launched
  struct Undo {
    ndt_context* c;
    bool armed = true;
    ~Undo() { if (armed) server_mark(c, false); }
  } undo{h};
  const int n = h->source->k2_n();
  const size_t mb_bytes = ndt::server_mailbox_bytes();
  if (!h->server_host_mbs) {
    // Where the host posts its commands.  On a large-BAR system: fine-grained DEVICE memory, written by the CPU
    // through the BAR (posted PCIe writes) and polled by the relay wave in its own memory -- a poll of pinned
    // host memory is a PCIe read round trip per look (tools/probes/bar_probe.cpp: 2.6 -> 2.0 us per host-GPU-host
    // ping-pong).  Otherwise, or with NDT_MAILBOX=host: pinned host memory.
    const char* where = std::getenv("NDT_MAILBOX");
    hipDeviceProp_t prop;
    const bool large_bar = hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.isLargeBar != 0;
    if (large_bar && !(where && std::strcmp(where, "host") == 0) &&
        hipExtMallocWithFlags(&h->server_host_mbs, 2 * mb_bytes, hipDeviceMallocFinegrained) == hipSuccess) {
      h->server_mbs_on_device = true;
      HIP_TRY(hipMemset(h->server_host_mbs, 0, 2 * mb_bytes));
    } else {
      h->server_host_mbs = nullptr;
      (void)hipGetLastError();
      HIP_TRY(hipHostMalloc(&h->server_host_mbs, 2 * mb_bytes, hipHostMallocDefault));
      std::memset(h->server_host_mbs, 0, 2 * mb_bytes);
    }
  }
  if (!h->server_dev_mb.p) {
    HIP_TRY(h->server_dev_mb.reserve(2 * mb_bytes));
    HIP_TRY(hipMemsetAsync(h->server_dev_mb.p, 0, 2 * mb_bytes, h->stream));
  }
  h->server_flip ^= 1;
  h->server_host_mb = static_cast<unsigned char*>(h->server_host_mbs) + h->server_flip * mb_bytes;
  ndt::server_reset_mailbox(h->server_host_mb);
  void* dev_mb = h->server_dev_mb.p + h->server_flip * mb_bytes;
  // two sets of shard counters (one per 128-B line), used alternately: a launch zeroes the set of the next one
  {
    const unsigned* before = h->server_counter.p;
    HIP_TRY(h->server_counter.reserve(2 * ndt::kServerCounterWords));
    if (h->server_counter.p != before) {  // fresh storage
      HIP_TRY(hipMemsetAsync(h->server_counter.p, 0, 2 * ndt::kServerCounterWords * sizeof(unsigned), h->stream));
      h->server_counter_set = 0;
    } else {
      h->server_counter_set ^= 1;
    }
  }
  unsigned* const counter_now = h->server_counter.p + h->server_counter_set * ndt::kServerCounterWords;
  unsigned* const counter_next = h->server_counter.p + (h->server_counter_set ^ 1) * ndt::kServerCounterWords;
  // one 512-thread block per CU at most: every block must be resident for the round to complete
  const int ppb = ndt::points_per_block(n, h->cu_count);
  int nblk = std::max(1, std::min(h->cu_count > 0 ? h->cu_count : 64, (n + ppb - 1) / ppb));
  h->server_blocks = nblk;
  HIP_TRY(h->partials.reserve(static_cast<size_t>(nblk) * ndt::kEvalStride));
  HIP_TRY(ensure_host_rows(h, 1) == NDT_OK ? hipSuccess : hipErrorOutOfMemory);
  const int n_out = static_cast<int>(h->source->n);
  HIP_TRY(h->out_cloud.reserve(n_out));
  const unsigned long long idle_ticks = 2000000ull;  // 20 ms of s_memrealtime (100 MHz)
  const ndt::Gauss gs = ndt::gauss_constants(h->resolution, h->outlier_ratio);
  const float r2 = kd_radius2(h->resolution);
  int pad_bits;
  std::memcpy(&pad_bits, &r2, sizeof(int));
  h->server_timed = h->profile_server;
  if (h->server_timed) HIP_TRY(hipEventRecord(h->ev_a, h->stream));
  HIP_TRY(ndt::launch_eval_server(h->source->k2_pts(), n, h->grid->view(), h->search, h->server_host_mb, dev_mb, nblk, ppb,
                                  h->partials.p, counter_now, h->host_pub, h->eval_seq + 1, idle_ticks, gs.d1,
                                  gs.d2, pad_bits, h->source->pts.p, h->out_cloud.p, n_out, h->stream,
                                  h->server_want_dbg ? h->server_dbg.p : nullptr,
                                  (h->server_mbs_on_device && !(std::getenv("NDT_SERVER_DIRECT") && std::atoi(std::getenv("NDT_SERVER_DIRECT")) == 0)) ? 1 : 0,
                                  h->server_out_host, counter_next));
  h->server_wrote_host = h->server_out_host != nullptr;
  undo.armed = false;
  return NDT_OK;
}

// last command of a registration: the server writes the aligned cloud (source x T) and exits; the
// caller does not wait (everything later on h->stream is ordered behind the server kernel)
void server_finish(ndt_context* h, const float* T_colmajor) {
  float T12[12];
  colmajor_to_T12(T_colmajor, T12);
  ndt::server_post(h->server_host_mb, ++h->eval_seq, ndt::kServerCmdTransformExit, T12, nullptr);
  server_mark(h, false);
}

// one evaluation through the running server; *served = false means the server had given up
// (idle time-out) and the caller must use the launch path
ndt_status server_evaluate(ndt_context* h, const ndt::EvalRequest& rq, const ndt::Gauss& gs, ndt::EvalResult& res,
                           double* nn_total, bool* served) {
  *served = false;
  (void)gs;
  float T12[12];
  colmajor_to_T12(rq.T, T12);
  double cs[6];
  ndt::snapped_cos_sin(rq.p, cs);
  const unsigned long long seq = ++h->eval_seq;
  ndt::server_post(h->server_host_mb, seq, static_cast<int>(rq.kind), T12, cs);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  // one tagged row per part of the fixed-order sum comes back (ndt_latency.hip, the server's epilogue)
  const int n_parts = std::min(ndt::kServerParts, h->server_blocks);
  unsigned arrived = 0;  // bit p: part p complete
  const unsigned all = (n_parts >= 32) ? ~0u : ((1u << n_parts) - 1u);
  auto parts_ready = [&] {
    for (int p = 0; p < n_parts; p++)
      if (!(arrived & (1u << p)) && pub_ready(h->host_pub + static_cast<size_t>(p) * ndt::kPublishSlots, seq)) arrived |= 1u << p;
    return arrived == all;
  };
  static const bool gap_probe = [] { const char* v = getenv("NDT_TIMING"); return v && atoi(v) >= 2; }();
  auto t_prev = t0;
  while (!parts_ready()) {
    __builtin_ia32_pause();
    if (gap_probe) {  // diagnostics: the longest time this thread was away from its poll loop
      const auto t_now = std::chrono::steady_clock::now();
      const double gap = std::chrono::duration<double>(t_now - t_prev).count();
      if (gap > h->t_gap) h->t_gap = gap;
      t_prev = t_now;
    }
    if ((++spins & 0x3FFF) == 0) {
      if (ndt::server_dead_word(h->server_host_mb) != 0 || hipStreamQuery(h->stream) != hipErrorNotReady) {
        // the server left (idle time-out or error): drain and let the caller relaunch
        static const bool timing = [] { const char* v = getenv("NDT_TIMING"); return v && atoi(v) != 0; }();
        if (timing)
          std::fprintf(stderr, "[ndt timing] server left: seq %llu dead word %llu arrived %#x of %#x after %.1f us, kind %d, blocks %d\n", seq,
                       ndt::server_dead_word(h->server_host_mb), arrived, all,
                       std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6, static_cast<int>(rq.kind), h->server_blocks);
        server_mark(h, false);
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (parts_ready()) break;
        ndt::server_reset_mailbox(h->server_host_mb);
        return NDT_OK;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20))
        return fail(NDT_ERR_HIP, "timed out waiting for the evaluation server");
    }
  }
  {  // second stage of the fixed-order sum: t = ((0 + part 0) + part 1) + ... over all kServerParts (absent parts are 0.0)
    double part[ndt::kEvalStride];
    for (int k = 0; k < ndt::kEvalStride; k++) h->host_result[k] = 0.0;
    for (int p = 0; p < ndt::kServerParts; p++) {
      if (p < n_parts) pub_gather(h->host_pub + static_cast<size_t>(p) * ndt::kPublishSlots, part);
      for (int k = 0; k < ndt::kEvalStride; k++) h->host_result[k] += (p < n_parts) ? part[k] : 0.0;
    }
  }
  unpack_row(h->host_result, rq.kind != ndt::EVAL_NO_HESSIAN, res, nn_total);
  *served = true;
  return NDT_OK;
}

}  // namespace ndtc

extern "C" {

ndt_status ndt_align(ndt_handle h, const float* guess, float* final_transformation, int* has_converged,
                     int* final_num_iteration, double* transformation_probability, void* out_cloud,
                     size_t out_stride_bytes) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  ndt_status s = check_ready(h);
  if (s) return s;
  s = maybe_compact_records(h, false);  // a grid that is registered against a second time gets dense, cell-ordered records
  if (s) return s;
  ndt::ScanSolver solver;
  size_t n_total = h->source->n;
  if (h->comm) {  // point-sharded scan: transformation_probability = score / N over ALL shards
    ndt_status sc = ensure_host_rows(h, 1);
    if (sc) return sc;
    HIP_TRY(h->batch_out.reserve(ndt::kEvalStride));
    double row[ndt::kEvalStride] = {static_cast<double>(h->source->n)};
    HIP_TRY(hipMemcpyAsync(h->batch_out.p, row, sizeof(row), hipMemcpyHostToDevice, h->stream));
    sc = comm_allreduce(h, h->batch_out.p, ndt::kEvalStride);
    if (sc) return sc;
    HIP_TRY(hipMemcpyAsync(row, h->batch_out.p, sizeof(row), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    n_total = static_cast<size_t>(row[0]);
  }
  solver.start(guess, n_total, solver_params(h));
  double nn = 0;
  const ndt::Gauss gs_align = ndt::gauss_constants(h->resolution, h->outlier_ratio);
  struct ServerGuard {  // whatever path leaves align, the server is told to exit
    ndt_context* c;
    ~ServerGuard() { if (c->server_running) (void)server_stop(c); }
  } server_guard{h};
  // Several host threads registering on handles of their own (one device): the server holds every CU it runs on for a whole
  // registration and handles take turns at it, so N threads get one thread's throughput; one launch per evaluation lets
  // the registrations' kernels share the chip (measured, 100k-point scans, 1 / 2 / 4 threads: 1 880 / 1 740 / 1 940 reg/s
  // through the server, 1 290 / 2 090 / 2 880 through the launch path).  A registration that starts while another one is
  // in flight on the device therefore uses the launch path; a lone caller keeps the server.  Same bits either way.
  struct ActiveAligns {
    std::atomic<int>* n;
    int mine;
    explicit ActiveAligns(int device) : n(&active_aligns(device)), mine(n->fetch_add(1) + 1) {}
    ~ActiveAligns() { n->fetch_sub(1); }
  } active(h->device);
  const bool alone = active.mine == 1 || h->persistent > 0;  // (ndt_set_evaluation_path(h, 1) insists on the server)
  // Multi-million-point scans: the launch path's kernel runs two blocks per CU where the server has one, and at ~60 us per
  // evaluation a launch is cheap -- 10M-point target at 0.5 m, server / launch path: 200k points 0.59 / 0.77 ms, 500k
  // 0.81 / 0.95, 1M 1.23 / 1.23, 2M 2.07 / 1.83 (NOTES.md, round 2).
  const bool big_scan = h->source->k2_n() >= 1500000 && h->persistent <= 0;
  const bool use_server = (h->persistent < 0 ? server_enabled() : h->persistent != 0) && alone && !big_scan && !h->profiling && !h->allreduce && !h->comm &&
                          ndt::derivative_variant() == 0 && h->source->k2_n() > 0 && !h->grid->empty;
  // the caller wants the aligned cloud on the host: the server's last command writes it into page-locked memory as well
  h->server_out_host = nullptr;
  h->server_wrote_host = false;
  if (out_cloud && h->source->n) {
    if (out_stride_bytes < 16) return fail(NDT_ERR_INVALID, "out_stride_bytes must be >= 16");
    const size_t bytes = h->source->n * sizeof(float4);
    if (h->out_pinned_bytes < bytes) {
      if (h->out_pinned) (void)hipHostFree(h->out_pinned);
      h->out_pinned = nullptr;
      h->out_pinned_bytes = 0;
      HIP_TRY(hipHostMalloc(&h->out_pinned, bytes + bytes / 4, hipHostMallocDefault));
      h->out_pinned_bytes = bytes + bytes / 4;
    }
    if (use_server) h->server_out_host = static_cast<float4*>(h->out_pinned);
  }
  while (!solver.done()) {
    ndt::EvalResult r;
    double nn_step = 0;
    const bool counts_neighbors = solver.request().kind != ndt::EVAL_HESSIAN_F64;
    bool served = false;
    if (use_server) {
      const auto tl0 = std::chrono::steady_clock::now();
      s = server_start(h);
      if (s) return s;
      const auto tl1 = std::chrono::steady_clock::now();
      s = server_evaluate(h, solver.request(), gs_align, r, &nn_step, &served);
      if (s) return s;
      h->t_launch += std::chrono::duration<double>(tl1 - tl0).count();
      h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tl1).count();
      if (h->t_fill == 0) h->t_fill = std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();  // first evaluation, launch included
    }
    if (!served) {
      s = evaluate_single(h, solver.request(), r, &nn_step);
      if (s) return s;
    }
    if (counts_neighbors) nn = nn_step;
    static const bool trace = [] { const char* v = getenv("NDT_ALIGN_TRACE"); return v && atoi(v) != 0; }();
    if (trace) {
      const ndt::EvalRequest& rq = solver.request();
      std::fprintf(stderr, "[trace] kind %d p %.17g %.17g %.17g %.17g %.17g %.17g -> score %.17g g0 %.17g g5 %.17g H00 %.17g nn %.17g\n", static_cast<int>(rq.kind),
                   rq.p[0], rq.p[1], rq.p[2], rq.p[3], rq.p[4], rq.p[5], r.score, r.g[0], r.g[5], r.H[0], nn_step);
    }
    const auto ts0 = std::chrono::steady_clock::now();
    solver.feed(r);
    h->t_solver += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
  }
  static const bool timing = [] { const char* v = getenv("NDT_TIMING"); return v && atoi(v) != 0; }();
  if (timing) {
    std::fprintf(stderr, "[ndt timing] evals=%d launch=%.1fus wait=%.1fus solver=%.1fus first-eval=%.1fus longest-poll-gap=%.1fus (per align)\n",
                 solver.n_evals + solver.n_hess, h->t_launch * 1e6, h->t_wait * 1e6, h->t_solver * 1e6, h->t_fill * 1e6, h->t_gap * 1e6);
    h->t_launch = h->t_wait = h->t_solver = h->t_fill = h->t_gap = 0;
  }
  std::memcpy(h->final_T, solver.final_T, sizeof(h->final_T));
  h->converged = solver.converged ? 1 : 0;
  h->nr_iterations = solver.nr_iterations;
  h->trans_probability = solver.trans_probability;
  h->n_evals = solver.n_evals;
  h->n_hess = solver.n_hess;
  h->mean_neighbors = n_total ? nn / static_cast<double>(n_total) : 0.0;
  // the aligned cloud = source transformed by the last trial's matrix (trans_cloud of :833/:878)
  const int n = static_cast<int>(h->source->n);
  bool wrote_host_copy = false;
  if (h->server_running) {
    wrote_host_copy = h->server_wrote_host;
    server_finish(h, h->final_T);  // the server writes it on its way out
    if (h->server_timed) {  // ndt_profile_enable(h, 2): duration of this registration's kernel
      HIP_TRY(hipEventRecord(h->ev_b, h->stream));
      HIP_TRY(hipEventSynchronize(h->ev_b));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
      h->prof_n[3]++;
      h->prof_ms[3] += ms;
      h->server_timed = false;
    }
  } else {
    HIP_TRY(h->out_cloud.reserve(n));
    float T12[12];
    colmajor_to_T12(h->final_T, T12);
    HIP_TRY(ndt::launch_transform(h->source->pts.p, n, T12, h->out_cloud.p, h->stream));
  }
  h->out_n = n;
  if (out_cloud && n) {
    // device -> page-locked staging (one contiguous DMA) -> the caller's records: a strided copy straight into the
    // caller's pageable buffer goes through the runtime's own staging in small pieces (measured 74 us for the
    // 256 KB of a 16k-point cloud, 2x the rest of the registration)
    const size_t bytes = static_cast<size_t>(n) * sizeof(float4);
    // (the staging buffer was sized at the top of ndt_align)  The server that finished THIS registration has written the
    // cloud there itself; any other path copies it over
    if (!(wrote_host_copy)) HIP_TRY(hipMemcpyAsync(h->out_pinned, h->out_cloud.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (out_stride_bytes == sizeof(float4)) {
      std::memcpy(out_cloud, h->out_pinned, bytes);
    } else {
      const unsigned char* src = static_cast<const unsigned char*>(h->out_pinned);
      unsigned char* dst = static_cast<unsigned char*>(out_cloud);
      for (int i = 0; i < n; i++) std::memcpy(dst + static_cast<size_t>(i) * out_stride_bytes, src + static_cast<size_t>(i) * sizeof(float4), sizeof(float4));
    }
  }
  // without a host copy nothing waits here: the cloud is complete in stream order (ndt_get_output_device
  // synchronises before handing the pointer out)
  return ndt_get_result(h, final_transformation, has_converged, final_num_iteration, transformation_probability);
}

ndt_status ndt_get_output_device(ndt_handle h, const void** d_cloud, size_t* n) {
  if (!h || !d_cloud || !n) return fail(NDT_ERR_INVALID, "bad arguments");
  if (h->device_ready) HIP_TRY(hipStreamSynchronize(h->stream));  // ndt_align does not wait for the cloud
  *d_cloud = h->out_cloud.p;
  *n = h->out_n;
  return NDT_OK;
}

// ---- inspection ------------------------------------------------------------
static ndt_status eval_impl(ndt_handle h, const float* T, const double* p, double* score, double* g, double* H,
                            double* mean_nn, int kind) {
  if (!h || !p) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = check_ready(h);
  if (s) return s;
  ndt::EvalRequest rq;
  rq.kind = static_cast<ndt::EvalKind>(kind);
  std::memcpy(rq.p, p, sizeof(rq.p));
  if (T) std::memcpy(rq.T, T, sizeof(rq.T));
  else ndt::pose_to_matrix(p, rq.T);
  ndt::EvalResult r;
  double nn = 0;
  s = evaluate_single(h, rq, r, &nn);
  if (s) return s;
  if (score) *score = r.score;
  if (g) std::memcpy(g, r.g, sizeof(r.g));
  if (H) std::memcpy(H, r.H, sizeof(r.H));
  if (mean_nn) *mean_nn = h->source->n ? nn / static_cast<double>(h->source->n) : 0.0;
  return NDT_OK;
}

ndt_status ndt_eval(ndt_handle h, const double* p, double* score, double* g, double* H, double* mean_nn) {
  return eval_impl(h, nullptr, p, score, g, H, mean_nn, H ? ndt::EVAL_WITH_HESSIAN : ndt::EVAL_NO_HESSIAN);
}
ndt_status ndt_eval_with_matrix(ndt_handle h, const float* T, const double* p, double* score, double* g, double* H,
                                double* mean_nn) {
  if (!T) return fail(NDT_ERR_INVALID, "null matrix");
  return eval_impl(h, T, p, score, g, H, mean_nn, H ? ndt::EVAL_WITH_HESSIAN : ndt::EVAL_NO_HESSIAN);
}
ndt_status ndt_eval_hessian_f64(ndt_handle h, const double* p, double* H) {
  if (!H) return fail(NDT_ERR_INVALID, "null output");
  return eval_impl(h, nullptr, p, nullptr, nullptr, H, nullptr, ndt::EVAL_HESSIAN_F64);
}

ndt_status ndt_diag_stamps(ndt_handle h, const double* p, unsigned long long* stamps, size_t* n_waves) {
  if (!h || !p || !stamps || !n_waves) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = check_ready(h);
  if (s) return s;
  const int n = h->source->k2_n();
  if (n == 0 || h->grid->empty) { *n_waves = 0; return NDT_OK; }
  const int nblk = ndt::derivative_blocks(n, NDT_DIRECT1);
  const size_t waves = static_cast<size_t>(nblk) * 4;
  if (*n_waves < waves) return fail(NDT_ERR_INVALID, "stamp buffer too small");
  ndt::EvalRequest rq;
  rq.kind = ndt::EVAL_WITH_HESSIAN;
  std::memcpy(rq.p, p, sizeof(rq.p));
  ndt::pose_to_matrix(p, rq.T);
  ndt::EvalParams P;
  fill_eval_params(rq, ndt::gauss_constants(h->resolution, h->outlier_ratio), kd_radius2(h->resolution), P);
  DevBuf<unsigned long long> d;
  HIP_TRY(d.reserve(waves * 8));
  HIP_TRY(h->partials.reserve(static_cast<size_t>(nblk) * ndt::kEvalStride));
  HIP_TRY(ndt::launch_derivatives_stamped(h->source->k2_pts(), n, h->grid->view(), P, nblk, h->partials.p, d.p, h->stream));
  HIP_TRY(hipMemcpyAsync(stamps, d.p, waves * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *n_waves = waves;
  return NDT_OK;
}

// Diagnostic: `rounds` evaluations at pose p with the host out of the loop (k_selfdrive: the last arriver adds the part
// sums and posts the next command itself).  us[0]: per round without the per-point body (protocol only), us[1]: with the
// with-Hessian body.  What a device-side Newton / More-Thuente step would start from, before the solver's own time.
ndt_status ndt_diag_selfdrive(ndt_handle h, const double* p, int rounds, double* us) {
  if (!h || !p || !us || rounds <= 0) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = check_ready(h);
  if (s) return s;
  if (h->source->k2_n() == 0 || h->grid->empty) return fail(NDT_ERR_INVALID, "empty inputs");
  if (h->search != 2) return fail(NDT_ERR_INVALID, "DIRECT7 only");
  s = server_stop(h);
  if (s) return s;
  std::lock_guard<std::mutex> turn(server_device_mutex(h->device));
  const int n = h->source->k2_n();
  const int ppb = ndt::points_per_block(n, h->cu_count);
  const int nblk = std::max(1, std::min(h->cu_count > 0 ? h->cu_count : 64, (n + ppb - 1) / ppb));
  const size_t mb_bytes = ndt::server_mailbox_bytes();
  DevBuf<unsigned char> mb;
  DevBuf<double> parts;
  HIP_TRY(mb.reserve(mb_bytes));
  HIP_TRY(parts.reserve(static_cast<size_t>(ndt::kServerParts) * ndt::kEvalStride));
  HIP_TRY(h->partials.reserve(static_cast<size_t>(nblk) * ndt::kEvalStride));
  HIP_TRY(h->server_counter.reserve(2 * ndt::kServerCounterWords));
  HIP_TRY(ensure_host_rows(h, 1) == NDT_OK ? hipSuccess : hipErrorOutOfMemory);
  const ndt::Gauss gs = ndt::gauss_constants(h->resolution, h->outlier_ratio);
  const float r2 = kd_radius2(h->resolution);
  int pad_bits;
  std::memcpy(&pad_bits, &r2, sizeof(int));
  float T[16], T12[12];
  ndt::pose_to_matrix(p, T);
  colmajor_to_T12(T, T12);
  double cs[6];
  ndt::snapped_cos_sin(p, cs);
  alignas(64) unsigned char staged[1024];
  if (mb_bytes > sizeof(staged)) return fail(NDT_ERR_INVALID, "mailbox larger than the staging block");
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  for (int with_body = 0; with_body < 2; with_body++) {
    const unsigned long long first_seq = h->eval_seq + 1;
    h->eval_seq += static_cast<unsigned long long>(rounds);
    std::memset(staged, 0, sizeof(staged));
    ndt::server_post(staged, first_seq, 0, T12, cs);
    HIP_TRY(hipMemcpyAsync(mb.p, staged, mb_bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->server_counter.p, 0, 32 * (1 + ndt::kServerParts) * sizeof(unsigned), h->stream));
    HIP_TRY(hipEventRecord(e0, h->stream));
    HIP_TRY(ndt::launch_selfdrive(h->source->k2_pts(), n, h->grid->view(), h->search, mb.p, nblk, ppb, h->partials.p, h->server_counter.p, parts.p,
                                  h->host_pub, first_seq, rounds, with_body, gs.d1, gs.d2, pad_bits, h->stream));
    HIP_TRY(hipEventRecord(e1, h->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    us[with_body] = static_cast<double>(ms) * 1e3 / rounds;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  // (the evaluation server expects both of its counter sets at zero)
  HIP_TRY(hipMemsetAsync(h->server_counter.p, 0, 2 * ndt::kServerCounterWords * sizeof(unsigned), h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  // the last round's row must be what the evaluation paths compute at this pose (the caller compares with ndt_eval)
  if (!pub_ready(h->host_pub, h->eval_seq)) return fail(NDT_ERR_HIP, "self-driven rounds did not publish their last row");
  pub_gather(h->host_pub, h->host_result);
  return NDT_OK;
}

ndt_status ndt_diag_server_roundtrip(ndt_handle h, const double* p, int n_iter, double* us) {
  if (!h || !p || !us || n_iter <= 0) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = check_ready(h);
  if (s) return s;
  if (h->source->k2_n() == 0 || h->grid->empty) return fail(NDT_ERR_INVALID, "empty inputs");
  const ndt::Gauss gs = ndt::gauss_constants(h->resolution, h->outlier_ratio);
  ndt::EvalRequest rq;
  std::memcpy(rq.p, p, sizeof(rq.p));
  ndt::pose_to_matrix(p, rq.T);
  HIP_TRY(h->server_dbg.reserve(8 + 10 * 1024));
  HIP_TRY(hipMemsetAsync(h->server_dbg.p, 0, (8 + 10 * 1024) * sizeof(unsigned long long), h->stream));
  h->server_want_dbg = true;
  s = server_start(h);
  h->server_want_dbg = false;
  if (s) return s;
  const int kinds[3] = {3, 1, 0};
  // NDT_DIAG_ONLY_KIND=3|1|0 (development aid, tools/pmc_valu_split.sh): only that kind of round in this launch, so that a
  // counter pass around the process attributes the launch's instructions to one kind
  static const int only_kind = [] { const char* e = getenv("NDT_DIAG_ONLY_KIND"); return e ? atoi(e) : -1; }();
  us[0] = us[1] = us[2] = 0.0;
  for (int v = 0; v < 3; v++) {
    if (only_kind >= 0 && kinds[v] != only_kind) continue;
    rq.kind = static_cast<ndt::EvalKind>(kinds[v]);
    double total = 0;
    for (int it = 0; it < n_iter + 5; it++) {
      ndt::EvalResult r;
      bool served = false;
      const auto t0 = std::chrono::steady_clock::now();
      s = server_evaluate(h, rq, gs, r, nullptr, &served);
      const auto t1 = std::chrono::steady_clock::now();
      if (s) { (void)server_stop(h); return s; }
      if (!served) { (void)server_stop(h); return fail(NDT_ERR_HIP, "server stopped serving"); }
      if (it >= 5) total += std::chrono::duration<double>(t1 - t0).count();
    }
    us[v] = total / n_iter * 1e6;
  }
  s = server_stop(h);
  if (s) return s;
  {  // device-side stamps of the LAST round (with Hessian), s_memrealtime ticks of 10 ns
    std::vector<unsigned long long> d(8 + 10 * 1024);
    HIP_TRY(hipMemcpy(d.data(), h->server_dbg.p, d.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const int ppb_diag = ndt::points_per_block(h->source->k2_n(), h->cu_count);
    const int nblk = std::max(1, std::min(h->cu_count > 0 ? h->cu_count : 64, (h->source->k2_n() + ppb_diag - 1) / ppb_diag));
    unsigned long long got_min = ~0ull, got_max = 0, tk_min = ~0ull, tk_max = 0;
    for (int b = 0; b < nblk; b++) {
      got_min = std::min(got_min, d[8 + 2 * b]); got_max = std::max(got_max, d[8 + 2 * b]);
      tk_min = std::min(tk_min, d[9 + 2 * b]); tk_max = std::max(tk_max, d[9 + 2 * b]);
    }
    auto us_of = [&](unsigned long long t) { return (static_cast<double>(t) - static_cast<double>(d[0])) * 0.01; };
    if (const char* path = getenv("NDT_DIAG_DUMP")) {  // raw stamps for offline analysis
      if (FILE* f = std::fopen(path, "wb")) {
        std::fwrite(d.data(), sizeof(unsigned long long), d.size(), f);
        std::fclose(f);
      }
    }
    std::fprintf(stderr, "[server stamps, us after the relay (direct mailbox: block 0) saw the command] relayed %.2f | blocks have params %.2f..%.2f | tickets %.2f..%.2f | final sum starts %.2f | published %.2f  (%d blocks)\n",
                 us_of(d[1]), us_of(got_min), us_of(got_max), us_of(tk_min), us_of(tk_max), us_of(d[2]), us_of(d[3]), nblk);
  }
  return NDT_OK;
}

// Liveness self-test of the evaluation server: serve one evaluation, let the host go quiet for stall_ms (the
// server's patience is 20 ms: it must tell the host and leave on its own), ask again -- the request must come back
// unserved, not hang -- then evaluate through the launch path and through a FRESH server.  scores[3]: before the
// stall (server), after it (launch path), fresh server; *served_after_stall: whether the stalled server still answered.
ndt_status ndt_selftest_server_idle(ndt_handle h, const double* p, int stall_ms, int* served_after_stall, double* scores) {
  if (!h || !p || !served_after_stall || !scores) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = check_ready(h);
  if (s) return s;
  if (h->source->k2_n() == 0 || h->grid->empty) return fail(NDT_ERR_INVALID, "empty inputs");
  const ndt::Gauss gs = ndt::gauss_constants(h->resolution, h->outlier_ratio);
  ndt::EvalRequest rq;
  rq.kind = ndt::EVAL_WITH_HESSIAN;
  std::memcpy(rq.p, p, sizeof(rq.p));
  ndt::pose_to_matrix(p, rq.T);
  ndt::EvalResult r;
  bool served = false;
  s = server_start(h);
  if (s) return s;
  s = server_evaluate(h, rq, gs, r, nullptr, &served);
  if (s || !served) {
    (void)server_stop(h);
    return s ? s : fail(NDT_ERR_HIP, "server did not serve its first evaluation");
  }
  scores[0] = r.score;
  std::this_thread::sleep_for(std::chrono::milliseconds(stall_ms));
  s = server_evaluate(h, rq, gs, r, nullptr, &served);
  if (s) return s;
  *served_after_stall = served ? 1 : 0;
  if (served) {
    s = server_stop(h);
    if (s) return s;
  }
  s = evaluate_single(h, rq, r, nullptr);
  if (s) return s;
  scores[1] = r.score;
  s = server_start(h);
  if (s) return s;
  s = server_evaluate(h, rq, gs, r, nullptr, &served);
  if (s || !served) {
    (void)server_stop(h);
    return s ? s : fail(NDT_ERR_HIP, "fresh server did not serve");
  }
  scores[2] = r.score;
  return server_stop(h);
}

ndt_status ndt_selftest_reduce(ndt_handle h, int n_blocks, double* block_sums) {
  if (!h || n_blocks <= 0 || !block_sums) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = ensure_device(h);
  if (s) return s;
  DevBuf<double> d;
  HIP_TRY(d.reserve(static_cast<size_t>(n_blocks) * ndt::kEvalStride));
  HIP_TRY(hipMemsetAsync(d.p, 0, static_cast<size_t>(n_blocks) * ndt::kEvalStride * sizeof(double), h->stream));
  HIP_TRY(ndt::launch_selftest_reduce(n_blocks, d.p, h->stream));
  HIP_TRY(hipMemcpyAsync(block_sums, d.p, static_cast<size_t>(n_blocks) * ndt::kEvalStride * sizeof(double),
                         hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return NDT_OK;
}

}  // extern "C"
