// ndt_grid.hip -- cloud upload, bounding boxes, spatial ordering of scans, the K1 target grid build (VoxelGridCovariance::filter), the N1 voxel
// filter, the N2 map accumulation, getFitnessScore, calculateScore and the grid inspection entry points.
// (split out of the former single C-ABI unit; shared state in ndt_internal.hpp)
#include "ndt_internal.hpp"

#include <immintrin.h>

#include <type_traits>

namespace ndtc {

// Host counterpart of k_repack_bbox for small clouds: records of `stride` bytes (x y z first) -> dense (x, y, z, 1) in dst,
// and the two bounding boxes of the cloud -- [0]: NaN coordinates dropped (what min / max do with them), [1]: finite
// points only (pcl::getMinMax3D for a cloud that is not dense).  min / max are exact and order-free, so the boxes are the
// ones the kernel's per-block rows reduce to.
static void host_repack_bbox(const unsigned char* src, size_t n, size_t stride, float* dst, float bb_min[2][3], float bb_max[2][3]) {
  __m128 mn0 = _mm_set1_ps(FLT_MAX), mx0 = _mm_set1_ps(-FLT_MAX), mn1 = mn0, mx1 = mx0;
  const __m128 keep_xyz = _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1)), one_w = _mm_set_ps(1.0f, 0.0f, 0.0f, 0.0f);
  const __m128 abs_mask = _mm_castsi128_ps(_mm_set1_epi32(0x7fffffff)), inf = _mm_set1_ps(INFINITY);
  auto take = [&](__m128 v, float* out) {
    v = _mm_or_ps(_mm_and_ps(v, keep_xyz), one_w);
    _mm_store_ps(out, v);
    mn0 = _mm_min_ps(v, mn0);  // (min / max hand back their SECOND operand when the first is NaN)
    mx0 = _mm_max_ps(v, mx0);
    if ((_mm_movemask_ps(_mm_cmplt_ps(_mm_and_ps(v, abs_mask), inf)) & 7) == 7) {
      mn1 = _mm_min_ps(v, mn1);
      mx1 = _mm_max_ps(v, mx1);
    }
  };
  const size_t n_wide = (stride >= 16) ? n : (n ? n - 1 : 0);  // 12-B records: a 16-B load of the last one would leave the buffer
  for (size_t i = 0; i < n_wide; i++) take(_mm_loadu_ps(reinterpret_cast<const float*>(src + i * stride)), dst + 4 * i);
  for (size_t i = n_wide; i < n; i++) {
    const float* p = reinterpret_cast<const float*>(src + i * stride);
    take(_mm_set_ps(0.0f, p[2], p[1], p[0]), dst + 4 * i);
  }
  alignas(16) float a[4], b[4], c[4], d[4];
  _mm_store_ps(a, mn0); _mm_store_ps(b, mx0); _mm_store_ps(c, mn1); _mm_store_ps(d, mx1);
  for (int k = 0; k < 3; k++) {
    bb_min[0][k] = a[k]; bb_max[0][k] = b[k];
    bb_min[1][k] = c[k]; bb_max[1][k] = d[k];
  }
}

// The same, two points per instruction (AVX2; chosen at run time): 16 k points 16 -> ~9 us.  min / max are exact and order-free,
// so the boxes are the ones the one-point loop gives.
__attribute__((target("avx2"))) static void host_repack_bbox_avx2(const unsigned char* src, size_t n, size_t stride, float* dst,
                                                                  float bb_min[2][3], float bb_max[2][3]) {
  const __m256 big = _mm256_set1_ps(FLT_MAX), small = _mm256_set1_ps(-FLT_MAX);
  __m256 mn0 = big, mx0 = small, mn1 = big, mx1 = small;
  const __m256 keep_xyz = _mm256_castsi256_ps(_mm256_set_epi32(0, -1, -1, -1, 0, -1, -1, -1));
  const __m256 one_w = _mm256_set_ps(1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f);
  const __m256 abs_mask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff)), inf = _mm256_set1_ps(INFINITY);
  const size_t n_wide = (stride >= 16) ? n : (n ? n - 1 : 0);  // 12-B records: a 16-B load of the last one would leave the buffer
  size_t i = 0;
  for (; i + 2 <= n_wide; i += 2) {
    __m256 v = _mm256_castps128_ps256(_mm_loadu_ps(reinterpret_cast<const float*>(src + i * stride)));
    v = _mm256_insertf128_ps(v, _mm_loadu_ps(reinterpret_cast<const float*>(src + (i + 1) * stride)), 1);
    v = _mm256_or_ps(_mm256_and_ps(v, keep_xyz), one_w);
    _mm256_storeu_ps(dst + 4 * i, v);
    mn0 = _mm256_min_ps(v, mn0);  // (min / max hand back their SECOND operand when the first is NaN)
    mx0 = _mm256_max_ps(v, mx0);
    const int fin = _mm256_movemask_ps(_mm256_cmp_ps(_mm256_and_ps(v, abs_mask), inf, _CMP_LT_OQ));
    if ((fin & 0x77) == 0x77) {  // both points finite (the usual case)
      mn1 = _mm256_min_ps(v, mn1);
      mx1 = _mm256_max_ps(v, mx1);
    } else {
      // one of the two (or neither): the other half is replaced by the neutral values
      const __m256 lo_ok = _mm256_castsi256_ps(_mm256_set_epi32(0, 0, 0, 0, -1, -1, -1, -1)), hi_ok = _mm256_castsi256_ps(_mm256_set_epi32(-1, -1, -1, -1, 0, 0, 0, 0));
      __m256 ok = _mm256_setzero_ps();
      if ((fin & 0x07) == 0x07) ok = _mm256_or_ps(ok, lo_ok);
      if ((fin & 0x70) == 0x70) ok = _mm256_or_ps(ok, hi_ok);
      mn1 = _mm256_min_ps(_mm256_or_ps(_mm256_and_ps(ok, v), _mm256_andnot_ps(ok, big)), mn1);
      mx1 = _mm256_max_ps(_mm256_or_ps(_mm256_and_ps(ok, v), _mm256_andnot_ps(ok, small)), mx1);
    }
  }
  float rest_min[2][3], rest_max[2][3];
  host_repack_bbox(src + i * stride, n - i, stride, dst + 4 * i, rest_min, rest_max);  // the odd point, the last 12-byte record
  alignas(32) float a[8], b[8], c[8], d[8];
  _mm256_store_ps(a, mn0); _mm256_store_ps(b, mx0); _mm256_store_ps(c, mn1); _mm256_store_ps(d, mx1);
  for (int k = 0; k < 3; k++) {
    bb_min[0][k] = std::min(std::min(a[k], a[4 + k]), rest_min[0][k]);
    bb_max[0][k] = std::max(std::max(b[k], b[4 + k]), rest_max[0][k]);
    bb_min[1][k] = std::min(std::min(c[k], c[4 + k]), rest_min[1][k]);
    bb_max[1][k] = std::max(std::max(d[k], d[4 + k]), rest_max[1][k]);
  }
}

// n dense float4 records from HBM into the caller's records of out_stride bytes: device -> the handle's page-locked
// staging (one contiguous DMA) -> the caller's buffer by the CPU.  A strided copy straight into pageable memory goes
// through the runtime's own staging in small pieces (measured 74 us for 256 KB; ~0.4 ms for the 1 MB of a filtered
// 70 k-point scan, most of the N1 call).  Synchronises the handle's stream.
ndt_status download_records(ndt_context* h, const float4* d_src, size_t n, void* out, size_t out_stride) {
  if (n == 0) return NDT_OK;
  const size_t bytes = n * sizeof(float4);
  if (h->out_pinned_bytes < bytes) {
    HIP_TRY(hipStreamSynchronize(h->stream));  // (an earlier download may still be reading the old block)
    if (h->out_pinned) (void)hipHostFree(h->out_pinned);
    h->out_pinned = nullptr;
    h->out_pinned_bytes = 0;
    HIP_TRY(hipHostMalloc(&h->out_pinned, bytes + bytes / 4, hipHostMallocDefault));
    h->out_pinned_bytes = bytes + bytes / 4;
  }
  HIP_TRY(hipMemcpyAsync(h->out_pinned, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (out_stride == sizeof(float4)) {
    std::memcpy(out, h->out_pinned, bytes);
  } else {
    const unsigned char* src = static_cast<const unsigned char*>(h->out_pinned);
    unsigned char* dst = static_cast<unsigned char*>(out);
    for (size_t i = 0; i < n; i++) std::memcpy(dst + i * out_stride, src + i * sizeof(float4), sizeof(float4));
  }
  return NDT_OK;
}

// upload + repack to dense float4
ndt_status upload_cloud(ndt_context* h, const void* pts, size_t n, size_t stride, bool on_device,
                        std::shared_ptr<DeviceCloud>& out, bool by_reference) {
  if (n > 0 && !pts) return fail(NDT_ERR_INVALID, "null point buffer");
  if (stride < 12 || stride % 4) return fail(NDT_ERR_INVALID, "stride_bytes must be a multiple of 4 and >= 12");
  if (n > static_cast<size_t>(std::numeric_limits<int>::max())) return fail(NDT_ERR_INVALID, "too many points");
  ndt_status s = ensure_device(h);
  if (s) return s;
  auto c = std::make_shared<DeviceCloud>();
  // by reference: dense 16-byte records already in HBM are used where they lie -- the caller keeps them alive and unchanged
  // while they are an input of this handle (what pcl::Registration's ConstPtr inputs promise); only the boxes are computed
  const bool borrowed = by_reference && on_device && n > 0 && stride == sizeof(float4) && (reinterpret_cast<uintptr_t>(pts) & 15) == 0;
  if (by_reference && !borrowed && n > 0) return fail(NDT_ERR_INVALID, "a cloud by reference must be device memory of 16-byte records on a 16-byte boundary");
  if (borrowed) c->pts.borrow(const_cast<float4*>(static_cast<const float4*>(pts)), n);
  else HIP_TRY(c->pts.reserve(n));
  c->n = n;
  // clouds of at most this many points take the host route (NDT_HOST_STAGE_MAX, 0 = never): the repack + bounding box pass
  // costs the host ~1 ns per point, the device route a blocking pageable copy, a kernel and a wait (~35 us whatever the size)
  static const size_t host_stage_max = [] {
    const char* v = getenv("NDT_HOST_STAGE_MAX");
    return std::min<size_t>(ndt_context::kStageSlotPoints, v ? static_cast<size_t>(std::max(0, atoi(v))) : 20480);
  }();
  if (n && !on_device && n <= host_stage_max) {
    const int slot = h->stage_next;
    h->stage_next = (slot + 1) % ndt_context::kStageSlots;
    if (!h->stage_host[slot]) {
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->stage_host[slot]), ndt_context::kStageSlotPoints * sizeof(float4), hipHostMallocDefault));
      HIP_TRY(hipEventCreateWithFlags(&h->stage_done[slot], hipEventDisableTiming));
    } else {
      HIP_TRY(hipEventSynchronize(h->stage_done[slot]));  // (four uploads ago: long done)
    }
    static const bool avx2 = [] { const char* v = getenv("NDT_HOST_AVX2"); return (!v || atoi(v) != 0) && __builtin_cpu_supports("avx2"); }();
    if (avx2) host_repack_bbox_avx2(static_cast<const unsigned char*>(pts), n, stride, h->stage_host[slot], c->bb_min, c->bb_max);
    else host_repack_bbox(static_cast<const unsigned char*>(pts), n, stride, h->stage_host[slot], c->bb_min, c->bb_max);
    HIP_TRY(ndt::launch_copy_records(reinterpret_cast<const float4*>(h->stage_host[slot]), c->pts.p, static_cast<int>(n), h->stream));
    HIP_TRY(hipEventRecord(h->stage_done[slot], h->stream));
  } else if (n) {
    const void* d_src = pts;
    // NDT_ZERO_COPY=1 (measured and left off): page-locked host memory of 16-byte records read by the repack kernel itself,
    // over the link -- one kernel and a poll of its rows instead of a copy into the staging buffer, the kernel and a stream
    // synchronisation.  The kernel's reads over PCIe run at 37 GB/s (26.5 us per 1 MB scan) against the copy engine's
    // ~50 GB/s plus a 5.6 us kernel: the node loop's prefilter 0.155 against 0.13-0.145 ms per scan on one box.
    static const bool zero_copy = [] { const char* v = getenv("NDT_ZERO_COPY"); return v && atoi(v) != 0; }();
    if (!on_device && zero_copy && stride == sizeof(float4) && (reinterpret_cast<uintptr_t>(pts) & 15) == 0) {
      hipPointerAttribute_t attr{};
      if (hipPointerGetAttributes(&attr, pts) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
        d_src = attr.devicePointer;
        on_device = true;  // (for what follows: a source the device reads where it lies, copied by the kernel)
      } else {
        (void)hipGetLastError();  // pageable memory: not an error, the staging copy takes it
      }
    }
    if (!on_device) {
      HIP_TRY(h->staging.reserve(n * stride));
      HIP_TRY(hipMemcpyAsync(h->staging.p, pts, n * stride, hipMemcpyHostToDevice, h->stream));
      d_src = h->staging.p;
    }
    // repack and bounding boxes in one pass; the per-block rows come back behind the synchronisation
    // the upload needs anyway (the caller's buffer must be free to go when this returns)
    // (16-byte records: a block per CU and eight 16-byte loads in flight per thread; the rows travel over PCIe one by one,
    // so fewer, fatter blocks also mean fewer of those writes at the end of the kernel)
    const bool rec16 = stride == sizeof(float4) && (reinterpret_cast<uintptr_t>(d_src) & 15) == 0;
    const int nb = static_cast<int>(std::min<size_t>(rec16 ? 256 : 1024, (n + 255) / 256));
    if (!h->bbox_rows) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->bbox_rows), 1024 * 12 * sizeof(float), hipHostMallocDefault));
    // the kernel stores its per-block rows straight into pinned host memory (no D2H copy to queue)
    static const bool poll_rows = [] { const char* v = getenv("NDT_BBOX_POLL"); return !v || atoi(v) != 0; }();
    bool polled = false;
    if (on_device && rec16 && poll_rows) {
      // A cloud used where it lies: nothing is copied, so nothing has to be waited for but the rows themselves -- tagged
      // word by word and polled here (a stream synchronisation costs several microseconds beyond the kernel's end).
      // A device cloud the library copies: a block writes its row after its last read of the caller's records, so all rows
      // in = the caller's buffer is free; the copy's own stores are ordered before whatever this stream runs next.
      if (!h->bbox_tagged) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->bbox_tagged), 256 * 12 * sizeof(unsigned long long), hipHostMallocDefault));
        std::memset(h->bbox_tagged, 0, 256 * 12 * sizeof(unsigned long long));
      }
      if (++h->bbox_tag == 0) h->bbox_tag = 1;
      const unsigned tag = h->bbox_tag;
      HIP_TRY(ndt::launch_repack_bbox(d_src, n, stride, borrowed ? nullptr : c->pts.p, reinterpret_cast<float*>(h->bbox_tagged), nb, h->stream, tag));
      const volatile unsigned long long* w = h->bbox_tagged;
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      polled = true;
      for (int i = nb * 12 - 1; i >= 0 && polled; i--)
        while (static_cast<unsigned>(w[i]) != tag) {
          __builtin_ia32_pause();
          if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { polled = false; break; }
        }
      if (!polled) {
        // two seconds without the rows: a stream that is merely slow (a profiler serialising it, long work queued ahead)
        // or a launch that failed / a device that hung.  Let the runtime say which: after a successful synchronisation the
        // kernel HAS run and its rows are valid.
        HIP_TRY(hipStreamSynchronize(h->stream));
        polled = true;
        for (int i = 0; i < nb * 12 && polled; i++) polled = static_cast<unsigned>(w[i]) == tag;
        if (!polled) return fail(NDT_ERR_HIP, "bounding-box rows did not arrive");
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      for (int i = 0; i < nb * 12; i++) {
        const unsigned bits = static_cast<unsigned>(w[i] >> 32);
        std::memcpy(&h->bbox_rows[i], &bits, sizeof(float));
      }
    }
    if (!polled) {
      HIP_TRY(ndt::launch_repack_bbox(d_src, n, stride, borrowed ? nullptr : c->pts.p, h->bbox_rows, nb, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
    }
    const float* mm = h->bbox_rows;
    for (int b = 0; b < nb; b++)
      for (int v = 0; v < 2; v++)
        for (int k = 0; k < 3; k++) {
          c->bb_min[v][k] = std::min(c->bb_min[v][k], mm[b * 12 + v * 6 + k]);
          c->bb_max[v][k] = std::max(c->bb_max[v][k], mm[b * 12 + v * 6 + 3 + k]);
        }
  }
  out = c;
  return NDT_OK;
}

// bounding box of a dense float4 device cloud: taken from the upload when the cloud came through
// upload_cloud (no kernel, no wait), else computed here (one kernel + one host round trip)

BBox bbox_of(const DeviceCloud& c, int dense) {
  BBox b;
  const int v = dense ? 0 : 1;
  for (int k = 0; k < 3; k++) {
    b.mn[k] = c.bb_min[v][k];
    b.mx[k] = c.bb_max[v][k];
  }
  return b;
}
ndt_status bbox_compute(ndt_context* h, const float4* d_pts, int n, int dense, BBox& out) {
  const int nb = std::min(1024, (n + 255) / 256);
  DevBuf<float> d_mm;
  HIP_TRY(d_mm.reserve(static_cast<size_t>(nb) * 6));
  HIP_TRY(ndt::launch_bbox(d_pts, n, dense, d_mm.p, nb, h->stream));
  std::vector<float> mm(static_cast<size_t>(nb) * 6);
  HIP_TRY(hipMemcpyAsync(mm.data(), d_mm.p, mm.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int k = 0; k < 3; k++) {
    out.mn[k] = FLT_MAX;
    out.mx[k] = -FLT_MAX;
  }
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 3; k++) {
      out.mn[k] = std::min(out.mn[k], mm[b * 6 + k]);
      out.mx[k] = std::max(out.mx[k], mm[b * 6 + 3 + k]);
    }
  return NDT_OK;
}

// Spatial ordering of a source range: counting sort by the cell of a lattice of pitch ~resolution
// laid over the range's own bounding box (x fastest), stable inside a cell.  Rigid transforms
// preserve locality, so whatever the pose, consecutive lanes of the derivative kernels land in
// the same or adjacent target voxels.  Only the order of the f64 summation changes.
ndt_status order_range(ndt_context* h, const float4* d_pts, size_t n, float pitch, float4* d_out, size_t* n_out,
                       const BBox* known_bbox) {
  *n_out = 0;
  if (n == 0) return NDT_OK;
  hipStream_t st = h->stream;
  const int ni = static_cast<int>(n);
  BBox bb;
  if (known_bbox) bb = *known_bbox;
  else { ndt_status sb = bbox_compute(h, d_pts, ni, 0, bb); if (sb) return sb; }
  const float* min_p = bb.mn;
  const float* max_p = bb.mx;
  if (!(min_p[0] <= max_p[0])) return NDT_OK;  // no finite point
  ndt::GridGeom geo{};
  for (;; pitch *= 2.0f) {
    double cells = 1;
    for (int k = 0; k < 3; k++) {
      geo.leaf[k] = pitch;
      geo.inv_leaf[k] = 1.0f / pitch;
      geo.min_b[k] = static_cast<int>(std::floor(min_p[k] * geo.inv_leaf[k]));
      geo.max_b[k] = static_cast<int>(std::floor(max_p[k] * geo.inv_leaf[k]));
      geo.div_b[k] = geo.max_b[k] - geo.min_b[k] + 1;
      cells *= geo.div_b[k];
    }
    if (cells <= 4.0e6) break;
  }
  geo.mul[0] = 1;
  geo.mul[1] = geo.div_b[0];
  geo.mul[2] = geo.div_b[0] * geo.div_b[1];
  geo.n_cells = static_cast<long long>(geo.div_b[0]) * geo.div_b[1] * geo.div_b[2];
  // Big clouds: stable radix passes of K1's order-preserving scatter (launch_order_radix) -- the same order, point for point,
  // as the counting sort below (NDT_ORDER=chain: that one always).
  static const bool radix_on = [] { const char* v = getenv("NDT_ORDER"); return !v || std::strcmp(v, "chain") != 0; }();
  static const size_t radix_from = [] { const char* v = getenv("NDT_ORDER_RADIX_FROM"); return v ? static_cast<size_t>(std::max(0, atoi(v))) : static_cast<size_t>(65536); }();
  if (radix_on && n >= radix_from && ndt::order_radix_passes(geo.n_cells) <= 3) {
    int digit_bits = 0;
    const size_t words = ndt::order_radix_cntmat_words(geo.n_cells, ni, nullptr, nullptr, &digit_bits);
    const int passes = ndt::order_radix_passes(geo.n_cells);
    DevBuf<unsigned> cntmat, bucket_base, counts;
    DevBuf<float4> tmp;
    HIP_TRY(cntmat.reserve(words));
    HIP_TRY(bucket_base.reserve((static_cast<size_t>(1) << digit_bits) + 1));
    HIP_TRY(counts.reserve(4));
    if (passes > 1) HIP_TRY(tmp.reserve(n));
    HIP_TRY(ndt::launch_order_radix(d_pts, ni, geo, cntmat.p, bucket_base.p, tmp.p, d_out, counts.p, st));
    unsigned kept = 0;
    HIP_TRY(hipMemcpyAsync(&kept, counts.p + (passes - 1), sizeof(kept), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *n_out = kept;
    return NDT_OK;
  }
  DevBuf<unsigned> cell_count, block_sums, totals, leaf_start, rank;
  DevBuf<int> key, leaf_cell, leaf_count, leaf_rec, sorted_idx;
  HIP_TRY(cell_count.reserve(static_cast<size_t>(geo.n_cells)));
  HIP_TRY(key.reserve(n));
  HIP_TRY(rank.reserve(n));
  HIP_TRY(hipMemsetAsync(cell_count.p, 0, static_cast<size_t>(geo.n_cells) * sizeof(unsigned), st));
  HIP_TRY(ndt::launch_count(d_pts, ni, 0, geo, key.p, rank.p, cell_count.p, st));
  const int n_tiles = ndt::scan_tiles(geo.n_cells);
  HIP_TRY(block_sums.reserve(static_cast<size_t>(n_tiles) * 3));
  HIP_TRY(totals.reserve(4));
  HIP_TRY(ndt::launch_scan_reduce(cell_count.p, geo.n_cells, 1, block_sums.p, n_tiles, st));
  HIP_TRY(ndt::launch_scan_blocks(block_sums.p, n_tiles, totals.p, st));
  // the leaf count stays on the device (the kernels read it there): leaf arrays are sized for the
  // worst case and the host learns the totals once, at the end, instead of in the middle
  const size_t n_leaves = std::min<size_t>(n, static_cast<size_t>(geo.n_cells));
  HIP_TRY(leaf_cell.reserve(n_leaves));
  HIP_TRY(leaf_start.reserve(n_leaves));
  HIP_TRY(leaf_count.reserve(n_leaves));
  HIP_TRY(leaf_rec.reserve(n_leaves));
  HIP_TRY(sorted_idx.reserve(n));
  HIP_TRY(ndt::launch_scan_apply(cell_count.p, geo.n_cells, 1, block_sums.p, n_tiles, leaf_cell.p, leaf_start.p,
                                 leaf_count.p, leaf_rec.p, st));
  HIP_TRY(ndt::launch_scatter(key.p, rank.p, ni, cell_count.p, sorted_idx.p, st));
  HIP_TRY(ndt::launch_sort_gather(d_pts, leaf_start.p, leaf_count.p, static_cast<int>(n_leaves), sorted_idx.p, d_out, st, totals.p));
  unsigned tot[3];
  HIP_TRY(hipMemcpyAsync(tot, totals.p, sizeof(tot), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *n_out = tot[0];
  return NDT_OK;
}

// The scans of a batch in count / scan / scatter passes (composite key: the scan's first counter + its cell); the ordered
// points of scan k end up contiguous at scan_starts[k] (non-finite points are dropped, so the segments are compacted).
// A scan's order must not depend on the batch around it -- a member of a lock-step batch gets the same sums, bit for bit,
// whichever group or rank it is registered in (tools/fuzz_batch.py) -- so every scan is ordered on a lattice of its own:
// pitch `resolution` (doubled only while that ONE scan's box has more than kMaxCounters cells), box from its own points.
// Scans go through in passes of at most kMaxCounters counters.
ndt_status order_batch(ndt_context* h, DeviceCloud* c, const size_t* offsets, size_t n_scans) {
  hipStream_t st = h->stream;
  c->scan_counts.assign(n_scans, 0);
  c->scan_starts.assign(n_scans + 1, 0);
  c->n_sorted = 0;
  if (n_scans == 0 || c->n == 0) return NDT_OK;
  constexpr double kMaxCounters = 32.0e6;
  // ---- per-scan bounding boxes (one kernel, one small copy back)
  std::vector<int> off(n_scans + 1);
  size_t max_scan = 0;
  for (size_t k = 0; k <= n_scans; k++) off[k] = static_cast<int>(offsets[k] - offsets[0]);
  for (size_t k = 0; k < n_scans; k++) max_scan = std::max(max_scan, offsets[k + 1] - offsets[k]);
  DevBuf<int> d_off, d_box;
  HIP_TRY(d_off.reserve(n_scans + 1));
  HIP_TRY(d_box.reserve(6 * n_scans));
  std::vector<int> box(6 * n_scans);
  for (size_t k = 0; k < n_scans; k++)
    for (int j = 0; j < 6; j++) box[6 * k + j] = j < 3 ? std::numeric_limits<int>::max() : std::numeric_limits<int>::min();
  HIP_TRY(hipMemcpyAsync(d_off.p, off.data(), (n_scans + 1) * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_box.p, box.data(), box.size() * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_TRY(ndt::launch_scan_bboxes(c->pts.p, d_off.p, static_cast<int>(n_scans), static_cast<int>(max_scan), d_box.p, st));
  HIP_TRY(hipMemcpyAsync(box.data(), d_box.p, box.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // ---- every scan's own lattice
  std::vector<ndt::ScanLattice> lat(n_scans);
  for (size_t k = 0; k < n_scans; k++) {
    ndt::ScanLattice& L = lat[k];
    L = ndt::ScanLattice{};
    if (box[6 * k] > box[6 * k + 3]) continue;  // no finite point: n_cells 0
    float mn[3], mx[3];
    for (int j = 0; j < 3; j++) { mn[j] = ndt::scan_bbox_decode(box[6 * k + j]); mx[j] = ndt::scan_bbox_decode(box[6 * k + 3 + j]); }
    for (float pitch = h->resolution;; pitch *= 2.0f) {
      const float inv = 1.0f / pitch;
      double cells = 1;
      int div[3];
      for (int j = 0; j < 3; j++) {
        L.min_b[j] = static_cast<int>(std::floor(mn[j] * inv));
        div[j] = static_cast<int>(std::floor(mx[j] * inv)) - L.min_b[j] + 1;
        cells *= div[j];
      }
      if (cells <= kMaxCounters) {
        L.inv_leaf = inv;
        L.mul1 = div[0];
        L.mul2 = div[0] * div[1];
        L.n_cells = static_cast<int>(cells);
        break;
      }
    }
  }
  DevBuf<unsigned> cell_count, block_sums, totals, leaf_start, rank, d_starts;
  DevBuf<int> key, leaf_cell, leaf_count, leaf_rec, sorted_idx;
  DevBuf<ndt::ScanLattice> d_lat;
  DevBuf<long long> d_bases;
  size_t out_base = 0;
  for (size_t s0 = 0; s0 < n_scans;) {
    // this pass: scans [s0, s0 + ns) while their counters fit
    size_t ns = 0;
    long long total_cells = 0;
    while (s0 + ns < n_scans && (ns == 0 || static_cast<double>(total_cells + lat[s0 + ns].n_cells) <= kMaxCounters)) {
      lat[s0 + ns].base = total_cells;
      total_cells += lat[s0 + ns].n_cells;
      ns++;
    }
    const size_t first_pt = offsets[s0] - offsets[0], n_pts = offsets[s0 + ns] - offsets[s0];
    if (n_pts == 0 || total_cells == 0) {
      for (size_t k = 0; k < ns; k++) c->scan_starts[s0 + k] = out_base;
      s0 += ns;
      continue;
    }
    const int ni = static_cast<int>(n_pts);
    const float4* in = c->pts.p + first_pt;
    std::vector<int> poff(ns + 1);
    std::vector<long long> bases(ns + 1);
    size_t pass_max = 0;
    for (size_t k = 0; k <= ns; k++) poff[k] = static_cast<int>(offsets[s0 + k] - offsets[s0]);
    for (size_t k = 0; k < ns; k++) {
      pass_max = std::max(pass_max, offsets[s0 + k + 1] - offsets[s0 + k]);
      bases[k] = lat[s0 + k].base;
    }
    bases[ns] = total_cells;  // the sentinel cell: the pass's total
    HIP_TRY(d_off.reserve(ns + 1));
    HIP_TRY(d_lat.reserve(ns));
    HIP_TRY(d_bases.reserve(ns + 1));
    HIP_TRY(d_starts.reserve(ns + 1));
    HIP_TRY(hipMemcpyAsync(d_off.p, poff.data(), (ns + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_lat.p, lat.data() + s0, ns * sizeof(ndt::ScanLattice), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_bases.p, bases.data(), (ns + 1) * sizeof(long long), hipMemcpyHostToDevice, st));
    HIP_TRY(cell_count.reserve(static_cast<size_t>(total_cells) + 1));
    HIP_TRY(key.reserve(n_pts));
    HIP_TRY(rank.reserve(n_pts));
    HIP_TRY(hipMemsetAsync(cell_count.p, 0, (static_cast<size_t>(total_cells) + 1) * sizeof(unsigned), st));
    HIP_TRY(ndt::launch_count_batch(in, d_off.p, static_cast<int>(ns), static_cast<int>(pass_max), d_lat.p, key.p, rank.p, cell_count.p, st));
    // one extra (always empty) cell at the end so that its start offset is the pass's total
    const long long scan_cells = total_cells + 1;
    const int n_tiles = ndt::scan_tiles(scan_cells);
    HIP_TRY(block_sums.reserve(static_cast<size_t>(n_tiles) * 3));
    HIP_TRY(totals.reserve(4));
    HIP_TRY(ndt::launch_scan_reduce(cell_count.p, scan_cells, 1, block_sums.p, n_tiles, st));
    HIP_TRY(ndt::launch_scan_blocks(block_sums.p, n_tiles, totals.p, st));
    unsigned tot[3];
    HIP_TRY(hipMemcpyAsync(tot, totals.p, sizeof(tot), hipMemcpyDeviceToHost, st));  // read after the pass's synchronise
    const size_t n_leaves = std::min<size_t>(n_pts, static_cast<size_t>(scan_cells));  // upper bound; the count stays on the device
    HIP_TRY(leaf_cell.reserve(n_leaves));
    HIP_TRY(leaf_start.reserve(n_leaves));
    HIP_TRY(leaf_count.reserve(n_leaves));
    HIP_TRY(leaf_rec.reserve(n_leaves));
    HIP_TRY(sorted_idx.reserve(n_pts));
    HIP_TRY(ndt::launch_scan_apply(cell_count.p, scan_cells, 1, block_sums.p, n_tiles, leaf_cell.p, leaf_start.p, leaf_count.p, leaf_rec.p, st));
    // start offset of every scan's first cell (+ the sentinel cell)
    std::vector<unsigned> starts(ns + 1);
    HIP_TRY(ndt::launch_pick(cell_count.p, d_bases.p, d_starts.p, static_cast<int>(ns + 1), st));
    HIP_TRY(hipMemcpyAsync(starts.data(), d_starts.p, (ns + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(ndt::launch_scatter(key.p, rank.p, ni, cell_count.p, sorted_idx.p, st));
    HIP_TRY(ndt::launch_sort_gather(in, leaf_start.p, leaf_count.p, static_cast<int>(n_leaves), sorted_idx.p, c->sorted.p + out_base, st, totals.p));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t k = 0; k < ns; k++) {
      c->scan_starts[s0 + k] = out_base + starts[k];
      c->scan_counts[s0 + k] = starts[k + 1] - starts[k];
    }
    out_base += starts[ns];
    c->n_sorted += tot[0];
    s0 += ns;
  }
  c->scan_starts[n_scans] = out_base;
  return NDT_OK;
}

ndt_status order_cloud(ndt_context* h, DeviceCloud* c, const size_t* offsets, size_t n_scans) {
  // Spatial ordering pays for itself only on big scans (measured: 5-6 us per evaluation at 100k points
  // against a 1M-point target, nothing at <= 60k points where the voxel records stay in L2 anyway,
  // for 85-170 us of ordering work).  NDT_SORT_SOURCE=0 / 1 forces it off / on; a lock-step batch is
  // always ordered (its points are concatenated scan by scan).
  static const int mode = [] { const char* v = getenv("NDT_SORT_SOURCE"); return v ? (atoi(v) != 0 ? 1 : 0) : -1; }();
  constexpr size_t kOrderFrom = 65536;
  c->n_sorted = 0;
  const bool enabled = mode < 0 ? (offsets != nullptr || c->n >= kOrderFrom) : mode != 0;
  if (!enabled || c->n == 0) return NDT_OK;
  HIP_TRY(c->sorted.reserve(c->n));
  if (!offsets) {
    size_t got = 0;
    const BBox bb = bbox_of(*c, 0);
    ndt_status s = order_range(h, c->pts.p, c->n, h->resolution, c->sorted.p, &got, &bb);
    if (s) return s;
    c->n_sorted = got;
  } else {
    ndt_status s = order_batch(h, c, offsets, n_scans);
    if (s) return s;
  }
  return NDT_OK;
}

static ndt_status compact_records_now(ndt_context* h, DeviceGrid* g);

// VoxelGridCovariance::filter(true) on the GPU.
ndt_status build_grid(ndt_context* h) {
  if (!h->target) return fail(NDT_ERR_NO_INPUT, "no target");
  auto g = std::make_shared<DeviceGrid>();
  g->target = h->target;
  g->resolution = h->resolution;
  g->min_pts = h->min_pts;
  g->eig_ratio = h->eig_ratio;
  const int n = static_cast<int>(h->target->n);
  ndt::GridGeom& geo = g->geom;
  for (int k = 0; k < 3; k++) {
    geo.leaf[k] = h->resolution;
    geo.inv_leaf[k] = 1.0f / h->resolution;  // [PCL] VoxelGrid::setLeafSize
  }
  if (n == 0) {
    h->grid = g;
    return NDT_OK;
  }
  hipStream_t st = h->stream;
  // ---- bbox
  const BBox bb = bbox_of(*h->target, h->target_dense);  // computed during the upload: no kernel, no wait
  const float* min_p = bb.mn;
  const float* max_p = bb.mx;
  if (!(min_p[0] <= max_p[0])) {  // no finite point at all
    h->grid = g;
    return NDT_OK;
  }
  // ---- geometry, voxel_grid_covariance_omp_impl.hpp:75-103
  long long d[3];
  for (int k = 0; k < 3; k++) d[k] = static_cast<long long>((max_p[k] - min_p[k]) * geo.inv_leaf[k]) + 1;
  if (d[0] * d[1] * d[2] > static_cast<long long>(std::numeric_limits<int32_t>::max())) {
    h->grid = g;  // the reference warns and leaves an empty grid (:79-84)
    return fail(NDT_ERR_GRID_OVERFLOW, "leaf size is too small for the input dataset: integer indices would overflow");
  }
  for (int k = 0; k < 3; k++) {
    geo.min_b[k] = static_cast<int>(std::floor(min_p[k] * geo.inv_leaf[k]));
    geo.max_b[k] = static_cast<int>(std::floor(max_p[k] * geo.inv_leaf[k]));
    geo.div_b[k] = geo.max_b[k] - geo.min_b[k] + 1;
  }
  geo.mul[0] = 1;
  geo.mul[1] = geo.div_b[0];
  geo.mul[2] = geo.div_b[0] * geo.div_b[1];
  geo.n_cells = static_cast<long long>(geo.div_b[0]) * geo.div_b[1] * geo.div_b[2];
  if (geo.n_cells <= 0 || geo.n_cells > static_cast<long long>(std::numeric_limits<int32_t>::max()))
    return fail(NDT_ERR_GRID_OVERFLOW, "voxel grid too large");

  const size_t max_leaves = std::min<size_t>(static_cast<size_t>(n), static_cast<size_t>(geo.n_cells));
  const size_t max_cand = std::min<size_t>(max_leaves, static_cast<size_t>(n) / static_cast<size_t>(std::max(1, h->min_pts)) + 1);
  ndt::set_padded_lut(geo);
  // Dense or sparse voxel index?  Dense (a table over the whole bounding box, one dependent load per probe) as long as the
  // cell count stays within reach of the point count; sparse (sort-based build, hash look-up: ndt_sparse.hip) when the box
  // is mostly empty -- the regime the reference's std::map handles for free.  ndt_set_voxel_index overrides.
  const bool sparse = !h->index_only && (h->voxel_index == 2 || (h->voxel_index == 0 && (geo.n_cells > (1ll << 25) || geo.n_cells > 64ll * n + (1ll << 22))));
  HIP_TRY(g->counts.reserve(8));  // [points binned, occupied voxels, candidate voxels (>= min_pts), valid voxels, points in crowded cells]
  HIP_TRY(g->leaf_cell.reserve(max_leaves));
  HIP_TRY(g->leaf_start.reserve(max_leaves));
  HIP_TRY(g->leaf_count.reserve(max_leaves));
  HIP_TRY(g->leaf_rec.reserve(max_leaves));
  HIP_TRY(g->sorted_idx.reserve(n));
  if (sparse) {
    const size_t rec_slots = static_cast<size_t>(n) / static_cast<size_t>(std::max(1, h->min_pts)) + 1;  // slot = segment start / min_pts
    HIP_TRY(g->recs.reserve(rec_slots));
    HIP_TRY(g->centroids.reserve(rec_slots));
    int bits = 10;
    while ((static_cast<size_t>(1) << bits) < 2 * rec_slots) bits++;
    geo.hash_bits = bits;
    HIP_TRY(g->lut.reserve(static_cast<size_t>(2) << bits));  // int2 slots
    HIP_TRY(hipMemsetAsync(g->lut.p, 0xFF, (static_cast<size_t>(2) << bits) * sizeof(int), st));
    const size_t tb = ndt::sparse_index_temp_bytes(n);
    DevBuf<unsigned char> temp;
    DevBuf<unsigned> w;  // keys_a, keys_b, flags, ord
    DevBuf<int> vals;
    HIP_TRY(temp.reserve(tb));
    HIP_TRY(w.reserve(4 * static_cast<size_t>(n)));
    HIP_TRY(vals.reserve(n));
    HIP_TRY(ndt::launch_sparse_index(h->target->pts.p, n, h->target_dense, geo, h->min_pts, temp.p, tb, w.p, w.p + n, vals.p, w.p + 2 * static_cast<size_t>(n),
                                     w.p + 3 * static_cast<size_t>(n), g->leaf_cell.p, g->leaf_start.p, g->leaf_count.p, g->leaf_rec.p, g->sorted_idx.p,
                                     g->counts.p, st));
    ndt::FinalizeDump nodump{nullptr, nullptr, nullptr, nullptr, nullptr};
    DevBuf<float4> big_pts;
    HIP_TRY(big_pts.reserve(n));
    HIP_TRY(ndt::launch_finalize(h->target->pts.p, g->leaf_cell.p, g->leaf_start.p, g->leaf_count.p, g->leaf_rec.p, static_cast<int>(max_leaves),
                                 g->sorted_idx.p, h->min_pts, h->eig_ratio, g->recs.p, g->centroids.p, g->lut.p, geo, g->counts.p + 3, nodump, st,
                                 g->counts.p, big_pts.p));
    g->counts_known = false;
    g->empty = false;
    h->grid = g;
    return NDT_OK;
  }
  HIP_TRY(g->lut.reserve(static_cast<size_t>(geo.lut_cells)));
  static const int k1_mode = [] { const char* v = getenv("NDT_K1"); return !v ? 0 : std::strcmp(v, "old") == 0 ? 1 : std::strcmp(v, "new") == 0 ? 2 : 0; }();
  // The bucket form builds every dense grid (NDT_K1=old: the general chain, kept for index-only builds -- GICP's search
  // index -- and as the cross-check of tools/fuzz_grid.py).  With its cells dealt to the buckets in short runs (k1_bucket)
  // it is the faster form for every cloud shape measured, uniform to heavily clustered (tools/time_k1_forms.py).
  const bool buckets_on = k1_mode != 1;
  ndt::GridBuildPlan plan{};
  if (buckets_on && !h->index_only && ndt::grid_build_plan(geo.n_cells, n, plan)) {
    // ---- bucket form (ndt_kernels.hip "K1, bucket form"): no per-point global atomic, per-voxel work staged through LDS
    const size_t K = static_cast<size_t>(plan.n_buckets);
    const size_t rec_slots = static_cast<size_t>(n) / static_cast<size_t>(std::max(1, h->min_pts)) + 1;  // slot = segment start / min_pts
    HIP_TRY(g->recs.reserve(rec_slots));
    HIP_TRY(g->centroids.reserve(rec_slots));
    DevBuf<unsigned> cntmat, order;
    HIP_TRY(g->bucket_base.reserve(2 * K + 1));  // [K + 1] bucket bases, [K] valid voxels per bucket (k1_finalize -> k1_count)
    HIP_TRY(cntmat.reserve((static_cast<size_t>(plan.n_blocks) + 1) * K));
    HIP_TRY(order.reserve(5 * static_cast<size_t>(n)));
    HIP_TRY(g->bpts.reserve(n));
    ndt::GridBuildScratch S{};
    S.cntmat = cntmat.p;
    S.bucket_base = g->bucket_base.p;
    S.bpts = g->bpts.p;
    static const bool index_form_env = [] { const char* v = getenv("NDT_K1_INDEX"); return v && atoi(v) != 0; }();
    S.index_form = index_form_env;
    S.order = order.p;
    static const bool want_stamps = [] { const char* v = getenv("NDT_K1_STAMPS"); return v && atoi(v) != 0; }();
    DevBuf<unsigned long long> stamps;
    if (want_stamps) {
      HIP_TRY(stamps.reserve(8 * (K + std::max(K, static_cast<size_t>(plan.n_blocks)))));
      HIP_TRY(hipMemsetAsync(stamps.p, 0, 8 * (K + std::max(K, static_cast<size_t>(plan.n_blocks))) * sizeof(unsigned long long), st));
      S.stamps = stamps.p;
    }
    // Records dense and in ascending cell order (maybe_compact_records: two small launches, ~13 us) pay for themselves as
    // soon as a few scans are registered against the grid: +8 % on lock-step batches, +1-5 % on a single 100k-point scan.
    // A mapping node registers ONE scan against every target it builds (ndt_omp_mapping_node.cpp:151-169), so the build
    // itself leaves k1_finalize's numbering and the compaction runs when the grid is seen to be reused: before the second
    // registration against it, or before the first lock-step batch (NDT_K1_COMPACT=eager: at once, as round 2 did; off: never).
    // The mapping nodes' 16 k-point clouds (records that fit L2 many times over) never compact.
    static const int compact_mode = [] { const char* v = getenv("NDT_K1_COMPACT"); return !v ? 1 : std::strcmp(v, "eager") == 0 ? 2 : std::strcmp(v, "off") == 0 ? 0 : 1; }();
    static const bool small_on = [] { const char* v = getenv("NDT_K1_SMALL"); return !v || atoi(v) != 0; }();
    const bool small_form = small_on && ndt::grid_build_small_applies(n, plan);
    if (small_form) S.index_form = false;
    g->index_form = S.index_form;
    if (small_form)
      HIP_TRY(ndt::launch_grid_build_small(h->target->pts.p, n, h->target_dense, geo, plan, h->min_pts, h->eig_ratio, S, g->sorted_idx.p,
                                           g->recs.p, g->centroids.p, g->lut.p, g->counts.p, st));
    else
    HIP_TRY(ndt::launch_grid_build_buckets(h->target->pts.p, n, h->target_dense, geo, plan, h->min_pts, h->eig_ratio, S, g->sorted_idx.p,
                                           g->recs.p, g->centroids.p, g->lut.p, g->counts.p, st));
    g->compact_pending = n > 65536 && compact_mode != 0;
    if (want_stamps) {  // k1_finalize's phase clocks: per phase the median and the maximum over the buckets (shader cycles)
      std::vector<unsigned long long> hst(8 * K);
      HIP_TRY(hipMemcpyAsync(hst.data(), stamps.p, 8 * K * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      static const char* names[8] = {"load+rank", "scans+select", "place", "teams", "sums+finish", "passes", "points", "total"};
      std::fprintf(stderr, "[k1_finalize clocks, %zu buckets] ", K);
      for (int q = 0; q < 8; q++) {
        std::vector<unsigned long long> d;
        for (size_t b = 0; b < K; b++)
          if (hst[8 * b + 7]) d.push_back(hst[8 * b + q]);
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        std::fprintf(stderr, "%s %llu/%llu  ", names[q], d[d.size() / 2], d.back());
      }
      std::fprintf(stderr, "\n");
      if (small_form) {  // k1_small: cycles from the block's start to the end of the scan / the publication / the end
        std::vector<unsigned long long> hk(4 * K);
        HIP_TRY(hipMemcpy(hk.data(), stamps.p + 8 * K, 4 * K * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[k1_small clocks, %zu blocks, cycles since the block's start: scan / published / end] ", K);
        for (int q = 1; q < 4; q++) {
          std::vector<unsigned long long> d;
          for (size_t b = 0; b < K; b++) d.push_back(hk[4 * b + q] - hk[4 * b]);
          std::sort(d.begin(), d.end());
          std::fprintf(stderr, "%llu/%llu  ", d[d.size() / 2], d.back());
        }
        unsigned long long t_lo = ~0ull, t_hi = 0;
        for (size_t b = 0; b < K; b++) { t_lo = std::min(t_lo, hk[4 * b]); t_hi = std::max(t_hi, hk[4 * b + 3]); }
        std::fprintf(stderr, " first start -> last end %llu\n", t_hi - t_lo);
      }
      const size_t B = small_form ? 0 : static_cast<size_t>(plan.n_blocks);
      std::vector<unsigned long long> hs(8 * B + 1);
      HIP_TRY(hipMemcpy(hs.data(), stamps.p + 8 * K, 8 * B * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      if (B) std::fprintf(stderr, "[k1_scatter clocks, %zu blocks, cycles since the block's start: tables / ranks / column scan / stores] ", B);
      for (int q = 1; q < 5; q++) {
        std::vector<unsigned long long> d;
        for (size_t bq = 0; bq < B; bq++)
          if (hs[8 * bq] && hs[8 * bq + q]) d.push_back(hs[8 * bq + q] - hs[8 * bq]);
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        std::fprintf(stderr, "%llu/%llu  ", d[d.size() / 2], d.back());
      }
      std::fprintf(stderr, "\n");
    }
    g->plan = plan;
    g->leaves_pending = true;  // leaf arrays and the occupied / candidate counts: on demand (grid_counts)
    if (compact_mode == 2 && g->compact_pending) {
      ndt_status cs = compact_records_now(h, g.get());
      if (cs) return cs;
    }
  } else {
  // every cell of the padded table starts out empty (kLutEmpty = -1 = all bits set); the finalize pass fills in the
  // voxels that reached min_points_per_voxel
  HIP_TRY(hipMemsetAsync(g->lut.p, 0xFF, static_cast<size_t>(geo.lut_cells) * sizeof(int), st));
  HIP_TRY(g->recs.reserve(max_cand));
  HIP_TRY(g->centroids.reserve(max_cand));
  // ---- count
  DevBuf<unsigned> cell_count, block_sums, rank;
  DevBuf<int> key;
  HIP_TRY(cell_count.reserve(static_cast<size_t>(geo.n_cells)));
  HIP_TRY(key.reserve(n));
  HIP_TRY(rank.reserve(n));
  HIP_TRY(hipMemsetAsync(cell_count.p, 0, static_cast<size_t>(geo.n_cells) * sizeof(unsigned), st));
  HIP_TRY(ndt::launch_count(h->target->pts.p, n, h->target_dense, geo, key.p, rank.p, cell_count.p, st));
  // ---- scan
  const int n_tiles = ndt::scan_tiles(geo.n_cells);
  HIP_TRY(block_sums.reserve(static_cast<size_t>(n_tiles) * 3));
  HIP_TRY(ndt::launch_scan_reduce(cell_count.p, geo.n_cells, h->min_pts, block_sums.p, n_tiles, st));
  HIP_TRY(ndt::launch_scan_blocks(block_sums.p, n_tiles, g->counts.p, st));
  // The counts stay on the device: the later kernels read the voxel count there, the arrays are sized
  // for the worst case, and the host fetches the four numbers only if somebody asks (grid_counts()).
  // Two host round trips (~30 us each) less per target; nothing below waits for the GPU.
  HIP_TRY(ndt::launch_scan_apply(cell_count.p, geo.n_cells, h->min_pts, block_sums.p, n_tiles, g->leaf_cell.p,
                                 g->leaf_start.p, g->leaf_count.p, g->leaf_rec.p, st));
  // ---- scatter + finalize
  HIP_TRY(ndt::launch_scatter(key.p, rank.p, n, cell_count.p, g->sorted_idx.p, st));
  HIP_TRY(hipMemsetAsync(g->counts.p + 3, 0, 2 * sizeof(unsigned), st));
  ndt::FinalizeDump nodump{nullptr, nullptr, nullptr, nullptr, nullptr};
  DevBuf<float4> big_pts;  // scratch of the crowded-leaf path (k_presort_large)
  if (!h->index_only) {
    HIP_TRY(big_pts.reserve(n));
    HIP_TRY(ndt::launch_finalize(h->target->pts.p, g->leaf_cell.p, g->leaf_start.p, g->leaf_count.p, g->leaf_rec.p,
                                 static_cast<int>(max_leaves), g->sorted_idx.p, h->min_pts, h->eig_ratio, g->recs.p, g->centroids.p,
                                 g->lut.p, geo, g->counts.p + 3, nodump, st, g->counts.p, big_pts.p));
  }
  }
  // the temporaries (cell_count, key, rank, block_sums) go back to the caching pool at scope exit; the
  // pool hands memory out again only to work queued on the same stream, i.e. after these kernels
  g->counts_known = false;
  g->empty = false;
  h->grid = g;
  return NDT_OK;
}

// Records of a bucket-form build -> dense, ascending cell order (launch_compact_records), in place: only while this handle is
// the grid's single holder (a clone may be evaluating on it) and before anybody has asked for the leaf arrays (they quote
// record numbers).  eager: now; otherwise from the second registration against the grid on.
static ndt_status compact_records_now(ndt_context* h, DeviceGrid* g) {
  g->compact_pending = false;
  if (!g->leaves_pending || g->empty) return NDT_OK;
  const size_t rec_slots = g->recs.cap;
  DevBuf<ndt::VoxelRec> recs_new;
  DevBuf<ndt::VoxelSide> cent_new;
  DevBuf<unsigned> tile_sums;
  HIP_TRY(recs_new.reserve(rec_slots));
  HIP_TRY(cent_new.reserve(rec_slots));
  HIP_TRY(tile_sums.reserve(ndt::record_compaction_tiles(g->geom.lut_cells) + 1));
  HIP_TRY(ndt::launch_compact_records(g->lut.p, g->geom.lut_cells, g->recs.p, g->centroids.p, recs_new.p, cent_new.p, tile_sums.p, h->stream));
  g->recs.swap(recs_new);  // (the old arrays go back to the pool at scope exit: reused only behind these launches, stream order)
  g->centroids.swap(cent_new);
  return NDT_OK;
}
ndt_status maybe_compact_records(ndt_context* h, bool eager) {
  DeviceGrid* g = h->grid.get();
  if (!g || !g->compact_pending) return NDT_OK;
  if (!eager && g->n_registrations++ == 0) return NDT_OK;
  if (h->grid.use_count() != 1) {  // shared with a clone: leave it alone for good
    g->compact_pending = false;
    return NDT_OK;
  }
  return compact_records_now(h, g);
}

// occupied / candidate / valid voxel counts of a built grid (fetched from the device on first use)
ndt_status grid_counts(ndt_context* h, DeviceGrid* g) {
  if (g->counts_known || g->empty) return NDT_OK;
  std::lock_guard<std::mutex> lock(g->fit_mu);
  if (g->counts_known) return NDT_OK;
  if (g->leaves_pending) {  // bucket-form build: number the leaves now that somebody wants them
    const size_t K = static_cast<size_t>(g->plan.n_buckets);
    DevBuf<unsigned> scratch;
    HIP_TRY(scratch.reserve(4 * K + 4));
    HIP_TRY(ndt::launch_grid_leaves(g->geom, g->plan, g->min_pts, g->bpts.p, g->bucket_base.p, scratch.p, g->leaf_cell.p, g->leaf_start.p,
                                    g->leaf_count.p, g->leaf_rec.p, g->counts.p, g->lut.p, h->stream, g->index_form ? g->target->pts.p : nullptr));
    HIP_TRY(hipStreamSynchronize(h->stream));
    g->leaves_pending = false;
    g->bpts.release();
    g->bucket_base.release();
  }
  unsigned c[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(c, g->counts.p, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  g->n_sorted = c[0];
  g->n_leaves = c[1];
  g->n_cand = c[2];
  g->n_valid = c[3];
  g->counts_known = true;
  return NDT_OK;
}

// cell -> occupied-cell ordinal table of a built grid (the nearest-neighbour searches walk it), built on first use
ndt_status ensure_cell2leaf(ndt_context* h, DeviceGrid* g) {
  std::lock_guard<std::mutex> lock(g->fit_mu);
  if (!g->have_cell2leaf) {
    HIP_TRY(g->cell_range.reserve(static_cast<size_t>(g->geom.n_cells)));
    HIP_TRY(hipMemsetAsync(g->cell_range.p, 0, static_cast<size_t>(g->geom.n_cells) * sizeof(uint2), h->stream));
    const size_t n_rows = static_cast<size_t>(g->geom.div_b[1]) * static_cast<size_t>(g->geom.div_b[2]);
    HIP_TRY(g->row_any.reserve(n_rows));
    HIP_TRY(hipMemsetAsync(g->row_any.p, 0, n_rows * sizeof(int), h->stream));
    HIP_TRY(ndt::launch_cell_ranges(g->leaf_cell.p, g->leaf_start.p, g->leaf_count.p, static_cast<int>(g->n_leaves), g->cell_range.p,
                                    g->geom.div_b[0], g->row_any.p, h->stream));
    HIP_TRY(g->cell_pts.reserve(g->target->n));
    HIP_TRY(ndt::launch_gather_points(g->target->pts.p, g->sorted_idx.p, g->counts.p, static_cast<int>(g->target->n), g->cell_pts.p,
                                      h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    g->have_cell2leaf = true;
  }
  return NDT_OK;
}
// slack of the shell bound: the build-time and search-time cell indices of a coordinate can differ
// at cell borders by rounding (SURVEY 8a trap 2) -- a few ulps of the largest coordinate
float index_slack(const DeviceGrid* g) {
  float max_abs = 0.f;
  for (int k = 0; k < 3; k++)
    max_abs = std::max(max_abs, std::max(std::fabs(g->geom.min_b[k] * g->geom.leaf[k]), std::fabs((g->geom.max_b[k] + 1) * g->geom.leaf[k])));
  return 1e-3f * g->resolution + 4e-6f * max_abs;
}
// the search structure over a built grid's target (after ensure_cell2leaf + grid_counts)
void fill_point_index(const DeviceGrid* g, ndt::PointIndex& ix) {
  ix.pts = g->target->pts.p;
  ix.n = static_cast<int>(g->target->n);
  ix.geom = g->geom;
  ix.cell_range = g->cell_range.p;
  ix.row_any = g->row_any.p;
  ix.sorted_idx = g->sorted_idx.p;
  ix.sorted_pts = g->cell_pts.p;
  ix.n_sorted = static_cast<int>(g->n_sorted);
  ix.slack = index_slack(g);
}
// [PCL] Registration::getFitnessScore of the dense device cloud d_src moved by T against h's target
ndt_status fitness_impl(ndt_context* h, const float4* d_src, int n, const float* T_colmajor, double max_range, double* fitness) {
  *fitness = std::numeric_limits<double>::max();  // nr == 0 in the reference
  DeviceGrid* g = h->grid.get();
  ndt_status s = grid_counts(h, g);
  if (s) return s;
  if (n == 0 || g->empty || g->n_sorted == 0) return NDT_OK;
  s = ensure_cell2leaf(h, g);
  if (s) return s;
  s = ensure_host_rows(h, 1);
  if (s) return s;
  float T12[12];
  colmajor_to_T12(T_colmajor, T12);
  const int nblk = std::max(1, std::min(2048, (n + 31) / 32));  // 32 query teams per block
  HIP_TRY(h->partials.reserve(static_cast<size_t>(nblk) * ndt::kEvalStride));
  ndt::PointIndex ix;
  fill_point_index(g, ix);
  HIP_TRY(ndt::launch_fitness(d_src, n, T12, ix, max_range, nblk, h->partials.p, h->stream));
  HIP_TRY(ndt::launch_reduce(h->partials.p, nblk, 1, nullptr, h->host_result, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->host_result[1] > 0) *fitness = h->host_result[0] / h->host_result[1];
  return NDT_OK;
}

// ---- N1: voxel-grid centroid down-sample -----------------------------------
// [PCL] VoxelGrid::applyFilter on a dense float4 device cloud: d_out (capacity n) receives one centroid per occupied voxel
// in ascending voxel-index order.  Two halves: voxel_filter_enqueue queues the whole chain on a stream and returns -- the
// count and the per-block rows of the result's bounding boxes travel to page-locked memory behind the last kernel --
// voxel_filter_finish reads them once that stream has been waited for.  voxel_filter_device is the two with a
// synchronisation in between (N1); the map update (N2) leaves the wait to whoever next needs the map.
static constexpr int kOutBoxBlocks = 64;
struct PoolStreamGuard {  // temporaries allocated (and given back) inside the scope belong to `s`'s pool
  hipStream_t keep;
  explicit PoolStreamGuard(hipStream_t s) : keep(tls_pool_stream) { tls_pool_stream = s; }
  ~PoolStreamGuard() { tls_pool_stream = keep; }
};

ndt_status voxel_filter_enqueue(ndt_handle h, hipStream_t st, const float4* d_in, size_t n, int is_dense, float leaf, float4* d_out,
                                const BBox& bb, FilterPending& P) {
  P.n_max = n;
  P.fixed_n = 0;
  P.from_device = false;
  P.overflow = false;
  if (n == 0) return NDT_OK;
  const PoolStreamGuard guard(st);
  const int ni = static_cast<int>(n);
  const float* min_p = bb.mn;
  const float* max_p = bb.mx;
  if (!(min_p[0] <= max_p[0])) {  // no finite point: empty output
    for (int i = 0; i < kOutBoxBlocks * 12; i++) P.rows[i] = (i % 6) < 3 ? FLT_MAX : -FLT_MAX;
    return NDT_OK;
  }
  const int nb_rows = static_cast<int>(std::min<size_t>(kOutBoxBlocks, (n + 255) / 256));
  ndt::GridGeom geo{};
  long long d[3];
  for (int k = 0; k < 3; k++) {
    geo.leaf[k] = leaf;
    geo.inv_leaf[k] = 1.0f / leaf;
    d[k] = static_cast<long long>((max_p[k] - min_p[k]) * geo.inv_leaf[k]) + 1;
  }
  if (d[0] * d[1] * d[2] > static_cast<long long>(std::numeric_limits<int32_t>::max())) {
    HIP_TRY(hipMemcpyAsync(d_out, d_in, n * sizeof(float4), hipMemcpyDeviceToDevice, st));  // output = *input_
    HIP_TRY(ndt::launch_repack_bbox(d_out, n, sizeof(float4), nullptr, P.rows, nb_rows, st, 0, nullptr));
    P.fixed_n = n;
    P.overflow = true;
    return NDT_OK;
  }
  for (int k = 0; k < 3; k++) {
    geo.min_b[k] = static_cast<int>(std::floor(min_p[k] * geo.inv_leaf[k]));
    geo.max_b[k] = static_cast<int>(std::floor(max_p[k] * geo.inv_leaf[k]));
    geo.div_b[k] = geo.max_b[k] - geo.min_b[k] + 1;
  }
  geo.mul[0] = 1;
  geo.mul[1] = geo.div_b[0];
  geo.mul[2] = geo.div_b[0] * geo.div_b[1];
  geo.n_cells = static_cast<long long>(geo.div_b[0]) * geo.div_b[1] * geo.div_b[2];
  DevBuf<unsigned> cell_count, block_sums, totals, leaf_start, rank;
  DevBuf<int> key, leaf_cell, leaf_count, leaf_rec, sorted_idx;
  P.from_device = true;
  if (h->voxel_index == 2 || (h->voxel_index == 0 && geo.n_cells > 16ll * static_cast<long long>(n) + (1ll << 22))) {
    // a fine leaf over a wide box (apps/align.cpp: 0.1 m over a whole scan): per-point work only (ndt_sparse.hip)
    const size_t max_l = std::min<size_t>(n, static_cast<size_t>(geo.n_cells));
    const size_t tb = ndt::sparse_index_temp_bytes(ni);
    DevBuf<unsigned char> temp;
    DevBuf<unsigned> w;
    DevBuf<int> vals;
    DevBuf<float4> big2;
    HIP_TRY(temp.reserve(tb));
    HIP_TRY(w.reserve(4 * n));
    HIP_TRY(vals.reserve(n));
    HIP_TRY(totals.reserve(8));
    HIP_TRY(leaf_cell.reserve(max_l));
    HIP_TRY(leaf_start.reserve(max_l));
    HIP_TRY(leaf_count.reserve(max_l));
    HIP_TRY(leaf_rec.reserve(max_l));
    HIP_TRY(sorted_idx.reserve(n));
    HIP_TRY(big2.reserve(n));
    HIP_TRY(ndt::launch_sparse_index(d_in, ni, is_dense, geo, 1, temp.p, tb, w.p, w.p + n, vals.p, w.p + 2 * n, w.p + 3 * n, leaf_cell.p, leaf_start.p,
                                     leaf_count.p, leaf_rec.p, sorted_idx.p, totals.p, st));
    HIP_TRY(ndt::launch_voxel_centroids(d_in, leaf_start.p, leaf_count.p, static_cast<int>(max_l), sorted_idx.p, d_out, st, totals.p, big2.p));
    HIP_TRY(hipMemcpyAsync(P.tot, totals.p, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(ndt::launch_repack_bbox(d_out, n, sizeof(float4), nullptr, P.rows, nb_rows, st, 0, totals.p + 1));
    return NDT_OK;  // (the temporaries go back to the stream's pool: reused only behind these launches)
  }
  // Dense grids within the bucket plan's range: the order-preserving bucket front end of K1 + vf_finalize / vf_bitmap_prefix /
  // vf_place (ndt_grid_kernels.hip): the cell space is never walked, no point is gathered through an index.
  // NDT_VF=chain: the general chain below for every grid (the cross-check).
  static const bool vf_buckets = [] { const char* v = getenv("NDT_VF"); return !(v && std::strcmp(v, "chain") == 0); }();
  ndt::GridBuildPlan plan{};
  // (from 128 k points and for boxes with at most four cells per point: below / beyond, a bucket's share of the cell space --
  // thousands of cells for a hundred points -- makes vf_finalize cost what the chain's scans cost: 60 k points 81 against 76 us,
  // a 250 k-point map 29 us for that kernel alone; 300 k-point scan 81 against 111, 1 M 106 against 228.  NDT_VF_FROM=0: always.)
  static const long long vf_from = [] { const char* v = getenv("NDT_VF_FROM"); return v ? static_cast<long long>(std::max(0, atoi(v))) : 131072ll; }();
  const bool vf_dense = vf_from == 0 || (static_cast<long long>(n) >= vf_from && geo.n_cells <= 4ll * static_cast<long long>(n));
  if (vf_buckets && vf_dense && ndt::filter_buckets_plan(geo.n_cells, ni, plan)) {
    const size_t K = static_cast<size_t>(plan.n_buckets);
    const size_t bw = ndt::filter_buckets_bitmap_words(geo.n_cells);
    DevBuf<unsigned> cntmat, order, bucket_base, bitmap, wprefix;
    DevBuf<float4> bpts, st_cent;
    DevBuf<int> st_cell;
    HIP_TRY(cntmat.reserve((static_cast<size_t>(plan.n_blocks) + 1) * K));
    HIP_TRY(order.reserve(5 * n));
    HIP_TRY(bucket_base.reserve(2 * K + 1));
    HIP_TRY(bpts.reserve(n));
    HIP_TRY(st_cell.reserve(n));
    HIP_TRY(st_cent.reserve(n));
    HIP_TRY(bitmap.reserve(bw));
    HIP_TRY(wprefix.reserve(bw));
    HIP_TRY(totals.reserve(4));
    ndt::GridBuildScratch S{};
    S.cntmat = cntmat.p;
    S.bucket_base = bucket_base.p;
    S.bpts = bpts.p;
    S.order = order.p;
    HIP_TRY(ndt::launch_filter_buckets(d_in, ni, is_dense, geo, plan, S, st_cell.p, st_cent.p, bitmap.p, wprefix.p, totals.p, d_out, st));
    HIP_TRY(hipMemcpyAsync(P.tot, totals.p, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(ndt::launch_repack_bbox(d_out, n, sizeof(float4), nullptr, P.rows, nb_rows, st, 0, totals.p + 1));
    return NDT_OK;  // (the temporaries go back to the stream's pool: reused only behind these launches)
  }
  HIP_TRY(cell_count.reserve(static_cast<size_t>(geo.n_cells)));
  HIP_TRY(key.reserve(n));
  HIP_TRY(rank.reserve(n));
  HIP_TRY(hipMemsetAsync(cell_count.p, 0, static_cast<size_t>(geo.n_cells) * sizeof(unsigned), st));
  HIP_TRY(ndt::launch_count(d_in, ni, is_dense, geo, key.p, rank.p, cell_count.p, st));
  const int n_tiles = ndt::scan_tiles(geo.n_cells);
  HIP_TRY(block_sums.reserve(static_cast<size_t>(n_tiles) * 3));
  HIP_TRY(totals.reserve(4));
  HIP_TRY(ndt::launch_scan_reduce(cell_count.p, geo.n_cells, 1, block_sums.p, n_tiles, st));
  HIP_TRY(ndt::launch_scan_blocks(block_sums.p, n_tiles, totals.p, st));
  const size_t n_leaves = std::min<size_t>(n, static_cast<size_t>(geo.n_cells));  // upper bound; the count stays on the device
  HIP_TRY(leaf_cell.reserve(n_leaves));
  HIP_TRY(leaf_start.reserve(n_leaves));
  HIP_TRY(leaf_count.reserve(n_leaves));
  HIP_TRY(leaf_rec.reserve(n_leaves));
  HIP_TRY(sorted_idx.reserve(n));
  HIP_TRY(ndt::launch_scan_apply(cell_count.p, geo.n_cells, 1, block_sums.p, n_tiles, leaf_cell.p, leaf_start.p,
                                 leaf_count.p, leaf_rec.p, st));
  HIP_TRY(ndt::launch_scatter(key.p, rank.p, ni, cell_count.p, sorted_idx.p, st));
  DevBuf<float4> big_pts;  // scratch of the crowded-voxel path (k_presort_large)
  HIP_TRY(big_pts.reserve(n));
  HIP_TRY(ndt::launch_voxel_centroids(d_in, leaf_start.p, leaf_count.p, static_cast<int>(n_leaves), sorted_idx.p, d_out, st, totals.p, big_pts.p));
  HIP_TRY(hipMemcpyAsync(P.tot, totals.p, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
  HIP_TRY(ndt::launch_repack_bbox(d_out, n, sizeof(float4), nullptr, P.rows, nb_rows, st, 0, totals.p + 1));
  return NDT_OK;
}

// after the stream of voxel_filter_enqueue has been waited for: the count, and the boxes of the result
void voxel_filter_finish(const FilterPending& P, size_t* n_out, DeviceCloud* boxes) {
  *n_out = P.from_device ? P.tot[1] : P.fixed_n;
  if (!boxes) return;
  for (int v = 0; v < 2; v++)
    for (int k = 0; k < 3; k++) {
      boxes->bb_min[v][k] = FLT_MAX;
      boxes->bb_max[v][k] = -FLT_MAX;
    }
  if (*n_out == 0) return;
  const int nb = static_cast<int>(std::min<size_t>(kOutBoxBlocks, (P.n_max + 255) / 256));
  for (int b = 0; b < nb; b++)
    for (int v = 0; v < 2; v++)
      for (int k = 0; k < 3; k++) {
        boxes->bb_min[v][k] = std::min(boxes->bb_min[v][k], P.rows[b * 12 + v * 6 + k]);
        boxes->bb_max[v][k] = std::max(boxes->bb_max[v][k], P.rows[b * 12 + v * 6 + 3 + k]);
      }
}

ndt_status filter_slots(ndt_handle h, int which, FilterPending& P) {
  if (!h->filter_slots) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->filter_slots), 3 * (kOutBoxBlocks * 12 + 4) * sizeof(float), hipHostMallocDefault));
  }
  float* base = h->filter_slots + which * (kOutBoxBlocks * 12 + 4);
  P.rows = base;
  P.tot = reinterpret_cast<unsigned*>(base + kOutBoxBlocks * 12);
  return NDT_OK;
}

// the synchronous form (N1): *overflow = the leaf is too small for the bounding box and, as PCL does, the input was copied
// through.  Synchronises h->stream.
ndt_status voxel_filter_device(ndt_handle h, const float4* d_in, size_t n, int is_dense, float leaf, float4* d_out,
                                      size_t* n_out, bool* overflow, const BBox* known_bbox, DeviceCloud* out_boxes) {
  *n_out = 0;
  *overflow = false;
  if (n == 0) return NDT_OK;
  BBox bb;
  if (known_bbox) bb = *known_bbox;
  else { ndt_status sb = bbox_compute(h, d_in, static_cast<int>(n), is_dense, bb); if (sb) return sb; }
  FilterPending P;
  ndt_status s = filter_slots(h, 0, P);
  if (!s) s = voxel_filter_enqueue(h, h->stream, d_in, n, is_dense, leaf, d_out, bb, P);
  if (s) return s;
  HIP_TRY(hipStreamSynchronize(h->stream));
  voxel_filter_finish(P, n_out, out_boxes);
  *overflow = P.overflow;
  return NDT_OK;
}

}  // namespace ndtc

extern "C" {

static ndt_status set_target_impl(ndt_handle h, const void* pts, size_t n, size_t stride, int is_dense, bool on_device, bool by_ref = false) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  std::shared_ptr<DeviceCloud> c;
  ndt_status s = upload_cloud(h, pts, n, stride, on_device, c, by_ref);
  if (s) return s;
  h->target = c;
  h->target_dense = is_dense ? 1 : 0;
  return build_grid(h);  // init(), ndt_omp.h:276-283
}
ndt_status ndt_set_input_target(ndt_handle h, const void* pts, size_t n, size_t stride, int is_dense) {
  return set_target_impl(h, pts, n, stride, is_dense, false);
}
ndt_status ndt_set_input_target_device(ndt_handle h, const void* pts, size_t n, size_t stride, int is_dense) {
  return set_target_impl(h, pts, n, stride, is_dense, true);
}
ndt_status ndt_set_input_target_device_ref(ndt_handle h, const void* d_pts, size_t n, int is_dense) {
  return set_target_impl(h, d_pts, n, sizeof(float4), is_dense, true, true);
}
static ndt_status set_source_impl(ndt_handle h, const void* pts, size_t n, size_t stride, bool on_device, bool by_ref = false) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  std::shared_ptr<DeviceCloud> c;
  ndt_status s = upload_cloud(h, pts, n, stride, on_device, c, by_ref);
  if (s) return s;
  s = order_cloud(h, c.get(), nullptr, 0);
  if (s) return s;
  h->source = c;
  return NDT_OK;
}
ndt_status ndt_set_input_source(ndt_handle h, const void* pts, size_t n, size_t stride) {
  return set_source_impl(h, pts, n, stride, false);
}
ndt_status ndt_set_input_source_device(ndt_handle h, const void* pts, size_t n, size_t stride) {
  return set_source_impl(h, pts, n, stride, true);
}
ndt_status ndt_set_input_source_device_ref(ndt_handle h, const void* d_pts, size_t n) {
  return set_source_impl(h, d_pts, n, sizeof(float4), true, true);
}

ndt_status ndt_set_voxel_index(ndt_handle h, int mode) {
  if (!h || mode < 0 || mode > 2) return fail(NDT_ERR_INVALID, "bad arguments");
  h->voxel_index = mode;
  return NDT_OK;
}

ndt_status ndt_share_input_source(ndt_handle dst, ndt_handle src) {
  if (!dst || !src) return fail(NDT_ERR_INVALID, "null handle");
  if (!src->source) return fail(NDT_ERR_NO_INPUT, "the donor handle has no input source");
  if (dst == src) return NDT_OK;
  if (dst->device != src->device) return fail(NDT_ERR_INVALID, "handles on different devices");
  // the cloud was uploaded and ordered on the donor's stream; what `dst` still runs on its old source must be over
  // before that cloud can go back to the pool
  HIP_TRY(hipSetDevice(src->device));
  if (src->device_ready) HIP_TRY(hipStreamSynchronize(src->stream));
  ndt_status s = ensure_device(dst);
  if (s) return s;
  HIP_TRY(hipStreamSynchronize(dst->stream));
  dst->source = src->source;
  return NDT_OK;
}

// the target cloud and its built grid, shared like the source above: a prep handle (side partition) builds the next target
// while the registration handle works; no copy, no rebuild
ndt_status ndt_share_input_target(ndt_handle dst, ndt_handle src) {
  if (!dst || !src) return fail(NDT_ERR_INVALID, "null handle");
  if (!src->target || !src->grid) return fail(NDT_ERR_NO_INPUT, "the donor handle has no input target");
  if (dst == src) return NDT_OK;
  if (dst->device != src->device) return fail(NDT_ERR_INVALID, "handles on different devices");
  HIP_TRY(hipSetDevice(src->device));
  if (src->device_ready) HIP_TRY(hipStreamSynchronize(src->stream));  // the grid may still be under construction there
  ndt_status s = ensure_device(dst);
  if (s) return s;
  HIP_TRY(hipStreamSynchronize(dst->stream));
  dst->target = src->target;
  dst->target_dense = src->target_dense;
  dst->grid = src->grid;
  dst->resolution = src->grid->resolution;  // the grid's parameters come with it (the Gauss constants follow the resolution)
  dst->min_pts = src->grid->min_pts;
  dst->eig_ratio = src->grid->eig_ratio;
  return NDT_OK;
}

ndt_status ndt_calculate_score(ndt_handle h, const void* cloud, size_t n, size_t stride, double* score) {
  if (!h || !score) return fail(NDT_ERR_INVALID, "bad arguments");
  if (!h->grid || !h->target) return fail(NDT_ERR_NO_INPUT, "no input target");
  std::shared_ptr<DeviceCloud> c;
  ndt_status s = upload_cloud(h, cloud, n, stride, false, c);
  if (s) return s;
  if (n == 0 || h->grid->empty) {
    *score = n ? 0.0 : std::numeric_limits<double>::quiet_NaN();  // 0/0 in the reference
    return NDT_OK;
  }
  s = ensure_host_rows(h, 1);
  if (s) return s;
  const ndt::Gauss gs = ndt::gauss_constants(h->resolution, h->outlier_ratio);
  const int nblk = ndt::derivative_blocks(static_cast<int>(n), NDT_DIRECT1);
  HIP_TRY(h->partials.reserve(static_cast<size_t>(nblk) * ndt::kEvalStride));
  HIP_TRY(hipMemsetAsync(h->partials.p, 0, static_cast<size_t>(nblk) * ndt::kEvalStride * sizeof(double), h->stream));
  HIP_TRY(ndt::launch_calc_score(c->pts.p, static_cast<int>(n), h->grid->view(), gs.d1, gs.d2, gs.d3, h->search, kd_radius2(h->resolution), nblk,
                                 h->partials.p, h->stream));
  HIP_TRY(ndt::launch_reduce(h->partials.p, nblk, 1, nullptr, h->host_result, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *score = h->host_result[0] / static_cast<double>(n);
  return NDT_OK;
}

ndt_status ndt_get_fitness_score(ndt_handle h, double max_range, double* fitness) {
  if (!h || !fitness) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = check_ready(h);
  if (s) return s;
  return fitness_impl(h, h->source->pts.p, static_cast<int>(h->source->n), h->final_T, max_range, fitness);
}

static ndt_status voxel_filter_impl(ndt_handle h, const void* pts, size_t n, size_t stride, int is_dense, float leaf,
                                    bool on_device, void* out, size_t out_stride, size_t* n_out) {
  if (!h || !n_out || (n && !out) || !(leaf > 0)) return fail(NDT_ERR_INVALID, "bad arguments");
  *n_out = 0;
  std::shared_ptr<DeviceCloud> c;
  ndt_status s = upload_cloud(h, pts, n, stride, on_device, c);
  if (s) return s;
  if (n == 0) return NDT_OK;
  float4* d_out = on_device ? static_cast<float4*>(out) : nullptr;
  if (!on_device) {
    // Host buffer out: the centroid kernel writes straight into the handle's page-locked block (posted writes over the link,
    // inside the kernel's own time) instead of into HBM followed by a copy and a second synchronisation; the CPU then
    // spreads the records into the caller's buffer.  (60 k-point scan from a C++ caller, tools/probes/time_filter.cpp: 178-216 -> 158-182 us.)
    if (out_stride < 16) return fail(NDT_ERR_INVALID, "out_stride_bytes must be >= 16");
    const size_t bytes = n * sizeof(float4);
    if (h->out_pinned_bytes < bytes) {
      HIP_TRY(hipStreamSynchronize(h->stream));  // (an earlier download may still be using the old block)
      if (h->out_pinned) (void)hipHostFree(h->out_pinned);
      h->out_pinned = nullptr;
      h->out_pinned_bytes = 0;
      HIP_TRY(hipHostMalloc(&h->out_pinned, bytes + bytes / 4, hipHostMallocDefault));
      h->out_pinned_bytes = bytes + bytes / 4;
    }
    d_out = static_cast<float4*>(h->out_pinned);
  }
  size_t n_written = 0;
  bool overflow = false;
  const BBox bb = bbox_of(*c, is_dense);
  s = voxel_filter_device(h, c->pts.p, n, is_dense, leaf, d_out, &n_written, &overflow, &bb);  // (synchronises the stream)
  if (s) return s;
  if (!on_device && n_written) {
    if (out_stride == sizeof(float4)) {
      std::memcpy(out, h->out_pinned, n_written * sizeof(float4));
    } else {
      const unsigned char* src = static_cast<const unsigned char*>(h->out_pinned);
      unsigned char* dst = static_cast<unsigned char*>(out);
      for (size_t i = 0; i < n_written; i++) std::memcpy(dst + i * out_stride, src + i * sizeof(float4), sizeof(float4));
    }
  }
  *n_out = n_written;
  if (overflow) return fail(NDT_ERR_GRID_OVERFLOW, "leaf size is too small for the input dataset: integer indices would overflow");
  return NDT_OK;
}

ndt_status ndt_voxel_grid_filter(ndt_handle h, const void* pts, size_t n, size_t stride, int is_dense, float leaf, void* out,
                                 size_t out_stride, size_t* n_out) {
  return voxel_filter_impl(h, pts, n, stride, is_dense, leaf, false, out, out_stride, n_out);
}
ndt_status ndt_voxel_grid_filter_device(ndt_handle h, const void* d_pts, size_t n, size_t stride, int is_dense, float leaf,
                                        void* d_out, size_t* n_out) {
  return voxel_filter_impl(h, d_pts, n, stride, is_dense, leaf, true, d_out, 16, n_out);
}

// ---- N2: global map accumulation --------------------------------------------
// update_global_map of the mapping nodes (ndt_omp_mapping_node.cpp:195-211,
// ndt_rosbag_mapping_node.cpp:146-161): transformPointCloud(scan, pose); global_map += it;
// global_map = VoxelGrid(leaf).filter(global_map).  The map stays in HBM.
// the map's stream (created on first use) and the completion of a queued update
static ndt_status map_stream_of(ndt_handle h) {
  if (!h->map_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&h->map_stream, hipStreamNonBlocking));
    DevPool::instance().adopt_stream(h->map_stream);
    HIP_TRY(hipEventCreateWithFlags(&h->map_ready, hipEventDisableTiming));
  }
  return NDT_OK;
}
// waits for a queued map update: the map's size and boxes are current afterwards
static ndt_status map_complete(ndt_handle h) {
  if (!h->map_pending) return NDT_OK;
  h->map_pending = false;
  HIP_TRY(hipStreamSynchronize(h->map_stream));
  h->map_scan.reset();
  size_t n_new = 0;
  voxel_filter_finish(h->map_filter, &n_new, &h->map_boxes);
  h->map_boxes_known = true;
  h->map_n = n_new;
  return NDT_OK;
}

static ndt_status map_update_impl(ndt_handle h, const void* scan, size_t n, size_t stride, int is_dense, bool on_device,
                                  const float* pose, float leaf, int* overflowed, const std::shared_ptr<DeviceCloud>* resident = nullptr) {
  if (!h || !(leaf > 0)) return fail(NDT_ERR_INVALID, "bad arguments");
  if (overflowed) *overflowed = 0;
  std::shared_ptr<DeviceCloud> c;
  ndt_status s = NDT_OK;
  if (resident) c = *resident;  // an ndt_cloud: read where it lies
  else s = upload_cloud(h, scan, n, stride, on_device, c);
  if (s) return s;
  s = map_stream_of(h);
  if (!s) s = map_complete(h);  // the map as the previous update left it: its size and its boxes
  if (s) return s;
  const size_t total = h->map_n + n;
  if (total > static_cast<size_t>(std::numeric_limits<int>::max())) return fail(NDT_ERR_INVALID, "map too large");
  if (total == 0) return NDT_OK;
  hipStream_t ms = h->map_stream;
  // the scan was made on the handle's stream (an upload, a filter): the map's stream starts behind it; a resident cloud is
  // read by the map's stream from now on (its memory is not recycled before that stream has been waited for)
  HIP_TRY(hipEventRecord(h->map_ready, h->stream));
  HIP_TRY(hipStreamWaitEvent(ms, h->map_ready, 0));
  if (resident && c->made_on && c->made_on != ms && std::find(c->used_on.begin(), c->used_on.end(), ms) == c->used_on.end()) c->used_on.push_back(ms);
  const PoolStreamGuard guard(ms);  // the map's buffers come from (and go back to) the map stream's pool
  // concatenation [map | transformed scan] (operator+= keeps the map's points first): the scan is transformed straight into
  // the room behind the map -- the map is not copied
  if (h->map_pts.cap < total) {
    DevBuf<float4> bigger;
    HIP_TRY(bigger.reserve(total + total / 2 + n));
    if (h->map_n) HIP_TRY(hipMemcpyAsync(bigger.p, h->map_pts.p, h->map_n * sizeof(float4), hipMemcpyDeviceToDevice, ms));
    h->map_pts.swap(bigger);  // (the old block goes back to the pool behind the copy, stream order)
  }
  float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float* P = pose ? pose : I;
  if (n) {
    float T12[12];
    colmajor_to_T12(P, T12);
    HIP_TRY(ndt::launch_transform(c->pts.p, static_cast<int>(n), T12, h->map_pts.p + h->map_n, ms, is_dense));
  }
  HIP_TRY(h->map_alt.reserve(total + total / 2 + n));
  // the accumulated map is dense only if every scan was; PCL carries is_dense through operator+=
  h->map_dense = (h->map_n == 0 ? 1 : h->map_dense) && is_dense;
  // A box for the filter without a pass over the points: the map's own box (the last pass left it) joined with the box of the
  // scan's box under the pose, padded for the f32 rounding of the transform.  ANY box that holds the points gives the same
  // voxels in the same order -- a voxel is floor(x / leaf) whatever min_b is, and the linear index orders the voxels by
  // (z, y, x) for every box -- so the result is PCL's bit for bit; only the index-overflow test wants the exact box, and it
  // is computed (one pass, one wait) when the padded one comes near overflowing.
  BBox guess{};
  bool have_guess = (h->map_n == 0 || h->map_boxes_known);
  const int v = h->map_dense ? 0 : 1;
  if (have_guess) {
    for (int k = 0; k < 3; k++) {
      guess.mn[k] = h->map_n ? h->map_boxes.bb_min[v][k] : FLT_MAX;
      guess.mx[k] = h->map_n ? h->map_boxes.bb_max[v][k] : -FLT_MAX;
    }
    if (n) {
      const BBox sb = bbox_of(*c, is_dense);
      if (sb.mn[0] <= sb.mx[0]) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, mag = 0;
        for (int corner = 0; corner < 8; corner++) {
          const double q[3] = {(corner & 1) ? sb.mx[0] : sb.mn[0], (corner & 2) ? sb.mx[1] : sb.mn[1], (corner & 4) ? sb.mx[2] : sb.mn[2]};
          for (int r = 0; r < 3; r++) {
            double a = P[12 + r], m = std::fabs(a);
            for (int k = 0; k < 3; k++) {
              a += static_cast<double>(P[4 * k + r]) * q[k];
              m += std::fabs(static_cast<double>(P[4 * k + r]) * q[k]);
            }
            lo[r] = std::min(lo[r], a);
            hi[r] = std::max(hi[r], a);
            mag = std::max(mag, m);
          }
        }
        const double pad = 1e-5 * mag + 1e-6;  // (a transformed coordinate is three f32 multiply-adds: a few ulps of the terms)
        for (int r = 0; r < 3; r++) {
          guess.mn[r] = std::min(guess.mn[r], static_cast<float>(lo[r] - pad));
          guess.mx[r] = std::max(guess.mx[r], static_cast<float>(hi[r] + pad));
        }
        for (int r = 0; r < 3; r++) have_guess = have_guess && std::isfinite(guess.mn[r]) && std::isfinite(guess.mx[r]);
      }
    }
    if (have_guess && guess.mn[0] <= guess.mx[0]) {  // would the padded box overflow the index space?  then the exact one decides
      long long d[3];
      for (int k = 0; k < 3; k++) d[k] = static_cast<long long>((guess.mx[k] - guess.mn[k]) * (1.0f / leaf)) + 1;
      if (d[0] * d[1] * d[2] > static_cast<long long>(std::numeric_limits<int32_t>::max()) / 2) have_guess = false;
    } else {
      have_guess = false;
    }
  }
  if (!have_guess) {  // the exact box: one pass over [map | scan] and a wait for it
    BBox exact;
    HIP_TRY(hipStreamSynchronize(ms));
    const hipStream_t keep_stream = h->stream;
    h->stream = ms;  // (bbox_compute launches on and waits for the handle's stream)
    s = bbox_compute(h, h->map_pts.p, static_cast<int>(total), h->map_dense, exact);
    h->stream = keep_stream;
    if (s) return s;
    guess = exact;
  }
  s = filter_slots(h, 1, h->map_filter);
  if (!s) s = voxel_filter_enqueue(h, ms, h->map_pts.p, total, h->map_dense, leaf, h->map_alt.p, guess, h->map_filter);
  if (s) return s;
  h->map_pts.swap(h->map_alt);
  h->map_scan = c;
  h->map_pending = true;  // (its size and boxes: map_complete, when somebody needs them)
  if (overflowed) *overflowed = h->map_filter.overflow ? 1 : 0;
  return NDT_OK;
}

ndt_status ndt_map_clear(ndt_handle h) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  if (h->map_pending) {
    ndt_status s = ensure_device(h);
    if (!s) s = map_complete(h);
    if (s) return s;
  }
  h->map_n = 0;
  h->map_dense = 1;
  h->map_boxes_known = false;
  return NDT_OK;
}
// ---- ndt_cloud: clouds that stay in HBM between the steps of a node's loop ----------------------------------------------
// the cloud is about to be read by work on h's stream: order that stream behind the cloud's making, remember it for the
// cloud's release
static ndt_status cloud_use_on(ndt_handle h, DeviceCloud* c) {
  if (c->made_on && c->made_on != h->stream) {
    if (c->device != h->device) return fail(NDT_ERR_INVALID, "the cloud lives on another device");
    HIP_TRY(hipStreamSynchronize(c->made_on));
    if (std::find(c->used_on.begin(), c->used_on.end(), h->stream) == c->used_on.end()) c->used_on.push_back(h->stream);
  }
  return NDT_OK;
}

ndt_status ndt_cloud_voxel_filter(ndt_handle h, const void* pts, size_t n, size_t stride, int is_dense, float leaf, int on_device,
                                  ndt_cloud* out, int* overflowed) {
  if (!h || !out || !(leaf > 0)) return fail(NDT_ERR_INVALID, "bad arguments");
  *out = nullptr;
  if (overflowed) *overflowed = 0;
  std::shared_ptr<DeviceCloud> in;
  // (a device cloud of 16-byte records is read where it lies; everything it is needed for is over when this returns)
  const bool ref_ok = on_device && n > 0 && stride == sizeof(float4) && (reinterpret_cast<uintptr_t>(pts) & 15) == 0;
  ndt_status s = upload_cloud(h, pts, n, stride, on_device != 0, in, ref_ok);
  if (s) return s;
  auto c = std::make_shared<DeviceCloud>();
  c->device = h->device;
  c->made_on = h->stream;
  HIP_TRY(c->pts.reserve(std::max<size_t>(n, 1)));
  size_t n_written = 0;
  bool overflow = false;
  if (n) {
    const BBox bb = bbox_of(*in, is_dense);
    s = voxel_filter_device(h, in->pts.p, n, is_dense, leaf, c->pts.p, &n_written, &overflow, &bb, c.get());
    if (s) return s;
  }
  c->n = n_written;
  if (overflowed) *overflowed = overflow ? 1 : 0;
  *out = new ndt_cloud_s{c};
  return NDT_OK;
}

// N1 of an ndt_cloud, in two halves: begin queues the whole chain on the handle's FILTER stream and returns (the input's boxes
// are known: nothing has to come back from the device before the chain can be queued); end waits for it.  Between the two the
// caller registers the previous scan on the handle's own stream.
ndt_status ndt_cloud_voxel_filter_begin(ndt_handle h, ndt_cloud in, int is_dense, float leaf) {
  if (!h || !in || !(leaf > 0)) return fail(NDT_ERR_INVALID, "bad arguments");
  if (h->n1_pending) return fail(NDT_ERR_INVALID, "a prefilter has been begun and not ended");
  ndt_status s = ensure_device(h);
  if (s) return s;
  if (!h->filter_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&h->filter_stream, hipStreamNonBlocking));
    DevPool::instance().adopt_stream(h->filter_stream);
  }
  DeviceCloud* ic = in->c.get();
  if (ic->made_on && ic->made_on != h->filter_stream) {  // made elsewhere: complete before the filter stream reads it
    if (ic->device != h->device) return fail(NDT_ERR_INVALID, "the cloud lives on another device");
    HIP_TRY(hipStreamSynchronize(ic->made_on));
    if (std::find(ic->used_on.begin(), ic->used_on.end(), h->filter_stream) == ic->used_on.end()) ic->used_on.push_back(h->filter_stream);
  }
  auto c = std::make_shared<DeviceCloud>();
  c->device = h->device;
  c->made_on = h->filter_stream;
  {
    const PoolStreamGuard guard(h->filter_stream);
    HIP_TRY(c->pts.reserve(std::max<size_t>(ic->n, 1)));
  }
  s = filter_slots(h, 2, h->n1_filter);
  if (!s) s = voxel_filter_enqueue(h, h->filter_stream, ic->pts.p, ic->n, is_dense, leaf, c->pts.p, bbox_of(*ic, is_dense), h->n1_filter);
  if (s) return s;
  h->n1_in = in->c;
  h->n1_out = c;
  h->n1_pending = true;
  return NDT_OK;
}
ndt_status ndt_cloud_voxel_filter_end(ndt_handle h, ndt_cloud* out, int* overflowed) {
  if (!h || !out) return fail(NDT_ERR_INVALID, "bad arguments");
  *out = nullptr;
  if (!h->n1_pending) return fail(NDT_ERR_INVALID, "no prefilter has been begun");
  ndt_status s = ensure_device(h);
  if (s) return s;
  h->n1_pending = false;
  HIP_TRY(hipStreamSynchronize(h->filter_stream));
  size_t n_written = 0;
  voxel_filter_finish(h->n1_filter, &n_written, h->n1_out.get());
  h->n1_out->n = n_written;
  if (overflowed) *overflowed = h->n1_filter.overflow ? 1 : 0;
  *out = new ndt_cloud_s{h->n1_out};
  h->n1_in.reset();
  h->n1_out.reset();
  return NDT_OK;
}

ndt_status ndt_cloud_upload(ndt_handle h, const void* pts, size_t n, size_t stride, ndt_cloud* out) {
  if (!h || !out) return fail(NDT_ERR_INVALID, "bad arguments");
  *out = nullptr;
  std::shared_ptr<DeviceCloud> c;
  ndt_status s = upload_cloud(h, pts, n, stride, false, c);
  if (s) return s;
  c->device = h->device;
  c->made_on = h->stream;
  *out = new ndt_cloud_s{c};
  return NDT_OK;
}

ndt_status ndt_cloud_size(ndt_cloud c, size_t* n) {
  if (!c || !n) return fail(NDT_ERR_INVALID, "bad arguments");
  *n = c->c->n;
  return NDT_OK;
}
ndt_status ndt_cloud_data(ndt_cloud c, const void** d_pts, size_t* n) {
  if (!c || !d_pts || !n) return fail(NDT_ERR_INVALID, "bad arguments");
  *d_pts = c->c->pts.p;
  *n = c->c->n;
  return NDT_OK;
}
ndt_status ndt_cloud_download(ndt_handle h, ndt_cloud c, void* out, size_t out_stride) {
  if (!h || !c || (c->c->n && !out)) return fail(NDT_ERR_INVALID, "bad arguments");
  if (out_stride < 16) return fail(NDT_ERR_INVALID, "out_stride_bytes must be >= 16");
  ndt_status s = ensure_device(h);
  if (!s) s = cloud_use_on(h, c->c.get());
  if (s) return s;
  return download_records(h, c->c->pts.p, c->c->n, out, out_stride);
}
void ndt_cloud_release(ndt_cloud c) { delete c; }

ndt_status ndt_set_input_source_cloud(ndt_handle h, ndt_cloud c) {
  if (!h || !c) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = ensure_device(h);
  if (!s) s = cloud_use_on(h, c->c.get());
  if (s) return s;
  // Big scans are registered from a copy in lattice order whose pitch is this handle's resolution (order_cloud): that copy
  // belongs to the handle, not to the shared cloud -- a view of the cloud's points with an ordered copy of its own.
  auto view = std::make_shared<DeviceCloud>();
  view->pts.borrow(c->c->pts.p, c->c->n);
  view->n = c->c->n;
  std::memcpy(view->bb_min, c->c->bb_min, sizeof(view->bb_min));
  std::memcpy(view->bb_max, c->c->bb_max, sizeof(view->bb_max));
  s = order_cloud(h, view.get(), nullptr, 0);
  if (s) return s;
  if (view->n_sorted == 0) {  // (the usual case at the nodes' size: nothing to order, the cloud itself is the source)
    h->source = c->c;
    return NDT_OK;
  }
  view->parent = c->c;
  h->source = view;
  return NDT_OK;
}
ndt_status ndt_set_input_target_cloud(ndt_handle h, ndt_cloud c, int is_dense) {
  if (!h || !c) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = ensure_device(h);
  if (!s) s = cloud_use_on(h, c->c.get());
  if (s) return s;
  h->target = c->c;
  h->target_dense = is_dense ? 1 : 0;
  return build_grid(h);
}
ndt_status ndt_promote_source_to_target(ndt_handle h, int is_dense) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  if (!h->source) return fail(NDT_ERR_NO_INPUT, "no input source to promote");
  ndt_status s = ensure_device(h);
  if (s) return s;
  // the resident points and their boxes: no upload, no repack, no bounding-box pass (a view made for ordering: its parent)
  h->target = h->source->parent ? h->source->parent : h->source;
  h->target_dense = is_dense ? 1 : 0;
  return build_grid(h);
}

ndt_status ndt_map_update_cloud(ndt_handle h, ndt_cloud scan, int is_dense, const float* pose, float leaf, int* overflowed) {
  if (!h || !scan) return fail(NDT_ERR_INVALID, "bad arguments");
  ndt_status s = ensure_device(h);
  if (!s) s = cloud_use_on(h, scan->c.get());
  if (s) return s;
  return map_update_impl(h, nullptr, scan->c->n, sizeof(float4), is_dense, true, pose, leaf, overflowed, &scan->c);
}

ndt_status ndt_warm_up(ndt_handle h, size_t expected_scan_points) {
  if (!h) return fail(NDT_ERR_INVALID, "null handle");
  ndt_status s = ensure_device(h);
  if (!s) s = ensure_host_rows(h, 1);
  if (s) return s;
  if (!h->bbox_rows) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->bbox_rows), 1024 * 12 * sizeof(float), hipHostMallocDefault));
  // The loop's calls once on this handle's stream (its memory pool, its staging buffers), on a synthetic scan of the expected
  // size -- a slab of 100 m x 100 m x 10 m, roughly a lidar sweep's extent: code object, kernels, page-locked slots, and pool
  // blocks of the sizes the real scans will ask for.  The handle's inputs, results and map are put back afterwards.
  const size_t n = std::max<size_t>(expected_scan_points, 1152);
  std::vector<float> pts(4 * n);
  unsigned long long z = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {  // splitmix64 -> [0, 1)
    z += 0x9E3779B97F4A7C15ull;
    unsigned long long x = z;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return static_cast<float>(x >> 40) * (1.0f / 16777216.0f);
  };
  for (size_t i = 0; i < n; i++) {
    pts[4 * i] = 100.0f * rnd() - 50.0f;
    pts[4 * i + 1] = 100.0f * rnd() - 50.0f;
    pts[4 * i + 2] = (i & 3) ? 0.2f * rnd() : 10.0f * rnd() - 5.0f;  // mostly a ground sheet: voxels with enough points
    pts[4 * i + 3] = 1.0f;
  }
  const auto keep_source = h->source;
  const auto keep_target = h->target;
  const auto keep_grid = h->grid;
  const int keep_dense = h->target_dense, keep_iter = h->max_iter;
  const int keep_conv = h->converged, keep_nr = h->nr_iterations, keep_evals = h->n_evals, keep_hess = h->n_hess;
  const double keep_prob = h->trans_probability, keep_nn = h->mean_neighbors;
  float keep_T[16];
  std::memcpy(keep_T, h->final_T, sizeof(keep_T));
  s = map_complete(h);
  if (s) return s;
  DevBuf<float4> keep_map, keep_alt;
  keep_map.swap(h->map_pts);
  keep_alt.swap(h->map_alt);
  const size_t keep_map_n = h->map_n;
  const int keep_map_dense = h->map_dense;
  const bool keep_boxes_known = h->map_boxes_known;
  h->map_n = 0;
  h->map_boxes_known = false;
  h->max_iter = 2;
  ndt_cloud c = nullptr, d = nullptr;
  int ov = 0, conv = 0, it = 0;
  float T[16];
  double prob = 0;
  for (int round = 0; round < 2 && !s; round++) {  // (twice: the second round finds the pool's blocks and leaves them sized)
    s = ndt_cloud_voxel_filter(h, pts.data(), n, 16, 1, 0.5f, 0, &c, &ov);
    if (!s) s = ndt_cloud_voxel_filter(h, pts.data(), n, 16, 1, 0.3f, 0, &d, &ov);
    if (!s) s = ndt_set_input_target_cloud(h, c, 1);
    if (!s) s = ndt_set_input_source_cloud(h, d);
    if (!s) s = ndt_align(h, nullptr, T, &conv, &it, &prob, nullptr, 0);
    if (!s) s = ndt_map_update_cloud(h, c, 1, nullptr, 0.5f, &ov);
    if (!s) s = ndt_map_update_cloud(h, d, 1, nullptr, 0.5f, &ov);
    ndt_cloud_release(c);
    ndt_cloud_release(d);
    c = d = nullptr;
    if (!s) {  // the two-halves prefilter: its stream, and that stream's pool
      ndt_cloud raw = nullptr;
      s = ndt_cloud_upload(h, pts.data(), n, 16, &raw);
      if (!s) s = ndt_cloud_voxel_filter_begin(h, raw, 1, 0.5f);
      if (!s) s = ndt_cloud_voxel_filter_end(h, &c, &ov);
      ndt_cloud_release(raw);
      ndt_cloud_release(c);
      c = nullptr;
    }
  }
  if (!s) s = map_complete(h);
  if (!s) HIP_TRY(hipStreamSynchronize(h->stream));
  {  // the scratch map's buffers go back to the map stream's pool, the handle's own map comes back
    const PoolStreamGuard guard(h->map_stream ? h->map_stream : h->stream);
    h->map_pts.release();
    h->map_alt.release();
  }
  h->source = keep_source;
  h->target = keep_target;
  h->grid = keep_grid;
  h->target_dense = keep_dense;
  h->max_iter = keep_iter;
  h->map_pts.swap(keep_map);
  h->map_alt.swap(keep_alt);
  h->map_n = keep_map_n;
  h->map_dense = keep_map_dense;
  h->map_boxes_known = keep_boxes_known && keep_map_n == 0 ? false : keep_boxes_known;
  h->converged = keep_conv;
  h->nr_iterations = keep_nr;
  h->n_evals = keep_evals;
  h->n_hess = keep_hess;
  h->trans_probability = keep_prob;
  h->mean_neighbors = keep_nn;
  std::memcpy(h->final_T, keep_T, sizeof(keep_T));
  return s;
}

ndt_status ndt_map_update(ndt_handle h, const void* scan, size_t n, size_t stride, int is_dense, const float* pose, float leaf,
                          int* overflowed) {
  return map_update_impl(h, scan, n, stride, is_dense, false, pose, leaf, overflowed);
}
ndt_status ndt_map_update_device(ndt_handle h, const void* d_scan, size_t n, size_t stride, int is_dense, const float* pose,
                                 float leaf, int* overflowed) {
  return map_update_impl(h, d_scan, n, stride, is_dense, true, pose, leaf, overflowed);
}
ndt_status ndt_map_size(ndt_handle h, size_t* n) {
  if (!h || !n) return fail(NDT_ERR_INVALID, "bad arguments");
  if (h->map_pending) {
    ndt_status s = ensure_device(h);
    if (!s) s = map_complete(h);
    if (s) return s;
  }
  *n = h->map_n;
  return NDT_OK;
}
ndt_status ndt_map_get(ndt_handle h, void* out, size_t out_stride) {
  if (!h) return fail(NDT_ERR_INVALID, "bad arguments");
  if (out_stride < 16) return fail(NDT_ERR_INVALID, "out_stride_bytes must be >= 16");
  if (h->map_pending || h->map_stream) {
    ndt_status s = ensure_device(h);
    if (!s) s = map_complete(h);
    if (s) return s;
    HIP_TRY(hipStreamSynchronize(h->map_stream));  // (the download runs on the handle's stream)
  }
  if (h->map_n && !out) return fail(NDT_ERR_INVALID, "bad arguments");
  if (h->map_n == 0) return NDT_OK;
  return download_records(h, h->map_pts.p, h->map_n, out, out_stride);
}
ndt_status ndt_map_get_device(ndt_handle h, const void** d_pts, size_t* n) {
  if (!h || !d_pts || !n) return fail(NDT_ERR_INVALID, "bad arguments");
  if (h->device_ready) HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->map_stream) {
    ndt_status s = ensure_device(h);
    if (!s) s = map_complete(h);
    if (s) return s;
    HIP_TRY(hipStreamSynchronize(h->map_stream));
  }
  *d_pts = h->map_pts.p;
  *n = h->map_n;
  return NDT_OK;
}
void ndt_host_chain_pose(const float* pose, const float* transform, float* out) { ndt::chain_pose(pose, transform, out); }

ndt_status ndt_grid_size(ndt_handle h, size_t* n_leaves, size_t* n_valid) {
  if (!h || !h->grid) return fail(NDT_ERR_NO_INPUT, "no grid");
  if (!h->grid->empty) {
    ndt_status s = ensure_device(h);
    if (!s) s = grid_counts(h, h->grid.get());
    if (s) return s;
  }
  if (n_leaves) *n_leaves = h->grid->n_leaves;
  if (n_valid) *n_valid = h->grid->n_valid;
  return NDT_OK;
}

ndt_status ndt_grid_info(ndt_handle h, int* min_b, int* max_b, int* div_b) {
  if (!h || !h->grid) return fail(NDT_ERR_NO_INPUT, "no grid");
  for (int k = 0; k < 3; k++) {
    if (min_b) min_b[k] = h->grid->geom.min_b[k];
    if (max_b) max_b[k] = h->grid->geom.max_b[k];
    if (div_b) div_b[k] = h->grid->geom.div_b[k];
  }
  return NDT_OK;
}

// Re-runs the finalize pass in dump mode (the records and LUT it rewrites are
// bit-identical, so sharing handles stay valid).
ndt_status ndt_grid_dump(ndt_handle h, int64_t* idx, int* nr_points, double* mean, double* cov, double* icov,
                         double* evals) {
  if (!h || !h->grid) return fail(NDT_ERR_NO_INPUT, "no grid");
  DeviceGrid* g = h->grid.get();
  if (!g->empty) {
    ndt_status sc = ensure_device(h);
    if (!sc) sc = grid_counts(h, g);
    if (sc) return sc;
  }
  const size_t V = g->n_leaves;
  if (V == 0) return NDT_OK;
  ndt_status s = ensure_device(h);
  if (s) return s;
  DevBuf<int> d_n;
  DevBuf<double> d_mean, d_cov, d_icov, d_evals;
  DevBuf<unsigned> d_cnt;
  HIP_TRY(d_n.reserve(V));
  HIP_TRY(d_mean.reserve(V * 3));
  HIP_TRY(d_cov.reserve(V * 9));
  HIP_TRY(d_icov.reserve(V * 9));
  HIP_TRY(d_evals.reserve(V * 3));
  HIP_TRY(d_cnt.reserve(1));
  HIP_TRY(hipMemsetAsync(d_cnt.p, 0, sizeof(unsigned), h->stream));
  ndt::FinalizeDump dump{d_n.p, d_mean.p, d_cov.p, d_icov.p, d_evals.p};
  HIP_TRY(ndt::launch_finalize(g->target->pts.p, g->leaf_cell.p, g->leaf_start.p, g->leaf_count.p, g->leaf_rec.p,
                               static_cast<int>(V), g->sorted_idx.p, g->min_pts, g->eig_ratio, g->recs.p, g->centroids.p, g->lut.p,
                               g->geom, d_cnt.p, dump, h->stream));
  std::vector<int> cell(V);
  HIP_TRY(hipMemcpyAsync(cell.data(), g->leaf_cell.p, V * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (nr_points) HIP_TRY(hipMemcpyAsync(nr_points, d_n.p, V * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (mean) HIP_TRY(hipMemcpyAsync(mean, d_mean.p, V * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (cov) HIP_TRY(hipMemcpyAsync(cov, d_cov.p, V * 9 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (icov) HIP_TRY(hipMemcpyAsync(icov, d_icov.p, V * 9 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (evals) HIP_TRY(hipMemcpyAsync(evals, d_evals.p, V * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  // ascending voxel index, the order of the reference's std::map (a bucket-form build numbers its leaves bucket by bucket)
  bool ascending = true;
  for (size_t i = 1; i < V && ascending; i++) ascending = cell[i - 1] < cell[i];
  if (!ascending) {
    std::vector<size_t> perm(V);
    for (size_t i = 0; i < V; i++) perm[i] = i;
    std::sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return cell[a] < cell[b]; });
    auto apply = [&](auto* arr, size_t width) {
      if (!arr) return;
      using T = std::remove_pointer_t<decltype(arr)>;
      std::vector<T> tmp(arr, arr + V * width);
      for (size_t i = 0; i < V; i++) std::copy(tmp.begin() + perm[i] * width, tmp.begin() + (perm[i] + 1) * width, arr + i * width);
    };
    apply(nr_points, 1);
    apply(mean, 3);
    apply(cov, 9);
    apply(icov, 9);
    apply(evals, 3);
    std::vector<int> sorted_cell(V);
    for (size_t i = 0; i < V; i++) sorted_cell[i] = cell[perm[i]];
    cell.swap(sorted_cell);
  }
  if (idx)
    for (size_t i = 0; i < V; i++) idx[i] = cell[i];
  return NDT_OK;
}

}  // extern "C"
