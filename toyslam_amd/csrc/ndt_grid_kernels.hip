// ndt_grid_kernels.hip -- K1: everything that turns a cloud into a voxel structure (gfx950, wave64).
//
//  target voxel grid  (VoxelGridCovariance::applyFilter, voxel_grid_covariance_omp_impl.hpp:48-370)
//    general chain:  bbox -> per-cell count -> 3-phase exclusive scan over cells (leaf ordinals, segment offsets, record
//                    ordinals, LUT init) -> counting-sort scatter of point indices -> k_presort_large (crowded voxels) ->
//                    k_finalize
//    bucket form:    k1_hist -> k1_colscan -> k1_scatter -> k1_finalize (order-preserving: no global atomics, no sort by
//                    point index; crowded cells summed by lane teams), k1_count / k1_leaves on demand
//    (the sort-based sparse form is ndt_sparse.hip; all three end in finish_voxel: index-ordered f64 sums -- bit-identical
//    to the reference's sequential accumulation --, mean, covariance with the reference's quirks, 3x3 symmetric
//    eigen-solve, eigenvalue inflation, inverse, validity -> 64-B VoxelRec)
//  The same count / scan / scatter machinery serves the scan prefilter (N1, k_voxel_centroids), the global-map update (N2)
//  and the spatial ordering of source scans (k_sort_gather; lock-step batches: one lattice per scan, k_scan_bboxes /
//  k_count_batch).
//
// The evaluation kernels (K2) are in ndt_kernels.hip (throughput side) and ndt_latency.hip (single-scan latency path).
#include "ndt_device.hpp"

namespace ndt {

namespace {

// ---------------------------------------------------------------------------
// repack: arbitrary-stride xyz records -> dense float4 (x,y,z,1)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_repack(const unsigned char* __restrict__ src, size_t n, size_t stride,
                                                   float4* __restrict__ dst) {
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const float* p = reinterpret_cast<const float*>(src + i * stride);
    dst[i] = make_float4(p[0], p[1], p[2], 1.0f);
  }
}

// ---------------------------------------------------------------------------
// K1.a  bounding box  ([PCL] getMinMax3D, _impl.hpp:72)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_bbox(const float4* __restrict__ pts, int n, int dense,
                                                 float* __restrict__ block_minmax) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 p = pts[i];
    if (!dense && !finite3(p.x, p.y, p.z)) continue;
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
  }
  __shared__ float s[kBlock / kWave][6];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float a = wave_min(mn[k]), b = wave_max(mx[k]);
    if (lane == 0) { s[wave][k] = a; s[wave][3 + k] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s[0][threadIdx.x];
    for (int w = 1; w < kBlock / kWave; w++) v = (threadIdx.x < 3) ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
    block_minmax[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

// repack + both bounding boxes in one pass over the upload: block_minmax[block][12] =
// {min xyz, max xyz} over the points that are not NaN (what getMinMax3D sees for an is_dense cloud) and
// {min xyz, max xyz} over the finite points (the !is_dense rule).  The host reduces the per-block rows
// behind the synchronisation the upload needs anyway, so no consumer launches k_bbox or waits again.
__global__ __launch_bounds__(kBlock) void k_repack_bbox(const unsigned char* __restrict__ src, size_t n, size_t stride,
                                                        float4* __restrict__ dst, float* __restrict__ block_minmax) {
  float mn[6] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX};
  float mx[6] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const float* p = reinterpret_cast<const float*>(src + i * stride);
    const float x = p[0], y = p[1], z = p[2];
    dst[i] = make_float4(x, y, z, 1.0f);
    mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);  // fminf / fmaxf drop NaN operands
    mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
    mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
    if (finite3(x, y, z)) {
      mn[3] = fminf(mn[3], x); mx[3] = fmaxf(mx[3], x);
      mn[4] = fminf(mn[4], y); mx[4] = fmaxf(mx[4], y);
      mn[5] = fminf(mn[5], z); mx[5] = fmaxf(mx[5], z);
    }
  }
  __shared__ float s[kBlock / kWave][12];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float a = wave_min(mn[k]), b = wave_max(mx[k]);
    const int base = (k < 3) ? 0 : 6, c = k % 3;
    if (lane == 0) { s[wave][base + c] = a; s[wave][base + 3 + c] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const bool is_min = (threadIdx.x % 6) < 3;
    float v = s[0][threadIdx.x];
    for (int w = 1; w < kBlock / kWave; w++) v = is_min ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
    block_minmax[blockIdx.x * 12 + threadIdx.x] = v;
  }
}

// The same for clouds that already are dense 16-byte records (pcl::PointXYZ in HBM): 16-byte loads, and -- COPY false --
// no copy at all: a cloud handed over by reference only needs its bounding boxes (ndt_set_input_*_device_ref).
// tag != 0: the row goes out as 12 self-validating words, (value bits << 32) | tag, so that the host can POLL the pinned rows
// instead of synchronising the stream (a cloud handed over by reference has no copy the host would have to wait for anyway)
template <bool COPY>
__global__ __launch_bounds__(kBlock) void k_bbox16(const float4* __restrict__ src, size_t n, float4* __restrict__ dst,
                                                  float* __restrict__ block_minmax, unsigned tag, const unsigned* __restrict__ n_dev) {
  if (n_dev) n = *n_dev;  // (a count that is still on the device: the voxel filter's output)
  float mn[6] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX};
  float mx[6] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  const size_t stride = static_cast<size_t>(gridDim.x) * kBlock;
  for (size_t i0 = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i0 < n; i0 += 8 * stride) {  // eight loads in flight
    float4 p[8];
#pragma unroll
    for (int u = 0; u < 8; u++) p[u] = (i0 + u * stride < n) ? src[i0 + u * stride] : make_float4(NAN, NAN, NAN, 1.0f);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (i0 + u * stride >= n) continue;
      const float x = p[u].x, y = p[u].y, z = p[u].z;
      if (COPY) dst[i0 + u * stride] = make_float4(x, y, z, 1.0f);
      mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);  // fminf / fmaxf drop NaN operands
      mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
      mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
      if (finite3(x, y, z)) {
        mn[3] = fminf(mn[3], x); mx[3] = fmaxf(mx[3], x);
        mn[4] = fminf(mn[4], y); mx[4] = fmaxf(mx[4], y);
        mn[5] = fminf(mn[5], z); mx[5] = fmaxf(mx[5], z);
      }
    }
  }
  __shared__ float s[kBlock / kWave][12];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float a = wave_min(mn[k]), b = wave_max(mx[k]);
    const int base = (k < 3) ? 0 : 6, c = k % 3;
    if (lane == 0) { s[wave][base + c] = a; s[wave][base + 3 + c] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const bool is_min = (threadIdx.x % 6) < 3;
    float v = s[0][threadIdx.x];
    for (int w = 1; w < kBlock / kWave; w++) v = is_min ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
    if (tag)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(block_minmax) + blockIdx.x * 12 + threadIdx.x,
                         (static_cast<unsigned long long>(__float_as_uint(v)) << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
      block_minmax[blockIdx.x * 12 + threadIdx.x] = v;
  }
}

// linear voxel index of a target point while BUILDING the grid:
// floor(x * inv_leaf) - float(min_b), _impl.hpp:218-223 (f32, trap 2)
__device__ __forceinline__ int build_cell(const GridGeom& g, float x, float y, float z) {
  // plain operators under contract(off): the product must be rounded to f32 before floor()
#pragma clang fp contract(off)
  const float fx = x * g.inv_leaf[0], fy = y * g.inv_leaf[1], fz = z * g.inv_leaf[2];
  const int i0 = static_cast<int>(floorf(fx) - static_cast<float>(g.min_b[0]));
  const int i1 = static_cast<int>(floorf(fy) - static_cast<float>(g.min_b[1]));
  const int i2 = static_cast<int>(floorf(fz) - static_cast<float>(g.min_b[2]));
  return i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2];
}

// ---------------------------------------------------------------------------
// K1.b  per-cell point count
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_count(const float4* __restrict__ pts, int n, int dense, GridGeom g,
                                                  int* __restrict__ key, unsigned* __restrict__ rank,
                                                  unsigned* __restrict__ cell_count) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 p = pts[i];
    int c = -1;
    if (dense || finite3(p.x, p.y, p.z)) {
      c = build_cell(g, p.x, p.y, p.z);
      // points are inside the bbox by construction; guard against NaN/garbage
      if (c < 0 || static_cast<long long>(c) >= g.n_cells) c = -1;
    }
    key[i] = c;
    // the returned count is the point's arrival rank inside its cell: the scatter needs no second
    // round of atomics
    if (c >= 0) rank[i] = atomicAdd(&cell_count[c], 1u);
  }
}

// Batch variant for the source ordering of many scans at once: blockIdx.y = scan, composite key = the scan's base + its
// cell on the scan's OWN lattice (ScanLattice), so one count / scan / scatter pass orders every scan inside its own
// segment and a scan's order is a function of its own points only.
__device__ __forceinline__ int enc_f32(float f) {  // order-preserving: a < b  <=>  enc(a) < enc(b)  (finite values)
  const int b = __float_as_int(f);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__global__ __launch_bounds__(kBlock) void k_scan_bboxes(const float4* __restrict__ pts, const int* __restrict__ scan_off, int* __restrict__ out) {
  const int lo = scan_off[blockIdx.y], hi = scan_off[blockIdx.y + 1];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = lo + blockIdx.x * kBlock + threadIdx.x; i < hi; i += gridDim.x * kBlock) {
    const float4 p = pts[i];
    if (!finite3(p.x, p.y, p.z)) continue;
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, kWave));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, kWave));
    }
  }
  // one set of atomics per block, not per wave (hundreds of waves meet on the six words of a scan)
  __shared__ float s_mm[kBlock / kWave][6];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0)
    for (int k = 0; k < 3; k++) { s_mm[wave][k] = mn[k]; s_mm[wave][3 + k] = mx[k]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s_mm[0][threadIdx.x];
    for (int w = 1; w < kBlock / kWave; w++) v = (threadIdx.x < 3) ? fminf(v, s_mm[w][threadIdx.x]) : fmaxf(v, s_mm[w][threadIdx.x]);
    int* o = out + 6 * blockIdx.y;
    if (threadIdx.x < 3) { if (v < INFINITY) atomicMin(o + threadIdx.x, enc_f32(v)); }
    else if (v > -INFINITY) atomicMax(o + threadIdx.x, enc_f32(v));
  }
}

__global__ __launch_bounds__(kBlock) void k_count_batch(const float4* __restrict__ pts, const int* __restrict__ scan_off,
                                                        const ScanLattice* __restrict__ lat, int* __restrict__ key,
                                                        unsigned* __restrict__ rank, unsigned* __restrict__ cell_count) {
#pragma clang fp contract(off)
  const int lo = scan_off[blockIdx.y], hi = scan_off[blockIdx.y + 1];
  const ScanLattice L = lat[blockIdx.y];
  for (int i = lo + blockIdx.x * kBlock + threadIdx.x; i < hi; i += gridDim.x * kBlock) {
    const float4 p = pts[i];
    int c = -1;
    if (L.n_cells > 0 && finite3(p.x, p.y, p.z)) {
      const int i0 = static_cast<int>(floorf(p.x * L.inv_leaf)) - L.min_b[0];
      const int i1 = static_cast<int>(floorf(p.y * L.inv_leaf)) - L.min_b[1];
      const int i2 = static_cast<int>(floorf(p.z * L.inv_leaf)) - L.min_b[2];
      const int cell = i0 + i1 * L.mul1 + i2 * L.mul2;
      if (i0 >= 0 && i1 >= 0 && i2 >= 0 && cell >= 0 && cell < L.n_cells) c = static_cast<int>(L.base + cell);
    }
    key[i] = c;
    if (c >= 0) rank[i] = atomicAdd(&cell_count[c], 1u);
  }
}

__global__ __launch_bounds__(kBlock) void k_pick(const unsigned* __restrict__ cell_count, const long long* __restrict__ bases,
                                                 unsigned* __restrict__ out, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = cell_count[bases[i]];
}

// ---------------------------------------------------------------------------
// K1.c  exclusive scan over cells of {points, occupied, candidate} counters
// ---------------------------------------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;  // 2048 cells per block

struct U3 {
  unsigned pts, occ, cand;
};
__device__ __forceinline__ U3 operator+(const U3& a, const U3& b) { return {a.pts + b.pts, a.occ + b.occ, a.cand + b.cand}; }

// exclusive block scan of one U3 per thread; returns the exclusive prefix and the block total
__device__ __forceinline__ U3 block_exclusive_scan(U3 v, U3& total, U3* lds /*[kBlock/kWave]*/) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  U3 inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    unsigned a = __shfl_up(inc.pts, off, kWave), b = __shfl_up(inc.occ, off, kWave), c = __shfl_up(inc.cand, off, kWave);
    if (lane >= off) { inc.pts += a; inc.occ += b; inc.cand += c; }
  }
  if (lane == kWave - 1) lds[wave] = inc;
  __syncthreads();
  U3 wave_off = {0, 0, 0};
  U3 tot = {0, 0, 0};
#pragma unroll
  for (int w = 0; w < kBlock / kWave; w++) {
    if (w < wave) wave_off = wave_off + lds[w];
    tot = tot + lds[w];
  }
  __syncthreads();
  total = tot;
  return {wave_off.pts + inc.pts - v.pts, wave_off.occ + inc.occ - v.occ, wave_off.cand + inc.cand - v.cand};
}

__global__ __launch_bounds__(kBlock) void k_scan_reduce(const unsigned* __restrict__ cell_count, long long n_cells,
                                                        unsigned min_pts, unsigned* __restrict__ block_sums) {
  const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
  U3 t = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const long long c = base + k;
    if (c < n_cells) {
      const unsigned v = cell_count[c];
      t.pts += v;
      t.occ += (v > 0);
      t.cand += (v >= min_pts);
    }
  }
  __shared__ U3 lds[kBlock / kWave];
  U3 total;
  block_exclusive_scan(t, total, lds);
  if (threadIdx.x == 0) {
    block_sums[blockIdx.x * 3 + 0] = total.pts;
    block_sums[blockIdx.x * 3 + 1] = total.occ;
    block_sums[blockIdx.x * 3 + 2] = total.cand;
  }
}

// single block: in-place exclusive scan of the per-tile sums; totals[3] out
__global__ __launch_bounds__(kBlock) void k_scan_blocks(unsigned* __restrict__ block_sums, int n_tiles,
                                                        unsigned* __restrict__ totals) {
  __shared__ U3 lds[kBlock / kWave];
  U3 carry = {0, 0, 0};
  for (int base = 0; base < n_tiles; base += kBlock) {
    const int i = base + threadIdx.x;
    U3 v = {0, 0, 0};
    if (i < n_tiles) v = {block_sums[i * 3 + 0], block_sums[i * 3 + 1], block_sums[i * 3 + 2]};
    U3 total;
    U3 ex = block_exclusive_scan(v, total, lds);
    if (i < n_tiles) {
      block_sums[i * 3 + 0] = carry.pts + ex.pts;
      block_sums[i * 3 + 1] = carry.occ + ex.occ;
      block_sums[i * 3 + 2] = carry.cand + ex.cand;
    }
    carry = carry + total;
  }
  if (threadIdx.x == 0) {
    totals[0] = carry.pts;
    totals[1] = carry.occ;
    totals[2] = carry.cand;
  }
}

__global__ __launch_bounds__(kBlock) void k_scan_apply(unsigned* __restrict__ cell_count /* -> cursor */,
                                                       long long n_cells, unsigned min_pts,
                                                       const unsigned* __restrict__ block_sums,
                                                       int* __restrict__ leaf_cell, unsigned* __restrict__ leaf_start,
                                                       int* __restrict__ leaf_count, int* __restrict__ leaf_rec) {
  const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
  unsigned cnt[kScanItems];
  U3 t = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const long long c = base + k;
    cnt[k] = (c < n_cells) ? cell_count[c] : 0u;
    t.pts += cnt[k];
    t.occ += (cnt[k] > 0);
    t.cand += (cnt[k] >= min_pts);
  }
  __shared__ U3 lds[kBlock / kWave];
  U3 total;
  U3 ex = block_exclusive_scan(t, total, lds);
  U3 run = {block_sums[blockIdx.x * 3 + 0] + ex.pts, block_sums[blockIdx.x * 3 + 1] + ex.occ,
            block_sums[blockIdx.x * 3 + 2] + ex.cand};
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const long long c = base + k;
    if (c < n_cells) {
      cell_count[c] = run.pts;  // scatter cursor
      if (cnt[k] > 0) {
        leaf_cell[run.occ] = static_cast<int>(c);
        leaf_start[run.occ] = run.pts;
        leaf_count[run.occ] = static_cast<int>(cnt[k]);
        leaf_rec[run.occ] = (cnt[k] >= min_pts) ? static_cast<int>(run.cand) : -1;
      }
      run.pts += cnt[k];
      run.occ += (cnt[k] > 0);
      run.cand += (cnt[k] >= min_pts);
    }
  }
}

// ---------------------------------------------------------------------------
// K1.d  counting-sort scatter of point indices into per-cell segments
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_scatter(const int* __restrict__ key, const unsigned* __restrict__ rank, int n,
                                                    const unsigned* __restrict__ cell_start, int* __restrict__ sorted_idx) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const int c = key[i];
    if (c >= 0) sorted_idx[cell_start[c] + rank[i]] = i;
  }
}

// ---------------------------------------------------------------------------
// K1.e  per-leaf finalize (second pass of applyFilter, _impl.hpp:282-367)
// ---------------------------------------------------------------------------
struct Sym3 {
  double xx, xy, xz, yy, yz, zz;
};

constexpr int kSortLimit = 64;      // up to here: insertion sort
constexpr int kSortGiveUp = 65536;  // beyond: left in arrival order (one thread would stall for too long)

// in-place ascending sort of a small index segment by ONE thread
__device__ __forceinline__ void sort_segment(int* seg, int cnt) {
  if (cnt <= kSortLimit) {
    for (int i = 1; i < cnt; i++) {
      const int v = seg[i];
      int j = i - 1;
      while (j >= 0 && seg[j] > v) { seg[j + 1] = seg[j]; j--; }
      seg[j + 1] = v;
    }
  } else if (cnt <= kSortGiveUp) {  // heap sort
    auto sift = [&](int root, int end) {
      for (;;) {
        int child = 2 * root + 1;
        if (child > end) break;
        if (child + 1 <= end && seg[child] < seg[child + 1]) child++;
        if (seg[root] < seg[child]) { const int t = seg[root]; seg[root] = seg[child]; seg[child] = t; root = child; }
        else break;
      }
    };
    for (int s0 = (cnt - 2) / 2; s0 >= 0; s0--) sift(s0, cnt - 1);
    for (int end = cnt - 1; end > 0; end--) {
      const int t = seg[0]; seg[0] = seg[end]; seg[end] = t;
      sift(0, end - 1);
    }
  }
}

// Leaves with more points than the register path of k_finalize takes (real scans: a 1 m voxel of a 0.1 m-filtered
// cloud holds hundreds): one WAVE per leaf restores ascending point order -- rank sort in LDS, every lane places its
// elements by counting the smaller ones -- and gathers the points into `big_pts` in that order, so that k_finalize's
// strictly sequential f64 sums read contiguous memory with many loads in flight instead of sorting the segment in
// global memory with one thread and chasing index -> point per addition (measured on the reference pair: 583 us per
// target build, almost all of it in that one-thread path).
constexpr int kPresortMin = 16;    // <= this many points: k_finalize's register path
constexpr int kPresortLds = 8192;  // segments up to here are sorted in LDS; longer ones by lane 0 (heap sort) as before
constexpr int kPresortRank = 256;  // up to here: rank sort (one element per thread, n comparisons each); above: bitonic network
__global__ __launch_bounds__(kBlock) void k_presort_large(const float4* __restrict__ pts, const unsigned* __restrict__ leaf_start,
                                                         const int* __restrict__ leaf_count, int n_leaves_host,
                                                         const unsigned* __restrict__ d_totals, int* __restrict__ sorted_idx,
                                                         float4* __restrict__ big_pts, int chunk) {
  __shared__ int s_idx[kPresortLds];
  __shared__ int s_cnt[kWave];
  const int n_leaves = d_totals ? static_cast<int>(d_totals[1]) : n_leaves_host;
  const int tid = threadIdx.x;
  // `chunk` (1..64, the launcher picks it so that the grid stays within 8192 blocks) leaves are looked at per step (one
  // load of their counts); the crowded ones among them are taken one after the other by the whole block -- few leaves:
  // about one crowded leaf per block; many leaves without crowded ones: a short pass over leaf_count
  for (int base = blockIdx.x * chunk; base < n_leaves; base += gridDim.x * chunk) {
    __syncthreads();  // s_cnt / s_idx of the previous step are done with
    const int mine = (tid < chunk && base + tid < n_leaves) ? leaf_count[base + tid] : 0;
    if (tid < chunk) s_cnt[tid] = mine;
    // (the barrier doubles as the vote: most steps of a cloud with ~1 point per cell -- a source scan being ordered -- hold
    // no crowded leaf at all, and walking their 64 counts one by one was 200 us per 6.4 M leaves)
    if (!__syncthreads_or(mine > kPresortMin)) continue;
    for (int pick = 0; pick < chunk; pick++) {
      const int cnt = s_cnt[pick];  // uniform across the block
      if (cnt <= kPresortMin) continue;
      const int leaf = base + pick;
      const unsigned start = leaf_start[leaf];
      int* seg = sorted_idx + start;
      if (cnt <= kPresortRank) {
        for (int i = tid; i < cnt; i += kBlock) s_idx[i] = seg[i];
        __syncthreads();
        for (int i = tid; i < cnt; i += kBlock) {
          const int v = s_idx[i];
          int rank = 0;
          for (int j = 0; j < cnt; j++) rank += (s_idx[j] < v) ? 1 : 0;  // point indices are unique
          seg[rank] = v;
          big_pts[start + rank] = pts[v];
        }
        __syncthreads();  // s_idx is reused by the next leaf
      } else if (cnt <= kPresortLds) {
        int m = 1;
        while (m < cnt) m <<= 1;  // padded with INT_MAX to a power of two
        for (int i = tid; i < m; i += kBlock) s_idx[i] = (i < cnt) ? seg[i] : 0x7fffffff;
        __syncthreads();
        for (int span = 2; span <= m; span <<= 1)
          for (int j = span >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m; i += kBlock) {
              const int partner = i ^ j;
              if (partner > i) {
                const int a = s_idx[i], b = s_idx[partner];
                const bool ascending = (i & span) == 0;
                if ((a > b) == ascending) {
                  s_idx[i] = b;
                  s_idx[partner] = a;
                }
              }
            }
            __syncthreads();
          }
        for (int i = tid; i < cnt; i += kBlock) {
          const int v = s_idx[i];
          seg[i] = v;
          big_pts[start + i] = pts[v];
        }
        __syncthreads();
      } else {
        if (tid == 0) sort_segment(seg, cnt);
        __threadfence();
        __syncthreads();
        for (int i = tid; i < cnt; i += kBlock) big_pts[start + i] = pts[__hip_atomic_load(seg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)];
      }
    }
  }
}

// Source ordering: after the counting sort by lattice cell, sort each cell's indices (stable,
// hence deterministic) and gather the points, so that consecutive lanes of K2 touch the same or
// adjacent target voxels (coalesced LUT probes and record gathers).
__global__ __launch_bounds__(kBlock) void k_sort_gather(const float4* __restrict__ pts, const unsigned* __restrict__ leaf_start,
                                                        const int* __restrict__ leaf_count, int n_leaves_host, const unsigned* __restrict__ d_totals,
                                                        int* __restrict__ sorted_idx, float4* __restrict__ out) {
  const int o = blockIdx.x * kBlock + threadIdx.x;
  const int n_leaves = d_totals ? static_cast<int>(d_totals[1]) : n_leaves_host;  // device-side count: no host round trip
  if (o >= n_leaves) return;
  const unsigned start = leaf_start[o];
  const int cnt = leaf_count[o];
  if (cnt > kPresortMin) return;  // crowded cells: k_presort_large has sorted and gathered them straight into `out`
  int* seg = sorted_idx + start;
  sort_segment(seg, cnt);
  for (int i = 0; i < cnt; i++) out[start + i] = pts[seg[i]];
}

// ---------------------------------------------------------------------------
// N1  centroid voxel down-sample -- [PCL] pcl::VoxelGrid<PointT>::applyFilter, the prefilter every
// caller runs before NDT (ndt_omp/apps/align.cpp:60-69, ndt_omp_mapping_node.cpp:142-148,203-210).
// Same count / scan / scatter machinery as K1; one thread per occupied voxel sums its points in
// f32 (CentroidPoint / AccumulatorXYZ) in ascending point order and divides by the count.  Leaves
// are enumerated in cell order, so the output is in ascending voxel-index order like PCL's.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_voxel_centroids(const float4* __restrict__ pts, const unsigned* __restrict__ leaf_start,
                                                            const int* __restrict__ leaf_count, int n_leaves_host, const unsigned* __restrict__ d_totals,
                                                            int* __restrict__ sorted_idx, float4* __restrict__ out,
                                                            const float4* __restrict__ big_pts) {
  const int o = blockIdx.x * kBlock + threadIdx.x;
  const int n_leaves = d_totals ? static_cast<int>(d_totals[1]) : n_leaves_host;  // device-side count: no host round trip
  if (o >= n_leaves) return;
  const unsigned start = leaf_start[o];
  const int cnt = leaf_count[o];
  int* seg = sorted_idx + start;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  int i = 0;
  if (big_pts && cnt > kPresortMin) {  // crowded voxel: sorted and laid out in order by k_presort_large
    const float4* bp = big_pts + start;
    for (; i + 8 <= cnt; i += 8) {
      const float4 p0 = bp[i], p1 = bp[i + 1], p2 = bp[i + 2], p3 = bp[i + 3], p4 = bp[i + 4], p5 = bp[i + 5], p6 = bp[i + 6], p7 = bp[i + 7];
      sx += p0.x; sy += p0.y; sz += p0.z;
      sx += p1.x; sy += p1.y; sz += p1.z;
      sx += p2.x; sy += p2.y; sz += p2.z;
      sx += p3.x; sy += p3.y; sz += p3.z;
      sx += p4.x; sy += p4.y; sz += p4.z;
      sx += p5.x; sy += p5.y; sz += p5.z;
      sx += p6.x; sy += p6.y; sz += p6.z;
      sx += p7.x; sy += p7.y; sz += p7.z;
    }
    for (; i < cnt; i++) {
      const float4 p = bp[i];
      sx += p.x; sy += p.y; sz += p.z;
    }
  } else {
    sort_segment(seg, cnt);
  }
  for (; i + 4 <= cnt; i += 4) {
    const float4 p0 = pts[seg[i]], p1 = pts[seg[i + 1]], p2 = pts[seg[i + 2]], p3 = pts[seg[i + 3]];
    sx += p0.x; sy += p0.y; sz += p0.z;
    sx += p1.x; sy += p1.y; sz += p1.z;
    sx += p2.x; sy += p2.y; sz += p2.z;
    sx += p3.x; sy += p3.y; sz += p3.z;
  }
  for (; i < cnt; i++) {
    const float4 p = pts[seg[i]];
    sx += p.x; sy += p.y; sz += p.z;
  }
  const float nf = static_cast<float>(cnt);
  out[o] = make_float4(sx / nf, sy / nf, sz / nf, 1.0f);
}

// First-pass sums of one voxel (applyFilter's first loop, _impl.hpp:209-263): mean_ += pt ; cov_ += pt*pt^T with cov_
// seeded Identity (.h:107); centroid.head<4>() += pt in f32 (:240-244).  Points must be added in ascending point
// order: the f64 sums then round exactly like the reference's sequential pass.
struct VoxelSums {
  double sx = 0, sy = 0, sz = 0;
  double cxx = 1, cxy = 0, cxz = 0, cyy = 1, cyz = 0, czz = 1;
  float fx = 0, fy = 0, fz = 0;
  __device__ __forceinline__ void add(float px, float py, float pz) {
#pragma clang fp contract(off)
    const double x = px, y = py, z = pz;
    sx += x; sy += y; sz += z;
    cxx += x * x; cxy += x * y; cxz += x * z; cyy += y * y; cyz += y * z; czz += z * z;
    fx += px; fy += py; fz += pz;
  }
};

// Second pass of applyFilter for one voxel (_impl.hpp:282-367): mean, covariance with the reference's quirks, 3x3
// eigen-solve, eigenvalue inflation, inverse, validity; writes the 64-B record, the centroid, the look-up table slot
// and (dump mode) the per-leaf outputs.  o: leaf ordinal, r: record ordinal (-1: fewer than min_pts points).
// Returns whether the voxel is valid for the DIRECT searches.
__device__ __forceinline__ bool finish_voxel(const VoxelSums& S, int cnt, int o, int r, int cell, int min_pts, double eig_ratio,
                                             VoxelRec* __restrict__ recs, VoxelSide* __restrict__ centroids, int* __restrict__ lut,
                                             const GridGeom& geom, const FinalizeDump& dump) {
  // No FMA contraction: the reference target (SSE4.2) never fuses, and its covariance formula (_impl.hpp:329-330)
  // cancels catastrophically when the coordinates are large against the voxel size, so a single fused multiply-add
  // shows up in the 7th digit of cov / icov.
#pragma clang fp contract(off)
  const double sx = S.sx, sy = S.sy, sz = S.sz;
  const double cxx = S.cxx, cxy = S.cxy, cxz = S.cxz, cyy = S.cyy, cyz = S.cyz, czz = S.czz;
  float fx = S.fx, fy = S.fy, fz = S.fz;
  const double n = cnt;
  const double ps[3] = {sx, sy, sz};
  const double mean[3] = {sx / n, sy / n, sz / n};  // :293
  fx /= static_cast<float>(cnt); fy /= static_cast<float>(cnt); fz /= static_cast<float>(cnt);  // :289

  double cov[3][3] = {{cxx, cxy, cxz}, {cxy, cyy, cyz}, {cxz, cyz, czz}};
  double icov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double evals[3] = {0, 0, 0};
  int nr_points = cnt;
  bool is_valid = false;

  if (cnt >= min_pts) {
    // :329-330
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) cov[i][j] = (cov[i][j] - 2 * (ps[i] * mean[j])) / n + mean[i] * mean[j];
    const double f = (n - 1.0) / n;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) cov[i][j] *= f;
    double w[3] = {0, 0, 0}, V[3][3];
    // The eigen-decomposition is needed only (a) to reject a voxel with a non-positive eigenvalue and (b) to inflate
    // the small eigenvalues of a flat or thin one; a voxel that is PROVABLY positive definite with
    // lambda_min >= eig_ratio lambda_max goes straight to the inverse of the untouched covariance -- bit for bit what
    // the full path computes for it.  Proof used: leading minors > 0 (Sylvester); lambda_max <= trace;
    // lambda_min = det / (lambda_mid lambda_max) >= det / (trace / 2)^2.  (Dump mode reports the eigenvalues: full path.)
    bool well_conditioned = false;
    if (!dump.nr_points) {
      const double m2 = cov[0][0] * cov[1][1] - cov[0][1] * cov[0][1];
      const double det = cov[0][0] * (cov[1][1] * cov[2][2] - cov[1][2] * cov[1][2]) - cov[0][1] * (cov[0][1] * cov[2][2] - cov[1][2] * cov[0][2]) +
                         cov[0][2] * (cov[0][1] * cov[1][2] - cov[1][1] * cov[0][2]);
      const double tr = cov[0][0] + cov[1][1] + cov[2][2];
      well_conditioned = cov[0][0] > 0 && m2 > 1e-12 * cov[0][0] * cov[1][1] && det > 0 && 4.0 * det > 1.05 * eig_ratio * tr * tr * tr && eig_ratio < 0.9;
    }
    if (well_conditioned) {
      w[0] = w[1] = w[2] = 1.0;  // (placeholders: positive, no inflation)
    } else {
      eig3_jacobi(cov, w, V);
    }
    if (w[0] < 0 || w[1] < 0 || w[2] <= 0) {  // :337-341
      nr_points = -1;
    } else {
      const double min_ev = eig_ratio * w[2];  // :345-356
      if (!well_conditioned && w[0] < min_ev) {
        w[0] = min_ev;
        if (w[1] < min_ev) w[1] = min_ev;
        double Vi[3][3], VL[3][3];
        inv3_cofactor(V, Vi);
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) VL[i][j] = V[i][j] * w[j];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) cov[i][j] = (VL[i][0] * Vi[0][j] + VL[i][1] * Vi[1][j]) + VL[i][2] * Vi[2][j];
      }
      evals[0] = w[0]; evals[1] = w[1]; evals[2] = w[2];
      inv3_cofactor(cov, icov);  // :359
      double mx = -DBL_MAX, mn = DBL_MAX;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { mx = fmax(mx, icov[i][j]); mn = fmin(mn, icov[i][j]); }
      if (mx == static_cast<double>(INFINITY) || mn == -static_cast<double>(INFINITY)) nr_points = -1;  // :360-364
    }
    {
      // Every voxel that reached min_points_per_voxel gets a record: the reference pushes its
      // centroid to the KD-tree BEFORE the eigenvalue / inverse checks (_impl.hpp:302-326 vs
      // :337-341,:360-364), so KDTREE search still returns a rejected voxel (trap 7), with the
      // icov_ it was left with (zero, or the inf-bearing inverse).  DIRECT searches skip it
      // (nr_points = -1): the LUT entry is lut_rejected(r).
      VoxelRec rec;
      rec.mean[0] = mean[0]; rec.mean[1] = mean[1]; rec.mean[2] = mean[2];
      const float c00 = static_cast<float>(icov[0][0]), c01 = static_cast<float>(icov[0][1]), c02 = static_cast<float>(icov[0][2]);
      const float c11 = static_cast<float>(icov[1][1]), c12 = static_cast<float>(icov[1][2]), c22 = static_cast<float>(icov[2][2]);
      rec.c[0] = c00; rec.c[1] = c01; rec.c[2] = c02;
      rec.c[3] = c12; rec.c[4] = c11; rec.c[5] = c22;
      rec.n = cnt;
      rec.pad = 0;
      rec.c01c11[0] = c01; rec.c01c11[1] = c11;
      recs[r] = rec;
      VoxelSide side;
      side.cx = fx; side.cy = fy; side.cz = fz; side.pad = 0.0f;
      side.icov[0] = icov[0][0]; side.icov[1] = icov[0][1]; side.icov[2] = icov[0][2];
      side.icov[3] = icov[1][1]; side.icov[4] = icov[1][2]; side.icov[5] = icov[2][2];
      centroids[r] = side;
      const int entry = (nr_points >= min_pts) ? r : lut_rejected(r);
      is_valid = nr_points >= min_pts;
      if (geom.hash_bits) {
        // sparse grid: claim a slot of the hash table (keys are unique: one insert per voxel)
        int2* tab = reinterpret_cast<int2*>(lut);
        const unsigned mask = (1u << geom.hash_bits) - 1u;
        for (unsigned hslot = hash_slot(cell, geom.hash_bits);; hslot = (hslot + 1u) & mask) {
          const int seen = atomicCAS(&tab[hslot].x, -1, cell);
          if (seen == -1 || seen == cell) {
            tab[hslot].y = entry;
            break;
          }
        }
      } else {
        // the cell's slot in the padded look-up table
        const int c = cell;
        const int cz = c / geom.mul[2], cy = (c - cz * geom.mul[2]) / geom.mul[1], cx = c - cz * geom.mul[2] - cy * geom.mul[1];
        const long long slot = static_cast<long long>(cx + kLutBorder) + static_cast<long long>(cy + kLutBorder) * geom.pmul[1] +
                               static_cast<long long>(cz + kLutBorder) * geom.pmul[2];
        lut[slot] = entry;
      }
    }
  }
  if (dump.nr_points) {
    dump.nr_points[o] = nr_points;
    for (int k = 0; k < 3; k++) {
      dump.mean[o * 3 + k] = mean[k];
      dump.evals[o * 3 + k] = evals[k];
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        dump.cov[o * 9 + i * 3 + j] = cov[i][j];
        dump.icov[o * 9 + i * 3 + j] = icov[i][j];
      }
  }
  return is_valid;
}

__global__ __launch_bounds__(kBlock) void k_finalize(const float4* __restrict__ pts, const int* __restrict__ leaf_cell,
                                                     const unsigned* __restrict__ leaf_start,
                                                     const int* __restrict__ leaf_count,
                                                     const int* __restrict__ leaf_rec, int n_leaves_host, const unsigned* __restrict__ d_totals,
                                                     int* __restrict__ sorted_idx, int min_pts, double eig_ratio,
                                                     VoxelRec* __restrict__ recs, VoxelSide* __restrict__ centroids, int* __restrict__ lut,
                                                     GridGeom geom, unsigned* __restrict__ n_valid, FinalizeDump dump,
                                                     const float4* __restrict__ big_pts, unsigned* __restrict__ /*unused*/) {
  // No FMA contraction anywhere in this kernel: the reference target (SSE4.2) never fuses, and its
  // covariance formula (_impl.hpp:329-330) cancels catastrophically when the coordinates are large
  // against the voxel size (sum of squares ~ n x^2 against a spread of millimetres), so a single fused
  // multiply-add in the sums shows up in the 7th digit of cov / icov.  With it off, the f64 sums and
  // the covariance are bit-identical to the reference's sequential pass.
#pragma clang fp contract(off)
  const int o = blockIdx.x * kBlock + threadIdx.x;
  const int n_leaves = d_totals ? static_cast<int>(d_totals[1]) : n_leaves_host;  // device-side count: no host round trip
  if (o >= n_leaves) return;
  const unsigned start = leaf_start[o];
  const int cnt = leaf_count[o];
  int* seg = sorted_idx + start;

  // The counting sort leaves the segment in arrival order; restore ascending point order so
  // the f64 sums below round exactly like the reference's sequential first pass
  // (_impl.hpp:209-263).
  // first-pass sums: mean_ += pt ; cov_ += pt*pt^T with cov_ seeded Identity (.h:107)
  VoxelSums S;
  auto add_point = [&](const float4& p) { S.add(p.x, p.y, p.z); };
  constexpr int kReg = 16;
  if (cnt <= kReg) {
    // typical voxel: indices in registers (all loads in flight at once), odd-even transposition
    // sort, then all point gathers in flight at once -- two memory latencies per voxel instead
    // of two per point
    int idx[kReg];
#pragma unroll
    for (int i = 0; i < kReg; i++) idx[i] = (i < cnt) ? seg[i] : 0x7fffffff;
#pragma unroll
    for (int pass = 0; pass < kReg; pass++) {
#pragma unroll
      for (int i = pass & 1; i + 1 < kReg; i += 2) {
        const int a = idx[i], b = idx[i + 1];
        idx[i] = min(a, b);
        idx[i + 1] = max(a, b);
      }
    }
    float4 pp[kReg];
#pragma unroll
    for (int i = 0; i < kReg; i++) pp[i] = pts[(i < cnt) ? idx[i] : idx[0]];
#pragma unroll
    for (int i = 0; i < kReg; i++) {
      if (i < cnt) {
        seg[i] = idx[i];  // keep the sorted order for the dump pass
        add_point(pp[i]);
      }
    }
  } else if (big_pts) {
    // k_presort_large has sorted the segment and laid its points out in order: eight contiguous loads in flight
    const float4* bp = big_pts + start;
    int i = 0;
    for (; i + 8 <= cnt; i += 8) {
      const float4 p0 = bp[i], p1 = bp[i + 1], p2 = bp[i + 2], p3 = bp[i + 3], p4 = bp[i + 4], p5 = bp[i + 5], p6 = bp[i + 6], p7 = bp[i + 7];
      add_point(p0); add_point(p1); add_point(p2); add_point(p3); add_point(p4); add_point(p5); add_point(p6); add_point(p7);
    }
    for (; i < cnt; i++) add_point(bp[i]);
  } else {
    sort_segment(seg, cnt);
    int i = 0;
    for (; i + 4 <= cnt; i += 4) {  // four gathers in flight
      const float4 p0 = pts[seg[i]], p1 = pts[seg[i + 1]], p2 = pts[seg[i + 2]], p3 = pts[seg[i + 3]];
      add_point(p0); add_point(p1); add_point(p2); add_point(p3);
    }
    for (; i < cnt; i++) add_point(pts[seg[i]]);
  }
  const bool is_valid = finish_voxel(S, cnt, o, leaf_rec[o], leaf_cell[o], min_pts, eig_ratio, recs, centroids, lut, geom, dump);
  {  // one counter update per wave instead of ~10^5 atomics on one word
    const unsigned long long vm = __ballot(is_valid);
    if (vm != 0 && (threadIdx.x & (kWave - 1)) == static_cast<unsigned>(__ffsll(static_cast<long long>(vm)) - 1))
      atomicAdd(n_valid, static_cast<unsigned>(__popcll(vm)));
  }
}

// ---------------------------------------------------------------------------
// K1, bucket form (the default for dense grids; the count / scan / scatter / finalize chain above stays as the general
// path and serves the prefilter and the scan ordering).
//
// Binning 1M points into 10^5 voxel counters with global atomics costs ~43 us whatever their scope -- integer atomics
// execute at the memory side on gfx950, ~23 G/s (tools/probes/atomic_probe.cpp) -- and the per-voxel pass then gathers
// its points at random from the whole cloud (64-B sectors for 16-B points).  Here the voxel index space is dealt out to
// K buckets of C cells each (short runs of consecutive cells, round-robin: k1_bucket below) and everything per-voxel is
// staged through LDS:
//   k1_hist     per block of points: LDS histogram over the buckets -> the block's row of the count matrix [blocks][K]
//   k1_colscan  the matrix column by column: exclusive prefix down the rows (a block's base inside every bucket), bucket sizes
//   k1_scatter  the same blocks move their points (x, y, z, point index) to their buckets ORDER-PRESERVING: block after
//               block, and inside a block by stable ranks (wave_rank) -- a bucket holds its points in ascending point index
//   k1_finalize one block per bucket: per-cell counts in LDS, then a STABLE placement of the points into their cells' LDS
//               segments (the same ranking) -- every cell's points in ascending point index, the order the reference adds
//               them in, with no sort (round 2 sorted every cell by point index: quadratic in a cell's points, half of the
//               kernel on crowded scenes) -- then one thread per cell (a lane team for crowded cells): sums, second pass
//               of applyFilter -> record, centroid, look-up table slot, sorted_idx
//   k1_count / k1_leaves  on demand: occupied / candidate counts, leaf arrays
// Points are read three times and written once, contiguously; no global atomic at all.
// ---------------------------------------------------------------------------
// Cell <-> (bucket, local cell).  Buckets are NOT ranges of the linear cell index: a clustered scene (a ground plane) would
// fill a few of those with many times the mean and leave the rest empty.  Runs of 2^rb consecutive cells (neighbours in x,
// whose points a spatially ordered cloud delivers together) are dealt round-robin to the K buckets:
//   run = cell >> rb,   bucket = run mod K,   local = (run / K) << rb | (cell mod 2^rb)
// K is any number (grid_build_plan picks a multiple of the blocks the chip holds at once, so that k1_finalize's one block
// per bucket runs in whole rounds); the division is a multiply-high by floor(2^32 / K) and one correction step.
struct K1Deal {
  int rb;
  unsigned K, M;
  __device__ __forceinline__ K1Deal(int rb_, int K_) : rb(rb_), K(static_cast<unsigned>(K_)), M(0xffffffffu / static_cast<unsigned>(K_)) {}
  __device__ __forceinline__ void divmod(unsigned run, unsigned& q, unsigned& r) const {
    q = __umulhi(run, M);  // floor(run / K) or one less
    r = run - q * K;
    if (r >= K) {
      r -= K;
      q++;
    }
  }
};
__device__ __forceinline__ int k1_bucket(int cell, const K1Deal& d) {
  unsigned q, r;
  d.divmod(static_cast<unsigned>(cell) >> d.rb, q, r);
  return static_cast<int>(r);
}
__device__ __forceinline__ int k1_local(int cell, const K1Deal& d) {
  unsigned q, r;
  d.divmod(static_cast<unsigned>(cell) >> d.rb, q, r);
  return static_cast<int>((q << d.rb) | (static_cast<unsigned>(cell) & ((1u << d.rb) - 1u)));
}
__device__ __forceinline__ int k1_cell(int bucket, int local, const K1Deal& d) {
  const unsigned l = static_cast<unsigned>(local);
  return static_cast<int>((((l >> d.rb) * d.K + static_cast<unsigned>(bucket)) << d.rb) | (l & ((1u << d.rb) - 1u)));
}
__device__ __forceinline__ int k1_key_bits(int K) {  // bits of the largest bucket number
  int b = 0;
  while ((1 << b) < K) b++;
  return b;
}

constexpr int kK1Threads = 256;              // k1_hist / k1_scatter
constexpr int kK1Waves = kK1Threads / kWave;  // 4
constexpr int kK1Round = 8 * kK1Threads;     // points one block of k1_scatter ranks per round (eight 64-point chunks per wave)

// Which slice of the cloud a block of k1_hist / k1_scatter takes.  Workgroups are dealt to the eight XCDs round-robin (block b
// runs on XCD b mod 8) and every XCD has an L2 of its own: with slice = block index, the 32-byte runs that neighbouring
// slices add to a bucket would come from eight different L2s and reach memory as eight partial lines.  XCD x takes the
// x-th eighth of the slices instead, so the runs one L2 collects for a bucket are adjacent and leave it as whole lines.
// (B, the grid size, is a multiple of 8.)
__device__ __forceinline__ int k1_slice(int b, int B) { return (b & 7) * (B >> 3) + (b >> 3); }

__device__ __forceinline__ int key_of(const GridGeom& g, const float4& p, int dense) {
  int c = -1;
  if (dense || finite3(p.x, p.y, p.z)) {
    c = build_cell(g, p.x, p.y, p.z);
    if (c < 0 || static_cast<long long>(c) >= g.n_cells) c = -1;  // inside the box by construction; NaN / garbage guard
  }
  return c;
}

// exclusive scan of n (<= 32 * nthreads) u32 values src[] -> dst[] by one block; dst[n] = total.  lds: nthreads / 64 words
__device__ __forceinline__ void block_scan_array(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int n, int nthreads,
                                                 unsigned* lds, bool agent_loads) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int per = (n + nthreads - 1) / nthreads;
  const int lo = tid * per, hi = min(n, lo + per);
  unsigned sum = 0;
  for (int i = lo; i < hi; i++) sum += agent_loads ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[i];
  unsigned inc = sum;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned a = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += a;
  }
  __syncthreads();
  if (lane == kWave - 1) lds[wave] = inc;
  __syncthreads();
  unsigned base = 0, total = 0;
  for (int w = 0; w < nthreads / kWave; w++) {
    if (w < wave) base += lds[w];
    total += lds[w];
  }
  unsigned run = base + inc - sum;
  for (int i = lo; i < hi; i++) {
    const unsigned v = agent_loads ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[i];
    dst[i] = run;
    run += v;
  }
  if (tid == 0) dst[n] = total;
}

// Stable rank of a lane's key among the lanes of its wave that hold the same key, plus the wave's running count of that
// key.  The lanes with equal keys find each other through LDS: every lane ORs its lane bit into the 64-bit mask of slot
// (key mod 256) of the wave's private mask table and reads the mask back -- one DS atomic and one DS read where a ballot
// per key bit (with its select) costs eight to thirteen times that; key bits above the eighth, if any, are settled with
// ballots.  The lowest lane of a slot clears it again, so the table is all zero between calls.  The lowest lane of a group
// of equal keys (the leader) bumps the wave's counter row by their number; everybody has read the old value in the same
// instruction.  Returns old count + number of equal-key lanes below this one: the position of the lane's element among
// ALL elements of that key the wave has ranked so far, in the order the wave met them.  `row`: this wave's counters (u16,
// one per key); `mtab`: this wave's kMatchSlots masks.  DS operations of one wave execute in order, so a later call sees
// this call's counter update and the cleared slot.
constexpr int kMatchSlots = 256;
__device__ __forceinline__ unsigned wave_rank(int key, bool valid, unsigned long long* mtab, unsigned short* row, int bits) {
  const int lane = threadIdx.x & (kWave - 1);
  unsigned long long* slot = mtab + (key & (kMatchSlots - 1));
  if (valid) (void)__hip_atomic_fetch_or(slot, 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const unsigned long long slotmask = valid ? __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
  const unsigned old = valid ? row[key] : 0u;
  unsigned long long peers = slotmask;
#pragma unroll
  for (int b = 8; b < 14; b++) {
    if (b < bits) {  // (uniform)
      const bool bit = ((key >> b) & 1) != 0;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
  }
  const unsigned first_of_slot = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(slotmask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(slotmask), 0u));
  if (valid && first_of_slot == 0) __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const unsigned below = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(peers >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(peers), 0u));
  if (valid && below == 0) row[key] = static_cast<unsigned short>(old + static_cast<unsigned>(__popcll(peers)));
  return old + below;
}

// rows = blocks of points, columns = buckets: the block's row of bucket counts (plain stores: round 2 claimed a run per
// (block, bucket) with a returning global atomic -- 250 k of them at 1 M points, 11 us of memory-side atomics, and runs in
// the order the blocks happened to arrive); the look-up table is cleared on the side (nothing reads it before k1_finalize)
__global__ __launch_bounds__(kK1Threads) void k1_hist(const float4* __restrict__ pts, int n, int dense, GridGeom g, int map, int K,
                                                      int ppb, unsigned* __restrict__ cntmat, int* __restrict__ lut, long long lut_cells,
                                                      const unsigned* __restrict__ n_dev) {
  extern __shared__ unsigned k1_lds[];
  if (n_dev) n = min(n, static_cast<int>(*n_dev));  // (a pass of launch_order_radix: the points an earlier pass kept)
  const K1Deal deal(map & 255, map >> 8);
  unsigned* h = k1_lds;
  const int slice = k1_slice(blockIdx.x, gridDim.x);
  const int lo = min(static_cast<long long>(n), static_cast<long long>(slice) * ppb), hi = min(static_cast<long long>(n), static_cast<long long>(lo) + ppb);
  float4 p[8];  // the first eight points per thread: requested before the clearing below
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int i = lo + threadIdx.x + u * kK1Threads;
    p[u] = (i < hi) ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
  }
  for (int k = threadIdx.x; k < K; k += kK1Threads) h[k] = 0;
  {  // the padded look-up table starts out empty: every block clears its slice
    const long long n4 = lut_cells / 4, per = (n4 + gridDim.x - 1) / gridDim.x;
    const long long lo4 = static_cast<long long>(blockIdx.x) * per, hi4 = min(n4, lo4 + per);
    int4* l4 = reinterpret_cast<int4*>(lut);
    for (long long i = lo4 + threadIdx.x; i < hi4; i += kK1Threads) l4[i] = make_int4(kLutEmpty, kLutEmpty, kLutEmpty, kLutEmpty);
    if (blockIdx.x == 0)
      for (long long i = n4 * 4 + threadIdx.x; i < lut_cells; i += kK1Threads) lut[i] = kLutEmpty;
  }
  __syncthreads();
  for (int base = lo + threadIdx.x; base < hi; base += 8 * kK1Threads) {  // eight loads in flight per thread
    if (base != lo + static_cast<int>(threadIdx.x)) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = base + u * kK1Threads;
        p[u] = (i < hi) ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int c = (base + u * kK1Threads < hi) ? key_of(g, p[u], dense) : -1;
      if (c >= 0) atomicAdd(&h[k1_bucket(c, deal)], 1u);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += kK1Threads) cntmat[static_cast<size_t>(slice) * K + k] = h[k];
}

// The count matrix, column by column: cntmat[b][k] <- points of bucket k in the blocks BEFORE b (exclusive prefix down the
// rows, in place), total[k] <- the bucket's size.  A block takes 16 columns; its 256 threads are 16 row groups x 16 columns,
// a thread keeps its rows (at most kColRows) in registers between the two sweeps.
constexpr int kColCols = 16, kColGroups = kK1Threads / kColCols, kColRows = 32;  // B <= kColGroups * kColRows = 512 rows
__global__ __launch_bounds__(kK1Threads) void k1_colscan(unsigned* __restrict__ cntmat, int B, int K, unsigned* __restrict__ total) {
  __shared__ unsigned s_part[kColGroups][kColCols + 1];
  const int c = threadIdx.x % kColCols, rg = threadIdx.x / kColCols;
  const int col = blockIdx.x * kColCols + c;
  const int R = (B + kColGroups - 1) / kColGroups;  // rows per group, <= kColRows
  const int r0 = rg * R;
  unsigned v[kColRows];
  unsigned sum = 0;
#pragma unroll
  for (int i = 0; i < kColRows; i++) {
    const int r = r0 + i;
    v[i] = (i < R && r < B && col < K) ? cntmat[static_cast<size_t>(r) * K + col] : 0u;
  }
#pragma unroll
  for (int i = 0; i < kColRows; i++) sum += v[i];
  s_part[rg][c] = sum;
  __syncthreads();
  unsigned base = 0, all = 0;
  for (int gq = 0; gq < kColGroups; gq++) {
    const unsigned t = s_part[gq][c];
    if (gq < rg) base += t;
    all += t;
  }
#pragma unroll
  for (int i = 0; i < kColRows; i++) {
    const int r = r0 + i;
    if (i < R && r < B && col < K) cntmat[static_cast<size_t>(r) * K + col] = base;
    base += v[i];
  }
  if (rg == 0 && col < K) total[col] = all;
}

// The same blocks move their points to their buckets, ORDER-PRESERVING: bucket k receives the points of block 0, block 1,
// ... (the column prefixes above), each block's in ascending point index (ranks inside a 64-point chunk from wave_rank, the
// waves' running counts in LDS) -- so every bucket, and after k1_finalize's stable placement every CELL, holds its points
// in ascending point index, the order the reference adds them in (_impl.hpp:233-244), and nothing has to be sorted.
// LDS (dynamic): cursor u32 [K + 1] | tot u16 [K] | tab u16 [kK1Waves][K]; static: the mask tables of wave_rank
__global__ __launch_bounds__(kK1Threads) void k1_scatter(const float4* __restrict__ pts, int n, int dense, GridGeom g, int map, int K,
                                                         int ppb, const unsigned* __restrict__ cntmat, const unsigned* __restrict__ total,
                                                         unsigned* __restrict__ bucket_base, float4* __restrict__ bpts,
                                                         unsigned* __restrict__ counts, unsigned long long* __restrict__ st, int index_form,
                                                         const unsigned* __restrict__ n_dev) {
  extern __shared__ unsigned k1_lds[];
  if (n_dev) n = min(n, static_cast<int>(*n_dev));
  const K1Deal deal(map & 255, map >> 8);
  __shared__ unsigned s_scan[kK1Waves];
  auto mark = [&](int q) {  // development aid (NDT_K1_STAMPS): thread 0's clock at the phase boundaries
    if (st && threadIdx.x == 0) st[8 * blockIdx.x + q] = stamp();
  };
  mark(0);
  __shared__ unsigned long long s_mtab[kK1Waves * kMatchSlots];  // (static: 8-byte aligned whatever precedes the dynamic part)
  unsigned long long* mtab_all = s_mtab;
  unsigned* cursor = k1_lds;                                                         // [K + 1] (+ 1 pad word)
  unsigned short* tot = reinterpret_cast<unsigned short*>(cursor + K + 2);           // [K]
  unsigned short* tab = tot + K;                                                     // [kK1Waves][K]
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int b = k1_slice(blockIdx.x, gridDim.x);  // the slice = the row of the count matrix
  const int kbits = k1_key_bits(K);
  const int lo = min(static_cast<long long>(n), static_cast<long long>(b) * ppb), hi = min(static_cast<long long>(n), static_cast<long long>(lo) + ppb);
  // the first round's points: requested before anything else, so that they arrive while the tables below are set up
  float4 p[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int i = lo + wave * (8 * kWave) + u * kWave + lane;
    p[u] = (i < hi) ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
  }
  for (int i = threadIdx.x; i < kK1Waves * kMatchSlots; i += kK1Threads) mtab_all[i] = 0ull;
  for (int i = threadIdx.x; i < kK1Waves * K / 2; i += kK1Threads) reinterpret_cast<unsigned*>(tab)[i] = 0u;
  // bucket bases = exclusive scan of the bucket sizes, by every block for itself; block 0 keeps them for k1_finalize
  block_scan_array(total, cursor, K, kK1Threads, s_scan, false);
  __syncthreads();
  if (b == 0) {
    for (int k = threadIdx.x; k <= K; k += kK1Threads) bucket_base[k] = cursor[k];
    if (threadIdx.x == 0) counts[0] = cursor[K];  // points binned
  }
  for (int k = threadIdx.x; k < K; k += kK1Threads) cursor[k] += cntmat[static_cast<size_t>(b) * K + k];
  __syncthreads();
  mark(1);
  // ---- stable split: rounds of 2048 points; wave w ranks the eight 64-point chunks [w * 512, (w + 1) * 512) of the round
  unsigned short* row = tab + wave * K;
  unsigned long long* mtab = mtab_all + wave * kMatchSlots;
  for (int r0 = lo; r0 < hi; r0 += kK1Round) {
    int key[8];
    unsigned rk[8];
    if (r0 != lo) {  // (uniform; the first round's points are on their way already)
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = r0 + wave * (8 * kWave) + u * kWave + lane;
        p[u] = (i < hi) ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = r0 + wave * (8 * kWave) + u * kWave + lane;
      const int c = (i < hi) ? key_of(g, p[u], dense) : -1;
      key[u] = (c >= 0) ? k1_bucket(c, deal) : -1;
      rk[u] = 0;
      if (r0 + wave * (8 * kWave) + u * kWave < hi) rk[u] = wave_rank(key[u], key[u] >= 0, mtab, row, kbits);  // (uniform: the chunk has points)
    }
    __syncthreads();
    if (r0 == lo) mark(2);
    // per bucket: the waves' counts -> exclusive prefix over the waves (in place), the round's total -> the cursor afterwards
    for (int k = threadIdx.x; k < K; k += kK1Threads) {
      unsigned s_ = 0;
#pragma unroll
      for (int w = 0; w < kK1Waves; w++) {
        const unsigned t = tab[w * K + k];
        tab[w * K + k] = static_cast<unsigned short>(s_);
        s_ += t;
      }
      tot[k] = static_cast<unsigned short>(s_);
    }
    __syncthreads();
    if (r0 == lo) mark(3);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (key[u] >= 0) {
        const int i = r0 + wave * (8 * kWave) + u * kWave + lane;
        const unsigned pos = cursor[key[u]] + row[key[u]] + rk[u];
        if (index_form) reinterpret_cast<int*>(bpts)[pos] = i;  // (uniform)
        else bpts[pos] = make_float4(p[u].x, p[u].y, p[u].z, __int_as_float(i));
      }
    }
    __syncthreads();
    if (r0 == lo) mark(4);
    if (r0 + kK1Round < hi) {  // (uniform) another round: advance the cursors, clear the counters
      for (int k = threadIdx.x; k < K; k += kK1Threads) cursor[k] += tot[k];
      for (int i = threadIdx.x; i < kK1Waves * K / 2; i += kK1Threads) reinterpret_cast<unsigned*>(tab)[i] = 0u;
      __syncthreads();
    }
  }
}

// per-bucket cell histogram in LDS (cnt[C], zeroed here)
// Where a bucket's points come from: its slice of the bucketed cloud in global memory (k1_scatter wrote it), or the lists a
// block of k1_small collected them in (LDS).  src(j) = point j of the bucket, j < nb, w = point index.
struct K1GlobalSrc {
  const float4* __restrict__ p;  // bpts + the bucket's base -- or, index form, the cloud itself
  const int* __restrict__ idx;   // index form (NDT_K1_INDEX=1): the bucket's slice of point indices; else null
  __device__ __forceinline__ float4 operator()(unsigned j) const {
    if (idx) {  // (uniform)
      const int i = idx[j];
      float4 q = p[i];
      q.w = __int_as_float(i);
      return q;
    }
    return p[j];
  }
};
// the bucket's source from a kernel's (bpts, cloud) pair: cloud non-null = bpts holds 4-byte point indices
__device__ __forceinline__ K1GlobalSrc k1_src(const float4* bpts, const float4* cloud, unsigned bb) {
  return cloud ? K1GlobalSrc{cloud, reinterpret_cast<const int*>(bpts) + bb} : K1GlobalSrc{bpts + bb, nullptr};
}
template <class Src>
__device__ __forceinline__ void k1_cell_histogram(const Src& src, unsigned nb, const GridGeom& g,
                                                  const K1Deal& deal, int C, unsigned* cnt) {
  for (int c = threadIdx.x; c < C; c += kBlock) cnt[c] = 0;
  __syncthreads();
  for (unsigned j = threadIdx.x; j < nb; j += kBlock) {
    const float4 p = src(j);
    atomicAdd(&cnt[k1_local(build_cell(g, p.x, p.y, p.z), deal)], 1u);
  }
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k1_count(const float4* __restrict__ bpts, GridGeom g, int map, int K, int C, unsigned min_pts,
                                                   const unsigned* __restrict__ bucket_base, unsigned* __restrict__ tot,
                                                   unsigned* __restrict__ ticket, unsigned* __restrict__ occ_base,
                                                   unsigned* __restrict__ cand_base, unsigned* __restrict__ counts, const float4* __restrict__ cloud) {
  extern __shared__ unsigned k1_lds[];
  const K1Deal deal(map & 255, map >> 8);
  __shared__ U3 s_u3[kBlock / kWave];
  __shared__ unsigned s_scan[kBlock / kWave];
  __shared__ int s_last;
  const int k = blockIdx.x;
  const unsigned bb = bucket_base[k], be = bucket_base[k + 1];
  U3 t = {0, 0, 0};
  if (be > bb) {  // uniform
    k1_cell_histogram(k1_src(bpts, cloud, bb), be - bb, g, deal, C, k1_lds);
    for (int c = threadIdx.x; c < C; c += kBlock) {
      const unsigned v = k1_lds[c];
      t.occ += (v > 0);
      t.cand += (v >= min_pts);
    }
  }
  U3 total;
  block_exclusive_scan(t, total, s_u3);
  if (threadIdx.x == 0) {
    __hip_atomic_store(tot + 2 * k, total.occ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(tot + 2 * k + 1, total.cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // exclusive scans of the per-bucket (occupied, candidate) counts, interleaved in tot[]
  {
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int per = (K + kBlock - 1) / kBlock, lo = tid * per, hi = min(K, lo + per);
    for (int which = 0; which < 2; which++) {
      unsigned* dst = which ? cand_base : occ_base;
      unsigned sum = 0;
      for (int i = lo; i < hi; i++) sum += __hip_atomic_load(tot + 2 * i + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned inc = sum;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const unsigned a = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += a;
      }
      __syncthreads();
      if (lane == kWave - 1) s_scan[wave] = inc;
      __syncthreads();
      unsigned base = 0, all = 0;
      for (int w = 0; w < kBlock / kWave; w++) {
        if (w < wave) base += s_scan[w];
        all += s_scan[w];
      }
      unsigned run = base + inc - sum;
      for (int i = lo; i < hi; i++) {
        dst[i] = run;
        run += __hip_atomic_load(tot + 2 * i + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid == 0) {
        dst[K] = all;
        counts[1 + which] = all;  // [1] occupied voxels, [2] candidates (>= min_pts)
      }
    }
    // [3] valid voxels: the words k1_finalize left behind the bucket bases
    const unsigned* bucket_valid = bucket_base + K + 1;
    unsigned v = 0;
    for (int i = lo; i < hi; i++) v += bucket_valid[i];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    __syncthreads();
    if (lane == 0) s_scan[wave] = v;
    __syncthreads();
    if (tid == 0) {
      unsigned t = 0;
      for (int w = 0; w < kBlock / kWave; w++) t += s_scan[w];
      counts[3] = t;
    }
  }
}

// exclusive scan of cnt[0..C) into cend[0..C) by the block (C a power of two >= 32)
__device__ __forceinline__ void k1_scan_cells(const unsigned* cnt, unsigned* cend, int C, U3* s_u3) {
  const int per = C / kBlock > 0 ? C / kBlock : 1;
  const int lo = threadIdx.x * per;
  U3 t = {0, 0, 0};
  for (int c = lo; c < lo + per && c < C; c++) t.pts += cnt[c];
  U3 total;
  U3 run = block_exclusive_scan(t, total, s_u3);
  for (int c = lo; c < lo + per && c < C; c++) {
    cend[c] = run.pts;
    run.pts += cnt[c];
  }
  __syncthreads();
}

constexpr int kK1PerThread = 8;                   // points per thread of one LDS pass
constexpr int kK1LdsCap = 2 * kK1PerThread * kBlock;  // 4096: points one LDS pass can hold (two rounds of the stable placement)
// Cells with more points than this are summed by a TEAM of 16 lanes, one accumulator per lane: the nine f64 sums and the
// three f32 centroid sums of a voxel are twelve independent chains of strictly ordered additions (the reference's order,
// _impl.hpp:233-244) -- one thread walking a 400-point voxel of a real scan issues 15 f64 instructions per point by
// itself (~20 us per voxel), a lane per chain issues two.
constexpr int kTeamCell = 32, kTeamLanes = 16;
constexpr size_t kK1MaxDynamicLds = 136 * 1024;  // of the CU's 160 KB (the kernels also hold up to ~20 KB of static LDS)
constexpr int kMaxTeamCells = kK1LdsCap / (kTeamCell + 1) + 1;  // team cells one LDS pass can hold

// What happens with a cell's sums once its points have been added up in order: the NDT grid turns them into a voxel record
// (finish_voxel); the voxel FILTER (vf_finalize) emits the centroid.  fin(S, points, slot, cell) -> counted as "valid".
struct K1RecordFin {
  int min_pts;
  double eig_ratio;
  VoxelRec* __restrict__ recs;
  VoxelSide* __restrict__ centroids;
  int* __restrict__ lut;
  const GridGeom* g;
  __device__ __forceinline__ bool operator()(const VoxelSums& S, int n_c, int r, int cell) const {
    const FinalizeDump nodump{nullptr, nullptr, nullptr, nullptr, nullptr};
    return finish_voxel(S, n_c, 0, r, cell, min_pts, eig_ratio, recs, centroids, lut, *g, nodump);
  }
};

// One bucket, finished by one block: bucket k holds nb points (src), the first of them is point bb of the bucket order.
// k1_lds: the dynamic LDS described at the launch (3 C words of per-cell state, 3 words per point of a pass, 5 rows of wmax
// u16 counters).
template <class Src, class Fin>
__device__ __forceinline__ void k1_finalize_bucket(const Src& src, const int k, const unsigned bb, const unsigned nb, unsigned* k1_lds,
                                                   const GridGeom& g, const K1Deal& deal, int C, int min_pts,
                                                   int lds_cap, int wmax /* cells one pass may span (power of two <= C) */,
                                                   int* __restrict__ sorted_idx /* or null */, const Fin& fin,
                                                   unsigned* __restrict__ bucket_valid /* [K]: valid voxels of every bucket */,
                                                   unsigned* __restrict__ scratch /* 5 x n words */, unsigned n_total,
                                                   unsigned long long* __restrict__ st /* development aid: 8 words per bucket, or null */) {
  __shared__ U3 s_u3[kBlock / kWave];
  __shared__ int s_hi;
  // phase clocks of thread 0 (NDT_K1_STAMPS): cycles spent in 0 loads + ranks (or the histogram pass), 1 column / cell scans
  // and the passes' selection, 2 placement, 3 lane teams, 4 sums + finish_voxel; 5 = passes, 6 = points, 7 = total
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, t_mark = 0, t_begin = 0;
  unsigned n_passes = 0;
  if (st && threadIdx.x == 0) t_mark = t_begin = stamp();
  auto lap = [&](int phase) {
    if (st && threadIdx.x == 0) {
      const unsigned long long t = stamp();
      ph[phase] += t - t_mark;
      t_mark = t;
    }
  };
  __shared__ int s_ncand;                     // cells of the current pass that get a record
  __shared__ int s_nteam;                     // team cells of the current pass ...
  __shared__ int s_team_cell[kMaxTeamCells];  // ... their cells.  A team leaves its sums where the cell's points were: the nine f64
                                              // (sx sy sz cxx cxy cxz cyy cyz czz) as 18 words at the head of the cell's x segment,
                                              // fx fy fz at the head of its y segment (a team cell has more than 32 points)
  __shared__ float s_one;
  if (nb == 0) {  // empty bucket (uniform)
    if (threadIdx.x == 0) bucket_valid[k] = 0u;
    return;
  }
  if (threadIdx.x == 0) s_one = 1.0f;
  unsigned* cnt = k1_lds;          // [C] points per cell
  unsigned* cstart = k1_lds + C;   // [C] start of the cell's segment inside the bucket (exclusive prefix of cnt)
  unsigned* cur = k1_lds + 2 * C;  // [C] of the current pass: points already placed in the cell's segment
  // the points of the current pass in (cell, point index) order
  float* ox = reinterpret_cast<float*>(k1_lds + 3 * C);
  float* oy = ox + lds_cap;
  float* oz = oy + lds_cap;
  __shared__ unsigned long long s_mtab[(kBlock / kWave) * kMatchSlots];  // wave_rank's mask tables (static: 8-byte aligned whatever precedes the dynamic part)
  unsigned long long* mtab_all = s_mtab;
  unsigned short* tab = reinterpret_cast<unsigned short*>(oz + lds_cap);  // [4 waves + totals][wmax], indexed by cell - c_lo
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  int wbits = 0;
  while ((1 << wbits) < wmax) wbits++;
  for (int i = threadIdx.x; i < (kBlock / kWave) * kMatchSlots; i += kBlock) mtab_all[i] = 0ull;
  for (int i = threadIdx.x; i < (kBlock / kWave + 1) * wmax / 2; i += kBlock) reinterpret_cast<unsigned*>(tab)[i] = 0u;
  // A bucket that fits one round of the stable placement (2048 points; every bucket of a uniform cloud) reads its points
  // ONCE: the ranking's per-wave counts ARE the per-cell histogram, so the separate counting pass over the bucket and
  // its LDS atomics are skipped -- rank, column scan (-> counts), scan over the cells (-> segment starts), place.
  const bool one_shot = nb <= static_cast<unsigned>(kK1PerThread * kBlock) && nb <= static_cast<unsigned>(lds_cap) && C <= wmax;
  if (one_shot) {
    unsigned short* row = tab + wave * wmax;
    unsigned long long* mtab = mtab_all + wave * kMatchSlots;
    const int per_wave = static_cast<int>(((nb + kWave - 1) / kWave + kBlock / kWave - 1) / (kBlock / kWave));  // <= kK1PerThread
    float4 p[kK1PerThread];
    int cc[kK1PerThread];
    unsigned rk[kK1PerThread];
#pragma unroll
    for (int u = 0; u < kK1PerThread; u++) {
      const unsigned j = static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
      p[u] = (u < per_wave && j < nb) ? src(j) : make_float4(NAN, NAN, NAN, 0.f);
    }
    __syncthreads();  // (the counter rows are cleared)
#pragma unroll
    for (int u = 0; u < kK1PerThread; u++) {
      const unsigned j = static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
      const int c = (u < per_wave && j < nb) ? k1_local(build_cell(g, p[u].x, p[u].y, p[u].z), deal) : -1;
      cc[u] = c;
      rk[u] = 0;
      if (u < per_wave) rk[u] = wave_rank(c, c >= 0, mtab, row, wbits);  // (uniform)
    }
    __syncthreads();
    lap(0);
    for (int c = threadIdx.x; c < C; c += kBlock) {
      unsigned s_ = 0;
#pragma unroll
      for (int w = 0; w < kBlock / kWave; w++) {
        const unsigned t = tab[w * wmax + c];
        tab[w * wmax + c] = static_cast<unsigned short>(s_);
        s_ += t;
      }
      cnt[c] = s_;
    }
    __syncthreads();
    k1_scan_cells(cnt, cstart, C, s_u3);
    lap(1);
#pragma unroll
    for (int u = 0; u < kK1PerThread; u++) {
      if (cc[u] >= 0) {
        const unsigned q = cstart[cc[u]] + row[cc[u]] + rk[u];
        ox[q] = p[u].x;
        oy[q] = p[u].y;
        oz[q] = p[u].z;
        if (sorted_idx) sorted_idx[bb + q] = __float_as_int(p[u].w);
      }
    }
    __syncthreads();
    lap(2);
  } else {
    k1_cell_histogram(src, nb, g, deal, C, cnt);
    k1_scan_cells(cnt, cstart, C, s_u3);
    lap(0);
  }
  unsigned n_ok = 0;
  // The bucket is finished in passes over runs of cells [c_lo, c_hi) that hold at most lds_cap points -- one pass for a
  // bucket of a uniform cloud, several for a crowded one (clustered data: a ground plane fills "its" buckets with many
  // times the mean).  A pass selects its points from the bucket, sorts them by (cell, point index) in LDS and finishes
  // its cells.
  for (int c_lo = 0; c_lo < C;) {
    if (threadIdx.x == 0) {
      int c = c_lo;
      if (one_shot || (nb <= static_cast<unsigned>(lds_cap) && C <= wmax)) {
        c = C;
      } else {
        // the last cell whose END stays within lds_cap points of c_lo's start (and within wmax cells: the counter rows of
        // the stable placement are that wide): binary search in the prefix sums
        // (cstart[c] = points before cell c; a linear walk by one thread cost 30 us per pass at C = 4096)
        const unsigned limit = cstart[c_lo] + static_cast<unsigned>(lds_cap);
        int lo = c_lo, hi = min(C, c_lo + wmax);  // invariant: cells [c_lo, lo) fit; answer in [lo, hi]
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;  // candidate: cells [c_lo, mid) -- they end at cstart[mid] (mid < C) or nb
          const unsigned end = (mid < C) ? cstart[mid] : nb;
          if (end <= limit) lo = mid; else hi = mid - 1;
        }
        c = lo;
        if (c == c_lo) c = c_lo + 1;  // a single cell with more points than a pass holds: the crowded-cell path below
      }
      s_hi = c;
    }
    __syncthreads();
    const int c_hi = s_hi;
    const unsigned base = cstart[c_lo];
    const unsigned n_pass = ((c_hi < C) ? cstart[c_hi] : nb) - base;
    if (n_pass == 0) {  // (uniform)
      c_lo = c_hi;
      __syncthreads();
      continue;
    }
    const bool giant = n_pass > static_cast<unsigned>(lds_cap);  // then c_hi == c_lo + 1
    // arrays of this pass: LDS, or (one cell too crowded for LDS) the bucket's slices of the global scratch
    float* px = giant ? reinterpret_cast<float*>(scratch + n_total + bb + base) : ox;
    float* py = giant ? reinterpret_cast<float*>(scratch + 2 * static_cast<size_t>(n_total) + bb + base) : oy;
    float* pz = giant ? reinterpret_cast<float*>(scratch + 3 * static_cast<size_t>(n_total) + bb + base) : oz;
    if (!one_shot) {
    for (int c = c_lo + threadIdx.x; c < c_hi; c += kBlock) cur[c] = cstart[c] - base;
    __syncthreads();
    // Select this pass's points, ORDER-PRESERVING.  The bucket holds its points in ascending point index (k1_scatter), and
    // a point's slot inside its cell's segment is its stable rank: points of the cell placed by earlier rounds (cur) +
    // points of the cell in the waves before mine this round (tab, after the column scan) + rank inside my wave
    // (wave_rank).  Every cell's segment therefore ends up in ascending point index -- the order the reference adds the
    // points in -- without the sort by index that round 2 spent a quarter (uniform cloud) to half (crowded cells: it is
    // quadratic in a cell's points) of this kernel on.  Rounds of 2048 points; wave w takes the eight 64-point chunks
    // [w * 512, (w + 1) * 512) of a round.
    {
      unsigned short* row = tab + wave * wmax;
      unsigned long long* mtab = mtab_all + wave * kMatchSlots;
      for (unsigned j0 = 0; j0 < nb; j0 += kK1PerThread * kBlock) {
        // the round's points (at most 2048) in four equal, contiguous shares of whole chunks: wave w ranks chunks
        // [w * per_wave, (w + 1) * per_wave) -- a bucket of 1000 points keeps all four waves busy, not the first two
        const unsigned round_n = min(static_cast<unsigned>(kK1PerThread * kBlock), nb - j0);
        const int per_wave = static_cast<int>(((round_n + kWave - 1) / kWave + kBlock / kWave - 1) / (kBlock / kWave));  // <= kK1PerThread
        float4 p[kK1PerThread];
        int cc[kK1PerThread];
        unsigned rk[kK1PerThread];
#pragma unroll
        for (int u = 0; u < kK1PerThread; u++) {
          const unsigned j = j0 + static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
          p[u] = (u < per_wave && j < nb) ? src(j) : make_float4(NAN, NAN, NAN, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kK1PerThread; u++) {
          const unsigned j = j0 + static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
          int c = -1;
          if (u < per_wave && j < nb) {
            c = k1_local(build_cell(g, p[u].x, p[u].y, p[u].z), deal);
            if (c < c_lo || c >= c_hi) c = -1;
          }
          cc[u] = c;
          rk[u] = 0;
          if (u < per_wave) rk[u] = wave_rank(c - c_lo, c >= 0, mtab, row, wbits);  // (uniform)
        }
        __syncthreads();
        for (int c = threadIdx.x; c < c_hi - c_lo; c += kBlock) {  // the waves' counts of a cell -> exclusive prefix over the waves, total
          unsigned s_ = 0;
#pragma unroll
          for (int w = 0; w < kBlock / kWave; w++) {
            const unsigned t = tab[w * wmax + c];
            tab[w * wmax + c] = static_cast<unsigned short>(s_);
            s_ += t;
          }
          tab[(kBlock / kWave) * wmax + c] = static_cast<unsigned short>(s_);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kK1PerThread; u++) {
          if (cc[u] >= 0) {
            const unsigned q = cur[cc[u]] + row[cc[u] - c_lo] + rk[u];
            px[q] = p[u].x;
            py[q] = p[u].y;
            pz[q] = p[u].z;
            if (sorted_idx) sorted_idx[bb + base + q] = __float_as_int(p[u].w);
          }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < c_hi - c_lo; c += kBlock) {  // next round: behind what this one placed; counters cleared
          cur[c_lo + c] += tab[(kBlock / kWave) * wmax + c];
#pragma unroll
          for (int w = 0; w < kBlock / kWave; w++) tab[w * wmax + c] = 0;
        }
        __syncthreads();
      }
    }
    }  // !one_shot
    n_passes++;
    lap(1);
    if (giant) {
      // one cell with more points than LDS holds (thousands per voxel): its points lie in order in the global scratch; one
      // thread adds them up.  Slow, and rare.
      __threadfence_block();
      __syncthreads();
      if (threadIdx.x == 0 && static_cast<int>(n_pass) >= min_pts) {
        VoxelSums S;
        unsigned i = 0;
        for (; i + 8 <= n_pass; i += 8) {
          float x[8], y[8], z[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { x[u] = px[i + u]; y[u] = py[i + u]; z[u] = pz[i + u]; }
#pragma unroll
          for (int u = 0; u < 8; u++) S.add(x[u], y[u], z[u]);
        }
        for (; i < n_pass; i++) S.add(px[i], py[i], pz[i]);
        const int r = static_cast<int>((bb + base) / static_cast<unsigned>(min_pts));
        n_ok += fin(S, static_cast<int>(n_pass), r, k1_cell(k, c_lo, deal)) ? 1u : 0u;
      }
      c_lo = c_hi;
      __syncthreads();
      continue;
    }
    lap(2);
    // ---- crowded cells of this pass: a team of 16 lanes per cell, a lane per accumulator (see kTeamCell) ----
    if (threadIdx.x == 0) s_nteam = 0;
    __syncthreads();
    for (int c = c_lo + threadIdx.x; c < c_hi; c += kBlock) {
      const int n_c = static_cast<int>(cnt[c]);
      if (n_c > kTeamCell && n_c >= min_pts) {
        const int slot = atomicAdd(&s_nteam, 1);
        s_team_cell[slot] = c;
      }
    }
    __syncthreads();
    if (s_nteam > 0) {
#pragma clang fp contract(off)
      const int tl = threadIdx.x & (kTeamLanes - 1), team = threadIdx.x / kTeamLanes;
      // lane -> (a, b): 0-2 mean sums a * 1; 3-8 the products xx xy xz yy yz zz; 9-11 the f32 centroid sums
      const int ia = (tl < 3) ? tl : (tl < 6) ? 0 : (tl < 8) ? 1 : (tl == 8) ? 2 : (tl < 12) ? tl - 9 : 0;
      const int ib = (tl == 3) ? 0 : (tl == 4 || tl == 6) ? 1 : (tl == 5 || tl == 7 || tl == 8) ? 2 : -1;
      const float* pa = (ia == 0) ? ox : (ia == 1) ? oy : oz;
      const float* pb = (ib == 0) ? ox : (ib == 1) ? oy : (ib == 2) ? oz : &s_one;
      const unsigned sb = (ib < 0) ? 0u : 1u;  // mean / centroid lanes multiply by the constant 1.0f (exact)
      const int n_team = s_nteam;
      for (int s = team; s < n_team; s += kBlock / kTeamLanes) {
        const int c = s_team_cell[s];
        const unsigned beg = cstart[c] - base, n_c = cnt[c];
        double acc = (tl == 3 || tl == 6 || tl == 8) ? 1.0 : 0.0;  // cov_ starts as Identity (voxel_grid_covariance_omp.h:107)
        float acc32 = 0.f;
        unsigned i = 0;
        for (; i + 4 <= n_c; i += 4) {
          float a[4], b[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { a[u] = pa[beg + i + u]; b[u] = pb[(beg + i + u) * sb]; }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const double prod = static_cast<double>(a[u]) * static_cast<double>(b[u]);
            acc += prod;
            acc32 += a[u];
          }
        }
        for (; i < n_c; i++) {
          const float a = pa[beg + i], b = pb[(beg + i) * sb];
          const double prod = static_cast<double>(a) * static_cast<double>(b);
          acc += prod;
          acc32 += a;
        }
        // (the 16 lanes of a team are lanes of one wave and walk the same n_c points: every read of the segment above has
        // been issued before these stores, and a wave's DS operations execute in order)
        if (tl < 9) {
          ox[beg + 2 * tl] = __int_as_float(__double2loint(acc));
          ox[beg + 2 * tl + 1] = __int_as_float(__double2hiint(acc));
        } else if (tl < 12) {
          oy[beg + tl - 9] = acc32;
        }
      }
    }
    __syncthreads();
    lap(3);
    // The cells that get a record (min_pts points and more: the reference skips the others at look-up, _impl.hpp:395), compacted
    // into a list first (cur[] is free after the placement): on a sparsely occupied grid -- 4096 cells per bucket for ~200
    // occupied ones at 10 M points / 0.5 m -- a thread per CELL left one lane in twenty with work and every wave ran
    // finish_voxel sixteen times over.  (The order of the list varies from run to run; nothing depends on it.)
    if (threadIdx.x == 0) s_ncand = 0;
    __syncthreads();
    for (int c = c_lo + threadIdx.x; c < c_hi; c += kBlock)
      if (static_cast<int>(cnt[c]) >= min_pts) cur[atomicAdd(&s_ncand, 1)] = static_cast<unsigned>(c);
    __syncthreads();
    const int n_cand = s_ncand;
    for (int ci = threadIdx.x; ci < n_cand; ci += kBlock) {
      const int c = static_cast<int>(cur[ci]);
      const int n_c = static_cast<int>(cnt[c]);
      const unsigned beg = cstart[c] - base;
      // record slot: candidates' segments start at least min_pts apart, so start / min_pts is unique per candidate --
      // no scan over the buckets is needed to number the records (k1_leaves numbers the LEAVES when somebody asks)
      const int r = static_cast<int>((bb + cstart[c]) / static_cast<unsigned>(min_pts));
      VoxelSums S;
      if (n_c > kTeamCell) {
        auto f64_at = [&](int q) { return __hiloint2double(__float_as_int(ox[beg + 2 * q + 1]), __float_as_int(ox[beg + 2 * q])); };
        S.sx = f64_at(0); S.sy = f64_at(1); S.sz = f64_at(2);
        S.cxx = f64_at(3); S.cxy = f64_at(4); S.cxz = f64_at(5);
        S.cyy = f64_at(6); S.cyz = f64_at(7); S.czz = f64_at(8);
        S.fx = oy[beg]; S.fy = oy[beg + 1]; S.fz = oy[beg + 2];
      } else {
        int i = 0;
        for (; i + 4 <= n_c; i += 4) {  // ascending point order, contiguous; twelve reads in flight per step
          float x[4], y[4], z[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { x[u] = ox[beg + i + u]; y[u] = oy[beg + i + u]; z[u] = oz[beg + i + u]; }
#pragma unroll
          for (int u = 0; u < 4; u++) S.add(x[u], y[u], z[u]);
        }
        for (; i < n_c; i++) S.add(ox[beg + i], oy[beg + i], oz[beg + i]);
      }
      n_ok += fin(S, n_c, r, k1_cell(k, c, deal)) ? 1u : 0u;
    }
    c_lo = c_hi;
    __syncthreads();  // the LDS arrays are reused by the next pass
    lap(4);
  }
  if (st && threadIdx.x == 0) {
    for (int q = 0; q < 5; q++) st[8 * k + q] = ph[q];
    st[8 * k + 5] = n_passes;
    st[8 * k + 6] = nb;
    st[8 * k + 7] = stamp() - t_begin;
  }
  {  // the bucket's valid voxels -> its own word; k1_count adds the words up when somebody asks (grid_counts).  One atomic
     // per WAVE on a single counter was 8 of this kernel's 41 us at 1 M points: same-address atomics serialise at the
     // memory side, ~15 ns each
    unsigned v = n_ok;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    __syncthreads();  // (s_u3 is free)
    if ((threadIdx.x & (kWave - 1)) == 0) s_u3[threadIdx.x / kWave].pts = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int w = 0; w < kBlock / kWave; w++) t += s_u3[w].pts;
      bucket_valid[k] = t;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k1_finalize(const float4* __restrict__ bpts, GridGeom g, int map, int K, int C, int min_pts,
                                                      double eig_ratio, int lds_cap, int wmax,
                                                      const unsigned* __restrict__ bucket_base, int* __restrict__ sorted_idx,
                                                      VoxelRec* __restrict__ recs, VoxelSide* __restrict__ centroids, int* __restrict__ lut,
                                                      unsigned* __restrict__ bucket_valid, unsigned* __restrict__ scratch, unsigned n_total,
                                                      unsigned long long* __restrict__ st, const float4* __restrict__ cloud) {
  extern __shared__ unsigned k1_lds[];
  const K1Deal deal(map & 255, map >> 8);
  const int k = blockIdx.x;
  const unsigned bb = bucket_base[k], be = bucket_base[k + 1];
  k1_finalize_bucket(k1_src(bpts, cloud, bb), k, bb, be - bb, k1_lds, g, deal, C, min_pts, lds_cap, wmax, sorted_idx,
                     K1RecordFin{min_pts, eig_ratio, recs, centroids, lut, &g}, bucket_valid, scratch, n_total, st);
}

// ---------------------------------------------------------------------------
// N1 / N2 on the bucket front end: pcl::VoxelGrid's centroid filter for dense grids.  k1_hist / k1_colscan / k1_scatter as
// for the NDT grid (order-preserving: every bucket, and after the stable placement every cell, holds its points in
// ascending point index -- the order PCL adds them in), then
//   vf_finalize      one block per bucket (k1_finalize_bucket with a finish of its own): a cell's f32 sums in point order ->
//                    its centroid, staged at the position of the cell's first point in the bucket order (unique, no scan
//                    over voxels); the positions of a bucket that start no cell are marked; and one BYTE of an occupancy
//                    bitmap per run of eight cells -- the block owns its runs, so the bitmap is written whole, without a
//                    clearing pass and without atomics
//   vf_bitmap_prefix voxels before every 32-cell word of the bitmap: chunk c's block counts the bits of all the chunks
//                    before it by itself (the bitmap is a few hundred KB out of L2: cheaper than a second launch for the
//                    chunk sums), then scans its own chunk; the last chunk leaves the total
//   vf_place         every staged centroid to its ordinal: voxels before its word + set bits below its own in the word ->
//                    the output is dense and in ascending voxel index, PCL's order
// The general chain (k_count ... k_voxel_centroids) walks the whole cell space three times and gathers every point through
// an index; this form reads the points three times and the cell space never: 1 M points 228 -> 106 us, 2 M 346 -> 148 us per
// call.  For clouds with about a point per cell (an accumulated map) a bucket's share of the cell space -- thousands of cells --
// makes vf_finalize LDS-bound (two blocks per CU) and the chain is as fast: the host takes this form for dense clouds only.
// ---------------------------------------------------------------------------
struct VfCentroidFin {
  int* __restrict__ st_cell;      // [n] voxel index at the position of a cell's first point, -1 elsewhere
  float4* __restrict__ st_cent;   // [n] its centroid
  __device__ __forceinline__ bool operator()(const VoxelSums& S, int n_c, int r, int cell) const {
    const float nf = static_cast<float>(n_c);
    st_cent[r] = make_float4(S.fx / nf, S.fy / nf, S.fz / nf, 1.0f);  // [PCL] centroid /= n, f32 sums in point order
    st_cell[r] = cell;
    return true;
  }
};
__global__ __launch_bounds__(kBlock) void vf_finalize(const float4* __restrict__ bpts, GridGeom g, int map, int K, int C, int lds_cap, int wmax,
                                                      const unsigned* __restrict__ bucket_base, int* __restrict__ st_cell,
                                                      float4* __restrict__ st_cent, unsigned char* __restrict__ bitmap,
                                                      unsigned* __restrict__ bucket_valid, unsigned* __restrict__ scratch, unsigned n_total) {
  extern __shared__ unsigned k1_lds[];
  const K1Deal deal(map & 255, map >> 8);  // (run length 8: a run is a byte of the bitmap)
  const int k = blockIdx.x;
  const unsigned bb = bucket_base[k], be = bucket_base[k + 1], nb = be - bb;
  for (unsigned j = threadIdx.x; j < nb; j += kBlock) st_cell[bb + j] = -1;
  __syncthreads();  // (the marks are out before the finish writes the cells' own)
  k1_finalize_bucket(K1GlobalSrc{bpts + bb, nullptr}, k, bb, nb, k1_lds, g, deal, C, 1, lds_cap, wmax, nullptr, VfCentroidFin{st_cell, st_cent},
                     bucket_valid, scratch, n_total, nullptr);
  __syncthreads();
  // the bucket's runs -> their bytes of the bitmap (k1_lds[0 .. C): the cells' point counts, left by the finish)
  const unsigned* cnt = k1_lds;
  const long long n_runs = (g.n_cells + 7) >> 3;
  for (int r = threadIdx.x; r < (C >> 3); r += kBlock) {
    const long long run = static_cast<long long>(r) * K + k;  // k1_cell: cell = (run << 3) | i
    if (run >= n_runs) continue;
    unsigned bits = 0;
    if (nb) {
#pragma unroll
      for (int i = 0; i < 8; i++) bits |= (cnt[8 * r + i] > 0u ? 1u : 0u) << i;
    }
    bitmap[run] = static_cast<unsigned char>(bits);
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) {  // the bytes behind the last run that still belong to the last word
    const long long idx = n_runs + threadIdx.x, n_bytes = ((n_runs + 3) >> 2) << 2;
    if (idx < n_bytes) bitmap[idx] = 0;
  }
}

constexpr int kVfChunkWords = 4096;  // words (of 32 cells) one block of vf_bitmap_prefix scans
__global__ __launch_bounds__(kBlock) void vf_bitmap_prefix(const unsigned* __restrict__ words, int n_words, unsigned* __restrict__ wprefix,
                                                           unsigned* __restrict__ total /* [2]: -, voxels */) {
  __shared__ unsigned s_w[kBlock / kWave];
  __shared__ unsigned s_base;
  const int lo = blockIdx.x * kVfChunkWords, hi = min(n_words, lo + kVfChunkWords);
  // voxels in the chunks before mine (16-byte loads, eight in flight per thread: the loop is nothing but their latency)
  unsigned before = 0;
  {
    const uint4* w4 = reinterpret_cast<const uint4*>(words);
    const int n4 = lo / 4;  // (lo is a multiple of the chunk size)
    for (int w0 = threadIdx.x; w0 < n4; w0 += 8 * kBlock) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = (w0 + u * kBlock < n4) ? w4[w0 + u * kBlock] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int u = 0; u < 8; u++) before += __popc(v[u].x) + __popc(v[u].y) + __popc(v[u].z) + __popc(v[u].w);
    }
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) before += __shfl_xor(before, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = before;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < kBlock / kWave; w++) t += s_w[w];
    s_base = t;
  }
  __syncthreads();
  // my chunk: 16 consecutive words per thread
  constexpr int kPer = kVfChunkWords / kBlock;
  unsigned v[kPer], sum = 0;
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    const int w = lo + threadIdx.x * kPer + i;
    v[i] = (w < hi) ? __popc(words[w]) : 0u;
    sum += v[i];
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  unsigned inc = sum;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned a = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += a;
  }
  __syncthreads();
  if (lane == kWave - 1) s_w[wave] = inc;
  __syncthreads();
  unsigned run = s_base + inc - sum, chunk_total = 0;
  for (int w = 0; w < kBlock / kWave; w++) {
    if (w < wave) run += s_w[w];
    chunk_total += s_w[w];
  }
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    const int w = lo + threadIdx.x * kPer + i;
    if (w < hi) wprefix[w] = run;
    run += v[i];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) total[1] = s_base + chunk_total;
}

__global__ __launch_bounds__(kBlock) void vf_place(const int* __restrict__ st_cell, const float4* __restrict__ st_cent, const unsigned* __restrict__ n_binned,
                                                   const unsigned* __restrict__ words, const unsigned* __restrict__ wprefix, float4* __restrict__ out) {
  const unsigned n = *n_binned;
  for (unsigned r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
    const int cell = st_cell[r];
    if (cell < 0) continue;
    const unsigned w = static_cast<unsigned>(cell) >> 5, bit = static_cast<unsigned>(cell) & 31u;
    out[wprefix[w] + __popc(words[w] & ((1u << bit) - 1u))] = st_cent[r];
  }
}

// A bucket of at most kSmallFinish points, finished without any per-cell table: a thread per point, the point's place in
// (cell, point index) order counted directly -- every thread reads all nb cells of the bucket out of LDS (broadcast
// reads, four cells each) and counts the points of smaller cells, the points of its own cell, and those of them before
// itself.  The first point of a cell owns the cell: it adds the cell's points up in order and finishes the voxel.  Four block
// barriers in all; k1_finalize_bucket on the same bucket (tables to clear, four scans over the C cells, the passes' selection)
// is ~14 k cycles of which 8 k are such overhead -- at the mapping nodes' size (16 k points over 256 buckets) every bucket is
// this small.  Same outputs as k1_finalize_bucket, bit for bit.  LDS: 4 x nb words from k1_lds.
constexpr int kSmallFinish = 128;
template <class Src>
__device__ __forceinline__ void k1_finish_small(const Src& src, const int k, const unsigned bb, const unsigned nb, unsigned* k1_lds,
                                                const GridGeom& g, const K1Deal& deal, int min_pts, double eig_ratio,
                                                int* __restrict__ sorted_idx, VoxelRec* __restrict__ recs, VoxelSide* __restrict__ centroids,
                                                int* __restrict__ lut, unsigned* __restrict__ bucket_valid) {
  __shared__ unsigned s_valid[kBlock / kWave];
  const unsigned nb4 = (nb + 3u) & ~3u;
  unsigned* cellof = k1_lds;  // [nb4]
  float* sx = reinterpret_cast<float*>(k1_lds + nb4);
  float* sy = sx + nb4;
  float* sz = sy + nb4;
  const unsigned j = threadIdx.x;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned c = 0xffffffffu;
  if (j < nb) {
    p = src(j);
    c = static_cast<unsigned>(k1_local(build_cell(g, p.x, p.y, p.z), deal));
  }
  if (j < nb4) cellof[j] = c;  // (the padding: larger than any cell)
  __syncthreads();
  unsigned lt = 0, eq = 0, eq_before = 0;
  if (j < nb) {
    for (unsigned i = 0; i < nb4; i += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(cellof + i);
      lt += (v.x < c) + (v.y < c) + (v.z < c) + (v.w < c);
      const unsigned e0 = v.x == c, e1 = v.y == c, e2 = v.z == c, e3 = v.w == c;
      eq += e0 + e1 + e2 + e3;
      eq_before += (e0 & (i < j)) + (e1 & (i + 1 < j)) + (e2 & (i + 2 < j)) + (e3 & (i + 3 < j));
    }
    const unsigned q = lt + eq_before;
    sx[q] = p.x;
    sy[q] = p.y;
    sz[q] = p.z;
    sorted_idx[bb + q] = __float_as_int(p.w);
  }
  __syncthreads();
  unsigned ok = 0;
  if (j < nb && eq_before == 0 && static_cast<int>(eq) >= min_pts) {
    const FinalizeDump nodump{nullptr, nullptr, nullptr, nullptr, nullptr};
    VoxelSums S;
    for (unsigned i = 0; i < eq; i++) S.add(sx[lt + i], sy[lt + i], sz[lt + i]);
    const int r = static_cast<int>((bb + lt) / static_cast<unsigned>(min_pts));
    ok = finish_voxel(S, static_cast<int>(eq), 0, r, k1_cell(k, static_cast<int>(c), deal), min_pts, eig_ratio, recs, centroids, lut, g, nodump) ? 1u : 0u;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) ok += __shfl_xor(ok, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) s_valid[threadIdx.x / kWave] = ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < kBlock / kWave; w++) t += s_valid[w];
    bucket_valid[k] = t;
  }
}

// ---------------------------------------------------------------------------
// K1 for small clouds (the mapping nodes' 16 k points): ONE launch.  The four-kernel chain above is, at that size, four launch
// boundaries around ~3 us of work each.  Here every block reads the WHOLE cloud (256 KB out of L2), keeps the points of its
// own bucket and only counts the others, and then finishes its bucket exactly as k1_finalize does:
//   scan     eight waves (two per SIMD: one wave alone issues a dependent instruction every ~8 cycles, two share the SIMD
//            without slowing each other); wave w walks the w-th eighth of the cloud in 64-point chunks: cell, bucket; ballot of "bucket below mine" ->
//            running count (the bucket's base in the bucket order, without any histogram, scan or exchange between blocks),
//            ballot of "mine" -> the point goes to the wave's list in LDS (x, y, z, index).  The lists of waves 0..7 one
//            after the other hold the bucket's points in ascending point index -- the order k1_scatter produces.
//   publish  bucket_base[k], the bucketed points (the leaf pass and grid_counts read both later on)
//   table    the block owns the look-up table slots of its bucket's cells (all written: record or empty) and the border
//            slots of the k-th slice of the padded table -- every slot has one writer, nothing is cleared beforehand
//   finish   by the first four waves (the others have ended): k1_finish_small / k1_finalize_bucket on the lists
// No block waits for another one: no grid barrier, no co-residency requirement, fine on a CU-masked stream or beside a
// resident evaluation server.  The price is the redundant scan, n x K cell computations (~5 us per 16 k points on every
// CU: issue-bound, not memory-bound -- the same loop over one L1-resident kilobyte is 8 % faster) -- hence small clouds
// only.  A wave whose list overflows (a bucket with more than list_cap points from one eighth of the cloud) is detected by the whole block; the block then scans a second time and writes its points straight to their
// final places in the bucketed cloud, and finishes from there.
// ---------------------------------------------------------------------------
constexpr int kSmallWaves = 8;                      // waves of the scan (two per SIMD); the first kBlock / kWave of them finish the bucket
constexpr int kSmallThreads = kSmallWaves * kWave;  // 512
struct K1ListSrc {
  const float4* list;    // [kSmallWaves][cap]
  const unsigned* seg;   // LDS [kSmallWaves + 1]: first bucket position of every wave's list, then nb
  int cap;
  const float4* spilled;  // after an overflow: the bucket's slice of the bucketed cloud instead (else null)
  __device__ __forceinline__ float4 operator()(unsigned j) const {
    if (spilled) return spilled[j];  // (uniform)
    unsigned w = 0;
#pragma unroll
    for (int i = 1; i < kSmallWaves; i++) w += (j >= seg[i]) ? 1u : 0u;
    return list[w * static_cast<unsigned>(cap) + (j - seg[w])];
  }
};
constexpr int kSmallBatch = 8;               // 64-point chunks a wave has in flight
// The bucket of a point for the redundant scan, in as few instructions as it takes (every block does this for EVERY point of
// the cloud): the per-axis indices exactly as build_cell computes them, each tested against its axis (so that the 24-bit
// multiply-adds below are exact and the cell needs no further range test: a small grid has fewer than 2^24 cells), the run's
// bucket by a mask when K is a power of two.  -1: not a point of the grid (non-finite, or outside the box: garbage in a
// cloud declared dense) -- such a point is dropped here, so the finish never sees it.
template <bool DENSE, bool POW2>
__device__ __forceinline__ int small_bucket(const GridGeom& g, const K1Deal& deal, float x, float y, float z) {
#pragma clang fp contract(off)
  const float fx = x * g.inv_leaf[0], fy = y * g.inv_leaf[1], fz = z * g.inv_leaf[2];
  const int i0 = static_cast<int>(floorf(fx) - static_cast<float>(g.min_b[0]));
  const int i1 = static_cast<int>(floorf(fy) - static_cast<float>(g.min_b[1]));
  const int i2 = static_cast<int>(floorf(fz) - static_cast<float>(g.min_b[2]));
  // The box is the box of these very points, so a finite point is inside it -- 0 <= i < div_b < 2^24 on every axis, the
  // 24-bit multiply-adds are exact and equal build_cell's -- and needs no range test.  What is not finite: in a cloud
  // declared dense nothing that gets here (an infinity makes the box overflow, the host stops before any kernel; a NaN
  // converts to index 0 on every axis here as in build_cell); otherwise the point is dropped.
  const unsigned cell = __umul24(static_cast<unsigned>(i2), static_cast<unsigned>(g.mul[2])) +
                        (__umul24(static_cast<unsigned>(i1), static_cast<unsigned>(g.mul[1])) + static_cast<unsigned>(i0));
  const unsigned run = cell >> deal.rb;
  int b;
  if (POW2) {
    b = static_cast<int>(run & (deal.K - 1u));
  } else {
    unsigned q, r;
    deal.divmod(run, q, r);
    b = static_cast<int>(r);
  }
  if (!DENSE) {
    const float t = (fx + fy) + fz;  // not finite as soon as one coordinate is not (inf - inf = NaN)
    if (!(fabsf(t) < INFINITY)) b = -1;
  }
  return b;
}

// One scan of this wave's chunks [c_lo, c_hi).  PASS 0: count the points of the buckets below k (-> below), collect the
// points of bucket k in the wave's list (own = how many there are, also beyond the list's capacity).  PASS 1: the points of
// bucket k go to out[own++] (their final places in the bucketed cloud).
template <bool DENSE, bool POW2, int PASS>
__device__ __forceinline__ void small_scan(const float4* __restrict__ pts, int n, const GridGeom& g, const K1Deal& deal, int k, int c_lo, int c_hi,
                                           float4* mylist, int list_cap, float4* __restrict__ out, unsigned& own_out, unsigned& below_out) {
  const int lane = threadIdx.x & (kWave - 1);
  unsigned own = 0, below = 0;  // wave-uniform
  const int full_hi = min(c_hi, n / kWave);  // chunks below this one are whole: no bounds test per point
  auto visit = [&](const float4& p, int i, bool in_range) {
    int b = small_bucket<DENSE, POW2>(g, deal, p.x, p.y, p.z);
    if (!in_range) b = -1;
    if (PASS == 0) below += static_cast<unsigned>(__popcll(__ballot(static_cast<unsigned>(b) < static_cast<unsigned>(k))));
    const unsigned long long mine = __ballot(b == k);
    if (mine) {  // (uniform, and rare: one point in K)
      const unsigned pos = own + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mine >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mine), 0u));
      if (b == k) {
        const float4 q = make_float4(p.x, p.y, p.z, __int_as_float(i));
        if (PASS == 0) {
          if (pos < static_cast<unsigned>(list_cap)) mylist[pos] = q;
        } else {
          out[pos] = q;
        }
      }
      own += static_cast<unsigned>(__popcll(mine));
    }
  };
  int c0 = c_lo;
  if (c0 + kSmallBatch <= full_hi) {
    // whole batches, two register sets: while one batch is looked at, the loads of the next one are in flight (a copy from a
    // "next" into a "current" set made the compiler wait for the loads it had just issued)
    const int nbat = (full_hi - c_lo) / kSmallBatch;
    float4 pa[kSmallBatch], pb[kSmallBatch];
    auto fetch = [&](float4* dst, int t) {
      const int cb = c_lo + t * kSmallBatch;
#pragma unroll
      for (int u = 0; u < kSmallBatch; u++) dst[u] = pts[(cb + u) * kWave + lane];
    };
    auto look = [&](const float4* src, int t) {
      const int cb = c_lo + t * kSmallBatch;
#pragma unroll
      for (int u = 0; u < kSmallBatch; u++) visit(src[u], (cb + u) * kWave + lane, true);
    };
    // (every fetch unconditional -- the last ones read a batch again -- so that the compiler knows how many loads are
    // outstanding at every point and waits for exactly the set it is about to look at)
    fetch(pa, 0);
    for (int t = 0; t < nbat; t += 2) {
      fetch(pb, min(t + 1, nbat - 1));
      look(pa, t);
      fetch(pa, min(t + 2, nbat - 1));
      if (t + 1 < nbat) look(pb, t + 1);  // (uniform)
    }
    c0 = c_lo + nbat * kSmallBatch;
  }
  for (; c0 < c_hi; c0++) {  // the rest, chunk by chunk
    const int i = c0 * kWave + lane;
    const float4 p = pts[min(i, n - 1)];
    visit(p, i, i < n);
  }
  own_out = own;
  below_out = below;
}

__global__ __launch_bounds__(kSmallThreads) void k1_small(const float4* __restrict__ pts, int n, int dense, GridGeom g, int map, int K, int C, int min_pts,
                                                   double eig_ratio, int lds_cap, int wmax, int list_cap, unsigned fin_words /* LDS words of the finish */,
                                                   int small_finish /* buckets up to this size: k1_finish_small (0: never) */,
                                                   unsigned* __restrict__ bucket_base, float4* __restrict__ bpts, int* __restrict__ sorted_idx,
                                                   VoxelRec* __restrict__ recs, VoxelSide* __restrict__ centroids, int* __restrict__ lut,
                                                   unsigned* __restrict__ bucket_valid, unsigned* __restrict__ scratch, unsigned* __restrict__ counts,
                                                   unsigned long long* __restrict__ st /* development aid (NDT_K1_STAMPS), or null */) {
  extern __shared__ unsigned k1_lds[];
  auto mark = [&](int q) {  // thread 0's clock: 0 start, 1 scan done, 2 published + table slots cleared, 3 end
    if (st && threadIdx.x == 0) st[8 * static_cast<size_t>(K) + 4 * blockIdx.x + q] = stamp();
  };
  mark(0);
  const K1Deal deal(map & 255, map >> 8);
  __shared__ unsigned s_own[kSmallWaves], s_below[kSmallWaves], s_seg[kSmallWaves + 1];
  float4* list = reinterpret_cast<float4*>(k1_lds + fin_words);  // (fin_words is a multiple of 4)
  const int k = blockIdx.x;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int chunks = (n + kWave - 1) / kWave, per = (chunks + kSmallWaves - 1) / kSmallWaves;  // chunks per wave
  const int c_lo = min(chunks, wave * per), c_hi = min(chunks, c_lo + per);
  float4* mylist = list + wave * list_cap;
  const bool pow2 = (K & (K - 1)) == 0;
  // ---- scan ----
  unsigned own = 0, below = 0;
  if (dense) {
    if (pow2) small_scan<true, true, 0>(pts, n, g, deal, k, c_lo, c_hi, mylist, list_cap, nullptr, own, below);
    else small_scan<true, false, 0>(pts, n, g, deal, k, c_lo, c_hi, mylist, list_cap, nullptr, own, below);
  } else {
    if (pow2) small_scan<false, true, 0>(pts, n, g, deal, k, c_lo, c_hi, mylist, list_cap, nullptr, own, below);
    else small_scan<false, false, 0>(pts, n, g, deal, k, c_lo, c_hi, mylist, list_cap, nullptr, own, below);
  }
  if (lane == 0) {
    s_own[wave] = own;
    s_below[wave] = below;
  }
  __syncthreads();
  mark(1);
  unsigned base_w = 0, bb = 0, nb = 0;
  bool overflow = false;
  for (int w = 0; w < kSmallWaves; w++) {
    bb += s_below[w];
    if (w < wave) base_w += s_own[w];
    nb += s_own[w];
    overflow = overflow || s_own[w] > static_cast<unsigned>(list_cap);
  }
  if (overflow) {  // (uniform, rare) a list was too short: a second scan, the points straight to their final places
    unsigned own2 = 0, dummy = 0;
    if (dense) small_scan<true, false, 1>(pts, n, g, deal, k, c_lo, c_hi, nullptr, 0, bpts + bb + base_w, own2, dummy);
    else small_scan<false, false, 1>(pts, n, g, deal, k, c_lo, c_hi, nullptr, 0, bpts + bb + base_w, own2, dummy);
  }
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int w = 0; w < kSmallWaves; w++) {
      s_seg[w] = run;
      run += s_own[w];
    }
    s_seg[kSmallWaves] = run;
    bucket_base[k] = bb;
    if (k == K - 1) {
      bucket_base[K] = bb + nb;
      counts[0] = bb + nb;  // points binned
    }
  }
  const K1ListSrc lsrc{list, s_seg, list_cap, overflow ? bpts + bb : nullptr};
  if (!overflow) {  // (uniform)
    __syncthreads();  // s_seg
    for (unsigned j = threadIdx.x; j < nb; j += kSmallThreads) bpts[bb + j] = lsrc(j);
  }
  // ---- the look-up table: border slots of my slice, and every cell of my bucket starts out empty ----
  {
    const long long per_blk = (g.lut_cells + K - 1) / K;
    const long long lo = static_cast<long long>(k) * per_blk, hi = min(g.lut_cells, lo + per_blk);
    const int ex = g.div_b[0] + kLutBorder, ey = g.div_b[1] + kLutBorder, ez = g.div_b[2] + kLutBorder;
    // (a / b for a < 2^31 and a quotient below 2^22 -- an axis of the grid: the product with the f32 reciprocal is within
    // one of it; two 32-bit integer divisions per slot were a third of this phase)
    auto divq = [](unsigned a, unsigned b, float rb) {
      unsigned q = static_cast<unsigned>(__uint2float_rz(a) * rb);
      int r = static_cast<int>(a - q * b);
      if (r < 0) { q--; r += static_cast<int>(b); }
      if (r >= static_cast<int>(b)) q++;
      return q;
    };
    const float rp2 = 1.0f / static_cast<float>(g.pmul[2]), rp1 = 1.0f / static_cast<float>(g.pmul[1]);
    const float rm2 = 1.0f / static_cast<float>(g.mul[2]), rm1 = 1.0f / static_cast<float>(g.mul[1]);
    for (long long sl = lo + threadIdx.x; sl < hi; sl += kSmallThreads) {
      const unsigned u = static_cast<unsigned>(sl);  // (a small grid's padded table has fewer than 2^31 slots)
      const int pz = static_cast<int>(divq(u, static_cast<unsigned>(g.pmul[2]), rp2));
      const unsigned rem = u - static_cast<unsigned>(pz) * static_cast<unsigned>(g.pmul[2]);
      const int py = static_cast<int>(divq(rem, static_cast<unsigned>(g.pmul[1]), rp1)), px = static_cast<int>(rem) - py * g.pmul[1];
      if (px < kLutBorder || px >= ex || py < kLutBorder || py >= ey || pz < kLutBorder || pz >= ez) lut[sl] = kLutEmpty;
    }
    for (int lc = threadIdx.x; lc < C; lc += kSmallThreads) {
      const long long cell = static_cast<unsigned>(k1_cell(k, lc, deal));
      if (cell < g.n_cells) {
        const unsigned c = static_cast<unsigned>(cell);
        const int cz = static_cast<int>(divq(c, static_cast<unsigned>(g.mul[2]), rm2));
        const unsigned rem = c - static_cast<unsigned>(cz) * static_cast<unsigned>(g.mul[2]);
        const int cy = static_cast<int>(divq(rem, static_cast<unsigned>(g.mul[1]), rm1)), cx = static_cast<int>(rem) - cy * g.mul[1];
        lut[static_cast<long long>(cx + kLutBorder) + static_cast<long long>(cy + kLutBorder) * g.pmul[1] + static_cast<long long>(cz + kLutBorder) * g.pmul[2]] = kLutEmpty;
      }
    }
  }
  __syncthreads();  // the bucketed points and the empty slots are out before anything of the finish reads / overwrites them
  mark(2);
  // The finish is written for kBlock threads: the scan's other waves end here.  (A barrier waits for the waves of the
  // workgroup that have not terminated -- s_barrier, CDNA3 ISA 4.4 -- so the barriers of the finish are the four waves' own.)
  if (threadIdx.x >= kBlock) return;
  if (nb <= static_cast<unsigned>(small_finish))  // (uniform)
    k1_finish_small(lsrc, k, bb, nb, k1_lds, g, deal, min_pts, eig_ratio, sorted_idx, recs, centroids, lut, bucket_valid);
  else
    k1_finalize_bucket(lsrc, k, bb, nb, k1_lds, g, deal, C, min_pts, lds_cap, wmax, sorted_idx, K1RecordFin{min_pts, eig_ratio, recs, centroids, lut, &g},
                       bucket_valid, scratch, static_cast<unsigned>(n), st);
  mark(3);
}

// ---------------------------------------------------------------------------
// Record compaction of a bucket-form build.  k1_finalize numbers a voxel's record by where its points sit in the bucket
// order (unique without a scan over voxels) -- slots with gaps, bucket by bucket.  The evaluation kernels gather records
// through the look-up table for points that arrive in lattice order, and they run measurably faster (+5 % on the headline
// registration) when the records are dense and in ascending cell order, as the general chain leaves them: neighbouring
// voxels then share cache lines and a wave's gathers walk the array forwards.  Two launches over the padded table (three
// for tables of more than kRcPrefixTiles tiles): count the records per tile; then every block sums the counts before its
// own tile, gives the tile's records their ordinals in table order, moves the 64-B records + side sectors there and
// rewrites the table entries.  (Measured and dropped: counting per tile from k1_finalize with one atomic per voxel --
// 100 k atomics on 512 addresses took k1_finalize from 42 to 97 us; a single launch with a block ticket and look-back
// over published tile counts -- the 512 same-address ticket atomics alone cost 10 us.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool lut_has_record(int e) { return e >= 0 || e <= -2; }
// records per tile of 256 x ITEMS table entries
template <int ITEMS>
__global__ __launch_bounds__(kBlock) void k_rc_count(const int* __restrict__ lut, long long n, unsigned* __restrict__ tile_sums) {
  __shared__ unsigned s_w[kBlock / kWave];
  const long long base = (static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x) * ITEMS;
  unsigned c = 0;
#pragma unroll
  for (int u = 0; u < ITEMS; u++) c += (base + u < n && lut_has_record(lut[base + u])) ? 1u : 0u;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) c += __shfl_xor(c, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < kBlock / kWave; w++) t += s_w[w];
    tile_sums[blockIdx.x] = t;
  }
}
// big tables (more than kRcPrefixTiles tiles): exclusive scan of the tile sums in place, by one block
__global__ __launch_bounds__(kBlock) void k_rc_scan(unsigned* __restrict__ tile_sums, int n_tiles) {
  __shared__ unsigned s_scan[kBlock / kWave];
  block_scan_array(tile_sums, tile_sums, n_tiles, kBlock, s_scan, false);
}
// One tile per block.  SCANNED: tile_sums[] already holds the tiles' first record ordinals (k_rc_scan); otherwise the block
// adds up the counts of the tiles before its own (at most kRcPrefixTiles words, out of L2).
template <int ITEMS, bool SCANNED>
__global__ __launch_bounds__(kBlock) void k_rc_apply(int* __restrict__ lut, long long n, const unsigned* __restrict__ tile_sums,
                                                     const VoxelRec* __restrict__ recs_in, const VoxelSide* __restrict__ cent_in,
                                                     VoxelRec* __restrict__ recs_out, VoxelSide* __restrict__ cent_out) {
  __shared__ unsigned s_w[kBlock / kWave], s_pre[kBlock / kWave];
  __shared__ int s_old[kBlock * ITEMS];  // old slot of the tile's k-th record
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (!SCANNED) {
    unsigned pre = 0;
    for (unsigned t = threadIdx.x; t < blockIdx.x; t += kBlock) pre += tile_sums[t];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) pre += __shfl_xor(pre, off, kWave);
    if (lane == 0) s_pre[wave] = pre;
  }
  const long long base = (static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x) * ITEMS;
  int e[ITEMS];
  unsigned c = 0;
  if (ITEMS == 8 && base + ITEMS <= n) {  // (base is a multiple of ITEMS: two aligned 16-B loads)
    const int4 lo = *reinterpret_cast<const int4*>(lut + base), hi = *reinterpret_cast<const int4*>(lut + base + 4);
    e[0] = lo.x; e[1 % ITEMS] = lo.y; e[2 % ITEMS] = lo.z; e[3 % ITEMS] = lo.w;
    e[4 % ITEMS] = hi.x; e[5 % ITEMS] = hi.y; e[6 % ITEMS] = hi.z; e[7 % ITEMS] = hi.w;
  } else {
#pragma unroll
    for (int u = 0; u < ITEMS; u++) e[u] = (base + u < n) ? lut[base + u] : kLutEmpty;
  }
#pragma unroll
  for (int u = 0; u < ITEMS; u++) c += lut_has_record(e[u]) ? 1u : 0u;
  // exclusive scan of c over the block: wave scan, then the wave totals
  unsigned inc = c;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned a = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += a;
  }
  if (lane == kWave - 1) s_w[wave] = inc;
  __syncthreads();
  unsigned before = 0, total = 0, first = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; w++) {
    if (w < wave) before += s_w[w];
    total += s_w[w];
    if (!SCANNED) first += s_pre[w];
  }
  if (SCANNED) first = tile_sums[blockIdx.x];
  unsigned k = before + inc - c;  // ordinal inside the tile
#pragma unroll
  for (int u = 0; u < ITEMS; u++) {
    if (!lut_has_record(e[u])) continue;
    s_old[k] = (e[u] >= 0) ? e[u] : -(e[u] + 2);
    const int r_new = static_cast<int>(first + k);
    lut[base + u] = (e[u] >= 0) ? r_new : lut_rejected(r_new);
    k++;
  }
  __syncthreads();
  // the moves, four lanes per 64-B record: a wave writes 1 KiB of consecutive bytes
  const float4* rin = reinterpret_cast<const float4*>(recs_in);
  const float4* cin = reinterpret_cast<const float4*>(cent_in);
  float4* rout = reinterpret_cast<float4*>(recs_out) + static_cast<size_t>(first) * 4;
  float4* cout = reinterpret_cast<float4*>(cent_out) + static_cast<size_t>(first) * 4;
  for (unsigned j = threadIdx.x; j < total * 4; j += kBlock) {
    const size_t from = static_cast<size_t>(s_old[j >> 2]) * 4 + (j & 3);
    const float4 a = rin[from], b = cin[from];
    rout[j] = a;
    cout[j] = b;
  }
}

// Leaf arrays of a bucket-form build (ascending cell order: leaf_cell / leaf_start / leaf_count / leaf_rec), written only
// when somebody asks for them (ndt_grid_dump, getFitnessScore's index, ndt_grid_size): after k1_count has numbered
// the buckets' occupied cells.
__global__ __launch_bounds__(kBlock) void k1_leaves(const float4* __restrict__ bpts, GridGeom g, int map, int C, int min_pts,
                                                    const unsigned* __restrict__ bucket_base, const unsigned* __restrict__ occ_base,
                                                    int* __restrict__ leaf_cell, unsigned* __restrict__ leaf_start,
                                                    int* __restrict__ leaf_count, int* __restrict__ leaf_rec, const int* __restrict__ lut,
                                                    const float4* __restrict__ cloud) {
  extern __shared__ unsigned k1_lds[];
  const K1Deal deal(map & 255, map >> 8);
  __shared__ U3 s_u3[kBlock / kWave];
  const int k = blockIdx.x;
  const unsigned bb = bucket_base[k], be = bucket_base[k + 1];
  if (be == bb) return;
  unsigned* cnt = k1_lds;
  k1_cell_histogram(k1_src(bpts, cloud, bb), be - bb, g, deal, C, cnt);
  const int per = C / kBlock > 0 ? C / kBlock : 1;
  const int lo = threadIdx.x * per;
  U3 t = {0, 0, 0};
  for (int c = lo; c < lo + per && c < C; c++) {
    t.pts += cnt[c];
    t.occ += (cnt[c] > 0);
  }
  U3 total;
  U3 run = block_exclusive_scan(t, total, s_u3);
  const unsigned ob = occ_base[k];
  for (int c = lo; c < lo + per && c < C; c++) {
    const unsigned v = cnt[c];
    if (v > 0) {
      const unsigned o = ob + run.occ;
      leaf_cell[o] = k1_cell(k, c, deal);
      leaf_start[o] = bb + run.pts;
      leaf_count[o] = static_cast<int>(v);
      int rec = -1;
      if (v >= static_cast<unsigned>(min_pts)) {  // the voxel's record: wherever the compaction put it (read back from the table)
        const int cell = k1_cell(k, c, deal);
        const int cz = cell / g.mul[2], cy = (cell - cz * g.mul[2]) / g.mul[1], cx = cell - cz * g.mul[2] - cy * g.mul[1];
        const int e = lut[static_cast<long long>(cx + kLutBorder) + static_cast<long long>(cy + kLutBorder) * g.pmul[1] +
                          static_cast<long long>(cz + kLutBorder) * g.pmul[2]];
        rec = (e >= 0) ? e : (e <= -2 ? -(e + 2) : -1);
      }
      leaf_rec[o] = rec;
    }
    run.pts += v;
    run.occ += (v > 0);
  }
}


// k_presort_large: leaves per wave-step and grid (one leaf per wave while that stays within 8192 waves)
inline int presort_chunk(int n_leaves) { return max(1, min(64, (n_leaves + 8191) / 8192)); }
inline int presort_grid(int n_leaves) { return max(1, min(8192, (n_leaves + presort_chunk(n_leaves) - 1) / presort_chunk(n_leaves))); }

inline int grid_for(size_t n, int max_blocks) {
  size_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > static_cast<size_t>(max_blocks)) b = max_blocks;
  return static_cast<int>(b);
}


}  // namespace

// ===========================================================================
// launchers
// ===========================================================================
hipError_t launch_repack(const void* d_src, size_t n, size_t stride_bytes, float4* d_dst, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_repack, dim3(grid_for(n, 2048)), dim3(kBlock), 0, stream,
                     static_cast<const unsigned char*>(d_src), n, stride_bytes, d_dst);
  return hipGetLastError();
}

hipError_t launch_repack_bbox(const void* d_src, size_t n, size_t stride_bytes, float4* d_dst, float* d_block_minmax,
                              int n_blocks, hipStream_t stream, unsigned tag, const unsigned* n_dev) {
  if (n == 0) return hipSuccess;
  const bool rec16 = stride_bytes == 16 && (reinterpret_cast<uintptr_t>(d_src) & 15) == 0;
  if ((tag || n_dev) && !rec16) return hipErrorInvalidValue;  // tagged rows, device-side counts: the 16-byte-record forms only
  if (rec16 && !d_dst)
    hipLaunchKernelGGL(k_bbox16<false>, dim3(n_blocks), dim3(kBlock), 0, stream, static_cast<const float4*>(d_src), n, nullptr, d_block_minmax, tag, n_dev);
  else if (rec16)
    hipLaunchKernelGGL(k_bbox16<true>, dim3(n_blocks), dim3(kBlock), 0, stream, static_cast<const float4*>(d_src), n, d_dst, d_block_minmax, tag, n_dev);
  else
    hipLaunchKernelGGL(k_repack_bbox, dim3(n_blocks), dim3(kBlock), 0, stream, static_cast<const unsigned char*>(d_src), n,
                       stride_bytes, d_dst, d_block_minmax);
  return hipGetLastError();
}

hipError_t launch_bbox(const float4* pts, int n, int dense, float* d_block_minmax, int n_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(k_bbox, dim3(n_blocks), dim3(kBlock), 0, stream, pts, n, dense, d_block_minmax);
  return hipGetLastError();
}

hipError_t launch_count(const float4* pts, int n, int dense, const GridGeom& g, int* d_key, unsigned* d_rank,
                        unsigned* d_cell_count, hipStream_t stream) {
  hipLaunchKernelGGL(k_count, dim3(grid_for(n, 2048)), dim3(kBlock), 0, stream, pts, n, dense, g, d_key, d_rank, d_cell_count);
  return hipGetLastError();
}

hipError_t launch_scan_reduce(const unsigned* d_cell_count, long long n_cells, int min_pts, unsigned* d_block_sums,
                              int n_tiles, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_reduce, dim3(n_tiles), dim3(kBlock), 0, stream, d_cell_count, n_cells,
                     static_cast<unsigned>(min_pts), d_block_sums);
  return hipGetLastError();
}

hipError_t launch_scan_blocks(unsigned* d_block_sums, int n_tiles, unsigned* d_totals, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(kBlock), 0, stream, d_block_sums, n_tiles, d_totals);
  return hipGetLastError();
}

hipError_t launch_scan_apply(unsigned* d_cell_count_to_cursor, long long n_cells, int min_pts,
                             const unsigned* d_block_sums, int n_tiles, int* d_leaf_cell,
                             unsigned* d_leaf_start, int* d_leaf_count, int* d_leaf_rec, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_apply, dim3(n_tiles), dim3(kBlock), 0, stream, d_cell_count_to_cursor, n_cells,
                     static_cast<unsigned>(min_pts), d_block_sums, d_leaf_cell, d_leaf_start, d_leaf_count,
                     d_leaf_rec);
  return hipGetLastError();
}

hipError_t launch_scatter(const int* d_key, const unsigned* d_rank, int n, const unsigned* d_cell_start, int* d_sorted_idx,
                          hipStream_t stream) {
  hipLaunchKernelGGL(k_scatter, dim3(grid_for(n, 2048)), dim3(kBlock), 0, stream, d_key, d_rank, n, d_cell_start, d_sorted_idx);
  return hipGetLastError();
}

hipError_t launch_finalize(const float4* pts, const int* d_leaf_cell, const unsigned* d_leaf_start,
                           const int* d_leaf_count, const int* d_leaf_rec, int n_leaves, int* d_sorted_idx,
                           int min_pts, double eig_ratio, VoxelRec* d_recs, VoxelSide* d_centroids, int* d_lut, const GridGeom& geom,
                           unsigned* d_n_valid, FinalizeDump dump, hipStream_t stream, const unsigned* d_totals, float4* d_big_pts) {
  // d_totals != nullptr: n_leaves is an upper bound (grid size); the kernel reads the count itself
  if (n_leaves == 0) return hipSuccess;
  if (d_big_pts)  // leaves with many points: sorted and gathered by one wave each, ahead of the per-leaf pass
    hipLaunchKernelGGL(k_presort_large, dim3(presort_grid(n_leaves)), dim3(kBlock), 0, stream, pts, d_leaf_start, d_leaf_count, n_leaves,
                       d_totals, d_sorted_idx, d_big_pts, presort_chunk(n_leaves));
  hipLaunchKernelGGL(k_finalize, dim3((n_leaves + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, pts, d_leaf_cell,
                     d_leaf_start, d_leaf_count, d_leaf_rec, n_leaves, d_totals, d_sorted_idx, min_pts, eig_ratio, d_recs, d_centroids,
                     d_lut, geom, d_n_valid, dump, d_big_pts, dump.nr_points ? nullptr : d_n_valid + 1);
  return hipGetLastError();
}

static int pow2_ceil(long long v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

bool grid_build_plan(long long n_cells, int n_points, GridBuildPlan& P) {
  constexpr int kMaxBuckets = kK1MaxBuckets, kMaxCells = 4096;
  if (n_points <= 0 || n_cells <= 0 || n_cells > static_cast<long long>(kMaxBuckets) * kMaxCells) return false;
  // ~1000 points per bucket: a bucket's per-point arrays then live in LDS and there are several blocks per CU
  // ... and at least a bucket per CU; small clouds (the mapping nodes' 16 k points: latency-bound on their fullest bucket)
  // get ~256 points per bucket (measured with the interleaved buckets: 16 k / 60 k / 200 k points 41 / 46 / 71 us per build,
  // against 54 / 74 / 117 us with ~8 points per bucket and 67 / 80 / 100 us for the general chain)
  static const int small_div = [] { const char* v = getenv("NDT_K1_SMALL_DIV"); return v ? std::max(1, atoi(v)) : 256; }();
  const long long k_small = std::min<long long>(4096, n_points / small_div);
  static const int big_div = [] { const char* v = getenv("NDT_K1_BUCKET_POINTS"); return v ? std::max(64, atoi(v)) : 1024; }();
  const long long k_target = std::max<long long>(std::max<long long>(256, n_points <= 262144 ? k_small : 0), std::min<long long>(kMaxBuckets, n_points / big_div));
  // cells are dealt to the buckets in runs of 2^rb (k1_bucket); C = slots per bucket << rb
  static const int rb_env = [] { const char* v = getenv("NDT_K1_RUN_BITS"); return v ? std::max(0, std::min(8, atoi(v))) : 3; }();
  const int rb = rb_env;
  // K: a power of two by default; NDT_K1_BUCKETS forces any multiple of 8.  (Measured and dropped: 768 buckets at 1 M points,
  // one block per bucket = exactly the three blocks per CU that k1_finalize's registers allow, so that the kernel runs in
  // one round instead of one and a third -- 27.4 against 26.0 us on the uniform scene, 54.7 against 41.5 us on surfaces:
  // the blocks of a round are in the same phase at the same time, and fewer, longer blocks overlap less.)
  static const int k_env = [] { const char* v = getenv("NDT_K1_BUCKETS"); return v ? std::max(8, atoi(v)) / 8 * 8 : 0; }();
  int K, k_step;
  if (k_env > 0) {
    K = std::min(kMaxBuckets, k_env);
    k_step = 8;
  } else {
    K = std::min(kMaxBuckets, pow2_ceil(k_target));
    k_step = -1;  // (doubling)
  }
  const long long runs = (n_cells + (1ll << rb) - 1) >> rb;
  int C;
  for (;;) {
    C = std::max(32, pow2_ceil((runs + K - 1) / K) << rb);
    if (C <= kMaxCells) break;
    const int next = k_step < 0 ? K * 2 : K + k_step;
    if (next > kMaxBuckets) return false;
    K = next;
  }
  P.cells_per_bucket = C;
  P.shift = rb | (K << 8);  // the packed cell <-> (bucket, local) map of the kernels: run bits, bucket count (K1Deal)
  P.n_buckets = K;
  // blocks of k1_hist / k1_scatter: two per CU at 1 M points; at most 512 rows in the count matrix (k1_colscan)
  long long ppb = std::max(512, std::min(kK1Round, pow2_ceil((n_points + 511) / 512)));
  static const int ppb_env = [] { const char* v = getenv("NDT_K1_PPB"); return v ? atoi(v) : 0; }();
  if (ppb_env > 0) ppb = ppb_env;
  while ((n_points + ppb - 1) / ppb > kColGroups * kColRows) ppb += kK1Round;
  P.pts_per_block = static_cast<int>(ppb);
  P.n_blocks = ((n_points + P.pts_per_block - 1) / P.pts_per_block + 7) / 8 * 8;  // a multiple of 8: k1_slice (empty slices cost nothing)
  return true;
}

hipError_t launch_grid_build_buckets(const float4* pts, int n, int dense, const GridGeom& g, const GridBuildPlan& P, int min_pts,
                                     double eig_ratio, const GridBuildScratch& S, int* sorted_idx, VoxelRec* recs, VoxelSide* centroids,
                                     int* lut, unsigned* counts, hipStream_t stream) {
  const int K = P.n_buckets, C = P.cells_per_bucket;
  const size_t lds_scatter = (static_cast<size_t>(K) + 2) * sizeof(unsigned) +
                             static_cast<size_t>(K) * sizeof(unsigned short) + static_cast<size_t>(kK1Waves) * K * sizeof(unsigned short);
  if (lds_scatter > kK1MaxDynamicLds || P.n_blocks > kColGroups * kColRows) return hipErrorInvalidValue;  // (grid_build_plan never asks for this)
  static bool once = [] {  // more than 64 KB of dynamic LDS has to be asked for
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kK1MaxDynamicLds));
    return true;
  }();
  (void)once;
  unsigned* total = S.cntmat + static_cast<size_t>(P.n_blocks) * K;
  hipLaunchKernelGGL(k1_hist, dim3(P.n_blocks), dim3(kK1Threads), static_cast<size_t>(K) * sizeof(unsigned), stream, pts, n, dense, g, P.shift, K,
                     P.pts_per_block, S.cntmat, lut, g.lut_cells, static_cast<const unsigned*>(nullptr));
  hipLaunchKernelGGL(k1_colscan, dim3((K + kColCols - 1) / kColCols), dim3(kK1Threads), 0, stream, S.cntmat, P.n_blocks, K, total);
  hipLaunchKernelGGL(k1_scatter, dim3(P.n_blocks), dim3(kK1Threads), lds_scatter, stream, pts, n, dense, g, P.shift, K, P.pts_per_block, S.cntmat,
                     total, S.bucket_base, S.bpts, counts, S.stamps ? S.stamps + 8 * static_cast<size_t>(K) : nullptr, S.index_form ? 1 : 0,
                     static_cast<const unsigned*>(nullptr));
  // LDS of k1_finalize: 3 C words of per-cell state + 5 words per point of a bucket that fits (at most kK1LdsCap points: eight
  // per thread, held in registers); bigger buckets (clustered data) go through their slices of the global scratch.
  // The kernel's registers allow three blocks per CU, so a pass gets what a third of the CU's LDS holds (the host does not
  // know the fullest bucket: on clustered scenes it has three times the mean, and every pass it needs beyond the first reads
  // and ranks the whole bucket again) -- or half / all of it where the per-cell state of a big C leaves less than 1024 points.
  // LDS: 3 C words of per-cell state, 3 words per point, 5 rows of wmax u16 counters (+ ~9 KB static: wave_rank's mask tables)
  const int wmax = std::min(C, 1024);
  auto fin_lds = [&](int cap) { return (static_cast<size_t>(3) * C + 3 * static_cast<size_t>(cap)) * sizeof(unsigned) + 5 * static_cast<size_t>(wmax) * 2; };
  int lds_cap = 256;
  for (const size_t budget : {static_cast<size_t>(44 * 1024), static_cast<size_t>(70 * 1024), kK1MaxDynamicLds}) {
    int cap = kK1LdsCap;
    while (cap > 256 && fin_lds(cap) > budget) cap -= 256;
    lds_cap = cap;
    if (fin_lds(cap) <= budget && cap >= 1024) break;
  }
  static const int cap_env = [] { const char* v = getenv("NDT_K1_LDS_CAP"); return v ? std::max(256, atoi(v)) / 256 * 256 : 0; }();
  if (cap_env > 0) lds_cap = std::min(lds_cap, cap_env);  // (tests: small passes, so that ordinary clouds take the multi-pass and crowded-cell paths)
  if (fin_lds(lds_cap) > kK1MaxDynamicLds) return hipErrorInvalidValue;
  static bool once_f = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_finalize), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kK1MaxDynamicLds));
    return true;
  }();
  (void)once_f;
  hipLaunchKernelGGL(k1_finalize, dim3(K), dim3(kBlock), fin_lds(lds_cap), stream,
                     S.bpts, g, P.shift, K, C, min_pts, eig_ratio, lds_cap, wmax, S.bucket_base, sorted_idx, recs, centroids, lut,
                     S.bucket_base + K + 1, S.order, static_cast<unsigned>(n), S.stamps, S.index_form ? pts : nullptr);
  return hipGetLastError();
}

// K1 for small clouds in one launch (k1_small).  false: not for this cloud / plan (the caller takes the chain).
bool grid_build_small_applies(int n, const GridBuildPlan& P) {
  // every block scans the whole cloud: n x K cell computations.  Measured against the chain (tools/time_k1_forms.py): see NOTES.
  static const int small_max = [] { const char* v = getenv("NDT_K1_SMALL_MAX"); return v ? atoi(v) : 49152; }();
  return n > 0 && n <= small_max && P.n_buckets <= 512;
}
hipError_t launch_grid_build_small(const float4* pts, int n, int dense, const GridGeom& g, const GridBuildPlan& P, int min_pts,
                                   double eig_ratio, const GridBuildScratch& S, int* sorted_idx, VoxelRec* recs, VoxelSide* centroids,
                                   int* lut, unsigned* counts, hipStream_t stream) {
  const int K = P.n_buckets, C = P.cells_per_bucket;
  const int wmax = std::min(C, 1024);
  // lists: eight waves x list_cap points of 16 bytes (NDT_K1_SMALL_LIST: tests force the overflow path with a tiny capacity)
  static const int list_env = [] { const char* v = getenv("NDT_K1_SMALL_LIST"); return v ? std::max(1, atoi(v)) : 0; }();
  const int list_cap = list_env > 0 ? list_env : 384;
  const size_t list_bytes = static_cast<size_t>(kSmallWaves) * list_cap * sizeof(float4);
  auto fin_lds = [&](int cap) { return ((static_cast<size_t>(3) * C + 3 * static_cast<size_t>(cap)) * sizeof(unsigned) + 5 * static_cast<size_t>(wmax) * 2 + 15) / 16 * 16; };
  int lds_cap = kK1LdsCap;
  while (lds_cap > 256 && fin_lds(lds_cap) + list_bytes > kK1MaxDynamicLds) lds_cap -= 256;
  static const int cap_env = [] { const char* v = getenv("NDT_K1_LDS_CAP"); return v ? std::max(256, atoi(v)) / 256 * 256 : 0; }();
  if (cap_env > 0) lds_cap = std::min(lds_cap, cap_env);
  if (fin_lds(lds_cap) + list_bytes > kK1MaxDynamicLds) return hipErrorInvalidValue;
  // (NDT_K1_SMALL_FINISH=0: every bucket through k1_finalize_bucket -- the cross-check of the two finishes)
  static const int small_finish = [] { const char* v = getenv("NDT_K1_SMALL_FINISH"); return v ? std::max(0, std::min(kSmallFinish, atoi(v))) : kSmallFinish; }();
  static bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_small), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kK1MaxDynamicLds));
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL(k1_small, dim3(K), dim3(kSmallThreads), fin_lds(lds_cap) + list_bytes, stream, pts, n, dense, g, P.shift, K, C, min_pts, eig_ratio,
                     lds_cap, wmax, list_cap, static_cast<unsigned>(fin_lds(lds_cap) / sizeof(unsigned)), small_finish, S.bucket_base, S.bpts, sorted_idx, recs,
                     centroids, lut, S.bucket_base + K + 1, S.order, counts, S.stamps);
  return hipGetLastError();
}

// The voxel filter on the bucket front end (vf_finalize's header).  false: not for this grid (the caller takes the general chain).
bool filter_buckets_plan(long long n_cells, int n_points, GridBuildPlan& P) {
  if (!grid_build_plan(n_cells, n_points, P)) return false;
  return (P.shift & 255) == 3;  // a run of eight cells is a byte of the occupancy bitmap
}
size_t filter_buckets_bitmap_words(long long n_cells) { return static_cast<size_t>(((n_cells + 7) / 8 + 3) / 4 + 1); }
hipError_t launch_filter_buckets(const float4* pts, int n, int dense, const GridGeom& g, const GridBuildPlan& P, const GridBuildScratch& S,
                                 int* st_cell, float4* st_cent, unsigned* bitmap_words, unsigned* wprefix, unsigned* counts /* [0] binned, [1] voxels */,
                                 float4* out, hipStream_t stream) {
  const int K = P.n_buckets, C = P.cells_per_bucket;
  const size_t lds_scatter = (static_cast<size_t>(K) + 2) * sizeof(unsigned) +
                             static_cast<size_t>(K) * sizeof(unsigned short) + static_cast<size_t>(kK1Waves) * K * sizeof(unsigned short);
  if (lds_scatter > kK1MaxDynamicLds || P.n_blocks > kColGroups * kColRows) return hipErrorInvalidValue;
  static bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kK1MaxDynamicLds));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vf_finalize), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kK1MaxDynamicLds));
    return true;
  }();
  (void)once;
  unsigned* total = S.cntmat + static_cast<size_t>(P.n_blocks) * K;
  hipLaunchKernelGGL(k1_hist, dim3(P.n_blocks), dim3(kK1Threads), static_cast<size_t>(K) * sizeof(unsigned), stream, pts, n, dense, g, P.shift, K,
                     P.pts_per_block, S.cntmat, static_cast<int*>(nullptr), 0ll, static_cast<const unsigned*>(nullptr));
  hipLaunchKernelGGL(k1_colscan, dim3((K + kColCols - 1) / kColCols), dim3(kK1Threads), 0, stream, S.cntmat, P.n_blocks, K, total);
  hipLaunchKernelGGL(k1_scatter, dim3(P.n_blocks), dim3(kK1Threads), lds_scatter, stream, pts, n, dense, g, P.shift, K, P.pts_per_block, S.cntmat,
                     total, S.bucket_base, S.bpts, counts, static_cast<unsigned long long*>(nullptr), 0, static_cast<const unsigned*>(nullptr));
  const int wmax = std::min(C, 1024);
  auto fin_lds = [&](int cap) { return (static_cast<size_t>(3) * C + 3 * static_cast<size_t>(cap)) * sizeof(unsigned) + 5 * static_cast<size_t>(wmax) * 2; };
  int lds_cap = 256;
  for (const size_t budget : {static_cast<size_t>(44 * 1024), static_cast<size_t>(70 * 1024), kK1MaxDynamicLds}) {
    int cap = kK1LdsCap;
    while (cap > 256 && fin_lds(cap) > budget) cap -= 256;
    lds_cap = cap;
    if (fin_lds(cap) <= budget && cap >= 1024) break;
  }
  static const int cap_env = [] { const char* v = getenv("NDT_K1_LDS_CAP"); return v ? std::max(256, atoi(v)) / 256 * 256 : 0; }();
  if (cap_env > 0) lds_cap = std::min(lds_cap, cap_env);
  if (fin_lds(lds_cap) > kK1MaxDynamicLds) return hipErrorInvalidValue;
  hipLaunchKernelGGL(vf_finalize, dim3(K), dim3(kBlock), fin_lds(lds_cap), stream, S.bpts, g, P.shift, K, C, lds_cap, wmax, S.bucket_base, st_cell, st_cent,
                     reinterpret_cast<unsigned char*>(bitmap_words), S.bucket_base + K + 1, S.order, static_cast<unsigned>(n));
  const int n_words = static_cast<int>(((g.n_cells + 7) / 8 + 3) / 4);
  const int n_chunks = (n_words + kVfChunkWords - 1) / kVfChunkWords;
  hipLaunchKernelGGL(vf_bitmap_prefix, dim3(n_chunks), dim3(kBlock), 0, stream, bitmap_words, n_words, wprefix, counts);
  hipLaunchKernelGGL(vf_place, dim3(std::max(1, std::min(2048, (n + kBlock - 1) / kBlock))), dim3(kBlock), 0, stream, st_cell, st_cent, counts, bitmap_words,
                     wprefix, out);
  return hipGetLastError();
}

// ---- a cloud in lattice-cell order by stable radix passes (order_range) ---------------------------------------------------
// The points sorted by their cell of the lattice g (x fastest), the points of a cell in the order they came in, non-finite
// points dropped: least-significant-digit-first passes of K1's order-preserving front end (k1_hist / k1_colscan / k1_scatter
// with bucket = digit: K1Deal's (cell >> rb) mod K is a digit when K is a power of two).  Two passes of <= 11 bits for up to
// 4 M cells, 32 bytes read and 16 written per point and pass -- instead of a rank per point by a returning atomic, a scatter
// of indices, a sort of every crowded cell back into point order and a gather.  counts: [passes] words, the points kept;
// counts[passes - 1] is the output's size.  tmp: n points (unused by a single pass); scratch as for a grid build.
int order_radix_passes(long long n_cells) {
  int bits = 1;
  while ((1ll << bits) < n_cells) bits++;
  return (bits + 10) / 11;
}
size_t order_radix_cntmat_words(long long n_cells, int n, int* n_blocks_out, int* ppb_out, int* digit_bits_out) {
  int bits = 1;
  while ((1ll << bits) < n_cells) bits++;
  const int passes = (bits + 10) / 11, db = (bits + passes - 1) / passes;
  long long ppb = std::max(512, std::min(kK1Round, pow2_ceil((n + 511) / 512)));
  while ((n + ppb - 1) / ppb > kColGroups * kColRows) ppb += kK1Round;
  const int nb = static_cast<int>(((n + ppb - 1) / ppb + 7) / 8 * 8);
  if (n_blocks_out) *n_blocks_out = nb;
  if (ppb_out) *ppb_out = static_cast<int>(ppb);
  if (digit_bits_out) *digit_bits_out = db;
  return (static_cast<size_t>(nb) + 1) * (static_cast<size_t>(1) << db);
}
hipError_t launch_order_radix(const float4* pts, int n, const GridGeom& g, unsigned* cntmat, unsigned* bucket_base, float4* tmp, float4* out,
                              unsigned* counts, hipStream_t stream) {
  int nb = 0, ppb = 0, db = 0;
  (void)order_radix_cntmat_words(g.n_cells, n, &nb, &ppb, &db);
  const int passes = order_radix_passes(g.n_cells), K = 1 << db;
  const size_t lds_scatter = (static_cast<size_t>(K) + 2) * sizeof(unsigned) + static_cast<size_t>(K) * sizeof(unsigned short) +
                             static_cast<size_t>(kK1Waves) * K * sizeof(unsigned short);
  if (passes > 3 || lds_scatter > kK1MaxDynamicLds || nb > kColGroups * kColRows) return hipErrorInvalidValue;
  static bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kK1MaxDynamicLds));
    return true;
  }();
  (void)once;
  unsigned* total = cntmat + static_cast<size_t>(nb) * K;
  const float4* in = pts;
  for (int p = 0; p < passes; p++) {
    // (an odd number of passes ends in `out` when it starts there)
    float4* dst = ((passes - p) & 1) ? out : tmp;
    const int map = (p * db) | (K << 8);
    const unsigned* n_dev = p ? counts + (p - 1) : nullptr;
    hipLaunchKernelGGL(k1_hist, dim3(nb), dim3(kK1Threads), static_cast<size_t>(K) * sizeof(unsigned), stream, in, n, 0, g, map, K, ppb, cntmat,
                       static_cast<int*>(nullptr), 0ll, n_dev);
    hipLaunchKernelGGL(k1_colscan, dim3((K + kColCols - 1) / kColCols), dim3(kK1Threads), 0, stream, cntmat, nb, K, total);
    hipLaunchKernelGGL(k1_scatter, dim3(nb), dim3(kK1Threads), lds_scatter, stream, in, n, 0, g, map, K, ppb, cntmat, total, bucket_base, dst, counts + p,
                       static_cast<unsigned long long*>(nullptr), 0, n_dev);
    in = dst;
  }
  return hipGetLastError();
}

// tile of the record compaction: 256 x ITEMS table entries, ITEMS in {1, 2, 4, 8} so that a table of a few 100 k cells still
// gives every CU a block (64 blocks of 2048 records each were 18 of the build's 125 us at 1 M points)
constexpr int kRcPrefixTiles = 2048;
static int rc_items(long long lut_cells) {
  int items = 1;
  while (items < 8 && lut_cells / (static_cast<long long>(kBlock) * items) >= kRcPrefixTiles) items <<= 1;
  return items;
}
size_t record_compaction_tiles(long long lut_cells) {
  const long long tile = static_cast<long long>(kBlock) * rc_items(lut_cells);
  return static_cast<size_t>((lut_cells + tile - 1) / tile);
}
template <int ITEMS>
static void launch_rc(int n_tiles, hipStream_t stream, int* lut, long long lut_cells, unsigned* tile_sums, const VoxelRec* recs_in,
                      const VoxelSide* cent_in, VoxelRec* recs_out, VoxelSide* cent_out) {
  hipLaunchKernelGGL((k_rc_count<ITEMS>), dim3(n_tiles), dim3(kBlock), 0, stream, lut, lut_cells, tile_sums);
  if (n_tiles > kRcPrefixTiles) {
    hipLaunchKernelGGL(k_rc_scan, dim3(1), dim3(kBlock), 0, stream, tile_sums, n_tiles);
    hipLaunchKernelGGL((k_rc_apply<ITEMS, true>), dim3(n_tiles), dim3(kBlock), 0, stream, lut, lut_cells, tile_sums, recs_in, cent_in, recs_out, cent_out);
  } else {
    hipLaunchKernelGGL((k_rc_apply<ITEMS, false>), dim3(n_tiles), dim3(kBlock), 0, stream, lut, lut_cells, tile_sums, recs_in, cent_in, recs_out, cent_out);
  }
}
hipError_t launch_compact_records(int* lut, long long lut_cells, const VoxelRec* recs_in, const VoxelSide* cent_in, VoxelRec* recs_out,
                                  VoxelSide* cent_out, unsigned* tile_sums, hipStream_t stream) {
  const int n_tiles = static_cast<int>(record_compaction_tiles(lut_cells));
  switch (rc_items(lut_cells)) {
    case 1: launch_rc<1>(n_tiles, stream, lut, lut_cells, tile_sums, recs_in, cent_in, recs_out, cent_out); break;
    case 2: launch_rc<2>(n_tiles, stream, lut, lut_cells, tile_sums, recs_in, cent_in, recs_out, cent_out); break;
    case 4: launch_rc<4>(n_tiles, stream, lut, lut_cells, tile_sums, recs_in, cent_in, recs_out, cent_out); break;
    default: launch_rc<8>(n_tiles, stream, lut, lut_cells, tile_sums, recs_in, cent_in, recs_out, cent_out); break;
  }
  return hipGetLastError();
}

hipError_t launch_grid_leaves(const GridGeom& g, const GridBuildPlan& P, int min_pts, const float4* bpts, const unsigned* bucket_base,
                              unsigned* scratch /* 4 K + 4 words */, int* leaf_cell, unsigned* leaf_start, int* leaf_count, int* leaf_rec,
                              unsigned* counts, const int* lut, hipStream_t stream, const float4* cloud) {
  const int K = P.n_buckets, C = P.cells_per_bucket;
  unsigned* ticket = scratch;
  unsigned* tot = scratch + 2;
  unsigned* occ_base = tot + 2 * K;
  unsigned* cand_base = occ_base + (K + 1);
  hipError_t e = hipMemsetAsync(ticket, 0, 2 * sizeof(unsigned), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k1_count, dim3(K), dim3(kBlock), static_cast<size_t>(C) * sizeof(unsigned), stream, bpts, g, P.shift, K, C,
                     static_cast<unsigned>(min_pts), bucket_base, tot, ticket, occ_base, cand_base, counts, cloud);
  hipLaunchKernelGGL(k1_leaves, dim3(K), dim3(kBlock), static_cast<size_t>(C) * sizeof(unsigned), stream, bpts, g, P.shift, C, min_pts,
                     bucket_base, occ_base, leaf_cell, leaf_start, leaf_count, leaf_rec, lut, cloud);
  return hipGetLastError();
}

hipError_t launch_sort_gather(const float4* pts, const unsigned* leaf_start, const int* leaf_count, int n_leaves,
                              int* sorted_idx, float4* out, hipStream_t stream, const unsigned* d_totals) {
  if (n_leaves == 0) return hipSuccess;
  // crowded cells first, one wave each (sorted and gathered straight into `out`); k_sort_gather takes the rest
  hipLaunchKernelGGL(k_presort_large, dim3(presort_grid(n_leaves)), dim3(kBlock), 0, stream, pts, leaf_start, leaf_count, n_leaves, d_totals,
                     sorted_idx, out, presort_chunk(n_leaves));
  hipLaunchKernelGGL(k_sort_gather, dim3((n_leaves + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, pts, leaf_start,
                     leaf_count, n_leaves, d_totals, sorted_idx, out);
  return hipGetLastError();
}

hipError_t launch_voxel_centroids(const float4* pts, const unsigned* leaf_start, const int* leaf_count, int n_leaves,
                                  int* sorted_idx, float4* out, hipStream_t stream, const unsigned* d_totals, float4* d_big_pts) {
  if (n_leaves == 0) return hipSuccess;
  if (d_big_pts)
    hipLaunchKernelGGL(k_presort_large, dim3(presort_grid(n_leaves)), dim3(kBlock), 0, stream, pts, leaf_start, leaf_count, n_leaves, d_totals,
                       sorted_idx, d_big_pts, presort_chunk(n_leaves));
  hipLaunchKernelGGL(k_voxel_centroids, dim3((n_leaves + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, pts, leaf_start,
                     leaf_count, n_leaves, d_totals, sorted_idx, out, d_big_pts);
  return hipGetLastError();
}

// small host clouds: the dense float4 records straight out of the page-locked slot the host repacked them into (a kernel
// reading 256 KB over PCIe is done before a DMA engine has started: measured 8 vs 20 us at 16 k points)
__global__ __launch_bounds__(kBlock) void k_copy_records(const float4* __restrict__ src, float4* __restrict__ dst, int n) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) dst[i] = src[i];
}
hipError_t launch_copy_records(const float4* src_host_pinned, float4* dst, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_copy_records, dim3(grid_for(n, 1024)), dim3(kBlock), 0, stream, src_host_pinned, dst, n);
  return hipGetLastError();
}

hipError_t launch_scan_bboxes(const float4* pts, const int* d_scan_off, int n_scans, int max_scan_points, int* d_out, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan_bboxes, dim3(grid_for(max_scan_points, 64), n_scans), dim3(kBlock), 0, stream, pts, d_scan_off, d_out);
  return hipGetLastError();
}

float scan_bbox_decode(int v) {
  const int b = v ^ ((v >> 31) & 0x7fffffff);
  float f;
  std::memcpy(&f, &b, sizeof(f));
  return f;
}

hipError_t launch_count_batch(const float4* pts, const int* d_scan_off, int n_scans, int max_scan_points, const ScanLattice* d_lat,
                              int* d_key, unsigned* d_rank, unsigned* d_cell_count, hipStream_t stream) {
  hipLaunchKernelGGL(k_count_batch, dim3(grid_for(max_scan_points, 256), n_scans), dim3(kBlock), 0, stream, pts, d_scan_off, d_lat,
                     d_key, d_rank, d_cell_count);
  return hipGetLastError();
}

hipError_t launch_pick(const unsigned* cell_count, const long long* d_bases, unsigned* d_out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(k_pick, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, cell_count, d_bases, d_out, n);
  return hipGetLastError();
}

int scan_tiles(long long n_cells) { return static_cast<int>((n_cells + kScanTile - 1) / kScanTile); }

}  // namespace ndt
