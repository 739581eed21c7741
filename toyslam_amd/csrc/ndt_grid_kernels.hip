// ndt_grid_kernels.hip -- K1: everything that turns a cloud into a voxel structure (gfx950, wave64).
//
//  target voxel grid  (VoxelGridCovariance::applyFilter, voxel_grid_covariance_omp_impl.hpp:48-370)
//    general chain:  bbox -> per-cell count -> 3-phase exclusive scan over cells (leaf ordinals, segment offsets, record
//                    ordinals, LUT init) -> counting-sort scatter of point indices -> k_presort_large (crowded voxels) ->
//                    k_finalize
//    bucket form:    k1_hist -> k1_scatter -> k1_finalize (3 launches, LDS histograms, no per-point global
//                    atomics; crowded cells summed by lane teams), k1_count / k1_leaves on demand
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
      rec.p0[0] = c00; rec.p0[1] = c01;
      rec.p1[0] = c01; rec.p1[1] = c11;
      rec.p2[0] = c02; rec.p2[1] = c12;
      rec.p3[0] = c11; rec.p3[1] = c22;
      rec.n = cnt;
      rec.pad = 0;
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
// staged through LDS.  Round 3: the whole chain is ORDER-PRESERVING -- a bucket holds its points in ascending point index,
// and so does every cell after k1_finalize's LDS sort -- because the reference adds a voxel's points in index order
// (_impl.hpp:233-244) and a sort by index per cell was a third of the old k1_finalize:
//   k1_hist     per block of points: LDS histogram over the buckets -> the block's row of the count matrix [blocks][K]
//               (plain stores; round 2 claimed a run per (block, bucket) with a returning global atomic: 250 k of them at
//               1 M points = 11 us of memory-side atomics); also clears the look-up table
//   k1_scatter  the same blocks: column sums of the count matrix (rows before mine = my base inside every bucket, all rows
//               = the bucket sizes), then a STABLE split of the block's points over the buckets: ranks inside a 64-point
//               chunk from ballots (wave_rank), per-wave running counts in LDS -- bucket k then holds the points of block 0,
//               block 1, ... each in index order
//   k1_finalize one block per bucket: the same stable ranking by CELL into LDS (no atomics, no sort), then one thread per
//               cell: sums in ascending point order (bit-identical to the reference's sequential pass), second pass of
//               applyFilter -> record, centroid, look-up table slot, sorted_idx
//   k1_count / k1_leaves  on demand: occupied / candidate counts, leaf arrays
// Points are read three times (k1_hist, k1_scatter: the second read comes from the Infinity Cache) and written once.
// ---------------------------------------------------------------------------
// Cell <-> (bucket, local cell).  Buckets are NOT ranges of the linear cell index: a clustered scene (a ground plane) would
// fill a few of those with many times the mean and leave the rest empty.  Runs of 2^rb consecutive cells (neighbours in x,
// whose points a spatially ordered cloud delivers together) are dealt round-robin to the K = 2^kb buckets:
//   bucket = (cell >> rb) mod K,   local = ((cell >> rb) / K) << rb | (cell mod 2^rb)
// `map` packs rb (bits 0-7) and kb (bits 8-15).
__device__ __forceinline__ int k1_bucket(int cell, int map) { return (cell >> (map & 255)) & ((1 << (map >> 8)) - 1); }
__device__ __forceinline__ int k1_local(int cell, int map) {
  const int rb = map & 255, kb = map >> 8;
  return ((cell >> (rb + kb)) << rb) | (cell & ((1 << rb) - 1));
}
__device__ __forceinline__ int k1_cell(int bucket, int local, int map) {
  const int rb = map & 255, kb = map >> 8;
  return ((((local >> rb) << kb) | bucket) << rb) | (local & ((1 << rb) - 1));
}

constexpr int kK1Threads = 512;              // k1_hist / k1_scatter / k1_finalize
constexpr int kK1Waves = kK1Threads / kWave;  // 8
constexpr int kK1Round = 8 * kK1Threads;     // points one block ranks per round in k1_scatter (eight 64-point chunks per wave)

__device__ __forceinline__ int key_of(const GridGeom& g, const float4& p, int dense) {
  int c = -1;
  if (dense || finite3(p.x, p.y, p.z)) {
    c = build_cell(g, p.x, p.y, p.z);
    if (c < 0 || static_cast<long long>(c) >= g.n_cells) c = -1;  // inside the box by construction; NaN / garbage guard
  }
  return c;
}

// exclusive scan of n (<= 16 * nthreads) u32 values src[] -> dst[] by one block; dst[n] = total.  lds: nthreads / 64 words
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
// key: the lanes with equal keys are found with one ballot per key bit; the lowest of them (the leader) bumps the wave's
// counter row in LDS by their number and hands the old value to the others.  Returns old count + number of equal-key
// lanes below this one -- the position of the lane's element among ALL elements of that key the wave has ranked so far,
// in the order the wave met them.  `row` is this wave's private row of counters (u16, one per key).  Invalid lanes (key
// ignored) take part in the ballots only.
__device__ __forceinline__ unsigned wave_rank(int key, bool valid, unsigned short* row, int bits) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 13; b++) {
    if (b < bits) {
      const bool bit = ((key >> b) & 1) != 0;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
  }
  const unsigned below = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(peers >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(peers), 0u));
  unsigned old = 0;
  if (valid && below == 0) {
    old = row[key];
    row[key] = static_cast<unsigned short>(old + static_cast<unsigned>(__popcll(peers)));
  }
  const int leader = valid ? (__ffsll(static_cast<long long>(peers)) - 1) : 0;
  old = __shfl(old, leader, kWave);
  return old + below;
}

// rows = blocks of points, columns = buckets: the block's row of bucket counts (plain stores); the look-up table is cleared
// on the side (nothing reads it before k1_finalize)
__global__ __launch_bounds__(kK1Threads) void k1_hist(const float4* __restrict__ pts, int n, int dense, GridGeom g, int map, int K,
                                                      int ppb, unsigned* __restrict__ cntmat, int* __restrict__ lut, long long lut_cells) {
  extern __shared__ unsigned k1_lds[];
  unsigned* h = k1_lds;
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
  const int lo = blockIdx.x * ppb, hi = min(n, lo + ppb);
  for (int base = lo + threadIdx.x; base < hi; base += 8 * kK1Threads) {  // eight loads in flight per thread
    float4 p[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = base + u * kK1Threads;
      p[u] = (i < hi) ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int c = (base + u * kK1Threads < hi) ? key_of(g, p[u], dense) : -1;
      if (c >= 0) atomicAdd(&h[k1_bucket(c, map)], 1u);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += kK1Threads) cntmat[static_cast<size_t>(blockIdx.x) * K + k] = h[k];
}

// The count matrix, column by column: cntmat[b][k] <- points of bucket k in the blocks BEFORE b (exclusive prefix down the
// rows, in place), total[k] <- the bucket's size.  One pass over the matrix instead of every block of k1_scatter summing
// all the rows above its own (at 10 M points: 489 x 4096 entries, read 489 times).  A block takes 16 columns; its 512
// threads are 32 row groups x 16 columns, a thread keeps its rows (at most kColRows) in registers between the two sweeps.
constexpr int kColCols = 16, kColGroups = kK1Threads / kColCols, kColRows = 16;  // B <= kColGroups * kColRows = 512 rows
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
  for (int g = 0; g < kColGroups; g++) {
    const unsigned t = s_part[g][c];
    if (g < rg) base += t;
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

// phase stamps (development aid, NDT_K1_STAMPS=1): thread 0 of every block leaves the shader clock at each phase boundary
__device__ __forceinline__ void k1_stamp(unsigned long long* st, int n_phases, int phase) {
  if (st && threadIdx.x == 0) st[static_cast<size_t>(blockIdx.x) * n_phases + phase] = stamp();
}
constexpr int kK1StampPhases = 12;

// LDS of k1_scatter: cursor[K + 1] u32, part_lt[K] u32, part_all[K] u32, tab[kK1Waves][K] u16
__global__ __launch_bounds__(kK1Threads) void k1_scatter(const float4* __restrict__ pts, int n, int dense, GridGeom g, int map, int K,
                                                         int ppb, const unsigned* __restrict__ cntmat, const unsigned* __restrict__ total,
                                                         unsigned* __restrict__ bucket_base, float4* __restrict__ bpts,
                                                         unsigned* __restrict__ counts, unsigned long long* __restrict__ st) {
  extern __shared__ unsigned k1_lds[];
  __shared__ unsigned s_scan[kK1Waves];
  unsigned* cursor = k1_lds;            // [K + 1]
  unsigned* part_lt = cursor + K + 1;   // [K]: points of this bucket in the blocks before this one
  unsigned* part_all = part_lt + K;     // [K]: bucket sizes; later the current round's counts
  unsigned short* tab = reinterpret_cast<unsigned short*>(part_all + K);  // [kK1Waves][K]
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int b = blockIdx.x;
  const int kbits = map >> 8;
  k1_stamp(st, kK1StampPhases, 0);
  for (int k = threadIdx.x; k < K; k += kK1Threads) {
    part_lt[k] = cntmat[static_cast<size_t>(b) * K + k];
    part_all[k] = total[k];
  }
  __syncthreads();
  k1_stamp(st, kK1StampPhases, 1);
  // bucket bases = exclusive scan of the bucket sizes, by every block for itself; block 0 keeps them for k1_finalize
  block_scan_array(part_all, cursor, K, kK1Threads, s_scan, false);
  __syncthreads();
  if (b == 0) {
    for (int k = threadIdx.x; k <= K; k += kK1Threads) bucket_base[k] = cursor[k];
    if (threadIdx.x == 0) counts[0] = cursor[K];  // points binned
  }
  for (int k = threadIdx.x; k < K; k += kK1Threads) cursor[k] += part_lt[k];
  for (int i = threadIdx.x; i < kK1Waves * K / 2; i += kK1Threads) reinterpret_cast<unsigned*>(tab)[i] = 0u;
  __syncthreads();
  k1_stamp(st, kK1StampPhases, 2);
  // ---- stable split: rounds of 4096 points; wave w ranks the eight 64-point chunks [w * 512, (w + 1) * 512) of the round
  const int lo = b * ppb, hi = min(n, lo + ppb);
  unsigned short* row = tab + wave * K;
  for (int r0 = lo; r0 < hi; r0 += kK1Round) {
    float4 p[8];
    int key[8];
    unsigned rk[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = r0 + wave * (8 * kWave) + u * kWave + lane;
      p[u] = (i < hi) ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = r0 + wave * (8 * kWave) + u * kWave + lane;
      const int c = (i < hi) ? key_of(g, p[u], dense) : -1;
      key[u] = (c >= 0) ? k1_bucket(c, map) : -1;
      rk[u] = wave_rank(key[u], key[u] >= 0, row, kbits);
    }
    __syncthreads();
    if (r0 == lo) k1_stamp(st, kK1StampPhases, 3);
    // per bucket: the waves' counts -> exclusive prefix over the waves (in place), the round's total -> cursor afterwards
    for (int k = threadIdx.x; k < K; k += kK1Threads) {
      unsigned s = 0;
#pragma unroll
      for (int w = 0; w < kK1Waves; w++) {
        const unsigned t = tab[w * K + k];
        tab[w * K + k] = static_cast<unsigned short>(s);
        s += t;
      }
      part_all[k] = s;
    }
    __syncthreads();
    if (r0 == lo) k1_stamp(st, kK1StampPhases, 4);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (key[u] >= 0) {
        const int i = r0 + wave * (8 * kWave) + u * kWave + lane;
        bpts[cursor[key[u]] + row[key[u]] + rk[u]] = make_float4(p[u].x, p[u].y, p[u].z, __int_as_float(i));
      }
    }
    __syncthreads();
    if (r0 == lo) k1_stamp(st, kK1StampPhases, 5);
    if (r0 + kK1Round < hi) {  // (uniform) another round: advance the cursors, clear the counters
      for (int k = threadIdx.x; k < K; k += kK1Threads) cursor[k] += part_all[k];
      for (int i = threadIdx.x; i < kK1Waves * K / 2; i += kK1Threads) reinterpret_cast<unsigned*>(tab)[i] = 0u;
      __syncthreads();
    }
  }
}

// per-bucket cell histogram in LDS (cnt[C], zeroed here)
__device__ __forceinline__ void k1_cell_histogram(const float4* __restrict__ bpts, unsigned bb, unsigned be, const GridGeom& g,
                                                  int map, int C, unsigned* cnt) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) cnt[c] = 0;
  __syncthreads();
  for (unsigned j = bb + threadIdx.x; j < be; j += blockDim.x) {
    const float4 p = bpts[j];
    atomicAdd(&cnt[k1_local(build_cell(g, p.x, p.y, p.z), map)], 1u);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// k1_finalize: one block per bucket.  The bucket holds its points in ascending point index (k1_scatter is order-preserving);
// a STABLE sort by local cell -- LSD radix sort, digits of dbits <= 11 bits, ranks from ballots (wave_rank) -- therefore
// leaves every cell's points in ascending point index, the order the reference adds them in, with no comparison sort at all.
// Cells are the runs of equal keys of the sorted sequence: nothing here is sized by the number of cells a bucket may hold.
//   fast path  the bucket fits the block's LDS (PT x 512 points): points stay in registers while they are ranked, one or two
//              digits, one scatter of x / y / z / key into LDS;
//   slow path  (a clustered cloud: more points than that) the same radix sort streamed in rounds through global scratch
//              (key, position-in-bucket pairs; sweep 1 counts a digit, sweep 2 places), the sums gather through the sorted
//              positions.  Runs longer than a round are summed by a lane team straight from global memory.
// Per run: sums in ascending point order (a team of 16 lanes, a lane per accumulator, for runs of more than 32 points),
// finish_voxel -> record, side sector, look-up table slot; the run itself (cell, start, count) goes to leaf_slots[bb + run
// ordinal] for the leaf pass (k1_leaves, on demand).
// ---------------------------------------------------------------------------
constexpr int kTeamCell = 32, kTeamLanes = 16;
constexpr size_t kK1MaxDynamicLds = 148 * 1024;  // of the CU's 160 KB; k1_finalize also holds ~10 KB of static team sums
constexpr int kMaxTeams = 96;   // long runs one round can hold in its team list (the rest are summed by their own thread)

struct K1Team {
  double s64[9];
  float s32[3];
  unsigned beg, cnt;
};

// sums of the points at sorted positions [beg, beg + cnt) by the 16-lane team `tl` belongs to: lane 0-2 the mean sums, 3-8
// the products xx xy xz yy yz zz (seeded Identity, voxel_grid_covariance_omp.h:107), 9-11 the f32 centroid sums
template <class Pts>
__device__ __forceinline__ void team_sum(const Pts& P, unsigned beg, unsigned cnt, int tl, K1Team* out) {
#pragma clang fp contract(off)
  const int ia = (tl < 3) ? tl : (tl < 6) ? 0 : (tl < 8) ? 1 : (tl == 8) ? 2 : (tl < 12) ? tl - 9 : 0;
  const int ib = (tl == 3) ? 0 : (tl == 4 || tl == 6) ? 1 : (tl == 5 || tl == 7 || tl == 8) ? 2 : -1;
  double acc = (tl == 3 || tl == 6 || tl == 8) ? 1.0 : 0.0;
  float acc32 = 0.f;
  unsigned i = 0;
  for (; i + 4 <= cnt; i += 4) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      a[u] = P.coord(beg + i + u, ia);
      b[u] = (ib < 0) ? 1.0f : P.coord(beg + i + u, ib);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const double prod = static_cast<double>(a[u]) * static_cast<double>(b[u]);  // (x * 1.0 is exact)
      acc += prod;
      acc32 += a[u];
    }
  }
  for (; i < cnt; i++) {
    const float a = P.coord(beg + i, ia), b = (ib < 0) ? 1.0f : P.coord(beg + i, ib);
    const double prod = static_cast<double>(a) * static_cast<double>(b);
    acc += prod;
    acc32 += a;
  }
  if (tl < 9) out->s64[tl] = acc;
  else if (tl < 12) out->s32[tl - 9] = acc32;
}

// sorted points in LDS (fast path) ...
struct K1LdsPts {
  const float* x;
  const float* y;
  const float* z;
  __device__ __forceinline__ float coord(unsigned pos, int a) const { return (a == 0 ? x : a == 1 ? y : z)[pos]; }
  __device__ __forceinline__ void get(unsigned pos, float& px, float& py, float& pz) const { px = x[pos]; py = y[pos]; pz = z[pos]; }
};
// ... or reached through the sorted positions in global memory (slow path): pos -> position in the bucket -> point
struct K1GlobalPts {
  const float4* b;      // the bucket's points
  const unsigned* jpos;  // sorted position -> position in the bucket (offset by `off`)
  unsigned off;
  __device__ __forceinline__ float coord(unsigned pos, int a) const {
    const float4 p = b[jpos[off + pos]];
    return a == 0 ? p.x : a == 1 ? p.y : p.z;
  }
  __device__ __forceinline__ void get(unsigned pos, float& px, float& py, float& pz) const {
    const float4 p = b[jpos[off + pos]];
    px = p.x; py = p.y; pz = p.z;
  }
};

// The runs [runstart[r], runstart[r + 1]) of one round, r < n_runs: long ones by lane teams, then a thread per run.
// key(r) = local cell of run r; slot0 = position of the round's first point in the bucket order (for the record slot and
// the leaf entry); run0 = ordinal of the round's first run inside the bucket.  Returns this thread's count of valid voxels.
template <class Pts, class KeyFn>
__device__ __forceinline__ unsigned k1_process_runs(const Pts& P, const KeyFn& key, const unsigned short* runstart, int n_runs, unsigned bb,
                                                    unsigned slot0, unsigned run0, int bucket, int map, int min_pts, double eig_ratio,
                                                    const GridGeom& g, VoxelRec* __restrict__ recs, VoxelSide* __restrict__ centroids,
                                                    int* __restrict__ lut, uint4* __restrict__ leaf_slots, K1Team* s_team,
                                                    unsigned short* s_team_of /* [runs]: team slot + 1, or 0 */, int* s_nteam,
                                                    unsigned long long* st = nullptr) {
  if (threadIdx.x == 0) *s_nteam = 0;
  __syncthreads();
  for (int r = threadIdx.x; r < n_runs; r += kK1Threads) {
    const int n_c = static_cast<int>(runstart[r + 1]) - static_cast<int>(runstart[r]);
    unsigned short t = 0;
    if (n_c > kTeamCell && n_c >= min_pts) {
      const int slot = atomicAdd(s_nteam, 1);
      if (slot < kMaxTeams) {
        s_team[slot].beg = runstart[r];
        s_team[slot].cnt = static_cast<unsigned>(n_c);
        t = static_cast<unsigned short>(slot + 1);
      }
    }
    s_team_of[r] = t;
  }
  __syncthreads();
  {
    const int n_team = min(*s_nteam, kMaxTeams);
    const int tl = threadIdx.x & (kTeamLanes - 1), team = threadIdx.x / kTeamLanes;
    for (int s = team; s < n_team; s += kK1Threads / kTeamLanes) team_sum(P, s_team[s].beg, s_team[s].cnt, tl, &s_team[s]);
  }
  __syncthreads();
  k1_stamp(st, kK1StampPhases, 6);
  const FinalizeDump nodump{nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned n_ok = 0;
  for (int r = threadIdx.x; r < n_runs; r += kK1Threads) {
    const unsigned beg = runstart[r];
    const int n_c = static_cast<int>(runstart[r + 1]) - static_cast<int>(beg);
    const int local = key(r, beg);
    const int cell = k1_cell(bucket, local, map);
    leaf_slots[bb + run0 + r] = make_uint4(static_cast<unsigned>(cell), bb + slot0 + beg, static_cast<unsigned>(n_c), 0u);
    if (n_c < min_pts) continue;  // (cells with fewer points get no record: the reference skips them at look-up, _impl.hpp:395)
    // record slot: candidates' segments start at least min_pts apart, so start / min_pts is unique per candidate --
    // no scan over the buckets is needed to number the records
    const int rec = static_cast<int>((bb + slot0 + beg) / static_cast<unsigned>(min_pts));
    VoxelSums S;
    const unsigned short t = s_team_of[r];
    if (t) {
      const K1Team& T = s_team[t - 1];
      S.sx = T.s64[0]; S.sy = T.s64[1]; S.sz = T.s64[2];
      S.cxx = T.s64[3]; S.cxy = T.s64[4]; S.cxz = T.s64[5];
      S.cyy = T.s64[6]; S.cyz = T.s64[7]; S.czz = T.s64[8];
      S.fx = T.s32[0]; S.fy = T.s32[1]; S.fz = T.s32[2];
    } else {
      int i = 0;
      for (; i + 4 <= n_c; i += 4) {  // ascending point order, contiguous; twelve reads in flight per step
        float x[4], y[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; u++) P.get(beg + i + u, x[u], y[u], z[u]);
#pragma unroll
        for (int u = 0; u < 4; u++) S.add(x[u], y[u], z[u]);
      }
      for (; i < n_c; i++) {
        float x, y, z;
        P.get(beg + i, x, y, z);
        S.add(x, y, z);
      }
    }
    n_ok += finish_voxel(S, n_c, 0, rec, cell, min_pts, eig_ratio, recs, centroids, lut, g, nodump) ? 1u : 0u;
  }
  return n_ok;
}

// exclusive scan of the D digit counts that the eight wave rows of `tab` hold: tab[w][d] <- count of digit d in the waves
// before w, dstart[d] <- positions before digit d (dstart[D] = total).  `s_scan`: kK1Waves words.
__device__ __forceinline__ void k1_digit_offsets(unsigned short* tab, unsigned* dstart, int D, unsigned* s_scan) {
  const int per = (D + kK1Threads - 1) / kK1Threads;  // D <= 2048: at most four digits per thread, consecutive
  const int lo = threadIdx.x * per, hi = min(D, lo + per);
  unsigned cnt[4] = {0, 0, 0, 0};
  unsigned sum = 0;
  for (int d = lo; d < hi; d++) {
    unsigned s = 0;
#pragma unroll
    for (int w = 0; w < kK1Waves; w++) {
      const unsigned t = tab[w * D + d];
      tab[w * D + d] = static_cast<unsigned short>(s);
      s += t;
    }
    cnt[d - lo] = s;
    sum += s;
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  unsigned inc = sum;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned a = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += a;
  }
  if (lane == kWave - 1) s_scan[wave] = inc;
  __syncthreads();
  unsigned base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kK1Waves; w++) {
    if (w < wave) base += s_scan[w];
    total += s_scan[w];
  }
  unsigned run = base + inc - sum;
  for (int d = lo; d < hi; d++) {
    dstart[d] = run;
    run += cnt[d - lo];
  }
  if (threadIdx.x == 0) dstart[D] = total;
  __syncthreads();
}

// heads of the runs of equal keys among the `n` sorted positions of one round -> runstart[0 .. n_runs], returns n_runs
// (uniform).  keyat(pos) reads the key at a sorted position.  PTR = positions per thread.
template <int PTR, class KeyAt>
__device__ __forceinline__ int k1_find_runs(const KeyAt& keyat, unsigned n, unsigned short* runstart, unsigned* s_scan, int* s_total) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  // wave w looks at the PTR consecutive chunks [w * PTR, (w + 1) * PTR) of 64 positions
  unsigned long long heads[PTR];
  unsigned mine = 0;
#pragma unroll
  for (int u = 0; u < PTR; u++) {
    const unsigned pos = static_cast<unsigned>((wave * PTR + u) * kWave + lane);
    const bool h = pos < n && (pos == 0 || keyat(pos) != keyat(pos - 1));
    heads[u] = __ballot(h);
    mine += static_cast<unsigned>(__popcll(heads[u]));  // (uniform per wave)
  }
  if (lane == 0) s_scan[wave] = mine;
  __syncthreads();
  unsigned base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kK1Waves; w++) {
    if (w < wave) base += s_scan[w];
    total += s_scan[w];
  }
#pragma unroll
  for (int u = 0; u < PTR; u++) {
    const unsigned pos = static_cast<unsigned>((wave * PTR + u) * kWave + lane);
    if ((heads[u] >> lane) & 1ull) {
      const unsigned below = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(heads[u] >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(heads[u]), 0u));
      runstart[base + below] = static_cast<unsigned short>(pos);
    }
    base += static_cast<unsigned>(__popcll(heads[u]));
  }
  if (threadIdx.x == 0) {
    runstart[total] = static_cast<unsigned short>(n);
    *s_total = static_cast<int>(total);
  }
  __syncthreads();
  return *s_total;
}

// PT: points per thread the fast path holds (capacity PT x 512); TWO: the local cell index needs two digits.
// LDS (dynamic): tab u16 [8][D] | dstart u32 [D + 2] | ox oy oz f32 [cap] (at least 2 D words: the slow path's histograms)
//                | skey u32 [cap] | runstart u16 [cap + 2] | team_of u16 [cap], or (TWO) item u32 [cap] (team_of on top) + posmap u16 [cap]
template <int PT, bool TWO>
__global__ __launch_bounds__(kK1Threads) void k1_finalize(const float4* __restrict__ bpts, GridGeom g, int map, int K, int cbits, int dbits,
                                                          int min_pts, double eig_ratio, const unsigned* __restrict__ bucket_base,
                                                          int* __restrict__ sorted_idx, VoxelRec* __restrict__ recs,
                                                          VoxelSide* __restrict__ centroids, int* __restrict__ lut,
                                                          unsigned* __restrict__ bucket_stat /* [K][4]: valid, occupied, candidate voxels, 0 */,
                                                          uint4* __restrict__ leaf_slots, unsigned* __restrict__ scratch /* 4 words per point */,
                                                          unsigned n_total, unsigned long long* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) unsigned char k1_smem[];
  __shared__ unsigned s_scan[kK1Waves];
  __shared__ int s_total, s_nteam;
  __shared__ unsigned s_u[4];
  __shared__ K1Team s_team[kMaxTeams];
  constexpr int cap = PT * kK1Threads;
  const int D = 1 << dbits;
  unsigned short* tab = reinterpret_cast<unsigned short*>(k1_smem);
  unsigned* dstart = reinterpret_cast<unsigned*>(tab + kK1Waves * D);
  float* ox = reinterpret_cast<float*>(dstart + D + 2);
  float* oy = ox + cap;
  float* oz = oy + cap;
  unsigned* skey = reinterpret_cast<unsigned*>(ox + max(3 * cap, 2 * D));
  unsigned short* runstart = reinterpret_cast<unsigned short*>(skey + cap);
  unsigned* item = reinterpret_cast<unsigned*>(runstart + cap + 2);        // TWO only: (high key bits, position) pairs ...
  unsigned short* posmap = reinterpret_cast<unsigned short*>(item + cap);  // ... and where each position ended up
  unsigned short* team_of = reinterpret_cast<unsigned short*>(item);       // (the pairs are dead by then)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int k = blockIdx.x;
  const unsigned bb = bucket_base[k], be = bucket_base[k + 1];
  const unsigned nb = be - bb;
  if (nb == 0) {  // empty bucket (uniform)
    if (threadIdx.x < 4) bucket_stat[4 * k + threadIdx.x] = 0u;
    return;
  }
  const float4* bp = bpts + bb;
  unsigned n_ok = 0, n_occ = 0, n_cand = 0;
  k1_stamp(st, kK1StampPhases, 0);
  const unsigned dmask = static_cast<unsigned>(D - 1);

  if (nb <= static_cast<unsigned>(cap)) {
    // =============================== fast path: everything in LDS ===============================
    for (int i = threadIdx.x; i < kK1Waves * D / 2; i += kK1Threads) reinterpret_cast<unsigned*>(tab)[i] = 0u;
    // wave w owns the consecutive chunks [w * per_wave, (w + 1) * per_wave) of 64 points: contiguous, ascending ranges
    const int per_wave = static_cast<int>(((nb + kWave - 1) / kWave + kK1Waves - 1) / kK1Waves);  // <= PT
    float4 p[PT];
    unsigned key[PT], rk[PT];
#pragma unroll
    for (int u = 0; u < PT; u++) {
      const unsigned j = static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
      p[u] = (u < per_wave && j < nb) ? bp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    k1_stamp(st, kK1StampPhases, 1);
    unsigned short* row = tab + wave * D;
#pragma unroll
    for (int u = 0; u < PT; u++) {
      const unsigned j = static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
      const bool valid = u < per_wave && j < nb;
      key[u] = valid ? static_cast<unsigned>(k1_local(build_cell(g, p[u].x, p[u].y, p[u].z), map)) : 0u;
      if (u < per_wave) rk[u] = wave_rank(static_cast<int>(key[u] & dmask), valid, row, dbits);  // (uniform condition)
    }
    __syncthreads();
    k1_stamp(st, kK1StampPhases, 2);
    k1_digit_offsets(tab, dstart, D, s_scan);
    k1_stamp(st, kK1StampPhases, 3);
    if (!TWO) {
#pragma unroll
      for (int u = 0; u < PT; u++) {
        const unsigned j = static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
        if (u < per_wave && j < nb) {
          const unsigned d = key[u] & dmask;
          const unsigned pos = dstart[d] + row[d] + rk[u];
          ox[pos] = p[u].x;
          oy[pos] = p[u].y;
          oz[pos] = p[u].z;
          skey[pos] = key[u];
          sorted_idx[bb + pos] = __float_as_int(p[u].w);
        }
      }
      __syncthreads();
    } else {
      // second digit: the (high key bits, position in bucket) pairs in first-digit order, ranked again in that order
#pragma unroll
      for (int u = 0; u < PT; u++) {
        const unsigned j = static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
        if (u < per_wave && j < nb) {
          const unsigned d = key[u] & dmask;
          item[dstart[d] + row[d] + rk[u]] = ((key[u] >> dbits) << 13) | j;
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < kK1Waves * D / 2; i += kK1Threads) reinterpret_cast<unsigned*>(tab)[i] = 0u;
      __syncthreads();
      unsigned it[PT], rk2[PT];
#pragma unroll
      for (int u = 0; u < PT; u++) {
        const unsigned s = static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
        const bool valid = u < per_wave && s < nb;
        it[u] = valid ? item[s] : 0u;
        if (u < per_wave) rk2[u] = wave_rank(static_cast<int>((it[u] >> 13) & dmask), valid, row, dbits);
      }
      __syncthreads();
      k1_digit_offsets(tab, dstart, D, s_scan);
#pragma unroll
      for (int u = 0; u < PT; u++) {
        const unsigned s = static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
        if (u < per_wave && s < nb) {
          const unsigned d = (it[u] >> 13) & dmask;
          posmap[it[u] & 8191u] = static_cast<unsigned short>(dstart[d] + row[d] + rk2[u]);
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < PT; u++) {
        const unsigned j = static_cast<unsigned>((wave * per_wave + u) * kWave + lane);
        if (u < per_wave && j < nb) {
          const unsigned pos = posmap[j];
          ox[pos] = p[u].x;
          oy[pos] = p[u].y;
          oz[pos] = p[u].z;
          skey[pos] = key[u];
          sorted_idx[bb + pos] = __float_as_int(p[u].w);
        }
      }
      __syncthreads();
    }
    k1_stamp(st, kK1StampPhases, 4);
    const int n_runs = k1_find_runs<PT>([&](unsigned pos) { return skey[pos]; }, nb, runstart, s_scan, &s_total);
    k1_stamp(st, kK1StampPhases, 5);
    const K1LdsPts P{ox, oy, oz};
    n_ok = k1_process_runs(P, [&](int, unsigned beg) { return static_cast<int>(skey[beg]); }, runstart, n_runs, bb, 0u, 0u, k, map, min_pts,
                           eig_ratio, g, recs, centroids, lut, leaf_slots, s_team, team_of, &s_nteam, st);
    k1_stamp(st, kK1StampPhases, 7);
    for (int r = threadIdx.x; r < n_runs; r += kK1Threads) n_cand += (static_cast<int>(runstart[r + 1]) - static_cast<int>(runstart[r]) >= min_pts) ? 1u : 0u;
    if (threadIdx.x == 0) n_occ = static_cast<unsigned>(n_runs);
    __syncthreads();
    k1_stamp(st, kK1StampPhases, 8);
  } else {
    // =============================== slow path: the radix sort streamed through global scratch ===============================
    // scratch: (key, jpos) pairs, two copies (ping-pong), each n_total words: keyA | posA | keyB | posB
    unsigned* keyA = scratch + bb;
    unsigned* posA = scratch + n_total + bb;
    unsigned* keyB = scratch + 2 * static_cast<size_t>(n_total) + bb;
    unsigned* posB = scratch + 3 * static_cast<size_t>(n_total) + bb;
    unsigned* dcur = dstart;  // [D + 1]
    unsigned* hist = reinterpret_cast<unsigned*>(ox);  // [D] (the point arrays are free during the sort)
    unsigned* tot = hist + D;                           // [D]
    constexpr int R = 8;  // chunks per wave per round
    constexpr unsigned round_n = R * kK1Threads;
    const int n_dig = (cbits + dbits - 1) / dbits;
    for (int dig = 0; dig < n_dig; dig++) {
      const unsigned* ksrc = (dig == 0) ? nullptr : ((dig & 1) ? keyA : keyB);
      const unsigned* psrc = (dig & 1) ? posA : posB;
      unsigned* kdst = (dig & 1) ? keyB : keyA;
      unsigned* pdst = (dig & 1) ? posB : posA;
      const int shift = dig * dbits;
      // sweep 1: histogram of this digit over the whole bucket
      for (int d = threadIdx.x; d < D; d += kK1Threads) hist[d] = 0u;
      __syncthreads();
      for (unsigned j = threadIdx.x; j < nb; j += kK1Threads) {
        unsigned key;
        if (dig == 0) {
          const float4 q = bp[j];
          key = static_cast<unsigned>(k1_local(build_cell(g, q.x, q.y, q.z), map));
        } else {
          key = ksrc[j];
        }
        atomicAdd(&hist[(key >> shift) & dmask], 1u);
      }
      __syncthreads();
      block_scan_array(hist, dcur, D, kK1Threads, s_scan, false);
      __syncthreads();
      // sweep 2: rounds of 4096 elements in order, ranked stably, placed behind what the earlier rounds placed
      for (unsigned r0 = 0; r0 < nb; r0 += round_n) {
        for (int i = threadIdx.x; i < kK1Waves * D / 2; i += kK1Threads) reinterpret_cast<unsigned*>(tab)[i] = 0u;
        __syncthreads();
        unsigned key[R], pj[R], rk[R];
        unsigned short* row = tab + wave * D;
#pragma unroll
        for (int u = 0; u < R; u++) {
          const unsigned j = r0 + static_cast<unsigned>((wave * R + u) * kWave + lane);
          const bool valid = j < nb;
          if (dig == 0) {
            const float4 q = valid ? bp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            key[u] = valid ? static_cast<unsigned>(k1_local(build_cell(g, q.x, q.y, q.z), map)) : 0u;
            pj[u] = j;
          } else {
            key[u] = valid ? ksrc[j] : 0u;
            pj[u] = valid ? psrc[j] : 0u;
          }
          rk[u] = wave_rank(static_cast<int>((key[u] >> shift) & dmask), valid, row, dbits);
        }
        __syncthreads();
        for (int d = threadIdx.x; d < D; d += kK1Threads) {
          unsigned s = 0;
#pragma unroll
          for (int w = 0; w < kK1Waves; w++) {
            const unsigned t = tab[w * D + d];
            tab[w * D + d] = static_cast<unsigned short>(s);
            s += t;
          }
          tot[d] = s;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < R; u++) {
          const unsigned j = r0 + static_cast<unsigned>((wave * R + u) * kWave + lane);
          if (j < nb) {
            const unsigned d = (key[u] >> shift) & dmask;
            const unsigned pos = dcur[d] + row[d] + rk[u];
            kdst[pos] = key[u];
            pdst[pos] = pj[u];
            if (dig == n_dig - 1) sorted_idx[bb + pos] = __float_as_int(bp[pj[u]].w);
          }
        }
        __syncthreads();
        for (int d = threadIdx.x; d < D; d += kK1Threads) dcur[d] += tot[d];
        __syncthreads();
      }
      __threadfence_block();
      __syncthreads();
    }
    const unsigned* skeyg = (n_dig & 1) ? keyA : keyB;  // where the last digit's pass left the sorted pairs
    const unsigned* sposg = (n_dig & 1) ? posA : posB;
    // ---- the runs, a window of at most `cap` sorted positions at a time, windows ending on a run boundary
    unsigned s0 = 0, run0 = 0;
    while (s0 < nb) {
      const unsigned wn = min(static_cast<unsigned>(cap), nb - s0);
      int n_runs = k1_find_runs<PT>([&](unsigned pos) { return skeyg[s0 + pos]; }, wn, runstart, s_scan, &s_total);
      // (position 0 of a window is always a head: windows start on run boundaries)
      unsigned used = wn;
      if (s0 + wn < nb) {  // the window's last run may continue behind it: leave it to the next window ...
        if (n_runs > 1) {
          used = runstart[n_runs - 1];
          n_runs -= 1;
        } else {
          // ... unless it IS the window: one run longer than a window -- a lane team sums it straight from global memory
          if (threadIdx.x == 0) s_u[0] = nb;
          __syncthreads();
          const unsigned kk = skeyg[s0];
          for (unsigned q = s0 + wn + threadIdx.x; q < nb; q += kK1Threads) {  // first position behind the run
            if (skeyg[q] != kk) {
              atomicMin(&s_u[0], q);
              break;
            }
          }
          __syncthreads();
          const unsigned e = s_u[0];
          const unsigned cnt = e - s0;
          const K1GlobalPts PG{bp, sposg, s0};
          if (threadIdx.x < kTeamLanes) team_sum(PG, 0u, cnt, static_cast<int>(threadIdx.x), &s_team[0]);
          __syncthreads();
          if (threadIdx.x == 0) {
            const int cell = k1_cell(k, static_cast<int>(kk), map);
            leaf_slots[bb + run0] = make_uint4(static_cast<unsigned>(cell), bb + s0, cnt, 0u);
            n_occ += 1;
            if (static_cast<int>(cnt) >= min_pts) {
              n_cand += 1;
              const K1Team& T = s_team[0];
              VoxelSums S;
              S.sx = T.s64[0]; S.sy = T.s64[1]; S.sz = T.s64[2];
              S.cxx = T.s64[3]; S.cxy = T.s64[4]; S.cxz = T.s64[5];
              S.cyy = T.s64[6]; S.cyz = T.s64[7]; S.czz = T.s64[8];
              S.fx = T.s32[0]; S.fy = T.s32[1]; S.fz = T.s32[2];
              const FinalizeDump nodump{nullptr, nullptr, nullptr, nullptr, nullptr};
              n_ok += finish_voxel(S, static_cast<int>(cnt), 0, static_cast<int>((bb + s0) / static_cast<unsigned>(min_pts)), cell, min_pts,
                                   eig_ratio, recs, centroids, lut, g, nodump) ? 1u : 0u;
            }
          }
          __syncthreads();
          s0 = e;
          run0 += 1;
          continue;
        }
      }
      if (threadIdx.x == 0) runstart[n_runs] = static_cast<unsigned short>(used);
      __syncthreads();
      const K1GlobalPts PG{bp, sposg, s0};
      n_ok += k1_process_runs(PG, [&](int, unsigned beg) { return static_cast<int>(skeyg[s0 + beg]); }, runstart, n_runs, bb, s0, run0, k, map,
                              min_pts, eig_ratio, g, recs, centroids, lut, leaf_slots, s_team, team_of, &s_nteam);
      for (int r = threadIdx.x; r < n_runs; r += kK1Threads) n_cand += (static_cast<int>(runstart[r + 1]) - static_cast<int>(runstart[r]) >= min_pts) ? 1u : 0u;
      if (threadIdx.x == 0) n_occ += static_cast<unsigned>(n_runs);
      __syncthreads();
      s0 += used;
      run0 += static_cast<unsigned>(n_runs);
    }
  }
  {  // the bucket's voxel counts -> its own words (k1_leaves adds them up when somebody asks: same-address atomics from
     // every block were a sixth of this kernel once)
    unsigned v[3] = {n_ok, n_occ, n_cand};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; q++) {
      unsigned x = v[q];
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
      if (lane == 0) s_scan[wave] = x;
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < kK1Waves; w++) t += s_scan[w];
        bucket_stat[4 * k + q] = t;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) bucket_stat[4 * k + 3] = 0u;
  }
}

// Leaf arrays of a bucket-form build (leaf_cell / leaf_start / leaf_count / leaf_rec: bucket by bucket, ascending local cell
// inside a bucket) and the voxel counts, written only when somebody asks for them (ndt_grid_dump, getFitnessScore's index,
// ndt_grid_size): k1_finalize left every bucket's runs in leaf_slots[bucket base + ordinal] and its counts in bucket_stat.
//   k1_leaf_scan  one block: exclusive scan of the buckets' occupied-cell counts -> occ_base[K + 1]; totals -> counts[1..3]
//   k1_leaves     one block per bucket: its runs -> the leaf arrays at occ_base[bucket]
__global__ __launch_bounds__(kBlock) void k1_leaf_scan(const unsigned* __restrict__ bucket_stat, int K, unsigned* __restrict__ occ_base,
                                                       unsigned* __restrict__ counts) {
  __shared__ unsigned s_scan[kBlock / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int per = (K + kBlock - 1) / kBlock, lo = tid * per, hi = min(K, lo + per);
  unsigned sum[3] = {0, 0, 0};
  for (int i = lo; i < hi; i++)
    for (int q = 0; q < 3; q++) sum[q] += bucket_stat[4 * i + q];
  // occupied cells: exclusive scan
  unsigned inc = sum[1];
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned a = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += a;
  }
  if (lane == kWave - 1) s_scan[wave] = inc;
  __syncthreads();
  unsigned base = 0, all = 0;
  for (int w = 0; w < kBlock / kWave; w++) {
    if (w < wave) base += s_scan[w];
    all += s_scan[w];
  }
  unsigned run = base + inc - sum[1];
  for (int i = lo; i < hi; i++) {
    occ_base[i] = run;
    run += bucket_stat[4 * i + 1];
  }
  if (tid == 0) {
    occ_base[K] = all;
    counts[1] = all;  // occupied voxels
  }
  __syncthreads();
  for (int q = 0; q < 3; q += 2) {  // [3] valid voxels (stat 0), [2] candidates (stat 2)
    unsigned x = sum[q];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    if (lane == 0) s_scan[wave] = x;
    __syncthreads();
    if (tid == 0) {
      unsigned t = 0;
      for (int w = 0; w < kBlock / kWave; w++) t += s_scan[w];
      counts[q == 0 ? 3 : 2] = t;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void k1_leaves(const uint4* __restrict__ leaf_slots, GridGeom g, int min_pts,
                                                    const unsigned* __restrict__ bucket_base, const unsigned* __restrict__ bucket_stat,
                                                    const unsigned* __restrict__ occ_base, int* __restrict__ leaf_cell,
                                                    unsigned* __restrict__ leaf_start, int* __restrict__ leaf_count, int* __restrict__ leaf_rec,
                                                    const int* __restrict__ lut) {
  const int k = blockIdx.x;
  const unsigned bb = bucket_base[k], n_runs = bucket_stat[4 * k + 1], ob = occ_base[k];
  for (unsigned r = threadIdx.x; r < n_runs; r += kBlock) {
    const uint4 e = leaf_slots[bb + r];
    const unsigned o = ob + r;
    const int cell = static_cast<int>(e.x);
    leaf_cell[o] = cell;
    leaf_start[o] = e.y;
    leaf_count[o] = static_cast<int>(e.z);
    int rec = -1;
    if (e.z >= static_cast<unsigned>(min_pts)) {  // the voxel's record: wherever the compaction put it (read back from the table)
      const int cz = cell / g.mul[2], cy = (cell - cz * g.mul[2]) / g.mul[1], cx = cell - cz * g.mul[2] - cy * g.mul[1];
      const int t = lut[static_cast<long long>(cx + kLutBorder) + static_cast<long long>(cy + kLutBorder) * g.pmul[1] +
                        static_cast<long long>(cz + kLutBorder) * g.pmul[2]];
      rec = (t >= 0) ? t : (t <= -2 ? -(t + 2) : -1);
    }
    leaf_rec[o] = rec;
  }
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
                              int n_blocks, hipStream_t stream) {
  if (n == 0) return hipSuccess;
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

static size_t k1_finalize_lds(int pt, bool two, int dbits) {
  const size_t D = static_cast<size_t>(1) << dbits, cap = static_cast<size_t>(pt) * kK1Threads;
  size_t b = kK1Waves * D * 2 + (D + 2) * 4 + std::max(3 * cap, 2 * D) * 4 + cap * 4 + (cap + 2) * 2;
  b += two ? cap * 4 + cap * 2 : cap * 2;
  return (b + 15) & ~static_cast<size_t>(15);
}

bool grid_build_plan(long long n_cells, int n_points, GridBuildPlan& P) {
  if (n_points <= 0 || n_cells <= 0) return false;
  // cells are dealt to the buckets in runs of 2^rb (k1_bucket): K a power of two
  static const int rb_env = [] { const char* v = getenv("NDT_K1_RUN_BITS"); return v ? std::max(0, std::min(8, atoi(v))) : 3; }();
  const int rb = rb_env;
  // Buckets: as many points as one block of k1_finalize sorts in LDS (~4500 of its 6144 on a uniform cloud: a bucket per CU
  // at 1 M points); at least 256 of them, and small clouds (the mapping nodes' 16 k points: latency-bound on their fullest
  // bucket) get ~256 points per bucket
  static const int small_div = [] { const char* v = getenv("NDT_K1_SMALL_DIV"); return v ? std::max(1, atoi(v)) : 256; }();
  static const int big_div = [] { const char* v = getenv("NDT_K1_BUCKET_POINTS"); return v ? std::max(64, atoi(v)) : 4500; }();
  long long k_target = n_points <= 262144 ? std::min<long long>(1024, n_points / small_div) : n_points / big_div;
  int K = std::max(256, std::min(kK1MaxBuckets, pow2_ceil(k_target)));
  const long long runs = (n_cells + (1ll << rb) - 1) >> rb;
  int cbits, dbits;
  for (;;) {
    const long long runs_per_bucket = (runs + K - 1) / K;
    cbits = rb;
    while ((1ll << (cbits - rb)) < runs_per_bucket) cbits++;
    cbits = std::max(cbits, 1);
    dbits = std::min(cbits, kK1DigitBits);
    // two digits: 4096 points per block at most; keep the mean bucket within ~80 % of that
    const bool two = cbits > dbits;
    if (cbits > 2 * kK1DigitBits || (two && n_points / K > 3300)) {
      if (K >= kK1MaxBuckets) {
        if (cbits > 2 * kK1DigitBits) return false;  // (more than 4096 x 4 M cells: the general path)
        break;  // crowded buckets take k1_finalize's streamed path
      }
      K <<= 1;
      continue;
    }
    break;
  }
  P.cbits = cbits;
  P.dbits = dbits;
  int kb = 0;
  while ((1 << kb) < K) kb++;
  P.shift = rb | (kb << 8);  // the packed cell <-> (bucket, local) map of the kernels
  P.n_buckets = K;
  // points per thread of k1_finalize's LDS path: the mean bucket + 12 % + 160 (one-digit keys: up to 12 x 512; two digits: 8 x 512)
  const long long want = static_cast<long long>(n_points) / K * 9 / 8 + 160;
  const bool two = cbits > dbits;
  P.fin_pt = want <= 1024 ? 2 : (want <= 4096 || two) ? 8 : 12;
  static const int pt_env = [] { const char* v = getenv("NDT_K1_FIN_PT"); return v ? atoi(v) : 0; }();
  if (pt_env == 2 || pt_env == 8 || (pt_env == 12 && !two)) P.fin_pt = pt_env;
  if (P.fin_pt == 12 && k1_finalize_lds(12, two, dbits) > kK1MaxDynamicLds) P.fin_pt = 8;  // (wide digits: the counter rows take the room)
  // blocks of k1_hist / k1_scatter: at most 512 rows in the count matrix; a block ranks its points in rounds of 4096
  long long ppb = std::max(512, std::min(kK1Round, pow2_ceil((n_points + 255) / 256)));
  if (static_cast<long long>(n_points) > 512ll * kK1Round) ppb = ((n_points + 511) / 512 + kK1Round - 1) / kK1Round * kK1Round;
  static const int ppb_env = [] { const char* v = getenv("NDT_K1_PPB"); return v ? atoi(v) : 0; }();
  if (ppb_env > 0) ppb = ppb_env;
  while ((n_points + ppb - 1) / ppb > kColGroups * kColRows) ppb += kK1Round;  // k1_colscan holds a column's rows in registers
  P.pts_per_block = static_cast<int>(ppb);
  P.n_blocks = static_cast<int>((n_points + ppb - 1) / ppb);
  return true;
}

template <int PT, bool TWO>
static void launch_k1_finalize(const GridGeom& g, const GridBuildPlan& P, int min_pts, double eig_ratio, const GridBuildScratch& S, int n,
                               int* sorted_idx, VoxelRec* recs, VoxelSide* centroids, int* lut, hipStream_t stream) {
  const size_t lds = k1_finalize_lds(PT, TWO, P.dbits);
  static bool once = [] {  // more than 64 KB of dynamic LDS has to be asked for
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_finalize<PT, TWO>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kK1MaxDynamicLds));
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL((k1_finalize<PT, TWO>), dim3(P.n_buckets), dim3(kK1Threads), lds, stream, S.bpts, g, P.shift, P.n_buckets, P.cbits, P.dbits,
                     min_pts, eig_ratio, S.bucket_base, sorted_idx, recs, centroids, lut, S.bucket_stat, S.leaf_slots, S.order,
                     static_cast<unsigned>(n), S.stamps ? S.stamps + static_cast<size_t>(P.n_blocks) * kK1StampPhases : nullptr);
}

hipError_t launch_grid_build_buckets(const float4* pts, int n, int dense, const GridGeom& g, const GridBuildPlan& P, int min_pts,
                                     double eig_ratio, const GridBuildScratch& S, int* sorted_idx, VoxelRec* recs, VoxelSide* centroids,
                                     int* lut, unsigned* counts, hipStream_t stream) {
  const int K = P.n_buckets;
  hipLaunchKernelGGL(k1_hist, dim3(P.n_blocks), dim3(kK1Threads), static_cast<size_t>(K) * sizeof(unsigned), stream, pts, n, dense, g, P.shift, K,
                     P.pts_per_block, S.cntmat, lut, g.lut_cells);
  const size_t lds_scatter = (static_cast<size_t>(K) + 1 + 2 * static_cast<size_t>(K)) * sizeof(unsigned) + static_cast<size_t>(kK1Waves) * K * sizeof(unsigned short);
  static bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kK1MaxDynamicLds));
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL(k1_colscan, dim3((K + kColCols - 1) / kColCols), dim3(kK1Threads), 0, stream, S.cntmat, P.n_blocks, K, S.cntmat + static_cast<size_t>(P.n_blocks) * K);
  hipLaunchKernelGGL(k1_scatter, dim3(P.n_blocks), dim3(kK1Threads), lds_scatter, stream, pts, n, dense, g, P.shift, K, P.pts_per_block, S.cntmat,
                     S.cntmat + static_cast<size_t>(P.n_blocks) * K, S.bucket_base, S.bpts, counts, S.stamps);
  const bool two = P.cbits > P.dbits;
  if (two) {
    if (P.fin_pt == 2) launch_k1_finalize<2, true>(g, P, min_pts, eig_ratio, S, n, sorted_idx, recs, centroids, lut, stream);
    else launch_k1_finalize<8, true>(g, P, min_pts, eig_ratio, S, n, sorted_idx, recs, centroids, lut, stream);
  } else {
    if (P.fin_pt == 2) launch_k1_finalize<2, false>(g, P, min_pts, eig_ratio, S, n, sorted_idx, recs, centroids, lut, stream);
    else if (P.fin_pt == 8) launch_k1_finalize<8, false>(g, P, min_pts, eig_ratio, S, n, sorted_idx, recs, centroids, lut, stream);
    else launch_k1_finalize<12, false>(g, P, min_pts, eig_ratio, S, n, sorted_idx, recs, centroids, lut, stream);
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

hipError_t launch_grid_leaves(const GridGeom& g, const GridBuildPlan& P, int min_pts, const uint4* leaf_slots, const unsigned* bucket_base,
                              const unsigned* bucket_stat, unsigned* occ_base /* n_buckets + 1 words */, int* leaf_cell, unsigned* leaf_start,
                              int* leaf_count, int* leaf_rec, unsigned* counts, const int* lut, hipStream_t stream) {
  const int K = P.n_buckets;
  hipLaunchKernelGGL(k1_leaf_scan, dim3(1), dim3(kBlock), 0, stream, bucket_stat, K, occ_base, counts);
  hipLaunchKernelGGL(k1_leaves, dim3(K), dim3(kBlock), 0, stream, leaf_slots, g, min_pts, bucket_base, bucket_stat, occ_base, leaf_cell, leaf_start,
                     leaf_count, leaf_rec, lut);
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
