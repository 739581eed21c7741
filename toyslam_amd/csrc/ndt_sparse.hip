// ndt_sparse.hip -- voxel index for grids whose cell count dwarfs their point count (a 0.1 m prefilter over 150 m is
// 7 x 10^8 cells for 10^5 points; a kilometre-sized map at 1 m voxels): what the reference gets for free from its
// std::map<size_t, Leaf> keyed by the linear voxel index (voxel_grid_covariance_omp.h:201, _impl.hpp:218-237).
//
// The dense forms of K1 keep per-CELL arrays (counters, the look-up table); here everything is per POINT:
//   keys      linear voxel index of every point (the build's index math, trap 2), invalid points last
//   sort      stable LSD radix sort of (key, point index) pairs -- rocPRIM's device-wide sort through hipCUB, the one
//             library primitive of this repository's kernels: a plain sort, nothing NDT-specific to fuse into it.
//             Stable, so the points of a voxel come out in ascending point order: the order the reference adds them in
//   segments  heads of equal-key runs -> leaf arrays (cell, start, count, record slot) in ascending voxel order
// and the derivative kernels find a voxel through an open-addressing hash table keyed by the same linear index
// (GridView::hash, filled by the finalize pass) instead of the padded dense table.
#include <hipcub/hipcub.hpp>

#include "ndt_device.hpp"

namespace ndt {

namespace {

constexpr unsigned kInvalidKey = 0x7fffffffu;

__global__ __launch_bounds__(kBlock) void k_sp_keys(const float4* __restrict__ pts, int n, int dense, GridGeom g, unsigned* __restrict__ keys,
                                                    int* __restrict__ vals) {
#pragma clang fp contract(off)
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 p = pts[i];
    unsigned key = kInvalidKey;
    if (dense || finite3(p.x, p.y, p.z)) {
      // floor(x * inv_leaf) - float(min_b), _impl.hpp:218-223 (f32; the product rounded before floor())
      const float fx = p.x * g.inv_leaf[0], fy = p.y * g.inv_leaf[1], fz = p.z * g.inv_leaf[2];
      const int i0 = static_cast<int>(floorf(fx) - static_cast<float>(g.min_b[0]));
      const int i1 = static_cast<int>(floorf(fy) - static_cast<float>(g.min_b[1]));
      const int i2 = static_cast<int>(floorf(fz) - static_cast<float>(g.min_b[2]));
      const long long c = static_cast<long long>(i0) * g.mul[0] + static_cast<long long>(i1) * g.mul[1] + static_cast<long long>(i2) * g.mul[2];
      if (i0 >= 0 && i1 >= 0 && i2 >= 0 && c >= 0 && c < g.n_cells) key = static_cast<unsigned>(c);
    }
    keys[i] = key;
    vals[i] = i;
  }
}

// head flags of the runs of equal keys in the sorted key array (invalid keys form no run)
__global__ __launch_bounds__(kBlock) void k_sp_heads(const unsigned* __restrict__ keys, int n, unsigned* __restrict__ flags,
                                                     unsigned* __restrict__ counts) {
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
    const unsigned k = keys[j];
    const unsigned prev = j ? keys[j - 1] : kInvalidKey;
    flags[j] = (k != kInvalidKey && (j == 0 || k != prev)) ? 1u : 0u;
    if (k == kInvalidKey && (j == 0 || prev != kInvalidKey)) counts[0] = static_cast<unsigned>(j);  // points binned
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && n > 0 && keys[n - 1] != kInvalidKey) counts[0] = static_cast<unsigned>(n);
}

__global__ __launch_bounds__(kBlock) void k_sp_leaves(const unsigned* __restrict__ keys, const unsigned* __restrict__ flags,
                                                      const unsigned* __restrict__ ord /* exclusive scan of flags */, int n,
                                                      int* __restrict__ leaf_cell, unsigned* __restrict__ leaf_start, unsigned* __restrict__ counts) {
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
    if (flags[j]) {
      leaf_cell[ord[j]] = static_cast<int>(keys[j]);
      leaf_start[ord[j]] = static_cast<unsigned>(j);
    }
    if (j == n - 1) counts[1] = ord[j] + flags[j];  // occupied voxels
  }
}

__global__ __launch_bounds__(kBlock) void k_sp_counts(const unsigned* __restrict__ leaf_start, const unsigned* __restrict__ counts_in,
                                                      int min_pts, int* __restrict__ leaf_count, int* __restrict__ leaf_rec,
                                                      unsigned* __restrict__ counts) {
  const int n_leaves = static_cast<int>(counts_in[1]);
  const unsigned n_binned = counts_in[0];
  unsigned cand = 0;
  for (int o = blockIdx.x * kBlock + threadIdx.x; o < n_leaves; o += gridDim.x * kBlock) {
    const unsigned s = leaf_start[o], e = (o + 1 < n_leaves) ? leaf_start[o + 1] : n_binned;
    const int cnt = static_cast<int>(e - s);
    leaf_count[o] = cnt;
    // record slot = segment start / min_pts: unique per candidate (their segments start at least min_pts apart)
    leaf_rec[o] = (cnt >= min_pts) ? static_cast<int>(s / static_cast<unsigned>(min_pts)) : -1;
    cand += (cnt >= min_pts) ? 1u : 0u;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) cand += __shfl_xor(cand, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0 && cand) atomicAdd(counts + 2, cand);
}

inline int grid_for_n(size_t n, int cap) {
  size_t b = (n + kBlock - 1) / kBlock;
  return static_cast<int>(std::max<size_t>(1, std::min<size_t>(b, static_cast<size_t>(cap))));
}

}  // namespace

size_t sparse_index_temp_bytes(int n) {
  size_t a = 0, b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, static_cast<const unsigned*>(nullptr), static_cast<unsigned*>(nullptr),
                                           static_cast<const int*>(nullptr), static_cast<int*>(nullptr), n, 0, 31);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, static_cast<const unsigned*>(nullptr), static_cast<unsigned*>(nullptr), n);
  return std::max(a, b) + 256;
}

hipError_t launch_sparse_index(const float4* pts, int n, int dense, const GridGeom& g, int min_pts, void* temp, size_t temp_bytes,
                               unsigned* keys_a, unsigned* keys_b, int* vals_a, unsigned* flags, unsigned* ord, int* leaf_cell,
                               unsigned* leaf_start, int* leaf_count, int* leaf_rec, int* sorted_idx, unsigned* counts,
                               hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(counts, 0, 5 * sizeof(unsigned), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_sp_keys, dim3(grid_for_n(n, 2048)), dim3(kBlock), 0, stream, pts, n, dense, g, keys_a, vals_a);
  int bits = 1;
  while (bits < 31 && (1ll << bits) < g.n_cells) bits++;
  bits = 31;  // the invalid key (0x7fffffff) must sort last: all 31 bits take part
  size_t tb = temp_bytes;
  e = hipcub::DeviceRadixSort::SortPairs(temp, tb, keys_a, keys_b, vals_a, sorted_idx, n, 0, bits, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_sp_heads, dim3(grid_for_n(n, 2048)), dim3(kBlock), 0, stream, keys_b, n, flags, counts);
  tb = temp_bytes;
  e = hipcub::DeviceScan::ExclusiveSum(temp, tb, flags, ord, n, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_sp_leaves, dim3(grid_for_n(n, 2048)), dim3(kBlock), 0, stream, keys_b, flags, ord, n, leaf_cell, leaf_start, counts);
  hipLaunchKernelGGL(k_sp_counts, dim3(grid_for_n(n, 1024)), dim3(kBlock), 0, stream, leaf_start, counts, min_pts, leaf_count, leaf_rec, counts);
  return hipGetLastError();
}

}  // namespace ndt
