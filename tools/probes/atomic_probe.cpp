// atomic_probe.cpp -- how fast can 1M points be binned into ~100k voxel counters on gfx950?
// (K1's k_count: returning device-scope atomics at ~33 G/s.)  Variants:
//   agent-ret   : __hip_atomic_fetch_add(..., AGENT), value used          (what k_count does)
//   agent-noret : same, value unused
//   wg-ret      : workgroup-scope fetch_add on a per-XCD copy of the counters (index = HW_REG_XCC_ID); executes in that
//                 XCD's L2 if the hardware keeps workgroup-scope atomics there
//   lds-bucket  : per-block LDS histogram of 1024 coarse buckets + row store (pass 1 of an MSD bucket sort, no global atomics)
// build: hipcc -O3 --offload-arch=gfx950 -o atomic_probe atomic_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  // s_getreg_b32 HW_REG_XCC_ID (id 20), offset 0, size 4
  return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_bin(const unsigned* __restrict__ keys, int n, unsigned* __restrict__ cnt, int n_cells,
                                             unsigned* __restrict__ rank) {
  const unsigned x = (MODE == 2) ? xcc_id() : 0u;
  unsigned* c = cnt + static_cast<size_t>(x) * n_cells;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const unsigned k = keys[i];
    if (MODE == 0) rank[i] = __hip_atomic_fetch_add(c + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 1) (void)__hip_atomic_fetch_add(c + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else rank[i] = (x << 28) | __hip_atomic_fetch_add(c + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

__global__ __launch_bounds__(512) void k_lds_hist(const unsigned* __restrict__ keys, int n, int per_block, int shift,
                                                  unsigned* __restrict__ hist /*[blocks][1024]*/) {
  __shared__ unsigned h[1024];
  for (int t = threadIdx.x; t < 1024; t += 512) h[t] = 0;
  __syncthreads();
  const int lo = blockIdx.x * per_block, hi = min(n, lo + per_block);
  for (int i = lo + threadIdx.x; i < hi; i += 512) atomicAdd(&h[keys[i] >> shift], 1u);
  __syncthreads();
  for (int t = threadIdx.x; t < 1024; t += 512) hist[static_cast<size_t>(blockIdx.x) * 1024 + t] = h[t];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000000, n_cells = argc > 2 ? atoi(argv[2]) : 100000;
  std::vector<unsigned> keys(n);
  unsigned s = 12345;
  for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; keys[i] = (s >> 8) % n_cells; }
  unsigned *d_keys, *d_cnt, *d_rank, *d_hist;
  CK(hipMalloc(&d_keys, n * 4)); CK(hipMalloc(&d_cnt, size_t(8) * n_cells * 4)); CK(hipMalloc(&d_rank, n * 4));
  CK(hipMalloc(&d_hist, size_t(4096) * 1024 * 4));
  CK(hipMemcpy(d_keys, keys.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[3] = {"agent-ret", "agent-noret", "wg-ret(per-XCD)"};
  for (int grid : {1024, 2048, 4096}) {
    for (int mode = 0; mode < 3; mode++) {
      float best = 1e9;
      for (int rep = 0; rep < 6; rep++) {
        CK(hipMemsetAsync(d_cnt, 0, size_t(8) * n_cells * 4, 0));
        CK(hipEventRecord(a, 0));
        if (mode == 0) hipLaunchKernelGGL(k_bin<0>, dim3(grid), dim3(256), 0, 0, d_keys, n, d_cnt, n_cells, d_rank);
        if (mode == 1) hipLaunchKernelGGL(k_bin<1>, dim3(grid), dim3(256), 0, 0, d_keys, n, d_cnt, n_cells, d_rank);
        if (mode == 2) hipLaunchKernelGGL(k_bin<2>, dim3(grid), dim3(256), 0, 0, d_keys, n, d_cnt, n_cells, d_rank);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
      }
      // correctness: total over the 8 copies per cell must equal the true histogram
      std::vector<unsigned> c(size_t(8) * n_cells); CK(hipMemcpy(c.data(), d_cnt, c.size() * 4, hipMemcpyDeviceToHost));
      std::vector<unsigned> ref(n_cells, 0); for (int i = 0; i < n; i++) ref[keys[i]]++;
      size_t bad = 0; unsigned used = 0;
      for (int k = 0; k < n_cells; k++) { unsigned t = 0; for (int x = 0; x < 8; x++) t += c[size_t(x) * n_cells + k]; bad += (t != ref[k]); }
      for (int x = 0; x < 8; x++) { unsigned long long t = 0; for (int k = 0; k < n_cells; k++) t += c[size_t(x) * n_cells + k]; used += (t != 0); }
      printf("grid %4d  %-16s %8.2f us  %6.1f G atomics/s  wrong cells %zu  copies used %u\n", grid, names[mode], best * 1e3, n / (best * 1e-3) / 1e9, bad, used);
    }
  }
  for (int per_block : {2048, 4096, 8192}) {
    const int blocks = (n + per_block - 1) / per_block;
    int shift = 0; while ((n_cells >> shift) > 1024) shift++;
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(k_lds_hist, dim3(blocks), dim3(512), 0, 0, d_keys, n, per_block, shift, d_hist);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("lds-bucket per_block %5d blocks %4d shift %d: %8.2f us\n", per_block, blocks, shift, best * 1e3);
  }
  return 0;
}
