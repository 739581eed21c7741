// LDS-DMA semantics used by derivatives_body_ahead, checked on the device: where the lanes' 4 / 12 / 16 bytes land relative to
// the M0 base, and whether a ds_read right behind s_waitcnt vmcnt(0) (no barrier) sees them.  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned lds_off(const void* p) { return __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p))); }
#define DMA(NAME, INSN) \
  __device__ __forceinline__ void NAME(const void* g, unsigned b) { unsigned keep; \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t" INSN " %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(b) : "memory"); }
DMA(dma32, "global_load_lds_dword")
DMA(dma96, "global_load_lds_dwordx3")
DMA(dma128, "global_load_lds_dwordx4")
__global__ __launch_bounds__(256) void k(const int* src, int* out, int rounds, int* dump) {
  __shared__ int pad[64];  // so that the bases are not zero
  __shared__ int a[4][64], b[4][64 * 3], c[4][64 * 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  pad[lane] = 0;
  int bad = 0;
  for (int r = 0; r < rounds; r++) {
    const int* s = src + ((r * 4 + w) * 64 + (63 - lane)) * 4;  // a gather: lane l reads record 63 - l
    dma32(s, lds_off(&a[w][0]));
    dma96(s, lds_off(&b[w][0]));
    dma128(s, lds_off(&c[w][0]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int va = a[w][lane], vb0 = b[w][lane * 3], vb2 = b[w][lane * 3 + 2], vc3 = c[w][lane * 4 + 3];
    bad += (va != s[0]) + ((vb0 != s[0]) << 8) + ((vb2 != s[2]) << 16) + ((vc3 != s[3]) << 24);
    if (r == 0 && blockIdx.x == 0 && w == 1) {
      dump[lane] = va; dump[64 + lane] = s[0];
      for (int q = 0; q < 3; q++) dump[128 + lane * 3 + q] = b[w][lane * 3 + q];
      for (int q = 0; q < 4; q++) dump[320 + lane * 4 + q] = c[w][lane * 4 + q];
      for (int q = 0; q < 4; q++) dump[576 + lane * 4 + q] = s[q];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  out[blockIdx.x * 256 + threadIdx.x] = bad;
}
int main() {
  const int rounds = 64, n = rounds * 4 * 64 * 4;
  std::vector<int> h(n);
  for (int i = 0; i < n; i++) h[i] = i * 2654435761u;
  int *d, *o;
  hipMalloc(&d, n * sizeof(int));
  hipMalloc(&o, 512 * 256 * sizeof(int));
  hipMemcpy(d, h.data(), n * sizeof(int), hipMemcpyHostToDevice);
  int* dd; hipMalloc(&dd, 1024 * sizeof(int)); hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, o, rounds, dd);
  std::vector<int> ho(512 * 256);
  hipError_t e = hipMemcpy(ho.data(), o, ho.size() * sizeof(int), hipMemcpyDeviceToHost);
  long bad = 0, t[4] = {0, 0, 0, 0};
  for (int v : ho) { bad += v; for (int q = 0; q < 4; q++) t[q] += (v >> (8 * q)) & 255; }
  std::printf("per kind (b32, b96 first, b96 third, b128 fourth): %ld %ld %ld %ld\n", t[0], t[1], t[2], t[3]);
  std::vector<int> hd(1024); hipMemcpy(hd.data(), dd, 4096, hipMemcpyDeviceToHost);
  // value -> index in src (record, word)
  auto where = [&](int v) { for (int i = 0; i < n; i++) if (h[i] == v) return i; return -1; };
  std::printf("b32 lanes 0..7 got src word index: "); for (int l = 0; l < 8; l++) std::printf("%d ", where(hd[l])); std::printf(" expected "); for (int l = 0; l < 8; l++) std::printf("%d ", where(hd[64 + l])); std::printf("\n");
  std::printf("b96 image words 0..23: "); for (int l = 0; l < 24; l++) std::printf("%d ", where(hd[128 + l])); std::printf("\n");
  std::printf("b128 image words 0..15: "); for (int l = 0; l < 16; l++) std::printf("%d ", where(hd[320 + l])); std::printf("\n");
  std::printf("expected per lane (4 words) lanes 0..3: "); for (int l = 0; l < 16; l++) std::printf("%d ", where(hd[576 + l])); std::printf("\n");
  std::printf("%s: mismatches %ld of %d checks\n", hipGetErrorString(e), bad, 512 * 256 * rounds * 4);
  return bad != 0;
}
