// How much of a chain of small dependent kernels is launch gap, and what a hipGraph of the same chain saves (development aid).
// hipcc --offload-arch=gfx950 -O2 tools/probes/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_small(float* p, int n, float a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * a + 1.0f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int n = 60000, chain = 10, reps = 200;
  float* d;
  hipMalloc(&d, n * sizeof(float));
  hipMemset(d, 0, n * sizeof(float));
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  auto run_chain = [&] { for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, st, d, n, 0.5f); };
  for (int w = 0; w < 20; w++) run_chain();
  hipStreamSynchronize(st);
  double t0 = now_us();
  for (int r = 0; r < reps; r++) { run_chain(); hipStreamSynchronize(st); }
  const double plain_sync = (now_us() - t0) / reps;
  t0 = now_us();
  for (int r = 0; r < reps; r++) run_chain();
  hipStreamSynchronize(st);
  const double plain_b2b = (now_us() - t0) / reps;
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  run_chain();
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int w = 0; w < 20; w++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  t0 = now_us();
  for (int r = 0; r < reps; r++) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
  const double graph_sync = (now_us() - t0) / reps;
  t0 = now_us();
  for (int r = 0; r < reps; r++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  const double graph_b2b = (now_us() - t0) / reps;
  // one kernel alone, for the kernel's own time
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int r = 0; r < 100; r++) hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, st, d, n, 0.5f);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::printf("chain of %d dependent kernels over %d floats: plain launches %.1f us (launch + sync each chain), %.1f us back to back; graph %.1f us / %.1f us; per kernel in a long run %.2f us\n",
              chain, n, plain_sync, plain_b2b, graph_sync, graph_b2b, ms * 1e3 / 100);
  return 0;
}
