// Developer probe: what does one dependent kernel boundary cost on this box, eager vs hipGraph, for trivial kernels of
// the decode chain's geometry?  hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o gpurun_out/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1.0f;
}
// one 16-byte load per lane from a 346 KB "weight" + dependent store: the memory round trip of a decode GEMM
__global__ void k_round(const float4* __restrict__ w, const float* __restrict__ in, float* __restrict__ out, int n4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float4 v = w[i % n4];
  out[i] = in[i] + v.x + v.y + v.z + v.w;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *a, *b; float4* w;
  const int n = 208 * 256;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&w, 346112));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(w, 0, 346112));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 456;  // 8 layers x 57
  auto chain = [&](int kind, int grid, int block) {
    for (int i = 0; i < N; ++i) {
      float* in = (i & 1) ? b : a; float* out = (i & 1) ? a : b;
      if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, s);
      else if (kind == 1) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(block), 0, s, in, out, n);
      else hipLaunchKernelGGL(k_round, dim3(grid), dim3(block), 0, s, w, in, out, 346112 / 16);
    }
  };
  struct Cfg { const char* name; int kind, grid, block; } cfgs[] = {
      {"empty 1x64", 0, 1, 64}, {"empty 208x256", 0, 208, 256}, {"empty 2048x256", 0, 2048, 256},
      {"touch 208x256", 1, 208, 256}, {"round 208x256", 2, 208, 256}};
  for (auto& c : cfgs) {
    // eager
    chain(c.kind, c.grid, c.block); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int r = 0; r < 5; ++r) chain(c.kind, c.grid, c.block); CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const float eager = ms * 1e3f / (5 * N);
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(c.kind, c.grid, c.block); CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-18s eager %.2f us/kernel   graph %.2f us/kernel\n", c.name, eager, ms * 1e3f / (5 * N));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
