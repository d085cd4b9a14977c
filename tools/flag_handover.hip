// Developer probe for DESIGN.md 3b ("the decode chain: what was analysed and not built"): what does a producer -> consumer
// hand-over INSIDE one kernel cost (device-scope release / acquire through a counter in global memory, consumers of stage s
// spin until all G workgroups of stage s - 1 have signalled), against a dependent kernel boundary in a replayed hipGraph?
// Every stage does one 16-byte load per lane from a 346 KB "weight", one from the previous stage's output tile (and one from
// a tile written by ANOTHER workgroup, so that the data really crosses CUs / XCDs) and a store -- the memory round trip of a
// decode GEMM.  Workgroups are dispatched in blockIdx order and a workgroup only ever waits for lower-numbered ones; the
// spin is bounded anyway (a consumer that gives up sets a flag and the run is reported as failed).
//   hipcc --offload-arch=gfx950 -O3 tools/flag_handover.hip -o tools/build/flag_handover
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int TPB = 256;

__device__ __forceinline__ void stage_work(const float4* __restrict__ w, int n4, const float4* __restrict__ in, float4* __restrict__ out,
                                           int j, int G) {
  const int t = threadIdx.x;
  const float4 a = in[(long)j * TPB + t];
  const float4 b = in[(long)((j + G / 2) % G) * TPB + t];   // a tile another workgroup wrote
  const float4 v = w[((long)j * TPB + t) % n4];
  out[(long)j * TPB + t] = make_float4(a.x + b.x * 0.5f + v.x, a.y + b.y * 0.5f + v.y, a.z + b.z * 0.5f + v.z, a.w + b.w * 0.5f + v.w);
}

__global__ void k_stage(const float4* w, int n4, const float4* in, float4* out, int G) { stage_work(w, n4, in, out, blockIdx.x, G); }

// grid = S * G workgroups; stage s reads buffer (s & 1) and writes the other
__global__ void k_dataflow(const float4* w, int n4, float4* buf0, float4* buf1, int G, int S, unsigned* cnt, unsigned target,
                           int* failed, int prefetch, int naps) {
  const int s = blockIdx.x / G, j = blockIdx.x - s * G;
  const float4* in = (s & 1) ? buf1 : buf0;
  float4* out = (s & 1) ? buf0 : buf1;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (prefetch) v = w[((long)j * TPB + threadIdx.x) % n4];   // the weight does not depend on the previous stage
  if (s > 0) {
    if (threadIdx.x == 0) {
      int spins = 0;
      while (__hip_atomic_load(&cnt[s - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(127);   // 127 x 64 clocks per nap: fewer polls in flight
        if (++spins > (1 << 20)) {
          *failed = 1;
          break;
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  const int t = threadIdx.x;
  const float4 a = in[(long)j * TPB + t];
  const float4 b = in[(long)((j + G / 2) % G) * TPB + t];
  if (!prefetch) v = w[((long)j * TPB + t) % n4];
  out[(long)j * TPB + t] = make_float4(a.x + b.x * 0.5f + v.x, a.y + b.y * 0.5f + v.y, a.z + b.z * 0.5f + v.z, a.w + b.w * 0.5f + v.w);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&cnt[s], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// the same chain with NO bulk fences: the tiles are written and read with agent-scope relaxed atomics (per-access coherence:
// the stores go through to memory, the loads do not hit a stale L2 line), the producer waits for its stores (vmcnt) before it
// signals, the counter itself is relaxed
__global__ void k_dataflow_atomic(const float4* w, int n4, float* buf0, float* buf1, int G, int S, unsigned* cnt, unsigned target,
                                  int* failed) {
  const int s = blockIdx.x / G, j = blockIdx.x - s * G;
  const float* in = (s & 1) ? buf1 : buf0;
  float* out = (s & 1) ? buf0 : buf1;
  const int t = threadIdx.x;
  const float4 v = w[((long)j * TPB + t) % n4];
  if (s > 0) {
    if (threadIdx.x == 0) {
      int spins = 0;
      while (__hip_atomic_load(&cnt[s - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) {
          *failed = 1;
          break;
        }
      }
    }
    __syncthreads();
  }
  float a[4], b[4];
  const long ia = ((long)j * TPB + t) * 4, ib = ((long)((j + G / 2) % G) * TPB + t) * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = __hip_atomic_load(&in[ia + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b[e] = __hip_atomic_load(&in[ib + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) __hip_atomic_store(&out[ia + e], a[e] + b[e] * 0.5f + vv[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&cnt[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int G = 208, S = 64, n4 = 346112 / 16;
  float4 *b0, *b1, *w;
  unsigned* cnt;
  int* failed;
  CK(hipMalloc(&b0, (size_t)G * TPB * 16)); CK(hipMalloc(&b1, (size_t)G * TPB * 16)); CK(hipMalloc(&w, 346112));
  CK(hipMalloc(&cnt, S * sizeof(unsigned))); CK(hipMalloc(&failed, sizeof(int)));
  CK(hipMemset(b0, 0, (size_t)G * TPB * 16)); CK(hipMemset(b1, 0, (size_t)G * TPB * 16)); CK(hipMemset(w, 0, 346112));
  CK(hipMemset(cnt, 0, S * sizeof(unsigned))); CK(hipMemset(failed, 0, sizeof(int)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  // (a) S dependent kernels in a replayed graph
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int s = 0; s < S; ++s) hipLaunchKernelGGL(k_stage, dim3(G), dim3(TPB), 0, st, w, n4, (s & 1) ? b1 : b0, (s & 1) ? b0 : b1, G);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st)); for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%d stages x %d workgroups: graph of dependent kernels  %.2f us per stage\n", S, G, ms * 1e3f / (20 * S));
  // (b) one kernel, flag hand-over between the stages
  for (int cfg = 0; cfg < 6; ++cfg) {
    const int prefetch = cfg & 1, naps = cfg < 2 ? 0 : cfg < 4 ? 1 : 4;
    unsigned epoch = 0;
    auto run = [&] { ++epoch; hipLaunchKernelGGL(k_dataflow, dim3(S * G), dim3(TPB), 0, st, w, n4, b0, b1, G, S, cnt, epoch * (unsigned)G, failed, prefetch, naps); };
    CK(hipMemset(cnt, 0, S * sizeof(unsigned)));
    run(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); for (int r = 0; r < 20; ++r) run(); CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
    int f = 0; CK(hipMemcpy(&f, failed, sizeof(int), hipMemcpyDeviceToHost));
    printf("%d stages x %d workgroups: ONE kernel, counter hand-over, %d naps of 8 k clocks between polls%s  %.2f us per stage%s\n", S, G, naps,
           prefetch ? ", weight load issued before the wait" : "", ms * 1e3f / (20 * S), f ? "  (a consumer gave up waiting: INVALID)" : "");
  }
  {
    unsigned epoch = 0;
    auto run = [&] { ++epoch; hipLaunchKernelGGL(k_dataflow_atomic, dim3(S * G), dim3(TPB), 0, st, w, n4, (float*)b0, (float*)b1, G, S, cnt, epoch * (unsigned)G, failed); };
    CK(hipMemset(cnt, 0, S * sizeof(unsigned)));
    CK(hipMemset(failed, 0, sizeof(int)));
    run(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); for (int r = 0; r < 20; ++r) run(); CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
    int f = 0; CK(hipMemcpy(&f, failed, sizeof(int), hipMemcpyDeviceToHost));
    printf("%d stages x %d workgroups: ONE kernel, tiles through agent-scope atomics, no fences  %.2f us per stage%s\n", S, G,
           ms * 1e3f / (20 * S), f ? "  (a consumer gave up waiting: INVALID)" : "");
  }
  return 0;
}
