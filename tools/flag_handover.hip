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


// ---- RESIDENT grid (round 5; the form MI355X_MICROARCH.md prices as barrier-counter / barrier-xcd): G workgroups, one per CU,
// each walking all S stages -- nobody waits for a workgroup that has not been dispatched.  (The kernels above launch S x G =
// 13,312 workgroups of which ~2 k are resident: what they measure is dispatch starvation, not a hand-over.) ----
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ bool poll_until(const unsigned* p, unsigned target, int* failed) {
  int spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    // bounded: ~20 ms for the first workgroup that gives up, and everybody else stops waiting as soon as somebody has
    if (++spins > (1 << 18) || ((spins & 255) == 0 && __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return true;
}
// mode 0: plain stores, lane-0 release fence -> ONE counter -> relaxed poll -> acquire fence (barrier-counter)
// mode 1: sc1 (write-through) 16-byte stores and sc1 loads, drained flag, no fences at all
// mode 2: plain stores, XCD-hierarchical barrier: per-XCC arrival counter, the XCD's last arriver releases and arrives at the top
//         counter, waits for all XCDs, acquires, bumps the XCD's generation; the others poll the generation and acquire (barrier-xcd)
__global__ __launch_bounds__(256) void k_resident(const float4* w, int n4, float4* buf0, float4* buf1, int G, int S, unsigned* cnt,
                                                  unsigned* xcnt, unsigned* xgen, const unsigned* xpop, unsigned epoch, int* failed,
                                                  int mode) {
  const int j = blockIdx.x, t = threadIdx.x;
  const unsigned x = xcc_id();
  for (int s = 0; s < S; ++s) {
    const float4* in = (s & 1) ? buf1 : buf0;
    float4* out = (s & 1) ? buf0 : buf1;
    const float4 v = w[((long)j * TPB + t) % n4];   // does not depend on the previous stage: requested before the wait
    if (s > 0) {
      const unsigned stage_id = (epoch - 1) * (unsigned)S + (unsigned)(s - 1) + 1;   // monotonic over launches
      if (mode == 2) {
        if (t == 0) poll_until(&xgen[x], stage_id, failed);
      } else {
        if (t == 0) poll_until(&cnt[0], stage_id * (unsigned)G, failed);
      }
      if (mode != 1 && t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
    }
    float4 a, b;
    const float4* pa = in + (long)j * TPB + t;
    const float4* pb = in + (long)((j + G / 2) % G) * TPB + t;   // a tile another workgroup wrote
    if (mode == 1) {
      f4v ra, rb;
      asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(ra), "=&v"(rb) : "v"(pa), "v"(pb) : "memory");
      a = make_float4(ra[0], ra[1], ra[2], ra[3]);
      b = make_float4(rb[0], rb[1], rb[2], rb[3]);
    } else {
      a = *pa;
      b = *pb;
    }
    const f4v r = {a.x + b.x * 0.5f + v.x, a.y + b.y * 0.5f + v.y, a.z + b.z * 0.5f + v.z, a.w + b.w * 0.5f + v.w};
    float4* po = out + (long)j * TPB + t;
    if (mode == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(po), "v"(r) : "memory");
    else *po = make_float4(r[0], r[1], r[2], r[3]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      const unsigned stage_id = (epoch - 1) * (unsigned)S + (unsigned)s + 1;
      if (mode == 2) {
        const unsigned ticket = __hip_atomic_fetch_add(&xcnt[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket + 1 == stage_id * xpop[x]) {   // the XCD's last arriver of this stage
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_fetch_add(&cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          poll_until(&cnt[0], stage_id * xpop[8], failed);   // xpop[8] = XCDs that hold workgroups
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(&xgen[x], stage_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        if (mode == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_fetch_add(&cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
// which XCD every workgroup of a G-workgroup launch lands on (the barrier needs the population of each)
__global__ void k_census(unsigned* xpop) {
  if (threadIdx.x == 0) atomicAdd(&xpop[xcc_id()], 1u);
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
  // (c) the RESIDENT forms: G workgroups walk the S stages themselves
  {
    unsigned *xcnt, *xgen, *xpop;
    CK(hipMalloc(&xcnt, 64)); CK(hipMalloc(&xgen, 64)); CK(hipMalloc(&xpop, 64));
    CK(hipMemset(xpop, 0, 64));
    hipLaunchKernelGGL(k_census, dim3(G), dim3(TPB), 0, st, xpop);
    CK(hipStreamSynchronize(st));
    unsigned hp[16] = {0};
    CK(hipMemcpy(hp, xpop, 32, hipMemcpyDeviceToHost));
    unsigned live = 0;
    for (int i = 0; i < 8; ++i) live += hp[i] > 0;
    hp[8] = live;
    CK(hipMemcpy(xpop, hp, 36, hipMemcpyHostToDevice));
    printf("census of a %d-workgroup launch by XCD: %u %u %u %u %u %u %u %u\n", G, hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7]);
    const char* names[3] = {"plain stores, release fence -> one counter -> relaxed poll -> acquire fence (barrier-counter)",
                            "sc1 stores / sc1 loads, drained counter, no fences",
                            "plain stores, XCD-hierarchical barrier (per-XCC counters, leaders release / acquire; barrier-xcd)"};
    for (int mode = 0; mode < 3; ++mode) {
      unsigned epoch = 0;
      auto run = [&] { ++epoch; hipLaunchKernelGGL(k_resident, dim3(G), dim3(TPB), 0, st, w, n4, b0, b1, G, S, cnt, xcnt, xgen, xpop, epoch, failed, mode); };
      CK(hipMemset(cnt, 0, S * sizeof(unsigned))); CK(hipMemset(xcnt, 0, 64)); CK(hipMemset(xgen, 0, 64)); CK(hipMemset(failed, 0, sizeof(int)));
      run(); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st)); for (int r = 0; r < 20; ++r) run(); CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
      int f = 0; CK(hipMemcpy(&f, failed, sizeof(int), hipMemcpyDeviceToHost));
      printf("%d stages, %d RESIDENT workgroups, ONE kernel: %s  %.2f us per stage%s\n", S, G, names[mode], ms * 1e3f / (20 * S),
             f ? "  (a workgroup gave up waiting: INVALID)" : "");
    }
  }
  return 0;
}
