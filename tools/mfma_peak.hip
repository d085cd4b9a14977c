// Developer probe: what does the matrix pipe of an MI355X SUSTAIN in bf16 -- nothing but MFMAs issued from registers, on
// non-zero data, at the occupancies the encoder kernels run at?  (the fused MLP kernel with everything but its
// MFMAs removed runs at 0.40 of the 2.5 PFLOP/s headline peak; this says how much of the missing 0.60 is the schedule's and
// how much is the chip's.)   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/build/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// CHAINS independent accumulators, each fed ITERS times in turn: CHAINS = 1 is one dependent chain (the next MFMA reads the
// accumulator the previous one writes), larger values leave CHAINS - 1 independent MFMAs between two dependent ones.
template <int CHAINS, bool BIG>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ seed, float* __restrict__ out, int iters,
                                                 unsigned long long* __restrict__ cyc) {
  const int lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();   // shader-clock ticks (MI355X_MICROARCH.md: tick = shader cycle)
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(seed[(lane * 8 + e) & 1023] * 0.01f);
    b[e] = (__bf16)(seed[(lane * 8 + e + 512) & 1023] * 0.01f);
  }
  float keep = 0.f;
  if constexpr (BIG) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    for (int c = 0; c < CHAINS; ++c)
      for (int i = 0; i < 16; ++i) keep += acc[c][i];
  } else {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
      for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    for (int c = 0; c < CHAINS; ++c)
      for (int i = 0; i < 4; ++i) keep += acc[c][i];
  }
  if (keep == 12345.678f) out[threadIdx.x] = keep;   // never true in practice: keeps the accumulators alive
  if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

template <int CHAINS, bool BIG>
int run(const char* what, const float* seed, float* out, int wgs, int threads, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  static unsigned long long* cyc = nullptr;
  if (cyc == nullptr) CK(hipMalloc(&cyc, 1024 * sizeof(unsigned long long)));
  hipLaunchKernelGGL((mfma_loop<CHAINS, BIG>), dim3(wgs), dim3(threads), 0, 0, seed, out, iters / 8, cyc);   // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((mfma_loop<CHAINS, BIG>), dim3(wgs), dim3(threads), 0, 0, seed, out, iters, cyc);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> hc(wgs);
  CK(hipMemcpy(hc.data(), cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double csum = 0;
  for (int i = 0; i < wgs; ++i) csum += (double)hc[i];
  const double cycles = csum / wgs;   // shader cycles a wave spent in the kernel
  const double flops = (double)wgs * (threads / 64) * (double)iters * CHAINS * (BIG ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
  // effective shader clock = cycles / wall (the event pair also holds the launch, ~10 us of a >= 10 ms kernel); cycles per MFMA
  // per SIMD = cycles / MFMAs issued on that SIMD; what the same cycle count would give at the 2.4 GHz the 2.5 PFLOP/s assume
  const double ghz = cycles / (ms * 1e6), waves_per_simd = (double)wgs * (threads / 64) / 1024.0;
  const double cyc_per_mfma = cycles / ((double)iters * CHAINS * (waves_per_simd < 1 ? 1 : waves_per_simd));
  printf("%-62s %4d wg x %d  %8.3f ms  %7.1f TFLOP/s = %.3f of 2.5 PF | clock %.2f GHz, %.1f cycles per MFMA per SIMD (floor %d) -> %.3f at 2.4 GHz\n",
         what, wgs, threads / 64, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15, ghz, cyc_per_mfma, BIG ? 32 : 16,
         flops / (ms * 1e-3) / 2.5e15 * 2.4 / ghz);
  return 0;
}

int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 1000u) - 500.f;
  float *seed = nullptr, *out = nullptr;
  CK(hipMalloc(&seed, 4096));
  CK(hipMalloc(&out, 4096));
  CK(hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice));
  float* zeros = nullptr;   // the same loops on all-zero operands: the guide's 2,495 TFLOP/s figure is a zero / trivial-data number,
  CK(hipMalloc(&zeros, 4096));   // and the chip clocks to its power budget (MI355X_MICROARCH.md, DVFS give-back)
  CK(hipMemset(zeros, 0, 4096));
  const int IT = 40000;
  run<13, true>("ZERO DATA 32x32x16, 1 wave/SIMD, 13 chains", zeros, out, 256, 256, IT / 13);
  run<4, true>("ZERO DATA 32x32x16, 2 waves/SIMD, 4 chains each", zeros, out, 512, 256, IT / 4);
  run<8, false>("ZERO DATA 16x16x32, 2 waves/SIMD, 8 chains each", zeros, out, 512, 256, IT / 8);
  // one wave per SIMD (the fused MLP kernel's occupancy): 256 workgroups of 4 waves
  run<1, true>("32x32x16, 1 wave/SIMD, one dependent chain", seed, out, 256, 256, IT);
  run<2, true>("32x32x16, 1 wave/SIMD, 2 chains", seed, out, 256, 256, IT / 2);
  run<4, true>("32x32x16, 1 wave/SIMD, 4 chains", seed, out, 256, 256, IT / 4);
  run<13, true>("32x32x16, 1 wave/SIMD, 13 chains (the MLP's fc2 tiles)", seed, out, 256, 256, IT / 13);
  // two waves per SIMD (the tiled GEMM / QKV panel / attention occupancy): 512 workgroups of 4 waves
  run<1, true>("32x32x16, 2 waves/SIMD, one chain each", seed, out, 512, 256, IT);
  run<4, true>("32x32x16, 2 waves/SIMD, 4 chains each", seed, out, 512, 256, IT / 4);
  run<1, false>("16x16x32, 1 wave/SIMD, one dependent chain", seed, out, 256, 256, IT);
  run<4, false>("16x16x32, 1 wave/SIMD, 4 chains", seed, out, 256, 256, IT / 4);
  run<4, false>("16x16x32, 2 waves/SIMD, 4 chains each", seed, out, 512, 256, IT / 4);
  run<8, false>("16x16x32, 2 waves/SIMD, 8 chains each", seed, out, 512, 256, IT / 8);
  // a short launch (the length of one MLP round, ~0.1 ms) against the long ones above: does the clock hold?
  run<13, true>("32x32x16, 1 wave/SIMD, 13 chains, ~0.1 ms launch", seed, out, 256, 256, 180);
  return 0;
}
