// Developer probe for the review's "2 waves per SIMD" question: the fused MLP kernel runs ONE wave per SIMD
// (512 registers), and its weight stream -- LDS-DMA pieces issued by the same waves that issue the MFMAs -- barely overlaps the
// MFMAs (0.061 + 0.038 -> 0.089 ms per round).  Would two waves per SIMD, each with half the MFMAs and half the pieces, hide one
// wave's DMA issue under the other's MFMAs?  This is the MLP's inner loop reduced to its traffic: per chunk and workgroup 48 KiB
// of weights HBM/L2 -> LDS by `global_load_lds_dwordx4` (3-stage ring, counted vmcnt, one barrier per chunk), every wave reads
// its A fragments with ds_read_b128 and issues 32x32x16 MFMAs on independent accumulators; B operands stay in registers.
//   shape A: 4 waves (1 per SIMD), 48 MFMAs + 12 pieces per wave and chunk, 13 accumulators  (the product's ratio 52 : 13)
//   shape B: 8 waves (2 per SIMD), 24 MFMAs +  6 pieces per wave and chunk,  7 accumulators  (what fits 256 registers)
// hipcc --offload-arch=gfx950 -O3 tools/mfma_dma_overlap.hip -o tools/build/mfma_dma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) char lds_char_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(lds_char_t*)(p); }
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE: 0 = both, 1 = MFMAs + fragment reads only (no DMA after the prologue), 2 = the weight stream only (DMA, waits, barrier)
template <int NW, int NM, int P, int CH, int MODE>
__global__ __launch_bounds__(64 * NW) void loop_kernel(const uint4* __restrict__ w, long w_pieces, float* __restrict__ out,
                                                       int chunks) {
  extern __shared__ __attribute__((aligned(16))) uint4 ring[];   // 3 stages x NW * P pieces of 64 uint4
  constexpr int SP = NW * P;   // pieces per stage
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(&ring[0]));
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  bf16x8 b;
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.01f * (float)((lane * 8 + e) % 37 - 18));
  auto issue = [&](int c) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      long piece = ((long)blockIdx.x * 7 + (long)c * SP + wave * P + p) % w_pieces;   // every CU walks the whole buffer, staggered
      dma16(w + piece * 64 + lane, base + (unsigned)(((c % 3) * SP + wave * P + p) * 1024));
    }
  };
  issue(0);
  issue(1);
#pragma unroll 1
  for (int c = 0; c < chunks; ++c) {
    if (c + 1 < chunks) wait_vmcnt<P>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (MODE != 1 && c + 2 < chunks) issue(c + 2);
    if (MODE != 2) {
      const uint4* st = ring + (c % 3) * SP * 64;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const uint4 q = st[((m * 5 + wave * 3) % SP) * 64 + lane];
        acc[m % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&q), b, acc[m % CH], 0, 0, 0);
      }
    }
  }
  float keep = 0.f;
  for (int c = 0; c < CH; ++c)
    for (int i = 0; i < 16; ++i) keep += acc[c][i];
  if (keep == 12345.678f) out[threadIdx.x] = keep;
}

template <int NW, int NM, int P, int CH, int MODE>
int run(const char* what, const uint4* w, long w_pieces, float* out) {
  const int chunks = 52 * 8, lds = 3 * NW * P * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<NW, NM, P, CH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((loop_kernel<NW, NM, P, CH, MODE>), dim3(256), dim3(64 * NW), lds, 0, w, w_pieces, out, chunks);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((loop_kernel<NW, NM, P, CH, MODE>), dim3(256), dim3(64 * NW), lds, 0, w, w_pieces, out, chunks);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double us_chunk = ms * 1e3 / chunks;
  const double flops = MODE == 2 ? 0.0 : 256.0 * NW * NM * chunks * 2.0 * 32 * 32 * 16;
  printf("%-72s %7.3f us per chunk  %7.1f TFLOP/s = %.3f of 2.5 PF   %5.1f GB/s of weights per CU\n", what, us_chunk, flops / (ms * 1e-3) / 1e12,
         flops / (ms * 1e-3) / 2.5e15, MODE == 1 ? 0.0 : NW * P * 1024.0 / (us_chunk * 1e-6) / 1e9);
  return 0;
}

int main() {
  const long w_pieces = 22L * 1024;   // 22 MiB of "weights" (the base encoder's MLP weights of 8 layers are 22 MB)
  uint4* w = nullptr;
  float* out = nullptr;
  CK(hipMalloc(&w, w_pieces * 1024));
  CK(hipMalloc(&out, 4096));
  std::vector<unsigned short> h(w_pieces * 512);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 20) % 0x300);   // bf16 values around 0.01 .. 0.1
  CK(hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) {
    run<4, 48, 12, 13, 0>("A: 1 wave/SIMD, 48 MFMA + 12 pieces per wave and chunk", w, w_pieces, out);
    run<4, 48, 12, 13, 1>("A: MFMAs + fragment reads only", w, w_pieces, out);
    run<4, 48, 12, 13, 2>("A: weight stream only (DMA + waits + barrier)", w, w_pieces, out);
    run<8, 24, 6, 7, 0>("B: 2 waves/SIMD, 24 MFMA + 6 pieces per wave and chunk", w, w_pieces, out);
    run<8, 24, 6, 7, 1>("B: MFMAs + fragment reads only", w, w_pieces, out);
    run<8, 24, 6, 7, 2>("B: weight stream only (DMA + waits + barrier)", w, w_pieces, out);
    run<8, 24, 6, 13, 0>("B': 2 waves/SIMD, 13 accumulators (as if registers allowed)", w, w_pieces, out);
  }
  return 0;
}
