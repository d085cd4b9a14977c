// Developer probe: wave-level timeline of the decode GEMM kernels at the batch-256 shapes (where does a 3-13 us kernel
// spend its time?).  Builds the product kernel source with MSH_TIMELINE:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imoonshine_amd/csrc -fno-slp-vectorize \
//         -Xclang -target-feature -Xclang -packed-fp32-ops tools/dec_gemm_timeline.hip -o tools/build/dec_gemm_timeline
#define MSH_TIMELINE 1
#include "../moonshine_amd/csrc/k_gemm_dec.hip"

#include <stdio.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using namespace msh;

static void report(const char* name, const std::vector<unsigned long long>& t, int blocks) {
  // per point: min / median / max over waves, relative to the earliest start of the launch (10 ns ticks -> us)
  unsigned long long t0 = ~0ull;
  for (int b = 0; b < blocks * 8; ++b) if (t[(size_t)b * 8] != 0) t0 = std::min(t0, t[(size_t)b * 8]);
  printf("%-28s blocks %4d |", name, blocks);
  static const char* pt[] = {"start", "loads issued", "LN sums", "LN sync", "mfma+lds", "sync", "stored"};
  for (int p = 0; p < 7; ++p) {
    std::vector<double> v;
    for (int b = 0; b < blocks * 8; ++b) {
      const unsigned long long x = t[(size_t)b * 8 + p];
      if (x != 0) v.push_back((double)(x - t0) * 0.01);
    }
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    printf(" %s %.2f/%.2f/%.2f |", pt[p], v.front(), v[v.size() / 2], v.back());
  }
  printf("\n");
}

int main() {
  const int M = 256, D = 416, F = 1664;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *H, *q, *bias;
  bf16_t *W, *ao, *z, *cK, *cV;
  int* pos;
  CK(hipMalloc(&H, M * D * 4)); CK(hipMalloc(&q, M * D * 4)); CK(hipMalloc(&bias, 2 * F * 4));
  CK(hipMalloc(&W, (size_t)2 * F * D * 2)); CK(hipMalloc(&ao, M * D * 2)); CK(hipMalloc(&z, M * F * 2));
  CK(hipMalloc(&cK, (size_t)M * D * 72 * 2)); CK(hipMalloc(&cV, (size_t)M * D * 72 * 2)); CK(hipMalloc(&pos, 64));
  CK(hipMemset(H, 0, M * D * 4)); CK(hipMemset(bias, 0, 2 * F * 4)); CK(hipMemset(W, 0, (size_t)2 * F * D * 2));
  CK(hipMemset(ao, 0, M * D * 2)); CK(hipMemset(z, 0, M * F * 2)); CK(hipMemset(pos, 0, 64));
  float* rope; CK(hipMalloc(&rope, 8192 * 23 * 4 * 2)); CK(hipMemset(rope, 0, 8192 * 23 * 4 * 2));
  RopeParams rp{rope, rope + 8192 * 23, 23, 52, D};
  unsigned long long* tl;
  const int max_blocks = 4096;
  CK(hipMalloc(&tl, (size_t)max_blocks * 64 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &tl, sizeof(tl)));
  std::vector<unsigned long long> host((size_t)max_blocks * 64);
  struct Case { const char* name; int blocks; int id; } cases[] = {
      {"o-proj  N=416 K=416", 16 * 13, 0}, {"cross-q N=416 K=416 LN", 16 * 13, 1}, {"qkv N=1248 K=416 LN", 16 * 20, 2},
      {"fc1 N=3328 K=416 LN tm2", 8 * 52, 3}, {"fc2 N=416 K=1664", 16 * 13, 4},
      {"o-proj nw8", 16 * 13, 10}, {"cross-q nw8", 16 * 13, 11}, {"qkv tn4 nw8", 16 * 20, 12}, {"fc1 tm2 tn4 nw8", 8 * 52, 13},
      {"fc1 tm2 tn8 nw4", 8 * 26, 14}, {"fc1 tm2 tn8 nw8", 8 * 26, 15}, {"fc2 tn2 nw8", 16 * 13, 16}, {"qkv tn6 nw8", 16 * 13, 17},
      {"fc1 tm1 tn8 nw8", 16 * 26, 18}};
  for (auto& c : cases) {
    auto launch = [&] {
      switch (c.id) {
        case 0: dec_gemm_resid(ao, W, nullptr, M, D, D, H, s); break;
        case 1: dec_gemm_ln_f32(H, W, M, D, D, q, s); break;
        case 2: dec_gemm_qkv(H, W, M, D, pos, rp, q, cK, cV, 72, s); break;
        case 3: dec_gemm_ln_swiglu(H, W, bias, M, F, D, z, s); break;
        case 4: dec_gemm_resid(z, W, bias, M, D, F, H, s); break;
        case 10: launch_fm<2, false, EpiDecResidFm<false>, 1, 8>(ao, W, M, D, D, EpiDecResidFm<false>{H, D / 32, nullptr}, s); break;
        case 11: launch_fm<2, true, EpiF32, 1, 8>(H, W, M, D, D, EpiF32{q, D}, s); break;
        case 12: launch_fm<4, true, EpiDecQkv, 1, 8>(H, W, M, 3 * D, D, EpiDecQkv{q, cK, cV, pos, rp, 72}, s); break;
        case 13: launch_fm<4, true, EpiSwiGLUFm, 2, 8>(H, W, M, 2 * F, D, EpiSwiGLUFm{z, F / 32, bias}, s); break;
        case 14: launch_fm<8, true, EpiSwiGLUFm, 2, 4>(H, W, M, 2 * F, D, EpiSwiGLUFm{z, F / 32, bias}, s); break;
        case 15: launch_fm<8, true, EpiSwiGLUFm, 2, 8>(H, W, M, 2 * F, D, EpiSwiGLUFm{z, F / 32, bias}, s); break;
        case 16: launch_fm<2, false, EpiDecResidFm<true>, 1, 8>(z, W, M, D, F, EpiDecResidFm<true>{H, D / 32, bias}, s); break;
        case 17: launch_fm<6, true, EpiDecQkv, 1, 8>(H, W, M, 3 * D, D, EpiDecQkv{q, cK, cV, pos, rp, 72}, s); break;
        case 18: launch_fm<8, true, EpiSwiGLUFm, 1, 8>(H, W, M, 2 * F, D, EpiSwiGLUFm{z, F / 32, bias}, s); break;
      }
    };
    for (int i = 0; i < 20; ++i) launch();   // warm; the last launch's stamps are what is read back
    CK(hipStreamSynchronize(s));
    CK(hipMemsetAsync(tl, 0, (size_t)max_blocks * 64 * 8, s));
    for (int i = 0; i < 8; ++i) launch();
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(host.data(), tl, host.size() * 8, hipMemcpyDeviceToHost));
    report(c.name, host, c.blocks);
  }
  return 0;
}
