// Keys-side queries of the absorbed decode cross-attention (k_xattn.hip) in two stages, for gfx950.
//
// The absorbed form attends over the encoder output E itself, so its query is qt_h = Wk_h^T q_h: the head's key
// projection (reference graph: encoder_attn.k_proj, transformers modeling_moonshine.py:265-330) applied to the head's
// ordinary query q_h = scale * Wq_h LN(x).  The first version multiplied the two matrices at load (Wqk = Wk_h^T Wq_h,
// [heads * D][D]: fc1's shape) and ran ONE decode GEMM with 8x the flops and 8x the weight bytes of the two factors
// (dec_gemm_ln_qt, 6.8 us per layer at 256 clips: the GEMM's load phase is bound by what a CU's vector-memory path
// ingests, 160 KB per workgroup).  The product has rank head_dim, so this kernel keeps the factors apart:
//
//   stage 1   q_h  [16 rows x 64]   = LN(x) [16 x D] . Wq_h'^T        (Wq' = scale * Wq * diag(gamma); rows j >= head_dim are zero)
//   stage 2   qt_h [16 rows x D/S]  = q_h [16 x 64] . Wk_h[:, d-range]  (S = 2 workgroups split the D output columns of a head)
//
// One workgroup = (16-row tile, head, half of D): 4 waves.  Stage 1 splits K = D over the waves exactly as gemm_dec_kernel
// does (same LayerNorm: shifted single-pass moments, one exchange), the four partial 64 x 16 tiles are summed through LDS
// in a fixed order and EVERY wave keeps the whole q_h.  The MFMA accumulator layout of stage 1 (lane = row, 4 consecutive j)
// is, up to the order of the k slots, the operand layout of stage 2, and that order is folded into how Wk is packed at
// load (pack_crossq_wk) -- nothing is transposed on the device.  q_h goes into stage 2 as two bf16 halves (value and
// rounding residual: ~16 mantissa bits), Wk as stored bf16: the same roundings as the projected-K/V path, where q stays
// fp32 and K = E Wk^T is formed from bf16 Wk.  All loads of both stages are issued up front (one memory round trip);
// 104 KB per workgroup instead of 160, a quarter of the weight bytes in HBM.  Output: EpiQtFrag's operand order.
#include <vector>

#include "gemm_common.h"

namespace msh {
namespace {

template <int D, int DSPLIT>
__global__ __launch_bounds__(256) void dec_crossq2_kernel(const float* __restrict__ H,     // FM32 [M16][D]
                                                         const bf16_t* __restrict__ W1,   // FM [heads * 64][D]
                                                         const bf16_t* __restrict__ W2,   // [heads][D / 16][2][64][8]
                                                         int M, EpiQtFrag epi) {
  constexpr int KS = D / 32, KW = (KS + 3) / 4, KFULL = KS / 4;
  constexpr int DT = D / 16, DTW = DT / DSPLIT, TW = (DTW + 3) / 4;   // d tiles of the head / of the workgroup / per wave
  static_assert(DT % DSPLIT == 0, "the head's column tiles must split evenly");
  __shared__ __attribute__((aligned(16))) float4 part[4][4][64];
  __shared__ float2 stat[4][16];
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // block -> (head, half, row tile): the head is the XCD (workgroup b runs on XCD b % 8: observed placement, used for speed
  // only), so a head's two weight factors are fetched into ONE XCD's L2
  const int h = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int half = rest % DSPLIT, mt = rest / DSPLIT;
  const int m0 = mt * 16;

  int ks[KW];
  bool kv[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int s = wave + 4 * i;
    kv[i] = i < KFULL ? true : s < KS;
    ks[i] = kv[i] ? s : KS - 1;
  }
  // ---- every load of this wave, unconditionally (masked k-steps / tiles re-read a valid one) ----
  uint4 w1[KW][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int i = 0; i < KW; ++i)
      w1[i][nt] = *reinterpret_cast<const uint4*>(W1 + ((((long)h * 4 + nt) * KS + ks[i]) * 64 + lane) * 8);
  float xv[KW][8];
  const float* xt = H + (long)mt * KS * 512;
  const float x0 = xt[li * 4];   // element (row, 0)
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const float* xf = xt + ((long)ks[i] * 128 + lane) * 4;
    const float4 a = *reinterpret_cast<const float4*>(xf);
    const float4 b = *reinterpret_cast<const float4*>(xf + 256);
    xv[i][0] = a.x; xv[i][1] = a.y; xv[i][2] = a.z; xv[i][3] = a.w;
    xv[i][4] = b.x; xv[i][5] = b.y; xv[i][6] = b.z; xv[i][7] = b.w;
  }
  int dts[TW];
  bool dv[TW];
  uint4 w2[TW][2];
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const int o = wave + 4 * t;
    dv[t] = o < DTW;
    dts[t] = half * DTW + (dv[t] ? o : DTW - 1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
      w2[t][s] = *reinterpret_cast<const uint4*>(W2 + ((((long)h * DT + dts[t]) * 2 + s) * 64 + lane) * 8);
  }
  __builtin_amdgcn_sched_barrier(0);   // no load sinks below the LayerNorm arithmetic (a second round trip otherwise)

  // ---- LayerNorm of the wave's k-slices: shifted single-pass moments, one exchange (gemm_dec_kernel's) ----
  {
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      const float keep = kv[i] ? 1.0f : 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (xv[i][e] - x0) * keep;
        xv[i][e] = d;
        sum += d;
        sq += d * d;
      }
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    if (kg == 0) stat[wave][li] = make_float2(sum, sq);
  }
  __syncthreads();
  bf16x8 afrag[KW];
  {
    const float2 s0 = stat[0][li], s1 = stat[1][li], s2 = stat[2][li], s3 = stat[3][li];
    const float mean = ((s0.x + s1.x) + (s2.x + s3.x)) * (1.0f / (float)D);
    float var = ((s0.y + s1.y) + (s2.y + s3.y)) * (1.0f / (float)D) - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      const float scale = kv[i] ? rstd : 0.0f;
      uint4 q;
      q.x = pack_bf16x2((xv[i][0] - mean) * scale, (xv[i][1] - mean) * scale);
      q.y = pack_bf16x2((xv[i][2] - mean) * scale, (xv[i][3] - mean) * scale);
      q.z = pack_bf16x2((xv[i][4] - mean) * scale, (xv[i][5] - mean) * scale);
      q.w = pack_bf16x2((xv[i][6] - mean) * scale, (xv[i][7] - mean) * scale);
      afrag[i] = *reinterpret_cast<bf16x8*>(&q);
    }
  }
  // ---- stage 1: partial q_h^T [64 j x 16 rows] over this wave's k-steps ----
  f32x4 acc1[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc1[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KW; ++i)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w1[i][nt]), afrag[i], acc1[nt], 0, 0, 0);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) part[wave][nt][lane] = make_float4(acc1[nt][0], acc1[nt][1], acc1[nt][2], acc1[nt][3]);
  __syncthreads();
  // every wave sums the four partials of all four tiles in the same fixed order: lane (li, kg) then holds
  // q[row li][j = 16 nt + 4 kg + r], r = 0..3
  float q[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 p0 = part[0][nt][lane], p1 = part[1][nt][lane], p2 = part[2][nt][lane], p3 = part[3][nt][lane];
    q[nt][0] = (p0.x + p1.x) + (p2.x + p3.x);
    q[nt][1] = (p0.y + p1.y) + (p2.y + p3.y);
    q[nt][2] = (p0.z + p1.z) + (p2.z + p3.z);
    q[nt][3] = (p0.w + p1.w) + (p2.w + p3.w);
  }
  // stage 2's operand: k-step s holds the j of tiles 2 s (slots 0-3) and 2 s + 1 (slots 4-7) -- pack_crossq_wk's order
  bf16x8 bhi[2], blo[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    uint4 hi, lo;
    hi.x = pack_bf16x2(q[2 * s][0], q[2 * s][1]);
    hi.y = pack_bf16x2(q[2 * s][2], q[2 * s][3]);
    hi.z = pack_bf16x2(q[2 * s + 1][0], q[2 * s + 1][1]);
    hi.w = pack_bf16x2(q[2 * s + 1][2], q[2 * s + 1][3]);
    lo.x = pack_bf16x2(q[2 * s][0] - __uint_as_float(hi.x << 16), q[2 * s][1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack_bf16x2(q[2 * s][2] - __uint_as_float(hi.y << 16), q[2 * s][3] - __uint_as_float(hi.y & 0xffff0000u));
    lo.z = pack_bf16x2(q[2 * s + 1][0] - __uint_as_float(hi.z << 16), q[2 * s + 1][1] - __uint_as_float(hi.z & 0xffff0000u));
    lo.w = pack_bf16x2(q[2 * s + 1][2] - __uint_as_float(hi.w << 16), q[2 * s + 1][3] - __uint_as_float(hi.w & 0xffff0000u));
    bhi[s] = *reinterpret_cast<bf16x8*>(&hi);
    blo[s] = *reinterpret_cast<bf16x8*>(&lo);
  }
  // ---- stage 2: qt_h^T [16 d x 16 rows] for this wave's d tiles ----
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w2[t][s]), bhi[s], a, 0, 0, 0);
      b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w2[t][s]), blo[s], b, 0, 0, 0);
    }
    f32x4 v;
    v[0] = a[0] + b[0]; v[1] = a[1] + b[1]; v[2] = a[2] + b[2]; v[3] = a[3] + b[3];
    const int m = m0 + li, n = h * D + dts[t] * 16 + kg * 4;
    if (dv[t] && m < M) epi.n4(m, n, v);
  }
}

}  // namespace

// Wk [D rows (h, j)][D] fp32 -> the stage-2 operand order [heads][D / 16 tiles][2 k-steps][64 lanes][8]: lane (li, kg) of
// tile dt, k-step s, slot e holds Wk[h * dh + j][dt * 16 + li] with j = 32 s + 16 (e / 4) + 4 kg + e % 4 (zero for j >= dh)
void pack_crossq_wk(const float* Wk, int D, int heads, bf16_t* out) {
  const int dh = D / heads, DT = D / 16;
  for (int h = 0; h < heads; ++h)
    for (int dt = 0; dt < DT; ++dt)
      for (int s = 0; s < 2; ++s)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int li = lane & 15, kg = lane >> 4;
            const int j = 32 * s + 16 * (e >> 2) + 4 * kg + (e & 3);
            const float v = j < dh ? Wk[(size_t)(h * dh + j) * D + dt * 16 + li] : 0.f;
            out[(((size_t)(h * DT + dt) * 2 + s) * 64 + lane) * 8 + e] = f32_to_bf16(v);
          }
}

bool crossq2_supported(int D, int heads) { return heads == 8 && (D == 416 || D == 288) && D / heads <= 64; }

void dec_crossq2(const float* H, const bf16_t* W1, const bf16_t* W2, int M, int heads, int D, bf16_t* qf, hipStream_t s) {
  if (!crossq2_supported(D, heads)) throw std::runtime_error("dec_crossq2: unsupported shape");
  const int m_tiles = (M + 15) / 16;
  EpiQtFrag epi{qf, D};
  if (D == 416) {
    MSH_LAUNCH((dec_crossq2_kernel<416, 2>), dim3(8 * 2 * m_tiles), dim3(256), 0, s, H, W1, W2, M, epi);
  } else {
    MSH_LAUNCH((dec_crossq2_kernel<288, 2>), dim3(8 * 2 * m_tiles), dim3(256), 0, s, H, W1, W2, M, epi);
  }
}


// Test / microbenchmark hook (msh_test_crossq2): x [M][D] fp32 row-major (the residual stream), wq [D][D] = scale * Wq *
// diag(gamma) rows (h, j), wk [D][D] rows (h, j); qt_out [M][8 * D] fp32 = value + rounding residual of the kernel's output.
// Returns ms per launch over `iters` launches (0 = one launch, not timed).
float crossq2_host(const float* x, const float* wq, const float* wk, int M, int D, float* qt_out, int iters) {
  const int heads = 8;
  if (!crossq2_supported(D, heads)) throw std::runtime_error("crossq2: unsupported width");
  const int dh = D / heads, KS = D / 32, M16 = (M + 15) / 16 * 16;
  std::vector<float> hfm((size_t)M16 * D, 0.f);
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < D; ++k) hfm[(size_t)fm32(m, k, KS)] = x[(size_t)m * D + k];
  std::vector<bf16_t> w1((size_t)heads * 64 * D, f32_to_bf16(0.f)), w2((size_t)heads * (D / 16) * 2 * 64 * 8);
  for (int h = 0; h < heads; ++h)
    for (int j = 0; j < dh; ++j)
      for (int k = 0; k < D; ++k) w1[(size_t)fm16(h * 64 + j, k, KS)] = f32_to_bf16(wq[(size_t)(h * dh + j) * D + k]);
  pack_crossq_wk(wk, D, heads, w2.data());
  float* dH = nullptr;
  bf16_t *d1 = nullptr, *d2 = nullptr, *dq = nullptr;
  const size_t qn = (size_t)M16 * heads * D * 2;
  MSH_HIP(hipMalloc(&dH, hfm.size() * 4));
  MSH_HIP(hipMalloc(&d1, w1.size() * 2));
  MSH_HIP(hipMalloc(&d2, w2.size() * 2));
  MSH_HIP(hipMalloc(&dq, qn * 2));
  MSH_HIP(hipMemcpy(dH, hfm.data(), hfm.size() * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(d1, w1.data(), w1.size() * 2, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(d2, w2.data(), w2.size() * 2, hipMemcpyHostToDevice));
  MSH_HIP(hipMemset(dq, 0, qn * 2));
  dec_crossq2(dH, d1, d2, M, heads, D, dq, 0);
  MSH_HIP(hipDeviceSynchronize());
  float ms = 0.f;
  if (iters > 0) {
    hipEvent_t a, b;
    MSH_HIP(hipEventCreate(&a));
    MSH_HIP(hipEventCreate(&b));
    MSH_HIP(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) dec_crossq2(dH, d1, d2, M, heads, D, dq, 0);
    MSH_HIP(hipEventRecord(b, 0));
    MSH_HIP(hipEventSynchronize(b));
    MSH_HIP(hipEventElapsedTime(&ms, a, b));
    ms /= (float)iters;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
  std::vector<bf16_t> q16(qn);
  MSH_HIP(hipMemcpy(q16.data(), dq, qn * 2, hipMemcpyDeviceToHost));
  for (int m = 0; m < M; ++m)
    for (int h = 0; h < heads; ++h)
      for (int d = 0; d < D; ++d) {
        const size_t frag = ((size_t)m * KS + d / 32) * 512 + d % 32;   // EpiQtFrag's order
        qt_out[((size_t)m * heads + h) * D + d] = bf16_to_f32(q16[frag + h * 32]) + bf16_to_f32(q16[frag + (h + 8) * 32]);
      }
  (void)hipFree(dH);
  (void)hipFree(d1);
  (void)hipFree(d2);
  (void)hipFree(dq);
  return ms;
}

}  // namespace msh
