// Decode-path GEMMs for gfx950 (M = batch rows, weights streamed): see the two kernels below.
#include <stdlib.h>
#include <string.h>

#include "gemm_common.h"

namespace msh {
namespace {

// Developer instrumentation (tools/dec_gemm_timeline.hip builds this file with MSH_TIMELINE): wave-level time stamps
// (100 MHz s_memrealtime, comparable across CUs) at the phase boundaries of gemm_dec_kernel.  Compiled out otherwise.
#ifdef MSH_TIMELINE
__device__ unsigned long long* g_timeline = nullptr;   // [blocks][8 waves][8 points]
#define MSH_TL(i)                                                                                          \
  do {                                                                                                     \
    if (lane == 0 && g_timeline != nullptr)                                                                \
      g_timeline[((size_t)blockIdx.x * 8 + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();            \
  } while (0)
#else
#define MSH_TL(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Decode GEMM (M = batch rows): fragment-direct, split-K inside the workgroup.
//
// A workgroup owns one 16 x (16*TN) output tile; its 4 waves split the K loop (wave w takes the 32-wide
// k-steps w, w+4, ...), each loading its MFMA fragments straight from global memory.  The kernel is written
// so that it has exactly ONE memory round trip on its critical path:
//   * every load of a wave (W fragments, A fragments, the epilogue's residual / bias / RoPE factors) is issued
//     up front and UNCONDITIONALLY -- a k-step or an output tile that does not exist for this wave (KS or the
//     tile count not a multiple of 4), a row beyond M or a column beyond N is clamped to a valid address and
//     its contribution zeroed / its result dropped.  The first version guarded those loads with `if (s < KS)`
//     on the (runtime) wave index: hipcc turned that into a branchy CFG with `s_waitcnt vmcnt(0)` between the
//     load groups -- two to three serialised round trips per kernel (4.8 us for a GEMM whose chain floor,
//     tools/launch_floor.hip, is 1.7 us).
//   * LN = true: A is the fp32 residual stream [M][K]; LayerNorm (no bias, eps 1e-5) is fused into the
//     fragment build with ONE cross-wave exchange: shifted single-pass moments (d = x - x[row][0], so a large
//     common offset cannot cancel catastrophically), sum(d) and sum(d^2) combined through LDS in a fixed order.
//     The LayerNorm scale gamma is folded into W at load time (W' = W * diag(gamma)).
// Partial tiles are summed through LDS in a fixed order (deterministic, independent of TN / TM).
// ------------------------------------------------------------------------------------------------
// FM = true: W, A and (through the epilogue) the outputs are in the MFMA-fragment-major layouts of kernels.h: every
// wave-level load is one contiguous 1 KiB run.  FM = false: row-major operands (the streaming decoder's path).
// NW = waves per workgroup (4 or 8): the K loop is split NW ways.  With 8 waves every wave issues half as many loads,
// which is what the first microsecond of these kernels consists of (tools/dec_gemm_timeline.hip).
template <int KS, int TN, bool LN, class Epi, int TM = 1, bool FM = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_dec_kernel(const void* __restrict__ Aptr, long lda,
                                                       const float* __restrict__ gamma,
                                                       const bf16_t* __restrict__ W, int M, int N, int n_tiles,
                                                       Epi epi) {
  // TM = 2: the workgroup owns 32 rows (two MFMA row tiles) and every W fragment it loads feeds two MFMAs -- the
  // weights are re-read M/32 instead of M/16 times.
  constexpr int K = 32 * KS;
  constexpr int KW = (KS + NW - 1) / NW;   // k-steps per wave (upper bound)
  constexpr int KFULL = KS / NW;           // k-steps i < KFULL exist for every wave
  __shared__ __attribute__((aligned(16))) float4 part[NW][TM * TN][64];
  __shared__ float2 stat[NW][16 * TM];
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Block -> tile.  n_tiles > 0: XCD-aware order (launch grid dec_grid()): workgroup b runs on XCD b % 8 (observed placement,
  // used for speed only), and column tile nt is given to XCD nt % 8 with ALL its row tiles -- every weight byte then crosses
  // the fabric into ONE XCD's L2 instead of into four to eight of them (the decode weights, 77-88 MB per step, never stay in
  // the 4 MB L2s from one step to the next: each GEMM starts L2-cold and its first round trip is the fabric's).  Blocks
  // beyond an XCD's share exit at once.  n_tiles < 0: plain row-major order over -n_tiles column tiles (MSH_DEC_XCD=0).
  int mt_, nt_;
  if (n_tiles > 0) {
    const int m_tiles = (M + 16 * TM - 1) / (16 * TM);
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int cnt = (n_tiles - x + 7) >> 3;   // column tiles with nt % 8 == x
    if (i >= cnt * m_tiles) return;
    const int c = i / m_tiles;
    mt_ = i - c * m_tiles;
    nt_ = x + 8 * c;
  } else {
    n_tiles = -n_tiles;
    mt_ = blockIdx.x / n_tiles;
    nt_ = blockIdx.x - mt_ * n_tiles;
  }
  const int m0 = mt_ * (16 * TM), n0 = nt_ * (16 * TN);
  MSH_TL(0);
  int gm[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    gm[t] = m0 + t * 16 + li;
    gm[t] = gm[t] < M ? gm[t] : M - 1;
  }
  // k-step i of this wave: s = wave + NW i (wave-uniform); a step past KS reads step KS - 1 again and is masked
  int ks[KW];
  bool kv[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int s = wave + NW * i;
    kv[i] = i < KFULL ? true : s < KS;
    ks[i] = kv[i] ? s : KS - 1;
  }

  // ---- every load of this wave, unconditionally ----
  uint4 wreg[KW][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    if constexpr (FM) {
      int tn = (n0 >> 4) + j;                     // N is a multiple of 16 (checked by the launcher)
      tn = tn < (N >> 4) ? tn : (N >> 4) - 1;
#pragma unroll
      for (int i = 0; i < KW; ++i)
        wreg[i][j] = *reinterpret_cast<const uint4*>(W + (((long)tn * KS + ks[i]) * 64 + lane) * 8);
    } else {
      int gn = n0 + j * 16 + li;
      gn = gn < N ? gn : N - 1;
      const bf16_t* wrow = W + (long)gn * K + kg * 8;
#pragma unroll
      for (int i = 0; i < KW; ++i) wreg[i][j] = *reinterpret_cast<const uint4*>(wrow + ks[i] * 32);
    }
  }
  float xv[LN ? TM : 1][LN ? KW : 1][8];
  float x0[LN ? TM : 1];
  uint4 araw[LN ? 1 : TM][LN ? 1 : KW];
  int mt[TM];   // FM: row tile of (m0, t), clamped to the last tile that holds a valid row
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    mt[t] = (m0 >> 4) + t;
    mt[t] = mt[t] <= ((M - 1) >> 4) ? mt[t] : ((M - 1) >> 4);
  }
  if constexpr (LN) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      if constexpr (FM) {
        const float* xt = reinterpret_cast<const float*>(Aptr) + (long)mt[t] * KS * 512;
        x0[t] = xt[li * 4];   // element (row, 0): k-step 0, half 0, lane = row % 16
#pragma unroll
        for (int i = 0; i < KW; ++i) {
          const float* xf = xt + ((long)ks[i] * 128 + lane) * 4;
          const float4 a = *reinterpret_cast<const float4*>(xf);
          const float4 b = *reinterpret_cast<const float4*>(xf + 256);
          xv[t][i][0] = a.x; xv[t][i][1] = a.y; xv[t][i][2] = a.z; xv[t][i][3] = a.w;
          xv[t][i][4] = b.x; xv[t][i][5] = b.y; xv[t][i][6] = b.z; xv[t][i][7] = b.w;
        }
      } else {
        const float* x = reinterpret_cast<const float*>(Aptr) + (long)gm[t] * lda;
        x0[t] = x[0];
#pragma unroll
        for (int i = 0; i < KW; ++i) {
          const float4 a = *reinterpret_cast<const float4*>(x + kg * 8 + ks[i] * 32);
          const float4 b = *reinterpret_cast<const float4*>(x + kg * 8 + ks[i] * 32 + 4);
          xv[t][i][0] = a.x; xv[t][i][1] = a.y; xv[t][i][2] = a.z; xv[t][i][3] = a.w;
          xv[t][i][4] = b.x; xv[t][i][5] = b.y; xv[t][i][6] = b.z; xv[t][i][7] = b.w;
        }
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      if constexpr (FM) {
        const bf16_t* at = reinterpret_cast<const bf16_t*>(Aptr) + (long)mt[t] * KS * 512;
#pragma unroll
        for (int i = 0; i < KW; ++i) araw[t][i] = *reinterpret_cast<const uint4*>(at + ((long)ks[i] * 64 + lane) * 8);
      } else {
        const bf16_t* a = reinterpret_cast<const bf16_t*>(Aptr) + (long)gm[t] * lda + kg * 8;
#pragma unroll
        for (int i = 0; i < KW; ++i) araw[t][i] = *reinterpret_cast<const uint4*>(a + ks[i] * 32);
      }
    }
  }
  // inputs of the epilogue this wave will run at the end (residual, bias, RoPE factors)
  constexpr int NT = TM * TN;            // output tiles of the workgroup
  constexpr int NE = (NT + NW - 1) / NW;
  typename Epi::Pre epre[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    int o = wave + NW * e;
    o = o < NT ? o : NT - 1;
    const int t = o / TN, j = o - t * TN;
    int m = m0 + t * 16 + li, n = n0 + j * 16 + kg * 4;
    m = m < M ? m : M - 1;
    n = n < N ? n : N - 4;
    epre[e] = epi.pre(m, n);
  }
  // nothing below may be scheduled above this point and no load above may sink below it: without the fence hipcc
  // moved the W loads behind the first LayerNorm arithmetic (= behind a wait for the A loads): a second round trip
  __builtin_amdgcn_sched_barrier(0);
  MSH_TL(1);

  bf16x8 afrag[TM][KW];
  if constexpr (LN) {
    // shifted single-pass moments: d = x - x0 (x0 = the row's first element)
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      float sum = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < KW; ++i) {
        const float keep = kv[i] ? 1.0f : 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (xv[t][i][e] - x0[t]) * keep;
          xv[t][i][e] = d;
          sum += d;
          sq += d * d;
        }
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      sq += __shfl_xor(sq, 16);
      sq += __shfl_xor(sq, 32);
      if (kg == 0) stat[wave][t * 16 + li] = make_float2(sum, sq);
    }
    MSH_TL(2);
    __syncthreads();
    MSH_TL(3);
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int r = t * 16 + li;
      float2 s0 = stat[0][r], s1 = stat[1][r], s2 = stat[2][r], s3 = stat[3][r];
      if constexpr (NW == 8) {
        const float2 s4 = stat[4][r], s5 = stat[5][r], s6 = stat[6][r], s7 = stat[7][r];
        s0.x += s4.x; s0.y += s4.y; s1.x += s5.x; s1.y += s5.y; s2.x += s6.x; s2.y += s6.y; s3.x += s7.x; s3.y += s7.y;
      }
      const float mean = ((s0.x + s1.x) + (s2.x + s3.x)) * (1.0f / (float)K);
      float var = ((s0.y + s1.y) + (s2.y + s3.y)) * (1.0f / (float)K) - mean * mean;
      var = var > 0.f ? var : 0.f;
      const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
      for (int i = 0; i < KW; ++i) {
        const float scale = kv[i] ? rstd : 0.0f;   // a masked k-step contributes exact zeros
        uint4 q;
        q.x = pack_bf16x2((xv[t][i][0] - mean) * scale, (xv[t][i][1] - mean) * scale);
        q.y = pack_bf16x2((xv[t][i][2] - mean) * scale, (xv[t][i][3] - mean) * scale);
        q.z = pack_bf16x2((xv[t][i][4] - mean) * scale, (xv[t][i][5] - mean) * scale);
        q.w = pack_bf16x2((xv[t][i][6] - mean) * scale, (xv[t][i][7] - mean) * scale);
        afrag[t][i] = *reinterpret_cast<bf16x8*>(&q);
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int i = 0; i < KW; ++i) {
        uint4 q = araw[t][i];
        if (i >= KFULL) {  // wave-uniform select, no branch
          q.x = kv[i] ? q.x : 0u; q.y = kv[i] ? q.y : 0u; q.z = kv[i] ? q.z : 0u; q.w = kv[i] ? q.w : 0u;
        }
        afrag[t][i] = *reinterpret_cast<bf16x8*>(&q);
      }
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KW; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int t = 0; t < TM; ++t)
        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wreg[i][j]), afrag[t][i],
                                                             acc[t][j], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      part[wave][t * TN + j][lane] = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
  MSH_TL(4);
  __syncthreads();
  MSH_TL(5);
  // the waves finish the output tiles round-robin: fixed summation order
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int o = wave + NW * e;
    const int oc = o < NT ? o : NT - 1;
    const int t = oc / TN, j = oc - t * TN;
    float4 p0 = part[0][oc][lane], p1 = part[1][oc][lane], p2 = part[2][oc][lane], p3 = part[3][oc][lane];
    if constexpr (NW == 8) {
      const float4 p4 = part[4][oc][lane], p5 = part[5][oc][lane], p6 = part[6][oc][lane], p7 = part[7][oc][lane];
      p0.x += p4.x; p0.y += p4.y; p0.z += p4.z; p0.w += p4.w;
      p1.x += p5.x; p1.y += p5.y; p1.z += p5.z; p1.w += p5.w;
      p2.x += p6.x; p2.y += p6.y; p2.z += p6.z; p2.w += p6.w;
      p3.x += p7.x; p3.y += p7.y; p3.z += p7.z; p3.w += p7.w;
    }
    f32x4 v;
    v[0] = (p0.x + p1.x) + (p2.x + p3.x);
    v[1] = (p0.y + p1.y) + (p2.y + p3.y);
    v[2] = (p0.z + p1.z) + (p2.z + p3.z);
    v[3] = (p0.w + p1.w) + (p2.w + p3.w);
    const int m = m0 + t * 16 + li, n = n0 + j * 16 + kg * 4;
    if (o < NT && m < M && n < N) epi.n4p(m, n, v, epre[e]);
  }
  MSH_TL(6);
}

static bool dec_xcd_order() {
  static const bool on = [] {
    const char* e = dev_getenv("MSH_DEC_XCD");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}
// launch grid and the n_tiles argument of gemm_dec_kernel (see "Block -> tile" there)
static inline unsigned dec_grid(int m_tiles, int n_tiles) {
  return dec_xcd_order() ? 8u * (unsigned)((n_tiles + 7) / 8) * (unsigned)m_tiles : (unsigned)(m_tiles * n_tiles);
}
static inline int dec_ntiles_arg(int n_tiles) { return dec_xcd_order() ? n_tiles : -n_tiles; }

template <int KS, int TN, bool LN, class Epi, int TM = 1>
void launch_dec_cfg(const void* A, long lda, const float* gamma, const bf16_t* W, int M, int N, Epi epi,
                    hipStream_t s) {
  const int m_tiles = (M + 16 * TM - 1) / (16 * TM), n_tiles = (N + 16 * TN - 1) / (16 * TN);
  MSH_LAUNCH((gemm_dec_kernel<KS, TN, LN, Epi, TM>), dim3(dec_grid(m_tiles, n_tiles)), dim3(256), 0, s, A, lda, gamma, W,
                     M, N, dec_ntiles_arg(n_tiles), epi);
}
static int dec_tm2_threshold() {
  static int thr = [] {
    const char* e = dev_getenv("MSH_DEC_TM2_M");
    return e ? atoi(e) : 128;
  }();
  return thr;
}

// K is a compile-time multiple of 32: D (LN-fused and attention-output GEMMs) or F (fc2)
template <int TN, bool LN, class Epi>
void launch_dec(const void* A, long lda, const float* gamma, const bf16_t* W, int M, int N, int K, Epi epi,
                hipStream_t s) {
  if ((N & 3) != 0) throw std::runtime_error("gemm_dec: N must be a multiple of 4");
  switch (K) {
    case 416: return launch_dec_cfg<13, TN, LN, Epi>(A, lda, gamma, W, M, N, epi, s);
    case 1664: return launch_dec_cfg<52, TN, LN, Epi>(A, lda, gamma, W, M, N, epi, s);
    case 288: return launch_dec_cfg<9, TN, LN, Epi>(A, lda, gamma, W, M, N, epi, s);
    case 1152: return launch_dec_cfg<36, TN, LN, Epi>(A, lda, gamma, W, M, N, epi, s);
    case 64: return launch_dec_cfg<2, TN, LN, Epi>(A, lda, gamma, W, M, N, epi, s);
    case 256: return launch_dec_cfg<8, TN, LN, Epi>(A, lda, gamma, W, M, N, epi, s);
    default: throw std::runtime_error("gemm_dec: unsupported K " + std::to_string(K));
  }
}



// FM operands (offline decoder): K = D or F of the offline architectures
template <int TN, bool LN, class Epi, int TM = 1, int NW = 4>
void launch_fm(const void* A, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  if ((N & 15) != 0) throw std::runtime_error("gemm_dec (FM): N must be a multiple of 16");
  const int m_tiles = (M + 16 * TM - 1) / (16 * TM), n_tiles = (N + 16 * TN - 1) / (16 * TN);
#define MSH_FM_CASE(KK)                                                                                              \
  case KK:                                                                                                           \
    MSH_LAUNCH((gemm_dec_kernel<KK / 32, TN, LN, Epi, TM, true, NW>), dim3(dec_grid(m_tiles, n_tiles)), dim3(64 * NW), 0, s, A, \
                       (long)0, (const float*)nullptr, W, M, N, dec_ntiles_arg(n_tiles), epi);                       \
    return;
  switch (K) {
    MSH_FM_CASE(416)
    MSH_FM_CASE(1664)
    MSH_FM_CASE(288)
    MSH_FM_CASE(1152)
    MSH_FM_CASE(64)
    MSH_FM_CASE(256)
    default: throw std::runtime_error("gemm_dec (FM): unsupported K " + std::to_string(K));
  }
#undef MSH_FM_CASE
}

// FM operands at the streaming decoder's widths (AR steps of decode_full): K = decoder width or its ffn
template <int TN, bool LN, class Epi, int TM = 1, int NW = 4>
bool launch_sfm(const void* A, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  if ((N & 15) != 0) return false;
  const int m_tiles = (M + 16 * TM - 1) / (16 * TM), n_tiles = (N + 16 * TN - 1) / (16 * TN);
#define MSH_SFM_CASE(KK)                                                                                             \
  case KK:                                                                                                           \
    MSH_LAUNCH((gemm_dec_kernel<KK / 32, TN, LN, Epi, TM, true, NW>), dim3(dec_grid(m_tiles, n_tiles)), dim3(64 * NW), 0, s, A, \
                       (long)0, (const float*)nullptr, W, M, N, dec_ntiles_arg(n_tiles), epi);                       \
    return true;
  switch (K) {
    MSH_SFM_CASE(96)
    MSH_SFM_CASE(320)
    MSH_SFM_CASE(640)
    default: break;
  }
  if constexpr (!LN) {
    switch (K) {
      MSH_SFM_CASE(192)
      MSH_SFM_CASE(1280)
      MSH_SFM_CASE(2560)
      default: break;
    }
  }
#undef MSH_SFM_CASE
  return false;
}

}  // namespace

// Narrow outputs (N = decoder width) at streaming batch sizes give too few 16 x 32 tiles to occupy the chip (M = 64,
// N = 640: 80 workgroups on 256 CUs, each streaming its whole K x 32 weight slice alone): halve the tile width then.
// The per-element summation order does not depend on the tile width, so results are bit-identical.
static bool few_tiles(int M, int N) {
  static const int thr = [] {
    const char* e = dev_getenv("MSH_DEC_NARROW_TILES");
    return e ? atoi(e) : 192;
  }();
  return ((M + 15) / 16) * ((N + 31) / 32) < thr;
}
// Throughput shapes.  The tile shapes below were chosen for the LATENCY of a decode step that has the GPU to itself: about one
// round of ~200 workgroups.  Beside other lanes the step is not alone, and what a launch costs the others is the operand traffic
// it pulls through L2 and the CUs it holds: 32-row workgroups (half the W re-reads, half the workgroups) measure -3 % serial and
// +1.8 % with four batches in flight at 256 clips (93.8 -> 95.5 k audio-s/s).  The engine says which regime a step is enqueued
// for (dec_gemm_prefer_throughput, per host thread: every lane enqueues and captures on its own); the summation order per
// output element does not depend on the shape, so ids are the same either way.  MSH_DEC_FAT_TILES=0 / 1 forces it.
static thread_local bool t_dec_fat_tiles = false;
void dec_gemm_prefer_throughput(bool on) { t_dec_fat_tiles = on; }
static bool dec_fat_tiles() {
  static const int forced = [] {
    const char* e = dev_getenv("MSH_DEC_FAT_TILES");
    return e == nullptr ? -1 : (e[0] == '1' ? 1 : 0);
  }();
  return forced >= 0 ? forced == 1 : t_dec_fat_tiles;
}
// One row tile (M <= 16, the single-clip latency case): 16-column tiles double the workgroups that share a GEMM's weight
// stream (o-proj 13 -> 26, qkv 39 -> 78, fc1 104 -> 208); measured p50 of a 10 s clip 20.8 -> 19.0 ms (decode 19.4 -> 17.6).
static bool narrow_small_batch(int M) {
  static const int thr = [] {
    const char* e = dev_getenv("MSH_DEC_NARROW_M");
    return e ? atoi(e) : 16;
  }();
  return M <= thr;
}

// ---- offline decoder: FM operands.  Tile shapes by batch: a GEMM wants >= ~200 workgroups (all 256 CUs pulling their
// share of the weights) but no second round of workgroups; the per-element summation order does not depend on the tile
// shape, so ids are identical across batch sizes. ----
void dec_gemm_qkv(const float* H, const bf16_t* W, int M, int D, const int* pos_ptr, RopeParams rp, float* q,
                  bf16_t* cacheK, bf16_t* cacheV, int Smax, hipStream_t s) {
  EpiDecQkv epi{q, cacheK, cacheV, pos_ptr, rp, Smax};
  // (timeline, M = 256: 96-column tiles = 208 workgroups, one round on 256 CUs, finish 0.4 us before 64-column tiles)
  const bool qkv_tm2 = dec_fat_tiles();   // 32 x 96 tiles beside other lanes
  if (M >= 192 && (3 * D) % 96 == 0 && qkv_tm2)
    launch_fm<6, true, EpiDecQkv, 2>(H, W, M, 3 * D, D, epi, s);
  else if (M >= 192 && (3 * D) % 96 == 0)
    launch_fm<6, true>(H, W, M, 3 * D, D, epi, s);
  else if (M >= 96)  // wide column tiles (64) halve the per-row-tile LayerNorm / A reloads of the big-N GEMMs
    launch_fm<4, true>(H, W, M, 3 * D, D, epi, s);
  else if (narrow_small_batch(M) || few_tiles(M, 3 * D))
    launch_fm<1, true>(H, W, M, 3 * D, D, epi, s);
  else
    launch_fm<2, true>(H, W, M, 3 * D, D, epi, s);
}
void dec_gemm_ln_f32(const float* H, const bf16_t* W, int M, int N, int D, float* out, hipStream_t s) {
  EpiF32 epi{out, N};
  if (few_tiles(M, N))
    launch_fm<1, true>(H, W, M, N, D, epi, s);
  else
    launch_fm<2, true>(H, W, M, N, D, epi, s);
}
void dec_gemm_ln_qt(const float* H, const bf16_t* W, int M, int heads, int D, bf16_t* qf, hipStream_t s) {
  // N = heads * D: fc1's shape, fc1's tiles (see dec_gemm_ln_swiglu)
  const int N = heads * D;
  EpiQtFrag epi{qf, D};
  if (M >= 192 && N % 128 == 0)
    launch_fm<8, true, EpiQtFrag, 2>(H, W, M, N, D, epi, s);
  else if (M >= 96)
    launch_fm<4, true>(H, W, M, N, D, epi, s);
  else
    launch_fm<2, true>(H, W, M, N, D, epi, s);
}
void dec_gemm_ln_swiglu(const float* H, const bf16_t* W, const float* bias, int M, int F, int D, bf16_t* z,
                        hipStream_t s) {
  EpiSwiGLUFm epi{z, F / 32, bias};
  // 32-row workgroups pay off only where there are many column tiles (r01t, M = 256: fc1 14.1 -> 12.4 us, but
  // qkv 9.8 -> 11.1 and cross-q 7.9 -> 11.1 us with half as many workgroups in flight)
  // (timeline, M = 256: 32 x 128 tiles = 208 workgroups in one round beat 32 x 64 = 416 in two by 0.4 us)
  if (M >= 192 && (2 * F) % 128 == 0)
    launch_fm<8, true, EpiSwiGLUFm, 2>(H, W, M, 2 * F, D, epi, s);
  else if (M >= dec_tm2_threshold())
    launch_fm<4, true, EpiSwiGLUFm, 2>(H, W, M, 2 * F, D, epi, s);
  else if (M >= 96)
    launch_fm<4, true>(H, W, M, 2 * F, D, epi, s);
  else if (narrow_small_batch(M))
    launch_fm<1, true>(H, W, M, 2 * F, D, epi, s);
  else
    launch_fm<2, true>(H, W, M, 2 * F, D, epi, s);
}
// K = heads * D (the context of the absorbed cross-attention, k_xattn.hip): 72 / 104 k-steps, split over EIGHT waves so that a
// wave's up-front load phase stays at 13 k-steps of W and A (the 4-wave form would hold 26 x (TN + 1) fragments in registers).
template <int TN, class Epi, int TM = 1>
static void launch_fm_wide_k(const bf16_t* A, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  if ((N & 15) != 0) throw std::runtime_error("gemm_dec (FM, wide K): N must be a multiple of 16");
  const int m_tiles = (M + 16 * TM - 1) / (16 * TM), n_tiles = (N + 16 * TN - 1) / (16 * TN);
  switch (K) {
    case 3328:
      MSH_LAUNCH((gemm_dec_kernel<104, TN, false, Epi, TM, true, 8>), dim3(dec_grid(m_tiles, n_tiles)), dim3(512), 0, s, (const void*)A,
                 (long)0, (const float*)nullptr, W, M, N, dec_ntiles_arg(n_tiles), epi);
      return;
    case 2304:
      MSH_LAUNCH((gemm_dec_kernel<72, TN, false, Epi, TM, true, 8>), dim3(dec_grid(m_tiles, n_tiles)), dim3(512), 0, s, (const void*)A,
                 (long)0, (const float*)nullptr, W, M, N, dec_ntiles_arg(n_tiles), epi);
      return;
    default: throw std::runtime_error("gemm_dec (FM, wide K): unsupported K " + std::to_string(K));
  }
}
static int wide_k_tn() {   // developer knob: column tiles per workgroup of the wide-K residual GEMM (1 or 2)
  static const int v = [] {
    const char* e = dev_getenv("MSH_XATTN_G2_TN");
    // round 6, M = 256 in a replayed graph: 5.2 us with two column tiles per workgroup (208 workgroups, half the A re-reads
    // through L2) against 5.9 with one (416) -- round 4 had measured the opposite on the kernel of that time
    return e != nullptr && e[0] == '1' ? 1 : 2;
  }();
  return v;
}
template <bool BIAS>
static void dec_gemm_resid_t(const bf16_t* A, const bf16_t* W, const float* bias, int M, int N, int K, float* H,
                             hipStream_t s) {
  EpiDecResidFm<BIAS> epi{H, N / 32, bias};
  if (K == 3328 || K == 2304) {
    // 32-row workgroups from 64 rows on: x 16 columns for a lone engine (4.9 us in the decode graph at 256 clips; 16 x 16:
    // 5.9, 16 x 32: 5.2), x 32 columns beside other lanes (5.8 us alone, but half the operand re-reads through L2)
    static const int wide_tm = [] {   // MSH_XATTN_G2_TM=1: the 16-row shapes of round 4 (MSH_XATTN_G2_TN columns tiles)
      const char* e = dev_getenv("MSH_XATTN_G2_TM");
      return e != nullptr && e[0] == '1' ? 1 : 2;
    }();
    if (wide_tm == 2 && M >= 64) {
      if (!dec_fat_tiles()) launch_fm_wide_k<1, EpiDecResidFm<BIAS>, 2>(A, W, M, N, K, epi, s);
      else launch_fm_wide_k<2, EpiDecResidFm<BIAS>, 2>(A, W, M, N, K, epi, s);
      return;
    }
    if (wide_k_tn() == 1 || few_tiles(M, N)) launch_fm_wide_k<1>(A, W, M, N, K, epi, s);
    else launch_fm_wide_k<2>(A, W, M, N, K, epi, s);
    return;
  }
  const bool resid_tm2 = dec_fat_tiles();   // 32 x 32 tiles at large batches (o-proj, fc2) beside other lanes
  if (narrow_small_batch(M) || few_tiles(M, N))
    launch_fm<1, false>(A, W, M, N, K, epi, s);
  else if (resid_tm2 && M >= 192)
    launch_fm<2, false, EpiDecResidFm<BIAS>, 2>(A, W, M, N, K, epi, s);
  else
    launch_fm<2, false>(A, W, M, N, K, epi, s);
}
void dec_gemm_resid(const bf16_t* A, const bf16_t* W, const float* bias, int M, int N, int K, float* H, hipStream_t s) {
  if ((N & 31) != 0) throw std::runtime_error("dec_gemm_resid: the residual width must be a multiple of 32");
  if (bias != nullptr) dec_gemm_resid_t<true>(A, W, bias, M, N, K, H, s);
  else dec_gemm_resid_t<false>(A, W, bias, M, N, K, H, s);
}
// ---- bf16-input small-batch GEMMs of the streaming decoder (row-based passes with M <= 256) ----
// Same split-K kernel, LayerNorm done by the caller; K covers the streaming widths (320 / 640 tiny / assumed-medium,
// 1280 / 2560 their ffn, 96 / 192 the test model) next to the offline ones (incl. 64 / 256, the offline test model: the
// offline ENCODER runs its layers' GEMMs here when a call holds few rows, Engine::run_encoder).
template <int TN, class Epi>
bool launch_dec_bf16(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  if ((N & 3) != 0) return false;
  switch (K) {
    case 64: launch_dec_cfg<2, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 256: launch_dec_cfg<8, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 96: launch_dec_cfg<3, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 192: launch_dec_cfg<6, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 288: launch_dec_cfg<9, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 320: launch_dec_cfg<10, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 416: launch_dec_cfg<13, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 640: launch_dec_cfg<20, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 1152: launch_dec_cfg<36, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 1280: launch_dec_cfg<40, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 1664: launch_dec_cfg<52, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 2560: launch_dec_cfg<80, TN, false, Epi>(A, lda, nullptr, W, M, N, epi, s); return true;
    default: return false;
  }
}
// LayerNorm-fused forms (A = fp32 residual stream, LayerNorm scale folded into W): K = decoder width only
template <int TN, class Epi>
bool launch_dec_ln(const float* H, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  if ((N & 3) != 0) return false;
  switch (K) {
    case 96: launch_dec_cfg<3, TN, true, Epi>(H, K, nullptr, W, M, N, epi, s); return true;
    case 288: launch_dec_cfg<9, TN, true, Epi>(H, K, nullptr, W, M, N, epi, s); return true;
    case 320: launch_dec_cfg<10, TN, true, Epi>(H, K, nullptr, W, M, N, epi, s); return true;
    case 416: launch_dec_cfg<13, TN, true, Epi>(H, K, nullptr, W, M, N, epi, s); return true;
    case 640: launch_dec_cfg<20, TN, true, Epi>(H, K, nullptr, W, M, N, epi, s); return true;
    default: return false;
  }
}
bool small_ln_gemm_stream_qkv(const float* H, const bf16_t* Wf, int M, int D, bf16_t* q_out, bf16_t* cacheK,
                              bf16_t* cacheV, const int* row_slot, const int* row_pos, RopeParams rp, int layer, int L,
                              int Scap, hipStream_t s) {
  return launch_dec_ln<2>(H, Wf, M, 3 * D, D, EpiStreamQkv{q_out, cacheK, cacheV, row_slot, row_pos, rp, layer, L, Scap}, s);
}
bool small_ln_gemm_bf16(const float* H, const bf16_t* Wf, int M, int N, int D, bf16_t* out, hipStream_t s) {
  if (few_tiles(M, N)) return launch_dec_ln<1>(H, Wf, M, N, D, EpiAct{out, nullptr, N, nullptr, 0}, s);
  return launch_dec_ln<2>(H, Wf, M, N, D, EpiAct{out, nullptr, N, nullptr, 0}, s);
}
bool small_ln_gemm_swiglu(const float* H, const bf16_t* Wf, const float* bias, int M, int N, int D, bf16_t* z,
                          hipStream_t s) {
  return launch_dec_ln<2>(H, Wf, M, N, D, EpiSwiGLU{z, N / 2, bias}, s);
}
bool small_ln_gemm_logits(const float* H, const bf16_t* Wf, int M, int N, int D, float* out, hipStream_t s) {
  return launch_dec_ln<4>(H, Wf, M, N, D, EpiF32{out, N}, s);
}
bool small_gemm_act(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int act, int M, int N, int K,
                    bf16_t* out_bf16, float* out_f32, hipStream_t s) {
  return launch_dec_bf16<2>(A, lda, W, M, N, K, EpiAct{out_bf16, out_f32, N, bias, act}, s);
}
bool small_gemm_qkv_rope_bf16(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, const int* row_pos,
                              RopeParams rp, bf16_t* out, hipStream_t s) {
  return launch_dec_bf16<2>(A, lda, W, M, N, K, EpiQkvRopeBf16{out, N, row_pos, rp}, s);
}
bool small_gemm_swiglu_bf16(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K,
                            bf16_t* z, hipStream_t s) {
  return launch_dec_bf16<2>(A, lda, W, M, N, K, EpiSwiGLU{z, N / 2, bias}, s);
}
bool small_gemm_resid_f32(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, float* H,
                          hipStream_t s) {
  if (bias != nullptr) {
    if (few_tiles(M, N)) return launch_dec_bf16<1>(A, lda, W, M, N, K, EpiDecResid<true>{H, N, bias}, s);
    return launch_dec_bf16<2>(A, lda, W, M, N, K, EpiDecResid<true>{H, N, bias}, s);
  }
  if (few_tiles(M, N)) return launch_dec_bf16<1>(A, lda, W, M, N, K, EpiDecResid<false>{H, N, bias}, s);
  return launch_dec_bf16<2>(A, lda, W, M, N, K, EpiDecResid<false>{H, N, bias}, s);
}
bool small_gemm_logits_f32(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, float* out, hipStream_t s) {
  return launch_dec_bf16<4>(A, lda, W, M, N, K, EpiF32{out, N}, s);
}

// ---- streaming decoder, AR steps: FM operands (weights packed at load, H / attention outputs / z in FM between the
// kernels).  Per element the k-split and the MFMA order are those of the row-major forms above: bit-identical results. ----
static int sfm_cfg(int digit, int dflt) {   // developer knob: MSH_SFM_CFG = digits (qkv, cross-q, fc1, resid), 0 = default
  static const char* e = dev_getenv("MSH_SFM_CFG");
  if (e == nullptr || (int)strlen(e) <= digit || e[digit] == '0') return dflt;
  return e[digit] - '0';
}
bool stream_fm_supported(int D, int F) {
  const bool d_ok = D == 96 || D == 320 || D == 640, f_ok = F == 192 || F == 1280 || F == 2560;
  return d_ok && f_ok;
}
bool stream_fm_qkv(const float* H, const bf16_t* Wfm, int M, int D, bf16_t* q_out, bf16_t* cacheK, bf16_t* cacheV,
                   const int* row_slot, const int* row_pos, RopeParams rp, int layer, int L, int Scap, hipStream_t s) {
  const EpiStreamQkv epi{q_out, cacheK, cacheV, row_slot, row_pos, rp, layer, L, Scap};
  switch (sfm_cfg(0, M >= 32 ? 2 : 1)) {
    case 1: return launch_sfm<1, true>(H, Wfm, M, 3 * D, D, epi, s);
    case 4: return launch_sfm<4, true>(H, Wfm, M, 3 * D, D, epi, s);
    default: return launch_sfm<2, true>(H, Wfm, M, 3 * D, D, epi, s);
  }
}
bool stream_fm_ln_bf16(const float* H, const bf16_t* Wfm, int M, int N, int D, bf16_t* out, hipStream_t s) {
  const EpiAct epi{out, nullptr, N, nullptr, 0};
  if (sfm_cfg(1, few_tiles(M, N) ? 1 : 2) == 1) return launch_sfm<1, true>(H, Wfm, M, N, D, epi, s);
  return launch_sfm<2, true>(H, Wfm, M, N, D, epi, s);
}
bool stream_fm_ln_swiglu(const float* H, const bf16_t* Wfm, const float* bias, int M, int F, int D, bf16_t* z,
                         hipStream_t s) {
  const EpiSwiGLUFm epi{z, F / 32, bias};
  switch (sfm_cfg(2, M >= 32 ? 3 : 2)) {
    case 1: return launch_sfm<1, true>(H, Wfm, M, 2 * F, D, epi, s);
    case 3: return launch_sfm<4, true, EpiSwiGLUFm, 2>(H, Wfm, M, 2 * F, D, epi, s);
    case 4: return launch_sfm<4, true>(H, Wfm, M, 2 * F, D, epi, s);
    case 5: return launch_sfm<2, true, EpiSwiGLUFm, 2>(H, Wfm, M, 2 * F, D, epi, s);
    default: return launch_sfm<2, true>(H, Wfm, M, 2 * F, D, epi, s);
  }
}
template <bool BIAS>
static bool stream_fm_resid_t(const bf16_t* A, const bf16_t* Wfm, const float* bias, int M, int N, int K, float* H,
                              hipStream_t s) {
  const EpiDecResidFm<BIAS> epi{H, N / 32, bias};
  if (sfm_cfg(3, few_tiles(M, N) ? 1 : 2) == 1) return launch_sfm<1, false>(A, Wfm, M, N, K, epi, s);
  return launch_sfm<2, false>(A, Wfm, M, N, K, epi, s);
}
bool stream_fm_resid(const bf16_t* A, const bf16_t* Wfm, const float* bias, int M, int N, int K, float* H, hipStream_t s) {
  if ((N & 31) != 0) return false;
  return bias != nullptr ? stream_fm_resid_t<true>(A, Wfm, bias, M, N, K, H, s) : stream_fm_resid_t<false>(A, Wfm, bias, M, N, K, H, s);
}

void dec_gemm_logits(const float* H, const bf16_t* E, int M, int V, int D, float* logits, hipStream_t s) {
  launch_fm<4, true>(H, E, M, V, D, EpiF32{logits, V}, s);
}

}  // namespace msh
