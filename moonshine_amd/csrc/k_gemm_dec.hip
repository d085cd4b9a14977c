// Decode-path GEMMs for gfx950 (M = batch rows, weights streamed): see the two kernels below.
#include <stdlib.h>

#include "gemm_common.h"

namespace msh {
namespace {

// ------------------------------------------------------------------------------------------------
// Decode GEMM (M = batch rows): fragment-direct, split-K inside the workgroup.
//
// A workgroup owns one 16 x (16*TN) output tile; its 4 waves split the K loop (wave w takes the 32-wide
// k-steps w, w+4, ...), each loading its MFMA fragments straight from global memory -- all loads of a
// wave are independent and issued up front, so the kernel costs about one memory round trip instead of a
// K/32-long dependent chain.  Partial tiles are summed through LDS in a fixed order (deterministic).
// LN = true: A is the fp32 residual stream [M][K]; LayerNorm (no bias, eps 1e-5, exact two-pass) is
// fused into the fragment build: row sums are combined across the 4 waves through LDS.  The LayerNorm
// scale gamma is folded into W at load time (W' = W * diag(gamma)), so the kernel only normalises.
// ------------------------------------------------------------------------------------------------
template <int KS, int TN, bool LN, class Epi, int TM = 1>
__global__ __launch_bounds__(256) void gemm_dec_kernel(const void* __restrict__ Aptr, long lda,
                                                       const float* __restrict__ gamma,
                                                       const bf16_t* __restrict__ W, int M, int N, int n_tiles,
                                                       Epi epi) {
  // TM = 2: the workgroup owns 32 rows (two MFMA row tiles) and every W fragment it loads feeds two MFMAs -- the
  // weights are re-read M/32 instead of M/16 times.  At M = 256 these GEMMs are bound by exactly that L2 -> CU
  // re-read traffic (fc1: 832 workgroups x 66 KB = 55 MB per launch at ~4 TB/s), not by HBM or latency.
  constexpr int K = 32 * KS;
  constexpr int KW = (KS + 3) / 4;  // k-steps per wave (upper bound)
  __shared__ __attribute__((aligned(16))) float4 part[4][TM * TN][64];
  __shared__ float stat[2][4][16 * TM];
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = (blockIdx.x / n_tiles) * (16 * TM), n0 = (blockIdx.x % n_tiles) * (16 * TN);
  int gm[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    gm[t] = m0 + t * 16 + li;
    gm[t] = gm[t] < M ? gm[t] : M - 1;
  }

  // ---- issue every load of this wave first ----
  uint4 wreg[KW][TN];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int s = wave + 4 * i;
    if (s < KS) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int gn = n0 + j * 16 + li;
        gn = gn < N ? gn : N - 1;
        wreg[i][j] = *reinterpret_cast<const uint4*>(W + (long)gn * K + s * 32 + kg * 8);
      }
    }
  }
  // inputs of the epilogue this wave will run at the end (residual, bias, RoPE factors): fetched now so
  // that the kernel has ONE memory round trip on its critical path, not one per dependent stage
  constexpr int NT = TM * TN;            // output tiles of the workgroup
  constexpr int NE = (NT + 3) / 4;
  typename Epi::Pre epre[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int o = wave + 4 * e, t = o / TN, j = o - t * TN;
    const int m = m0 + t * 16 + li, n = n0 + j * 16 + kg * 4;
    if (o < NT && m < M && n < N) epre[e] = epi.pre(m, n);
  }
  bf16x8 afrag[TM][KW];
  if constexpr (LN) {
    float xv[TM][KW][8];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const float* x = reinterpret_cast<const float*>(Aptr) + (long)gm[t] * lda + kg * 8;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < KW; ++i) {
        const int s = wave + 4 * i;
        if (s < KS) {
          const float4 a = *reinterpret_cast<const float4*>(x + s * 32);
          const float4 b = *reinterpret_cast<const float4*>(x + s * 32 + 4);
          xv[t][i][0] = a.x; xv[t][i][1] = a.y; xv[t][i][2] = a.z; xv[t][i][3] = a.w;
          xv[t][i][4] = b.x; xv[t][i][5] = b.y; xv[t][i][6] = b.z; xv[t][i][7] = b.w;
#pragma unroll
          for (int e = 0; e < 8; ++e) sum += xv[t][i][e];
        }
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      if (kg == 0) stat[0][wave][t * 16 + li] = sum;
    }
    __syncthreads();
    float mean[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int r = t * 16 + li;
      mean[t] = ((stat[0][0][r] + stat[0][1][r]) + (stat[0][2][r] + stat[0][3][r])) * (1.0f / (float)K);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < KW; ++i) {
        if (wave + 4 * i < KS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = xv[t][i][e] - mean[t];
            sq += d * d;
          }
        }
      }
      sq += __shfl_xor(sq, 16);
      sq += __shfl_xor(sq, 32);
      if (kg == 0) stat[1][wave][r] = sq;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int r = t * 16 + li;
      const float var = ((stat[1][0][r] + stat[1][1][r]) + (stat[1][2][r] + stat[1][3][r])) * (1.0f / (float)K);
      const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
      for (int i = 0; i < KW; ++i) {
        const int s = wave + 4 * i;
        if (s < KS) {
          uint4 q;
          q.x = pack_bf16x2((xv[t][i][0] - mean[t]) * rstd, (xv[t][i][1] - mean[t]) * rstd);
          q.y = pack_bf16x2((xv[t][i][2] - mean[t]) * rstd, (xv[t][i][3] - mean[t]) * rstd);
          q.z = pack_bf16x2((xv[t][i][4] - mean[t]) * rstd, (xv[t][i][5] - mean[t]) * rstd);
          q.w = pack_bf16x2((xv[t][i][6] - mean[t]) * rstd, (xv[t][i][7] - mean[t]) * rstd);
          afrag[t][i] = *reinterpret_cast<bf16x8*>(&q);
        }
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const bf16_t* a = reinterpret_cast<const bf16_t*>(Aptr) + (long)gm[t] * lda + kg * 8;
#pragma unroll
      for (int i = 0; i < KW; ++i) {
        const int s = wave + 4 * i;
        if (s < KS) {
          uint4 q = *reinterpret_cast<const uint4*>(a + s * 32);
          afrag[t][i] = *reinterpret_cast<bf16x8*>(&q);
        }
      }
    }
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    if (wave + 4 * i < KS) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int t = 0; t < TM; ++t)
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wreg[i][j]), afrag[t][i],
                                                               acc[t][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      part[wave][t * TN + j][lane] = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
  __syncthreads();
  // the waves finish the output tiles round-robin: fixed summation order
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int o = wave + 4 * e;
    if (o < NT) {
      const int t = o / TN, j = o - t * TN;
      const float4 p0 = part[0][o][lane], p1 = part[1][o][lane], p2 = part[2][o][lane], p3 = part[3][o][lane];
      f32x4 v;
      v[0] = (p0.x + p1.x) + (p2.x + p3.x);
      v[1] = (p0.y + p1.y) + (p2.y + p3.y);
      v[2] = (p0.z + p1.z) + (p2.z + p3.z);
      v[3] = (p0.w + p1.w) + (p2.w + p3.w);
      const int m = m0 + t * 16 + li, n = n0 + j * 16 + kg * 4;
      if (m < M && n < N) epi.n4p(m, n, v, epre[e]);
    }
  }
}

template <int KS, int TN, bool LN, class Epi, int TM = 1>
void launch_dec_cfg(const void* A, long lda, const float* gamma, const bf16_t* W, int M, int N, Epi epi,
                    hipStream_t s) {
  const int m_tiles = (M + 16 * TM - 1) / (16 * TM), n_tiles = (N + 16 * TN - 1) / (16 * TN);
  hipLaunchKernelGGL((gemm_dec_kernel<KS, TN, LN, Epi, TM>), dim3(m_tiles * n_tiles), dim3(256), 0, s, A, lda, gamma, W,
                     M, N, n_tiles, epi);
}
// 32-row workgroups for the K = D GEMMs of large batches (see gemm_dec_kernel); false if K has no such instance
template <int TN, bool LN, class Epi>
bool launch_dec_tm2(const void* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  switch (K) {
    case 416: launch_dec_cfg<13, TN, LN, Epi, 2>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 288: launch_dec_cfg<9, TN, LN, Epi, 2>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 64: launch_dec_cfg<2, TN, LN, Epi, 2>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 1664: launch_dec_cfg<52, TN, LN, Epi, 2>(A, lda, nullptr, W, M, N, epi, s); return true;
    case 1152: launch_dec_cfg<36, TN, LN, Epi, 2>(A, lda, nullptr, W, M, N, epi, s); return true;
    default: return false;
  }
}
static int dec_tm2_threshold() {
  static int thr = [] {
    const char* e = getenv("MSH_DEC_TM2_M");
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


// ------------------------------------------------------------------------------------------------
// Decode GEMM for larger batches (M >= 64): A fragments resident in registers, W streamed by LDS-DMA.
//
// A workgroup owns a 64 x (16*TN) tile: wave w holds the MFMA A fragments of its own 16 rows for the
// whole K = 32*KS (LayerNorm fused: the row is normalised in registers, two-pass, gamma folded into W),
// so the only operand that moves during the k-loop is the W slice -- TN KiB per 32-wide k-step, fetched
// once per workgroup by `global_load_lds` into a 6-deep ring and shared by the 4 waves.  Compared with
// gemm_dec_kernel (16-row tiles, W fragments loaded per wave) the weights are re-read M/64 instead of
// M/16 times and the LayerNorm is recomputed N/(16*TN) / 4 times less often.
// ------------------------------------------------------------------------------------------------
template <int KS, int TN, bool LN, class Epi>
__global__ __launch_bounds__(256) void gemm_dec64_kernel(const void* __restrict__ Aptr, long lda,
                                                         const bf16_t* __restrict__ W, int M, int N, int n_tiles,
                                                         Epi epi) {
  constexpr int K = 32 * KS;
  constexpr int NST = 6;                      // ring depth (k-slices)
  constexpr int AHEAD = NST - 1;
  constexpr int PW = (TN + 3) / 4;            // 1-KiB pieces a wave issues per k-slice (upper bound)
  __shared__ __attribute__((aligned(16))) uint4 lds[NST * TN * 64];
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = (blockIdx.x / n_tiles) * 64 + wave * 16, n0 = (blockIdx.x % n_tiles) * (16 * TN);
  const int m = m0 + li;
  const int gm = m < M ? m : M - 1;
  const int my_pieces = (TN - wave + 3) / 4;  // wave-uniform, 0 when TN < 4 and wave >= TN

  // epilogue inputs first (oldest loads), then the A rows, then the DMA prologue
  typename Epi::Pre epre[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + j * 16 + kg * 4;
    if (m < M && n < N) epre[j] = epi.pre(m, n);
  }
  float xv[LN ? KS : 1][8];
  uint4 araw[LN ? 1 : KS];
  if constexpr (LN) {
    const float* x = reinterpret_cast<const float*>(Aptr) + (long)gm * lda + kg * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(x + s * 32);
      const float4 b = *reinterpret_cast<const float4*>(x + s * 32 + 4);
      xv[s][0] = a.x; xv[s][1] = a.y; xv[s][2] = a.z; xv[s][3] = a.w;
      xv[s][4] = b.x; xv[s][5] = b.y; xv[s][6] = b.z; xv[s][7] = b.w;
    }
  } else {
    const bf16_t* a = reinterpret_cast<const bf16_t*>(Aptr) + (long)gm * lda + kg * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) araw[s] = *reinterpret_cast<const uint4*>(a + s * 32);
  }
  const bf16_t* src[PW];
  unsigned dst[PW];
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_offset_of(&lds[0]));
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int p = wave + 4 * i;               // column tile this wave fetches
    const int r = lane >> 2, pos = lane & 3, row = p * 16 + r;
    int gn = n0 + row;
    gn = gn < N ? gn : N - 1;
    src[i] = W + (long)gn * K + ((pos ^ swz(row)) << 3);
    dst[i] = lds_base + (unsigned)p * 1024u;
  }
  auto issue = [&](int kt) {
    const unsigned sb = (unsigned)(kt % NST) * (TN * 1024u);
#pragma unroll
    for (int i = 0; i < PW; ++i)
      if (i < my_pieces) dma16(src[i] + (kt << 5), dst[i] + sb);
  };
#pragma unroll
  for (int s = 0; s < AHEAD; ++s)
    if (s < KS) issue(s);

  bf16x8 afr[KS];
  if constexpr (LN) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += xv[s][e];
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / (float)K);
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = xv[s][e] - mean;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq * (1.0f / (float)K) + 1e-5f);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint4 t;
      t.x = pack_bf16x2((xv[s][0] - mean) * rstd, (xv[s][1] - mean) * rstd);
      t.y = pack_bf16x2((xv[s][2] - mean) * rstd, (xv[s][3] - mean) * rstd);
      t.z = pack_bf16x2((xv[s][4] - mean) * rstd, (xv[s][5] - mean) * rstd);
      t.w = pack_bf16x2((xv[s][6] - mean) * rstd, (xv[s][7] - mean) * rstd);
      afr[s] = *reinterpret_cast<bf16x8*>(&t);
    }
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) afr[s] = *reinterpret_cast<bf16x8*>(&araw[s]);
  }

  f32x4 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // every compiler-issued load above is older than the DMAs, and vector-memory ops retire in order: a
  // counted wait that leaves only this wave's newest DMA pieces outstanding therefore covers them too
#pragma unroll
  for (int kt = 0; kt < KS; ++kt) {
    const int inflight = (kt + AHEAD - 1 < KS ? AHEAD - 1 : KS - 1 - kt);  // compile-time after unrolling
    if (my_pieces == PW) {
      switch (inflight) {
        case 4: wait_vmcnt<4 * PW>(); break;
        case 3: wait_vmcnt<3 * PW>(); break;
        case 2: wait_vmcnt<2 * PW>(); break;
        case 1: wait_vmcnt<PW>(); break;
        default: wait_vmcnt<0>(); break;
      }
    } else {
      switch (inflight) {
        case 4: wait_vmcnt<4 * (PW - 1)>(); break;
        case 3: wait_vmcnt<3 * (PW - 1)>(); break;
        case 2: wait_vmcnt<2 * (PW - 1)>(); break;
        case 1: wait_vmcnt<(PW - 1)>(); break;
        default: wait_vmcnt<0>(); break;
      }
    }
    __builtin_amdgcn_s_barrier();
    if (kt + AHEAD < KS) issue(kt + AHEAD);
    const uint4* st = lds + (kt % NST) * (TN * 64);
    uint4 bfr[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = j * 16 + li;
      bfr[j] = st[row * 4 + (kg ^ swz(row))];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&bfr[j]), afr[kt], acc[j], 0, 0, 0);
  }
  wait_vmcnt<0>();
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + j * 16 + kg * 4;
    if (m < M && n < N) epi.n4p(m, n, acc[j], epre[j]);
  }
}

template <int KS, int TN, bool LN, class Epi>
void launch_dec64_cfg(const void* A, long lda, const bf16_t* W, int M, int N, Epi epi, hipStream_t s) {
  const int m_tiles = (M + 63) / 64, n_tiles = (N + 16 * TN - 1) / (16 * TN);
  hipLaunchKernelGGL((gemm_dec64_kernel<KS, TN, LN, Epi>), dim3(m_tiles * n_tiles), dim3(256), 0, s, A, lda, W, M, N,
                     n_tiles, epi);
}

// register-resident A needs K <= 416 (13 fragments): the hidden size of every supported model
template <int TN, bool LN, class Epi>
bool launch_dec64(const void* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  switch (K) {
    case 416: launch_dec64_cfg<13, TN, LN, Epi>(A, lda, W, M, N, epi, s); return true;
    case 288: launch_dec64_cfg<9, TN, LN, Epi>(A, lda, W, M, N, epi, s); return true;
    case 64: launch_dec64_cfg<2, TN, LN, Epi>(A, lda, W, M, N, epi, s); return true;
    default: return false;
  }
}

}  // namespace

// Batch threshold for the 64-row register-resident-A kernel.  Off by default (MSH_DEC64_M=0): on MI355X it
// measured slower than the split-K kernel at M = 256 (qkv 23 vs 12 us) -- kept for further work.
static bool use_dec64(int M) {
  static int thr = [] {
    const char* e = getenv("MSH_DEC64_M");
    return e ? atoi(e) : 0;
  }();
  return thr > 0 && M >= thr;
}

// Narrow outputs (N = decoder width) at streaming batch sizes give too few 16 x 32 tiles to occupy the chip (M = 64,
// N = 640: 80 workgroups on 256 CUs, each streaming its whole K x 32 weight slice alone): halve the tile width then.
// The per-element summation order does not depend on the tile width, so results are bit-identical.
static bool few_tiles(int M, int N) {
  static const int thr = [] {
    const char* e = getenv("MSH_DEC_NARROW_TILES");
    return e ? atoi(e) : 192;
  }();
  return ((M + 15) / 16) * ((N + 31) / 32) < thr;
}
// One row tile (M <= 16, the single-clip latency case): 16-column tiles double the workgroups that share a GEMM's weight
// stream (o-proj 13 -> 26, qkv 39 -> 78, fc1 104 -> 208); measured p50 of a 10 s clip 20.8 -> 19.0 ms (decode 19.4 -> 17.6).
static bool narrow_small_batch(int M) {
  static const int thr = [] {
    const char* e = getenv("MSH_DEC_NARROW_M");
    return e ? atoi(e) : 16;
  }();
  return M <= thr;
}

void dec_gemm_qkv(const float* H, const bf16_t* W, int M, int D, const int* pos_ptr, RopeParams rp, float* q,
                  bf16_t* cacheK, bf16_t* cacheV, int Smax, hipStream_t s) {
  EpiDecQkv epi{q, cacheK, cacheV, pos_ptr, rp, Smax};
  if (use_dec64(M) && launch_dec64<4, true>(H, D, W, M, 3 * D, D, epi, s)) return;
  if (M >= 96)  // wide column tiles (64) halve the per-row-tile LayerNorm / A reloads of the big-N GEMMs
    launch_dec<4, true>(H, D, nullptr, W, M, 3 * D, D, epi, s);
  else if (narrow_small_batch(M) || few_tiles(M, 3 * D))
    launch_dec<1, true>(H, D, nullptr, W, M, 3 * D, D, epi, s);
  else
    launch_dec<2, true>(H, D, nullptr, W, M, 3 * D, D, epi, s);
}
void dec_gemm_ln_f32(const float* H, const bf16_t* W, int M, int N, int D, float* out, hipStream_t s) {
  EpiF32 epi{out, N};
  if (use_dec64(M) && launch_dec64<4, true>(H, D, W, M, N, D, epi, s)) return;
  if (few_tiles(M, N))
    launch_dec<1, true>(H, D, nullptr, W, M, N, D, epi, s);
  else
    launch_dec<2, true>(H, D, nullptr, W, M, N, D, epi, s);
}
void dec_gemm_ln_swiglu(const float* H, const bf16_t* W, const float* bias, int M, int F, int D, bf16_t* z,
                        hipStream_t s) {
  EpiSwiGLU epi{z, F, bias};
  if (use_dec64(M) && launch_dec64<4, true>(H, D, W, M, 2 * F, D, epi, s)) return;
  // 32-row workgroups pay off only where there are many column tiles (r01t, M = 256: fc1 14.1 -> 12.4 us, but
  // qkv 9.8 -> 11.1 and cross-q 7.9 -> 11.1 us with half as many workgroups in flight)
  if (M >= dec_tm2_threshold() && launch_dec_tm2<4, true>(H, D, W, M, 2 * F, D, epi, s)) return;
  if (M >= 96)
    launch_dec<4, true>(H, D, nullptr, W, M, 2 * F, D, epi, s);
  else if (narrow_small_batch(M))
    launch_dec<1, true>(H, D, nullptr, W, M, 2 * F, D, epi, s);
  else
    launch_dec<2, true>(H, D, nullptr, W, M, 2 * F, D, epi, s);
}
void dec_gemm_resid(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, float* H,
                    hipStream_t s) {
  EpiResidF32 epi{H, N, bias};
  if (use_dec64(M) && launch_dec64<4, false>(A, lda, W, M, N, K, epi, s)) return;  // K = D only
  {
    static const bool fc2_tm2 = [] {
      const char* e = getenv("MSH_DEC_FC2_TM2");
      return e != nullptr && e[0] == '1';
    }();
    if (fc2_tm2 && K > N && M >= dec_tm2_threshold() && launch_dec_tm2<2, false>(A, lda, W, M, N, K, epi, s)) return;
  }
  if (narrow_small_batch(M) || few_tiles(M, N))
    launch_dec<1, false>(A, lda, nullptr, W, M, N, K, epi, s);
  else
    launch_dec<2, false>(A, lda, nullptr, W, M, N, K, epi, s);
}
// ---- bf16-input small-batch GEMMs of the streaming decoder (row-based passes with M <= 256) ----
// Same split-K kernel, LayerNorm done by the caller; K covers the streaming widths (320 / 640 tiny / assumed-medium,
// 1280 / 2560 their ffn, 96 / 192 the test model) next to the offline ones.
template <int TN, class Epi>
bool launch_dec_bf16(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  if ((N & 3) != 0) return false;
  switch (K) {
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
  if (few_tiles(M, N)) return launch_dec_bf16<1>(A, lda, W, M, N, K, EpiResidF32{H, N, bias}, s);
  return launch_dec_bf16<2>(A, lda, W, M, N, K, EpiResidF32{H, N, bias}, s);
}
bool small_gemm_logits_f32(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, float* out, hipStream_t s) {
  return launch_dec_bf16<4>(A, lda, W, M, N, K, EpiF32{out, N}, s);
}

void dec_gemm_logits(const float* H, const bf16_t* E, int M, int V, int D, float* logits, hipStream_t s) {
  launch_dec<4, true>(H, D, nullptr, E, M, V, D, EpiF32{logits, V}, s);
}

}  // namespace msh
