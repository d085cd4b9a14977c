// bf16 MFMA GEMMs for gfx950 (MI355X): C[M,N] = A[M,K](row stride lda) * W[N,K]^T, fp32 accumulate.
//
// Two kernels:
//  * gemm_tiled_kernel  -- encoder / conv-stem / cross-KV GEMMs (M = 10^3..10^5 rows).  One workgroup of
//    4 waves owns a (64*TM) x (16*TN) output tile; A and W k-slices of 32 are staged through LDS
//    (register-staged, double-buffered, one barrier per k-step, XOR-swizzled 16-B slots so the MFMA
//    fragment reads (ds_read_b128) are bank-conflict free).  Every Moonshine width is a multiple of
//    D/2 = 13*16 (base) so TN = 13 wastes nothing on N; the conv layers run as strided views of the
//    channels-last activation (lda = stride*C), never materialising im2col.
//  * gemm_small_kernel  -- decode GEMMs (M = batch).  Each wave owns a 16 x (16*TN) tile and loads its
//    MFMA fragments straight from global memory (weights are streamed once; no LDS round trip);
//    LayerNorm of the fp32 residual row is fused into the A-fragment load.
//
// MFMA 16x16x32 bf16 fragment map (cdna_hip_programming.md section 3): A: lane l holds row (l&15),
// k-chunk (l>>4)*8..+8; B: lane l holds column (l&15), same k-chunk; C/D: col = l&15,
// row = (l>>4)*4 + reg.  SWAP = true computes C^T tiles (W fragment as the A operand) so a lane
// ends up with 4 consecutive n for one m (vector stores along n, RoPE / SwiGLU pairs in-lane);
// SWAP = false gives 4 consecutive m for one n (used to write K^T / V^T along t).
#include "kernels.h"

namespace msh {
namespace {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------
// Epilogues.  n4(m, n, v): v[i] = C[m][n+i].   m4(m, n, v): v[i] = C[m+i][n].
// ------------------------------------------------------------------------------------------------
struct EpiTanhF32 {
  float* out;
  long ldc;
  __device__ void n4(int m, int n, f32x4 v) const {
    float4 o = make_float4(tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3]));
    *reinterpret_cast<float4*>(out + (long)m * ldc + n) = o;
  }
};

struct EpiBiasGeluBf16 {
  bf16_t* out;
  long ldc;
  const float* bias;
  __device__ void n4(int m, int n, f32x4 v) const {
    float4 b = *reinterpret_cast<const float4*>(bias + n);
    uint2 o;
    o.x = pack_bf16x2(gelu_erf(v[0] + b.x), gelu_erf(v[1] + b.y));
    o.y = pack_bf16x2(gelu_erf(v[2] + b.z), gelu_erf(v[3] + b.w));
    *reinterpret_cast<uint2*>(out + (long)m * ldc + n) = o;
  }
};

struct EpiBiasGeluF32 {
  float* out;
  long ldc;
  const float* bias;
  __device__ void n4(int m, int n, f32x4 v) const {
    float4 b = *reinterpret_cast<const float4*>(bias + n);
    float4 o = make_float4(gelu_erf(v[0] + b.x), gelu_erf(v[1] + b.y), gelu_erf(v[2] + b.z), gelu_erf(v[3] + b.w));
    *reinterpret_cast<float4*>(out + (long)m * ldc + n) = o;
  }
};

// rotate the two (even, odd) pairs held in v for head-dim offsets d, d+2
__device__ __forceinline__ void rope4(f32x4& v, int d, int pos, const RopeParams& rp) {
  const int j0 = d >> 1;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int j = j0 + p;
    if (j < rp.rot_pairs) {
      const float c = rp.cos[(long)pos * rp.rot_pairs + j];
      const float s = rp.sin[(long)pos * rp.rot_pairs + j];
      const float x0 = v[2 * p], x1 = v[2 * p + 1];
      v[2 * p] = x0 * c - x1 * s;
      v[2 * p + 1] = x1 * c + x0 * s;
    }
  }
}

struct EpiQkvRopeBf16 {
  bf16_t* out;
  long ldc;
  const int* row_pos;
  RopeParams rp;
  __device__ void n4(int m, int n, f32x4 v) const {
    if (n < 2 * rp.hidden) {  // q and k are rotated, v passes through
      int pos = row_pos[m];
      pos = pos < 0 ? 0 : pos;
      const int d = (n % rp.hidden) % rp.head_dim;
      rope4(v, d, pos, rp);
    }
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(out + (long)m * ldc + n) = o;
  }
};

struct EpiResidF32 {
  float* H;
  long ldc;
  const float* bias;  // nullable
  __device__ void n4(int m, int n, f32x4 v) const {
    float4* p = reinterpret_cast<float4*>(H + (long)m * ldc + n);
    float4 h = *p;
    if (bias != nullptr) {
      float4 b = *reinterpret_cast<const float4*>(bias + n);
      h.x += b.x;
      h.y += b.y;
      h.z += b.z;
      h.w += b.w;
    }
    h.x += v[0];
    h.y += v[1];
    h.z += v[2];
    h.w += v[3];
    *p = h;
  }
};

// cross K/V, all decoder layers in one GEMM: n = layer*2D + which*D + c  ->  K^T/V^T[layer][clip][c][t]
struct EpiCrossKV {
  bf16_t* KT;
  bf16_t* VT;
  const int* row_clip;
  const ClipMeta* clips;
  int D;
  long layer_stride;  // elements per layer = D * sum(Tk)
  __device__ void m4(int m, int n, f32x4 v) const {
    const int b = row_clip[m];
    const ClipMeta cm = clips[b];
    const int t = m - cm.row_start;
    if (t >= cm.Tk) return;  // Tk is a multiple of 8 and t of 4: the group is all in or all out
    const int layer = n / (2 * D);
    const int r = n - layer * 2 * D;
    const int which = r / D;
    const int c = r - which * D;
    bf16_t* base = (which ? VT : KT) + layer * layer_stride + (long)cm.kv_start * D + (long)c * cm.Tk + t;
    float x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (t + i < cm.T) ? v[i] : 0.0f;  // padding keys are exact zeros
    uint2 o;
    o.x = pack_bf16x2(x[0], x[1]);
    o.y = pack_bf16x2(x[2], x[3]);
    *reinterpret_cast<uint2*>(base) = o;
  }
};

struct EpiDecQkv {
  float* q;         // [M][D]
  bf16_t* cacheK;   // [M][H][Smax][dh]
  bf16_t* cacheV;
  const int* pos_ptr;
  RopeParams rp;
  int Smax;
  __device__ void n4(int m, int n, f32x4 v) const {
    const int D = rp.hidden, dh = rp.head_dim;
    const int pos = *pos_ptr;
    const int which = n / D;
    const int c = n - which * D;
    const int h = c / dh, d = c - h * dh;
    if (which < 2) rope4(v, d, pos, rp);
    if (which == 0) {
      *reinterpret_cast<float4*>(q + (long)m * D + c) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      bf16_t* dst = (which == 1 ? cacheK : cacheV) + (((long)m * (D / dh) + h) * Smax + pos) * dh + d;
      uint2 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(dst) = o;
    }
  }
};

struct EpiF32 {
  float* out;
  long ldc;
  __device__ void n4(int m, int n, f32x4 v) const {
    *reinterpret_cast<float4*>(out + (long)m * ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// rows of W / bias interleaved as (value_j, gate_j): modeling_moonshine.py:92-96 chunk order
struct EpiSwiGLU {
  bf16_t* z;
  long ldz;  // F
  const float* bias;
  __device__ void n4(int m, int n, f32x4 v) const {
    float4 b = *reinterpret_cast<const float4*>(bias + n);
    const float val0 = v[0] + b.x, gate0 = v[1] + b.y, val1 = v[2] + b.z, gate1 = v[3] + b.w;
    uint32_t o = pack_bf16x2(silu_f(gate0) * val0, silu_f(gate1) * val1);
    *reinterpret_cast<uint32_t*>(z + (long)m * ldz + (n >> 1)) = o;
  }
};

// ------------------------------------------------------------------------------------------------
// Tiled kernel
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }

template <int TM, int TN, bool SWAP, class Epi>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(const bf16_t* __restrict__ A, long lda,
                                                         const bf16_t* __restrict__ W, int M, int N, int K, int ntn,
                                                         int nblocks, Epi epi) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  constexpr int NA = TM;                      // 16-B chunks of A per thread per k-step
  constexpr int NB = (BN * 4 + 255) / 256;    // 16-B chunks of W per thread per k-step
  __shared__ __attribute__((aligned(16))) uint4 lds[2][(BM + BN) * 4];

  // XCD-aware tile order: workgroup b runs on XCD b%8; give every XCD a contiguous run of tiles with the
  // n-tile fastest so the A panel of an m-tile is re-read from that XCD's L2 (bijective for any grid size).
  const int bid = blockIdx.x;
  const int q8 = nblocks >> 3, r8 = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
  const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int m0 = (vid / ntn) * BM, n0 = (vid % ntn) * BN;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kg = lane >> 4;

  // per-thread staging assignment: 16-B chunk c = tid + 256*i  ->  (row = c >> 2, k-chunk = c & 3)
  const int srow = tid >> 2, sch = tid & 3;
  const bf16_t* abase = A + sch * 8;
  const bf16_t* wbase = W + sch * 8;
  // Staging registers are named scalars (not arrays): hipcc otherwise keeps them in an alloca that its
  // promote-alloca pass moves into LDS (+16 KB and a ds round trip per k-step).
  static_assert(NA <= 4 && NB <= 4, "staging registers are unrolled by hand up to 4 chunks");
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define MSH_LDA(i, k0)                                                                \
  if constexpr (NA > i) {                                                             \
    int gm = m0 + srow + 64 * i;                                                      \
    gm = gm < M ? gm : M - 1;                                                         \
    ra##i = *reinterpret_cast<const uint4*>(abase + (long)gm * lda + (k0));           \
  }
#define MSH_LDB(i, k0)                                                                \
  if constexpr (NB > i) {                                                             \
    if (srow + 64 * i < BN) {                                                         \
      int gn = n0 + srow + 64 * i;                                                    \
      gn = gn < N ? gn : N - 1;                                                       \
      rb##i = *reinterpret_cast<const uint4*>(wbase + (long)gn * K + (k0));           \
    }                                                                                 \
  }
#define MSH_STA(i, buf)                                                               \
  if constexpr (NA > i) {                                                             \
    const int row = srow + 64 * i;                                                    \
    lds[buf][row * 4 + (sch ^ swz(row))] = ra##i;                                     \
  }
#define MSH_STB(i, buf)                                                               \
  if constexpr (NB > i) {                                                             \
    const int row = srow + 64 * i;                                                    \
    if (row < BN) lds[buf][BM * 4 + row * 4 + (sch ^ swz(row))] = rb##i;              \
  }
#define MSH_GLOAD(k0)                                                                 \
  {                                                                                   \
    MSH_LDA(0, k0) MSH_LDA(1, k0) MSH_LDA(2, k0) MSH_LDA(3, k0)                       \
    MSH_LDB(0, k0) MSH_LDB(1, k0) MSH_LDB(2, k0) MSH_LDB(3, k0)                       \
  }
#define MSH_SSTORE(buf)                                                               \
  {                                                                                   \
    MSH_STA(0, buf) MSH_STA(1, buf) MSH_STA(2, buf) MSH_STA(3, buf)                   \
    MSH_STB(0, buf) MSH_STB(1, buf) MSH_STB(2, buf) MSH_STB(3, buf)                   \
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K >> 5;
  MSH_GLOAD(0);
  MSH_SSTORE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) MSH_GLOAD((kt + 1) << 5);  // next k-slice in flight during the MFMAs
    bf16x8 af[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wave * 16 * TM + i * 16 + li;
      uint4 t = lds[buf][row * 4 + (kg ^ swz(row))];
      af[i] = *reinterpret_cast<bf16x8*>(&t);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = j * 16 + li;
      uint4 t = lds[buf][BM * 4 + row * 4 + (kg ^ swz(row))];
      bf16x8 bf = *reinterpret_cast<bf16x8*>(&t);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (SWAP)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af[i], acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) MSH_SSTORE(buf ^ 1);
    __syncthreads();
  }
#undef MSH_GLOAD
#undef MSH_SSTORE
#undef MSH_LDA
#undef MSH_LDB
#undef MSH_STA
#undef MSH_STB

#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (SWAP) {
        const int m = m0 + wave * 16 * TM + i * 16 + li, n = n0 + j * 16 + kg * 4;
        if (m < M && n < N) epi.n4(m, n, acc[i][j]);
      } else {
        const int m = m0 + wave * 16 * TM + i * 16 + kg * 4, n = n0 + j * 16 + li;
        if (m < M && n < N) epi.m4(m, n, acc[i][j]);
      }
    }
  }
}

template <int TM, int TN, bool SWAP, class Epi>
void launch_tiled_cfg(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  const int nblocks = ntm * ntn;
  hipLaunchKernelGGL((gemm_tiled_kernel<TM, TN, SWAP, Epi>), dim3(nblocks), dim3(256), 0, s, A, lda, W, M, N, K, ntn,
                     nblocks, epi);
}

// Tile choice: TN = 13 (208 columns) divides every base-model width; widths that are multiples of 144
// (tiny, D = 288) use TN = 9; anything else falls back to TN = 4 with column predication.  TM = 4
// (256 rows, 208 accumulator registers, one workgroup per CU) once the grid still fills the chip.
template <bool SWAP, class Epi>
void launch_tiled(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  if ((K & 31) != 0 || (N & 3) != 0 || (lda & 7) != 0) throw std::runtime_error("gemm_tiled: unsupported shape");
  const bool big = (long)((M + 255) / 256) * ((N + 207) / 208) >= 512;
  if (N % 208 == 0) {
    if (big)
      launch_tiled_cfg<4, 13, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    else
      launch_tiled_cfg<2, 13, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
  } else if (N % 144 == 0) {
    launch_tiled_cfg<2, 9, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
  } else {
    launch_tiled_cfg<1, 4, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
  }
}

// ------------------------------------------------------------------------------------------------
// Fragment-direct kernel (decode)
// ------------------------------------------------------------------------------------------------
// KS > 0: A is the fp32 residual [M][K = 32*KS]; the wave normalises its 16 rows (LayerNorm, no bias,
// eps 1e-5, two-pass in registers) while building the bf16 A fragments.  KS == 0: A is bf16 [M][lda].
template <int KS, int TN, class Epi>
__global__ __launch_bounds__(256) void gemm_small_kernel(const void* __restrict__ Aptr, long lda,
                                                         const float* __restrict__ gamma,
                                                         const bf16_t* __restrict__ W, int M, int N, int K,
                                                         int n_tiles, int total_tiles, Epi epi) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= total_tiles) return;
  const int m0 = (tile / n_tiles) * 16, n0 = (tile % n_tiles) * (16 * TN);
  int gm = m0 + li;
  gm = gm < M ? gm : M - 1;

  f32x4 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* wrow[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int gn = n0 + j * 16 + li;
    gn = gn < N ? gn : N - 1;
    wrow[j] = W + (long)gn * K + kg * 8;
  }

  if constexpr (KS > 0) {
    const float* x = reinterpret_cast<const float*>(Aptr) + (long)gm * lda + kg * 8;
    float xv[KS][8];
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float4 a = *reinterpret_cast<const float4*>(x + s * 32);
      float4 b = *reinterpret_cast<const float4*>(x + s * 32 + 4);
      xv[s][0] = a.x; xv[s][1] = a.y; xv[s][2] = a.z; xv[s][3] = a.w;
      xv[s][4] = b.x; xv[s][5] = b.y; xv[s][6] = b.z; xv[s][7] = b.w;
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += xv[s][e];
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum / (float)K;
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
    const float rstd = rsqrtf(sq / (float)K + 1e-5f);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float4 g0 = *reinterpret_cast<const float4*>(gamma + s * 32 + kg * 8);
      float4 g1 = *reinterpret_cast<const float4*>(gamma + s * 32 + kg * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      uint4 t;
      t.x = pack_bf16x2((xv[s][0] - mean) * rstd * g[0], (xv[s][1] - mean) * rstd * g[1]);
      t.y = pack_bf16x2((xv[s][2] - mean) * rstd * g[2], (xv[s][3] - mean) * rstd * g[3]);
      t.z = pack_bf16x2((xv[s][4] - mean) * rstd * g[4], (xv[s][5] - mean) * rstd * g[5]);
      t.w = pack_bf16x2((xv[s][6] - mean) * rstd * g[6], (xv[s][7] - mean) * rstd * g[7]);
      const bf16x8 af = *reinterpret_cast<bf16x8*>(&t);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        uint4 wv = *reinterpret_cast<const uint4*>(wrow[j] + s * 32);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv), af, acc[j], 0, 0, 0);
      }
    }
  } else {
    const bf16_t* a = reinterpret_cast<const bf16_t*>(Aptr) + (long)gm * lda + kg * 8;
    const int nk = K >> 5;
#pragma unroll 4
    for (int s = 0; s < nk; ++s) {
      uint4 av = *reinterpret_cast<const uint4*>(a + s * 32);
      const bf16x8 af = *reinterpret_cast<bf16x8*>(&av);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        uint4 wv = *reinterpret_cast<const uint4*>(wrow[j] + s * 32);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv), af, acc[j], 0, 0, 0);
      }
    }
  }
  const int m = m0 + li;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + j * 16 + kg * 4;
    if (m < M && n < N) epi.n4(m, n, acc[j]);
  }
}

template <int KS, int TN, class Epi>
void launch_small_cfg(const void* A, long lda, const float* gamma, const bf16_t* W, int M, int N, int K, Epi epi,
                      hipStream_t s) {
  const int m_tiles = (M + 15) / 16, n_tiles = (N + 16 * TN - 1) / (16 * TN);
  const int total = m_tiles * n_tiles;
  hipLaunchKernelGGL((gemm_small_kernel<KS, TN, Epi>), dim3((total + 3) / 4), dim3(256), 0, s, A, lda, gamma, W, M, N,
                     K, n_tiles, total, epi);
}

template <int TN, class Epi>
void launch_small_ln(const float* H, const float* gamma, const bf16_t* W, int M, int N, int D, Epi epi,
                     hipStream_t s) {
  if ((N & 3) != 0) throw std::runtime_error("gemm_small: N must be a multiple of 4");
  switch (D / 32) {
    case 13: if (D == 416) return launch_small_cfg<13, TN, Epi>(H, D, gamma, W, M, N, D, epi, s); break;
    case 9:  if (D == 288) return launch_small_cfg<9, TN, Epi>(H, D, gamma, W, M, N, D, epi, s); break;
    case 2:  if (D == 64)  return launch_small_cfg<2, TN, Epi>(H, D, gamma, W, M, N, D, epi, s); break;
    default: break;
  }
  throw std::runtime_error("gemm_small: unsupported hidden size " + std::to_string(D));
}

}  // namespace

// ------------------------------------------------------------------------------------------------
void gemm_tanh_f32(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, float* out, hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiTanhF32{out, N}, s);
}
void gemm_bias_gelu_bf16(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K,
                         bf16_t* out, hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiBiasGeluBf16{out, N, bias}, s);
}
void gemm_bias_gelu_f32(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, float* out,
                        hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiBiasGeluF32{out, N, bias}, s);
}
void gemm_qkv_rope_bf16(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, const int* row_pos,
                        RopeParams rp, bf16_t* out, hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiQkvRopeBf16{out, N, row_pos, rp}, s);
}
void gemm_resid_f32(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, float* H,
                    hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiResidF32{H, N, bias}, s);
}
void gemm_cross_kv(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, const int* row_clip,
                   const ClipMeta* clips, int D, long layer_stride, bf16_t* KT, bf16_t* VT, hipStream_t s) {
  launch_tiled<false>(A, lda, W, M, N, K, EpiCrossKV{KT, VT, row_clip, clips, D, layer_stride}, s);
}

void dec_gemm_qkv(const float* H, const float* gamma, const bf16_t* W, int M, int D, const int* pos_ptr, RopeParams rp,
                  float* q, bf16_t* cacheK, bf16_t* cacheV, int Smax, hipStream_t s) {
  launch_small_ln<2>(H, gamma, W, M, 3 * D, D, EpiDecQkv{q, cacheK, cacheV, pos_ptr, rp, Smax}, s);
}
void dec_gemm_ln_f32(const float* H, const float* gamma, const bf16_t* W, int M, int N, int D, float* out,
                     hipStream_t s) {
  launch_small_ln<2>(H, gamma, W, M, N, D, EpiF32{out, N}, s);
}
void dec_gemm_ln_swiglu(const float* H, const float* gamma, const bf16_t* W, const float* bias, int M, int F, int D,
                        bf16_t* z, hipStream_t s) {
  launch_small_ln<2>(H, gamma, W, M, 2 * F, D, EpiSwiGLU{z, F, bias}, s);
}
void dec_gemm_resid(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, float* H,
                    hipStream_t s) {
  if ((K & 31) != 0 || (N & 3) != 0) throw std::runtime_error("dec_gemm_resid: unsupported shape");
  launch_small_cfg<0, 2>(A, lda, nullptr, W, M, N, K, EpiResidF32{H, N, bias}, s);
}
void dec_gemm_logits(const float* H, const float* gamma, const bf16_t* E, int M, int V, int D, float* logits,
                     hipStream_t s) {
  if (M > 32)
    launch_small_ln<8>(H, gamma, E, M, V, D, EpiF32{logits, V}, s);
  else
    launch_small_ln<2>(H, gamma, E, M, V, D, EpiF32{logits, V}, s);
}

}  // namespace msh
