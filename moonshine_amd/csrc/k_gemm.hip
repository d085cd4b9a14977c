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
#include <stdlib.h>

#include "kernels.h"

namespace msh {
namespace {

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 output rounding): one v_exp,
// one v_rcp and a degree-5 Horner chain instead of libm's branchy erff (which cost as much as the
// K = 416 main loop in the fc1 epilogue).
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float e = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
// tanh(x) = 1 - 2 / (1 + e^{2x}); saturates cleanly (e^{2x} -> inf gives 1, -> 0 gives -1)
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------
// Epilogues.  n4(m, n, v): v[i] = C[m][n+i].   m4(m, n, v): v[i] = C[m+i][n].
// ------------------------------------------------------------------------------------------------
struct EpiTanhF32 {
  float* out;
  long ldc;
  __device__ void n4(int m, int n, f32x4 v) const {
    float4 o = make_float4(tanh_fast(v[0]), tanh_fast(v[1]), tanh_fast(v[2]), tanh_fast(v[3]));
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
  // decode path: the old residual value and the bias are fetched before the GEMM, not after it
  struct Pre {
    float4 h, b;
  };
  __device__ Pre pre(int m, int n) const {
    Pre p;
    p.h = *reinterpret_cast<const float4*>(H + (long)m * ldc + n);
    p.b = bias != nullptr ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    return p;
  }
  __device__ void n4p(int m, int n, f32x4 v, const Pre& p) const {
    *reinterpret_cast<float4*>(H + (long)m * ldc + n) =
        make_float4(p.h.x + p.b.x + v[0], p.h.y + p.b.y + v[1], p.h.z + p.b.z + v[2], p.h.w + p.b.w + v[3]);
  }
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
  struct Pre {
    int pos;
    float c0, s0, c1, s1;
  };
  __device__ Pre pre(int /*m*/, int n) const {
    Pre p;
    p.pos = *pos_ptr;
    const int d = (n % rp.hidden) % rp.head_dim, j0 = d >> 1;
    p.c0 = p.c1 = 1.f;
    p.s0 = p.s1 = 0.f;
    if (n < 2 * rp.hidden) {  // q / k: rotation factors of the lane's two pairs (identity beyond rot_pairs)
      if (j0 < rp.rot_pairs) {
        p.c0 = rp.cos[(long)p.pos * rp.rot_pairs + j0];
        p.s0 = rp.sin[(long)p.pos * rp.rot_pairs + j0];
      }
      if (j0 + 1 < rp.rot_pairs) {
        p.c1 = rp.cos[(long)p.pos * rp.rot_pairs + j0 + 1];
        p.s1 = rp.sin[(long)p.pos * rp.rot_pairs + j0 + 1];
      }
    }
    return p;
  }
  __device__ void n4p(int m, int n, f32x4 v, const Pre& p) const {
    const int D = rp.hidden, dh = rp.head_dim;
    const int which = n / D;
    const int c = n - which * D;
    const int h = c / dh, d = c - h * dh;
    const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
    v[0] = x0 * p.c0 - x1 * p.s0;
    v[1] = x1 * p.c0 + x0 * p.s0;
    v[2] = x2 * p.c1 - x3 * p.s1;
    v[3] = x3 * p.c1 + x2 * p.s1;
    if (which == 0) {
      *reinterpret_cast<float4*>(q + (long)m * D + c) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      bf16_t* dst = (which == 1 ? cacheK : cacheV) + (((long)m * (D / dh) + h) * Smax + p.pos) * dh + d;
      uint2 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(dst) = o;
    }
  }
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
  struct Pre {};
  __device__ Pre pre(int, int) const { return Pre{}; }
  __device__ void n4p(int m, int n, f32x4 v, const Pre&) const { n4(m, n, v); }
  __device__ void n4(int m, int n, f32x4 v) const {
    *reinterpret_cast<float4*>(out + (long)m * ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// rows of W / bias interleaved as (value_j, gate_j): modeling_moonshine.py:92-96 chunk order
struct EpiSwiGLU {
  bf16_t* z;
  long ldz;  // F
  const float* bias;
  struct Pre {
    float4 b;
  };
  __device__ Pre pre(int, int n) const { return Pre{*reinterpret_cast<const float4*>(bias + n)}; }
  __device__ void n4p(int m, int n, f32x4 v, const Pre& p) const {
    const float val0 = v[0] + p.b.x, gate0 = v[1] + p.b.y, val1 = v[2] + p.b.z, gate1 = v[3] + p.b.w;
    uint32_t o = pack_bf16x2(silu_f(gate0) * val0, silu_f(gate1) * val1);
    *reinterpret_cast<uint32_t*>(z + (long)m * ldz + (n >> 1)) = o;
  }
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

// ------------------------------------------------------------------------------------------------
// Tiled kernel, LDS-DMA pipeline.  Same tile / fragment / epilogue structure as gemm_tiled_kernel, but
// the k-slices travel HBM -> LDS with `global_load_lds_dwordx4` (no staging registers), NSTAGE buffers
// deep with up to NSTAGE-1 slices in flight across the per-step barrier (counted vmcnt, raw s_barrier).
// The DMA writes LDS lane-linearly (wave-uniform base + lane*16 B), so the XOR swizzle of the slot layout
// is applied on the per-lane SOURCE address: a 1-KiB piece = 16 rows x 4 slots, lane l fills slot
// (row = l>>2, pos = l&3) with global k-chunk pos ^ swz(row) -- still one 64-B segment per row.
// The DMA is issued from inline asm: hipcc otherwise treats it as an LDS store that may alias the
// fragment reads and drains vmcnt(0) before every ds_read (cdna_hip_programming.md section 5).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char lds_char_t;
__device__ __forceinline__ unsigned lds_offset_of(const void* p) { return (unsigned)(unsigned long)(lds_char_t*)(p); }
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NW, int TM, int TN, int NSTAGE, bool SWAP, class Epi>
__global__ __launch_bounds__(64 * NW) void gemm_tiled_dma_kernel(const bf16_t* __restrict__ A, long lda,
                                                             const bf16_t* __restrict__ W, int M, int N, int K,
                                                             int ntn, int nblocks, Epi epi) {
  constexpr int BM = 16 * NW * TM, BN = 16 * TN;
  constexpr int PA = BM / 16, P = PA + TN;     // 1-KiB pieces per k-slice (A rows, then W rows)
  constexpr int PMAX = (P + NW - 1) / NW;      // pieces per wave (waves take pieces w, w+NW, ...)
  constexpr int STAGE_SLOTS = (BM + BN) * 4;   // 16-B slots per k-slice
  __shared__ __attribute__((aligned(16))) uint4 lds[NSTAGE * STAGE_SLOTS];

  const int bid = blockIdx.x;
  const int q8 = nblocks >> 3, r8 = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
  const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int m0 = (vid / ntn) * BM, n0 = (vid % ntn) * BN;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int my_pieces = (P - wave + NW - 1) / NW;  // wave-uniform

  // per-lane source pointers of this wave's pieces (k0 = 0)
  const bf16_t* src[PMAX];
  unsigned dst[PMAX];
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_offset_of(&lds[0]));
#pragma unroll
  for (int i = 0; i < PMAX; ++i) {
    const int p = wave + NW * i;
    const int r = lane >> 2, pos = lane & 3;
    if (p < PA) {
      const int row = p * 16 + r;
      int gm = m0 + row;
      gm = gm < M ? gm : M - 1;
      src[i] = A + (long)gm * lda + ((pos ^ swz(row)) << 3);
    } else {
      const int row = (p - PA) * 16 + r;
      int gn = n0 + row;
      gn = gn < N ? gn : N - 1;
      src[i] = W + (long)gn * K + ((pos ^ swz(row)) << 3);
    }
    dst[i] = lds_base + (unsigned)p * 1024u;   // piece p starts at slot 64*p (A pieces first, then W)
  }
  auto issue = [&](int kt) {
    const unsigned sb = (unsigned)(kt % NSTAGE) * (STAGE_SLOTS * 16u);
#pragma unroll
    for (int i = 0; i < PMAX; ++i)
      if (i < my_pieces) dma16(src[i] + (kt << 5), dst[i] + sb);
  };
  // wait until at most `stages` of this wave's k-slices are still in flight
  auto wait_stages = [&](int stages) {
    if (my_pieces == PMAX) {
      if (stages >= 2) wait_vmcnt<2 * PMAX>();
      else if (stages == 1) wait_vmcnt<PMAX>();
      else wait_vmcnt<0>();
    } else {
      if (stages >= 2) wait_vmcnt<2 * (PMAX - 1)>();
      else if (stages == 1) wait_vmcnt<PMAX - 1>();
      else wait_vmcnt<0>();
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K >> 5;
  constexpr int AHEAD = NSTAGE - 1;            // k-slices issued ahead of the one being consumed
#pragma unroll
  for (int s = 0; s < AHEAD; ++s)
    if (s < nk) issue(s);
  for (int kt = 0; kt < nk; ++kt) {
    // slices kt+1 .. min(kt+AHEAD-1, nk-1) may stay in flight; AHEAD is 2 or 3
    const int inflight = (kt + AHEAD - 1 < nk ? AHEAD - 1 : nk - 1 - kt);
    wait_stages(inflight);
    __builtin_amdgcn_s_barrier();              // every wave's pieces of slice kt have landed; slice kt-1 is consumed
    if (kt + AHEAD < nk) issue(kt + AHEAD);
    const uint4* st = lds + (kt % NSTAGE) * STAGE_SLOTS;
    // all fragment reads of the k-slice are issued before the first MFMA: one LDS latency per slice
    // instead of one per column tile (the two waves of a SIMD then cover each other's read phase)
    uint4 afr[TM], bfr[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wave * 16 * TM + i * 16 + li;
      afr[i] = st[row * 4 + (kg ^ swz(row))];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = j * 16 + li;
      bfr[j] = st[BM * 4 + row * 4 + (kg ^ swz(row))];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (SWAP)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&bfr[j]),
                                                               *reinterpret_cast<bf16x8*>(&afr[i]), acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&afr[i]),
                                                               *reinterpret_cast<bf16x8*>(&bfr[j]), acc[i][j], 0, 0, 0);
      }
    }
  }
  wait_vmcnt<0>();

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

template <int NW, int TM, int TN, int NSTAGE, bool SWAP, class Epi>
void launch_tiled_dma_cfg(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  constexpr int BM = 16 * NW * TM, BN = 16 * TN;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  const int nblocks = ntm * ntn;
  hipLaunchKernelGGL((gemm_tiled_dma_kernel<NW, TM, TN, NSTAGE, SWAP, Epi>), dim3(nblocks), dim3(64 * NW), 0, s, A, lda,
                     W, M, N, K, ntn, nblocks, epi);
}

// MSH_GEMM_MODE (debug / A-B switch): 0 = register-staged double buffer; 1 = LDS-DMA, 256x208 tile, 4 waves,
// 4 stages (one workgroup per CU: measured ~2x slower than 2, nothing hides a wave's ds_read latency);
// 2 = LDS-DMA, 128x208 tile, 4 waves, 3 stages, two workgroups per CU (default);
// 3 = LDS-DMA, 256x208 tile, 8 waves (two per SIMD), 4 stages.
inline int gemm_mode() {
  static int mode = [] {
    const char* e = getenv("MSH_GEMM_MODE");
    return e ? atoi(e) : 2;
  }();
  return mode;
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
  if (N % 208 == 0 || (N % 144 != 0 && N >= 416)) {  // ragged last column tile (e.g. the 32768-wide LM head) is predicated
    const int mode = gemm_mode();
    if (mode == 0) {
      if (big)
        launch_tiled_cfg<4, 13, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
      else
        launch_tiled_cfg<2, 13, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    } else if (mode == 1 && big) {
      launch_tiled_dma_cfg<4, 4, 13, 4, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    } else if (mode == 3 && big) {
      launch_tiled_dma_cfg<8, 2, 13, 4, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    } else {
      launch_tiled_dma_cfg<4, 2, 13, 3, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    }
  } else if (N % 144 == 0) {
    launch_tiled_cfg<2, 9, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
  } else {
    launch_tiled_cfg<1, 4, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
  }
}

// ------------------------------------------------------------------------------------------------
// Decode GEMM (M = batch rows): fragment-direct, split-K inside the workgroup.
//
// A workgroup owns one 16 x (16*TN) output tile; its 4 waves split the K loop (wave w takes the 32-wide
// k-steps w, w+4, ...), each loading its MFMA fragments straight from global memory -- all loads of a
// wave are independent and issued up front, so the kernel costs about one memory round trip instead of a
// K/32-long dependent chain.  Partial tiles are summed through LDS in a fixed order (deterministic).
// LN = true: A is the fp32 residual stream [M][K]; LayerNorm (no bias, eps 1e-5, exact two-pass) is
// fused into the fragment build: row sums are combined across the 4 waves through LDS.
// ------------------------------------------------------------------------------------------------
template <int KS, int TN, bool LN, class Epi>
__global__ __launch_bounds__(256) void gemm_dec_kernel(const void* __restrict__ Aptr, long lda,
                                                       const float* __restrict__ gamma,
                                                       const bf16_t* __restrict__ W, int M, int N, int n_tiles,
                                                       Epi epi) {
  constexpr int K = 32 * KS;
  constexpr int KW = (KS + 3) / 4;  // k-steps per wave (upper bound)
  __shared__ __attribute__((aligned(16))) float4 part[4][TN][64];
  __shared__ float stat[2][4][16];
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = (blockIdx.x / n_tiles) * 16, n0 = (blockIdx.x % n_tiles) * (16 * TN);
  int gm = m0 + li;
  gm = gm < M ? gm : M - 1;

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
  constexpr int NE = (TN + 3) / 4;
  typename Epi::Pre epre[NE];
  {
    const int m = m0 + li;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int j = wave + 4 * e;
      const int n = n0 + j * 16 + kg * 4;
      if (j < TN && m < M && n < N) epre[e] = epi.pre(m, n);
    }
  }
  bf16x8 afrag[KW];
  if constexpr (LN) {
    const float* x = reinterpret_cast<const float*>(Aptr) + (long)gm * lda + kg * 8;
    float xv[KW][8];
    float4 gam[KW][2];
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      const int s = wave + 4 * i;
      if (s < KS) {
        gam[i][0] = *reinterpret_cast<const float4*>(gamma + s * 32 + kg * 8);
        gam[i][1] = *reinterpret_cast<const float4*>(gamma + s * 32 + kg * 8 + 4);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      const int s = wave + 4 * i;
      if (s < KS) {
        const float4 a = *reinterpret_cast<const float4*>(x + s * 32);
        const float4 b = *reinterpret_cast<const float4*>(x + s * 32 + 4);
        xv[i][0] = a.x; xv[i][1] = a.y; xv[i][2] = a.z; xv[i][3] = a.w;
        xv[i][4] = b.x; xv[i][5] = b.y; xv[i][6] = b.z; xv[i][7] = b.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += xv[i][e];
      }
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (kg == 0) stat[0][wave][li] = sum;
    __syncthreads();
    const float mean = ((stat[0][0][li] + stat[0][1][li]) + (stat[0][2][li] + stat[0][3][li])) * (1.0f / (float)K);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      if (wave + 4 * i < KS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = xv[i][e] - mean;
          sq += d * d;
        }
      }
    }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    if (kg == 0) stat[1][wave][li] = sq;
    __syncthreads();
    const float var = ((stat[1][0][li] + stat[1][1][li]) + (stat[1][2][li] + stat[1][3][li])) * (1.0f / (float)K);
    const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      const int s = wave + 4 * i;
      if (s < KS) {
        const float4 g0 = gam[i][0], g1 = gam[i][1];
        uint4 t;
        t.x = pack_bf16x2((xv[i][0] - mean) * rstd * g0.x, (xv[i][1] - mean) * rstd * g0.y);
        t.y = pack_bf16x2((xv[i][2] - mean) * rstd * g0.z, (xv[i][3] - mean) * rstd * g0.w);
        t.z = pack_bf16x2((xv[i][4] - mean) * rstd * g1.x, (xv[i][5] - mean) * rstd * g1.y);
        t.w = pack_bf16x2((xv[i][6] - mean) * rstd * g1.z, (xv[i][7] - mean) * rstd * g1.w);
        afrag[i] = *reinterpret_cast<bf16x8*>(&t);
      }
    }
  } else {
    const bf16_t* a = reinterpret_cast<const bf16_t*>(Aptr) + (long)gm * lda + kg * 8;
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      const int s = wave + 4 * i;
      if (s < KS) {
        uint4 t = *reinterpret_cast<const uint4*>(a + s * 32);
        afrag[i] = *reinterpret_cast<bf16x8*>(&t);
      }
    }
  }

  f32x4 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    if (wave + 4 * i < KS) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wreg[i][j]), afrag[i], acc[j], 0,
                                                         0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) part[wave][j][lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  __syncthreads();
  // waves 0..TN-1 (round-robin when TN > 4) finish one column tile each: fixed summation order
  const int m = m0 + li;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int j = wave + 4 * e;
    if (j < TN) {
      const float4 p0 = part[0][j][lane], p1 = part[1][j][lane], p2 = part[2][j][lane], p3 = part[3][j][lane];
      f32x4 v;
      v[0] = (p0.x + p1.x) + (p2.x + p3.x);
      v[1] = (p0.y + p1.y) + (p2.y + p3.y);
      v[2] = (p0.z + p1.z) + (p2.z + p3.z);
      v[3] = (p0.w + p1.w) + (p2.w + p3.w);
      const int n = n0 + j * 16 + kg * 4;
      if (m < M && n < N) epi.n4p(m, n, v, epre[e]);
    }
  }
}

template <int KS, int TN, bool LN, class Epi>
void launch_dec_cfg(const void* A, long lda, const float* gamma, const bf16_t* W, int M, int N, Epi epi,
                    hipStream_t s) {
  const int m_tiles = (M + 15) / 16, n_tiles = (N + 16 * TN - 1) / (16 * TN);
  hipLaunchKernelGGL((gemm_dec_kernel<KS, TN, LN, Epi>), dim3(m_tiles * n_tiles), dim3(256), 0, s, A, lda, gamma, W, M,
                     N, n_tiles, epi);
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
  launch_dec<2, true>(H, D, gamma, W, M, 3 * D, D, EpiDecQkv{q, cacheK, cacheV, pos_ptr, rp, Smax}, s);
}
void dec_gemm_ln_f32(const float* H, const float* gamma, const bf16_t* W, int M, int N, int D, float* out,
                     hipStream_t s) {
  launch_dec<2, true>(H, D, gamma, W, M, N, D, EpiF32{out, N}, s);
}
void dec_gemm_ln_swiglu(const float* H, const float* gamma, const bf16_t* W, const float* bias, int M, int F, int D,
                        bf16_t* z, hipStream_t s) {
  launch_dec<2, true>(H, D, gamma, W, M, 2 * F, D, EpiSwiGLU{z, F, bias}, s);
}
void dec_gemm_resid(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, float* H,
                    hipStream_t s) {
  launch_dec<2, false>(A, lda, nullptr, W, M, N, K, EpiResidF32{H, N, bias}, s);
}
void dec_gemm_logits(const float* H, const float* gamma, const bf16_t* E, int M, int V, int D, float* logits,
                     hipStream_t s) {
  launch_dec<4, true>(H, D, gamma, E, M, V, D, EpiF32{logits, V}, s);
}
void gemm_logits_f32(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, float* out, hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiF32{out, N}, s);
}

}  // namespace msh
