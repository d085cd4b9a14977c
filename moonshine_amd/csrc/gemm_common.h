// Shared pieces of the gfx950 GEMM kernels: fast activations, epilogue functors, LDS-DMA helpers.
// Included by k_gemm.hip (tiled encoder GEMMs) and k_gemm_dec.hip (decode GEMMs); everything is
// internal-linkage (anonymous namespace) device code.
#pragma once

#include <type_traits>

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
// gelu(x) = x * Phi(x) with Phi(x) ~ sigmoid(x * (a0 + a1 x^2 + a2 x^4)), x^2 clamped at 81 (beyond |x| = 9 the result is
// x or 0 to 1e-18): minimax fit against the erf form, max abs error 2.8e-5 over the whole line (tools/fit_gelu.py) -- a
// tenth of the bf16 rounding of the result -- in 9 instructions (2 transcendental) where the erf formula of gemm_common.h
// takes 19.  In the fused MLP kernel (one wave per SIMD) GELU shares the wave's issue slots with the MFMAs it overlaps; in
// the tiled GEMMs' epilogues it is plain VALU time behind the main loop.
__device__ __forceinline__ float gelu_sig(float x) {
  constexpr float kL2e = -1.4426950408889634f;   // sigmoid(t) = 1 / (1 + 2^(-t log2 e))
  constexpr float c0 = 1.5949708004086212f * kL2e, c1 = 0.07405211422714464f * kL2e, c2 = -0.000709652592810915f * kL2e;
  const float s = fminf(x * x, 81.0f);
  float p = fmaf(s, c2, c1);
  p = fmaf(p, s, c0);
  const float e = __builtin_amdgcn_exp2f(x * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// tanh(x) = 1 - 2 / (1 + e^{2x}); saturates cleanly (e^{2x} -> inf gives 1, -> 0 gives -1)
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------
// Epilogues.  n4(m, n, v): v[i] = C[m][n+i].   m4(m, n, v): v[i] = C[m+i][n].
// ------------------------------------------------------------------------------------------------
// conv1: tanh(acc) as bf16 (the GroupNorm that follows is folded into conv2, see EpiGnBiasGeluBf16)
struct EpiTanhBf16 {
  static constexpr bool kStagedBf16 = true;
  static constexpr bool kRowSums = true;
  bf16_t* out;
  long ldc;
  int nt = 0;   // 1: the output goes out with the non-temporal policy (staged_store_tile; MSH_STEM_STORE_NT)
  // GroupNorm statistics without a pass over the output: every (row, column tile) leaves the sum and the sum of squares of
  // the bf16 values it stored -- [M][column tiles] -- and groupnorm_stats_rows (k_misc.hip) adds up each clip's valid rows
  // in a fixed order (10 MB instead of the 532 MB of conv1 output per 256 clips).  null: no sums.
  float2* rowsum = nullptr;
  struct RowCtx {};
  struct ColCtx {};
  __device__ RowCtx row_ctx(int) const { return RowCtx{}; }
  __device__ ColCtx col_ctx(int) const { return ColCtx{}; }
  __device__ void col_next16(ColCtx&) const {}
  __device__ uint2 pack4(const RowCtx&, const ColCtx&, int, f32x4 v) const {
    uint2 o;
    o.x = pack_bf16x2(tanh_fast(v[0]), tanh_fast(v[1]));
    o.y = pack_bf16x2(tanh_fast(v[2]), tanh_fast(v[3]));
    return o;
  }
  __device__ void n4(int m, int n, f32x4 v) const {
    *reinterpret_cast<uint2*>(out + (long)m * ldc + n) = pack4(RowCtx{}, ColCtx{}, n, v);
  }
};

// conv2 with GroupNorm(1 group) folded in.  GN(x)[t,c] = (x - mu_b) * rstd_b * gamma_c + beta_c is affine per clip
// and per input channel, so  conv2(GN(x))[t,n] = rstd_b * acc[t,n] + (S2[n] + bias[n] - mu_b * rstd_b * S1[n])  where acc is
// the convolution of the RAW tanh output with W' = W * gamma (folded at load), S1[n] = sum W'[n,:,:] and
// S2[n] = sum W[n,:,c] * beta_c.  The normalised copy of the conv1 output (1.6 GB of traffic per 256 clips) is never
// materialised.  Row m of the conv2 output belongs to the clip of stream row m / 2.
struct EpiGnBiasGeluBf16 {
  static constexpr bool kStagedBf16 = true;
  static constexpr int kTaps = 7;   // conv2: K = 7 taps x C channels (conv_k_offset below)
  bf16_t* out;
  long ldc;
  const float* table;   // [clips][N]: S2[n] + bias[n] - mean_b * rstd_b * S1[n]   (gn_fold_table, once per batch)
  const float2* stats;  // per clip {mean, rstd}
  const int* row_clip;  // per stream row
  int nt = 0;           // 1: non-temporal output stores (see EpiTanhBf16)
  int kperm = 0;        // 1: W is stored in the tap-inner k-order
  struct RowCtx {
    float rstd;
    const float* trow;
  };
  struct ColCtx {};
  __device__ RowCtx row_ctx(int m) const {
    const int b = row_clip[m >> 1];
    return RowCtx{stats[b].y, table + (long)b * ldc};
  }
  __device__ ColCtx col_ctx(int) const { return ColCtx{}; }
  __device__ void col_next16(ColCtx&) const {}
  __device__ uint2 pack4(const RowCtx& r, const ColCtx&, int n, f32x4 v) const {
    const float4 t = *reinterpret_cast<const float4*>(r.trow + n);
    uint2 o;
    o.x = pack_bf16x2(gelu_sig(v[0] * r.rstd + t.x), gelu_sig(v[1] * r.rstd + t.y));
    o.y = pack_bf16x2(gelu_sig(v[2] * r.rstd + t.z), gelu_sig(v[3] * r.rstd + t.w));
    return o;
  }
  __device__ void n4(int m, int n, f32x4 v) const {
    *reinterpret_cast<uint2*>(out + (long)m * ldc + n) = pack4(row_ctx(m), ColCtx{}, n, v);
  }
};

struct EpiBiasGeluBf16 {
  static constexpr bool kStagedBf16 = true;
  bf16_t* out;
  long ldc;
  const float* bias;
  struct RowCtx {};
  struct ColCtx {};
  __device__ RowCtx row_ctx(int) const { return RowCtx{}; }
  __device__ ColCtx col_ctx(int) const { return ColCtx{}; }
  __device__ void col_next16(ColCtx&) const {}
  __device__ uint2 pack4(const RowCtx&, const ColCtx&, int n, f32x4 v) const {
    float4 b = *reinterpret_cast<const float4*>(bias + n);
    uint2 o;
    o.x = pack_bf16x2(gelu_sig(v[0] + b.x), gelu_sig(v[1] + b.y));
    o.y = pack_bf16x2(gelu_sig(v[2] + b.z), gelu_sig(v[3] + b.w));
    return o;
  }
  __device__ void n4(int m, int n, f32x4 v) const {
    *reinterpret_cast<uint2*>(out + (long)m * ldc + n) = pack4(RowCtx{}, ColCtx{}, n, v);
  }
};

struct EpiBiasGeluF32 {
  static constexpr int kTaps = 3;   // conv3: K = 3 taps x C channels
  float* out;
  long ldc;
  const float* bias;
  int kperm = 0;        // 1: W is stored in the tap-inner k-order
  __device__ void n4(int m, int n, f32x4 v) const {
    float4 b = *reinterpret_cast<const float4*>(bias + n);
    float4 o = make_float4(gelu_sig(v[0] + b.x), gelu_sig(v[1] + b.y), gelu_sig(v[2] + b.z), gelu_sig(v[3] + b.w));
    *reinterpret_cast<float4*>(out + (long)m * ldc + n) = o;
  }
};

// k-ORDER OF THE CONV GEMMS.  A stride-s convolution of k taps over channels-last rows is a GEMM whose A row m is the
// contiguous window of k * C elements that starts at input row s * m: element (m, tap t, channel c) IS element (m + 1, tap
// t - s, channel c).  Walking K tap-major (t outer, c inner -- the order of the window in memory) puts 13 .. 39 k-slices between
// the two uses of a byte: with 16 row tiles of 323 KB resident per XCD the L2 has dropped it by then, and conv2 fetched 3.1 x
// its algorithmic bytes (PMC).  Walking K channel-block-major (32 channels outer, taps inner) puts THREE
// slices between them.  Only the order of the k-slices changes: slice kt reads A at (kt % TAPS) * C + (kt / TAPS) * 32 instead
// of 32 kt, and W is stored in that order at load (Engine::load_weights), so its slices stay contiguous.  Epilogues that
// belong to a conv name their tap count (kTaps) and carry the switch (kperm; MSH_CONV_KORDER=0 at load: tap-major as before).
template <class E, class = void>
struct epi_taps : std::integral_constant<int, 1> {};
template <class E>
struct epi_taps<E, std::void_t<decltype(E::kTaps)>> : std::integral_constant<int, E::kTaps> {};
template <class Epi>
__device__ __forceinline__ int epi_kperm(const Epi& e) {
  if constexpr (epi_taps<Epi>::value > 1) return e.kperm;
  else return 0;
}
// element offset of k-slice kt inside an A row; kc = channels per tap
template <int TAPS>
__device__ __forceinline__ int conv_k_offset(int kt, int kc, int kperm) {
  if constexpr (TAPS > 1) {
    if (kperm) return (kt % TAPS) * kc + (kt / TAPS) * 32;
  }
  return kt << 5;
}

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
  struct Pre {};
  __device__ Pre pre(int, int) const { return Pre{}; }
  __device__ void n4p(int m, int n, f32x4 v, const Pre&) const { n4(m, n, v); }
  static constexpr bool kStagedBf16 = true;
  // RoPE factors of a tile's rows come from a per-wave LDS table (16 rows x [cos | sin] x 28 pairs, identity beyond
  // rot_pairs) that the wave fills with coalesced loads before it packs the tile: reading the global cos / sin
  // tables per accumulator group cost four scattered 4-byte loads per group, ~100 us of the 300 us kernel.
  static constexpr bool kRowTable = true;
  static constexpr int kTabRow = 56;  // floats per row: 28 cos + 28 sin
  struct RowCtx {
    const float* tab;  // this lane's row of the LDS table, or nullptr: factors come from the global tables
    int pos;
  };
  // (the table holds 28 pairs per row: heads wider than 56 keep the global-table path)
  __device__ bool tile_needs_table(int n0) const { return n0 < 2 * rp.hidden && rp.head_dim <= 56; }
  __device__ void fill_row_table(float* tab, int mbase, int M, int lane) const {
    for (int idx = lane; idx < 16 * kTabRow; idx += 64) {
      const int r = idx / kTabRow, c = idx - r * kTabRow;
      const int which = c >= 28, p = c - which * 28;
      int m = mbase + r;
      m = m < M ? m : M - 1;
      int pos = row_pos[m];
      pos = pos < 0 ? 0 : pos;
      float v = which ? 0.f : 1.f;
      if (p < rp.rot_pairs) v = (which ? rp.sin : rp.cos)[(long)pos * rp.rot_pairs + p];
      tab[idx] = v;
    }
  }
  __device__ RowCtx row_ctx_tab(const float* tab, int li) const { return RowCtx{tab + li * kTabRow, 0}; }
  __device__ RowCtx row_ctx(int m) const {
    const int pos = row_pos[m];
    return RowCtx{nullptr, pos < 0 ? 0 : pos};
  }
  struct ColCtx {
    int c, d;  // column within q / k / v, offset within its head
  };
  __device__ ColCtx col_ctx(int n) const {
    const int c = n % rp.hidden;
    return ColCtx{c, c % rp.head_dim};
  }
  __device__ void col_next16(ColCtx& k) const {  // head_dim >= 16
    k.c += 16;
    k.d += 16;
    if (k.c >= rp.hidden) {
      k.c -= rp.hidden;
      k.d = k.c % rp.head_dim;
    } else if (k.d >= rp.head_dim) {
      k.d -= rp.head_dim;
    }
  }
  __device__ uint2 pack4(const RowCtx& c, const ColCtx& k, int n, f32x4 v) const {
    if (n < 2 * rp.hidden) {  // q and k are rotated, v passes through
      const int d = k.d;
      if (c.tab != nullptr) {
        const int j0 = d >> 1;  // pairs j0, j0 + 1 (head_dim % 4 == 0)
        const float c0 = c.tab[j0], c1 = c.tab[j0 + 1], s0 = c.tab[28 + j0], s1 = c.tab[28 + j0 + 1];
        const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
        v[0] = x0 * c0 - x1 * s0;
        v[1] = x1 * c0 + x0 * s0;
        v[2] = x2 * c1 - x3 * s1;
        v[3] = x3 * c1 + x2 * s1;
      } else {
        rope4(v, d, c.pos, rp);
      }
    }
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    return o;
  }
  __device__ void n4(int m, int n, f32x4 v) const {  // unstaged paths: factors straight from the global tables
    if (n < 2 * rp.hidden) {
      int pos = row_pos[m];
      pos = pos < 0 ? 0 : pos;
      rope4(v, (n % rp.hidden) % rp.head_dim, pos, rp);
    }
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(out + (long)m * ldc + n) = o;
  }
};
template <class E, class = void>
struct has_row_table : std::false_type {};
template <class E>
struct has_row_table<E, std::void_t<decltype(E::kRowTable)>> : std::true_type {};

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

// Decode-kernel form of the residual epilogue: whether there is a bias is a template parameter, so `pre` is straight-line
// code (a runtime `bias != nullptr` test put a branch -- and with it a vmcnt(0) -- into the kernel's load phase).
template <bool BIAS>
struct EpiDecResid {
  float* H;
  long ldc;
  const float* bias;
  struct Pre {
    float4 h, b;
  };
  __device__ Pre pre(int m, int n) const {
    Pre p;
    p.h = *reinterpret_cast<const float4*>(H + (long)m * ldc + n);
    if constexpr (BIAS) p.b = *reinterpret_cast<const float4*>(bias + n);
    else p.b = make_float4(0.f, 0.f, 0.f, 0.f);
    return p;
  }
  __device__ void n4p(int m, int n, f32x4 v, const Pre& p) const {
    *reinterpret_cast<float4*>(H + (long)m * ldc + n) =
        make_float4(p.h.x + p.b.x + v[0], p.h.y + p.b.y + v[1], p.h.z + p.b.z + v[2], p.h.w + p.b.w + v[3]);
  }
};

// The same on the fragment-major residual stream of the offline decoder (kernels.h fm32): H is FM fp32 [M16][N],
// ksteps = N / 32.  A lane's 4 consecutive columns are contiguous in FM as well (n % 4 == 0).
template <bool BIAS>
struct EpiDecResidFm {
  float* H;
  int ksteps;
  const float* bias;
  struct Pre {
    float4 h, b;
  };
  __device__ Pre pre(int m, int n) const {
    Pre p;
    p.h = *reinterpret_cast<const float4*>(H + fm32(m, n, ksteps));
    if constexpr (BIAS) p.b = *reinterpret_cast<const float4*>(bias + n);
    else p.b = make_float4(0.f, 0.f, 0.f, 0.f);
    return p;
  }
  __device__ void n4p(int m, int n, f32x4 v, const Pre& p) const {
    *reinterpret_cast<float4*>(H + fm32(m, n, ksteps)) =
        make_float4(p.h.x + p.b.x + v[0], p.h.y + p.b.y + v[1], p.h.z + p.b.z + v[2], p.h.w + p.b.w + v[3]);
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
  // Paired form for kernels whose wave owns 32 consecutive keys (two 16-row tiles): lanes kg and kg^1 swap halves so
  // that every lane stores 8 consecutive keys (16 bytes) and the four lanes of a column cover one full 64-byte
  // sector.  With 8-byte stores the PMC pass showed 2.4 GB written per launch for 1.43 GB of K^T / V^T.
  // Needs clip row_start and Tk to be multiples of 8 (plan_batch pads rows to 8).
  static constexpr bool kPairedKeys = true;
  struct KeyRow {  // per-lane row bookkeeping, looked up once per tile instead of once per column
    int t, T, Tk;
    long off;  // kv_start * D + t
  };
  __device__ KeyRow key_row(int m) const {
    const ClipMeta cm = clips[row_clip[m]];
    KeyRow k;
    k.t = m - cm.row_start;
    k.T = cm.T;
    k.Tk = cm.Tk;
    k.off = (long)cm.kv_start * D + k.t;
    return k;
  }
  __device__ uint2 pack_keys4(const KeyRow& k, f32x4 v) const {  // this lane's 4 keys, zeroed past the frame count
    uint2 o;
    o.x = pack_bf16x2(k.t < k.T ? v[0] : 0.f, k.t + 1 < k.T ? v[1] : 0.f);
    o.y = pack_bf16x2(k.t + 2 < k.T ? v[2] : 0.f, k.t + 3 < k.T ? v[3] : 0.f);
    return o;
  }
  __device__ void store_keys8(const KeyRow& k, int n, uint4 v) const {  // k = first of 8 consecutive keys of one clip
    if (k.t >= k.Tk) return;
    const int layer = n / (2 * D);
    const int r = n - layer * 2 * D;
    const int which = r / D;
    const int c = r - which * D;
    *reinterpret_cast<uint4*>((which ? VT : KT) + layer * layer_stride + k.off + (long)c * k.Tk) = v;
  }
};
// The same tensor as e4m3 bytes (engine option kv_dtype = fp8): value * qscale[n], qscale fixed per output column at load
// from a bound on |value| (Engine::load_weights), so nothing data-dependent has to be known before the store.  The decode
// kernel multiplies the query (K) and the output (V) by 1 / qscale.  Layout as above with one byte per key.
struct EpiCrossKVFp8 {
  uint8_t* KT;
  uint8_t* VT;
  const int* row_clip;
  const ClipMeta* clips;
  int D;
  long layer_stride;     // bytes per layer = D * sum(Tk)
  const float* qscale;   // [N]
  static __device__ __forceinline__ uint32_t pack4(float a, float b, float c, float d) {
    // v_cvt_pk_fp8_f32 does not saturate: keep the (already bounded) values inside e4m3's +-448
    a = fminf(fmaxf(a, -448.f), 448.f);
    b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f);
    d = fminf(fmaxf(d, -448.f), 448.f);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (uint32_t)p;
  }
  __device__ void m4(int m, int n, f32x4 v) const {
    const int b = row_clip[m];
    const ClipMeta cm = clips[b];
    const int t = m - cm.row_start;
    if (t >= cm.Tk) return;
    const int layer = n / (2 * D);
    const int r = n - layer * 2 * D;
    const int which = r / D;
    const int c = r - which * D;
    const float qs = qscale[n];
    uint8_t* base = (which ? VT : KT) + layer * layer_stride + (long)cm.kv_start * D + (long)c * cm.Tk + t;
    *reinterpret_cast<uint32_t*>(base) = pack4(t < cm.T ? v[0] * qs : 0.f, t + 1 < cm.T ? v[1] * qs : 0.f,
                                               t + 2 < cm.T ? v[2] * qs : 0.f, t + 3 < cm.T ? v[3] * qs : 0.f);
  }
  static constexpr bool kPairedKeysFp8 = true;
  struct KeyRow {
    int t, T, Tk;
    long off;  // kv_start * D + t
  };
  __device__ KeyRow key_row(int m) const {
    const ClipMeta cm = clips[row_clip[m]];
    KeyRow k;
    k.t = m - cm.row_start;
    k.T = cm.T;
    k.Tk = cm.Tk;
    k.off = (long)cm.kv_start * D + k.t;
    return k;
  }
  __device__ uint32_t pack_keys4(const KeyRow& k, f32x4 v, float qs) const {
    return pack4(k.t < k.T ? v[0] * qs : 0.f, k.t + 1 < k.T ? v[1] * qs : 0.f, k.t + 2 < k.T ? v[2] * qs : 0.f,
                 k.t + 3 < k.T ? v[3] * qs : 0.f);
  }
  __device__ void store_keys8(const KeyRow& k, int n, uint2 v) const {  // k = first of 8 consecutive keys of one clip
    if (k.t >= k.Tk) return;
    const int layer = n / (2 * D);
    const int r = n - layer * 2 * D;
    const int which = r / D;
    const int c = r - which * D;
    *reinterpret_cast<uint2*>((which ? VT : KT) + layer * layer_stride + k.off + (long)c * k.Tk) = v;
  }
};
template <class E, class = void>
struct is_paired_keys_fp8 : std::false_type {};
template <class E>
struct is_paired_keys_fp8<E, std::void_t<decltype(E::kPairedKeysFp8)>> : std::true_type {};

template <class E, class = void>
struct is_paired_keys : std::false_type {};
template <class E>
struct is_paired_keys<E, std::void_t<decltype(E::kPairedKeys)>> : std::true_type {};

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
    // straight-line: the factor loads are unconditional (index clamped into the table row) and the identity is
    // selected afterwards for v columns and for pairs beyond rot_pairs -- no branch in the kernel's load phase
    Pre p;
    p.pos = *pos_ptr;
    const int d = (n % rp.hidden) % rp.head_dim, j0 = d >> 1;
    const int ja = j0 < rp.rot_pairs ? j0 : rp.rot_pairs - 1, jb = j0 + 1 < rp.rot_pairs ? j0 + 1 : rp.rot_pairs - 1;
    const float* cr = rp.cos + (long)p.pos * rp.rot_pairs;
    const float* sr = rp.sin + (long)p.pos * rp.rot_pairs;
    const float c0 = cr[ja], s0 = sr[ja], c1 = cr[jb], s1 = sr[jb];
    const bool rot = n < 2 * rp.hidden;   // q / k are rotated, v passes through
    const bool r0 = rot && j0 < rp.rot_pairs, r1 = rot && j0 + 1 < rp.rot_pairs;
    p.c0 = r0 ? c0 : 1.f;
    p.s0 = r0 ? s0 : 0.f;
    p.c1 = r1 ? c1 : 1.f;
    p.s1 = r1 ? s1 : 0.f;
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

// Keys-side queries of the absorbed cross-attention (k_xattn.hip): column n = head * D + d of row m (a clip) goes out as
// TWO bf16 values, x rounded and the rounding residual x - bf16(x), in the order the attention kernel's score product takes
// its B operand, k-step by k-step: qf[clip][d / 32][16 rows][4][8] with row = head (value) or head + 8 (residual) and the
// 32 features of a k-step contiguous per row -- a lane (row, kg) of that kernel loads its 16 bytes at (row * 4 + kg) * 16 of
// the k-step's KiB (the wave's load is the whole KiB), and a 16-column tile of this GEMM lands in one 32-byte run per clip.
struct EpiQtFrag {
  bf16_t* qf;
  int D;   // model width; N = heads * D, D % 32 == 0
  struct Pre {};
  __device__ Pre pre(int, int) const { return Pre{}; }
  __device__ void n4p(int m, int n, f32x4 v, const Pre&) const { n4(m, n, v); }
  __device__ void n4(int m, int n, f32x4 v) const {
    const int h = n / D, d = n - h * D;
    bf16_t* frag = qf + ((size_t)m * (D >> 5) + (d >> 5)) * 512 + (d & 31);
    uint2 hi, lo;
    hi.x = pack_bf16x2(v[0], v[1]);
    hi.y = pack_bf16x2(v[2], v[3]);
    lo.x = pack_bf16x2(v[0] - __uint_as_float(hi.x << 16), v[1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack_bf16x2(v[2] - __uint_as_float(hi.y << 16), v[3] - __uint_as_float(hi.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(frag + h * 32) = hi;
    *reinterpret_cast<uint2*>(frag + (h + 8) * 32) = lo;
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

// SwiGLU writing the fragment-major bf16 activation of the offline decoder's fc2 (kernels.h fm16): z is FM [M16][F],
// ksteps = F / 32; the lane's two outputs (columns n/2, n/2 + 1) share one 16-byte chunk.
struct EpiSwiGLUFm {
  bf16_t* z;
  int ksteps;
  const float* bias;
  struct Pre {
    float4 b;
  };
  __device__ Pre pre(int, int n) const { return Pre{*reinterpret_cast<const float4*>(bias + n)}; }
  __device__ void n4p(int m, int n, f32x4 v, const Pre& p) const {
    const float val0 = v[0] + p.b.x, gate0 = v[1] + p.b.y, val1 = v[2] + p.b.z, gate1 = v[3] + p.b.w;
    *reinterpret_cast<uint32_t*>(z + fm16(m, n >> 1, ksteps)) = pack_bf16x2(silu_f(gate0) * val0, silu_f(gate1) * val1);
  }
};

// Streaming decoder QKV: row m belongs to stream row_slot[m] at position row_pos[m].  q (RoPE applied) goes to
// q_out [M][D]; k (RoPE) and v go straight into the stream's self-attention cache [slot][L][Scap][D] at that
// position -- no [M][3D] round trip and no separate append kernel.
struct EpiStreamQkv {
  bf16_t* q_out;
  bf16_t* cacheK;
  bf16_t* cacheV;
  const int* row_slot;
  const int* row_pos;
  RopeParams rp;
  int layer, L, Scap;
  struct Pre {};
  __device__ Pre pre(int, int) const { return Pre{}; }
  __device__ void n4p(int m, int n, f32x4 v, const Pre&) const { n4(m, n, v); }
  __device__ void n4(int m, int n, f32x4 v) const {
    const int D = rp.hidden;
    const int which = n / D, c = n - which * D;
    int pos = row_pos[m];
    pos = pos < 0 ? 0 : pos;
    if (which < 2) rope4(v, c % rp.head_dim, pos, rp);
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    bf16_t* dst;
    if (which == 0)
      dst = q_out + (long)m * D + c;
    else
      dst = (which == 1 ? cacheK : cacheV) + (((long)row_slot[m] * L + layer) * Scap + pos) * D + c;
    *reinterpret_cast<uint2*>(dst) = o;
  }
};

// Generic epilogue of the streaming path (frontend linear / causal convs, projections without a fused
// consumer): out = act(acc + bias) written as bf16 and / or fp32.  act: 0 none, 1 SiLU, 2 GELU(erf).
struct EpiAct {
  bf16_t* out16;     // nullable
  float* out32;      // nullable
  long ldc;
  const float* bias;  // nullable
  int act;
  struct Pre {};
  __device__ Pre pre(int, int) const { return Pre{}; }
  __device__ void n4p(int m, int n, f32x4 v, const Pre&) const { n4(m, n, v); }
  __device__ void n4(int m, int n, f32x4 v) const {
    if (bias != nullptr) {
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      v[0] += b.x;
      v[1] += b.y;
      v[2] += b.z;
      v[3] += b.w;
    }
    if (act == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = silu_f(v[i]);
    } else if (act == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = gelu_sig(v[i]);
    }
    if (out16 != nullptr) {
      uint2 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(out16 + (long)m * ldc + n) = o;
    }
    if (out32 != nullptr) *reinterpret_cast<float4*>(out32 + (long)m * ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// LM head without the logits round trip: every workgroup reduces its 128 x 208 tile to one (max, first index)
// pair per row; the decode bookkeeping kernel then reduces N/208 pairs per row instead of scanning V logits.
// Handled inside gemm_tiled_dma_kernel (kRowArgmax), not through n4.
struct EpiArgmaxPartial {
  static constexpr bool kRowArgmax = true;
  float* pval;  // [M][ntn]
  int* pidx;
  int ntn;
  __device__ void n4(int, int, f32x4) const {}
};
template <class E, class = void>
struct is_row_argmax : std::false_type {};
template <class E>
struct is_row_argmax<E, std::void_t<decltype(E::kRowArgmax)>> : std::true_type {};

// Row-major bf16 outputs: the accumulator layout gives every store instruction sixteen 32-byte row segments, and the
// microbenchmark shows those stores, not the MFMA loop, holding the wide-N GEMMs back (fc1: 0.166 ms without the
// epilogue, 0.295 ms with it -- profiles/r01z_gemm_epilogue_probe.txt).  Epilogues that declare kStagedBf16 hand back
// packed values (pack4) and the kernel transposes a 16-row slab through LDS so that each lane stores 16 contiguous
// bytes and a wave covers whole rows of the tile.
template <class E, class = void>
struct is_staged_bf16 : std::false_type {};
template <class E>
struct is_staged_bf16<E, std::void_t<decltype(E::kStagedBf16)>> : std::true_type {};

template <int TN>
struct StagedRow {
  static constexpr int ROWP = TN * 4 + 2;  // uint2 per staged row: 16 * TN columns + 16 bytes of padding
};
// acc = the TN accumulators of one 16-row tile of this wave; stg = this wave's [16][ROWP] uint2 staging slab
// epilogues that carry a run-time `nt` flag (non-temporal output stores)
template <class E, class = void>
struct has_nt_flag : std::false_type {};
template <class E>
struct has_nt_flag<E, std::void_t<decltype(std::declval<const E&>().nt)>> : std::true_type {};
typedef unsigned int u32x4_native __attribute__((ext_vector_type(4)));

// epilogues that leave per-(row, column tile) sums of what they stored (EpiTanhBf16::rowsum)
template <class E, class = void>
struct has_row_sums : std::false_type {};
template <class E>
struct has_row_sums<E, std::void_t<decltype(E::kRowSums)>> : std::true_type {};
// the four bf16 values of a packed group, as stored, into a row's running sums
__device__ __forceinline__ void row_sums_add(uint2 pk, float& s, float& q) {
  const float a = __uint_as_float(pk.x << 16), b = __uint_as_float(pk.x & 0xffff0000u);
  const float c = __uint_as_float(pk.y << 16), d = __uint_as_float(pk.y & 0xffff0000u);
  s += (a + b) + (c + d);
  q += (a * a + b * b) + (c * c + d * d);
}
// the four lane groups (kg = lane >> 4) of a row hold its partial sums: fixed-order exchange, every lane ends with the total
__device__ __forceinline__ void row_sums_reduce(float& s, float& q) {
  s += __shfl_xor(s, 16, 64);
  q += __shfl_xor(q, 16, 64);
  s += __shfl_xor(s, 32, 64);
  q += __shfl_xor(q, 32, 64);
}

template <int TN, class Epi>
__device__ __forceinline__ void staged_store_tile(const Epi& epi, uint2* __restrict__ stg, const f32x4 (&acc)[TN],
                                                  int mbase, int n0, int M, int N, int lane, float* rowtab = nullptr,
                                                  int ntn = 0, int tile_n = 0) {
  constexpr int ROWP = StagedRow<TN>::ROWP;
  const int li = lane & 15, kg = lane >> 4;
  auto ctx = epi.row_ctx(mbase + li < M ? mbase + li : M - 1);  // per-row inputs, fetched once per tile
  if constexpr (has_row_table<Epi>::value) {
    if (rowtab != nullptr && epi.tile_needs_table(n0)) {
      epi.fill_row_table(rowtab, mbase, M, lane);
      __builtin_amdgcn_wave_barrier();
      ctx = epi.row_ctx_tab(rowtab, li);
    }
  }
  // per-column inputs advance by 16 columns per accumulator: no integer division inside the loop (the two runtime
  // modulos per group that RoPE needs cost more than its arithmetic: ~100 us of the 300 us QKV kernel)
  auto col = epi.col_ctx(n0 + kg * 4);
  [[maybe_unused]] float rs_s = 0.f, rs_q = 0.f;   // row sums of what this lane packs (has_row_sums epilogues)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    // a column group beyond N (ragged last tile, e.g. N = 3072 on 208-wide tiles) is dropped by the stores below, but
    // pack4 READS per-column inputs (bias, folded GroupNorm table): it gets the last valid group's address instead of one
    // up to 188 bytes past the end of a 1-D parameter -- a GPU memory fault when that parameter ends on a mapping boundary
    const int n = n0 + j * 16 + kg * 4;
    const uint2 pk = epi.pack4(ctx, col, n < N ? n : N - 4, acc[j]);
    stg[li * ROWP + j * 4 + kg] = pk;
    if constexpr (has_row_sums<Epi>::value) {
      if (n < N) row_sums_add(pk, rs_s, rs_q);
    }
    epi.col_next16(col);
  }
  if constexpr (has_row_sums<Epi>::value) {
    if (epi.rowsum != nullptr) {   // (wave-uniform)
      row_sums_reduce(rs_s, rs_q);
      if (kg == 0 && mbase + li < M) epi.rowsum[(long)(mbase + li) * ntn + tile_n] = make_float2(rs_s, rs_q);
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q0 = 0; q0 < 16 * TN * 2; q0 += 64) {
    const int q = q0 + lane;
    if (q < 16 * TN * 2) {
      const int r = q / (TN * 2), c = q - r * (TN * 2);
      const uint4 v = *reinterpret_cast<const uint4*>(stg + r * ROWP + c * 2);
      const int m = mbase + r, n = n0 + c * 8;
      if (m < M) {
        bf16_t* dst = epi.out + (long)m * epi.ldc + n;
        bool nt = false;
        if constexpr (has_nt_flag<Epi>::value) nt = epi.nt != 0;
        if (n + 8 <= N) {
          if (nt) __builtin_nontemporal_store(u32x4_native{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_native*>(dst));
          else *reinterpret_cast<uint4*>(dst) = v;
        } else if (n + 4 <= N)
          *reinterpret_cast<uint2*>(dst) = make_uint2(v.x, v.y);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// The unstaged n4 epilogue of an epilogue with row sums (EpiTanhBf16): what its n4 stores, plus the sums.  mwave0 = first row
// of this wave's TM 16-row tiles; the shuffles run on every lane (rows >= M contribute zeros and store nothing).
template <int TM, int TN, class Epi>
__device__ __forceinline__ void store_rows_with_sums(const Epi& epi, const f32x4 (&acc)[TM][TN], int mwave0, int n0, int M, int N,
                                                     int ntn, int tile_n, int li, int kg) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mwave0 + i * 16 + li;
    float rs = 0.f, rq = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + j * 16 + kg * 4;
      if (m < M && n < N) {
        const uint2 pk = epi.pack4(typename Epi::RowCtx{}, typename Epi::ColCtx{}, n, acc[i][j]);
        *reinterpret_cast<uint2*>(epi.out + (long)m * epi.ldc + n) = pk;
        row_sums_add(pk, rs, rq);
      }
    }
    if (epi.rowsum != nullptr) {   // (wave-uniform)
      row_sums_reduce(rs, rq);
      if (kg == 0 && m < M) epi.rowsum[(long)m * ntn + tile_n] = make_float2(rs, rq);
    }
  }
}

// XOR swizzle of the 16-B k-chunk position inside a 64-B row of an LDS k-slice (conflict-free
// ds_read_b128 fragment reads for the lane groups of gfx950; verified: SQ_LDS_BANK_CONFLICT = 0)
__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }

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
__device__ __forceinline__ void dma16_nt(const void* gsrc, unsigned lds_base) {   // the same with the non-temporal policy
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
// Up to four consecutive KiB from gsrc (per lane: + 16 * lane) to LDS at lds_base (wave-uniform) + 16 * lane in ONE setting of
// M0 and of the address register: the instruction's offset field moves both the global and the LDS address, so pieces that lie
// 1 KiB apart on both sides need nothing between them -- 8 instructions for 4 KiB where four dma16 calls (M0 saved / set /
// restored and a 64-bit address add each) are 28.
template <int N>
__device__ __forceinline__ void dma16_run(const void* gsrc, unsigned lds_base) {
  static_assert(N >= 1 && N <= 4, "the offset field reaches 4095");
  unsigned keep;
  if constexpr (N == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
  else if constexpr (N == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace
}  // namespace msh
