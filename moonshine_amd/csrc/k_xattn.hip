// Decode cross-attention over the ENCODER OUTPUT itself ("absorbed" projections) for gfx950.
//
// The classic form (k_attn.hip, dec_cross_attention_kernel) streams K^T and V^T of a clip: 2 * T * D bf16 per layer
// and step, 1.4 GB per decode step at 256 x 10 s -- the kernel that bounds batched decode (5.5 MB per
// clip per step).  But K and V are both linear images of the same T x D encoder output E (reference graph:
// encoder_attn.k_proj / v_proj, transformers modeling_moonshine.py:265-330), and D = heads * head_dim, so K | V is a
// 2D-wide expansion of a D-wide row.  Moving the two projections to the other side of the products,
//
//     score_h[t] = q_h . K_h[t]        = (Wk_h^T q_h) . E[t]          =: qt_h . E[t]
//     out        = sum_h Wo_h (P_h V_h) = sum_h (Wo_h Wv_h) (P_h E)    =: sum_h Wvo_h ctx_h,   ctx_h = sum_t p_h[t] E[t]
//
// the attention of ALL heads of a clip needs ONE pass over E: T * D bf16 = half the bytes, the same for every decoder
// layer, and no cross-K/V projection in the encoder at all.  The price is arithmetic (every head works on D-wide rows
// instead of head_dim-wide ones: 8x the flops), which is why both products run on the matrix pipe here, and two wider
// decode GEMMs around the kernel: qt = LN(h) Wqk^T with Wqk = [heads * D][D] (Wk_h^T Wq_h stacked, softmax scale and the
// LayerNorm scale folded in) and the residual update h += ctx Wvo^T with Wvo = [D][heads * D] (both built at load,
// engine.cpp).  In exact arithmetic the result is the reference's; in bf16 it rounds at different points (E is read
// as stored, qt and p are split into two bf16 halves so both products see ~16 mantissa bits of them, ctx is rounded to
// bf16 where the classic form rounded P V), and it is held to the same parity gates (tests/test_gpu_xattn.py).
//
// Kernel: one workgroup per clip, 4 waves.  Wave w owns the 16-key tiles w, w + 4, ... of the clip, each fetched by
// global_load_lds_dwordx4 into a wave-private 2-slot LDS ring (rows at a pitch of 2D + 32 bytes: conflict-free for both
// read patterns below), and keeps its own online-softmax state -- no barrier inside the loop.  Per tile:
//   S^T [16 keys x 16]   = E_tile (A: rows = keys, ds_read_b128) x Qt^T (B: columns 0-7 = high halves of the 8 heads' qt,
//                          8-15 = low halves; resident in registers), MFMA 16x16x32, D/32 steps;
//   hi + lo columns added (DPP row rotate), fp32 online softmax per head column, P split hi / lo the same way;
//   C^T [D x 16]        += E_tile^T (A: rows = feature d, ds_read_b64_tr_b16: the hardware 4x4 transposing read) x P^T
//                          (B: the S^T accumulator layout IS the 16x16x16 B-operand layout), MFMA 16x16x16, D/16 tiles.
// At the end the four waves' (m, l, C) are merged through LDS (the ring is dead by then) and ctx is written as the bf16
// fragment-major A operand of the residual GEMM (kernels.h fm16, K = heads * D).
#include <stdlib.h>

#include <vector>

#include "gemm_common.h"

namespace msh {
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

template <int D>
struct XaCfg {
  static constexpr int KS = D / 32;                        // k-steps of the score product
  static constexpr int DT = D / 16;                        // 16-row tiles of the context product
  static constexpr int ROWB = D * 2 + 32;                  // LDS row pitch in bytes (= 96 mod 256 for D = 416 and 288)
  static constexpr int TILE = 16 * ROWB;                   // bytes a 16-key tile occupies
  static constexpr int PIECES = (TILE + 1023) / 1024;      // 1 KiB DMA instructions per tile
  static constexpr int SLOT = PIECES * 1024;
  static constexpr int NSLOT = (160 * 1024 / 4 / SLOT) >= 3 ? 3 : 2;
  static constexpr int RING = 4 * NSLOT * SLOT;
  static constexpr int MERGE = 4 * 8 * D * 4 + 4 * 8 * 8;  // [wave][head][D] fp32 + [wave][head] {m, l}
  static constexpr int LDS = RING > MERGE ? RING : MERGE;
  static_assert(D % 32 == 0, "D must be a multiple of 32");
  static_assert(ROWB % 256 == 96, "row pitch chosen for conflict-free ds_read_b128 / ds_read_b64_tr_b16");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ float dpp_ror8(float v) {   // value of lane (l ^ 8) within every row of 16 lanes
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, true));
}
// reductions across the four 16-lane rows of a wave (lane ^ 16, lane ^ 32): v_permlane{16,32}_swap, pure VALU
__device__ __forceinline__ float xa_rows_max(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const unsigned w = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float xa_rows_sum(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned w = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float bf16_round(float x) {   // x rounded to bf16 (RNE), as fp32
  return __uint_as_float(pack_bf16x2(x, 0.f) << 16);
}

// TR = true: the context product's A fragments come from ds_read_b64_tr_b16; false: four 2-byte reads per fragment (the
// plain formulation of the same gather, kept as the check of the transposing read: MSH_XATTN_TR=0)
template <int D, bool TR>
__global__ __launch_bounds__(256, 2) void dec_cross_absorbed_kernel(const float* __restrict__ qt,      // [M][8 * D]
                                                               const bf16_t* __restrict__ enc,    // [R][D]
                                                               const ClipMeta* __restrict__ clips,
                                                               bf16_t* __restrict__ ctx) {       // FM [M16][8 * D]
  using C = XaCfg<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  const ClipMeta cm = clips[b];
  const int T = cm.T;
  const int n_tiles = (T + 15) >> 4;
  const char* ebase = reinterpret_cast<const char*>(enc + (long)cm.row_start * D);
  const unsigned ring = lds_offset_of(smem) + (unsigned)wave * (C::NSLOT * C::SLOT);

  // byte offset (within a tile's 16 x D block of E) that DMA piece j of this lane fetches; pad lanes re-read offset 0
  int goff[C::PIECES];
#pragma unroll
  for (int j = 0; j < C::PIECES; ++j) {
    const unsigned o = (unsigned)(j * 1024 + lane * 16);
    const unsigned r = o / (unsigned)C::ROWB, c = o - r * (unsigned)C::ROWB;
    goff[j] = (r < 16u && c < (unsigned)(D * 2)) ? (int)(r * (unsigned)(D * 2) + c) : 0;
  }
  auto issue_tile = [&](int tile, int slot) {
    const char* src = ebase + (long)tile * (16 * D * 2);
    const unsigned dst = ring + (unsigned)slot * C::SLOT;
    if (tile * 16 + 16 <= T) {
#pragma unroll
      for (int j = 0; j < C::PIECES; ++j) dma16(src + goff[j], dst + (unsigned)j * 1024u);
    } else {   // last tile of a clip whose frame count is not a multiple of 16: rows past the end re-read the last valid row
      const int last = T - 1 - tile * 16;   // >= 0
#pragma unroll
      for (int j = 0; j < C::PIECES; ++j) {
        const unsigned o = (unsigned)(j * 1024 + lane * 16);
        unsigned r = o / (unsigned)C::ROWB;
        const unsigned c = o - r * (unsigned)C::ROWB;
        const bool ok = r < 16u && c < (unsigned)(D * 2);
        r = r < (unsigned)last ? r : (unsigned)last;
        dma16(src + (ok ? (int)(r * (unsigned)(D * 2) + c) : 0), dst + (unsigned)j * 1024u);
      }
    }
  };
  // the first NSLOT tiles of this wave go out before anything else
#pragma unroll
  for (int sl = 0; sl < C::NSLOT; ++sl)
    if (wave + 4 * sl < n_tiles) issue_tile(wave + 4 * sl, sl);

  // B operand of the score product: column li = head (li & 7), high half (li < 8) or low half of qt; k = kg * 8 .. + 8
  bf16x8 qfrag[C::KS];
  {
    const float* qrow = qt + ((long)b * 8 + (li & 7)) * D + kg * 8;
    float4 qa[C::KS], qb[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      qa[ks] = *reinterpret_cast<const float4*>(qrow + ks * 32);
      qb[ks] = *reinterpret_cast<const float4*>(qrow + ks * 32 + 4);
    }
    // low-half lanes keep x - bf16(x), high-half lanes x itself (branch-free: the subtrahend is scaled by 0 or 1)
    const float lo = li >= 8 ? 1.0f : 0.0f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      float x[8] = {qa[ks].x, qa[ks].y, qa[ks].z, qa[ks].w, qb[ks].x, qb[ks].y, qb[ks].z, qb[ks].w};
      uint4 u;
      uint32_t* up = reinterpret_cast<uint32_t*>(&u);
#pragma unroll
      for (int e = 0; e < 8; e += 2)
        up[e >> 1] = pack_bf16x2(fmaf(-lo, bf16_round(x[e]), x[e]), fmaf(-lo, bf16_round(x[e + 1]), x[e + 1]));
      qfrag[ks] = *reinterpret_cast<bf16x8*>(&u);
    }
  }

  f32x4 acc[C::DT];
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_part = 0.f;
  const float lo_lane = li >= 8 ? 1.0f : 0.0f;   // see qfrag: low-half columns carry x - bf16(x)
  // per-lane LDS offsets inside a slot: score A fragment (row = key li, 16 bytes at kg) and context A fragment
  const unsigned a1_off = (unsigned)(li * C::ROWB + kg * 16);
  const unsigned a2_off = TR ? (unsigned)((kg * 4 + (li >> 2)) * C::ROWB + (li & 3) * 8)
                             : (unsigned)(kg * 4 * C::ROWB + li * 2);

  int slot = 0;
#pragma unroll 1
  for (int tile = wave; tile < n_tiles; tile += 4) {
    if constexpr (C::NSLOT == 2) {
      if (tile + 4 < n_tiles) wait_vmcnt<C::PIECES>();
      else wait_vmcnt<0>();
    } else {
      if (tile + 8 < n_tiles) wait_vmcnt<2 * C::PIECES>();
      else if (tile + 4 < n_tiles) wait_vmcnt<C::PIECES>();
      else wait_vmcnt<0>();
    }
    const char* sbase = smem + (size_t)wave * (C::NSLOT * C::SLOT) + (size_t)slot * C::SLOT;

    // ---- scores: S^T[key][col] over D, two accumulation chains ----
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(sbase + a1_off + ks * 64);
      if (ks & 1) s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qfrag[ks], s1, 0, 0, 0);
      else s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qfrag[ks], s0, 0, 0, 0);
    }
    float s[4];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = s0[r] + s1[r];
      const float t = v + dpp_ror8(v);   // high-half column + low-half column of the same head
      s[r] = (tile * 16 + kg * 4 + r < T) ? t : -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = xa_rows_max(mx);
    // The softmax reference m_run moves only when a column's maximum outgrows it by more than 2^16 (always on the wave's
    // first tile: m_run = -inf): p <= 2^16 is harmless in fp32 / split bf16, and the 4 * D/16 accumulators are rescaled a
    // couple of times per clip instead of once per tile.
    if (__builtin_amdgcn_ballot_w64(mx > m_run + 16.0f) != 0) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_part *= alpha;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        acc[dt][0] *= alpha; acc[dt][1] *= alpha; acc[dt][2] *= alpha; acc[dt][3] *= alpha;
      }
    }
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(s[r] - m_run);
    l_part += (p[0] + p[1]) + (p[2] + p[3]);
    s16x4 pfrag;
    {
      uint2 u;
      u.x = pack_bf16x2(fmaf(-lo_lane, bf16_round(p[0]), p[0]), fmaf(-lo_lane, bf16_round(p[1]), p[1]));
      u.y = pack_bf16x2(fmaf(-lo_lane, bf16_round(p[2]), p[2]), fmaf(-lo_lane, bf16_round(p[3]), p[3]));
      pfrag = *reinterpret_cast<s16x4*>(&u);
    }
    // ---- context: C^T[d][col] += E_tile^T x P^T ----
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      s16x4 a;
      if constexpr (TR) {
        a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(sbase + a2_off + dt * 32));
      } else {
        const char* g = sbase + a2_off + dt * 32;
        a[0] = *reinterpret_cast<const short*>(g);
        a[1] = *reinterpret_cast<const short*>(g + C::ROWB);
        a[2] = *reinterpret_cast<const short*>(g + 2 * C::ROWB);
        a[3] = *reinterpret_cast<const short*>(g + 3 * C::ROWB);
      }
      acc[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, pfrag, acc[dt], 0, 0, 0);
    }
    // the slot is free once every read of it has returned: fetch the tile NSLOT rounds ahead into it
    if (tile + 4 * C::NSLOT < n_tiles) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_tile(tile + 4 * C::NSLOT, slot);
    }
    slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
  }

  // ---- merge the four waves ----
  const float l_wave = xa_rows_sum(l_part);   // all four key groups of the column
  __syncthreads();                             // every wave is done with its ring: the merge area overlays it
  float* mc = reinterpret_cast<float*>(smem);                 // [wave][head][D]
  float2* ml = reinterpret_cast<float2*>(smem + 4 * 8 * D * 4);   // [wave][head]
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt) {
    float4 v;
    v.x = acc[dt][0] + dpp_ror8(acc[dt][0]);
    v.y = acc[dt][1] + dpp_ror8(acc[dt][1]);
    v.z = acc[dt][2] + dpp_ror8(acc[dt][2]);
    v.w = acc[dt][3] + dpp_ror8(acc[dt][3]);
    if (li < 8) *reinterpret_cast<float4*>(mc + ((size_t)(wave * 8 + li) * D + dt * 16 + kg * 4)) = v;
  }
  if (lane < 8) ml[wave * 8 + lane] = make_float2(m_run, l_wave);
  __syncthreads();
  constexpr int CHUNKS = 8 * D / 8;   // 16-byte output chunks of the clip's row
  for (int ch = threadIdx.x; ch < CHUNKS; ch += 256) {
    const int h = ch / (D / 8), d0 = (ch - h * (D / 8)) * 8;
    const float2 e0 = ml[h], e1 = ml[8 + h], e2 = ml[16 + h], e3 = ml[24 + h];
    float m = fmaxf(fmaxf(e0.x, e1.x), fmaxf(e2.x, e3.x));
    m = m > -INFINITY ? m : 0.f;   // a clip without frames: every weight 0, output 0
    const float f0 = __builtin_amdgcn_exp2f(e0.x - m), f1 = __builtin_amdgcn_exp2f(e1.x - m);
    const float f2 = __builtin_amdgcn_exp2f(e2.x - m), f3 = __builtin_amdgcn_exp2f(e3.x - m);
    const float l = (f0 * e0.y + f1 * e1.y) + (f2 * e2.y + f3 * e3.y);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float o[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float4 c0 = *reinterpret_cast<const float4*>(mc + (size_t)(0 * 8 + h) * D + d0 + half * 4);
      const float4 c1 = *reinterpret_cast<const float4*>(mc + (size_t)(1 * 8 + h) * D + d0 + half * 4);
      const float4 c2 = *reinterpret_cast<const float4*>(mc + (size_t)(2 * 8 + h) * D + d0 + half * 4);
      const float4 c3 = *reinterpret_cast<const float4*>(mc + (size_t)(3 * 8 + h) * D + d0 + half * 4);
      o[half * 4 + 0] = ((f0 * c0.x + f1 * c1.x) + (f2 * c2.x + f3 * c3.x)) * inv;
      o[half * 4 + 1] = ((f0 * c0.y + f1 * c1.y) + (f2 * c2.y + f3 * c3.y)) * inv;
      o[half * 4 + 2] = ((f0 * c0.z + f1 * c1.z) + (f2 * c2.z + f3 * c3.z)) * inv;
      o[half * 4 + 3] = ((f0 * c0.w + f1 * c1.w) + (f2 * c2.w + f3 * c3.w)) * inv;
    }
    uint4 u;
    u.x = pack_bf16x2(o[0], o[1]);
    u.y = pack_bf16x2(o[2], o[3]);
    u.z = pack_bf16x2(o[4], o[5]);
    u.w = pack_bf16x2(o[6], o[7]);
    *reinterpret_cast<uint4*>(ctx + fm16(b, h * D + d0, 8 * D / 32)) = u;
  }
}

bool xattn_use_tr() {
  static const bool on = [] {
    const char* e = getenv("MSH_XATTN_TR");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}

template <int D>
void launch_absorbed(const float* qt, const bf16_t* enc, const ClipMeta* clips, int M, bf16_t* ctx, hipStream_t s) {
  using C = XaCfg<D>;
  static const bool attr = [] {
    MSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_cross_absorbed_kernel<D, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    MSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_cross_absorbed_kernel<D, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    return true;
  }();
  (void)attr;
  if (xattn_use_tr())
    MSH_LAUNCH((dec_cross_absorbed_kernel<D, true>), dim3(M), dim3(256), C::LDS, s, qt, enc, clips, ctx);
  else
    MSH_LAUNCH((dec_cross_absorbed_kernel<D, false>), dim3(M), dim3(256), C::LDS, s, qt, enc, clips, ctx);
}

}  // namespace

bool cross_absorbed_supported(int D, int heads) { return heads == 8 && (D == 416 || D == 288); }

void dec_cross_absorbed(const float* qt, const bf16_t* enc, const ClipMeta* clips, int M, int D, int heads, bf16_t* ctx,
                        hipStream_t s) {
  if (!cross_absorbed_supported(D, heads)) throw std::runtime_error("dec_cross_absorbed: unsupported shape");
  if (D == 416) launch_absorbed<416>(qt, enc, clips, M, ctx, s);
  else launch_absorbed<288>(qt, enc, clips, M, ctx, s);
}

// Test / microbenchmark hook (msh_test_cross_absorbed): M clips of Ts[b] frames, qt [M][8 D] fp32, enc rows as fp32
// (rounded to bf16 here; clip b's rows start at row_starts[b]); ctx_out [M][8 D] fp32 = the kernel's bf16 output.  Returns
// ms per launch over `iters` launches (0 = one launch, not timed).
float cross_absorbed_host(const float* qt, const float* enc_f32, long R, const int* Ts, const int* row_starts, int M, int D,
                          float* ctx_out, int iters) {
  if (!cross_absorbed_supported(D, 8)) throw std::runtime_error("cross_absorbed: unsupported width");
  std::vector<bf16_t> e16((size_t)R * D);
  for (size_t i = 0; i < e16.size(); ++i) e16[i] = f32_to_bf16(enc_f32[i]);
  std::vector<ClipMeta> cm(M);
  for (int b = 0; b < M; ++b) {
    memset(&cm[b], 0, sizeof(ClipMeta));
    cm[b].row_start = row_starts[b];
    cm[b].T = Ts[b];
  }
  const int M16 = (M + 15) / 16 * 16;
  float* dq = nullptr;
  bf16_t *de = nullptr, *dc = nullptr;
  ClipMeta* dm = nullptr;
  MSH_HIP(hipMalloc(&dq, (size_t)M * 8 * D * 4));
  MSH_HIP(hipMalloc(&de, e16.size() * 2));
  MSH_HIP(hipMalloc(&dc, (size_t)M16 * 8 * D * 2));
  MSH_HIP(hipMalloc(&dm, (size_t)M * sizeof(ClipMeta)));
  MSH_HIP(hipMemcpy(dq, qt, (size_t)M * 8 * D * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(de, e16.data(), e16.size() * 2, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(dm, cm.data(), (size_t)M * sizeof(ClipMeta), hipMemcpyHostToDevice));
  MSH_HIP(hipMemset(dc, 0, (size_t)M16 * 8 * D * 2));
  dec_cross_absorbed(dq, de, dm, M, D, 8, dc, 0);
  MSH_HIP(hipDeviceSynchronize());
  float ms = 0.f;
  if (iters > 0) {
    hipEvent_t a, b2;
    MSH_HIP(hipEventCreate(&a));
    MSH_HIP(hipEventCreate(&b2));
    MSH_HIP(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) dec_cross_absorbed(dq, de, dm, M, D, 8, dc, 0);
    MSH_HIP(hipEventRecord(b2, 0));
    MSH_HIP(hipEventSynchronize(b2));
    MSH_HIP(hipEventElapsedTime(&ms, a, b2));
    ms /= (float)iters;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b2);
  }
  std::vector<bf16_t> c16((size_t)M16 * 8 * D);
  MSH_HIP(hipMemcpy(c16.data(), dc, c16.size() * 2, hipMemcpyDeviceToHost));
  for (int b = 0; b < M; ++b)
    for (int k = 0; k < 8 * D; ++k) ctx_out[(size_t)b * 8 * D + k] = bf16_to_f32(c16[(size_t)fm16(b, k, 8 * D / 32)]);
  (void)hipFree(dq);
  (void)hipFree(de);
  (void)hipFree(dc);
  (void)hipFree(dm);
  return ms;
}

}  // namespace msh
