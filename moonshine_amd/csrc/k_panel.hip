// A-stationary "panel" GEMMs of the encoder for gfx950: C[R][N] = f(A[R][D]) * W[N][D]^T with D = the model width.
//
// The encoder's projections have a short K (D = 416: 13 k-slices of the tiled kernel) and a long M (106 k rows at
// 256 x 10 s): as tiled GEMMs every 128 x 128 tile pays a pipeline fill, an accumulator drain and a store tail per 13
// slices and ran at 0.10-0.13 of the MFMA peak (enc_qkv_rope_gemm, enc_oproj_gemm).  Here, as in k_mlp.hip, a workgroup
// owns a PANEL of 128 rows for the whole N:
//   * 4 waves x 32 rows, v_mfma_f32_32x32x16_bf16; the wave's 32 x D activation block is built ONCE (LayerNorm of the
//     fp32 residual stream in registers, or bf16 rows as they are) and stays in registers as the B operand;
//   * N is walked in chunks of 32 columns; the only operand that moves is W, packed at load into fragment order, one
//     chunk = D/16 KiB fetched by global_load_lds_dwordx4 into a 3-stage LDS ring, one workgroup barrier per chunk;
//   * the chunk's accumulator (row = lane, 16 columns per lane) is finished -- RoPE, bf16 pack, transposes -- beside the
//     MFMAs of the next chunk and stored behind that chunk's barrier;
//   * 78 KiB of LDS and < 256 registers: TWO workgroups per CU, so one's LayerNorm prologue / stores overlap the other's
//     MFMAs and the two waves per SIMD interleave their dependent accumulator chains.
// Instance so far: the encoder QKV projection (LayerNorm + q | k with RoPE row-major + V transposed).  The kernel is
// written against an epilogue interface (sections of D columns of two compile-time kinds) so that the other K = D
// projections can follow (cross-attention K / V of all decoder layers: its transposed store is the kind-1 path).
#include <stdlib.h>

#include <stdexcept>
#include <utility>
#include <vector>

#include "gemm_common.h"
#include "panel_rows.h"

namespace msh {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D>
struct PanelGeom {
  static constexpr int KS = D / 16;   // k-steps = W fragments (1 KiB each) per chunk
  static constexpr int CT = D / 32;   // chunks per D output columns ("section")
  static constexpr int NST = 3;       // ring stages: being read, published, being filled
  static constexpr int PMAX = (KS + 3) / 4;
  static_assert(D % 32 == 0, "width must be a multiple of 32");
  static_assert(NST * KS <= 80, "two workgroups' rings must fit the LDS");
};

__device__ __forceinline__ bf16x8 frag_of(const uint4& v) { return *reinterpret_cast<const bf16x8*>(&v); }

template <class Body, int... I>
__device__ __forceinline__ void sfor_impl(Body&& body, std::integer_sequence<int, I...>) {
  (body(std::integral_constant<int, I>{}), ...);
}
template <int N, class Body>
__device__ __forceinline__ void sfor(Body&& body) {
  sfor_impl(static_cast<Body&&>(body), std::make_integer_sequence<int, N>{});
}

// lanes l and l + 32 hold the two halves of a row: a[32..63] <-> b[0..31] (inline asm: see k_mlp.hip on the builtin)
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// ---------------------------------------------------------------------------------------------------------------
// Epilogue of the encoder QKV projection.  Sections of D columns: 0 = q, 1 = k (RoPE, row-major [R][2D] bf16), 2 = v
// (transposed: vt[c][row], the P.V operand of enc_attention).  A lane's row position is fixed for the whole panel: its
// cos / sin factors (RP pairs) are loaded once into registers; which pair a value needs is a compile-time function of
// (chunk in section, register) up to the lane half, which selects between two constants.
// NTS: q | k and V^T go out with non-temporal stores (see qkv_panel)
template <int D, int DH, int RP, bool NTS = false>
struct EpiQkvPanel {
  bf16_t* qk;         // [R][2D]
  bf16_t* vt;         // [D][vt_ld]
  long vt_ld;
  const int* row_pos;
  const float* cos_t;  // [pos][RP]
  const float* sin_t;
  struct Row {
    float cs[RP], sn[RP];
  };
  // the RoPE factors of the wave's 32 rows: fetched with lanes running along the table rows (23 consecutive floats per row)
  // into the wave's LDS scratch, then every lane picks up its row -- per-lane reads of the global table were 46 loads of
  // 64 different lines each
  __device__ void init(Row& r, int row0, int R, float* tab, int lane) const {
    constexpr int TW = 2 * RP + 1;
    int row = row0 + (lane & 31);
    row = row < R ? row : R - 1;
    int pos = row_pos[row];
    pos = pos < 0 ? 0 : pos;
    for (int idx = lane; idx < 32 * RP; idx += 64) {
      const int rr = idx / RP, j = idx - rr * RP;
      const int pr = __shfl(pos, rr, 64);
      tab[rr * TW + j] = cos_t[(long)pr * RP + j];
      tab[rr * TW + RP + j] = sin_t[(long)pr * RP + j];
    }
    __builtin_amdgcn_wave_barrier();
    const float* mine = tab + (lane & 31) * TW;
#pragma unroll
    for (int j = 0; j < RP; ++j) {
      r.cs[j] = mine[j];
      r.sn[j] = mine[RP + j];
    }
    __builtin_amdgcn_wave_barrier();
  }
  __device__ void init_dummy(Row& r, int lane) const {
#pragma unroll
    for (int j = 0; j < RP; ++j) {
      r.cs[j] = 0.5f + j + lane;
      r.sn[j] = 0.25f * j;
    }
  }
  // KIND 0 = a q / k section (sec = 0 / 1), KIND 1 = the v section.  Group Q = accumulator registers 4Q..4Q+3 = columns
  // 32 C + 8 Q + 4 hh + 0..3 of the section.  A group is finished in two halves (half A behind one MFMA, half B behind the
  // next: short VALU bursts keep the matrix pipe fed), the values in between live in x.
  template <int KIND, int C, int Q, int HALF>
  __device__ void compute(const f32x16& z, float (&x)[4], uint32_t (&pk)[8], const Row& r, int /*sec*/, int hh, int lane) const {
    if constexpr (HALF == 0) {
      x[0] = z[4 * Q];
      x[1] = z[4 * Q + 1];
      x[2] = z[4 * Q + 2];
      x[3] = z[4 * Q + 3];
    }
    if constexpr (KIND == 0) {
      constexpr int d0 = (32 * C + 8 * Q) % DH, d1 = (32 * C + 8 * Q + 4) % DH;   // head-dim offset for hh = 0 / 1
      constexpr int p = HALF, j0 = d0 / 2 + p, j1 = d1 / 2 + p;
      float c0 = 1.f, s0 = 0.f, c1 = 1.f, s1 = 0.f;   // pairs beyond the rotary part pass through
      if constexpr (j0 < RP) {
        c0 = r.cs[j0];
        s0 = r.sn[j0];
      }
      if constexpr (j1 < RP) {
        c1 = r.cs[j1];
        s1 = r.sn[j1];
      }
      if constexpr (j0 < RP || j1 < RP) {
        const float c = hh ? c1 : c0, s = hh ? s1 : s0;
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        x[2 * p] = x0 * c - x1 * s;
        x[2 * p + 1] = x1 * c + x0 * s;
      }
      pk[2 * Q + p] = pack_bf16x2(x[2 * p], x[2 * p + 1]);
    } else {
      // 4 x 4 transpose inside every quad of lanes (rows) in two exchange steps: afterwards lane r = lane & 3 holds column
      // 8 Q + 4 hh + r for the quad's 4 rows
      if constexpr (HALF == 0) {
        const bool o1 = (lane & 1) != 0;
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const float t = dpp_f<0xB1>(o1 ? x[i] : x[i + 1]);   // quad_perm [1, 0, 3, 2]
          if (o1) x[i] = t; else x[i + 1] = t;
        }
      } else {
        const bool o2 = (lane & 2) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float t = dpp_f<0x4E>(o2 ? x[i] : x[i + 2]);   // quad_perm [2, 3, 0, 1]
          if (o2) x[i] = t; else x[i + 2] = t;
        }
        pk[2 * Q] = pack_bf16x2(x[0], x[1]);
        pk[2 * Q + 1] = pack_bf16x2(x[2], x[3]);
      }
    }
  }
  // Stores are UNCONDITIONAL (the kernel counts them in its vmcnt waits): rows past R land in the buffers' padding -- qk
  // holds ceil(R / 128) * 128 rows and vt_ld >= that.
  static constexpr int kMinStores = 2;   // the fewest store instructions a chunk issues
  template <int KIND, int C>
  __device__ void store(uint32_t (&pk)[8], const Row&, int sec, int row, int hh, int lane) const {
    if constexpr (KIND == 0) {
      // swap32(a = group 2p, b = group 2p + 1): lane hh = 1's a (columns 16p + 4..7) <-> lane hh = 0's b (16p + 8..11);
      // afterwards hh = 0 holds columns 16p + 0..7 and hh = 1 columns 16p + 8..15: 16-byte stores
      swap32(pk[0], pk[2]);
      swap32(pk[1], pk[3]);
      swap32(pk[4], pk[6]);
      swap32(pk[5], pk[7]);
      bf16_t* o = qk + (long)row * (2 * D) + sec * D + 32 * C + 8 * hh;
      typedef unsigned int u32x4_nat __attribute__((ext_vector_type(4)));
      if constexpr (NTS) {
        __builtin_nontemporal_store(u32x4_nat{pk[0], pk[1], pk[2], pk[3]}, reinterpret_cast<u32x4_nat*>(o));
        __builtin_nontemporal_store(u32x4_nat{pk[4], pk[5], pk[6], pk[7]}, reinterpret_cast<u32x4_nat*>(o + 16));
      } else {
        *reinterpret_cast<uint4*>(o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(o + 16) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
    } else {
      const int r0 = row & ~3;   // the quad's first row
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * C + 8 * q + 4 * hh + (lane & 3);
        typedef unsigned int u32x2_nat __attribute__((ext_vector_type(2)));
        if constexpr (NTS) __builtin_nontemporal_store(u32x2_nat{pk[2 * q], pk[2 * q + 1]}, reinterpret_cast<u32x2_nat*>(vt + (long)c * vt_ld + r0));
        else *reinterpret_cast<uint2*>(vt + (long)c * vt_ld + r0) = make_uint2(pk[2 * q], pk[2 * q + 1]);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Epilogue of the cross-attention K / V projection of ALL decoder layers: section sec = 2 * layer + (0 = K, 1 = V), every
// section stored TRANSPOSED into K^T / V^T [layer][clip][c][t] (keys contiguous: what the decode kernel streams), zeros
// for keys past the clip's frame count, nothing for rows past its padded key count.  After the quad transpose a lane
// holds 4 consecutive keys of one column: 8-byte stores (bf16), 4-byte stores (fp8: value * qscale[column], e4m3).
template <int D, bool FP8>
struct EpiCrossKvPanel {
  void* KT;
  void* VT;
  const int* row_clip;
  const ClipMeta* clips;
  long layer_stride;      // elements (= bytes for fp8) per layer: D * total keys
  const float* qscale;    // [L * 2 * D], fp8 only
  static constexpr int kMinStores = 0;   // stores are conditional: the mid-stage wait is a full vmcnt(0)
  struct Row {
    int t0, T, Tk;   // first key of the lane's quad within its clip, the clip's frames and padded key count
    long off;        // kv_start * D + t0
  };
  __device__ void init(Row& r, int row0, int R, float*, int lane) const {
    const int row = row0 + (lane & 31);
    const ClipMeta cm = clips[row_clip[row < R ? row : R - 1]];
    r.t0 = (row & ~3) - cm.row_start;   // (a row past R lands past the last clip's keys and is never stored)
    r.T = cm.T;
    r.Tk = cm.Tk;
    r.off = (long)cm.kv_start * D + r.t0;
  }
  __device__ void init_dummy(Row& r, int lane) const {
    r.t0 = lane & 28;
    r.T = 400;
    r.Tk = 0;
    r.off = 0;
  }
  template <int KIND, int C, int Q, int HALF>
  __device__ void compute(const f32x16& z, float (&x)[4], uint32_t (&pk)[8], const Row& r, int sec, int hh, int lane) const {
    if constexpr (HALF == 0) {
      x[0] = z[4 * Q];
      x[1] = z[4 * Q + 1];
      x[2] = z[4 * Q + 2];
      x[3] = z[4 * Q + 3];
      const bool o1 = (lane & 1) != 0;
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const float t = dpp_f<0xB1>(o1 ? x[i] : x[i + 1]);
        if (o1) x[i] = t; else x[i + 1] = t;
      }
    } else {
      const bool o2 = (lane & 2) != 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float t = dpp_f<0x4E>(o2 ? x[i] : x[i + 2]);
        if (o2) x[i] = t; else x[i + 2] = t;
      }
      // the lane now holds keys t0 .. t0 + 3 of column 32 C + 8 Q + 4 hh + (lane & 3); padding keys are exact zeros
      if constexpr (!FP8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = r.t0 + k < r.T ? x[k] : 0.f;
        pk[2 * Q] = pack_bf16x2(x[0], x[1]);
        pk[2 * Q + 1] = pack_bf16x2(x[2], x[3]);
      } else {
        const float qs = qscale[sec * D + 32 * C + 8 * Q + 4 * hh + (lane & 3)];
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = r.t0 + k < r.T ? fminf(fmaxf(x[k] * qs, -448.f), 448.f) : 0.f;   // (the cvt does not saturate)
        int p = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], 0, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], p, true);
        pk[2 * Q] = (uint32_t)p;
        pk[2 * Q + 1] = 0u;
      }
    }
  }
  template <int KIND, int C>
  __device__ void store(uint32_t (&pk)[8], const Row& r, int sec, int row, int hh, int lane) const {
    (void)row;
    if (r.t0 >= r.Tk) return;
    const int layer = sec >> 1, which = sec & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 32 * C + 8 * q + 4 * hh + (lane & 3);
      const long at = layer * layer_stride + r.off + (long)c * r.Tk;
      if constexpr (!FP8) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(which ? VT : KT) + at) = make_uint2(pk[2 * q], pk[2 * q + 1]);
      else *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(which ? VT : KT) + at) = pk[2 * q];
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// rows [128 blockIdx.x, +128) of A; Wp: NC chunks of KS KiB (pack_panel_weights).  AM = how the activation block arrives:
// 1 = A is the fp32 residual stream and LayerNorm (no bias, eps 1e-5; gamma folded into W) is computed here; 0 = A is bf16
// [R][D]; 2 (round 6) = A is bf16 in FRAGMENT-MAJOR order, already normalised by the kernel that produced the rows
// (mlp_fused_kernel YOUT): the B operand is KS coalesced 1-KiB loads, no LDS staging, no arithmetic.
// ABL (microbenchmark ablations, MSH_PANEL_ABL; results are garbage): 1 = no stores, 2 = no finish arithmetic, 4 = no DMA
// after the first two stages, 8 = no activation loads / LayerNorm, 16 = no per-row epilogue context (RoPE factors)
template <int D, int AM, class Epi, int ABL = 0>
__global__ __launch_bounds__(256, 2) void panel_gemm_kernel(const void* __restrict__ Aptr, const bf16_t* __restrict__ Wp,
                                                            Epi epi, int R, int n0, int n1) {
  using G = PanelGeom<D>;
  constexpr int KS = G::KS, CT = G::CT, NST = G::NST, PMAX = G::PMAX, PF = 4, MID = KS / 2;
  __shared__ __attribute__((aligned(16))) uint4 lds[NST * KS * 64];

  const int tid = threadIdx.x, lane = tid & 63, mrow = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int my_pieces = (KS - wave + 3) / 4;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_offset_of(&lds[0]));
  const int nsec = n0 + n1, nstages = nsec * CT;   // n0 sections of kind 0, then n1 of kind 1 (see the epilogue struct)

  const bf16_t* wsrc = Wp + (long)wave * 512 + lane * 8;
  auto stage_src = [&](int st) { return wsrc + (long)(st < nstages ? st : nstages - 1) * (KS * 512); };
  auto stage_dst = [&](int buf) { return lds_base + (unsigned)(buf * KS + wave) * 1024u; };
  auto issue_piece = [&](const bf16_t* src, unsigned dst, int q) {
    if (q < PMAX - 1 || my_pieces == PMAX) dma16(src + (long)q * 2048, dst + (unsigned)q * 4096u);
  };
  // ---- the wave's 32 x D activation block as the MFMA B operand: lane (mrow, hh) holds columns 16 s + 8 hh + 0..7 ----
  const int row = blockIdx.x * 128 + wave * 32 + mrow;
  const int lrow = row < R ? row : R - 1;
  bf16x8 yf[KS];
  if constexpr ((ABL & 8) != 0) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 p = make_uint4(0x3f803f80u + lane, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u + s);
      yf[s] = frag_of(p);
    }
  } else if constexpr (AM == 2) {
    const uint4* yp = reinterpret_cast<const uint4*>(Aptr) + ((long)(blockIdx.x * 4 + wave) * KS) * 64 + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) yf[s] = frag_of(yp[s * 64]);
  } else if constexpr (AM == 1) {
    // the wave's rows through LDS (panel_rows.h), twice: moments (shifted by the lane's first value, the two half rows merged
    // with Chan's formula: no cancellation whatever the row's mean), then -- the block comes from L2 now -- the operand
    using RV = RowsViaLds<D, (D >= 416 ? 4 : 3)>;
    static_assert(4 * RV::BYTES <= (int)sizeof(lds), "the staging regions live in the (still empty) weight ring");
    const unsigned roff = lds_base + (unsigned)(wave * RV::BYTES);
    const unsigned char* rptr = reinterpret_cast<const unsigned char*>(lds) + wave * RV::BYTES;
    const float* Hf = reinterpret_cast<const float*>(Aptr);
    const int row0w = blockIdx.x * 128 + wave * 32;
    float k0 = 0.f, s1 = 0.f, s2 = 0.f;
    RV::run(Hf, row0w, R, roff, rptr, lane, [&](auto sc, const float4 a, const float4 b) {
      if constexpr (decltype(sc)::value == 0) k0 = a.x;
      const float d0 = a.x - k0, d1 = a.y - k0, d2 = a.z - k0, d3 = a.w - k0, d4 = b.x - k0, d5 = b.y - k0, d6 = b.z - k0, d7 = b.w - k0;
      s1 += (d0 + d1) + (d2 + d3) + (d4 + d5) + (d6 + d7);
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) + (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
    });
    constexpr float kHalf = D / 2;
    const float mean_l = k0 + s1 * (1.0f / kHalf), m2_l = s2 - s1 * s1 * (1.0f / kHalf);
    const float mean_o = __shfl_xor(mean_l, 32, 64), m2_o = __shfl_xor(m2_l, 32, 64);
    const float mean = 0.5f * (mean_l + mean_o), dm = mean_l - mean_o;
    const float var = (m2_l + m2_o + dm * dm * (0.5f * kHalf)) * (1.0f / D);
    const float rstd = rsqrtf(var + 1e-5f);
    RV::run(Hf, row0w, R, roff, rptr, lane, [&](auto sc, const float4 xa, const float4 xb) {
      uint4 p;
      p.x = pack_bf16x2((xa.x - mean) * rstd, (xa.y - mean) * rstd);
      p.y = pack_bf16x2((xa.z - mean) * rstd, (xa.w - mean) * rstd);
      p.z = pack_bf16x2((xb.x - mean) * rstd, (xb.y - mean) * rstd);
      p.w = pack_bf16x2((xb.z - mean) * rstd, (xb.w - mean) * rstd);
      yf[decltype(sc)::value] = frag_of(p);
    });
  } else {
    const bf16_t* ap = reinterpret_cast<const bf16_t*>(Aptr) + (long)lrow * D + hh * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 p = *reinterpret_cast<const uint4*>(ap + s * 16);
      yf[s] = frag_of(p);
    }
  }
  typename Epi::Row rctx;
  if constexpr ((ABL & 16) == 0) {
    epi.init(rctx, blockIdx.x * 128 + wave * 32, R, reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(lds) + wave * (int)(sizeof(lds) / 4)), lane);
  } else {
    epi.init_dummy(rctx, lane);
  }
  __syncthreads();   // every wave is done with its staging region: the weight ring may be filled
#pragma unroll
  for (int s = 0; s < 2; ++s) {   // stages 0 and 1
    const bf16_t* src = stage_src(s);
    const unsigned dst = stage_dst(s);
#pragma unroll
    for (int q = 0; q < PMAX; ++q) issue_piece(src, dst, q);
  }

  uint4 fr[PF];
  f32x16 za, zp;        // accumulator of the chunk being multiplied / of the finished chunk being stored
  uint32_t pk[8];
#pragma unroll
  for (int r = 0; r < 16; ++r) zp[r] = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) pk[r] = 0u;

  // One stage = the KS MFMAs of chunk j.  Beside them: the fragment ring (PF reads ahead, across stages), the finish of
  // chunk j - 1 (VALU, first half), ONE barrier in the middle (publishes stage j + 1, retires stage j - 1), then chunk
  // j - 1's stores and the DMAs of stage j + 2 into the retired buffer.  Order pinned as in k_mlp.hip.
  // kp_c: kind of the PREVIOUS chunk (-1: there is none); sec_p: its section.  first_stores: this is the first stage whose
  // predecessor issued stores (only then may the mid-stage wait leave stores in flight).
  float xs[4];
  auto stage = [&](int j, int buf, auto c_c, auto kp_c, int sec_p, bool prev_stored) {
    constexpr int C = decltype(c_c)::value, CP = (C + CT - 1) % CT, KP = decltype(kp_c)::value;
    const int nbuf = buf + 1 == NST ? 0 : buf + 1;
    const bf16_t* nsrc = stage_src(j + 2);
    const unsigned ndst = stage_dst(buf == 0 ? NST - 1 : buf - 1);
    const uint4* st = lds + buf * (KS * 64) + lane;
    const uint4* stn = lds + nbuf * (KS * 64) + lane;
    constexpr int RO = (C * KS) % PF;   // ring slot of this stage's first step (the ring restarts with every section)
    constexpr int DS = (KS - MID - 4) / PMAX > 0 ? (KS - MID - 4) / PMAX : 1;
    constexpr int STORE_AT = KS - 2;
    static_assert(MID >= 9, "the eight finish half-groups sit behind steps 1..8 of the first half");
    static_assert(MID + 1 + (PMAX - 1) * DS < STORE_AT, "the stores go behind the last DMA");
    sfor<KS>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      if constexpr (f == 0) {
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        za = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(fr[(RO + f) % PF]), yf[f], zero, 0, 0, 0);
      } else {
        za = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(fr[(RO + f) % PF]), yf[f], za, 0, 0, 0);
      }
      // finish of the previous chunk (VALU only): half (f - 1) % 2 of group (f - 1) / 2 behind step f = 1..8
      if constexpr ((ABL & 2) == 0 && KP >= 0 && f >= 1 && f <= 8) {
        constexpr int Q = (f - 1) / 2, HALF = (f - 1) % 2;
        epi.template compute<KP, CP, Q, HALF>(zp, xs, pk, rctx, sec_p, hh, lane);
        if constexpr (HALF == 0) asm volatile("" : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xs[3]));
        else asm volatile("" : "+v"(pk[2 * Q]), "+v"(pk[2 * Q + 1]));
      }
      if constexpr (f == MID) {
        // this wave's pieces of stage j + 1 were requested BEFORE the previous stage's stores: loads and stores retire in
        // order, so the pieces are in once at most that chunk's stores are outstanding -- the stores themselves (hundreds
        // of cycles of HBM write latency under load) are never waited for
        if constexpr ((ABL & 32) != 0) {
          wait_vmcnt<0>();
        } else {
          if (prev_stored) wait_vmcnt<Epi::kMinStores>();
          else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
      }
      if constexpr ((ABL & 4) == 0 && f > MID) {
        sfor<PMAX>([&](auto qc) {
          constexpr int q = decltype(qc)::value, at = MID + 1 + q * DS;
          if constexpr (at == f) issue_piece(nsrc, ndst, q);
        });
      }
      if constexpr ((ABL & 1) == 0 && KP >= 0 && f == STORE_AT) epi.template store<KP, CP>(pk, rctx, sec_p, row, hh, lane);
      if constexpr (f + PF < KS) fr[(RO + f) % PF] = st[(f + PF) * 64];
      else if constexpr (C + 1 < CT) fr[(RO + f) % PF] = stn[(f + PF - KS) * 64];   // next stage's first fragments (published at MID)
      __builtin_amdgcn_sched_barrier(0);
    });
    zp = za;
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using KN = std::integral_constant<int, -1>;
  int buf = 0, j = 0;
  auto ring_restart = [&] {   // the fragment ring restarts with every section: stage j's buffer is published by now
    const uint4* st = lds + buf * (KS * 64) + lane;
#pragma unroll
    for (int p = 0; p < PF; ++p) fr[p] = st[p * 64];
    __builtin_amdgcn_sched_barrier(0);
  };
  auto advance = [&] {
    buf = buf + 1 == NST ? 0 : buf + 1;
    ++j;
  };
  // a section's chunks 1 .. CT - 1 (their previous chunk is in the same section)
  auto rest_of_section = [&](auto kind_c, int sec) {
    sfor<CT - 1>([&](auto cc) {
      constexpr int C = decltype(cc)::value + 1;
      // stage j - 1 issued stores unless it was the very first stage
      stage(j, buf, std::integral_constant<int, C>{}, kind_c, sec, C > 1 || sec > 0);
      advance();
    });
  };

  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();   // stages 0 and 1 are in the ring
  for (int sec = 0; sec < n0; ++sec) {
    ring_restart();
    if (sec == 0) stage(j, buf, K0{}, KN{}, 0, false);
    else stage(j, buf, K0{}, K0{}, sec - 1, true);
    advance();
    rest_of_section(K0{}, sec);
  }
  for (int sec = n0; sec < nsec; ++sec) {
    ring_restart();
    if (sec == 0) stage(j, buf, K0{}, KN{}, 0, false);
    else if (sec == n0) stage(j, buf, K0{}, K0{}, sec - 1, true);
    else stage(j, buf, K0{}, K1{}, sec - 1, true);
    advance();
    rest_of_section(K1{}, sec);
  }
  wait_vmcnt<0>();   // trailing re-fetches must not outlive the workgroup's LDS
  // the last chunk
  if (n1 > 0) {
    sfor<8>([&](auto uc) { epi.template compute<1, CT - 1, decltype(uc)::value / 2, decltype(uc)::value % 2>(zp, xs, pk, rctx, nsec - 1, hh, lane); });
    epi.template store<1, CT - 1>(pk, rctx, nsec - 1, row, hh, lane);
  } else {
    sfor<8>([&](auto uc) { epi.template compute<0, CT - 1, decltype(uc)::value / 2, decltype(uc)::value % 2>(zp, xs, pk, rctx, nsec - 1, hh, lane); });
    epi.template store<0, CT - 1>(pk, rctx, nsec - 1, row, hh, lane);
  }
}

template <int D, int DH, int RP>
void launch_qkv_panel(const float* H, const bf16_t* Wp, int R, const int* row_pos, RopeParams rp, bf16_t* qk, bf16_t* vt,
                      long vt_ld, hipStream_t s, bool store_nt) {
  if (rp.rot_pairs != RP || rp.head_dim != DH) throw std::runtime_error("qkv_panel: rotary layout not compiled");
  if constexpr (D == 416) {
    if (store_nt) {
      EpiQkvPanel<D, DH, RP, true> e2{qk, vt, vt_ld, row_pos, rp.cos, rp.sin};
      MSH_LAUNCH((panel_gemm_kernel<D, 1, EpiQkvPanel<D, DH, RP, true>>), dim3((R + 127) / 128), dim3(256), 0, s, H, Wp, e2, R, 2, 1);
      return;
    }
  }
  EpiQkvPanel<D, DH, RP> epi{qk, vt, vt_ld, row_pos, rp.cos, rp.sin};
  static const int abl = [] {
    const char* e = dev_getenv("MSH_PANEL_ABL");
    return e ? atoi(e) : 0;
  }();
  using E = EpiQkvPanel<D, DH, RP>;
  const dim3 grid((R + 127) / 128);
  if constexpr (D == 416) {
    switch (abl) {
      case 0: break;
#define MSH_PABL(A) case A: MSH_LAUNCH((panel_gemm_kernel<D, 1, E, A>), grid, dim3(256), 0, s, H, Wp, epi, R, 2, 1); return;
      MSH_PABL(1) MSH_PABL(2) MSH_PABL(3) MSH_PABL(4) MSH_PABL(8) MSH_PABL(16) MSH_PABL(32)
#undef MSH_PABL
      default: throw std::runtime_error("qkv_panel: ablation not compiled");
    }
  }
  MSH_LAUNCH((panel_gemm_kernel<D, 1, E>), grid, dim3(256), 0, s, H, Wp, epi, R, 2, 1);
}

template <int D, int DH, int RP>
void launch_qkv_panel_prenorm(const bf16_t* Yfm, const bf16_t* Wp, int R, const int* row_pos, RopeParams rp, bf16_t* qk, bf16_t* vt,
                              long vt_ld, hipStream_t s, bool store_nt) {
  if (rp.rot_pairs != RP || rp.head_dim != DH) throw std::runtime_error("qkv_panel: rotary layout not compiled");
  const dim3 grid((R + 127) / 128);
  if constexpr (D == 416) {
    if (store_nt) {
      EpiQkvPanel<D, DH, RP, true> e2{qk, vt, vt_ld, row_pos, rp.cos, rp.sin};
      MSH_LAUNCH((panel_gemm_kernel<D, 2, EpiQkvPanel<D, DH, RP, true>>), grid, dim3(256), 0, s, Yfm, Wp, e2, R, 2, 1);
      return;
    }
  }
  EpiQkvPanel<D, DH, RP> epi{qk, vt, vt_ld, row_pos, rp.cos, rp.sin};
  MSH_LAUNCH((panel_gemm_kernel<D, 2, EpiQkvPanel<D, DH, RP>>), grid, dim3(256), 0, s, Yfm, Wp, epi, R, 2, 1);
}

template <int D, bool FP8>
void launch_cross_kv_panel(const bf16_t* A, const bf16_t* Wp, int R, int L, const int* row_clip, const ClipMeta* clips,
                           long layer_stride, const float* qscale, void* KT, void* VT, hipStream_t s) {
  using E = EpiCrossKvPanel<D, FP8>;
  E epi{KT, VT, row_clip, clips, layer_stride, qscale};
  MSH_LAUNCH((panel_gemm_kernel<D, 0, E>), dim3((R + 127) / 128), dim3(256), 0, s, A, Wp, epi, R, 0, 2 * L);
}

}  // namespace

bool cross_kv_panel_supported(int D) { return D == 416 || D == 288; }

// K^T / V^T of all L decoder layers from the encoder output A [R][D] bf16; Wp = pack_panel_weights of the fused
// [L * 2 * D][D] weight; qscale non-null = e4m3 output (layer_stride then in bytes)
void cross_kv_panel(const bf16_t* A, const bf16_t* Wp, int R, int D, int L, const int* row_clip, const ClipMeta* clips,
                    long layer_stride, const float* qscale, void* KT, void* VT, hipStream_t s) {
  if (R <= 0) return;
  const bool fp8 = qscale != nullptr;
  switch (D) {
    case 416:
      return fp8 ? launch_cross_kv_panel<416, true>(A, Wp, R, L, row_clip, clips, layer_stride, qscale, KT, VT, s)
                 : launch_cross_kv_panel<416, false>(A, Wp, R, L, row_clip, clips, layer_stride, qscale, KT, VT, s);
    case 288:
      return fp8 ? launch_cross_kv_panel<288, true>(A, Wp, R, L, row_clip, clips, layer_stride, qscale, KT, VT, s)
                 : launch_cross_kv_panel<288, false>(A, Wp, R, L, row_clip, clips, layer_stride, qscale, KT, VT, s);
    default: throw std::runtime_error("cross_kv_panel: unsupported width");
  }
}

bool qkv_panel_supported(int D, int head_dim, int rot_pairs) {
  return (D == 416 && head_dim == 52 && rot_pairs == 23) || (D == 288 && head_dim == 36 && rot_pairs == 16);
}

size_t panel_packed_elems(int N, int D) { return (size_t)(N / 32) * (D / 16) * 512; }

// w [N][D] (row n = output column), gamma nullable [D] -> N/32 chunks of D/16 fragments: fragment s of chunk j, lane l,
// element e = w[32 j + (l & 31)][16 s + 8 (l >> 5) + e] * gamma
void pack_panel_weights(const float* w, const float* gamma, int N, int D, bf16_t* out) {
  if ((N & 31) != 0 || (D & 31) != 0) throw std::runtime_error("pack_panel_weights: N and D must be multiples of 32");
  const int KS = D / 16, NC = N / 32;
  for (int j = 0; j < NC; ++j)
    for (int s = 0; s < KS; ++s)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) {
          const int n = 32 * j + (l & 31), k = 16 * s + 8 * (l >> 5) + e;
          out[(((size_t)j * KS + s) * 64 + l) * 8 + e] = f32_to_bf16(w[(size_t)n * D + k] * (gamma != nullptr ? gamma[k] : 1.0f));
        }
}

void qkv_panel(const float* H, const bf16_t* Wp, int R, int D, const int* row_pos, RopeParams rp, bf16_t* qk, bf16_t* vt,
               long vt_ld, hipStream_t s, bool store_nt) {
  if (R <= 0) return;
  if ((R & 7) != 0) throw std::runtime_error("qkv_panel: the row count must be a multiple of 8");
  switch (D) {
    case 416: return launch_qkv_panel<416, 52, 23>(H, Wp, R, row_pos, rp, qk, vt, vt_ld, s, store_nt);
    case 288: return launch_qkv_panel<288, 36, 16>(H, Wp, R, row_pos, rp, qk, vt, vt_ld, s, false);
    default: throw std::runtime_error("qkv_panel: unsupported width");
  }
}

// The same projection on rows that arrive normalised and in fragment-major order (mlp_fused_oproj's yfm output)
void qkv_panel_prenorm(const bf16_t* Yfm, const bf16_t* Wp, int R, int D, const int* row_pos, RopeParams rp, bf16_t* qk, bf16_t* vt,
                       long vt_ld, hipStream_t s, bool store_nt) {
  if (R <= 0) return;
  if ((R & 7) != 0) throw std::runtime_error("qkv_panel: the row count must be a multiple of 8");
  switch (D) {
    case 416: return launch_qkv_panel_prenorm<416, 52, 23>(Yfm, Wp, R, row_pos, rp, qk, vt, vt_ld, s, store_nt);
    case 288: return launch_qkv_panel_prenorm<288, 36, 16>(Yfm, Wp, R, row_pos, rp, qk, vt, vt_ld, s, false);
    default: throw std::runtime_error("qkv_panel: unsupported width");
  }
}

// Microbenchmark / test hook: random data, R rows; returns ms per launch.  When out_qk / out_vt are non-null the result of
// the last launch is copied out ([R][2D] and [D][R] bf16 bit patterns) together with the inputs used (h [R][D] fp32,
// w [3D][D] fp32 with gamma already folded in, pos [R]).
float qkv_panel_microbench(int R, int D, int iters, uint16_t* out_qk, uint16_t* out_vt, float* out_h, float* out_w, int* out_pos) {
  const int DH = D == 416 ? 52 : 36, RP = D == 416 ? 23 : 16, MAXPOS = 4096;
  if (!qkv_panel_supported(D, DH, RP) || (R & 7) != 0) throw std::runtime_error("qkv_panel_microbench: unsupported shape");
  std::vector<float> w((size_t)3 * D * D), h((size_t)R * D), cs((size_t)MAXPOS * RP), sn((size_t)MAXPOS * RP);
  std::vector<int> pos(R);
  unsigned x = 4242u;
  auto rnd = [&] {
    x = x * 1664525u + 1013904223u;
    return (float)((x >> 8) & 0xffff) / 32768.0f - 1.0f;
  };
  for (auto& v : w) v = rnd() * 0.05f;
  for (auto& v : h) v = rnd() * 2.0f + 0.3f;
  for (int r = 0; r < R; ++r) pos[r] = (r % 416 == 415) ? -1 : r % 416;
  for (int p = 0; p < MAXPOS; ++p)
    for (int j = 0; j < RP; ++j) {
      const float a = (float)p / powf(10000.f, (float)(2 * j) / (float)(2 * RP));
      cs[(size_t)p * RP + j] = cosf(a);
      sn[(size_t)p * RP + j] = sinf(a);
    }
  std::vector<bf16_t> packed(panel_packed_elems(3 * D, D));
  pack_panel_weights(w.data(), nullptr, 3 * D, D, packed.data());
  float *Hd = nullptr, *Cd = nullptr, *Sd = nullptr;
  bf16_t *Wd = nullptr, *QK = nullptr, *VT = nullptr;
  int* Pd = nullptr;
  MSH_HIP(hipMalloc(&Hd, h.size() * 4));
  MSH_HIP(hipMalloc(&Cd, cs.size() * 4));
  MSH_HIP(hipMalloc(&Sd, sn.size() * 4));
  MSH_HIP(hipMalloc(&Wd, packed.size() * 2));
  const long Rp = (R + 127) / 128 * 128;   // rows past R are stored too (into the padding)
  MSH_HIP(hipMalloc(&QK, (size_t)Rp * 2 * D * 2));
  MSH_HIP(hipMalloc(&VT, (size_t)D * Rp * 2));
  MSH_HIP(hipMalloc(&Pd, (size_t)R * 4));
  MSH_HIP(hipMemcpy(Hd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(Cd, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(Sd, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(Wd, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(Pd, pos.data(), (size_t)R * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemset(QK, 0, (size_t)Rp * 2 * D * 2));
  MSH_HIP(hipMemset(VT, 0, (size_t)D * Rp * 2));
  const RopeParams rp{Cd, Sd, RP, DH, D};
  qkv_panel(Hd, Wd, R, D, Pd, rp, QK, VT, Rp, 0);
  MSH_HIP(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  MSH_HIP(hipEventCreate(&e0));
  MSH_HIP(hipEventCreate(&e1));
  MSH_HIP(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) qkv_panel(Hd, Wd, R, D, Pd, rp, QK, VT, Rp, 0);
  MSH_HIP(hipEventRecord(e1, 0));
  MSH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MSH_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (out_qk != nullptr) MSH_HIP(hipMemcpy(out_qk, QK, (size_t)R * 2 * D * 2, hipMemcpyDeviceToHost));
  if (out_vt != nullptr)
    MSH_HIP(hipMemcpy2D(out_vt, (size_t)R * 2, VT, (size_t)Rp * 2, (size_t)R * 2, D, hipMemcpyDeviceToHost));
  if (out_h != nullptr) memcpy(out_h, h.data(), h.size() * 4);
  if (out_w != nullptr) memcpy(out_w, w.data(), w.size() * 4);
  if (out_pos != nullptr) memcpy(out_pos, pos.data(), (size_t)R * 4);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(Hd);
  (void)hipFree(Cd);
  (void)hipFree(Sd);
  (void)hipFree(Wd);
  (void)hipFree(QK);
  (void)hipFree(VT);
  (void)hipFree(Pd);
  return iters > 0 ? ms / iters : 0.f;
}

}  // namespace msh
