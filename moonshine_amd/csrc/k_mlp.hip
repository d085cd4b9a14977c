// Encoder MLP block as ONE kernel for gfx950:   H += fc2( gelu( fc1( LayerNorm(H) ) + b1 ) ) + b2
// (modeling_moonshine.py:69-85 MoonshineEncoderMLP inside the pre-LN layer wiring :382-411; the ORT encoder graph the
// reference runs at core/moonshine-model.cpp:270-274).
//
// Why fused.  As two GEMMs the block writes the [R][F] intermediate (354 MB per layer at 256 x 10 s) and reads it back,
// and each GEMM pays its pipeline fill, accumulator drain and store tail every 13 k-slices (K = 416): fc1 ran at 0.17 and
// fc2 at 0.19 of the MFMA peak.  Here a workgroup owns a PANEL of 128 rows for the whole block:
//   * 4 waves x 32 rows, one wave per SIMD, v_mfma_f32_32x32x16_bf16 (a W fragment read from LDS feeds 32 rows: half the
//     LDS traffic per flop of the 16-row shape);
//   * LayerNorm of the wave's rows is computed in registers straight from the fp32 residual stream and stays there as
//     the 32 x D bf16 B-operand of fc1 for the whole kernel (gamma is folded into W1 at load);
//   * the hidden dimension F is walked in chunks of 32: fc1 gives Z^T[32 n][32 m] in the accumulator layout, bias rides
//     in as the accumulator's initial value, GELU runs on the accumulator registers, and the packed bf16 result IS the
//     B-operand of fc2 for that chunk -- the k-order of W2 inside each 32-chunk is permuted at load to the order the
//     accumulator layout hands the values over in, so the intermediate never touches LDS or HBM;
//   * fc2 accumulates all D output columns of the wave's rows in registers (D/32 tiles x 16 = 208 VGPRs at D = 416)
//     over the whole F: ONE epilogue per panel (bias + residual add into H) instead of one per GEMM tile;
//   * the only operand that moves is W: W1 and W2 are packed at load into the exact fragment order, chunk by chunk, so a
//     stage of the LDS ring is (D/8 + 1) contiguous KiB fetched by global_load_lds_dwordx4 (whole 128-B lines, no
//     per-row 64-B segments), three stages (159 KiB) deep, one workgroup barrier per chunk; 123 flop per ingested byte.
// fc2 of chunk j-1 and GELU of chunk j are independent and sit in one basic block: matrix and vector pipes overlap.
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
struct MlpGeom {
  static constexpr int KS = D / 16;         // fc1 k-steps = W1 fragments per chunk
  static constexpr int CT = D / 32;         // output column tiles of fc2
  static constexpr int WP = KS + 2 * CT;    // weight fragments (1 KiB each) per stage
  static constexpr int PIECES = WP + 1;     // + the bias piece (first 32 floats = b1 of the chunk)
  static constexpr int NST = 3;             // ring stages: being read, published, being filled (159 KiB at D = 416)
  static constexpr int PMAX = (PIECES + 3) / 4;             // pieces per wave per stage: a contiguous run of the stage
  static constexpr int NG = (PMAX + 3) / 4;                 // DMA groups of up to four pieces (one M0 / address setting each)
  static_assert(D % 32 == 0 && KS == 2 * CT, "hidden size must be a multiple of 32");
  static_assert(NST * PIECES <= 160, "ring does not fit the LDS");
};

__device__ __forceinline__ bf16x8 as_frag(const uint4& v) { return *reinterpret_cast<const bf16x8*>(&v); }

// compile-time loop: body(std::integral_constant<int, I>) for I in [0, N).  Every index derived from I is a constant in
// the FRONT END (if constexpr, fixed register-array slots) -- with `#pragma unroll` and a runtime-looking index the kernel
// below came out with its accumulators in scratch (1408 bytes per lane) whenever the GELU placement was not trivial.
template <class Body, int... I>
__device__ __forceinline__ void static_for_impl(Body&& body, std::integer_sequence<int, I...>) {
  (body(std::integral_constant<int, I>{}), ...);
}
template <int N, class Body>
__device__ __forceinline__ void static_for(Body&& body) {
  static_for_impl(static_cast<Body&&>(body), std::make_integer_sequence<int, N>{});
}

// rows [128 * blockIdx.x, +128) of H [R][D] fp32, in place.  Wp: (NC + 1) stages x PIECES KiB, see pack_mlp_weights.
// ABL (microbenchmark ablations, 0 in the product; results are garbage unless noted): bit 0 = no DMA / no vmcnt waits after
// the prologue (compute side alone), bit 1 = no GELU arithmetic, bit 2 = fc1 alternating between TWO accumulators instead
// of one chain (correct), bit 3 = ring of 2 fragment registers instead of 4 (correct), bit 4 = a stage's DMAs issued
// together behind the barrier instead of spread over the MFMAs that follow it (correct), bit 5 = no stages at all
// (LayerNorm prologue + residual epilogue only), bit 6 = no fragment reads, bit 7 = no workgroup barrier, bit 8 = no MFMAs
// (with bits 1 and 6: the weight stream alone -- DMA issue, the waits and the barrier: what the CU INGESTS per stage).
// OP: the attention output projection of the layer rides in front: H' = H + AO Wo^T is formed in the fc2 accumulators
// (NOP extra stages of two 32-column tiles each at the head of Wp, AO [R][D] bf16 as their B operand), LayerNorm is then
// taken from those registers, and H is read ONCE and written once per layer for o-proj + MLP together.
// NTS: the residual stream is written back with non-temporal stores (see mlp_fused_oproj)
// YOUT (round 6): the epilogue also hands the NEXT layer's QKV projection its operand: LayerNorm (no scale: gamma is folded
// into that projection's weights) of the rows it just produced, taken from the accumulators, as bf16 in the fragment-major
// order panel_gemm_kernel consumes (Yfm: per 32-row wave block, D/16 k-steps of 64 lanes x 8 values = one contiguous KiB per
// wave-level load).  The QKV panel kernel then starts with D/16 coalesced loads instead of fetching the fp32 rows twice
// through LDS and normalising them again: a third of its time (profiles/rd5_final_qkv_panel_ablations.txt, ablation 8).
template <int D, int ABL = 0, bool OP = false, bool NTS = false, bool YOUT = false>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(float* __restrict__ H, const bf16_t* __restrict__ Wp,
                                                           const float* __restrict__ b2, int R, int NC,
                                                           const bf16_t* __restrict__ AO, bf16_t* __restrict__ Yfm) {
  using G = MlpGeom<D>;
  constexpr int KS = G::KS, CT = G::CT, WP = G::WP, PIECES = G::PIECES, NST = G::NST, PMAX = G::PMAX, NG = G::NG;
  constexpr int NOP = OP ? (CT + 1) / 2 : 0;   // o-proj stages
  __shared__ __attribute__((aligned(16))) uint4 lds[NST * PIECES * 64];

  const int tid = threadIdx.x, lane = tid & 63, mrow = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave w moves PMAX consecutive pieces of a stage from piece min(w PMAX, PIECES - PMAX) on (the last wave's run overlaps its
  // neighbour's by 4 PMAX - PIECES pieces: the same bytes to the same place, harmless, and every wave issues the same static
  // pattern): consecutive KiB on both sides, so four of them go out behind ONE setting of M0 and of the address register
  // (dma16_run).  The first version dealt the pieces round-robin (piece q of wave w = 4 q + w, 4 KiB apart: beyond the offset
  // field) and paid 7 instructions per piece, ~98 of a stage's 472.
  const int my_first = wave * PMAX < PIECES - PMAX ? wave * PMAX : PIECES - PMAX;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_offset_of(&lds[0]));
  const int nstages = NOP + NC + 1;

  // group g of this wave for stage `st` into ring buffer `buf` (a stage past the end re-fetches the last one into a
  // buffer nobody reads any more: the issue / wait pattern is the same for every stage)
  const bf16_t* wsrc = Wp + (long)my_first * 512 + lane * 8;
  auto stage_src = [&](int st) { return wsrc + (long)(st < nstages ? st : nstages - 1) * (PIECES * 512); };
  auto stage_dst = [&](int buf) { return lds_base + (unsigned)(buf * PIECES + my_first) * 1024u; };
  auto issue_group = [&](const bf16_t* src, unsigned dst, auto gc) {
    constexpr int g = decltype(gc)::value, n = PMAX - 4 * g >= 4 ? 4 : PMAX - 4 * g;
    static_assert(n >= 1, "group beyond the wave's run");
    dma16_run<n>(src + (long)g * 2048, dst + (unsigned)g * 4096u);
  };
#pragma unroll
  for (int s = 0; s < 2; ++s) {   // stages 0 and 1 up front
    const bf16_t* src = stage_src(s);
    const unsigned dst = stage_dst(s);
    static_for<NG>([&](auto gc) { issue_group(src, dst, gc); });
  }

  // ---- LayerNorm of this lane's row straight from the residual stream, kept as the fc1 B-operand ----
  // lane (mrow, hh) holds columns s*16 + hh*8 + 0..7 of row m0 + mrow for every k-step s: half a row
  const int row = blockIdx.x * 128 + wave * 32 + mrow;
  bf16x8 yf[KS];
  f32x16 oacc[CT];   // fc2 accumulators, started from the residual itself (below): H leaves HBM ONCE
  {
    // The wave's 32 rows come through LDS (panel_rows.h: fetched with lanes running along the rows, read back in the operand
    // pattern -- a lane reading its own row from global memory costs 64 cache-line lookups per load instruction), staged in
    // the ring's third buffer, which is empty until the first stage's barrier.
    constexpr int kPerWave = PIECES * 1024 / 4;   // bytes of the third ring buffer per wave
    constexpr int SL = kPerWave >= 17 * 1024 ? 4 : kPerWave >= 13 * 1024 ? 3 : kPerWave >= 9 * 1024 ? 2 : kPerWave >= 5 * 1024 ? 1 : 0;
    using RV = RowsViaLds<D, (SL > 0 ? SL : 1)>;
    static_assert(SL == 0 || 4 * RV::BYTES <= PIECES * 1024, "the row staging must fit one ring buffer");
    const unsigned roff = lds_base + (unsigned)(2 * PIECES * 1024 + wave * RV::BYTES);
    const unsigned char* rptr = reinterpret_cast<const unsigned char*>(lds) + 2 * PIECES * 1024 + wave * RV::BYTES;
    const int row0w = blockIdx.x * 128 + wave * 32;
    auto for_rows = [&](auto&& f) {
      if constexpr (SL > 0) {
        RV::run(H, row0w, R, roff, rptr, lane, f);
      } else {   // (a ring buffer too small to stage a slice: the lane reads its row itself)
        const float* hp = H + (long)(row < R ? row : R - 1) * D + hh * 8;
        static_for<KS>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          f(sc, *reinterpret_cast<const float4*>(hp + s * 16), *reinterpret_cast<const float4*>(hp + s * 16 + 4));
        });
      }
    };
    if constexpr (OP) {
      // only the residual is taken here (one pass): LayerNorm follows the o-proj stages, from registers
      for_rows([&](auto sc, const float4 xa, const float4 xb) {
        constexpr int s = decltype(sc)::value;
        float a[4] = {xa.x, xa.y, xa.z, xa.w}, b[4] = {xb.x, xb.y, xb.z, xb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[e]), "+v"(b[e]));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          oacc[s >> 1][4 * (2 * (s & 1)) + e] = a[e];
          oacc[s >> 1][4 * (2 * (s & 1) + 1) + e] = b[e];
        }
      });
    } else {
    // Pass 1: the row's moments, nothing kept (half a row is 208 fp32 values per lane: keeping them next to the bf16
    // operand and the accumulators they turn into does not fit the register file).  Shifted sums per lane half, merged
    // with the partner lane's (Chan): no cancellation whatever the row's mean.
    float k0 = 0.f, s1 = 0.f, s2 = 0.f;
    for_rows([&](auto sc, const float4 a, const float4 b) {
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
    // Pass 2 (the panel is in L2 now): normalise into the fc1 operand, and hand the raw values to the fc2 accumulators --
    // the residual add without a second HBM read of H.  This lane pair holds columns 16s + 8hh + 0..7; the accumulator
    // layout wants lane hh to hold columns 8g + 4hh + 0..3 of every group of 8: lane 0 keeps its first four of each eight
    // and takes lane 1's first four, lane 1 keeps its last four and takes lane 0's -- one v_permlane32_swap per register
    // pair (lanes l and l + 32 are the two halves of a row).
    for_rows([&](auto sc, const float4 xa, const float4 xb) {
      constexpr int s = decltype(sc)::value;
      uint4 p;
      p.x = pack_bf16x2((xa.x - mean) * rstd, (xa.y - mean) * rstd);
      p.y = pack_bf16x2((xa.z - mean) * rstd, (xa.w - mean) * rstd);
      p.z = pack_bf16x2((xb.x - mean) * rstd, (xb.y - mean) * rstd);
      p.w = pack_bf16x2((xb.z - mean) * rstd, (xb.w - mean) * rstd);
      yf[s] = as_frag(p);
      float a[4] = {xa.x, xa.y, xa.z, xa.w}, b[4] = {xb.x, xb.y, xb.z, xb.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // lanes 32-63 of the first operand <-> lanes 0-31 of the second: a[32..63] <-> b[0..31].  Inline asm on purpose:
        // __builtin_amdgcn_permlane32_swap on bit-cast floats came back with BOTH result elements in the first operand's
        // register (hipcc 7.2; columns 8..15 of every 16 were copies of 0..7).  s_nop 1 = the two wait states a VALU write
        // of an operand needs before the swap reads it (nothing inside an asm statement is padded by the compiler).
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[e]), "+v"(b[e]));
      }
      // columns 16s + 4hh + e (q even) and 16s + 8 + 4hh + e (q odd) of tile s / 2
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        oacc[s >> 1][4 * (2 * (s & 1)) + e] = a[e];
        oacc[s >> 1][4 * (2 * (s & 1) + 1) + e] = b[e];
      }
    });
    }
  }

  bf16x8 zb0, zb1;   // gelu(fc1) of the previous chunk: the two 16-deep k-steps of fc2's B operand
  {
    uint4 zero = make_uint4(0, 0, 0, 0);
    zb0 = as_frag(zero);
    zb1 = as_frag(zero);
  }

  // One stage = fc1 of chunk j (W1 pieces 0..KS-1 + the bias piece) then fc2 of chunk j-1 (W2 pieces KS..WP-1, k-step
  // major: every accumulator is touched once per half).  What runs beside the 2 KS MFMAs, all from this one wave:
  //   * the fragments, read from LDS in consumption order through a ring of PF register sets (PF reads ahead, counted
  //     lgkmcnt, piece offset in the instruction's immediate) -- ACROSS stage boundaries: the ring never drains;
  //   * GELU of chunk j (vector pipe), spread over the fc2 MFMAs of chunk j-1 (matrix pipe), which do not depend on it;
  //   * ONE workgroup barrier, in the MIDDLE of the stage: it publishes stage j+1 (every wave has waited for its own
  //     pieces, issued half a stage earlier) and retires stage j-1, whose buffer the DMAs of stage j+2 then refill, one
  //     every other MFMA of the second half.  At the stage boundary itself there is no synchronisation at all (a barrier
  //     there cost ~450 cycles per stage: the matrix pipe ran dry while the first fragments of the new stage were read).
  // The instruction order is pinned (compile-time loop, sched_barrier after every step): left to itself hipcc issued
  // every ds_read right before its MFMA behind an lgkmcnt(0) -- one LDS latency per MFMA with one wave per SIMD.
  constexpr int PF = (ABL & 8) ? 2 : 4;
  constexpr bool TWO_ACC = (ABL & 4) != 0;
  uint4 fr[PF];
  f32x16 za, zc;   // fc1 accumulator(s); za arrives at a stage holding the chunk's bias
  auto load_bias = [&](int buf) {
    const float4* bp = reinterpret_cast<const float4*>(lds + buf * (PIECES * 64) + WP * 64) + hh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // accumulator row of register 4q + e is n = 8q + 4hh + e
      const float4 b = bp[2 * q];
      za[4 * q] = b.x;
      za[4 * q + 1] = b.y;
      za[4 * q + 2] = b.z;
      za[4 * q + 3] = b.w;
    }
  };
  // kind: 0 = first stage (fc1 only), 1 = steady, 2 = last stage (fc2 only); next_f0 = first piece the NEXT stage reads
  auto stage = [&](int j, int buf, auto kind_c, int next_f0) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool DO1 = KIND != 2, DO2 = KIND != 0, LAST = KIND == 2;
    constexpr int F0 = DO1 ? 0 : KS, NF = (DO1 ? KS : 0) + (DO2 ? KS : 0), MID = NF / 2;
    // slot of step f in the fragment ring: the ring runs on across stages, the first stage is KS steps long, every other
    // one 2 KS (a multiple of PF), the last one KS again
    constexpr int RO = KIND == 0 ? 0 : KS % PF;
    static_assert((2 * KS) % PF == 0, "a steady stage must leave the ring's phase unchanged");
    // GELU values [16 (i - G0) / GN, 16 (i + 1 - G0) / GN) (rounded up) are computed behind fc2 step i
    constexpr int G0 = KS > 2 ? 1 : 0, GN = KS - G0 - (KS > 3 ? 1 : 0);
    // DMA group g of this wave goes behind step MID + 1 + g * DS (all behind MID with ABL bit 4)
    constexpr int DS = (ABL & 16) ? 0 : ((NF - MID - 2) / NG > 0 ? (NF - MID - 2) / NG : 1);
    const int nbuf = buf + 1 == NST ? 0 : buf + 1;           // stage j + 1
    const bf16_t* nsrc = stage_src(NOP + j + 2);
    const unsigned ndst = stage_dst(buf == 0 ? NST - 1 : buf - 1);   // stage j + 2 goes where stage j - 1 was
    const uint4* st = lds + buf * (PIECES * 64) + lane;
    const uint4* stn = lds + nbuf * (PIECES * 64) + next_f0 * 64 + lane;
    if constexpr (DO1 && TWO_ACC) {
#pragma unroll
      for (int r = 0; r < 16; ++r) zc[r] = 0.f;
    }
    float gv[16];
    uint32_t zn[8];
    static_for<NF>([&](auto fc) {
      constexpr int f = decltype(fc)::value, piece = F0 + f;
      if constexpr (piece < KS) {
        if constexpr ((ABL & 256) != 0) {
        } else if constexpr (!TWO_ACC || (piece & 1) == 0)
          za = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fr[(RO + f) % PF]), yf[piece], za, 0, 0, 0);
        else
          zc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fr[(RO + f) % PF]), yf[piece], zc, 0, 0, 0);
      } else {
        constexpr int i = piece - KS, u = i / CT, t = i % CT;
        if constexpr ((ABL & 256) == 0)
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fr[(RO + f) % PF]), u ? zb1 : zb0, oacc[t], 0, 0, 0);
        if constexpr (DO1) {
          // (the empty asm pins each value HERE: the IR-level passes otherwise sink the whole GELU below the last MFMA,
          // where nothing overlaps it -- sched_barrier only binds the machine scheduler)
          constexpr int r0 = i < G0 ? 0 : (16 * (i - G0) + GN - 1) / GN, r1x = i + 1 < G0 ? 0 : (16 * (i + 1 - G0) + GN - 1) / GN;
          constexpr int r1 = r1x > 16 ? 16 : r1x;
          static_for<(r1 > r0 ? r1 - r0 : 0)>([&](auto rc) {
            constexpr int r = r0 + decltype(rc)::value;
            const float x = TWO_ACC ? za[r] + zc[r] : za[r];
            gv[r] = (ABL & 2) ? x : gelu_sig(x);
            if constexpr ((r & 1) != 0) {
              zn[r >> 1] = pack_bf16x2(gv[r - 1], gv[r]);
              asm volatile("" : "+v"(zn[r >> 1]));
            } else {
              asm volatile("" : "+v"(gv[r]));
            }
          });
        }
      }
      if constexpr (f == MID) {
        if constexpr ((ABL & 1) == 0) wait_vmcnt<0>();            // this wave's pieces of stage j + 1 (nothing younger is in flight)
        if constexpr ((ABL & 128) == 0) __builtin_amdgcn_s_barrier();   // stage j + 1 complete for everyone; stage j - 1 read by everyone
      }
      if constexpr ((ABL & 1) == 0 && f > MID) {
        static_for<NG>([&](auto qc) {
          constexpr int q = decltype(qc)::value, at = (MID + 1 + q * DS) < NF ? (MID + 1 + q * DS) : NF - 1;
          if constexpr (at == f) issue_group(nsrc, ndst, qc);
        });
      }
      if constexpr ((ABL & 64) == 0) {
        if constexpr (f + PF < NF) fr[(RO + f) % PF] = st[(F0 + f + PF) * 64];
        else if constexpr (!LAST) fr[(RO + f) % PF] = stn[(f + PF - NF) * 64];   // the next stage's first fragments (published at MID)
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (KIND == 0) {   // first stage: nothing to overlap with
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float x0 = TWO_ACC ? za[r] + zc[r] : za[r], x1 = TWO_ACC ? za[r + 1] + zc[r + 1] : za[r + 1];
        zn[r >> 1] = pack_bf16x2(gelu_sig(x0), gelu_sig(x1));
      }
    }
    if constexpr (DO1) {
      const uint4 p0 = make_uint4(zn[0], zn[1], zn[2], zn[3]), p1 = make_uint4(zn[4], zn[5], zn[6], zn[7]);
      zb0 = as_frag(p0);
      zb1 = as_frag(p1);
    }
    if constexpr (!LAST) load_bias(nbuf);   // next chunk's bias into the fc1 accumulator (only read when the next stage has an fc1)
  };
  if constexpr ((ABL & 32) == 0) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // stages 0 and 1 are in the ring
    {
      const uint4* st = lds + lane;
#pragma unroll
      for (int p = 0; p < PF; ++p) fr[p] = st[p * 64];
      load_bias(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    int buf = 0;
    if constexpr (OP) {
      // ---- o-proj: stage I holds the Wo fragments of output tiles 2I and 2I + 1 (k-step major inside a tile); the B operand
      // is the wave's AO block, the accumulators are fc2's and already hold the residual ----
      bf16x8 yao[KS];
      {
        const bf16_t* ap = AO + (long)(row < R ? row : R - 1) * D + hh * 8;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
          const uint4 p = *reinterpret_cast<const uint4*>(ap + s2 * 16);
          yao[s2] = as_frag(p);
        }
      }
      static_for<NOP>([&](auto ic) {
        constexpr int I = decltype(ic)::value, T0 = 2 * I, T1 = 2 * I + 1 < CT ? 2 * I + 1 : CT - 1;   // (a missing last tile: zero weights)
        constexpr int NF = 2 * KS, MID = NF / 2;
        static_assert(NF % PF == 0, "an o-proj stage must leave the fragment ring's phase unchanged");
        constexpr int DS = (ABL & 16) ? 0 : ((NF - MID - 2) / NG > 0 ? (NF - MID - 2) / NG : 1);
        const int nbuf = buf + 1 == NST ? 0 : buf + 1;
        const bf16_t* nsrc = stage_src(I + 2);
        const unsigned ndst = stage_dst(buf == 0 ? NST - 1 : buf - 1);
        const uint4* st = lds + buf * (PIECES * 64) + lane;
        const uint4* stn = lds + nbuf * (PIECES * 64) + lane;
        static_for<NF>([&](auto fc) {
          constexpr int f = decltype(fc)::value, T = f < KS ? T0 : T1, s2 = f % KS;
          oacc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fr[f % PF]), yao[s2], oacc[T], 0, 0, 0);
          if constexpr (f == MID) {
            if constexpr ((ABL & 1) == 0) wait_vmcnt<0>();
            if constexpr ((ABL & 128) == 0) __builtin_amdgcn_s_barrier();
          }
          if constexpr ((ABL & 1) == 0 && f > MID) {
            static_for<NG>([&](auto qc) {
              constexpr int q = decltype(qc)::value, at = (MID + 1 + q * DS) < NF ? (MID + 1 + q * DS) : NF - 1;
              if constexpr (at == f) issue_group(nsrc, ndst, qc);
            });
          }
          if constexpr (f + PF < NF) fr[f % PF] = st[(f + PF) * 64];
          else fr[f % PF] = stn[(f + PF - NF) * 64];   // the next stage's first fragments (published at MID)
          __builtin_amdgcn_sched_barrier(0);
        });
        buf = nbuf;
      });
      // ---- LayerNorm of H' from the accumulators: this lane holds half a row (columns 32t + 8q + 4hh + e) ----
      {
        const float k0 = oacc[0][0];
        float s1 = 0.f, s2v = 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = oacc[t][r] - k0;
            s1 += d;
            s2v += d * d;
          }
        constexpr float kHalf = D / 2;
        const float mean_l = k0 + s1 * (1.0f / kHalf), m2_l = s2v - s1 * s1 * (1.0f / kHalf);
        const float mean_o = __shfl_xor(mean_l, 32, 64), m2_o = __shfl_xor(m2_l, 32, 64);
        const float mean = 0.5f * (mean_l + mean_o), dm = mean_l - mean_o;
        const float var = (m2_l + m2_o + dm * dm * (0.5f * kHalf)) * (1.0f / D);
        const float rstd = rsqrtf(var + 1e-5f);
        // back to the operand layout (columns 16s + 8hh + 0..7): the swap of the prologue is its own inverse
        static_for<KS>([&](auto sc) {
          constexpr int s3 = decltype(sc)::value, t = s3 >> 1, q0 = 2 * (s3 & 1);
          float a[4], b[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = oacc[t][4 * q0 + e];
            b[e] = oacc[t][4 * (q0 + 1) + e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[e]), "+v"(b[e]));
          uint4 p;
          p.x = pack_bf16x2((a[0] - mean) * rstd, (a[1] - mean) * rstd);
          p.y = pack_bf16x2((a[2] - mean) * rstd, (a[3] - mean) * rstd);
          p.z = pack_bf16x2((b[0] - mean) * rstd, (b[1] - mean) * rstd);
          p.w = pack_bf16x2((b[2] - mean) * rstd, (b[3] - mean) * rstd);
          yf[s3] = as_frag(p);
        });
      }
      load_bias(buf);   // b1 of chunk 0 (its stage was published at the last o-proj stage's barrier)
      __builtin_amdgcn_sched_barrier(0);
    }
    stage(0, buf, std::integral_constant<int, 0>{}, NC > 1 ? 0 : KS);
    for (int j = 1; j < NC; ++j) {
      buf = buf + 1 == NST ? 0 : buf + 1;
      stage(j, buf, std::integral_constant<int, 1>{}, j + 1 < NC ? 0 : KS);
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
    stage(NC, buf, std::integral_constant<int, 2>{}, 0);
    wait_vmcnt<0>();   // the trailing re-fetches land in buffers nobody reads; they must not outlive the workgroup's LDS
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" ::"v"(yf[s]));
  }

  // ---- epilogue: H[row][c] = acc + b2, accumulator rows are output columns c = 32t + 8q + 4hh + e ----
  {
    const float* bp = b2 + hh * 4;
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(bp + t * 32 + q * 8);
        oacc[t][4 * q] += b.x;
        oacc[t][4 * q + 1] += b.y;
        oacc[t][4 * q + 2] += b.z;
        oacc[t][4 * q + 3] += b.w;
      }
  }
  if (row < R) {
    float* op = H + (long)row * D + hh * 4;
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = t * 32 + q * 8;
        typedef float f32x4_native __attribute__((ext_vector_type(4)));
        const f32x4_native v = {oacc[t][4 * q], oacc[t][4 * q + 1], oacc[t][4 * q + 2], oacc[t][4 * q + 3]};
        if constexpr (NTS) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_native*>(op + c));
        else *reinterpret_cast<f32x4_native*>(op + c) = v;
      }
  }
  if constexpr (YOUT) {
    // LayerNorm of the new rows from the accumulators (this lane holds half a row: columns 32t + 8q + 4hh + e), shifted
    // moments per lane half merged with the partner lane's (Chan) as in the prologue, then back to the operand layout
    // (columns 16s + 8hh + 0..7: the swap is its own inverse) and out as one KiB per k-step and wave.  Rows beyond R are
    // written too (garbage, inside the buffer's padding to whole panels): nobody reads them as valid rows.
    const float k0 = oacc[0][0];
    float s1 = 0.f, s2v = 0.f;
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = oacc[t][r] - k0;
        s1 += d;
        s2v += d * d;
      }
    constexpr float kHalf = D / 2;
    const float mean_l = k0 + s1 * (1.0f / kHalf), m2_l = s2v - s1 * s1 * (1.0f / kHalf);
    const float mean_o = __shfl_xor(mean_l, 32, 64), m2_o = __shfl_xor(m2_l, 32, 64);
    const float mean = 0.5f * (mean_l + mean_o), dm = mean_l - mean_o;
    const float var = (m2_l + m2_o + dm * dm * (0.5f * kHalf)) * (1.0f / D);
    const float rstd = rsqrtf(var + 1e-5f);
    uint4* yp = reinterpret_cast<uint4*>(Yfm) + ((long)(blockIdx.x * 4 + wave) * KS) * 64 + lane;
    static_for<KS>([&](auto sc) {
      constexpr int s3 = decltype(sc)::value, t = s3 >> 1, q0 = 2 * (s3 & 1);
      float a[4], b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = oacc[t][4 * q0 + e];
        b[e] = oacc[t][4 * (q0 + 1) + e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[e]), "+v"(b[e]));
      uint4 p;
      p.x = pack_bf16x2((a[0] - mean) * rstd, (a[1] - mean) * rstd);
      p.y = pack_bf16x2((a[2] - mean) * rstd, (a[3] - mean) * rstd);
      p.z = pack_bf16x2((b[0] - mean) * rstd, (b[1] - mean) * rstd);
      p.w = pack_bf16x2((b[2] - mean) * rstd, (b[3] - mean) * rstd);
      yp[s3 * 64] = p;
    });
  }
}

template <int D, int ABL = 0>
void launch_mlp(float* H, const bf16_t* Wp, const float* b2, int R, int F, hipStream_t s) {
  MSH_LAUNCH((mlp_fused_kernel<D, ABL, false>), dim3((R + 127) / 128), dim3(256), 0, s, H, Wp, b2, R, F / 32, (const bf16_t*)nullptr,
             (bf16_t*)nullptr);
}
template <int D>
void launch_mlp_o(float* H, const bf16_t* AO, const bf16_t* Wp, const float* b2, int R, int F, hipStream_t s, bool store_nt, bf16_t* yfm) {
  const dim3 grid((R + 127) / 128);
  if (yfm != nullptr) {
    if constexpr (D == 416) {
      if (store_nt) {
        MSH_LAUNCH((mlp_fused_kernel<D, 0, true, true, true>), grid, dim3(256), 0, s, H, Wp, b2, R, F / 32, AO, yfm);
        return;
      }
    }
    MSH_LAUNCH((mlp_fused_kernel<D, 0, true, false, true>), grid, dim3(256), 0, s, H, Wp, b2, R, F / 32, AO, yfm);
    return;
  }
  if constexpr (D == 416) {
    if (store_nt) {
      MSH_LAUNCH((mlp_fused_kernel<D, 0, true, true>), grid, dim3(256), 0, s, H, Wp, b2, R, F / 32, AO, yfm);
      return;
    }
  }
  MSH_LAUNCH((mlp_fused_kernel<D, 0, true>), grid, dim3(256), 0, s, H, Wp, b2, R, F / 32, AO, yfm);
}

}  // namespace

bool mlp_fused_supported(int D, int F) { return (D == 416 || D == 288 || D == 64) && F % 32 == 0 && F >= 32; }

size_t mlp_packed_elems(int D, int F, bool with_oproj) {
  return (size_t)(F / 32 + 1 + (with_oproj ? (D / 32 + 1) / 2 : 0)) * (D / 8 + 1) * 512;
}

// Host-side packing (once, at load).  w1 [F][D] (gamma is folded in here), b1 [F], w2 [D][F] -> (F/32 + 1) stages of
// (D/8 + 1) KiB: stage j = { W1 fragments of hidden rows 32j..32j+31 (k-step s: lane l holds row 32j + (l & 31), columns
// 16s + 8(l >> 5) + 0..7) | W2 fragments of chunk j-1 (tile t, k-step u: lane l holds output row 32t + (l & 31), hidden
// columns 32(j-1) + 8(2u + (e >> 2)) + 4(l >> 5) + (e & 3) for e = 0..7 -- the order the 32x32 accumulator layout hands
// gelu(fc1) over in) | 32 floats of b1 }.  Stage 0 has no W2 part and stage F/32 no W1 part (zeros).
// wo (nullable) [D][D]: the attention output projection; its (CT + 1) / 2 stages come first: stage I = { Wo fragments of
// output tile 2I, k-steps 0..KS-1 | of tile 2I + 1 (zeros when there is none) | 32 + ... floats of zeros }.
void pack_mlp_weights(const float* w1, const float* gamma, const float* b1, const float* w2, int D, int F, bf16_t* out,
                      const float* wo) {
  const int KS = D / 16, CT = D / 32, WP = KS + 2 * CT, PIECES = WP + 1, NC = F / 32;
  const bf16_t zero = f32_to_bf16(0.f);
  if (wo != nullptr) {
    const int NOP = (CT + 1) / 2;
    for (int I = 0; I < NOP; ++I) {
      bf16_t* stage = out + (size_t)I * PIECES * 512;
      for (int h2 = 0; h2 < 2; ++h2)
        for (int s = 0; s < KS; ++s)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
              const int t = 2 * I + h2, n = 32 * t + (l & 31), k = 16 * s + 8 * (l >> 5) + e;
              stage[((size_t)(h2 * KS + s) * 64 + l) * 8 + e] = t < CT ? f32_to_bf16(wo[(size_t)n * D + k]) : zero;
            }
      float* bias = reinterpret_cast<float*>(stage + (size_t)WP * 512);
      for (int i = 0; i < 256; ++i) bias[i] = 0.f;
    }
    out += (size_t)NOP * PIECES * 512;
  }
  for (int j = 0; j <= NC; ++j) {
    bf16_t* stage = out + (size_t)j * PIECES * 512;
    for (int s = 0; s < KS; ++s)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) {
          const int n = 32 * j + (l & 31), k = 16 * s + 8 * (l >> 5) + e;
          stage[((size_t)s * 64 + l) * 8 + e] = j < NC ? f32_to_bf16(w1[(size_t)n * D + k] * gamma[k]) : zero;
        }
    for (int t = 0; t < CT; ++t)
      for (int u = 0; u < 2; ++u)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            const int c = 32 * t + (l & 31), n = 32 * (j - 1) + 8 * (2 * u + (e >> 2)) + 4 * (l >> 5) + (e & 3);
            stage[((size_t)(KS + u * CT + t) * 64 + l) * 8 + e] = j >= 1 ? f32_to_bf16(w2[(size_t)c * F + n]) : zero;
          }
    float* bias = reinterpret_cast<float*>(stage + (size_t)WP * 512);
    for (int i = 0; i < 256; ++i) bias[i] = (i < 32 && j < NC) ? b1[32 * j + i] : 0.f;
  }
}

// yfm (nullable): [ceil(R / 128) * 128][D] bf16, receives LayerNorm (no scale) of the new rows in fragment-major order, the
// operand of the next layer's qkv_panel_prenorm
void mlp_fused_oproj(float* H, const bf16_t* AO, const bf16_t* Wp, const float* b2, int R, int D, int F, hipStream_t s, bool store_nt,
                     bf16_t* yfm) {
  if (R <= 0) return;
  switch (D) {
    case 416: return launch_mlp_o<416>(H, AO, Wp, b2, R, F, s, store_nt, yfm);
    case 288: return launch_mlp_o<288>(H, AO, Wp, b2, R, F, s, false, yfm);
    case 64: return launch_mlp_o<64>(H, AO, Wp, b2, R, F, s, false, yfm);
    default: throw std::runtime_error("mlp_fused_oproj: unsupported hidden size");
  }
}

void mlp_fused(float* H, const bf16_t* Wp, const float* b2, int R, int D, int F, hipStream_t s) {
  if (R <= 0) return;
  switch (D) {
    case 416: return launch_mlp<416>(H, Wp, b2, R, F, s);
    case 288: return launch_mlp<288>(H, Wp, b2, R, F, s);
    case 64: return launch_mlp<64>(H, Wp, b2, R, F, s);
    default: throw std::runtime_error("mlp_fused: unsupported hidden size");
  }
}

// Test hook (tests/test_gpu_mlp.py): packs the weights and runs the kernel once on h [R][D] (host, in / out).
void mlp_fused_host(float* h, int R, int D, int F, const float* w1, const float* gamma, const float* b1, const float* w2,
                    const float* b2, const float* ao, const float* wo, uint16_t* y_fm) {
  if (!mlp_fused_supported(D, F)) throw std::runtime_error("mlp_fused: unsupported shape");
  const bool op = ao != nullptr && wo != nullptr;
  std::vector<bf16_t> packed(mlp_packed_elems(D, F, op));
  pack_mlp_weights(w1, gamma, b1, w2, D, F, packed.data(), op ? wo : nullptr);
  float *H = nullptr, *B2 = nullptr;
  bf16_t *Wp = nullptr, *AOd = nullptr;
  if (op) {
    std::vector<bf16_t> a16((size_t)R * D);
    for (size_t i = 0; i < a16.size(); ++i) a16[i] = f32_to_bf16(ao[i]);
    MSH_HIP(hipMalloc(&AOd, a16.size() * 2));
    MSH_HIP(hipMemcpy(AOd, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
  }
  MSH_HIP(hipMalloc(&H, (size_t)R * D * 4));
  MSH_HIP(hipMalloc(&B2, (size_t)D * 4));
  MSH_HIP(hipMalloc(&Wp, packed.size() * 2));
  MSH_HIP(hipMemcpy(H, h, (size_t)R * D * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(B2, b2, (size_t)D * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(Wp, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  bf16_t* Yd = nullptr;
  const size_t ybytes = (size_t)((R + 127) / 128 * 128) * D * 2;
  if (y_fm != nullptr) {
    if (!op) throw std::runtime_error("mlp_fused: the fragment-major LayerNorm output rides on the o-proj form");
    MSH_HIP(hipMalloc(&Yd, ybytes));
  }
  if (op) mlp_fused_oproj(H, AOd, Wp, B2, R, D, F, 0, false, Yd);
  else mlp_fused(H, Wp, B2, R, D, F, 0);
  MSH_HIP(hipDeviceSynchronize());
  MSH_HIP(hipMemcpy(h, H, (size_t)R * D * 4, hipMemcpyDeviceToHost));
  if (Yd != nullptr) {
    MSH_HIP(hipMemcpy(y_fm, Yd, ybytes, hipMemcpyDeviceToHost));
    (void)hipFree(Yd);
  }
  if (AOd != nullptr) (void)hipFree(AOd);
  (void)hipFree(H);
  (void)hipFree(B2);
  (void)hipFree(Wp);
}

// Microbenchmark (tools/mlp_microbench.py): ms per launch on uniform random [-1, 1) data, R rows.
float mlp_microbench(int R, int D, int F, int iters, int abl) {
  if (!mlp_fused_supported(D, F)) throw std::runtime_error("mlp_microbench: unsupported shape");
  std::vector<float> w1((size_t)F * D), w2((size_t)D * F), g(D, 1.f), b1(F), b2(D), h((size_t)R * D);
  unsigned x = 12345u;
  auto rnd = [&] {
    x = x * 1664525u + 1013904223u;
    return (float)((x >> 8) & 0xffff) / 32768.0f - 1.0f;
  };
  for (auto& v : w1) v = rnd() * 0.05f;
  for (auto& v : w2) v = rnd() * 0.025f;
  for (auto& v : b1) v = rnd() * 0.1f;
  for (auto& v : b2) v = rnd() * 0.1f;
  for (auto& v : h) v = rnd();
  std::vector<bf16_t> packed(mlp_packed_elems(D, F, false));
  pack_mlp_weights(w1.data(), g.data(), b1.data(), w2.data(), D, F, packed.data(), nullptr);
  float *H = nullptr, *B2 = nullptr;
  bf16_t* Wp = nullptr;
  MSH_HIP(hipMalloc(&H, h.size() * 4));
  MSH_HIP(hipMalloc(&B2, b2.size() * 4));
  MSH_HIP(hipMalloc(&Wp, packed.size() * 2));
  MSH_HIP(hipMemcpy(H, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(B2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(Wp, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  auto run = [&] {
    if (abl == 0) return mlp_fused(H, Wp, B2, R, D, F, 0);
    if (D != 416) throw std::runtime_error("mlp_microbench: ablations are compiled for D = 416");
    switch (abl) {
      case 1: return launch_mlp<416, 1>(H, Wp, B2, R, F, 0);
      case 2: return launch_mlp<416, 2>(H, Wp, B2, R, F, 0);
      case 3: return launch_mlp<416, 3>(H, Wp, B2, R, F, 0);
      case 4: return launch_mlp<416, 4>(H, Wp, B2, R, F, 0);
      case 8: return launch_mlp<416, 8>(H, Wp, B2, R, F, 0);
      case 16: return launch_mlp<416, 16>(H, Wp, B2, R, F, 0);
      case 32: return launch_mlp<416, 32>(H, Wp, B2, R, F, 0);
      case 67: return launch_mlp<416, 67>(H, Wp, B2, R, F, 0);
      case 195: return launch_mlp<416, 195>(H, Wp, B2, R, F, 0);
      case 199: return launch_mlp<416, 199>(H, Wp, B2, R, F, 0);
      case 322: return launch_mlp<416, 322>(H, Wp, B2, R, F, 0);   // 256 + 64 + 2: the weight stream alone
      default: throw std::runtime_error("mlp_microbench: bad ablation");
    }
  };
  run();
  MSH_HIP(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  MSH_HIP(hipEventCreate(&e0));
  MSH_HIP(hipEventCreate(&e1));
  MSH_HIP(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) run();   // (H stays bounded: every pass normalises its input)
  MSH_HIP(hipEventRecord(e1, 0));
  MSH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MSH_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(H);
  (void)hipFree(B2);
  (void)hipFree(Wp);
  return ms / iters;
}

}  // namespace msh
