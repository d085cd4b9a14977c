// Attention kernels for gfx950.
//
//  enc_attention        non-causal MHA over the packed encoder stream (flash-style: 64-key blocks staged
//                       through LDS, online softmax in fp32, bf16 MFMA 16x16x32 for QK^T and PV).
//                       Both products are computed transposed (S^T = K Q^T, O^T = V^T P^T) so that the
//                       query index is the MFMA column (lane & 15) everywhere: softmax statistics, the
//                       P fragment and the O accumulator all live in the same lane, no cross-lane moves
//                       other than two xor-shuffles for the row max.
//  dec_self_attention   one wave per (clip, head): <= Smax cached keys, latency-bound, plain VALU.
//  dec_cross_attention  one workgroup per (clip, head), 4 waves splitting the head dim, streaming K^T / V^T
//                       ([dh][Tk] bf16, keys contiguous) with 16-byte non-temporal buffer loads: the
//                       HBM-bound kernel that dominates batched decode (5.5 MB per clip per step at
//                       Moonshine-base).
#include <math.h>
#include <stdlib.h>

#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "kernels.h"

namespace msh {
namespace {

constexpr float kLog2e = 1.4426950408889634f;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// cache-policy bits of the buffer-load aux operand: 2 = nt (non-temporal).  The cross K^T/V^T stream
// (1.4 GB per decode step at batch 256) is read exactly once per step; marking it streaming keeps it from
// evicting the 77 MB of decoder weights that every step re-reads (MI355X_MICROARCH.md, row nt-weights).
#ifndef MSH_CROSS_KV_AUX
#define MSH_CROSS_KV_AUX 2
#endif
constexpr int kStreamAux = MSH_CROSS_KV_AUX;
#ifndef MSH_FP8_ATT_OCC
#define MSH_FP8_ATT_OCC 2   // minimum workgroups per CU the fp8 variant is compiled for: it lands at 114 VGPRs = 4 per CU by itself;
                          // forcing 5 (96 VGPRs, 13 spilled) measured 26.5 us per launch against 19.2
#endif

// compile-time loop: body(std::integral_constant<int, I>) for I in [0, N)
template <class Body, int... I>
__device__ __forceinline__ void static_for_att_impl(Body&& body, std::integer_sequence<int, I...>) {
  (body(std::integral_constant<int, I>{}), ...);
}
template <int N, class Body>
__device__ __forceinline__ void static_for_att(Body&& body) {
  static_for_att_impl(static_cast<Body&&>(body), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Reductions across the four 16-lane rows of a wave (lane ^ 16, lane ^ 32) with the gfx950 row-swap instructions: pure
// VALU, no trip through the LDS crossbar (__shfl_xor lowers to ds_bpermute + a wait on lgkmcnt).
// v_permlane16_swap(a, b): rows 1, 3 of a <-> rows 0, 2 of b;  v_permlane32_swap(a, b): upper half of a <-> lower half of b.
__device__ __forceinline__ float rows_max(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const unsigned w = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned w = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
// Full-wave reductions without the LDS crossbar: four DPP row rotations inside the 16-lane rows, then the row swaps above.
// Every lane ends with the result.  (Fixed order: deterministic; not the order of wave_max / wave_sum.)
#define MSH_DPP_ROR(v, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xf, 0xf, true))
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, MSH_DPP_ROR(v, 8));
  v = fmaxf(v, MSH_DPP_ROR(v, 4));
  v = fmaxf(v, MSH_DPP_ROR(v, 2));
  v = fmaxf(v, MSH_DPP_ROR(v, 1));
  return rows_max(v);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += MSH_DPP_ROR(v, 8);
  v += MSH_DPP_ROR(v, 4);
  v += MSH_DPP_ROR(v, 2);
  v += MSH_DPP_ROR(v, 1);
  return rows_sum(v);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// ------------------------------------------------------------------------------------------------
// One 64-key block for the first NA query tiles of a wave -- the arithmetic BOTH encoder attention kernels below run, so
// that a clip's output does not depend on which of them a batch's longest clip selected (identical bits).
//   Kb   the block's K rows in LDS: [key][8 x 16 B], chunk ^= (key >> 1) & 7, head dim zero-padded to 64
//   Vb   the block's V^T in LDS: [d][VLD], rows DH .. padded with zeros, row DH = ones when DH % 16 != 0 (the softmax
//        denominator then comes out of the P.V MFMA as "dimension DH" of O: the same bf16 P that forms the numerator)
//   qf   the tiles' query fragments, PRE-SCALED by rsqrt(dh) * log2(e) (folded into the q projection at load)
//   nm   MINUS the reference point of each tile's exponent, as the accumulator the score MFMAs start from: the MFMA output
//        is already `score - reference` in the exp2 domain, no multiply-add per score
//   first  this is the clip's first key block: the reference (0 so far) always moves to the block's maximum
//   keys_left  valid keys from the block's first one on (only read when MASKED: the block holds the clip's last valid key)
// The block is a PIPELINE over the tiles: while tile i's scores go through max / (rare) reference move / exp2 / bf16 packing
// on the vector pipe, tile i + 1's score MFMAs run on the matrix pipe, and tile i's P.V MFMAs run beside tile i + 1's
// maxima.  Between two tiles there is exactly one rarely taken wave-uniform branch (does this tile's reference point move:
// did a query find a score more than kTau above it?  per lane -- the cross-lane maximum is only formed inside the branch);
// everything else of a tile is one basic block of 16 MFMAs and ~36 VALU instructions for the scheduler to interleave.
// ABL (microbenchmark only): 4 = no exp2, 8 = no MFMAs.
// ------------------------------------------------------------------------------------------------
constexpr int KB = 64;        // keys per block
constexpr float kAttTau = 8.0f;

template <int DH, int VLD, int EQT, int NA, bool MASKED, int ABL = 0>
__device__ __forceinline__ void att_key_block(const uint4* __restrict__ Kb, const bf16_t* __restrict__ Vb, bool first, int keys_left,
                                              int li, int kg, const bf16x8 (&qf)[EQT][2], f32x4 (&nm)[EQT], f32x4 (&o)[EQT][4],
                                              float (&l_run)[EQT]) {
  constexpr bool ONES_ROW = (DH % 16) != 0;
  uint4 kf[4][2];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = kt * 16 + li;
#pragma unroll
    for (int s = 0; s < 2; ++s) kf[kt][s] = Kb[key * 8 + ((s * 4 + kg) ^ ((key >> 1) & 7))];
  }
  // st[kt][r] = score(q = li of the tile, key = kt*16 + kg*4 + r of the block) - reference, in the exp2 domain
  auto scores = [&](auto qc, f32x4 (&st)[4]) {
    constexpr int qi = decltype(qc)::value;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 a = nm[qi];
      if constexpr ((ABL & 8) == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&kf[kt][s]), qf[qi][s], a, 0, 0, 0);
      } else {
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(kf[kt][0].x), "v"(kf[kt][1].w), "v"(qf[qi][0][0]));
      }
      st[kt] = a;
    }
  };
  f32x4 sa[4], sb[4];
  scores(std::integral_constant<int, 0>{}, sa);
  static_for_att<NA>([&](auto qc) {
    constexpr int qi = decltype(qc)::value;
    f32x4(&st)[4] = (qi & 1) ? sb : sa;
    f32x4(&sn)[4] = (qi & 1) ? sa : sb;
    if constexpr (MASKED) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt][r] = (kt * 16 + kg * 4 + r >= keys_left) ? -INFINITY : st[kt][r];
    }
    float m = -INFINITY;   // (this sequential form compiles to eight v_max3_f32; nested pairs came out as 21 v_max_f32 + 4 v_max3)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, st[kt][r]);
    // The reference point moves a few times per clip (and always in the first block).  Numerator and denominator use the
    // same reference; p <= 2^kAttTau keeps bf16 / fp32 in range.
    if (first || __any(m > kAttTau)) {
      const float delta = rows_max(m);                      // per query: the block's maximum above the old reference
      const float alpha = __builtin_amdgcn_exp2f(-delta);   // (delta is finite: the block's first key is valid for every query)
      const float nmv = nm[qi][0] - delta;
      nm[qi] = f32x4{nmv, nmv, nmv, nmv};
      if (!first) {   // (o and l are zero in the first block, and alpha may be 2^(+large) = inf there)
        if constexpr (!ONES_ROW) l_run[qi] *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[qi][i] *= alpha;
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt][r] -= delta;
    }
    // ---- one basic block from here to the next tile's branch ----
    if constexpr (qi + 1 < NA) scores(std::integral_constant<int, qi + 1>{}, sn);   // matrix pipe, independent of this tile
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr ((ABL & 4) == 0) st[kt][r] = __builtin_amdgcn_exp2f(st[kt][r]);
        if constexpr (!ONES_ROW) psum += st[kt][r];
      }
    if constexpr (!ONES_ROW) l_run[qi] += psum;
    // P^T fragments.  MFMA k-slot (kg, e): e < 4 -> key ks*32 + kg*4 + e, e >= 4 -> key ks*32 + 16 + kg*4 + e-4
    bf16x8 pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 pt;
      pt.x = pack_bf16x2(st[2 * ks][0], st[2 * ks][1]);
      pt.y = pack_bf16x2(st[2 * ks][2], st[2 * ks][3]);
      pt.z = pack_bf16x2(st[2 * ks + 1][0], st[2 * ks + 1][1]);
      pt.w = pack_bf16x2(st[2 * ks + 1][2], st[2 * ks + 1][3]);
      pf[ks] = *reinterpret_cast<bf16x8*>(&pt);
    }
    // O^T += V^T P^T
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if (dt * 16 >= DH) continue;
        const bf16_t* vr = Vb + (dt * 16 + li) * VLD + ks * 32 + kg * 4;
        const uint2 v0 = *reinterpret_cast<const uint2*>(vr);
        const uint2 v1 = *reinterpret_cast<const uint2*>(vr + 16);
        uint4 vtf = make_uint4(v0.x, v0.y, v1.x, v1.y);
        if constexpr ((ABL & 8) == 0) {
          o[qi][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&vtf), pf[ks], o[qi][dt], 0, 0, 0);
        } else {
          asm volatile("" : "+v"(o[qi][dt][0]), "+v"(o[qi][dt][3]) : "v"(vtf.x), "v"(vtf.w), "v"(pf[ks][0]), "v"(pf[ks][7]));
        }
      }
    }
  });
}

// ------------------------------------------------------------------------------------------------
// Encoder self-attention (modeling_moonshine.py:171-193, :283-362 with RoPE already applied by the QKV kernel; the encoder graph
// the reference runs at core/moonshine-model.cpp:270-274): the keys and values of a (clip, head) RESIDENT in LDS.
//
// Rounds 1-5 walked the keys in 64-key blocks (fetch into registers, commit to LDS, workgroup barrier, compute: seven times per
// workgroup for a 10 s clip, two workgroups per (clip, head) each staging every block) with a per-tile softmax of ~62 VALU
// instructions per 16 x 64 score tile.  PMC of that kernel and of the first resident version (profiles/r6c_pmc_attn_*): a VALU
// instruction costs the SIMD 4.6 cycles whichever wave issues it (v_exp_f32 8), 109 of them per tile and block with staging
// and epilogue against 16 MFMAs of 16 cycles; VALU issue was 42 % of the kernel's SIMD time, the matrix pipe 19 %, and four
// waves per SIMD instead of two moved nothing.  So the instruction count is the lever:
//   * a 10 s clip has 415 keys: K [448][64] + V^T [64][448] bf16 is 113 KiB, which one workgroup per CU can hold.  One
//     workgroup of NW waves per (clip, head, 512 queries) copies KMAX keys' K rows and V^T rows into LDS at a time -- every
//     load of the thread in flight together, thread -> (key, piece) fixed so that every LDS address is one base + an
//     immediate -- one barrier, and then every wave runs its query tiles (wave w: tiles w, w + NW, ...) over the chunk's key
//     blocks with no barrier, no global load and no staging register in the loop.  A 10 s clip is ONE chunk; longer clips
//     walk their keys in chunks of KMAX with two barriers per chunk (the online softmax state carries over);
//   * the queries arrive PRE-SCALED, the reference point rides in the accumulator init of the score MFMA, the "does the
//     reference move?" test is per lane, a block is a pipeline over the wave's tiles: att_key_block above.
// Per tile and block: 16 MFMAs, 8 v_max3 + 16 v_exp + 8 v_cvt_pk + ~5 (was ~62 in the loop, ~109 with staging and epilogue):
// 235 -> 188 us per layer at 256 x 10 s (tools/enc_attention_microbench.py, profiles/r6*_enc_att_microbench.txt).  What is
// left is not an instruction count: the same kernel on all-zero keys and values runs in 130 us with an identical instruction
// stream (the loads still waited for: ablation 2) -- with real operands the chip is power-limited here.
// The staging phase is not a latency to hide either: touching the NEXT item's lines from inside the key loop (one dword per
// 128-byte line, LDS-DMA into a sink, so that the next workgroup's staging hits the XCD's L2) made the kernel slower, and the
// same touches added to the "no global loads" ablation cost it 33 us = the 271 MB of q | k | v at the HBM rate -- bytes moved
// cost time here whenever they move (HISTORY.md, round 6).  What helps is moving fewer of them: the XCD-aware item order below.
// ABL (microbenchmark only, garbage results): 1 = no global loads while staging, 2 = loads waited for but zeroed, 4 = no exp2,
// 8 = no MFMAs.
// ------------------------------------------------------------------------------------------------
// MULTI: clips of more than KMAX frames exist in the batch (the chunk loop is compiled in; it costs the kernel ~300 bytes per
// lane of scratch, so batches of short clips run the instantiation without it -- the same att_key_block calls in the same
// order for every clip either way: identical bits).
template <int DH, int KMAX, int ABL = 0, int NW = 8, int EQT = 4, bool MULTI = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void enc_attention_res_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ vt,
                                                                   long vt_ld, bf16_t* __restrict__ out,
                                                                   const ClipMeta* __restrict__ clips, int n_clips, int gx,
                                                                   int D) {
  static_assert(DH % 4 == 0 && DH <= 64, "head_dim must be a multiple of 4, at most 64");
  static_assert(KMAX % 128 == 64, "V^T row stride (KMAX + 8) must be 72 mod 128 elements: conflict-free ds_read_b64");
  static_assert(NW % 4 == 0, "whole waves per SIMD");
  constexpr int NT = 64 * NW;
  constexpr int PIECES = DH / 4;            // 8-byte pieces per K row
  constexpr int VLD = KMAX + 8;             // V^T row stride in bf16
  constexpr int VROWS = (DH + 15) / 16 * 16;
  constexpr int VCH = KMAX / 8;             // 16-byte chunks (8 keys) per V^T row
  constexpr bool ONES_ROW = (DH % 16) != 0;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[KMAX * 128 + VROWS * VLD * 2];
  unsigned char* const Ks = lds_raw;                                   // [key][8 x 16 B], chunk ^= (key >> 1) & 7
  bf16_t* const Vt = reinterpret_cast<bf16_t*>(lds_raw + KMAX * 128);   // [d][key]

  // Workgroup b runs on XCD b % 8 (observed placement, used for speed only).  A head's K and Q slices are 104 of the 1664 bytes
  // of a q | k row: 128-byte lines are shared between neighbouring heads, so the heads of a clip belong on ONE XCD -- its L2
  // then fetches every line once (PMC, 256 x 10 s: 572 MB read per layer with head h on XCD h against 271 MB of q | k | v).
  // The 1-D grid (a multiple of 8, att_grid1) is dealt so that every XCD takes a contiguous run of (clip, query block, head).
  const int heads = D / DH, per = gridDim.x >> 3, vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= n_clips * gx * heads) return;
  const int clip = vb / (gx * heads), rr = vb - clip * (gx * heads), qx = rr / heads, h = rr - qx * heads;
  const ClipMeta cm = clips[clip];
  const int tile0 = qx * (NW * EQT);   // this workgroup's run of 16-query tiles
  if (tile0 * 16 >= cm.rows) return;
  const int T = cm.T;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long ld = 2L * D;   // [row][q | k]
  const bf16_t* base = qk + (long)cm.row_start * ld + h * DH;
  const bf16_t* vbase = vt + (long)(h * DH) * vt_ld + cm.row_start;
  const int nchunks = (T + KMAX - 1) / KMAX;

  // thread -> staging slots.  K: groups of 16 keys x PIECES pieces; iteration i adds 16 KG keys: the swizzle term
  // (key >> 1) & 7 does not change, the LDS address is base + 2048 KG i.  V^T: (row within VR, 16-byte chunk); iteration i
  // adds VR rows.
  constexpr int KG = NT / (16 * PIECES);          // 16-key groups per iteration
  constexpr int KIT = (KMAX + 16 * KG - 1) / (16 * KG);
  constexpr int VR = NT / VCH, VIT = (DH + VR - 1) / VR;

  // head-dim padding of K (pieces PIECES .. 15 of every key) and of V^T (rows DH .. VROWS - 1; row DH holds ones when it exists):
  // written once, the chunks' staging never touches it
  {
    const int vd0 = tid / VCH, vch = tid - vd0 * VCH;
    const bool vthread = vd0 < VR;
    const int npad = nchunks > 1 ? KMAX : (T + KB - 1) / KB * KB;
    for (int p = tid; p < npad * (16 - PIECES); p += NT) {
      const int key = p / (16 - PIECES), piece = PIECES + (p - key * (16 - PIECES));
      *reinterpret_cast<uint2*>(Ks + key * 128 + (((piece >> 1) ^ ((key >> 1) & 7)) * 16 + (piece & 1) * 8)) = make_uint2(0u, 0u);
    }
    if (vthread && vch * 8 < npad) {
#pragma unroll
      for (int r = 0; r < VROWS - DH; r += VR)
        if (r + vd0 < VROWS - DH) {
          const unsigned fill = (ONES_ROW && r + vd0 == 0) ? 0x3f803f80u : 0u;
          *reinterpret_cast<uint4*>(Vt + (DH + r + vd0) * VLD + vch * 8) = make_uint4(fill, fill, fill, fill);
        }
    }
  }

  // One chunk of KMAX keys: stage (every load of this thread first, then the LDS writes) ...
  // `between` runs after the chunk's loads are issued and before they are waited for: the first chunk issues the query loads
  // there, so that keys, values and queries arrive in ONE memory round trip (two dependent ones before; for a 1 s clip that
  // second round trip was a quarter of the workgroup's life)
  auto stage = [&](int ch, auto&& between) {
    const int c0 = ch * KMAX;                       // first key of the chunk
    const int Tc = T - c0 < KMAX ? T - c0 : KMAX;   // valid keys in it
    const int nkeys = (Tc + KB - 1) / KB * KB;
    {
      // (the thread's slots are re-derived from an opaque copy of the thread index in every chunk: hoisted out of the chunk
      // loop they stayed live through the key blocks and pushed the kernel 324 bytes per lane into scratch)
      int t2 = tid;
      asm volatile("" : "+v"(t2));
      const int kgrp = t2 / (16 * PIECES), ku = t2 - kgrp * (16 * PIECES);
      const int kkey0 = kgrp * 16 + ku / PIECES, kpiece = ku % PIECES;
      const bool kthread = kgrp < KG;
      const int vd0 = t2 / VCH, vch = t2 - vd0 * VCH;
      const bool vthread = vd0 < VR;
      uint2 kreg[KIT];
      uint4 vreg[VIT];
      {
        const bf16_t* kp = base + (long)(c0 + kkey0) * ld + D + kpiece * 4;
#pragma unroll
        for (int i = 0; i < KIT; ++i) {
          kreg[i] = make_uint2(0u, 0u);
          if ((ABL & 1) == 0 && kthread && kkey0 + i * 16 * KG < Tc) kreg[i] = *reinterpret_cast<const uint2*>(kp + (long)i * (16 * KG) * ld);
        }
        const bf16_t* vp = vbase + (long)vd0 * vt_ld + c0 + vch * 8;
#pragma unroll
        for (int i = 0; i < VIT; ++i) {
          vreg[i] = make_uint4(0u, 0u, 0u, 0u);
          if ((ABL & 1) == 0 && vthread && vd0 + i * VR < DH && vch * 8 < Tc)   // (row_start, c0 and the chunk are multiples of 8 keys: 16-byte aligned)
            vreg[i] = *reinterpret_cast<const uint4*>(vp + (long)i * VR * vt_ld);
        }
      }
      between();
      // keys >= T inside the last chunk of 8 that holds a valid one: exact zeros (the clip's padding rows hold arbitrary values)
      unsigned vmask[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int nv = Tc - vch * 8;   // valid keys of this thread's chunk
        vmask[e] = 2 * e + 1 < nv ? 0xffffffffu : (2 * e < nv ? 0xffffu : 0u);
      }
      if constexpr ((ABL & 2) != 0) {   // ablation: the loads are waited for, the kernel computes on zeros
#pragma unroll
        for (int e = 0; e < 4; ++e) vmask[e] = 0u;
#pragma unroll
        for (int i = 0; i < KIT; ++i) kreg[i] = make_uint2(kreg[i].x & vmask[0], kreg[i].y & vmask[1]);
      }
      if (ch > 0) __syncthreads();   // every wave has finished the previous chunk's blocks
      if (kthread) {
        unsigned char* kd = Ks + kkey0 * 128 + (((kpiece >> 1) ^ ((kkey0 >> 1) & 7)) * 16 + (kpiece & 1) * 8);
#pragma unroll
        for (int i = 0; i < KIT; ++i)
          if (kkey0 + i * 16 * KG < nkeys) *reinterpret_cast<uint2*>(kd + i * (16 * KG) * 128) = kreg[i];
      }
      if (vthread && vch * 8 < nkeys) {
        bf16_t* vd = Vt + vd0 * VLD + vch * 8;
#pragma unroll
        for (int i = 0; i < VIT; ++i)
          if (vd0 + i * VR < DH)
            *reinterpret_cast<uint4*>(vd + i * VR * VLD) = make_uint4(vreg[i].x & vmask[0], vreg[i].y & vmask[1], vreg[i].z & vmask[2], vreg[i].w & vmask[3]);
      }
    }
  };
  // this wave's query tiles; a tile that starts at or beyond the clip's last valid frame does no work
  bf16x8 qf[EQT][2];
  int nact = 0;   // the wave's active tiles are its first `nact` (the tile index grows with qi)
  auto load_queries = [&] {
#pragma unroll
    for (int qi = 0; qi < EQT; ++qi) {
      const int tile = tile0 + qi * NW + wave;
      nact += (int)(tile * 16 < T);   // wave-uniform
      const int qrow = tile * 16 + li;
      const int qrow_ld = qrow < cm.rows ? qrow : cm.rows - 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int d = s * 32 + kg * 8;
        uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
        if (d + 4 <= DH) lo = *reinterpret_cast<const uint2*>(base + (long)qrow_ld * ld + d);
        if (d + 8 <= DH) hi = *reinterpret_cast<const uint2*>(base + (long)qrow_ld * ld + d + 4);
        uint4 t = make_uint4(lo.x, lo.y, hi.x, hi.y);
        qf[qi][s] = *reinterpret_cast<bf16x8*>(&t);
      }
    }
  };
  stage(0, load_queries);   // (before the accumulators exist: the 52 staging registers of a 10 s clip's one chunk are free then)

  // nm[qi] = MINUS the reference point of tile qi's exponent (att_key_block): 0 until the first block moves it
  f32x4 nm[EQT];
  float l_run[EQT];
  f32x4 o[EQT][4];
#pragma unroll
  for (int qi = 0; qi < EQT; ++qi) {
    nm[qi] = f32x4{0.f, 0.f, 0.f, 0.f};
    l_run[qi] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[qi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ... and compute: whole blocks first, then the one that holds the clip's last valid key
  auto compute = [&](int ch) {
    const int c0 = ch * KMAX;
    const int Tc = T - c0 < KMAX ? T - c0 : KMAX;
    const int nkb = (Tc + KB - 1) / KB, nfull = Tc / KB;

    // whole blocks first, then the one that holds the clip's last valid key
    auto run = [&](auto na_c) {
      constexpr int NA = decltype(na_c)::value;
      const uint4* K4 = reinterpret_cast<const uint4*>(Ks);
      for (int kb = 0; kb < nfull; ++kb)
        att_key_block<DH, VLD, EQT, NA, false, ABL>(K4 + kb * (KB * 8), Vt + kb * KB, ch == 0 && kb == 0, 0, li, kg, qf, nm, o, l_run);
      if (nfull < nkb)
        att_key_block<DH, VLD, EQT, NA, true, ABL>(K4 + nfull * (KB * 8), Vt + nfull * KB, ch == 0 && nfull == 0, Tc - nfull * KB, li, kg, qf, nm, o, l_run);
    };
    switch (nact) {
      case 4: if constexpr (EQT >= 4) run(std::integral_constant<int, 4>{}); break;
      case 3: if constexpr (EQT >= 3) run(std::integral_constant<int, 3>{}); break;
      case 2: if constexpr (EQT >= 2) run(std::integral_constant<int, 2>{}); break;
      case 1: run(std::integral_constant<int, 1>{}); break;
      default: break;
    }
  };
  __syncthreads();   // the first chunk's K and V^T are in LDS
  compute(0);
  if constexpr (MULTI) {
    for (int ch = 1; ch < nchunks; ++ch) {   // clips of more than KMAX frames (10.8 s)
      stage(ch, [] {});
      __syncthreads();
      compute(ch);
    }
  }

#pragma unroll
  for (int qi = 0; qi < EQT; ++qi) {
    float l;
    if constexpr (ONES_ROW) {   // row DH of O^T: tile DH / 16, lane group (DH % 16) / 4, register 0
      l = __shfl(o[qi][DH / 16][0], ((DH % 16) / 4) * 16 + li);
    } else {
      l = rows_sum(l_run[qi]);
    }
    const float inv = 1.0f / l;
    const int qrow = (tile0 + qi * NW + wave) * 16 + li;
    if (qrow < cm.rows) {
      const bool valid = qrow < T;   // padding rows of the clip are written as zeros
      bf16_t* orow = out + (long)(cm.row_start + qrow) * D + h * DH;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + kg * 4;
        if (d < DH) {
          uint2 w = make_uint2(0u, 0u);
          if (valid) {
            w.x = pack_bf16x2(o[qi][dt][0] * inv, o[qi][dt][1] * inv);
            w.y = pack_bf16x2(o[qi][dt][2] * inv, o[qi][dt][3] * inv);
          }
          *reinterpret_cast<uint2*>(orow + d) = w;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Decode self-attention: one wave per (clip, head), keys in blocks of KB staged through LDS.
//
// The cache rows of a (clip, head) are one contiguous run of S x dh bf16.  The first version had every lane fetch "its"
// key row with thirteen 8-byte loads at a 104-byte stride (64 cache lines per wave instruction) and spent its 8.6 us
// issuing ~45 such loads per wave.  Here the wave copies the K and V runs of a key block to its private LDS slab with
// LDS-DMA (`global_load_lds_dwordx4`: 1 KiB of consecutive bytes per instruction, 8 + 8 instructions for a 72-key block
// at dh = 52) and computes from LDS: scores with lane = key (row stride 26 dwords: at most 2-way bank conflicts), fp32
// softmax in the exp2 domain (online across key blocks, so any S <= Smax works; a 65-step decode is ONE block), P.V with
// lane = (key group, 4-dim piece) and a fixed-order reduction.  The output row goes to the FM activation buffer the
// o-proj GEMM reads (kernels.h fm16).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char lds_char_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(lds_char_t*)(p); }
// 16 bytes per ACTIVE lane from gsrc (per lane) to LDS at lds_base (wave-uniform) + 16 * lane.  NT: non-temporal policy.
template <bool NT = false>
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  if constexpr (NT)
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_base)
        : "memory");
  else
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_base)
        : "memory");
}

template <int DH>
struct SelfAttnCfg {
  static constexpr int KB = 72;                           // keys per block: a 10 s clip's 65 steps fit in one
  static constexpr int TILE_BYTES = KB * DH * 2;
  static constexpr int NCH = (TILE_BYTES + 1023) / 1024;  // 1 KiB DMA pieces per tile
  static constexpr int TILE_PAD = NCH * 1024;
  static constexpr int PIECES = DH / 4, G = 64 / PIECES;
};

// NT: the K / V cache stream with the non-temporal policy (see dec_self_attention)
// ABL (probe, 0 in the product): 1 = no K / V copies (the arithmetic on whatever the LDS holds), 2 = the copies alone
template <int DH, bool NT = false, int ABL = 0>
__global__ __launch_bounds__(256) void dec_self_attention_kernel(const float* __restrict__ q,
                                                                 const bf16_t* __restrict__ cacheK,
                                                                 const bf16_t* __restrict__ cacheV,
                                                                 const int* __restrict__ pos_ptr, int M, int D,
                                                                 int heads, int Smax, bf16_t* __restrict__ out) {
  using C = SelfAttnCfg<DH>;
  constexpr int KB = C::KB, NCH = C::NCH, PIECES = C::PIECES, G = C::G;
  __shared__ __attribute__((aligned(16))) unsigned char tiles[4][2][C::TILE_PAD];
  __shared__ float sc[4][KB + 8];
  __shared__ float4 red[4][G][PIECES];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + wave;
  if (pair >= M * heads) return;   // wave-uniform; the waves of a workgroup never synchronise with each other
  const int b = pair / heads, h = pair - b * heads;
  const int S = *pos_ptr + 1;
  const float* qp = q + (long)b * D + h * DH;
  const unsigned char* kp = reinterpret_cast<const unsigned char*>(cacheK + (long)pair * Smax * DH);
  const unsigned char* vp = reinterpret_cast<const unsigned char*>(cacheV + (long)pair * Smax * DH);
  const long run_bytes = (long)S * DH * 2;   // only the rows written so far are fetched (S <= Smax rows exist)
  const float c = rsqrtf((float)DH) * kLog2e;
  const unsigned kt = __builtin_amdgcn_readfirstlane(lds_addr(&tiles[wave][0][0]));
  const unsigned vt = __builtin_amdgcn_readfirstlane(lds_addr(&tiles[wave][1][0]));
  const bf16_t* Kl = reinterpret_cast<const bf16_t*>(&tiles[wave][0][0]);
  const bf16_t* Vl = reinterpret_cast<const bf16_t*>(&tiles[wave][1][0]);

  auto stage = [&](int k0) {   // keys [k0, k0 + KB) of K and V -> LDS (nothing beyond row S - 1 is read)
    const long base = (long)k0 * DH * 2;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const long off = base + i * 1024 + lane * 16;
      if (ABL != 1 && i * 1024 + lane * 16 < C::TILE_BYTES && off < run_bytes) {
        lds_dma16<NT>(kp + off, kt + i * 1024);
        lds_dma16<NT>(vp + off, vt + i * 1024);
      }
    }
  };
  stage(0);
  float qreg[DH];   // every lane holds the whole query (same address in all lanes: one request per load)
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    const float4 t = *reinterpret_cast<const float4*>(qp + d);
    qreg[d] = t.x * c; qreg[d + 1] = t.y * c; qreg[d + 2] = t.z * c; qreg[d + 3] = t.w * c;
  }
  const int g = lane / PIECES, piece = lane - g * PIECES;
  float m_run = -INFINITY, l_run = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
  for (int k0 = 0; k0 < S; k0 += KB) {
    if (k0 > 0) stage(k0);   // (the previous block's LDS reads are complete: their values were consumed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int nk = S - k0 < KB ? S - k0 : KB;
    if constexpr (ABL == 2) {
      acc.x += Kl[lane] == 0x7fc1 ? 1.f : 0.f;
      l_run = 1.f;
      continue;
    }
    // scores: lane = key (and key + 64 for the block's tail)
    float s0 = -INFINITY, s1 = -INFINITY;
    if (lane < nk) {
      const uint2* kr = reinterpret_cast<const uint2*>(Kl + lane * DH);
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < DH; d += 4) {
        const uint2 u = kr[d >> 2];
        a += qreg[d] * bf_lo(u.x) + qreg[d + 1] * bf_hi(u.x) + qreg[d + 2] * bf_lo(u.y) + qreg[d + 3] * bf_hi(u.y);
      }
      s0 = a;
    }
    if (lane + 64 < nk) {
      const uint2* kr = reinterpret_cast<const uint2*>(Kl + (lane + 64) * DH);
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < DH; d += 4) {
        const uint2 u = kr[d >> 2];
        a += qreg[d] * bf_lo(u.x) + qreg[d + 1] * bf_hi(u.x) + qreg[d + 2] * bf_lo(u.y) + qreg[d + 3] * bf_hi(u.y);
      }
      s1 = a;
    }
    const float m_new = fmaxf(m_run, wave_max_dpp(fmaxf(s0, s1)));
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first block: exp2(-inf) = 0
    m_run = m_new;
    const float p0 = __builtin_amdgcn_exp2f(s0 - m_new), p1 = __builtin_amdgcn_exp2f(s1 - m_new);   // masked keys: 0
    sc[wave][lane] = p0;
    if (lane < KB - 64) sc[wave][lane + 64] = p1;
    l_run = l_run * alpha + wave_sum_dpp(p0 + p1);
    __builtin_amdgcn_wave_barrier();
    // P.V: lane = (key group g, 4-dim piece); group g walks keys g, g + G, ...
    acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
    if (g < G) {
#pragma unroll 6
      for (int s = g; s < nk; s += G) {
        const uint2 u = *reinterpret_cast<const uint2*>(Vl + s * DH + piece * 4);
        const float p = sc[wave][s];
        acc.x += p * bf_lo(u.x);
        acc.y += p * bf_hi(u.x);
        acc.z += p * bf_lo(u.y);
        acc.w += p * bf_hi(u.y);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (g < G) red[wave][g][piece] = acc;
  __builtin_amdgcn_wave_barrier();
  if (lane < PIECES) {
    float4 t = red[wave][0][lane];
#pragma unroll
    for (int k = 1; k < G; ++k) {
      const float4 r = red[wave][k][lane];
      t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
    }
    const float inv = 1.0f / l_run;
    uint2 o;
    o.x = pack_bf16x2(t.x * inv, t.y * inv);
    o.y = pack_bf16x2(t.z * inv, t.w * inv);
    *reinterpret_cast<uint2*>(out + fm16(b, h * DH + lane * 4, D >> 5)) = o;   // FM: the o-proj GEMM's A operand
  }
}

// ------------------------------------------------------------------------------------------------
// Decode cross-attention: one workgroup per (clip, head); its 4 waves split the head dim.
//
// K^T / V^T are [dh][Tk] bf16 (keys contiguous).  Lane l owns keys 8l..8l+7 of a 512-key chunk and
// streams 16 B per row; wave w covers rows d in [w*dh/4, (w+1)*dh/4): partial q.k sums are combined
// through LDS (fixed order), every wave then runs the same fp32 online softmax and accumulates its own
// dh/4 output dims, reduced across lanes at the end.  Each wave issues 2*dh/4 independent 16-B loads per
// chunk -- short enough to be latency-friendly at batch 1 and, with 4x more waves in flight than a
// one-wave-per-head layout, enough bytes in flight to stream at HBM rate at batch 256.
// ------------------------------------------------------------------------------------------------
// FUSEQ: the query projection runs here too.  `q` is then the fp32 residual stream H [M][D] and Wq the cross-q
// weight [D][D] with the LayerNorm scale folded in (the layout of dec_gemm_ln_f32); each wave normalises the clip's
// row (lane l owns k = 8l..8l+7) and reduces its own dh/4 query dims across lanes -- 13 extra 16-byte loads and a
// few dozen shuffles that sit under the latency of the first K loads, instead of a separate 7 us GEMM launch.
// Only for small batches: every (clip, head) workgroup re-reads its 43 KB slice of Wq, which at batch 256 adds
// 88 MB of L2 traffic per layer next to the 177 MB K/V stream and costs more (31 -> 48 us) than the GEMM saved.
// FP8: K^T / V^T are e4m3 bytes (engine option kv_dtype = fp8: half the bytes of the kernel that bounds a decode step),
// written by the cross-KV GEMM as value * qscale[column] with a per-column scale fixed at load (gemm_common.h
// EpiCrossKVFp8).  Same lane / key mapping with 8-byte loads; the K scale is folded into the query (kdq[d] = 1 / qscale of
// K row d), the V scale into the output (vdq).  Scores, softmax and accumulation stay fp32.
template <int DH, bool FUSEQ, bool FP8 = false>
__global__ __launch_bounds__(256, FP8 ? MSH_FP8_ATT_OCC : 2) void dec_cross_attention_kernel(const float* __restrict__ q,
                                                                  const bf16_t* __restrict__ Wq,
                                                                  const bf16_t* __restrict__ KT,
                                                                  const bf16_t* __restrict__ VT,
                                                                  const ClipMeta* __restrict__ clips, int D,
                                                                  int heads, bf16_t* __restrict__ out,
                                                                  const float* __restrict__ kdq = nullptr,
                                                                  const float* __restrict__ vdq = nullptr) {
  constexpr int DQ = DH / 4;
  constexpr int EB = FP8 ? 1 : 2;   // bytes per stored key
  __shared__ float sp[4][512];
  __shared__ float red[4][DQ][65];   // (a butterfly reduction of the tail instead of this slab measured 1 us SLOWER at bf16)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const ClipMeta cm = clips[b];
  const int T = cm.T, Tk = cm.Tk;
  const float* qp = q + (long)b * D + h * DH + wave * DQ;
  const long off = (long)cm.kv_start * D + (long)(h * DH + wave * DQ) * Tk;
  // buffer descriptors over this wave's dh/4 rows: row d sits at scalar offset d*Tk*2, the lane's keys at
  // one shared 32-bit vector offset -> no per-row 64-bit addresses in VGPRs; reads past the last row return 0
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)KT + off * EB), 0, DQ * Tk * EB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)VT + off * EB), 0, DQ * Tk * EB, 0x00020000);
  const float c = rsqrtf((float)DH) * kLog2e;

  // FUSEQ (the single-clip latency path): the first chunk's K rows are requested BEFORE the query is formed -- they do not
  // depend on it, and behind the LayerNorm / projection chain their round trip was a third one in a row
  u32x4 kr0[(FUSEQ && !FP8) ? DQ : 1];
  if constexpr (FUSEQ && !FP8) {
#pragma unroll
    for (int d = 0; d < DQ; ++d) kr0[d] = __builtin_amdgcn_raw_buffer_load_b128(rk, lane * 16, d * Tk * 2, kStreamAux);
  }
  float qd[DQ];
  if constexpr (FUSEQ) {
    const int k0q = lane * 8;
    const bool act = k0q < D;  // D / 8 lanes hold the row
    float xv[8];
    if (act) {   // the residual stream is FM (kernels.h fm32): columns k0q..k0q+3 and k0q+4..k0q+7 are two float4 halves
      const float4 a = *reinterpret_cast<const float4*>(q + fm32(b, k0q, D >> 5));
      const float4 c4 = *reinterpret_cast<const float4*>(q + fm32(b, k0q + 4, D >> 5));
      xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w; xv[4] = c4.x; xv[5] = c4.y; xv[6] = c4.z; xv[7] = c4.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = 0.f;
    }
    u32x4 wr[DQ];
    const bf16_t* wrow = Wq + (long)(h * DH + wave * DQ) * D + k0q;
#pragma unroll
    for (int d = 0; d < DQ; ++d)
      wr[d] = act ? *reinterpret_cast<const u32x4*>(wrow + (long)d * D) : u32x4{0u, 0u, 0u, 0u};
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += xv[e];
    const float mean = wave_sum_dpp(sum) / (float)D;
    float sq = 0.f;
    if (act) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xv[e] -= mean;
        sq += xv[e] * xv[e];
      }
    }
    const float rstd = rsqrtf(wave_sum_dpp(sq) / (float)D + 1e-5f);
    // the decode GEMM this replaces rounds the normalised row to bf16 before the MFMA: do the same, so the two
    // paths differ only in summation order
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const uint32_t pk = pack_bf16x2(xv[e] * rstd, xv[e + 1] * rstd);
      xv[e] = bf_lo(pk);
      xv[e + 1] = bf_hi(pk);
    }
#pragma unroll
    for (int d = 0; d < DQ; ++d) {
      const u32x4 u = wr[d];
      const float part = xv[0] * bf_lo(u.x) + xv[1] * bf_hi(u.x) + xv[2] * bf_lo(u.y) + xv[3] * bf_hi(u.y) +
                         xv[4] * bf_lo(u.z) + xv[5] * bf_hi(u.z) + xv[6] * bf_lo(u.w) + xv[7] * bf_hi(u.w);
      qd[d] = wave_sum_dpp(part);   // (DPP instead of thirteen ds_bpermute butterflies, and the K request above: 8.2 -> 7.65 us per launch)
    }
  } else {
#pragma unroll
    for (int d = 0; d < DQ; ++d) qd[d] = qp[d];
  }
  // (fp8: the K scales are multiplied into q only BEHIND the first chunk's K / V requests: as a statement up here the
  // multiply made the wave wait for q and the scales before it issued a single K / V load -- one more memory round trip)
  float kq[FP8 ? DQ : 1];
  if constexpr (FP8) {
#pragma unroll
    for (int d = 0; d < DQ; ++d) kq[d] = kdq[h * DH + wave * DQ + d];
  }
  float opart[DQ];
#pragma unroll
  for (int d = 0; d < DQ; ++d) opart[d] = 0.f;
  float m_run = -INFINITY, l_part = 0.f;

#pragma unroll 1
  for (int k0 = 0; k0 < Tk; k0 += 512) {
    const int key = k0 + lane * 8;
    const bool in = key < Tk;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    u32x4 vr[DQ];     // bf16: 8 keys in 4 dwords; fp8: 8 keys in .x / .y
    if constexpr (FP8) {
      // K and V rows are requested together: at one byte per key both fit the register file (2 x 26 VGPRs), and the
      // workgroup sits through ONE memory latency per chunk instead of two
      u32x2 kr[DQ];
#pragma unroll
      for (int d = 0; d < DQ; ++d) kr[d] = __builtin_amdgcn_raw_buffer_load_b64(rk, key, d * Tk, kStreamAux);
#pragma unroll
      for (int d = 0; d < DQ; ++d) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rv, key, d * Tk, kStreamAux);
        vr[d] = u32x4{t.x, t.y, 0u, 0u};
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k0 == 0) {
#pragma unroll
        for (int d = 0; d < DQ; ++d) qd[d] *= kq[d];
      }
#pragma unroll
      for (int d = 0; d < DQ; ++d) {
        const f32x2 k01 = __builtin_amdgcn_cvt_pk_f32_fp8(kr[d].x, false), k23 = __builtin_amdgcn_cvt_pk_f32_fp8(kr[d].x, true);
        const f32x2 k45 = __builtin_amdgcn_cvt_pk_f32_fp8(kr[d].y, false), k67 = __builtin_amdgcn_cvt_pk_f32_fp8(kr[d].y, true);
        s[0] += qd[d] * k01[0]; s[1] += qd[d] * k01[1];
        s[2] += qd[d] * k23[0]; s[3] += qd[d] * k23[1];
        s[4] += qd[d] * k45[0]; s[5] += qd[d] * k45[1];
        s[6] += qd[d] * k67[0]; s[7] += qd[d] * k67[1];
      }
    } else {
      u32x4 kr[DQ];
      if constexpr (FUSEQ) {
        if (k0 == 0) {
#pragma unroll
          for (int d = 0; d < DQ; ++d) kr[d] = kr0[d];
        } else {
#pragma unroll
          for (int d = 0; d < DQ; ++d) kr[d] = __builtin_amdgcn_raw_buffer_load_b128(rk, key * 2, d * Tk * 2, kStreamAux);
        }
      } else {
#pragma unroll
        for (int d = 0; d < DQ; ++d) kr[d] = __builtin_amdgcn_raw_buffer_load_b128(rk, key * 2, d * Tk * 2, kStreamAux);
      }
#pragma unroll
      for (int d = 0; d < DQ; ++d) {
        const u32x4 u = kr[d];
        s[0] += qd[d] * bf_lo(u.x); s[1] += qd[d] * bf_hi(u.x);
        s[2] += qd[d] * bf_lo(u.y); s[3] += qd[d] * bf_hi(u.y);
        s[4] += qd[d] * bf_lo(u.z); s[5] += qd[d] * bf_hi(u.z);
        s[6] += qd[d] * bf_lo(u.w); s[7] += qd[d] * bf_hi(u.w);
      }
      // the V rows are requested only now (K registers are dead): they fly during the score exchange
      __builtin_amdgcn_sched_barrier(0);
      // (requesting the first chunk's V rows ahead as well -- at the top or behind the projection -- costs a spilled
      // register at two workgroups per CU and measured SLOWER: 8.3 against 7.65 us for the single-clip launch)
#pragma unroll
      for (int d = 0; d < DQ; ++d) vr[d] = __builtin_amdgcn_raw_buffer_load_b128(rv, key * 2, d * Tk * 2, kStreamAux);
    }
    if (k0 > 0) __syncthreads();  // previous chunk's partial scores fully consumed
    *reinterpret_cast<float4*>(&sp[wave][lane * 8]) = make_float4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<float4*>(&sp[wave][lane * 8 + 4]) = make_float4(s[4], s[5], s[6], s[7]);
    __syncthreads();
    float mloc = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = lane * 8 + e;
      const float t = (sp[0][i] + sp[1][i]) + (sp[2][i] + sp[3][i]);
      s[e] = (in && key + e < T) ? t * c : -INFINITY;
      mloc = fmaxf(mloc, s[e]);
    }
    mloc = wave_max(mloc);
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] = __builtin_amdgcn_exp2f(s[e] - m_new);
      psum += s[e];
    }
    l_part = l_part * alpha + psum;
    {
#pragma unroll
      for (int d = 0; d < DQ; ++d) {
        const u32x4 u = vr[d];
        float part;
        if constexpr (FP8) {
          const f32x2 v01 = __builtin_amdgcn_cvt_pk_f32_fp8(u.x, false), v23 = __builtin_amdgcn_cvt_pk_f32_fp8(u.x, true);
          const f32x2 v45 = __builtin_amdgcn_cvt_pk_f32_fp8(u.y, false), v67 = __builtin_amdgcn_cvt_pk_f32_fp8(u.y, true);
          part = s[0] * v01[0] + s[1] * v01[1] + s[2] * v23[0] + s[3] * v23[1] + s[4] * v45[0] + s[5] * v45[1] + s[6] * v67[0] + s[7] * v67[1];
        } else {
          part = s[0] * bf_lo(u.x) + s[1] * bf_hi(u.x) + s[2] * bf_lo(u.y) + s[3] * bf_hi(u.y) +
                 s[4] * bf_lo(u.z) + s[5] * bf_hi(u.z) + s[6] * bf_lo(u.w) + s[7] * bf_hi(u.w);
        }
        opart[d] = opart[d] * alpha + part;
      }
    }
  }
  const float l = wave_sum(l_part);
#pragma unroll
  for (int d = 0; d < DQ; ++d) red[wave][d][lane] = opart[d];
  __builtin_amdgcn_wave_barrier();
  if (lane < DQ) {
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) acc += red[wave][lane][i];
    if constexpr (FP8) acc *= vdq[h * DH + wave * DQ + lane];
    out[fm16(b, h * DH + wave * DQ + lane, D >> 5)] = f32_to_bf16(acc / l);   // FM: the o-proj GEMM's A operand
  }
}

// ------------------------------------------------------------------------------------------------
// Cross-attention PROBABILITIES of one decode step, for word timestamps (the `cross_attentions.{l}` outputs of the
// reference's attention-exporting decoder, core/moonshine-model.cpp:480-500): one workgroup per (clip, head), plain
// two-pass softmax over the clip's T valid frames, written to out[clip][layer][head][pos][0..T).  Runs only when the
// capture is switched on; the regular kernel above never materialises the probabilities.
// ------------------------------------------------------------------------------------------------
constexpr int PROBS_TMAX = 2048;
__global__ __launch_bounds__(256) void dec_cross_probs_kernel(const float* __restrict__ q, const bf16_t* __restrict__ KT,
                                                              const ClipMeta* __restrict__ clips,
                                                              const int* __restrict__ pos_ptr, int D, int heads, int dh,
                                                              int layers, int layer, int Smax, int Tcap,
                                                              float* __restrict__ out) {
  __shared__ float sc[PROBS_TMAX];
  __shared__ float red[4];
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const ClipMeta cm = clips[b];
  const int T = cm.T < PROBS_TMAX ? cm.T : PROBS_TMAX, Tk = cm.Tk;
  const int tid = threadIdx.x, wave = tid >> 6;
  const float* qp = q + (long)b * D + h * dh;
  const bf16_t* kp = KT + (long)cm.kv_start * D + (long)(h * dh) * Tk;
  const float scale = rsqrtf((float)dh);
  float mloc = -INFINITY;
  for (int key = tid; key < T; key += 256) {
    float acc = 0.f;
    for (int d = 0; d < dh; ++d) acc += qp[d] * bf16_to_f32(kp[(long)d * Tk + key]);
    acc *= scale;
    sc[key] = acc;
    mloc = fmaxf(mloc, acc);
  }
  mloc = wave_max(mloc);
  if ((tid & 63) == 0) red[wave] = mloc;
  __syncthreads();
  const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float lsum = 0.f;
  for (int key = tid; key < T; key += 256) {
    const float p = expf(sc[key] - mx);
    sc[key] = p;
    lsum += p;
  }
  lsum = wave_sum(lsum);
  if ((tid & 63) == 0) red[wave] = lsum;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  float* o = out + ((((long)b * layers + layer) * heads + h) * Smax + *pos_ptr) * Tcap;
  for (int key = tid; key < T; key += 256) o[key] = sc[key] * inv;
}

}  // namespace

void dec_cross_attention_probs(const float* q, const bf16_t* KT, const ClipMeta* clips, const int* pos_ptr, int M, int D,
                               int heads, int layers, int layer, int Smax, int Tcap, float* out, hipStream_t s) {
  MSH_LAUNCH(dec_cross_probs_kernel, dim3(M * heads), dim3(256), 0, s, q, KT, clips, pos_ptr, D, heads, D / heads,
                     layers, layer, Smax, Tcap, out);
}

// (query blocks, heads, clips) -> the 1-D grid of enc_attention_res_kernel, padded to a multiple of 8 (its XCD-aware item order)
static inline dim3 att_grid1(dim3 g) { return dim3((g.x * g.y * g.z + 7) / 8 * 8); }

void enc_attention(const bf16_t* qk, const bf16_t* vt, long vt_ld, bf16_t* out, const ClipMeta* clips, int n_clips, int max_rows,
                   int D, int heads, hipStream_t s) {
  const int dh = D / heads;
  // One 8-wave workgroup per (clip, head, 512 queries): a 10 s clip (415 frames = 26 tiles of 16 queries) is one workgroup
  // and one chunk of keys; longer clips get more workgroups along x, each walking all keys in chunks of 448.
  if (n_clips > 65535) throw std::runtime_error("enc_attention: more than 65535 clips in one batch");
  const int ntiles = (max_rows + 15) / 16;
  dim3 grid((ntiles + 31) / 32, heads, n_clips);
  const bool multi = max_rows > 448;   // (rows >= frames: no clip of the batch has more than one chunk of keys otherwise)
  // A few clips (the latency case): (clip, head) workgroups are 8 per clip on 256 CUs, each walking 26 query tiles.  One tile per
  // wave instead of four gives four workgroups per (clip, head) -- each stages the clip's keys and values again, which nobody
  // else needs the CU for -- and the same bits (a query tile's arithmetic does not depend on its neighbours): 18.6 -> 11.7 us per
  // launch (event scope) at one 10 s clip, encode 0.655 -> 0.60 ms.  MSH_ENC_ATT_FEW=0: off.
  static const bool few_off = [] {
    const char* e = dev_getenv("MSH_ENC_ATT_FEW");
    return e != nullptr && e[0] == '0';
  }();
  if (!few_off && dh == 52 && (long)n_clips * heads * ((ntiles + 31) / 32) <= 64) {
    dim3 g1((ntiles + 7) / 8, heads, n_clips);
    if (multi) MSH_LAUNCH((enc_attention_res_kernel<52, 448, 0, 8, 1, true>), att_grid1(g1), dim3(512), 0, s, qk, vt, vt_ld, out, clips, (int)g1.z, (int)g1.x, D);
    else MSH_LAUNCH((enc_attention_res_kernel<52, 448, 0, 8, 1, false>), att_grid1(g1), dim3(512), 0, s, qk, vt, vt_ld, out, clips, (int)g1.z, (int)g1.x, D);
    return;
  }
#define MSH_EATT(DHV)                                                                                                                  \
  case DHV:                                                                                                                            \
    if (multi) MSH_LAUNCH((enc_attention_res_kernel<DHV, 448, 0, 8, 4, true>), att_grid1(grid), dim3(512), 0, s, qk, vt, vt_ld, out, clips, (int)grid.z, (int)grid.x, D);   \
    else MSH_LAUNCH((enc_attention_res_kernel<DHV, 448>), att_grid1(grid), dim3(512), 0, s, qk, vt, vt_ld, out, clips, (int)grid.z, (int)grid.x, D);                        \
    break
  switch (dh) {
    MSH_EATT(52);
    MSH_EATT(36);
    MSH_EATT(16);
    default: throw std::runtime_error("enc_attention: unsupported head_dim " + std::to_string(dh));
  }
#undef MSH_EATT
}

// Microbenchmark / test hook (include/moonshine_hip_dev.h msh_test_enc_attention): n_clips clips of T frames at width D on
// uniform random q | k / V^T; variant 0 = the product kernel (8 waves x 4 tiles, clips of up to 448 frames), 1 = its
// instantiation with the chunk loop (any length), 50 / 51 = 12 waves x 3 tiles / 16 waves x 2 tiles, 100 + abl = the product
// shape with ablation bits `abl`.  Returns ms per launch; out (nullable) [R][D] receives the
// last launch's output as bf16 bit patterns.
float enc_attention_microbench(int variant, int n_clips, int T, int D, int heads, int iters, uint16_t* out_host) {
  const int dh = D / heads;
  if (dh != 52 || T < 1 || n_clips < 1 || heads * dh != D) throw std::runtime_error("enc_attention_microbench: head_dim 52 only");
  const int rows = (T + 7) / 8 * 8;
  const long R = (long)rows * n_clips, vt_ld = (R + 127) / 128 * 128;
  std::vector<ClipMeta> cm(n_clips);
  for (int b = 0; b < n_clips; ++b) {
    cm[b] = ClipMeta{};
    cm[b].row_start = b * rows;
    cm[b].rows = rows;
    cm[b].T = T;
  }
  std::vector<bf16_t> qk((size_t)R * 2 * D), vtv((size_t)D * vt_ld);
  unsigned x = 2463534242u;
  auto rnd = [&] {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
  };
  // q | k uniform in [-2, 2); the q half pre-scaled by rsqrt(dh) * log2(e) as the engine's weights deliver it
  const float qscale = 1.0f / sqrtf((float)dh) * kLog2e;
  for (size_t i = 0; i < qk.size(); ++i) qk[i] = f32_to_bf16(rnd() * 2.0f * ((int)(i % (size_t)(2 * D)) < D ? qscale : 1.0f));
  for (auto& v : vtv) v = f32_to_bf16(rnd());
  bf16_t *QK = nullptr, *VT = nullptr, *O = nullptr;
  ClipMeta* CM = nullptr;
  MSH_HIP(hipMalloc(&QK, qk.size() * 2));
  MSH_HIP(hipMalloc(&VT, vtv.size() * 2));
  MSH_HIP(hipMalloc(&O, (size_t)R * D * 2));
  MSH_HIP(hipMalloc(&CM, cm.size() * sizeof(ClipMeta)));
  MSH_HIP(hipMemcpy(QK, qk.data(), qk.size() * 2, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(VT, vtv.data(), vtv.size() * 2, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(CM, cm.data(), cm.size() * sizeof(ClipMeta), hipMemcpyHostToDevice));
  MSH_HIP(hipMemset(O, 0xff, (size_t)R * D * 2));
  const int ntiles = (rows + 15) / 16;
  auto run = [&] {
    dim3 grid((ntiles + 31) / 32, heads, n_clips);
#define MSH_RES(A)                                                                                                      \
  case A:                                                                                                               \
    MSH_LAUNCH((enc_attention_res_kernel<52, 448, A>), att_grid1(grid), dim3(512), 0, (hipStream_t)0, QK, VT, vt_ld, O, CM, (int)grid.z, (int)grid.x, D);   \
    break
    if (variant == 50) {
      dim3 g2((ntiles + 35) / 36, heads, n_clips);
      MSH_LAUNCH((enc_attention_res_kernel<52, 448, 0, 12, 3>), att_grid1(g2), dim3(768), 0, (hipStream_t)0, QK, VT, vt_ld, O, CM, (int)g2.z, (int)g2.x, D);
      return;
    }
    if (variant == 51) {
      MSH_LAUNCH((enc_attention_res_kernel<52, 448, 0, 16, 2>), att_grid1(grid), dim3(1024), 0, (hipStream_t)0, QK, VT, vt_ld, O, CM, (int)grid.z, (int)grid.x, D);
      return;
    }
    if (variant == 1) {   // the instantiation with the chunk loop (any clip length)
      MSH_LAUNCH((enc_attention_res_kernel<52, 448, 0, 8, 4, true>), att_grid1(grid), dim3(512), 0, (hipStream_t)0, QK, VT, vt_ld, O, CM, (int)grid.z, (int)grid.x, D);
      return;
    }
    if (variant != 1 && rows > 448) throw std::runtime_error("enc_attention_microbench: only variant 1 walks more than 448 keys");
    switch (variant >= 100 ? variant - 100 : variant) {
      MSH_RES(0);
      MSH_RES(1);
      MSH_RES(2);
      MSH_RES(4);
      MSH_RES(8);
      MSH_RES(12);
      default: throw std::runtime_error("enc_attention_microbench: variant / ablation not compiled");
    }
#undef MSH_RES
  };
  run();
  MSH_HIP(hipDeviceSynchronize());
  float ms = 0.f;
  if (iters > 0) {
    hipEvent_t e0, e1;
    MSH_HIP(hipEventCreate(&e0));
    MSH_HIP(hipEventCreate(&e1));
    MSH_HIP(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) run();
    MSH_HIP(hipEventRecord(e1, 0));
    MSH_HIP(hipEventSynchronize(e1));
    MSH_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ms /= iters;
  }
  if (out_host != nullptr) MSH_HIP(hipMemcpy(out_host, O, (size_t)R * D * 2, hipMemcpyDeviceToHost));
  (void)hipFree(QK);
  (void)hipFree(VT);
  (void)hipFree(O);
  (void)hipFree(CM);
  return ms;
}

void dec_self_attention(const float* q, const bf16_t* cacheK, const bf16_t* cacheV, const int* pos_ptr, int M, int D,
                        int heads, int Smax, bf16_t* out, hipStream_t s) {
  const int dh = D / heads;
  dim3 grid((M * heads + 3) / 4);
  // the K / V cache stream goes with the non-temporal policy (MSH_SELF_NT=0: default policy): a decode step walks ~100 MB of
  // weights, the caches (up to 226 MB at 256 clips) and the encoder output through a 256 MB memory-side cache, and the
  // caches are the part nobody reads again before the next step -- keeping them from displacing the weights makes a layer's
  // kernels IN SEQUENCE 57 instead of 62 us (tools/chain_masks.py), the serial batch 5 % faster
  static const bool nt = [] {
    const char* e = dev_getenv("MSH_SELF_NT");
    return !(e != nullptr && e[0] == '0');
  }();
  static const int abl = [] {
    const char* e = dev_getenv("MSH_SELF_ABL");
    return e != nullptr ? atoi(e) : 0;
  }();
  if (abl != 0 && dh == 52) {
    if (abl == 1) MSH_LAUNCH((dec_self_attention_kernel<52, true, 1>), grid, dim3(256), 0, s, q, cacheK, cacheV, pos_ptr, M, D, heads, Smax, out);
    else MSH_LAUNCH((dec_self_attention_kernel<52, true, 2>), grid, dim3(256), 0, s, q, cacheK, cacheV, pos_ptr, M, D, heads, Smax, out);
    return;
  }
#define MSH_SELF(DHV)                                                                                                              \
  case DHV:                                                                                                                        \
    if (nt)                                                                                                                        \
      MSH_LAUNCH((dec_self_attention_kernel<DHV, true>), grid, dim3(256), 0, s, q, cacheK, cacheV, pos_ptr, M, D, heads, Smax, out); \
    else                                                                                                                           \
      MSH_LAUNCH((dec_self_attention_kernel<DHV, false>), grid, dim3(256), 0, s, q, cacheK, cacheV, pos_ptr, M, D, heads, Smax, out); \
    break
  switch (dh) {
    MSH_SELF(52);
    MSH_SELF(36);
    MSH_SELF(16);
    default: throw std::runtime_error("dec_self_attention: unsupported head_dim " + std::to_string(dh));
  }
#undef MSH_SELF
}

void dec_cross_attention(const float* q, const bf16_t* KT, const bf16_t* VT, const ClipMeta* clips, int M, int D,
                         int heads, bf16_t* out, hipStream_t s, const float* kdq, const float* vdq) {
  const int dh = D / heads;
  dim3 grid(M * heads);
  if (kdq != nullptr) {   // fp8 K^T / V^T
    switch (dh) {
      case 52: MSH_LAUNCH((dec_cross_attention_kernel<52, false, true>), grid, dim3(256), 0, s, q, nullptr, KT, VT, clips, D, heads, out, kdq, vdq); break;
      case 36: MSH_LAUNCH((dec_cross_attention_kernel<36, false, true>), grid, dim3(256), 0, s, q, nullptr, KT, VT, clips, D, heads, out, kdq, vdq); break;
      case 16: MSH_LAUNCH((dec_cross_attention_kernel<16, false, true>), grid, dim3(256), 0, s, q, nullptr, KT, VT, clips, D, heads, out, kdq, vdq); break;
      default: throw std::runtime_error("dec_cross_attention: unsupported head_dim");
    }
    return;
  }
  switch (dh) {
    case 52: MSH_LAUNCH((dec_cross_attention_kernel<52, false>), grid, dim3(256), 0, s, q, nullptr, KT, VT, clips, D, heads, out); break;
    case 36: MSH_LAUNCH((dec_cross_attention_kernel<36, false>), grid, dim3(256), 0, s, q, nullptr, KT, VT, clips, D, heads, out); break;
    case 16: MSH_LAUNCH((dec_cross_attention_kernel<16, false>), grid, dim3(256), 0, s, q, nullptr, KT, VT, clips, D, heads, out); break;
    default: throw std::runtime_error("dec_cross_attention: unsupported head_dim");
  }
}

void dec_cross_attention_fused_q(const float* H, const bf16_t* Wq, const bf16_t* KT, const bf16_t* VT,
                                 const ClipMeta* clips, int M, int D, int heads, bf16_t* out, hipStream_t s, const float* kdq,
                                 const float* vdq) {
  const int dh = D / heads;
  if (D > 512 || (D & 7) != 0) throw std::runtime_error("dec_cross_attention_fused_q: unsupported width");
  dim3 grid(M * heads);
  if (kdq != nullptr) {   // fp8 K^T / V^T
    switch (dh) {
      case 52: MSH_LAUNCH((dec_cross_attention_kernel<52, true, true>), grid, dim3(256), 0, s, H, Wq, KT, VT, clips, D, heads, out, kdq, vdq); break;
      case 36: MSH_LAUNCH((dec_cross_attention_kernel<36, true, true>), grid, dim3(256), 0, s, H, Wq, KT, VT, clips, D, heads, out, kdq, vdq); break;
      case 16: MSH_LAUNCH((dec_cross_attention_kernel<16, true, true>), grid, dim3(256), 0, s, H, Wq, KT, VT, clips, D, heads, out, kdq, vdq); break;
      default: throw std::runtime_error("dec_cross_attention: unsupported head_dim");
    }
    return;
  }
  switch (dh) {
    case 52: MSH_LAUNCH((dec_cross_attention_kernel<52, true>), grid, dim3(256), 0, s, H, Wq, KT, VT, clips, D, heads, out); break;
    case 36: MSH_LAUNCH((dec_cross_attention_kernel<36, true>), grid, dim3(256), 0, s, H, Wq, KT, VT, clips, D, heads, out); break;
    case 16: MSH_LAUNCH((dec_cross_attention_kernel<16, true>), grid, dim3(256), 0, s, H, Wq, KT, VT, clips, D, heads, out); break;
    default: throw std::runtime_error("dec_cross_attention: unsupported head_dim");
  }
}

}  // namespace msh
