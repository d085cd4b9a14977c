// The 32 x D fp32 row block of a wave, delivered in the MFMA operand pattern -- lane (mrow, hh) gets columns
// 16 s + 8 hh + 0..7 of row mrow for every k-step s -- through LDS.
//
// Read straight from global memory that pattern is one row per lane: every wave-level load touches 64 different 128-byte
// lines, and the CU's vector memory pipe retires about one line per clock -- the LayerNorm prologues of the panel kernels
// (k_mlp.hip, k_panel.hip: two passes over the block) cost 0.03-0.07 ms per 256 panels that way, as much as their MFMAs.
// Here the block is fetched by global_load_lds_dwordx4 with lanes running ALONG the rows (a row slice of SL k-steps is
// SL * 64 contiguous bytes: ~9 lines per instruction) into a wave-private LDS region, [row][SL * 64 + 16 bytes] -- the
// 16-byte pad puts the lanes' 16-byte reads on distinct banks -- and read back per lane.  Slices are double-buffered:
// slice t + 1 is in flight while slice t is consumed; no workgroup barrier is involved (the region belongs to the wave).
#pragma once

#include <utility>

#include "gemm_common.h"

namespace msh {
namespace {

template <int D, int SL>
struct RowsViaLds {
  static constexpr int KS = D / 16;
  static constexpr int PPR = SL * 4 + 1;                 // 16-byte pieces per row in LDS (the last one is padding)
  static constexpr int ROWB = PPR * 16;
  static constexpr int HALF = 32 * ROWB;                 // bytes of one slice buffer
  static constexpr int BYTES = 2 * HALF;                 // per wave
  static constexpr int NI = (32 * PPR + 63) / 64;        // DMA instructions per slice
  static constexpr int NS = (KS + SL - 1) / SL;          // slices
  static_assert(D % 16 == 0, "width must be a multiple of 16");

  // rows row0 .. row0 + 31 of H [R][D] (rows past R read row R - 1 again).  region_off / region: the wave's LDS region as
  // byte offset and as pointer.  f(std::integral_constant<int, s>, float4 lo, float4 hi) is called for s = 0 .. KS - 1 in
  // order with columns 16 s + 8 hh + 0..3 / + 4..7 of the lane's row.  Other vector-memory operations of the wave that are
  // still in flight when this is called are waited for as well (loads retire in order).
  template <class F>
  static __device__ __forceinline__ void run(const float* __restrict__ H, int row0, int R, unsigned region_off,
                                             const unsigned char* region, int lane, F&& f) {
    const int mrow = lane & 31, hh = lane >> 5;
    auto issue = [&](int t, int half) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int g = i * 64 + lane;
        const int r = g / PPR, p = g - r * PPR;
        int row = row0 + r;
        row = row < R ? row : R - 1;
        int c4 = t * SL * 4 + (p < SL * 4 ? p : 0);      // column / 4; the pad piece re-fetches the row's first piece
        c4 = c4 < D / 4 ? c4 : D / 4 - 1;                // (the last slice may be short)
        if (g < 32 * PPR) dma16(H + (long)row * D + c4 * 4, region_off + (unsigned)(half * HALF + i * 1024));
      }
    };
    issue(0, 0);
    static_for_rows<NS>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if constexpr (t + 1 < NS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slice t - 1's reads are done before its buffer is refilled
        issue(t + 1, (t + 1) & 1);
        wait_vmcnt<NI>();                                    // slice t is in (the NI younger requests are slice t + 1)
      } else {
        wait_vmcnt<0>();
      }
      const unsigned char* base = region + (t & 1) * HALF + mrow * ROWB + hh * 32;
      static_for_rows<SL>([&](auto sc) {
        constexpr int s = t * SL + decltype(sc)::value;
        if constexpr (s < KS) {
          const float4* p = reinterpret_cast<const float4*>(base + decltype(sc)::value * 64);
          f(std::integral_constant<int, s>{}, p[0], p[1]);
        }
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  template <class Body, int... I>
  static __device__ __forceinline__ void sfr_impl(Body&& body, std::integer_sequence<int, I...>) {
    (body(std::integral_constant<int, I>{}), ...);
  }
  template <int N, class Body>
  static __device__ __forceinline__ void static_for_rows(Body&& body) {
    sfr_impl(static_cast<Body&&>(body), std::make_integer_sequence<int, N>{});
  }
};

}  // namespace
}  // namespace msh
