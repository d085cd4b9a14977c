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
// global_load_lds_dwordx4 into a wave-private 3-slot LDS ring (rows at a pitch of 2D + 16 bytes), and keeps its own online-softmax state -- no barrier inside the loop.  Per tile:
//   S^T [16 keys x 16]   = E_tile (A: rows = keys, ds_read_b128) x Qt^T (B: columns 0-7 = high halves of the 8 heads' qt,
//                          8-15 = low halves; resident in registers), MFMA 16x16x32, D/32 steps;
//   hi + lo columns added (DPP row rotate), fp32 online softmax per head column, P split hi / lo the same way;
//   C^T [D x 16]        += E_tile^T (A: rows = feature d, ds_read_b64_tr_b16: the hardware 4x4 transposing read) x P^T
//                          (B: the S^T accumulator layout IS the 16x16x16 B-operand layout), MFMA 16x16x16, D/16 tiles.
// At the end the four waves' (m, l, C) are merged through LDS (the ring is dead by then) and ctx is written as the bf16
// fragment-major A operand of the residual GEMM (kernels.h fm16, K = heads * D).
#include <atomic>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "gemm_common.h"

namespace msh {
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

template <int D, int NWAVES, int NS, int XSLOTS = 0>
struct XaCfg {
  static constexpr int NW = NWAVES;                        // waves per workgroup: tile i of the clip belongs to wave i % NW
  static constexpr int KS = D / 32;                        // k-steps of the score product
  static constexpr int DT = D / 16;                        // 16-row tiles of the context product
  // A ring slot is the tile as it lies in memory: 16 rows of 2 D bytes, exactly D / 32 KiB -- one DMA piece per KiB, the
  // same lane offset for every piece.  (Both read patterns below see 2-way bank conflicts at this pitch, a few cycles per
  // tile; a padded pitch costs per-lane address tables and, at D = 416, the twelfth slot.)
  static constexpr int ROWB = D * 2;
  static constexpr int TILE = 16 * ROWB;
  static constexpr int PIECES = TILE / 1024;
  static constexpr int SLOT = TILE;
  static constexpr int NSLOT = NS;
  // XS > 0 (with NS = 1): waves 0 .. XS - 1 own TWO slots, the others one.  The tiles of a clip rarely divide evenly (10 s:
  // 26 tiles on 8 waves), and with one slot each the waves that hold the extra tile fetch it alone in a last, latency-bound
  // round; with a second slot they request it while the others request their last one.
  static constexpr int XS = XSLOTS;
  static constexpr int RING = (NW * NSLOT + XS) * SLOT;
  static_assert(XS == 0 || NS == 1, "extra slots go with the one-slot ring");
  static_assert(XS <= NW, "at most one extra slot per wave");
  // byte offset of wave w's ring (wave-uniform)
  __host__ __device__ static constexpr int wring_off(int w) { return XS > 0 ? (w < XS ? 2 * w : w + XS) * SLOT : w * NSLOT * SLOT; }
  // Merge area: a wave's fp32 partial context [8 heads][D] is exactly one ring slot (16 rows x 2 D bytes = 8 x D x 4), so
  // every wave overlays ITS OWN ring with it -- no barrier between the last tile's reads and the partial's writes, and the
  // waves that own one tile less than the others have written theirs before the last tiles land.  The heads' rows start
  // on the same bank (D * 4 bytes = a multiple of 128), so column c of head h is kept at c ^ (h << 2): a 16-lane group of
  // the 8-byte writes then covers 32 distinct banks, and the output pass's 16-byte reads stay aligned.
  static constexpr int MST = D;
  static constexpr int ML = RING;                          // [wave][head] {m, l} behind the ring
  static constexpr int LDS = RING + NW * 8 * 8;
  static_assert(8 * MST * 4 <= NSLOT * SLOT, "a wave's partial must fit its own ring");
  static_assert(D % 32 == 0 && TILE % 1024 == 0, "a tile must be whole KiB pieces");
  static_assert(NS >= 1 && NS <= 3, "ring depth");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ float dpp_ror8(float v) {   // value of lane (l ^ 8) within every row of 16 lanes
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, true));
}
// x + (x of lane ^ 8) in one VALU operation (the compiler keeps a v_mov_b32_dpp and a v_add_f32 apart once a select sits
// between them)
__device__ __forceinline__ float add_ror8(float v) {
  float r;
  asm("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
  return r;
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

// One 16-key tile of E -> one ring slot: PIECES x buffer_load_dwordx4 ... lds (1 KiB each: lane l writes 16 bytes at
// M0 + 16 l).  The lane's byte offset (16 l) is one VGPR, the piece's offset inside the clip is the SGPR soffset and its LDS
// address is M0, both stepped by SALU adds: per piece two scalar adds and one VMEM issue, no vector arithmetic.  M0 is
// restored inside the statement.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
#define MSH_XA_P(POL) "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[so], %[so], 0x400\n\tbuffer_load_dwordx4 %[vo], %[rs], %[so] offen" POL " lds\n\t"
#define MSH_XA_HEAD(POL) "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[vo], %[rs], %[so] offen" POL " lds\n\t"
#define MSH_XA_TAIL "s_mov_b32 m0, %[keep]"
#define MSH_XA_13(POL) MSH_XA_HEAD(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) \
    MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_TAIL
#define MSH_XA_9(POL) MSH_XA_HEAD(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) MSH_XA_P(POL) \
    MSH_XA_P(POL) MSH_XA_TAIL
// NT: the non-temporal cache policy on the stream (every byte of a clip is read once per launch, by one CU)
template <int P, bool NT>
__device__ __forceinline__ void dma_tile(int vo, i32x4_t rs, int so, unsigned lds) {
  static_assert(P == 13 || P == 9, "pieces per tile");
  unsigned keep;
  if constexpr (P == 13 && NT)
    asm volatile(MSH_XA_13(" nt") : [keep] "=&s"(keep), [so] "+s"(so) : [vo] "v"(vo), [rs] "s"(rs), [lds] "s"(lds) : "memory", "scc");
  else if constexpr (P == 13)
    asm volatile(MSH_XA_13("") : [keep] "=&s"(keep), [so] "+s"(so) : [vo] "v"(vo), [rs] "s"(rs), [lds] "s"(lds) : "memory", "scc");
  else if constexpr (NT)
    asm volatile(MSH_XA_9(" nt") : [keep] "=&s"(keep), [so] "+s"(so) : [vo] "v"(vo), [rs] "s"(rs), [lds] "s"(lds) : "memory", "scc");
  else
    asm volatile(MSH_XA_9("") : [keep] "=&s"(keep), [so] "+s"(so) : [vo] "v"(vo), [rs] "s"(rs), [lds] "s"(lds) : "memory", "scc");
}
#undef MSH_XA_P
#undef MSH_XA_HEAD
#undef MSH_XA_TAIL
#undef MSH_XA_13
#undef MSH_XA_9
// one piece with its own lane offsets (the ragged last tile of a clip)
__device__ __forceinline__ void dma_piece(int vo, i32x4_t rs, int so, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[vo], %[rs], %[so] offen lds\n\t"
               "s_mov_b32 m0, %[keep]"
               : [keep] "=&s"(keep)
               : [vo] "v"(vo), [rs] "s"(rs), [so] "s"(so), [lds] "s"(lds)
               : "memory");
}

// TR = true: the context product's A fragments come from ds_read_b64_tr_b16; false: four 2-byte reads per fragment (the
// plain formulation of the same gather, kept as the check of the transposing read: MSH_XATTN_CFG=40).
// ABL (developer ablations, tools/gpu_r4r.sh; garbage results): 1 = no DMA (the products run on whatever the LDS holds),
// 2 = no products / softmax (the tiles are only fetched and waited for), 4 = no merge / output, 8 = no output pass,
// 16 = no merge writes, 32 / 64 = one k-step / one row tile per group of the score / context product, 128 = time stamps, 256 = non-temporal stream
// (correct results: a candidate, not an ablation).
template <int D, bool TR, int NWAVES, int NS, int ABL = 0, int XS = 0>
__global__ __launch_bounds__(64 * NWAVES, 2) void dec_cross_absorbed_kernel(const bf16_t* __restrict__ qf,     // [M][D / 32][16][32]
                                                                          const bf16_t* __restrict__ enc,    // [R][D]
                                                                          const ClipMeta* __restrict__ clips, int M, int xcd_group,
                                                                          bf16_t* __restrict__ ctx,         // FM [M16][8 * D]
                                                                          unsigned long long* __restrict__ dbg) {
  using C = XaCfg<D, NWAVES, NS, XS>;
  // ABL & 128: wave time stamps (s_memtime, shader clock) into dbg[(block * NW + wave) * 16 + point]
#define MSH_XA_TL(i)                                                                                              \
  do {                                                                                                            \
    if constexpr ((ABL & 128) != 0)                                                                               \
      if ((threadIdx.x & 63) == 0) dbg[((size_t)blockIdx.x * NWAVES + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  MSH_XA_TL(0);
  constexpr int NW = C::NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Block -> clip.  xcd_group: the 16 clips of an output row tile (16 consecutive rows of the fragment-major context, whose
  // 128-byte lines interleave 16-byte pieces of all 16 rows) run on ONE XCD (workgroup i runs on XCD i % 8: observed
  // placement, used for speed only), so the pieces meet in that XCD's L2 and leave it as whole lines; in plain order the
  // 16 writers of a line sit on 8 XCDs and every L2 writes back its own masked copy of every line at the kernel's end.
  int b = blockIdx.x;
  if (xcd_group != 0) {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    b = ((x + 8 * (i >> 4)) << 4) + (i & 15);
    if (b >= M) return;
  }
  const ClipMeta cm = clips[b];
  const int T = cm.T;
  const int n_tiles = (T + 15) >> 4;
  // buffer descriptor over the clip's T rows of E (raw buffer: byte offsets)
  i32x4_t rs;
  {
    const unsigned long long a = reinterpret_cast<unsigned long long>(enc + (long)cm.row_start * D);
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffull));
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    rs[2] = __builtin_amdgcn_readfirstlane(T * D * 2);
    rs[3] = 0x00020000;
  }
  const int wring = __builtin_amdgcn_readfirstlane(C::wring_off(wave));
  const unsigned ring = lds_offset_of(smem) + (unsigned)wring;   // this wave's slots
  const int nslot = XS > 0 ? (wave < XS ? 2 : 1) : C::NSLOT;      // wave-uniform
  const int lane16 = lane * 16;
  auto issue_tile = [&](int tile, int slot) {
    if constexpr ((ABL & 1) != 0) return;
    const int so = tile * C::TILE;
    const unsigned dst = ring + (unsigned)slot * C::SLOT;
    if (tile * 16 + 16 <= T) {
      dma_tile<C::PIECES, (ABL & 256) != 0>(lane16, rs, so, dst);
    } else {   // last tile of a clip whose frame count is not a multiple of 16: rows past the end re-read the last valid row
      const unsigned last = (unsigned)(T - 1 - tile * 16);
#pragma unroll
      for (int j = 0; j < C::PIECES; ++j) {
        const unsigned o = (unsigned)(j * 1024 + lane16);
        unsigned r = o / (unsigned)C::ROWB;
        const unsigned c = o - r * (unsigned)C::ROWB;
        r = r < last ? r : last;
        dma_piece((int)(r * (unsigned)C::ROWB + c), rs, so, dst + (unsigned)j * 1024u);
      }
    }
  };
  // the first NSLOT tiles of this wave go out before anything else
#pragma unroll
  for (int sl = 0; sl < (XS > 0 ? 2 : C::NSLOT); ++sl)
    if (sl < nslot && wave + NW * sl < n_tiles) issue_tile(wave + NW * sl, sl);

  // B operand of the score product: column li = head (li & 7), value (li < 8) or rounding residual (li >= 8) of the head's
  // keys-side query, k = kg * 8 .. + 8 -- already in that order in memory (EpiQtFrag): one contiguous KiB per wave load
  bf16x8 qfrag[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks)
    qfrag[ks] = *reinterpret_cast<const bf16x8*>(qf + (((size_t)b * C::KS + ks) * 64 + (li * 4 + kg)) * 8);
  MSH_XA_TL(1);   // qfrag built (its loads waited for)
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
  for (int tile = wave; tile < n_tiles; tile += NW) {
    // tiles of this wave still in flight BEHIND this one: up to NSLOT - 1
    if constexpr ((ABL & 1) != 0) {
    } else if constexpr (XS > 0) {
      if (nslot == 2 && tile + NW < n_tiles) wait_vmcnt<C::PIECES>();
      else wait_vmcnt<0>();
    } else if constexpr (C::NSLOT == 1) {
      wait_vmcnt<0>();
    } else if constexpr (C::NSLOT == 2) {
      if (tile + NW < n_tiles) wait_vmcnt<C::PIECES>();
      else wait_vmcnt<0>();
    } else {
      if (tile + 2 * NW < n_tiles) wait_vmcnt<2 * C::PIECES>();
      else if (tile + NW < n_tiles) wait_vmcnt<C::PIECES>();
      else wait_vmcnt<0>();
    }
    if (tile == wave) MSH_XA_TL(2);   // first tile landed
    const char* sbase = smem + wring + (size_t)slot * C::SLOT;
    if constexpr ((ABL & 2) != 0) {
      if (tile + NW * nslot < n_tiles) issue_tile(tile + NW * nslot, slot);
      slot = slot + 1 == nslot ? 0 : slot + 1;
      continue;
    }

    // ---- scores: S^T[key][col] over D.  The A fragments (rows = keys) are requested in three groups ahead of the MFMAs
    // that consume them, FOUR accumulation chains: a read that lands in the registers an MFMA is still reading, or an MFMA
    // waiting on its predecessor's result, is what the first version of this loop spent most of its time in. ----
    constexpr int KSE = (ABL & 32) != 0 ? 3 : C::KS;
    constexpr int P0 = (KSE + 2) / 3, P1 = (KSE + 1) / 3, P2 = KSE / 3;
    auto read_s = [&](int ks) { return *reinterpret_cast<const bf16x8*>(sbase + a1_off + ks * 64); };
    bf16x8 e0[P0], e1[P1], e2[P2];
#pragma unroll
    for (int i = 0; i < P0; ++i) e0[i] = read_s(i);
#pragma unroll
    for (int i = 0; i < P1; ++i) e1[i] = read_s(P0 + i);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 sc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < P0; ++i) sc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e0[i], qfrag[i], sc[i & 3], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < P2; ++i) e2[i] = read_s(P0 + P1 + i);
    // ---- context: C^T[d][col] += E_tile^T x P^T, the A fragments read in four groups: group g + 1 is requested before
    // group g's MFMAs are issued, the first group before the softmax arithmetic (it does not depend on it) ----
    constexpr int DTE = (ABL & 64) != 0 ? 4 : C::DT;
    constexpr int G0 = (DTE + 3) / 4, G1 = (DTE + 2) / 4, G2 = (DTE + 1) / 4, G3 = DTE / 4;
    auto read_a = [&](int dt) {
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
      return a;
    };
    s16x4 a0[G0], a1[G1], a2[G2], a3[G3];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < P1; ++i)
      sc[(P0 + i) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e1[i], qfrag[P0 + i], sc[(P0 + i) & 3], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < G0; ++i) a0[i] = read_a(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < P2; ++i)
      sc[(P0 + P1 + i) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e2[i], qfrag[P0 + P1 + i], sc[(P0 + P1 + i) & 3], 0, 0, 0);
    if (tile == wave) MSH_XA_TL(3);   // score MFMAs issued
    float s[4];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = (sc[0][r] + sc[1][r]) + (sc[2][r] + sc[3][r]);
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
    if (tile == wave) MSH_XA_TL(4);   // softmax done
#pragma unroll
    for (int i = 0; i < G1; ++i) a1[i] = read_a(G0 + i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < G0; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0[i], pfrag, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < G2; ++i) a2[i] = read_a(G0 + G1 + i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < G1; ++i) acc[G0 + i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1[i], pfrag, acc[G0 + i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < G3; ++i) a3[i] = read_a(G0 + G1 + G2 + i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < G2; ++i)
      acc[G0 + G1 + i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2[i], pfrag, acc[G0 + G1 + i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < G3; ++i)
      acc[G0 + G1 + G2 + i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a3[i], pfrag, acc[G0 + G1 + G2 + i], 0, 0, 0);
    if (tile == wave) MSH_XA_TL(5);   // context MFMAs issued
    // the slot is free once every read of it has returned: fetch the tile NSLOT rounds ahead into it
    if (tile + NW * nslot < n_tiles) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_tile(tile + NW * nslot, slot);
    }
    if (tile == wave) MSH_XA_TL(6);   // next tile requested
    if (tile == wave + NW) MSH_XA_TL(7);   // second tile of the wave finished
    slot = slot + 1 == nslot ? 0 : slot + 1;
  }

  if constexpr ((ABL & 4) != 0) {
    float keep = l_part;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) keep += (acc[dt][0] + acc[dt][1]) + (acc[dt][2] + acc[dt][3]);
    if (keep == 12345.f) ctx[threadIdx.x] = (bf16_t)1;   // keeps every accumulator alive, never true in practice
    return;
  }
  // ---- merge the waves ----
  MSH_XA_TL(8);    // loop done
  const float l_wave = xa_rows_sum(l_part);   // all four key groups of the column
  // (the wave's own LDS operations run in order: its last tile's reads are ahead of these writes, and it has no DMA in flight)
  float* mine = reinterpret_cast<float*>(smem + wring);   // [head][MST], column c at c ^ (head << 2)
  float2* ml = reinterpret_cast<float2*>(smem + C::ML);                     // [wave][head]
  {
    const int hh = li & 7;
    const int sw = hh << 2;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      // high-half column + low-half column of the head: both lanes (li, li ^ 8) then hold the four sums, the lower lane
      // stores the first two, the upper one the last two
      const float v0 = add_ror8(acc[dt][0]);
      const float v1 = add_ror8(acc[dt][1]);
      const float v2 = add_ror8(acc[dt][2]);
      const float v3 = add_ror8(acc[dt][3]);
      const float2 w2 = li < 8 ? make_float2(v0, v1) : make_float2(v2, v3);
      const int c = dt * 16 + kg * 4 + (li < 8 ? 0 : 2);
      if constexpr ((ABL & 16) == 0) *reinterpret_cast<float2*>(mine + hh * C::MST + (c ^ sw)) = w2;
      if constexpr ((ABL & 16) != 0)
        if (w2.x == 12345.f) ctx[0] = 1;
    }
  }
  if (lane < 8) ml[wave * 8 + lane] = make_float2(m_run, l_wave);
  MSH_XA_TL(9);    // partials written
  __syncthreads();
  MSH_XA_TL(10);   // barrier passed
  constexpr int CHUNKS = 8 * D / 8;   // 16-byte output chunks of the clip's row
  if constexpr ((ABL & 8) != 0) return;
  for (int ch = threadIdx.x; ch < CHUNKS; ch += 64 * NW) {
    const int h = ch / (D / 8), d0 = (ch - h * (D / 8)) * 8;
    float mw[NW], f[NW];
    float m = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      mw[w] = ml[w * 8 + h].x;
      m = fmaxf(m, mw[w]);
    }
    m = m > -INFINITY ? m : 0.f;   // a clip without frames: every weight 0, output 0
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      f[w] = __builtin_amdgcn_exp2f(mw[w] - m);
      l += f[w] * ml[w * 8 + h].y;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    const int sw = h << 2;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* row = reinterpret_cast<const float*>(smem + C::wring_off(w)) + h * C::MST;
      const float4 c0 = *reinterpret_cast<const float4*>(row + (d0 ^ sw));
      const float4 c1 = *reinterpret_cast<const float4*>(row + ((d0 + 4) ^ sw));
      o[0] += f[w] * c0.x; o[1] += f[w] * c0.y; o[2] += f[w] * c0.z; o[3] += f[w] * c0.w;
      o[4] += f[w] * c1.x; o[5] += f[w] * c1.y; o[6] += f[w] * c1.z; o[7] += f[w] * c1.w;
    }
    uint4 u;
    u.x = pack_bf16x2(o[0] * inv, o[1] * inv);
    u.y = pack_bf16x2(o[2] * inv, o[3] * inv);
    u.z = pack_bf16x2(o[4] * inv, o[5] * inv);
    u.w = pack_bf16x2(o[6] * inv, o[7] * inv);
    *reinterpret_cast<uint4*>(ctx + fm16(b, h * D + d0, 8 * D / 32)) = u;
  }
  MSH_XA_TL(11);   // output stores issued
#undef MSH_XA_TL
}

// Shape of the workgroup (developer knob MSH_XATTN_CFG): 84 = eight waves, one ring slot each plus a second one for waves
// 0-3 (default; 156 KiB of LDS -- the kernel's 246 registers per lane keep other kernels off its CUs anyway); 81 = eight
// waves with one slot each (a wave that sits in the back-pressure of its own DMA issue does not keep the CU from
// computing: seven others are there); 43 / 42 = four waves with three / two slots each; 40 = 42 with the context product's
// fragments gathered by 2-byte reads instead of ds_read_b64_tr_b16 (the check of that read).
int xattn_cfg() {
  static const int v = [] {
    const char* e = dev_getenv("MSH_XATTN_CFG");
    const int c = e != nullptr ? atoi(e) : 84;
    return c == 81 || c == 43 || c == 42 || c == 40 ? c : 84;
  }();
  return v;
}

template <int D, bool TR, int NWAVES, int NS, int ABL = 0, int XS = 0>
void launch_absorbed_cfg(const bf16_t* qt, const bf16_t* enc, const ClipMeta* clips, int M, bf16_t* ctx, hipStream_t s,
                         unsigned long long* dbg = nullptr) {
  using C = XaCfg<D, NWAVES, NS, XS>;
  // the attribute belongs to the (function, device) pair: engines on several GPUs of one process (options num_gpus / devices)
  // each need it before their first launch
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  MSH_HIP(hipGetDevice(&dev));
  if (dev >= 64 || ((attr_done.load(std::memory_order_acquire) >> dev) & 1ull) == 0) {
    MSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_cross_absorbed_kernel<D, TR, NWAVES, NS, ABL, XS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    if (dev < 64) attr_done.fetch_or(1ull << dev, std::memory_order_release);
  }
  // MSH_XATTN_XCD=0: plain block -> clip order (developer knob; see "Block -> clip" in the kernel)
  static const bool group = [] {
    const char* e = dev_getenv("MSH_XATTN_XCD");
    return !(e != nullptr && e[0] == '0');
  }();
  const unsigned grid = group ? 128u * (unsigned)(((M + 15) / 16 + 7) / 8) : (unsigned)M;
  MSH_LAUNCH((dec_cross_absorbed_kernel<D, TR, NWAVES, NS, ABL, XS>), dim3(grid), dim3(64 * NWAVES), C::LDS, s, qt, enc, clips, M,
             group ? 1 : 0, ctx, dbg);
}
template <int D>
void launch_absorbed(const bf16_t* qt, const bf16_t* enc, const ClipMeta* clips, int M, bf16_t* ctx, hipStream_t s, bool stream_nt) {
  static const int abl = [] {
    const char* e = dev_getenv("MSH_XATTN_ABL");
    return e != nullptr ? atoi(e) : 0;
  }();
  if constexpr (D == 416) {
    switch (abl) {
      case 1: return launch_absorbed_cfg<D, true, 8, 1, 1, 4>(qt, enc, clips, M, ctx, s);
      case 2: return launch_absorbed_cfg<D, true, 8, 1, 2, 4>(qt, enc, clips, M, ctx, s);
      case 4: return launch_absorbed_cfg<D, true, 8, 1, 4, 4>(qt, enc, clips, M, ctx, s);
      case 6: return launch_absorbed_cfg<D, true, 8, 1, 6, 4>(qt, enc, clips, M, ctx, s);
      case 10: return launch_absorbed_cfg<D, true, 8, 1, 10, 4>(qt, enc, clips, M, ctx, s);
      case 256: return launch_absorbed_cfg<D, true, 8, 1, 256, 4>(qt, enc, clips, M, ctx, s);
      case 262: return launch_absorbed_cfg<D, true, 8, 1, 262, 4>(qt, enc, clips, M, ctx, s);
      default: break;
    }
  }
  switch (xattn_cfg()) {
    case 43: return launch_absorbed_cfg<D, true, 4, 3>(qt, enc, clips, M, ctx, s);
    case 42: return launch_absorbed_cfg<D, true, 4, 2>(qt, enc, clips, M, ctx, s);
    case 40: return launch_absorbed_cfg<D, false, 4, 2>(qt, enc, clips, M, ctx, s);
    case 81: return launch_absorbed_cfg<D, true, 8, 1>(qt, enc, clips, M, ctx, s);
    default:
      // ABL bit 256 = the encoder rows with the non-temporal policy (correct results; see dec_cross_absorbed)
      if (stream_nt) return launch_absorbed_cfg<D, true, 8, 1, 256, 4>(qt, enc, clips, M, ctx, s);
      return launch_absorbed_cfg<D, true, 8, 1, 0, 4>(qt, enc, clips, M, ctx, s);
  }
}

}  // namespace

bool cross_absorbed_supported(int D, int heads) { return heads == 8 && (D == 416 || D == 288); }

void dec_cross_absorbed(const bf16_t* qt, const bf16_t* enc, const ClipMeta* clips, int M, int D, int heads, bf16_t* ctx,
                        hipStream_t s, bool stream_nt) {
  if (!cross_absorbed_supported(D, heads)) throw std::runtime_error("dec_cross_absorbed: unsupported shape");
  if (D == 416) launch_absorbed<416>(qt, enc, clips, M, ctx, s, stream_nt);
  else launch_absorbed<288>(qt, enc, clips, M, ctx, s, stream_nt);
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
  // the queries in the operand order the GEMM epilogue writes (gemm_common.h EpiQtFrag): value and rounding residual
  std::vector<bf16_t> qf16((size_t)M * 8 * D * 2);
  for (int b = 0; b < M; ++b)
    for (int h = 0; h < 8; ++h)
      for (int d = 0; d < D; ++d) {
        const float x = qt[((size_t)b * 8 + h) * D + d];
        const bf16_t hi = f32_to_bf16(x);
        const size_t frag = ((size_t)b * (D / 32) + d / 32) * 512 + d % 32;
        qf16[frag + h * 32] = hi;
        qf16[frag + (h + 8) * 32] = f32_to_bf16(x - bf16_to_f32(hi));
      }
  bf16_t* dq = nullptr;
  bf16_t *de = nullptr, *dc = nullptr;
  ClipMeta* dm = nullptr;
  MSH_HIP(hipMalloc(&dq, (size_t)M * 8 * D * 4));
  MSH_HIP(hipMalloc(&de, e16.size() * 2));
  MSH_HIP(hipMalloc(&dc, (size_t)M16 * 8 * D * 2));
  MSH_HIP(hipMalloc(&dm, (size_t)M * sizeof(ClipMeta)));
  MSH_HIP(hipMemcpy(dq, qf16.data(), qf16.size() * 2, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(de, e16.data(), e16.size() * 2, hipMemcpyHostToDevice));
  MSH_HIP(hipMemcpy(dm, cm.data(), (size_t)M * sizeof(ClipMeta), hipMemcpyHostToDevice));
  MSH_HIP(hipMemset(dc, 0, (size_t)M16 * 8 * D * 2));
  dec_cross_absorbed(dq, de, dm, M, D, 8, dc, 0, false);
  MSH_HIP(hipDeviceSynchronize());
  float ms = 0.f;
  if (iters > 0) {
    hipEvent_t a, b2;
    MSH_HIP(hipEventCreate(&a));
    MSH_HIP(hipEventCreate(&b2));
    MSH_HIP(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) dec_cross_absorbed(dq, de, dm, M, D, 8, dc, 0, false);
    MSH_HIP(hipEventRecord(b2, 0));
    MSH_HIP(hipEventSynchronize(b2));
    MSH_HIP(hipEventElapsedTime(&ms, a, b2));
    ms /= (float)iters;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b2);
  }
  if (const char* tl = dev_getenv("MSH_XATTN_TIMELINE"); tl != nullptr && tl[0] == '1' && D == 416) {
    // developer: per-wave s_memtime stamps of the default shape (tools/xattn_microbench.py prints nothing else for it)
    const size_t nb = 128u * (size_t)(((M + 15) / 16 + 7) / 8) + (size_t)M;   // blocks of either block -> clip order
    const size_t n = nb * 8 * 16;
    unsigned long long* dd = nullptr;
    MSH_HIP(hipMalloc(&dd, n * 8));
    MSH_HIP(hipMemset(dd, 0, n * 8));
    launch_absorbed_cfg<416, true, 8, 1, 128, 4>(dq, de, dm, M, dc, 0, dd);   // warm (code object, caches)
    MSH_HIP(hipDeviceSynchronize());
    MSH_HIP(hipMemset(dd, 0, n * 8));
    launch_absorbed_cfg<416, true, 8, 1, 128, 4>(dq, de, dm, M, dc, 0, dd);
    MSH_HIP(hipDeviceSynchronize());
    std::vector<unsigned long long> h(n);
    MSH_HIP(hipMemcpy(h.data(), dd, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dd);
    // the shader clocks of different XCDs have different origins: every workgroup's stamps are taken relative to the
    // earliest start among its own waves
    static const char* names[12] = {"start", "qfrag", "tile0 landed", "scores issued", "softmax", "context issued", "next dma issued",
                                    "2nd tile done", "loop done", "partials written", "barrier", "stores issued"};
    std::vector<std::vector<unsigned long long>> pts(12);
    for (size_t blk = 0; blk < nb; ++blk) {
      unsigned long long t0 = ~0ull;
      for (int w = 0; w < 8; ++w)
        if (h[(blk * 8 + w) * 16] != 0) t0 = std::min(t0, h[(blk * 8 + w) * 16]);
      if (t0 == ~0ull) continue;
      for (int w = 0; w < 8; ++w)
        for (int i = 0; i < 12; ++i)
          if (h[(blk * 8 + w) * 16 + i] != 0) pts[i].push_back(h[(blk * 8 + w) * 16 + i] - t0);
    }
    fprintf(stderr, "[xattn timeline] %d clips, shader cycles since the workgroup's first wave started\n", M);
    for (int i = 0; i < 12; ++i) {
      std::vector<unsigned long long>& v = pts[i];
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      fprintf(stderr, "  %-18s min %7llu  p10 %7llu  p50 %7llu  p90 %7llu  max %7llu  (%zu waves)\n", names[i], v.front(), v[v.size() / 10],
              v[v.size() / 2], v[v.size() * 9 / 10], v.back(), v.size());
    }
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
