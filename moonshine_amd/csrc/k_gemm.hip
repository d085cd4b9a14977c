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

#include "gemm_common.h"

namespace msh {
namespace {

// ------------------------------------------------------------------------------------------------
// Tiled kernel
// ------------------------------------------------------------------------------------------------

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
#define MSH_GLOAD(kt_)                                                                \
  {                                                                                   \
    const int ka_ = conv_k_offset<TAPS>(kt_, kc, kperm), kw_ = (kt_) << 5;            \
    MSH_LDA(0, ka_) MSH_LDA(1, ka_) MSH_LDA(2, ka_) MSH_LDA(3, ka_)                   \
    MSH_LDB(0, kw_) MSH_LDB(1, kw_) MSH_LDB(2, kw_) MSH_LDB(3, kw_)                   \
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
  constexpr int TAPS = epi_taps<Epi>::value;   // conv GEMMs may walk K channel-block-major (conv_k_offset, gemm_common.h)
  const int kc = K / TAPS, kperm = epi_kperm(epi);
  MSH_GLOAD(0);
  MSH_SSTORE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) MSH_GLOAD(kt + 1);  // next k-slice in flight during the MFMAs
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

  if constexpr (SWAP && has_row_sums<Epi>::value) {   // conv1: the stored values' row sums go out with them (GroupNorm statistics)
    store_rows_with_sums<TM, TN>(epi, acc, m0 + wave * 16 * TM, n0, M, N, ntn, n0 / BN, li, kg);
    return;
  }
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
// ABL (microbenchmark ablations, 0 in the product): bit 0 = no DMA after the prologue, bit 1 = no MFMA,
// bit 2 = no fragment reads, bit 3 = no epilogue stores, bit 4 = direct (unstaged) stores, bit 5 = the W stream non-temporal
// (not an ablation: correct results, used by the LM head under MSH_LMHEAD_NT=1); ABL >> 8 = b + 1: workgroups whose id has bit b set start
// ~10 us late (probe for co-resident workgroups running their main loops and epilogues in lockstep).
template <int NW, int TM, int TN, int NSTAGE, bool SWAP, class Epi, int ABL = 0>
__global__ __launch_bounds__(64 * NW, (NW == 4 && TM * TN * 4 <= 104) ? 2 : 1) void gemm_tiled_dma_kernel(const bf16_t* __restrict__ A, long lda,
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
  if constexpr (((ABL >> 8) & 0xff) != 0) {
    if ((bid >> (((ABL >> 8) & 0xff) - 1)) & 1) {
#pragma unroll 1
      for (int i = 0; i < ((ABL >> 16) ? (ABL >> 16) : 3); ++i) __builtin_amdgcn_s_sleep(127);
    }
  }

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
  constexpr int TAPS = epi_taps<Epi>::value;   // conv GEMMs may walk K channel-block-major (conv_k_offset, gemm_common.h)
  const int kc = K / TAPS, kperm = epi_kperm(epi);
  auto issue = [&](int kt) {
    const unsigned sb = (unsigned)(kt % NSTAGE) * (STAGE_SLOTS * 16u);
    const int ka = conv_k_offset<TAPS>(kt, kc, kperm), kw = kt << 5;   // A slices may be permuted, W slices are stored in order
#pragma unroll
    for (int i = 0; i < PMAX; ++i)
      if (i < my_pieces) {
        const int ko = (TAPS > 1 && wave + NW * i < PA) ? ka : kw;
        if constexpr ((ABL & 32) != 0) {   // W pieces with the non-temporal policy (a weight read once per launch: the LM head)
          if (wave + NW * i >= PA) dma16_nt(src[i] + ko, dst[i] + sb);
          else dma16(src[i] + ko, dst[i] + sb);
        } else {
          dma16(src[i] + ko, dst[i] + sb);
        }
      }
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
    if (kt + AHEAD < nk && !(ABL & 1)) issue(kt + AHEAD);
    const uint4* st = lds + (kt % NSTAGE) * STAGE_SLOTS;
    // all fragment reads of the k-slice are issued before the first MFMA: one LDS latency per slice
    // instead of one per column tile (the two waves of a SIMD then cover each other's read phase)
    uint4 afr[TM], bfr[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wave * 16 * TM + i * 16 + li;
      afr[i] = (ABL & 4) ? make_uint4(kt, lane, 1, 2) : st[row * 4 + (kg ^ swz(row))];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = j * 16 + li;
      bfr[j] = (ABL & 4) ? make_uint4(lane, kt, 3, 4) : st[BM * 4 + row * 4 + (kg ^ swz(row))];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((ABL & 2) != 0) {  // keep the fragments live without issuing MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(afr[i].x), "v"(afr[i].y), "v"(afr[i].z), "v"(afr[i].w));
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bfr[j].x), "v"(bfr[j].y), "v"(bfr[j].z), "v"(bfr[j].w));
      continue;
    }
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

  if constexpr ((ABL & 8) != 0) {  // keep the accumulators live, store nothing
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) epi.n4(m0, n0, acc[0][0]);
    return;
  }
  if constexpr (is_row_argmax<Epi>::value) {
    // row m = li of this wave's row tile; its columns of the tile sit in j (TN), kg (4 lane groups) and the 4
    // accumulator registers: reduce locally, then across the 4 lane groups.  Ties keep the lowest column.
    static_assert(SWAP, "row argmax needs the n4 orientation");
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float best = -INFINITY;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 16 + kg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][j][r];
          if (n + r < N && v > best) {  // columns are visited in ascending order: '>' keeps the first maximum
            best = v;
            bidx = n + r;
          }
        }
      }
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) {
          best = ov;
          bidx = oi;
        }
      }
      const int m = m0 + wave * 16 * TM + i * 16 + li;
      if (kg == 0 && m < M) {
        const int tile_n = n0 / (16 * TN);
        epi.pval[(long)m * epi.ntn + tile_n] = best;
        epi.pidx[(long)m * epi.ntn + tile_n] = bidx;
      }
    }
    return;
  }
  if constexpr (is_staged_bf16<Epi>::value && SWAP && (ABL & 16) == 0) {
    // the pipeline ring is dead once every wave has read the last k-slice: reuse it as the store staging area
    constexpr int STG_BYTES = NW * 16 * StagedRow<TN>::ROWP * 8, TAB_BYTES = 16 * 56 * 4;  // per-wave 16-row RoPE table
    static_assert(STG_BYTES + NW * TAB_BYTES <= NSTAGE * STAGE_SLOTS * 16, "staging does not fit the ring");
    __builtin_amdgcn_s_barrier();
    uint2* stg = reinterpret_cast<uint2*>(lds) + wave * (16 * StagedRow<TN>::ROWP);
    float* rowtab = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + STG_BYTES) + wave * (TAB_BYTES / 4);
#pragma unroll
    for (int i = 0; i < TM; ++i)
      staged_store_tile<TN>(epi, stg, acc[i], m0 + wave * 16 * TM + i * 16, n0, M, N, lane, rowtab, ntn, n0 / BN);
    return;
  }
  if constexpr (SWAP && has_row_sums<Epi>::value) {   // (the direct-store ablation of an epilogue with row sums)
    store_rows_with_sums<TM, TN>(epi, acc, m0 + wave * 16 * TM, n0, M, N, ntn, n0 / BN, li, kg);
    return;
  }
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

template <int NW, int TM, int TN, int NSTAGE, bool SWAP, class Epi, int ABL = 0>
int launch_tiled_dma_cfg(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  constexpr int BM = 16 * NW * TM, BN = 16 * TN;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  const int nblocks = ntm * ntn;
  MSH_LAUNCH((gemm_tiled_dma_kernel<NW, TM, TN, NSTAGE, SWAP, Epi, ABL>), dim3(nblocks), dim3(64 * NW), 0, s, A, lda,
                     W, M, N, K, ntn, nblocks, epi);
  return ntn;   // column tiles (the row-sum layout of EpiTanhBf16)
}

// ------------------------------------------------------------------------------------------------
// A-stationary tiled kernel for short K (K = 32*KS <= 416: qkv, o-proj, fc1, cross-KV).
//
// The microbenchmark ablations (profiles/r01g_gemm_microbench_ablations.txt) show the LDS-DMA kernel is
// bound by per-CU ingest (its "DMA only" time equals its full time), and for K = 416 it also refills its
// pipeline every 13 k-slices.  Here a workgroup keeps the MFMA A fragments of its 128 rows (4 waves x 32
// rows x K) in registers for its whole life and walks over column tiles of 208: only W k-slices (13 KiB)
// move through the LDS ring -- 128 flop per ingested byte instead of 79 -- and the ring keeps running
// across column tiles, so the next tile's first slices are already in flight while a tile's epilogue
// stores drain (one vmcnt(0) per tile, after the epilogue, covers both).
// ------------------------------------------------------------------------------------------------
template <int KS, int TN, int NST, bool SWAP, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_astat_kernel(const bf16_t* __restrict__ A, long lda,
                                                            const bf16_t* __restrict__ W, int M, int N, int ntn,
                                                            int nsplit, Epi epi) {
  constexpr int K = 32 * KS, BM = 128, BN = 16 * TN;
  constexpr int PMAX = (TN + 3) / 4;          // 1-KiB W pieces per wave per k-slice
  constexpr int AHEAD = NST - 1;
  __shared__ __attribute__((aligned(16))) uint4 lds[NST * TN * 64];
  // Staged (LDS-transposed) stores are off here: the ring keeps running across column tiles, so they would need
  // their own slab, and the extra live state spilled 46 VGPRs -- fc1 went from 339 to 460 us (r02b).
  constexpr bool STAGED = false;
  __shared__ __attribute__((aligned(16))) uint2 stage_lds[STAGED ? 4 * 16 * StagedRow<TN>::ROWP : 1];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = blockIdx.x / nsplit, part = blockIdx.x - mt * nsplit;
  const int t0 = (int)((long)ntn * part / nsplit), t1 = (int)((long)ntn * (part + 1) / nsplit);
  const int m0 = mt * BM + wave * 32;
  const int my_pieces = (TN - wave + 3) / 4;

  // A fragments: rows m0 + i*16 + li, k-chunk kg of every 32-wide slice
  bf16x8 afr[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int gm = m0 + i * 16 + li;
    gm = gm < M ? gm : M - 1;
    const bf16_t* a = A + (long)gm * lda + kg * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 t = *reinterpret_cast<const uint4*>(a + s * 32);
      afr[i][s] = *reinterpret_cast<const bf16x8*>(&t);
    }
  }
  // W piece sources: row (within the column tile) and swizzled chunk are fixed per lane
  // (N is a multiple of the 208-wide column tile here, so no row clamping is needed)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_offset_of(&lds[0])) + (unsigned)wave * 1024u;
  const int wrow = wave * 16 + (lane >> 2);
  const bf16_t* wsrc = W + (long)wrow * K + (((lane & 3) ^ swz(wrow)) << 3);  // swz(row + 64*i) == swz(row)
  const int n_slices = (t1 - t0) * KS;
  int it = t0, ik = 0, islice = 0;             // issue cursor (tile, k-slice, running index)
  auto issue_next = [&]() {
    if (islice < n_slices) {
      const unsigned sb = (unsigned)(islice % NST) * (TN * 1024u);
      const bf16_t* src = wsrc + (long)(it * BN) * K + (ik << 5);
#pragma unroll
      for (int i = 0; i < PMAX; ++i) {
        if (i < my_pieces) dma16(src + (long)i * 64 * K, lds_base + sb + (unsigned)i * 4096u);
      }
      ++islice;
      if (++ik == KS) {
        ik = 0;
        ++it;
      }
    }
  };
#pragma unroll
  for (int s = 0; s < AHEAD; ++s) issue_next();

  int cslice = 0;
  for (int t = t0; t < t1; ++t) {
    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KS; ++kt) {
      // slices cslice+1 .. cslice+AHEAD-1 may stay in flight (fewer at the very end)
      const int left = n_slices - 1 - cslice;
      const int inflight = left < AHEAD - 1 ? left : AHEAD - 1;
      if (my_pieces == PMAX) {
        if (inflight >= 2) wait_vmcnt<2 * PMAX>();
        else if (inflight == 1) wait_vmcnt<PMAX>();
        else wait_vmcnt<0>();
      } else {
        if (inflight >= 2) wait_vmcnt<2 * (PMAX - 1)>();
        else if (inflight == 1) wait_vmcnt<PMAX - 1>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      issue_next();
      const uint4* st = lds + (cslice % NST) * (TN * 64);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = j * 16 + li;
        const uint4 b = st[row * 4 + (kg ^ swz(row))];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (SWAP)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&b), afr[i][kt],
                                                                 acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[i][kt], *reinterpret_cast<const bf16x8*>(&b),
                                                                 acc[i][j], 0, 0, 0);
        }
      }
      ++cslice;
    }
    const int n0 = t * BN;
    if constexpr (STAGED) {
      uint2* stg = stage_lds + wave * (16 * StagedRow<TN>::ROWP);
#pragma unroll
      for (int i = 0; i < 2; ++i) staged_store_tile<TN>(epi, stg, acc[i], m0 + i * 16, n0, M, N, lane);
    } else if constexpr (!SWAP && is_paired_keys<Epi>::value) {
      // rows m0 .. m0+31 are this wave's keys: lane (li, kg) holds keys kg*4..+3 of both 16-row tiles for column li
      const int mr = m0 + kg * 4;
      const bool rows_ok = m0 + 31 < M;  // M (total rows) is a multiple of 8; a ragged last tile takes the 4-key path
      if (rows_ok) {
        const auto klo = epi.key_row(mr), khi = epi.key_row(mr + 16);
        // the 8-key run this lane stores: even kg -> keys kg*4.. of tile 0, odd kg -> keys (kg-1)*4.. of tile 1
        const auto kst = epi.key_row((kg & 1) ? m0 + 16 + (kg - 1) * 4 : mr);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + j * 16 + li;
          const uint2 lo = epi.pack_keys4(klo, acc[0][j]), hi = epi.pack_keys4(khi, acc[1][j]);
          const uint2 send = (kg & 1) ? lo : hi;  // even kg keeps tile 0, odd kg keeps tile 1
          uint2 recv;
          recv.x = __shfl_xor(send.x, 16, 64);
          recv.y = __shfl_xor(send.y, 16, 64);
          if (n < N) epi.store_keys8(kst, n, (kg & 1) ? make_uint4(recv.x, recv.y, hi.x, hi.y) : make_uint4(lo.x, lo.y, recv.x, recv.y));
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + j * 16 + li;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int m = m0 + i * 16 + kg * 4;
            if (m < M && n < N) epi.m4(m, n, acc[i][j]);
          }
        }
      }
    } else if constexpr (!SWAP && is_paired_keys_fp8<Epi>::value) {
      // the same key pairing with one byte per key: a lane stores 8 consecutive keys as 8 bytes
      const int mr = m0 + kg * 4;
      if (m0 + 31 < M) {
        const auto klo = epi.key_row(mr), khi = epi.key_row(mr + 16);
        const auto kst = epi.key_row((kg & 1) ? m0 + 16 + (kg - 1) * 4 : mr);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + j * 16 + li;
          const float qs = epi.qscale[n < N ? n : N - 1];
          const uint32_t lo = epi.pack_keys4(klo, acc[0][j], qs), hi = epi.pack_keys4(khi, acc[1][j], qs);
          const uint32_t recv = __shfl_xor((kg & 1) ? lo : hi, 16, 64);
          if (n < N) epi.store_keys8(kst, n, (kg & 1) ? make_uint2(recv, hi) : make_uint2(lo, recv));
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + j * 16 + li;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int m = m0 + i * 16 + kg * 4;
            if (m < M && n < N) epi.m4(m, n, acc[i][j]);
          }
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (SWAP) {
          const int m = m0 + i * 16 + li, n = n0 + j * 16 + kg * 4;
          if (m < M && n < N) epi.n4(m, n, acc[i][j]);
        } else {
          const int m = m0 + i * 16 + kg * 4, n = n0 + j * 16 + li;
          if (m < M && n < N) epi.m4(m, n, acc[i][j]);
        }
      }
    }
    }
    // the epilogue's loads / stores share the vmcnt counter with the DMAs: drain everything once per tile
    // (the ring was refilled before the epilogue, so its latency overlaps the stores)
    wait_vmcnt<0>();
  }
}

template <int KS, int TN, int NST, bool SWAP, class Epi>
int launch_astat_cfg(const bf16_t* A, long lda, const bf16_t* W, int M, int N, Epi epi, hipStream_t s) {
  const int ntm = (M + 127) / 128, ntn = (N + 16 * TN - 1) / (16 * TN);
  int nsplit = ntm >= 1024 ? 1 : (1024 + ntm - 1) / ntm;
  if (nsplit > ntn) nsplit = ntn;
  MSH_LAUNCH((gemm_astat_kernel<KS, TN, NST, SWAP, Epi>), dim3(ntm * nsplit), dim3(256), 0, s, A, lda, W, M, N,
                     ntn, nsplit, epi);
  return ntn;
}

// MSH_GEMM_MODE (debug / A-B switch): 0 = register-staged double buffer; 1 = LDS-DMA, 256x208 tile, 4 waves,
// 4 stages (one workgroup per CU: measured ~2x slower than 2, nothing hides a wave's ds_read latency);
// 2 = LDS-DMA, 128x208 tile, 4 waves, 3 stages, two workgroups per CU (default);
// 3 = LDS-DMA, 256x208 tile, 8 waves (two per SIMD), 4 stages.
inline int gemm_mode() {
  static int mode = [] {
    const char* e = dev_getenv("MSH_GEMM_MODE");
    return e ? atoi(e) : 2;
  }();
  return mode;
}

template <int TM, int TN, bool SWAP, class Epi>
int launch_tiled_cfg(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  const int nblocks = ntm * ntn;
  MSH_LAUNCH((gemm_tiled_kernel<TM, TN, SWAP, Epi>), dim3(nblocks), dim3(256), 0, s, A, lda, W, M, N, K, ntn,
                     nblocks, epi);
  return ntn;
}

// Tile choice: TN = 13 (208 columns) divides every base-model width; widths that are multiples of 144
// (tiny, D = 288) use TN = 9; anything else falls back to TN = 4 with column predication.  TM = 4
// (256 rows, 208 accumulator registers, one workgroup per CU) once the grid still fills the chip.
// Epilogues that also exist on 64-row tiles (the GEMMs of a streaming verify pass: a few thousand rows x 640 .. 1920 columns
// are 60 .. 130 tiles of 128 x 208 on 256 CUs; 64 x 80 / 64 x 160 tiles give 256 .. 400 workgroups, two to four per CU).
// The k-slices and the MFMA order per output element are those of the large tile: the same bits.
template <class Epi> constexpr bool kMidTiles = false;
template <> constexpr bool kMidTiles<EpiResidF32> = true;
template <> constexpr bool kMidTiles<EpiAct> = true;
template <> constexpr bool kMidTiles<EpiQkvRopeBf16> = true;
template <> constexpr bool kMidTiles<EpiSwiGLU> = true;

template <bool SWAP, class Epi>
int launch_tiled(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, Epi epi, hipStream_t s) {
  if ((K & 31) != 0 || (N & 3) != 0 || (lda & 7) != 0) throw std::runtime_error("gemm_tiled: unsupported shape");
  if constexpr (kMidTiles<Epi>) {
    static const bool mid_off = [] {   // A/B switch: MSH_GEMM_MID_TILES=0
      const char* e = dev_getenv("MSH_GEMM_MID_TILES");
      return e != nullptr && e[0] == '0';
    }();
    const long grid128 = (long)((M + 127) / 128) * ((N + 207) / 208);
    if (!mid_off && gemm_mode() == 2 && grid128 < 192 && M >= 64) {
      const long ntm = (M + 63) / 64;
      if (N % 160 == 0 && ntm * (N / 160) >= 192) return launch_tiled_dma_cfg<4, 1, 10, 3, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
      if (N % 80 == 0) return launch_tiled_dma_cfg<4, 1, 5, 3, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    }
  }
  const bool big = (long)((M + 255) / 256) * ((N + 207) / 208) >= 512;
  if (N % 208 == 0 || (N % 144 != 0 && N >= 416)) {  // ragged last column tile (e.g. the 32768-wide LM head) is predicated
    const int mode = gemm_mode();
    // short K with many column tiles (fc1, cross-KV): the A-stationary kernel wins (r01k: fc1 386 -> 452,
    // cross-KV 355 -> 389 TFLOP/s); with few column tiles (qkv, o-proj) its A preload is not amortised
    if (epi_taps<Epi>::value == 1 && !has_row_sums<Epi>::value && (mode == 4 || (mode == 2 && N >= 1664)) && K == 416 && M >= 2048 && N % 208 == 0)   // (conv GEMMs: the tiled kernels know their k-order)
      return launch_astat_cfg<13, 13, 3, SWAP, Epi>(A, lda, W, M, N, epi, s);
    if (mode == 0) {
      if (big)
        return launch_tiled_cfg<4, 13, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
      else
        return launch_tiled_cfg<2, 13, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    } else if (mode == 1 && big) {
      return launch_tiled_dma_cfg<4, 4, 13, 4, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    } else if (mode == 3 && big) {
      return launch_tiled_dma_cfg<8, 2, 13, 4, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    } else {
      return launch_tiled_dma_cfg<4, 2, 13, 3, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
    }
  } else if (N % 144 == 0) {
    return launch_tiled_cfg<2, 9, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
  } else {
    return launch_tiled_cfg<1, 4, SWAP, Epi>(A, lda, W, M, N, K, epi, s);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// developer probe MSH_STEM_STORE_NT=1: the conv1 / conv2 outputs (0.9 GB per 256 clips, written once, read by the next stem
// kernel) with the non-temporal policy, so that one lane's stem does not flush the decoder weights the other lanes are reading
static int stem_store_nt() {
  static const int v = [] {
    const char* e = dev_getenv("MSH_STEM_STORE_NT");
    return e != nullptr && e[0] == '1' ? 1 : 0;
  }();
  return v;
}
int gemm_tanh_bf16(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, bf16_t* out, float2* rowsum, hipStream_t s) {
  return launch_tiled<true>(A, lda, W, M, N, K, EpiTanhBf16{out, N, stem_store_nt(), rowsum}, s);
}
void gemm_gn_bias_gelu_bf16(const bf16_t* A, long lda, const bf16_t* W, const float* table, const float2* stats,
                            const int* row_clip, int M, int N, int K, bf16_t* out, int kperm, hipStream_t s) {
  if (K % (EpiGnBiasGeluBf16::kTaps * 32) != 0) throw std::runtime_error("conv2 GEMM: K must be 7 taps of a multiple of 32 channels");
  launch_tiled<true>(A, lda, W, M, N, K, EpiGnBiasGeluBf16{out, N, table, stats, row_clip, stem_store_nt(), kperm}, s);
}
void gemm_bias_gelu_bf16(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K,
                         bf16_t* out, hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiBiasGeluBf16{out, N, bias}, s);
}
void gemm_bias_gelu_f32(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, float* out,
                        int kperm, hipStream_t s) {
  if (K % (EpiBiasGeluF32::kTaps * 32) != 0) throw std::runtime_error("conv3 GEMM: K must be 3 taps of a multiple of 32 channels");
  launch_tiled<true>(A, lda, W, M, N, K, EpiBiasGeluF32{out, N, bias, kperm}, s);
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

void gemm_cross_kv_fp8(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, const int* row_clip,
                       const ClipMeta* clips, int D, long layer_stride, const float* qscale, uint8_t* KT, uint8_t* VT,
                       hipStream_t s) {
  launch_tiled<false>(A, lda, W, M, N, K, EpiCrossKVFp8{KT, VT, row_clip, clips, D, layer_stride, qscale}, s);
}

void gemm_act(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int act, int M, int N, int K,
              bf16_t* out_bf16, float* out_f32, hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiAct{out_bf16, out_f32, N, bias, act}, s);
}
void gemm_swiglu_bf16(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, bf16_t* z,
                      hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiSwiGLU{z, N / 2, bias}, s);
}
// LM head tile width: 128 columns at V = 32768, M = 256 gives 2 x 256 = 512 workgroups = exactly two per CU (the 208-wide
// tile of the encoder GEMMs gives 316: 60 CUs carry two workgroups, 196 carry one, and the launch waits for the 60):
// 19.2 -> 15.8 us per decode step
static constexpr int kArgmaxTN = 8;
int gemm_argmax_tiles(int N) { return (N + 16 * kArgmaxTN - 1) / (16 * kArgmaxTN); }
void gemm_argmax_partials(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, float* pval, int* pidx,
                          hipStream_t s) {
  if ((K & 31) != 0 || (lda & 7) != 0) throw std::runtime_error("gemm_argmax_partials: unsupported shape");
  static const bool w_nt = [] {   // developer probe: the 27 MB embedding with the non-temporal policy
    const char* e = dev_getenv("MSH_LMHEAD_NT");
    return e != nullptr && e[0] == '1';
  }();
  // 256 x 128 tiles from 129 rows on: at 256 clips the 27 MB embedding then crosses L2 once instead of once per 128-row tile
  // (15.4 -> 15.2 us alone, +0.4-0.8 % with four batches in flight; MSH_LMHEAD_TALL=0: the 128-row tile)
  static const bool tall = [] {
    const char* e = dev_getenv("MSH_LMHEAD_TALL");
    return !(e != nullptr && e[0] == '0');
  }();
  if (tall && M > 128) {
    launch_tiled_dma_cfg<4, 4, kArgmaxTN, 3, true, EpiArgmaxPartial>(A, lda, W, M, N, K,
                                                                     EpiArgmaxPartial{pval, pidx, gemm_argmax_tiles(N)}, s);
    return;
  }
  if (w_nt)
    launch_tiled_dma_cfg<4, 2, kArgmaxTN, 3, true, EpiArgmaxPartial, 32>(A, lda, W, M, N, K,
                                                                         EpiArgmaxPartial{pval, pidx, gemm_argmax_tiles(N)}, s);
  else
    launch_tiled_dma_cfg<4, 2, kArgmaxTN, 3, true, EpiArgmaxPartial>(A, lda, W, M, N, K,
                                                                     EpiArgmaxPartial{pval, pidx, gemm_argmax_tiles(N)}, s);
}
void gemm_logits_f32(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, float* out, hipStream_t s) {
  launch_tiled<true>(A, lda, W, M, N, K, EpiF32{out, N}, s);
}


// ------------------------------------------------------------------------------------------------
// Microbenchmark hook (tools/gemm_microbench.py): times one tiled-GEMM configuration on synthetic
// operands with a plain bf16 store epilogue.  cfg: 0 = 4 waves 128x208 3 stages (product default),
// 1 = 8 waves 256x208 4 stages, 2 = 4 waves 256x208 4 stages, 3 = 4 waves 128x208 2 stages,
// 4 = 4 waves 128x208 4 stages, 5 / 6 = A-stationary kernel with 3 / 4 stages (K = 416 only).  abl: see gemm_tiled_dma_kernel.
// ------------------------------------------------------------------------------------------------
namespace {
struct EpiBf16 {
  static constexpr bool kStagedBf16 = true;
  bf16_t* out;
  long ldc;
  struct RowCtx {};
  struct ColCtx {};
  __device__ RowCtx row_ctx(int) const { return RowCtx{}; }
  __device__ ColCtx col_ctx(int) const { return ColCtx{}; }
  __device__ void col_next16(ColCtx&) const {}
  __device__ uint2 pack4(const RowCtx&, const ColCtx&, int, f32x4 v) const {
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    return o;
  }
  __device__ void n4(int m, int n, f32x4 v) const {
    *reinterpret_cast<uint2*>(out + (long)m * ldc + n) = pack4(RowCtx{}, ColCtx{}, 0, v);
  }
};
__global__ void fill_bf16_kernel(bf16_t* p, long n, unsigned seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = f32_to_bf16(((float)(h & 0xffff) / 32768.0f - 1.0f));  // uniform [-1, 1)
  }
}
template <int NW, int TM, int NSTAGE>
void bench_launch(int abl, const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, bf16_t* C, hipStream_t s) {
  constexpr int BM = 16 * NW * TM, BN = 208;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN, nb = ntm * ntn;
  EpiBf16 epi{C, N};
#define MSH_BL(X)                                                                                                     \
  MSH_LAUNCH((gemm_tiled_dma_kernel<NW, TM, 13, NSTAGE, true, EpiBf16, X>), dim3(nb), dim3(64 * NW), 0, s, A, \
                     lda, W, M, N, K, ntn, nb, epi)
  switch (abl) {
    case 0: MSH_BL(0); break;
    case 1: MSH_BL(1); break;
    case 2: MSH_BL(2); break;
    case 3: MSH_BL(3); break;
    case 5: MSH_BL(5); break;
    case 6: MSH_BL(6); break;
    case 8: MSH_BL(8); break;
    case 16: MSH_BL(16); break;
    case 0x10400: MSH_BL(0x10400); break;
    case 0x20400: MSH_BL(0x20400); break;
    case 0x40400: MSH_BL(0x40400); break;
    case 0x10C00: MSH_BL(0x10C00); break;
    case 0x20C00: MSH_BL(0x20C00); break;
    case 0x40C00: MSH_BL(0x40C00); break;
    case 0x10100: MSH_BL(0x10100); break;
    case 0x20100: MSH_BL(0x20100); break;
    default: throw std::runtime_error("bad ablation");
  }
#undef MSH_BL
}
}  // namespace

// Probe: the same GEMM launched with `abl` bytes of extra dynamic LDS per workgroup, which moves the second
// co-resident workgroup's pipeline ring up in the CU's 160 KB; returns the number of output elements that differ
// from the launch without it.
static float gemm_lds_placement_probe(int M, int N, int K, long lda, int dyn_lds, int iters) {
  bf16_t *A = nullptr, *W = nullptr, *C = nullptr, *C2 = nullptr;
  const long a_elems = (long)M * lda + K + 64;
  MSH_HIP(hipMalloc(&A, a_elems * 2));
  MSH_HIP(hipMalloc(&W, (long)N * K * 2));
  MSH_HIP(hipMalloc(&C, (long)M * N * 2));
  MSH_HIP(hipMalloc(&C2, (long)M * N * 2));
  MSH_LAUNCH(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, A, a_elems, 1u);
  MSH_LAUNCH(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, W, (long)N * K, 7u);
  constexpr int BM = 128, BN = 208;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN, nb = ntm * ntn;
  MSH_LAUNCH((gemm_tiled_dma_kernel<4, 2, 13, 3, true, EpiBf16, 0>), dim3(nb), dim3(256), 0, 0, A, lda, W, M, N, K,
                     ntn, nb, EpiBf16{C, N});
  MSH_HIP(hipDeviceSynchronize());
  std::vector<uint16_t> ref((size_t)M * N), got((size_t)M * N);
  MSH_HIP(hipMemcpy(ref.data(), C, ref.size() * 2, hipMemcpyDeviceToHost));
  long bad = 0;
  for (int it = 0; it < iters; ++it) {
    MSH_HIP(hipMemset(C2, 0, (size_t)M * N * 2));
    MSH_LAUNCH((gemm_tiled_dma_kernel<4, 2, 13, 3, true, EpiBf16, 0>), dim3(nb), dim3(256), dyn_lds, 0, A, lda, W, M,
                       N, K, ntn, nb, EpiBf16{C2, N});
    MSH_HIP(hipDeviceSynchronize());
    MSH_HIP(hipMemcpy(got.data(), C2, got.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i];
  }
  (void)hipFree(A);
  (void)hipFree(W);
  (void)hipFree(C);
  (void)hipFree(C2);
  return (float)bad;
}

float gemm_microbench(int M, int N, int K, long lda, int cfg, int abl, int iters) {
  if (cfg == 10) return gemm_lds_placement_probe(M, N, K, lda, abl, iters);
  bf16_t *A = nullptr, *W = nullptr, *C = nullptr;
  const long a_elems = (long)M * lda + K + 64;
  MSH_HIP(hipMalloc(&A, a_elems * 2));
  MSH_HIP(hipMalloc(&W, (long)N * K * 2));
  MSH_HIP(hipMalloc(&C, (long)M * N * 2));
  MSH_LAUNCH(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, A, a_elems, 1u);
  MSH_LAUNCH(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, W, (long)N * K, 7u);
  MSH_HIP(hipDeviceSynchronize());
  auto run = [&] {
    switch (cfg) {
      case 0: bench_launch<4, 2, 3>(abl, A, lda, W, M, N, K, C, 0); break;
      case 1: bench_launch<8, 2, 4>(abl, A, lda, W, M, N, K, C, 0); break;
      case 2: bench_launch<4, 4, 4>(abl, A, lda, W, M, N, K, C, 0); break;
      case 3: bench_launch<4, 2, 2>(abl, A, lda, W, M, N, K, C, 0); break;
      case 4: bench_launch<4, 2, 4>(abl, A, lda, W, M, N, K, C, 0); break;
      case 5: launch_astat_cfg<13, 13, 3, true>(A, lda, W, M, N, EpiBf16{C, N}, 0); break;  // K must be 416
      case 6: launch_astat_cfg<13, 13, 4, true>(A, lda, W, M, N, EpiBf16{C, N}, 0); break;
      default: throw std::runtime_error("bad config");
    }
  };
  run();
  MSH_HIP(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  MSH_HIP(hipEventCreate(&e0));
  MSH_HIP(hipEventCreate(&e1));
  MSH_HIP(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) run();
  MSH_HIP(hipEventRecord(e1, 0));
  MSH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MSH_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(A);
  (void)hipFree(W);
  (void)hipFree(C);
  return ms / iters;
}

}  // namespace msh
