// Single-clip latency path of the offline decoder (batch <= 8) for gfx950.
//
// BASELINE.json quotes two things: throughput at 256 clips per GPU and the latency of ONE 10 s clip.  At one clip a decode
// step is ~58 dependent launches of which none fills the chip; what a step costs is (launch floor ~1.5 us + the memory round
// trips and the dependent instruction chain of each kernel) x launches.  The projected-form cross-attention was the outlier:
// one workgroup per (clip, head) = 8 workgroups on 256 CUs, each walking 415 keys behind a LayerNorm + query projection -- its
// chain was K request, 13 wave reductions for the query, two workgroup barriers around the score exchange, a wave maximum by
// ds_bpermute, the V request, an LDS reduction: 7.6-8.2 us per launch, 62 of the 217 us of a step.
//
//   dec_cross_split_kernel   one workgroup of 4 waves per (64-key slice, head, clip): 7 x 8 = 56 workgroups for a 10 s clip.
//                            Every load is requested up front (the clip's residual row, wave w's 16 rows of Wq, wave w's 16
//                            keys of K^T and V^T): ONE round trip.  LayerNorm on eight values per lane, the normalised row
//                            handed to the MFMA operand order through LDS, the query projection as 13 MFMAs per wave, q
//                            exchanged through LDS (barrier 1), then wave w: scores / softmax / P.V of ITS 16 keys on the VALU
//                            with DPP and row-swap reductions; the four waves' (max, sum, output) merged by wave 0 (barrier 2)
//                            into the slice's record -- flash-decoding's split over the keys.
//                            (The first version did all of it in ONE wave: no barrier at all, but ~2000 instructions issued by a
//                            single wave, 416 registers: 6.3 us, of which 4.1 us with everything but the loads ablated.)
//   dec_merge_resid_kernel   the cross-attention output projection with the merge of those records as its A-operand
//                            prologue: H += merge(records) Wo^T.  Same split-K MFMA body, FM weight loads and fixed-order
//                            reduction as gemm_dec_kernel (k_gemm_dec.hip).  Slices beyond a clip's frames are written as empty
//                            records, so the merge needs no clip metadata: its loads are one round trip.
// Reference: the decoder's cross-attention inside the ORT graph run at core/moonshine-model.cpp:380-517; float definition
// modeling_moonshine.py:265-330 (encoder_attn of MoonshineDecoderLayer).  Numerics: the same roundings as the one-workgroup kernel
// of k_attn.hip (LayerNorm output and attention output rounded to bf16, fp32 softmax in the exp2 domain); the sums over k and over
// the keys are taken in a different order (logits agree to ~3e-3, tests/test_gpu_dec_small.py).
#include <stdlib.h>

#include <stdexcept>
#include <string>

#include "gemm_common.h"

namespace msh {
namespace {

constexpr int XS_KEYS = 64;     // keys per slice
constexpr int XS_REC = 64;      // floats per record: [0] = max (log2 domain; -inf = empty), [1] = sum, [4 .. 4 + DH) = output

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bflo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
#define MSH_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, true))
constexpr int kRor1 = 0x121, kRor2 = 0x122, kRor4 = 0x124, kRor8 = 0x128;   // row_ror:n (inside the 16-lane row)
constexpr int kXor1 = 0xB1;                                                  // quad_perm [1, 0, 3, 2]
constexpr int kXor2 = 0x4E;                                                  // quad_perm [2, 3, 0, 1]
// sum over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48), every lane gets the result
__device__ __forceinline__ float rows4_sum(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned w = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float wave_sum(float v) {   // fixed order; every lane gets the result
  v += MSH_DPP(v, kRor8);
  v += MSH_DPP(v, kRor4);
  v += MSH_DPP(v, kRor2);
  v += MSH_DPP(v, kRor1);
  return rows4_sum(v);
}
// lanes are (rs = lane >> 1, kg = lane & 1): sum over rs, i.e. over lane bits 1..5, separately for the two kg
__device__ __forceinline__ float sum_over_rs(float v) {
  v += MSH_DPP(v, kXor2);   // bit 1
  v += MSH_DPP(v, kRor4);   // bits 2 and 3: the four lanes i, i - 4, i - 8, i - 12 of the row
  v += MSH_DPP(v, kRor8);
  return rows4_sum(v);      // bits 4 and 5
}

// The merge over a (clip, head)'s slice records, four consecutive output dims per caller: (sum_s w_s o_s[d .. d + 3]) / sum_s w_s l_s
// with w_s = 2^(m_s - max), as packed bf16.  ONE definition for both callers -- the merging projection reads the records from
// global memory (n = the launch's slice count, empty records beyond a clip's frames), the looped kernel from LDS (n = the clip's
// own count): empty records add exact zeros and a round of eight without a new maximum rescales by exactly 1, so the two give
// the same bits.
__device__ __forceinline__ uint2 merge_slices(const float* rec, int n, int d) {
  float mx = -INFINITY, L = 0.f;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < n; s0 += 8) {   // eight slices per round: their loads are independent
    float ms[8], ls[8];
    float4 os[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool in = s0 + j < n;
      const float* r = rec + (long)(in ? s0 + j : s0) * XS_REC;
      const float2 ml = *reinterpret_cast<const float2*>(r);
      ms[j] = in ? ml.x : -INFINITY;
      ls[j] = ml.y;
      os[j] = *reinterpret_cast<const float4*>(r + 4 + d);
    }
    float mn = mx;
#pragma unroll
    for (int j = 0; j < 8; ++j) mn = fmaxf(mn, ms[j]);
    const float mr = mn > -INFINITY ? mn : 0.f;
    const float sc = __builtin_amdgcn_exp2f(mx - mr);   // first round: 2^-inf = 0
    L *= sc;
    a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w = __builtin_amdgcn_exp2f(ms[j] - mr);   // empty / absent slices: 0
      L += w * ls[j];
      a.x += w * os[j].x; a.y += w * os[j].y; a.z += w * os[j].z; a.w += w * os[j].w;
    }
    mx = mn;
  }
  const float inv = 1.0f / L;   // slice 0 of a clip is never empty: L > 0
  uint2 pk;
  pk.x = pack_bf16x2(a.x * inv, a.y * inv);
  pk.y = pack_bf16x2(a.z * inv, a.w * inv);
  return pk;
}

// grid (ns_max, heads, clips), 256 threads.  H: FM fp32 residual stream [M16][D]; Wq: row-major [D][D] bf16 with the LayerNorm
// scale folded in; KT / VT: [dh][Tk] bf16 per (clip, head), keys contiguous; part: [clip][head][ns_max][XS_REC] fp32.
// LOOP (batches too large for one workgroup per slice: 5 .. 63 clips): grid (1, heads, clips); the workgroup walks the clip's
// slices itself (the next slice's K / V requested while the current one is computed), keeps the records in LDS, merges them
// with the projection kernel's own function and writes the attention output as the FM bf16 operand of dec_gemm_resid -- the
// same arithmetic slice by slice, hence the same bits as the one-workgroup-per-slice form.
constexpr int XS_LOOP_MAX = 32;   // slices a looping workgroup keeps (2048 frames = 49 s of audio; longer clips: k_attn.hip's kernel)
template <int DH, int KS, bool LOOP>
__global__ __launch_bounds__(256) void dec_cross_split_kernel(const float* __restrict__ H, const bf16_t* __restrict__ Wq,
                                                              const bf16_t* __restrict__ KT, const bf16_t* __restrict__ VT,
                                                              const ClipMeta* __restrict__ clips, int heads, int ns_max,
                                                              float* __restrict__ part, bf16_t* __restrict__ out) {
  constexpr int D = 32 * KS;
  constexpr int NTQ = (DH + 15) / 16;   // 16-row tiles of the head's query: wave w computes tile w
  constexpr int NR = (DH + 31) / 32;    // K / V rows per lane: d = rs + 32 i
  static_assert(DH + 4 <= XS_REC && NTQ <= 4 && D / 8 <= 64, "shape not covered");
  __shared__ __attribute__((aligned(16))) bf16_t xs[4][D];   // the normalised row, one private copy per wave
  __shared__ float qs[64];
  // the waves' partials: one set per slice when looping (merged after the loop, no barrier inside it), else one set
  __shared__ float wp[LOOP ? XS_LOOP_MAX : 1][4][XS_REC];
  __shared__ __attribute__((aligned(16))) float recs[LOOP ? XS_LOOP_MAX : 1][XS_REC];
  const int s_first = LOOP ? 0 : blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kg4 = lane >> 4, rs = lane >> 1, kg = lane & 1;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- every load of the wave, up front; the clip's geometry (a scalar load) is only needed for the K / V addresses ----
  const bool xact = lane * 8 < D;
  float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
  if (xact) {
    xa = *reinterpret_cast<const float4*>(H + fm32(b, lane * 8, KS));
    xb = *reinterpret_cast<const float4*>(H + fm32(b, lane * 8 + 4, KS));
  }
  uint4 wq[KS];
  {
    int d = 16 * wave + li;
    d = d < DH ? d : DH - 1;
    const bf16_t* wrow = Wq + (long)(h * DH + d) * D + 8 * kg4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wq[ks] = *reinterpret_cast<const uint4*>(wrow + 32 * ks);
  }
  const ClipMeta cm = clips[b];
  const int T = cm.T, Tk = cm.Tk;
  const long off = (long)cm.kv_start * D + (long)(h * DH) * Tk;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(KT + off), 0, DH * Tk * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(VT + off), 0, DH * Tk * 2, 0x00020000);
  const int s_end = LOOP ? (T + XS_KEYS - 1) / XS_KEYS : s_first + 1;
  const int kw0 = wave * 16 + kg * 8;   // the lane's 8 keys inside a slice
  u32x4v kr[NR], vr[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (rs + 32 * i) * Tk * 2 + (s_first * XS_KEYS + kw0) * 2, 0, 0);   // rows >= DH: out of range = 0
#pragma unroll
  for (int i = 0; i < NR; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, (rs + 32 * i) * Tk * 2 + (s_first * XS_KEYS + kw0) * 2, 0, 0);
  __builtin_amdgcn_sched_barrier(0);

  // ---- LayerNorm of the clip's row (every wave, redundantly: lane l holds columns 8 l .. 8 l + 7), two-pass ----
  float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
  const float mean = wave_sum(((xv[0] + xv[1]) + (xv[2] + xv[3])) + ((xv[4] + xv[5]) + (xv[6] + xv[7]))) * (1.0f / (float)D);
  float sq = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    xv[e] = xact ? xv[e] - mean : 0.f;
    sq += xv[e] * xv[e];
  }
  const float rstd = rsqrtf(wave_sum(sq) * (1.0f / (float)D) + 1e-5f);
  if (xact) {
    uint4 p;
    p.x = pack_bf16x2(xv[0] * rstd, xv[1] * rstd);
    p.y = pack_bf16x2(xv[2] * rstd, xv[3] * rstd);
    p.z = pack_bf16x2(xv[4] * rstd, xv[5] * rstd);
    p.w = pack_bf16x2(xv[6] * rstd, xv[7] * rstd);
    *reinterpret_cast<uint4*>(&xs[wave][lane * 8]) = p;
  }
  __builtin_amdgcn_wave_barrier();   // (LDS operations of one wave complete in order: the reads below see the row)
  // ---- q tile of this wave: rows 16 wave + ..., the row as the B operand (the same in all 16 columns) ----
  if (wave < NTQ) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 bq = *reinterpret_cast<const uint4*>(&xs[wave][32 * ks + 8 * kg4]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wq[ks]), *reinterpret_cast<const bf16x8*>(&bq), acc, 0, 0, 0);
    }
    // lane (column li, row group kg4) holds q[16 wave + 4 kg4 + r], equal in every column: column 0 hands them over
    const float c = rsqrtf((float)DH) * 1.4426950408889634f;
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) qs[16 * wave + 4 * kg4 + r] = acc[r] * c;
    }
  }
  __syncthreads();

  // ---- wave w: its 16 keys of every slice.  Lane (rs, kg): rows rs, rs + 32 of keys key0 .. key0 + 7 ----
  float qd[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) qd[i] = rs + 32 * i < DH ? qs[rs + 32 * i] : 0.f;
#pragma unroll 1
  for (int s = s_first; s < s_end; ++s) {
    u32x4v kn[NR], vn[NR];
    if constexpr (LOOP) {   // the next slice's rows fly while this one is computed (past the last slice: in range, masked, unused)
#pragma unroll
      for (int i = 0; i < NR; ++i) kn[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (rs + 32 * i) * Tk * 2 + ((s + 1) * XS_KEYS + kw0) * 2, 0, 0);
#pragma unroll
      for (int i = 0; i < NR; ++i) vn[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, (rs + 32 * i) * Tk * 2 + ((s + 1) * XS_KEYS + kw0) * 2, 0, 0);
    }
    const int key0 = s * XS_KEYS + kw0;
    float sc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const u32x4v u = kr[i];
      sc[0] += qd[i] * bflo(u.x); sc[1] += qd[i] * bfhi(u.x);
      sc[2] += qd[i] * bflo(u.y); sc[3] += qd[i] * bfhi(u.y);
      sc[4] += qd[i] * bflo(u.z); sc[5] += qd[i] * bfhi(u.z);
      sc[6] += qd[i] * bflo(u.w); sc[7] += qd[i] * bfhi(u.w);
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = sum_over_rs(sc[e]);
      sc[e] = key0 + e < T ? sc[e] : -INFINITY;
      mloc = fmaxf(mloc, sc[e]);
    }
    const float m = fmaxf(mloc, MSH_DPP(mloc, kXor1));   // the other 8 keys of the wave
    const bool any = m > -INFINITY;                      // (a wave whose 16 keys all lie beyond the clip: an empty partial)
    const float mref = any ? m : 0.f;
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = __builtin_amdgcn_exp2f(sc[e] - mref);   // masked keys: 2^-inf = 0
      psum += sc[e];
    }
    const float l = psum + MSH_DPP(psum, kXor1);
    float (*wps)[XS_REC] = wp[LOOP ? s : 0];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const u32x4v u = vr[i];
      float o = sc[0] * bflo(u.x) + sc[1] * bfhi(u.x) + sc[2] * bflo(u.y) + sc[3] * bfhi(u.y) +
                sc[4] * bflo(u.z) + sc[5] * bfhi(u.z) + sc[6] * bflo(u.w) + sc[7] * bfhi(u.w);
      o += MSH_DPP(o, kXor1);
      if (kg == 0 && rs + 32 * i < DH) wps[wave][4 + rs + 32 * i] = o;
    }
    if (lane == 0) {
      wps[wave][0] = m;
      wps[wave][1] = l;
    }
    if constexpr (LOOP) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        kr[i] = kn[i];
        vr[i] = vn[i];
      }
    }
  }
  __syncthreads();
  // ---- a slice's record: the four waves' partials merged (fixed order); looping: wave w takes slices w, w + 4, ... ----
  for (int s = s_first + (LOOP ? wave : 0); s < s_end; s += LOOP ? 4 : 1) {
    if (!LOOP && wave != 0) break;
    float (*wps)[XS_REC] = wp[LOOP ? s : 0];
    const float m0 = wps[0][0], m1 = wps[1][0], m2 = wps[2][0], m3 = wps[3][0];
    const float mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    const float mr = mx > -INFINITY ? mx : 0.f;
    const float w0 = __builtin_amdgcn_exp2f(m0 - mr), w1 = __builtin_amdgcn_exp2f(m1 - mr);
    const float w2 = __builtin_amdgcn_exp2f(m2 - mr), w3 = __builtin_amdgcn_exp2f(m3 - mr);
    float* rec = LOOP ? &recs[s][0] : part + (((long)b * heads + h) * ns_max + s) * XS_REC;
    if (lane < DH) rec[4 + lane] = (w0 * wps[0][4 + lane] + w1 * wps[1][4 + lane]) + (w2 * wps[2][4 + lane] + w3 * wps[3][4 + lane]);
    if (lane == 0) {
      rec[0] = mx;   // -inf: a slice beyond the clip's frames (weight 0 in the merge)
      rec[1] = (w0 * wps[0][1] + w1 * wps[1][1]) + (w2 * wps[2][1] + w3 * wps[3][1]);
    }
  }
  if constexpr (LOOP) {
    __syncthreads();   // the records are complete
    if (tid < DH / 4) {
      const uint2 pk = merge_slices(&recs[0][0], s_end, 4 * tid);
      *reinterpret_cast<uint2*>(out + fm16(b, h * DH + 4 * tid, KS)) = pk;   // FM: the o-proj GEMM's A operand
    }
  }
}

// H[M16][D] (FM fp32) += merge(records)[M][D] x Wo^T.  grid = (D / 16 column tiles, M / 16 row tiles), 256 threads.
template <int DH, int KS>
__global__ __launch_bounds__(256) void dec_merge_resid_kernel(const float* __restrict__ part, const bf16_t* __restrict__ W, int M,
                                                              int heads, int ns_max, float* __restrict__ H) {
  constexpr int D = 32 * KS, NW = 4;
  constexpr int KW = (KS + NW - 1) / NW, KFULL = KS / NW;
  constexpr int LDA = D + 8;   // bf16 elements per staged row: 16-byte aligned rows, 4-bank skew between them
  __shared__ __attribute__((aligned(16))) bf16_t as[16][LDA];
  __shared__ __attribute__((aligned(16))) float4 psum[NW][64];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x, n0 = nt * 16, m0 = blockIdx.y * 16;
  const int Mt = M - m0 < 16 ? M - m0 : 16;   // rows of this tile
  int ks[KW];
  bool kv[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int s = wave + NW * i;
    kv[i] = i < KFULL ? true : s < KS;
    ks[i] = kv[i] ? s : KS - 1;
  }
  // ---- loads that do not depend on the records: the weight fragments and the residual ----
  uint4 wreg[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) wreg[i] = *reinterpret_cast<const uint4*>(W + (((long)nt * KS + ks[i]) * 64 + lane) * 8);
  const int mrow = li < Mt ? li : Mt - 1, ncol = n0 + kg * 4;
  const float4 hres = *reinterpret_cast<const float4*>(H + fm32(m0 + mrow, ncol, KS));
  // ---- merge: element (m, k) = sum_s w_s o_s[d] / sum_s w_s l_s, w_s = 2^(m_s - max), head = k / DH; a thread takes four
  // consecutive columns (DH % 4 == 0: one head) ----
  static_assert(DH % 4 == 0, "a thread's four columns must lie in one head");
  for (int idx = tid; idx < Mt * (D / 4); idx += 256) {
    const int m = idx / (D / 4), k = 4 * (idx - m * (D / 4));
    const int hh = k / DH, d = k - hh * DH;
    const float* rec = part + (((long)(m0 + m) * heads + hh) * ns_max) * XS_REC;
    const uint2 pk = merge_slices(rec, ns_max, d);
    *reinterpret_cast<uint2*>(&as[m][k]) = pk;
  }
  __syncthreads();
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    uint4 q = *reinterpret_cast<const uint4*>(&as[mrow][32 * ks[i] + 8 * kg]);   // rows >= M: a copy of the last one, never stored
    if (i >= KFULL) {
      q.x = kv[i] ? q.x : 0u; q.y = kv[i] ? q.y : 0u; q.z = kv[i] ? q.z : 0u; q.w = kv[i] ? q.w : 0u;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wreg[i]), *reinterpret_cast<const bf16x8*>(&q), acc, 0, 0, 0);
  }
  psum[wave][lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (wave == 0 && li < Mt) {
    const float4 p0 = psum[0][lane], p1 = psum[1][lane], p2 = psum[2][lane], p3 = psum[3][lane];
    *reinterpret_cast<float4*>(H + fm32(m0 + li, ncol, KS)) =
        make_float4(hres.x + ((p0.x + p1.x) + (p2.x + p3.x)), hres.y + ((p0.y + p1.y) + (p2.y + p3.y)),
                    hres.z + ((p0.z + p1.z) + (p2.z + p3.z)), hres.w + ((p0.w + p1.w) + (p2.w + p3.w)));
  }
}


// ------------------------------------------------------------------------------------------------
// Self-attention + output projection in ONE launch (one clip by default; the kernel takes two): H += selfattn(q, cache) Wo^T.
// grid = D / 16 column tiles, 512 threads.  Wave h of EVERY workgroup computes head h of the clip(s) -- at one clip that is 26 x
// the work of the separate kernel, all of it out of the L2 (a layer's cache at 66 keys is 110 KB), and it removes a launch and a
// round trip through HBM from a chain whose every link costs >= 2 us.  The wave-level attention is that of k_attn.hip's
// dec_self_attention_kernel (K / V runs of a 72-key block copied to a wave-private LDS slab by LDS-DMA, scores with lane = key,
// fp32 softmax in the exp2 domain, online across blocks, P.V with lane = (key group, 4-dim piece), fixed-order reduction); its
// output row goes to LDS as the A operand of the split-K MFMA body (8 waves over the k-steps, fixed-order reduction).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char lds_char_s;
__device__ __forceinline__ unsigned lds_addr_s(const void* p) { return (unsigned)(unsigned long)(lds_char_s*)(p); }
__device__ __forceinline__ void lds_dma16_s(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_base)
               : "memory");
}
__device__ __forceinline__ float rows4_max(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const unsigned w = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, MSH_DPP(v, kRor8));
  v = fmaxf(v, MSH_DPP(v, kRor4));
  v = fmaxf(v, MSH_DPP(v, kRor2));
  v = fmaxf(v, MSH_DPP(v, kRor1));
  return rows4_max(v);
}

template <int DH>
struct SelfCfg {
  static constexpr int KB = 72;
  static constexpr int TILE_BYTES = KB * DH * 2;
  static constexpr int NCH = (TILE_BYTES + 1023) / 1024;
  static constexpr int TILE_PAD = NCH * 1024;
  static constexpr int PIECES = DH / 4, G = 64 / PIECES;
};

template <int DH, int KS>
__global__ __launch_bounds__(512) void dec_self_oproj_kernel(const float* __restrict__ q, const bf16_t* __restrict__ cacheK,
                                                             const bf16_t* __restrict__ cacheV, const int* __restrict__ pos_ptr,
                                                             const bf16_t* __restrict__ W, int M, int heads, int Smax,
                                                             float* __restrict__ H) {
  using C = SelfCfg<DH>;
  // NA = attention waves (one per head), NW = waves of the GEMM body: the first four, with gemm_dec_kernel's assignment of
  // k-steps and its summation order -- H comes out bit-identical to dec_self_attention + dec_gemm_resid, so a clip decoded
  // alone (this kernel) and inside a larger batch (those two) gives the same ids
  constexpr int D = 32 * KS, NA = 8, NW = 4, KB = C::KB, NCH = C::NCH, PIECES = C::PIECES, G = C::G;
  constexpr int KW = (KS + NW - 1) / NW, KFULL = KS / NW;
  constexpr int LDA = D + 8;
  __shared__ __attribute__((aligned(16))) unsigned char tiles[NA][2][C::TILE_PAD];
  __shared__ float scs[NA][KB + 8];
  __shared__ float4 red[NA][G][PIECES];
  __shared__ __attribute__((aligned(16))) bf16_t as[2][LDA];
  __shared__ __attribute__((aligned(16))) float4 psum[NW][64];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x, n0 = nt * 16;
  const int gw = wave < NW ? wave : NW - 1;   // (waves 4..7 mirror wave 3's loads and drop the result)
  int ks[KW];
  bool kv[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int s = gw + NW * i;
    kv[i] = i < KFULL ? true : s < KS;
    ks[i] = kv[i] ? s : KS - 1;
  }
  uint4 wreg[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) wreg[i] = *reinterpret_cast<const uint4*>(W + (((long)nt * KS + ks[i]) * 64 + lane) * 8);
  const int mrow = li < M ? li : M - 1, ncol = n0 + kg * 4;
  const float4 hres = *reinterpret_cast<const float4*>(H + fm32(mrow, ncol, KS));
  const int S = *pos_ptr + 1;

  if (wave < heads) {
    const int h = wave;
    const unsigned kt = __builtin_amdgcn_readfirstlane(lds_addr_s(&tiles[wave][0][0]));
    const unsigned vt = __builtin_amdgcn_readfirstlane(lds_addr_s(&tiles[wave][1][0]));
    const bf16_t* Kl = reinterpret_cast<const bf16_t*>(&tiles[wave][0][0]);
    const bf16_t* Vl = reinterpret_cast<const bf16_t*>(&tiles[wave][1][0]);
    const float c = rsqrtf((float)DH) * 1.4426950408889634f;
    const long run_bytes = (long)S * DH * 2;
    for (int m = 0; m < M; ++m) {
      const long pair = (long)m * heads + h;
      const unsigned char* kp = reinterpret_cast<const unsigned char*>(cacheK + pair * Smax * DH);
      const unsigned char* vp = reinterpret_cast<const unsigned char*>(cacheV + pair * Smax * DH);
      auto stage = [&](int k0) {
        const long base = (long)k0 * DH * 2;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const long off = base + i * 1024 + lane * 16;
          if (i * 1024 + lane * 16 < C::TILE_BYTES && off < run_bytes) {
            lds_dma16_s(kp + off, kt + i * 1024);
            lds_dma16_s(vp + off, vt + i * 1024);
          }
        }
      };
      stage(0);
      const float* qp = q + (long)m * D + h * DH;
      float qreg[DH];
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
        if (k0 > 0) stage(k0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int nk = S - k0 < KB ? S - k0 : KB;
        float s0 = -INFINITY, s1 = -INFINITY;
        if (lane < nk) {
          const uint2* kr = reinterpret_cast<const uint2*>(Kl + lane * DH);
          float a = 0.f;
#pragma unroll
          for (int d = 0; d < DH; d += 4) {
            const uint2 u = kr[d >> 2];
            a += qreg[d] * bflo(u.x) + qreg[d + 1] * bfhi(u.x) + qreg[d + 2] * bflo(u.y) + qreg[d + 3] * bfhi(u.y);
          }
          s0 = a;
        }
        if (lane + 64 < nk) {
          const uint2* kr = reinterpret_cast<const uint2*>(Kl + (lane + 64) * DH);
          float a = 0.f;
#pragma unroll
          for (int d = 0; d < DH; d += 4) {
            const uint2 u = kr[d >> 2];
            a += qreg[d] * bflo(u.x) + qreg[d + 1] * bfhi(u.x) + qreg[d + 2] * bflo(u.y) + qreg[d + 3] * bfhi(u.y);
          }
          s1 = a;
        }
        const float m_new = fmaxf(m_run, wave_max(fmaxf(s0, s1)));
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        const float p0 = __builtin_amdgcn_exp2f(s0 - m_new), p1 = __builtin_amdgcn_exp2f(s1 - m_new);
        scs[wave][lane] = p0;
        if (lane < KB - 64) scs[wave][lane + 64] = p1;
        l_run = l_run * alpha + wave_sum(p0 + p1);
        __builtin_amdgcn_wave_barrier();
        acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
        if (g < G) {
#pragma unroll 6
          for (int s = g; s < nk; s += G) {
            const uint2 u = *reinterpret_cast<const uint2*>(Vl + s * DH + piece * 4);
            const float p = scs[wave][s];
            acc.x += p * bflo(u.x);
            acc.y += p * bfhi(u.x);
            acc.z += p * bflo(u.y);
            acc.w += p * bfhi(u.y);
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
        *reinterpret_cast<uint2*>(&as[m][h * DH + lane * 4]) = o;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    uint4 qf = *reinterpret_cast<const uint4*>(&as[mrow][32 * ks[i] + 8 * kg]);
    if (i >= KFULL) {
      qf.x = kv[i] ? qf.x : 0u; qf.y = kv[i] ? qf.y : 0u; qf.z = kv[i] ? qf.z : 0u; qf.w = kv[i] ? qf.w : 0u;
    }
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wreg[i]), *reinterpret_cast<const bf16x8*>(&qf), acc2, 0, 0, 0);
  }
  if (wave < NW) psum[wave][lane] = make_float4(acc2[0], acc2[1], acc2[2], acc2[3]);
  __syncthreads();
  if (wave == 0 && li < M) {
    const float4 p0 = psum[0][lane], p1 = psum[1][lane], p2 = psum[2][lane], p3 = psum[3][lane];
    // EpiDecResidFm's association: residual + bias (none) + sum
    *reinterpret_cast<float4*>(H + fm32(li, ncol, KS)) =
        make_float4(hres.x + 0.f + ((p0.x + p1.x) + (p2.x + p3.x)), hres.y + 0.f + ((p0.y + p1.y) + (p2.y + p3.y)),
                    hres.z + 0.f + ((p0.z + p1.z) + (p2.z + p3.z)), hres.w + 0.f + ((p0.w + p1.w) + (p2.w + p3.w)));
  }
}

}  // namespace

bool dec_cross_split_supported(int D, int heads) {
  const int dh = heads > 0 ? D / heads : 0;
  return (D == 416 && dh == 52) || (D == 288 && dh == 36) || (D == 64 && dh == 16);
}
int dec_cross_split_slices(int T) { return (T + XS_KEYS - 1) / XS_KEYS; }
size_t dec_cross_split_part_floats(int M, int heads, int ns_max) { return (size_t)M * heads * ns_max * XS_REC; }

void dec_cross_split(const float* H, const bf16_t* Wq_rm, const bf16_t* KT, const bf16_t* VT, const ClipMeta* clips, int M, int D,
                     int heads, int ns_max, float* part, hipStream_t s) {
  if (M <= 0) return;
  const dim3 grid(ns_max, heads, M);
  bf16_t* none = nullptr;
  switch (D) {
    case 416: MSH_LAUNCH((dec_cross_split_kernel<52, 13, false>), grid, dim3(256), 0, s, H, Wq_rm, KT, VT, clips, heads, ns_max, part, none); break;
    case 288: MSH_LAUNCH((dec_cross_split_kernel<36, 9, false>), grid, dim3(256), 0, s, H, Wq_rm, KT, VT, clips, heads, ns_max, part, none); break;
    case 64: MSH_LAUNCH((dec_cross_split_kernel<16, 2, false>), grid, dim3(256), 0, s, H, Wq_rm, KT, VT, clips, heads, ns_max, part, none); break;
    default: throw std::runtime_error("dec_cross_split: unsupported width " + std::to_string(D));
  }
}

int dec_cross_looped_max_slices() { return XS_LOOP_MAX; }
void dec_cross_looped(const float* H, const bf16_t* Wq_rm, const bf16_t* KT, const bf16_t* VT, const ClipMeta* clips, int M, int D,
                      int heads, bf16_t* out_fm, hipStream_t s) {
  if (M <= 0) return;
  const dim3 grid(1, heads, M);
  float* none = nullptr;
  switch (D) {
    case 416: MSH_LAUNCH((dec_cross_split_kernel<52, 13, true>), grid, dim3(256), 0, s, H, Wq_rm, KT, VT, clips, heads, 0, none, out_fm); break;
    case 288: MSH_LAUNCH((dec_cross_split_kernel<36, 9, true>), grid, dim3(256), 0, s, H, Wq_rm, KT, VT, clips, heads, 0, none, out_fm); break;
    case 64: MSH_LAUNCH((dec_cross_split_kernel<16, 2, true>), grid, dim3(256), 0, s, H, Wq_rm, KT, VT, clips, heads, 0, none, out_fm); break;
    default: throw std::runtime_error("dec_cross_looped: unsupported width " + std::to_string(D));
  }
}

void dec_merge_resid(const float* part, const bf16_t* Wo_fm, int M, int D, int heads, int ns_max, float* H, hipStream_t s) {
  if (M <= 0) return;
  const dim3 grid(D / 16, (M + 15) / 16);
  switch (D) {
    case 416: MSH_LAUNCH((dec_merge_resid_kernel<52, 13>), grid, dim3(256), 0, s, part, Wo_fm, M, heads, ns_max, H); break;
    case 288: MSH_LAUNCH((dec_merge_resid_kernel<36, 9>), grid, dim3(256), 0, s, part, Wo_fm, M, heads, ns_max, H); break;
    case 64: MSH_LAUNCH((dec_merge_resid_kernel<16, 2>), grid, dim3(256), 0, s, part, Wo_fm, M, heads, ns_max, H); break;
    default: throw std::runtime_error("dec_merge_resid: unsupported width " + std::to_string(D));
  }
}

bool dec_self_oproj_supported(int D, int heads, int M) {
  return dec_cross_split_supported(D, heads) && heads <= 8 && M >= 1 && M <= 2;
}
void dec_self_oproj(const float* q, const bf16_t* cacheK, const bf16_t* cacheV, const int* pos_ptr, const bf16_t* Wo_fm, int M, int D,
                    int heads, int Smax, float* H, hipStream_t s) {
  if (!dec_self_oproj_supported(D, heads, M)) throw std::runtime_error("dec_self_oproj: unsupported shape");
  const dim3 grid(D / 16);
  switch (D) {
    case 416: MSH_LAUNCH((dec_self_oproj_kernel<52, 13>), grid, dim3(512), 0, s, q, cacheK, cacheV, pos_ptr, Wo_fm, M, heads, Smax, H); break;
    case 288: MSH_LAUNCH((dec_self_oproj_kernel<36, 9>), grid, dim3(512), 0, s, q, cacheK, cacheV, pos_ptr, Wo_fm, M, heads, Smax, H); break;
    case 64: MSH_LAUNCH((dec_self_oproj_kernel<16, 2>), grid, dim3(512), 0, s, q, cacheK, cacheV, pos_ptr, Wo_fm, M, heads, Smax, H); break;
    default: throw std::runtime_error("dec_self_oproj: unsupported width " + std::to_string(D));
  }
}

}  // namespace msh
