// Streaming-path kernels for gfx950 (everything that is not a GEMM): frame normalisation, the sliding-window
// encoder attention, row-based decoder attention over per-stream caches, and the device-side bookkeeping
// of the speculative decode.  Rows of a decoder pass are (stream, position) pairs, so one launch serves
// the wide verify pass (many rows per stream) and the auto-regressive steps (one row per stream) alike.
//
// These kernels are small HBM/latency-bound pieces: one 64-lane wave per (row, head) for the short
// attentions, one workgroup per (row, head) for cross-attention, wave-level shuffles for every reduction.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include <stdlib.h>

#include "stream_kernels.h"

namespace msh {
namespace {

// Full-wave reductions on the VALU: four DPP row rotations inside the 16-lane rows, then the gfx950 row swaps (permlane16 /
// permlane32) across them -- no trip through the LDS crossbar (__shfl_xor is ds_bpermute + a wait on lgkmcnt, six times per
// reduction, on the critical path of every attention kernel here).  Fixed order, every lane gets the result; all callers reduce
// at wave-uniform points.  (The same forms as k_attn.hip's wave_sum_dpp / wave_max_dpp.)
#define MSH_S_ROR(v, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xf, 0xf, true))
__device__ __forceinline__ float wsum(float v) {
  v += MSH_S_ROR(v, 8);
  v += MSH_S_ROR(v, 4);
  v += MSH_S_ROR(v, 2);
  v += MSH_S_ROR(v, 1);
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned w = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float wmax(float v) {
  v = fmaxf(v, MSH_S_ROR(v, 8));
  v = fmaxf(v, MSH_S_ROR(v, 4));
  v = fmaxf(v, MSH_S_ROR(v, 2));
  v = fmaxf(v, MSH_S_ROR(v, 1));
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const unsigned w = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float bf(bf16_t h) { return bf16_to_f32(h); }

// dot of an fp32 vector in LDS with a bf16 row in global memory, n % 4 == 0
__device__ __forceinline__ float dot_row(const float* __restrict__ q, const bf16_t* __restrict__ k, int n) {
  float acc = 0.f;
  for (int d = 0; d < n; d += 4) {
    const uint2 r = *reinterpret_cast<const uint2*>(k + d);
    acc += q[d] * __uint_as_float(r.x << 16) + q[d + 1] * __uint_as_float(r.x & 0xffff0000u) +
           q[d + 2] * __uint_as_float(r.y << 16) + q[d + 3] * __uint_as_float(r.y & 0xffff0000u);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// one wave per 80-sample frame: mean / rms over the frame, asinh compression, bf16 row of 96
__global__ __launch_bounds__(256) void frames_kernel(const float* __restrict__ audio, const FrameJob* __restrict__ jobs,
                                                     float k, bf16_t* __restrict__ frames) {
  const FrameJob job = jobs[blockIdx.y];
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (f >= job.n_frames) return;
  const float* a = audio + job.audio_off + (long)f * 80;
  const float x0 = a[lane];
  const float x1 = lane < 16 ? a[64 + lane] : 0.f;
  const float mean = wsum(x0 + x1) * (1.0f / 80.0f);
  const float c0 = x0 - mean, c1 = lane < 16 ? x1 - mean : 0.f;
  const float rms = sqrtf(wsum(c0 * c0 + c1 * c1) * (1.0f / 80.0f) + 1e-6f);
  bf16_t* o = frames + (long)(job.row0 + f) * 96;
  o[lane] = f32_to_bf16(asinhf(k * (c0 / rms)));
  if (lane < 32) o[64 + lane] = lane < 16 ? f32_to_bf16(asinhf(k * (c1 / rms))) : (bf16_t)0;
}

__global__ __launch_bounds__(256) void copy_segments_kernel(const StreamSeg* __restrict__ segs) {
  const StreamSeg sg = segs[blockIdx.x];
  const uint4* s = reinterpret_cast<const uint4*>(sg.src);
  uint4* d = reinterpret_cast<uint4*>(sg.dst);
  const long n = sg.bytes >> 4;
  for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.y * blockDim.x) d[i] = s[i];
}

// ---------------------------------------------------------------------------------------------
// sliding-window encoder attention: one wave per (row, head), at most 64 keys in the window
__global__ __launch_bounds__(256) void enc_window_attention_kernel(const bf16_t* __restrict__ qkv,
                                                                   const int* __restrict__ row_lo,
                                                                   const int* __restrict__ row_hi, int R, int D,
                                                                   int heads, int past, int future,
                                                                   bf16_t* __restrict__ out) {
  __shared__ float sq[4][128];
  __shared__ float sp[4][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + w;
  if (item >= R * heads) return;
  const int row = item / heads, head = item - row * heads, dh = D / heads;
  const float scale = rsqrtf((float)dh);
  const bf16_t* q = qkv + (long)row * 3 * D + head * dh;
  for (int d = lane; d < dh; d += 64) sq[w][d] = bf(q[d]) * scale;
  int jlo = row - past, jhi = row + future;
  jlo = jlo < row_lo[row] ? row_lo[row] : jlo;
  jhi = jhi > row_hi[row] - 1 ? row_hi[row] - 1 : jhi;
  const int nk = jhi - jlo + 1;
  __builtin_amdgcn_wave_barrier();
  float sc = -INFINITY;
  if (lane < nk) sc = dot_row(sq[w], qkv + (long)(jlo + lane) * 3 * D + D + head * dh, dh);
  const float mx = wmax(sc);
  const float e = lane < nk ? __expf(sc - mx) : 0.f;
  const float inv = 1.0f / wsum(e);
  sp[w][lane] = e * inv;
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < dh; d += 64) {
    float acc = 0.f;
    const bf16_t* v = qkv + (long)jlo * 3 * D + 2 * D + head * dh + d;
    for (int j = 0; j < nk; ++j) acc += sp[w][j] * bf(v[(long)j * 3 * D]);
    out[(long)row * D + head * dh + d] = f32_to_bf16(acc);
  }
}

// The same attention on the matrix pipe: one wave per (tile of 16 consecutive rows of ONE stream, head).  The tile's queries
// see keys [r0 - past, r0 + 15 + future] = NK = 32 KT key slots (36 -> 64 for the (16, 4) layers, 32 for (16, 0)):
//   S^T [key x query]   = K_tile (A: rows = keys, 16-byte loads along the head dim) x Q^T (B), DHP = 32 KS dims zero-padded;
//   softmax per query column in fp32 (exp2 domain), window / stream bounds as -inf, the four lane groups of a column meet
//   with the two row swaps; the S^T accumulator layout IS the B operand of the second product (slot 8 g + i of a 32-key
//   step = key 4 g + i of its first 16-key tile, slot 8 g + 4 + i of its second);
//   O^T [dim x query]   = V^T (A: the same slot order, gathered by 2-byte reads from the wave's row-major LDS copy of the
//   V rows, row stride DHP + 8: the four lane groups land 16 banks apart) x P^T.
// Tiles start at the stream's first row of the call (tile_row0, built by the host), so a query's key slots -- and with them
// the summation order inside the MFMAs -- depend on the stream alone: its rows do not change with what else is in the batch.
template <int KS, int KT>
__global__ __launch_bounds__(256) void enc_window_attention_mfma_kernel(const bf16_t* __restrict__ qkv,
                                                                        const int* __restrict__ tile_row0, int n_tiles,
                                                                        const int* __restrict__ row_lo,
                                                                        const int* __restrict__ row_hi, int R, int D, int heads,
                                                                        int dh, int past, int future, bf16_t* __restrict__ out) {
  constexpr int DHP = 32 * KS, NK = 32 * KT, VLD = DHP + 8, DT = DHP / 16, CH = DHP / 8;
  __shared__ __attribute__((aligned(16))) bf16_t vs[4][NK * VLD];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  // Workgroup b runs on XCD b % 8 (observed placement, used for speed only): the workgroups of one XCD take a CONTIGUOUS run of
  // (tile, head) items, so the key / value rows that neighbouring tiles share (a tile reads 32-64 rows for its 16 queries) are
  // fetched into ONE L2 instead of two to four (PMC: 94 / 169 MB read per launch for the 32- / 64-slot layers against 49 MB
  // of q | k | v, before this mapping)
  // (the grid is a multiple of 8)
  const int per = gridDim.x >> 3, vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int item = vb * 4 + w;
  if (item >= n_tiles * heads) return;
  const int tile = item / heads, head = item - tile * heads;
  const int r0 = tile_row0[tile];
  const int lo = row_lo[r0], hi = row_hi[r0];
  const int kb = r0 - past;
  const long ld = 3L * D;
  const bf16_t* base = qkv + head * dh;
  union Frag {
    uint4 u;
    bf16x8 v;
    uint32_t w[4];
    bf16_t h[8];
  };
  // the V rows of the tile's key range, row-major, dims beyond dh zero
  bf16_t* vw = vs[w];
#pragma unroll
  for (int c0 = 0; c0 < NK * CH; c0 += 64) {
    const int c = c0 + lane, kk = c / CH, d8 = (c - kk * CH) * 8;
    int row = kb + kk;
    row = row < 0 ? 0 : (row >= R ? R - 1 : row);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (d8 < dh) v = *reinterpret_cast<const uint4*>(base + row * ld + 2 * D + d8);
    *reinterpret_cast<uint4*>(vw + kk * VLD + d8) = v;
  }
  Frag qf[KS];
  {
    int row = r0 + n;
    row = row >= R ? R - 1 : row;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d = 32 * ks + 8 * g;
      qf[ks].u = d < dh ? *reinterpret_cast<const uint4*>(base + row * ld + d) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  f32x4 st[2 * KT];
#pragma unroll
  for (int kt = 0; kt < 2 * KT; ++kt) {
    int row = kb + 16 * kt + n;
    row = row < 0 ? 0 : (row >= R ? R - 1 : row);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d = 32 * ks + 8 * g;
      Frag kf;
      kf.u = d < dh ? *reinterpret_cast<const uint4*>(base + row * ld + D + d) : make_uint4(0u, 0u, 0u, 0u);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, qf[ks].v, acc, 0, 0, 0);
    }
    st[kt] = acc;
  }
  const int q = r0 + n;
  int jlo = q - past, jhi = q + future;
  jlo = jlo < lo ? lo : jlo;
  jhi = jhi > hi - 1 ? hi - 1 : jhi;
  const float c = rsqrtf((float)dh) * 1.4426950408889634f;
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 2 * KT; ++kt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = kb + 16 * kt + 4 * g + i;
      const float t = (j >= jlo && j <= jhi) ? st[kt][i] * c : -INFINITY;
      st[kt][i] = t;
      mx = fmaxf(mx, t);
    }
  {  // the column's four lane groups (lanes n, n + 16, n + 32, n + 48)
    const unsigned u = __float_as_uint(mx);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    mx = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const unsigned u2 = __float_as_uint(mx);
    auto r2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    mx = fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
  }
  if (mx == -INFINITY) mx = 0.f;   // (a column past the stream's last row: nothing is stored for it)
  float sum = 0.f;
  Frag pf[KT];
#pragma unroll
  for (int s = 0; s < KT; ++s) {
    float p0[4], p1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      p0[i] = exp2f(st[2 * s][i] - mx);
      p1[i] = exp2f(st[2 * s + 1][i] - mx);
      sum += p0[i] + p1[i];
    }
    pf[s].w[0] = pack_bf16x2(p0[0], p0[1]);
    pf[s].w[1] = pack_bf16x2(p0[2], p0[3]);
    pf[s].w[2] = pack_bf16x2(p1[0], p1[1]);
    pf[s].w[3] = pack_bf16x2(p1[2], p1[3]);
  }
  {
    const unsigned u = __float_as_uint(sum);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    sum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned u2 = __float_as_uint(sum);
    auto r2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    sum = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
  }
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  __builtin_amdgcn_wave_barrier();   // (the wave's own LDS writes are ordered before its reads; this pins the compiler)
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    if (16 * dt >= dh) break;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KT; ++s) {
      Frag vf;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vf.h[i] = vw[(32 * s + 4 * g + i) * VLD + 16 * dt + n];
        vf.h[4 + i] = vw[(32 * s + 16 + 4 * g + i) * VLD + 16 * dt + n];
      }
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[s].v, o, 0, 0, 0);
    }
    const int d = 16 * dt + 4 * g;
    if (q < hi && d < dh) {
      uint2 pk;
      pk.x = pack_bf16x2(o[0] * inv, o[1] * inv);
      pk.y = pack_bf16x2(o[2] * inv, o[3] * inv);
      *reinterpret_cast<uint2*>(out + (long)q * D + head * dh + d) = pk;
    }
  }
}

__global__ void adapter_in_kernel(const float* __restrict__ y32, const int* __restrict__ rows,
                                  const int* __restrict__ pos, int D, const float* __restrict__ pos_emb,
                                  bf16_t* __restrict__ out16, float* __restrict__ out32) {
  const int i = blockIdx.x;
  const float* y = y32 + (long)rows[i] * D;
  const float* p = pos_emb + (long)pos[i] * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float v = y[d] + p[d];
    out16[(long)i * D + d] = f32_to_bf16(v);
    out32[(long)i * D + d] = v;
  }
}

// K and V go in transposed ([slot][L][D][Mcap], keys contiguous): the cross-attention kernel streams 16 bytes
// (8 keys) per lane per head-dim row, for the scores and for the values alike.
__global__ void scatter_cross_kernel(const bf16_t* __restrict__ tmp, const int* __restrict__ slot,
                                     const int* __restrict__ idx, int L, int D, int Mcap, bf16_t* __restrict__ crossK,
                                     bf16_t* __restrict__ crossV) {
  const int i = blockIdx.x, l = blockIdx.y;
  const bf16_t* src = tmp + ((long)i * L + l) * 2 * D;
  const long base = ((long)slot[i] * L + l) * Mcap * D;
  const int key = idx[i];
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    crossK[base + (long)d * Mcap + key] = src[d];
    crossV[base + (long)d * Mcap + key] = src[D + d];
  }
}

__global__ void embed_kernel(const int* __restrict__ tokens, const float* __restrict__ embed, int D,
                             float* __restrict__ H) {
  const int r = blockIdx.x;
  const float4* e = reinterpret_cast<const float4*>(embed + (long)tokens[r] * D);
  float4* h = reinterpret_cast<float4*>(H + (long)r * D);
  for (int d = threadIdx.x; d < D / 4; d += blockDim.x) h[d] = e[d];
}

// ---------------------------------------------------------------------------------------------
__global__ void self_append_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ row_slot,
                                   const int* __restrict__ row_pos, int D, int layer, int L, int Scap,
                                   bf16_t* __restrict__ cacheK, bf16_t* __restrict__ cacheV) {
  const int r = blockIdx.x;
  const long dst = (((long)row_slot[r] * L + layer) * Scap + row_pos[r]) * D;
  const bf16_t* src = qkv + (long)r * 3 * D;
  for (int d = threadIdx.x * 8; d < D; d += blockDim.x * 8) {
    *reinterpret_cast<uint4*>(cacheK + dst + d) = *reinterpret_cast<const uint4*>(src + D + d);
    *reinterpret_cast<uint4*>(cacheV + dst + d) = *reinterpret_cast<const uint4*>(src + 2 * D + d);
  }
}

// one wave per (row, head): keys [0, pos], scores through LDS (at most SMAX keys).  Value pass: dh/4 lanes share one
// cached row (8-byte loads), 64 / (dh/4) keys in flight per step, the key groups' partial sums meet in LDS.
constexpr int SELF_SMAX = 512;
__global__ __launch_bounds__(256) void self_attention_kernel(const bf16_t* __restrict__ q, int q_stride,
                                                             const int* __restrict__ row_slot,
                                                             const int* __restrict__ row_pos, int M, int D, int heads,
                                                             int layer, int L, int Scap,
                                                             const bf16_t* __restrict__ cacheK,
                                                             const bf16_t* __restrict__ cacheV,
                                                             bf16_t* __restrict__ out, int fm) {
  __shared__ float sq[4][128];
  __shared__ float sp[4][SELF_SMAX];
  __shared__ float part[4][16][128];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + w;
  if (item >= M * heads) return;
  const int row = item / heads, head = item - row * heads, dh = D / heads;
  const float scale = rsqrtf((float)dh);
  const bf16_t* qr = q + (long)row * q_stride + head * dh;
  for (int d = lane; d < dh; d += 64) sq[w][d] = bf(qr[d]) * scale;
  const int nk = row_pos[row] + 1;
  const long base = (((long)row_slot[row] * L + layer) * Scap) * D + head * dh;
  __builtin_amdgcn_wave_barrier();
  float mx = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    const float sc = dot_row(sq[w], cacheK + base + (long)j * D, dh);
    sp[w][j] = sc;
    mx = fmaxf(mx, sc);
  }
  mx = wmax(mx);
  float sum = 0.f;
  for (int j = lane; j < nk; j += 64) {
    const float e = __expf(sp[w][j] - mx);
    sp[w][j] = e;
    sum += e;
  }
  const float inv = 1.0f / wsum(sum);
  __builtin_amdgcn_wave_barrier();
  const int tpk = dh >> 2;
  int G = 64 / tpk;
  G = G > 16 ? 16 : G;
  const int g = lane / tpk, c = lane - g * tpk;
  if (g < G) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bf16_t* v = cacheV + base + c * 4;
#pragma unroll 4
    for (int j = g; j < nk; j += G) {
      const uint2 r = *reinterpret_cast<const uint2*>(v + (long)j * D);
      const float p = sp[w][j];
      acc.x += p * __uint_as_float(r.x << 16);
      acc.y += p * __uint_as_float(r.x & 0xffff0000u);
      acc.z += p * __uint_as_float(r.y << 16);
      acc.w += p * __uint_as_float(r.y & 0xffff0000u);
    }
    *reinterpret_cast<float4*>(&part[w][g][c * 4]) = acc;
  }
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < dh; d += 64) {
    float t = 0.f;
    for (int k = 0; k < G; ++k) t += part[w][k][d];
    out[fm ? fm16(row, head * dh + d, D >> 5) : (long)row * D + head * dh + d] = f32_to_bf16(t * inv);
  }
}

// The same attention for the AR steps of decode_full (one new row per stream, at most SELF_AR_KEYS keys): one wave per
// (row, head), and everything it needs -- the query, the head's cached K rows (one key per lane, a second one from 64 keys
// on) and all V rows (16-byte chunks, staged through LDS) -- is requested in ONE batch of loads; self_attention_kernel
// above walks K and then V in several dependent round trips (10.3 us per launch at 64 streams, all of it latency).
// Arithmetic and summation order per output element are those of self_attention_kernel: scores as dot_row adds them, the
// value sum in G interleaved partial sums (G = that kernel's key groups) added in group order.
constexpr int SELF_AR_KEYS = 128;
// NT (probe for the next round, MSH_STREAM_SELF_NT=1, head_dim 80 only): the cached K / V rows with the non-temporal policy
typedef unsigned int su32x4_nt __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ld16(const bf16_t* p) {
  if constexpr (NT) {
    const su32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const su32x4_nt*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *reinterpret_cast<const uint4*>(p);
  }
}
template <int DH, bool NT = false>
__global__ __launch_bounds__(64) void self_attention_ar_kernel(const bf16_t* __restrict__ q, const int* __restrict__ row_slot,
                                                               const int* __restrict__ row_pos, int D, int heads, int layer,
                                                               int L, int Scap, const bf16_t* __restrict__ cacheK,
                                                               const bf16_t* __restrict__ cacheV, bf16_t* __restrict__ out,
                                                               int fm) {
  constexpr int C8 = DH / 8;                 // 16-byte chunks of a cached row
  constexpr int VROW = DH + 8;               // LDS row stride (bf16): 16-byte aligned, rows 4 banks apart
  constexpr int NV = (SELF_AR_KEYS * C8 + 63) / 64;
  constexpr int TPK = DH / 4;
  constexpr int G = 64 / TPK > 16 ? 16 : 64 / TPK;
  constexpr int ND = (DH + 63) / 64;
  __shared__ float sq[DH];
  __shared__ float sp[SELF_AR_KEYS];
  __shared__ __attribute__((aligned(16))) bf16_t sv[SELF_AR_KEYS * VROW];
  const int lane = threadIdx.x;
  const int row = blockIdx.x / heads, head = blockIdx.x - row * heads;
  int nk = row_pos[row] + 1;
  nk = nk > SELF_AR_KEYS ? SELF_AR_KEYS : nk;   // the host only picks this kernel when the pass cannot get there
  const long base = (((long)row_slot[row] * L + layer) * Scap) * D + head * DH;
  // ---- every load of the wave ----
  float qv[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int d = lane + 64 * i;
    qv[i] = d < DH ? bf(q[(long)row * D + head * DH + d]) : 0.f;
  }
  uint4 kr[2][C8];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int j = lane + 64 * t;
    const bf16_t* kp = cacheK + base + (long)(j < nk ? j : 0) * D;
    if (t == 0 || nk > 64) {   // (wave-uniform: a row of at most 64 keys has no second key per lane)
#pragma unroll
      for (int c = 0; c < C8; ++c) kr[t][c] = ld16<NT>(kp + c * 8);
    } else {
#pragma unroll
      for (int c = 0; c < C8; ++c) kr[t][c] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  uint4 vr[NV];
  const int nchunks = nk * C8;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    const int j = idx / C8, c = idx - j * C8;
    vr[i] = idx < nchunks ? ld16<NT>(cacheV + base + (long)j * D + c * 8) : make_uint4(0u, 0u, 0u, 0u);
  }
  __builtin_amdgcn_sched_barrier(0);
  const float scale = rsqrtf((float)DH);
#pragma unroll
  for (int i = 0; i < ND; ++i)
    if (lane + 64 * i < DH) sq[lane + 64 * i] = qv[i] * scale;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    const int j = idx / C8, c = idx - j * C8;
    if (idx < nchunks) *reinterpret_cast<uint4*>(&sv[j * VROW + c * 8]) = vr[i];
  }
  __syncthreads();
  float sc[2], mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t == 1 && nk <= 64) {   // wave-uniform
      sc[t] = -INFINITY;
      continue;
    }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < C8; ++c) {
      const uint4 r = kr[t][c];
      const float* qq = &sq[c * 8];
      acc += qq[0] * __uint_as_float(r.x << 16) + qq[1] * __uint_as_float(r.x & 0xffff0000u) +
             qq[2] * __uint_as_float(r.y << 16) + qq[3] * __uint_as_float(r.y & 0xffff0000u);
      acc += qq[4] * __uint_as_float(r.z << 16) + qq[5] * __uint_as_float(r.z & 0xffff0000u) +
             qq[6] * __uint_as_float(r.w << 16) + qq[7] * __uint_as_float(r.w & 0xffff0000u);
    }
    sc[t] = lane + 64 * t < nk ? acc : -INFINITY;
    mx = fmaxf(mx, sc[t]);
  }
  mx = wmax(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (lane + 64 * t < nk) {
      const float e = __expf(sc[t] - mx);
      sp[lane + 64 * t] = e;
      sum += e;
    }
  }
  const float inv = 1.0f / wsum(sum);
  __syncthreads();
  // P.V with lane = (key group g, 4-dim piece c), as self_attention_kernel has it: group g walks keys g, g + G, ... with 8-byte
  // LDS reads, the G partial sums of a dim are added in group order -- the same sums in the same order as the first version of
  // this kernel (one lane per dim walking ALL keys with 2-byte reads, a second pass for dims 64..79), a third of the iterations
  __shared__ __attribute__((aligned(16))) float pt[G][DH];
  {
    const int g = lane / TPK, c = lane - g * TPK;
    if (g < G) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int jk = g; jk < nk; jk += G) {
        const uint2 u = *reinterpret_cast<const uint2*>(&sv[jk * VROW + c * 4]);
        const float pw = sp[jk];
        acc.x += pw * __uint_as_float(u.x << 16);
        acc.y += pw * __uint_as_float(u.x & 0xffff0000u);
        acc.z += pw * __uint_as_float(u.y << 16);
        acc.w += pw * __uint_as_float(u.y & 0xffff0000u);
      }
      *reinterpret_cast<float4*>(&pt[g][c * 4]) = acc;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int d = lane + 64 * i;
    if (d < DH) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) t += pt[g][d];
      const int col = head * DH + d;
      out[fm ? fm16(row, col, D >> 5) : (long)row * D + col] = f32_to_bf16(t * inv);
    }
  }
}

// Cross-attention: one workgroup per (row, head), its 4 waves split the head dim (the layout of the offline
// dec_cross_attention_kernel).  K^T / V^T rows are keys-contiguous: lane l owns keys 8l..8l+7 of a 512-key chunk
// and loads 16 bytes per head-dim row; wave w covers rows d in [w*dh/4, (w+1)*dh/4).  All 2*dh/4 loads of a chunk
// are independent and issued back to back -- one memory round trip per chunk instead of one per head dim --
// partial q.k sums meet in LDS, every wave runs the same fp32 online softmax (exp2 domain) and accumulates its
// own output dims, reduced across lanes at the end.
constexpr int CROSS_MMAX = 4096;
typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float s_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float s_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// NT: the memory K / V rows of an auto-regressive step with the non-temporal policy -- the offline decoder's finding (DESIGN.md
// 3c: keep what is read once per step out of the memory-side cache, so that it keeps the weights) on the streaming decoder,
// whose weights (~240 MB at the medium dims) are what every step re-reads: config 5 931-934 -> 954-958 audio-s/s
// (profiles/r5x_*).  Default for head_dim 80; MSH_STREAM_XATTN_NT=0 switches it off.
// ABL (probe, 0 in the product): 1 = no K / V loads (the arithmetic on zeros), 2 = the loads alone (their values OR-ed into the
// output so that they stay)
template <int DH, bool NT = false, int ABL = 0>
__global__ __launch_bounds__(256, 2) void cross_attention_kernel(const bf16_t* __restrict__ q,
                                                                 const int* __restrict__ row_slot,
                                                                 const SlotDev* __restrict__ slots, int D, int heads,
                                                                 int layer, int L, int Mcap,
                                                                 const bf16_t* __restrict__ crossKT,
                                                                 const bf16_t* __restrict__ crossVT,
                                                                 bf16_t* __restrict__ out, int fm,
                                                                 const int* __restrict__ row_mem) {
  constexpr int DQ = DH / 4;
  __shared__ float sp[4][512];
  __shared__ float red[4][DQ][65];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int row = blockIdx.x, h = blockIdx.y;
  const int slot = row_slot[row];
  // row_mem (optional): the rows' memory lengths as the host knows them -- read beside row_slot instead of behind it
  // (row -> slot -> slots[slot].mem_len is two dependent loads in front of the first K / V request)
  const int nk = row_mem != nullptr ? row_mem[row] : slots[slot].mem_len;
  const long base = (((long)slot * L + layer) * D + h * DH + wave * DQ) * Mcap;
  const bf16_t* kt = crossKT + base;
  const bf16_t* vt = crossVT + base;
  const float c = rsqrtf((float)DH) * 1.4426950408889634f;
  float qd[DQ];
#pragma unroll
  for (int d = 0; d < DQ; ++d) qd[d] = bf(q[(long)row * D + h * DH + wave * DQ + d]);
  float opart[DQ];
#pragma unroll
  for (int d = 0; d < DQ; ++d) opart[d] = 0.f;
  float m_run = -INFINITY, l_part = 0.f;
#pragma unroll 1
  for (int k0 = 0; k0 < nk; k0 += 512) {
    const int key = k0 + lane * 8;
    const bool in = key < nk && ABL != 1;  // nk <= Mcap and Mcap % 8 == 0: an "in" lane's 8 keys are inside the row
    // K and V rows of the chunk are requested together: one memory round trip per chunk (2 x DQ 16-byte loads in flight)
    su32x4 kr[DQ], vr[DQ];
#pragma unroll
    for (int d = 0; d < DQ; ++d)
      if constexpr (NT) kr[d] = in ? __builtin_nontemporal_load(reinterpret_cast<const su32x4*>(kt + (long)d * Mcap + key)) : su32x4{0u, 0u, 0u, 0u};
      else kr[d] = in ? *reinterpret_cast<const su32x4*>(kt + (long)d * Mcap + key) : su32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int d = 0; d < DQ; ++d)
      if constexpr (NT) vr[d] = in ? __builtin_nontemporal_load(reinterpret_cast<const su32x4*>(vt + (long)d * Mcap + key)) : su32x4{0u, 0u, 0u, 0u};
      else vr[d] = in ? *reinterpret_cast<const su32x4*>(vt + (long)d * Mcap + key) : su32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABL == 2) {
      unsigned acc = 0;
#pragma unroll
      for (int d = 0; d < DQ; ++d) acc |= kr[d].x | kr[d].y | kr[d].z | kr[d].w | vr[d].x | vr[d].y | vr[d].z | vr[d].w;
      opart[0] += __uint_as_float(acc & 0x3f800000u);
      l_part = 1.f;
      continue;
    }
    float sc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sc[e] = 0.f;
#pragma unroll
    for (int d = 0; d < DQ; ++d) {
      const su32x4 u = kr[d];
      sc[0] += qd[d] * s_lo(u.x); sc[1] += qd[d] * s_hi(u.x);
      sc[2] += qd[d] * s_lo(u.y); sc[3] += qd[d] * s_hi(u.y);
      sc[4] += qd[d] * s_lo(u.z); sc[5] += qd[d] * s_hi(u.z);
      sc[6] += qd[d] * s_lo(u.w); sc[7] += qd[d] * s_hi(u.w);
    }
    if (k0 > 0) __syncthreads();
    *reinterpret_cast<float4*>(&sp[wave][lane * 8]) = make_float4(sc[0], sc[1], sc[2], sc[3]);
    *reinterpret_cast<float4*>(&sp[wave][lane * 8 + 4]) = make_float4(sc[4], sc[5], sc[6], sc[7]);
    __syncthreads();
    float mloc = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = lane * 8 + e;
      const float t = (sp[0][i] + sp[1][i]) + (sp[2][i] + sp[3][i]);
      sc[e] = (key + e < nk) ? t * c : -INFINITY;
      mloc = fmaxf(mloc, sc[e]);
    }
    mloc = wmax(mloc);
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = exp2f(sc[e] - m_new);
      psum += sc[e];
    }
    l_part = l_part * alpha + psum;
#pragma unroll
    for (int d = 0; d < DQ; ++d) {
      const su32x4 u = vr[d];
      const float part = sc[0] * s_lo(u.x) + sc[1] * s_hi(u.x) + sc[2] * s_lo(u.y) + sc[3] * s_hi(u.y) +
                         sc[4] * s_lo(u.z) + sc[5] * s_hi(u.z) + sc[6] * s_lo(u.w) + sc[7] * s_hi(u.w);
      opart[d] = opart[d] * alpha + part;
    }
  }
  const float l = wsum(l_part);
#pragma unroll
  for (int d = 0; d < DQ; ++d) red[wave][d][lane] = opart[d];
  __builtin_amdgcn_wave_barrier();
  if (lane < DQ) {
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) acc += red[wave][lane][i];
    const int col = h * DH + wave * DQ + lane;
    out[fm ? fm16(row, col, D >> 5) : (long)row * D + col] = f32_to_bf16(acc / l);
  }
}

// The same attention for RUNS of rows that belong to one stream (the wide verify pass: BOS + draft of a stream are up to
// 66 consecutive rows over ONE memory).  A workgroup takes up to RB consecutive rows of a stream and streams the K^T / V^T
// chunk once for all of them -- the per-row kernel above re-read a stream's 1.3 MB of K / V for every row (5.4 GB per layer
// at 64 streams x 66 rows: 396 us per launch, half of the wide pass).  Per row the arithmetic, its order and therefore the
// result are those of cross_attention_kernel.
struct RowRun {
  int row0, n;   // rows [row0, row0 + n) of the pass, all of one stream, 1 <= n <= RB
};
// NT (probe for the next round, MSH_STREAM_XRUNS_NT=1, head_dim 80 only): the memory rows of the wide pass non-temporal
template <int DH, int RB, bool NT = false>
__global__ __launch_bounds__(256, 2) void cross_attention_runs_kernel(const bf16_t* __restrict__ q,
                                                                      const int* __restrict__ row_slot,
                                                                      const RowRun* __restrict__ runs,
                                                                      const SlotDev* __restrict__ slots, int D, int heads,
                                                                      int layer, int L, int Mcap,
                                                                      const bf16_t* __restrict__ crossKT,
                                                                      const bf16_t* __restrict__ crossVT,
                                                                      bf16_t* __restrict__ out) {
  constexpr int DQ = DH / 4;
  __shared__ float sp[RB][4][512];
  __shared__ float red[4][DQ][65];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const RowRun run = runs[blockIdx.x];
  const int h = blockIdx.y;
  const int slot = row_slot[run.row0];
  const int nk = slots[slot].mem_len;
  const long base = (((long)slot * L + layer) * D + h * DH + wave * DQ) * Mcap;
  // buffer descriptors over this wave's dh/4 rows (row d at SCALAR offset d * Mcap * 2, the lane's keys at one shared 32-bit
  // vector offset): no per-row 64-bit addresses in registers; a read past the last row returns 0.  Lanes whose keys lie
  // beyond mem_len read stale (finite) cache contents; their scores are masked and their probabilities are exactly 0.
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(crossKT + base), 0, DQ * Mcap * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(crossVT + base), 0, DQ * Mcap * 2, 0x00020000);
  const float c = rsqrtf((float)DH) * 1.4426950408889634f;
  // the queries of the run sit in LDS (every lane of a wave uses the same value: broadcast reads), not in 80 registers
  __shared__ __attribute__((aligned(16))) float sq[4][DQ][RB];   // rows fastest: one 16-byte read per head-dim row
  float opart[RB][DQ], m_run[RB], l_part[RB];
  if (lane < RB * DQ) {
    const int r = lane / DQ, d = lane - r * DQ;
    const int row = run.row0 + (r < run.n ? r : run.n - 1);   // a short run repeats its last row; the copy is not stored
    sq[wave][d][r] = bf(q[(long)row * D + h * DH + wave * DQ + d]);
  }
  if constexpr (RB * DQ > 64) {
    const int i = lane + 64;
    if (i < RB * DQ) {
      const int r = i / DQ, d = i - r * DQ;
      const int row = run.row0 + (r < run.n ? r : run.n - 1);
      sq[wave][d][r] = bf(q[(long)row * D + h * DH + wave * DQ + d]);
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
#pragma unroll
    for (int d = 0; d < DQ; ++d) opart[r][d] = 0.f;
    m_run[r] = -INFINITY;
    l_part[r] = 0.f;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll 1
  for (int k0 = 0; k0 < nk; k0 += 512) {
    const int key = k0 + lane * 8;
    su32x4 kv[DQ];
#pragma unroll
    for (int d = 0; d < DQ; ++d) kv[d] = __builtin_amdgcn_raw_buffer_load_b128(rk, key * 2, d * Mcap * 2, NT ? 2 : 0);
    float sc[RB][8];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[r][e] = 0.f;
    // d outside, rows inside: one row of K is unpacked once and used by every row of the run (per row the sum still runs
    // over d in ascending order, as in the single-row kernel)
#pragma unroll
    for (int d = 0; d < DQ; ++d) {
      const su32x4 u = kv[d];
      const float k0f = s_lo(u.x), k1f = s_hi(u.x), k2f = s_lo(u.y), k3f = s_hi(u.y), k4f = s_lo(u.z), k5f = s_hi(u.z),
                  k6f = s_lo(u.w), k7f = s_hi(u.w);
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const float qv = sq[wave][d][r];
        sc[r][0] += qv * k0f; sc[r][1] += qv * k1f; sc[r][2] += qv * k2f; sc[r][3] += qv * k3f;
        sc[r][4] += qv * k4f; sc[r][5] += qv * k5f; sc[r][6] += qv * k6f; sc[r][7] += qv * k7f;
      }
      if ((d & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four query reads ahead at most, not all dh/4 of them
    }
    // the V rows are requested only now, when the K registers are dead: the scores are pinned in registers first (the
    // empty asm makes them "used" here), so neither the optimiser nor the scheduler can float the V loads above the FMAs
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(sc[r][e]));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < DQ; ++d) kv[d] = __builtin_amdgcn_raw_buffer_load_b128(rv, key * 2, d * Mcap * 2, NT ? 2 : 0);
    if (k0 > 0) __syncthreads();
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      *reinterpret_cast<float4*>(&sp[r][wave][lane * 8]) = make_float4(sc[r][0], sc[r][1], sc[r][2], sc[r][3]);
      *reinterpret_cast<float4*>(&sp[r][wave][lane * 8 + 4]) = make_float4(sc[r][4], sc[r][5], sc[r][6], sc[r][7]);
    }
    __syncthreads();
    float alpha[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      float mloc = -INFINITY;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = lane * 8 + e;
        const float t = (sp[r][0][i] + sp[r][1][i]) + (sp[r][2][i] + sp[r][3][i]);
        sc[r][e] = (key + e < nk) ? t * c : -INFINITY;
        mloc = fmaxf(mloc, sc[r][e]);
      }
      mloc = wmax(mloc);
      const float m_new = fmaxf(m_run[r], mloc);
      alpha[r] = exp2f(m_run[r] - m_new);
      m_run[r] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc[r][e] = exp2f(sc[r][e] - m_new);
        psum += sc[r][e];
      }
      l_part[r] = l_part[r] * alpha[r] + psum;
      __builtin_amdgcn_sched_barrier(0);   // one row's 32 LDS reads at a time (hoisting all rows' reads spilled 200 registers)
    }
#pragma unroll
    for (int d = 0; d < DQ; ++d) {
      const su32x4 u = kv[d];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const float part = sc[r][0] * s_lo(u.x) + sc[r][1] * s_hi(u.x) + sc[r][2] * s_lo(u.y) + sc[r][3] * s_hi(u.y) +
                           sc[r][4] * s_lo(u.z) + sc[r][5] * s_hi(u.z) + sc[r][6] * s_lo(u.w) + sc[r][7] * s_hi(u.w);
        opart[r][d] = opart[r][d] * alpha[r] + part;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const float l = wsum(l_part[r]);
    __builtin_amdgcn_wave_barrier();   // the previous row's reads of this wave's slab are done
#pragma unroll
    for (int d = 0; d < DQ; ++d) red[wave][d][lane] = opart[r][d];
    __builtin_amdgcn_wave_barrier();
    if (lane < DQ && r < run.n) {
      float acc = 0.f;
#pragma unroll 8
      for (int i = 0; i < 64; ++i) acc += red[wave][lane][i];
      out[(long)(run.row0 + r) * D + h * DH + wave * DQ + lane] = f32_to_bf16(acc / l);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Cross-attention of a LONG run of rows of one stream (the verify pass: BOS + draft = up to 66 rows over one memory) on the
// matrix pipe: one 8-wave workgroup per (run of <= 128 rows, head), wave w owns query tile w (16 rows); the stream's K^T / V^T
// pass through LDS ONCE per head, in chunks of 64 keys (the runs kernel above re-reads them for every four rows).
//   staging: every thread fetches 16-byte pieces (8 keys of one head-dim row) of the NEXT chunk into registers while the
//     current one is computed; keys >= nk and padding dims arrive as zeros.  V^T goes to LDS as it is ([dim][72]); K^T is
//     written in MFMA-fragment order (a piece = 8 two-byte writes) so that every wave reads an A fragment with one
//     ds_read_b128 -- the transposition is done once per workgroup, not once per wave;
//   S^T [key x query] = K (A) x Q^T (B: 16-byte loads from the row-major q rows, kept in registers for all chunks);
//   fp32 online softmax per query column in the exp2 domain; the S^T accumulator layout IS the B operand of
//   O^T [dim x query] += V^T (A: two ds_read_b64 per fragment, the slot order of the accumulators) x P^T.
// P is rounded to bf16 (as in every MFMA attention of this library); the one-row kernels keep it in fp32 -- the wide pass
// and the auto-regressive steps already round differently elsewhere (tiled vs split-K GEMMs).
template <int KS, int DT>
__global__ __launch_bounds__(512) void cross_attention_wide_kernel(const bf16_t* __restrict__ q, const int* __restrict__ row_slot,
                                                                   const RowRun* __restrict__ runs,
                                                                   const SlotDev* __restrict__ slots, int D, int heads, int dh,
                                                                   int layer, int L, int Mcap,
                                                                   const bf16_t* __restrict__ crossKT,
                                                                   const bf16_t* __restrict__ crossVT, bf16_t* __restrict__ out) {
  constexpr int KC = 64, VLD = 72, DHP = 32 * KS, VR = 16 * DT;
  constexpr int KPIECES = DHP * (KC / 8), VPIECES = VR * (KC / 8);   // 16-byte pieces per chunk
  constexpr int KPT = (KPIECES + 511) / 512, VPT = (VPIECES + 511) / 512;
  __shared__ __attribute__((aligned(16))) bf16_t kfm[(KC / 16) * KS * 512];   // [key tile][k-step][lane][8]
  __shared__ __attribute__((aligned(16))) bf16_t vimg[VR * VLD];
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, g = lane >> 4;
  const RowRun run = runs[blockIdx.x];
  const int h = blockIdx.y;
  const int slot = row_slot[run.row0];
  const int nk = slots[slot].mem_len;
  const long base = (((long)slot * L + layer) * D + h * dh) * Mcap;
  const bf16_t* kt_g = crossKT + base;
  const bf16_t* vt_g = crossVT + base;
  union Frag {
    uint4 u;
    bf16x8 v;
    uint32_t w[4];
    bf16_t h[8];
    uint2 d[2];
  };
  // the wave's query tile as B fragments
  const bool active = 16 * w < run.n;
  const int qrow = run.row0 + 16 * w + n;
  const bool qok = 16 * w + n < run.n;
  Frag qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d = 32 * ks + 8 * g;
    qf[ks].u = (qok && d < dh) ? *reinterpret_cast<const uint4*>(q + (long)qrow * D + h * dh + d) : make_uint4(0u, 0u, 0u, 0u);
  }
  uint4 kreg[KPT], vreg[VPT];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int p = tid + 512 * i, d = p >> 3, key = k0 + 8 * (p & 7);
      kreg[i] = (p < KPIECES && d < dh && key < nk) ? *reinterpret_cast<const uint4*>(kt_g + (long)d * Mcap + key) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int p = tid + 512 * i, d = p >> 3, key = k0 + 8 * (p & 7);
      vreg[i] = (p < VPIECES && d < dh && key < nk) ? *reinterpret_cast<const uint4*>(vt_g + (long)d * Mcap + key) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  // nk % 8 == 0 is not guaranteed: a piece that straddles nk brings keys >= nk with it -- they are masked in the scores
  // (-inf -> p = 0) and their V values, finite numbers of an earlier, longer memory or zeros of the allocation, meet p = 0
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int p = tid + 512 * i;
      if (p < KPIECES) {
        const int d = p >> 3, kk0 = 8 * (p & 7);
        Frag f;
        f.u = kreg[i];
        const int ks = d >> 5, gg = (d & 31) >> 3, e = d & 7;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kk = kk0 + j;
          kfm[(((kk >> 4) * KS + ks) * 64 + (kk & 15) + 16 * gg) * 8 + e] = f.h[j];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int p = tid + 512 * i;
      if (p < VPIECES) *reinterpret_cast<uint4*>(vimg + (p >> 3) * VLD + 8 * (p & 7)) = vreg[i];
    }
  };
  const float c = rsqrtf((float)dh) * 1.4426950408889634f;
  float m_run = -INFINITY, l_part = 0.f;
  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch(0);
#pragma unroll 1
  for (int k0 = 0; k0 < nk; k0 += KC) {
    if (k0 > 0) __syncthreads();   // every wave is done with the previous chunk's images
    stash();
    __syncthreads();
    if (k0 + KC < nk) fetch(k0 + KC);
    if (active) {
      f32x4 st[KC / 16];
#pragma unroll
      for (int kt = 0; kt < KC / 16; ++kt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Frag kf;
          kf.u = *reinterpret_cast<const uint4*>(kfm + ((kt * KS + ks) * 64 + lane) * 8);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, qf[ks].v, acc, 0, 0, 0);
        }
        st[kt] = acc;
      }
      float cm = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KC / 16; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key = k0 + 16 * kt + 4 * g + i;
          const float t = key < nk ? st[kt][i] * c : -INFINITY;
          st[kt][i] = t;
          cm = fmaxf(cm, t);
        }
      {
        const unsigned u = __float_as_uint(cm);
        auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        cm = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        const unsigned u2 = __float_as_uint(cm);
        auto r2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
        cm = fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
      }
      const float m_new = fmaxf(m_run, cm);   // (finite: key k0 of the chunk is < nk)
      const float alpha = exp2f(m_run - m_new);
      m_run = m_new;
      Frag pf[KC / 32];
      float psum = 0.f;
#pragma unroll
      for (int s = 0; s < KC / 32; ++s) {
        float p0[4], p1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          p0[i] = exp2f(st[2 * s][i] - m_new);
          p1[i] = exp2f(st[2 * s + 1][i] - m_new);
          psum += p0[i] + p1[i];
        }
        pf[s].w[0] = pack_bf16x2(p0[0], p0[1]);
        pf[s].w[1] = pack_bf16x2(p0[2], p0[3]);
        pf[s].w[2] = pack_bf16x2(p1[0], p1[1]);
        pf[s].w[3] = pack_bf16x2(p1[2], p1[3]);
      }
      l_part = l_part * alpha + psum;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f32x4 acc = o[dt];
        acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
#pragma unroll
        for (int s = 0; s < KC / 32; ++s) {
          Frag vf;
          const bf16_t* vrow = vimg + (16 * dt + n) * VLD + 32 * s + 4 * g;
          vf.d[0] = *reinterpret_cast<const uint2*>(vrow);
          vf.d[1] = *reinterpret_cast<const uint2*>(vrow + 16);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[s].v, acc, 0, 0, 0);
        }
        o[dt] = acc;
      }
    }
  }
  if (!active) return;
  float l = l_part;
  {
    const unsigned u = __float_as_uint(l);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    l = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned u2 = __float_as_uint(l);
    auto r2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    l = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  if (qok) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = 16 * dt + 4 * g;
      if (d < dh) {
        uint2 pk;
        pk.x = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
        pk.y = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
        *reinterpret_cast<uint2*>(out + (long)qrow * D + h * dh + d) = pk;
      }
    }
  }
}

// Any other head_dim (multiple of 4, <= 128): plain two-pass kernel over the same transposed layouts.
__global__ __launch_bounds__(256) void cross_attention_generic_kernel(const bf16_t* __restrict__ q,
                                                                      const int* __restrict__ row_slot,
                                                                      const SlotDev* __restrict__ slots, int D,
                                                                      int heads, int layer, int L, int Mcap,
                                                                      const bf16_t* __restrict__ crossKT,
                                                                      const bf16_t* __restrict__ crossVT,
                                                                      bf16_t* __restrict__ out, int fm) {
  __shared__ float sq[128];
  __shared__ float sp[CROSS_MMAX];
  __shared__ float red[8];
  const int row = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int dh = D / heads;
  const float scale = rsqrtf((float)dh);
  const int slot = row_slot[row];
  const int nk = slots[slot].mem_len;
  for (int d = tid; d < dh; d += 256) sq[d] = bf(q[(long)row * D + head * dh + d]) * scale;
  __syncthreads();
  const long base = (((long)slot * L + layer) * D + head * dh) * Mcap;
  float mx = -INFINITY;
  for (int j = tid; j < nk; j += 256) {
    float a = 0.f;
    for (int d = 0; d < dh; ++d) a += sq[d] * bf(crossKT[base + (long)d * Mcap + j]);
    sp[j] = a;
    mx = fmaxf(mx, a);
  }
  mx = wmax(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int j = tid; j < nk; j += 256) {
    const float e = __expf(sp[j] - mx);
    sp[j] = e;
    sum += e;
  }
  sum = wsum(sum);
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  // one wave per head dim (round-robin), lanes over keys
  for (int d = w; d < dh; d += 4) {
    float a = 0.f;
    for (int j = lane; j < nk; j += 64) a += sp[j] * bf(crossVT[base + (long)d * Mcap + j]);
    a = wsum(a);
    if (lane == 0) out[fm ? fm16(row, head * dh + d, D >> 5) : (long)row * D + head * dh + d] = f32_to_bf16(a * inv);
  }
}

// Cross-attention PROBABILITIES of a decoder pass, for word timestamps on the streaming architectures (the
// `cross_attentions.{l}` outputs of the reference's decoder_kv_with_attention graph,
// core/moonshine-streaming-model.cpp:946-1066): plain two-pass softmax per (row, head) over the stream's memory frames,
// written to out[row][layer][head][0..mem_len) with row stride L * heads * Ecap.  Only launched while a capture is on.
__global__ __launch_bounds__(256) void cross_probs_kernel(const bf16_t* __restrict__ q, const int* __restrict__ row_slot,
                                                          const SlotDev* __restrict__ slots, int D, int heads, int layer,
                                                          int L, int Mcap, const bf16_t* __restrict__ crossKT, int Ecap,
                                                          float* __restrict__ out) {
  __shared__ float sq[128];
  __shared__ float sp[CROSS_MMAX];
  __shared__ float red[8];
  const int row = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int dh = D / heads;
  const float scale = rsqrtf((float)dh);
  const int slot = row_slot[row];
  const int nk = slots[slot].mem_len;
  for (int d = tid; d < dh; d += 256) sq[d] = bf(q[(long)row * D + head * dh + d]) * scale;
  __syncthreads();
  const long base = (((long)slot * L + layer) * D + head * dh) * Mcap;
  float mx = -INFINITY;
  for (int j = tid; j < nk; j += 256) {
    float a = 0.f;
    for (int d = 0; d < dh; ++d) a += sq[d] * bf(crossKT[base + (long)d * Mcap + j]);
    sp[j] = a;
    mx = fmaxf(mx, a);
  }
  mx = wmax(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int j = tid; j < nk; j += 256) {
    const float e = __expf(sp[j] - mx);
    sp[j] = e;
    sum += e;
  }
  sum = wsum(sum);
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  float* o = out + (((long)row * L + layer) * heads + head) * Ecap;
  for (int j = tid; j < nk; j += 256) o[j] = sp[j] * inv;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int V, int* __restrict__ pred) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float4* x = reinterpret_cast<const float4*>(logits + (long)r * V);
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int c = tid; c < V / 4; c += 256) {
    const float4 v = x[c];
    const float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (vals[k] > best) {  // strictly greater: within a thread indices only grow, so the first max wins
        best = vals[k];
        idx = c * 4 + k;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) {
      best = ov;
      idx = oi;
    }
  }
  if ((tid & 63) == 0) {
    bv[tid >> 6] = best;
    bi[tid >> 6] = idx;
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < 4; ++k)
      if (bv[k] > best || (bv[k] == best && bi[k] < idx)) {
        best = bv[k];
        idx = bi[k];
      }
    pred[r] = idx == 0x7fffffff ? 0 : idx;  // all-NaN / -inf row: the reference's scan stays at index 0
  }
}

__global__ void verify_kernel(const DecJob* __restrict__ jobs, const int* __restrict__ pred,
                              const int* __restrict__ draft, SlotDev* __restrict__ slots, int* __restrict__ result,
                              int result_stride, int eos, const float* __restrict__ embed, int D, float* __restrict__ H,
                              int* __restrict__ step_pos, int* __restrict__ n_active, int fm) {
  __shared__ int s_cur, s_fin;
  const DecJob job = jobs[blockIdx.x];
  if (threadIdx.x == 0) {
    SlotDev sd = slots[job.slot];
    int* res = result + (long)job.slot * result_stride;
    int d = 0;
    for (int i = 0; i < job.draft_len; ++i) {
      if (pred[job.row0 + i] == draft[job.draft_off + i])
        d = i + 1;
      else
        break;
    }
    for (int i = 0; i < d; ++i) res[i] = draft[job.draft_off + i];
    sd.accepted = d;
    sd.count = d;
    sd.cache_len = d + 1;  // BOS + the accepted prefix; later cache rows are stale and get overwritten
    sd.max_tokens = job.max_tokens;
    sd.current = pred[job.row0 + d];
    sd.finished = (sd.current == eos || sd.count >= sd.max_tokens) ? 1 : 0;
    if (!sd.finished) {
      res[sd.count++] = sd.current;
      atomicAdd(n_active, 1);
    }
    step_pos[blockIdx.x] = sd.cache_len;
    slots[job.slot] = sd;
    s_cur = sd.current;
    s_fin = sd.finished;
  }
  __syncthreads();
  if (!s_fin) {
    const float* e = embed + (long)s_cur * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) H[fm ? fm32(blockIdx.x, d, D >> 5) : (long)blockIdx.x * D + d] = e[d];
  }
}

// The same step when the LM head ran as the tiled GEMM with the per-tile argmax epilogue (gemm_argmax_partials): the row's
// token is the first maximum over its ntn (max, lowest index) pairs -- tiles are in ascending column order, ties keep the
// lowest index, a row without any value above -inf (all NaN) yields token 0 like argmax_kernel.  No logits, no argmax launch.
__global__ __launch_bounds__(128) void advance_partials_kernel(const DecJob* __restrict__ jobs,
                                                               const float* __restrict__ pval, const int* __restrict__ pidx,
                                                               int ntn, SlotDev* __restrict__ slots, int* __restrict__ result,
                                                               int result_stride, int eos, const float* __restrict__ embed,
                                                               int D, float* __restrict__ H, int* __restrict__ step_pos,
                                                               int* __restrict__ n_active, int fm) {
  __shared__ int s_cur, s_fin;
  __shared__ float bv[2];
  __shared__ int bi[2];
  const DecJob job = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int i = tid; i < ntn; i += 128) {
    const float v = pval[(long)blockIdx.x * ntn + i];
    const int ix = pidx[(long)blockIdx.x * ntn + i];
    if (v > best || (v == best && ix < besti)) {
      best = v;
      besti = ix;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) {
      best = ov;
      besti = oi;
    }
  }
  if ((tid & 63) == 0) {
    bv[tid >> 6] = best;
    bi[tid >> 6] = besti;
  }
  __syncthreads();
  if (tid == 0) {
    if (bv[1] > best || (bv[1] == best && bi[1] < besti)) besti = bi[1];
    const int token = besti == 0x7fffffff ? 0 : besti;
    SlotDev sd = slots[job.slot];
    s_fin = 1;
    if (!sd.finished) {
      sd.cache_len += 1;  // the token just fed is now cached
      sd.current = token;
      if (sd.current == eos || sd.count >= sd.max_tokens) {
        sd.finished = 1;
        atomicSub(n_active, 1);
      } else {
        result[(long)job.slot * result_stride + sd.count++] = sd.current;
        step_pos[blockIdx.x] = sd.cache_len;
        s_fin = 0;
      }
      slots[job.slot] = sd;
      s_cur = sd.current;
    }
  }
  __syncthreads();
  if (!s_fin) {
    const float* e = embed + (long)s_cur * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) H[fm ? fm32(blockIdx.x, d, D >> 5) : (long)blockIdx.x * D + d] = e[d];
  }
}

__global__ void advance_kernel(const DecJob* __restrict__ jobs, const int* __restrict__ pred,
                               SlotDev* __restrict__ slots, int* __restrict__ result, int result_stride, int eos,
                               const float* __restrict__ embed, int D, float* __restrict__ H,
                               int* __restrict__ step_pos, int* __restrict__ n_active, int fm) {
  __shared__ int s_cur, s_fin;
  const DecJob job = jobs[blockIdx.x];
  if (threadIdx.x == 0) {
    SlotDev sd = slots[job.slot];
    s_fin = 1;
    if (!sd.finished) {
      sd.cache_len += 1;  // the token just fed is now cached
      sd.current = pred[blockIdx.x];
      if (sd.current == eos || sd.count >= sd.max_tokens) {
        sd.finished = 1;
        atomicSub(n_active, 1);
      } else {
        result[(long)job.slot * result_stride + sd.count++] = sd.current;
        step_pos[blockIdx.x] = sd.cache_len;
        s_fin = 0;
      }
      slots[job.slot] = sd;
      s_cur = sd.current;
    }
  }
  __syncthreads();
  if (!s_fin) {
    const float* e = embed + (long)s_cur * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) H[fm ? fm32(blockIdx.x, d, D >> 5) : (long)blockIdx.x * D + d] = e[d];
  }
}

// ---------------------------------------------------------------------------------------------
// Contextual biasing: one 64-lane wave per logits row.
constexpr int BIAS_MAX_ACTIVE = 64;
__device__ __forceinline__ int trie_child(const BiasTrie& T, int node, int token) {
  int lo = T.child_off[node], hi = T.child_off[node + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int t = T.child_tok[mid];
    if (t == token) return T.child_node[mid];
    if (t < token)
      lo = mid + 1;
    else
      hi = mid;
  }
  return -1;
}
__global__ __launch_bounds__(64) void bias_rows_kernel(BiasTrie T, const int2* __restrict__ prefix,
                                                       const int* __restrict__ tokens, const DecJob* __restrict__ jobs,
                                                       const SlotDev* __restrict__ slots, const int* __restrict__ result,
                                                       int result_stride, float* __restrict__ logits, int V) {
  __shared__ int active[2][BIAS_MAX_ACTIVE];
  __shared__ int n_active_s;
  const int row = blockIdx.x, lane = threadIdx.x;
  const int* pre;
  int n_pre;
  if (jobs != nullptr) {
    const int slot = jobs[row].slot;
    const SlotDev sd = slots[slot];
    if (sd.finished) return;
    pre = result + (long)slot * result_stride;
    n_pre = sd.count;
  } else {
    pre = tokens + prefix[row].x;
    n_pre = prefix[row].y;
  }
  if (lane == 0) {  // the walk of ContextBiaser::advance: root always active, plus every continued path
    int cur = 0, n = 1;
    active[0][0] = 0;
    for (int i = 0; i < n_pre; ++i) {
      const int tok = pre[i];
      int m = 1;
      active[cur ^ 1][0] = 0;
      for (int a = 0; a < n; ++a) {
        const int c = trie_child(T, active[cur][a], tok);
        if (c >= 0 && m < BIAS_MAX_ACTIVE) active[cur ^ 1][m++] = c;
      }
      cur ^= 1;
      n = m;
    }
    if (cur == 1)
      for (int a = 0; a < n; ++a) active[0][a] = active[1][a];
    n_active_s = n;
  }
  __syncthreads();
  const int n = n_active_s;
  float* lg = logits + (long)row * V;
  // every (active node, child) pair proposes a token; the largest bonus among the active nodes that propose it is
  // added once -- by the first such node in the active list
  for (int a = 0; a < n; ++a) {
    const int node = active[0][a];
    const int beg = T.child_off[node], end = T.child_off[node + 1];
    const float mine = T.depth_bonus[T.depth[node] + 1];
    for (int c = beg + lane; c < end; c += 64) {
      const int tok = T.child_tok[c];
      if (tok < 0 || tok >= V) continue;
      float best = mine;
      bool owner = true;
      for (int b = 0; b < n; ++b) {
        if (b == a) continue;
        if (trie_child(T, active[0][b], tok) >= 0) {
          if (b < a) owner = false;
          best = fmaxf(best, T.depth_bonus[T.depth[active[0][b]] + 1]);
        }
      }
      if (owner) lg[tok] += best;
    }
  }
}

__global__ void slot_update_kernel(const int4* __restrict__ upd, int n, SlotDev* __restrict__ slots) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int4 u = upd[j];  // (slot, mem_len or -1, cache_len or -1, zero-the-decode-fields flag)
  SlotDev sd = slots[u.x];
  if (u.y >= 0) sd.mem_len = u.y;
  if (u.z >= 0) sd.cache_len = u.z;
  if (u.w) sd.count = sd.finished = sd.current = sd.accepted = sd.max_tokens = 0;
  slots[u.x] = sd;
}

__global__ void bump_cache_kernel(const DecJob* __restrict__ jobs, int n, SlotDev* __restrict__ slots) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) slots[jobs[j].slot].cache_len += jobs[j].n_rows;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
void stream_frames(const float* audio, const FrameJob* jobs, int n_jobs, int max_frames, float k, bf16_t* frames,
                   hipStream_t s) {
  if (n_jobs <= 0 || max_frames <= 0) return;
  MSH_LAUNCH(frames_kernel, dim3((max_frames + 3) / 4, n_jobs), dim3(256), 0, s, audio, jobs, k, frames);
}
void copy_segments(const StreamSeg* segs, int n, hipStream_t s) {
  if (n <= 0) return;
  MSH_LAUNCH(copy_segments_kernel, dim3(n, 4), dim3(256), 0, s, segs);
}
void stream_enc_attention(const bf16_t* qkv, const int* row_lo, const int* row_hi, int R, int D, int heads, int past,
                          int future, bf16_t* out, hipStream_t s, const int* tile_row0, int n_tiles) {
  const int dh = D / heads;
  if (past + future + 1 > 64 || dh > 128 || (dh & 3) != 0)
    throw std::runtime_error("stream_enc_attention: window wider than 64 keys or unsupported head_dim");
  if (R <= 0) return;
  static const bool no_mfma = [] {   // A/B switch: MSH_STREAM_WINDOW_VALU=1 keeps the one-wave-per-(row, head) VALU kernel
    const char* e = dev_getenv("MSH_STREAM_WINDOW_VALU");
    return e != nullptr && e[0] == '1';
  }();
  const int ks = (dh + 31) / 32, kt = (past + future + 16 + 31) / 32;
  if (!no_mfma && tile_row0 != nullptr && n_tiles > 0 && (dh & 7) == 0 && (D & 7) == 0 && ks <= 4 && kt <= 3) {
    const dim3 grid(((n_tiles * heads + 3) / 4 + 7) / 8 * 8);   // (XCD-aware item order: see the kernel)
#define MSH_WATT(KSV, KTV)                                                                                                       \
  if (ks == KSV && kt == KTV) {                                                                                                  \
    MSH_LAUNCH((enc_window_attention_mfma_kernel<KSV, KTV>), grid, dim3(256), 0, s, qkv, tile_row0, n_tiles, row_lo, row_hi, R, D, \
               heads, dh, past, future, out);                                                                                    \
    return;                                                                                                                      \
  }
    MSH_WATT(1, 1) MSH_WATT(1, 2) MSH_WATT(1, 3) MSH_WATT(2, 1) MSH_WATT(2, 2) MSH_WATT(2, 3)
    MSH_WATT(3, 1) MSH_WATT(3, 2) MSH_WATT(3, 3) MSH_WATT(4, 1) MSH_WATT(4, 2) MSH_WATT(4, 3)
#undef MSH_WATT
  }
  MSH_LAUNCH(enc_window_attention_kernel, dim3((R * heads + 3) / 4), dim3(256), 0, s, qkv, row_lo, row_hi, R,
                     D, heads, past, future, out);
}
void stream_adapter_in(const float* y32, const int* rows, const int* pos, int n, int D, const float* pos_emb,
                       bf16_t* out16, float* out32, hipStream_t s) {
  if (n <= 0) return;
  MSH_LAUNCH(adapter_in_kernel, dim3(n), dim3(256), 0, s, y32, rows, pos, D, pos_emb, out16, out32);
}
void stream_scatter_cross(const bf16_t* tmp, const int* slot, const int* idx, int n, int L, int D, int Mcap,
                          bf16_t* crossK, bf16_t* crossV, hipStream_t s) {
  if (n <= 0) return;
  if ((D & 7) != 0) throw std::runtime_error("stream_scatter_cross: width must be a multiple of 8");
  MSH_LAUNCH(scatter_cross_kernel, dim3(n, L), dim3(128), 0, s, tmp, slot, idx, L, D, Mcap, crossK, crossV);
}
void stream_embed(const int* tokens, int M, const float* embed, int D, float* H, hipStream_t s) {
  if (M <= 0) return;
  MSH_LAUNCH(embed_kernel, dim3(M), dim3(128), 0, s, tokens, embed, D, H);
}
void stream_self_attention(const bf16_t* qkv, const int* row_slot, const int* row_pos, int M, int D, int heads, int layer,
                           int L, int Scap, bf16_t* cacheK, bf16_t* cacheV, bf16_t* out, hipStream_t s) {
  const int dh = D / heads;
  if (Scap > SELF_SMAX || dh > 128 || (dh & 3) != 0 || (D & 7) != 0)
    throw std::runtime_error("stream_self_attention: unsupported cache length or head_dim");
  if (M <= 0) return;
  MSH_LAUNCH(self_append_kernel, dim3(M), dim3(128), 0, s, qkv, row_slot, row_pos, D, layer, L, Scap, cacheK,
                     cacheV);
  MSH_LAUNCH(self_attention_kernel, dim3((M * heads + 3) / 4), dim3(256), 0, s, qkv, 3 * D, row_slot, row_pos, M,
                     D, heads, layer, L, Scap, cacheK, cacheV, out, 0);
}
void stream_self_attention_cached(const bf16_t* q, const int* row_slot, const int* row_pos, int M, int D, int heads,
                                  int layer, int L, int Scap, const bf16_t* cacheK, const bf16_t* cacheV, bf16_t* out,
                                  hipStream_t s, bool fm, int max_keys) {
  const int dh = D / heads;
  if (Scap > SELF_SMAX || dh > 128 || (dh & 3) != 0 || (D & 7) != 0)
    throw std::runtime_error("stream_self_attention: unsupported cache length or head_dim");
  if (M <= 0) return;
  if (fm && (D & 31) != 0) throw std::runtime_error("stream_self_attention: FM output needs D % 32 == 0");
  static const bool no_ar_kernel = [] {
    const char* e = dev_getenv("MSH_STREAM_SELF_AR");
    return e != nullptr && e[0] == '0';
  }();
  static const bool self_nt = [] {   // probe for the next round (default off: not measured yet)
    const char* e = dev_getenv("MSH_STREAM_SELF_NT");
    return e != nullptr && e[0] == '1';
  }();
  if (max_keys > 0 && max_keys <= SELF_AR_KEYS && !no_ar_kernel && dh == 80 && self_nt) {
    MSH_LAUNCH((self_attention_ar_kernel<80, true>), dim3(M * heads), dim3(64), 0, s, q, row_slot, row_pos, D, heads, layer, L, Scap,
               cacheK, cacheV, out, fm ? 1 : 0);
    return;
  }
  if (max_keys > 0 && max_keys <= SELF_AR_KEYS && !no_ar_kernel && (dh == 24 || dh == 40 || dh == 80)) {
#define MSH_SELF_AR(DHV)                                                                                              \
  case DHV:                                                                                                           \
    MSH_LAUNCH(self_attention_ar_kernel<DHV>, dim3(M * heads), dim3(64), 0, s, q, row_slot, row_pos, D, heads, layer, \
               L, Scap, cacheK, cacheV, out, fm ? 1 : 0);                                                            \
    return
    switch (dh) {
      MSH_SELF_AR(24);
      MSH_SELF_AR(40);
      MSH_SELF_AR(80);
    }
#undef MSH_SELF_AR
  }
  MSH_LAUNCH(self_attention_kernel, dim3((M * heads + 3) / 4), dim3(256), 0, s, q, D, row_slot, row_pos, M, D,
                     heads, layer, L, Scap, cacheK, cacheV, out, fm ? 1 : 0);
}
void stream_cross_attention(const bf16_t* q, const int* row_slot, const SlotDev* slots, int M, int D, int heads,
                            int layer, int L, int Mcap, const bf16_t* crossK, const bf16_t* crossV, bf16_t* out,
                            hipStream_t s, bool fm, const int* row_mem) {
  const int dh = D / heads;
  if (Mcap > CROSS_MMAX || dh > 128 || (dh & 3) != 0 || (Mcap & 7) != 0)
    throw std::runtime_error("stream_cross_attention: unsupported memory length or head_dim");
  if (fm && (D & 31) != 0) throw std::runtime_error("stream_cross_attention: FM output needs D % 32 == 0");
  if (M <= 0) return;
  const int fmi = fm ? 1 : 0;
  static const bool nt = [] {
    const char* e = dev_getenv("MSH_STREAM_XATTN_NT");
    return !(e != nullptr && e[0] == '0');
  }();
  if (nt && dh == 80) {
    static const int abl = [] {
      const char* e = dev_getenv("MSH_STREAM_XATTN_ABL");
      return e != nullptr ? atoi(e) : 0;
    }();
    if (abl == 1)
      MSH_LAUNCH((cross_attention_kernel<80, true, 1>), dim3(M, heads), dim3(256), 0, s, q, row_slot, slots, D, heads, layer, L, Mcap,
                 crossK, crossV, out, fmi, row_mem);
    else if (abl == 2)
      MSH_LAUNCH((cross_attention_kernel<80, true, 2>), dim3(M, heads), dim3(256), 0, s, q, row_slot, slots, D, heads, layer, L, Mcap,
                 crossK, crossV, out, fmi, row_mem);
    else
      MSH_LAUNCH((cross_attention_kernel<80, true>), dim3(M, heads), dim3(256), 0, s, q, row_slot, slots, D, heads, layer, L, Mcap,
                 crossK, crossV, out, fmi, row_mem);
    return;
  }
#define MSH_XATT(DHV)                                                                                                  \
  case DHV:                                                                                                            \
    MSH_LAUNCH(cross_attention_kernel<DHV>, dim3(M, heads), dim3(256), 0, s, q, row_slot, slots, D, heads,     \
                       layer, L, Mcap, crossK, crossV, out, fmi, row_mem);                                            \
    break
  switch (dh) {
    MSH_XATT(16);
    MSH_XATT(24);
    MSH_XATT(36);
    MSH_XATT(40);
    MSH_XATT(52);
    MSH_XATT(64);
    MSH_XATT(80);
    default:
      MSH_LAUNCH(cross_attention_generic_kernel, dim3(M, heads), dim3(256), 0, s, q, row_slot, slots, D, heads,
                         layer, L, Mcap, crossK, crossV, out, fmi);
  }
#undef MSH_XATT
}
void stream_cross_attention_runs(const bf16_t* q, const int* row_slot, const int2* runs, int n_runs, const SlotDev* slots, int D,
                                 int heads, int layer, int L, int Mcap, const bf16_t* crossK, const bf16_t* crossV, bf16_t* out,
                                 hipStream_t s) {
  if (n_runs <= 0) return;
  static_assert(sizeof(RowRun) == sizeof(int2), "RowRun is passed as int2");
  const RowRun* rr = reinterpret_cast<const RowRun*>(runs);
  static const bool runs_nt = [] {   // probe for the next round (default off: not measured yet)
    const char* e = dev_getenv("MSH_STREAM_XRUNS_NT");
    return e != nullptr && e[0] == '1';
  }();
  if (runs_nt && D / heads == 80) {
    MSH_LAUNCH((cross_attention_runs_kernel<80, kCrossRunRows, true>), dim3(n_runs, heads), dim3(256), 0, s, q, row_slot, rr, slots, D,
               heads, layer, L, Mcap, crossK, crossV, out);
    return;
  }
#define MSH_XRUN(DHV)                                                                                                   \
  case DHV:                                                                                                             \
    MSH_LAUNCH((cross_attention_runs_kernel<DHV, kCrossRunRows>), dim3(n_runs, heads), dim3(256), 0, s, q, row_slot, rr, \
               slots, D, heads, layer, L, Mcap, crossK, crossV, out);                                                   \
    break
  switch (D / heads) {
    MSH_XRUN(16);
    MSH_XRUN(24);
    MSH_XRUN(36);
    MSH_XRUN(40);
    MSH_XRUN(52);
    MSH_XRUN(64);
    MSH_XRUN(80);
    default: throw std::runtime_error("stream_cross_attention_runs: unsupported head_dim");
  }
#undef MSH_XRUN
}
bool stream_cross_attention_wide_supported(int D, int heads, int Mcap) {
  const int dh = D / heads;
  return (dh & 7) == 0 && dh >= 16 && dh <= 96 && (D & 7) == 0 && (Mcap & 7) == 0;
}
void stream_cross_attention_wide(const bf16_t* q, const int* row_slot, const int2* runs, int n_runs, const SlotDev* slots, int D,
                                 int heads, int layer, int L, int Mcap, const bf16_t* crossK, const bf16_t* crossV, bf16_t* out,
                                 hipStream_t s) {
  if (n_runs <= 0) return;
  const RowRun* rr = reinterpret_cast<const RowRun*>(runs);
  const int dh = D / heads, ks = (dh + 31) / 32, dt = (dh + 15) / 16;
#define MSH_XWIDE(KSV, DTV)                                                                                                     \
  if (ks == KSV && dt == DTV) {                                                                                                 \
    MSH_LAUNCH((cross_attention_wide_kernel<KSV, DTV>), dim3(n_runs, heads), dim3(512), 0, s, q, row_slot, rr, slots, D, heads, \
               dh, layer, L, Mcap, crossK, crossV, out);                                                                        \
    return;                                                                                                                     \
  }
  MSH_XWIDE(1, 1) MSH_XWIDE(1, 2) MSH_XWIDE(2, 3) MSH_XWIDE(2, 4) MSH_XWIDE(3, 5) MSH_XWIDE(3, 6)
#undef MSH_XWIDE
  throw std::runtime_error("stream_cross_attention_wide: unsupported head_dim");
}
bool stream_cross_attention_runs_supported(int D, int heads, int Mcap) {
  const int dh = D / heads;
  return (dh == 16 || dh == 24 || dh == 36 || dh == 40 || dh == 52 || dh == 64 || dh == 80) && Mcap <= CROSS_MMAX && (Mcap & 7) == 0;
}
void stream_cross_probs(const bf16_t* q, const int* row_slot, const SlotDev* slots, int M, int D, int heads, int layer,
                        int L, int Mcap, const bf16_t* crossK, int Ecap, float* out, hipStream_t s) {
  if (Mcap > CROSS_MMAX || D / heads > 128) throw std::runtime_error("stream_cross_probs: unsupported memory length or head_dim");
  if (M <= 0) return;
  MSH_LAUNCH(cross_probs_kernel, dim3(M, heads), dim3(256), 0, s, q, row_slot, slots, D, heads, layer, L, Mcap,
                     crossK, Ecap, out);
}
void stream_argmax(const float* logits, int M, int V, int* pred, hipStream_t s) {
  if ((V & 3) != 0) throw std::runtime_error("stream_argmax: vocabulary must be a multiple of 4");
  if (M <= 0) return;
  MSH_LAUNCH(argmax_kernel, dim3(M), dim3(256), 0, s, logits, V, pred);
}
void stream_verify(const DecJob* jobs, int n_jobs, const int* pred, const int* draft, SlotDev* slots, int* result,
                   int result_stride, int eos, const float* embed, int D, float* H, int* step_pos, int* n_active,
                   hipStream_t s, bool fm) {
  if (n_jobs <= 0) return;
  MSH_LAUNCH(verify_kernel, dim3(n_jobs), dim3(128), 0, s, jobs, pred, draft, slots, result, result_stride, eos,
                     embed, D, H, step_pos, n_active, fm ? 1 : 0);
}
void stream_advance(const DecJob* jobs, int n_jobs, const int* pred, SlotDev* slots, int* result, int result_stride,
                    int eos, const float* embed, int D, float* H, int* step_pos, int* n_active, hipStream_t s, bool fm) {
  if (n_jobs <= 0) return;
  MSH_LAUNCH(advance_kernel, dim3(n_jobs), dim3(128), 0, s, jobs, pred, slots, result, result_stride, eos, embed,
                     D, H, step_pos, n_active, fm ? 1 : 0);
}
void stream_advance_partials(const DecJob* jobs, int n_jobs, const float* pval, const int* pidx, int ntn, SlotDev* slots,
                             int* result, int result_stride, int eos, const float* embed, int D, float* H, int* step_pos,
                             int* n_active, hipStream_t s, bool fm) {
  if (n_jobs <= 0) return;
  MSH_LAUNCH(advance_partials_kernel, dim3(n_jobs), dim3(128), 0, s, jobs, pval, pidx, ntn, slots, result, result_stride, eos,
             embed, D, H, step_pos, n_active, fm ? 1 : 0);
}
void stream_bias_rows(BiasTrie trie, const int2* prefix, const int* tokens, const DecJob* jobs, const SlotDev* slots,
                      const int* result, int result_stride, int rows, float* logits, int V, hipStream_t s) {
  if (rows <= 0 || trie.n_nodes <= 0) return;
  MSH_LAUNCH(bias_rows_kernel, dim3(rows), dim3(64), 0, s, trie, prefix, tokens, jobs, slots, result,
                     result_stride, logits, V);
}
void stream_slot_update(const int4* upd, int n, SlotDev* slots, hipStream_t s) {
  if (n <= 0) return;
  MSH_LAUNCH(slot_update_kernel, dim3((n + 63) / 64), dim3(64), 0, s, upd, n, slots);
}
void stream_bump_cache(const DecJob* jobs, int n_jobs, SlotDev* slots, hipStream_t s) {
  if (n_jobs <= 0) return;
  MSH_LAUNCH(bump_cache_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, s, jobs, n_jobs, slots);
}

}  // namespace msh
