// Memory-bound helper kernels: audio packing, GroupNorm, LayerNorm, decode bookkeeping (argmax).
// All are coalesced 16-byte-per-lane streams; reductions are wave shuffles + one LDS hop, and are
// deterministic (no floating-point atomics) so the same batch always yields the same token ids.
#include "kernels.h"

namespace msh {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_audio_kernel(const float* const* __restrict__ clip_ptrs,
                                                         const ClipMeta* __restrict__ clips,
                                                         bf16_t* __restrict__ out) {
  const ClipMeta cm = clips[blockIdx.y];
  const float* src = clip_ptrs[blockIdx.y];
  const long total = 384L * cm.rows;  // this clip's slot in the conv1 input stream
  bf16_t* dst = out + 384L * cm.row_start;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 1024) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (i + e < cm.n_samples) ? src[i + e] : 0.0f;
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(dst + i) = o;
  }
}

__global__ __launch_bounds__(256) void build_row_meta_kernel(const ClipMeta* __restrict__ clips,
                                                             int* __restrict__ row_pos, int* __restrict__ row_clip) {
  const ClipMeta cm = clips[blockIdx.y];
  for (int t = blockIdx.x * 256 + threadIdx.x; t < cm.rows; t += gridDim.x * 256) {
    row_pos[cm.row_start + t] = t < cm.T ? t : -1;
    row_clip[cm.row_start + t] = blockIdx.y;
  }
}

// ---------------------------------------------------------------------------------------------
constexpr int GN_CHUNKS = 64;

__global__ __launch_bounds__(256) void groupnorm_partial_kernel(const bf16_t* __restrict__ x1,
                                                                const ClipMeta* __restrict__ clips, int D,
                                                                float2* __restrict__ partials) {
  __shared__ float2 red[4];
  const ClipMeta cm = clips[blockIdx.y];
  const long n8 = (long)cm.L1 * D / 8;  // valid block is contiguous: rows [0, L1) x D (D % 8 == 0)
  const uint4* p = reinterpret_cast<const uint4*>(x1 + 6L * cm.row_start * D);
  const long per = (n8 + GN_CHUNKS - 1) / GN_CHUNKS;
  const long lo = blockIdx.x * per, hi = (lo + per < n8) ? lo + per : n8;
  float s = 0.f, ss = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const uint4 u = p[i];
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = __uint_as_float(w[k] << 16), b = __uint_as_float(w[k] & 0xffff0000u);
      s += a + b;
      ss += a * a + b * b;
    }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(s, ss);
  __syncthreads();
  if (threadIdx.x == 0) {
    float2 a = red[0];
    for (int i = 1; i < 4; ++i) {
      a.x += red[i].x;
      a.y += red[i].y;
    }
    partials[blockIdx.y * GN_CHUNKS + blockIdx.x] = a;
  }
}

// The same partial sums from the row sums conv1's epilogue left (EpiTanhBf16::rowsum, [6 R][ntn] {sum, sum of squares}): a clip's
// valid rows are one contiguous run of L1 * ntn entries.  80 KB per 10 s clip instead of 4 MB of bf16 values.
__global__ __launch_bounds__(256) void groupnorm_rows_partial_kernel(const float2* __restrict__ rowsum, int ntn,
                                                                     const ClipMeta* __restrict__ clips,
                                                                     float2* __restrict__ partials) {
  __shared__ float2 red[4];
  const ClipMeta cm = clips[blockIdx.y];
  const long n = (long)cm.L1 * ntn;
  const float2* p = rowsum + 6L * cm.row_start * ntn;
  const long per = (n + GN_CHUNKS - 1) / GN_CHUNKS;
  const long lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  float s = 0.f, ss = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float2 v = p[i];
    s += v.x;
    ss += v.y;
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(s, ss);
  __syncthreads();
  if (threadIdx.x == 0) {
    float2 a = red[0];
    for (int i = 1; i < 4; ++i) {
      a.x += red[i].x;
      a.y += red[i].y;
    }
    partials[blockIdx.y * GN_CHUNKS + blockIdx.x] = a;
  }
}

__global__ void gn_fold_table_kernel(const float2* __restrict__ stats, const float* __restrict__ s1,
                                     const float* __restrict__ b2, int N, float* __restrict__ table) {
  const float2 st = stats[blockIdx.x];
  const float c = st.x * st.y;
  for (int n = threadIdx.x; n < N; n += blockDim.x) table[(long)blockIdx.x * N + n] = b2[n] - c * s1[n];
}

__global__ void groupnorm_final_kernel(const float2* __restrict__ partials, const ClipMeta* __restrict__ clips,
                                       int n_clips, int D, float2* __restrict__ stats) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_clips) return;
  double s = 0.0, ss = 0.0;
  for (int i = 0; i < GN_CHUNKS; ++i) {
    s += (double)partials[b * GN_CHUNKS + i].x;
    ss += (double)partials[b * GN_CHUNKS + i].y;
  }
  const double n = (double)clips[b].L1 * D;
  const double mean = s / n;
  double var = ss / n - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  stats[b] = make_float2((float)mean, (float)(1.0 / sqrt(var + 1e-5)));
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row lives in registers (two-pass mean / variance, eps 1e-5).
// FMIN: x is the decode path's fragment-major residual stream (kernels.h fm32); the output stays row-major.
template <int NCH, bool FMIN = false>  // float4 chunks per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        int rows, int D, bf16_t* __restrict__ y,
                                                        float* __restrict__ y32) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int D4 = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (long)row * D);
  float4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if constexpr (FMIN)
      v[i] = c < D4 ? *reinterpret_cast<const float4*>(x + fm32(row, 4 * c, D >> 5)) : make_float4(0.f, 0.f, 0.f, 0.f);
    else
      v[i] = c < D4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (lane + 64 * i < D4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + 1e-5f);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < D4) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c];
      const float4 o = make_float4((v[i].x - mean) * rstd * g.x, (v[i].y - mean) * rstd * g.y,
                                   (v[i].z - mean) * rstd * g.z, (v[i].w - mean) * rstd * g.w);
      uint2 w;
      w.x = pack_bf16x2(o.x, o.y);
      w.y = pack_bf16x2(o.z, o.w);
      reinterpret_cast<uint2*>(y + (long)row * D)[c] = w;
      if (y32 != nullptr) reinterpret_cast<float4*>(y32 + (long)row * D)[c] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
__global__ void decode_begin_kernel(int M, DecodeState st, int bos, const float* __restrict__ embed, int D,
                                    float* __restrict__ H) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    st.tokens[(long)b * st.stride] = bos;
    st.counts[b] = 1;
    st.finished[b] = 0;
    if (b == 0) {
      *st.pos = 0;
      *st.n_active = M;
    }
  }
  for (int d = threadIdx.x; d < D; d += blockDim.x) H[fm32(b, d, D >> 5)] = embed[(long)bos * D + d];   // H is FM
}

// Loop bookkeeping of reference core/moonshine-model.cpp:511-516 for clip b once its token is chosen: append, stop
// on EOS / step budget; returns the id the next step feeds.
__device__ __forceinline__ int advance_bookkeeping(int besti, int b, const ClipMeta* __restrict__ clips, DecodeState st) {
  if (besti == 0x7fffffff) besti = 0;  // all-NaN row: behave like the linear scan (index 0)
  const int cnt = st.counts[b];
  st.tokens[(long)b * st.stride + cnt] = besti;
  st.counts[b] = cnt + 1;
  int nxt = besti;
  if (st.forced != nullptr && cnt < st.stride) nxt = st.forced[(long)b * st.stride + cnt];
  // cnt generated tokens so far (BOS excluded): stop on EOS or when the step budget is used up
  if ((besti == st.eos && !st.ignore_eos) || cnt >= clips[b].max_len) {
    st.finished[b] = 1;
    atomicSub(st.n_active, 1);
  }
  return nxt;
}

// One block per clip: first-max argmax (strict '>' scan order, ties -> lowest index, the rule of
// reference core/ort-utils/moonshine-tensor-view.cpp:222-236), then the loop bookkeeping of
// reference core/moonshine-model.cpp:511-516 (append, stop on EOS / step budget, next input id).
__global__ __launch_bounds__(1024) void decode_advance_kernel(const float* __restrict__ logits, int V,
                                                             const ClipMeta* __restrict__ clips, DecodeState st,
                                                             const float* __restrict__ embed, int D,
                                                             float* __restrict__ H) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  __shared__ int next_tok;
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool done = st.finished[b] != 0;
  if (!done) {
    const float4* lp = reinterpret_cast<const float4*>(logits + (long)b * V);
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int i = tid; i < (V >> 2); i += 1024) {
      const float4 v = lp[i];
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // indices visited by one thread increase monotonically, so '>' keeps the first maximum
        if (e[k] > best) {
          best = e[k];
          besti = i * 4 + k;
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ov = __shfl_xor(best, o);
      const int oi = __shfl_xor(besti, o);
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
      for (int w = 1; w < 16; ++w)
        if (bv[w] > best || (bv[w] == best && bi[w] < besti)) {
          best = bv[w];
          besti = bi[w];
        }
      next_tok = advance_bookkeeping(besti, b, clips, st);
    }
    __syncthreads();
    const int nt = next_tok;
    for (int d = tid; d < D; d += 1024) H[fm32(b, d, D >> 5)] = embed[(long)nt * D + d];
  }
  if (b == 0 && tid == 0) *st.pos += 1;
}

// The same step from per-tile (max, first index) pairs: tiles are in ascending column order, so among equal maxima
// the lowest index is the first maximum of the whole row.
__global__ __launch_bounds__(256) void decode_advance_partials_kernel(const float* __restrict__ pval,
                                                                      const int* __restrict__ pidx, int ntn,
                                                                      const ClipMeta* __restrict__ clips,
                                                                      DecodeState st, const float* __restrict__ embed,
                                                                      int D, float* __restrict__ H) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  __shared__ int next_tok;
  const int b = blockIdx.x, tid = threadIdx.x;
  // the partial maxima are fetched before the "finished" flag is known: one memory round trip instead of two
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int i = tid; i < ntn; i += 256) {
    const float v = pval[(long)b * ntn + i];
    const int ix = pidx[(long)b * ntn + i];
    if (v > best || (v == best && ix < besti)) {
      best = v;
      besti = ix;
    }
  }
  if (st.finished[b] == 0) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ov = __shfl_xor(best, o);
      const int oi = __shfl_xor(besti, o);
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
      for (int w = 1; w < 4; ++w)
        if (bv[w] > best || (bv[w] == best && bi[w] < besti)) {
          best = bv[w];
          besti = bi[w];
        }
      next_tok = advance_bookkeeping(besti, b, clips, st);
    }
    __syncthreads();
    const int nt = next_tok;
    for (int d = tid; d < D; d += 256) H[fm32(b, d, D >> 5)] = embed[(long)nt * D + d];
  }
  if (b == 0 && tid == 0) *st.pos += 1;
}

}  // namespace

void pack_audio(const float* const* clip_ptrs, const ClipMeta* clips, int n_clips, bf16_t* out, long /*out_elems*/,
                hipStream_t s) {
  MSH_LAUNCH(pack_audio_kernel, dim3(32, n_clips), dim3(256), 0, s, clip_ptrs, clips, out);
}

void build_row_meta(const ClipMeta* clips, int n_clips, int* row_pos, int* row_clip, hipStream_t s) {
  MSH_LAUNCH(build_row_meta_kernel, dim3(2, n_clips), dim3(256), 0, s, clips, row_pos, row_clip);
}

void groupnorm_stats(const bf16_t* x1, const ClipMeta* clips, int n_clips, int D, float* partials, float2* stats,
                     hipStream_t s) {
  if ((D & 7) != 0) throw std::runtime_error("groupnorm_stats: width must be a multiple of 8");
  MSH_LAUNCH(groupnorm_partial_kernel, dim3(GN_CHUNKS, n_clips), dim3(256), 0, s, x1, clips, D,
                     reinterpret_cast<float2*>(partials));
  MSH_LAUNCH(groupnorm_final_kernel, dim3((n_clips + 63) / 64), dim3(64), 0, s,
                     reinterpret_cast<const float2*>(partials), clips, n_clips, D, stats);
}

void groupnorm_stats_rows(const float2* rowsum, int ntn, const ClipMeta* clips, int n_clips, int D, float* partials,
                          float2* stats, hipStream_t s) {
  MSH_LAUNCH(groupnorm_rows_partial_kernel, dim3(GN_CHUNKS, n_clips), dim3(256), 0, s, rowsum, ntn, clips,
                     reinterpret_cast<float2*>(partials));
  MSH_LAUNCH(groupnorm_final_kernel, dim3((n_clips + 63) / 64), dim3(64), 0, s,
                     reinterpret_cast<const float2*>(partials), clips, n_clips, D, stats);
}

void gn_fold_table(const float2* stats, const float* s1, const float* b2, int n_clips, int N, float* table,
                   hipStream_t s) {
  MSH_LAUNCH(gn_fold_table_kernel, dim3(n_clips), dim3(256), 0, s, stats, s1, b2, N, table);
}

void layernorm_bf16(const float* x, const float* gamma, int rows, int D, bf16_t* y, float* y_f32, hipStream_t s) {
  const int nch = (D / 4 + 63) / 64;
  dim3 grid((rows + 3) / 4);
  if ((D & 3) != 0 || nch > 4) throw std::runtime_error("layernorm: unsupported width");
  switch (nch) {
    case 1: MSH_LAUNCH(layernorm_kernel<1>, grid, dim3(256), 0, s, x, gamma, rows, D, y, y_f32); break;
    case 2: MSH_LAUNCH(layernorm_kernel<2>, grid, dim3(256), 0, s, x, gamma, rows, D, y, y_f32); break;
    default: MSH_LAUNCH(layernorm_kernel<4>, grid, dim3(256), 0, s, x, gamma, rows, D, y, y_f32); break;
  }
}

void dec_final_layernorm(const float* H, const float* gamma, int M, int D, bf16_t* y, hipStream_t s) {
  const int nch = (D / 4 + 63) / 64;
  dim3 grid((M + 3) / 4);
  if ((D & 31) != 0 || nch > 4) throw std::runtime_error("dec_final_layernorm: unsupported width");
  switch (nch) {
    case 1: MSH_LAUNCH((layernorm_kernel<1, true>), grid, dim3(256), 0, s, H, gamma, M, D, y, (float*)nullptr); break;
    case 2: MSH_LAUNCH((layernorm_kernel<2, true>), grid, dim3(256), 0, s, H, gamma, M, D, y, (float*)nullptr); break;
    default: MSH_LAUNCH((layernorm_kernel<4, true>), grid, dim3(256), 0, s, H, gamma, M, D, y, (float*)nullptr); break;
  }
}

void decode_begin(int M, DecodeState st, int bos, const float* embed_f32, int D, float* H, hipStream_t s) {
  MSH_LAUNCH(decode_begin_kernel, dim3(M), dim3(128), 0, s, M, st, bos, embed_f32, D, H);
}

void decode_advance(const float* logits, int M, int V, const ClipMeta* clips, DecodeState st, const float* embed_f32,
                    int D, float* H, hipStream_t s) {
  MSH_LAUNCH(decode_advance_kernel, dim3(M), dim3(1024), 0, s, logits, V, clips, st, embed_f32, D, H);
}

void decode_advance_partials(const float* pval, const int* pidx, int ntn, int M, const ClipMeta* clips, DecodeState st,
                             const float* embed_f32, int D, float* H, hipStream_t s) {
  MSH_LAUNCH(decode_advance_partials_kernel, dim3(M), dim3(256), 0, s, pval, pidx, ntn, clips, st, embed_f32, D,
                     H);
}

}  // namespace msh
