// Silero VAD (16 kHz v5 network) on the device, batched over the hops of many clips: the kernels behind
// SileroDevice (silero_device.{h,cpp}).  The network is fp32 end to end (its LSTM carries state over hundreds of hops and the
// detector compares the output with a threshold: bf16 MFMA operands would move probabilities by 1e-2), so these are plain
// fp32 kernels: a tiled FMA GEMM for everything that does not depend on the recurrent state -- |STFT| basis, four
// convolutions (as im2col GEMMs), the input half of the LSTM gates: 1.4 MFLOP per 32 ms hop -- and one persistent
// workgroup per clip for the recurrence, with the recurrent weight matrix held in registers.
//
// Host reference: silero_vad.cpp (same arithmetic, other summation order); network: core/silero-vad.cpp:78-173 of the
// reference drives the published model through ONNX Runtime one hop at a time.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "silero_kernels.h"

#include <algorithm>

namespace msh {
namespace {

// frames[(hop * 4 + t)][256] = padded[t * 128 .. t * 128 + 256) of the hop's network input: 64 samples of context (the 64
// samples stored in front of the hop; zeros in front of a clip's first hop), the 512-sample hop, and 64 samples of reflect
// padding on the right (padded[576 + i] = padded[574 - i]).  hop_base[h] = index of the hop's context start in `audio`.
__global__ __launch_bounds__(256) void silero_frames_kernel(const float* __restrict__ audio, const long* __restrict__ hop_base,
                                                            float* __restrict__ frames) {
  const long h = blockIdx.x;
  const float* a = audio + hop_base[h];
  const int tid = threadIdx.x;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int idx = t * 128 + tid;
    frames[(h * 4 + t) * 256 + tid] = idx < 576 ? a[idx] : a[1150 - idx];
  }
}

// mag[hop][b * 4 + t] = |re + i im| from the STFT GEMM's [frame][258] output (129 real then 129 imaginary columns)
__global__ __launch_bounds__(256) void silero_mag_kernel(const float* __restrict__ stft, long n_hops, float* __restrict__ mag) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over hops * 516
  if (i >= n_hops * 516) return;
  const long h = i / 516;
  const int r = (int)(i - h * 516), b = r >> 2, t = r & 3;
  const float* row = stft + (h * 4 + t) * 258;
  const float re = row[b], im = row[129 + b];
  mag[i] = sqrtf(re * re + im * im);
}

// cols[(hop * tout + t)][kpad]: the 3 * cin inputs of output frame t in the weights' [c][k] order (zeros for the conv padding
// and for the columns that pad K to a multiple of 16); in = [hop][cin][tin]
__global__ __launch_bounds__(256) void silero_im2col_kernel(const float* __restrict__ in, int cin, int tin, int stride, int tout,
                                                            int kpad, long n_rows, float* __restrict__ cols) {
  const long row = blockIdx.x;   // hop * tout + t
  if (row >= n_rows) return;
  const long h = row / tout;
  const int t = (int)(row - h * tout);
  const float* x = in + h * cin * tin;
  const int c0 = t * stride - 1;
  for (int j = threadIdx.x; j < kpad; j += 256) {
    float v = 0.f;
    if (j < cin * 3) {
      const int c = j / 3, k = j - c * 3, p = c0 + k;
      if (p >= 0 && p < tin) v = x[c * tin + p];
    }
    cols[row * kpad + j] = v;
  }
}

// ---- epilogues of the GEMM ----
struct EpiPlain {   // C[m][n]
  float* out;
  long ldc;
  __device__ void store(long m, int n, float v) const { out[m * ldc + n] = v; }
};
struct EpiBias {    // C[m][n] + bias[n]
  float* out;
  long ldc;
  const float* bias;
  __device__ void store(long m, int n, float v) const { out[m * ldc + n] = v + bias[n]; }
};
struct EpiBiasReluCT {   // relu(C + bias) into the next layer's [hop][channel][frame] layout; m = hop * tout + t
  float* out;
  const float* bias;
  int tout, cout;
  __device__ void store(long m, int n, float v) const {
    const long h = m / tout;
    const int t = (int)(m - h * tout);
    const float y = v + bias[n];
    out[(h * cout + n) * tout + t] = y > 0.f ? y : 0.f;
  }
};

// C = A[M][K](lda) * B[N][K]^T, fp32, K % 16 == 0, lda % 4 == 0, 16-byte aligned rows.  64 x 64 tile per workgroup, 256
// threads with 4 x 4 outputs each, k-slices of 16 through LDS (stored k-major so that the inner loop reads float4 runs).
template <class Epi>
__global__ __launch_bounds__(256) void sgemm_nt_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                       long ldb, long M, int N, int K, Epi epi) {
  __shared__ __attribute__((aligned(16))) float As[16][64 + 4];
  __shared__ __attribute__((aligned(16))) float Bs[16][64 + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const long m0 = (long)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  const int lrow = tid >> 2, lk = (tid & 3) * 4;   // this thread's float4 of the 64 x 16 tile loads
  long am = m0 + lrow;
  am = am < M ? am : M - 1;
  int bn = n0 + lrow;
  bn = bn < N ? bn : N - 1;
  const float* ap = A + am * lda + lk;
  const float* bp = B + (long)bn * ldb + lk;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 av = *reinterpret_cast<const float4*>(ap + k0);
    const float4 bv = *reinterpret_cast<const float4*>(bp + k0);
    __syncthreads();   // the previous slice is consumed
    As[lk + 0][lrow] = av.x; As[lk + 1][lrow] = av.y; As[lk + 2][lrow] = av.z; As[lk + 3][lrow] = av.w;
    Bs[lk + 0][lrow] = bv.x; Bs[lk + 1][lrow] = bv.y; Bs[lk + 2][lrow] = bv.z; Bs[lk + 3][lrow] = bv.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
      const float ar[4] = {a.x, a.y, a.z, a.w}, br[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += ar[i] * br[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n < N) epi.store(m, n, acc[i][j]);
    }
  }
}

template <class Epi>
void launch_sgemm(const float* A, long lda, const float* B, long ldb, long M, int N, int K, Epi epi, hipStream_t s) {
  if (M <= 0) return;
  if ((K & 15) != 0 || (lda & 3) != 0 || (ldb & 3) != 0) throw std::runtime_error("silero sgemm: K % 16, lda % 4, ldb % 4 required");
  const long mt = (M + 63) / 64;
  if (mt > 0x7fffffffL) throw std::runtime_error("silero sgemm: too many rows");
  MSH_LAUNCH((sgemm_nt_kernel<Epi>), dim3((unsigned)mt, (unsigned)((N + 63) / 64)), dim3(256), 0, s, A, lda, B, ldb, M, N, K, epi);
}

// The recurrence.  One workgroup of 512 threads walks the hops of a clip in order (and then takes the next clip): thread r
// keeps row r of the recurrent weight matrix [512][128] in registers for its whole life, h sits in LDS.  Per hop: 128 FMAs
// per thread, the gate exchange, the cell update on 128 threads, the output unit.  gin = input half of the gates + both
// biases (from the GEMM).  clip_hop0[c] .. clip_hop0[c + 1] = the clip's hops.
__global__ __launch_bounds__(512) void silero_lstm_kernel(const float* __restrict__ gin, const float* __restrict__ w_hh,
                                                          const float* __restrict__ out_w, float out_b,
                                                          const long* __restrict__ clip_hop0, int n_clips,
                                                          float* __restrict__ probs) {
  __shared__ __attribute__((aligned(16))) float h[128];
  __shared__ float gates[512];
  __shared__ float part[2];
  const int r = threadIdx.x;
  float w[128];
#pragma unroll
  for (int k = 0; k < 128; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(w_hh + (long)r * 128 + k);
    w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
  }
  const float ow = r < 128 ? out_w[r] : 0.f;
  for (int c = blockIdx.x; c < n_clips; c += gridDim.x) {
    const long h0 = clip_hop0[c], h1 = clip_hop0[c + 1];
    float cell = 0.f;
    if (r < 128) h[r] = 0.f;
    __syncthreads();
    for (long hop = h0; hop < h1; ++hop) {
      float g = gin[hop * 512 + r];
#pragma unroll
      for (int k = 0; k < 128; k += 4) {
        const float4 hv = *reinterpret_cast<const float4*>(&h[k]);
        g += w[k] * hv.x + w[k + 1] * hv.y + w[k + 2] * hv.z + w[k + 3] * hv.w;
      }
      gates[r] = g;
      __syncthreads();   // every row has read h; the gates are complete
      if (r < 128) {
        const float ig = 1.0f / (1.0f + expf(-gates[r])), fg = 1.0f / (1.0f + expf(-gates[128 + r]));
        const float gg = tanhf(gates[256 + r]), og = 1.0f / (1.0f + expf(-gates[384 + r]));
        cell = fg * cell + ig * gg;
        const float hn = og * tanhf(cell);
        h[r] = hn;
        float p = ow * (hn > 0.f ? hn : 0.f);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        if ((r & 63) == 0) part[r >> 6] = p;
      }
      __syncthreads();   // the new h and the two partial sums are visible
      if (r == 0) probs[hop] = 1.0f / (1.0f + expf(-(out_b + (part[0] + part[1]))));
    }
    __syncthreads();     // thread 0 has read part[] before the next clip's first step rewrites it
  }
}

// 16-bit PCM -> fp32 in [-1, 1): x / 32768, exact (the host detectors convert the same way)
__global__ __launch_bounds__(256) void pcm16_to_f32_kernel(const short* __restrict__ src, float* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const short4 v = reinterpret_cast<const short4*>(src)[i];
    reinterpret_cast<float4*>(dst)[i] = make_float4((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f),
                                                    (float)v.z * (1.0f / 32768.0f), (float)v.w * (1.0f / 32768.0f));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = (float)src[n4 * 4 + threadIdx.x] * (1.0f / 32768.0f);
}

}  // namespace

void silero_pcm16_to_f32(const int16_t* src, float* dst, long n, hipStream_t s) {
  if (n <= 0) return;
  const long blocks = std::min<long>((n / 4 + 255) / 256 + 1, 4096);
  MSH_LAUNCH(pcm16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const short*>(src), dst, n);
}

void silero_frames(const float* audio, const long* hop_base, long n_hops, float* frames, hipStream_t s) {
  if (n_hops <= 0) return;
  MSH_LAUNCH(silero_frames_kernel, dim3((unsigned)n_hops), dim3(256), 0, s, audio, hop_base, frames);
}
void silero_stft_mag(const float* frames, const float* basis, long n_hops, float* stft_tmp, float* mag, hipStream_t s) {
  if (n_hops <= 0) return;
  launch_sgemm(frames, 256, basis, 256, n_hops * 4, 258, 256, EpiPlain{stft_tmp, 258}, s);
  MSH_LAUNCH(silero_mag_kernel, dim3((unsigned)((n_hops * 516 + 255) / 256)), dim3(256), 0, s, stft_tmp, n_hops, mag);
}
void silero_conv_relu(const float* in, int cin, int tin, int stride, const float* w_padded, int kpad, const float* bias, int cout,
                      long n_hops, float* cols, float* out, hipStream_t s) {
  if (n_hops <= 0) return;
  const int tout = (tin - 1) / stride + 1;
  const long rows = n_hops * tout;
  MSH_LAUNCH(silero_im2col_kernel, dim3((unsigned)rows), dim3(256), 0, s, in, cin, tin, stride, tout, kpad, rows, cols);
  launch_sgemm(cols, kpad, w_padded, kpad, rows, cout, kpad, EpiBiasReluCT{out, bias, tout, cout}, s);
}
void silero_gate_inputs(const float* feat, const float* w_ih, const float* bias_sum, long n_hops, float* gin, hipStream_t s) {
  launch_sgemm(feat, 128, w_ih, 128, n_hops, 512, 128, EpiBias{gin, 512, bias_sum}, s);
}
void silero_lstm(const float* gin, const float* w_hh, const float* out_w, float out_b, const long* clip_hop0, int n_clips,
                 float* probs, hipStream_t s) {
  if (n_clips <= 0) return;
  const int wgs = n_clips < 512 ? n_clips : 512;   // two workgroups per CU at most; a workgroup takes clips round-robin
  MSH_LAUNCH(silero_lstm_kernel, dim3(wgs), dim3(512), 0, s, gin, w_hh, out_w, out_b, clip_hop0, n_clips, probs);
}

}  // namespace msh
