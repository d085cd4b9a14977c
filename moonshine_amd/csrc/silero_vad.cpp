#include "silero_vad.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <stdexcept>

#include "safetensors.h"

namespace msh_host {
namespace {
constexpr int kBins = 129, kFft = 256, kStftHop = 128, kFrames = 4, kPad = 64;
const int kConvIn[4] = {129, 128, 64, 64}, kConvOut[4] = {128, 64, 64, 128}, kConvStride[4] = {1, 2, 2, 1};

std::vector<float> take(const msh::SafeTensors& st, const std::string& name, std::initializer_list<int64_t> shape) {
  std::string key = name;
  if (!st.has(key)) key = "_model." + name;
  if (!st.has(key)) throw std::runtime_error("Silero VAD weights: tensor '" + name + "' is missing");
  const msh::StTensor& t = st.get(key);
  int64_t n = 1;
  for (int64_t d : shape) n *= d;
  if (t.numel() != n)
    throw std::runtime_error("Silero VAD weights: tensor '" + name + "' has " + std::to_string(t.numel()) + " elements, expected " +
                             std::to_string(n) + " (is this the 16 kHz v5 model?)");
  return st.to_f32(key);
}

void fill(SileroWeights* w, const msh::SafeTensors& st) {
  w->stft = take(st, "stft.forward_basis_buffer", {2 * kBins, 1, kFft});
  for (int i = 0; i < 4; ++i) {
    const std::string p = "encoder." + std::to_string(i) + ".reparam_conv.";
    w->conv_w[i] = take(st, p + "weight", {kConvOut[i], kConvIn[i], 3});
    w->conv_b[i] = take(st, p + "bias", {kConvOut[i]});
  }
  w->w_ih = take(st, "decoder.rnn.weight_ih", {4 * SileroVad::kState, SileroVad::kState});
  w->w_hh = take(st, "decoder.rnn.weight_hh", {4 * SileroVad::kState, SileroVad::kState});
  w->b_ih = take(st, "decoder.rnn.bias_ih", {4 * SileroVad::kState});
  w->b_hh = take(st, "decoder.rnn.bias_hh", {4 * SileroVad::kState});
  w->out_w = take(st, "decoder.decoder.2.weight", {1, SileroVad::kState, 1});
  w->out_b = take(st, "decoder.decoder.2.bias", {1})[0];
}

inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// exp for the LSTM cell's 640 activations per hop, written so that the loops over the 128 units vectorise (libm's expf /
// tanhf are calls: a quarter of a hop's time).  Cody-Waite range reduction by ln 2, degree-6 Taylor polynomial on
// |r| <= ln 2 / 2 (relative error < 2e-7), scaling by 2^n through the exponent bits.  Inputs are clamped to +-87.
static inline __attribute__((always_inline)) float exp_fast(float x) {
  x = x < -87.0f ? -87.0f : (x > 87.0f ? 87.0f : x);
  const float t = x * 1.44269504088896341f;
  const float nf = (t + 12582912.0f) - 12582912.0f;          // round to nearest (|t| < 2^22)
  const float r = (x - nf * 0.693145751953125f) - nf * 1.42860682030941723212e-6f;
  float p = 1.0f / 720.0f;
  p = p * r + 1.0f / 120.0f;
  p = p * r + 1.0f / 24.0f;
  p = p * r + 1.0f / 6.0f;
  p = p * r + 0.5f;
  p = p * r + 1.0f;
  p = p * r + 1.0f;
  int32_t bits;
  const int32_t n = (int32_t)nf;
  bits = (n + 127) << 23;
  float scale;
  memcpy(&scale, &bits, sizeof(scale));
  return p * scale;
}
static inline __attribute__((always_inline)) float sigmoid_fast(float x) { return 1.0f / (1.0f + exp_fast(-x)); }
// tanh(x) = 1 - 2 / (1 + e^{2x}): absolute error ~1e-7 everywhere, saturates cleanly
static inline __attribute__((always_inline)) float tanh_fast(float x) { return 1.0f - 2.0f / (1.0f + exp_fast(2.0f * x)); }

// Dot products with eight running sums in a fixed order (one 256-bit or two 128-bit vector accumulators): the plain
// `acc += a[k] * b[k]` loop is a serial dependency the compiler may not reassociate, i.e. scalar code -- 218 us per
// 32 ms hop, which made the VAD (not the GPU) the slowest stage of a batch call.  MSH_SIMD_CLONES builds the hop function
// for AVX2+FMA as well and lets the loader pick (the baseline x86-64 build is SSE2).
typedef float v8f __attribute__((vector_size(32), aligned(4)));
#define MSH_INLINE static inline __attribute__((always_inline))
MSH_INLINE float hsum(v8f s) { return ((s[0] + s[4]) + (s[2] + s[6])) + ((s[1] + s[5]) + (s[3] + s[7])); }
MSH_INLINE float dot(const float* a, const float* b, int n) {
  v8f s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= n; k += 8) s += *reinterpret_cast<const v8f*>(a + k) * *reinterpret_cast<const v8f*>(b + k);
  float tail = 0.f;
  for (; k < n; ++k) tail += a[k] * b[k];
  return hsum(s) + tail;
}

// NB dots of ONE weight row with NB input vectors: the row is loaded once per 8 floats and feeds NB accumulators.  Every
// result is the bit pattern dot() gives (same eight running sums, same order): only the loads are shared.  A hop's network
// reads ~1.5 MB of weights; one hop at a time that is 1.5 MB of L2 traffic per 32 ms of audio and two loads per multiply-add.
template <int NB>
MSH_INLINE void dot_block(const float* w, const float* const* x, int n, float* out) {
  v8f s[NB];
  for (int j = 0; j < NB; ++j) s[j] = v8f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    const v8f wv = *reinterpret_cast<const v8f*>(w + k);
    for (int j = 0; j < NB; ++j) s[j] += wv * *reinterpret_cast<const v8f*>(x[j] + k);
  }
  for (int j = 0; j < NB; ++j) {
    float tail = 0.f;
    for (int kk = k; kk < n; ++kk) tail += w[kk] * x[j][kk];
    out[j] = hsum(s[j]) + tail;
  }
}
// out[j] = dot(w, x[j], n) for j < count (count vectors, any number)
MSH_INLINE void dot_many(const float* w, const float* const* x, int count, int n, float* out) {
  int j = 0;
  for (; j + 8 <= count; j += 8) dot_block<8>(w, x + j, n, out + j);   // 8 independent FMA chains keep both FMA ports busy
  for (; j + 4 <= count; j += 4) dot_block<4>(w, x + j, n, out + j);
  for (; j < count; ++j) out[j] = dot(w, x[j], n);
}

// Conv1d(k = 3, padding = 1, stride s) + ReLU over [C_in][T_in] -> [C_out][T_out], T_out = (T_in - 1) / s + 1.
// Per output frame the 3 * C_in inputs are gathered once in the weights' [c][k] order (zeros for the padding), then every
// output channel is one contiguous dot product.
MSH_INLINE int conv_relu(const float* x, int cin, int tin, const float* w, const float* b, int cout, int stride, float* y) {
  const int tout = (tin - 1) / stride + 1;
  float col[129 * 3];
  for (int t = 0; t < tout; ++t) {
    const int c0 = t * stride - 1;
    for (int c = 0; c < cin; ++c)
      for (int k = 0; k < 3; ++k) {
        const int p = c0 + k;
        col[c * 3 + k] = (p >= 0 && p < tin) ? x[(size_t)c * tin + p] : 0.f;
      }
    for (int o = 0; o < cout; ++o) {
      const float acc = b[o] + dot(w + (size_t)o * cin * 3, col, cin * 3);
      y[(size_t)o * tout + t] = acc > 0.f ? acc : 0.f;
    }
  }
  return tout;
}
}  // namespace

void SileroWeights::load_file(const std::string& path) {
  msh::SafeTensors st;
  st.load_file(path);
  fill(this, st);
}
void SileroWeights::load_memory(const uint8_t* data, size_t size) {
  msh::SafeTensors st;
  st.parse(data, size);
  fill(this, st);
}

SileroVad::SileroVad(std::shared_ptr<const SileroWeights> w) : w_(std::move(w)) {
  if (!w_) throw std::runtime_error("Silero VAD: no weights");
  reset();
}

void SileroVad::reset() {
  memset(context_, 0, sizeof(context_));
  memset(state_, 0, sizeof(state_));
}

#if defined(__x86_64__) && defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
#define MSH_SIMD_CLONES __attribute__((target_clones("avx2,fma", "default")))
#else
#define MSH_SIMD_CLONES
#endif

namespace {
constexpr int kGroup = SileroVad::kGroup;
constexpr int kIn = SileroVad::kContext + SileroVad::kHop + kPad;   // 640 samples per hop: context + hop + reflect padding

// G <= kGroup consecutive hops through the network.  x = [G][640] padded inputs, state = [h | c] (updated), probs[G].
// The convolutional front (|STFT| and four conv + ReLU blocks) and the input half of the LSTM gates do not depend on the
// recurrent state: they run for the whole group at once, every weight row loaded once per group (dot_many); only the
// recurrent half of the gates and the cell update walk the hops in order.  Per hop the arithmetic and its order are those
// of a group of one, so chunked and whole-buffer feeding give identical probabilities.
MSH_SIMD_CLONES void silero_group(const SileroWeights& W, const float* x, int G, float* state, float* probs, float* scratch) {
  constexpr int S = SileroVad::kState;
  const int NF = G * kFrames;
  float* mag = scratch;                               // [G][129 * 4]
  float* bufa = mag + kGroup * kBins * kFrames;       // [G][128 * 4]
  float* bufc = bufa + kGroup * 128 * kFrames;        // [G][128 * 4]
  float* cols = bufc + kGroup * 128 * kFrames;        // [G * 4][129 * 3]
  float* gin = cols + kGroup * kFrames * kBins * 3;   // [G][512]
  const float* ptr[kGroup * kFrames];
  float re[kGroup * kFrames], im[kGroup * kFrames];
  // |STFT|: conv1d with the basis, stride 128 -> [258][4] per hop; magnitude over (real, imag) -> [129][4]
  for (int g = 0; g < G; ++g)
    for (int t = 0; t < kFrames; ++t) ptr[g * kFrames + t] = x + (size_t)g * kIn + t * kStftHop;
  const float* basis = W.stft.data();
  for (int b = 0; b < kBins; ++b) {
    dot_many(basis + (size_t)b * kFft, ptr, NF, kFft, re);
    dot_many(basis + (size_t)(b + kBins) * kFft, ptr, NF, kFft, im);
    for (int f = 0; f < NF; ++f)
      mag[(size_t)(f / kFrames) * kBins * kFrames + b * kFrames + (f % kFrames)] = sqrtf(re[f] * re[f] + im[f] * im[f]);
  }
  // four Conv1d(k = 3, padding = 1, stride s) + ReLU blocks: per output frame the 3 * C_in inputs are gathered once in the
  // weights' [c][k] order (zeros for the padding), then every output channel is one dot product per (hop, frame)
  int tin = kFrames;
  const float* in = mag;
  int in_stride = kBins * kFrames;
  float* outs[2] = {bufa, bufc};
  for (int i = 0; i < 4; ++i) {
    const int cin = kConvIn[i], cout = kConvOut[i], stride = kConvStride[i];
    const int tout = (tin - 1) / stride + 1, klen = cin * 3;
    float* y = outs[i & 1];
    const int y_stride = 128 * kFrames;
    for (int g = 0; g < G; ++g)
      for (int t = 0; t < tout; ++t) {
        float* col = cols + (size_t)(g * tout + t) * klen;
        const float* xg = in + (size_t)g * in_stride;
        const int c0 = t * stride - 1;
        for (int c = 0; c < cin; ++c)
          for (int k = 0; k < 3; ++k) {
            const int p = c0 + k;
            col[c * 3 + k] = (p >= 0 && p < tin) ? xg[(size_t)c * tin + p] : 0.f;
          }
        ptr[g * tout + t] = col;
      }
    const float* w = W.conv_w[i].data();
    const float* b = W.conv_b[i].data();
    for (int o = 0; o < cout; ++o) {
      dot_many(w + (size_t)o * klen, ptr, G * tout, klen, re);
      for (int g = 0; g < G; ++g)
        for (int t = 0; t < tout; ++t) {
          const float acc = b[o] + re[g * tout + t];
          y[(size_t)g * y_stride + (size_t)o * tout + t] = acc > 0.f ? acc : 0.f;
        }
    }
    in = y;
    in_stride = y_stride;
    tin = tout;
  }
  // tin == 1: `in` holds [128] features per hop.  Input half of the LSTM gates for the whole group
  for (int g = 0; g < G; ++g) ptr[g] = in + (size_t)g * in_stride;
  for (int r = 0; r < 4 * S; ++r) {
    dot_many(W.w_ih.data() + (size_t)r * S, ptr, G, S, re);
    for (int g = 0; g < G; ++g) gin[(size_t)g * 4 * S + r] = re[g];
  }
  // the recurrence, hop by hop: gates i, f, g, o
  float* h = state;
  float* cs = state + S;
  float gates[4 * S];
  for (int g = 0; g < G; ++g) {
    const float* gi = gin + (size_t)g * 4 * S;
    for (int r = 0; r < 4 * S; ++r) gates[r] = (W.b_ih[r] + W.b_hh[r]) + (gi[r] + dot(W.w_hh.data() + (size_t)r * S, h, S));
    for (int k = 0; k < S; ++k) {   // (vectorises: no calls, no cross-iteration dependency)
      const float ig = sigmoid_fast(gates[k]), fg = sigmoid_fast(gates[S + k]), gg = tanh_fast(gates[2 * S + k]),
                  og = sigmoid_fast(gates[3 * S + k]);
      const float cn = fg * cs[k] + ig * gg;
      cs[k] = cn;
      h[k] = og * tanh_fast(cn);    // every gate row above has read the old h already
    }
    float logit = W.out_b;
    for (int k = 0; k < S; ++k) logit += W.out_w[k] * (h[k] > 0.f ? h[k] : 0.f);   // in unit order, as before
    probs[g] = sigmoidf(logit);
  }
}
}  // namespace

void SileroVad::predict_many(const float* hops, size_t n_hops, float* probs) {
  constexpr int n = kContext + kHop;
  if (scratch_.empty())
    scratch_.resize((size_t)kGroup * kIn + (size_t)kGroup * (kBins * kFrames + 2 * 128 * kFrames + kFrames * kBins * 3 + 4 * kState));
  float* x = scratch_.data();
  float* work = x + (size_t)kGroup * kIn;
  for (size_t done = 0; done < n_hops;) {
    const int G = (int)std::min<size_t>(kGroup, n_hops - done);
    for (int g = 0; g < G; ++g) {
      // input = context (64) + hop (512), reflect-padded by 64 on the right: padded[576 + i] = input[574 - i]
      float* xg = x + (size_t)g * kIn;
      const float* hop = hops + (done + g) * kHop;
      if (g == 0) memcpy(xg, context_, sizeof(context_));
      else memcpy(xg, hop - kContext, sizeof(context_));   // the last 64 samples of the previous hop (silero-vad.cpp:163-164)
      memcpy(xg + kContext, hop, kHop * sizeof(float));
      for (int i = 0; i < kPad; ++i) xg[n + i] = xg[n - 2 - i];
    }
    memcpy(context_, hops + (done + G) * kHop - kContext, sizeof(context_));
    silero_group(*w_, x, G, state_, probs + done, work);
    done += G;
  }
}

float SileroVad::predict(const float* hop) {
  float p = 0.f;
  predict_many(hop, 1, &p);
  return p;
}

}  // namespace msh_host
