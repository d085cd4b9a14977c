#include "silero_vad.h"

#include <math.h>
#include <string.h>

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
// one hop through the network; x = context + hop + reflect padding (640 samples), state = [h | c]
MSH_SIMD_CLONES float silero_hop(const SileroWeights& W, const float* x, float* state) {
  // |STFT|: conv1d with the basis, stride 128 -> [258][4]; magnitude over (real, imag) -> [129][4]
  float mag[kBins * kFrames];
  const float* basis = W.stft.data();
  for (int b = 0; b < kBins; ++b) {
    const float* br = basis + (size_t)b * kFft;
    const float* bi = basis + (size_t)(b + kBins) * kFft;
    for (int t = 0; t < kFrames; ++t) {
      const float* xs = x + t * kStftHop;
      const float re = dot(br, xs, kFft), im = dot(bi, xs, kFft);
      mag[b * kFrames + t] = sqrtf(re * re + im * im);
    }
  }
  float a[128 * kFrames], c[128 * kFrames];
  int t = kFrames;
  const float* in = mag;
  float* bufs[2] = {a, c};
  for (int i = 0; i < 4; ++i) {
    float* out = bufs[i & 1];
    t = conv_relu(in, kConvIn[i], t, W.conv_w[i].data(), W.conv_b[i].data(), kConvOut[i], kConvStride[i], out);
    in = out;
  }
  // t == 1: in = [128] features.  LSTM cell, gates i, f, g, o
  constexpr int S = SileroVad::kState;
  float* h = state;
  float* cs = state + S;
  float gates[4 * S];
  for (int g = 0; g < 4 * S; ++g)
    gates[g] = (W.b_ih[g] + W.b_hh[g]) + (dot(W.w_ih.data() + (size_t)g * S, in, S) + dot(W.w_hh.data() + (size_t)g * S, h, S));
  float logit = W.out_b;
  for (int k = 0; k < S; ++k) {
    const float ig = sigmoidf(gates[k]), fg = sigmoidf(gates[S + k]), gg = tanhf(gates[2 * S + k]), og = sigmoidf(gates[3 * S + k]);
    const float cn = fg * cs[k] + ig * gg;
    const float hn = og * tanhf(cn);
    cs[k] = cn;
    gates[k] = hn;  // h is read by every gate row above; commit after the loop
    logit += W.out_w[k] * (hn > 0.f ? hn : 0.f);
  }
  memcpy(h, gates, S * sizeof(float));
  return sigmoidf(logit);
}
}  // namespace

float SileroVad::predict(const float* hop) {
  // input = context (64) + hop (512), reflect-padded by 64 on the right: padded[576 + i] = input[574 - i]
  float x[kContext + kHop + kPad];
  memcpy(x, context_, sizeof(context_));
  memcpy(x + kContext, hop, kHop * sizeof(float));
  constexpr int n = kContext + kHop;
  for (int i = 0; i < kPad; ++i) x[n + i] = x[n - 2 - i];
  memcpy(context_, x + n - kContext, sizeof(context_));  // last 64 samples of the un-padded input (silero-vad.cpp:163-164)
  return silero_hop(*w_, x, state_);
}

}  // namespace msh_host
