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

// Conv1d(k = 3, padding = 1, stride s) + ReLU over [C_in][T_in] -> [C_out][T_out], T_out = (T_in - 1) / s + 1
int conv_relu(const float* x, int cin, int tin, const float* w, const float* b, int cout, int stride, float* y) {
  const int tout = (tin - 1) / stride + 1;
  for (int o = 0; o < cout; ++o) {
    const float* wo = w + (size_t)o * cin * 3;
    for (int t = 0; t < tout; ++t) {
      const int c0 = t * stride - 1;
      float acc = b[o];
      for (int c = 0; c < cin; ++c) {
        const float* xc = x + (size_t)c * tin;
        const float* wc = wo + c * 3;
        for (int k = 0; k < 3; ++k) {
          const int p = c0 + k;
          if (p >= 0 && p < tin) acc += wc[k] * xc[p];
        }
      }
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

float SileroVad::predict(const float* hop) {
  // input = context (64) + hop (512), reflect-padded by 64 on the right: padded[576 + i] = input[574 - i]
  float x[kContext + kHop + kPad];
  memcpy(x, context_, sizeof(context_));
  memcpy(x + kContext, hop, kHop * sizeof(float));
  constexpr int n = kContext + kHop;
  for (int i = 0; i < kPad; ++i) x[n + i] = x[n - 2 - i];
  memcpy(context_, x + n - kContext, sizeof(context_));  // last 64 samples of the un-padded input (silero-vad.cpp:163-164)

  // |STFT|: conv1d with the basis, stride 128 -> [258][4]; magnitude over (real, imag) -> [129][4]
  float mag[kBins * kFrames];
  const float* basis = w_->stft.data();
  for (int b = 0; b < kBins; ++b) {
    const float* br = basis + (size_t)b * kFft;
    const float* bi = basis + (size_t)(b + kBins) * kFft;
    for (int t = 0; t < kFrames; ++t) {
      const float* xs = x + t * kStftHop;
      float re = 0.f, im = 0.f;
      for (int k = 0; k < kFft; ++k) {
        re += br[k] * xs[k];
        im += bi[k] * xs[k];
      }
      mag[b * kFrames + t] = sqrtf(re * re + im * im);
    }
  }
  float a[128 * kFrames], c[128 * kFrames];
  int t = kFrames;
  const float* in = mag;
  float* bufs[2] = {a, c};
  for (int i = 0; i < 4; ++i) {
    float* out = bufs[i & 1];
    t = conv_relu(in, kConvIn[i], t, w_->conv_w[i].data(), w_->conv_b[i].data(), kConvOut[i], kConvStride[i], out);
    in = out;
  }
  // t == 1: in = [128] features.  LSTM cell, gates i, f, g, o
  float* h = state_;
  float* cs = state_ + kState;
  float gates[4 * kState];
  for (int g = 0; g < 4 * kState; ++g) {
    const float* wi = w_->w_ih.data() + (size_t)g * kState;
    const float* wh = w_->w_hh.data() + (size_t)g * kState;
    float acc = w_->b_ih[g] + w_->b_hh[g];
    for (int k = 0; k < kState; ++k) acc += wi[k] * in[k] + wh[k] * h[k];
    gates[g] = acc;
  }
  float logit = w_->out_b;
  for (int k = 0; k < kState; ++k) {
    const float ig = sigmoidf(gates[k]), fg = sigmoidf(gates[kState + k]), gg = tanhf(gates[2 * kState + k]),
                og = sigmoidf(gates[3 * kState + k]);
    const float cn = fg * cs[k] + ig * gg;
    const float hn = og * tanhf(cn);
    cs[k] = cn;
    gates[k] = hn;  // h is read by every gate row above; commit after the loop
    logit += w_->out_w[k] * (hn > 0.f ? hn : 0.f);
  }
  memcpy(h, gates, kState * sizeof(float));
  return sigmoidf(logit);
}

}  // namespace msh_host
