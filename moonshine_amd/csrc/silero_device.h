// Silero VAD on the GPU for batch calls: the probabilities of every 512-sample hop of many clips in one go.
//
// The host implementation (silero_vad.{h,cpp}) costs ~20-50 us of one core per 32 ms hop; with the reference's default
// vad_threshold a 2048-clip batch call spends 14 CPU-seconds there -- 0.8-1.4 s on the 16 CPUs of the benchmark box, against
// 0.3-0.45 s of GPU time for the transcription itself.  Here the state-independent part of the network runs as fp32 GEMMs
// over all hops of a wave of clips, and the LSTM recurrence as one persistent workgroup per clip (k_silero.hip).  Every
// clip starts from a fresh state (zero context, zero h / c), as a detector does after start().
// The VoiceActivityDetector state machine (probability ring, look-behind, fade, cuts) stays on the host and consumes the
// probabilities (process_audio_with_probs).
#pragma once

#include <memory>
#include <vector>

#include "engine.h"
#include "silero_vad.h"

namespace msh {

class SileroDevice {
 public:
  SileroDevice(int device, const msh_host::SileroWeights& w);
  ~SileroDevice();
  SileroDevice(const SileroDevice&) = delete;
  SileroDevice& operator=(const SileroDevice&) = delete;
  // pcm[i]: n[i] samples of 16 kHz audio in HOST memory.  probs[i] gets n[i] / 512 values: what SileroVad::predict returns
  // hop after hop from a fresh state (up to fp32 summation order).  Clips are processed in chunks that bound the workspace.
  void probabilities(const float* const* pcm, const uint64_t* n, size_t count, std::vector<std::vector<float>>* probs);

 private:
  void upload_weights(const msh_host::SileroWeights& w);
  void run_chunk(const float* const* pcm, const uint64_t* n, size_t c0, size_t c1, std::vector<std::vector<float>>* probs);
  int device_;
  hipStream_t stream_ = nullptr;
  std::vector<void*> weights_;
  float *basis_ = nullptr, *conv_w_[4] = {nullptr, nullptr, nullptr, nullptr}, *conv_b_[4] = {nullptr, nullptr, nullptr, nullptr};
  float *w_ih_ = nullptr, *w_hh_ = nullptr, *bias_sum_ = nullptr, *out_w_ = nullptr;
  float out_b_ = 0.f;
  int kpad_[4] = {0, 0, 0, 0};
  DevBuf audio_, hop_base_, clip_hop0_, frames_, stft_, act_[2], cols_, gin_, probs_;
  void* pinned_ = nullptr;
  size_t pinned_cap_ = 0;
};

}  // namespace msh
