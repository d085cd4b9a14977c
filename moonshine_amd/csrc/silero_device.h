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

#include <chrono>
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
  // resident (optional): the uploaded audio stays on the device -- (*resident)[i] = DEVICE pointer to the whole hops of clip i
  // (n[i] / 512 * 512 floats, the caller's samples verbatim), or nullptr once this object holds kArenaBudget bytes of kept
  // audio; valid until release_audio().  A batch call hands segments to the engine as slices of it (Engine::encode,
  // on_device) instead of sending the same PCM over PCIe a second time.
  void probabilities(const float* const* pcm, const uint64_t* n, size_t count, std::vector<std::vector<float>>* probs,
                     std::vector<const float*>* resident = nullptr);
  // The same in two halves: submit() stages one chunk of clips (at most kMaxHopsPerSubmit whole hops, unless it is a single
  // clip) and enqueues its upload and network; collect() waits for it and returns the probabilities of its clips back to
  // back (+ the device pointers, as above).  Two submissions may be outstanding: chunk k + 1 is gathered, and its upload queued
  // behind chunk k's network, while the caller consumes chunk k - 1.  Tickets are collected in order.
  static constexpr long kMaxHopsPerSubmit = 65536;
  int64_t submit(const float* const* pcm, const uint64_t* n, size_t count, bool keep_audio);
  // the same for 16-bit PCM: the clips cross PCIe at two bytes per sample and become fp32 (x / 32768) on the device
  int64_t submit_pcm16(const int16_t* const* pcm16, const uint64_t* n, size_t count, bool keep_audio);
  void collect(int64_t ticket, std::vector<float>* probs, std::vector<const float*>* resident);
  void abandon();   // after a failure: waits for the streams and forgets the outstanding tickets
  // the kept audio may be overwritten by later calls (the buffers themselves stay allocated for them)
  // (submissions never collected -- a caller that gave up half-way -- are waited for and forgotten)
  void release_audio();
  static constexpr size_t kArenaBudget = (size_t)8 << 30;
  static constexpr size_t kArenaKeep = (size_t)2 << 30;   // kept-audio buffers that survive release_audio() for the next call

 private:
  void upload_weights(const msh_host::SileroWeights& w);
  void sync_streams();
  int64_t submit_any(const float* const* pcm, const int16_t* const* pcm16, const uint64_t* n, size_t nc, bool keep_audio);
  static constexpr int kSlots = 2;
  struct Slot {   // one submission in flight
    bool busy = false, kept = false;
    int64_t ticket = -1;
    size_t nc = 0, bytes = 0;
    long hops = 0;
    std::vector<long> clip_off, clip_hop0;
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    float* probs_host = nullptr;   // inside `pinned`
    DevBuf audio, audio16, hop_base, clip_hop0_d;   // audio16: the 16-bit upload of submit_pcm16, converted into audio / the arena
    DevBuf* abuf = nullptr;
    hipEvent_t done = nullptr;
    double gather_ms = 0.0;
    std::chrono::steady_clock::time_point t_enqueued;
  };
  Slot slots_[kSlots];
  int64_t next_ticket_ = 0, next_collect_ = 0;
  int device_;
  hipStream_t stream_ = nullptr;
  std::vector<void*> weights_;
  float *basis_ = nullptr, *conv_w_[4] = {nullptr, nullptr, nullptr, nullptr}, *conv_b_[4] = {nullptr, nullptr, nullptr, nullptr};
  float *w_ih_ = nullptr, *w_hh_ = nullptr, *bias_sum_ = nullptr, *out_w_ = nullptr;
  float out_b_ = 0.f;
  int kpad_[4] = {0, 0, 0, 0};
  DevBuf frames_, stft_, act_[2], cols_, gin_, probs_;   // the network's workspaces, shared by the slots
  std::vector<std::unique_ptr<DevBuf>> arena_;   // audio of chunks whose caller asked for residency, one buffer per chunk
  size_t arena_used_ = 0, arena_live_bytes_ = 0;
};

}  // namespace msh
