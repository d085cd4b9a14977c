// Host-side pre/post steps of the transcription path: tokenizer.bin reader (ids -> text), UTF-8 repair,
// and the hop-based voice-activity segmenter.  Byte-exact with the reference for the paths it keeps.
#pragma once

#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "silero_vad.h"

namespace msh_host {

// tokenizer.bin = concatenated length-prefixed byte strings, one per id
// (reference core/bin-tokenizer/bin-tokenizer.cpp:46-66).
class BinTokenizer {
 public:
  BinTokenizer(const uint8_t* data, size_t size, const std::string& space_marker = "\xE2\x96\x81");
  static BinTokenizer* from_file(const std::string& path);
  size_t vocab_size() const { return tokens_.size(); }
  const std::string& token_bytes(size_t id) const { return tokens_[id]; }
  // ids -> text: concatenate, skip `<...>` specials, marker -> ' ', trim spaces/tabs
  // (reference core/bin-tokenizer/bin-tokenizer.cpp:406-426).  Throws on an id with no bytes.
  std::string tokens_to_text(const int32_t* ids, size_t count, bool skip_specials = true) const;

  // text -> ids (reference core/bin-tokenizer/bin-tokenizer.cpp:277-402).  bpe = true replays the merges with a
  // piece's id as its rank and falls back to raw bytes for what no merge spells; it needs the vocabulary's block of
  // 256 single-byte entries and degrades to longest-match (which throws on an unspellable byte) without one.
  std::vector<int32_t> text_to_tokens(const std::string& text, bool bpe) const;
  bool has_byte_fallback() const { return byte_base_ >= 0; }

 private:
  std::vector<int32_t> encode_longest_match(const std::string& text) const;
  std::vector<int32_t> encode_bpe(const std::string& text) const;
  void build_indexes();
  std::vector<std::string> tokens_;
  std::string space_;
  std::vector<std::vector<int32_t>> by_first_byte_;
  std::unordered_map<std::string, int32_t> merge_ids_;
  int32_t byte_base_ = -1;
};

// Replace every byte that does not start a structurally valid UTF-8 sequence by '?'
// (reference core/transcriber.cpp:1489-1543).
std::string sanitize_utf8(const std::string& text);

struct VadSegment {
  std::vector<float> audio;  // 16 kHz samples of the segment so far
  float start_time = 0.f, end_time = 0.f;
  bool is_complete = false, just_updated = false;
  // first sample of the segment in the detector's 16 kHz input, counted from start(): the audio is always the verbatim
  // slice [src_offset, src_offset + audio.size()) of it (the max-length fade scales the probability, never the samples),
  // which lets a batch call hand the engine a slice of the PCM the device VAD already uploaded
  size_t src_offset = 0;
};

// Segmenter with the reference's state machine (reference core/voice-activity-detector.cpp:69-199):
// whole hops only, look-behind on voice start, max-segment fade.  threshold == 0 means "always voice" (what the
// reference's own evaluation / benchmark scripts use, scripts/eval-librispeech.py:381-388); threshold > 0 runs
// Silero VAD on every hop (silero_vad.h) and smooths its probability over the last `window_size` hops (:139-151).
//
// Two deliberate differences from the reference:
//  * every detector owns its Silero context / LSTM state, reset by start(); the reference shares ONE global
//    SileroVad between all detectors of the process and never resets it (voice-activity-detector.cpp:22,35-37), so
//    its probabilities depend on whatever audio any stream saw before.  The first stream of a process is identical.
//  * `hard_cap` (samples, 0 = none): a segment that would grow beyond it is closed and the next hop opens a new one
//    without look-behind.  The reference has no such cap (with threshold 0 its fade factor stays positive, so a
//    segment never ends, Appendix A.2 of SURVEY.md); the engine behind this build has a finite position table and
//    token budget, and the Transcriber passes that capacity here instead of failing on long audio.
class VoiceActivityDetector {
 public:
  VoiceActivityDetector(float threshold, int32_t window_size, int32_t hop_size, size_t look_behind, size_t max_segment,
                        std::shared_ptr<const SileroWeights> silero = nullptr, size_t hard_cap = 0);
  void start();
  void stop();
  bool is_active() const { return active_; }
  // silero_probs (optional): the Silero probability of every whole hop of THIS call, computed elsewhere (the device network
  // of batch calls, silero_device.h); only for 16 kHz audio handed over on a hop boundary (nothing waiting from an earlier
  // call), n_probs = count / hop.  The detector's own network state is then not advanced.
  void process_audio(const float* audio, size_t count, int32_t sample_rate, const float* silero_probs = nullptr,
                     size_t n_probs = 0);
  const std::vector<VadSegment>& segments() const { return segments_; }
  std::vector<VadSegment> take_segments() { return std::move(segments_); }  // after stop(): hands the audio over, no copy
  void clear_completed_audio();

 private:
  void process_hop(const float* hop, const float* silero_prob = nullptr);   // silero_prob: precomputed for this hop
  void append_history(const float* p, size_t count);
  float threshold_;
  int32_t hop_;
  size_t look_behind_, max_segment_, hard_cap_;
  std::unique_ptr<SileroVad> silero_;
  std::vector<float> prob_window_;
  size_t prob_index_ = 0;
  bool active_ = false, prev_voice_ = false, forced_cut_ = false;
  size_t processed_ = 0;
  // the open segment's samples live in segments_.back().audio only (the reference re-copies its growing buffer into the
  // segment on every hop -- quadratic in the segment length: ~100 MB of memcpy for one 10 s clip)
  size_t open_size() const { return prev_voice_ && !segments_.empty() ? segments_.back().audio.size() : 0; }
  std::vector<float> look_buf_, remainder_, hop_probs_;   // look_buf_: the look_behind_ samples in front of region_
  const float* region_ = nullptr;   // consecutive samples of the current call, ending with the hop being processed
  size_t region_len_ = 0;
  size_t call_remaining_ = 0;  // samples of the current process_audio call not yet consumed (a reserve() hint)
  std::vector<VadSegment> segments_;
};

}  // namespace msh_host
