// TEST INFRASTRUCTURE ONLY -- never linked into libmoonshine.so, never on the product path.
//
// A C-ABI shim over the REFERENCE's own host sources for the byte / integer / sample rows of the transcription path,
// so that tests can diff `libmoonshine.so`'s msh_host_* helpers (the code the MI355X Transcriber runs) against the
// reference implementation byte for byte.  The reference sources are compiled where they lie under /root/reference/core
// (nothing is copied into this repository) by oracle/build_ref.py into oracle/_ref/libmoonshine_ref_host.so:
//   resampler.cpp, word-alignment.cpp, context-biaser.cpp, context-extractor.cpp, bin-tokenizer/bin-tokenizer.cpp,
//   moonshine-utils/{debug-utils,string-utils,file-utils}.cpp, voice-activity-detector.cpp
// Two rows need a seam:
//   * VoiceActivityDetector (core/voice-activity-detector.cpp:69-199) calls SileroVad::predict, whose implementation
//     (core/silero-vad.cpp) is the ONNX Runtime driver.  The detector is compiled as it is; the SileroVad member functions
//     are DEFINED HERE as a stub that hands out precomputed probabilities, one per hop -- the counterpart of
//     msh_host_vad_segments_from_probs (the device-VAD path of batch calls feeds the detector the same way).
//   * Transcriber::sanitize_text (core/transcriber.cpp:1489-1543) sits in a translation unit that needs the whole ORT
//     model stack to link.  oracle/build_ref.py lifts that one function's text out of the reference file at build time
//     into the git-ignored oracle/_ref/sanitize_text_extracted.inc (never committed, never edited), included below.
// Every ref_host_* function below has the argument list of the msh_host_* function of include/moonshine_hip.h it checks.
// The reference's arithmetic hot path (ONNX Runtime + .ort graphs) is NOT buildable here; see DESIGN.md section 4.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bin-tokenizer/bin-tokenizer.h"   // core/bin-tokenizer/bin-tokenizer.h
#include "context-biaser.h"                // core/context-biaser.h
#include "context-extractor.h"             // core/context-extractor.h
#include "debug-utils.h"                   // core/moonshine-utils/debug-utils.h (load_wav_data / save_wav_data)
#include "resampler.h"                     // core/resampler.h
#include "voice-activity-detector.h"       // core/voice-activity-detector.h (pulls in core/silero-vad.h + the ORT C header)
#include "word-alignment.h"                // core/word-alignment.h

#if __has_include("_ref/sanitize_text_extracted.inc")
#define REF_HAS_SANITIZE 1
namespace refx {
struct Transcriber {   // the extracted definition is a static member function of this name
  static std::string* sanitize_text(const char* text);
};
#include "_ref/sanitize_text_extracted.inc"
}  // namespace refx
#endif

// ---- SileroVad stub: precomputed probabilities, one per predict() call, in order ----
namespace {
std::vector<float> g_probs;
size_t g_prob_next = 0;
}  // namespace
SileroVad::SileroVad(int, int, float, int, int, int, float) {}
SileroVad::~SileroVad() {}
void SileroVad::predict(const std::vector<float>&, float* out_probability, int* out_flag) {
  const float p = g_prob_next < g_probs.size() ? g_probs[g_prob_next] : 0.0f;
  ++g_prob_next;
  if (out_probability) *out_probability = p;
  if (out_flag) *out_flag = p > 0.5f;
}

namespace {
int64_t copy_out(const std::string& r, char* out, uint64_t cap) {
  if (out != nullptr && cap > 0) {
    const size_t n = r.size() < cap - 1 ? r.size() : (size_t)cap - 1;
    memcpy(out, r.data(), n);
    out[n] = 0;
  }
  return (int64_t)r.size();
}
}  // namespace

extern "C" {

// BinTokenizer::tokens_to_text<int32_t> (bin-tokenizer.cpp:406-426)
int64_t ref_host_tokens_to_text(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const int32_t* ids, uint64_t n_ids,
                                char* out, uint64_t out_cap) {
  try {
    BinTokenizer tok(tokenizer_bin, (size_t)tokenizer_size);
    return copy_out(tok.tokens_to_text<int32_t>(std::vector<int32_t>(ids, ids + n_ids)), out, out_cap);
  } catch (const std::exception&) {
    return -3;
  }
}

// BinTokenizer::text_to_tokens<int32_t> (bin-tokenizer.cpp:277-402)
int64_t ref_host_text_to_tokens(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const char* text, uint64_t text_len,
                                const char* space_marker, int32_t bpe, int32_t* out, uint64_t out_cap) {
  try {
    BinTokenizer tok(tokenizer_bin, (size_t)tokenizer_size, space_marker ? space_marker : "\xE2\x96\x81",
                     bpe ? BinTokenizerEncoding::kBpe : BinTokenizerEncoding::kLongestMatch);
    const std::vector<int32_t> ids = tok.text_to_tokens<int32_t>(std::string(text ? text : "", (size_t)text_len));
    for (size_t i = 0; i < ids.size() && i < out_cap; ++i) out[i] = ids[i];
    return (int64_t)ids.size();
  } catch (const std::exception&) {
    return -3;
  }
}

// ContextBiaser::add_token_sequence / advance / apply (context-biaser.cpp:44-149)
int64_t ref_host_biaser_bonuses(const int32_t* flat_tokens, const int32_t* seq_lens, uint64_t n_seqs, float boost,
                                const int32_t* prefix, uint64_t n_prefix, float* out, uint64_t vocab) {
  try {
    ContextBiaser b;
    b.set_boost(boost);
    size_t off = 0;
    for (uint64_t i = 0; i < n_seqs; ++i) {
      b.add_token_sequence(std::vector<int32_t>(flat_tokens + off, flat_tokens + off + seq_lens[i]));
      off += seq_lens[i];
    }
    b.reset();
    for (uint64_t i = 0; i < n_prefix; ++i) b.advance(prefix[i]);
    b.apply(out, (int)vocab);
    return (int64_t)b.sequence_count_for_test();
  } catch (const std::exception&) {
    return -3;
  }
}

// ContextExtractor::extract (context-extractor.cpp:152-226) with the subword count the Transcriber injects
// (transcriber.cpp:233-243: the streaming model's BPE tokenizer, 0 for a word it cannot spell)
int64_t ref_host_context_terms(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const char* context,
                               uint64_t context_len, int32_t max_terms, char* out, uint64_t out_cap) {
  try {
    BinTokenizer tok(tokenizer_bin, (size_t)tokenizer_size, "\xE2\x96\x81", BinTokenizerEncoding::kBpe);
    const std::vector<std::string> terms = ContextExtractor::extract(
        std::string(context ? context : "", (size_t)context_len), max_terms, [&](const std::string& w) -> size_t {
          try {
            return tok.text_to_tokens<int32_t>(w).size();
          } catch (const std::exception&) {
            return 0;
          }
        });
    std::string joined;
    for (size_t i = 0; i < terms.size(); ++i) joined += (i ? "\n" : "") + terms[i];
    return copy_out(joined, out, out_cap);
  } catch (const std::exception&) {
    return -3;
  }
}

// dtw (word-alignment.cpp:12-88)
int64_t ref_host_dtw(const float* cost, int32_t n_text, int32_t n_time, int32_t* text_idx, int32_t* time_idx, uint64_t cap) {
  std::vector<int> a, b;
  dtw(std::vector<float>(cost, cost + (size_t)n_text * n_time), n_text, n_time, a, b);
  for (size_t i = 0; i < a.size() && i < cap; ++i) {
    if (text_idx) text_idx[i] = a[i];
    if (time_idx) time_idx[i] = b[i];
  }
  return (int64_t)a.size();
}

// median_filter (word-alignment.cpp:98-153): [rows][row_len] is its [channels = rows][height = 1][width = row_len]
int32_t ref_host_median_filter(float* data, uint64_t rows, int32_t row_len, int32_t width) {
  std::vector<float> v(data, data + rows * (size_t)row_len);
  median_filter(v, (int)rows, 1, row_len, width);
  memcpy(data, v.data(), v.size() * sizeof(float));
  return 0;
}

// align_words (word-alignment.cpp:181-394).  The reference takes layers and heads separately and only ever uses
// their product: heads_total is passed as (1 layer x heads_total heads).
int64_t ref_host_align_words(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const float* att, int32_t heads_total,
                             int32_t n_steps, int32_t frames, const int32_t* tokens, uint64_t n_tokens,
                             float seconds_per_frame, char* text_out, uint64_t text_cap, float* times_out, uint64_t max_words) {
  try {
    BinTokenizer tok(tokenizer_bin, (size_t)tokenizer_size);
    const std::vector<TranscriberWord> words = align_words(att, 1, heads_total, n_steps, frames,
                                                           std::vector<int>(tokens, tokens + n_tokens), seconds_per_frame, &tok);
    std::string joined;
    for (size_t i = 0; i < words.size(); ++i) {
      joined += (i ? "\n" : "") + words[i].text;
      if (times_out != nullptr && i < max_words) {
        times_out[3 * i] = words[i].start;
        times_out[3 * i + 1] = words[i].end;
        times_out[3 * i + 2] = words[i].confidence;
      }
    }
    copy_out(joined, text_out, text_cap);
    return (int64_t)words.size();
  } catch (const std::exception&) {
    return -3;
  }
}

// load_wav_data / save_wav_data (moonshine-utils/debug-utils.cpp:52-250)
int64_t ref_host_load_wav(const char* path, float* out, uint64_t out_cap, int32_t* sample_rate) {
  float* data = nullptr;
  size_t n = 0;
  int32_t rate = 0;
  if (!load_wav_data(path, &data, &n, &rate)) return -3;
  if (sample_rate) *sample_rate = rate;
  if (out != nullptr) memcpy(out, data, sizeof(float) * (n < out_cap ? n : (size_t)out_cap));
  free(data);
  return (int64_t)n;
}
int32_t ref_host_save_wav(const char* path, const float* samples, uint64_t count, int32_t sample_rate) {
  return save_wav_data(path, samples, (size_t)count, (uint32_t)sample_rate) ? 0 : -3;
}

// resample_audio (resampler.cpp:5-86)
int64_t ref_host_resample(const float* in, uint64_t n, float in_rate, float out_rate, float* out, uint64_t out_cap) {
  const std::vector<float> r = resample_audio(std::vector<float>(in, in + n), in_rate, out_rate);
  if (out != nullptr) memcpy(out, r.data(), sizeof(float) * (r.size() < out_cap ? r.size() : (size_t)out_cap));
  return (int64_t)r.size();
}

// VoiceActivityDetector start / process_audio (in `chunk`-sample calls, chunk = 0: one call) / stop, with the Silero
// probabilities of the hops supplied (ignored by the detector when threshold == 0).  out[4i..4i+3] = (round(start_time *
// 16000), audio sample count, is_complete, FNV-1a of the segment's audio bytes); returns the segment count.
int64_t ref_host_vad_segments(float threshold, int32_t window, int32_t hop, uint64_t look_behind, uint64_t max_segment,
                              const float* audio, uint64_t n_samples, int32_t sample_rate, uint64_t chunk, const float* probs,
                              uint64_t n_probs, int64_t* out, uint64_t max_segments) {
  g_probs.assign(probs, probs + (probs ? n_probs : 0));
  g_prob_next = 0;
  VoiceActivityDetector vad(threshold, window, hop, (size_t)look_behind, (size_t)max_segment);
  vad.start();
  if (chunk == 0) chunk = n_samples ? n_samples : 1;
  for (uint64_t i = 0; i < n_samples; i += chunk) vad.process_audio(audio + i, (size_t)((n_samples - i) < chunk ? (n_samples - i) : chunk), sample_rate);
  vad.stop();
  const std::vector<VoiceActivitySegment>* segs = vad.get_segments();
  uint64_t k = 0;
  for (const VoiceActivitySegment& s : *segs) {
    if (k < max_segments) {
      uint64_t h = 1469598103934665603ull;
      const uint8_t* b = reinterpret_cast<const uint8_t*>(s.audio_data.data());
      for (size_t i = 0; i < s.audio_data.size() * sizeof(float); ++i) h = (h ^ b[i]) * 1099511628211ull;
      out[4 * k] = (int64_t)llround((double)s.start_time * 16000.0);
      out[4 * k + 1] = (int64_t)s.audio_data.size();
      out[4 * k + 2] = s.is_complete ? 1 : 0;
      out[4 * k + 3] = (int64_t)h;
    }
    ++k;
  }
  return (int64_t)k;
}

// Transcriber::sanitize_text (core/transcriber.cpp:1489-1543), same return rule as msh_host_sanitize_utf8
int64_t ref_host_sanitize_utf8(const char* text, uint64_t n, char* out, uint64_t out_cap) {
#ifdef REF_HAS_SANITIZE
  const std::string in(text, (size_t)n);   // the reference takes a C string: the tests keep NUL out of the input
  std::string* r = refx::Transcriber::sanitize_text(in.c_str());
  const int64_t len = copy_out(*r, out, out_cap);
  delete r;
  return len;
#else
  (void)text; (void)n; (void)out; (void)out_cap;
  return -1;
#endif
}

}  // extern "C"
