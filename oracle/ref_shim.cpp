// TEST INFRASTRUCTURE ONLY -- never linked into libmoonshine.so, never on the product path.
//
// A C-ABI shim over the REFERENCE's own host sources for the byte / integer / sample rows of the transcription path,
// so that tests can diff `libmoonshine.so`'s msh_host_* helpers (the code the MI355X Transcriber runs) against the
// reference implementation byte for byte.  The reference sources are compiled where they lie under /root/reference/core
// (nothing is copied into this repository) by oracle/build_ref.py into oracle/_ref/libmoonshine_ref_host.so:
//   resampler.cpp, word-alignment.cpp, context-biaser.cpp, context-extractor.cpp, bin-tokenizer/bin-tokenizer.cpp,
//   moonshine-utils/{debug-utils,string-utils,file-utils}.cpp
// Every ref_host_* function below has the argument list of the msh_host_* function of include/moonshine_hip.h it checks.
// The reference's arithmetic hot path (ONNX Runtime + .ort graphs) is NOT buildable here; see DESIGN.md section 4.
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
#include "word-alignment.h"                // core/word-alignment.h

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

}  // extern "C"
