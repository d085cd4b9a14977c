// The exported moonshine_* C API (include/moonshine-c-api.h): handle map, option parsing and the
// exception barrier, following the conventions of reference core/moonshine-c-api.cpp:74-80 (handle check),
// :87-97 (option names lower-cased), :129-198 (recognised options; unknown names fail the load),
// :200-216 (int32 handles into a process-global map), :439-446 (exceptions -> MOONSHINE_ERROR_UNKNOWN).
#include "../../include/moonshine-c-api.h"
#include "../../include/moonshine_hip.h"

#include <inttypes.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_utils.h"
#include "context_biaser.h"
#include "context_extractor.h"
#include "transcriber.h"
#include "word_alignment.h"

using namespace msh_host;

namespace {

typedef std::vector<std::pair<std::string, std::string>> OptionList;
bool g_log_api_calls = false;

OptionList read_options(const moonshine_option_t* options, uint64_t count) {
  OptionList out;
  for (uint64_t i = 0; i < count; ++i) {
    if (options[i].name == nullptr || options[i].value == nullptr) throw std::runtime_error("option with null name or value");
    out.emplace_back(to_lower(options[i].name), options[i].value);
  }
  return out;
}

// Options the reference accepts for features this build does not have.  They are accepted when they
// ask for the default (feature off) and refused otherwise, so a caller never silently loses a feature.
void require_off(const std::string& name, const std::string& value) {
  if (parse_bool(value))
    throw std::runtime_error("option '" + name + "' needs a component that is not part of the MI355X build");
}

// comma-separated list, each term trimmed, empties dropped (reference core/moonshine-c-api.cpp:114-127)
std::vector<std::string> parse_keyterms(const std::string& v) {
  std::vector<std::string> out;
  size_t pos = 0;
  while (pos <= v.size()) {
    const size_t c = v.find(',', pos);
    const std::string term = trim(v.substr(pos, c == std::string::npos ? std::string::npos : c - pos));
    if (!term.empty()) out.push_back(term);
    if (c == std::string::npos) break;
    pos = c + 1;
  }
  return out;
}

void apply_options(const OptionList& options, TranscriberOptions* o) {
  for (const auto& kv : options) {
    const std::string& k = kv.first;
    const std::string& v = kv.second;
    if (k == "log_api_calls") g_log_api_calls = parse_bool(v);
    else if (k == "skip_transcription") o->model_source = TranscriberOptions::NONE;
    else if (k == "transcription_interval") o->transcription_interval = parse_float(v);
    else if (k == "vad_threshold") o->vad_threshold = parse_float(v);
    else if (k == "vad_window_duration") o->vad_window_duration = parse_float(v);
    else if (k == "vad_hop_size") o->vad_hop_size = parse_int32(v);
    else if (k == "vad_look_behind_sample_count") o->vad_look_behind_sample_count = parse_size(v);
    else if (k == "vad_max_segment_duration") o->vad_max_segment_duration = parse_float(v);
    else if (k == "vad_model_path") o->vad_model_path = v;                          // additive: Silero weights (safetensors)
    else if (k == "max_tokens_per_second") o->max_tokens_per_second = parse_float(v);
    else if (k == "decode_incomplete_lines") o->decode_incomplete_lines = parse_bool(v);
    else if (k == "return_audio_data") o->return_audio_data = parse_bool(v);
    else if (k == "log_output_text") o->log_output_text = parse_bool(v);
    else if (k == "log_ort_run") o->log_ort_run = parse_bool(v);
    else if (k == "save_input_wav_path") o->save_input_wav_path = v;
    else if (k == "device") o->device = parse_int32(v);
    else if (k == "use_speculative_decoding") o->use_speculative_decoding = parse_bool(v);
    else if (k == "max_streams") o->max_streams = parse_int32(v);                  // additive (streaming archs)
    else if (k == "vad_device") o->vad_device = parse_int32(v);                    // additive: Silero on the GPU for batch calls
    else if (k == "host_threads") o->host_threads = parse_int32(v);                // additive: VAD threads of batch calls
    else if (k == "max_stream_seconds") o->max_stream_seconds = parse_float(v);     // additive (streaming archs)
    else if (k == "kv_dtype") {   // additive: cross K / V on the device as bf16 (default) or fp8 e4m3
      std::string t = v;
      for (char& ch : t) ch = (char)tolower((unsigned char)ch);
      if (t == "bf16") o->kv_dtype = 0;
      else if (t == "fp8" || t == "fp8_e4m3" || t == "e4m3") o->kv_dtype = 1;
      else throw std::runtime_error("kv_dtype must be bf16 or fp8, got '" + v + "'");
    }
    else if (k == "cross_attention") {   // additive: auto (default) | kv | absorbed (msh_set_cross_mode)
      std::string t = v;
      for (char& ch : t) ch = (char)tolower((unsigned char)ch);
      if (t == "auto") o->cross_attention = 0;
      else if (t == "kv" || t == "projected") o->cross_attention = 1;
      else if (t == "absorbed") o->cross_attention = 2;
      else throw std::runtime_error("cross_attention must be auto, kv or absorbed, got '" + v + "'");
    }
    else if (k == "kernel_set") {   // additive: auto (default) | per_call | uniform (msh_set_uniform_kernels)
      std::string t = v;
      for (char& ch : t) ch = (char)tolower((unsigned char)ch);
      if (t == "auto") o->kernel_set = 0;
      else if (t == "per_call" || t == "latency") o->kernel_set = 1;
      else if (t == "uniform" || t == "throughput") o->kernel_set = 2;
      else throw std::runtime_error("kernel_set must be auto, per_call or uniform, got '" + v + "'");
    }
    else if (k == "batch_clips" || k == "max_batch_size") {   // additive (batch calls; SURVEY 8b names it max_batch_size)
      o->batch_clips = parse_int32(v);
      o->batch_clips_given = true;   // asking for large sub-batches is what switches cross_attention=auto to the absorbed form
    }
    else if (k == "num_gpus") o->num_gpus = parse_int32(v);                         // additive: shard batch calls over GPUs device .. device+n-1 (-1 = all)
    else if (k == "devices") {                                                      // additive: explicit GPU list, e.g. "0,1,2,3"
      o->device_ids.clear();
      for (const std::string& t : parse_keyterms(v)) o->device_ids.push_back(parse_int32(t));
    }
    else if (k == "batches_in_flight") o->batches_in_flight = parse_int32(v);       // additive (batch calls)
    else if (k == "hw_queues") {                                                    // additive: see msh_set_hw_queues
      if (msh_set_hw_queues(parse_int32(v)) != MSH_OK) throw std::runtime_error("option 'hw_queues' must be 1..64");
    }
    else if (k == "word_timestamps") o->word_timestamps = parse_bool(v);
    else if (k == "identify_speakers") require_off(k, v);
    else if (k == "keyterms") {
      o->keyterms = parse_keyterms(v);
    } else if (k == "keyterm_boost") {
      o->keyterm_boost = parse_float(v);
    } else if (k == "context") {
      o->context = v;
    } else if (k == "context_max_terms") {
      o->context_max_terms = parse_int32(v);
    } else if ( k == "diarization_cluster_cadence" ||
               k == "diarization_analyze_cadence" || k == "diarization_cluster_window_sec" ||
               k == "diarization_model_dir" || k == "coreml_cache_dir") {
      // tuning knobs of features that are off: nothing to do
    } else if (k == "ort_providers" || k == "ort_provider") {
      // there is no ONNX Runtime here; the only execution target is the MI355X
    } else if (k == "spelling_model_path") {
      if (!v.empty()) throw std::runtime_error("option 'spelling_model_path' needs the spelling model, not part of the MI355X build");
    } else {
      throw std::runtime_error("Unknown transcriber option: '" + k + "', value=" + v);
    }
  }
}

std::mutex g_map_mutex;
// Handles own their transcriber through a shared_ptr: a call in flight keeps it alive, so moonshine_free_transcriber
// racing another thread's call (the reference documents the API as thread-safe, moonshine-c-api.h:64-67) ends with the
// object destroyed by whoever finishes last -- never under a running call.
std::map<int32_t, std::shared_ptr<Transcriber>> g_transcribers;
int32_t g_next_handle = 0;

int32_t register_transcriber(Transcriber* t) {
  std::lock_guard<std::mutex> lock(g_map_mutex);
  const int32_t h = g_next_handle++;
  g_transcribers[h].reset(t);
  return h;
}

std::shared_ptr<Transcriber> lookup(int32_t handle) {
  std::lock_guard<std::mutex> lock(g_map_mutex);
  auto it = g_transcribers.find(handle);
  return (handle < 0 || it == g_transcribers.end()) ? nullptr : it->second;
}

template <class F>
int32_t with_transcriber(int32_t handle, const char* what, F&& f) {
  const std::shared_ptr<Transcriber> keep = lookup(handle);  // alive until this call returns
  Transcriber* t = keep.get();
  if (t == nullptr) {
    MSH_LOGF("Moonshine transcriber handle is invalid: handle %d", handle);
    return MOONSHINE_ERROR_INVALID_HANDLE;
  }
  try {
    return f(t);
  } catch (const std::exception& e) {
    MSH_LOGF("Failed to %s: %s", what, e.what());
    return MOONSHINE_ERROR_UNKNOWN;
  }
}

int32_t load_common(TranscriberOptions& o, const moonshine_option_t* options, uint64_t options_count) {
  try {
    if (options_count > 0 && options == nullptr) return MOONSHINE_ERROR_INVALID_ARGUMENT;
    apply_options(read_options(options, options_count), &o);
    return register_transcriber(new Transcriber(o));
  } catch (const std::exception& e) {
    MSH_LOGF("Failed to load transcriber: %s", e.what());
    return MOONSHINE_ERROR_UNKNOWN;
  }
}

}  // namespace

extern "C" {

int32_t moonshine_get_version(void) { return MOONSHINE_HEADER_VERSION; }

const char* moonshine_error_to_string(int32_t error) {
  switch (error) {
    case MOONSHINE_ERROR_NONE: return "Success";
    case MOONSHINE_ERROR_INVALID_HANDLE: return "Invalid handle";
    case MOONSHINE_ERROR_INVALID_ARGUMENT: return "Invalid argument";
    default: return "Unknown error";
  }
}

void moonshine_free_buffer(void* ptr) { free(ptr); }

const char* moonshine_transcript_to_string(const struct transcript_t* transcript) {
  static std::string description;  // static buffer, like the reference (core/moonshine-c-api.cpp:560-562)
  description = transcript == nullptr ? std::string("<null transcript>") : Transcriber::transcript_to_string(transcript);
  return description.c_str();
}

int32_t moonshine_transcriber_set_keyterms(int32_t handle, const char* keyterms) {
  return with_transcriber(handle, "set keyterms", [&](Transcriber* t) -> int32_t {
    t->set_keyterms(keyterms == nullptr ? std::vector<std::string>() : parse_keyterms(keyterms));
    return MOONSHINE_ERROR_NONE;
  });
}

int32_t moonshine_transcriber_set_context(int32_t handle, const char* context, int32_t max_terms) {
  return with_transcriber(handle, "set context", [&](Transcriber* t) -> int32_t {
    t->set_context(context == nullptr ? std::string() : std::string(context), max_terms);
    return MOONSHINE_ERROR_NONE;
  });
}

int32_t moonshine_load_transcriber_from_files(const char* path, uint32_t model_arch, const moonshine_option_t* options,
                                              uint64_t options_count, int32_t moonshine_version) {
  if (g_log_api_calls)
    MSH_LOGF("moonshine_load_transcriber_from_files(path=%s, model_arch=%u, options_count=%" PRIu64 ", version=%d)",
             path ? path : "(null)", model_arch, options_count, moonshine_version);
  TranscriberOptions o;
  o.model_source = TranscriberOptions::FILES;
  o.model_path = path ? path : "";
  o.model_arch = model_arch;
  return load_common(o, options, options_count);
}

int32_t moonshine_load_transcriber_from_memory(const uint8_t*, size_t, const uint8_t*, size_t, const uint8_t*, size_t,
                                               const uint8_t*, size_t, uint32_t, const moonshine_option_t*, uint64_t,
                                               int32_t moonshine_version) {
  // The fixed encoder/decoder/tokenizer triple describes ORT graphs.  Callers built against >= 3.0.0 get the
  // reference's answer (core/moonshine-c-api.cpp:313-320); older ones are told the same thing.
  MSH_LOGF("moonshine_load_transcriber_from_memory() is not supported (caller version %d): use "
           "moonshine_load_transcriber_from_memory_files() with model.safetensors + tokenizer.bin",
           moonshine_version);
  return MOONSHINE_ERROR_INVALID_ARGUMENT;
}

int32_t moonshine_load_transcriber_from_memory_files(const char** filenames, const uint8_t** memory,
                                                     const uint64_t* memory_sizes, uint64_t file_count,
                                                     uint32_t model_arch, const moonshine_option_t* options,
                                                     uint64_t options_count, int32_t /*moonshine_version*/) {
  if (file_count > 0 && (filenames == nullptr || memory == nullptr || memory_sizes == nullptr))
    return MOONSHINE_ERROR_INVALID_ARGUMENT;
  TranscriberOptions o;
  o.model_source = TranscriberOptions::MEMORY_FILES;
  o.model_arch = model_arch;
  for (uint64_t i = 0; i < file_count; ++i) {
    if (filenames[i] == nullptr) return MOONSHINE_ERROR_INVALID_ARGUMENT;
    const std::string key(filenames[i]);
    if (key != "model.safetensors" && key != "tokenizer.bin" && key != "streaming_config.json" && key != "silero_vad.safetensors") {
      MSH_LOGF("moonshine_load_transcriber_from_memory_files(): '%s' is not a model asset this loader recognizes. "
               "Canonical filenames: model.safetensors, tokenizer.bin, streaming_config.json, silero_vad.safetensors",
               key.c_str());
      return MOONSHINE_ERROR_INVALID_ARGUMENT;
    }
    o.memory_files[key] = std::make_pair(memory[i], (size_t)memory_sizes[i]);
  }
  return load_common(o, options, options_count);
}

void moonshine_free_transcriber(int32_t handle) {
  std::shared_ptr<Transcriber> t;
  {
    std::lock_guard<std::mutex> lock(g_map_mutex);
    auto it = g_transcribers.find(handle);
    if (it == g_transcribers.end()) return;
    t = std::move(it->second);
    g_transcribers.erase(it);
  }
  t.reset();  // destroyed here unless a call on another thread still holds it
}

int32_t moonshine_transcribe_without_streaming(int32_t handle, float* audio_data, uint64_t audio_length,
                                               int32_t sample_rate, uint32_t flags, struct transcript_t** out) {
  if (g_log_api_calls)
    MSH_LOGF("moonshine_transcribe_without_streaming(handle=%d, audio_length=%" PRIu64 ", sample_rate=%d, flags=%u)", handle,
             audio_length, sample_rate, flags);
  return with_transcriber(handle, "transcribe without streaming", [&](Transcriber* t) -> int32_t {
    if (audio_data == nullptr && audio_length > 0) return MOONSHINE_ERROR_INVALID_ARGUMENT;
    t->transcribe_without_streaming(audio_data, audio_length, sample_rate, flags, out);
    return MOONSHINE_ERROR_NONE;
  });
}

int32_t moonshine_transcribe_batch_without_streaming(int32_t handle, const float* const* audio_data,
                                                     const uint64_t* audio_lengths, uint64_t count, int32_t sample_rate,
                                                     uint32_t flags, struct transcript_t** out_transcripts) {
  return with_transcriber(handle, "transcribe batch", [&](Transcriber* t) -> int32_t {
    if (count > 0 && (audio_data == nullptr || audio_lengths == nullptr || out_transcripts == nullptr))
      return MOONSHINE_ERROR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < count; ++i)
      if (audio_data[i] == nullptr && audio_lengths[i] > 0) return MOONSHINE_ERROR_INVALID_ARGUMENT;
    t->transcribe_batch_without_streaming(audio_data, audio_lengths, count, sample_rate, flags, out_transcripts);
    return MOONSHINE_ERROR_NONE;
  });
}

int32_t moonshine_transcribe_batch_without_streaming_pcm16(int32_t handle, const int16_t* const* audio_data,
                                                           const uint64_t* audio_lengths, uint64_t count, int32_t sample_rate,
                                                           uint32_t flags, struct transcript_t** out_transcripts) {
  return with_transcriber(handle, "transcribe batch (16-bit PCM)", [&](Transcriber* t) -> int32_t {
    if (count > 0 && (audio_data == nullptr || audio_lengths == nullptr || out_transcripts == nullptr))
      return MOONSHINE_ERROR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < count; ++i)
      if (audio_data[i] == nullptr && audio_lengths[i] > 0) return MOONSHINE_ERROR_INVALID_ARGUMENT;
    t->transcribe_batch_without_streaming_pcm16(audio_data, audio_lengths, count, sample_rate, flags, out_transcripts);
    return MOONSHINE_ERROR_NONE;
  });
}

int32_t moonshine_create_stream(int32_t handle, uint32_t /*flags*/) {
  return with_transcriber(handle, "create stream", [&](Transcriber* t) -> int32_t { return t->create_stream(); });
}

int32_t moonshine_free_stream(int32_t handle, int32_t stream) {
  return with_transcriber(handle, "free stream", [&](Transcriber* t) -> int32_t {
    t->free_stream(stream);
    return MOONSHINE_ERROR_NONE;
  });
}

int32_t moonshine_start_stream(int32_t handle, int32_t stream) {
  return with_transcriber(handle, "start stream", [&](Transcriber* t) -> int32_t {
    t->start_stream(stream);
    return MOONSHINE_ERROR_NONE;
  });
}

int32_t moonshine_stop_stream(int32_t handle, int32_t stream) {
  return with_transcriber(handle, "stop stream", [&](Transcriber* t) -> int32_t {
    t->stop_stream(stream);
    return MOONSHINE_ERROR_NONE;
  });
}

int32_t moonshine_transcribe_add_audio_to_stream(int32_t handle, int32_t stream, const float* new_audio_data,
                                                 uint64_t audio_length, int32_t sample_rate, uint32_t /*flags*/) {
  return with_transcriber(handle, "add audio to stream", [&](Transcriber* t) -> int32_t {
    if (new_audio_data == nullptr && audio_length > 0) return MOONSHINE_ERROR_INVALID_ARGUMENT;
    t->add_audio_to_stream(stream, new_audio_data, audio_length, sample_rate);
    return MOONSHINE_ERROR_NONE;
  });
}

int32_t moonshine_transcribe_stream(int32_t handle, int32_t stream, uint32_t flags, struct transcript_t** out) {
  return with_transcriber(handle, "transcribe stream", [&](Transcriber* t) -> int32_t {
    t->transcribe_stream(stream, flags, out);
    return MOONSHINE_ERROR_NONE;
  });
}

// ---- host helpers (include/moonshine_hip.h) ----
static int64_t copy_out(const std::string& r, char* out, uint64_t cap) {
  if (out != nullptr && cap > 0) {
    const size_t n = r.size() < cap - 1 ? r.size() : (size_t)cap - 1;
    memcpy(out, r.data(), n);
    out[n] = 0;
  }
  return (int64_t)r.size();
}

int64_t msh_host_tokens_to_text(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const int32_t* ids, uint64_t n_ids,
                                char* out, uint64_t out_cap) {
  try {
    BinTokenizer tok(tokenizer_bin, (size_t)tokenizer_size);
    return copy_out(tok.tokens_to_text(ids, (size_t)n_ids), out, out_cap);
  } catch (const std::exception& e) {
    MSH_LOGF("tokens_to_text failed: %s", e.what());
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

int64_t msh_host_text_to_tokens(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const char* text, uint64_t text_len,
                                const char* space_marker, int32_t bpe, int32_t* out, uint64_t out_cap) {
  try {
    BinTokenizer tok(tokenizer_bin, (size_t)tokenizer_size, space_marker ? space_marker : "\xE2\x96\x81");
    const std::vector<int32_t> ids = tok.text_to_tokens(std::string(text ? text : "", (size_t)text_len), bpe != 0);
    for (size_t i = 0; i < ids.size() && i < out_cap; ++i) out[i] = ids[i];
    return (int64_t)ids.size();
  } catch (const std::exception& e) {
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

int64_t msh_host_biaser_bonuses(const int32_t* flat_tokens, const int32_t* seq_lens, uint64_t n_seqs, float boost,
                                const int32_t* prefix, uint64_t n_prefix, float* out, uint64_t vocab) {
  try {
    ContextBiaser b;
    b.set_boost(boost);
    size_t off = 0;
    for (uint64_t i = 0; i < n_seqs; ++i) {
      b.add_token_sequence(std::vector<int32_t>(flat_tokens + off, flat_tokens + off + seq_lens[i]));
      off += seq_lens[i];
    }
    for (uint64_t i = 0; i < n_prefix; ++i) b.advance(prefix[i]);
    b.apply(out, (int)vocab);
    return (int64_t)b.sequence_count();
  } catch (const std::exception& e) {
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

int64_t msh_host_context_terms(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const char* context,
                               uint64_t context_len, int32_t max_terms, char* out, uint64_t out_cap) {
  try {
    BinTokenizer tok(tokenizer_bin, (size_t)tokenizer_size);
    const std::vector<std::string> terms = ContextExtractor::extract(
        std::string(context ? context : "", (size_t)context_len), max_terms, [&](const std::string& w) -> size_t {
          try {
            return tok.text_to_tokens(w, true).size();
          } catch (const std::exception&) {
            return 0;
          }
        });
    std::string joined;
    for (size_t i = 0; i < terms.size(); ++i) joined += (i ? "\n" : "") + terms[i];
    return copy_out(joined, out, out_cap);
  } catch (const std::exception& e) {
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

int64_t msh_host_dtw(const float* cost, int32_t n_text, int32_t n_time, int32_t* text_idx, int32_t* time_idx, uint64_t cap) {
  if (cost == nullptr || n_text <= 0 || n_time <= 0) return MSH_ERR_INVALID_ARGUMENT;
  std::vector<int> a, b;
  dtw_path(cost, n_text, n_time, &a, &b);
  for (size_t i = 0; i < a.size() && i < cap; ++i) {
    if (text_idx) text_idx[i] = a[i];
    if (time_idx) time_idx[i] = b[i];
  }
  return (int64_t)a.size();
}

int32_t msh_host_median_filter(float* data, uint64_t rows, int32_t row_len, int32_t width) {
  if (data == nullptr || row_len <= 0) return MSH_ERR_INVALID_ARGUMENT;
  median_filter_rows(data, (size_t)rows, row_len, width);
  return MSH_OK;
}

int64_t msh_host_align_words(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const float* att, int32_t heads_total,
                             int32_t n_steps, int32_t frames, const int32_t* tokens, uint64_t n_tokens,
                             float seconds_per_frame, char* text_out, uint64_t text_cap, float* times_out, uint64_t max_words) {
  try {
    BinTokenizer tok(tokenizer_bin, (size_t)tokenizer_size);
    const std::vector<TranscriberWord> words = align_words(att, heads_total, n_steps, frames, std::vector<int32_t>(tokens, tokens + n_tokens),
                                                           seconds_per_frame, tok);
    std::string joined;
    for (size_t i = 0; i < words.size(); ++i) {
      joined += (i ? "\n" : "") + words[i].text;
      if (times_out != nullptr && i < max_words) {
        times_out[3 * i] = words[i].start;
        times_out[3 * i + 1] = words[i].end;
        times_out[3 * i + 2] = words[i].confidence;
      }
    }
    if (copy_out(joined, text_out, text_cap) < 0) return MSH_ERR_INVALID_ARGUMENT;
    return (int64_t)words.size();
  } catch (const std::exception& e) {
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

int64_t msh_host_load_wav(const char* path, float* out, uint64_t out_cap, int32_t* sample_rate) {
  if (path == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  std::vector<float> samples;
  int32_t rate = 0;
  if (!load_wav(path, &samples, &rate)) return MSH_ERR_INVALID_ARGUMENT;
  if (sample_rate != nullptr) *sample_rate = rate;
  if (out != nullptr) memcpy(out, samples.data(), sizeof(float) * (samples.size() < out_cap ? samples.size() : (size_t)out_cap));
  return (int64_t)samples.size();
}

int32_t msh_host_save_wav(const char* path, const float* samples, uint64_t count, int32_t sample_rate) {
  if (path == nullptr || (samples == nullptr && count > 0)) return MSH_ERR_INVALID_ARGUMENT;
  return save_wav(path, samples, (size_t)count, sample_rate) ? MSH_OK : MSH_ERR_INVALID_ARGUMENT;
}

// Silero VAD + segmenter, exactly as the Transcriber's streams run them (tests; bindings that want segments only)
int64_t msh_host_silero_probabilities(const uint8_t* weights, uint64_t weights_size, const float* audio, uint64_t n_samples,
                                      float* probs, uint64_t cap, float* state_out) {
  try {
    if (weights == nullptr || (audio == nullptr && n_samples > 0)) return MSH_ERR_INVALID_ARGUMENT;
    std::shared_ptr<SileroWeights> w(new SileroWeights());
    w->load_memory(weights, (size_t)weights_size);
    SileroVad vad(w);
    const uint64_t hops = n_samples / SileroVad::kHop;
    for (uint64_t i = 0; i < hops; ++i) {
      const float p = vad.predict(audio + i * SileroVad::kHop);
      if (probs != nullptr && i < cap) probs[i] = p;
    }
    if (state_out != nullptr) memcpy(state_out, vad.state(), sizeof(float) * 2 * SileroVad::kState);
    return (int64_t)hops;
  } catch (const std::exception& e) {
    MSH_LOGF("silero_probabilities failed: %s", e.what());
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

// The rolling batch's plan on the host alone (rolling_plan.h): clips of lens[] arrive in pieces of piece_sizes[]; sub_of_clip[i]
// receives the sub-batch clip i went out in, piece_of_sub[s] the piece after which sub-batch s was submitted, first_of_sub[s]
// its first (= longest) clip.  Returns the number of sub-batches (more than max_subs: only the first max_subs are described).
int64_t msh_host_rolling_plan(const uint64_t* lens, const uint64_t* piece_sizes, uint64_t n_pieces, int32_t batch_clips,
                              float short_frac, int32_t narrow_runs, int32_t* sub_of_clip, int32_t* piece_of_sub,
                              int32_t* first_of_sub, uint64_t max_subs) {
  try {
    if ((lens == nullptr || piece_sizes == nullptr) && n_pieces > 0) return MSH_ERR_INVALID_ARGUMENT;
    RollingPlanner plan(batch_clips, short_frac, narrow_runs != 0);
    uint64_t off = 0, subs = 0;
    for (uint64_t p = 0; p < n_pieces; ++p) {
      for (const std::vector<uint32_t>& ids : plan.add(lens + off, (size_t)piece_sizes[p], p + 1 == n_pieces)) {
        if (sub_of_clip != nullptr)
          for (uint32_t id : ids) sub_of_clip[id] = (int32_t)subs;
        if (subs < max_subs) {
          if (piece_of_sub != nullptr) piece_of_sub[subs] = (int32_t)p;
          if (first_of_sub != nullptr) first_of_sub[subs] = ids.empty() ? -1 : (int32_t)ids[0];
        }
        ++subs;
      }
      off += piece_sizes[p];
    }
    if (plan.waiting() != 0) return MSH_ERR_UNKNOWN;
    return (int64_t)subs;
  } catch (const std::exception& e) {
    MSH_LOGF("rolling_plan failed: %s", e.what());
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

int32_t msh_host_effective_cpus(void) { return (int32_t)msh_host::effective_cpus(); }
int64_t msh_host_parse_cpu_list(const char* text, int32_t* out, uint64_t cap) {
  const std::vector<int> v = msh_host::parse_cpu_list(text != nullptr ? text : "");
  for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
  return (int64_t)v.size();
}

int64_t msh_host_vad_segments(const uint8_t* weights, uint64_t weights_size, float threshold, int32_t window, int32_t hop,
                              uint64_t look_behind, uint64_t max_segment, uint64_t hard_cap, const float* audio,
                              uint64_t n_samples, int32_t sample_rate, uint64_t chunk, int64_t* bounds, uint64_t max_segments) {
  try {
    std::shared_ptr<SileroWeights> w;
    if (weights != nullptr && weights_size > 0) {
      w.reset(new SileroWeights());
      w->load_memory(weights, (size_t)weights_size);
    }
    VoiceActivityDetector vad(threshold, window, hop, (size_t)look_behind, (size_t)max_segment, w, (size_t)hard_cap);
    vad.start();
    if (chunk == 0) chunk = n_samples ? n_samples : 1;
    for (uint64_t off = 0; off < n_samples; off += chunk)
      vad.process_audio(audio + off, (size_t)(n_samples - off < chunk ? n_samples - off : chunk), sample_rate);
    vad.stop();
    const std::vector<VadSegment>& segs = vad.segments();
    // the invariant the batch call's device-audio slices rest on (VadSegment::src_offset): a segment's audio is the verbatim
    // slice of the 16 kHz input, however the input was cut into calls (look-behind across call boundaries included)
    if (sample_rate == kSampleRate)
      for (const VadSegment& sg : segs)
        if (sg.src_offset + sg.audio.size() > n_samples ||
            memcmp(sg.audio.data(), audio + sg.src_offset, sg.audio.size() * sizeof(float)) != 0) {
          MSH_LOGF("vad_segments: a segment's audio is not the input slice at its src_offset %zu (+%zu)", sg.src_offset, sg.audio.size());
          return MSH_ERR_UNKNOWN;
        }
    for (size_t i = 0; i < segs.size() && i < max_segments; ++i) {
      // start (the detector's own sample count: VadSegment::src_offset, which the batch call slices the device VAD's audio
      // by) / length in samples of the 16 kHz stream + completeness
      bounds[3 * i] = (int64_t)segs[i].src_offset;
      bounds[3 * i + 1] = (int64_t)segs[i].audio.size();
      bounds[3 * i + 2] = segs[i].is_complete ? 1 : 0;
    }
    return (int64_t)segs.size();
  } catch (const std::exception& e) {
    MSH_LOGF("vad_segments failed: %s", e.what());
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

// The same detector fed with PRECOMPUTED Silero probabilities (one per whole hop of the 16 kHz clip, given in one
// process_audio call): the host half of the device-VAD path of batch calls.
int64_t msh_host_vad_segments_from_probs(const uint8_t* weights, uint64_t weights_size, float threshold, int32_t window, int32_t hop,
                                         uint64_t look_behind, uint64_t max_segment, uint64_t hard_cap, const float* audio,
                                         uint64_t n_samples, const float* probs, uint64_t n_probs, int64_t* bounds,
                                         uint64_t max_segments) {
  try {
    if (weights == nullptr || probs == nullptr) return MSH_ERR_INVALID_ARGUMENT;
    std::shared_ptr<SileroWeights> w(new SileroWeights());
    w->load_memory(weights, (size_t)weights_size);
    VoiceActivityDetector vad(threshold, window, hop, (size_t)look_behind, (size_t)max_segment, w, (size_t)hard_cap);
    vad.start();
    vad.process_audio(audio, (size_t)n_samples, kSampleRate, probs, (size_t)n_probs);
    vad.stop();
    const std::vector<VadSegment>& segs = vad.segments();
    for (size_t i = 0; i < segs.size() && i < max_segments; ++i) {
      bounds[3 * i] = (int64_t)segs[i].src_offset;
      bounds[3 * i + 1] = (int64_t)segs[i].audio.size();
      bounds[3 * i + 2] = segs[i].is_complete ? 1 : 0;
    }
    return (int64_t)segs.size();
  } catch (const std::exception& e) {
    MSH_LOGF("vad_segments_from_probs failed: %s", e.what());
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

int64_t msh_host_sanitize_utf8(const char* text, uint64_t n, char* out, uint64_t out_cap) {
  if (text == nullptr && n > 0) return MSH_ERR_INVALID_ARGUMENT;
  return copy_out(sanitize_utf8(std::string(text ? text : "", (size_t)n)), out, out_cap);
}

int64_t msh_host_resample(const float* in, uint64_t n, float in_rate, float out_rate, float* out, uint64_t out_cap) {
  if (in == nullptr && n > 0) return MSH_ERR_INVALID_ARGUMENT;
  std::vector<float> r = resample(std::vector<float>(in, in + n), in_rate, out_rate);
  if (out != nullptr) memcpy(out, r.data(), sizeof(float) * (r.size() < out_cap ? r.size() : (size_t)out_cap));
  return (int64_t)r.size();
}

}  // extern "C"
