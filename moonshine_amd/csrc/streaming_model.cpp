#include "streaming_model.h"

#include <stdlib.h>
#include <string.h>

#include <stdexcept>

#include "host_utils.h"

namespace msh_host {

namespace {
int json_int_field(const std::string& json, const char* key, int dflt) {
  // the reference's own reader: first "key": followed by an integer (streaming-model.cpp:78-95)
  const std::string search = std::string("\"") + key + "\":";
  size_t pos = json.find(search);
  if (pos == std::string::npos) return dflt;
  pos += search.size();
  while (pos < json.size() && (json[pos] == ' ' || json[pos] == '\t')) ++pos;
  bool neg = false;
  if (pos < json.size() && json[pos] == '-') {
    neg = true;
    ++pos;
  }
  int v = 0;
  bool any = false;
  while (pos < json.size() && json[pos] >= '0' && json[pos] <= '9') {
    v = v * 10 + (json[pos] - '0');
    ++pos;
    any = true;
  }
  if (!any) return dflt;
  return neg ? -v : v;
}
}  // namespace

int MoonshineStreamingState::memory_len() const { return msh_stream_query(owner->engine, slot, 0); }
int MoonshineStreamingState::accumulated_feature_count() const { return msh_stream_query(owner->engine, slot, 1); }
int MoonshineStreamingState::cache_seq_len() const { return msh_stream_query(owner->engine, slot, 2); }
int MoonshineStreamingState::encoder_frames_emitted() const { return msh_stream_query(owner->engine, slot, 3); }

MoonshineStreamingModel::MoonshineStreamingModel(int dev, int streams, int frames)
    : device(dev), max_streams(streams), max_memory_frames(frames) {}

MoonshineStreamingModel::~MoonshineStreamingModel() {
  msh_stream_destroy(engine);
  delete tokenizer;
}

int MoonshineStreamingModel::fail(int32_t rc) {
  last_error = msh_stream_last_error(engine);
  MSH_LOGF("streaming engine call failed (status %d): %s", rc, last_error.c_str());
  return 1;
}

int MoonshineStreamingModel::finish_load(const uint8_t* tokenizer_data, size_t tokenizer_size,
                                         const std::string& json) {
  msh_stream_info info;
  if (msh_stream_info_get(engine, &info) != MSH_OK) return 1;
  config.encoder_dim = info.encoder_dim;
  config.decoder_dim = info.decoder_dim;
  config.depth = info.depth;
  config.nheads = info.nheads;
  config.head_dim = info.head_dim;
  config.vocab_size = info.vocab_size;
  config.bos_id = info.bos_id;
  config.eos_id = info.eos_id;
  config.frame_len = info.frame_len;
  config.total_lookahead = info.total_lookahead;
  config.max_seq_len = info.max_seq_len;
  config.d_model_frontend = json_int_field(json, "d_model_frontend", info.encoder_dim);
  config.c1 = json_int_field(json, "c1", 2 * info.encoder_dim);
  config.c2 = json_int_field(json, "c2", info.encoder_dim);
  if (tokenizer_data != nullptr) tokenizer = new BinTokenizer(tokenizer_data, tokenizer_size);
  return 0;
}

int MoonshineStreamingModel::load(const char* model_dir, const char* tokenizer_path, int32_t /*model_type*/) {
  if (model_dir == nullptr || tokenizer_path == nullptr) return 1;
  std::vector<uint8_t> cfg;
  const std::string cfg_path = join_path(model_dir, "streaming_config.json");
  if (!read_file(cfg_path, &cfg)) {
    last_error = "Failed to read config file: " + cfg_path;  // streaming-model.cpp:218-221
    return 1;
  }
  const std::string json(cfg.begin(), cfg.end());
  const std::string wts = join_path(model_dir, "model.safetensors");
  const int32_t rc = msh_stream_create(device, wts.c_str(), json.c_str(), max_streams, max_memory_frames, &engine);
  if (rc != MSH_OK) {
    last_error = msh_stream_last_error(nullptr);
    MSH_LOGF("Failed to load streaming model from '%s' (status %d): %s", model_dir, rc, last_error.c_str());
    return 1;
  }
  if (finish_load(nullptr, 0, json) != 0) return 1;
  tokenizer = BinTokenizer::from_file(tokenizer_path);
  return 0;
}

int MoonshineStreamingModel::load_from_memory(const uint8_t* weights, size_t weights_size, const std::string& json,
                                              const uint8_t* tokenizer_data, size_t tokenizer_size,
                                              int32_t /*model_type*/) {
  if (weights == nullptr || tokenizer_data == nullptr) return 1;
  const int32_t rc = msh_stream_create_from_memory(device, weights, weights_size, json.c_str(), max_streams,
                                                   max_memory_frames, &engine);
  if (rc != MSH_OK) {
    last_error = msh_stream_last_error(nullptr);
    MSH_LOGF("Failed to load streaming model from memory (status %d): %s", rc, last_error.c_str());
    return 1;
  }
  return finish_load(tokenizer_data, tokenizer_size, json);
}

MoonshineStreamingState* MoonshineStreamingModel::create_state() {
  std::lock_guard<std::mutex> lock(processing_mutex);
  const int32_t slot = msh_stream_open(engine);
  if (slot < 0) {
    fail(slot);
    return nullptr;
  }
  MoonshineStreamingState* s = new MoonshineStreamingState();
  s->owner = this;
  s->slot = slot;
  ++states_live_;
  return s;
}

void MoonshineStreamingModel::free_state(MoonshineStreamingState* state) {
  if (state == nullptr) return;
  std::lock_guard<std::mutex> lock(processing_mutex);
  msh_stream_close(engine, state->slot);
  --states_live_;
  delete state;
}

int MoonshineStreamingModel::reset_state(MoonshineStreamingState* state) {
  if (state == nullptr) return 1;
  std::lock_guard<std::mutex> lock(processing_mutex);
  const int32_t rc = msh_stream_reset(engine, state->slot);
  return rc == MSH_OK ? 0 : fail(rc);
}

int MoonshineStreamingModel::process_audio_chunk(MoonshineStreamingState* state, const float* audio_chunk,
                                                 size_t chunk_len, int* features_out) {
  if (state == nullptr) return 1;                            // streaming-model.cpp:445-448
  if (audio_chunk == nullptr && chunk_len > 0) return 1;     // :449-452
  if (chunk_len == 0) {                                      // :454-457
    if (features_out) *features_out = 0;
    return 0;
  }
  std::lock_guard<std::mutex> lock(processing_mutex);
  const float* ptrs[1] = {audio_chunk};
  const uint64_t lens[1] = {chunk_len};
  int32_t feats = 0;
  const int32_t rc = msh_stream_process_audio(engine, 1, &state->slot, ptrs, lens, &feats);
  if (rc != MSH_OK) return fail(rc);
  if (features_out) *features_out = feats;
  return 0;
}

int MoonshineStreamingModel::encode(MoonshineStreamingState* state, bool is_final, int* new_frames_out) {
  if (state == nullptr) return 1;
  std::lock_guard<std::mutex> lock(processing_mutex);
  const uint8_t fin = is_final ? 1 : 0;
  int32_t fresh = 0;
  const int32_t rc = msh_stream_encode(engine, 1, &state->slot, &fin, &fresh);
  if (rc != MSH_OK) return fail(rc);
  if (new_frames_out) *new_frames_out = fresh;
  return 0;
}

int MoonshineStreamingModel::decode_step(MoonshineStreamingState* state, int token, float* logits_out) {
  if (state == nullptr || logits_out == nullptr) return 1;
  return decode_tokens(state, &token, 1, logits_out);
}

int MoonshineStreamingModel::decode_tokens(MoonshineStreamingState* state, const int* tokens, int tokens_len,
                                           float* logits_out) {
  if (state == nullptr || tokens == nullptr || tokens_len <= 0 || logits_out == nullptr) return 1;
  std::lock_guard<std::mutex> lock(processing_mutex);
  const int32_t* tp[1] = {tokens};
  const int32_t n[1] = {tokens_len};
  const int32_t rc = msh_stream_decode_tokens(engine, 1, &state->slot, tp, n, logits_out);
  return rc == MSH_OK ? 0 : fail(rc);
}

int MoonshineStreamingModel::decode_full(MoonshineStreamingState* state, const int* speculative_tokens,
                                         int speculative_len, int** tokens_out, int* tokens_len_out) {
  if (state == nullptr || tokens_out == nullptr || tokens_len_out == nullptr) return 1;
  std::vector<std::vector<int>> drafts(1), out;
  if (speculative_tokens != nullptr && speculative_len > 0)
    drafts[0].assign(speculative_tokens, speculative_tokens + speculative_len);
  if (decode_full_batch({state}, drafts, {-1}, &out) != 0) return 1;
  *tokens_len_out = (int)out[0].size();
  *tokens_out = nullptr;
  if (!out[0].empty()) {
    *tokens_out = static_cast<int*>(malloc(out[0].size() * sizeof(int)));  // caller frees (:1387-1394)
    if (*tokens_out == nullptr) return 1;
    memcpy(*tokens_out, out[0].data(), out[0].size() * sizeof(int));
  }
  return 0;
}

void MoonshineStreamingModel::decoder_reset(MoonshineStreamingState* state) {
  if (state == nullptr) return;
  decoder_reset_batch({state});
}

std::string MoonshineStreamingModel::tokens_to_text(const std::vector<int64_t>& tokens) {
  std::vector<int32_t> ids(tokens.begin(), tokens.end());
  return tokenizer->tokens_to_text(ids.data(), ids.size());
}

std::vector<int32_t> MoonshineStreamingModel::text_to_tokens(const std::string& text) {
  if (tokenizer == nullptr) return {};
  return tokenizer->text_to_tokens(text, /*bpe=*/true);
}

int MoonshineStreamingModel::set_biaser(const ContextBiaser& biaser) {
  std::lock_guard<std::mutex> lock(processing_mutex);
  int32_t rc;
  if (biaser.empty()) {
    rc = msh_stream_set_bias(engine, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  } else {
    const ContextBiaser::Flat f = biaser.flatten();
    rc = msh_stream_set_bias(engine, (int32_t)f.depth.size(), f.child_off.data(), f.child_tok.data(), f.child_node.data(),
                             f.depth.data(), f.depth_bonus.data(), (int32_t)f.depth_bonus.size());
  }
  return rc == MSH_OK ? 0 : fail(rc);
}

// ---- batched forms ----
int MoonshineStreamingModel::process_audio_batch(const std::vector<MoonshineStreamingState*>& states,
                                                 const std::vector<const float*>& audio,
                                                 const std::vector<size_t>& lens) {
  if (states.empty()) return 0;
  std::lock_guard<std::mutex> lock(processing_mutex);
  std::vector<int32_t> slots;
  std::vector<uint64_t> n(lens.begin(), lens.end());
  for (auto* s : states) slots.push_back(s->slot);
  const int32_t rc = msh_stream_process_audio(engine, (int32_t)slots.size(), slots.data(), audio.data(), n.data(), nullptr);
  return rc == MSH_OK ? 0 : fail(rc);
}

int MoonshineStreamingModel::encode_batch(const std::vector<MoonshineStreamingState*>& states,
                                          const std::vector<uint8_t>& is_final) {
  if (states.empty()) return 0;
  std::lock_guard<std::mutex> lock(processing_mutex);
  std::vector<int32_t> slots;
  for (auto* s : states) slots.push_back(s->slot);
  const int32_t rc = msh_stream_encode(engine, (int32_t)slots.size(), slots.data(), is_final.data(), nullptr);
  return rc == MSH_OK ? 0 : fail(rc);
}

int MoonshineStreamingModel::decoder_reset_batch(const std::vector<MoonshineStreamingState*>& states) {
  if (states.empty()) return 0;
  std::lock_guard<std::mutex> lock(processing_mutex);
  std::vector<int32_t> slots;
  for (auto* s : states) slots.push_back(s->slot);
  const int32_t rc = msh_stream_decoder_reset(engine, (int32_t)slots.size(), slots.data());
  return rc == MSH_OK ? 0 : fail(rc);
}

int MoonshineStreamingModel::cross_attention(MoonshineStreamingState* state, const std::vector<int>& tokens,
                                             std::vector<float>* out, int dims[3]) {
  std::lock_guard<std::mutex> lock(processing_mutex);
  std::vector<int32_t> t(tokens.begin(), tokens.end());
  int32_t d[3] = {0, 0, 0};
  const int64_t need = msh_stream_cross_attention(engine, state->slot, t.data(), (int32_t)t.size(), nullptr, 0, d);
  if (need < 0) return fail((int32_t)need);
  out->assign((size_t)need, 0.f);
  if (need > 0) {
    const int64_t rc = msh_stream_cross_attention(engine, state->slot, t.data(), (int32_t)t.size(), out->data(), (uint64_t)need, d);
    if (rc < 0) return fail((int32_t)rc);
  }
  dims[0] = d[0], dims[1] = d[1], dims[2] = d[2];
  return 0;
}

int MoonshineStreamingModel::decode_full_batch(const std::vector<MoonshineStreamingState*>& states,
                                               const std::vector<std::vector<int>>& drafts,
                                               const std::vector<int>& max_tokens,
                                               std::vector<std::vector<int>>* tokens_out) {
  tokens_out->assign(states.size(), {});
  if (states.empty()) return 0;
  std::lock_guard<std::mutex> lock(processing_mutex);
  const int32_t n = (int32_t)states.size();
  std::vector<int32_t> slots, dl(n), counts(n), mt(max_tokens.begin(), max_tokens.end());
  std::vector<const int32_t*> dp(n);
  for (int32_t i = 0; i < n; ++i) {
    slots.push_back(states[i]->slot);
    dp[i] = drafts[i].empty() ? nullptr : drafts[i].data();
    dl[i] = (int32_t)drafts[i].size();
  }
  const int32_t stride = config.max_seq_len + 8;
  std::vector<int32_t> toks((size_t)n * stride);
  const int32_t rc = msh_stream_decode_full(engine, n, slots.data(), dp.data(), dl.data(), mt.data(), toks.data(),
                                            counts.data(), stride, nullptr);
  if (rc != MSH_OK) return fail(rc);
  for (int32_t i = 0; i < n; ++i) (*tokens_out)[i].assign(toks.begin() + (size_t)i * stride,
                                                          toks.begin() + (size_t)i * stride + counts[i]);
  return 0;
}

}  // namespace msh_host
