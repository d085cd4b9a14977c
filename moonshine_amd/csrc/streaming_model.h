// Host mirror of the reference's MoonshineStreamingModel / MoonshineStreamingState
// (reference core/moonshine-streaming-model.h:18-201) over the MI355X streaming engine
// (include/moonshine_hip.h, msh_stream_*).  Same method names, argument meaning and status-code
// convention (0 = success) as the reference struct the Transcriber drives; the state's tensors live in
// a device slot instead of host vectors, and every method has a *_batch form that takes many states
// at once because the GPU wants them together.
#pragma once

#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/moonshine_hip.h"
#include "context_biaser.h"
#include "host_text_vad.h"

namespace msh_host {

struct MoonshineStreamingConfig {  // reference core/moonshine-streaming-model.h:18-33
  int encoder_dim = 0, decoder_dim = 0, depth = 0, nheads = 0, head_dim = 0, vocab_size = 0;
  int bos_id = 1, eos_id = 2, frame_len = 80, total_lookahead = 0, d_model_frontend = 0, c1 = 0, c2 = 0;
  int max_seq_len = 448;
};

struct MoonshineStreamingModel;

struct MoonshineStreamingState {  // reference :36-71; the buffers themselves are in device slot `slot`
  MoonshineStreamingModel* owner = nullptr;
  int32_t slot = -1;
  int memory_len() const;
  int cache_seq_len() const;
  int accumulated_feature_count() const;
  int encoder_frames_emitted() const;
};

struct MoonshineStreamingModel {
  msh_stream_engine* engine = nullptr;
  BinTokenizer* tokenizer = nullptr;
  std::mutex processing_mutex;
  MoonshineStreamingConfig config;
  std::string last_error;
  int device = 0, max_streams = 64, max_memory_frames = 2048;
  int states_live_ = 0;

  MoonshineStreamingModel(int device, int max_streams, int max_memory_frames);
  ~MoonshineStreamingModel();

  // model_dir holds model.safetensors + streaming_config.json (reference :118-119 loads five .ort graphs
  // from the same directory)
  int load(const char* model_dir, const char* tokenizer_path, int32_t model_type);
  int load_from_memory(const uint8_t* weights, size_t weights_size, const std::string& config_json,
                       const uint8_t* tokenizer_data, size_t tokenizer_size, int32_t model_type);

  MoonshineStreamingState* create_state();          // reference :181
  void free_state(MoonshineStreamingState* state);
  int states_in_use() {   // device slots held by live states (what the Transcriber balances streams over devices with)
    std::lock_guard<std::mutex> lock(processing_mutex);
    return states_live_;
  }
  int reset_state(MoonshineStreamingState* state);   // MoonshineStreamingState::reset

  int process_audio_chunk(MoonshineStreamingState* state, const float* audio_chunk, size_t chunk_len,
                          int* features_out);                                           // :145
  int encode(MoonshineStreamingState* state, bool is_final, int* new_frames_out);       // :149
  int decode_step(MoonshineStreamingState* state, int token, float* logits_out);        // :153
  int decode_tokens(MoonshineStreamingState* state, const int* tokens, int tokens_len, float* logits_out);  // :159
  // tokens_out is malloc'd, the caller frees it (:165); no ContextBiaser in this build
  int decode_full(MoonshineStreamingState* state, const int* speculative_tokens, int speculative_len,
                  int** tokens_out, int* tokens_len_out);                               // :174
  void decoder_reset(MoonshineStreamingState* state);                                   // :178
  // word timestamps (reference :946-1066 + core/transcriber.cpp:1028-1068): cross-attention of `tokens` fed from an
  // empty self cache as [depth*heads][tokens][memory_len]; dims = {depth*heads, tokens, memory_len}
  int cross_attention(MoonshineStreamingState* state, const std::vector<int>& tokens, std::vector<float>* out, int dims[3]);
  std::string tokens_to_text(const std::vector<int64_t>& tokens);                       // :184
  // :189 -- byte-pair encoding (kTokenizerEncoding = kBpe, streaming-model.cpp:57); empty without a tokenizer
  std::vector<int32_t> text_to_tokens(const std::string& text);
  // Hands the compiled key terms to the device (the reference passes a ContextBiaser* into decode_full, :174-176;
  // here token choice runs on the GPU, so the trie lives there).  An empty biaser switches biasing off.
  int set_biaser(const ContextBiaser& biaser);

  // batched forms (no reference counterpart)
  int process_audio_batch(const std::vector<MoonshineStreamingState*>& states, const std::vector<const float*>& audio,
                          const std::vector<size_t>& lens);
  int encode_batch(const std::vector<MoonshineStreamingState*>& states, const std::vector<uint8_t>& is_final);
  int decoder_reset_batch(const std::vector<MoonshineStreamingState*>& states);
  // drafts[i] empty = decode from BOS; max_tokens[i] < 0 = the reference rule from the memory length
  int decode_full_batch(const std::vector<MoonshineStreamingState*>& states, const std::vector<std::vector<int>>& drafts,
                        const std::vector<int>& max_tokens, std::vector<std::vector<int>>* tokens_out);

 private:
  int finish_load(const uint8_t* tokenizer_data, size_t tokenizer_size, const std::string& config_json);
  int fail(int32_t rc);
};

}  // namespace msh_host
