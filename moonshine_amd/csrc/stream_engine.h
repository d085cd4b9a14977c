// MI355X streaming Moonshine engine: the device-side replacement of the five ORT sessions behind
// reference core/moonshine-streaming-model.{h,cpp} (frontend, encoder, adapter, cross_kv, decoder_kv),
// batched over streams.
//
// Every stream owns a slot of fixed-capacity device state (what MoonshineStreamingState keeps in host
// vectors, streaming-model.h:36-71):
//   conv1_buf [4][De] / conv2_buf [4][2De] bf16    the frontend's carried frames
//   features  [Mcap][De] fp32                       accumulated_features
//   memory    [Mcap][Dd] fp32                       adapter output (kept for inspection)
//   crossK/V  [L][Dd][Mcap] bf16 (keys contiguous)   cross-attention keys / values, transposed; appended per update
//   selfK/V   [L][Scap][Dd] bf16                    decoder self-attention cache
//   result    [Scap] int32 + SlotDev                decode_full bookkeeping
// Calls take a list of slots and work on all of them at once: rows of every GEMM are the concatenation of
// the streams' rows, attention kernels look the owning stream up per row.
//
// Differences from the reference driver that do not change results:
//   * cross K/V are appended for the new memory frames only (the reference re-projects the whole memory on
//     every update, streaming-model.cpp:779-860; a key's projection does not depend on later frames);
//   * a rejected draft suffix is rolled back by truncating the self cache instead of re-running the accepted
//     prefix (streaming-model.cpp:1337-1365): causal attention makes the two identical.
#pragma once

#include <string>
#include <vector>

#include "engine.h"
#include "stream_kernels.h"

namespace msh {

struct StreamingConfig {  // streaming_config.json (streaming-model.cpp:97-113) + additive keys
  int encoder_dim = 0, decoder_dim = 0, depth = 0, nheads = 0, head_dim = 0, vocab_size = 0;
  int bos_id = 1, eos_id = 2, frame_len = 80, total_lookahead = 0, max_seq_len = 448;
  int encoder_heads = 8, enc_layers = 0, enc_ffn = 0, dec_ffn = 0, max_pos = 0;
  float rope_theta = 10000.f, partial_rotary = 0.8f;
  std::vector<std::pair<int, int>> windows;
};

class StreamingEngine {
 public:
  StreamingEngine(int device, int max_slots, int max_memory_frames);
  ~StreamingEngine();

  void load(const SafeTensors& st, const std::string& config_json);
  const StreamingConfig& config() const { return cfg_; }
  bool loaded() const { return loaded_; }
  int max_slots() const { return max_slots_; }
  int memory_capacity() const { return Mcap_; }

  int create_stream();        // returns a slot id
  void free_stream(int slot);
  void reset_stream(int slot);  // MoonshineStreamingState::reset

  // frontend over the new audio of n streams (host pointers); features_out[i] = new feature frames
  void process_audio(int n, const int* slots, const float* const* pcm, const uint64_t* lens, int* features_out);
  // window encoder + adapter + cross-K/V append (MoonshineStreamingModel::encode)
  void encode(int n, const int* slots, const uint8_t* is_final, int* new_frames_out);
  void decoder_reset(int n, const int* slots);
  // wide decoder pass: tokens of stream i appended to its cache; logits_out (nullable) gets the rows of all
  // streams concatenated, [sum(lens)][V]
  void decode_tokens(int n, const int* slots, const int32_t* const* tokens, const int* lens, float* logits_out);
  // decode_full for n streams at once.  max_tokens[i] < 0 = the reference rule from the memory length.
  // word timestamps: resets the slot's decoder, feeds `tokens` and returns their cross-attention [depth*heads][n][memory_len]
  void cross_attention(int slot, const int32_t* tokens, int n, float* out, size_t cap, int dims[3]);
  void decode_full(int n, const int* slots, const int32_t* const* drafts, const int* draft_lens, const int* max_tokens,
                   int32_t* tokens_out, int32_t* counts_out, int tokens_stride, int32_t* accepted_out);

  // contextual biasing trie (copied to the device); n_nodes == 0 switches it off
  void set_bias(int n_nodes, const int32_t* child_off, const int32_t* child_tok, const int32_t* child_node,
                const int32_t* depth, const float* depth_bonus, int n_depth_bonus);

  int memory_len(int slot) const { return st(slot).mem_len; }
  int feature_count(int slot) const { return st(slot).feat_count; }
  int cache_len(int slot) const { return st(slot).cache_len; }
  int frames_emitted(int slot) const { return st(slot).emitted; }
  void get_memory(int slot, float* out);    // [memory_len][Dd]
  void get_features(int slot, float* out);  // [feature_count][De]
  int max_tokens_for(int slot) const;       // streaming-model.cpp:1217-1219
  // decode_full statistics since the last reset (GPU time between HIP events on the engine's stream, no extra synchronisation):
  // 0 = auto-regressive passes run, 1 = wide (verify) passes run, 2 = microseconds in AR loops, 3 = microseconds in verify passes
  // (embedding + pass + bias + argmax + verify); 4 resets.  What a pass costs does not depend on the draft's acceptance rate.
  long decode_stat(int what);
  void synchronize();
  void profile_enable(bool on) { prof_.enable(on, stream_); }
  void profile_reset() { prof_.reset(stream_); }
  std::vector<ProfEntry> profile_get() { return prof_.get(stream_); }
  hipStream_t stream() const { return stream_; }

 private:
  struct SlotHost {
    bool used = false;
    std::vector<float> pending;  // samples that do not fill a 320-sample period yet
    int feat_count = 0, emitted = 0, pos_offset = 0, mem_len = 0, cache_len = 0;
  };
  const SlotHost& st(int slot) const;
  SlotHost& st(int slot);
  void check_slots(int n, const int* slots) const;
  void upload(const std::vector<float>& src, float** dst);
  void upload_bf16(const std::vector<float>& src, bf16_t** dst);
  void upload_bf16_fm(const std::vector<float>& src, int rows, int K, bf16_t** dst);
  void reserve_decoder_buffers(int rows);
  bool fm_ok_ = false;   // AR steps on FM operands
  int ar_keys_bound_ = 0;  // while decode_full runs its AR steps: an upper bound of any row's key count (0 = unknown)
  const int* ar_row_mem_d_ = nullptr;   // ... and the rows' memory lengths on the device (null outside the AR steps)
  template <class T>
  T* stage(DevBuf& buf, const std::vector<T>& host);  // async H2D of a small descriptor array
  // runs_d (optional): the pass's rows as runs of consecutive rows of one stream, for the shared-K/V cross-attention
  // pval / pidx (optional, then logits may be null): the LM head as the tiled GEMM whose epilogue keeps only each 128-column
  // tile's (max, lowest index) per row -- [M][gemm_argmax_tiles(V)] -- instead of M x V logits
  void decoder_pass(int M, const int* row_slot_d, const int* row_pos_d, float* logits, const int2* runs_d = nullptr,
                    int n_runs = 0, float* pval = nullptr, int* pidx = nullptr, bool fm = false);
  // rows of `rs` (slot per row) -> runs of <= kCrossRunRows consecutive rows with the same slot, staged on the device;
  // nullptr when the pass has no run longer than one row (the auto-regressive steps) or the kernel does not cover the shape
  const int2* stage_runs(const std::vector<int>& rs, int* n_runs);
  bool runs_wide_ = false;   // the staged runs are whole runs of <= kCrossWideRows rows for the MFMA cross-attention
  void push_slot_state(int slot);

  struct EncW {
    float *ln1, *ln2, *b1, *b2;
    bf16_t *wqkv, *wo, *fc1, *fc2;
  };
  struct DecW {
    float *ln1, *ln2, *ln3, *b1, *b2;
    bf16_t *wqkv, *wo, *wq_c, *wo_c, *fc1, *fc2;
    bf16_t *wqkv_f, *wq_c_f, *fc1_f;  // LayerNorm scale folded in (LN-fused small-batch GEMMs of the AR steps)
    // the six AR-step weights in the fragment-major order of the decode GEMMs (kernels.h fm16), LayerNorm scales folded
    // in where the GEMM is LN-fused; null when the widths are not FM-compiled (stream_fm_supported)
    bf16_t *wqkv_fm = nullptr, *wo_fm = nullptr, *wq_c_fm = nullptr, *wo_c_fm = nullptr, *fc1_fm = nullptr, *fc2_fm = nullptr;
  };

  // per-kernel-group timing (profiler.h); while it is on, the AR steps run eagerly instead of from their hipGraph.
  // pass_cross_bytes_: algorithmic K / V bytes of the decoder pass being enqueued (rows x their stream's memory), set by
  // the callers of decoder_pass for the profiler
  ScopeProfiler prof_;
  double pass_cross_bytes_ = 0.0;
  hipGraphExec_t ar_graph_ = nullptr;  // one autoregressive decode step (decode_full), replayed
  hipEvent_t stat_ev_[3] = {nullptr, nullptr, nullptr};
  // Pinned host memory for the engine's small transfers.  A copy between PAGEABLE host memory and the device goes through the
  // runtime's bounce buffers and blocks the caller (~30-50 us each): decode_full read its results back with two such copies per
  // stream (128 per call at 64 streams: 6 of the 8.5 ms of host time a call cost beside its GPU work) and staged nine descriptor
  // arrays with a copy + stream drain each.  pin_: a ring for host -> device staging (slices stay untouched until the stream has
  // been drained since they were handed out); rb_: the read-back area (slot records + token table + the active counter).
  unsigned char* pin_ = nullptr;
  size_t pin_cap_ = 0, pin_off_ = 0, pin_live_ = 0;
  void* pin_take(size_t bytes);
  unsigned char* rb_ = nullptr;
  size_t rb_cap_ = 0;
  void* rb_area(size_t bytes);
  long stat_ar_passes_ = 0, stat_verify_passes_ = 0;
  double stat_ar_us_ = 0.0, stat_verify_us_ = 0.0;
  std::string ar_key_;
  float* capture_probs_ = nullptr;  // set while cross_attention() runs its pass
  int capture_ecap_ = 0;
  DevBuf probs_;
  int device_;
  hipStream_t stream_ = nullptr;
  bool loaded_ = false;
  StreamingConfig cfg_;
  int max_slots_, Mcap_, Scap_ = 0;
  std::vector<void*> allocs_;

  // weights
  float k_scale_ = 0.75f;
  bf16_t *lin_w_ = nullptr, *conv1_w_ = nullptr, *conv2_w_ = nullptr, *proj_w_ = nullptr, *cross_w_ = nullptr,
         *head_w_ = nullptr, *head_wf_ = nullptr;
  float *conv1_b_ = nullptr, *conv2_b_ = nullptr, *enc_ln_ = nullptr, *pos_emb_ = nullptr, *embed_f32_ = nullptr,
        *dec_ln_ = nullptr, *rope_cos_ = nullptr, *rope_sin_ = nullptr;
  int rot_pairs_ = 0;
  std::vector<EncW> enc_;
  std::vector<DecW> dec_;

  // per-slot device state (slabs over max_slots)
  bf16_t *conv1_buf_ = nullptr, *conv2_buf_ = nullptr, *crossK_ = nullptr, *crossV_ = nullptr, *selfK_ = nullptr,
         *selfV_ = nullptr;
  float *features_ = nullptr, *memory_ = nullptr;
  int32_t* result_ = nullptr;
  SlotDev* slots_d_ = nullptr;
  int32_t* n_active_d_ = nullptr;
  std::vector<SlotHost> slots_;

  // workspace (grow-only)
  DevBuf audio_, frames_, hidden_, c1out_, feat_pk_, segs_, jobs_, H_, Y_, Y32_, QKV_, AO_, Z_, Q_, rowlo_, rowhi_,
      newrows_, newpos_, newslot_, newidx_, adp16_, adp32_, mem16_, mem32_, crosstmp_, rowslot_, rowpos_, tokens_, jobmem_,
      logits_, pred_, draft_, decjobs_, stepH_, steppos_;
  DevBuf runs_, pval_, pidx_, tiles_;
  DevBuf bias_off_, bias_tok_, bias_node_, bias_depth_, bias_bonus_, bias_prefix_;
  BiasTrie bias_{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
};

}  // namespace msh
