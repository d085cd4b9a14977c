// MI355X Moonshine engine: weights resident in HBM, batched encoder, lock-step batched greedy decoder.
// This is the device-side replacement for the reference's ORT sessions
// (reference core/moonshine-model.cpp:216-563 + core/ort-utils/): host code above it only sees
// "PCM clips in, token ids out".
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"
#include "profiler.h"
#include "safetensors.h"

namespace msh {

struct ModelConfig {
  std::string arch;
  int hidden = 0, ffn = 0, enc_layers = 0, dec_layers = 0, heads = 0, vocab = 0;
  int bos = 1, eos = 2;  // reference core/moonshine-model.cpp:56-57
  float rope_theta = 10000.f;
  float partial_rotary = 0.9f;
  int head_dim() const { return hidden / heads; }
  int rot_pairs() const { return (int)(head_dim() * partial_rotary) / 2; }
};

// A device allocation that only ever grows.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  void* raw = nullptr;   // the allocation p lies in (p = raw + a skew, see reserve)
  bool reserve(size_t bytes);  // returns true if reallocated (contents lost, zero-filled)
  void release();
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct EncLayerW {
  float *ln1, *ln2, *b1, *b2;
  bf16_t *wqkv, *wo, *fc1, *fc2;
  bf16_t* mlp = nullptr;   // fc1 (LayerNorm scale folded in) + b1 + fc2 packed for the fused MLP kernel (k_mlp.hip), or null
  bf16_t* qkv_panel = nullptr;   // q | k | v with the LayerNorm scale folded in, packed for the panel kernel (k_panel.hip), or null
};
struct DecLayerW {
  float *ln1, *ln2, *ln3, *b1, *b2;
  bf16_t *wqkv, *wo, *wq_c, *wo_c, *fc1, *fc2;  // MFMA-fragment-major (kernels.h fm16)
  bf16_t* wq_c_rm;                              // cross-q again, row-major (fused-q attention kernel, small batches)
  // absorbed cross-attention (k_xattn.hip), FM, or null when the shape is not supported:
  //   wqk [heads * D][D]  rows (h, d) = softmax scale * sum_j Wk[h j][d] * (Wq[h j][:] * gamma)   (query side, LayerNorm fused)
  //   wvo [D][heads * D]  columns (h, d) = sum_j Wo[:][h j] * Wv[h j][d]                          (output side, residual update)
  bf16_t *wqk = nullptr, *wvo = nullptr;
  // the two factors of wqk for the two-stage query kernel (k_crossq.hip): scale * Wq * diag(gamma) as FM [heads * 64][D]
  // and Wk in pack_crossq_wk's order; wqk itself is then only uploaded on request (MSH_XATTN_QT=1, the merged-weight GEMM)
  bf16_t *wq1 = nullptr, *wk2 = nullptr;
};

class Engine {
 public:
  explicit Engine(int device);
  ~Engine();
  // Checkpoint validation without a GPU (msh_host_check_weights): runs load_weights with every upload skipped and returns
  // the dimensions it found; throws what a real load would throw.
  static ModelConfig check_weights(const SafeTensors& st, int expect_arch);

  void load_weights(const SafeTensors& st, int expect_arch /* -1 any, 0 tiny, 1 base */);
  // Use the weight buffers of a loaded engine on the same device (read-only; `owner` must outlive this engine).
  void share_weights_from(const Engine& owner);
  const ModelConfig& config() const { return cfg_; }
  bool loaded() const { return loaded_; }

  // Encoder + cross-K/V projection for a batch of clips.  pcm[i] is a host pointer unless on_device.
  void encode(const float* const* pcm, const uint64_t* n_samples, uint32_t count, bool on_device,
              float max_tokens_per_second);
  // Lock-step greedy decode of the encoded batch.  forced_steps < 0: stop on EOS / per-clip budget
  // (reference loop); >= 0: ignore EOS and run exactly that many steps.  teacher (optional,
  // [count][teacher_stride], row starts with BOS) feeds given ids instead of the argmax.
  // logits_out (optional) receives [steps][count][V] fp32.  Returns the number of steps run.
  int decode(int forced_steps, const int32_t* teacher, int teacher_stride, float* logits_out, int max_logit_steps,
             int32_t* tokens_out, int32_t* counts_out, int tokens_stride);

  uint32_t batch_count() const { return n_clips_; }
  int clip_frames(uint32_t clip) const { return clips_h_.at(clip).T; }
  int max_decode_len() const { return max_steps_; }
  void get_encoder_output(uint32_t clip, float* out);  // [T][D] fp32 (needs keep_encoder_f32)
  void set_keep_encoder_f32(bool v) { keep_enc_f32_ = v; }

  // Word timestamps: keep the cross-attention probabilities of every decode step of the next decode() calls (eager
  // decode, one extra kernel per layer and step).  get_cross_attention copies clip `clip`'s [layers*heads][steps][T]
  // fp32 block (steps = tokens generated, T = encoder frames) to `out` if it holds `cap` >= that many floats, and
  // returns the three dimensions.
  void set_capture_cross_attention(bool on) {
    if (on && kv_fp8_) throw std::invalid_argument("cross-attention capture (word timestamps) needs kv_dtype = bf16");
    if (on && absorbed_) encoded_ = false;   // the capture reads K^T, which the absorbed form never writes: encode again
    capture_cross_ = on;
  }
  // How the decoder's cross-attention runs (k_xattn.hip): 1 (and 0, the default) = the projected K^T / V^T stream, the
  // reference's form; 2 = the absorbed form -- one pass over the encoder output for all heads, no cross K/V.  ONE form per
  // engine whatever the batch size: a clip's ids never depend on how many clips share its batch (until round 4 mode 0
  // switched by batch size; the host layer now resolves its `auto` ONCE at load, from the configured sub-batch size).
  // Mode 2 on an architecture without the absorbed operands, with the word-timestamp capture or with fp8 keys is an error
  // of the next encode, not a silent change of form.  Applies to the next encode; lanes take it when they are created.
  void set_cross_mode(int mode) {
    if (mode < 0 || mode > 2) throw std::invalid_argument("cross mode: 0 / 1 = projected K/V, 2 = absorbed");
    if (mode == 2 && loaded_ && !cross_absorbed_available())
      throw std::invalid_argument("cross mode 2 (absorbed): this architecture has no absorbed operands (8 heads, hidden 288 / 416 only)");
    if (mode != cross_mode_) {
      cross_mode_ = mode;
      encoded_ = false;
    }
  }
  int cross_mode() const { return cross_mode_; }
  // Kernel set: 0 (default) = chosen per call by its size (split-K encoder GEMMs for <= 1024 rows, tiled ones below 16 k
  // rows, panel / fused kernels above; decode kernels by clip count: 1-2 / <= 4 / 5..63 / >= 64, LM head tiled from 128) --
  // fastest at every size, but a clip's bits change across those lines.  1 = ONE kernel set, the large-batch one, for every
  // call: a clip's ids do not depend on how many clips share its call (the batch call's tail sub-batches, a single clip
  // through a throughput deployment); small calls run slower.  ONE per engine like the cross-attention form; lanes take it
  // when they are created.
  void set_uniform_kernels(bool on) {
    if (on != uniform_kernels_) {
      uniform_kernels_ = on;
      encoded_ = false;
    }
  }
  bool uniform_kernels() const { return uniform_kernels_; }
  bool cross_absorbed_available() const { return !dec_.empty() && dec_[0].wvo != nullptr; }
  // this engine is one of several lanes that decode at the same time on one GPU (BatchPipeline): its once-per-launch streams
  // go with the non-temporal policy (kernels.h dec_cross_absorbed)
  void set_shared_gpu(bool on) { shared_gpu_ = on; }
  bool cross_absorbed() const { return absorbed_; }   // of the batch encoded last
  // cross K^T / V^T storage: bf16 (default) or e4m3 bytes with per-column scales fixed at load; applies to the next encode.
  // Lanes (batches in flight) take the setting when they are created.
  void set_kv_fp8(bool on) {
    if (on && capture_cross_) throw std::invalid_argument("kv_dtype = fp8 cannot be combined with the cross-attention capture");
    if (on != kv_fp8_) {
      kv_fp8_ = on;
      encoded_ = false;
    }
  }
  bool kv_fp8() const { return kv_fp8_; }
  void get_cross_attention(uint32_t clip, float* out, size_t cap, int dims[3]);

  // per-kernel-group timing with HIP events on the engine stream
  void profile_enable(bool on);
  void profile_reset();
  std::vector<ProfEntry> profile_get();
  // average time between the two events of an EMPTY profiling scope on the engine stream: the part of every
  // per-launch figure that is event / dispatch bookkeeping rather than kernel time
  double profile_event_overhead_ms(int iters);
  // The decode cross-attention kernel launched back to back over the cross K/V of all layers (`rounds` sweeps, so every
  // launch streams its own 2*D*keys bytes from HBM like a real step does) between ONE pair of HIP events: average ms per
  // launch without per-launch event bookkeeping.  Needs a batch that was encoded and decoded at least once.
  double profile_cross_attention_ms(int rounds);
  // marginal cost of every decode kernel group inside a replayed hipGraph chain: fills "chain_*" profile entries
  void profile_decode_chain(int reps);
  // test hook: bytes of a named decode buffer of the last decode() call ("cache_k", "cache_v", "resid"); copies
  // min(bytes, size) to `dst` and returns the buffer's size
  size_t debug_read(const std::string& name, void* dst, size_t bytes);

  hipStream_t stream() const { return stream_; }
  void synchronize();

 private:
  struct ProfScope;
  struct DecodeGroup;
  void destroy_groups();
  void plan_batch(const uint64_t* n_samples, uint32_t count, float max_tokens_per_second);
  void run_encoder();
  void decode_step_enqueue(DecodeGroup& g);
  void* weight_alloc(size_t bytes);
  void upload(const std::vector<float>& src, float** dst);
  void upload_bf16(const std::vector<float>& src, bf16_t** dst);
  void upload_bf16_fm(const std::vector<float>& src, int rows, int K, bf16_t** dst);

  int device_;
  hipStream_t stream_ = nullptr;
  // second stream of the encoder's layer loop when the batch runs as two halves side by side (run_encoder)
  hipStream_t enc_stream2_ = nullptr;
  hipEvent_t enc_fork_ = nullptr, enc_join_ = nullptr;
  void* stream_probe_ = nullptr;
  struct DryRun {};
  explicit Engine(DryRun);   // no device: only load_weights' validation runs
  bool dry_run_ = false;
  ModelConfig cfg_;
  bool loaded_ = false;
  std::vector<void*> weight_allocs_;
  char* slab_ = nullptr;      // weight_alloc: the slab being filled
  size_t slab_used_ = 0;

  // weights
  bf16_t *conv1_w_ = nullptr, *conv2_w_ = nullptr, *conv3_w_ = nullptr;
  int conv_kperm_ = 0;   // 1: conv2_w_ / conv3_w_ are stored in the tap-inner k-order (conv_k_offset, gemm_common.h)
  float *conv2_s1_ = nullptr, *conv2_b2_ = nullptr, *conv3_b_ = nullptr, *enc_ln_ = nullptr;  // conv2_s1 / _b2: GroupNorm fold
  std::vector<EncLayerW> enc_;
  std::vector<DecLayerW> dec_;
  bf16_t *embed_bf16_ = nullptr, *embed_head_folded_ = nullptr, *cross_kv_w_ = nullptr;
  bf16_t* cross_kv_panel_w_ = nullptr;   // the fused cross K/V weight packed for the panel kernel (k_panel.hip), or null
  float *embed_f32_ = nullptr, *dec_ln_ = nullptr;
  float *rope_cos_ = nullptr, *rope_sin_ = nullptr;
  int rope_max_pos_ = 0;

  // current batch
  uint32_t n_clips_ = 0;
  std::vector<ClipMeta> clips_h_;
  long R_ = 0;        // packed rows
  long kv_keys_ = 0;  // sum of Tk
  int max_rows_ = 0, max_steps_ = 0;
  bool encoded_ = false, keep_enc_f32_ = false;
  bool capture_cross_ = false;
  bool kv_fp8_ = false;
  int cross_mode_ = 0;
  bool uniform_kernels_ = false;
  bool shared_gpu_ = false;
  bool absorbed_ = false;   // the encoded batch decodes with the absorbed cross-attention: K^T / V^T were not written
  float *kv_qscale_ = nullptr, *kv_dq_ = nullptr;   // [L * 2D]: e4m3 scale of every cross-KV column and its inverse
  size_t kv_bytes() const { return kv_fp8_ ? 1 : 2; }
  const bf16_t* kv_layer(const DevBuf& b, int l) const {
    return reinterpret_cast<const bf16_t*>(static_cast<const char*>(b.p) + (size_t)l * cfg_.hidden * kv_keys_ * kv_bytes());
  }
  const float* kdq(int l) const { return kv_fp8_ ? kv_dq_ + (size_t)l * 2 * cfg_.hidden : nullptr; }
  const float* vdq(int l) const { return kv_fp8_ ? kv_dq_ + (size_t)l * 2 * cfg_.hidden + cfg_.hidden : nullptr; }
  DevBuf cross_probs_;                // [clips][layers][heads][Smax][Tcap] fp32 (capture_cross_)
  int cross_tcap_ = 0, cross_smax_ = 0;
  std::vector<int32_t> cross_counts_;  // tokens per clip (incl. BOS) of the captured decode

  // workspace (grow-only)
  void* pcm_pinned_ = nullptr;  // host staging for clips handed over in pageable memory (see encode())
  size_t pcm_pinned_cap_ = 0;
  DevBuf clips_d_, clip_ptrs_d_, pcm_stage_, audio_bf16_, row_pos_, row_clip_, x1_, x2_, H_, Y_, QKV_, VTe_, AO_, Z_,
      ENC_, ENC32_, gn_part_, gn_rows_, gn_stats_, gn_table_, KT_, VT_;
  int Smax_ = 0;

  // decode groups (own stream + buffers + captured step graph each); group 0 runs on stream_
  std::vector<std::unique_ptr<DecodeGroup>> groups_;
  hipEvent_t enc_done_ = nullptr;
  int dec_groups_ = 0;  // 0 = auto (MSH_DEC_GROUPS overrides)
  uint64_t ws_gen_ = 0;
  bool use_graph_ = true;

  // profiling
  bool prof_on_ = false;
  int step_only_ = -1;  // decode_step_enqueue: enqueue only this kernel group (profile_decode_chain)
  unsigned step_mask_ = 0;  // != 0: enqueue the kernel groups whose bit is set (profile_decode_chain, MSH_CHAIN_MASKS)
  struct ProfRec {
    int idx;
    hipEvent_t a, b;
  };
  std::vector<ProfEntry> prof_;
  std::map<std::string, int> prof_idx_;
  std::vector<ProfRec> prof_pending_;
  std::vector<hipEvent_t> event_pool_;
  hipEvent_t get_event();
  void prof_flush();
};

}  // namespace msh
