// Host orchestration above the HIP engine: the MoonshineModel / Transcriber pair of the reference,
// re-implemented for the MI355X build.
//
//   MoonshineModel  mirrors reference core/moonshine-model.h:17-108 (load, load_from_memory, transcribe) --
//                   the struct the reference's Transcriber drives -- but runs on the engine of
//                   include/moonshine_hip.h instead of two ORT sessions, and adds transcribe_batch.
//   Transcriber     mirrors reference core/transcriber.h:231-366 for the offline architectures: VAD
//                   segmentation -> model call per just-updated segment -> transcript_t assembly, with
//                   the same line / flag / ownership semantics (core/transcriber.cpp:656-696, 775-891,
//                   989-1148, 1648-1726).  Segments of one call are decoded as ONE GPU batch.
#pragma once

#include <atomic>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/moonshine-c-api.h"
#include "../../include/moonshine_hip.h"
#include "context_extractor.h"
#include "host_text_vad.h"
#include "rolling_plan.h"
#include "streaming_model.h"
#include "word_alignment.h"

namespace msh_host {

struct MoonshineModel {
  msh_engine* engine = nullptr;          // devices[0]
  // utterance-batch data parallelism inside one process: one engine per GPU, every one holding
  // its own copy of the weights; a batch call shards its clips over them, one host thread per device, no collective
  struct DeviceShard {
    msh_engine* engine = nullptr;
    int device = 0;
    bool lanes_ready = false;
  };
  std::vector<DeviceShard> devices;
  BinTokenizer* tokenizer = nullptr;
  std::mutex processing_mutex;
  float max_tokens_per_second = 6.5f;  // reference core/moonshine-model.h:49
  bool log_ort_run = false;            // here: print per-kernel-group timings after each call
  // additive: a call with more than batch_clips clips is cut into sub-batches of that size and batches_in_flight of them
  // run on the GPU at once (include/moonshine_hip.h, msh_set_batches_in_flight); 1 = one sub-batch after the other
  int batch_clips = 256;
  int batches_in_flight = 2;
  // word timestamps (reference core/moonshine-model.cpp:600-645): with this on, transcribe_batch also keeps the device's
  // cross-attention and fills out_words[i] through align_words (times relative to the clip start)
  bool word_timestamps = false;
  std::string last_result;

  // device_ids: the GPUs to use (an id may repeat: two engines on one GPU, which is how the sharding is tested on a
  // single-GPU box)
  MoonshineModel(bool log_ort_run, float max_tokens_per_second, const std::vector<int>& device_ids);
  ~MoonshineModel();
  // model_type: MOONSHINE_MODEL_ARCH_TINY / _BASE.  Both return 0 on success (reference convention).
  int load(const char* weights_path, const char* tokenizer_path, int32_t model_type);
  int load_from_memory(const uint8_t* weights, size_t weights_size, const uint8_t* tokenizer_data,
                       size_t tokenizer_size, int32_t model_type);
  // One clip -> text; *out_text points into last_result and is valid until the next call
  // (reference core/moonshine-model.cpp:216-563).
  int transcribe(const float* audio, size_t n_samples, char** out_text);
  // Many clips -> texts in one GPU batch (no reference counterpart).
  int transcribe_batch(const std::vector<const float*>& audio, const std::vector<size_t>& n_samples,
                       std::vector<std::string>* out_texts, std::vector<std::vector<TranscriberWord>>* out_words = nullptr);
  std::string error() const;

  // The same batch, handed over in pieces while earlier pieces are already on the GPU (the batch call under the
  // reference's default VAD: segments keep arriving from the segmentation of later clips).  rolling_begin takes the
  // model; every rolling_add appends clips -- their indices continue from the previous add -- and submits sub-batches to
  // the devices' lanes as the plan of rolling_plan.h releases them (full sub-batches of one length or of the short clips
  // while pieces keep coming, the sorted rest on the `last` one); rolling_finish waits for everything and returns the texts in index order.
  // device_audio (nullable, entries nullable): the same samples already in the memory of `device_audio_gpu` -- a sub-batch
  // whose clips all have one and that runs on that GPU reads them there instead of uploading host_audio.
  // All three return 0 on success; after a failure rolling_finish still has to be called (it waits and releases).
  int rolling_begin();
  int rolling_add(const float* const* host_audio, const float* const* device_audio, int device_audio_gpu, const size_t* n_samples,
                  size_t count, bool last);
  int rolling_finish(std::vector<std::string>* out_texts);

 private:
  struct RollingSub {
    std::vector<uint32_t> idx;
    std::vector<const float*> pcm;
    std::vector<uint64_t> n;
    std::vector<int32_t> tokens, counts;
    int32_t stride = 0;
    int64_t ticket = -1;
    size_t dev = 0;
  };
  struct RollingClip {
    uint32_t idx;
    const float *host, *dev;
    uint64_t n;
  };
  struct Rolling {
    std::unique_lock<std::mutex> lock;
    std::unique_ptr<RollingPlanner> plan;   // which waiting clips go out when (rolling_plan.h)
    std::vector<RollingClip> clips;         // every clip handed over so far, by index
    std::deque<RollingSub> subs;            // submitted sub-batches (their arrays are written by the lanes: addresses must not move)
    size_t next_dev = 0;                    // tie-break of the device choice (rolling_submit)
    std::vector<double> assigned_ms;        // estimated GPU time handed to each device so far
    int device_audio_gpu = -1;
    bool failed = false;
  };
  std::unique_ptr<Rolling> rolling_;
  int rolling_submit(const RollingClip* clips, uint32_t m);
  // clips `idx` (indices into audio / lens) on one device: sub-batches of batch_clips, batches_in_flight of them at once;
  // ids[i] receives the token ids of clip i.  Returns 0 on success (the reference's status convention).
  int run_shard(DeviceShard& d, const std::vector<uint32_t>& idx, const std::vector<const float*>& audio,
                const std::vector<uint64_t>& lens, std::vector<std::vector<int32_t>>* ids);
};

struct TranscriberOptions {  // reference core/transcriber.h:129-229 (fields this build honours)
  enum ModelSource { FILES, MEMORY_FILES, NONE };
  ModelSource model_source = FILES;
  std::string model_path;
  uint32_t model_arch = (uint32_t)-1;
  std::map<std::string, std::pair<const uint8_t*, size_t>> memory_files;  // name -> client buffer
  float transcription_interval = 0.5f;
  float vad_threshold = 0.5f;
  float vad_window_duration = 0.5f;
  int32_t vad_hop_size = 512;
  size_t vad_look_behind_sample_count = 8192;
  float vad_max_segment_duration = 15.0f;
  std::string vad_model_path;            // additive: Silero VAD weights (safetensors); default <model_path>/silero_vad.safetensors
  float max_tokens_per_second = 6.5f;
  bool decode_incomplete_lines = true;
  bool use_speculative_decoding = true;  // streaming architectures (reference core/transcriber.h:191)
  std::vector<std::string> keyterms;     // contextual biasing (streaming architectures only)
  std::string context;                   // free-text passage the key terms are picked from (reference :context option)
  int32_t context_max_terms = 0;         // 0 = ContextExtractor::kDefaultMaxTerms
  float keyterm_boost = ContextBiaser::kDefaultBoost;
  int max_streams = 64;                  // additive: device slots for concurrent streaming lines
  int vad_device = 1;                    // additive: 1 = batch calls run the Silero network on the GPU (16 kHz input), 0 = host only
  int host_threads = 0;                  // additive: host threads for the per-clip VAD of batch calls (0 = twice the CPUs the process may use -- affinity and cgroup quota --, <= 128)
  float max_stream_seconds = 40.0f;      // additive: longest streaming line the device state is sized for
  bool word_timestamps = false;          // reference core/transcriber.h (word_timestamps): offline architectures here
  int kv_dtype = 0;                      // additive: storage of the decoder's cross K / V on the device: 0 = bf16, 1 = fp8 e4m3 (msh_set_kv_dtype)
  int cross_attention = 0;               // additive: form of the decoder's cross-attention (msh_set_cross_mode), ONE per transcriber, fixed at load: 0 = auto (absorbed when the caller ASKED for sub-batches of >= 192 clips -- batch_clips / max_batch_size passed -- and neither word_timestamps nor kv_dtype=fp8 is on; else projected K / V, the reference's order of operations and the form with the single-clip latency path), 1 = projected K / V, 2 = absorbed (an error where it cannot be honoured)
  int batch_clips = 256;                 // additive: clips per GPU sub-batch of a batch call
  bool batch_clips_given = false;        // the option was passed (the cross_attention=auto rule reads it)
  int kernel_set = 0;                    // additive: 0 = auto (the uniform set when the caller ASKED for sub-batches of >= 192 clips, like cross_attention=auto), 1 = per call (every call the kernels that are fastest at its size), 2 = uniform (ONE kernel set, the large-batch one: a clip's ids do not depend on what shares its call; msh_set_uniform_kernels)
  int batches_in_flight = 2;             // additive: sub-batches on the GPU at once (1 = strictly one after the other)
  bool return_audio_data = true;
  bool log_output_text = false;
  bool log_ort_run = false;
  std::string save_input_wav_path;
  int device = 0;
  int num_gpus = 1;                      // additive: GPUs device .. device + num_gpus - 1; -1 = every visible GPU
  std::vector<int> device_ids;           // additive: explicit GPU list ("devices" option), overrides device / num_gpus
};

struct TranscriberLine {
  bool has_text = false;
  std::string text;
  std::vector<float> audio;
  float start_time = 0.f, duration = 0.f;
  bool is_complete = false, just_updated = false, is_new = false, has_text_changed = false;
  uint64_t id = 0;
  uint32_t latency_ms = 0;
  std::vector<TranscriberWord> words;  // word_timestamps option; absolute times (segment start added)
};

struct TranscriptOutput {
  std::map<uint64_t, TranscriberLine> lines;
  std::vector<uint64_t> order;
  std::vector<transcript_line_t> c_lines;
  std::vector<std::vector<transcript_word_t>> c_words;  // per line, pointing into the lines' word texts
  transcript_t transcript{nullptr, 0};
  std::mutex mutex;
  void clear_update_flags();
  void mark_all_complete();
  void add_or_update(TranscriberLine& line);
  void rebuild();
};

struct TranscriberStream {
  std::unique_ptr<VoiceActivityDetector> vad;
  std::mutex vad_mutex;
  TranscriptOutput out;
  std::vector<float> new_audio;  // guarded by audio_mutex: add_audio (audio callback thread) vs transcribe_stream
  std::mutex audio_mutex;
  std::vector<float> saved_input;
  int32_t saved_rate = 0;
  int32_t id = -1;
  // streaming architectures: the state the reference keeps once per Transcriber
  // (core/transcriber.h:241,262-264) is kept per stream here, so streams do not evict each other
  MoonshineStreamingState* sstate = nullptr;
  uint64_t streaming_segment_id = UINT64_MAX;
  size_t streaming_samples_processed = 0;
  std::vector<int> last_streaming_tokens;
  MoonshineStreamingModel* sowner = nullptr;
  ~TranscriberStream() {
    if (sstate != nullptr && sowner != nullptr) sowner->free_state(sstate);
  }
};

class Transcriber {
 public:
  explicit Transcriber(const TranscriberOptions& options);
  ~Transcriber();
  void transcribe_without_streaming(const float* audio, uint64_t n, int32_t sample_rate, uint32_t flags,
                                    transcript_t** out);
  void transcribe_batch_without_streaming(const float* const* audio, const uint64_t* n, uint64_t count,
                                          int32_t sample_rate, uint32_t flags, transcript_t** out);
  // additive: the same for 16-bit PCM (sample value / 32768); with the device VAD the clips cross PCIe at two bytes per sample
  void transcribe_batch_without_streaming_pcm16(const int16_t* const* audio16, const uint64_t* n, uint64_t count, int32_t sample_rate,
                                                uint32_t flags, transcript_t** out);
  int32_t create_stream();
  void free_stream(int32_t id);
  void start_stream(int32_t id);
  void stop_stream(int32_t id);
  void add_audio_to_stream(int32_t id, const float* audio, uint64_t n, int32_t sample_rate);
  void transcribe_stream(int32_t id, uint32_t flags, transcript_t** out);
  static std::string transcript_to_string(const transcript_t* t);
  // reference core/transcriber.cpp:250-288: compile the key terms (each in its mid-sentence and utterance-initial
  // spelling) into the biasing trie; drops every stream's speculative draft
  void set_keyterms(const std::vector<std::string>& keyterms);
  // reference core/transcriber.cpp:217-248: pick the key terms out of a passage with ContextExtractor (a word
  // qualifies when the loaded tokenizer needs >= 2 subwords for it), then set_keyterms
  std::vector<std::string> keyterms_from_context(const std::string& context, int32_t max_terms);
  void set_context(const std::string& context, int32_t max_terms);

 private:
  void transcribe_batch_any(const float* const* audio, const int16_t* const* audio16, const uint64_t* n, uint64_t count,
                            int32_t sample_rate, uint32_t flags, transcript_t** out);
  TranscriberStream* new_stream(int32_t id);
  void load_streaming_model();
  std::shared_ptr<TranscriberStream> find_stream(int32_t id);  // the caller's copy keeps the stream alive against free_stream
  // transcribe every just-updated segment of `streams[i]` (all in one GPU batch), then rebuild outputs.  `segments` is the
  // caller's snapshot and is consumed: the audio of a segment moves into its transcript line (no copy per line)
  // given_texts (offline models; nullable): the texts of the pass's jobs, in job order -- stream by stream, segment by
  // segment, every segment is_offline_job() accepts --, already computed (the rolling batch call); the model is not called
  void update_from_segments(const std::vector<TranscriberStream*>& streams, std::vector<std::vector<VadSegment>>& segments,
                            transcript_t** outs, const std::vector<std::string>* given_texts = nullptr,
                            uint32_t given_latency_ms = 0);
  // a segment an offline model transcribes in this pass (reference core/transcriber.cpp:1075-1100)
  bool is_offline_job(const VadSegment& seg) const {
    return seg.just_updated && model_ != nullptr && (seg.is_complete || opt_.decode_incomplete_lines) &&
           seg.audio.size() >= 895;  // shorter than the conv stem's receptive field: empty text
  }
  void save_input(TranscriberStream* s, const float* audio, uint64_t n, int32_t rate, bool flush);
  struct StreamingJob {
    TranscriberStream* stream;
    const VadSegment* segment;
    uint64_t line_id;
    std::string text;
    std::vector<TranscriberWord> words;  // word_timestamps option: times relative to the segment start
  };
  // Transcriber::transcribe_segment_with_streaming_model (reference core/transcriber.cpp:1311-1487) for one
  // segment of each of several streams, as one GPU batch
  void transcribe_segments_with_streaming_model(std::vector<StreamingJob>& jobs);

  void load_vad_model();
  TranscriberOptions opt_;
  std::shared_ptr<const SileroWeights> silero_;  // shared by every stream's detector (each keeps its own state)
  std::vector<uint8_t> silero_blob_;             // the weights file as read: what the device network is created from
  msh_silero* silero_device_ = nullptr;          // batch calls: the network on the GPU (option vad_device, default on)
  bool silero_device_failed_ = false;
  size_t vad_hard_cap_ = 0;                      // longest segment (samples) the engine behind this transcriber takes
  std::unique_ptr<MoonshineModel> model_;
  std::unique_ptr<MoonshineStreamingModel> streaming_model_;   // device 0 of the list (tokenizer, config, the biaser's owner)
  // streaming architectures on more than one GPU (options num_gpus / devices): one model (engine + its slots, own copy of
  // the weights) per further device.  A stream's line state lives on ONE device -- the one with the fewest lines when its
  // first line starts -- so streams shard as clips do: no collective, per-device batches on per-device host threads.
  std::vector<std::unique_ptr<MoonshineStreamingModel>> streaming_more_;
  std::vector<MoonshineStreamingModel*> streaming_models() {
    std::vector<MoonshineStreamingModel*> v;
    if (streaming_model_) v.push_back(streaming_model_.get());
    for (auto& m : streaming_more_) v.push_back(m.get());
    return v;
  }
  void transcribe_segments_on_model(MoonshineStreamingModel* m, std::vector<StreamingJob*>& jobs);
  ContextBiaser context_biaser_;
  std::mutex context_biaser_mutex_;  // taken before model_mutex_ (reference core/transcriber.cpp:1394-1396)
  std::mutex model_mutex_, batch_mutex_, streams_mutex_;
  std::unique_ptr<TranscriberStream> batch_stream_;
  std::vector<std::unique_ptr<TranscriberStream>> batch_streams_;  // one per clip of the last batch call
  std::thread batch_retire_;   // frees the previous batch call's streams beside the current call (guarded by batch_mutex_)
  std::map<int32_t, std::shared_ptr<TranscriberStream>> streams_;
  std::atomic<uint64_t> next_line_id_;
  int32_t next_stream_id_ = 1;
};

}  // namespace msh_host
