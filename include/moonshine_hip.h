/* moonshine_hip.h -- thin C ABI over the MI355X (gfx950) Moonshine engine.
 *
 * This is the seam that replaces the reference's ONNX Runtime glue: where the reference's
 * MoonshineModel creates two ORT sessions and calls OrtApi::Run per token
 * (reference core/moonshine-model.cpp:145-161 load, :270-274 encoder Run, :443-447 decoder Run,
 * core/ort-utils/ort-utils.cpp:37-81 session creation, :256-288 ort_run), the MI355X build calls the
 * functions below.  Plain C types only (pointers, sizes, int status); no C++ or torch types cross it.
 * Every function returns 0 on success and a negative msh_status on failure; the message for the last
 * failure on an engine is available from msh_last_error().  There is no CPU fallback: msh_create fails
 * when no HIP device is present.
 *
 * Threading: calls on one engine must be serialised by the caller (the Transcriber above holds the
 * same mutex the reference holds around MoonshineModel::transcribe, core/transcriber.cpp:1078).
 */
#ifndef MOONSHINE_HIP_H
#define MOONSHINE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSH_EXPORT __attribute__((visibility("default")))

typedef struct msh_engine msh_engine;

enum msh_status {
  MSH_OK = 0,
  MSH_ERR_UNKNOWN = -1,
  MSH_ERR_INVALID_ARGUMENT = -3,
  MSH_ERR_NO_DEVICE = -10,
  MSH_ERR_HIP = -11,
};

typedef struct msh_model_info {
  int32_t hidden, ffn, enc_layers, dec_layers, heads, head_dim, vocab, bos, eos;
  char arch[16];
} msh_model_info;

typedef struct msh_profile_entry {
  char name[48];
  double ms;         /* summed HIP-event time of this kernel group */
  uint64_t launches;
  double flops;      /* algorithmic flops summed over the launches */
  double bytes;      /* algorithmic HBM bytes summed over the launches */
} msh_profile_entry;

/* Library / device queries (msh_device_count never fails: 0 when no GPU or no driver). */
MSH_EXPORT int32_t msh_device_count(void);
MSH_EXPORT const char* msh_version(void);

/* Engine life cycle.  Replaces MoonshineModel::MoonshineModel / ~MoonshineModel
 * (reference core/moonshine-model.cpp:82-143). */
MSH_EXPORT int32_t msh_create(int32_t device, msh_engine** out);
MSH_EXPORT void msh_destroy(msh_engine* e);
MSH_EXPORT const char* msh_last_error(const msh_engine* e);

/* Weights: a safetensors blob with HuggingFace Moonshine tensor names.  model_arch: 0 tiny, 1 base
 * (MOONSHINE_MODEL_ARCH_*), -1 = take the dimensions from the file.  Replaces MoonshineModel::load /
 * load_from_memory (reference core/moonshine-model.cpp:145-161, :163-186). */
MSH_EXPORT int32_t msh_load_weights_file(msh_engine* e, const char* safetensors_path, int32_t model_arch);
MSH_EXPORT int32_t msh_load_weights_memory(msh_engine* e, const void* data, uint64_t size, int32_t model_arch);
MSH_EXPORT int32_t msh_model_info_get(const msh_engine* e, msh_model_info* out);

/* The checks of msh_load_weights_* without a GPU: parses the safetensors blob, validates every tensor the architecture
 * needs (name, shape, dtype F32 / F16 / BF16), an optional `proj_out.weight` (must BE the tied embedding) and refuses tensors
 * the loader does not know; *out receives the dimensions found.  On failure returns MSH_ERR_INVALID_ARGUMENT and writes the
 * reason to err (NUL-terminated, at most err_cap bytes).  What tools/verify_real_checkpoint.py runs first on a download. */
MSH_EXPORT int32_t msh_host_check_weights(const void* safetensors, uint64_t size, int32_t model_arch, msh_model_info* out,
                                          char* err, uint64_t err_cap);

/* Encoder (+ cross-attention K/V projection) over a batch of 16 kHz mono float clips.
 * pcm[i] points to n_samples[i] floats in host memory, or in device memory when pcm_on_device != 0.
 * Replaces the encoder ORT_RUN of reference core/moonshine-model.cpp:245-290, batched. */
MSH_EXPORT int32_t msh_encode(msh_engine* e, const float* const* pcm, const uint64_t* n_samples, uint32_t count,
                              int32_t pcm_on_device, float max_tokens_per_second);

/* Greedy decode of the batch encoded last.  Replaces the decode loop of reference
 * core/moonshine-model.cpp:380-517 (BOS start, first-max argmax, stop on EOS or after
 * ceil(duration * max_tokens_per_second) steps), run in lock-step over the batch.
 *   forced_steps <  0 : reference semantics.
 *   forced_steps >= 0 : ignore EOS, run exactly forced_steps steps (benchmarks, parity tests).
 *   teacher           : optional [count][teacher_stride] ids (row starts with BOS) fed instead of the
 *                       argmax of the previous step (teacher forcing for logit parity).
 *   logits_out        : optional [logit_steps][count][vocab] fp32, logits of the first logit_steps steps.
 *   tokens_out        : [count][tokens_stride] ids incl. BOS (and EOS when emitted), -1 padded;
 *                       tokens_stride >= steps + 1 where steps = forced_steps or msh_max_decode_steps().
 *   counts_out        : [count] number of valid ids per clip. */
MSH_EXPORT int32_t msh_decode(msh_engine* e, int32_t forced_steps, const int32_t* teacher, int32_t teacher_stride,
                              float* logits_out, int32_t logit_steps, int32_t* tokens_out, int32_t* counts_out,
                              int32_t tokens_stride);

/* encode + decode in one call: the batched equivalent of MoonshineModel::transcribe up to, not
 * including, detokenisation (reference core/moonshine-model.cpp:216-563). */
MSH_EXPORT int32_t msh_transcribe_tokens(msh_engine* e, const float* const* pcm, const uint64_t* n_samples,
                                         uint32_t count, int32_t pcm_on_device, float max_tokens_per_second,
                                         int32_t forced_steps, int32_t* tokens_out, int32_t* counts_out,
                                         int32_t tokens_stride);

/* Batch introspection after msh_encode. */
MSH_EXPORT int32_t msh_max_decode_steps(const msh_engine* e);           /* max over clips of the step budget */
MSH_EXPORT int32_t msh_clip_frames(const msh_engine* e, uint32_t clip); /* encoder frames T of a clip */
/* fp32 copy of last_hidden_state [T][hidden] of one clip; needs msh_set_keep_encoder_output(e, 1)
 * before msh_encode (test / debugging hook: the reference keeps it for word alignment,
 * core/moonshine-model.cpp:292-302). */
MSH_EXPORT int32_t msh_set_keep_encoder_output(msh_engine* e, int32_t keep);
MSH_EXPORT int32_t msh_get_encoder_output(msh_engine* e, uint32_t clip, float* out);

/* Storage of the cross-attention K / V the decoder streams every step (the `past_key_values.{l}.encoder.{key,value}` tensors
 * of the reference's decoder graph, core/moonshine-model.cpp:354-368, fp32 there): 0 = bf16 (default; the parity
 * tolerances of tests/ are stated for it), 1 = fp8 e4m3 with one scale per head-dim row fixed at load -- half the bytes of
 * the HBM-bound kernel that dominates a decode step; scores, softmax and accumulation stay fp32.  Applies from the next
 * msh_encode; call it before msh_set_batches_in_flight.  Not combinable with the cross-attention capture. */
MSH_EXPORT int32_t msh_set_kv_dtype(msh_engine* e, int32_t dtype);

/* Per-kernel-group timing with HIP events on the engine's stream (the role of the reference's
 * log_ort_run option, core/ort-utils/ort-utils.cpp:256-288).  While enabled the decode step runs
 * eagerly instead of from its hipGraph. */
MSH_EXPORT int32_t msh_profile_enable(msh_engine* e, int32_t on);
MSH_EXPORT int32_t msh_profile_reset(msh_engine* e);
MSH_EXPORT int32_t msh_profile_count(msh_engine* e);
MSH_EXPORT int32_t msh_profile_get(msh_engine* e, int32_t index, msh_profile_entry* out);

MSH_EXPORT int32_t msh_synchronize(msh_engine* e);
/* Average duration (ms) of an EMPTY profiling scope (two event records back to back on the engine stream): what every
 * per-launch figure of msh_profile_get carries on top of the kernel's own run time.  Negative on error. */
MSH_EXPORT double msh_profile_event_overhead_ms(msh_engine* e, int32_t iters);
/* Average ms per launch of the decode cross-attention kernel (the HBM-bound kernel that dominates a decode step),
 * launched back to back over the cross K/V of every layer of the batch encoded + decoded last, `rounds` sweeps between
 * one pair of HIP events: the kernel's own duration, free of per-launch event bookkeeping.  < 0 on error. */
MSH_EXPORT double msh_profile_cross_attention_ms(msh_engine* e, int32_t rounds);
/* Marginal cost of every decode kernel group the way the decode loop runs it: for each group a hipGraph of `reps` decode
 * steps holding ONLY that group's launches (same arguments as the real step of the batch decoded last) is replayed and
 * timed as a whole; the result is added to the profile table as entries named "chain_<group>" (ms / launches = us per
 * launch inside a dependent chain).  HIP-event scopes (msh_profile_enable) add ~4.8 us per launch and rocprofv3 reports
 * >= 4.3 us for an empty kernel, so neither resolves kernels of 2-6 us.  Leaves the decode state undefined. */
MSH_EXPORT int32_t msh_profile_decode_chain(msh_engine* e, int32_t reps);
/* ---- word timestamps: the decoder's cross-attention ----
 * Replaces the `cross_attentions.{l}` outputs of the reference's attention-exporting decoder graph
 * (decoder_with_attention.ort, core/moonshine-model.cpp:480-500, buffer layout :616-640).  With the capture on, every
 * msh_decode keeps the cross-attention probabilities of all layers, heads and steps (eager decode, one extra kernel per
 * layer and step).  msh_get_cross_attention: dims3 = {layers*heads, steps, frames} of clip `clip` (steps = ids generated,
 * frames = encoder frames T); if `out` holds cap_floats >= their product it receives the fp32 block
 * [layers*heads][steps][frames] -- the layout align_words takes.  Returns the element count or a negative msh error. */
MSH_EXPORT int32_t msh_set_capture_cross_attention(msh_engine* e, int32_t on);
MSH_EXPORT int64_t msh_get_cross_attention(msh_engine* e, uint32_t clip, float* out, uint64_t cap_floats, int32_t* dims3);

/* ---- batches in flight (additive; the reference serialises calls on processing_mutex, core/moonshine-model.cpp:229) ----
 * n lanes (1..8), each an engine with its own HIP stream, workspace and host thread, sharing this engine's weights: the
 * encoder of one batch runs inside the idle gaps of another batch's decode loop.  n = 0 tears the lanes down.  The
 * synchronous calls above keep using the engine itself and may be mixed with submitted batches. */
MSH_EXPORT int32_t msh_set_batches_in_flight(msh_engine* e, int32_t n);
/* Lanes overlap only when their HIP streams sit on different hardware queues, and HIP sizes its queue pool from the
 * environment variable GPU_MAX_HW_QUEUES (default 4) when the runtime initialises.  The library never touches the
 * environment by itself: either the host exports GPU_MAX_HW_QUEUES (>= lanes + 3; 8 is what bench.py uses) or it calls
 * this BEFORE the first GPU call of the process (msh_create, or any HIP call of the host).  Later calls have no effect
 * on an initialised runtime.  The transcriber load option `hw_queues` calls it. */
MSH_EXPORT int32_t msh_set_hw_queues(int32_t n);
/* Queue msh_transcribe_tokens for one batch; returns a ticket >= 0 or a negative msh error.  The clips and the output
 * arrays must stay valid until msh_wait(ticket) returns (the pointer / length arrays themselves are copied). */
MSH_EXPORT int64_t msh_submit_transcribe_tokens(msh_engine* e, const float* const* pcm, const uint64_t* n_samples,
                                                uint32_t count, int32_t on_device, float max_tokens_per_second,
                                                int32_t forced_steps, int32_t* tokens_out, int32_t* counts_out,
                                                int32_t tokens_stride);
/* Block until that batch is done; returns its status (msh_last_error has the message).  One wait per ticket. */
MSH_EXPORT int32_t msh_wait(msh_engine* e, int64_t ticket);

/* Form of the decoder's cross-attention (additive; the reference's graphs always project K and V).  ONE form per engine,
 * whatever the batch size -- a clip's token ids never depend on how many clips share its batch:
 *   1 (and 0, the default) = the projected K^T / V^T stream, the reference's form;
 *   2 = the "absorbed" form -- the key projection moved onto the query (qt_h = Wk_h^T q_h) and the value projection onto the
 *       output projection (Wo_h Wv_h), so that a decode step reads the encoder output ONCE per layer for all heads instead
 *       of K^T and V^T: half the bytes of the kernel that bounds batched decode and no cross-K/V projection in the encoder
 *       (k_xattn.hip).  It pays from ~192 clips per batch on (one workgroup per clip); the host layer's
 *       `cross_attention=auto` picks it at LOAD when the caller passed `batch_clips` / `max_batch_size` >= 192.
 * Mode 2 is an error (of this call, or of the next msh_encode) where it cannot be honoured: an architecture without the
 * absorbed operands (msh_cross_absorbed_supported), the word-timestamp capture, kv_dtype = fp8.  Applies to the next
 * msh_encode; set it before msh_set_batches_in_flight. */
MSH_EXPORT int32_t msh_set_cross_mode(msh_engine* e, int32_t mode);
/* Kernel set (additive).  0 (default): every call takes the kernels that are fastest at ITS size -- split-K encoder GEMMs for
 * a call of <= 1024 rows, tiled ones below 16 k rows, the panel / fused kernels above; decode kernels by clip count (1-2, <= 4,
 * 5..63, >= 64; LM head on the tiled kernel from 128 clips) -- so a clip's token ids can change (near-tie flips, same
 * tolerance) with the number of clips in its call.  1: ONE kernel set, the large-batch one, for every call: a clip's ids do
 * not depend on what shares its call -- the tail sub-batches of a batch call, one clip through a throughput deployment.
 * Small calls then run slower (one 10 s clip: 16.4 instead of 12.9 ms).  The host layer's `kernel_set=auto` switches it on at
 * LOAD together with the absorbed cross-attention form (batch_clips / max_batch_size >= 192 passed).  Applies to the next
 * msh_encode; set it before msh_set_batches_in_flight.  (reference: one graph per model, core/moonshine-model.cpp:270-274,
 * :443-447 -- the reference's result for a clip never depends on other clips.) */
MSH_EXPORT int32_t msh_set_uniform_kernels(msh_engine* e, int32_t on);
MSH_EXPORT int32_t msh_uniform_kernels(const msh_engine* e);
/* 1 if the batch encoded last decodes (with lanes: if the engine's batches decode) with the absorbed form, else 0. */
MSH_EXPORT int32_t msh_cross_absorbed(const msh_engine* e);
/* 1 if the loaded weights include the absorbed form's operands (8 heads, hidden 288 / 416), else 0. */
MSH_EXPORT int32_t msh_cross_absorbed_supported(const msh_engine* e);
/* ---- host-side byte / sample helpers of the transcription path, exported so that parity tests (and
 * bindings that want them) can call exactly the code the Transcriber runs.  No GPU involved. ----
 * msh_host_tokens_to_text : tokenizer.bin blob + ids -> text (reference core/bin-tokenizer/bin-tokenizer.cpp:406-426).
 *                           Returns the text length (excluding the terminating NUL), or a negative status;
 *                           at most out_cap-1 bytes are written.
 * msh_host_sanitize_utf8  : invalid UTF-8 bytes -> '?' (reference core/transcriber.cpp:1489-1543); same return rule.
 * msh_host_effective_cpus : CPUs the process may use (affinity mask cut down to the cgroup CPU quota): what the host
 *                           layer sizes its default thread counts with (option host_threads = 0)
 * msh_host_resample       : box-filter down / linear up (reference core/resampler.cpp:5-86); returns the output
 *                           sample count (call with out == NULL to size the buffer). */
MSH_EXPORT int64_t msh_host_tokens_to_text(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const int32_t* ids,
                                           uint64_t n_ids, char* out, uint64_t out_cap);
MSH_EXPORT int64_t msh_host_sanitize_utf8(const char* text, uint64_t n, char* out, uint64_t out_cap);
MSH_EXPORT int32_t msh_host_effective_cpus(void);
/* A sysfs CPU list ("0-63,128-191") as integers: the parser behind the NUMA pinning of the lanes' host threads when one
 * process drives several GPUs (csrc/host_utils.h pin_thread_to_gpu_node).  Returns the number of CPUs named; writes
 * min(that, cap) of them. */
MSH_EXPORT int64_t msh_host_parse_cpu_list(const char* text, int32_t* out, uint64_t cap);
MSH_EXPORT int64_t msh_host_resample(const float* in, uint64_t n, float in_rate, float out_rate, float* out,
                                     uint64_t out_cap);
/* msh_host_text_to_tokens : BinTokenizer::text_to_tokens (reference core/bin-tokenizer/bin-tokenizer.cpp:277-402);
 *                           bpe != 0 = byte-pair encoding (falls back to longest match without a byte block).
 *                           Returns the id count (ids beyond out_cap are dropped) or a negative status.
 * msh_host_biaser_bonuses : builds a ContextBiaser (reference core/context-biaser.cpp) from n_seqs token sequences
 *                           (flat ids + lengths), advances it over `prefix`, and adds its bonuses to out[0..vocab). */
MSH_EXPORT int64_t msh_host_text_to_tokens(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const char* text,
                                           uint64_t text_len, const char* space_marker, int32_t bpe, int32_t* out,
                                           uint64_t out_cap);
/* msh_host_context_terms   : ContextExtractor::extract (reference core/context-extractor.cpp:152-226) with the subword count
 *                           of the given tokenizer; the chosen terms come back joined by '\n'. */
MSH_EXPORT int64_t msh_host_context_terms(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const char* context,
                                          uint64_t context_len, int32_t max_terms, char* out, uint64_t out_cap);
/* WAV files either side of the path (reference core/moonshine-utils/debug-utils.cpp:52-190 load_wav_data, :192-250
 * save_wav_data -- what its tests, benchmark and the save_input_wav_path option use): 16-bit PCM only, samples / 32768,
 * the channel count is ignored (interleaved samples stay interleaved), a data chunk longer than the file is clamped.
 * msh_host_load_wav returns the sample count (copies min(count, out_cap) floats if out != NULL) or a negative error. */
MSH_EXPORT int64_t msh_host_load_wav(const char* path, float* out, uint64_t out_cap, int32_t* sample_rate);
MSH_EXPORT int32_t msh_host_save_wav(const char* path, const float* samples, uint64_t count, int32_t sample_rate);
/* Word alignment (reference core/word-alignment.cpp): the host code the transcriber runs on the device's cross-attention.
 * msh_host_dtw           : dtw (:12-88); returns the path length, indices into text_idx / time_idx (cap entries each).
 * msh_host_median_filter : median_filter (:98-153) in place over [rows][row_len].
 * msh_host_align_words   : align_words (:181-394) on att [heads_total][n_steps][frames]; word texts joined by '\n' into
 *                          text_out, (start, end, confidence) triples into times_out; returns the word count. */
MSH_EXPORT int64_t msh_host_dtw(const float* cost, int32_t n_text, int32_t n_time, int32_t* text_idx, int32_t* time_idx,
                                uint64_t cap);
MSH_EXPORT int32_t msh_host_median_filter(float* data, uint64_t rows, int32_t row_len, int32_t width);
MSH_EXPORT int64_t msh_host_align_words(const uint8_t* tokenizer_bin, uint64_t tokenizer_size, const float* att,
                                        int32_t heads_total, int32_t n_steps, int32_t frames, const int32_t* tokens,
                                        uint64_t n_tokens, float seconds_per_frame, char* text_out, uint64_t text_cap,
                                        float* times_out, uint64_t max_words);
/* Silero VAD on the device for batch calls (no reference counterpart: the reference runs the published ONNX model one hop
 * at a time on the host).  msh_silero_probabilities takes 16 kHz clips in HOST memory and writes, concatenated, the
 * probability of every whole 512-sample hop of every clip (clip i at offset sum_{j<i} n[j] / 512), each clip from a fresh
 * state -- what msh_host_silero_probabilities returns clip by clip, up to fp32 summation order.  Returns the number of
 * probabilities, or a negative msh error (cap too small: MSH_ERR_INVALID_ARGUMENT).
 * msh_silero_probabilities_keep_audio does the same and leaves the uploaded audio on the device: device_audio_out[i]
 * receives a DEVICE pointer to clip i's whole hops (n_samples[i] / 512 * 512 floats, the caller's samples verbatim; NULL for
 * a clip with no whole hop, or once the handle keeps 8 GiB), valid until msh_silero_release_audio / msh_silero_destroy.  It is
 * what msh_encode / msh_submit_transcribe_tokens take with on_device = 1: the batch call under the reference's default
 * vad_threshold (core/transcriber.cpp:656-696,997 segments every clip, then transcribes the segments) hands the engine
 * slices of it instead of sending the same PCM over PCIe a second time.  Calls on one handle are not thread-safe. */
typedef struct msh_silero msh_silero;
MSH_EXPORT int32_t msh_silero_create(int32_t device, const uint8_t* weights, uint64_t weights_size, msh_silero** out);
MSH_EXPORT void msh_silero_destroy(msh_silero* s);
MSH_EXPORT int64_t msh_silero_probabilities(msh_silero* s, const float* const* pcm, const uint64_t* n_samples, uint64_t count,
                                            float* probs_out, uint64_t cap);
MSH_EXPORT int64_t msh_silero_probabilities_keep_audio(msh_silero* s, const float* const* pcm, const uint64_t* n_samples,
                                                       uint64_t count, float* probs_out, uint64_t cap,
                                                       const float** device_audio_out);
/* The same in two halves, for a caller that has work of its own per chunk of clips (the batch call: the detectors' state
 * machines, the submission of the segments): msh_silero_submit stages one chunk -- at most 65536 whole hops unless it is a
 * single clip -- and enqueues its upload and network, returning a ticket (>= 0) or a negative msh error; msh_silero_collect
 * waits for it and writes the probabilities of its clips back to back (return value = their number) and, when
 * device_audio_out is given (`count` entries, the submission's clip count; only meaningful after keep_audio != 0), the
 * device pointers described above.  Two submissions may be outstanding -- chunk k + 1 is gathered and uploaded while the
 * network of chunk k runs --; tickets are collected in the order they were given. */
MSH_EXPORT int64_t msh_silero_submit(msh_silero* s, const float* const* pcm, const uint64_t* n_samples, uint64_t count,
                                     int32_t keep_audio);
/* msh_silero_submit for 16-bit PCM: the clips cross PCIe at two bytes per sample and become fp32 on the device (x / 32768,
 * exact -- the value the fp32 entry points would be given for the same recording); everything else as msh_silero_submit, the
 * kept audio (keep_audio != 0) is fp32. */
MSH_EXPORT int64_t msh_silero_submit_pcm16(msh_silero* s, const int16_t* const* pcm16, const uint64_t* n_samples, uint64_t count,
                                           int32_t keep_audio);
MSH_EXPORT int64_t msh_silero_collect(msh_silero* s, int64_t ticket, float* probs_out, uint64_t cap,
                                      const float** device_audio_out, uint64_t count);
MSH_EXPORT int32_t msh_silero_release_audio(msh_silero* s);
MSH_EXPORT const char* msh_silero_last_error(msh_silero* s);

/* Voice activity detection (reference core/silero-vad.cpp:78-173, core/voice-activity-detector.cpp:125-199), as the
 * Transcriber's streams run it -- host code, no GPU:
 * msh_host_silero_probabilities : Silero VAD over whole 512-sample hops of 16 kHz audio from a fresh state; weights = a
 *                                 safetensors blob (tools/convert_silero_vad.py); probs[i] = SileroVad::predict of hop i;
 *                                 state_out (nullable) receives the final [2][128] LSTM state.  Returns the hop count.
 * msh_host_vad_segments         : VoiceActivityDetector start / process_audio (in `chunk`-sample calls) / stop; weights
 *                                 may be NULL when threshold == 0.  bounds[3i..3i+2] = (start sample, sample count,
 *                                 is_complete) of segment i at 16 kHz.  Returns the segment count. */
MSH_EXPORT int64_t msh_host_silero_probabilities(const uint8_t* weights, uint64_t weights_size, const float* audio,
                                                 uint64_t n_samples, float* probs, uint64_t cap, float* state_out);
MSH_EXPORT int64_t msh_host_vad_segments(const uint8_t* weights, uint64_t weights_size, float threshold, int32_t window,
                                         int32_t hop, uint64_t look_behind, uint64_t max_segment, uint64_t hard_cap,
                                         const float* audio, uint64_t n_samples, int32_t sample_rate, uint64_t chunk,
                                         int64_t* bounds, uint64_t max_segments);
/* the same detector fed with precomputed Silero probabilities (one per whole hop of a 16 kHz clip): the host half of the
 * device-VAD path of batch calls (msh_silero_probabilities computes them on the GPU) */
MSH_EXPORT int64_t msh_host_vad_segments_from_probs(const uint8_t* weights, uint64_t weights_size, float threshold, int32_t window,
                                                    int32_t hop, uint64_t look_behind, uint64_t max_segment, uint64_t hard_cap,
                                                    const float* audio, uint64_t n_samples, const float* probs, uint64_t n_probs,
                                                    int64_t* bounds, uint64_t max_segments);
/* The plan of the batch call's rolling batch (no reference counterpart; csrc/rolling_plan.h): clips of lens[] (16 kHz samples)
 * arrive in pieces of piece_sizes[] clips; which sub-batch each clip goes out in and after which piece -- before the last
 * piece only full sub-batches of nearly one length or of the short class (the shortest clips holding short_frac of the first
 * piece's audio), the rest sorted like a whole batch.  sub_of_clip[i] = sub-batch of clip i; piece_of_sub[s] / first_of_sub[s]
 * (max_subs entries) = the piece after which sub-batch s was submitted / its first, longest clip.  Returns the number of
 * sub-batches.  Host code, no GPU. */
MSH_EXPORT int64_t msh_host_rolling_plan(const uint64_t* lens, const uint64_t* piece_sizes, uint64_t n_pieces, int32_t batch_clips,
                                         float short_frac, int32_t narrow_runs, int32_t* sub_of_clip, int32_t* piece_of_sub,
                                         int32_t* first_of_sub, uint64_t max_subs);
MSH_EXPORT int64_t msh_host_biaser_bonuses(const int32_t* flat_tokens, const int32_t* seq_lens, uint64_t n_seqs,
                                           float boost, const int32_t* prefix, uint64_t n_prefix, float* out,
                                           uint64_t vocab);

/* ---------------------------------------------------------------------------------------------------
 * Streaming models (reference core/moonshine-streaming-model.h:73-201).  One msh_stream_engine replaces the
 * five ORT sessions of MoonshineStreamingModel (frontend / encoder / adapter / cross_kv / decoder_kv) and
 * holds the per-stream state the reference keeps in MoonshineStreamingState (:36-71) on the device, one
 * "slot" per stream.  Every call takes a list of slots and processes them as one batch.
 *
 *   msh_stream_create         MoonshineStreamingModel ctor + load (:112-119); weights = safetensors with the
 *                             HuggingFace MoonshineStreamingForConditionalGeneration names, config = the
 *                             streaming_config.json text (lora/export.py:455-463; additive keys
 *                             "encoder_heads", "windows", "rope_theta", "partial_rotary_factor").
 *   msh_stream_open / _close / _reset     create_state (:181) / delete / MoonshineStreamingState::reset (:70)
 *   msh_stream_process_audio  process_audio_chunk (:145) for n streams; whole 320-sample periods are consumed,
 *                             the rest waits for the next call
 *   msh_stream_encode         encode (:149) for n streams (+ the cross K/V the reference computes lazily, :200)
 *   msh_stream_decoder_reset  decoder_reset (:178)
 *   msh_stream_decode_tokens  decode_tokens (:159) / decode_step (:153): logits rows of all streams concatenated
 *   msh_stream_decode_full    decode_full (:174) for n streams: drafts[i] (nullable) is stream i's speculative
 *                             draft; tokens_out[i * stride ...] / counts_out[i] get the content tokens (no BOS /
 *                             EOS), accepted_out[i] (nullable) the number of draft tokens kept;
 *                             max_tokens[i] < 0 (or a null array) = the reference's rule from the memory length
 *                             (:1217-1219), otherwise an explicit budget.
 * ------------------------------------------------------------------------------------------------- */
typedef struct msh_stream_engine msh_stream_engine;

typedef struct msh_stream_info {
  int32_t encoder_dim, decoder_dim, depth, nheads, head_dim, vocab_size, bos_id, eos_id;
  int32_t frame_len, total_lookahead, max_seq_len, enc_layers, encoder_heads;
  int32_t max_slots, memory_capacity; /* engine limits: concurrent streams, memory frames (20 ms) per stream */
} msh_stream_info;

MSH_EXPORT int32_t msh_stream_create(int32_t device, const char* safetensors_path, const char* config_json,
                                     int32_t max_slots, int32_t max_memory_frames, msh_stream_engine** out);
MSH_EXPORT int32_t msh_stream_create_from_memory(int32_t device, const void* safetensors, uint64_t size,
                                                 const char* config_json, int32_t max_slots,
                                                 int32_t max_memory_frames, msh_stream_engine** out);
MSH_EXPORT void msh_stream_destroy(msh_stream_engine* e);
MSH_EXPORT const char* msh_stream_last_error(const msh_stream_engine* e);
MSH_EXPORT int32_t msh_stream_info_get(const msh_stream_engine* e, msh_stream_info* out);
MSH_EXPORT int32_t msh_stream_open(msh_stream_engine* e);                 /* slot id >= 0, or a negative status */
MSH_EXPORT int32_t msh_stream_close(msh_stream_engine* e, int32_t slot);
MSH_EXPORT int32_t msh_stream_reset(msh_stream_engine* e, int32_t slot);
MSH_EXPORT int32_t msh_stream_process_audio(msh_stream_engine* e, int32_t n, const int32_t* slots,
                                            const float* const* pcm, const uint64_t* n_samples,
                                            int32_t* features_out);
MSH_EXPORT int32_t msh_stream_encode(msh_stream_engine* e, int32_t n, const int32_t* slots, const uint8_t* is_final,
                                     int32_t* new_frames_out);
MSH_EXPORT int32_t msh_stream_decoder_reset(msh_stream_engine* e, int32_t n, const int32_t* slots);
/* Word timestamps on the streaming architectures: the cross-attention probabilities of every decoder layer, head and
 * position for `tokens` fed from an EMPTY self cache (the slot's decoder is reset first and holds these tokens afterwards)
 * -- what the reference collects call by call from its decoder_kv_with_attention graph
 * (core/moonshine-streaming-model.cpp:946-1066) and rearranges for align_words (core/transcriber.cpp:1028-1068).
 * dims3 = {depth*heads, n_tokens, memory_len}; if `out` holds cap_floats >= their product it receives the fp32 block
 * [depth*heads][n_tokens][memory_len].  Returns the element count or a negative msh error. */
MSH_EXPORT int64_t msh_stream_cross_attention(msh_stream_engine* e, int32_t slot, const int32_t* tokens, int32_t n_tokens,
                                              float* out, uint64_t cap_floats, int32_t* dims3);
MSH_EXPORT int32_t msh_stream_decode_tokens(msh_stream_engine* e, int32_t n, const int32_t* slots,
                                            const int32_t* const* tokens, const int32_t* n_tokens,
                                            float* logits_out);
MSH_EXPORT int32_t msh_stream_decode_full(msh_stream_engine* e, int32_t n, const int32_t* slots,
                                          const int32_t* const* drafts, const int32_t* draft_lens,
                                          const int32_t* max_tokens, int32_t* tokens_out, int32_t* counts_out,
                                          int32_t tokens_stride, int32_t* accepted_out);
/* Contextual biasing (reference core/context-biaser.cpp:88-149, applied inside decode_full,
 * core/moonshine-streaming-model.cpp:1234-1246): a flat trie over token ids -- children of node n are entries
 * [child_off[n], child_off[n+1]) of child_tok / child_node, sorted by token; node 0 is the root; depth[n] its depth;
 * depth_bonus[d] the logit bonus of a depth-d token.  n_nodes == 0 switches biasing off.  The trie is copied. */
MSH_EXPORT int32_t msh_stream_set_bias(msh_stream_engine* e, int32_t n_nodes, const int32_t* child_off,
                                       const int32_t* child_tok, const int32_t* child_node, const int32_t* depth,
                                       const float* depth_bonus, int32_t n_depth_bonus);
/* Per-kernel-group timing of the streaming engine, as msh_profile_* above (groups: stream_frontend, senc_* window encoder,
 * stream_adapter_cross_kv, sver_* wide verify pass, sdec_* auto-regressive steps).  While enabled the AR steps run eagerly
 * instead of from their hipGraph. */
MSH_EXPORT int32_t msh_stream_profile_enable(msh_stream_engine* e, int32_t on);
MSH_EXPORT int32_t msh_stream_profile_reset(msh_stream_engine* e);
MSH_EXPORT int32_t msh_stream_profile_count(msh_stream_engine* e);
MSH_EXPORT int32_t msh_stream_profile_get(msh_stream_engine* e, int32_t index, msh_profile_entry* out);
/* state queries: 0 memory_len, 1 feature_count, 2 cache_len, 3 frames_emitted, 4 max_tokens for the memory;
 * engine-wide decode_full statistics since the last reset (slot ignored): 10 auto-regressive passes run, 11 wide (verify)
 * passes run, 12 / 13 microseconds of GPU time in AR loops / verify passes (HIP events, no extra synchronisation), 14 resets --
 * what a pass costs does not depend on the draft acceptance of the workload */
MSH_EXPORT int32_t msh_stream_query(const msh_stream_engine* e, int32_t slot, int32_t what);
MSH_EXPORT int32_t msh_stream_get_memory(msh_stream_engine* e, int32_t slot, float* out);   /* [memory_len][Dd] */
MSH_EXPORT int32_t msh_stream_get_features(msh_stream_engine* e, int32_t slot, float* out); /* [features][De] */

#ifdef __cplusplus
}
#endif
#endif /* MOONSHINE_HIP_H */
