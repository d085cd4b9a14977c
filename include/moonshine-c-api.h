/* moonshine-c-api.h -- public C API of the MI355X Moonshine transcriber library.
 *
 * Binary-compatible with the transcriber subset of the reference header
 * (reference core/moonshine-c-api.h:95-149 constants and option struct, :203-288 transcript structs,
 * :295-364 utility calls, :453-756 transcriber / stream calls): same symbol names, argument order,
 * struct layouts, handle and ownership rules, so a program or language binding built against the
 * reference header links and runs against this library unchanged.  The text-to-speech, embedding,
 * G2P and model-catalog entry points of the reference (h:758-1258) are different products and are
 * not provided here.
 *
 * What differs behind the surface: the model directory holds `model.safetensors` (HuggingFace
 * Moonshine tensor names) + `tokenizer.bin` instead of `.ort` graphs, inference runs on one MI355X,
 * and there are two additive entry points for utterance batches (end of this file).
 */
#ifndef MOONSHINE_C_API_H
#define MOONSHINE_C_API_H

#include <stddef.h>
#include <stdint.h>

#define MOONSHINE_EXPORT __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

/* MAJOR * 10000 + MINOR * 100 + PATCH of the header the caller was built against. */
#define MOONSHINE_HEADER_VERSION (30000)
#define MOONSHINE_FROM_MEMORY_REMOVED_VERSION (30000)

#define MOONSHINE_MODEL_ARCH_TINY (0)
#define MOONSHINE_MODEL_ARCH_BASE (1)
#define MOONSHINE_MODEL_ARCH_TINY_STREAMING (2)
#define MOONSHINE_MODEL_ARCH_BASE_STREAMING (3)
#define MOONSHINE_MODEL_ARCH_SMALL_STREAMING (4)
#define MOONSHINE_MODEL_ARCH_MEDIUM_STREAMING (5)

#define MOONSHINE_ERROR_NONE (0)
#define MOONSHINE_ERROR_UNKNOWN (-1)
#define MOONSHINE_ERROR_INVALID_HANDLE (-2)
#define MOONSHINE_ERROR_INVALID_ARGUMENT (-3)

#define MOONSHINE_FLAG_FORCE_UPDATE (1 << 0)
#define MOONSHINE_FLAG_SPELLING_MODE (1 << 1) /* accepted, no effect: no spelling model in this build */

struct moonshine_option_t {
  const char *name;  /* case-insensitive */
  const char *value; /* always a string; bools are true/false/1/0 */
};

struct transcript_word_t {
  const char *text;
  float start;
  float end;
  float confidence;
};

struct speaker_span_t {
  float start_time;
  float duration;
  uint64_t speaker_id;
  uint32_t speaker_index;
  uint64_t start_char;
  uint64_t end_char;
};

/* One phrase of speech.  Everything a line points to is owned by the transcriber and stays valid
 * until the next call on that transcriber / stream, or until it is freed. */
struct transcript_line_t {
  const char *text;           /* UTF-8, NULL when no model is loaded */
  const float *audio_data;    /* 16 kHz mono PCM of the line (when return_audio_data) */
  size_t audio_data_count;
  float start_time;           /* seconds from the start of the audio */
  float duration;
  uint64_t id;                /* stable identifier of the line */
  int8_t is_complete;
  int8_t is_updated;
  int8_t is_new;
  int8_t has_text_changed;
  int8_t have_speakers_changed;
  const struct speaker_span_t *speaker_spans; /* always NULL / 0 here (no diarizer) */
  uint64_t speaker_span_count;
  uint32_t last_transcription_latency_ms;
  const struct transcript_word_t *words;      /* word_timestamps option (offline architectures); NULL / 0 otherwise */
  uint64_t word_count;
};

struct transcript_t {
  struct transcript_line_t *lines;
  uint64_t line_count;
};

MOONSHINE_EXPORT int32_t moonshine_get_version(void);
MOONSHINE_EXPORT const char *moonshine_error_to_string(int32_t error);
MOONSHINE_EXPORT void moonshine_free_buffer(void *ptr);
MOONSHINE_EXPORT const char *moonshine_transcript_to_string(const struct transcript_t *transcript);

/* Keyterm biasing only exists for the streaming architectures; these return an error code for the
 * offline models, as the reference does. */
MOONSHINE_EXPORT int32_t moonshine_transcriber_set_keyterms(int32_t transcriber_handle, const char *keyterms);
MOONSHINE_EXPORT int32_t moonshine_transcriber_set_context(int32_t transcriber_handle, const char *context,
                                                           int32_t max_terms);

/* Load a transcriber from a directory holding model.safetensors + tokenizer.bin.  Returns a handle
 * (>= 0) or a negative error code.  Options recognised (all optional): vad_threshold (must be 0 in
 * this build: the Silero model is not included), vad_window_duration, vad_hop_size,
 * vad_look_behind_sample_count, vad_max_segment_duration, max_tokens_per_second,
 * transcription_interval, return_audio_data, log_api_calls, log_output_text, log_ort_run (per-kernel
 * timings), save_input_wav_path, skip_transcription, decode_incomplete_lines, and the additive
 * `device` (GPU index, default 0).  Unknown option names make the load fail. */
MOONSHINE_EXPORT int32_t moonshine_load_transcriber_from_files(const char *path, uint32_t model_arch,
                                                               const struct moonshine_option_t *options,
                                                               uint64_t options_count, int32_t moonshine_version);
/* Deprecated fixed-asset loader: refused for callers built against version >= 30000. */
MOONSHINE_EXPORT int32_t moonshine_load_transcriber_from_memory(
    const uint8_t *encoder_model_data, size_t encoder_model_data_size, const uint8_t *decoder_model_data,
    size_t decoder_model_data_size, const uint8_t *tokenizer_data, size_t tokenizer_data_size,
    const uint8_t *spelling_model_data, size_t spelling_model_data_size, uint32_t model_arch,
    const struct moonshine_option_t *options, uint64_t options_count, int32_t moonshine_version);
/* Keyed in-memory loader: filenames[i] in {"model.safetensors", "tokenizer.bin"}; a NULL / empty
 * buffer means "read that name as a path".  Buffers are copied to the GPU during the call. */
MOONSHINE_EXPORT int32_t moonshine_load_transcriber_from_memory_files(
    const char **filenames, const uint8_t **memory, const uint64_t *memory_sizes, uint64_t file_count,
    uint32_t model_arch, const struct moonshine_option_t *options, uint64_t options_count,
    int32_t moonshine_version);
MOONSHINE_EXPORT void moonshine_free_transcriber(int32_t transcriber_handle);

/* Transcribe a whole clip (PCM floats in [-1, 1], any sample rate, mono). */
MOONSHINE_EXPORT int32_t moonshine_transcribe_without_streaming(int32_t transcriber_handle, float *audio_data,
                                                                uint64_t audio_length, int32_t sample_rate,
                                                                uint32_t flags,
                                                                struct transcript_t **out_transcript);

MOONSHINE_EXPORT int32_t moonshine_create_stream(int32_t transcriber_handle, uint32_t flags);
MOONSHINE_EXPORT int32_t moonshine_free_stream(int32_t transcriber_handle, int32_t stream_handle);
MOONSHINE_EXPORT int32_t moonshine_start_stream(int32_t transcriber_handle, int32_t stream_handle);
MOONSHINE_EXPORT int32_t moonshine_stop_stream(int32_t transcriber_handle, int32_t stream_handle);
MOONSHINE_EXPORT int32_t moonshine_transcribe_add_audio_to_stream(int32_t transcriber_handle,
                                                                  int32_t stream_handle,
                                                                  const float *new_audio_data,
                                                                  uint64_t audio_length, int32_t sample_rate,
                                                                  uint32_t flags);
MOONSHINE_EXPORT int32_t moonshine_transcribe_stream(int32_t transcriber_handle, int32_t stream_handle,
                                                     uint32_t flags, struct transcript_t **out_transcript);

/* ---- additive: utterance batches (not in the reference, whose batch dimension is fixed at 1,
 * reference core/moonshine-model.cpp:247) ----
 * Transcribe `count` independent clips in one GPU batch.  out_transcripts[i] receives what
 * moonshine_transcribe_without_streaming would return for clip i; all of it is owned by the
 * transcriber and valid until its next batch call or until it is freed. */
MOONSHINE_EXPORT int32_t moonshine_transcribe_batch_without_streaming(
    int32_t transcriber_handle, const float *const *audio_data, const uint64_t *audio_lengths, uint64_t count,
    int32_t sample_rate, uint32_t flags, struct transcript_t **out_transcripts);
/* The same for 16-bit PCM, the format most recordings are stored in: sample value / 32768 is what the float entry point
 * would be handed, so the transcripts (and the lines' float audio_data) are identical to a call with those floats.  Why it
 * exists: a batch call moves every sample from host memory to the GPU once -- 64 KB per audio second as fp32, 32 KB as
 * 16-bit -- and at 60-85 k audio seconds per second and GPU that is 4-5 GB/s per GPU of host reads and PCIe traffic (eight
 * GPUs of one host: 40 GB/s).  With the default options on 16 kHz audio the clips cross PCIe at two bytes per sample and are
 * widened on the GPU; on every other path they are widened on the host first. */
MOONSHINE_EXPORT int32_t moonshine_transcribe_batch_without_streaming_pcm16(
    int32_t transcriber_handle, const int16_t *const *audio_data, const uint64_t *audio_lengths, uint64_t count,
    int32_t sample_rate, uint32_t flags, struct transcript_t **out_transcripts);

/* ---- out of scope for this engine, exported as stubs (reference core/moonshine-c-api.h:758-1258) ----
 * Sentence embeddings, text to speech, grapheme-to-phoneme, model catalogs and download manifests are not part of the
 * MI355X transcription build.  The reference's bindings bind every symbol of the header when they load the library
 * (language-bindings/python/src/moonshine_voice/moonshine_api.py:972-1121), so the symbols exist: each call returns
 * MOONSHINE_ERROR_UNKNOWN and clears its outputs.  Prototypes are the reference's, byte for byte. */
#define MOONSHINE_EMBEDDING_MODEL_ARCH_GEMMA_300M (0)
struct moonshine_speech_clip_t { /* reference h:833-851 */
  float *audio_data;
  uint64_t audio_length;
  float start_time;
  float speech_duration;
  int32_t is_complete;
  char *transcript;
};
MOONSHINE_EXPORT int32_t moonshine_create_embedding_model(const char *model_path, uint32_t model_arch,
                                                          const char *model_variant);
MOONSHINE_EXPORT int32_t moonshine_create_embedding_model_from_memory(
    uint32_t model_arch, const char *model_variant, const char **filenames, uint64_t filenames_count,
    const uint8_t **memory, const uint64_t *memory_sizes, const struct moonshine_option_t *options,
    uint64_t options_count, int32_t moonshine_version);
MOONSHINE_EXPORT void moonshine_free_embedding_model(int32_t embedding_model_handle);
MOONSHINE_EXPORT int32_t moonshine_calculate_embedding(int32_t embedding_model_handle, const char *sentence,
                                                       float **out_embedding, uint64_t *out_embedding_size,
                                                       const char *model_name);
MOONSHINE_EXPORT void moonshine_free_embedding(float *embedding);
MOONSHINE_EXPORT int32_t moonshine_calculate_embedding_distance(int32_t embedding_model_handle,
                                                                const float *embedding_a, const float *embedding_b,
                                                                uint64_t embedding_size, float *out_similarity);
MOONSHINE_EXPORT int32_t moonshine_extract_speech_clip(const float *audio_data, uint64_t audio_length,
                                                       int32_t sample_rate, int32_t tts_synthesizer_handle,
                                                       const struct moonshine_option_t *options,
                                                       uint64_t options_count, struct moonshine_speech_clip_t *out_clip);
MOONSHINE_EXPORT int32_t moonshine_create_tts_synthesizer_from_files(const char *language, const char **filenames,
                                                                     uint64_t filenames_count,
                                                                     const struct moonshine_option_t *options,
                                                                     uint64_t options_count, int32_t moonshine_version);
MOONSHINE_EXPORT int32_t moonshine_create_tts_synthesizer_from_memory(
    const char *language, const char **filenames, const uint64_t filenames_count, const uint8_t **memory,
    const uint64_t *memory_sizes, const struct moonshine_option_t *options, uint64_t options_count,
    int32_t moonshine_version);
MOONSHINE_EXPORT void moonshine_free_tts_synthesizer(int32_t tts_synthesizer_handle);
MOONSHINE_EXPORT int32_t moonshine_get_g2p_dependencies(const char *languages, const struct moonshine_option_t *options,
                                                        uint64_t options_count, char **out_dependencies_json);
MOONSHINE_EXPORT int32_t moonshine_get_tts_dependencies(const char *languages, const struct moonshine_option_t *options,
                                                        uint64_t options_count, char **out_dependencies_json);
MOONSHINE_EXPORT int32_t moonshine_get_tts_voices(const char *languages, const struct moonshine_option_t *options,
                                                  uint64_t options_count, char **out_voices_json);
MOONSHINE_EXPORT int32_t moonshine_get_stt_dependencies(const char *language, const struct moonshine_option_t *options,
                                                        uint64_t options_count, char **out_dependencies_json);
MOONSHINE_EXPORT int32_t moonshine_get_embedding_dependencies(const char *model_name,
                                                              const struct moonshine_option_t *options,
                                                              uint64_t options_count, char **out_dependencies_json);
MOONSHINE_EXPORT int32_t moonshine_get_diarization_dependencies(char **out_dependencies_json);
MOONSHINE_EXPORT int32_t moonshine_get_stt_catalog(char **out_catalog_json);
MOONSHINE_EXPORT int32_t moonshine_get_embedding_catalog(char **out_catalog_json);
MOONSHINE_EXPORT int32_t moonshine_text_to_speech(int32_t tts_synthesizer_handle, const char *text,
                                                  const struct moonshine_option_t *options, uint64_t options_count,
                                                  float **out_audio_data, uint64_t *out_audio_data_size,
                                                  int32_t *out_sample_rate);
MOONSHINE_EXPORT int32_t moonshine_phonemes_to_speech(int32_t tts_synthesizer_handle, const char *phonemes,
                                                      const struct moonshine_option_t *options, uint64_t options_count,
                                                      float **out_audio_data, uint64_t *out_audio_data_size,
                                                      int32_t *out_sample_rate);
MOONSHINE_EXPORT int32_t moonshine_create_grapheme_to_phonemizer_from_files(
    const char *language, const char **filenames, uint64_t filenames_count, const struct moonshine_option_t *options,
    uint64_t options_count, int32_t moonshine_version);
MOONSHINE_EXPORT int32_t moonshine_create_grapheme_to_phonemizer_from_memory(
    const char *language, const char **filenames, const uint64_t filenames_count, const uint8_t **memory,
    const uint64_t *memory_sizes, const struct moonshine_option_t *options, uint64_t options_count,
    int32_t moonshine_version);
MOONSHINE_EXPORT void moonshine_free_grapheme_to_phonemizer(int32_t grapheme_to_phonemizer_handle);
MOONSHINE_EXPORT int32_t moonshine_text_to_phonemes(int32_t grapheme_to_phonemizer_handle, const char *text,
                                                    const struct moonshine_option_t *options, uint64_t options_count,
                                                    const char **out_phonemes, uint64_t *out_phonemes_count);

#ifdef __cplusplus
}
#endif
#endif /* MOONSHINE_C_API_H */
