// Export stubs for the part of the reference C API that is OUT OF SCOPE for this engine: text to
// speech, grapheme-to-phoneme, sentence embeddings, model catalogs / download manifests (reference
// core/moonshine-c-api.h:758-1258).  The reference's language bindings resolve every symbol of the header eagerly when
// they load the library (e.g. language-bindings/python/src/moonshine_voice/moonshine_api.py:972-1121), so a drop-in
// libmoonshine.so has to export them; each one reports MOONSHINE_ERROR_UNKNOWN ("not part of this build") and leaves
// its outputs empty.  Nothing here touches the GPU.
#include <stdlib.h>

#include "../../include/moonshine-c-api.h"
#include "host_utils.h"

namespace {
int32_t not_built(const char* what) {
  MSH_LOGF("%s is not part of the MI355X transcription build (only the transcriber API of moonshine-c-api.h is)", what);
  return MOONSHINE_ERROR_UNKNOWN;
}
template <class T>
void clear(T* p) {
  if (p != nullptr) *p = T();
}
}  // namespace

extern "C" {

int32_t moonshine_create_embedding_model(const char*, uint32_t, const char*) { return not_built("moonshine_create_embedding_model"); }
int32_t moonshine_create_embedding_model_from_memory(uint32_t, const char*, const char**, uint64_t, const uint8_t**, const uint64_t*,
                                                     const struct moonshine_option_t*, uint64_t, int32_t) {
  return not_built("moonshine_create_embedding_model_from_memory");
}
void moonshine_free_embedding_model(int32_t) {}
int32_t moonshine_calculate_embedding(int32_t, const char*, float** out_embedding, uint64_t* out_size, const char*) {
  clear(out_embedding);
  clear(out_size);
  return not_built("moonshine_calculate_embedding");
}
void moonshine_free_embedding(float* embedding) { free(embedding); }
int32_t moonshine_calculate_embedding_distance(int32_t, const float*, const float*, uint64_t, float* out_similarity) {
  clear(out_similarity);
  return not_built("moonshine_calculate_embedding_distance");
}
int32_t moonshine_extract_speech_clip(const float*, uint64_t, int32_t, int32_t, const struct moonshine_option_t*, uint64_t,
                                      struct moonshine_speech_clip_t* out_clip) {
  clear(out_clip);
  return not_built("moonshine_extract_speech_clip");
}
int32_t moonshine_create_tts_synthesizer_from_files(const char*, const char**, uint64_t, const struct moonshine_option_t*, uint64_t,
                                                    int32_t) {
  return not_built("moonshine_create_tts_synthesizer_from_files");
}
int32_t moonshine_create_tts_synthesizer_from_memory(const char*, const char**, const uint64_t, const uint8_t**, const uint64_t*,
                                                     const struct moonshine_option_t*, uint64_t, int32_t) {
  return not_built("moonshine_create_tts_synthesizer_from_memory");
}
void moonshine_free_tts_synthesizer(int32_t) {}
int32_t moonshine_get_g2p_dependencies(const char*, const struct moonshine_option_t*, uint64_t, char** out) {
  clear(out);
  return not_built("moonshine_get_g2p_dependencies");
}
int32_t moonshine_get_tts_dependencies(const char*, const struct moonshine_option_t*, uint64_t, char** out) {
  clear(out);
  return not_built("moonshine_get_tts_dependencies");
}
int32_t moonshine_get_tts_voices(const char*, const struct moonshine_option_t*, uint64_t, char** out) {
  clear(out);
  return not_built("moonshine_get_tts_voices");
}
int32_t moonshine_get_stt_dependencies(const char*, const struct moonshine_option_t*, uint64_t, char** out) {
  clear(out);
  return not_built("moonshine_get_stt_dependencies");
}
int32_t moonshine_get_embedding_dependencies(const char*, const struct moonshine_option_t*, uint64_t, char** out) {
  clear(out);
  return not_built("moonshine_get_embedding_dependencies");
}
int32_t moonshine_get_diarization_dependencies(char** out) {
  clear(out);
  return not_built("moonshine_get_diarization_dependencies");
}
int32_t moonshine_get_stt_catalog(char** out) {
  clear(out);
  return not_built("moonshine_get_stt_catalog");
}
int32_t moonshine_get_embedding_catalog(char** out) {
  clear(out);
  return not_built("moonshine_get_embedding_catalog");
}
int32_t moonshine_text_to_speech(int32_t, const char*, const struct moonshine_option_t*, uint64_t, float** out_audio, uint64_t* out_size,
                                 int32_t* out_rate) {
  clear(out_audio);
  clear(out_size);
  clear(out_rate);
  return not_built("moonshine_text_to_speech");
}
int32_t moonshine_phonemes_to_speech(int32_t, const char*, const struct moonshine_option_t*, uint64_t, float** out_audio,
                                     uint64_t* out_size, int32_t* out_rate) {
  clear(out_audio);
  clear(out_size);
  clear(out_rate);
  return not_built("moonshine_phonemes_to_speech");
}
int32_t moonshine_create_grapheme_to_phonemizer_from_files(const char*, const char**, uint64_t, const struct moonshine_option_t*, uint64_t,
                                                           int32_t) {
  return not_built("moonshine_create_grapheme_to_phonemizer_from_files");
}
int32_t moonshine_create_grapheme_to_phonemizer_from_memory(const char*, const char**, const uint64_t, const uint8_t**, const uint64_t*,
                                                            const struct moonshine_option_t*, uint64_t, int32_t) {
  return not_built("moonshine_create_grapheme_to_phonemizer_from_memory");
}
void moonshine_free_grapheme_to_phonemizer(int32_t) {}
int32_t moonshine_text_to_phonemes(int32_t, const char*, const struct moonshine_option_t*, uint64_t, const char** out_phonemes,
                                   uint64_t* out_count) {
  clear(out_phonemes);
  clear(out_count);
  return not_built("moonshine_text_to_phonemes");
}

}  // extern "C"
