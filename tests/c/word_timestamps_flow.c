/* The flow of the reference's core/word-alignment-test.cpp ("non-streaming-transcribe-with-word-timestamps") as a plain C
 * program against include/moonshine-c-api.h + libmoonshine.so: load a transcriber with word_timestamps on, transcribe one
 * clip, check what that test REQUIREs of every word.  Differences: the model directory and the clip are synthetic (the
 * shipped tiny-en model and beckett.wav are not available; the WAV is read with msh_host_load_wav, this library's
 * load_wav_data), vad_threshold is 0, and "end > start" is relaxed to ">=" (random weights can put two words on one frame).
 * usage: word_timestamps_flow <model dir> <16-bit PCM wav> */
#include <stdio.h>
#include <stdlib.h>

#include "moonshine-c-api.h"
#include "moonshine_hip.h"

#define REQUIRE(c)                                                     \
  do {                                                                 \
    if (!(c)) {                                                        \
      fprintf(stderr, "REQUIRE failed at line %d: %s\n", __LINE__, #c); \
      return 1;                                                        \
    }                                                                  \
  } while (0)

int main(int argc, char** argv) {
  REQUIRE(argc == 3);
  struct moonshine_option_t options[] = {
      {"word_timestamps", "true"},
      {"identify_speakers", "false"},
      {"vad_threshold", "0"},
  };
  int32_t handle = moonshine_load_transcriber_from_files(argv[1], MOONSHINE_MODEL_ARCH_TINY, options, 3, moonshine_get_version());
  REQUIRE(handle >= 0);

  /* Load WAV file (reference: load_wav_data) */
  int32_t sample_rate = 0;
  int64_t count = msh_host_load_wav(argv[2], NULL, 0, &sample_rate);
  REQUIRE(count > 0);
  size_t n = (size_t)count;
  float* pcm = (float*)malloc(n * sizeof(float));
  REQUIRE(pcm != NULL);
  REQUIRE(msh_host_load_wav(argv[2], pcm, n, &sample_rate) == count);

  struct transcript_t* transcript = NULL;
  int32_t err = moonshine_transcribe_without_streaming(handle, pcm, n, sample_rate, 0, &transcript);
  REQUIRE(err == 0);
  REQUIRE(transcript != NULL);
  REQUIRE(transcript->line_count > 0);

  int total_words = 0;
  float prev_start = -1.0f;
  for (uint64_t i = 0; i < transcript->line_count; i++) {
    struct transcript_line_t* line = &transcript->lines[i];
    REQUIRE(line->word_count > 0);
    REQUIRE(line->words != NULL);
    for (uint64_t j = 0; j < line->word_count; j++) {
      const struct transcript_word_t* word = &line->words[j];
      REQUIRE(word->text != NULL);
      REQUIRE(word->end >= word->start);
      REQUIRE(word->start >= prev_start); /* monotonic */
      REQUIRE(word->confidence >= 0.0f);
      REQUIRE(word->confidence <= 1.0f);
      prev_start = word->start;
      total_words++;
    }
  }
  REQUIRE(total_words > 0);
  printf("words: %d, first '%s' %.3f-%.3f\n", total_words, transcript->lines[0].words[0].text, transcript->lines[0].words[0].start,
         transcript->lines[0].words[0].end);
  moonshine_free_transcriber(handle);
  free(pcm);
  return 0;
}
