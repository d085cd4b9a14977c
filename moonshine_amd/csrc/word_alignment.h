// Word timestamps from the decoder's cross-attention (host side).  Same rules as the reference's
// core/word-alignment.{h,cpp}: per-(head, step) z-score over the encoder frames, 7-wide median along the frames, mean over
// heads, DTW on the negated matrix, words cut at the SentencePiece marker U+2581, one frame span per word, overlapping
// neighbours snapped to their midpoint.  The attention itself comes from the device (msh_get_cross_attention).
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

#include "host_text_vad.h"

namespace msh_host {

struct TranscriberWord {  // reference core/word-alignment.h:9-14
  std::string text;
  float start = 0.f;  // seconds
  float end = 0.f;
  float confidence = 1.f;
};

// Cheapest monotone path through cost [n_text][n_time] from (0, 0) to the last cell (reference :12-88; on ties the
// diagonal step wins, then the step that only moves in text, then the one that only moves in time).
void dtw_path(const float* cost, int n_text, int n_time, std::vector<int>* text_idx, std::vector<int>* time_idx);

// In-place median of `width` (made odd) neighbours along the last axis of [rows][width_of_row]; the borders are
// mirrored without repeating the edge sample (reference :98-153).
void median_filter_rows(float* data, size_t rows, int row_len, int width);

// att: [heads_total][n_steps][frames] fp32; tokens: BOS, generated ids ..., last id (EOS or the id the budget cut at);
// the first and the last id carry no word (reference :285-296).
std::vector<TranscriberWord> align_words(const float* att, int heads_total, int n_steps, int frames,
                                         const std::vector<int32_t>& tokens, float seconds_per_frame,
                                         const BinTokenizer& tokenizer);

}  // namespace msh_host
