#include "word_alignment.h"

#include <math.h>

#include <algorithm>
#include <limits>

namespace msh_host {

void dtw_path(const float* cost, int n_text, int n_time, std::vector<int>* text_idx, std::vector<int>* time_idx) {
  text_idx->clear();
  time_idx->clear();
  if (n_text <= 0 || n_time <= 0) return;
  const size_t W = (size_t)n_time + 1;
  std::vector<float> acc(((size_t)n_text + 1) * W, std::numeric_limits<float>::infinity());
  std::vector<uint8_t> came_from((size_t)n_text * n_time);  // 0 diagonal, 1 text only, 2 time only
  acc[0] = 0.f;
  for (int i = 0; i < n_text; ++i) {
    const float* above = &acc[(size_t)i * W];
    float* here = &acc[(size_t)(i + 1) * W];
    for (int j = 0; j < n_time; ++j) {
      const float diag = above[j], up = above[j + 1], left = here[j];
      uint8_t step;
      float best;
      if (diag <= up && diag <= left) {
        step = 0, best = diag;
      } else if (up <= diag && up <= left) {
        step = 1, best = up;
      } else {
        step = 2, best = left;
      }
      came_from[(size_t)i * n_time + j] = step;
      here[j + 1] = cost[(size_t)i * n_time + j] + best;
    }
  }
  for (int i = n_text - 1, j = n_time - 1;;) {
    text_idx->push_back(i);
    time_idx->push_back(j);
    if (i == 0 && j == 0) break;
    if (i < 0 || j < 0) break;  // cannot happen: row 0 / column 0 only have one finite predecessor
    const uint8_t step = came_from[(size_t)i * n_time + j];
    if (step != 2) --i;
    if (step != 1) --j;
  }
  std::reverse(text_idx->begin(), text_idx->end());
  std::reverse(time_idx->begin(), time_idx->end());
}

void median_filter_rows(float* data, size_t rows, int row_len, int width) {
  if (width <= 1 || row_len <= 0) return;
  width |= 1;
  const int half = width / 2;
  std::vector<float> ext((size_t)row_len + 2 * half), win(width), out(row_len);
  for (size_t r = 0; r < rows; ++r) {
    float* row = data + r * (size_t)row_len;
    for (int p = 0; p < half; ++p) {
      ext[p] = row[std::min(half - p, row_len - 1)];
      ext[half + row_len + p] = row[std::max(row_len - 2 - p, 0)];
    }
    std::copy(row, row + row_len, ext.begin() + half);
    for (int x = 0; x < row_len; ++x) {
      std::copy(ext.begin() + x, ext.begin() + x + width, win.begin());
      std::nth_element(win.begin(), win.begin() + half, win.end());
      out[x] = win[half];
    }
    std::copy(out.begin(), out.end(), row);
  }
}

namespace {
bool opens_word(const BinTokenizer& tk, int32_t id) {
  if (id < 0 || (size_t)id >= tk.vocab_size()) return false;
  const std::string& b = tk.token_bytes((size_t)id);
  return b.size() >= 3 && (uint8_t)b[0] == 0xE2 && (uint8_t)b[1] == 0x96 && (uint8_t)b[2] == 0x81;
}
std::string strip_ws(const std::string& s) {
  const char* ws = " \t\n\r";
  const size_t a = s.find_first_not_of(ws), b = s.find_last_not_of(ws);
  if (a == std::string::npos || b == std::string::npos) return s;  // all blank: left as is, like the reference (:330-335)
  return s.substr(a, b - a + 1);
}
}  // namespace

std::vector<TranscriberWord> align_words(const float* att, int heads_total, int n_steps, int frames,
                                         const std::vector<int32_t>& tokens, float seconds_per_frame,
                                         const BinTokenizer& tokenizer) {
  std::vector<TranscriberWord> words;
  if (att == nullptr || heads_total <= 0 || n_steps <= 0 || frames <= 0) return words;
  const size_t rows = (size_t)heads_total * n_steps;
  std::vector<float> z(att, att + rows * frames);
  for (size_t r = 0; r < rows; ++r) {  // z-score over the frames of one (head, step)
    float* v = &z[r * frames];
    float sum = 0.f;
    for (int f = 0; f < frames; ++f) sum += v[f];
    const float mean = sum / frames;
    float sq = 0.f;
    for (int f = 0; f < frames; ++f) {
      const float d = v[f] - mean;
      sq += d * d;
    }
    float sd = sqrtf(sq / frames);
    if (sd == 0.f) sd = 1e-10f;
    for (int f = 0; f < frames; ++f) v[f] = (v[f] - mean) / sd;
  }
  median_filter_rows(z.data(), rows, frames, 7);
  std::vector<float> neg((size_t)n_steps * frames, 0.f);
  for (int h = 0; h < heads_total; ++h)
    for (size_t i = 0; i < neg.size(); ++i) neg[i] += z[(size_t)h * neg.size() + i];
  const float inv = 1.0f / heads_total;
  for (float& x : neg) x = -(x * inv);
  std::vector<int> path_text, path_time;
  dtw_path(neg.data(), n_steps, frames, &path_text, &path_time);

  if (tokens.size() < 3) return words;  // BOS + last id only: no text token
  const size_t n_text = tokens.size() - 2;
  // row i of the attention matrix is the step that produced tokens[i + 1]
  std::vector<std::pair<size_t, size_t>> spans;  // [first, last] text-token index of each word
  for (size_t i = 0; i < n_text; ++i) {
    if (spans.empty() || opens_word(tokenizer, tokens[i + 1])) spans.emplace_back(i, i);
    spans.back().second = i;
  }
  for (const auto& sp : spans) {
    const std::string text = strip_ws(tokenizer.tokens_to_text(tokens.data() + 1 + sp.first, sp.second - sp.first + 1, true));
    if (text.empty()) continue;
    int lo = frames, hi = -1;
    for (size_t p = 0; p < path_text.size(); ++p)
      if ((size_t)path_text[p] >= sp.first && (size_t)path_text[p] <= sp.second) {
        lo = std::min(lo, path_time[p]);
        hi = std::max(hi, path_time[p]);
      }
    TranscriberWord w;
    w.text = text;
    if (hi >= 0) {
      w.start = lo * seconds_per_frame;
      w.end = (hi + 1) * seconds_per_frame;
    }
    words.push_back(w);
  }
  for (size_t i = 1; i < words.size(); ++i)
    if (words[i - 1].end > words[i].start) {
      const float mid = (words[i - 1].end + words[i].start) * 0.5f;
      words[i - 1].end = mid;
      words[i].start = mid;
    }
  return words;
}

}  // namespace msh_host
