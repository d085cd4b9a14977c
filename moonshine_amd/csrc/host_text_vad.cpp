#include "host_text_vad.h"

#include <algorithm>
#include <stdexcept>

#include "host_utils.h"

namespace msh_host {

// ---------------------------------------------------------------------------------------------
BinTokenizer::BinTokenizer(const uint8_t* data, size_t size, const std::string& space_marker) : space_(space_marker) {
  size_t i = 0;
  while (i < size) {
    const uint8_t first = data[i++];
    if (first == 0) {  // empty entry
      tokens_.emplace_back();
      continue;
    }
    size_t n = first;
    if (first >= 128) {  // two-byte length: first = (len % 128) + 128, second = len / 128
      if (i >= size) throw std::runtime_error("tokenizer.bin: truncated length prefix");
      n = (size_t)data[i++] * 128 + first - 128;
    }
    if (i + n > size) throw std::runtime_error("tokenizer.bin: truncated token bytes");
    tokens_.emplace_back(reinterpret_cast<const char*>(data + i), n);
    i += n;
  }
  if (tokens_.empty()) throw std::runtime_error("tokenizer.bin: no tokens found");
  build_indexes();
}

void BinTokenizer::build_indexes() {
  by_first_byte_.assign(256, {});
  for (size_t i = 0; i < tokens_.size(); ++i)
    if (!tokens_[i].empty()) by_first_byte_[(uint8_t)tokens_[i][0]].push_back((int32_t)i);
  // byte fallback block: 256 consecutive single-byte entries 0x00..0xFF (searched, not assumed at a fixed id)
  byte_base_ = -1;
  for (size_t start = 0; start + 256 <= tokens_.size() && byte_base_ < 0; ++start) {
    bool complete = true;
    for (size_t o = 0; o < 256 && complete; ++o)
      complete = tokens_[start + o].size() == 1 && (uint8_t)tokens_[start + o][0] == o;
    if (complete) byte_base_ = (int32_t)start;
  }
  if (byte_base_ < 0) return;
  // only entries after the byte block can be produced by a merge; the first spelling wins
  for (size_t i = (size_t)byte_base_ + 256; i < tokens_.size(); ++i)
    if (!tokens_[i].empty()) merge_ids_.emplace(tokens_[i], (int32_t)i);
}

std::vector<int32_t> BinTokenizer::text_to_tokens(const std::string& text, bool bpe) const {
  return (bpe && byte_base_ >= 0) ? encode_bpe(text) : encode_longest_match(text);
}

std::vector<int32_t> BinTokenizer::encode_longest_match(const std::string& text) const {
  std::vector<int32_t> out;
  const std::string s = replace_all(text, " ", space_);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t best_len = 0;
    int32_t best = -1;
    for (const int32_t i : by_first_byte_[(uint8_t)s[pos]]) {
      const std::string& t = tokens_[i];
      if (s.size() - pos < t.size()) continue;
      if (t.size() > best_len && s.compare(pos, t.size(), t) == 0) {  // strictly longer: the lowest id keeps a tie
        best_len = t.size();
        best = i;
      }
    }
    if (best < 0) throw std::runtime_error("No match found for remaining bytes " + s.substr(pos));
    out.push_back(best);
    pos += best_len;
  }
  return out;
}

std::vector<int32_t> BinTokenizer::encode_bpe(const std::string& text) const {
  const std::string s = replace_all(text, " ", space_);
  auto seq_len = [](uint8_t lead) -> size_t {
    if ((lead & 0x80) == 0x00) return 1;
    if ((lead & 0xE0) == 0xC0) return 2;
    if ((lead & 0xF0) == 0xE0) return 3;
    if ((lead & 0xF8) == 0xF0) return 4;
    return 1;
  };
  std::vector<std::string> pieces;
  for (size_t off = 0; off < s.size();) {
    const size_t n = std::min(seq_len((uint8_t)s[off]), s.size() - off);
    pieces.push_back(s.substr(off, n));
    off += n;
  }
  std::string cand;
  while (pieces.size() > 1) {
    int32_t best_id = -1;
    size_t best_pos = 0;
    for (size_t p = 0; p + 1 < pieces.size(); ++p) {
      cand.assign(pieces[p]);
      cand.append(pieces[p + 1]);
      const auto f = merge_ids_.find(cand);
      if (f != merge_ids_.end() && (best_id < 0 || f->second < best_id)) {
        best_id = f->second;
        best_pos = p;
      }
    }
    if (best_id < 0) break;
    pieces[best_pos].append(pieces[best_pos + 1]);
    pieces.erase(pieces.begin() + best_pos + 1);
  }
  std::vector<int32_t> out;
  for (const std::string& piece : pieces) {
    const auto f = merge_ids_.find(piece);
    if (f != merge_ids_.end()) {
      out.push_back(f->second);
      continue;
    }
    for (const char b : piece) out.push_back(byte_base_ + (int32_t)(uint8_t)b);
  }
  return out;
}

BinTokenizer* BinTokenizer::from_file(const std::string& path) {
  std::vector<uint8_t> blob;
  if (!read_file(path, &blob)) throw std::runtime_error("Failed to open tokenizer file at " + path);
  return new BinTokenizer(blob.data(), blob.size());
}

std::string BinTokenizer::tokens_to_text(const int32_t* ids, size_t count, bool skip_specials) const {
  std::string bytes;
  for (size_t k = 0; k < count; ++k) {
    const int32_t id = ids[k];
    if (id < 0 || (size_t)id >= tokens_.size()) throw std::out_of_range("token id " + std::to_string(id) + " out of range");
    const std::string& t = tokens_[id];
    if (t.empty()) throw std::runtime_error("Invalid token " + std::to_string(id));
    if (skip_specials && t.size() > 2 && t.front() == '<' && t.back() == '>') continue;
    bytes += t;
  }
  return trim(replace_all(bytes, space_, " "));
}

// ---------------------------------------------------------------------------------------------
std::string sanitize_utf8(const std::string& text) {
  std::string out;
  out.reserve(text.size());
  const size_t n = text.size();
  auto cont = [&](size_t j) { return (((uint8_t)text[j]) & 0xC0) == 0x80; };
  size_t i = 0;
  while (i < n) {
    const uint8_t c = (uint8_t)text[i];
    size_t need = 0;
    if (c < 0x80) need = 1;
    else if ((c & 0xE0) == 0xC0) need = 2;
    else if ((c & 0xF0) == 0xE0) need = 3;
    else if ((c & 0xF8) == 0xF0) need = 4;
    bool ok = need != 0 && n - i >= need;
    for (size_t k = 1; ok && k < need; ++k) ok = cont(i + k);
    if (ok) {
      out.append(text, i, need);
      i += need;
    } else {
      out.push_back('?');
      i += 1;
    }
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
VoiceActivityDetector::VoiceActivityDetector(float threshold, int32_t window_size, int32_t hop_size, size_t look_behind,
                                             size_t max_segment, std::shared_ptr<const SileroWeights> silero,
                                             size_t hard_cap)
    : threshold_(threshold), hop_(hop_size), look_behind_(look_behind), max_segment_(max_segment), hard_cap_(hard_cap) {
  if (hop_size <= 0) throw std::runtime_error("vad_hop_size must be positive");
  if (threshold > 0.0f) {
    if (!silero)
      throw std::runtime_error(
          "vad_threshold > 0 needs the Silero VAD weights: pass the option vad_model_path=<silero_vad.safetensors> "
          "(tools/convert_silero_vad.py writes it from the published model) or put silero_vad.safetensors into the "
          "model directory; vad_threshold=0 treats all audio as speech");
    if (hop_size != SileroVad::kHop)
      throw std::runtime_error("vad_hop_size must be 512 when vad_threshold > 0 (the Silero model's window)");
    silero_.reset(new SileroVad(std::move(silero)));
  }
  prob_window_.assign((size_t)(window_size > 0 ? window_size : 1), 0.f);
  look_buf_.assign(look_behind_, 0.f);
}

void VoiceActivityDetector::start() {
  active_ = true;
  processed_ = 0;
  segments_.clear();
  remainder_.clear();
  look_buf_.assign(look_behind_, 0.f);
  prev_voice_ = false;
  forced_cut_ = false;
  prob_window_.assign(prob_window_.size(), 0.f);
  prob_index_ = 0;
  if (silero_) silero_->reset();
}

void VoiceActivityDetector::stop() {
  active_ = false;
  if (prev_voice_ && !segments_.empty()) {  // close the open segment as it stands (the sub-hop tail is dropped)
    VadSegment& s = segments_.back();
    s.end_time = (float)processed_ / kSampleRate;
    s.is_complete = true;
    s.just_updated = true;
  }
}

void VoiceActivityDetector::process_audio(const float* audio, size_t count, int32_t sample_rate, const float* silero_probs,
                                          size_t n_probs) {
  if (!active_) return;
  if (silero_probs != nullptr &&
      (!silero_ || sample_rate != kSampleRate || !remainder_.empty() || n_probs != count / (size_t)hop_))
    throw std::runtime_error("precomputed VAD probabilities need 16 kHz audio on a hop boundary and one value per whole hop");
  for (VadSegment& s : segments_) s.just_updated = false;
  std::vector<float> resampled;
  const float* src = audio;
  size_t n = count;
  if (sample_rate != kSampleRate) {
    resampled = resample(std::vector<float>(audio, audio + count), (float)sample_rate, (float)kSampleRate);
    src = resampled.data();
    n = resampled.size();
  }
  // whole hops only; the sub-hop tail waits in remainder_ for the next call.  Hops are read in place (no staging copy), and
  // so is the look-behind: `region_` is the run of consecutive samples of THIS call that ends with the hop being
  // processed; look_buf_ holds what came before it and is brought up to date once per run (sliding it on every hop was a
  // 32 KB memmove per 2 KB hop: most of the detector's time with the network on the GPU).
  size_t off = 0;
  if (!remainder_.empty()) {
    const size_t take = std::min((size_t)hop_ - remainder_.size(), n);
    remainder_.insert(remainder_.end(), src, src + take);
    off = take;
    if (remainder_.size() < (size_t)hop_) return;
    call_remaining_ = n - off;
    region_ = remainder_.data(), region_len_ = (size_t)hop_;
    process_hop(remainder_.data());
    append_history(remainder_.data(), (size_t)hop_);
    region_ = nullptr, region_len_ = 0;
    remainder_.clear();
  }
  // Silero probabilities of all the whole hops of this call in one go (the network's state-independent part runs for
  // eight hops at a time, silero_vad.h predict_many; per hop the value is the one predict() would return)
  const size_t whole = (n - off) / (size_t)hop_;
  if (silero_ && whole > 0) {
    if (silero_probs != nullptr) {
      hop_probs_.assign(silero_probs, silero_probs + whole);
    } else {
      hop_probs_.resize(whole);
      silero_->predict_many(src + off, whole, hop_probs_.data());
    }
  }
  const size_t run0 = off;
  for (size_t i = 0; i < whole; ++i) {
    call_remaining_ = n - off - hop_;
    region_ = src + run0, region_len_ = off - run0 + (size_t)hop_;
    process_hop(src + off, silero_ ? &hop_probs_[i] : nullptr);
    off += hop_;
  }
  if (off > run0) append_history(src + run0, off - run0);
  region_ = nullptr, region_len_ = 0;
  call_remaining_ = 0;
  remainder_.assign(src + off, src + n);
}

// look_buf_ := the last look_behind_ samples of (look_buf_ | p[0 .. count))
void VoiceActivityDetector::append_history(const float* p, size_t count) {
  const size_t cap = look_buf_.size();
  if (cap == 0) return;
  if (count >= cap) {
    std::copy(p + count - cap, p + count, look_buf_.begin());
  } else {
    std::move(look_buf_.begin() + (long)count, look_buf_.end(), look_buf_.begin());
    std::copy(p, p + count, look_buf_.end() - (long)count);
  }
}

void VoiceActivityDetector::clear_completed_audio() {
  for (VadSegment& s : segments_)
    if (s.is_complete) std::vector<float>().swap(s.audio);
}

void VoiceActivityDetector::process_hop(const float* hop, const float* silero_prob) {
  processed_ += hop_;
  // threshold 0: probability 1; otherwise Silero's probability averaged over the last `window` hops (a ring that
  // starts at zero, reference :139-151).  Either is scaled by the max-length fade once the segment passes 2/3 of the cap.
  float p = 1.0f;
  if (silero_) {
    prob_window_[prob_index_] = silero_prob != nullptr ? *silero_prob : silero_->predict(hop);
    prob_index_ = (prob_index_ + 1) % prob_window_.size();
    float sum = 0.0f;  // std::accumulate(..., 0.0f) in the reference: fp32, in ring order
    for (float v : prob_window_) sum += v;
    p = sum / (float)prob_window_.size();
  }
  const size_t fade = (max_segment_ * 2) / 3;
  const size_t cur = open_size();
  if (max_segment_ && cur > fade) p = p * ((float)(cur - fade) / (float)fade);
  bool voice = p > threshold_;
  bool cut = false;
  if (voice && prev_voice_ && hard_cap_ && cur + 2 * (size_t)hop_ > hard_cap_) voice = false, cut = true;  // engine capacity
  const float now = (float)processed_ / kSampleRate;
  if (voice && !prev_voice_) {
    const size_t lb = forced_cut_ ? std::min((size_t)hop_, look_buf_.size()) : std::min(look_behind_, processed_);
    forced_cut_ = false;
    VadSegment s;
    // one allocation for what this call can still add (bounded by the caps), instead of regrowing hop by hop
    size_t want = lb + call_remaining_;
    if (max_segment_) want = std::min(want, max_segment_ + (size_t)hop_);
    if (hard_cap_) want = std::min(want, hard_cap_);
    s.audio.reserve(std::max(want, lb));
    // the last lb samples up to the end of this hop: from the run this hop ends, and what it lacks from the history before it
    if (lb <= region_len_) {
      s.audio.assign(region_ + region_len_ - lb, region_ + region_len_);
    } else {
      s.audio.assign(look_buf_.end() - (long)(lb - region_len_), look_buf_.end());
      s.audio.insert(s.audio.end(), region_, region_ + region_len_);
    }
    s.src_offset = processed_ - lb;
    s.start_time = now - (float)s.audio.size() / kSampleRate;
    s.end_time = now;
    s.just_updated = true;
    segments_.push_back(std::move(s));
  } else if (!voice && prev_voice_) {
    VadSegment& s = segments_.back();
    s.audio.insert(s.audio.end(), hop, hop + hop_);
    s.end_time = now;
    s.is_complete = true;
    s.just_updated = true;
    forced_cut_ = cut;   // (the reference's resize() of the look-behind buffer here is a no-op: it keeps its contents)
  } else if (voice && prev_voice_) {
    VadSegment& s = segments_.back();
    s.audio.insert(s.audio.end(), hop, hop + hop_);
    s.end_time = now;
    s.is_complete = false;
    s.just_updated = true;
  }
  prev_voice_ = voice;
}

}  // namespace msh_host
