#include "transcriber.h"

#include <math.h>
#include <sys/stat.h>

#include <chrono>
#include <iterator>
#include <random>
#include <stdexcept>
#include <future>
#include <thread>
#include <algorithm>

#include "host_utils.h"
#include "msh_common.h"

namespace msh_host {

// ------------------------------------------------------------------------------------------------
// MoonshineModel
// ------------------------------------------------------------------------------------------------
MoonshineModel::MoonshineModel(bool log_run, float mtps, const std::vector<int>& device_ids)
    : max_tokens_per_second(mtps), log_ort_run(log_run) {
  if (device_ids.empty()) throw std::runtime_error("no GPU selected");
  for (int dev : device_ids) {
    DeviceShard d;
    d.device = dev;
    const int32_t rc = msh_create(dev, &d.engine);
    if (rc != MSH_OK) {
      const std::string why = msh_last_error(nullptr);
      for (DeviceShard& o : devices) msh_destroy(o.engine);
      devices.clear();
      throw std::runtime_error("cannot create the MI355X engine on device " + std::to_string(dev) + " (status " +
                               std::to_string(rc) + "): " + why);
    }
    devices.push_back(d);
  }
  engine = devices[0].engine;
}

MoonshineModel::~MoonshineModel() {
  for (DeviceShard& d : devices) msh_destroy(d.engine);
  delete tokenizer;
}

std::string MoonshineModel::error() const { return msh_last_error(engine); }

int MoonshineModel::load(const char* weights_path, const char* tokenizer_path, int32_t model_type) {
  if (weights_path == nullptr || tokenizer_path == nullptr) return 1;
  for (DeviceShard& d : devices)   // replicated weights: every device reads the file itself, nothing is broadcast
    if (msh_load_weights_file(d.engine, weights_path, model_type) != MSH_OK) {
      MSH_LOGF("Failed to load weights from '%s' on device %d: %s", weights_path, d.device, msh_last_error(d.engine));
      return 1;
    }
  tokenizer = BinTokenizer::from_file(tokenizer_path);
  return 0;
}

int MoonshineModel::load_from_memory(const uint8_t* weights, size_t weights_size, const uint8_t* tokenizer_data,
                                     size_t tokenizer_size, int32_t model_type) {
  if (weights == nullptr || tokenizer_data == nullptr) return 1;
  for (DeviceShard& d : devices)
    if (msh_load_weights_memory(d.engine, weights, weights_size, model_type) != MSH_OK) {
      MSH_LOGF("Failed to load weights from memory on device %d: %s", d.device, msh_last_error(d.engine));
      return 1;
    }
  tokenizer = new BinTokenizer(tokenizer_data, tokenizer_size);
  return 0;
}

int MoonshineModel::run_shard(DeviceShard& d, const std::vector<uint32_t>& idx_in, const std::vector<const float*>& audio,
                              const std::vector<uint64_t>& lens, std::vector<std::vector<int32_t>>* ids) {
  const uint32_t count = (uint32_t)idx_in.size();
  if (count == 0) return 0;
  // Longest first: a sub-batch decodes until its LAST clip is done, and a decode step costs about the same for 16 rows as
  // for 256 (latency-bound GEMMs), so clips of similar length -- similar step budgets -- belong together.
  std::vector<uint32_t> idx(idx_in);
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return lens[a] > lens[b]; });
  std::vector<const float*> pcm(count);
  std::vector<uint64_t> n(count);
  for (uint32_t i = 0; i < count; ++i) {
    pcm[i] = audio[idx[i]];
    n[i] = lens[idx[i]];
  }
  const size_t longest = (size_t)n[0];
  // Sub-batches are sized by AUDIO, not by clip count: batch_clips x 10 s of it (what a sub-batch of batch_clips full
  // clips holds; the workspace and the HBM-bound part of a decode step scale with frames), at most 4 x batch_clips
  // clips.  With the reference's default VAD a 2048-clip call turns into ~4100 segments of ~5 s: cut by count (256) that
  // was 17 decode chains, by audio it is 9, each step of which costs little more.
  const uint32_t clip_cap = (uint32_t)std::min<long>(4L * std::max(1, batch_clips), 1024);
  const uint64_t audio_cap = (uint64_t)std::max(1, batch_clips) * 160000ull;
  std::vector<std::pair<uint32_t, uint32_t>> cuts;   // [lo, lo + m)
  for (uint32_t lo = 0; lo < count;) {
    uint64_t sum = 0;
    uint32_t m = 0;
    while (lo + m < count && m < clip_cap) {
      // the first batch_clips clips always go in (the old rule); further ones while the audio budget lasts
      if (m >= (uint32_t)std::max(1, batch_clips) && sum + n[lo + m] > audio_cap) break;
      sum += n[lo + m];
      ++m;
    }
    cuts.push_back({lo, m});
    lo += m;
  }
  // rows wide enough for the step budget of the longest clip (the engine's rule: ceil(seconds * tokens/s)) + BOS
  const int32_t stride = (int32_t)ceilf((float)longest / 16000.0f * max_tokens_per_second) + 2;
  std::vector<int32_t> tokens((size_t)count * stride, 0), counts(count, 0);
  if (cuts.size() <= 1 || batches_in_flight <= 1) {
    for (const auto& c : cuts) {  // one sub-batch after the other on the engine itself
      const uint32_t lo = c.first, m = c.second;
      if (msh_encode(d.engine, pcm.data() + lo, n.data() + lo, m, 0, max_tokens_per_second) != MSH_OK) {
        MSH_LOGF("encoder failed on device %d: %s", d.device, msh_last_error(d.engine));
        return 1;
      }
      if (msh_max_decode_steps(d.engine) + 1 > stride) {
        MSH_LOGF("internal: step budget %d above the token row width %d", msh_max_decode_steps(d.engine), stride);
        return 1;
      }
      if (msh_decode(d.engine, -1, nullptr, 0, nullptr, 0, tokens.data() + (size_t)lo * stride, counts.data() + lo, stride) != MSH_OK) {
        MSH_LOGF("decoder failed on device %d: %s", d.device, msh_last_error(d.engine));
        return 1;
      }
    }
  } else {
    if (!d.lanes_ready) {
      if (msh_set_batches_in_flight(d.engine, batches_in_flight) != MSH_OK) {
        MSH_LOGF("batches in flight on device %d: %s", d.device, msh_last_error(d.engine));
        return 1;
      }
      d.lanes_ready = true;
    }
    std::vector<int64_t> tickets;
    bool failed = false;
    for (const auto& c : cuts) {
      const uint32_t lo = c.first, m = c.second;
      const int64_t t = msh_submit_transcribe_tokens(d.engine, pcm.data() + lo, n.data() + lo, m, 0, max_tokens_per_second, -1,
                                                     tokens.data() + (size_t)lo * stride, counts.data() + lo, stride);
      if (t < 0) {
        MSH_LOGF("submit failed on device %d: %s", d.device, msh_last_error(d.engine));
        failed = true;
        break;
      }
      tickets.push_back(t);
    }
    for (int64_t t : tickets)  // every queued sub-batch is waited for, also after a failure: they write into `tokens`
      if (msh_wait(d.engine, t) != MSH_OK) {
        MSH_LOGF("sub-batch failed on device %d: %s", d.device, msh_last_error(d.engine));
        failed = true;
      }
    if (failed) return 1;
  }
  for (uint32_t i = 0; i < count; ++i)
    (*ids)[idx[i]].assign(tokens.begin() + (size_t)i * stride, tokens.begin() + (size_t)i * stride + counts[i]);
  return 0;
}

// ---- the batch in pieces (transcriber.h: rolling_begin / rolling_add / rolling_finish) ----
int MoonshineModel::rolling_begin() {
  std::unique_ptr<Rolling> r(new Rolling());
  r->lock = std::unique_lock<std::mutex>(processing_mutex);
  {
    const char* f = msh::dev_getenv("MSH_ROLLING_SHORT_FRAC");
    const char* nr = msh::dev_getenv("MSH_ROLLING_NARROW_RUNS");
    r->plan.reset(new RollingPlanner(batch_clips, f != nullptr ? atof(f) : 0.15, nr == nullptr || atoi(nr) != 0));
  }
  if (msh_set_capture_cross_attention(engine, 0) != MSH_OK) return 1;
  for (DeviceShard& d : devices)
    if (!d.lanes_ready) {
      if (msh_set_batches_in_flight(d.engine, std::max(1, batches_in_flight)) != MSH_OK) {
        MSH_LOGF("batches in flight on device %d: %s", d.device, msh_last_error(d.engine));
        return 1;
      }
      d.lanes_ready = true;
    }
  rolling_ = std::move(r);
  return 0;
}

int MoonshineModel::rolling_submit(const RollingClip* c, uint32_t m) {
  Rolling& r = *rolling_;
  r.subs.emplace_back();
  RollingSub& sb = r.subs.back();
  // Which device: the one with the least estimated work so far, a sub-batch whose clips all sit in the device VAD's kept audio
  // counting its PCIe upload on every OTHER device.  (Plain round-robin gave long-clip and short-clip sub-batches alternately to
  // the same devices -- their costs differ by the decode steps of the longest clip -- and sent (N - 1) / N of the resident
  // sub-batches to devices that had to upload the PCM again.)  Estimate, from the 256 x 10 s batch: encoder 4 us per audio
  // second, a decode step 0.45 ms at 6.5 steps per second of the LONGEST clip, upload 2.6 us per audio second.  The sub-batches
  // themselves do not depend on the device list (ids(N devices) == ids(1)).
  {
    double audio_s = 0.0, longest_s = 0.0;
    bool resident = true;
    for (uint32_t i = 0; i < m; ++i) {
      audio_s += (double)c[i].n / 16000.0;
      longest_s = std::max(longest_s, (double)c[i].n / 16000.0);
      resident = resident && c[i].dev != nullptr;
    }
    const double cost = 0.004 * audio_s + 0.45 * (double)max_tokens_per_second * longest_s, upload = 0.0026 * audio_s;
    if (r.assigned_ms.size() != devices.size()) r.assigned_ms.assign(devices.size(), 0.0);
    size_t best = r.next_dev % devices.size();
    double best_t = 1e300;
    for (size_t k = 0; k < devices.size(); ++k) {
      const size_t dv = (r.next_dev + k) % devices.size();   // ties: the round-robin order
      const double t = r.assigned_ms[dv] + cost + ((resident && devices[dv].device == r.device_audio_gpu) ? 0.0 : upload);
      if (t < best_t) best_t = t, best = dv;
    }
    sb.dev = best;
    r.assigned_ms[best] = best_t;
    r.next_dev = (best + 1) % devices.size();
  }
  DeviceShard& d = devices[sb.dev];
  bool on_device = d.device == r.device_audio_gpu;
  for (uint32_t i = 0; i < m && on_device; ++i) on_device = c[i].dev != nullptr;
  sb.idx.resize(m), sb.pcm.resize(m), sb.n.resize(m);
  uint64_t longest = 0;
  for (uint32_t i = 0; i < m; ++i) {
    sb.idx[i] = c[i].idx;
    sb.pcm[i] = on_device ? c[i].dev : c[i].host;
    sb.n[i] = c[i].n;
    longest = std::max(longest, c[i].n);
  }
  // rows wide enough for the step budget of the longest clip (the engine's rule: ceil(seconds * tokens/s)) + BOS
  sb.stride = (int32_t)ceilf((float)longest / 16000.0f * max_tokens_per_second) + 2;
  sb.tokens.assign((size_t)m * sb.stride, 0);
  sb.counts.assign(m, 0);
  sb.ticket = msh_submit_transcribe_tokens(d.engine, sb.pcm.data(), sb.n.data(), m, on_device ? 1 : 0, max_tokens_per_second, -1,
                                           sb.tokens.data(), sb.counts.data(), sb.stride);
  if (sb.ticket < 0) {
    MSH_LOGF("submit failed on device %d: %s", d.device, msh_last_error(d.engine));
    r.subs.pop_back();
    r.failed = true;
    return 1;
  }
  return 0;
}

int MoonshineModel::rolling_add(const float* const* host_audio, const float* const* device_audio, int device_audio_gpu,
                                const size_t* n_samples, size_t count, bool last) {
  if (!rolling_) return 1;
  Rolling& r = *rolling_;
  if (r.failed) return 1;
  if (device_audio != nullptr) r.device_audio_gpu = device_audio_gpu;
  std::vector<uint64_t> lens(count);
  for (size_t i = 0; i < count; ++i) {
    lens[i] = (uint64_t)n_samples[i];
    r.clips.push_back({(uint32_t)r.clips.size(), host_audio[i], device_audio != nullptr ? device_audio[i] : nullptr, lens[i]});
  }
  // which of the waiting clips go out now: rolling_plan.h
  std::vector<RollingClip> sub;
  for (const std::vector<uint32_t>& ids : r.plan->add(lens.data(), count, last)) {
    sub.clear();
    for (uint32_t id : ids) sub.push_back(r.clips[id]);
    if (rolling_submit(sub.data(), (uint32_t)sub.size()) != 0) return 1;
  }
  return 0;
}

int MoonshineModel::rolling_finish(std::vector<std::string>* out_texts) {
  std::unique_ptr<Rolling> r = std::move(rolling_);   // released (and the model unlocked) on every way out
  if (!r) return 1;
  bool failed = r->failed;
  for (RollingSub& sb : r->subs)   // every submitted sub-batch is waited for, also after a failure: the lanes write into it
    if (msh_wait(devices[sb.dev].engine, sb.ticket) != MSH_OK) {
      MSH_LOGF("sub-batch failed on device %d: %s", devices[sb.dev].device, msh_last_error(devices[sb.dev].engine));
      failed = true;
    }
  if (failed || out_texts == nullptr) return failed ? 1 : 0;
  if (r->plan->waiting() != 0) {
    MSH_LOGF("internal: %zu clips were never submitted (rolling_add without last)", r->plan->waiting());
    return 1;
  }
  out_texts->assign(r->clips.size(), std::string());
  for (const RollingSub& sb : r->subs)
    for (size_t i = 0; i < sb.idx.size(); ++i)
      (*out_texts)[sb.idx[i]] = tokenizer->tokens_to_text(sb.tokens.data() + i * (size_t)sb.stride, (size_t)sb.counts[i]);
  return 0;
}

int MoonshineModel::transcribe_batch(const std::vector<const float*>& audio, const std::vector<size_t>& n_samples,
                                     std::vector<std::string>* out_texts, std::vector<std::vector<TranscriberWord>>* out_words) {
  std::lock_guard<std::mutex> lock(processing_mutex);
  const uint32_t count = (uint32_t)audio.size();
  out_texts->assign(count, std::string());
  const bool want_words = word_timestamps && out_words != nullptr;
  if (out_words != nullptr) out_words->assign(count, {});
  if (msh_set_capture_cross_attention(engine, want_words ? 1 : 0) != MSH_OK) return 1;
  if (count == 0) return 0;
  std::vector<uint64_t> lens(n_samples.begin(), n_samples.end());
  std::vector<std::vector<int32_t>> ids(count);
  if (log_ort_run || want_words) {
    // diagnostic / word-timestamp calls: one sub-batch after the other on the first device
    if (log_ort_run) {
      msh_profile_reset(engine);
      msh_profile_enable(engine, 1);
    }
    const uint32_t chunk = (uint32_t)std::max(1, batch_clips);
    for (uint32_t lo = 0; lo < count; lo += chunk) {
      const uint32_t n = std::min(chunk, count - lo);
      if (msh_encode(engine, audio.data() + lo, lens.data() + lo, n, 0, max_tokens_per_second) != MSH_OK) {
        MSH_LOGF("encoder failed: %s", error().c_str());
        return 1;
      }
      const int32_t st = msh_max_decode_steps(engine) + 1;
      std::vector<int32_t> part((size_t)n * st), counts(n);
      if (msh_decode(engine, -1, nullptr, 0, nullptr, 0, part.data(), counts.data(), st) != MSH_OK) {
        MSH_LOGF("decoder failed: %s", error().c_str());
        return 1;
      }
      for (uint32_t i = 0; i < n; ++i) ids[lo + i].assign(part.begin() + (size_t)i * st, part.begin() + (size_t)i * st + counts[i]);
      if (want_words) {  // the attention of this sub-batch is only on the device until the next decode
        std::vector<float> att;
        for (uint32_t i = 0; i < n; ++i) {
          int32_t dims[3] = {0, 0, 0};
          const int64_t need = msh_get_cross_attention(engine, i, nullptr, 0, dims);
          if (need < 0) {
            MSH_LOGF("cross-attention not available: %s", error().c_str());
            return 1;
          }
          if (need == 0 || counts[i] < 2) continue;
          att.resize((size_t)need);
          if (msh_get_cross_attention(engine, i, att.data(), (uint64_t)att.size(), dims) < 0) return 1;
          // seconds per encoder frame: clip duration / frames (reference core/moonshine-model.cpp:636)
          const float spf = ((float)lens[lo + i] / 16000.0f) / (float)dims[2];
          (*out_words)[lo + i] = align_words(att.data(), dims[0], dims[1], dims[2], ids[lo + i], spf, *tokenizer);
        }
      }
    }
    if (log_ort_run) {
      const int32_t n = msh_profile_count(engine);
      for (int32_t i = 0; i < n; ++i) {
        msh_profile_entry pe;
        if (msh_profile_get(engine, i, &pe) == MSH_OK)
          MSH_LOGF("kernel group %-24s %8.3f ms over %llu launches", pe.name, pe.ms, (unsigned long long)pe.launches);
      }
      msh_profile_enable(engine, 0);
    }
  } else {
    // Sharding: clips are independent end to end, so the only "communication" is handing out
    // pointers and collecting ids.  The clip list is sorted by length (longest first) and dealt to the devices in
    // snake order -- every device gets the same mix of lengths, hence the same amount of audio within one clip -- and
    // each device cuts its share, still sorted, into sub-batches of batch_clips: clips of similar length end up in the
    // same sub-batch, so little is wasted on padding rows.  (A plain contiguous split of the sorted list, as the survey
    // words it, would give one GPU all the long clips.)  With one device the order is left as the caller gave it.
    const size_t nd = std::min<size_t>(devices.size(), count);
    std::vector<std::vector<uint32_t>> shard(std::max<size_t>(nd, 1));
    if (nd <= 1) {
      shard[0].resize(count);
      for (uint32_t i = 0; i < count; ++i) shard[0][i] = i;
    } else {
      std::vector<uint32_t> order(count);
      for (uint32_t i = 0; i < count; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lens[a] > lens[b]; });
      for (uint32_t k = 0; k < count; ++k) {
        const size_t round = k / nd, pos = k % nd;
        shard[(round & 1) ? nd - 1 - pos : pos].push_back(order[k]);
      }
    }
    std::vector<int> rc(shard.size(), 0);
    if (shard.size() == 1) {
      rc[0] = run_shard(devices[0], shard[0], audio, lens, &ids);
    } else {
      std::vector<std::thread> workers;
      for (size_t d = 0; d < shard.size(); ++d)
        workers.emplace_back([&, d] {
          try {
            rc[d] = run_shard(devices[d], shard[d], audio, lens, &ids);
          } catch (const std::exception& e) {
            MSH_LOGF("device %d: %s", devices[d].device, e.what());
            rc[d] = 1;
          }
        });
      for (std::thread& t : workers) t.join();
    }
    for (int r : rc)
      if (r != 0) return 1;
  }
  for (uint32_t i = 0; i < count; ++i) (*out_texts)[i] = tokenizer->tokens_to_text(ids[i].data(), ids[i].size());
  return 0;
}

int MoonshineModel::transcribe(const float* audio, size_t n, char** out_text) {
  *out_text = nullptr;
  if (audio == nullptr || n == 0) {
    MSH_LOGF("Audio data is nullptr or empty");
    return 1;
  }
  std::vector<std::string> texts;
  const int rc = transcribe_batch({audio}, {n}, &texts);
  if (rc != 0) return rc;
  last_result = texts[0];
  *out_text = const_cast<char*>(last_result.c_str());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// TranscriptOutput
// ------------------------------------------------------------------------------------------------
void TranscriptOutput::clear_update_flags() {
  std::lock_guard<std::mutex> lock(mutex);
  for (uint64_t id : order) {
    TranscriberLine& l = lines.at(id);
    l.just_updated = l.is_new = l.has_text_changed = false;
  }
  for (transcript_line_t& l : c_lines) l.is_updated = l.has_text_changed = l.is_new = l.have_speakers_changed = 0;
}

void TranscriptOutput::add_or_update(TranscriberLine& line) {
  auto it = lines.find(line.id);
  if (it != lines.end()) {
    line.is_new = false;
    const TranscriberLine& old = it->second;
    line.has_text_changed = (old.has_text != line.has_text) || (old.has_text && line.has_text && old.text != line.text);
  } else {
    line.is_new = true;
    line.has_text_changed = line.has_text;
  }
  lines[line.id] = std::move(line);
}

void TranscriptOutput::rebuild() {
  std::lock_guard<std::mutex> lock(mutex);
  c_lines.clear();
  c_words.assign(order.size(), {});
  size_t li = 0;
  for (uint64_t id : order) {
    const TranscriberLine& l = lines[id];
    transcript_line_t c{};
    std::vector<transcript_word_t>& cw = c_words[li++];
    for (const TranscriberWord& w : l.words) cw.push_back(transcript_word_t{w.text.c_str(), w.start, w.end, w.confidence});
    c.words = cw.empty() ? nullptr : cw.data();
    c.word_count = cw.size();
    c.text = l.has_text ? l.text.c_str() : nullptr;
    c.audio_data = l.audio.empty() ? nullptr : l.audio.data();
    c.audio_data_count = l.audio.size();
    c.start_time = l.start_time;
    c.duration = l.duration;
    c.id = l.id;
    c.is_complete = l.is_complete;
    c.is_updated = l.just_updated;
    c.is_new = l.is_new;
    c.has_text_changed = l.has_text_changed;
    c.last_transcription_latency_ms = l.latency_ms;
    c_lines.push_back(c);
  }
  transcript.lines = c_lines.data();
  transcript.line_count = c_lines.size();
}

void TranscriptOutput::mark_all_complete() {
  {
    std::lock_guard<std::mutex> lock(mutex);
    for (uint64_t id : order) {
      TranscriberLine& l = lines[id];
      if (!l.is_complete) {
        l.is_complete = true;
        l.just_updated = true;
      }
    }
  }
  rebuild();
}

// ------------------------------------------------------------------------------------------------
// Transcriber
// ------------------------------------------------------------------------------------------------
namespace {
bool is_streaming_arch(uint32_t a) { return a >= MOONSHINE_MODEL_ARCH_TINY_STREAMING && a <= MOONSHINE_MODEL_ARCH_MEDIUM_STREAMING; }
const char* kWeightsName = "model.safetensors";
const char* kTokenizerName = "tokenizer.bin";
bool is_dir_or_file(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}
}  // namespace

Transcriber::Transcriber(const TranscriberOptions& options) : opt_(options) {
  std::random_device rd;  // random 64-bit base for line ids (reference core/transcriber.cpp:112-117)
  next_line_id_ = ((uint64_t)rd() << 32) | (uint64_t)rd();
  load_vad_model();  // before the model: a transcriber that could never segment audio must not load (reference default 0.5)
  if (opt_.model_source == TranscriberOptions::NONE) return;
  if (opt_.model_arch > MOONSHINE_MODEL_ARCH_MEDIUM_STREAMING)
    throw std::runtime_error("Invalid model architecture: " + std::to_string(opt_.model_arch));
  if (!(opt_.max_tokens_per_second > 0.0f)) throw std::runtime_error("max_tokens_per_second must be positive");
  if (is_streaming_arch(opt_.model_arch)) {
    // one line = one device slot sized for max_stream_seconds of memory frames
    vad_hard_cap_ = (size_t)(opt_.max_stream_seconds * kSampleRate);
    load_streaming_model();
    // compiled last: this needs the tokenizer the load just brought up (reference core/transcriber.cpp:201-213)
    if (!opt_.context.empty()) {
      std::vector<std::string> terms = opt_.keyterms;  // terms named outright are kept next to the passage's
      for (const std::string& t : keyterms_from_context(opt_.context, opt_.context_max_terms)) terms.push_back(t);
      set_keyterms(terms);
    } else if (!opt_.keyterms.empty()) {
      set_keyterms(opt_.keyterms);
    }
    return;
  }
  if (!opt_.keyterms.empty() || !opt_.context.empty())
    throw std::runtime_error("Key-term biasing requires one of the streaming model architectures; the loaded model "
                             "does not decode through a path that can apply it.");
  {  // offline engine limits: 504 decode steps (include/moonshine_hip.h) and 8192 encoder frames of 384 samples
    const double by_steps = 504.0 / (double)opt_.max_tokens_per_second, by_frames = 8192.0 * 384.0 / kSampleRate;
    vad_hard_cap_ = (size_t)((by_steps < by_frames ? by_steps : by_frames) * kSampleRate);
  }
  {
    std::vector<int> ids = opt_.device_ids;
    if (ids.empty()) {
      int n = opt_.num_gpus;
      if (n < 0) n = msh_device_count() - opt_.device;   // every visible GPU from `device` on
      if (n < 1) n = 1;
      for (int i = 0; i < n; ++i) ids.push_back(opt_.device + i);
    }
    const int visible = msh_device_count();
    for (int d : ids)
      if (d < 0 || d >= visible)
        throw std::runtime_error("GPU " + std::to_string(d) + " requested (options device / num_gpus / devices) but only " +
                                 std::to_string(visible) + " visible");
    model_.reset(new MoonshineModel(opt_.log_ort_run, opt_.max_tokens_per_second, ids));
  }
  model_->batch_clips = opt_.batch_clips;
  model_->batches_in_flight = opt_.batches_in_flight;
  model_->word_timestamps = opt_.word_timestamps;
  if (opt_.kv_dtype != 0) {
    if (opt_.word_timestamps) throw std::runtime_error("kv_dtype=fp8 cannot be combined with word_timestamps (the capture reads bf16 keys)");
    for (MoonshineModel::DeviceShard& d : model_->devices)
      if (msh_set_kv_dtype(d.engine, opt_.kv_dtype) != MSH_OK)
        throw std::runtime_error(std::string("kv_dtype: ") + msh_last_error(d.engine));
  }
  if (opt_.model_source == TranscriberOptions::FILES) {
    if (opt_.model_path.empty()) throw std::runtime_error("Model path is null");
    if (!is_dir_or_file(opt_.model_path))
      throw std::runtime_error("Model directory does not exist at path '" + opt_.model_path + "'");
    const std::string tok = join_path(opt_.model_path, kTokenizerName);
    if (!file_exists(tok)) throw std::runtime_error("Required tokenizer file does not exist at path '" + tok + "'");
    const std::string wts = join_path(opt_.model_path, kWeightsName);
    if (!file_exists(wts)) {
      if (file_exists(join_path(opt_.model_path, "encoder_model.ort")))
        throw std::runtime_error("'" + opt_.model_path +
                                 "' holds ONNX Runtime .ort graphs; the MI355X build loads model.safetensors "
                                 "(HuggingFace Moonshine tensor names) -- see INTEGRATION.md");
      throw std::runtime_error("Required model file does not exist at path '" + wts + "'");
    }
    if (model_->load(wts.c_str(), tok.c_str(), (int32_t)opt_.model_arch) != 0)
      throw std::runtime_error("Failed to load model from '" + opt_.model_path + "': " + model_->error());
  } else {
    auto get = [&](const char* name, std::vector<uint8_t>* owned, const uint8_t** p, size_t* n) {
      auto it = opt_.memory_files.find(name);
      if (it == opt_.memory_files.end()) throw std::runtime_error(std::string("Required model asset missing: ") + name);
      if (it->second.first != nullptr && it->second.second > 0) {
        *p = it->second.first;
        *n = it->second.second;
      } else {  // no buffer: the key is a path
        if (!read_file(name, owned)) throw std::runtime_error(std::string("cannot read ") + name);
        *p = owned->data();
        *n = owned->size();
      }
    };
    std::vector<uint8_t> o1, o2;
    const uint8_t *w = nullptr, *t = nullptr;
    size_t wn = 0, tn = 0;
    get(kWeightsName, &o1, &w, &wn);
    get(kTokenizerName, &o2, &t, &tn);
    if (model_->load_from_memory(w, wn, t, tn, (int32_t)opt_.model_arch) != 0)
      throw std::runtime_error("Failed to load model from memory: " + model_->error());
  }
  // Form of the decoder's cross-attention: ONE per transcriber, fixed here -- a clip's transcript never depends on how many
  // clips shared its sub-batch.  `auto` = absorbed when the caller configured this transcriber for large sub-batches (the
  // batch_clips / max_batch_size option PASSED and >= 192: where that form pays, k_xattn.hip) and nothing needs the projected
  // keys; else the reference's projected form -- a transcriber loaded without options serves single clips through the
  // latency path (k_dec_small.hip), which exists for the projected form only.
  {
    int mode = opt_.cross_attention;
    const bool needs_keys = opt_.word_timestamps || opt_.kv_dtype != 0;
    if (mode == 2 && needs_keys)
      throw std::runtime_error("cross_attention=absorbed cannot be combined with word_timestamps or kv_dtype=fp8 (both read projected keys)");
    for (MoonshineModel::DeviceShard& d : model_->devices) {
      int m = mode;
      if (m == 0) m = (opt_.batch_clips_given && opt_.batch_clips >= 192 && !needs_keys && msh_cross_absorbed_supported(d.engine) == 1) ? 2 : 1;
      if (msh_set_cross_mode(d.engine, m) != MSH_OK)
        throw std::runtime_error(std::string("cross_attention: ") + msh_last_error(d.engine));
      // Kernel set, fixed here like the form: a transcriber configured for large sub-batches runs EVERY call -- the tail
      // sub-batches of a batch call, a single clip -- on the large-batch kernels, so a clip's transcript does not depend on
      // what shared its sub-batch; a transcriber loaded without batch options keeps the per-call choice (the latency path).
      const bool uniform = opt_.kernel_set == 2 || (opt_.kernel_set == 0 && opt_.batch_clips_given && opt_.batch_clips >= 192);
      if (msh_set_uniform_kernels(d.engine, uniform ? 1 : 0) != MSH_OK)
        throw std::runtime_error(std::string("kernel_set: ") + msh_last_error(d.engine));
      // (two deployments whose EFFECTIVE options are equal can still differ in this form -- `auto` looks at whether a sub-batch
      // size was asked for at all, INTEGRATION.md section A -- so the resolved form is said out loud where logging is on)
      if (opt_.log_ort_run)
        MSH_LOGF("device %d: decoder cross-attention form = %s (option cross_attention=%s, batch_clips %s = %d); kernel set = %s", d.device,
                 m == 2 ? "absorbed" : "projected K/V", mode == 0 ? "auto" : mode == 2 ? "absorbed" : "kv",
                 opt_.batch_clips_given ? "given" : "defaulted", opt_.batch_clips, uniform ? "uniform (large-batch kernels for every call)" : "per call");
    }
  }
}

// Streaming architectures: model directory = model.safetensors + streaming_config.json + tokenizer.bin
// (the reference's holds frontend / encoder / adapter / cross_kv / decoder_kv .ort graphs next to the same
// json and tokenizer, core/moonshine-streaming-model.cpp:233-300).
void Transcriber::load_streaming_model() {
  // (word_timestamps: the reference swaps in decoder_kv_with_attention.ort here, core/transcriber.cpp:327-345; this engine
  //  computes the attention of the final token sequence on request, StreamingEngine::cross_attention)
  const int frames = (int)ceilf(opt_.max_stream_seconds * 50.0f);
  std::vector<int> ids = opt_.device_ids;
  if (ids.empty()) {
    int n = opt_.num_gpus;
    if (n < 0) n = msh_device_count() - opt_.device;   // every visible GPU from `device` on
    if (n < 1) n = 1;
    for (int i = 0; i < n; ++i) ids.push_back(opt_.device + i);
  }
  {
    const int visible = msh_device_count();
    for (int d : ids)
      if (visible > 0 && (d < 0 || d >= visible))
        throw std::runtime_error("GPU " + std::to_string(d) + " requested (options device / num_gpus / devices) but only " +
                                 std::to_string(visible) + " visible");
  }
  streaming_model_.reset(new MoonshineStreamingModel(ids[0], opt_.max_streams, frames));
  for (size_t i = 1; i < ids.size(); ++i)
    streaming_more_.emplace_back(new MoonshineStreamingModel(ids[i], opt_.max_streams, frames));
  if (opt_.model_source == TranscriberOptions::FILES) {
    if (opt_.model_path.empty()) throw std::runtime_error("Model path is null");
    if (!is_dir_or_file(opt_.model_path))
      throw std::runtime_error("Model directory does not exist at path '" + opt_.model_path + "'");
    const std::string tok = join_path(opt_.model_path, kTokenizerName);
    if (!file_exists(tok)) throw std::runtime_error("Required tokenizer file does not exist at path '" + tok + "'");
    for (const char* name : {kWeightsName, "streaming_config.json"}) {
      const std::string f = join_path(opt_.model_path, name);
      if (!file_exists(f)) {
        if (file_exists(join_path(opt_.model_path, "decoder_kv.ort")))
          throw std::runtime_error("'" + opt_.model_path +
                                   "' holds ONNX Runtime .ort graphs; the MI355X build loads model.safetensors "
                                   "(HuggingFace MoonshineStreaming tensor names) -- see INTEGRATION.md");
        throw std::runtime_error("Required model file does not exist at path '" + f + "'");
      }
    }
    for (MoonshineStreamingModel* m : streaming_models())   // replicated weights: every device reads the files itself
      if (m->load(opt_.model_path.c_str(), tok.c_str(), (int32_t)opt_.model_arch) != 0)
        throw std::runtime_error("Failed to load streaming model from '" + opt_.model_path + "': " + m->last_error);
  } else {
    auto get = [&](const char* name, std::vector<uint8_t>* owned, const uint8_t** p, size_t* n) {
      auto it = opt_.memory_files.find(name);
      if (it == opt_.memory_files.end()) throw std::runtime_error(std::string("Required model asset missing: ") + name);
      if (it->second.first != nullptr && it->second.second > 0) {
        *p = it->second.first;
        *n = it->second.second;
      } else {
        if (!read_file(name, owned)) throw std::runtime_error(std::string("cannot read ") + name);
        *p = owned->data();
        *n = owned->size();
      }
    };
    std::vector<uint8_t> o1, o2, o3;
    const uint8_t *w = nullptr, *t = nullptr, *c = nullptr;
    size_t wn = 0, tn = 0, cn = 0;
    get(kWeightsName, &o1, &w, &wn);
    get(kTokenizerName, &o2, &t, &tn);
    get("streaming_config.json", &o3, &c, &cn);
    for (MoonshineStreamingModel* m : streaming_models())
      if (m->load_from_memory(w, wn, std::string((const char*)c, cn), t, tn, (int32_t)opt_.model_arch) != 0)
        throw std::runtime_error("Failed to load streaming model from memory: " + m->last_error);
  }
}

std::vector<std::string> Transcriber::keyterms_from_context(const std::string& context, int32_t max_terms) {
  if (streaming_model_ == nullptr) {
    if (model_ != nullptr)
      throw std::runtime_error("Key-term biasing requires one of the streaming model architectures; the loaded model "
                               "does not decode through a path that can apply it.");
    return {};  // no model at all (skip_transcription)
  }
  return ContextExtractor::extract(context, max_terms, [this](const std::string& word) -> size_t {
    try {
      return streaming_model_->text_to_tokens(word).size();
    } catch (const std::exception&) {
      return 0;  // a word the tokenizer cannot spell costs that word and nothing else
    }
  });
}

void Transcriber::set_context(const std::string& context, int32_t max_terms) {
  set_keyterms(keyterms_from_context(context, max_terms));
}

void Transcriber::set_keyterms(const std::vector<std::string>& keyterms) {
  std::lock_guard<std::mutex> lock(context_biaser_mutex_);
  opt_.keyterms = keyterms;
  context_biaser_.clear();
  context_biaser_.set_boost(opt_.keyterm_boost);
  {  // the drafts were decoded under the previous key terms: drop them (costs one re-decode from BOS)
    std::lock_guard<std::mutex> sl(streams_mutex_);
    for (auto& kv : streams_) kv.second->last_streaming_tokens.clear();
    if (batch_stream_) batch_stream_->last_streaming_tokens.clear();
  }
  if (streaming_model_ == nullptr) {
    if (!keyterms.empty() && model_ != nullptr)
      throw std::runtime_error("Key-term biasing requires one of the streaming model architectures; the loaded model "
                               "does not decode through a path that can apply it.");
    return;  // no model at all (skip_transcription): nothing to tokenize against
  }
  for (const std::string& term : keyterms)
    for (const std::string& variant : ContextBiaser::variants_for_term(term)) {
      const std::vector<int32_t> tokens = streaming_model_->text_to_tokens(variant);
      if (!tokens.empty()) context_biaser_.add_token_sequence(tokens);
    }
  std::lock_guard<std::mutex> ml(model_mutex_);
  for (MoonshineStreamingModel* m : streaming_models())
    if (m->set_biaser(context_biaser_) != 0) throw std::runtime_error("Failed to install the key terms: " + m->last_error);
  if (opt_.log_output_text)
    MSH_LOGF("Compiled %zu key terms for contextual biasing (boost %.2f)", keyterms.size(), context_biaser_.boost());
}

Transcriber::~Transcriber() {
  // streams own device slots of the streaming model: drop them before the model goes away
  streams_.clear();
  if (batch_retire_.joinable()) batch_retire_.join();
  batch_streams_.clear();
  batch_stream_.reset();
  if (silero_device_ != nullptr) msh_silero_destroy(silero_device_);
}

// reference core/transcriber.cpp:1311-1487, batched over streams: feed only the new whole 1280-sample chunks of
// each segment, encode (final = the segment is complete), then decode from scratch -- with the previous
// pass's tokens as a speculative draft when there is one -- and keep the tokens for the next pass.
void Transcriber::transcribe_segments_with_streaming_model(std::vector<StreamingJob>& jobs) {
  // Streams shard over the devices as clips do: a stream's device is fixed when its first line starts --
  // the one holding the fewest lines -- and every device then runs its own batch of this round on its own host thread.
  const std::vector<MoonshineStreamingModel*> models = streaming_models();
  std::vector<std::vector<StreamingJob*>> part(models.size());
  std::vector<int> load(models.size(), 0);
  for (size_t d = 0; d < models.size(); ++d) load[d] = models[d]->states_in_use();
  for (StreamingJob& j : jobs) {
    size_t d = 0;
    if (j.stream->sowner != nullptr) {
      for (size_t k = 0; k < models.size(); ++k)
        if (models[k] == j.stream->sowner) d = k;
    } else {
      for (size_t k = 1; k < models.size(); ++k)
        if (load[k] < load[d]) d = k;
      ++load[d];
      j.stream->sowner = models[d];   // the state itself is created on that device by transcribe_segments_on_model
    }
    part[d].push_back(&j);
  }
  if (models.size() == 1) return transcribe_segments_on_model(models[0], part[0]);
  std::vector<std::exception_ptr> errs(models.size());
  std::vector<std::thread> threads;
  for (size_t d = 0; d < models.size(); ++d) {
    if (part[d].empty()) continue;
    threads.emplace_back([&, d] {
      try {
        transcribe_segments_on_model(models[d], part[d]);
      } catch (...) {
        errs[d] = std::current_exception();
      }
    });
  }
  for (std::thread& t : threads) t.join();
  for (const std::exception_ptr& e : errs)
    if (e) std::rethrow_exception(e);
}

void Transcriber::transcribe_segments_on_model(MoonshineStreamingModel* m, std::vector<StreamingJob*>& jobs) {
  const MoonshineStreamingConfig& cfg = m->config;
  struct Work {
    StreamingJob* job;
    bool is_new = false, fed = false;
  };
  std::vector<Work> work;
  std::vector<MoonshineStreamingState*> feed_states, enc_states;
  std::vector<const float*> feed_audio;
  std::vector<size_t> feed_lens;
  std::vector<uint8_t> enc_final;
  for (StreamingJob* jp : jobs) {
    StreamingJob& j = *jp;
    j.text.clear();
    TranscriberStream* s = j.stream;
    const std::vector<float>& audio = j.segment->audio;
    if (audio.empty()) continue;  // :1314-1316
    if (s->sstate == nullptr) {
      s->sstate = m->create_state();
      s->sowner = m;
      if (s->sstate == nullptr) throw std::runtime_error("no free streaming slot: " + m->last_error);
    }
    Work w;
    w.job = &j;
    w.is_new = j.line_id != s->streaming_segment_id;  // :1321-1327
    if (w.is_new) {
      if (m->reset_state(s->sstate) != 0) throw std::runtime_error("Failed to reset streaming state: " + m->last_error);
      s->streaming_segment_id = j.line_id;
      s->streaming_samples_processed = 0;
      s->last_streaming_tokens.clear();
    }
    const size_t start = s->streaming_samples_processed;
    if (start < audio.size()) {  // :1332-1372
      const size_t chunk_count = (audio.size() - start) / 1280;
      if (chunk_count > 0) {
        feed_states.push_back(s->sstate);
        feed_audio.push_back(audio.data() + start);
        feed_lens.push_back(chunk_count * 1280);
      }
      enc_states.push_back(s->sstate);
      enc_final.push_back(j.segment->is_complete ? 1 : 0);
      s->streaming_samples_processed += chunk_count * 1280;
    }
    work.push_back(w);
  }
  if (m->process_audio_batch(feed_states, feed_audio, feed_lens) != 0)
    throw std::runtime_error("Failed to process audio chunk: " + m->last_error);
  if (m->encode_batch(enc_states, enc_final) != 0) throw std::runtime_error("Failed to encode: " + m->last_error);

  std::vector<Work*> dec;
  std::vector<MoonshineStreamingState*> dec_states;
  std::vector<std::vector<int>> drafts;
  std::vector<int> budgets;
  for (Work& w : work) {
    TranscriberStream* s = w.job->stream;
    if (s->sstate->memory_len() == 0) continue;                                  // :1375-1377
    if (!w.job->segment->is_complete && !opt_.decode_incomplete_lines) continue;   // :1379-1381
    std::vector<int> draft;
    int budget = -1;
    const bool speculative = opt_.use_speculative_decoding && !w.is_new && !s->last_streaming_tokens.empty();
    if (speculative) {
      for (int t : s->last_streaming_tokens)
        if (t != cfg.bos_id && t != cfg.eos_id) draft.push_back(t);              // :1407-1412
    } else {
      const float duration = (float)w.job->segment->audio.size() / (float)kSampleRate;  // :1388-1392
      budget = std::min((int)ceilf(duration * opt_.max_tokens_per_second), 256);
    }
    dec.push_back(&w);
    dec_states.push_back(s->sstate);
    drafts.push_back(std::move(draft));
    budgets.push_back(budget);
  }
  if (m->decoder_reset_batch(dec_states) != 0) throw std::runtime_error("Failed to reset decoder: " + m->last_error);
  std::vector<std::vector<int>> out;
  if (m->decode_full_batch(dec_states, drafts, budgets, &out) != 0)
    throw std::runtime_error("Streaming decode failed: " + m->last_error);
  for (size_t i = 0; i < dec.size(); ++i) {
    TranscriberStream* s = dec[i]->job->stream;
    std::vector<int64_t> tokens;
    tokens.push_back(cfg.bos_id);
    for (int t : out[i]) tokens.push_back(t);
    // the plain loop (:1441-1466) also records the EOS that ended it; it ran out of budget iff it produced
    // `budget` content tokens
    if (budgets[i] >= 0 && (int)out[i].size() < budgets[i]) tokens.push_back(cfg.eos_id);
    s->last_streaming_tokens.assign(tokens.begin(), tokens.end());
    std::string text = m->tokens_to_text(tokens);
    if (opt_.log_output_text) MSH_LOGF("Streaming model transcribed text: '%s'", text.c_str());
    dec[i]->job->text = sanitize_utf8(text);
    if (opt_.word_timestamps && tokens.size() >= 2) {
      // reference core/transcriber.cpp:1028-1068: the cross-attention of every decoder call of the pass (inputs BOS,
      // t1, ... = all tokens but the last), [layers*heads][steps][memory frames], seconds per frame = duration / frames
      std::vector<int> inputs(tokens.begin(), tokens.end() - 1);
      std::vector<float> att;
      int dims[3] = {0, 0, 0};
      if (m->cross_attention(s->sstate, inputs, &att, dims) != 0)
        throw std::runtime_error("Streaming cross-attention failed: " + m->last_error);
      if (dims[1] > 0 && dims[2] > 0 && m->tokenizer != nullptr) {
        const float spf = ((float)dec[i]->job->segment->audio.size() / (float)kSampleRate) / (float)dims[2];
        const std::vector<int32_t> ids(tokens.begin(), tokens.end());
        dec[i]->job->words = align_words(att.data(), dims[0], dims[1], dims[2], ids, spf, *m->tokenizer);
      }
    }
  }
}

// Silero weights for vad_threshold > 0 (the reference's default): option vad_model_path, else silero_vad.safetensors
// in the model directory / among the memory files.  Missing weights fail the LOAD with the explanation, not the first
// transcribe call.
void Transcriber::load_vad_model() {
  if (!(opt_.vad_threshold > 0.0f)) return;
  static const char* kName = "silero_vad.safetensors";
  std::shared_ptr<SileroWeights> w(new SileroWeights());
  auto mem = opt_.memory_files.find(kName);
  auto from_file = [&](const std::string& path) {   // the bytes are kept: the device network is built from them on demand
    FILE* f = fopen(path.c_str(), "rb");
    if (f == nullptr) throw std::runtime_error("cannot open the Silero VAD weights " + path);
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (size <= 0) {
      fclose(f);
      throw std::runtime_error("the Silero VAD weights file " + path + " is empty");
    }
    silero_blob_.resize((size_t)size);
    const size_t got = fread(silero_blob_.data(), 1, (size_t)size, f);
    fclose(f);
    if (got != (size_t)size) throw std::runtime_error("short read of the Silero VAD weights " + path);
    w->load_memory(silero_blob_.data(), silero_blob_.size());
  };
  if (!opt_.vad_model_path.empty()) {
    from_file(opt_.vad_model_path);
  } else if (mem != opt_.memory_files.end() && mem->second.first != nullptr) {
    silero_blob_.assign(mem->second.first, mem->second.first + mem->second.second);
    w->load_memory(silero_blob_.data(), silero_blob_.size());
  } else if (opt_.model_source == TranscriberOptions::FILES && !opt_.model_path.empty() &&
             file_exists(join_path(opt_.model_path, kName))) {
    from_file(join_path(opt_.model_path, kName));
  } else if (opt_.model_source == TranscriberOptions::NONE) {
    // no model at all (the reference loads such a transcriber with its embedded VAD): the load succeeds, and the first
    // stream that would need the network says what is missing (new_stream)
    return;
  } else {
    throw std::runtime_error(
        "vad_threshold=" + std::to_string(opt_.vad_threshold) + " (the default is 0.5) needs the Silero VAD weights: pass the "
        "option vad_model_path=<silero_vad.safetensors> (tools/convert_silero_vad.py writes it from the published model) or "
        "put silero_vad.safetensors into the model directory; vad_threshold=0 treats all audio as speech");
  }
  silero_ = std::move(w);
}

TranscriberStream* Transcriber::new_stream(int32_t id) {
  const int32_t window = (int32_t)ceilf((opt_.vad_window_duration * kSampleRate) / opt_.vad_hop_size);
  const size_t max_seg = (size_t)roundf(opt_.vad_max_segment_duration * kSampleRate);
  if (opt_.vad_threshold > 0.0f && !silero_)
    throw std::runtime_error("vad_threshold=" + std::to_string(opt_.vad_threshold) + " needs the Silero VAD weights (option "
                             "vad_model_path=<silero_vad.safetensors>); vad_threshold=0 treats all audio as speech");
  TranscriberStream* s = new TranscriberStream();
  s->vad.reset(new VoiceActivityDetector(opt_.vad_threshold, window, opt_.vad_hop_size, opt_.vad_look_behind_sample_count,
                                         max_seg, silero_, vad_hard_cap_));
  s->id = id;
  return s;
}

std::shared_ptr<TranscriberStream> Transcriber::find_stream(int32_t id) {
  std::lock_guard<std::mutex> lock(streams_mutex_);
  auto it = streams_.find(id);
  if (it == streams_.end())
    throw std::runtime_error("Stream with ID " + std::to_string(id) + " not found in " + std::to_string(streams_.size()) +
                             " streams");
  return it->second;
}

void Transcriber::save_input(TranscriberStream* s, const float* audio, uint64_t n, int32_t rate, bool flush) {
  if (opt_.save_input_wav_path.empty()) return;
  const size_t before = s->saved_input.size() / kSampleRate;
  if (audio != nullptr) {
    s->saved_input.insert(s->saved_input.end(), audio, audio + n);
    s->saved_rate = rate;
  }
  if (flush || s->saved_input.size() / kSampleRate != before) {
    mkdir(opt_.save_input_wav_path.c_str(), 0755);
    const std::string name = s->id == -1 ? "input_batch.wav" : "input_" + std::to_string(s->id) + ".wav";
    save_wav(join_path(opt_.save_input_wav_path, name), s->saved_input.data(), s->saved_input.size(), s->saved_rate);
  }
}

// The per-segment loop of reference core/transcriber.cpp:989-1148, with every model call of the pass
// gathered into one GPU batch.
void Transcriber::update_from_segments(const std::vector<TranscriberStream*>& streams,
                                       std::vector<std::vector<VadSegment>>& segments, transcript_t** outs,
                                       const std::vector<std::string>* given_texts, uint32_t given_latency_ms) {
  struct Job {
    size_t stream, segment;
  };
  std::vector<Job> jobs;
  std::vector<const float*> ptrs;
  std::vector<size_t> lens;
  std::vector<std::string> texts;
  std::vector<std::vector<TranscriberWord>> words;
  uint32_t latency_ms = 0;
  const bool streaming = streaming_model_ != nullptr;
  for (size_t si = 0; si < streams.size(); ++si) {
    streams[si]->out.clear_update_flags();
    for (size_t gi = 0; gi < segments[si].size(); ++gi) {
      const VadSegment& seg = segments[si][gi];
      if (!seg.just_updated || (model_ == nullptr && !streaming)) continue;
      if (streaming) {
        // the line id doubles as the streaming segment id (reference core/transcriber.cpp:1024-1027)
        std::lock_guard<std::mutex> lock(streams[si]->out.mutex);
        while (gi >= streams[si]->out.order.size()) streams[si]->out.order.push_back(next_line_id_.fetch_add(1));
      } else if (!is_offline_job(seg)) {
        continue;
      }
      jobs.push_back({si, gi});
      ptrs.push_back(seg.audio.data());
      lens.push_back(seg.audio.size());
    }
  }
  if (!jobs.empty() && streaming) {
    // one segment per stream per round: the segments of a stream share its device slot and run in order.
    // The biaser lock is held across the decode so a concurrent set_keyterms cannot swap the trie under it.
    std::lock_guard<std::mutex> biaser_lock(context_biaser_mutex_);
    std::lock_guard<std::mutex> lock(model_mutex_);
    const auto t0 = std::chrono::steady_clock::now();
    texts.assign(jobs.size(), std::string());
    std::vector<char> done(jobs.size(), 0);
    size_t remaining = jobs.size();
    while (remaining > 0) {
      std::vector<StreamingJob> round;
      std::vector<size_t> idx;
      std::vector<char> taken(streams.size(), 0);
      for (size_t j = 0; j < jobs.size(); ++j) {
        if (done[j] || taken[jobs[j].stream]) continue;
        taken[jobs[j].stream] = 1;
        TranscriberStream* s = streams[jobs[j].stream];
        round.push_back({s, &segments[jobs[j].stream][jobs[j].segment], s->out.order.at(jobs[j].segment), std::string()});
        idx.push_back(j);
      }
      transcribe_segments_with_streaming_model(round);
      if (opt_.word_timestamps) words.resize(jobs.size());
      for (size_t k = 0; k < idx.size(); ++k) {
        texts[idx[k]] = round[k].text;
        if (opt_.word_timestamps) words[idx[k]] = std::move(round[k].words);
        done[idx[k]] = 1;
      }
      remaining -= idx.size();
    }
    latency_ms = (uint32_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  } else if (!jobs.empty() && given_texts != nullptr) {
    if (given_texts->size() != jobs.size())
      throw std::runtime_error("internal: " + std::to_string(given_texts->size()) + " texts for " + std::to_string(jobs.size()) + " segments");
    texts = *given_texts;
    latency_ms = given_latency_ms;
  } else if (!jobs.empty()) {
    std::lock_guard<std::mutex> lock(model_mutex_);
    const auto t0 = std::chrono::steady_clock::now();
    if (model_->transcribe_batch(ptrs, lens, &texts, opt_.word_timestamps ? &words : nullptr) != 0)
      throw std::runtime_error("Failed to transcribe: " + model_->error());
    latency_ms = (uint32_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("MSH_HOST_TIMING") != nullptr)
      MSH_LOGF("update: model transcribe_batch of %zu segments in %.1f ms", jobs.size(),
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  size_t job = 0;
  for (size_t si = 0; si < streams.size(); ++si) {
    TranscriberStream* s = streams[si];
    for (size_t gi = 0; gi < segments[si].size(); ++gi) {
      VadSegment& seg = segments[si][gi];
      if (!seg.just_updated) continue;
      std::lock_guard<std::mutex> lock(s->out.mutex);
      TranscriberLine line;
      line.start_time = seg.start_time;
      line.duration = seg.end_time - seg.start_time;
      line.is_complete = seg.is_complete;
      line.just_updated = true;
      if (gi >= s->out.order.size()) s->out.order.push_back(next_line_id_.fetch_add(1));
      line.id = s->out.order.at(gi);
      if (model_ != nullptr || streaming) {
        line.has_text = true;
        if (job < jobs.size() && jobs[job].stream == si && jobs[job].segment == gi) {
          if (opt_.log_output_text && !streaming) MSH_LOGF("Transcribed text: '%s'", texts[job].c_str());
          line.text = streaming ? texts[job] : sanitize_utf8(texts[job]);
          line.latency_ms = latency_ms;
          // words only for finished lines: an open segment is re-transcribed on the next update anyway (reference
          // core/transcriber.cpp:1103-1117); times become absolute by adding the segment start
          // (the streaming path attaches them on every update, like core/transcriber.cpp:1028-1068)
          if (job < words.size() && (seg.is_complete || streaming)) {
            line.words = words[job];
            for (TranscriberWord& w : line.words) {
              w.start += seg.start_time;
              w.end += seg.start_time;
            }
          }
          ++job;
        }
      }
      if (opt_.return_audio_data) line.audio = std::move(seg.audio);
      s->out.add_or_update(line);
    }
    if (!s->vad->is_active()) s->out.mark_all_complete();
    s->out.rebuild();
    if (outs != nullptr && outs[si] == nullptr) outs[si] = &s->out.transcript;
  }
}

void Transcriber::transcribe_without_streaming(const float* audio, uint64_t n, int32_t sample_rate, uint32_t /*flags*/,
                                               transcript_t** out) {
  std::lock_guard<std::mutex> lock(batch_mutex_);
  if (!batch_stream_) batch_stream_.reset(new_stream(-1));
  TranscriberStream* s = batch_stream_.get();
  save_input(s, audio, n, sample_rate, true);
  std::vector<std::vector<VadSegment>> segs(1);
  {
    std::lock_guard<std::mutex> vl(s->vad_mutex);
    s->vad->start();
    {
      std::lock_guard<std::mutex> ol(s->out.mutex);
      s->out.lines.clear();
      s->out.order.clear();
    }
    s->vad->process_audio(audio, (size_t)n, sample_rate);
    s->vad->stop();
    segs[0] = s->vad->take_segments();
  }
  transcript_t* one = nullptr;
  update_from_segments({s}, segs, &one);
  if (out != nullptr) *out = one;
}

void Transcriber::transcribe_batch_without_streaming(const float* const* audio, const uint64_t* n, uint64_t count,
                                                     int32_t sample_rate, uint32_t flags, transcript_t** out) {
  transcribe_batch_any(audio, nullptr, n, count, sample_rate, flags, out);
}
void Transcriber::transcribe_batch_without_streaming_pcm16(const int16_t* const* audio16, const uint64_t* n, uint64_t count,
                                                           int32_t sample_rate, uint32_t flags, transcript_t** out) {
  transcribe_batch_any(nullptr, audio16, n, count, sample_rate, flags, out);
}

// Exactly one of audio / audio16 is given.  16-bit PCM means x / 32768 (exact in fp32): on the pipeline that keeps the device
// VAD's audio the clips cross PCIe at two bytes per sample and are converted on the GPU (msh_silero_submit_pcm16), the
// detectors convert a clip at a time on their host threads; every other path converts the whole call up front.
void Transcriber::transcribe_batch_any(const float* const* audio, const int16_t* const* audio16, const uint64_t* n, uint64_t count,
                                       int32_t sample_rate, uint32_t /*flags*/, transcript_t** out) {
  std::lock_guard<std::mutex> lock(batch_mutex_);
  static const bool timing = getenv("MSH_HOST_TIMING") != nullptr;   // phase times of the host layer, to the log
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
  auto t_phase = now();
  // The previous call's transcripts end here (moonshine-c-api.h: valid until the next call on the handle).  Freeing a
  // few thousand streams and their lines' audio is ~100 ms of allocator work: it goes to a helper thread and this call
  // starts at once.  Streams of a streaming model hold device slots the new streams need, so those are freed in place.
  if (batch_retire_.joinable()) batch_retire_.join();
  if (streaming_model_ || batch_streams_.size() < 64) {
    batch_streams_.clear();
  } else {
    auto* old = new std::vector<std::unique_ptr<TranscriberStream>>(std::move(batch_streams_));
    batch_streams_.clear();
    batch_retire_ = std::thread([old] { delete old; });
  }
  if (timing) MSH_LOGF("batch call: previous results retired in %.1f ms", ms_since(t_phase));
  t_phase = now();
  std::vector<TranscriberStream*> streams;
  std::vector<std::vector<VadSegment>> segs(count);
  batch_streams_.resize(count);
  streams.resize(count);
  {
    // (a stream = a detector with its buffers: ~5 us each, created on a few threads when there are thousands)
    const unsigned nt = count >= 512 ? std::min(8u, effective_cpus()) : 1u;
    std::vector<std::exception_ptr> errs(nt);
    auto make = [&](unsigned t) {
      try {
        for (uint64_t i = t; i < count; i += nt) {
          batch_streams_[i].reset(new_stream(-1));
          streams[i] = batch_streams_[i].get();
        }
      } catch (...) {
        errs[t] = std::current_exception();
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(make, t);
    make(0);
    for (auto& x : th) x.join();
    for (auto& e : errs)
      if (e) {
        batch_streams_.clear();
        std::rethrow_exception(e);
      }
  }
  if (timing) MSH_LOGF("batch call: %llu streams created in %.1f ms", (unsigned long long)count, ms_since(t_phase));
  t_phase = now();
  // Segmentation is per clip (its own detector state, shared read-only weights) and runs on the host: one clip per host
  // thread.  With Silero on, a 10 s clip costs ~15 ms of one core -- serially that is seconds for a batch the GPU
  // transcribes in tens of milliseconds (the reference walks the clips one after the other, transcriber.cpp:997).
  const unsigned vad_threads = opt_.host_threads > 0 ? (unsigned)opt_.host_threads
                                                     : std::min(128u, 2u * effective_cpus());   // 2x: the lanes' threads mostly wait
  // With Silero on and 16 kHz input the network runs on the GPU for the whole wave (silero_device.h: tens of milliseconds
  // for 2048 clips against ~1 s of host threads); the detectors' state machines then only consume the probabilities.
  const bool device_vad = opt_.vad_threshold > 0.0f && opt_.vad_device != 0 && sample_rate == kSampleRate && !silero_blob_.empty() &&
                          !silero_device_failed_;
  if (device_vad && silero_device_ == nullptr) {
    if (msh_silero_create(opt_.device, silero_blob_.data(), silero_blob_.size(), &silero_device_) != MSH_OK) {
      silero_device_failed_ = true;   // (logged by the library) -- the host network takes over
      silero_device_ = nullptr;
    }
  }
  // THIS call's decision: a device VAD left over from an earlier 16 kHz call must not see the un-resampled audio of a
  // call at another rate (process_audio would refuse the probabilities and the whole batch would fail)
  bool use_device_vad = device_vad && silero_device_ != nullptr;
  double detectors_ms = 0.0;   // of the segmentation time: the detectors' state machines (incl. the copy of every segment's audio)
  if (use_device_vad) msh_silero_release_audio(silero_device_);   // the previous call's audio
  // the detectors' state machines of clips [c0, c1), one clip per host thread; probs (nullable): the device network's
  // probabilities of their whole hops, clip k at poff[k]
  auto run_detectors = [&](uint64_t c0, uint64_t c1, const float* probs, const std::vector<size_t>& poff) {
    const auto t_det = now();
    parallel_for((size_t)(c1 - c0), [&](size_t k) {
      const size_t i = (size_t)c0 + k;
      TranscriberStream* s = streams[i];
      std::vector<float> from16;
      const float* a = nullptr;
      if (audio16 != nullptr) {
        from16.resize((size_t)n[i]);
        for (size_t j = 0; j < from16.size(); ++j) from16[j] = (float)audio16[i][j] * (1.0f / 32768.0f);
        a = from16.data();
      } else {
        a = audio[i];
      }
      s->vad->start();
      if (probs != nullptr)
        s->vad->process_audio(a, (size_t)n[i], sample_rate, probs + poff[k], poff[k + 1] - poff[k]);
      else
        s->vad->process_audio(a, (size_t)n[i], sample_rate);
      s->vad->stop();
      segs[i] = s->vad->take_segments();
    }, vad_threads);
    detectors_ms += ms_since(t_det);
  };
  auto hop_offsets = [&](uint64_t c0, uint64_t c1) {
    std::vector<size_t> poff((size_t)(c1 - c0) + 1, 0);
    for (uint64_t i = c0; i < c1; ++i) poff[i - c0 + 1] = poff[i - c0] + (size_t)(n[i] / (uint64_t)opt_.vad_hop_size);
    return poff;
  };
  // e.g. out of device memory: the host network does this chunk and every later one.  keep_device: the rolling call's
  // sub-batches may still read the audio the device VAD kept -- the handle is destroyed when they are done.
  auto device_vad_failed = [&](bool keep_device) {
    MSH_LOGF("device VAD failed (%s): falling back to the host network", msh_silero_last_error(silero_device_));
    if (!keep_device) {
      msh_silero_destroy(silero_device_);
      silero_device_ = nullptr;
    }
    silero_device_failed_ = true;
    use_device_vad = false;
  };
  auto segment = [&](uint64_t c0, uint64_t c1) {
    if (use_device_vad) {
      const std::vector<size_t> poff = hop_offsets(c0, c1);
      std::vector<float> probs(std::max<size_t>(poff.back(), 1));
      if (msh_silero_probabilities(silero_device_, audio + c0, n + c0, c1 - c0, probs.data(), poff.back()) == (int64_t)poff.back()) {
        run_detectors(c0, c1, probs.data(), poff);
        return;
      }
      device_vad_failed(false);
    }
    run_detectors(c0, c1, nullptr, {});
  };
  std::vector<transcript_t*> outs(count, nullptr);
  auto transcribe = [&](uint64_t w0, uint64_t w1) {
    std::vector<TranscriberStream*> sub(streams.begin() + w0, streams.begin() + w1);
    std::vector<std::vector<VadSegment>> sub_segs(std::make_move_iterator(segs.begin() + w0), std::make_move_iterator(segs.begin() + w1));
    update_from_segments(sub, sub_segs, outs.data() + w0);
    if (streaming_model_)
      for (TranscriberStream* s : sub)
        if (s->sstate != nullptr && s->sowner != nullptr) {
          s->sowner->free_state(s->sstate);
          s->sstate = nullptr;
        }
  };
  // Waves.  A streaming architecture keeps one device slot per line being decoded (max_streams of them): larger batches run
  // in waves of that size, and a wave's slots are handed back before the next one starts (the transcripts stay).  An offline
  // model with Silero on is VAD-bound on the host (the network runs for every 32 ms hop of every clip): clips go in waves of
  // two sub-batches and the GPU transcribes wave k while the host threads segment wave k + 1.  Without Silero segmentation
  // is a copy and everything is one wave.
  // ... and with the network on the GPU (tens of milliseconds per thousand clips) the same pipeline, in waves of eight
  // sub-batches after a first wave of one: the device VAD + the detectors' state machines of wave k + 1 run beside the
  // transcription of wave k, and there are few wave boundaries (each one is a tail of half-empty sub-batches on the GPU).
  const bool vad_on = !streaming_model_ && opt_.vad_threshold > 0.0f;
  // The rolling form of the pipeline (below): with Silero on, and without it for calls of more than one sub-batch (their
  // "segmentation" is the copy of every clip into its line, 12 ms for 2048 clips, which then runs beside the first sub-batches).
  // Word timestamps and the per-kernel log read one sub-batch at a time from the first engine and keep the waves
  // (MSH_BATCH_ROLLING=0: the waves / the single wave for everything, for A/B runs).
  const bool rolling = !streaming_model_ && model_ != nullptr && (vad_on || count > (uint64_t)std::max(1, opt_.batch_clips)) &&
                       !opt_.word_timestamps && !opt_.log_ort_run && [] {
    const char* e = msh::dev_getenv("MSH_BATCH_ROLLING");
    return e == nullptr || atoi(e) != 0;
  }();
  const bool pipelined = vad_on || rolling;
  const bool keep_audio = [] {   // MSH_VAD_KEEP_AUDIO=0: segments are uploaded from the host (A/B)
    const char* e = msh::dev_getenv("MSH_VAD_KEEP_AUDIO");
    return e == nullptr || atoi(e) != 0;
  }();
  // 16-bit input anywhere but on the device-audio pipeline: fp32 copies of the clips for the rest of the call
  std::vector<std::vector<float>> from16;
  std::vector<const float*> from16_ptrs;
  if (audio16 != nullptr && !(rolling && use_device_vad && keep_audio)) {
    from16.resize((size_t)count);
    from16_ptrs.resize((size_t)count);
    parallel_for((size_t)count, [&](size_t i) {
      from16[i].resize((size_t)n[i]);
      for (size_t j = 0; j < from16[i].size(); ++j) from16[i][j] = (float)audio16[i][j] * (1.0f / 32768.0f);
      from16_ptrs[i] = from16[i].data();
    }, vad_threads);
    audio = from16_ptrs.data();
    audio16 = nullptr;
  }
  const uint64_t wave = streaming_model_ ? (uint64_t)std::max(1, opt_.max_streams) * (1 + streaming_more_.size())
                        : pipelined      ? (uint64_t)std::max(1, opt_.batch_clips) * (use_device_vad ? 8 : 2)
                                         : std::max<uint64_t>(count, 1);
  double seg_ms = 0.0;
  if (!pipelined) {
    segment(0, count);
    seg_ms = ms_since(t_phase);
    if (timing) MSH_LOGF("batch call: segmentation in %.1f ms", seg_ms);
    t_phase = now();
    for (uint64_t w0 = 0; w0 < count; w0 += wave) transcribe(w0, std::min(count, w0 + wave));
    if (timing) MSH_LOGF("batch call: transcription + transcript assembly in %.1f ms", ms_since(t_phase));
  } else if (rolling) {
    // Clips are segmented chunk by chunk and the segments of every chunk join the model's rolling batch: the GPU starts on
    // the first chunk's segments and never waits for a wave to end.  A chunk = what one pass of the device VAD takes (64 Ki
    // hops, ~200 clips of 10 s; silero_device.cpp), or a few clips per host thread for the host network.
    const auto t_call = now();
    std::vector<std::string> texts;
    size_t n_jobs = 0, n_resident = 0;
    bool begun = false;
    try {
      {
        std::lock_guard<std::mutex> lock(model_mutex_);
        if (model_->rolling_begin() != 0) throw std::runtime_error("Failed to transcribe: " + model_->error());
      }
      begun = true;
      const uint64_t chunk_clips = [] {   // MSH_BATCH_CHUNK_CLIPS: clips per chunk (the tests force many small chunks)
        const char* e = msh::dev_getenv("MSH_BATCH_CHUNK_CLIPS");
        return e != nullptr && atoi(e) > 0 ? (uint64_t)atoi(e) : 0ull;
      }();
      std::vector<std::pair<uint64_t, uint64_t>> chunks;
      for (uint64_t c0 = 0, c1 = 0; c0 < count; c0 = c1) {
        if (chunk_clips > 0) {
          c1 = std::min<uint64_t>(count, c0 + chunk_clips);
        } else if (use_device_vad) {
          uint64_t hops = 0;
          for (c1 = c0; c1 < count; ++c1) {
            const uint64_t h = n[c1] / (uint64_t)opt_.vad_hop_size;
            if (c1 > c0 && hops + h > 65536) break;
            hops += h;
          }
        } else {
          // the host network: a few clips per thread; no network (vad_threshold 0, a clip is one segment): a sub-batch of clips
          c1 = std::min<uint64_t>(count, c0 + (vad_on ? (uint64_t)std::max(64u, 2u * vad_threads) : (uint64_t)std::max(64, opt_.batch_clips)));
        }
        chunks.push_back({c0, c1});
      }
      // The device network runs one chunk ahead (msh_silero_submit / _collect): chunk k + 1 is gathered and uploaded, and its
      // network runs, while this thread feeds chunk k's probabilities to the detectors and submits its segments.
      std::vector<int64_t> tickets(chunks.size(), -1);
      auto vad_submit = [&](size_t k) {
        if (!use_device_vad || k >= chunks.size()) return;
        const uint64_t c0 = chunks[k].first, c1 = chunks[k].second;
        tickets[k] = audio16 != nullptr ? msh_silero_submit_pcm16(silero_device_, audio16 + c0, n + c0, c1 - c0, keep_audio ? 1 : 0)
                                        : msh_silero_submit(silero_device_, audio + c0, n + c0, c1 - c0, keep_audio ? 1 : 0);
        if (tickets[k] < 0) device_vad_failed(true);
      };
      const auto ts0 = now();
      vad_submit(0);
      seg_ms += ms_since(ts0);
      for (size_t k = 0; k < chunks.size(); ++k) {
        const uint64_t c0 = chunks[k].first, c1 = chunks[k].second;
        const auto ts = now();
        vad_submit(k + 1);
        std::vector<const float*> resident;
        bool have_probs = false;
        if (use_device_vad && tickets[k] >= 0) {
          const std::vector<size_t> poff = hop_offsets(c0, c1);
          std::vector<float> probs(std::max<size_t>(poff.back(), 1));
          if (keep_audio) resident.assign((size_t)(c1 - c0), nullptr);
          have_probs = msh_silero_collect(silero_device_, tickets[k], probs.data(), poff.back(), keep_audio ? resident.data() : nullptr,
                                          c1 - c0) == (int64_t)poff.back();
          if (have_probs) {
            run_detectors(c0, c1, probs.data(), poff);
          } else {
            device_vad_failed(true);
            resident.clear();
          }
        }
        if (!have_probs) run_detectors(c0, c1, nullptr, {});
        seg_ms += ms_since(ts);
        std::vector<const float*> host, dev;
        std::vector<size_t> lens;
        for (uint64_t i = c0; i < c1; ++i) {
          const uint64_t whole = n[i] / (uint64_t)opt_.vad_hop_size * (uint64_t)opt_.vad_hop_size;
          for (const VadSegment& seg : segs[i]) {
            if (!is_offline_job(seg)) continue;
            host.push_back(seg.audio.data());
            lens.push_back(seg.audio.size());
            const float* base = resident.empty() ? nullptr : resident[i - c0];
            const bool slice = base != nullptr && seg.src_offset + seg.audio.size() <= whole;
            dev.push_back(slice ? base + seg.src_offset : nullptr);
            n_resident += slice ? 1 : 0;
          }
        }
        n_jobs += host.size();
        if (model_->rolling_add(host.data(), resident.empty() ? nullptr : dev.data(), opt_.device, lens.data(), host.size(),
                                k + 1 == chunks.size()) != 0)
          throw std::runtime_error("Failed to transcribe: " + model_->error());
      }
      begun = false;
      if (model_->rolling_finish(&texts) != 0) throw std::runtime_error("Failed to transcribe: " + model_->error());
    } catch (...) {
      if (begun) model_->rolling_finish(nullptr);   // waits for what was submitted: the lanes write into the batch's arrays
      if (silero_device_failed_ && silero_device_ != nullptr) msh_silero_destroy(silero_device_), silero_device_ = nullptr;
      throw;
    }
    if (silero_device_failed_ && silero_device_ != nullptr) msh_silero_destroy(silero_device_), silero_device_ = nullptr;
    const double call_ms = ms_since(t_call);
    if (timing)
      MSH_LOGF("batch call: %zu segments (%zu read from the device VAD's audio) transcribed in %.1f ms, of which segmentation %.1f ms "
               "(%.1f ms of that in the detectors' state machines on %u threads) beside the transcription", n_jobs, n_resident, call_ms, seg_ms,
               detectors_ms, vad_threads);
    t_phase = now();
    update_from_segments(streams, segs, outs.data(), &texts, (uint32_t)call_ms);
    if (timing) MSH_LOGF("batch call: transcript assembly in %.1f ms", ms_since(t_phase));
  } else {
    std::future<void> pending;   // the previous wave on the GPU
    try {
      // the first wave's segmentation has nothing to hide behind: keep it to one sub-batch of clips, then full waves
      // (waves growing 256 / 512 / 1024 measured slower -- 475 against 413 ms for 2048 clips: every wave is its own set of
      // sub-batches with its own tail)
      uint64_t w1 = 0;
      for (uint64_t w0 = 0; w0 < count; w0 = w1) {
        w1 = std::min(count, w0 + (w0 == 0 ? std::min<uint64_t>(wave, (uint64_t)std::max(1, opt_.batch_clips)) : wave));
        const auto ts = now();
        segment(w0, w1);
        seg_ms += ms_since(ts);
        if (pending.valid()) pending.get();
        pending = std::async(std::launch::async, [&transcribe, w0, w1] { transcribe(w0, w1); });
      }
      if (pending.valid()) pending.get();
    } catch (...) {
      if (pending.valid()) pending.wait();   // never leave the worker running on this frame's state
      throw;
    }
    if (timing)
      MSH_LOGF("batch call: %.1f ms, of which segmentation %.1f ms on %u threads with the previous wave's transcription beside it",
               ms_since(t_phase), seg_ms, vad_threads);
  }
  if (out != nullptr)
    for (uint64_t i = 0; i < count; ++i) out[i] = outs[i];
}

int32_t Transcriber::create_stream() {
  std::lock_guard<std::mutex> lock(streams_mutex_);
  const int32_t id = next_stream_id_++;
  streams_[id].reset(new_stream(id));
  return id;
}

void Transcriber::free_stream(int32_t id) {
  std::lock_guard<std::mutex> lock(streams_mutex_);
  streams_.erase(id);
}

void Transcriber::start_stream(int32_t id) {
  const std::shared_ptr<TranscriberStream> keep = find_stream(id);
  TranscriberStream* s = keep.get();
  std::lock_guard<std::mutex> ol(s->out.mutex);
  s->out.lines.clear();
  s->out.order.clear();
  s->out.transcript.lines = nullptr;  // earlier pointers handed to the client are invalid from here on
  s->out.transcript.line_count = 0;
  s->vad->start();
}

void Transcriber::stop_stream(int32_t id) {
  const std::shared_ptr<TranscriberStream> keep = find_stream(id);
  TranscriberStream* s = keep.get();
  s->vad->stop();
  save_input(s, nullptr, 0, 0, true);
}

void Transcriber::add_audio_to_stream(int32_t id, const float* audio, uint64_t n, int32_t sample_rate) {
  const std::shared_ptr<TranscriberStream> keep = find_stream(id);
  TranscriberStream* s = keep.get();
  if (!s->vad->is_active())
    throw std::runtime_error("Adding new audio for stream with ID " + std::to_string(id) +
                             " but VAD is not active. Did you call start_stream()?");
  save_input(s, audio, n, sample_rate, false);
  std::vector<float> in(audio, audio + n);
  std::vector<float> r = resample(in, (float)sample_rate, (float)kSampleRate);
  std::lock_guard<std::mutex> al(s->audio_mutex);
  s->new_audio.insert(s->new_audio.end(), r.begin(), r.end());
}

// reference core/transcriber.cpp:775-891: only re-run the model when enough new audio has arrived
// (or on FORCE_UPDATE); otherwise hand back the cached transcript with cleared update flags.
void Transcriber::transcribe_stream(int32_t id, uint32_t flags, transcript_t** out) {
  const std::shared_ptr<TranscriberStream> keep = find_stream(id);
  TranscriberStream* s = keep.get();
  std::vector<float> fresh;  // the audio this update consumes; add_audio may keep appending meanwhile
  {
    std::lock_guard<std::mutex> al(s->audio_mutex);
    const size_t n = s->new_audio.size();
    const bool has_new = n > 0;
    const bool long_enough = (float)n / (float)kSampleRate >= opt_.transcription_interval;
    const bool force = (flags & MOONSHINE_FLAG_FORCE_UPDATE) != 0;
    if ((long_enough || force) && has_new) fresh.swap(s->new_audio);
  }
  if (fresh.empty()) {
    s->out.clear_update_flags();
    if (!s->vad->is_active()) s->out.mark_all_complete();
    if (out != nullptr) *out = &s->out.transcript;
    return;
  }
  std::vector<std::vector<VadSegment>> segs(1);
  {
    std::lock_guard<std::mutex> vl(s->vad_mutex);
    s->vad->process_audio(fresh.data(), fresh.size(), kSampleRate);
    for (const VadSegment& seg : s->vad->segments()) {
      VadSegment c;
      c.start_time = seg.start_time;
      c.end_time = seg.end_time;
      c.is_complete = seg.is_complete;
      c.just_updated = seg.just_updated;
      if (seg.just_updated) c.audio = seg.audio;
      segs[0].push_back(std::move(c));
    }
  }
  transcript_t* one = nullptr;
  update_from_segments({s}, segs, &one);
  if (out != nullptr) *out = one;
  if (!opt_.return_audio_data) {
    std::lock_guard<std::mutex> vl(s->vad_mutex);
    s->vad->clear_completed_audio();
  }
}

std::string Transcriber::transcript_to_string(const transcript_t* t) {
  std::string r = std::to_string(t->line_count) + " lines\n";
  for (uint64_t i = 0; i < t->line_count; ++i) {
    char buf[32];
    snprintf(buf, sizeof(buf), "%.1fs: ", t->lines[i].start_time);
    r += buf;
    r += t->lines[i].text == nullptr ? "<null>" : t->lines[i].text;
    r += "\n";
  }
  return r;
}

}  // namespace msh_host
