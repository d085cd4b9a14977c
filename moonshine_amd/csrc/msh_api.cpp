// C ABI over msh::Engine (include/moonshine_hip.h).  C++ exceptions never cross this boundary:
// every entry catches, records the message on the engine and returns a status code -- the rule the
// reference applies at its own C boundary (reference core/moonshine-c-api.cpp:439-446).
#include "../../include/moonshine_hip.h"

#include <stdio.h>
#include <string.h>

#include <string>

#include <stdlib.h>

#include "engine.h"
#include "host_utils.h"
#include "pipeline.h"
#include "silero_device.h"
#include "stream_engine.h"

struct msh_engine {
  msh::Engine* eng = nullptr;
  int device = 0;
  std::unique_ptr<msh::BatchPipeline> pipe;  // batches in flight (msh_set_batches_in_flight)
  std::string last_error;
  std::vector<msh::ProfEntry> prof_cache;
};

struct msh_stream_engine {
  msh::StreamingEngine* eng = nullptr;
  std::string last_error;
  std::vector<msh::ProfEntry> prof_cache;
};

namespace {
// HIP binds streams to hardware queues out of a pool of GPU_MAX_HW_QUEUES (4 by default), by least use.  An engine with
// N lanes needs N + 3 streams that really run concurrently (lanes, the engine's own stream, the utility stream, the
// legacy stream of the host framework); with 4 queues two lanes can share one and then simply alternate (measured:
// 2 lanes 47.4k audio-s/s = no overlap at all, against 59.8k on separate queues).  The pool size is read once, when the
// HIP runtime initialises.  A drop-in library must not change the process environment behind the host's back, so this
// is an explicit call (msh_set_hw_queues / the load option `hw_queues`) -- or the host exports the variable itself.
bool hw_queue_note_given = false;
void note_hw_queues(int lanes) {
  if (lanes < 2 || hw_queue_note_given) return;
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  if (e != nullptr && atoi(e) >= lanes + 3) return;
  hw_queue_note_given = true;
  MSH_LOGF("%d batches in flight want %d HIP hardware queues; GPU_MAX_HW_QUEUES is %s: lanes may share a queue and "
           "serialise.  Export GPU_MAX_HW_QUEUES=8 or call msh_set_hw_queues(8) before the first GPU call.",
           lanes, lanes + 3, e ? e : "unset (4)");
}

thread_local std::string g_create_error;

template <class E, class F>
int32_t guarded(E* e, F&& f) {
  if (e == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  try {
    f();
    return MSH_OK;
  } catch (const std::invalid_argument& ex) {
    e->last_error = ex.what();
    return MSH_ERR_INVALID_ARGUMENT;
  } catch (const msh::HipError& ex) {
    e->last_error = ex.what();
    return MSH_ERR_HIP;
  } catch (const std::exception& ex) {
    e->last_error = ex.what();
    return MSH_ERR_UNKNOWN;
  } catch (...) {
    e->last_error = "unknown exception";
    return MSH_ERR_UNKNOWN;
  }
}
}  // namespace

extern "C" {

int32_t msh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* msh_version(void) { return "moonshine-mi355x 0.1.0 (gfx950)"; }

int32_t msh_create(int32_t device, msh_engine** out) {
  if (out == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  try {
    msh_engine* e = new msh_engine();
    try {
      e->eng = new msh::Engine(device);
      e->device = device;
    } catch (...) {
      delete e;
      throw;
    }
    *out = e;
    return MSH_OK;
  } catch (const msh::HipError& ex) {
    g_create_error = ex.what();
    return msh_device_count() == 0 ? MSH_ERR_NO_DEVICE : MSH_ERR_HIP;
  } catch (const std::exception& ex) {
    g_create_error = ex.what();
    return MSH_ERR_UNKNOWN;
  }
}

void msh_destroy(msh_engine* e) {
  if (e == nullptr) return;
  try {
    e->pipe.reset();  // the lanes borrow the engine's weights
    delete e->eng;
  } catch (...) {
  }
  delete e;
}

const char* msh_last_error(const msh_engine* e) {
  if (e == nullptr) return g_create_error.c_str();
  return e->last_error.c_str();
}

int32_t msh_load_weights_file(msh_engine* e, const char* path, int32_t model_arch) {
  return guarded(e, [&] {
    if (path == nullptr) throw std::invalid_argument("null path");
    msh::SafeTensors st;
    st.load_file(path);
    e->eng->load_weights(st, model_arch);
  });
}

int32_t msh_load_weights_memory(msh_engine* e, const void* data, uint64_t size, int32_t model_arch) {
  return guarded(e, [&] {
    if (data == nullptr) throw std::invalid_argument("null data");
    msh::SafeTensors st;
    st.parse(reinterpret_cast<const uint8_t*>(data), (size_t)size);
    e->eng->load_weights(st, model_arch);
  });
}

int32_t msh_model_info_get(const msh_engine* e, msh_model_info* out) {
  if (e == nullptr || out == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  if (!e->eng->loaded()) return MSH_ERR_INVALID_ARGUMENT;
  const msh::ModelConfig& c = e->eng->config();
  memset(out, 0, sizeof(*out));
  out->hidden = c.hidden;
  out->ffn = c.ffn;
  out->enc_layers = c.enc_layers;
  out->dec_layers = c.dec_layers;
  out->heads = c.heads;
  out->head_dim = c.head_dim();
  out->vocab = c.vocab;
  out->bos = c.bos;
  out->eos = c.eos;
  strncpy(out->arch, c.arch.c_str(), sizeof(out->arch) - 1);
  return MSH_OK;
}

int32_t msh_host_check_weights(const void* data, uint64_t size, int32_t model_arch, msh_model_info* out, char* err, uint64_t err_cap) {
  if (err != nullptr && err_cap > 0) err[0] = 0;
  try {
    if (data == nullptr || size == 0) throw std::invalid_argument("null / empty checkpoint");
    msh::SafeTensors st;
    st.parse(static_cast<const uint8_t*>(data), (size_t)size);
    const msh::ModelConfig c = msh::Engine::check_weights(st, model_arch);
    if (out != nullptr) {
      memset(out, 0, sizeof(*out));
      out->hidden = c.hidden;
      out->ffn = c.ffn;
      out->enc_layers = c.enc_layers;
      out->dec_layers = c.dec_layers;
      out->heads = c.heads;
      out->head_dim = c.head_dim();
      out->vocab = c.vocab;
      out->bos = c.bos;
      out->eos = c.eos;
      strncpy(out->arch, c.arch.c_str(), sizeof(out->arch) - 1);
    }
    return MSH_OK;
  } catch (const std::exception& ex) {
    if (err != nullptr && err_cap > 0) {
      strncpy(err, ex.what(), (size_t)err_cap - 1);
      err[err_cap - 1] = 0;
    }
    return MSH_ERR_INVALID_ARGUMENT;
  }
}

int32_t msh_encode(msh_engine* e, const float* const* pcm, const uint64_t* n_samples, uint32_t count,
                   int32_t pcm_on_device, float max_tokens_per_second) {
  return guarded(e, [&] {
    if (pcm == nullptr || n_samples == nullptr) throw std::invalid_argument("null input");
    for (uint32_t i = 0; i < count; ++i)
      if (pcm[i] == nullptr) throw std::invalid_argument("null clip pointer");
    e->eng->encode(pcm, n_samples, count, pcm_on_device != 0, max_tokens_per_second);
  });
}

int32_t msh_decode(msh_engine* e, int32_t forced_steps, const int32_t* teacher, int32_t teacher_stride,
                   float* logits_out, int32_t logit_steps, int32_t* tokens_out, int32_t* counts_out,
                   int32_t tokens_stride) {
  return guarded(e, [&] {
    e->eng->decode(forced_steps, teacher, teacher_stride, logits_out, logit_steps, tokens_out, counts_out,
                   tokens_stride);
  });
}

int32_t msh_transcribe_tokens(msh_engine* e, const float* const* pcm, const uint64_t* n_samples, uint32_t count,
                              int32_t pcm_on_device, float max_tokens_per_second, int32_t forced_steps,
                              int32_t* tokens_out, int32_t* counts_out, int32_t tokens_stride) {
  int32_t rc = msh_encode(e, pcm, n_samples, count, pcm_on_device, max_tokens_per_second);
  if (rc != MSH_OK) return rc;
  return msh_decode(e, forced_steps, nullptr, 0, nullptr, 0, tokens_out, counts_out, tokens_stride);
}

int32_t msh_max_decode_steps(const msh_engine* e) { return e ? e->eng->max_decode_len() : MSH_ERR_INVALID_ARGUMENT; }

int32_t msh_clip_frames(const msh_engine* e, uint32_t clip) {
  if (e == nullptr || clip >= e->eng->batch_count()) return MSH_ERR_INVALID_ARGUMENT;
  return e->eng->clip_frames(clip);
}

int32_t msh_set_keep_encoder_output(msh_engine* e, int32_t keep) {
  return guarded(e, [&] { e->eng->set_keep_encoder_f32(keep != 0); });
}

int32_t msh_set_kv_dtype(msh_engine* e, int32_t dtype) {
  return guarded(e, [&] {
    if (dtype != 0 && dtype != 1) throw std::invalid_argument("kv dtype: 0 = bf16, 1 = fp8 (e4m3)");
    if (e->pipe) throw std::invalid_argument("msh_set_kv_dtype: set it before msh_set_batches_in_flight (lanes take it at creation)");
    e->eng->set_kv_fp8(dtype == 1);
  });
}

int32_t msh_set_cross_mode(msh_engine* e, int32_t mode) {
  return guarded(e, [&] {
    if (e->pipe) throw std::invalid_argument("msh_set_cross_mode: set it before msh_set_batches_in_flight (lanes take it at creation)");
    e->eng->set_cross_mode(mode);
  });
}

int32_t msh_set_uniform_kernels(msh_engine* e, int32_t on) {
  return guarded(e, [&] {
    if (e->pipe) throw std::invalid_argument("msh_set_uniform_kernels: set it before msh_set_batches_in_flight (lanes take it at creation)");
    e->eng->set_uniform_kernels(on != 0);
  });
}
int32_t msh_uniform_kernels(const msh_engine* e) { return e != nullptr && e->eng->uniform_kernels() ? 1 : 0; }

int32_t msh_cross_absorbed(const msh_engine* e) {
  if (e == nullptr) return 0;
  // with lanes (msh_set_batches_in_flight) the batches are encoded by the lanes; the form is the engine's, all lanes share it
  if (e->pipe) return e->eng->cross_mode() == 2 ? 1 : 0;
  return e->eng->cross_absorbed() ? 1 : 0;
}
int32_t msh_cross_absorbed_supported(const msh_engine* e) { return e != nullptr && e->eng->cross_absorbed_available() ? 1 : 0; }

int32_t msh_get_encoder_output(msh_engine* e, uint32_t clip, float* out) {
  return guarded(e, [&] {
    if (out == nullptr || clip >= e->eng->batch_count()) throw std::invalid_argument("bad clip index / null output");
    e->eng->get_encoder_output(clip, out);
  });
}

int32_t msh_profile_enable(msh_engine* e, int32_t on) {
  return guarded(e, [&] { e->eng->profile_enable(on != 0); });
}
int32_t msh_profile_reset(msh_engine* e) {
  return guarded(e, [&] { e->eng->profile_reset(); });
}
int32_t msh_profile_count(msh_engine* e) {
  if (e == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  int32_t rc = guarded(e, [&] { e->prof_cache = e->eng->profile_get(); });
  return rc == MSH_OK ? (int32_t)e->prof_cache.size() : rc;
}
int32_t msh_profile_get(msh_engine* e, int32_t index, msh_profile_entry* out) {
  if (e == nullptr || out == nullptr || index < 0 || index >= (int32_t)e->prof_cache.size())
    return MSH_ERR_INVALID_ARGUMENT;
  const msh::ProfEntry& p = e->prof_cache[index];
  memset(out, 0, sizeof(*out));
  strncpy(out->name, p.name.c_str(), sizeof(out->name) - 1);
  out->ms = p.ms;
  out->launches = p.launches;
  out->flops = p.flops;
  out->bytes = p.bytes;
  return MSH_OK;
}

double msh_profile_event_overhead_ms(msh_engine* e, int32_t iters) {
  double v = -1.0;
  guarded(e, [&] { v = e->eng->profile_event_overhead_ms(iters); });
  return v;
}

int32_t msh_set_hw_queues(int32_t n) {
  if (n < 1 || n > 64) return MSH_ERR_INVALID_ARGUMENT;
  return setenv("GPU_MAX_HW_QUEUES", std::to_string(n).c_str(), 1) == 0 ? MSH_OK : MSH_ERR_UNKNOWN;
}

int32_t msh_set_batches_in_flight(msh_engine* e, int32_t n) {
  return guarded(e, [&] {
    note_hw_queues(n);
    e->pipe.reset();
    if (n > 0) e->pipe.reset(new msh::BatchPipeline(*e->eng, e->device, n));
  });
}

int64_t msh_submit_transcribe_tokens(msh_engine* e, const float* const* pcm, const uint64_t* n_samples, uint32_t count,
                                     int32_t on_device, float max_tokens_per_second, int32_t forced_steps,
                                     int32_t* tokens_out, int32_t* counts_out, int32_t tokens_stride) {
  int64_t ticket = -1;
  const int32_t rc = guarded(e, [&] {
    if (!e->pipe) throw std::invalid_argument("msh_submit_transcribe_tokens: call msh_set_batches_in_flight first");
    ticket = e->pipe->submit(pcm, n_samples, count, on_device != 0, max_tokens_per_second, forced_steps, tokens_out,
                             counts_out, tokens_stride);
  });
  return rc == MSH_OK ? ticket : (int64_t)rc;
}

int32_t msh_wait(msh_engine* e, int64_t ticket) {
  return guarded(e, [&] {
    if (!e->pipe) throw std::invalid_argument("msh_wait: no batches in flight");
    e->pipe->wait(ticket);
  });
}

int32_t msh_set_capture_cross_attention(msh_engine* e, int32_t on) {
  return guarded(e, [&] { e->eng->set_capture_cross_attention(on != 0); });
}

int64_t msh_get_cross_attention(msh_engine* e, uint32_t clip, float* out, uint64_t cap_floats, int32_t* dims3) {
  int d[3] = {0, 0, 0};
  const int32_t rc = guarded(e, [&] { e->eng->get_cross_attention(clip, out, (size_t)cap_floats, d); });
  if (rc != MSH_OK) return rc;
  if (dims3 != nullptr) dims3[0] = d[0], dims3[1] = d[1], dims3[2] = d[2];
  return (int64_t)d[0] * d[1] * d[2];
}

// (not exported: the development library's msh_test_debug_read, dev_hooks.cpp, reads decode buffers through this)
int64_t msh_internal_debug_read(msh_engine* e, const char* name, void* dst, uint64_t bytes) {
  int64_t v = -1;
  guarded(e, [&] { v = (int64_t)e->eng->debug_read(name ? name : "", dst, bytes); });
  return v;
}

int32_t msh_profile_decode_chain(msh_engine* e, int32_t reps) {
  return guarded(e, [&] { e->eng->profile_decode_chain(reps); });
}

double msh_profile_cross_attention_ms(msh_engine* e, int32_t rounds) {
  double v = -1.0;
  guarded(e, [&] { v = e->eng->profile_cross_attention_ms(rounds); });
  return v;
}

// ---- Silero VAD on the device ----
struct msh_silero {
  msh::SileroDevice* dev = nullptr;
  std::string last_error;
};

int32_t msh_silero_create(int32_t device, const uint8_t* weights, uint64_t weights_size, msh_silero** out) {
  if (out == nullptr || weights == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  try {
    msh_host::SileroWeights w;
    w.load_memory(weights, (size_t)weights_size);
    std::unique_ptr<msh_silero> h(new msh_silero());
    h->dev = new msh::SileroDevice(device, w);
    *out = h.release();
    return MSH_OK;
  } catch (const msh::HipError& ex) {
    fprintf(stderr, "msh_silero_create: %s\n", ex.what());
    return MSH_ERR_HIP;
  } catch (const std::exception& ex) {
    fprintf(stderr, "msh_silero_create: %s\n", ex.what());
    return MSH_ERR_INVALID_ARGUMENT;
  }
}
void msh_silero_destroy(msh_silero* s) {
  if (s == nullptr) return;
  delete s->dev;
  delete s;
}
namespace {
int64_t silero_probabilities(msh_silero* s, const float* const* pcm, const uint64_t* n_samples, uint64_t count, float* probs_out,
                             uint64_t cap, const float** device_audio_out) {
  if (s == nullptr || s->dev == nullptr || (count > 0 && (pcm == nullptr || n_samples == nullptr))) return MSH_ERR_INVALID_ARGUMENT;
  try {
    uint64_t total = 0;
    for (uint64_t i = 0; i < count; ++i) total += n_samples[i] / 512;
    if (total > cap || (total > 0 && probs_out == nullptr)) {
      s->last_error = "probs_out too small";
      return MSH_ERR_INVALID_ARGUMENT;
    }
    std::vector<std::vector<float>> probs;
    std::vector<const float*> resident;
    s->dev->probabilities(pcm, n_samples, (size_t)count, &probs, device_audio_out != nullptr ? &resident : nullptr);
    uint64_t off = 0;
    for (uint64_t i = 0; i < count; ++i) {
      if (!probs[i].empty()) memcpy(probs_out + off, probs[i].data(), probs[i].size() * sizeof(float));
      off += probs[i].size();
      if (device_audio_out != nullptr) device_audio_out[i] = resident[i];
    }
    return (int64_t)off;
  } catch (const msh::HipError& ex) {
    s->last_error = ex.what();
    return MSH_ERR_HIP;
  } catch (const std::exception& ex) {
    s->last_error = ex.what();
    return MSH_ERR_INVALID_ARGUMENT;
  }
}
}  // namespace
int64_t msh_silero_probabilities(msh_silero* s, const float* const* pcm, const uint64_t* n_samples, uint64_t count, float* probs_out,
                                 uint64_t cap) {
  return silero_probabilities(s, pcm, n_samples, count, probs_out, cap, nullptr);
}
int64_t msh_silero_probabilities_keep_audio(msh_silero* s, const float* const* pcm, const uint64_t* n_samples, uint64_t count,
                                            float* probs_out, uint64_t cap, const float** device_audio_out) {
  if (device_audio_out == nullptr && count > 0) return MSH_ERR_INVALID_ARGUMENT;
  return silero_probabilities(s, pcm, n_samples, count, probs_out, cap, device_audio_out);
}
int64_t msh_silero_submit(msh_silero* s, const float* const* pcm, const uint64_t* n_samples, uint64_t count, int32_t keep_audio) {
  if (s == nullptr || s->dev == nullptr || (count > 0 && (pcm == nullptr || n_samples == nullptr))) return MSH_ERR_INVALID_ARGUMENT;
  try {
    return s->dev->submit(pcm, n_samples, (size_t)count, keep_audio != 0);
  } catch (const msh::HipError& ex) {
    s->last_error = ex.what();
    return MSH_ERR_HIP;
  } catch (const std::exception& ex) {
    s->last_error = ex.what();
    return MSH_ERR_INVALID_ARGUMENT;
  }
}
int64_t msh_silero_submit_pcm16(msh_silero* s, const int16_t* const* pcm16, const uint64_t* n_samples, uint64_t count, int32_t keep_audio) {
  if (s == nullptr || s->dev == nullptr || (count > 0 && (pcm16 == nullptr || n_samples == nullptr))) return MSH_ERR_INVALID_ARGUMENT;
  try {
    return s->dev->submit_pcm16(pcm16, n_samples, (size_t)count, keep_audio != 0);
  } catch (const msh::HipError& ex) {
    s->last_error = ex.what();
    return MSH_ERR_HIP;
  } catch (const std::exception& ex) {
    s->last_error = ex.what();
    return MSH_ERR_INVALID_ARGUMENT;
  }
}
int64_t msh_silero_collect(msh_silero* s, int64_t ticket, float* probs_out, uint64_t cap, const float** device_audio_out,
                           uint64_t count) {
  if (s == nullptr || s->dev == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  try {
    std::vector<float> probs;
    std::vector<const float*> resident;
    s->dev->collect(ticket, &probs, device_audio_out != nullptr ? &resident : nullptr);
    if (probs.size() > cap || (!probs.empty() && probs_out == nullptr) || (device_audio_out != nullptr && resident.size() != count)) {
      s->last_error = "msh_silero_collect: output arrays do not match the submission";
      return MSH_ERR_INVALID_ARGUMENT;
    }
    if (!probs.empty()) memcpy(probs_out, probs.data(), probs.size() * sizeof(float));
    for (size_t i = 0; i < resident.size(); ++i) device_audio_out[i] = resident[i];
    return (int64_t)probs.size();
  } catch (const msh::HipError& ex) {
    s->last_error = ex.what();
    return MSH_ERR_HIP;
  } catch (const std::exception& ex) {
    s->last_error = ex.what();
    return MSH_ERR_INVALID_ARGUMENT;
  }
}
int32_t msh_silero_release_audio(msh_silero* s) {
  if (s == nullptr || s->dev == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  s->dev->release_audio();
  return MSH_OK;
}
const char* msh_silero_last_error(msh_silero* s) { return s != nullptr ? s->last_error.c_str() : ""; }

// ---- streaming ----
static int32_t stream_create_impl(int32_t device, msh::SafeTensors& st, const char* config_json, int32_t max_slots,
                                  int32_t max_memory_frames, msh_stream_engine** out) {
  try {
    if (config_json == nullptr) throw std::invalid_argument("null config json");
    msh_stream_engine* e = new msh_stream_engine();
    try {
      e->eng = new msh::StreamingEngine(device, max_slots > 0 ? max_slots : 64,
                                        max_memory_frames > 0 ? max_memory_frames : 2048);
      e->eng->load(st, config_json);
    } catch (...) {
      delete e->eng;
      delete e;
      throw;
    }
    *out = e;
    return MSH_OK;
  } catch (const std::invalid_argument& ex) {
    g_create_error = ex.what();
    return MSH_ERR_INVALID_ARGUMENT;
  } catch (const msh::HipError& ex) {
    g_create_error = ex.what();
    return msh_device_count() == 0 ? MSH_ERR_NO_DEVICE : MSH_ERR_HIP;
  } catch (const std::exception& ex) {
    g_create_error = ex.what();
    return MSH_ERR_UNKNOWN;
  }
}

int32_t msh_stream_create(int32_t device, const char* path, const char* config_json, int32_t max_slots,
                          int32_t max_memory_frames, msh_stream_engine** out) {
  if (out == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  msh::SafeTensors st;
  try {
    if (path == nullptr) throw std::invalid_argument("null path");
    st.load_file(path);
  } catch (const std::exception& ex) {
    g_create_error = ex.what();
    return MSH_ERR_INVALID_ARGUMENT;
  }
  return stream_create_impl(device, st, config_json, max_slots, max_memory_frames, out);
}

int32_t msh_stream_create_from_memory(int32_t device, const void* data, uint64_t size, const char* config_json,
                                      int32_t max_slots, int32_t max_memory_frames, msh_stream_engine** out) {
  if (out == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  msh::SafeTensors st;
  try {
    if (data == nullptr) throw std::invalid_argument("null data");
    st.parse(reinterpret_cast<const uint8_t*>(data), (size_t)size);
  } catch (const std::exception& ex) {
    g_create_error = ex.what();
    return MSH_ERR_INVALID_ARGUMENT;
  }
  return stream_create_impl(device, st, config_json, max_slots, max_memory_frames, out);
}

void msh_stream_destroy(msh_stream_engine* e) {
  if (e == nullptr) return;
  try {
    delete e->eng;
  } catch (...) {
  }
  delete e;
}

const char* msh_stream_last_error(const msh_stream_engine* e) {
  if (e == nullptr) return g_create_error.c_str();
  return e->last_error.c_str();
}

int32_t msh_stream_info_get(const msh_stream_engine* e, msh_stream_info* out) {
  if (e == nullptr || out == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  const msh::StreamingConfig& c = e->eng->config();
  memset(out, 0, sizeof(*out));
  out->encoder_dim = c.encoder_dim;
  out->decoder_dim = c.decoder_dim;
  out->depth = c.depth;
  out->nheads = c.nheads;
  out->head_dim = c.head_dim;
  out->vocab_size = c.vocab_size;
  out->bos_id = c.bos_id;
  out->eos_id = c.eos_id;
  out->frame_len = c.frame_len;
  out->total_lookahead = c.total_lookahead;
  out->max_seq_len = c.max_seq_len;
  out->enc_layers = c.enc_layers;
  out->encoder_heads = c.encoder_heads;
  out->max_slots = e->eng->max_slots();
  out->memory_capacity = e->eng->memory_capacity();
  return MSH_OK;
}

int32_t msh_stream_open(msh_stream_engine* e) {
  int32_t slot = -1;
  int32_t rc = guarded(e, [&] { slot = e->eng->create_stream(); });
  return rc == MSH_OK ? slot : rc;
}
int32_t msh_stream_close(msh_stream_engine* e, int32_t slot) {
  return guarded(e, [&] { e->eng->free_stream(slot); });
}
int32_t msh_stream_reset(msh_stream_engine* e, int32_t slot) {
  return guarded(e, [&] { e->eng->reset_stream(slot); });
}
int32_t msh_stream_process_audio(msh_stream_engine* e, int32_t n, const int32_t* slots, const float* const* pcm,
                                 const uint64_t* n_samples, int32_t* features_out) {
  return guarded(e, [&] {
    if (n > 0 && (pcm == nullptr || n_samples == nullptr)) throw std::invalid_argument("null input");
    e->eng->process_audio(n, slots, pcm, n_samples, features_out);
  });
}
int32_t msh_stream_encode(msh_stream_engine* e, int32_t n, const int32_t* slots, const uint8_t* is_final,
                          int32_t* new_frames_out) {
  return guarded(e, [&] { e->eng->encode(n, slots, is_final, new_frames_out); });
}
int32_t msh_stream_decoder_reset(msh_stream_engine* e, int32_t n, const int32_t* slots) {
  return guarded(e, [&] { e->eng->decoder_reset(n, slots); });
}
int32_t msh_stream_decode_tokens(msh_stream_engine* e, int32_t n, const int32_t* slots, const int32_t* const* tokens,
                                 const int32_t* n_tokens, float* logits_out) {
  return guarded(e, [&] {
    if (n > 0 && (tokens == nullptr || n_tokens == nullptr)) throw std::invalid_argument("null input");
    e->eng->decode_tokens(n, slots, tokens, n_tokens, logits_out);
  });
}
int64_t msh_stream_cross_attention(msh_stream_engine* e, int32_t slot, const int32_t* tokens, int32_t n_tokens, float* out,
                                   uint64_t cap_floats, int32_t* dims3) {
  int dims[3] = {0, 0, 0};
  const int32_t rc = guarded(e, [&] {
    if (tokens == nullptr && n_tokens > 0) throw std::invalid_argument("null tokens");
    e->eng->cross_attention(slot, tokens, n_tokens, out, (size_t)cap_floats, dims);
  });
  if (rc != MSH_OK) return rc;
  if (dims3 != nullptr) dims3[0] = dims[0], dims3[1] = dims[1], dims3[2] = dims[2];
  return (int64_t)dims[0] * dims[1] * dims[2];
}
int32_t msh_stream_decode_full(msh_stream_engine* e, int32_t n, const int32_t* slots, const int32_t* const* drafts,
                               const int32_t* draft_lens, const int32_t* max_tokens, int32_t* tokens_out,
                               int32_t* counts_out, int32_t tokens_stride, int32_t* accepted_out) {
  return guarded(e, [&] {
    e->eng->decode_full(n, slots, drafts, draft_lens, max_tokens, tokens_out, counts_out, tokens_stride, accepted_out);
  });
}
int32_t msh_stream_set_bias(msh_stream_engine* e, int32_t n_nodes, const int32_t* child_off, const int32_t* child_tok,
                            const int32_t* child_node, const int32_t* depth, const float* depth_bonus,
                            int32_t n_depth_bonus) {
  return guarded(e, [&] { e->eng->set_bias(n_nodes, child_off, child_tok, child_node, depth, depth_bonus, n_depth_bonus); });
}
int32_t msh_stream_profile_enable(msh_stream_engine* e, int32_t on) {
  return guarded(e, [&] { e->eng->profile_enable(on != 0); });
}
int32_t msh_stream_profile_reset(msh_stream_engine* e) {
  return guarded(e, [&] { e->eng->profile_reset(); });
}
int32_t msh_stream_profile_count(msh_stream_engine* e) {
  int32_t n = 0;
  const int32_t rc = guarded(e, [&] {
    e->prof_cache = e->eng->profile_get();
    n = (int32_t)e->prof_cache.size();
  });
  return rc == MSH_OK ? n : rc;
}
int32_t msh_stream_profile_get(msh_stream_engine* e, int32_t index, msh_profile_entry* out) {
  if (e == nullptr || out == nullptr || index < 0 || index >= (int32_t)e->prof_cache.size()) return MSH_ERR_INVALID_ARGUMENT;
  const msh::ProfEntry& p = e->prof_cache[index];
  memset(out, 0, sizeof(*out));
  strncpy(out->name, p.name.c_str(), sizeof(out->name) - 1);
  out->ms = p.ms;
  out->launches = p.launches;
  out->flops = p.flops;
  out->bytes = p.bytes;
  return MSH_OK;
}
int32_t msh_stream_query(const msh_stream_engine* e, int32_t slot, int32_t what) {
  if (e == nullptr) return MSH_ERR_INVALID_ARGUMENT;
  try {
    switch (what) {
      case 0: return e->eng->memory_len(slot);
      case 1: return e->eng->feature_count(slot);
      case 2: return e->eng->cache_len(slot);
      case 3: return e->eng->frames_emitted(slot);
      case 4: return e->eng->max_tokens_for(slot);
      case 10: case 11: case 12: case 13: case 14: {   // decode_full statistics of the ENGINE (slot ignored), clamped to int32
        const long v = const_cast<msh::StreamingEngine*>(e->eng)->decode_stat(what - 10);
        return (int32_t)std::min<long>(v, 0x7fffffffL);
      }
      default: return MSH_ERR_INVALID_ARGUMENT;
    }
  } catch (const std::exception&) {
    return MSH_ERR_INVALID_ARGUMENT;
  }
}
int32_t msh_stream_get_memory(msh_stream_engine* e, int32_t slot, float* out) {
  return guarded(e, [&] {
    if (out == nullptr) throw std::invalid_argument("null output");
    e->eng->get_memory(slot, out);
  });
}
int32_t msh_stream_get_features(msh_stream_engine* e, int32_t slot, float* out) {
  return guarded(e, [&] {
    if (out == nullptr) throw std::invalid_argument("null output");
    e->eng->get_features(slot, out);
  });
}

int32_t msh_synchronize(msh_engine* e) {
  return guarded(e, [&] { e->eng->synchronize(); });
}

}  // extern "C"
