// C ABI over msh::Engine (include/moonshine_hip.h).  C++ exceptions never cross this boundary:
// every entry catches, records the message on the engine and returns a status code -- the rule the
// reference applies at its own C boundary (reference core/moonshine-c-api.cpp:439-446).
#include "../../include/moonshine_hip.h"

#include <stdio.h>
#include <string.h>

#include <string>

#include "engine.h"

struct msh_engine {
  msh::Engine* eng = nullptr;
  std::string last_error;
  std::vector<msh::ProfEntry> prof_cache;
};

namespace {
thread_local std::string g_create_error;

template <class F>
int32_t guarded(msh_engine* e, F&& f) {
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

float msh_test_gemm_microbench(int32_t M, int32_t N, int32_t K, int64_t lda, int32_t cfg, int32_t abl, int32_t iters) {
  try {
    return msh::gemm_microbench(M, N, K, lda, cfg, abl, iters);
  } catch (const std::exception& ex) {
    fprintf(stderr, "gemm_microbench: %s\n", ex.what());
    return -1.0f;
  }
}

int32_t msh_synchronize(msh_engine* e) {
  return guarded(e, [&] { e->eng->synchronize(); });
}

}  // extern "C"
