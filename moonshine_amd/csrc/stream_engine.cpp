#include "stream_engine.h"

#include <chrono>

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <stdexcept>

namespace msh {

namespace {

// Minimal reader for the flat streaming_config.json (the reference parses it the same way:
// first "key": then an integer, core/moonshine-streaming-model.cpp:75-117).
bool json_number(const std::string& j, const char* key, double* out) {
  const std::string k = std::string("\"") + key + "\"";
  size_t p = j.find(k);
  if (p == std::string::npos) return false;
  p = j.find(':', p + k.size());
  if (p == std::string::npos) return false;
  ++p;
  while (p < j.size() && (j[p] == ' ' || j[p] == '\t' || j[p] == '\n' || j[p] == '\r')) ++p;
  char* end = nullptr;
  const double v = strtod(j.c_str() + p, &end);
  if (end == j.c_str() + p) return false;
  *out = v;
  return true;
}
int json_int(const std::string& j, const char* key, int dflt) {
  double v;
  return json_number(j, key, &v) ? (int)v : dflt;
}
// "windows": [[16, 4], [16, 0], ...]
std::vector<std::pair<int, int>> json_windows(const std::string& j) {
  std::vector<std::pair<int, int>> out;
  size_t p = j.find("\"windows\"");
  if (p == std::string::npos) return out;
  p = j.find('[', p);
  if (p == std::string::npos) return out;
  int depth = 0;
  std::vector<int> nums;
  for (size_t i = p; i < j.size(); ++i) {
    const char c = j[i];
    if (c == '[') {
      ++depth;
    } else if (c == ']') {
      if (--depth == 0) break;
    } else if ((c >= '0' && c <= '9') || c == '-') {
      char* end = nullptr;
      nums.push_back((int)strtol(j.c_str() + i, &end, 10));
      i = (size_t)(end - j.c_str()) - 1;
    }
  }
  for (size_t i = 0; i + 1 < nums.size(); i += 2) out.emplace_back(nums[i], nums[i + 1]);
  return out;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

// ------------------------------------------------------------------------------------------------
StreamingEngine::StreamingEngine(int device, int max_slots, int max_memory_frames)
    : device_(device), max_slots_(max_slots), Mcap_(round_up(std::max(max_memory_frames, 64), 8)) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    throw HipError("no HIP device available: the MI355X engine has no CPU fallback (" +
                   std::string(hipGetErrorString(e)) + ")");
  if (device < 0 || device >= n) throw HipError("invalid device index " + std::to_string(device));
  if (max_slots_ <= 0 || max_slots_ > 4096) throw std::invalid_argument("max_slots out of range");
  if (Mcap_ > 4096) throw std::invalid_argument("memory capacity above 4096 frames (81 s) is not supported");
  MSH_HIP(hipSetDevice(device_));
  MSH_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  slots_.resize(max_slots_);
}

StreamingEngine::~StreamingEngine() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (ar_graph_ != nullptr) (void)hipGraphExecDestroy(ar_graph_);
  for (hipEvent_t ev : stat_ev_)
    if (ev != nullptr) (void)hipEventDestroy(ev);
  if (pin_ != nullptr) (void)hipHostFree(pin_);
  if (rb_ != nullptr) (void)hipHostFree(rb_);
  {
    std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
    for (void* p : allocs_) device_free(p);
  }
  DevBuf* bufs[] = {&audio_, &frames_, &hidden_, &c1out_, &feat_pk_, &segs_, &jobs_, &H_, &Y_, &Y32_, &QKV_, &AO_,
                    &Z_, &Q_, &rowlo_, &rowhi_, &newrows_, &newpos_, &newslot_, &newidx_, &adp16_, &adp32_, &mem16_,
                    &mem32_, &crosstmp_, &rowslot_, &rowpos_, &tokens_, &logits_, &pred_, &draft_, &decjobs_, &stepH_,
                    &steppos_, &probs_, &runs_, &pval_, &pidx_, &tiles_, &bias_off_, &bias_tok_, &bias_node_, &bias_depth_, &bias_bonus_, &bias_prefix_};
  for (DevBuf* b : bufs) b->release();
  if (stream_) (void)hipStreamDestroy(stream_);
}

void StreamingEngine::synchronize() {
  MSH_HIP(hipSetDevice(device_));
  MSH_HIP(hipStreamSynchronize(stream_));
}

void StreamingEngine::upload(const std::vector<float>& src, float** dst) {
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
    p = device_alloc(std::max<size_t>(src.size(), 4) * sizeof(float));
  }
  allocs_.push_back(p);
  copy_blocking(p, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice);
  *dst = reinterpret_cast<float*>(p);
}
void StreamingEngine::upload_bf16(const std::vector<float>& src, bf16_t** dst) {
  std::vector<bf16_t> tmp(src.size());
  for (size_t i = 0; i < src.size(); ++i) tmp[i] = f32_to_bf16(src[i]);
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
    p = device_alloc(tmp.size() * sizeof(bf16_t));
  }
  allocs_.push_back(p);
  copy_blocking(p, tmp.data(), tmp.size() * sizeof(bf16_t), hipMemcpyHostToDevice);
  *dst = reinterpret_cast<bf16_t*>(p);
}

// bf16 upload of a [rows][K] matrix in the MFMA-fragment-major order of the decode GEMMs (kernels.h fm16)
void StreamingEngine::upload_bf16_fm(const std::vector<float>& src, int rows, int K, bf16_t** dst) {
  if ((rows & 15) != 0 || (K & 31) != 0 || src.size() != (size_t)rows * K)
    throw std::runtime_error("decoder weight [" + std::to_string(rows) + ", " + std::to_string(K) + "] cannot be packed fragment-major");
  std::vector<bf16_t> tmp(src.size());
  const int ks = K >> 5;
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < K; ++k) tmp[(size_t)fm16(r, k, ks)] = f32_to_bf16(src[(size_t)r * K + k]);
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
    p = device_alloc(tmp.size() * sizeof(bf16_t));
  }
  allocs_.push_back(p);
  copy_blocking(p, tmp.data(), tmp.size() * sizeof(bf16_t), hipMemcpyHostToDevice);
  *dst = reinterpret_cast<bf16_t*>(p);
}

// Small descriptor arrays go host -> device on the engine stream.  The copy is issued from pageable memory
// and the stream is drained before the host vector dies: these arrays are a few KiB and every public call
// ends with a host-visible result anyway.
template <class T>
T* StreamingEngine::stage(DevBuf& buf, const std::vector<T>& host) {
  buf.reserve(std::max<size_t>(host.size(), 1) * sizeof(T));
  if (!host.empty()) {   // through the pinned ring: a truly asynchronous copy, and the host vector may die at once
    const size_t bytes = host.size() * sizeof(T);
    void* h = pin_take(bytes);
    memcpy(h, host.data(), bytes);
    MSH_HIP(hipMemcpyAsync(buf.p, h, bytes, hipMemcpyHostToDevice, stream_));
  }
  return buf.as<T>();
}

void* StreamingEngine::pin_take(size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes > pin_cap_ / 2) {   // (first use, or a transfer the ring was not sized for: the PCM of a long update)
    MSH_HIP(hipStreamSynchronize(stream_));
    if (pin_ != nullptr) (void)hipHostFree(pin_);
    pin_ = nullptr;
    pin_cap_ = std::max<size_t>((size_t)8 << 20, 4 * bytes);
    MSH_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin_), pin_cap_, hipHostMallocDefault));
    pin_off_ = pin_live_ = 0;
  }
  if (pin_off_ + bytes > pin_cap_) {   // wrap: the skipped tail counts as handed out
    pin_live_ += pin_cap_ - pin_off_;
    pin_off_ = 0;
  }
  if (pin_live_ + bytes > pin_cap_) {   // the slice would overlap one a queued copy may still read: drain first
    MSH_HIP(hipStreamSynchronize(stream_));
    pin_live_ = 0;
  }
  void* p = pin_ + pin_off_;
  pin_off_ += bytes;
  pin_live_ += bytes;
  return p;
}

void* StreamingEngine::rb_area(size_t bytes) {
  if (bytes > rb_cap_) {
    MSH_HIP(hipStreamSynchronize(stream_));
    if (rb_ != nullptr) (void)hipHostFree(rb_);
    rb_ = nullptr;
    rb_cap_ = (bytes + 4095) & ~(size_t)4095;
    MSH_HIP(hipHostMalloc(reinterpret_cast<void**>(&rb_), rb_cap_, hipHostMallocDefault));
  }
  return rb_;
}

const StreamingEngine::SlotHost& StreamingEngine::st(int slot) const {
  if (slot < 0 || slot >= max_slots_ || !slots_[slot].used) throw std::invalid_argument("invalid stream slot");
  return slots_[slot];
}
StreamingEngine::SlotHost& StreamingEngine::st(int slot) {
  if (slot < 0 || slot >= max_slots_ || !slots_[slot].used) throw std::invalid_argument("invalid stream slot");
  return slots_[slot];
}
void StreamingEngine::check_slots(int n, const int* slots) const {
  if (n < 0 || (n > 0 && slots == nullptr)) throw std::invalid_argument("null slot list");
  std::vector<char> seen(max_slots_, 0);
  for (int i = 0; i < n; ++i) {
    (void)st(slots[i]);
    if (seen[slots[i]]) throw std::invalid_argument("a stream appears twice in one call");
    seen[slots[i]] = 1;
  }
}

// ------------------------------------------------------------------------------------------------
// Weights: HuggingFace MoonshineStreamingForConditionalGeneration state_dict
// (modeling_moonshine_streaming.py:283-296 embedder, :185-207 / :133-139 encoder layer, :595-604 / :455-456
// decoder layer, :782-794 decoder, :1005-1010 head).  Layout changes made once, here:
//   embedder.linear [De,80]      -> [De][96] (zero-padded K)
//   conv1 [2De,De,5] / conv2 [De,2De,5] -> tap-major [Cout][5][Cin]: a window of 5 channels-last rows is one
//                                   contiguous K = 5*Cin vector, so the causal convs are strided GEMMs
//   LayerNorm gamma (unit offset) -> gamma + 1
//   q,k,v -> fused [3D][D]; decoder fc1 rows interleaved (value_j, gate_j); cross k,v of all layers -> [L*2Dd][Dd]
// ------------------------------------------------------------------------------------------------
void StreamingEngine::load(const SafeTensors& st, const std::string& json) {
  MSH_HIP(hipSetDevice(device_));
  if (loaded_) throw std::runtime_error("weights already loaded");
  StreamingConfig c;
  c.encoder_dim = json_int(json, "encoder_dim", 0);
  c.decoder_dim = json_int(json, "decoder_dim", 0);
  c.depth = json_int(json, "depth", 0);
  c.nheads = json_int(json, "nheads", 0);
  c.head_dim = json_int(json, "head_dim", 0);
  c.vocab_size = json_int(json, "vocab_size", 0);
  c.bos_id = json_int(json, "bos_id", 1);
  c.eos_id = json_int(json, "eos_id", 2);
  c.frame_len = json_int(json, "frame_len", 80);
  c.total_lookahead = json_int(json, "total_lookahead", -1);
  const int msl = json_int(json, "max_seq_len", 0);
  c.max_seq_len = msl > 0 ? msl : 448;  // streaming-model.cpp:111-113
  c.encoder_heads = json_int(json, "encoder_heads", c.nheads);
  double v;
  if (json_number(json, "rope_theta", &v)) c.rope_theta = (float)v;
  if (json_number(json, "partial_rotary_factor", &v)) c.partial_rotary = (float)v;
  c.windows = json_windows(json);
  if (c.depth <= 0 || c.decoder_dim <= 0 || c.vocab_size <= 0 || c.encoder_dim <= 0 || c.nheads <= 0)
    throw std::runtime_error("streaming_config.json: missing depth / dims / vocab_size");

  const int De = c.encoder_dim, Dd = c.decoder_dim, V = c.vocab_size, L = c.depth;
  auto shape_is = [&](const std::string& name, std::vector<int64_t> shape) {
    if (st.get(name).shape != shape) throw std::runtime_error("unexpected shape for " + name);
  };
  auto vec = [&](const std::string& name, int64_t n) {  // 1-D parameters are checked like the matrices
    shape_is(name, {n});
    return st.to_f32(name);
  };
  while (st.has("model.encoder.layers." + std::to_string(c.enc_layers) + ".mlp.fc1.weight")) ++c.enc_layers;
  int dec_layers = 0;
  while (st.has("model.decoder.layers." + std::to_string(dec_layers) + ".mlp.fc1.weight")) ++dec_layers;
  if (dec_layers != L) throw std::runtime_error("decoder layer count differs from streaming_config depth");
  if (c.enc_layers == 0) throw std::runtime_error("no encoder layers in the checkpoint");
  c.enc_ffn = (int)st.get("model.encoder.layers.0.mlp.fc1.weight").shape[0];
  c.dec_ffn = (int)st.get("model.decoder.layers.0.mlp.fc2.weight").shape[1];
  c.max_pos = (int)st.get("model.decoder.pos_emb.weight").shape[0];
  if (c.windows.empty()) {
    // the reference's json does not carry the windows (they are baked into encoder.onnx); HF default pattern
    if (c.enc_layers != 6) throw std::runtime_error("streaming_config.json needs a \"windows\" array for this encoder depth");
    c.windows = {{16, 4}, {16, 4}, {16, 0}, {16, 0}, {16, 4}, {16, 4}};
  }
  if ((int)c.windows.size() != c.enc_layers) throw std::runtime_error("windows array does not match the encoder depth");
  int look = 0;
  for (auto& w : c.windows) look += w.second;
  if (c.total_lookahead < 0) c.total_lookahead = look;
  if (c.total_lookahead != look) throw std::runtime_error("total_lookahead disagrees with the windows array");
  if (c.head_dim * c.nheads != Dd) throw std::runtime_error("nheads * head_dim != decoder_dim");
  if (c.frame_len != 80) throw std::runtime_error("frame_len other than 80 is not supported");
  if (De % 32 || Dd % 32 || c.enc_ffn % 32 || c.dec_ffn % 32 || V % 4 || De % c.encoder_heads || (c.head_dim & 3) ||
      ((De / c.encoder_heads) & 3) || c.head_dim > 128 || De / c.encoder_heads > 128 || De > 1024 || Dd > 1024)
    throw std::runtime_error("unsupported streaming model dimensions");
  Scap_ = round_up(c.max_seq_len + 8, 8);
  if (Scap_ > 512) throw std::runtime_error("max_seq_len above 504 is not supported");
  cfg_ = c;
  const int Fe = c.enc_ffn, Fd = c.dec_ffn;

  {  // frontend
    k_scale_ = expf(st.to_f32("model.encoder.embedder.comp.log_k").at(0));
    shape_is("model.encoder.embedder.linear.weight", {De, 80});
    std::vector<float> w = st.to_f32("model.encoder.embedder.linear.weight"), r((size_t)De * 96, 0.f);
    for (int n = 0; n < De; ++n)
      for (int k = 0; k < 80; ++k) r[(size_t)n * 96 + k] = w[(size_t)n * 80 + k];
    upload_bf16(r, &lin_w_);
    shape_is("model.encoder.embedder.conv1.weight", {2 * De, De, 5});
    w = st.to_f32("model.encoder.embedder.conv1.weight");
    r.assign((size_t)2 * De * 5 * De, 0.f);
    for (int n = 0; n < 2 * De; ++n)
      for (int ch = 0; ch < De; ++ch)
        for (int k = 0; k < 5; ++k) r[((size_t)n * 5 + k) * De + ch] = w[((size_t)n * De + ch) * 5 + k];
    upload_bf16(r, &conv1_w_);
    shape_is("model.encoder.embedder.conv2.weight", {De, 2 * De, 5});
    w = st.to_f32("model.encoder.embedder.conv2.weight");
    r.assign((size_t)De * 5 * 2 * De, 0.f);
    for (int n = 0; n < De; ++n)
      for (int ch = 0; ch < 2 * De; ++ch)
        for (int k = 0; k < 5; ++k) r[((size_t)n * 5 + k) * 2 * De + ch] = w[((size_t)n * 2 * De + ch) * 5 + k];
    upload_bf16(r, &conv2_w_);
    upload(vec("model.encoder.embedder.conv1.bias", 2 * De), &conv1_b_);
    upload(vec("model.encoder.embedder.conv2.bias", De), &conv2_b_);
  }
  auto cat = [&](std::initializer_list<std::string> names, int rows, int cols) {
    std::vector<float> out;
    for (const std::string& n : names) {
      shape_is(n, {rows, cols});
      std::vector<float> w = st.to_f32(n);
      out.insert(out.end(), w.begin(), w.end());
    }
    return out;
  };
  auto gamma1 = [&](const std::string& name) {
    std::vector<float> g = vec(name, De);
    for (float& x : g) x += 1.0f;  // unit_offset LayerNorm, modeling_moonshine_streaming.py:127-130
    return g;
  };
  enc_.resize(c.enc_layers);
  for (int l = 0; l < c.enc_layers; ++l) {
    const std::string p = "model.encoder.layers." + std::to_string(l) + ".";
    EncW& E = enc_[l];
    upload_bf16(cat({p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"}, De, De),
                &E.wqkv);
    upload_bf16(cat({p + "self_attn.o_proj.weight"}, De, De), &E.wo);
    upload_bf16(cat({p + "mlp.fc1.weight"}, Fe, De), &E.fc1);
    upload_bf16(cat({p + "mlp.fc2.weight"}, De, Fe), &E.fc2);
    upload(vec(p + "mlp.fc1.bias", Fe), &E.b1);
    upload(vec(p + "mlp.fc2.bias", De), &E.b2);
    upload(gamma1(p + "input_layernorm.gamma"), &E.ln1);
    upload(gamma1(p + "post_attention_layernorm.gamma"), &E.ln2);
  }
  upload(gamma1("model.encoder.final_norm.gamma"), &enc_ln_);
  shape_is("model.decoder.pos_emb.weight", {c.max_pos, De});
  upload(st.to_f32("model.decoder.pos_emb.weight"), &pos_emb_);
  if (st.has("model.decoder.proj.weight")) {
    upload_bf16(cat({"model.decoder.proj.weight"}, Dd, De), &proj_w_);
  } else if (De != Dd) {
    throw std::runtime_error("encoder_dim != decoder_dim but model.decoder.proj.weight is missing");
  }
  {
    shape_is("model.decoder.embed_tokens.weight", {V, Dd});
    std::vector<float> e = st.to_f32("model.decoder.embed_tokens.weight");
    upload(e, &embed_f32_);
    // untied head when the checkpoint has one (lora/export.py:198-203), else the embedding
    std::vector<float> head = st.has("proj_out.weight") ? cat({"proj_out.weight"}, V, Dd) : e;
    upload_bf16(head, &head_w_);
    const std::vector<float> gn = vec("model.decoder.norm.weight", Dd);
    upload(gn, &dec_ln_);
    for (int v = 0; v < V; ++v)
      for (int d = 0; d < Dd; ++d) head[(size_t)v * Dd + d] *= gn[d];
    upload_bf16(head, &head_wf_);
  }
  dec_.resize(L);
  {
    const char* e = dev_getenv("MSH_STREAM_FM");   // developer switch: 0 = AR steps on the row-major operands
    fm_ok_ = !(e != nullptr && e[0] == '0') && stream_fm_supported(Dd, Fd);
  }
  std::vector<float> cross;
  for (int l = 0; l < L; ++l) {
    const std::string p = "model.decoder.layers." + std::to_string(l) + ".";
    DecW& W = dec_[l];
    // LN(x) * W^T = ((x - mu) * rstd) * (W * diag(gamma))^T: folded copies for the LN-fused small-batch kernels
    auto fold = [&](std::vector<float> wmat, const std::string& ln_name, int rows) {
      const std::vector<float> gam = vec(ln_name, Dd);
      for (int r = 0; r < rows; ++r)
        for (int d = 0; d < Dd; ++d) wmat[(size_t)r * Dd + d] *= gam[d];
      return wmat;
    };
    {
      std::vector<float> qkv = cat({p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"}, Dd, Dd);
      upload_bf16(qkv, &W.wqkv);
      const std::vector<float> qkv_f = fold(qkv, p + "input_layernorm.weight", 3 * Dd);
      upload_bf16(qkv_f, &W.wqkv_f);
      std::vector<float> qc = cat({p + "encoder_attn.q_proj.weight"}, Dd, Dd);
      upload_bf16(qc, &W.wq_c);
      const std::vector<float> qc_f = fold(qc, p + "post_attention_layernorm.weight", Dd);
      upload_bf16(qc_f, &W.wq_c_f);
      if (fm_ok_) {
        upload_bf16_fm(qkv_f, 3 * Dd, Dd, &W.wqkv_fm);
        upload_bf16_fm(qc_f, Dd, Dd, &W.wq_c_fm);
      }
    }
    {
      const std::vector<float> wo = cat({p + "self_attn.o_proj.weight"}, Dd, Dd), woc = cat({p + "encoder_attn.o_proj.weight"}, Dd, Dd);
      upload_bf16(wo, &W.wo);
      upload_bf16(woc, &W.wo_c);
      if (fm_ok_) {
        upload_bf16_fm(wo, Dd, Dd, &W.wo_fm);
        upload_bf16_fm(woc, Dd, Dd, &W.wo_c_fm);
      }
    }
    std::vector<float> kv = cat({p + "encoder_attn.k_proj.weight", p + "encoder_attn.v_proj.weight"}, Dd, Dd);
    cross.insert(cross.end(), kv.begin(), kv.end());
    shape_is(p + "mlp.fc1.weight", {2 * Fd, Dd});
    std::vector<float> f1 = st.to_f32(p + "mlp.fc1.weight"), b1 = vec(p + "mlp.fc1.bias", 2 * Fd);
    std::vector<float> f1i((size_t)2 * Fd * Dd), b1i((size_t)2 * Fd);
    for (int j = 0; j < Fd; ++j) {  // chunk(2): first half value, second half gate (modeling_moonshine_streaming.py:459-461)
      memcpy(&f1i[(size_t)(2 * j) * Dd], &f1[(size_t)j * Dd], Dd * sizeof(float));
      memcpy(&f1i[(size_t)(2 * j + 1) * Dd], &f1[(size_t)(Fd + j) * Dd], Dd * sizeof(float));
      b1i[2 * j] = b1[j];
      b1i[2 * j + 1] = b1[Fd + j];
    }
    upload_bf16(f1i, &W.fc1);
    const std::vector<float> f1f = fold(f1i, p + "final_layernorm.weight", 2 * Fd);
    upload_bf16(f1f, &W.fc1_f);
    upload(b1i, &W.b1);
    const std::vector<float> f2 = cat({p + "mlp.fc2.weight"}, Dd, Fd);
    upload_bf16(f2, &W.fc2);
    if (fm_ok_) {
      upload_bf16_fm(f1f, 2 * Fd, Dd, &W.fc1_fm);
      upload_bf16_fm(f2, Dd, Fd, &W.fc2_fm);
    }
    upload(vec(p + "mlp.fc2.bias", Dd), &W.b2);
    upload(vec(p + "input_layernorm.weight", Dd), &W.ln1);
    upload(vec(p + "post_attention_layernorm.weight", Dd), &W.ln2);
    upload(vec(p + "final_layernorm.weight", Dd), &W.ln3);
  }
  upload_bf16(cross, &cross_w_);
  {  // RoPE tables: inv_freq over dim = int(head_dim * factor); ceil(dim / 2) rotated pairs
     // (modeling_moonshine_streaming.py:498-506, 552-557)
    const int dim = (int)(c.head_dim * c.partial_rotary);
    rot_pairs_ = (dim + 1) / 2;
    std::vector<float> cs((size_t)Scap_ * rot_pairs_), sn((size_t)Scap_ * rot_pairs_);
    for (int j = 0; j < rot_pairs_; ++j) {
      const float inv = 1.0f / powf(c.rope_theta, (float)(2 * j) / (float)dim);
      for (int pos = 0; pos < Scap_; ++pos) {
        const float a = (float)pos * inv;
        cs[(size_t)pos * rot_pairs_ + j] = cosf(a);
        sn[(size_t)pos * rot_pairs_ + j] = sinf(a);
      }
    }
    upload(cs, &rope_cos_);
    upload(sn, &rope_sin_);
  }
  // per-slot state slabs
  auto slab = [&](size_t bytes) {
    void* p = nullptr;
    {
      std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
      p = device_alloc(bytes);
    }
    allocs_.push_back(p);
    zero_blocking(p, bytes);  // complete before anything on the engine stream touches the slab (see DevBuf::reserve)
    return p;
  };
  const size_t S = (size_t)max_slots_;
  conv1_buf_ = (bf16_t*)slab(S * 4 * De * 2);
  conv2_buf_ = (bf16_t*)slab(S * 4 * 2 * De * 2);
  features_ = (float*)slab(S * Mcap_ * De * 4);
  memory_ = (float*)slab(S * Mcap_ * Dd * 4);
  crossK_ = (bf16_t*)slab(S * L * Mcap_ * Dd * 2);
  crossV_ = (bf16_t*)slab(S * L * Mcap_ * Dd * 2);
  selfK_ = (bf16_t*)slab(S * L * Scap_ * Dd * 2);
  selfV_ = (bf16_t*)slab(S * L * Scap_ * Dd * 2);
  result_ = (int32_t*)slab(S * Scap_ * 4);
  slots_d_ = (SlotDev*)slab(S * sizeof(SlotDev));
  n_active_d_ = (int32_t*)slab(64);
  loaded_ = true;
}

// ------------------------------------------------------------------------------------------------
int StreamingEngine::create_stream() {
  if (!loaded_) throw std::runtime_error("weights not loaded");
  for (int s = 0; s < max_slots_; ++s)
    if (!slots_[s].used) {
      slots_[s].used = true;
      reset_stream(s);
      return s;
    }
  throw std::runtime_error("all " + std::to_string(max_slots_) + " stream slots are in use");
}
void StreamingEngine::free_stream(int slot) { st(slot).used = false; }

void StreamingEngine::reset_stream(int slot) {
  SlotHost& h = st(slot);
  h.pending.clear();
  h.feat_count = h.emitted = h.pos_offset = h.mem_len = h.cache_len = 0;
  MSH_HIP(hipSetDevice(device_));
  const int De = cfg_.encoder_dim;
  MSH_HIP(hipMemsetAsync(conv1_buf_ + (size_t)slot * 4 * De, 0, (size_t)4 * De * 2, stream_));
  MSH_HIP(hipMemsetAsync(conv2_buf_ + (size_t)slot * 8 * De, 0, (size_t)8 * De * 2, stream_));
  MSH_HIP(hipMemsetAsync(slots_d_ + slot, 0, sizeof(SlotDev), stream_));
}

int StreamingEngine::max_tokens_for(int slot) const {
  // streaming-model.cpp:1217-1219: float duration, double product, ceil, capped by max_seq_len
  const float duration = (float)st(slot).mem_len * 0.020f;
  return std::min((int)ceil((double)duration * 6.5), cfg_.max_seq_len);
}

void StreamingEngine::get_memory(int slot, float* out) {
  const SlotHost& h = st(slot);
  MSH_HIP(hipSetDevice(device_));
  MSH_HIP(hipStreamSynchronize(stream_));
  copy_blocking(out, memory_ + (size_t)slot * Mcap_ * cfg_.decoder_dim,
                    (size_t)h.mem_len * cfg_.decoder_dim * sizeof(float), hipMemcpyDeviceToHost);
}
void StreamingEngine::get_features(int slot, float* out) {
  const SlotHost& h = st(slot);
  MSH_HIP(hipSetDevice(device_));
  MSH_HIP(hipStreamSynchronize(stream_));
  copy_blocking(out, features_ + (size_t)slot * Mcap_ * cfg_.encoder_dim,
                    (size_t)h.feat_count * cfg_.encoder_dim * sizeof(float), hipMemcpyDeviceToHost);
}

// ------------------------------------------------------------------------------------------------
// Frontend (lora/export.py:67-97, driver streaming-model.cpp:441-602).  A "period" is 320 samples = 4 frames
// = 2 conv1 rows = 1 feature row.  Stream i gets periods [b_i, b_i + n_i + 2) of one packed row stream whose
// first two periods hold the carried context, so that every stage is a uniform-stride GEMM over all streams:
//   hidden  row 4b+8+f   = frame f          rows 4b+4..4b+7 = conv1_buf (previous 4 frames)
//   conv1   row 2b+4+c   = conv1 output c   rows 2b..2b+3   = conv2_buf (previous 4 outputs)
//   feature row  b+2+m   = new feature m
// conv1 row R reads hidden rows [2R-4, 2R], conv2 row R reads conv1 rows [2R-4, 2R]  (kernel 5, stride 2,
// left pad 4).  Samples that do not fill a period wait in `pending` (the reference keeps < 80 of them in
// sample_buffer and is only ever fed 1280-sample chunks, transcriber.cpp:1341-1357).
// ------------------------------------------------------------------------------------------------
void StreamingEngine::process_audio(int n, const int* slots, const float* const* pcm, const uint64_t* lens,
                                    int* features_out) {
  if (!loaded_) throw std::runtime_error("weights not loaded");
  check_slots(n, slots);
  MSH_HIP(hipSetDevice(device_));
  const int De = cfg_.encoder_dim;
  struct Job {
    int slot, n_per, base;
    long audio_off;
  };
  std::vector<Job> jobs;
  std::vector<float> audio;
  int P = 0, max_frames = 0;
  for (int i = 0; i < n; ++i) {
    SlotHost& h = st(slots[i]);
    if (lens[i] > 0 && pcm[i] == nullptr) throw std::invalid_argument("null audio pointer");
    h.pending.insert(h.pending.end(), pcm[i], pcm[i] + lens[i]);
    const int n_per = (int)(h.pending.size() / 320);
    if (features_out) features_out[i] = n_per;
    if (n_per == 0) continue;
    if (h.feat_count + n_per > Mcap_)
      throw std::runtime_error("stream exceeds the engine's memory capacity of " + std::to_string(Mcap_) + " frames");
    jobs.push_back({slots[i], n_per, P, (long)audio.size()});
    audio.insert(audio.end(), h.pending.begin(), h.pending.begin() + (size_t)n_per * 320);
    h.pending.erase(h.pending.begin(), h.pending.begin() + (size_t)n_per * 320);
    P += n_per + 2;
    max_frames = std::max(max_frames, 4 * n_per);
  }
  if (jobs.empty()) return;
  frames_.reserve((size_t)4 * P * 96 * 2);
  hidden_.reserve((size_t)4 * P * De * 2);
  c1out_.reserve((size_t)2 * P * 2 * De * 2);
  feat_pk_.reserve((size_t)P * De * 4);
  const float* audio_d = stage(audio_, audio);
  std::vector<FrameJob> fj;
  std::vector<StreamSeg> segA, segB, segC;
  bf16_t* hid = hidden_.as<bf16_t>();
  bf16_t* c1o = c1out_.as<bf16_t>();
  float* fpk = feat_pk_.as<float>();
  for (const Job& j : jobs) {
    const long b = j.base;
    const int nf = 4 * j.n_per, nc = 2 * j.n_per;
    fj.push_back({j.audio_off, nf, (int)(4 * b + 8)});
    bf16_t* c1b = conv1_buf_ + (size_t)j.slot * 4 * De;
    bf16_t* c2b = conv2_buf_ + (size_t)j.slot * 8 * De;
    segA.push_back({c1b, hid + (4 * b + 4) * De, (long)4 * De * 2});
    segB.push_back({hid + (4 * b + 8 + nf - 4) * De, c1b, (long)4 * De * 2});
    segB.push_back({c2b, c1o + (2 * b) * 2 * De, (long)8 * De * 2});
    segC.push_back({c1o + (2 * b + 4 + nc - 4) * 2 * De, c2b, (long)8 * De * 2});
    SlotHost& h = st(j.slot);
    segC.push_back({fpk + (b + 2) * De, features_ + ((size_t)j.slot * Mcap_ + h.feat_count) * De,
                    (long)j.n_per * De * 4});
    h.feat_count += j.n_per;
  }
  const FrameJob* fj_d = stage(jobs_, fj);
  std::vector<StreamSeg> all(segA);
  all.insert(all.end(), segB.begin(), segB.end());
  all.insert(all.end(), segC.begin(), segC.end());
  const StreamSeg* segs_d = stage(segs_, all);
  {
    // algorithmic: 80 -> De linear on 4P frames, conv1 (k5) on 2P rows, conv2 (k5) on P rows; bytes = samples in, features out
    ScopeProfiler::Scope sc(&prof_, stream_, "stream_frontend",
                            2.0 * 4 * P * De * 80 + 2.0 * 2 * P * 2 * De * 5 * De + 2.0 * P * De * 10 * De,
                            (double)P * 320 * 4 + (double)P * De * 4);
    stream_frames(audio_d, fj_d, (int)fj.size(), max_frames, k_scale_, frames_.as<bf16_t>(), stream_);
    gemm_act(frames_.as<bf16_t>(), 96, lin_w_, nullptr, 1, 4 * P, De, 96, hid, nullptr, stream_);
    copy_segments(segs_d, (int)segA.size(), stream_);
    gemm_act(hid, 2 * De, conv1_w_, conv1_b_, 1, 2 * P - 2, 2 * De, 5 * De, c1o + (size_t)2 * 2 * De, nullptr, stream_);
    copy_segments(segs_d + segA.size(), (int)segB.size(), stream_);
    gemm_act(c1o, 4 * De, conv2_w_, conv2_b_, 0, P - 2, De, 10 * De, nullptr, fpk + (size_t)2 * De, stream_);
    copy_segments(segs_d + segA.size() + segB.size(), (int)segC.size(), stream_);
  }
}

// ------------------------------------------------------------------------------------------------
// Encoder window + adapter + cross K/V (streaming-model.cpp:604-772 and :779-860).
// ------------------------------------------------------------------------------------------------
void StreamingEngine::encode(int n, const int* slots, const uint8_t* is_final, int* new_frames_out) {
  if (!loaded_) throw std::runtime_error("weights not loaded");
  check_slots(n, slots);
  MSH_HIP(hipSetDevice(device_));
  const int De = cfg_.encoder_dim, Dd = cfg_.decoder_dim, L = cfg_.depth, Fe = cfg_.enc_ffn;
  struct Job {
    int slot, start, total, stable, fresh, r0;
  };
  std::vector<Job> jobs;
  int R = 0, Nn = 0;
  for (int i = 0; i < n; ++i) {
    SlotHost& h = st(slots[i]);
    if (new_frames_out) new_frames_out[i] = 0;
    const int total = h.feat_count;
    if (total == 0) continue;
    const int stable = (is_final && is_final[i]) ? total : std::max(0, total - cfg_.total_lookahead);
    const int fresh = stable - h.emitted;
    if (fresh <= 0) continue;
    const int start = std::max(0, h.emitted - 16 * L);  // streaming-model.cpp:638-642 (16 * depth)
    if (h.pos_offset + fresh > cfg_.max_pos) throw std::runtime_error("stream exceeds max_position_embeddings");
    jobs.push_back({slots[i], start, total, stable, fresh, R});
    R += total - start;
    Nn += fresh;
    if (new_frames_out) new_frames_out[i] = fresh;
  }
  if (jobs.empty()) return;
  H_.reserve((size_t)R * De * 4);
  Y_.reserve((size_t)R * std::max(De, Dd) * 2);
  Y32_.reserve((size_t)R * De * 4);
  QKV_.reserve((size_t)R * 3 * std::max(De, Dd) * 2);
  AO_.reserve((size_t)R * std::max(De, Dd) * 2);
  Z_.reserve((size_t)R * std::max(Fe, cfg_.dec_ffn) * 2);
  adp16_.reserve((size_t)Nn * De * 2);
  adp32_.reserve((size_t)Nn * De * 4);
  mem16_.reserve((size_t)Nn * Dd * 2);
  mem32_.reserve((size_t)Nn * Dd * 4);
  crosstmp_.reserve((size_t)Nn * L * 2 * Dd * 2);
  std::vector<StreamSeg> gather, memseg;
  std::vector<int> lo(R), hi(R), nrow(Nn), npos(Nn), nslot(Nn), nidx(Nn), tiles;
  std::vector<int4> upd;
  float* H = H_.as<float>();
  int k = 0;
  for (const Job& j : jobs) {
    SlotHost& h = st(j.slot);
    const int W = j.total - j.start;
    gather.push_back({features_ + ((size_t)j.slot * Mcap_ + j.start) * De, H + (size_t)j.r0 * De, (long)W * De * 4});
    for (int r = 0; r < W; ++r) {
      lo[j.r0 + r] = j.r0;
      hi[j.r0 + r] = j.r0 + W;
    }
    for (int t0 = 0; t0 < W; t0 += 16) tiles.push_back(j.r0 + t0);   // window attention: 16-row tiles from the stream's first row
    memseg.push_back({mem32_.as<float>() + (size_t)k * Dd, memory_ + ((size_t)j.slot * Mcap_ + h.mem_len) * Dd,
                      (long)j.fresh * Dd * 4});
    for (int i = 0; i < j.fresh; ++i, ++k) {
      nrow[k] = j.r0 + (h.emitted - j.start) + i;
      npos[k] = h.pos_offset + i;
      nslot[k] = j.slot;
      nidx[k] = h.mem_len + i;
    }
    h.mem_len += j.fresh;
    h.emitted = j.stable;
    h.pos_offset += j.fresh;
    upd.push_back(make_int4(j.slot, h.mem_len, -1, 0));
  }
  std::vector<StreamSeg> all(gather);
  all.insert(all.end(), memseg.begin(), memseg.end());
  const StreamSeg* segs_d = stage(segs_, all);
  const int* lo_d = stage(rowlo_, lo);
  const int* hi_d = stage(rowhi_, hi);
  const int* tiles_d = stage(tiles_, tiles);
  const int* nrow_d = stage(newrows_, nrow);
  const int* npos_d = stage(newpos_, npos);
  const int* nslot_d = stage(newslot_, nslot);
  const int* nidx_d = stage(newidx_, nidx);
  const int4* upd_d = stage(jobs_, upd);

  copy_segments(segs_d, (int)gather.size(), stream_);
  bf16_t* Y = Y_.as<bf16_t>();
  bf16_t* QKV = QKV_.as<bf16_t>();
  bf16_t* AO = AO_.as<bf16_t>();
  bf16_t* Z = Z_.as<bf16_t>();
  using Sc = ScopeProfiler::Scope;
  for (int l = 0; l < cfg_.enc_layers; ++l) {
    const EncW& E = enc_[l];
    const double rd = (double)R * De;
    {
      Sc sc(&prof_, stream_, "senc_layernorm", 0, rd * 6);
      layernorm_bf16(H, E.ln1, R, De, Y, nullptr, stream_);
    }
    {
      Sc sc(&prof_, stream_, "senc_qkv_gemm", 2.0 * rd * 3 * De, rd * 2 * 4 + 6.0 * De * De);
      gemm_act(Y, De, E.wqkv, nullptr, 0, R, 3 * De, De, QKV, nullptr, stream_);
    }
    {
      const double keys = cfg_.windows[l].first + cfg_.windows[l].second + 1;
      Sc sc(&prof_, stream_, "senc_window_attention", 4.0 * rd * keys, rd * 2 * 4);
      stream_enc_attention(QKV, lo_d, hi_d, R, De, cfg_.encoder_heads, cfg_.windows[l].first, cfg_.windows[l].second, AO,
                           stream_, tiles_d, (int)tiles.size());
    }
    {
      Sc sc(&prof_, stream_, "senc_oproj_gemm", 2.0 * rd * De, rd * (2 + 8) + 2.0 * De * De);
      gemm_resid_f32(AO, De, E.wo, nullptr, R, De, De, H, stream_);
    }
    {
      Sc sc(&prof_, stream_, "senc_layernorm", 0, rd * 6);
      layernorm_bf16(H, E.ln2, R, De, Y, nullptr, stream_);
    }
    {
      Sc sc(&prof_, stream_, "senc_fc1_gelu_gemm", 2.0 * rd * Fe, (double)R * (De + Fe) * 2 + 2.0 * De * Fe);
      gemm_bias_gelu_bf16(Y, De, E.fc1, E.b1, R, Fe, De, Z, stream_);
    }
    {
      Sc sc(&prof_, stream_, "senc_fc2_gemm", 2.0 * rd * Fe, (double)R * (Fe * 2 + De * 8) + 2.0 * De * Fe);
      gemm_resid_f32(Z, Fe, E.fc2, E.b2, R, De, Fe, H, stream_);
    }
  }
  Sc sc_tail(&prof_, stream_, "stream_adapter_cross_kv", 2.0 * Nn * Dd * 2.0 * L * Dd + (proj_w_ != nullptr ? 2.0 * Nn * De * Dd : 0.0),
             (double)Nn * Dd * (2 + 4.0 * L) + 4.0 * L * Dd * Dd);
  layernorm_bf16(H, enc_ln_, R, De, Y, Y32_.as<float>(), stream_);
  stream_adapter_in(Y32_.as<float>(), nrow_d, npos_d, Nn, De, pos_emb_, adp16_.as<bf16_t>(), adp32_.as<float>(),
                    stream_);
  const bf16_t* mem16 = adp16_.as<bf16_t>();
  if (proj_w_ != nullptr) {
    gemm_act(adp16_.as<bf16_t>(), De, proj_w_, nullptr, 0, Nn, Dd, De, mem16_.as<bf16_t>(), mem32_.as<float>(), stream_);
    mem16 = mem16_.as<bf16_t>();
  } else {
    MSH_HIP(hipMemcpyAsync(mem32_.p, adp32_.p, (size_t)Nn * Dd * 4, hipMemcpyDeviceToDevice, stream_));
  }
  copy_segments(segs_d + gather.size(), (int)memseg.size(), stream_);
  gemm_act(mem16, Dd, cross_w_, nullptr, 0, Nn, L * 2 * Dd, Dd, crosstmp_.as<bf16_t>(), nullptr, stream_);
  stream_scatter_cross(crosstmp_.as<bf16_t>(), nslot_d, nidx_d, Nn, L, Dd, Mcap_, crossK_, crossV_, stream_);
  stream_slot_update(upd_d, (int)upd.size(), slots_d_, stream_);
}

// ------------------------------------------------------------------------------------------------
void StreamingEngine::decoder_reset(int n, const int* slots) {
  check_slots(n, slots);
  if (n == 0) return;
  MSH_HIP(hipSetDevice(device_));
  std::vector<int4> upd;
  for (int i = 0; i < n; ++i) {
    st(slots[i]).cache_len = 0;
    upd.push_back(make_int4(slots[i], -1, 0, 1));
  }
  stream_slot_update(stage(jobs_, upd), n, slots_d_, stream_);
}

// One pass of the decoder (lora/export.py:207-256) over M rows whose embeddings sit in H; row r belongs to
// stream row_slot[r] at position row_pos[r].
const int2* StreamingEngine::stage_runs(const std::vector<int>& rs, int* n_runs) {
  *n_runs = 0;
  static const bool off = [] {   // A/B switch: MSH_NO_CROSS_RUNS=1 keeps the one-row-per-workgroup kernel for every pass
    const char* e = dev_getenv("MSH_NO_CROSS_RUNS");
    return e != nullptr && e[0] == '1';
  }();
  runs_wide_ = false;
  if (off) return nullptr;
  // a stream with 8 or more consecutive rows in the pass (a verify pass): whole runs of up to 128 rows for the MFMA kernel
  // (MSH_STREAM_XWIDE=0: the four-row runs kernel for every pass)
  static const bool wide_off = [] {
    const char* e = dev_getenv("MSH_STREAM_XWIDE");
    return e != nullptr && e[0] == '0';
  }();
  size_t longest = 0;
  for (size_t r = 0; r < rs.size();) {
    size_t e = r + 1;
    while (e < rs.size() && rs[e] == rs[r]) ++e;
    longest = std::max(longest, e - r);
    r = e;
  }
  const bool wide = !wide_off && longest >= 8 && capture_probs_ == nullptr &&
                    stream_cross_attention_wide_supported(cfg_.decoder_dim, cfg_.nheads, Mcap_);
  if (!wide && !stream_cross_attention_runs_supported(cfg_.decoder_dim, cfg_.nheads, Mcap_)) return nullptr;
  const size_t cap = wide ? (size_t)kCrossWideRows : (size_t)kCrossRunRows;
  runs_wide_ = wide;
  std::vector<int2> runs;
  bool any_long = false;
  for (size_t r = 0; r < rs.size();) {
    size_t e = r + 1;
    while (e < rs.size() && rs[e] == rs[r] && e - r < cap) ++e;
    runs.push_back(make_int2((int)r, (int)(e - r)));
    any_long |= e - r > 1;
    r = e;
  }
  if (!any_long) return nullptr;
  *n_runs = (int)runs.size();
  return stage(runs_, runs);
}

void StreamingEngine::reserve_decoder_buffers(int rows) {
  const int Dd = cfg_.decoder_dim, Fd = cfg_.dec_ffn;
  stepH_.reserve((size_t)rows * Dd * 4);
  Y_.reserve((size_t)rows * Dd * 2);
  QKV_.reserve((size_t)rows * 3 * Dd * 2);
  AO_.reserve((size_t)rows * Dd * 2);
  Q_.reserve((size_t)rows * Dd * 2);
  Z_.reserve((size_t)rows * Fd * 2);
}

// fm: the pass runs on fragment-major operands (one row per stream, AR steps only): stepH_ is FM fp32 [M16][Dd], the
// attention outputs and the MLP activation are FM bf16, weights are the *_fm copies -- every wave-level operand load of
// the GEMMs is one contiguous 1 KiB run (kernels.h).  Results are bit-identical to the row-major pass.
void StreamingEngine::decoder_pass(int M, const int* row_slot, const int* row_pos, float* logits, const int2* runs_d,
                                   int n_runs, float* pval, int* pidx, bool fm) {
  const int Dd = cfg_.decoder_dim, L = cfg_.depth, Fd = cfg_.dec_ffn, V = cfg_.vocab_size;
  reserve_decoder_buffers(fm ? (M + 15) / 16 * 16 : M);
  if (fm && (runs_d != nullptr || capture_probs_ != nullptr || !fm_ok_)) throw std::logic_error("decoder_pass: FM is for AR steps");
  float* H = stepH_.as<float>();
  bf16_t* Y = Y_.as<bf16_t>();
  bf16_t* QKV = QKV_.as<bf16_t>();
  bf16_t* AO = AO_.as<bf16_t>();
  bf16_t* Q = Q_.as<bf16_t>();
  bf16_t* Z = Z_.as<bf16_t>();
  const RopeParams rp{rope_cos_, rope_sin_, rot_pairs_, cfg_.head_dim, Dd};
  const bool small = M <= 256;  // split-K decode kernel (16-row tiles) instead of 128-row MFMA tiles
  // developer probe (tools/gpu_stream_chain.sh): MSH_STREAM_AR_MASK = bit mask of the kernel groups an FM auto-regressive step
  // enqueues (bit 0 LN + QKV, 1 self-attention, 2 o-proj, 3 LN + cross-q, 4 cross-attention, 5 cross-o, 6 LN + fc1, 7 fc2,
  // 8 final LN + LM head); the replayed graph then times that chain alone.  Tokens are garbage under a partial mask.
  static const unsigned ar_mask = [] {
    const char* e = dev_getenv("MSH_STREAM_AR_MASK");
    return e != nullptr ? (unsigned)strtoul(e, nullptr, 0) : 0xffffffffu;
  }();
  auto on = [&](int bit) { return !fm || ((ar_mask >> bit) & 1u) != 0; };
  // profiler groups: "sdec_*" for the one-row-per-stream AR steps (weight-streaming: bytes = the weights), "sver_*" for
  // the wide verify pass (many rows per stream)
  using Sc = ScopeProfiler::Scope;
  const bool wide = runs_d != nullptr;
  const double md = (double)M * Dd, wdd = 2.0 * Dd * Dd;
  auto nm = [&](const char* ar, const char* ver) { return wide ? ver : ar; };
  for (int l = 0; l < L; ++l) {
    const DecW& W = dec_[l];
    // small passes: LayerNorm fused into the GEMM, q to its own buffer, k / v straight into the cache
    {
      Sc sc(&prof_, stream_, nm("sdec_qkv_self_attention", "sver_qkv_self_attention"), 2.0 * md * 3 * Dd, 3 * wdd + md * 12);
      if (fm) {
        if (on(0) && !stream_fm_qkv(H, W.wqkv_fm, M, Dd, Q, selfK_, selfV_, row_slot, row_pos, rp, l, L, Scap_, stream_))
          throw std::logic_error("decoder_pass: FM qkv width not compiled");
        if (on(1))
          stream_self_attention_cached(Q, row_slot, row_pos, M, Dd, cfg_.nheads, l, L, Scap_, selfK_, selfV_, AO, stream_, true,
                                       ar_keys_bound_);
      } else if (small && small_ln_gemm_stream_qkv(H, W.wqkv_f, M, Dd, Q, selfK_, selfV_, row_slot, row_pos, rp, l, L, Scap_, stream_)) {
        stream_self_attention_cached(Q, row_slot, row_pos, M, Dd, cfg_.nheads, l, L, Scap_, selfK_, selfV_, AO, stream_, false,
                                     wide ? 0 : ar_keys_bound_);
      } else {
        layernorm_bf16(H, W.ln1, M, Dd, Y, nullptr, stream_);
        gemm_qkv_rope_bf16(Y, Dd, W.wqkv, M, 3 * Dd, Dd, row_pos, rp, QKV, stream_);
        stream_self_attention(QKV, row_slot, row_pos, M, Dd, cfg_.nheads, l, L, Scap_, selfK_, selfV_, AO, stream_);
      }
    }
    {
      Sc sc(&prof_, stream_, nm("sdec_proj_gemms", "sver_proj_gemms"), 2.0 * md * Dd * 2, 2 * wdd + md * 16);
      if (fm) {
        if ((on(2) && !stream_fm_resid(AO, W.wo_fm, nullptr, M, Dd, Dd, H, stream_)) ||
            (on(3) && !stream_fm_ln_bf16(H, W.wq_c_fm, M, Dd, Dd, Q, stream_)))
          throw std::logic_error("decoder_pass: FM projection width not compiled");
      } else {
        if (!(small && small_gemm_resid_f32(AO, Dd, W.wo, nullptr, M, Dd, Dd, H, stream_)))
          gemm_resid_f32(AO, Dd, W.wo, nullptr, M, Dd, Dd, H, stream_);
        if (!(small && small_ln_gemm_bf16(H, W.wq_c_f, M, Dd, Dd, Q, stream_))) {
          layernorm_bf16(H, W.ln2, M, Dd, Y, nullptr, stream_);
          gemm_act(Y, Dd, W.wq_c, nullptr, 0, M, Dd, Dd, Q, nullptr, stream_);
        }
      }
    }
    if (capture_probs_ != nullptr)  // word timestamps: this pass's cross-attention probabilities (cross_attention())
      stream_cross_probs(Q, row_slot, slots_d_, M, Dd, cfg_.nheads, l, L, Mcap_, crossK_, capture_ecap_, capture_probs_, stream_);
    {
      Sc sc(&prof_, stream_, nm("sdec_cross_attention", "sver_cross_attention"), 0, pass_cross_bytes_ / L + md * 4);
      if (runs_d != nullptr && runs_wide_)
        stream_cross_attention_wide(Q, row_slot, runs_d, n_runs, slots_d_, Dd, cfg_.nheads, l, L, Mcap_, crossK_, crossV_, AO, stream_);
      else if (runs_d != nullptr)
        stream_cross_attention_runs(Q, row_slot, runs_d, n_runs, slots_d_, Dd, cfg_.nheads, l, L, Mcap_, crossK_, crossV_, AO, stream_);
      else if (on(4))
        stream_cross_attention(Q, row_slot, slots_d_, M, Dd, cfg_.nheads, l, L, Mcap_, crossK_, crossV_, AO, stream_, fm,
                               ar_row_mem_d_);
    }
    {
      Sc sc(&prof_, stream_, nm("sdec_crosso_mlp_gemms", "sver_crosso_mlp_gemms"), 2.0 * md * (Dd + 3.0 * Fd),
            wdd + 6.0 * Dd * Fd + md * 16);
      if (fm) {
        if ((on(5) && !stream_fm_resid(AO, W.wo_c_fm, nullptr, M, Dd, Dd, H, stream_)) ||
            (on(6) && !stream_fm_ln_swiglu(H, W.fc1_fm, W.b1, M, Fd, Dd, Z, stream_)) ||
            (on(7) && !stream_fm_resid(Z, W.fc2_fm, W.b2, M, Dd, Fd, H, stream_)))
          throw std::logic_error("decoder_pass: FM MLP width not compiled");
      } else {
        if (!(small && small_gemm_resid_f32(AO, Dd, W.wo_c, nullptr, M, Dd, Dd, H, stream_)))
          gemm_resid_f32(AO, Dd, W.wo_c, nullptr, M, Dd, Dd, H, stream_);
        if (!(small && small_ln_gemm_swiglu(H, W.fc1_f, W.b1, M, 2 * Fd, Dd, Z, stream_))) {
          layernorm_bf16(H, W.ln3, M, Dd, Y, nullptr, stream_);
          gemm_swiglu_bf16(Y, Dd, W.fc1, W.b1, M, 2 * Fd, Dd, Z, stream_);
        }
        if (!(small && small_gemm_resid_f32(Z, Fd, W.fc2, W.b2, M, Dd, Fd, H, stream_)))
          gemm_resid_f32(Z, Fd, W.fc2, W.b2, M, Dd, Fd, H, stream_);
      }
    }
  }
  // (an LN-fused head would redo the LayerNorm in each of its V / 64 column tiles: measured 46 vs 28 + 5 us)
  if (!on(8)) return;
  Sc sc(&prof_, stream_, nm("sdec_lm_head", "sver_lm_head"), 2.0 * md * V, 2.0 * V * Dd + (pval != nullptr ? 0.0 : 4.0 * M * V));
  if (fm) dec_final_layernorm(H, dec_ln_, M, Dd, Y, stream_);   // FM in, row-major out (the LM head's A operand)
  else layernorm_bf16(H, dec_ln_, M, Dd, Y, nullptr, stream_);
  if (pval != nullptr) {   // nobody reads the logits of this pass: only each column tile's maximum leaves the GEMM
    gemm_argmax_partials(Y, Dd, head_w_, M, V, Dd, pval, pidx, stream_);
    return;
  }
  if (!(small && small_gemm_logits_f32(Y, Dd, head_w_, M, V, Dd, logits, stream_)))
    gemm_logits_f32(Y, Dd, head_w_, M, V, Dd, logits, stream_);
}

long StreamingEngine::decode_stat(int what) {
  switch (what) {
    case 0: return stat_ar_passes_;
    case 1: return stat_verify_passes_;
    case 2: return (long)stat_ar_us_;
    case 3: return (long)stat_verify_us_;
    case 4:
      stat_ar_passes_ = stat_verify_passes_ = 0;
      stat_ar_us_ = stat_verify_us_ = 0.0;
      return 0;
    default: throw std::invalid_argument("decode_stat: unknown statistic");
  }
}

void StreamingEngine::decode_tokens(int n, const int* slots, const int32_t* const* tokens, const int* lens,
                                    float* logits_out) {
  if (!loaded_) throw std::runtime_error("weights not loaded");
  check_slots(n, slots);
  MSH_HIP(hipSetDevice(device_));
  const int Dd = cfg_.decoder_dim, V = cfg_.vocab_size;
  std::vector<int> rs, rpos, tok;
  std::vector<DecJob> jobs;
  for (int i = 0; i < n; ++i) {
    SlotHost& h = st(slots[i]);
    if (lens[i] <= 0 || tokens[i] == nullptr) throw std::invalid_argument("empty token list");
    if (h.mem_len == 0) throw std::invalid_argument("memory is empty");  // streaming-model.cpp:1151-1154
    if (h.cache_len + lens[i] > Scap_) throw std::invalid_argument("self-attention cache capacity exceeded");
    jobs.push_back({slots[i], (int)rs.size(), lens[i], 0, 0, 0, h.cache_len, 0});
    for (int t = 0; t < lens[i]; ++t) {
      if (tokens[i][t] < 0 || tokens[i][t] >= V) throw std::invalid_argument("token id out of range");
      rs.push_back(slots[i]);
      rpos.push_back(h.cache_len + t);
      tok.push_back(tokens[i][t]);
    }
    h.cache_len += lens[i];
  }
  const int M = (int)rs.size();
  if (M == 0) return;
  stepH_.reserve((size_t)M * Dd * 4);
  logits_.reserve((size_t)M * V * 4);
  const int* rs_d = stage(rowslot_, rs);
  const int* rp_d = stage(rowpos_, rpos);
  const int* tok_d = stage(tokens_, tok);
  const DecJob* jobs_d = stage(decjobs_, jobs);
  int n_runs = 0;
  const int2* runs_d = stage_runs(rs, &n_runs);
  stream_embed(tok_d, M, embed_f32_, Dd, stepH_.as<float>(), stream_);
  decoder_pass(M, rs_d, rp_d, logits_.as<float>(), runs_d, n_runs);
  stream_bump_cache(jobs_d, (int)jobs.size(), slots_d_, stream_);
  if (logits_out != nullptr)
    MSH_HIP(hipMemcpyAsync(logits_out, logits_.p, (size_t)M * V * sizeof(float), hipMemcpyDeviceToHost, stream_));
  MSH_HIP(hipStreamSynchronize(stream_));
}

// Cross-attention of a token sequence fed from an empty self cache: [depth * heads][n][memory_len] fp32, the layout
// align_words takes.  Replaces the `cross_attentions.{l}` outputs the reference collects call by call from its
// decoder_kv_with_attention graph (core/moonshine-streaming-model.cpp:946-1066, core/transcriber.cpp:1028-1068).
void StreamingEngine::cross_attention(int slot, const int32_t* tokens, int n, float* out, size_t cap, int dims[3]) {
  if (!loaded_) throw std::runtime_error("weights not loaded");
  check_slots(1, &slot);
  MSH_HIP(hipSetDevice(device_));
  const int L = cfg_.depth, Hh = cfg_.nheads, E = st(slot).mem_len;
  dims[0] = L * Hh, dims[1] = n, dims[2] = E;
  if (out == nullptr) return;
  if (n <= 0 || E <= 0) throw std::invalid_argument("cross_attention: no tokens or empty memory");
  if (cap < (size_t)L * Hh * n * E) throw std::invalid_argument("cross_attention: output buffer too small");
  const int ecap = (E + 3) & ~3;
  probs_.reserve((size_t)n * L * Hh * ecap * sizeof(float));
  decoder_reset(1, &slot);
  capture_probs_ = probs_.as<float>();
  capture_ecap_ = ecap;
  try {
    decode_tokens(1, &slot, &tokens, &n, nullptr);
  } catch (...) {
    capture_probs_ = nullptr;
    throw;
  }
  capture_probs_ = nullptr;
  std::vector<float> tmp((size_t)n * L * Hh * ecap);
  MSH_HIP(hipMemcpyAsync(tmp.data(), probs_.p, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, stream_));
  MSH_HIP(hipStreamSynchronize(stream_));
  for (int r = 0; r < n; ++r)          // device rows are [position][layer][head][ecap]
    for (int lh = 0; lh < L * Hh; ++lh)
      memcpy(out + ((size_t)lh * n + r) * E, tmp.data() + ((size_t)r * L * Hh + lh) * ecap, (size_t)E * sizeof(float));
}

void StreamingEngine::set_bias(int n_nodes, const int32_t* child_off, const int32_t* child_tok, const int32_t* child_node,
                               const int32_t* depth, const float* depth_bonus, int n_depth_bonus) {
  MSH_HIP(hipSetDevice(device_));
  MSH_HIP(hipStreamSynchronize(stream_));
  // the captured AR step holds the trie's device pointers: a new trie (its buffers may move when they grow) needs a new graph
  if (ar_graph_ != nullptr) {
    (void)hipGraphExecDestroy(ar_graph_);
    ar_graph_ = nullptr;
  }
  ar_key_.clear();
  bias_ = BiasTrie{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  if (n_nodes <= 0) return;
  if (child_off == nullptr || depth == nullptr || depth_bonus == nullptr) throw std::invalid_argument("null trie array");
  const int n_children = child_off[n_nodes];
  if (n_children > 0 && (child_tok == nullptr || child_node == nullptr)) throw std::invalid_argument("null trie array");
  int max_depth = 0;
  for (int i = 0; i < n_nodes; ++i) {
    if (child_off[i] > child_off[i + 1]) throw std::invalid_argument("trie offsets must be non-decreasing");
    for (int c = child_off[i]; c < child_off[i + 1]; ++c) {
      if (c > child_off[i] && child_tok[c] <= child_tok[c - 1]) throw std::invalid_argument("trie children must be sorted by token");
      if (child_node[c] <= 0 || child_node[c] >= n_nodes) throw std::invalid_argument("trie child index out of range");
    }
    max_depth = std::max(max_depth, depth[i]);
  }
  if (max_depth + 2 > n_depth_bonus) throw std::invalid_argument("depth_bonus table too short");
  if (max_depth + 1 > 64) throw std::invalid_argument("key terms longer than 63 tokens are not supported");
  stage(bias_off_, std::vector<int32_t>(child_off, child_off + n_nodes + 1));
  stage(bias_tok_, std::vector<int32_t>(child_tok, child_tok + n_children));
  stage(bias_node_, std::vector<int32_t>(child_node, child_node + n_children));
  stage(bias_depth_, std::vector<int32_t>(depth, depth + n_nodes));
  stage(bias_bonus_, std::vector<float>(depth_bonus, depth_bonus + n_depth_bonus));
  bias_ = BiasTrie{bias_off_.as<int>(), bias_tok_.as<int>(), bias_node_.as<int>(), bias_depth_.as<int>(),
                   bias_bonus_.as<float>(), n_nodes};
}

// ------------------------------------------------------------------------------------------------
// decode_full (streaming-model.cpp:1192-1397) for n streams: one wide pass over [BOS, draft...] of every
// stream, device-side verify, then lock-step auto-regressive steps (one row per stream) until every stream
// hit EOS or its budget.  Token choice, EOS / budget tests and the rollback all run on the device; the host
// only polls a counter of active streams.
// ------------------------------------------------------------------------------------------------
void StreamingEngine::decode_full(int n, const int* slots, const int32_t* const* drafts, const int* draft_lens,
                                  const int* max_tokens, int32_t* tokens_out, int32_t* counts_out, int tokens_stride,
                                  int32_t* accepted_out) {
  const auto t_call = std::chrono::steady_clock::now();
  if (!loaded_) throw std::runtime_error("weights not loaded");
  check_slots(n, slots);
  if (n > 0 && (tokens_out == nullptr || counts_out == nullptr)) throw std::invalid_argument("null output");
  MSH_HIP(hipSetDevice(device_));
  const int Dd = cfg_.decoder_dim, V = cfg_.vocab_size;
  std::vector<int> rs, rpos, tok, draft_flat, job_slot, job_index;
  std::vector<int2> prefix;  // per wide-pass row: (offset into draft_flat, tokens before the row) for the biaser walk
  std::vector<DecJob> jobs;
  int max_budget = 0, keys_bound = 0;
  for (int i = 0; i < n; ++i) {
    SlotHost& h = st(slots[i]);
    counts_out[i] = 0;
    if (accepted_out) accepted_out[i] = 0;
    if (h.mem_len == 0) continue;  // streaming-model.cpp:1205-1210: empty result, success
    if (h.cache_len != 0)
      throw std::invalid_argument("decode_full needs an empty self-attention cache: call decoder_reset first "
                                  "(reference transcriber.cpp:1385)");
    const int dl = (drafts != nullptr && draft_lens != nullptr && drafts[i] != nullptr) ? draft_lens[i] : 0;
    const int budget = (max_tokens != nullptr && max_tokens[i] >= 0) ? max_tokens[i] : max_tokens_for(slots[i]);
    if (1 + std::max(dl, budget) + 1 > Scap_) throw std::invalid_argument("draft or budget exceeds the cache capacity");
    if (std::max(dl, budget) > tokens_stride) throw std::invalid_argument("tokens_stride too small");
    jobs.push_back({slots[i], (int)rs.size(), 1 + dl, (int)draft_flat.size(), dl, budget, 0, 0});
    job_slot.push_back(slots[i]);
    job_index.push_back(i);
    rs.push_back(slots[i]);
    rpos.push_back(0);
    tok.push_back(cfg_.bos_id);
    const int doff = (int)draft_flat.size();
    prefix.push_back(make_int2(doff, 0));
    for (int t = 0; t < dl; ++t) {
      if (drafts[i][t] < 0 || drafts[i][t] >= V) throw std::invalid_argument("draft token out of range");
      rs.push_back(slots[i]);
      rpos.push_back(1 + t);
      tok.push_back(drafts[i][t]);
      draft_flat.push_back(drafts[i][t]);
      prefix.push_back(make_int2(doff, t + 1));
    }
    max_budget = std::max(max_budget, budget);
    keys_bound = std::max(keys_bound, 2 + std::max(dl, budget));   // BOS + draft / budget + the row being decoded
  }
  const int J = (int)jobs.size(), M = (int)rs.size();
  if (J == 0) return;
  // AR steps (one row per stream) on fragment-major operands; every buffer is sized before the graph capture below
  const bool fm = fm_ok_ && J <= 256;
  reserve_decoder_buffers(std::max(M, (J + 15) / 16 * 16));
  logits_.reserve((size_t)M * V * 4);
  pred_.reserve((size_t)M * 4);
  steppos_.reserve((size_t)J * 4);
  const int* rs_d = stage(rowslot_, rs);
  const int* rp_d = stage(rowpos_, rpos);
  const int* tok_d = stage(tokens_, tok);
  const int* draft_d = stage(draft_, draft_flat);
  const DecJob* jobs_d = stage(decjobs_, jobs);
  const int* jslot_d = stage(newslot_, job_slot);
  std::vector<int> job_mem(job_slot.size());
  for (size_t j = 0; j < job_slot.size(); ++j) job_mem[j] = st(job_slot[j]).mem_len;
  const int* jmem_d = stage(jobmem_, job_mem);   // the AR rows' memory lengths (fixed for the whole decode_full)
  MSH_HIP(hipMemsetAsync(n_active_d_, 0, sizeof(int32_t), stream_));
  for (hipEvent_t& ev : stat_ev_)
    if (ev == nullptr) MSH_HIP(hipEventCreate(&ev));
  const double us_staged = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count();
  MSH_HIP(hipEventRecord(stat_ev_[0], stream_));
  stream_embed(tok_d, M, embed_f32_, Dd, stepH_.as<float>(), stream_);
  const int2* prefix_d = bias_.n_nodes > 0 ? stage(bias_prefix_, prefix) : nullptr;
  int n_runs = 0;
  const int2* runs_d = stage_runs(rs, &n_runs);
  // algorithmic K / V bytes of a pass over all layers: every stream's memory once per pass (the run kernel's design), two
  // tensors, bf16
  double mem_rows = 0.0;
  for (int sl : job_slot) mem_rows += st(sl).mem_len;
  pass_cross_bytes_ = mem_rows * Dd * 2.0 * 2.0 * cfg_.depth;
  decoder_pass(M, rs_d, rp_d, logits_.as<float>(), runs_d, n_runs);
  // the biaser's bonuses go in before every token choice, the verify pass included (streaming-model.cpp:1241-1246,
  // 1304-1315); row t of a stream is conditioned on draft[0..t)
  stream_bias_rows(bias_, prefix_d, draft_d, nullptr, nullptr, nullptr, 0, M, logits_.as<float>(), V, stream_);
  stream_argmax(logits_.as<float>(), M, V, pred_.as<int>(), stream_);
  stream_verify(jobs_d, J, pred_.as<int>(), draft_d, slots_d_, result_, Scap_, cfg_.eos_id, embed_f32_, Dd,
                stepH_.as<float>(), steppos_.as<int>(), n_active_d_, stream_, fm);
  // One autoregressive step = ~100 short dependent kernels whose every argument is a device pointer (positions, ids and
  // stop flags live on the device): captured once per (row count, buffer addresses, trie) into a hipGraph and replayed -- an
  // eager launch costs the host >= 3.5 us per kernel, a graph node ~1.6 us of GPU time.
  // Without a bias trie nobody needs the logits of an AR step: from 32 streams on, the LM head runs as the tiled GEMM with the
  // per-tile argmax epilogue (W read once instead of once per 16-row tile: 40 -> 14 us at 64 streams) and the advance kernel
  // picks the token from the tile maxima; no logits, no argmax launch.  Same first-max rule.
  static const bool no_fused_head = [] {
    const char* e = dev_getenv("MSH_NO_FUSED_ARGMAX");
    return e != nullptr && e[0] == '1';
  }();
  const bool fused_head = !no_fused_head && bias_.n_nodes == 0 && J >= 32 && (Dd & 31) == 0;
  const int ntn = gemm_argmax_tiles(V);
  if (fused_head) {
    pval_.reserve((size_t)J * ntn * sizeof(float));
    pidx_.reserve((size_t)J * ntn * sizeof(int));
  }
  ar_keys_bound_ = keys_bound;   // lets the AR steps' self-attention take its one-round-trip form (<= 128 keys)
  ar_row_mem_d_ = jmem_d;        // and their cross-attention read the memory length beside the slot index
  struct ClearBound {
    int* p;
    const int** q;
    ~ClearBound() {
      *p = 0;
      *q = nullptr;
    }
  } clear_bound{&ar_keys_bound_, &ar_row_mem_d_};
  auto ar_step = [&] {
    if (fused_head) {
      decoder_pass(J, jslot_d, steppos_.as<int>(), nullptr, nullptr, 0, pval_.as<float>(), pidx_.as<int>(), fm);
      stream_advance_partials(jobs_d, J, pval_.as<float>(), pidx_.as<int>(), ntn, slots_d_, result_, Scap_, cfg_.eos_id, embed_f32_,
                              Dd, stepH_.as<float>(), steppos_.as<int>(), n_active_d_, stream_, fm);
      return;
    }
    decoder_pass(J, jslot_d, steppos_.as<int>(), logits_.as<float>(), nullptr, 0, nullptr, nullptr, fm);
    stream_bias_rows(bias_, nullptr, nullptr, jobs_d, slots_d_, result_, Scap_, J, logits_.as<float>(), V, stream_);
    stream_argmax(logits_.as<float>(), J, V, pred_.as<int>(), stream_);
    stream_advance(jobs_d, J, pred_.as<int>(), slots_d_, result_, Scap_, cfg_.eos_id, embed_f32_, Dd, stepH_.as<float>(),
                   steppos_.as<int>(), n_active_d_, stream_, fm);
  };
  static const bool use_graph = [] {
    const char* e = dev_getenv("MSH_NO_GRAPH");
    return !(e != nullptr && e[0] == '1');
  }();
  const bool graph_now = use_graph && !prof_.on();   // event scopes cannot sit inside a replayed graph
  if (graph_now && max_budget > 0) {
    char key[512];
    snprintf(key, sizeof(key), "%d:%p:%p:%p:%p:%p:%p:%p:%p:%p:%p:%p:%p:%d:%p:%d:%p:%p:%p", J, (void*)jslot_d, (void*)jobs_d, steppos_.p,
             logits_.p, pred_.p, stepH_.p, Y_.p, QKV_.p, AO_.p, Q_.p, Z_.p, (void*)result_, bias_.n_nodes, (void*)bias_off_.p,
             (int)fused_head + 2 * (int)fm + 4 * (int)(keys_bound <= 128), pval_.p, pidx_.p, (void*)jmem_d);
    if (ar_graph_ == nullptr || ar_key_ != key) {
      if (ar_graph_ != nullptr) {
        MSH_HIP(hipGraphExecDestroy(ar_graph_));
        ar_graph_ = nullptr;
      }
      hipGraph_t gr = nullptr;
      std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
      MSH_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
      try {
        ar_step();
      } catch (...) {   // never leave the stream in capture mode: end it, drop the partial graph, report the real error
        (void)hipStreamEndCapture(stream_, &gr);
        if (gr != nullptr) (void)hipGraphDestroy(gr);
        ar_key_.clear();
        throw;
      }
      MSH_HIP(hipStreamEndCapture(stream_, &gr));
      const hipError_t inst = hipGraphInstantiate(&ar_graph_, gr, nullptr, nullptr, 0);
      (void)hipGraphDestroy(gr);
      if (inst != hipSuccess) {
        ar_graph_ = nullptr;
        ar_key_.clear();
        MSH_HIP(inst);
      }
      ar_key_ = key;
    }
  }
  // developer timing (MSH_STREAM_TIMING=1): host clock around the verify pass and the AR loop, with the syncs that takes
  static const bool timing = getenv("MSH_STREAM_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::micro>(now() - t).count(); };
  if (timing) {
    const double us_enq = us_since(t_call);
    MSH_HIP(hipStreamSynchronize(stream_));
    fprintf(stderr, "[moonshine] decode_full: %d streams, %d rows: staging %.0f us, verify pass enqueued at %.0f us, done at %.0f us\n", J, M,
            us_staged, us_enq, us_since(t_call));
  }
  const auto t_ar = now();
  MSH_HIP(hipEventRecord(stat_ev_[1], stream_));
  // read-back area (pinned): [0, 64) the active-stream counter, then every slot's record, then the token table
  const size_t rb_slots = 64, rb_tokens = rb_slots + (((size_t)max_slots_ * sizeof(SlotDev) + 63) & ~(size_t)63);
  unsigned char* rb = static_cast<unsigned char*>(rb_area(rb_tokens + (size_t)max_slots_ * Scap_ * sizeof(int32_t)));
  volatile int32_t* active_h = reinterpret_cast<volatile int32_t*>(rb);
  *active_h = 1;
  int steps_run = 0;
  for (int step = 0; step < max_budget; ++step) {
    if (step % 8 == 0) {
      MSH_HIP(hipMemcpyAsync(rb, n_active_d_, sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
      MSH_HIP(hipStreamSynchronize(stream_));
      if (*active_h <= 0) break;
    }
    if (graph_now) MSH_HIP(hipGraphLaunch(ar_graph_, stream_));
    else ar_step();
    ++steps_run;
  }
  MSH_HIP(hipEventRecord(stat_ev_[2], stream_));
  if (timing) {
    MSH_HIP(hipStreamSynchronize(stream_));
    const double us = us_since(t_ar);
    fprintf(stderr, "[moonshine] decode_full: %d AR steps in %.0f us = %.1f us per step (%s)\n", steps_run, us, us / std::max(steps_run, 1),
             graph_now ? "graph" : "eager");
  }
  // results: every slot's record and the whole token table in TWO copies into pinned memory (they were two pageable copies
  // per stream, each of which blocks the caller for tens of microseconds)
  const SlotDev* sd = reinterpret_cast<const SlotDev*>(rb + rb_slots);
  const int32_t* tok_h = reinterpret_cast<const int32_t*>(rb + rb_tokens);
  MSH_HIP(hipMemcpyAsync(rb + rb_slots, slots_d_, (size_t)max_slots_ * sizeof(SlotDev), hipMemcpyDeviceToHost, stream_));
  MSH_HIP(hipMemcpyAsync(rb + rb_tokens, result_, (size_t)max_slots_ * Scap_ * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  MSH_HIP(hipStreamSynchronize(stream_));
  {
    float v_ms = 0.f, a_ms = 0.f;
    if (hipEventElapsedTime(&v_ms, stat_ev_[0], stat_ev_[1]) == hipSuccess && hipEventElapsedTime(&a_ms, stat_ev_[1], stat_ev_[2]) == hipSuccess) {
      stat_verify_us_ += (double)v_ms * 1e3;
      stat_ar_us_ += (double)a_ms * 1e3;
      stat_verify_passes_ += 1;
      stat_ar_passes_ += steps_run;
    }
  }
  if (timing) fprintf(stderr, "[moonshine] decode_full: results on the host %.0f us after the call\n", us_since(t_call));
  for (int j = 0; j < J; ++j) {
    const int i = job_index[j];
    const SlotDev& r = sd[jobs[j].slot];
    counts_out[i] = r.count;
    if (accepted_out) accepted_out[i] = r.accepted;
    st(jobs[j].slot).cache_len = r.cache_len;
    if (r.count > 0)
      memcpy(tokens_out + (size_t)i * tokens_stride, tok_h + (size_t)jobs[j].slot * Scap_, (size_t)r.count * sizeof(int32_t));
  }
}

}  // namespace msh
